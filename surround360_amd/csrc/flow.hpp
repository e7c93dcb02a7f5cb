// flow.hpp — host orchestration of PixFlow::computeOpticalFlow on the GPU.
//
// FlowEngine computes B flows over N images of identical size in one batched launch
// sequence (PixFlow.h:81-183). Flow b matches image idx.i0[b] against idx.i1[b]; all
// per-image work (downscale, grey/alpha, pyramids, gradients) is done once per image.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "core.hpp"
#include "flow_kernels.hpp"

namespace s360 {

// makeOpticalFlowByName (OpticalFlowFactory.h:23-64). Throws Error(-4) for unknown names.
PixFlowConsts pixflow_consts_by_name(const std::string& name);
// OpenCV getGaussianKernel(n, sigma, CV_32F) folded to centre + symmetric taps.
BlurTaps gaussian_taps(int ksize, double sigma);

struct FlowLevels {
  std::vector<int> w, h;          // level sizes, finest first (PixFlow.h:477-491)
  std::vector<size_t> off;        // pixel offset of each level inside a per-image pyramid plane
  size_t total = 0;               // pixels per image over all levels
  void build(int dw, int dh, float pyrScale);
};

// Which flows a batch computes and where its data lives. Flow b matches image i0[b] (I0) against i1[b] (I1). Images,
// previous flows and outputs are given per item as device pointers, so a batch may span buffers of several frames
// (s360_frame_render_batch); `contiguous` helpers cover the common one-allocation case.
struct FlowBatch {
  std::vector<int> i0, i1;                    // B entries
  std::vector<const uchar4*> images;          // N entries, each [h][w] uchar4
  std::vector<const uchar4*> prev_images;     // N entries or empty (no previous frame)
  std::vector<const float2*> prev_flow;       // B entries or empty
  std::vector<float2*> out;                   // B entries, each [h][w] float2
  void add_images(const uchar4* base, int n, size_t stride) { for (int k = 0; k < n; ++k) images.push_back(base + stride * k); }
  void add_prev_images(const uchar4* base, int n, size_t stride) { for (int k = 0; k < n; ++k) prev_images.push_back(base + stride * k); }
  void add_flow(int a, int b, float2* o, const float2* prev = nullptr) {
    i0.push_back(a); i1.push_back(b); out.push_back(o);
    if (prev) prev_flow.push_back(prev);
  }
};

// The device buffers of a compute() call: all of them are dead when the call's last kernel has run (the flows go to the
// batch's own output pointers), so engines whose calls can never overlap — the side, pole and pole-removal engines of a
// context without frame pipelining: one stream, one after the other — share ONE set, each buffer as large as its largest user
// (grow-only): 1.25 GB per 8K frame slot that the side engine no longer holds beside the pole engine's 2.4 GB.
struct FlowBufs {
  DevBuf down, prevdown, gray, pyrI, G, flowA, flowB, prevFlowDown, prevPyr, motionPyr, I1eq, rec, handoff;
};

class FlowEngine {
 public:
  explicit FlowEngine(Profiler* prof) : prof_(prof), bufs_(std::make_shared<FlowBufs>()) {}
  // use another engine's buffer set from now on (the caller guarantees that the two never compute at the same time)
  void share_buffers(const std::shared_ptr<FlowBufs>& b) { bufs_ = b; }
  const std::shared_ptr<FlowBufs>& buffers() const { return bufs_; }
  void compute(hipStream_t st, const PixFlowConsts& pc, const FlowBatch& batch, int w, int h, int hint);
  // debugging taps for parity tests (valid after compute() + stream sync)
  const uchar4* dbg_down() const { return bufs_->down.as<uchar4>(); }
  const FlowLevels& levels() const { return lv_; }
  int dw() const { return dw_; }
  int dh() const { return dh_; }
  // optional per-level flow capture (coarsest first), host side, for tests
  std::vector<std::vector<float>>* capture_levels = nullptr;

 private:
  Profiler* prof_;
  FlowLevels lv_;
  int dw_ = 0, dh_ = 0;
  // device copies of a batch's index arrays and pointer tables, cached by content (a video stream alternates between
  // the two halves of its double-buffered temporal state, so a few slots make the steady state upload-free)
  struct TabSlot { std::vector<unsigned long long> key; DevBuf buf; };
  TabSlot tabs_[4];
  int tab_next_ = 0;
  const unsigned long long* batch_tables(hipStream_t st, const FlowBatch& b);
  std::shared_ptr<FlowBufs> bufs_;
  DevBuf err_;
  int sweep_mode_ = 2;      // 2: lockstep kernel (latency, default), 3: quad kernel (throughput)
  int sweep_fast_ = -1;     // verified fast division / sqrt in the sweeps; S360_SWEEP_DIV=ieee selects the IEEE expansions (same bits)

 public:
  // 2 = lockstep (lowest latency of one flow), 3 = quad (highest chip-wide rate with many flows in flight)
  void set_sweep_mode(int m) { sweep_mode_ = (m == 3) ? 3 : 2; }
  // non-zero if a banded sweep timed out waiting for its neighbour band (results invalid); resets the flag
  unsigned take_error(hipStream_t st);
  // the device word behind take_error (nullptr before the first compute): frame_finish snapshots it per output buffer
  const unsigned* error_word() const { return err_.as<unsigned>(); }

 private:
};

}  // namespace s360
