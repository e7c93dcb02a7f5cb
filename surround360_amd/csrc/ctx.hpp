// ctx.hpp — the s360_ctx object behind the C ABI: one per device.
#pragma once
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "core.hpp"
#include "flow.hpp"
#include "rig.hpp"

namespace s360 {
struct FrameState;   // render.hpp
struct SlotScratch;  // render.hpp: buffers shared by the frame slots of a context
}

struct s360_ctx {
  // Every C-ABI entry point that takes a context holds this lock for its whole duration (api.hip `guard`): a context is
  // safe to call from any number of host threads — the 14 std::threads of TRSP:320-335 may share one — and the calls
  // execute one after the other in lock order. Recursive: batch entry points call the single-pair ones.
  mutable std::recursive_mutex mu;
  int device = 0;
  unsigned long long uid = 0;  // unique per process, never reused (what an ISP object binds to)
  hipStream_t st = nullptr;
  // what s360_stream() hands out: the stream created with the context, never swapped (frame_finish replaces `st` by `st2`
  // for its own duration when frame pipelining is on)
  hipStream_t st_user = nullptr;
  // Frame pipelining (s360_set_frame_pipelining): the pole stage / composite (frame_finish) runs on st2 so that it
  // overlaps the side stage (frame_render_pairs) of the NEXT frame of a video stream. The two stages share three
  // things, each guarded by an event: the strips, the pole source images, and the order side(k) -> finish(k).
  hipStream_t st2 = nullptr;
  bool pipeline = false;
  bool two_outputs = false;  // s360_set_output_double_buffer: finished frames alternate between two output buffers without pipelining
  hipEvent_t evSideDone = nullptr, evStripsFree = nullptr, evPoleSrcFree = nullptr;
  bool haveStripsFree = false, havePoleSrcFree = false;
  // Uploads (s360_frame_upload_*) never touch the render streams: the caller's buffer is copied through a small ring
  // of library-owned pinned chunks and sent on stUp, so a host thread can feed frame k+1 while frame k renders. Three
  // events order the two sides: evUploaded (render waits for the inputs), evSideSrcFree / evPoleSrcFree (the upload's
  // conversion kernels wait until the previous frame's projections have read the source images).
  hipStream_t stUp = nullptr;
  hipStream_t stDown = nullptr;  // s360_frame_download_equirect_of: device -> host copy of a finished frame
  hipEvent_t evDown = nullptr;   // ... and what its host thread sleeps on with the context lock released
  unsigned* downErr = nullptr;   // ... and that frame's sweep error words (pinned; copied on stDown with the pixels)
  bool png_encode = false;       // s360_set_png_encode: every finished frame is also encoded as a PNG on the device
  void* pngMetaHost = nullptr;   // pinned landing area of a frame's band table (s360_frame_download_png)
  size_t pngMetaHostBytes = 0;
  hipEvent_t evUpHost = nullptr; // s360_frame_uploads_complete
  static constexpr int kPinChunks = 4;
  static constexpr size_t kPinChunkBytes = (size_t)8 << 20;
  void* pin[kPinChunks] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t pinEv[kPinChunks] = {nullptr, nullptr, nullptr, nullptr};
  bool pinUsed[kPinChunks] = {false, false, false, false};
  int pinNext = 0;
  hipEvent_t evUploaded = nullptr, evSideSrcFree = nullptr;
  bool haveUploaded = false, haveSideSrcFree = false;
  // RCCL communicator of the sharded frame (comm.cpp); opaque here so that only comm.cpp needs rccl.h
  void* comm = nullptr;
  int comm_rank = 0, comm_size = 1;
  // what the two exchanges of the sharded frame have moved since the communicator was made (s360_comm_stats):
  // [0] s360_frame_exchange_strips, [1] s360_frame_gather_pole_layers
  struct CommStats { unsigned long long calls = 0, sent = 0, received = 0; } comm_stats[2];
  s360::Rig rig;
  s360_params P;
  s360_geometry g;
  s360::PoleRamp ramp;
  int top_idx = -1, bottom_idx = -1;
  s360::Profiler prof;
  std::unique_ptr<s360::FlowEngine> flow;       // operator-level calls + side flows
  std::unique_ptr<s360::FlowEngine> flow_pole;  // pole flows (different sizes: keeps both buffer sets resident)
  std::unique_ptr<s360::FlowEngine> flow_pr;    // pole-removal flow between the two bottom cameras
  int sweep_mode = -1;  // -1 default (lockstep), 2 latency, 3 throughput (s360_set_sweep_mode)
  std::string err;
  std::string frame_invalid;  // why the flags describe no renderable frame (s360_create); empty = fine
  // scratch for operator-level calls
  s360::DevBuf op_a, op_b, op_c, op_d, op_e, op_f;
  // Frame slots: slot 0 always exists; s360_set_frame_slots(n) adds more so that n independent frames (n streams of a
  // multi-stream job) are rendered by ONE launch sequence with the flows of all of them in the same batched kernels
  // (s360_frame_render_batch). Uploads / getters / downloads act on the selected slot.
  std::vector<std::shared_ptr<s360::FrameState>> slots;
  std::shared_ptr<s360::SlotScratch> slotScratch;
  int slot = 0;
  // warp maps of bicubicRemapToSpherical per rig camera: depend only on rig + sizes, shared by all slots
  s360::DevBuf sideMaps, topMap, botMap;
  // the same maps as the packed remap reads them (render_kernels.hip): per destination pixel one dword, per 64x16 tile
  // the box of source pixels; they also depend on the SOURCE size, so they are (re)built when that is first seen
  // (a small cache per map, keyed by the source size: slots or streams whose input sizes alternate do not re-pack, and a
  // new size never rebuilds buffers another stream of the context may still read; `ready` orders the users after the pack
  // kernel without a host wait)
  struct PackedMap { s360::DevBuf packed, tiles; int sw = -1, sh = -1; hipEvent_t ready = nullptr; };
  struct PackedCache { std::vector<std::unique_ptr<PackedMap>> e; };
  PackedCache sidePk, topPk, botPk;
  bool maps_ready = false;
  hipEvent_t evMaps = nullptr;  // behind the kernels that built the float maps (the pack kernels may run on another stream)
  void make_current() const { S360_HIP(hipSetDevice(device)); }
};
