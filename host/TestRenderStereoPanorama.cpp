// TestRenderStereoPanorama — drop-in host binary for the per-frame stereo-panorama render of surround360_render
// (reference: source/test/TestRenderStereoPanorama.cpp). Same flags (TRSP:44-70), same rig JSON, same input /
// output / state-file layout, so scripts/batch_process_video.py can call it unmodified; the work between
// "decoded images" and "stacked equirect" runs on one MI355X through the C ABI of libs360 (include/s360.h).
//
// Kept on the host, as in the reference: flag parsing, the rig loader (inside libs360: rig.cpp), directory
// scanning, PNG decode/encode (png_io.hpp instead of cv::imread/imwrite) and the flow-state files.
// Not supported here: --save_debug_images (debug PNGs only).
// Opt-in additions: --num_gpus G (one frame sharded over G GPUs, native RCCL strip gather) and --num_frames N (a
// stream of N consecutive frames in one process: device-resident temporal state, overlapped I/O).
#include <dirent.h>
#include <malloc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/s360.h"
#include "jpeg_io.hpp"
#include "footage.hpp"
#include "png_io.hpp"

namespace {

struct Flags {
  std::map<std::string, std::string> v;
  Flags() {
    // DEFINE_* of TRSP:44-70
    v = {{"rig_json_file", ""}, {"imgs_dir", ""}, {"frame_number", ""}, {"output_data_dir", ""},
         {"prev_frame_data_dir", "NONE"}, {"output_cubemap_path", ""}, {"output_equirect_path", ""},
         {"interpupilary_dist", "6.4"}, {"side_alpha_feather_size", "100"}, {"std_alpha_feather_size", "31"},
         {"save_debug_images", "false"}, {"sharpening", "0.0"}, {"enable_top", "false"}, {"enable_bottom", "false"},
         {"enable_pole_removal", "false"}, {"bottom_pole_masks_dir", ""}, {"side_flow_alg", "pixflow_low"},
         {"polar_flow_alg", "pixflow_low"}, {"poleremoval_flow_alg", "pixflow_low"}, {"zero_parallax_dist", "10000"},
         {"eqr_width", "256"}, {"eqr_height", "128"}, {"final_eqr_width", "3480"}, {"final_eqr_height", "960"},
         {"cubemap_width", "1536"}, {"cubemap_height", "1536"}, {"cubemap_format", "video"},
         // glog flags the caller passes (batch_process_video.py:31-34); accepted, only --v is used
         {"log_dir", ""}, {"stderrthreshold", "0"}, {"v", "0"}, {"logbuflevel", "0"}, {"logtostderr", "false"},
         {"alsologtostderr", "false"},
         // additions of this implementation (opt-in)
         {"device", "0"}, {"write_state", "true"},
         // --device_png (default on): the output equirect's PNG is filtered and deflated on the GPU behind the frame's last kernel
         // (s360_set_png_encode / s360_frame_download_png) and this program only writes the bytes; false: the frame's pixels come
         // back and host threads encode them (png_io.hpp). Same pixels in the file either way.
         {"device_png", "true"},
         // --num_gpus G: the 14 side pairs of the frame sharded over G GPUs, one RCCL strip gather
         {"num_gpus", "1"},
         // --num_frames N: frames frame_number .. +N-1 as ONE stream in this process (temporal state stays on the
         // device, decode/upload/render/download/encode overlapped); output_equirect_path must contain %s or {frame}
         // (the state files are written after the LAST frame only: that is what a later run resumes from)
         {"num_frames", "1"},
         // --num_streams S: the --num_frames frames as S independent streams, stream s = the s-th contiguous segment of the
         // frame range on GPU device + s (modulo the GPUs there are) — exactly what S invocations with that segment's
         // --frame_number / --num_frames on S GPUs produce (each segment's first frame has no previous frame, except the
         // first one's --prev_frame_data_dir), in one process. One stream cannot use more than one GPU: its pole flows
         // are one serial chain per frame and every frame needs its predecessor's flows
         {"num_streams", "1"},
         // ... on at most this many GPUs (0 = all there are). Streams that share a GPU are rendered as the FRAME SLOTS of one
         // context: frame k of all of them in one launch sequence, every stream's temporal state resident in its slot
         {"stream_gpus", "0"},
         // --bin_list a.bin,b.bin --isp_dir D: the cameras' frames come straight from the capture's .bin containers through the
         // ISP on the device ("the ISP feeding the GPU directly from .bin") instead of imgs_dir/<cam>/<frame>.png:
         // what `Unpacker --bin_list .. --isp_dir D --output_dir imgs_dir` followed by this program writes, with no file in
         // between. Cameras map as Unpacker names them: the n-th smallest serial number is the rig's camera "cam<n>"; D holds
         // <serial>.json; frame_number is the frame's index in the containers. --soft_isp as in host/Unpacker.
         {"bin_list", ""}, {"isp_dir", ""}, {"soft_isp", "false"}};
  }
  static bool is_bool(const std::string& k) {
    static const char* b[] = {"save_debug_images", "enable_top", "enable_bottom", "enable_pole_removal", "logtostderr",
                              "alsologtostderr", "write_state", "soft_isp", "device_png"};
    for (auto s : b)
      if (k == s) return true;
    return false;
  }
  // gflags syntax: --k=v, --k v, -k ..., --bool, --nobool. Unknown flags are an error, like gflags.
  void parse(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
      std::string a = argv[i];
      if (a.size() < 2 || a[0] != '-') fail("unexpected argument: " + a);
      a = a.substr(a[1] == '-' ? 2 : 1);
      std::string key = a, val;
      bool has = false;
      const size_t eq = a.find('=');
      if (eq != std::string::npos) { key = a.substr(0, eq); val = a.substr(eq + 1); has = true; }
      if (key == "help") { usage(); std::exit(0); }
      if (!v.count(key)) {
        if (key.rfind("no", 0) == 0 && v.count(key.substr(2)) && is_bool(key.substr(2)) && !has) { v[key.substr(2)] = "false"; continue; }
        fail("unknown command line flag '" + key + "'");
      }
      if (!has) {
        if (is_bool(key)) {
          // "--flag true/false" is accepted too when the next token is a literal boolean
          if (i + 1 < argc && (!std::strcmp(argv[i + 1], "true") || !std::strcmp(argv[i + 1], "false") ||
                               !std::strcmp(argv[i + 1], "1") || !std::strcmp(argv[i + 1], "0")))
            val = argv[++i];
          else val = "true";
        } else {
          if (i + 1 >= argc) fail("flag '" + key + "' is missing its argument");
          val = argv[++i];
        }
      }
      v[key] = val;
    }
  }
  std::string s(const std::string& k) const { return v.at(k); }
  double d(const std::string& k) const { return std::atof(v.at(k).c_str()); }
  int i(const std::string& k) const { return std::atoi(v.at(k).c_str()); }
  bool b(const std::string& k) const { const std::string& x = v.at(k); return x == "true" || x == "1" || x == "t" || x == "yes"; }
  [[noreturn]] static void fail(const std::string& m) {
    std::fprintf(stderr, "ERROR: %s\n", m.c_str());
    std::exit(1);
  }
  void usage() const {
    std::printf("TestRenderStereoPanorama (surround360_amd / MI355X): flags and defaults\n");
    for (auto& kv : v) std::printf("  --%s (default: \"%s\")\n", kv.first.c_str(), kv.second.c_str());
  }
};

double now_sec() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
[[noreturn]] void die(const std::string& m) {  // VrCamException -> terminate handler -> abort (SystemUtil.cpp:42-61)
  std::fprintf(stderr, "Terminated with exception: %s\n", m.c_str());
  std::abort();
}
void require_arg(const std::string& v, const char* name) {  // SystemUtil.h:45-49
  if (v.empty()) die(std::string("missing required command line argument: ") + name);
}
void ck(int rc, s360_ctx* ctx) {
  if (rc < 0) die(s360_last_error(ctx));
}
// getImageFileExtension (SystemUtil.h:96-105): extension of the first file in the camera's directory
std::string image_extension(const std::string& dir) {
  DIR* d = opendir(dir.c_str());
  if (!d) die("failed to read directory: " + dir);
  std::vector<std::string> names;
  while (dirent* e = readdir(d))
    if (e->d_name[0] != '.') names.push_back(e->d_name);
  closedir(d);
  if (names.empty()) die("no files in directory: " + dir);
  std::sort(names.begin(), names.end());
  const size_t dot = names[0].rfind('.');
  return dot == std::string::npos ? "" : names[0].substr(dot);
}
void mkdirs(const std::string& path) {
  std::string cur;
  for (size_t i = 0; i <= path.size(); ++i) {
    if (i == path.size() || path[i] == '/') {
      if (!cur.empty()) mkdir(cur.c_str(), 0775);
    }
    if (i < path.size()) cur += path[i];
  }
}
// imread: the decoder is chosen by the file's signature — PNG (every colour type / depth) or baseline JPEG (jpeg_io.hpp)
pngio::Image load_png(const std::string& path, bool keep_alpha) {
  try {
    return jpegio::read_any(path, keep_alpha);
  } catch (const std::exception& e) {
    die(e.what());
  }
}
void load_png_into(const std::string& path, bool keep_alpha, pngio::Image& im) {
  try {
    jpegio::read_any_into(path, keep_alpha, im);
  } catch (const std::exception& e) {
    die(e.what());
  }
}
// Deflate threads of one PNG encoder. A stream keeps up to three encoders running beside 51 decoder threads and the HIP
// runtime's own threads, which feed the GPU its ~1000 launches per frame: unbounded (one thread per 2 MB band, ~100 for an 8K
// equirect) the encoders crowd those out and the GPU waits for its host — on the GPU boxes the process has 16 CPUs of a
// 256-thread machine (profiles/r04_v2_end_to_end_*).
static bool g_fast_exit = false;
static int g_png_threads = 0;  // 0 = one per band, up to the hardware threads
// a PNG the device has encoded: nothing left to do but write the bytes (a short write leaves no file behind, like save_png)
void save_bytes(const std::string& path, const uint8_t* data, size_t n) {
  try {
    pngio::OutFile f(path);
    f.put(data, n);
    f.close();
  } catch (const std::exception& e) {
    die(e.what());
  }
}
void save_png(const std::string& path, const uint8_t* px, int w, int h, int c) {
  try {
    pngio::write(path, px, w, h, c, 1, g_png_threads);
  } catch (const std::exception& e) {
    die(e.what());
  }
}

// One decoded frame of the rig (rig.loadSideCameraImages: one thread per camera, RigDescription.cpp:80-108)
struct FrameInputs {
  std::string frame;
  std::vector<pngio::Image> side;
  pngio::Image top, bottom, bottom2, mask1, mask2;
  long binFrame = -1;  // --bin_list: the frame's index in the containers (nothing is decoded on the host)
};
// --bin_list: where a rig camera's frames are, and the ISP that develops them
struct BinCamera {
  const footage::Footage* file = nullptr;
  size_t cam = 0;
  uint32_t serial = 0;
  s360_isp* isp = nullptr;
};

struct Job {
  Flags F;
  std::vector<s360_camera> cams;
  std::vector<int> sideIdx;
  int P = 0, ncams = 0, ti = -1, bi = -1, b2 = -1;
  s360_params prm;
  s360_geometry g;
  std::vector<s360_ctx*> ctx;  // one per GPU; ctx[0] is the root (composite, output)
  std::vector<int> bounds;     // pairs [bounds[r], bounds[r+1]) are rendered by ctx[r]
  int owner[4] = {-1, -1, -1, -1};  // GPU of pole unit u (top_left, top_right, bottom_left, bottom_right); -1 = not enabled
  std::vector<int> unitMask, need;  // per GPU: its pole units; the eyes whose complete strips it assembles
  int extW = 0;
  std::vector<std::unique_ptr<footage::Footage>> bins;  // --bin_list
  std::map<std::string, BinCamera> binCam;              // rig camera id -> its frames
  bool from_bins() const { return !bins.empty(); }
  s360_ctx* owner_ctx(int u) const { return ctx[owner[u] < 0 ? 0 : owner[u]]; }
  int bottom_gpu() const { return owner[2] >= 0 ? owner[2] : 0; }
};

// Pole unit -> GPU (the reference runs the four units as four threads, TRSP:811-860). 4 or more GPUs: unit u
// on GPU u — with pole removal both bottom units on GPU 2, where the merged bottom image is then computed once. 2 or 3
// GPUs: the top units on GPU 0, the bottom units on GPU 1 (each needs one pole image, one FlowEngine batch of two flows).
void assign_pole_units(Job& J) {
  const int G = (int)J.ctx.size();
  J.unitMask.assign(G, 0);
  J.need.assign(G, 0);
  for (int u = 0; u < 4; ++u) {
    const bool on = u < 2 ? J.prm.enable_top : J.prm.enable_bottom;
    if (!on) { J.owner[u] = -1; continue; }
    int r = 0;
    if (G >= 4) r = (J.prm.enable_pole_removal && u == 3) ? 2 : u;
    else if (G >= 2) r = u < 2 ? 0 : 1;
    J.owner[u] = r;
    J.unitMask[r] |= 1 << u;
    J.need[r] |= 1 << (u & 1);  // poleToSideFlowThread reads the whole side panorama of its eye
  }
  J.need[0] = 3;  // the root composites both eyes
}

// `recycled`: the inputs of an earlier frame whose pixel buffers are reused (stream mode)
FrameInputs load_frame(const Job& J, const std::string& frame, FrameInputs recycled = FrameInputs()) {
  FrameInputs in = std::move(recycled);
  in.frame = frame;
  if (J.from_bins()) {  // the frames stay where they are (mmap); upload_frame sends their packed bytes
    char* end = nullptr;
    in.binFrame = std::strtol(frame.c_str(), &end, 10);
    if (end == frame.c_str() || *end || in.binFrame < 0) die("--bin_list: frame_number must be a frame index, got '" + frame + "'");
    return in;
  }
  in.side.resize(J.P);
  const std::string imgs = J.F.s("imgs_dir");
  std::vector<std::thread> th;
  for (int k = 0; k < J.P; ++k)
    th.emplace_back([&, k] {
      const std::string dir = imgs + "/" + J.cams[J.sideIdx[k]].id;
      load_png_into(dir + "/" + frame + image_extension(dir), false, in.side[k]);
    });
  if (J.prm.enable_top) th.emplace_back([&] { load_png_into(imgs + "/" + J.cams[J.ti].id + "/" + frame + ".png", false, in.top); });  // TRSP:652
  if (J.prm.enable_bottom) {
    th.emplace_back([&] { load_png_into(imgs + "/" + J.cams[J.bi].id + "/" + frame + ".png", false, in.bottom); });  // TRSP:602
    if (J.prm.enable_pole_removal) {  // PoleRemoval.cpp:48-66
      const std::string masks = J.F.s("bottom_pole_masks_dir");
      th.emplace_back([&] { in.bottom2 = load_png(imgs + "/" + J.cams[J.b2].id + "/" + frame + ".png", false); });
      th.emplace_back([&, masks] { in.mask1 = load_png(masks + "/" + J.cams[J.bi].id + ".png", false); });
      th.emplace_back([&, masks] { in.mask2 = load_png(masks + "/" + J.cams[J.b2].id + ".png", false); });
    }
  }
  for (auto& t : th) t.join();
  if (J.prm.enable_pole_removal) {
    const pngio::Image& im = in.bottom;
    if (in.bottom2.w != im.w || in.bottom2.h != im.h || in.mask1.w != im.w || in.mask1.h != im.h || in.mask2.w != im.w || in.mask2.h != im.h)
      die("missing or bad pole mask:" + J.F.s("bottom_pole_masks_dir") + "/" + J.cams[J.bi].id + ".png," + J.F.s("bottom_pole_masks_dir") + "/" + J.cams[J.b2].id + ".png");
  }
  return in;
}

// every GPU gets the side images its pairs touch and the pole images of its pole units (asynchronous: upload stream)
void upload_frame(const Job& J, const FrameInputs& in) {
  const int G = (int)J.ctx.size();
  if (J.from_bins()) {
    auto send = [&](const std::string& id, int camera) {
      const BinCamera& bc = J.binCam.at(id);
      const footage::Header& md = bc.file->md;
      const uint8_t* fr = nullptr;
      try {
        fr = bc.file->frame((size_t)in.binFrame, bc.cam);
      } catch (const std::exception& e) {
        die(e.what());
      }
      ck(s360_frame_upload_packed(J.ctx[0], bc.isp, camera, fr, (int)md.bitsPerPixel, (int)md.width, (int)md.height), J.ctx[0]);
    };
    for (int k = 0; k < J.P; ++k) send(J.cams[J.sideIdx[k]].id, k);
    if (J.prm.enable_top) send(J.cams[J.ti].id, S360_CAMERA_TOP);
    if (J.prm.enable_bottom) send(J.cams[J.bi].id, S360_CAMERA_BOTTOM);
    return;
  }
  for (int r = 0; r < G; ++r) {
    std::vector<char> need(J.P, 0);
    for (int p = J.bounds[r]; p < J.bounds[r + 1]; ++p) need[p] = need[(p + 1) % J.P] = 1;
    for (int k = 0; k < J.P; ++k)
      if (need[k]) ck(s360_frame_upload_side(J.ctx[r], k, in.side[k].px.data(), in.side[k].w, in.side[k].h, in.side[k].c), J.ctx[r]);
  }
  for (int r = 0; r < G; ++r) {  // the pole images go to the GPUs that run units of that pole
    s360_ctx* c = J.ctx[r];
    if (J.unitMask[r] & 3) ck(s360_frame_upload_top(c, in.top.px.data(), in.top.w, in.top.h), c);
    if (J.unitMask[r] & 12) {
      ck(s360_frame_upload_bottom(c, in.bottom.px.data(), in.bottom.w, in.bottom.h), c);
      if (J.prm.enable_pole_removal)
        ck(s360_frame_upload_pole_removal(c, in.bottom2.px.data(), in.mask1.px.data(), in.mask2.px.data(), in.bottom.w, in.bottom.h), c);
    }
  }
}

const char* const kEyeNames[4] = {"top_left", "top_right", "bottom_left", "bottom_right"};

// fn(0) .. fn(n - 1) on up to `workers` threads (file reads / PNG codecs of the per-frame state: 68 files per frame)
template <class Fn>
void parallel_items(int n, int workers, Fn fn) {
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  for (int w = 0; w < std::max(1, std::min(workers, n)); ++w)
    th.emplace_back([&] {
      for (int i = next++; i < n; i = next++) fn(i);
    });
  for (auto& t : th) t.join();
}
int state_workers() { return std::max(2, std::min(16, pngio::available_cpus())); }

// tasks pushed by one thread, run by a pool (the state files of a frame are coded and written while the next ones are fetched)
class TaskQueue {
 public:
  explicit TaskQueue(int workers) {
    for (int w = 0; w < std::max(1, workers); ++w)
      th_.emplace_back([this] {
        for (;;) {
          std::function<void()> t;
          {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return closed_ || !q_.empty(); });
            if (q_.empty()) return;
            t = std::move(q_.front());
            q_.pop_front();
          }
          t();
        }
      });
  }
  void push(std::function<void()> t) {
    { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(t)); }
    cv_.notify_one();
  }
  ~TaskQueue() {
    { std::lock_guard<std::mutex> lk(mu_); closed_ = true; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
 private:
  std::deque<std::function<void()>> q_;
  bool closed_ = false;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<std::thread> th_;
};

// fn(i) for i in [0, n) on a pool of threads in the background; wait(i) blocks until item i is done (the caller consumes the
// items in order while the later ones are still being produced)
class ItemPool {
 public:
  template <class Fn>
  ItemPool(int n, int workers, Fn fn) : done_(n) {
    for (auto& d : done_) d.store(false);
    for (int w = 0; w < std::max(1, std::min(workers, n)); ++w)
      th_.emplace_back([this, n, fn] {
        for (int i = next_++; i < n; i = next_++) {
          fn(i);
          { std::lock_guard<std::mutex> lk(mu_); done_[i].store(true); }
          cv_.notify_all();
        }
      });
  }
  void wait(int i) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return done_[i].load(); });
  }
  ~ItemPool() { for (auto& t : th_) t.join(); }
 private:
  std::vector<std::atomic<bool>> done_;
  std::atomic<int> next_{0};
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<std::thread> th_;
};

// previous frame's state from files (TRSP:215-235, 421-436), each pair to the GPU that renders it. The 68 files of an 8K frame
// (1.2 GB of flows, 0.9 GB of images behind their PNG coding) are read and decoded by a pool of threads — the reference reads
// them inside its per-pair / per-unit threads, TRSP:215-235 —, then handed to the library in order.
void load_prev_state(const Job& J, const std::string& prev) {
  const s360_geometry& g = J.g;
  const std::string outData = J.F.s("output_data_dir");
  const std::string flowPrevDir = outData + "/flow/" + prev, imgPrevDir = outData + "/debug/" + prev + "/flow_images/";
  const size_t on = (size_t)g.overlap_image_width * g.cam_image_height;
  struct PairState { std::vector<float> fl, fr; pngio::Image L, R; };
  struct UnitState { std::vector<float> pf; pngio::Image S, Fi; bool on = false; };
  std::vector<PairState> pairs(J.P);
  UnitState units[4];
  for (int u = 0; u < 4; ++u) units[u].on = !((u < 2 && !J.prm.enable_top) || (u >= 2 && !J.prm.enable_bottom));
  // the pole units' files first (the largest: 85 MB images, 170 MB flows), then the pairs in the order they are handed over
  ItemPool pool(J.P + 4, state_workers(), [&](int it) {
    const int item = it < 4 ? J.P + it : it - 4;
    if (item < J.P) {
      const int i = item;
      PairState& ps = pairs[i];
      ps.fl.resize(on * 2);
      ps.fr.resize(on * 2);
      int w = 0, h = 0;
      if (s360_read_flow_from_file((flowPrevDir + "/flowLtoR_" + std::to_string(i) + ".bin").c_str(), ps.fl.data(), &w, &h, ps.fl.size()) < 0 ||
          w != g.overlap_image_width || h != g.cam_image_height)
        die("bad previous flow file for pair " + std::to_string(i) + ": " + s360_last_error(nullptr));
      if (s360_read_flow_from_file((flowPrevDir + "/flowRtoL_" + std::to_string(i) + ".bin").c_str(), ps.fr.data(), &w, &h, ps.fr.size()) < 0 ||
          w != g.overlap_image_width || h != g.cam_image_height)
        die("bad previous flow file for pair " + std::to_string(i) + ": " + s360_last_error(nullptr));
      ps.L = load_png(imgPrevDir + "/overlap_" + std::to_string(i) + "_L.png", true);
      ps.R = load_png(imgPrevDir + "/overlap_" + std::to_string(i) + "_R.png", true);
      if (ps.L.c != 4 || ps.R.c != 4 || ps.L.w != g.overlap_image_width || ps.L.h != g.cam_image_height || ps.R.w != ps.L.w || ps.R.h != ps.L.h)
        die("previous overlap images have the wrong size/channels");
    } else {
      const int u = item - J.P;
      UnitState& us = units[u];
      if (!us.on) return;
      const int rows = u < 2 ? g.top_rows : g.bottom_rows;
      us.pf.resize((size_t)J.extW * rows * 2);
      int w = 0, h = 0;
      if (s360_read_flow_from_file((flowPrevDir + "/flow_" + kEyeNames[u] + ".bin").c_str(), us.pf.data(), &w, &h, us.pf.size()) < 0 || w != J.extW || h != rows)
        die(std::string("bad previous pole flow file: ") + kEyeNames[u]);
      us.S = load_png(imgPrevDir + "/extendedSideSpherical_" + kEyeNames[u] + ".png", true);
      us.Fi = load_png(imgPrevDir + "/extendedFisheyeSpherical_" + kEyeNames[u] + ".png", true);
      if (us.S.c != 4 || us.Fi.c != 4 || us.S.w != J.extW || us.S.h != rows || us.Fi.w != J.extW || us.Fi.h != rows)
        die("previous extended pole images have the wrong size/channels");
    }
  });
  for (size_t r = 0; r < J.ctx.size(); ++r) {
    ck(s360_frame_set_partition(J.ctx[r], J.bounds[r], J.bounds[r + 1]), J.ctx[r]);
    for (int i = J.bounds[r]; i < J.bounds[r + 1]; ++i) {
      pool.wait(4 + i);  // (handed to the device while the later pairs are still being read)
      PairState& ps = pairs[i];
      ck(s360_frame_set_prev_side(J.ctx[r], i, ps.fl.data(), ps.fr.data(), ps.L.px.data(), ps.R.px.data()), J.ctx[r]);
      ps = PairState();
    }
  }
  if (J.prm.enable_pole_removal) {  // PoleRemoval.cpp:95-110
    s360_ctx* root = J.ctx[J.bottom_gpu()];  // the GPU that merges the bottom cameras
    int w = 0, h = 0;
    if (s360_read_flow_from_file((flowPrevDir + "/flow_bottom_secondary.bin").c_str(), nullptr, &w, &h, 0) < 0)
      die(std::string("bad previous flow file: flow_bottom_secondary.bin: ") + s360_last_error(nullptr));
    std::vector<float> pf((size_t)w * h * 2);
    if (s360_read_flow_from_file((flowPrevDir + "/flow_bottom_secondary.bin").c_str(), pf.data(), &w, &h, pf.size()) < 0)
      die(std::string("bad previous flow file: flow_bottom_secondary.bin: ") + s360_last_error(nullptr));
    const pngio::Image b1 = load_png(imgPrevDir + "/bottomImage.png", true), b2 = load_png(imgPrevDir + "/bottomImage2.png", true);
    if (b1.c != 4 || b2.c != 4 || b1.w != w || b1.h != h || b2.w != w || b2.h != h)
      die("previous bottomImage / bottomImage2 have the wrong size/channels");
    ck(s360_frame_set_prev_pole_removal(root, pf.data(), b1.px.data(), b2.px.data(), w, h), root);
  }
  for (int u = 0; u < 4; ++u) {
    UnitState& us = units[u];
    pool.wait(u);
    if (!us.on) continue;
    ck(s360_frame_set_prev_pole(J.owner_ctx(u), u, us.pf.data(), us.S.px.data(), us.Fi.px.data()), J.owner_ctx(u));  // a unit's state lives where it runs
    us = UnitState();
  }
}

// Enqueue one frame. One GPU: the whole frame on its stream. G GPUs (TRSP:320-385): every GPU renders its
// block of pairs, ONE grouped RCCL exchange hands the strips to the GPUs that assemble an eye (the root: both; the owner
// of a pole unit: its eye), the pole units run where assign_pole_units put them, a second grouped exchange returns their
// warped layers to the root, the root composites. One host thread per GPU for the collective calls (single-process RCCL).
void render_frame(const Job& J, bool usePrev) {
  const int G = (int)J.ctx.size();
  if (G == 1) {
    ck(s360_frame_render(J.ctx[0], usePrev ? 1 : 0), J.ctx[0]);
    return;
  }
  std::vector<std::thread> th;
  for (int r = 0; r < G; ++r)
    th.emplace_back([&, r] {
      s360_ctx* c = J.ctx[r];
      ck(s360_frame_render_pairs(c, J.bounds[r], J.bounds[r + 1], usePrev ? 1 : 0), c);
      ck(s360_frame_exchange_strips(c, J.bounds.data(), J.need.data()), c);  // exchange 1: strips to whoever assembles that eye
      if (J.unitMask[r] || r == 0) ck(s360_frame_pole_units(c, J.unitMask[r], usePrev ? 1 : 0), c);
      ck(s360_frame_gather_pole_layers(c, J.owner, 0), c);                    // exchange 2: warped pole layers to the root
      if (r == 0) ck(s360_frame_composite(c, 15), c);
    });
  for (auto& t : th) t.join();
}

// state for the next frame: always written by the reference (TRSP:201-208, 247-255, 413-416, 451-452). Fetched from the device
// in order, PNG-coded / written by a pool of threads (36 images and 32 flow files per 8K frame: 2 GB).
void write_state(const Job& J, const std::string& frame) {
  const std::string outData = J.F.s("output_data_dir");
  const std::string flowDir = outData + "/flow/" + frame, flowImagesDir = outData + "/debug/" + frame + "/flow_images";
  mkdirs(flowDir);
  mkdirs(flowImagesDir);
  struct Item { std::string path; std::vector<uint8_t> img; std::vector<float> fl; int w = 0, h = 0; };
  TaskQueue coders(state_workers());  // (joined when this function returns: every file is complete then)
  auto hand_over = [&](std::shared_ptr<Item> it) {
    coders.push([it] {
      if (!it->img.empty()) {
        try {
          // (the pool is the parallelism; only the pole units' 85 MB images get deflate threads of their own)
          pngio::write(it->path, it->img.data(), it->w, it->h, 4, 1, it->img.size() > ((size_t)32 << 20) ? 4 : 1);
        } catch (const std::exception& e) {
          die(e.what());
        }
      } else {
        ck(s360_save_flow_to_file(it->path.c_str(), it->fl.data(), it->w, it->h), nullptr);
      }
    });
  };
  auto get_img = [&](s360_ctx* c, const char* what, int idx, const std::string& path) {
    int whc[3];
    ck(s360_frame_get_u8(c, what, idx, whc, nullptr), c);
    auto it = std::make_shared<Item>();
    it->path = path; it->w = whc[0]; it->h = whc[1];
    it->img.resize((size_t)whc[0] * whc[1] * 4);
    ck(s360_frame_get_u8(c, what, idx, whc, it->img.data()), c);
    hand_over(std::move(it));
  };
  auto get_flow = [&](s360_ctx* c, const char* what, int idx, const std::string& path) {
    int whc[3];
    ck(s360_frame_get_f32(c, what, idx, whc, nullptr), c);
    auto it = std::make_shared<Item>();
    it->path = path; it->w = whc[0]; it->h = whc[1];
    it->fl.resize((size_t)whc[0] * whc[1] * 2);
    ck(s360_frame_get_f32(c, what, idx, whc, it->fl.data()), c);
    hand_over(std::move(it));
  };
  // the pole units first: their images (85 MB each at 8K) take longest to code
  for (int u = 0; u < 4; ++u) {
    if ((u < 2 && !J.prm.enable_top) || (u >= 2 && !J.prm.enable_bottom)) continue;
    s360_ctx* root = J.owner_ctx(u);
    get_img(root, "extended_side", u, flowImagesDir + "/extendedSideSpherical_" + kEyeNames[u] + ".png");
    get_img(root, "extended_fisheye", u, flowImagesDir + "/extendedFisheyeSpherical_" + kEyeNames[u] + ".png");
    get_flow(root, "flow_pole", u, flowDir + "/flow_" + kEyeNames[u] + ".bin");
  }
  if (J.prm.enable_pole_removal) {  // PoleRemoval.cpp:118-126 (kSaveDataNextFrame, TRSP:581)
    s360_ctx* root = J.ctx[J.bottom_gpu()];
    get_img(root, "bottom_image", 0, flowImagesDir + "/bottomImage.png");
    get_img(root, "bottom_image2", 0, flowImagesDir + "/bottomImage2.png");
    get_flow(root, "flow_bottom_secondary", 0, flowDir + "/flow_bottom_secondary.bin");
  }
  for (size_t r = 0; r < J.ctx.size(); ++r) {
    s360_ctx* c = J.ctx[r];
    for (int i = J.bounds[r]; i < J.bounds[r + 1]; ++i) {
      get_img(c, "overlap_l", i, flowImagesDir + "/overlap_" + std::to_string(i) + "_L.png");
      get_img(c, "overlap_r", i, flowImagesDir + "/overlap_" + std::to_string(i) + "_R.png");
      get_flow(c, "flow_l_to_r", i, flowDir + "/flowLtoR_" + std::to_string(i) + ".bin");
      get_flow(c, "flow_r_to_l", i, flowDir + "/flowRtoL_" + std::to_string(i) + ".bin");
    }
  }
}

// "000123" + 1 -> "000124" (same width); frame names of the reference's datasets are zero-padded decimal numbers
std::string next_frame_name(const std::string& f) {
  char buf[64];
  std::snprintf(buf, sizeof buf, "%0*lld", (int)f.size(), std::atoll(f.c_str()) + 1);
  return buf;
}
// output path of one frame in stream mode: "%s" (or "{frame}") in the flag value is replaced by the frame name
std::string frame_path(const std::string& pattern, const std::string& frame) {
  std::string out = pattern;
  for (const char* key : {"%s", "{frame}"}) {
    const size_t at = out.find(key);
    if (at != std::string::npos) { out.replace(at, std::strlen(key), frame); return out; }
  }
  return out;
}

}  // namespace

// --bin_list: opens the containers, finds every camera's serial number (the second word of its frames), names the cameras
// like Unpacker (Unpacker.cpp:203-219: directories sorted by serial number become cam0, cam1, ...) and creates one ISP per
// rig camera from <isp_dir>/<serial>.json — the arithmetic Unpacker runs (CameraIspPipe at 16 bits; --soft_isp: CameraIsp).
static void open_bins(Job& J) {
  const Flags& F = J.F;
  if (J.ctx.size() != 1) die("--bin_list feeds one GPU (an ISP object feeds one context): not with --num_gpus");
  if (J.prm.enable_pole_removal) die("--bin_list is not available with --enable_pole_removal");
  std::istringstream list(F.s("bin_list"));
  std::string path;
  std::map<uint32_t, std::pair<const footage::Footage*, size_t>> bySerial;
  try {
    while (std::getline(list, path, ',')) {
      std::unique_ptr<footage::Footage> ff(new footage::Footage);
      ff->path = path;
      ff->open(false);
      if (ff->md.numberOfCameras == 0) continue;
      if (ff->md.bitsPerPixel != 8 && ff->md.bitsPerPixel != 12) throw std::runtime_error("unsupported bits per pixel in " + path);
      for (size_t cam = 0; cam < ff->md.numberOfCameras; ++cam) {
        uint32_t serial;
        std::memcpy(&serial, ff->frame(0, cam) + 4, 4);
        bySerial[serial] = {ff.get(), cam};
      }
      J.bins.push_back(std::move(ff));
    }
  } catch (const std::exception& e) {
    die(e.what());
  }
  if (J.bins.empty()) die("--bin_list: no camera in " + F.s("bin_list"));
  size_t ordinal = 0;
  for (const auto& kv : bySerial) {
    BinCamera bc;
    bc.file = kv.second.first;
    bc.cam = kv.second.second;
    bc.serial = kv.first;
    J.binCam["cam" + std::to_string(ordinal++)] = bc;
  }
  auto want = [&](int idx) {
    const std::string id = J.cams[idx].id;
    auto it = J.binCam.find(id);
    if (it == J.binCam.end()) die("--bin_list: the containers hold " + std::to_string(J.binCam.size()) + " cameras, none becomes '" + id + "'");
    BinCamera& bc = it->second;
    if (bc.isp) return;
    const std::string json_path = F.s("isp_dir") + "/" + std::to_string(bc.serial) + ".json";
    std::ifstream js(json_path);
    if (!js) die("--bin_list: no ISP configuration " + json_path);
    std::stringstream ss;
    ss << js.rdbuf();
    s360_isp_config cfg;
    s360_isp_config_defaults(&cfg);
    cfg.output_bpp = 16;                  // kOutputBpp (Unpacker.cpp:167)
    cfg.pipe = F.b("soft_isp") ? 0 : 1;   // CameraIspPipe, kFast = false (Unpacker.cpp:166-168)
    if (s360_isp_config_from_json(ss.str().c_str(), &cfg) < 0) die(s360_last_error(nullptr));
    if (s360_isp_create(&bc.isp, F.i("device"), &cfg) < 0) die(s360_last_error(nullptr));
  };
  for (int k = 0; k < J.P; ++k) want(J.sideIdx[k]);
  if (J.prm.enable_top) want(J.ti);
  if (J.prm.enable_bottom) want(J.bi);
}

// Flags -> rig, parameters, required arguments (TRSP:717-721): common to a job and to a batch of streams
static void init_job(Job& J, const Flags& flags) {
  J.F = flags;
  Flags& F = J.F;
  require_arg(F.s("rig_json_file"), "rig_json_file");  // TRSP:717-721
  if (F.s("bin_list").empty()) require_arg(F.s("imgs_dir"), "imgs_dir");
  else require_arg(F.s("isp_dir"), "isp_dir");
  require_arg(F.s("frame_number"), "frame_number");
  require_arg(F.s("output_data_dir"), "output_data_dir");
  require_arg(F.s("output_equirect_path"), "output_equirect_path");

  J.cams.resize(64);
  J.ncams = s360_rig_load_json(F.s("rig_json_file").c_str(), J.cams.data(), (int)J.cams.size());
  if (J.ncams < 0) die(s360_last_error(nullptr));
  J.cams.resize(J.ncams);
  for (int i = 0; i < J.ncams; ++i)
    if (J.cams[i].is_side) J.sideIdx.push_back(i);
  J.P = (int)J.sideIdx.size();

  s360_params& prm = J.prm;
  std::memset(&prm, 0, sizeof prm);
  prm.interpupilary_dist = F.d("interpupilary_dist");
  prm.zero_parallax_dist = F.d("zero_parallax_dist");
  prm.sharpening = F.d("sharpening");
  prm.side_alpha_feather_size = F.i("side_alpha_feather_size");
  prm.std_alpha_feather_size = F.i("std_alpha_feather_size");
  prm.enable_top = F.b("enable_top");
  prm.enable_bottom = F.b("enable_bottom");
  prm.eqr_width = F.i("eqr_width");
  prm.eqr_height = F.i("eqr_height");
  prm.final_eqr_width = F.i("final_eqr_width");
  prm.final_eqr_height = F.i("final_eqr_height");
  std::strncpy(prm.side_flow_alg, F.s("side_flow_alg").c_str(), sizeof(prm.side_flow_alg) - 1);
  std::strncpy(prm.polar_flow_alg, F.s("polar_flow_alg").c_str(), sizeof(prm.polar_flow_alg) - 1);
  prm.enable_pole_removal = F.b("enable_pole_removal") && F.b("enable_bottom");
  std::strncpy(prm.poleremoval_flow_alg, F.s("poleremoval_flow_alg").c_str(), sizeof(prm.poleremoval_flow_alg) - 1);
  if (prm.enable_pole_removal) require_arg(F.s("bottom_pole_masks_dir"), "bottom_pole_masks_dir");  // TRSP:571
  if (prm.enable_top && (J.ti = s360_rig_find_top(J.cams.data(), J.ncams)) < 0) die("no top camera in the rig");
  if (prm.enable_bottom && (J.bi = s360_rig_find_bottom(J.cams.data(), J.ncams)) < 0) die("no bottom camera in the rig");
  if (prm.enable_pole_removal) J.b2 = s360_rig_find_bottom2(J.cams.data(), J.ncams);
}

// One job = what one invocation of the reference's program does, or (--num_frames) one stream of consecutive frames.
static int run_job(const Flags& flags) {
  Job J;
  const double startTime = now_sec();
  init_job(J, flags);
  Flags& F = J.F;
  s360_params& prm = J.prm;
  const int verbose = F.i("v");
  const double rigTime = now_sec();
  // the first frame's 17 images are decoded (one thread per camera) while HIP starts up and the context is made: a process
  // per frame — how batch_process_video.py:29-62 runs the program — pays both for every frame
  std::string frame = F.s("frame_number");
  std::future<FrameInputs> firstDecode;
  if (F.s("bin_list").empty()) firstDecode = std::async(std::launch::async, [&J, frame] { return load_frame(J, frame); });

  // ---- GPUs: --num_gpus G uses devices device .. device+G-1 (never more GPUs than pairs)
  const int G = std::max(1, std::min(F.i("num_gpus"), J.P));
  const int numFrames = std::max(1, F.i("num_frames"));
  if (G > 1 && numFrames > 1) die("--num_gpus and --num_frames are separate modes (a stream keeps its temporal state on one GPU)");
  if (G > s360_device_count() - F.i("device")) die("--num_gpus: not that many HIP devices");
  J.ctx.resize(G, nullptr);
  for (int r = 0; r < G; ++r)
    if (s360_create(&J.ctx[r], F.i("device") + r, J.cams.data(), J.ncams, &prm) < 0) die(s360_last_error(nullptr));
  ck(s360_get_geometry(J.ctx[0], &J.g), J.ctx[0]);
  J.extW = int(float(prm.eqr_width) * 1.2f);
  J.bounds.assign(1, 0);
  for (int r = 0; r < G; ++r) J.bounds.push_back(J.bounds.back() + J.P / G + (r < J.P % G ? 1 : 0));  // 14 over 8 -> 2,2,2,2,2,2,1,1
  assign_pole_units(J);
  if (G > 1) {
    if (s360_comm_init_all(J.ctx.data(), G) < 0) die(s360_last_error(nullptr));
    for (int r = 0; r < G; ++r) ck(s360_frame_set_partition(J.ctx[r], J.bounds[r], J.bounds[r + 1]), J.ctx[r]);
  }
  if (numFrames > 1) {
    ck(s360_set_frame_pipelining(J.ctx[0], 1), J.ctx[0]);  // pole stage of frame k overlaps side stage of k+1
  }
  if (!F.s("bin_list").empty()) open_bins(J);
  const bool devPng = F.b("device_png");
  if (devPng) ck(s360_set_png_encode(J.ctx[0], 1), J.ctx[0]);  // (the root composites and holds the output)

  const s360_geometry& g = J.g;
  const std::string prev = F.s("prev_frame_data_dir");
  const bool cube = F.i("cubemap_width") > 0 && F.i("cubemap_height") > 0 && !F.s("output_cubemap_path").empty();
  if (numFrames > 1 && cube) die("--output_cubemap_path is not available with --num_frames > 1");

  // ---- frame 0: decode, upload, previous-frame state from files, render
  const double ctxTime = now_sec();
  FrameInputs in = firstDecode.valid() ? firstDecode.get() : load_frame(J, frame);
  const double decodeTime = now_sec();
  upload_frame(J, in);
  const double loadTime = now_sec();
  if (prev != "NONE") load_prev_state(J, prev);
  const double renderStart = now_sec();
  render_frame(J, prev != "NONE");

  // Up to two finished frames are PNG-encoded and written while the next one renders (one encoder per frame, parallel
  // deflate inside it): an 8192 x 8192 file takes longer to encode and write than the frame takes to render.
  const size_t outBytes = devPng ? s360_frame_png_bound(J.ctx[0]) : (size_t)g.out_width * g.out_height * 3;
  constexpr int kMaxEncoders = 4;
  const char* encEnv = std::getenv("S360_ENCODERS");  // (developer switch)
  const int kEncoders = std::max(1, std::min(kMaxEncoders, encEnv ? std::atoi(encEnv) : 3));
  // (page-locked in stream mode: the finished frame comes back in one DMA transfer instead of through the runtime's
  // staging buffers — 201 MB per 8K frame)
  pngio::Pixels outBuf[kMaxEncoders + 1];
  std::thread encoder[kMaxEncoders + 1];  // encoder[i] owns outBuf[i] while it runs
  for (int i = 0; i <= kEncoders; ++i) outBuf[i].resize(numFrames > 1 || i == 0 ? outBytes : 0);
  int cur = 0;  // the buffer the next download goes to
  double renderEnd = renderStart, stateEnd = renderStart;
  double tDecode = 0, tUpload = 0, tFetch = 0, tJoin = 0;  // where the host thread of a stream spends its time (--v 1)
  double tSteadyStart = 0;                                  // ... from its fourth frame on (the first ones build maps and buffers)
  int steadyFrames = 0;
  // A stream decodes ahead: the PNGs of up to three coming frames are read and decoded by threads of their own (17 per
  // frame) while this thread feeds and drains the GPU — decoding one 8K frame's inputs takes longer than rendering it.
  std::deque<std::future<FrameInputs>> decoding;
  std::string decodeCursor = frame;
  int decodesStarted = 0;
  std::vector<FrameInputs> spare;  // uploaded frames: their pixel buffers go to the next decodes
  std::vector<FrameInputs> leaving;  // enqueued uploads may still read these (page-locked buffers are sent in place)
  auto decode_ahead = [&] {
    while (decodesStarted < numFrames - 1 && decoding.size() < 3) {
      decodeCursor = next_frame_name(decodeCursor);
      auto recycled = std::make_shared<FrameInputs>();
      if (!spare.empty()) { *recycled = std::move(spare.back()); spare.pop_back(); }
      decoding.push_back(std::async(std::launch::async, [&J, name = decodeCursor, recycled] { return load_frame(J, name, std::move(*recycled)); }));
      ++decodesStarted;
    }
  };
  if (numFrames > 1) {
    // Page-locking memory costs ~0.4 ms per MB: the buffers of the frames that will be in flight (three decoding ahead, one
    // uploading, one whose uploads may still run) are made once, here, while the GPU works on frame 0 — not by the decoder
    // threads of the first frames one image at a time (they serialise on the runtime's allocation lock: measured, the host
    // thread of a 20-frame stream waited 71 ms per frame for its decoders).
    for (int i = 0; i < 4 && i < numFrames - 1; ++i) {
      FrameInputs fi;
      fi.side.resize(in.side.size());
      for (size_t k = 0; k < in.side.size(); ++k) fi.side[k].px.resize(in.side[k].px.size());
      fi.top.px.resize(in.top.px.size());
      fi.bottom.px.resize(in.bottom.px.size());
      fi.bottom2.px.resize(in.bottom2.px.size());
      spare.push_back(std::move(fi));
    }
  }
  decode_ahead();
  for (int k = 0; k < numFrames; ++k) {
    const bool last = k + 1 == numFrames;
    if (k == 3 && numFrames > 4) { tDecode = tUpload = tFetch = tJoin = 0; tSteadyStart = now_sec(); }
    if (k >= 3) ++steadyFrames;
    std::string nextName;
    if (!last) {  // feed frame k+1 behind frame k: the GPU never waits for the host
      nextName = next_frame_name(frame);
      const double t0 = now_sec();
      FrameInputs nin = decoding.front().get();
      decoding.pop_front();
      decode_ahead();
      const double t1 = now_sec();
      upload_frame(J, nin);     // upload stream: overlaps frame k
      render_frame(J, true);    // temporal state stays on the device
      leaving.push_back(std::move(nin));  // (sent in place: recycled once the uploads have run, below)
      tDecode += t1 - t0;
      tUpload += now_sec() - t1;
    }
    const double tf = now_sec();
    size_t pngBytes = 0;
    if (devPng) ck(s360_frame_download_png(J.ctx[0], last ? 0 : 1, outBuf[cur].data(), outBuf[cur].size(), &pngBytes), J.ctx[0]);
    else if (last) ck(s360_frame_download_equirect(J.ctx[0], outBuf[cur].data()), J.ctx[0]);
    else ck(s360_frame_download_equirect_of(J.ctx[0], 1, outBuf[cur].data()), J.ctx[0]);  // frame k, while k+1 renders
    if (!leaving.empty()) {  // frame k is complete, so frame k+1's uploads (enqueued before it rendered) are long done
      ck(s360_frame_uploads_complete(J.ctx[0]), J.ctx[0]);
      for (auto& fi : leaving) spare.push_back(std::move(fi));
      leaving.clear();
    }
    renderEnd = now_sec();
    tFetch += renderEnd - tf;
    // (%s / {frame} in the path stands for the frame name — needed by streams, harmless for a single frame: a stream
    // segment of one frame is named like the others)
    const std::string outPath = frame_path(F.s("output_equirect_path"), frame);
    const uint8_t* px = outBuf[cur].data();
    if (devPng) encoder[cur] = std::thread([px, outPath, pngBytes] { save_bytes(outPath, px, pngBytes); });
    else encoder[cur] = std::thread([px, outPath, &g] { save_png(outPath, px, g.out_width, g.out_height, 3); });  // TRSP:961
    // the reference writes the state of every frame; a stream only needs it to resume after its last frame. (Beside the
    // equirect's encoder, not in front of it.)
    if (F.b("write_state") && last) write_state(J, frame);
    stateEnd = now_sec();
    if (last && cube) {  // optional stereo cubemap (TRSP:917-935)
      int whc[3];
      ck(s360_frame_cubemap(J.ctx[0], F.i("cubemap_width"), F.i("cubemap_height"), F.s("cubemap_format").c_str(), whc, nullptr), J.ctx[0]);
      std::vector<uint8_t> cubeImg((size_t)whc[0] * whc[1] * 3);
      ck(s360_frame_cubemap(J.ctx[0], F.i("cubemap_width"), F.i("cubemap_height"), F.s("cubemap_format").c_str(), whc, cubeImg.data()), J.ctx[0]);
      save_png(F.s("output_cubemap_path"), cubeImg.data(), whc[0], whc[1], 3);
    }
    cur = numFrames > 1 ? (cur + 1) % (kEncoders + 1) : 0;
    const double tj = now_sec();
    if (encoder[cur].joinable()) encoder[cur].join();  // the oldest encoder: its buffer takes the next frame
    if (last)
      for (auto& e : encoder)
        if (e.joinable()) e.join();
    tJoin += now_sec() - tj;
    frame = nextName;
  }
  const double endTime = now_sec();
  if (verbose >= 1) {  // the reference's VLOG(1) runtime breakdown, TRSP:964-971
    std::fprintf(stderr, "--- Runtime breakdown (sec) ---\n");
    std::fprintf(stderr, "load + decode + upload:  %.3f  (flags + rig %.3f, HIP start-up + context %.3f beside the PNG decodes, waiting for them %.3f, upload %.3f)\n",
                 loadTime - startTime, rigTime - startTime, ctxTime - rigTime, decodeTime - ctxTime, loadTime - decodeTime);
    std::fprintf(stderr, "previous-frame state:    %.3f\n", renderStart - loadTime);
    if (numFrames == 1) {
      std::fprintf(stderr, "GPU render + download:   %.3f  (%d GPU%s)\n", renderEnd - renderStart, G, G > 1 ? "s, RCCL strip gather" : "");
      std::fprintf(stderr, "state files:             %.3f  (beside the equirect's PNG encoder)\n", stateEnd - renderEnd);
      std::fprintf(stderr, "equirect PNG encode:     %.3f  (what was left of it)\n", endTime - stateEnd);
    } else {
      std::fprintf(stderr, "stream of %d frames:      %.3f  (%.3f per frame: decode, upload, render, download, encode overlapped)\n",
                   numFrames, endTime - renderStart, (endTime - renderStart) / numFrames);
      const int nf = tSteadyStart > 0 ? steadyFrames : numFrames;
      const double per = tSteadyStart > 0 ? (endTime - tSteadyStart) / nf : (endTime - renderStart) / nf;
      std::fprintf(stderr, "host thread per frame:   decode %.3f  upload+enqueue %.3f  wait+fetch %.3f  wait for the encoder %.3f  other %.3f  of %.3f  (frames %d..%d; decode = waiting for the decode-ahead threads)\n",
                   tDecode / nf, tUpload / nf, tFetch / nf, tJoin / nf, per - (tDecode + tUpload + tFetch + tJoin) / nf, per,
                   numFrames - nf, numFrames - 1);
    }
    std::fprintf(stderr, "TOTAL:                   %.3f\n", endTime - startTime);
  }
  if (g_fast_exit) return 0;  // main leaves the process at once (one frame per process): the device memory goes with it
  for (auto& kv : J.binCam)
    if (kv.second.isp) s360_isp_destroy(kv.second.isp);
  for (s360_ctx* c : J.ctx) s360_destroy(c);
  return 0;
}

// --num_streams with more streams than GPUs: the streams that share a GPU are the FRAME SLOTS of one context there. Step k renders
// frame k of every stream that still has one as ONE launch sequence (s360_frame_render_slots): per-frame kernels slot by slot, the
// 28 side flows of every stream in one batch of the flow kernels, the 4 pole flows of every stream in another — each frame
// regularised toward its own stream's device-resident previous flows and images (TRSP:215-235, 421-436; PixFlow.h:101-118,
// 185-193). A single stream is bound by the latency of its pole flows' serial chain; S of them in one
// launch sequence share that chain's time. Every stream writes, file for file, what an invocation of its own with that segment's
// --frame_number / --num_frames writes: its equirects, and the state files behind its last frame.
// While step k renders, step k+1's images are decoded (one task per stream, one thread per camera inside it) and uploaded, and
// step k-1's equirects are PNG-encoded.
struct Segment { std::string first; int n = 0; std::string prev = "NONE"; };
static int run_stream_batch(const Flags& flags, const std::vector<Segment>& segs, int device) {
  Job J;
  const double startTime = now_sec();
  init_job(J, flags);
  Flags& F = J.F;
  const int verbose = F.i("v");
  if (F.i("cubemap_width") > 0 && F.i("cubemap_height") > 0 && !F.s("output_cubemap_path").empty())
    die("--output_cubemap_path is not available with --num_streams");
  const int S = (int)segs.size();
  J.ctx.assign(1, nullptr);
  if (s360_create(&J.ctx[0], device, J.cams.data(), J.ncams, &J.prm) < 0) die(s360_last_error(nullptr));
  s360_ctx* ctx = J.ctx[0];
  ck(s360_get_geometry(ctx, &J.g), ctx);
  J.extW = int(float(J.prm.eqr_width) * 1.2f);
  J.bounds = {0, J.P};
  assign_pole_units(J);
  ck(s360_set_frame_slots(ctx, S), ctx);
  ck(s360_set_output_double_buffer(ctx, 1), ctx);  // step k is fetched while step k+1 renders
  if (!F.s("bin_list").empty()) {  // every stream's frames straight from the capture's containers: no PNG is inflated
    F.v["device"] = std::to_string(device);  // (open_bins makes the ISP objects on the job's device)
    open_bins(J);
  }
  ck(s360_set_sweep_mode(ctx, "throughput"), ctx);  // many flows per launch: the kernel with the fewest instructions per pixel
  const s360_geometry& g = J.g;
  const bool devPng = F.b("device_png");
  if (devPng) ck(s360_set_png_encode(ctx, 1), ctx);
  const size_t outBytes = devPng ? s360_frame_png_bound(ctx) : (size_t)g.out_width * g.out_height * 3;
  std::vector<size_t> pngBytes(S, 0);

  int steps = 0;
  for (const Segment& sg : segs) steps = std::max(steps, sg.n);
  std::vector<std::string> name(S);  // the frame stream s renders in the current step
  for (int s = 0; s < S; ++s) name[s] = segs[s].first;
  std::vector<FrameInputs> cur(S), spare(S);
  std::vector<std::future<FrameInputs>> decoding(S);
  auto start_decode = [&](int k) {  // step k's frames
    for (int s = 0; s < S; ++s)
      if (k < segs[s].n) {
        auto recycled = std::make_shared<FrameInputs>(std::move(spare[s]));
        std::string nm = segs[s].first;
        for (int i = 0; i < k; ++i) nm = next_frame_name(nm);
        decoding[s] = std::async(std::launch::async, [&J, nm, recycled] { return load_frame(J, nm, std::move(*recycled)); });
      }
  };
  std::mutex selMu;
  auto upload_step = [&](int k, std::vector<FrameInputs>& into) {
    for (int s = 0; s < S; ++s)
      if (k < segs[s].n) {
        into[s] = decoding[s].get();
        std::lock_guard<std::mutex> sel(selMu);  // (slot selection is context state: the fetching thread selects too, for the state files)
        ck(s360_select_frame_slot(ctx, s), ctx);
        upload_frame(J, into[s]);
      }
  };
  auto render_step = [&](int k) {
    std::vector<int> resumed, rest;
    for (int s = 0; s < S; ++s)
      if (k < segs[s].n) (k == 0 && segs[s].prev != "NONE" ? resumed : rest).push_back(s);
    for (int s : resumed) {  // a stream whose first frame resumes from state files: that slot alone, with its state
      ck(s360_select_frame_slot(ctx, s), ctx);
      load_prev_state(J, segs[s].prev);
      ck(s360_frame_render_slots(ctx, &s, 1, 1), ctx);
    }
    if (!rest.empty()) ck(s360_frame_render_slots(ctx, rest.data(), (int)rest.size(), k > 0 ? 1 : 0), ctx);
  };
  std::vector<pngio::Pixels> outBuf(S);
  for (auto& b : outBuf) b.resize(outBytes);
  std::vector<std::thread> encoder(S);
  double tGpuWait = 0, tEncWait = 0, tDecWait = 0, tEnqueue = 0, tDrainWait = 0, tStep0 = 0;
  start_decode(0);
  std::vector<FrameInputs> next(S);
  upload_step(0, cur);
  render_step(0);
  start_decode(1);
  // Two host threads. THIS one feeds: step k+1 is uploaded and enqueued while step k renders — every slot has two output buffers
  // (s360_set_output_double_buffer), so the GPU goes from step k straight into step k+1. The OTHER one drains: it fetches step k's
  // frames (age 1: the slot's latest enqueued frame is k+1 by then) with the slot named, not selected, and hands them to the file
  // writers. (Measured on 8 streams from containers: fetch then enqueue on one thread 14 frames per second — the GPU idle for the
  // length of the fetch —, enqueue then fetch on one thread 31, the device renders 41.) Step k+1 is enqueued only when step k-1 has
  // been fetched: it composites into the buffers step k-1 left.
  std::mutex mu;
  std::condition_variable cv;
  int enqueued = 0, drained = -1;  // highest step enqueued / fetched
  std::thread drain([&] {
    for (int k = 0; k < steps; ++k) {
      const bool more = k + 1 < steps;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return enqueued >= std::min(k + 1, steps - 1); });
      }
      const double t1 = now_sec();
      for (auto& e : encoder)
        if (e.joinable()) e.join();  // step k-1's files are written: their buffers take step k's frames
      const double t2 = now_sec();
      tEncWait += t2 - t1;
      for (int s = 0; s < S; ++s)
        if (k < segs[s].n) {
          const int age = (more && k + 1 < segs[s].n) ? 1 : 0;  // the slot's latest enqueued frame is k+1 unless its stream has ended
          // (the first fetch of a step waits for the step)
          if (devPng) ck(s360_frame_download_png_slot(ctx, s, age, outBuf[s].data(), outBuf[s].size(), &pngBytes[s]), ctx);
          else ck(s360_frame_download_equirect_slot(ctx, s, age, outBuf[s].data()), ctx);
          if (F.b("write_state") && k + 1 == segs[s].n) {
            std::lock_guard<std::mutex> sel(selMu);
            ck(s360_select_frame_slot(ctx, s), ctx);
            write_state(J, name[s]);
          }
        }
      tGpuWait += now_sec() - t2;
      if (k == 0) tStep0 = now_sec();  // the first step's frames have arrived: the steady state is measured from here
      {
        std::lock_guard<std::mutex> lk(mu);
        drained = k;
      }
      cv.notify_all();
      for (int s = 0; s < S; ++s)
        if (k < segs[s].n) {
          const std::string outPath = frame_path(F.s("output_equirect_path"), name[s]);
          const uint8_t* px = outBuf[s].data();
          const size_t nb = pngBytes[s];
          if (devPng) encoder[s] = std::thread([px, outPath, nb] { save_bytes(outPath, px, nb); });
          else encoder[s] = std::thread([px, outPath, &g] { save_png(outPath, px, g.out_width, g.out_height, 3); });
          name[s] = next_frame_name(name[s]);
        }
    }
  });
  for (int k = 0; k < steps; ++k) {
    const bool more = k + 1 < steps;
    const double t0 = now_sec();
    if (more) upload_step(k + 1, next);  // (the uploads wait, on their own stream, for step k's projections)
    const double t1 = now_sec();
    tDecWait += t1 - t0;
    if (more) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return drained >= k - 1; });
      }
      const double t1a = now_sec();
      tDrainWait += t1a - t1;
      render_step(k + 1);
      {
        std::lock_guard<std::mutex> lk(mu);
        enqueued = k + 1;
      }
      cv.notify_all();
      tEnqueue += now_sec() - t1a;
      ck(s360_frame_uploads_complete(ctx), ctx);
    }
    for (int s = 0; s < S; ++s)
      if (k < segs[s].n) spare[s] = std::move(cur[s]);  // its uploads have run (s360_frame_uploads_complete of the previous round)
    std::swap(cur, next);
    if (k + 2 < steps) start_decode(k + 2);
  }
  drain.join();
  for (auto& e : encoder)
    if (e.joinable()) e.join();
  const double endTime = now_sec();
  if (verbose >= 1) {
    int frames = 0;
    for (const Segment& sg : segs) frames += sg.n;
    std::fprintf(stderr, "--- Runtime breakdown (sec) ---\n");
    std::fprintf(stderr, "%d streams as frame slots of one context, %d steps, %d frames: %.3f  (%.3f per frame)\n", S, steps, frames,
                 endTime - startTime, (endTime - startTime) / frames);
    std::fprintf(stderr, "host thread per step:    decode + upload %.3f  wait for the encoders %.3f  wait for the GPU + fetch %.3f  enqueue %.3f  wait for the fetching thread %.3f  (feeding thread: decode + upload, enqueue, wait for the fetching thread; fetching thread: the rest)\n",
                 tDecWait / steps, tEncWait / steps, tGpuWait / steps, tEnqueue / steps, tDrainWait / steps);
    int later = 0;  // frames of the steps behind the first one (which pays for maps, buffers and kernel loading)
    for (const Segment& sg : segs) later += std::max(0, sg.n - 1);
    if (later > 0 && endTime > tStep0)
      std::fprintf(stderr, "steady state:            %d frames of steps 1..%d in %.3f  (%.2f frames per second, files written)\n", later, steps - 1,
                   endTime - tStep0, later / (endTime - tStep0));
    std::fprintf(stderr, "TOTAL:                   %.3f\n", endTime - startTime);
  }
  for (auto& kv : J.binCam)
    if (kv.second.isp) s360_isp_destroy(kv.second.isp);
  s360_destroy(ctx);
  return 0;
}

int main(int argc, char** argv) {
  // glibc: keep freed blocks of up to 32 MB (decoded camera images, PNG scanline bands) in the heap instead of handing
  // every one back to the kernel — a stream's decoder and encoder threads would spend their time in page faults
  mallopt(M_MMAP_THRESHOLD, 32 << 20);
  mallopt(M_TRIM_THRESHOLD, 1 << 30);
  Flags F;
  F.parse(argc, argv);
  const int streams = std::max(1, F.i("num_streams")), frames = std::max(1, F.i("num_frames"));
  if (frames > 1) {
    const char* e = std::getenv("S360_PNG_THREADS");  // (developer switch)
    // three encoders and 51 decoder threads share the CPUs the process may use: half of them per encoder
    g_png_threads = e ? std::atoi(e) : std::max(2, pngio::available_cpus() / 2);
  }
  // (S360_HOST_PINNED=0: developer switch for timing the two ways against each other; the pixels do not depend on it)
  const char* pinEnv = std::getenv("S360_HOST_PINNED");
  if (frames > 1 && !(pinEnv && pinEnv[0] == '0')) {
    // a stream's decoders write into page-locked memory: the uploads are DMA transfers straight from the decoded images
    // (set once, before the first image exists; every pngio::Image of this process then lives in such memory)
    pngio::g_pixel_alloc = s360_host_alloc;
    pngio::g_pixel_free = s360_host_free;
  }
  if (streams == 1) {
    // One frame per process is how the reference's caller runs this program (batch_process_video.py:29-62). Every file is closed
    // and every device result fetched when run_job returns; freeing 12 GB of device memory buffer by buffer and unloading the
    // runtime only to exit costs ~0.2 s per frame, so run_job skips its destroy calls and the process leaves at once with
    // _exit — which runs no atexit handlers and no library destructors: S360_CLEAN_EXIT=1 tears down normally, and is what a
    // leak checker, rocprofv3 or any other tracer that flushes its output at exit needs.
    const char* ce = std::getenv("S360_CLEAN_EXIT");
    g_fast_exit = frames == 1 && !(ce && ce[0] == '1');
    const int rc = run_job(F);
    if (g_fast_exit) {
      std::fflush(nullptr);
      _exit(rc);
    }
    return rc;
  }
  if (F.i("num_gpus") > 1) die("--num_streams and --num_gpus are separate modes");
  if (streams > frames) die("--num_streams: more streams than frames");
  require_arg(F.s("frame_number"), "frame_number");
  const int devices = s360_device_count();
  if (devices < 1) die("no HIP device");
  // stream s = the s-th contiguous segment; streams go round the GPUs (--stream_gpus limits them); a GPU with one stream runs it
  // as a job (frame pipelining inside the stream), a GPU with several runs them as the frame slots of one context
  const int useDev = std::max(1, std::min(devices, F.i("stream_gpus") > 0 ? F.i("stream_gpus") : devices));
  std::vector<std::vector<Segment>> perDev(useDev);
  std::string first = F.s("frame_number");
  for (int s = 0; s < streams; ++s) {
    Segment sg;
    sg.first = first;
    sg.n = frames / streams + (s < frames % streams ? 1 : 0);
    sg.prev = s == 0 ? F.s("prev_frame_data_dir") : "NONE";
    perDev[s % useDev].push_back(sg);
    for (int k = 0; k < sg.n; ++k) first = next_frame_name(first);
  }
  const int perEncoder = std::max(1, (int)perDev[0].size());  // encoders running at once per GPU
  if (perEncoder > 1) g_png_threads = std::max(1, pngio::available_cpus() / std::min(perEncoder * useDev, 8));
  std::vector<std::thread> th;
  for (int d = 0; d < useDev; ++d) {
    const int dev = (F.i("device") + d) % devices;
    if (perDev[d].size() == 1) {
      Flags Fs = F;
      Fs.v["num_streams"] = "1";
      Fs.v["frame_number"] = perDev[d][0].first;
      Fs.v["num_frames"] = std::to_string(perDev[d][0].n);
      Fs.v["device"] = std::to_string(dev);
      Fs.v["prev_frame_data_dir"] = perDev[d][0].prev;
      th.emplace_back([Fs] { run_job(Fs); });  // (errors abort the process, like the reference's)
    } else if (!perDev[d].empty()) {
      th.emplace_back([&F, &perDev, d, dev] { run_stream_batch(F, perDev[d], dev); });
    }
  }
  for (auto& t : th) t.join();
  return 0;
}
