// TestOpticalFlow — the reference's single-pair flow harness (source/test/TestOpticalFlow.cpp:50-143, `--mode test`)
// on the GPU: reads --left_img / --right_img (relative to --test_dir, loaded "unchanged" like imread(path, -1), alpha
// added when missing), runs NovelViewGeneratorAsymmetricFlow::prepare — flowLtoR = flow(L, R, LEFT) and flowRtoL =
// flow(R, L, RIGHT), NovelView.cpp:270-299 — --repetitions times and logs "RUNTIME (sec) = ..." per repetition exactly
// where the reference does (TestOpticalFlow.cpp:78-81). This is the harness shape of BASELINE configs[1] (one
// 2048x2048 pair). The flow fields are written in the reference's .bin format (CvUtil.cpp:159-199) to
// <test_dir>/disparity/flow{LtoR,RtoL}_<flow_alg>.bin. The reference's debug visualisations and its non-lazy novel-view
// morph (generateNovelView) are test-only output and not produced.
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../include/s360.h"
#include "png_io.hpp"

namespace {
[[noreturn]] void die(const std::string& m) {  // VrCamException -> terminate handler -> abort (SystemUtil.cpp:42-61)
  std::fprintf(stderr, "Terminated with exception: %s\n", m.c_str());
  std::abort();
}
double now_sec() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
std::vector<uint8_t> load_bgra(const std::string& path, int* w, int* h) {
  pngio::Image im;
  try {
    im = pngio::read(path, true);
  } catch (const std::exception& e) {
    die(e.what());
  }
  *w = im.w;
  *h = im.h;
  if (im.c == 4) return std::vector<uint8_t>(im.px.begin(), im.px.end());
  std::vector<uint8_t> out((size_t)im.w * im.h * 4);  // cvtColor(BGR2BGRA): alpha = 255 (TestOpticalFlow.cpp:60-66)
  for (size_t i = 0, n = (size_t)im.w * im.h; i < n; ++i) {
    out[4 * i] = im.px[3 * i];
    out[4 * i + 1] = im.px[3 * i + 1];
    out[4 * i + 2] = im.px[3 * i + 2];
    out[4 * i + 3] = 255;
  }
  return out;
}
}  // namespace

int main(int argc, char** argv) {
  std::map<std::string, std::string> F = {{"mode", ""}, {"test_dir", ""}, {"left_img", ""}, {"right_img", ""},
                                          {"num_intermediate_views", "11"}, {"flow_alg", ""}, {"repetitions", "1"},
                                          {"save_asymmetric_novel_views", "false"}, {"show_interpolated_view", "false"},
                                          {"device", "0"}, {"log_dir", ""}, {"stderrthreshold", "0"}, {"v", "0"},
                                          {"logbuflevel", "0"}};
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.size() < 2 || a[0] != '-') die("unexpected argument: " + a);
    a = a.substr(a[1] == '-' ? 2 : 1);
    std::string key = a, val = "true";
    const size_t eq = a.find('=');
    if (eq != std::string::npos) {
      key = a.substr(0, eq);
      val = a.substr(eq + 1);
    } else if (key != "save_asymmetric_novel_views" && key != "show_interpolated_view") {
      if (i + 1 >= argc) die("flag '" + key + "' is missing its argument");
      val = argv[++i];
    }
    if (!F.count(key)) {
      std::fprintf(stderr, "ERROR: unknown command line flag '%s'\n", key.c_str());
      return 1;
    }
    F[key] = val;
  }
  auto require = [&](const char* k) {
    if (F[k].empty()) die(std::string("missing required command line argument: ") + k);
  };
  require("mode");
  if (F["mode"] != "test") die("unrecongized mode: " + F["mode"]);  // TestOpticalFlow.cpp:238 (this harness: test only)
  require("test_dir");
  require("left_img");
  require("right_img");
  require("flow_alg");

  int wl, hl, wr, hr;
  const std::vector<uint8_t> L = load_bgra(F["test_dir"] + "/" + F["left_img"], &wl, &hl);
  const std::vector<uint8_t> R = load_bgra(F["test_dir"] + "/" + F["right_img"], &wr, &hr);
  if (wl != wr || hl != hr) die("left and right images differ in size");

  // the flow operator needs no rig: a one-camera placeholder carries the context
  s360_camera cam;
  const double o[3] = {20, 0, 0}, fwd[3] = {1, 0, 0}, up[3] = {0, 0, 1}, right[3] = {0, -1, 0};
  const double res[2] = {(double)wl, (double)hl}, focal[2] = {1000, -1000};
  if (s360_camera_init(&cam, S360_CAM_RECTILINEAR, o, fwd, up, right, res, nullptr, nullptr, focal, nullptr, "side camera",
                       "cam0") < 0)
    die(s360_last_error(nullptr));
  s360_params P;
  std::memset(&P, 0, sizeof P);
  P.interpupilary_dist = 6.4;
  P.zero_parallax_dist = 10000;
  P.side_alpha_feather_size = 100;
  P.std_alpha_feather_size = 31;
  P.eqr_width = 256;
  P.eqr_height = 128;
  std::strncpy(P.side_flow_alg, "pixflow_low", sizeof(P.side_flow_alg) - 1);
  std::strncpy(P.polar_flow_alg, "pixflow_low", sizeof(P.polar_flow_alg) - 1);
  s360_ctx* ctx = nullptr;
  if (s360_create(&ctx, std::atoi(F["device"].c_str()), &cam, 1, &P) < 0) die(s360_last_error(nullptr));

  std::vector<float> flowLtoR((size_t)wl * hl * 2), flowRtoL((size_t)wl * hl * 2);
  const int reps = std::max(1, std::atoi(F["repetitions"].c_str()));
  const char* alg = F["flow_alg"].c_str();
  for (int rep = 0; rep < reps; ++rep) {
    std::fprintf(stderr, "---- repetition %d\n", rep);
    const double t0 = now_sec();
    // NovelViewGeneratorAsymmetricFlow::prepare (NovelView.cpp:282-297)
    if (s360_compute_optical_flow(ctx, alg, L.data(), R.data(), wl, hl, nullptr, nullptr, nullptr, S360_HINT_LEFT,
                                  flowLtoR.data()) < 0 ||
        s360_compute_optical_flow(ctx, alg, R.data(), L.data(), wl, hl, nullptr, nullptr, nullptr, S360_HINT_RIGHT,
                                  flowRtoL.data()) < 0)
      die(s360_last_error(ctx));
    std::fprintf(stderr, "RUNTIME (sec) = %g\n", now_sec() - t0);
  }
  const std::string dir = F["test_dir"] + "/disparity";
  mkdir(dir.c_str(), 0775);
  if (s360_save_flow_to_file((dir + "/flowLtoR_" + F["flow_alg"] + ".bin").c_str(), flowLtoR.data(), wl, hl) < 0 ||
      s360_save_flow_to_file((dir + "/flowRtoL_" + F["flow_alg"] + ".bin").c_str(), flowRtoL.data(), wl, hl) < 0)
    die(s360_last_error(nullptr));
  s360_destroy(ctx);
  return 0;
}
