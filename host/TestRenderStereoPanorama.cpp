// TestRenderStereoPanorama — drop-in host binary for the per-frame stereo-panorama render of surround360_render
// (reference: source/test/TestRenderStereoPanorama.cpp). Same flags (TRSP:44-70), same rig JSON, same input /
// output / state-file layout, so scripts/batch_process_video.py can call it unmodified; the work between
// "decoded images" and "stacked equirect" runs on one MI355X through the C ABI of libs360 (include/s360.h).
//
// Kept on the host, as in the reference: flag parsing, the rig loader (inside libs360: rig.cpp), directory
// scanning, PNG decode/encode (png_io.hpp instead of cv::imread/imwrite) and the flow-state files.
// Not supported here: --save_debug_images (debug PNGs only).
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../include/s360.h"
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
         {"device", "0"}, {"write_state", "true"}};
  }
  static bool is_bool(const std::string& k) {
    static const char* b[] = {"save_debug_images", "enable_top", "enable_bottom", "enable_pole_removal", "logtostderr",
                              "alsologtostderr", "write_state"};
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
pngio::Image load_png(const std::string& path, bool keep_alpha) {
  try {
    return pngio::read(path, keep_alpha);
  } catch (const std::exception& e) {
    die(e.what());
  }
}
void save_png(const std::string& path, const uint8_t* px, int w, int h, int c) {
  try {
    pngio::write(path, px, w, h, c);
  } catch (const std::exception& e) {
    die(e.what());
  }
}

}  // namespace

int main(int argc, char** argv) {
  Flags F;
  F.parse(argc, argv);
  require_arg(F.s("rig_json_file"), "rig_json_file");  // TRSP:717-721
  require_arg(F.s("imgs_dir"), "imgs_dir");
  require_arg(F.s("frame_number"), "frame_number");
  require_arg(F.s("output_data_dir"), "output_data_dir");
  require_arg(F.s("output_equirect_path"), "output_equirect_path");
  const int verbose = F.i("v");
  const double startTime = now_sec();

  std::vector<s360_camera> cams(64);
  const int ncams = s360_rig_load_json(F.s("rig_json_file").c_str(), cams.data(), (int)cams.size());
  if (ncams < 0) die(s360_last_error(nullptr));
  cams.resize(ncams);
  std::vector<int> sideIdx;
  for (int i = 0; i < ncams; ++i)
    if (cams[i].is_side) sideIdx.push_back(i);
  const int P = (int)sideIdx.size();

  s360_params prm;
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

  s360_ctx* ctx = nullptr;
  if (s360_create(&ctx, F.i("device"), cams.data(), ncams, &prm) < 0) die(s360_last_error(nullptr));
  s360_geometry g;
  ck(s360_get_geometry(ctx, &g), ctx);

  // ---- load + upload the camera images (rig.loadSideCameraImages: one thread per camera, RigDescription.cpp:80-108)
  const std::string frame = F.s("frame_number"), imgs = F.s("imgs_dir");
  std::vector<pngio::Image> sideImgs(P);
  {
    std::vector<std::thread> th;
    for (int k = 0; k < P; ++k)
      th.emplace_back([&, k] {
        const std::string dir = imgs + "/" + cams[sideIdx[k]].id;
        sideImgs[k] = load_png(dir + "/" + frame + image_extension(dir), false);
      });
    for (auto& t : th) t.join();
  }
  for (int k = 0; k < P; ++k) ck(s360_frame_upload_side(ctx, k, sideImgs[k].px.data(), sideImgs[k].w, sideImgs[k].h, sideImgs[k].c), ctx);
  if (prm.enable_top) {
    const int ti = s360_rig_find_top(cams.data(), ncams);
    if (ti < 0) die("no top camera in the rig");
    const pngio::Image im = load_png(imgs + "/" + cams[ti].id + "/" + frame + ".png", false);  // TRSP:652
    ck(s360_frame_upload_top(ctx, im.px.data(), im.w, im.h), ctx);
  }
  if (prm.enable_bottom) {
    const int bi = s360_rig_find_bottom(cams.data(), ncams);
    if (bi < 0) die("no bottom camera in the rig");
    const pngio::Image im = load_png(imgs + "/" + cams[bi].id + "/" + frame + ".png", false);  // TRSP:602
    ck(s360_frame_upload_bottom(ctx, im.px.data(), im.w, im.h), ctx);
    if (prm.enable_pole_removal) {  // PoleRemoval.cpp:48-66
      const int b2 = s360_rig_find_bottom2(cams.data(), ncams);
      const std::string masks = F.s("bottom_pole_masks_dir");
      const pngio::Image im2 = load_png(imgs + "/" + cams[b2].id + "/" + frame + ".png", false);
      const pngio::Image m1 = load_png(masks + "/" + cams[bi].id + ".png", false);
      const pngio::Image m2 = load_png(masks + "/" + cams[b2].id + ".png", false);
      if (im2.w != im.w || im2.h != im.h || m1.w != im.w || m1.h != im.h || m2.w != im.w || m2.h != im.h)
        die("missing or bad pole mask:" + masks + "/" + cams[bi].id + ".png," + masks + "/" + cams[b2].id + ".png");
      ck(s360_frame_upload_pole_removal(ctx, im2.px.data(), m1.px.data(), m2.px.data(), im.w, im.h), ctx);
    }
  }
  const double loadTime = now_sec();

  // ---- previous frame's state (TRSP:215-235, 421-436)
  const std::string outData = F.s("output_data_dir"), prev = F.s("prev_frame_data_dir");
  const bool usePrev = prev != "NONE";
  static const char* eyeNames[4] = {"top_left", "top_right", "bottom_left", "bottom_right"};
  const int extW = int(float(prm.eqr_width) * 1.2f);
  if (usePrev) {
    const std::string flowPrevDir = outData + "/flow/" + prev, imgPrevDir = outData + "/debug/" + prev + "/flow_images/";
    const size_t on = (size_t)g.overlap_image_width * g.cam_image_height;
    std::vector<float> fl(on * 2), fr(on * 2);
    for (int i = 0; i < P; ++i) {
      int w = 0, h = 0;
      if (s360_read_flow_from_file((flowPrevDir + "/flowLtoR_" + std::to_string(i) + ".bin").c_str(), fl.data(), &w, &h, fl.size()) < 0 ||
          w != g.overlap_image_width || h != g.cam_image_height)
        die("bad previous flow file for pair " + std::to_string(i) + ": " + s360_last_error(nullptr));
      if (s360_read_flow_from_file((flowPrevDir + "/flowRtoL_" + std::to_string(i) + ".bin").c_str(), fr.data(), &w, &h, fr.size()) < 0 ||
          w != g.overlap_image_width || h != g.cam_image_height)
        die("bad previous flow file for pair " + std::to_string(i) + ": " + s360_last_error(nullptr));
      const pngio::Image L = load_png(imgPrevDir + "/overlap_" + std::to_string(i) + "_L.png", true);
      const pngio::Image R = load_png(imgPrevDir + "/overlap_" + std::to_string(i) + "_R.png", true);
      if (L.c != 4 || R.c != 4 || L.w != g.overlap_image_width || L.h != g.cam_image_height || R.w != L.w || R.h != L.h)
        die("previous overlap images have the wrong size/channels");
      ck(s360_frame_set_prev_side(ctx, i, fl.data(), fr.data(), L.px.data(), R.px.data()), ctx);
    }
    if (prm.enable_pole_removal) {  // PoleRemoval.cpp:95-110
      int w = 0, h = 0;
      if (s360_read_flow_from_file((flowPrevDir + "/flow_bottom_secondary.bin").c_str(), nullptr, &w, &h, 0) < 0)
        die(std::string("bad previous flow file: flow_bottom_secondary.bin: ") + s360_last_error(nullptr));
      std::vector<float> pf((size_t)w * h * 2);
      if (s360_read_flow_from_file((flowPrevDir + "/flow_bottom_secondary.bin").c_str(), pf.data(), &w, &h, pf.size()) < 0)
        die(std::string("bad previous flow file: flow_bottom_secondary.bin: ") + s360_last_error(nullptr));
      const pngio::Image b1 = load_png(imgPrevDir + "/bottomImage.png", true), b2 = load_png(imgPrevDir + "/bottomImage2.png", true);
      if (b1.c != 4 || b2.c != 4 || b1.w != w || b1.h != h || b2.w != w || b2.h != h)
        die("previous bottomImage / bottomImage2 have the wrong size/channels");
      ck(s360_frame_set_prev_pole_removal(ctx, pf.data(), b1.px.data(), b2.px.data(), w, h), ctx);
    }
    for (int u = 0; u < 4; ++u) {
      if ((u < 2 && !prm.enable_top) || (u >= 2 && !prm.enable_bottom)) continue;
      const int rows = u < 2 ? g.top_rows : g.bottom_rows;
      std::vector<float> pf((size_t)extW * rows * 2);
      int w = 0, h = 0;
      if (s360_read_flow_from_file((flowPrevDir + "/flow_" + eyeNames[u] + ".bin").c_str(), pf.data(), &w, &h, pf.size()) < 0 || w != extW || h != rows)
        die(std::string("bad previous pole flow file: ") + eyeNames[u]);
      const pngio::Image S = load_png(imgPrevDir + "/extendedSideSpherical_" + eyeNames[u] + ".png", true);
      const pngio::Image Fi = load_png(imgPrevDir + "/extendedFisheyeSpherical_" + eyeNames[u] + ".png", true);
      if (S.c != 4 || Fi.c != 4 || S.w != extW || S.h != rows || Fi.w != extW || Fi.h != rows)
        die("previous extended pole images have the wrong size/channels");
      ck(s360_frame_set_prev_pole(ctx, u, pf.data(), S.px.data(), Fi.px.data()), ctx);
    }
  }

  // ---- render on the GPU
  const double renderStart = now_sec();
  ck(s360_frame_render(ctx, usePrev ? 1 : 0), ctx);
  std::vector<uint8_t> equirect((size_t)g.out_width * g.out_height * 3);
  ck(s360_frame_download_equirect(ctx, equirect.data()), ctx);
  const double renderEnd = now_sec();

  // ---- state for the next frame: always written by the reference (TRSP:201-208, 247-255, 413-416, 451-452)
  if (F.b("write_state")) {
    const std::string flowDir = outData + "/flow/" + frame, flowImagesDir = outData + "/debug/" + frame + "/flow_images";
    mkdirs(flowDir);
    mkdirs(flowImagesDir);
    int whc[3];
    const size_t on = (size_t)g.overlap_image_width * g.cam_image_height;
    std::vector<uint8_t> img(on * 4);
    std::vector<float> fl(on * 2);
    for (int i = 0; i < P; ++i) {
      ck(s360_frame_get_u8(ctx, "overlap_l", i, whc, img.data()), ctx);
      save_png(flowImagesDir + "/overlap_" + std::to_string(i) + "_L.png", img.data(), whc[0], whc[1], 4);
      ck(s360_frame_get_u8(ctx, "overlap_r", i, whc, img.data()), ctx);
      save_png(flowImagesDir + "/overlap_" + std::to_string(i) + "_R.png", img.data(), whc[0], whc[1], 4);
      ck(s360_frame_get_f32(ctx, "flow_l_to_r", i, whc, fl.data()), ctx);
      ck(s360_save_flow_to_file((flowDir + "/flowLtoR_" + std::to_string(i) + ".bin").c_str(), fl.data(), whc[0], whc[1]), nullptr);
      ck(s360_frame_get_f32(ctx, "flow_r_to_l", i, whc, fl.data()), ctx);
      ck(s360_save_flow_to_file((flowDir + "/flowRtoL_" + std::to_string(i) + ".bin").c_str(), fl.data(), whc[0], whc[1]), nullptr);
    }
    if (prm.enable_pole_removal) {  // PoleRemoval.cpp:118-126 (kSaveDataNextFrame, TRSP:581)
      ck(s360_frame_get_u8(ctx, "bottom_image", 0, whc, nullptr), ctx);
      std::vector<uint8_t> bimg((size_t)whc[0] * whc[1] * 4);
      std::vector<float> bfl((size_t)whc[0] * whc[1] * 2);
      ck(s360_frame_get_u8(ctx, "bottom_image", 0, whc, bimg.data()), ctx);
      save_png(flowImagesDir + "/bottomImage.png", bimg.data(), whc[0], whc[1], 4);
      ck(s360_frame_get_u8(ctx, "bottom_image2", 0, whc, bimg.data()), ctx);
      save_png(flowImagesDir + "/bottomImage2.png", bimg.data(), whc[0], whc[1], 4);
      ck(s360_frame_get_f32(ctx, "flow_bottom_secondary", 0, whc, bfl.data()), ctx);
      ck(s360_save_flow_to_file((flowDir + "/flow_bottom_secondary.bin").c_str(), bfl.data(), whc[0], whc[1]), nullptr);
    }
    for (int u = 0; u < 4; ++u) {
      if ((u < 2 && !prm.enable_top) || (u >= 2 && !prm.enable_bottom)) continue;
      const int rows = u < 2 ? g.top_rows : g.bottom_rows;
      std::vector<uint8_t> e((size_t)extW * rows * 4);
      std::vector<float> pf((size_t)extW * rows * 2);
      ck(s360_frame_get_u8(ctx, "extended_side", u, whc, e.data()), ctx);
      save_png(flowImagesDir + "/extendedSideSpherical_" + eyeNames[u] + ".png", e.data(), whc[0], whc[1], 4);
      ck(s360_frame_get_u8(ctx, "extended_fisheye", u, whc, e.data()), ctx);
      save_png(flowImagesDir + "/extendedFisheyeSpherical_" + eyeNames[u] + ".png", e.data(), whc[0], whc[1], 4);
      ck(s360_frame_get_f32(ctx, "flow_pole", u, whc, pf.data()), ctx);
      ck(s360_save_flow_to_file((flowDir + "/flow_" + eyeNames[u] + ".bin").c_str(), pf.data(), whc[0], whc[1]), nullptr);
    }
  }
  const double stateEnd = now_sec();
  // optional stereo cubemap (TRSP:917-935)
  if (F.i("cubemap_width") > 0 && F.i("cubemap_height") > 0 && !F.s("output_cubemap_path").empty()) {
    int whc[3];
    ck(s360_frame_cubemap(ctx, F.i("cubemap_width"), F.i("cubemap_height"), F.s("cubemap_format").c_str(), whc, nullptr), ctx);
    std::vector<uint8_t> cube((size_t)whc[0] * whc[1] * 3);
    ck(s360_frame_cubemap(ctx, F.i("cubemap_width"), F.i("cubemap_height"), F.s("cubemap_format").c_str(), whc, cube.data()), ctx);
    save_png(F.s("output_cubemap_path"), cube.data(), whc[0], whc[1], 3);
  }
  save_png(F.s("output_equirect_path"), equirect.data(), g.out_width, g.out_height, 3);  // TRSP:961
  const double endTime = now_sec();
  if (verbose >= 1) {  // the reference's VLOG(1) runtime breakdown, TRSP:964-971
    std::fprintf(stderr, "--- Runtime breakdown (sec) ---\n");
    std::fprintf(stderr, "load + decode + upload:  %.3f\n", loadTime - startTime);
    std::fprintf(stderr, "previous-frame state:    %.3f\n", renderStart - loadTime);
    std::fprintf(stderr, "GPU render + download:   %.3f\n", renderEnd - renderStart);
    std::fprintf(stderr, "state files:             %.3f\n", stateEnd - renderEnd);
    std::fprintf(stderr, "equirect PNG encode:     %.3f\n", endTime - stateEnd);
    std::fprintf(stderr, "TOTAL:                   %.3f\n", endTime - startTime);
  }
  s360_destroy(ctx);
  return 0;
}
