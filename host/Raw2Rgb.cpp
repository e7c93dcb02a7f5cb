// Raw2Rgb — the reference's raw-to-RGB converter (source/camera_isp/Raw2Rgb.cpp) on the GPU. Without --accelerate the soft
// ISP (CameraIsp.h, Raw2Rgb.cpp:441-456; pinned to the reference compiled); with --accelerate [--fast] the arithmetic of
// CameraIspPipe (Raw2Rgb.cpp:427-440: --demosaic_filter is not read there, --resize must be 1; restated from CameraIspGen.cpp,
// not pinned — include/s360.h, s360_isp_config.pipe). --input_image_path (an 8-/16-bit greyscale PNG, or a headerless 16-bit ".raw" whose size comes from the
// configuration's "width"/"height", Raw2Rgb.cpp:395-404), --isp_config_path (the ISP JSON), --output_image_path (8- or
// 16-bit RGB PNG by --output_bpp), --demosaic_filter (0 bilinear, 2 edge-aware; 1 = DCT is not available), --resize,
// --disable_tone_curve, --black_level_offset. 8-bit inputs are widened like convert8bitTo16bit (v << 8 | v,
// CvUtil.cpp:53-66). "Runtime = ... ms" is logged where the reference logs it (Raw2Rgb.cpp:369-373).
// Not produced: the DNG copy (--output_dng_path is accepted and ignored) and readRaw's "raw.tif" side file.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../include/s360.h"
#include "png_io.hpp"

namespace {
[[noreturn]] void die(const std::string& m) {  // VrCamException -> terminate handler -> abort (SystemUtil.cpp:42-61)
  std::fprintf(stderr, "Terminated with exception: %s\n", m.c_str());
  std::abort();
}
// the integer after "key" : in the "CameraIsp" object (getInteger(config, "CameraIsp", key)); -1 if absent
long json_int(const std::string& text, const char* key) {
  const std::string k = std::string("\"") + key + "\"";
  const size_t p = text.find(k);
  if (p == std::string::npos) return -1;
  const size_t c = text.find(':', p + k.size());
  if (c == std::string::npos) return -1;
  return std::strtol(text.c_str() + c + 1, nullptr, 10);
}
}  // namespace

int main(int argc, char** argv) {
  std::map<std::string, std::string> F = {{"input_image_path", ""}, {"output_image_path", ""}, {"output_dng_path", ""},
                                          {"isp_config_path", ""}, {"black_level_offset", "0"}, {"demosaic_filter", "2"},
                                          {"resize", "1"}, {"output_bpp", "8"}, {"disable_tone_curve", "false"},
                                          {"accelerate", "false"}, {"fast", "false"}, {"device", "0"}, {"log_dir", ""},
                                          {"stderrthreshold", "0"}, {"v", "0"}, {"logbuflevel", "0"}};
  const char* bools[] = {"disable_tone_curve", "accelerate", "fast"};
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.size() < 2 || a[0] != '-') die("unexpected argument: " + a);
    a = a.substr(a[1] == '-' ? 2 : 1);
    std::string key = a, val = "true";
    const size_t eq = a.find('=');
    bool is_bool = false;
    if (eq != std::string::npos) {
      key = a.substr(0, eq);
      val = a.substr(eq + 1);
    } else {
      for (const char* b : bools) is_bool = is_bool || key == b;
      if (!is_bool) {
        if (i + 1 >= argc) die("flag '" + key + "' is missing its argument");
        val = argv[++i];
      }
    }
    if (!F.count(key)) {
      std::fprintf(stderr, "ERROR: unknown command line flag '%s'\n", key.c_str());
      return 1;
    }
    F[key] = val;
  }
  for (const char* k : {"input_image_path", "output_image_path", "isp_config_path"})
    if (F[k].empty()) die(std::string("missing required command line argument: ") + k);
  const bool tone_off = F["disable_tone_curve"] == "true" || F["disable_tone_curve"] == "1";

  std::ifstream ifs(F["isp_config_path"]);
  if (!ifs) die("file read failed: " + F["isp_config_path"]);
  std::stringstream ss;
  ss << ifs.rdbuf();
  const std::string json = ss.str();

  s360_isp_config cfg;
  s360_isp_config_defaults(&cfg);
  cfg.output_bpp = std::atoi(F["output_bpp"].c_str());
  cfg.demosaic_filter = std::atoi(F["demosaic_filter"].c_str());
  cfg.resize = std::atoi(F["resize"].c_str());
  cfg.disable_tone_curve = tone_off ? 1 : 0;
  cfg.black_level_offset = std::atoi(F["black_level_offset"].c_str());
  const auto on = [&](const char* k) { return F[k] == "true" || F[k] == "1"; };
  cfg.pipe = on("accelerate") ? (on("fast") ? 2 : 1) : 0;  // CameraIspPipe(json, FLAGS_fast, FLAGS_output_bpp), Raw2Rgb.cpp:427-428
  if (s360_isp_config_from_json(json.c_str(), &cfg) < 0) die(s360_last_error(nullptr));

  // input: ".raw" = headerless 16-bit samples (readRaw), anything else is decoded as a greyscale image of unchanged depth
  int w = 0, h = 0;
  std::vector<uint16_t> raw;
  if (F["input_image_path"].find(".raw") != std::string::npos) {
    w = (int)json_int(json, "width");
    h = (int)json_int(json, "height");
    if (w <= 0 || h <= 0) die("the ISP configuration has no width / height for a .raw input");
    raw.assign((size_t)w * h, 0);
    std::ifstream in(F["input_image_path"], std::ios::binary);
    if (!in) die("file read failed: " + F["input_image_path"]);
    in.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)raw.size() * 2);
    if ((size_t)in.gcount() != raw.size() * 2)
      std::fprintf(stderr, "Warning: expected %zu but only read %zu\n", raw.size() * 2, (size_t)in.gcount());
  } else {
    int depth = 0;
    try {
      raw = pngio::read_gray(F["input_image_path"], &w, &h, &depth);
    } catch (const std::exception& e) {
      die(e.what());
    }
    if (depth == 8) {
      std::fprintf(stderr, "8 bit raw\n");
      for (uint16_t& v : raw) v = (uint16_t)(v << 8 | v);  // convert8bitTo16bit
    } else {
      std::fprintf(stderr, "16 bit raw\n");
    }
  }
  if (!(w > 2 && h > 2)) die("Unable to open " + F["input_image_path"]);

  s360_isp* isp = nullptr;
  if (s360_isp_create(&isp, std::atoi(F["device"].c_str()), &cfg) < 0) die(s360_last_error(nullptr));
  const int ow = w / cfg.resize, oh = h / cfg.resize;
  std::vector<uint8_t> out((size_t)ow * oh * 3 * (cfg.output_bpp == 8 ? 1 : 2));
  const auto t0 = std::chrono::steady_clock::now();
  if (s360_isp_process(isp, raw.data(), w, h, out.data()) < 0) die(s360_last_error(nullptr));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::fprintf(stderr, "Runtime = %gms\n", ms);
  try {
    if (cfg.output_bpp == 8) pngio::write(F["output_image_path"], out.data(), ow, oh, 3);
    else pngio::write16(F["output_image_path"], reinterpret_cast<const uint16_t*>(out.data()), ow, oh);
  } catch (const std::exception& e) {
    die(e.what());
  }
  s360_isp_destroy(isp);
  return 0;
}
