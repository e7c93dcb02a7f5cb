// Unpacker — the reference's footage unpacker (source/camera_isp/Unpacker.cpp) on the GPU: reads the capture's .bin
// containers (BinaryFootageFile.cpp: a 4096-byte metadata page {magic 0xfaceb00c, timestamp, fileIndex, fileCount,
// width, height, bitsPerPixel, numberOfCameras}, then frames interleaved by camera, 8 or 12 bits per pixel packed; the
// camera's serial number is the second 32-bit word of every frame), and for every camera and frame
//   * writes the widened 16-bit raw image to <output_raw_dir>/<serial>/<frame>.tiff (if --output_raw_dir is given),
//   * runs the ISP configured by <isp_dir>/<serial>.json — if that file exists — and writes the 16-bit result to
//     <output_dir>/<serial>/<frame>.png (Unpacker.cpp:136-183),
// then renames <output_dir>/<serial> to cam0, cam1, ... in serial order (Unpacker.cpp:203-219). One host thread per
// camera like the reference's std::async tasks; each camera owns one ISP object (the reference builds one per frame).
// The ISP is the reference Unpacker's: CameraIspPipe(json, fast = false, 16 bits) (Unpacker.cpp:165-183), i.e. the accelerated
// pipeline's arithmetic (s360_isp_config.pipe = 1: restated from CameraIspGen.cpp, not pinned — include/s360.h). --soft_isp
// runs the frames through the soft CameraIsp arithmetic instead (pinned bit for bit to CameraIsp.h).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/s360.h"
#include "footage.hpp"
#include "png_io.hpp"

namespace {
[[noreturn]] void die(const std::string& m) {
  std::fprintf(stderr, "Terminated with exception: %s\n", m.c_str());
  std::abort();
}
using footage::Footage;
// RawConverter::convert8Frame / convert12Frame on the host, for the raw dump only (the ISP widens on the device)
void widen(const uint8_t* frame, int bits, int w, int h, std::vector<uint16_t>& out) {
  out.resize((size_t)w * h);
  if (bits == 8) {
    for (size_t i = 0; i < out.size(); ++i) out[i] = (uint16_t)(frame[i] * 0x101);
    return;
  }
  size_t p = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const uint16_t lo = frame[p], hi = frame[p + 1];
      uint16_t u;
      if (x & 1) { p += 2; u = (uint16_t)(hi << 4 | lo >> 4); }
      else { p += 1; u = (uint16_t)(lo << 4 | (hi & 0xF)); }
      out[(size_t)y * w + x] = (uint16_t)(u << 4 | u >> 8);
    }
}
// baseline TIFF, one strip, 16-bit greyscale, no compression (imwrite(..., tiffParams = {259, 1}), CvUtil.h:27)
void write_tiff_gray16(const std::string& path, const uint16_t* px, int w, int h) {
  pngio::OutFile f(path);
  const uint32_t nbytes = (uint32_t)((size_t)w * h * 2), ifd = 8 + nbytes;
  uint8_t hdr[8] = {'I', 'I', 42, 0, (uint8_t)ifd, (uint8_t)(ifd >> 8), (uint8_t)(ifd >> 16), (uint8_t)(ifd >> 24)};
  f.put(hdr, 8);
  f.put(reinterpret_cast<const uint8_t*>(px), nbytes);  // little-endian samples on this host
  struct Tag { uint16_t id, type; uint32_t count, value; };
  const Tag tags[] = {{256, 4, 1, (uint32_t)w}, {257, 4, 1, (uint32_t)h}, {258, 3, 1, 16}, {259, 3, 1, 1}, {262, 3, 1, 1},
                      {273, 4, 1, 8}, {277, 3, 1, 1}, {278, 4, 1, (uint32_t)h}, {279, 4, 1, nbytes}};
  const uint16_t n = sizeof(tags) / sizeof(tags[0]);
  std::vector<uint8_t> d(2 + 12 * n + 4, 0);
  d[0] = (uint8_t)n; d[1] = (uint8_t)(n >> 8);
  for (int i = 0; i < n; ++i) {
    uint8_t* e = &d[2 + 12 * i];
    std::memcpy(e, &tags[i].id, 2); std::memcpy(e + 2, &tags[i].type, 2); std::memcpy(e + 4, &tags[i].count, 4);
    if (tags[i].type == 3) { const uint16_t v = (uint16_t)tags[i].value; std::memcpy(e + 8, &v, 2); }
    else std::memcpy(e + 8, &tags[i].value, 4);
  }
  f.put(d.data(), d.size());
  f.close();
}
std::string frame_path(const std::string& dir, uint32_t serial, size_t frame, const char* ext) {
  char buf[64];
  std::snprintf(buf, sizeof buf, "/%u/%06zu%s", serial, frame, ext);
  return dir + buf;
}
}  // namespace

int main(int argc, char** argv) {
  std::map<std::string, std::string> F = {{"isp_dir", ""}, {"output_dir", ""}, {"output_raw_dir", ""}, {"bin_list", ""},
                                          {"start_frame", "0"}, {"frame_count", "0"}, {"device", "0"}, {"soft_isp", "false"}, {"log_dir", ""},
                                          {"stderrthreshold", "0"}, {"v", "0"}, {"logbuflevel", "0"}};
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.size() < 2 || a[0] != '-') die("unexpected argument: " + a);
    a = a.substr(a[1] == '-' ? 2 : 1);
    std::string key = a, val;
    const size_t eq = a.find('=');
    if (eq != std::string::npos) { key = a.substr(0, eq); val = a.substr(eq + 1); }
    else if (key == "soft_isp") val = "true";  // (a boolean flag)
    else { if (i + 1 >= argc) die("flag '" + key + "' is missing its argument"); val = argv[++i]; }
    if (!F.count(key)) { std::fprintf(stderr, "ERROR: unknown command line flag '%s'\n", key.c_str()); return 1; }
    F[key] = val;
  }
  for (const char* k : {"isp_dir", "output_dir", "bin_list"})
    if (F[k].empty()) die(std::string("missing required command line argument: ") + k);
  const int device = std::atoi(F["device"].c_str());
  const long startFrame = std::atol(F["start_frame"].c_str()), frameCountFlag = std::atol(F["frame_count"].c_str());
  if (startFrame < 0 || frameCountFlag < 0) die("--start_frame and --frame_count must not be negative");

  std::set<uint32_t> serials;
  std::mutex mu;
  std::istringstream list(F["bin_list"]);
  std::string bin;
  try {
    while (std::getline(list, bin, ',')) {
      Footage ff;
      ff.path = bin;
      std::fprintf(stderr, "Reading %s...\n", bin.c_str());
      ff.open();
      const int ncam = (int)ff.md.numberOfCameras;
      if (ncam == 0) { std::fprintf(stderr, "No cameras found...\n"); continue; }
      if (ff.md.bitsPerPixel != 8 && ff.md.bitsPerPixel != 12) throw std::runtime_error("unsupported bits per pixel in " + bin);
      const long endMax = (long)ff.frames() - 1;
      long endFrame = frameCountFlag == 0 ? endMax : startFrame + frameCountFlag - 1;
      if (startFrame > endMax) {
        std::ostringstream ss;
        ss << "Start frame (" << startFrame << ") larger than total number of frames (" << endMax << ")";
        throw std::runtime_error(ss.str());
      }
      if (endFrame > endMax) {
        std::fprintf(stderr, "End frame (%ld) larger than total number of frames (%ld)\n", endFrame, endMax);
        endFrame = endMax;
      }
      const int w = (int)ff.md.width, h = (int)ff.md.height, bits = (int)ff.md.bitsPerPixel;
      std::vector<std::future<void>> tasks;
      for (int cam = 0; cam < ncam; ++cam)
        tasks.push_back(std::async(std::launch::async, [&, cam] {
          s360_isp* isp = nullptr;
          uint32_t ispSerial = 0;
          std::vector<uint16_t> raw16;
          std::vector<uint16_t> colored((size_t)w * h * 3);
          for (long f = startFrame; f <= endFrame; ++f) {
            const uint8_t* frame = ff.frame((size_t)f, (size_t)cam);
            uint32_t serial;
            std::memcpy(&serial, frame + 4, 4);
            {
              std::lock_guard<std::mutex> lk(mu);
              if (!serials.count(serial)) {
                serials.insert(serial);
                mkdir((F["output_dir"] + "/" + std::to_string(serial)).c_str(), 0755);
                if (!F["output_raw_dir"].empty()) mkdir((F["output_raw_dir"] + "/" + std::to_string(serial)).c_str(), 0755);
              }
            }
            if (!F["output_raw_dir"].empty()) {
              widen(frame, bits, w, h, raw16);
              write_tiff_gray16(frame_path(F["output_raw_dir"], serial, (size_t)f, ".tiff"), raw16.data(), w, h);
            }
            const std::string json_path = F["isp_dir"] + "/" + std::to_string(serial) + ".json";
            std::ifstream js(json_path);
            if (!js) {  // "we can still unpack raws"
              std::fprintf(stderr, "Cannot convert to RGB, file not found: %s\n", json_path.c_str());
              continue;
            }
            if (!isp || ispSerial != serial) {
              if (isp) s360_isp_destroy(isp);
              isp = nullptr;
              std::stringstream ss;
              ss << js.rdbuf();
              s360_isp_config cfg;
              s360_isp_config_defaults(&cfg);
              cfg.output_bpp = 16;  // kOutputBpp (Unpacker.cpp:167)
              cfg.pipe = (F["soft_isp"] == "true" || F["soft_isp"] == "1") ? 0 : 1;  // CameraIspPipe, kFast = false (:166-168)
              if (s360_isp_config_from_json(ss.str().c_str(), &cfg) < 0) throw std::runtime_error(s360_last_error(nullptr));
              if (s360_isp_create(&isp, device, &cfg) < 0) throw std::runtime_error(s360_last_error(nullptr));
              ispSerial = serial;
            }
            if (s360_isp_process_packed(isp, frame, bits, w, h, colored.data()) < 0) throw std::runtime_error(s360_last_error(nullptr));
            pngio::write16(frame_path(F["output_dir"], serial, (size_t)f, ".png"), colored.data(), w, h, 1, 1);
          }
          if (isp) s360_isp_destroy(isp);
        }));
      for (auto& t : tasks) t.get();
    }
  } catch (const std::exception& e) {
    die(e.what());
  }
  size_t ordinal = 0;  // rename <output_dir>/<serial> to camN, sorted by serial number
  for (uint32_t serial : serials) {
    const std::string from = F["output_dir"] + "/" + std::to_string(serial), to = F["output_dir"] + "/cam" + std::to_string(ordinal++);
    std::rename(from.c_str(), to.c_str());
  }
  return 0;
}
