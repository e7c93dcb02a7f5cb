// TEST INFRASTRUCTURE (oracle/_ref): the reference's Halide GENERATOR, camera_isp/CameraIspGen.cpp, compiled from /root/reference
// where it lies and EXECUTED: its main() (renamed) builds the pipelines over ref_shim/halide_eval/Halide.h — a lazy evaluator of the
// Halide front end it uses — and compile_to_static_library() files them under the generated functions' names
// (CameraIspGen.cpp:715-728). The four functions Halide would emit, with the generated signature the reference's CameraIspPipe.h
// calls (CameraIspPipe.h:143-175), are defined below as evaluations of those pipelines. Built by `make -C oracle ref` into
// oracle/_ref/libref_isppipe.so together with ref_isppipe.cpp (the reference's CameraIspPipe.h on top of these functions).
#define main cameraispgen_main
#include "CameraIspGen.cpp"
#undef main

#include <mutex>

namespace {
void build_pipelines() {
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* bpp : {"--output_bpp=8", "--output_bpp=16"}) {  // the reference's build runs the generator once per depth
      char a0[] = "CameraIspGen";
      std::string a1 = bpp;
      char* argv_[] = {a0, &a1[0], nullptr};
      char** argv = argv_;
      int argc = 2;
      std::streambuf* keep = std::cout.rdbuf(nullptr);  // "Halide: Generating .." lines
      const int rc = cameraispgen_main(argc, argv);
      std::cout.rdbuf(keep);
      if (rc != 0) throw std::runtime_error("CameraIspGen.cpp's main() failed");
    }
  });
}
int run(const char* name, buffer_t* input, int width, int height, buffer_t* vignetteH, buffer_t* vignetteV, const float* f17,
        buffer_t* ccm, buffer_t* toneTable, bool BGR, int bayerPattern, buffer_t* output) {
  build_pipelines();
  using Halide::Internal::CallArg;
  std::vector<CallArg> a;  // the order of `args` in the generator's main() (CameraIspGen.cpp:704-712)
  a.push_back(CallArg{input, 0});
  a.push_back(CallArg{nullptr, (double)width});
  a.push_back(CallArg{nullptr, (double)height});
  a.push_back(CallArg{vignetteH, 0});
  a.push_back(CallArg{vignetteV, 0});
  for (int i = 0; i < 17; ++i) a.push_back(CallArg{nullptr, (double)f17[i]});
  a.push_back(CallArg{ccm, 0});
  a.push_back(CallArg{toneTable, 0});
  a.push_back(CallArg{nullptr, BGR ? 1.0 : 0.0});
  a.push_back(CallArg{nullptr, (double)bayerPattern});
  return Halide::Internal::run_pipeline(name, a, output);
}
}  // namespace

#define REF_GENERATED(NAME)                                                                                                     \
  extern "C" int NAME(buffer_t* input, int width, int height, buffer_t* vignetteH, buffer_t* vignetteV, float blackLevelR,      \
                      float blackLevelG, float blackLevelB, float whiteBalanceGainR, float whiteBalanceGainG,                   \
                      float whiteBalanceGainB, float clampMinR, float clampMinG, float clampMinB, float clampMaxR,              \
                      float clampMaxG, float clampMaxB, float sharpeningR, float sharpeningG, float sharpeningB,                \
                      float sharpeningSupport, float noiseCore, buffer_t* ccm, buffer_t* toneTable, bool BGR, int bayerPattern, \
                      buffer_t* output) {                                                                                       \
    const float f[17] = {blackLevelR, blackLevelG, blackLevelB, whiteBalanceGainR, whiteBalanceGainG, whiteBalanceGainB,        \
                         clampMinR, clampMinG, clampMinB, clampMaxR, clampMaxG, clampMaxB, sharpeningR, sharpeningG,            \
                         sharpeningB, sharpeningSupport, noiseCore};                                                            \
    return run(#NAME, input, width, height, vignetteH, vignetteV, f, ccm, toneTable, BGR, bayerPattern, output);                \
  }
REF_GENERATED(CameraIspGen8)
REF_GENERATED(CameraIspGen16)
REF_GENERATED(CameraIspGenFast8)
REF_GENERATED(CameraIspGenFast16)
