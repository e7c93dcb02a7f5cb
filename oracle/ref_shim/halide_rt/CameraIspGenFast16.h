// oracle/ref_shim/halide_rt/CameraIspGenFast16.h — TEST INFRASTRUCTURE: the header compile_to_static_library("CameraIspGenFast16", ..) writes
// (CameraIspGen.cpp:715-728); the function is the generator's pipeline, evaluated (oracle/ref_ispgen.cpp).
#pragma once
#include "Halide.h"
extern "C" int CameraIspGenFast16(buffer_t* input, int width, int height, buffer_t* vignetteH, buffer_t* vignetteV, float blackLevelR,
    float blackLevelG, float blackLevelB, float whiteBalanceGainR, float whiteBalanceGainG, float whiteBalanceGainB,
    float clampMinR, float clampMinG, float clampMinB, float clampMaxR, float clampMaxG, float clampMaxB, float sharpeningR,
    float sharpeningG, float sharpeningB, float sharpeningSupport, float noiseCore, buffer_t* ccm, buffer_t* toneTable, bool BGR,
    int bayerPattern, buffer_t* output);
