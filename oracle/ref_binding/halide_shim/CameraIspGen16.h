// oracle/ref_binding/halide_shim/CameraIspGen16.h — TEST INFRASTRUCTURE: the header Halide's compile_to_static_library("CameraIspGen16", ..)
// writes (CameraIspGen.cpp:720-728), with the function implemented over libs360 instead of generated (camera_isp_gen_s360.h).
#pragma once
#include "camera_isp_gen_s360.h"
S360_HALIDE_GENERATED(CameraIspGen16, 0, 16)
