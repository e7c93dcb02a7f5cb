// oracle/ref_shim/halide_rt/Halide.h — TEST INFRASTRUCTURE (oracle/_ref). What a USER of a Halide-generated pipeline needs from
// "Halide.h": buffer_t and the namespace. The reference's camera_isp/CameraIspPipe.h compiles over this header and the four
// headers beside it, which declare the functions camera_isp/CameraIspGen.cpp generates; oracle/ref_ispgen.cpp defines them by
// EXECUTING that generator over ref_shim/halide_eval/Halide.h. (ref_binding/halide_shim/ is the other implementation of the same
// four functions: the library's kernels.)
#pragma once
#include "../halide_eval/halide_buffer_t.h"
namespace Halide {}
