// oracle/ref_binding/halide_shim/Halide.h — TEST INFRASTRUCTURE. What the reference's CameraIspPipe.h needs from "Halide.h" when
// it is compiled WITHOUT Halide: the runtime's buffer_t (the pre-2017 Halide ABI the reference is written against:
// HalideRuntime.h's `struct buffer_t`) and the namespace its `using namespace Halide;` names. With the four headers beside this
// one — the build outputs of CameraIspGen.cpp (CameraIspGen.cpp:720-728 compile_to_static_library), written here over the C ABI
// of libs360 — the reference's unmodified CameraIspPipe.h, Unpacker.cpp and Raw2Rgb.cpp (-DUSE_HALIDE) compile and run on the
// library: `make -C oracle ref_binding_pipe`.
#pragma once
#include <stdint.h>

typedef struct buffer_t {
  uint64_t dev;
  uint8_t* host;
  int32_t extent[4];
  int32_t stride[4];
  int32_t min[4];
  int32_t elem_size;
  bool host_dirty;
  bool dev_dirty;
} buffer_t;

namespace Halide {}
