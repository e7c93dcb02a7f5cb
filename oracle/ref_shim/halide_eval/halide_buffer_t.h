// TEST INFRASTRUCTURE (oracle/ref_shim). The buffer descriptor of the Halide runtime the reference is written against (HalideRuntime.h
// before 2017: `struct buffer_t`; CameraIspPipe.h:30-42 fills these fields).
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
