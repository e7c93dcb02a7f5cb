// valu_latency.hip — dependent / independent issue cost of common VALU ops for ONE wave on a SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float a, float b) {
  float x0 = a + threadIdx.x, x1 = b, x2 = a * 2, x3 = b * 3;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a, b}, p1 = {b, a};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {
    if (MODE == 0) {  // 16 dependent v_fma
#pragma unroll
      for (int k = 0; k < 16; ++k) x0 = __builtin_fmaf(x0, a, b);
    } else if (MODE == 1) {  // 16 fma, 4 independent chains
#pragma unroll
      for (int k = 0; k < 4; ++k) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b); }
    } else if (MODE == 2) {  // 16 dependent v_pk_fma
#pragma unroll
      for (int k = 0; k < 16; ++k) p0 = __builtin_elementwise_fma(p0, p1, p1);
    } else if (MODE == 3) {  // 16 dependent add (non-fma)
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(a));
    } else if (MODE == 4) {  // 16 dependent dpp movs
#pragma unroll
      for (int k = 0; k < 16; ++k) x0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x0), 0x151, 0xF, 0xF, true));
    } else if (MODE == 5) {  // 16 dependent: cmp + cndmask pairs (8 pairs)
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(x0) : "v"(x1), "v"(x2) : "vcc");
    } else if (MODE == 6) {  // 16 dependent v_sqrt
#pragma unroll
      for (int k = 0; k < 16; ++k) x0 = __builtin_amdgcn_sqrtf(x0);
    } else if (MODE == 7) {  // 16 dependent cvt pairs
#pragma unroll
      for (int k = 0; k < 8; ++k) { int q = (int)x0; x0 = (float)q + a; }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x0 + x1 + x2 + x3 + p0.x + p0.y;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 8));
  const char* names[] = {"16 dep v_fma_f32", "16 fma in 4 chains", "16 dep v_pk_fma_f32", "16 dep v_add_f32", "16 dep v_mov_dpp", "8 dep cmp+cndmask", "16 dep v_sqrt_f32", "8 dep cvt_i32+cvt_f32+add"};
  for (int m = 0; m < 8; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (m) {
        case 0: hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, out, cyc, 1.0001f, 0.5f); break;
        case 1: hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, out, cyc, 1.0001f, 0.5f); break;
        case 2: hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, out, cyc, 1.0001f, 0.5f); break;
        case 3: hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, out, cyc, 1.0001f, 0.5f); break;
        case 4: hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, out, cyc, 1.0001f, 0.5f); break;
        case 5: hipLaunchKernelGGL(k<5>, 1, 64, 0, 0, out, cyc, 1.0001f, 0.5f); break;
        case 6: hipLaunchKernelGGL(k<6>, 1, 64, 0, 0, out, cyc, 1.0001f, 0.5f); break;
        case 7: hipLaunchKernelGGL(k<7>, 1, 64, 0, 0, out, cyc, 1.0001f, 0.5f); break;
      }
      CK(hipDeviceSynchronize());
    }
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-28s %8.2f ticks per loop body (256 iterations)\n", names[m], c / 256.0);
  }
  // wall-clock calibration of the s_memtime tick
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a)); for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, out, cyc, 1.0001f, 0.5f); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("k<0>: %.2f us per launch wall, %llu ticks in-kernel\n", ms * 1000 / 50, c);
  return 0;
}
