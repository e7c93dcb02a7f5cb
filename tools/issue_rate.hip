// issue_rate.hip — how many instructions per cycle does one SIMD of gfx950 issue when k waves share it, for pure
// VALU, pure SALU and mixed streams? (Does a SALU / branch / s_nop of one wave issue beside a VALU of another?)
// Workgroups are 256 threads = one wave per SIMD of a CU; the grid is k workgroups per CU.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define V4 "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
#define S4 "s_add_u32 %4, %4, 1\n\ts_add_u32 %5, %5, 1\n\ts_add_u32 %6, %6, 1\n\ts_add_u32 %7, %7, 1\n\t"
#define VS4 "v_fma_f32 %0, %0, %8, %9\n\ts_add_u32 %4, %4, 1\n\tv_fma_f32 %1, %1, %8, %9\n\ts_add_u32 %5, %5, 1\n\tv_fma_f32 %2, %2, %8, %9\n\ts_add_u32 %6, %6, 1\n\tv_fma_f32 %3, %3, %8, %9\n\ts_add_u32 %7, %7, 1\n\t"
#define VN4 "v_fma_f32 %0, %0, %8, %9\n\ts_nop 0\n\tv_fma_f32 %1, %1, %8, %9\n\ts_nop 0\n\tv_fma_f32 %2, %2, %8, %9\n\ts_nop 0\n\tv_fma_f32 %3, %3, %8, %9\n\ts_nop 0\n\t"
#define VB4 "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\ts_cbranch_scc1 1f\n1:\n\t"
#define VW4 "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\ts_waitcnt vmcnt(0)\n\t"
#define VE4 "v_fma_f32 %0, %0, %8, %9\n\ts_and_b64 vcc, exec, vcc\n\tv_fma_f32 %1, %1, %8, %9\n\ts_or_b64 vcc, vcc, exec\n\tv_fma_f32 %2, %2, %8, %9\n\ts_and_b64 vcc, exec, vcc\n\tv_fma_f32 %3, %3, %8, %9\n\ts_or_b64 vcc, vcc, exec\n\t"

// per-opcode rates (round 5): 32 independent instructions of ONE opcode per loop body — does the opcode issue at v_fma_f32's rate?
#define P4 "v_perm_b32 %0, %0, %8, %9\n\tv_perm_b32 %1, %1, %8, %9\n\tv_perm_b32 %2, %2, %8, %9\n\tv_perm_b32 %3, %3, %8, %9\n\t"
#define D4 "v_dot2c_i32_i16 %0, %8, %9\n\tv_dot2c_i32_i16 %1, %8, %9\n\tv_dot2c_i32_i16 %2, %8, %9\n\tv_dot2c_i32_i16 %3, %8, %9\n\t"
#define M4 "v_min_f32 %0, %0, %8\n\tv_max_f32 %1, %1, %8\n\tv_min_f32 %2, %2, %9\n\tv_max_f32 %3, %3, %9\n\t"
#define A4 "v_add_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %9\n\tv_mul_f32 %3, %3, %9\n\t"
#define C4 "v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %9, vcc\n\tv_cndmask_b32 %3, %3, %9, vcc\n\t"
#define I4 "v_add_u32 %0, %0, %8\n\tv_lshlrev_b32 %1, 1, %1\n\tv_and_b32 %2, %2, %9\n\tv_mad_u32_u24 %3, %3, %8, %9\n\t"
#define DP4 "v_mov_b32_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %3 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0 = a + threadIdx.x, x1 = b, x2 = a * 2, x3 = b * 3;
  unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4;
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
    // every body is 32 instructions
    if (MODE == 0)
      asm volatile(V4 V4 V4 V4 V4 V4 V4 V4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 1)
      asm volatile(S4 S4 S4 S4 S4 S4 S4 S4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 2)
      asm volatile(VS4 VS4 VS4 VS4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 3)
      asm volatile(VN4 VN4 VN4 VN4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 4)
      asm volatile(VB4 VB4 VB4 VB4 VB4 VB4 VB4 VB4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 5)
      asm volatile(VW4 VW4 VW4 VW4 VW4 VW4 VW4 VW4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 7)
      asm volatile(P4 P4 P4 P4 P4 P4 P4 P4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 8)
      asm volatile(D4 D4 D4 D4 D4 D4 D4 D4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 9)
      asm volatile(M4 M4 M4 M4 M4 M4 M4 M4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 10)
      asm volatile(A4 A4 A4 A4 A4 A4 A4 A4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 11)
      asm volatile(C4 C4 C4 C4 C4 C4 C4 C4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 12)
      asm volatile(I4 I4 I4 I4 I4 I4 I4 I4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 13)
      asm volatile(DP4 DP4 DP4 DP4 DP4 DP4 DP4 DP4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
    else if (MODE == 6)
      asm volatile(VE4 VE4 VE4 VE4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a), "v"(b) : "scc", "vcc");
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (float)(s0 + s1 + s2 + s3);
}

template <int MODE>
static double run(float* out, int wgs, int iters) {
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, 16, 1.0001f, 0.5f);
  (void)hipDeviceSynchronize();
  const auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
  (void)hipDeviceSynchronize();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  float* out;
  CK(hipMalloc(&out, 256 * 8 * 256 * 4 * 2));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  printf("%d CUs, nominal %.2f GHz; 32-instruction loop bodies, instructions per nominal cycle per SIMD:\n", cus, ghz);
  const char* names[] = {"32 VALU", "32 SALU", "16 VALU + 16 SALU", "16 VALU + 16 s_nop", "24 VALU + 8 branch (not taken)",
                         "24 VALU + 8 s_waitcnt", "16 VALU + 16 s_and/or_b64", "32 v_perm_b32", "32 v_dot2c_i32_i16", "32 v_min/max_f32",
                         "32 v_add/mul_f32", "32 v_cndmask_b32", "32 int (add/lshl/and/mad24)", "32 v_mov_b32_dpp (row bcast)"};
  const int iters = 200000;
  for (int m = 0; m < 14; ++m) {
    printf("%-32s", names[m]);
    for (int kw : {1, 2, 4, 8}) {
      const int wgs = cus * kw;
      double sec = 0;
      switch (m) {
        case 0: sec = run<0>(out, wgs, iters); break;
        case 1: sec = run<1>(out, wgs, iters); break;
        case 2: sec = run<2>(out, wgs, iters); break;
        case 3: sec = run<3>(out, wgs, iters); break;
        case 4: sec = run<4>(out, wgs, iters); break;
        case 5: sec = run<5>(out, wgs, iters); break;
        case 6: sec = run<6>(out, wgs, iters); break;
        case 7: sec = run<7>(out, wgs, iters); break;
        case 8: sec = run<8>(out, wgs, iters); break;
        case 9: sec = run<9>(out, wgs, iters); break;
        case 10: sec = run<10>(out, wgs, iters); break;
        case 11: sec = run<11>(out, wgs, iters); break;
        case 12: sec = run<12>(out, wgs, iters); break;
        case 13: sec = run<13>(out, wgs, iters); break;
      }
      const double instr_per_simd = (double)kw * iters * 34.0;  // + loop counter and branch
      printf("  k=%d: %.3f", kw, instr_per_simd / (sec * ghz * 1e9));
    }
    printf("\n");
  }
  return 0;
}
