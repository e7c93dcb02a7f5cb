// DEVELOPER / TEST TOOL — not part of the product. A stand-in for <hip/hip_runtime.h> that runs the wave-synchronous
// HIP kernels of libs360 (the PixFlow sweeps: surround360_amd/csrc/sweep_lock.hip, sweep_quad.hip) on the CPU with the
// execution model they rely on:
//   * a workgroup is one OS thread; its GPU threads are coroutines on that thread, scheduled wave by wave;
//   * cross-lane operations (DPP, ballot, readfirstlane) and s_barrier are rendezvous points: every live lane of the
//     wave (workgroup) arrives before any continues — the kernels only use them under wave-uniform control flow, and
//     the scheduler aborts if it ever sees a wave split between two different rendezvous;
//   * between two rendezvous the lanes of a wave run one after the other (forward, reverse or shuffled order:
//     EMU_LANE_ORDER), so an LDS hand-over between lanes of one wave that the hardware orders by lockstep execution must
//     be marked in the source with S360_WAVE_SYNC() — running the tests under different lane orders finds missing marks;
//   * workgroups run concurrently (one OS thread each) and talk through global memory with real atomics, so the
//     granule hand-off between bands, its on-demand waits and the ticket counter run as they do on the GPU;
//   * __shared__ is per workgroup (static thread_local).
// Float arithmetic is IEEE on both sides (-ffp-contract=off, fmaf exact); v_sqrt_f32's 1-ulp result is replaced by the
// correctly rounded one, which the kernels' fix-up leaves unchanged. It is NOT a fallback: libs360 never links it.
// Build with clang++ (ext_vector_type, __builtin_nontemporal_load): see tools/Makefile, target sweep_emulate.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint3_ { unsigned x, y, z; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(4) uchar4 { unsigned char x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }

namespace emu {
enum Op { OP_BALLOT = 1, OP_FIRST, OP_DPP, OP_SYNC };
const uint3_& tid();
const uint3_& bid();
const dim3& bdim();
const dim3& gdim();
int lane_id();  // 0..63 within the wave
// deposit `v`, wait for the wave, then read any participant's value with peer()
struct Rendezvous {
  int slot;
  unsigned seq;
};
Rendezvous arrive(Op op, unsigned long long v);
bool peer(const Rendezvous& r, int lane, unsigned long long* v);  // false: that lane did not take part (has exited)
void barrier();
void nap();  // s_sleep: lets the other waves / workgroups run
inline void wave_sync() { (void)arrive(OP_SYNC, 0); }
void launch(std::function<void()> body, dim3 grid, dim3 block);
}  // namespace emu

#define threadIdx (emu::tid())
#define blockIdx (emu::bid())
#define blockDim (emu::bdim())
#define gridDim (emu::gdim())

// ---- host API (just enough for the launchers) ----
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
struct hipEvent_st;
typedef hipEvent_st* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* t, hipEvent_t, hipEvent_t) { *t = 0; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(1, n ? n : 1); return hipSuccess; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) {  // EMU_CUS: "compute units" of the emulated device
  const char* e = std::getenv("EMU_CUS");
  *v = e ? std::atoi(e) : 1;
  return hipSuccess;
}
template <typename K, typename... A>
void emu_launch_kernel(K kernel, dim3 grid, dim3 block, A... args) {
  emu::launch([=] { kernel(args...); }, grid, block);
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch_kernel(kernel, dim3(grid), dim3(block), __VA_ARGS__)

// ---- device functions ----
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
inline int __mul24(int a, int b) { return (int)((unsigned)((a << 8) >> 8) * (unsigned)((b << 8) >> 8)); }
inline int __ffsll(long long v) { return v ? __builtin_ctzll((unsigned long long)v) + 1 : 0; }
using std::max;
using std::min;
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, __ATOMIC_ACQUIRE)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, __ATOMIC_RELEASE)
inline void __syncthreads() { emu::barrier(); }

inline unsigned long long __ballot(int pred) {
  const emu::Rendezvous r = emu::arrive(emu::OP_BALLOT, pred ? 1u : 0u);
  unsigned long long m = 0, v;
  for (int l = 0; l < 64; ++l)
    if (emu::peer(r, l, &v) && v) m |= 1ull << l;
  return m;
}
template <typename T>
inline T emu_readfirstlane(T x) {
  static_assert(sizeof(T) == 4, "32-bit values only");
  unsigned u;
  std::memcpy(&u, &x, 4);
  const emu::Rendezvous r = emu::arrive(emu::OP_FIRST, u);
  unsigned long long v = 0;
  for (int l = 0; l < 64; ++l)
    if (emu::peer(r, l, &v)) break;
  u = (unsigned)v;
  std::memcpy(&x, &u, 4);
  return x;
}
#define __builtin_amdgcn_readfirstlane(x) emu_readfirstlane(x)
// v_mov_b32_dpp semantics (the controls the kernels use): quad_perm, row_shr:n, row_bcast:15, row_newbcast:n
inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const emu::Rendezvous r = emu::arrive(emu::OP_DPP, (unsigned)src);
  const int lane = emu::lane_id(), row = lane >> 4, inrow = lane & 15;
  if (!((row_mask >> row) & 1) || !((bank_mask >> (inrow >> 2)) & 1)) return old;
  int from = -1;
  if (ctrl >= 0 && ctrl <= 0xFF) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl >= 0x111 && ctrl <= 0x11F) from = inrow >= (ctrl & 15) ? lane - (ctrl & 15) : -1;
  else if (ctrl == 0x142) from = row > 0 ? row * 16 - 1 : -1;
  else if (ctrl >= 0x150 && ctrl <= 0x15F) from = (lane & ~15) | (ctrl & 15);
  else { std::fprintf(stderr, "emu: DPP control 0x%x is not implemented\n", ctrl); std::abort(); }
  unsigned long long v;
  if (from >= 0 && emu::peer(r, from, &v)) return (int)(unsigned)v;
  return bound_ctrl ? 0 : old;
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu_update_dpp(old, src, ctrl, rm, bm, bc)
inline float emu_fmed3f(float a, float b, float c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
#define __builtin_amdgcn_fmed3f(a, b, c) emu_fmed3f(a, b, c)
#define __builtin_amdgcn_fractf(x) ((x) - std::floor(x))
#define __builtin_amdgcn_sqrtf(x) std::sqrt((float)(x))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) emu::nap()
#define __builtin_amdgcn_s_memtime() 0ull
