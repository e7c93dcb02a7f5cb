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
#define EMU_VEC(T, name, al2, al4)                                                          \
  struct alignas(al2) name##2 { T x, y; };                                                   \
  struct name##3 { T x, y, z; };                                                             \
  struct alignas(al4) name##4 { T x, y, z, w; };                                             \
  inline name##2 make_##name##2(T x, T y) { return name##2{x, y}; }                          \
  inline name##3 make_##name##3(T x, T y, T z) { return name##3{x, y, z}; }                  \
  inline name##4 make_##name##4(T x, T y, T z, T w) { return name##4{x, y, z, w}; }
EMU_VEC(float, float, 8, 16)
EMU_VEC(double, double, 16, 32)
EMU_VEC(int, int, 8, 16)
EMU_VEC(unsigned, uint, 8, 16)
EMU_VEC(short, short, 4, 8)
EMU_VEC(unsigned short, ushort, 4, 8)
EMU_VEC(signed char, char, 2, 4)
EMU_VEC(unsigned char, uchar, 2, 4)
#undef EMU_VEC

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
int barrier_and(int pred);  // __syncthreads_and
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
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventBlockingSync = 1, hipHostMallocDefault = 0, hipHostMallocPortable = 1 };
// everything is synchronous here: a kernel has run when its launch returns, copies are memcpy, streams and events are names
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { const char* e = std::getenv("EMU_DEVICES"); *n = e && std::atoi(e) > 0 ? std::atoi(e) : 1; return hipSuccess; }  // EMU_DEVICES: emulated GPUs (multi-GPU host program runs)
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = std::calloc(1, n ? n : 1); return hipSuccess; }
template <typename T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)64 << 30; return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
struct hipEvent_st;
typedef hipEvent_st* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
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
inline int __mul24(int a, int b) {  // low 24 bits of each operand, sign-extended (shifts on the unsigned value: no undefined behaviour)
  const int sa = (int)((unsigned)a << 8) >> 8, sb = (int)((unsigned)b << 8) >> 8;
  return (int)((unsigned)sa * (unsigned)sb);
}
inline int __ffsll(long long v) { return v ? __builtin_ctzll((unsigned long long)v) + 1 : 0; }
using std::max;
using std::min;
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicMin(int* p, int v) { int o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
inline int atomicMax(int* p, int v) { int o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v > o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
inline unsigned atomicMin(unsigned* p, unsigned v) { unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v < o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v > o && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
inline long long __double_as_longlong(double d) { long long u; std::memcpy(&u, &d, 8); return u; }
inline double __longlong_as_double(long long u) { double d; std::memcpy(&d, &u, 8); return d; }
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, __ATOMIC_ACQUIRE)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, __ATOMIC_RELEASE)
inline void __syncthreads() { emu::barrier(); }
inline int __syncthreads_and(int pred) { return emu::barrier_and(pred); }
// cross-lane exchange by XOR of the lane index (width 64)
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  static_assert(sizeof(T) == 4, "32-bit values only");
  (void)width;
  unsigned u;
  std::memcpy(&u, &v, 4);
  const emu::Rendezvous r = emu::arrive(emu::OP_DPP, u);
  unsigned long long o;
  if (emu::peer(r, emu::lane_id() ^ mask, &o)) u = (unsigned)o;
  std::memcpy(&v, &u, 4);
  return v;
}
// ds_bpermute_b32: every lane reads the value of the lane whose byte address (lane * 4) it names (0 from a lane that is gone)
inline int emu_ds_bpermute(int addr, int v) {
  const emu::Rendezvous r = emu::arrive(emu::OP_DPP, (unsigned)v);
  unsigned long long o = 0;
  return emu::peer(r, (addr >> 2) & 63, &o) ? (int)(unsigned)o : 0;
}
#define __builtin_amdgcn_ds_bpermute(addr, v) emu_ds_bpermute(addr, v)
// v_perm_b32: bytes of {src0 (4..7), src1 (0..3)} picked by the selector's bytes; 8..11 sign replication, 12 zero, 13+ 0xFF
inline unsigned emu_perm(unsigned s0, unsigned s1, unsigned sel) {
  const unsigned long long in = ((unsigned long long)s0 << 32) | s1;
  unsigned r = 0;
  for (int i = 0; i < 4; ++i) {
    const unsigned k = (sel >> (8 * i)) & 0xFF;
    unsigned b;
    if (k <= 7) b = (unsigned)(in >> (8 * k)) & 0xFF;
    else if (k <= 11) b = ((in >> (16 * (k - 8) + 15)) & 1) ? 0xFF : 0x00;
    else if (k == 12) b = 0x00;
    else b = 0xFF;
    r |= b << (8 * i);
  }
  return r;
}
#define __builtin_amdgcn_perm(a, b, s) emu_perm((unsigned)(a), (unsigned)(b), (unsigned)(s))
// v_sad_u8 / v_dot4_u32_u8 (png.hip: sums over the four bytes of a dword)
inline unsigned emu_sad_u8(unsigned a, unsigned b, unsigned c) {
  for (int i = 0; i < 4; ++i) { const int x = (a >> (8 * i)) & 255, y = (b >> (8 * i)) & 255; c += (unsigned)(x > y ? x - y : y - x); }
  return c;
}
inline unsigned emu_udot4(unsigned a, unsigned b, unsigned c, bool) {
  for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 255u) * ((b >> (8 * i)) & 255u);
  return c;
}
#define __builtin_amdgcn_sad_u8(a, b, c) emu_sad_u8((unsigned)(a), (unsigned)(b), (unsigned)(c))
#define __builtin_amdgcn_udot4(a, b, c, clamp) emu_udot4((unsigned)(a), (unsigned)(b), (unsigned)(c), clamp)
// v_dot2_i32_i16
template <typename V>
inline int emu_sdot2(V a, V b, int c, bool) { return (int)a.x * (int)b.x + (int)a.y * (int)b.y + c; }
#define __builtin_amdgcn_sdot2(a, b, c, clamp) emu_sdot2(a, b, c, clamp)

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
// v_readlane_b32: the value of lane `lane` (which must be active), in every lane
inline int emu_readlane(int x, int lane) {
  const emu::Rendezvous r = emu::arrive(emu::OP_DPP, (unsigned)x);
  unsigned long long v = 0;
  if (!emu::peer(r, lane, &v)) { std::fprintf(stderr, "emu: v_readlane of an inactive lane\n"); std::abort(); }
  return (int)(unsigned)v;
}
#define __builtin_amdgcn_readlane(x, lane) emu_readlane(x, lane)
// v_mov_b32_dpp semantics (the controls the kernels use): quad_perm, row_shl:n, row_shr:n, row_bcast:15, row_newbcast:n
inline int emu_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const emu::Rendezvous r = emu::arrive(emu::OP_DPP, (unsigned)src);
  const int lane = emu::lane_id(), row = lane >> 4, inrow = lane & 15;
  if (!((row_mask >> row) & 1) || !((bank_mask >> (inrow >> 2)) & 1)) return old;
  int from = -1;
  if (ctrl >= 0 && ctrl <= 0xFF) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl >= 0x101 && ctrl <= 0x10F) from = inrow + (ctrl & 15) <= 15 ? lane + (ctrl & 15) : -1;
  else if (ctrl >= 0x111 && ctrl <= 0x11F) from = inrow >= (ctrl & 15) ? lane - (ctrl & 15) : -1;
  else if (ctrl == 0x142) from = row > 0 ? row * 16 - 1 : -1;
  else if (ctrl == 0x143) from = row >= 2 ? 31 : -1;  // row_bcast:31
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
