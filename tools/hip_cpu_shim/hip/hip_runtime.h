// DEVELOPER TOOL — not part of the product. A minimal stand-in for <hip/hip_runtime.h> that lets the ISP's HIP sources
// (surround360_amd/csrc/isp_kernels.hip, isp.cpp) be compiled with g++ and executed on the CPU, one std::thread per GPU
// thread of a block (blocks one after another, __syncthreads = a barrier). It exists to check kernel indexing and
// arithmetic where no GPU is attached (tools/isp_emulate.cpp, tests/test_cpu_isp.py::test_kernels_emulated_on_cpu);
// float arithmetic is IEEE on both sides and the sources are built with -ffp-contract=off, so results are comparable
// bit for bit. It is NOT a fallback: libs360 never links it.
#pragma once
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint3_ { unsigned x, y, z; };
extern thread_local uint3_ threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum { hipStreamNonBlocking = 1 };
inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(1, n ? n : 1); return hipSuccess; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
#define hipHostMallocDefault 0u
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = std::calloc(1, n ? n : 1); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
struct hipEvent_st; typedef hipEvent_st* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t*) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* t, hipEvent_t, hipEvent_t) { *t = 0; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }

inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline long long __double_as_longlong(double d) { long long u; std::memcpy(&u, &d, 8); return u; }
inline double __longlong_as_double(long long u) { double d; std::memcpy(&d, &u, 8); return d; }
using std::max;
using std::min;

// ---- block execution: every thread of a block is a std::thread; __syncthreads is a reusable barrier ----
struct EmuBarrier {
  std::mutex m;
  std::condition_variable cv;
  unsigned count = 0, waiting = 0, gen = 0;
  void wait() {
    std::unique_lock<std::mutex> l(m);
    const unsigned g = gen;
    if (++waiting == count) { waiting = 0; ++gen; cv.notify_all(); }
    else cv.wait(l, [&] { return g != gen; });
  }
  void leave() {  // a thread that returns early no longer takes part
    std::unique_lock<std::mutex> l(m);
    --count;
    if (count && waiting == count) { waiting = 0; ++gen; cv.notify_all(); }
  }
};
extern EmuBarrier g_emu_barrier;
inline void __syncthreads() { g_emu_barrier.wait(); }

template <typename K, typename... A>
void emu_launch(K kernel, dim3 grid, dim3 block, A... args) {
  const unsigned nt = block.x * block.y * block.z;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        g_emu_barrier.count = nt;
        g_emu_barrier.waiting = 0;
        std::vector<std::thread> th;
        th.reserve(nt);
        for (unsigned t = 0; t < nt; ++t)
          th.emplace_back([=] {
            threadIdx = uint3_{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
            blockIdx = uint3_{bx, by, bz};
            blockDim = block;
            gridDim = grid;
            kernel(args...);
            g_emu_barrier.leave();
          });
        for (auto& x : th) x.join();
      }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) emu_launch(kernel, dim3(grid), dim3(block), __VA_ARGS__)
