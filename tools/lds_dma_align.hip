// Developer probe: does global_load_lds_dwordx4 (LDS-DMA, 16 bytes per lane) accept source addresses that are only 8-byte
// aligned? (float2 texel pairs at odd columns.) Prints OK / MISMATCH per source misalignment.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* src, int mis, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 + 64];
  const int lane = threadIdx.x;
  const char* g = src + mis + lane * 24;  // lanes 24 bytes apart: a gather, not a contiguous run
  unsigned keep;
  const unsigned dst = (unsigned)(size_t)lds;  // LDS byte address of the buffer (wave-uniform)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}
int main() {
  const int n = 64 * 24 + 64;
  std::vector<float> h(n / 4 + 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  char* d; float* o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 64 * 4 * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int mis : {0, 8, 4, 12}) {
    hipMemset(o, 0, 64 * 16);
    k<<<1, 64>>>(d, mis, o);
    if (hipDeviceSynchronize() != hipSuccess) { printf("mis=%d: launch failed\n", mis); return 1; }
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) bad += r[l * 4 + i] != h[(mis + l * 24) / 4 + i];
    printf("global_load_lds_dwordx4, source misaligned by %2d bytes: %s (%d of 256 dwords differ; lane 1 got %g %g %g %g, expected %g..)\n",
           mis, bad ? "MISMATCH" : "OK", bad, r[4], r[5], r[6], r[7], h[(mis + 24) / 4]);
  }
  return 0;
}
