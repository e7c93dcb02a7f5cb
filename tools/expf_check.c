/* DEVELOPER TOOL. Exhaustive check of the expf sequence the ISP's noise coring uses on the GPU (expf_glibc_neg in
 * surround360_amd/csrc/isp_kernels.hip) against the host's libm: every float <= 0 (2 139 095 041 values).
 *   gcc -O2 -ffp-contract=off -o expf_check expf_check.c -lm && ./expf_check      (about 25 s)
 * glibc 2.35 on x86-64 with FMA: 0 mismatches. Without the fma in `r` exactly one input differs (0xc27c65d9). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
static uint64_t T[32];
static float expf_seq(float x) {
  const double InvLn2N = 0x1.71547652b82fep+0 * 32, SHIFT = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
  const uint32_t abstop = (asuint(x) >> 20) & 0x7ff;
  if (abstop >= (asuint(88.0f) >> 20)) {
    if (asuint(x) == asuint(-INFINITY)) return 0.0f;
    if (abstop >= (asuint(INFINITY) >> 20)) return x + x;
    if (x < -0x1.9fe368p6f) return 0.0f;
    if (x < -0x1.9d1d9ep6f) return 0x1p-149f;
  }
  const double xd = x;
  double kd = fma(InvLn2N, xd, SHIFT);
  const uint64_t ki = asuint64(kd);
  kd -= SHIFT;
  const double r = fma(InvLn2N, xd, -kd);
  const double s = asdouble(T[ki % 32] + (ki << 47));
  const double z = fma(C0, r, C1), r2 = r * r;
  double y = fma(C2, r, 1.0);
  y = fma(z, r2, y);
  return (float)(y * s);
}
int main(void) {
  for (int i = 0; i < 32; i++) T[i] = asuint64(exp2((double)i / 32)) - ((uint64_t)i << 47);
  unsigned long long bad = 0, n = 0;
  uint32_t first = 0;
  for (uint64_t u = 0x80000000ull; u <= 0xff800000ull; ++u, ++n) {
    volatile float x = asfloat((uint32_t)u);
    if (asuint(expf_seq(x)) != asuint(expf(x))) { if (!bad) first = (uint32_t)u; bad++; }
  }
  printf("%llu floats <= 0 checked, %llu mismatches (first %08x)\n", n, bad, first);
  return bad != 0;
}
