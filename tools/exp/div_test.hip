// Throw-away: a / 1.055f by  q0 = a * c; r = fma(-1.055f, q0, a); q = fma(r, c, q0)  with c = RN(1 / 1.055f): where does it
// equal the IEEE quotient?  Exhaustive over all 2^32 a; mismatches per exponent field of a.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void test(unsigned long long *bad_by_exp) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const float c = 1.0f / 1.055f;
  for (uint32_t k = 0; k < 256; k++) {
    uint32_t b = tid * 256u + k;
    float a = __uint_as_float(b);
    float want = a / 1.055f;
    float q0 = a * c;
    float r = __builtin_fmaf(-1.055f, q0, a);
    float q = __builtin_fmaf(r, c, q0);
    uint32_t w = __float_as_uint(want), g = __float_as_uint(q);
    bool nanw = (w & 0x7FFFFFFF) > 0x7F800000, nang = (g & 0x7FFFFFFF) > 0x7F800000;
    bool same = nanw ? nang : (g == w);
    if (!same) atomicAdd(&bad_by_exp[(b >> 23) & 0xFF], 1ull);
  }
}
int main() {
  unsigned long long *d, h[256];
  (void)hipMalloc(&d, 256 * 8); (void)hipMemset(d, 0, 256 * 8);
  test<<<65536, 256>>>(d);
  (void)hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
  unsigned long long tot = 0;
  for (int e = 0; e < 256; e++) { if (h[e]) printf("exponent field %3d: %llu mismatches\n", e, h[e]); tot += h[e]; }
  printf("total %llu\n", tot);
  return 0;
}
