// Throw-away: is  y = v_rcp_f32(x); y' = fma(fma(-x, y, 1), y, y)  the correctly rounded 1/x?  Exhaustive over all 2^32
// bit patterns against the compiler's IEEE division sequence (v_div_scale / v_div_fmas / v_div_fixup).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o rcp_test rcp_test.hip && ./rcp_test
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float rcp1(float x) {
  float y = __builtin_amdgcn_rcpf(x);
  float r = __builtin_fmaf(-x, y, 1.0f);
  return __builtin_fmaf(r, y, y);
}
__device__ __forceinline__ float rcp2(float x) {
  float y = rcp1(x);
  float r = __builtin_fmaf(-x, y, 1.0f);
  return __builtin_fmaf(r, y, y);
}

// categories: 0 zero/inf/nan, 1 denormal x, 2 normal with |x| < 2^-125, 3 normal 2^-125 <= |x| <= 2^125, 4 |x| > 2^125
__device__ int category(uint32_t b) {
  uint32_t e = (b >> 23) & 0xFF, m = b & 0x7FFFFF;
  if (e == 0xFF) return 0;
  if (e == 0) return m ? 1 : 0;
  if (e < 2) return 2;            // 2^-126 <= |x| < 2^-125
  if (e > 252) return 4;          // |x| >= 2^126
  return 3;
}

__global__ void test(unsigned long long *bad1, unsigned long long *bad2, uint32_t *ex1, uint32_t *ex2) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (uint32_t k = 0; k < 256; k++) {
    uint32_t b = tid * 256u + k;
    float x = __uint_as_float(b);
    float want = 1.0f / x;
    float g1 = rcp1(x), g2 = rcp2(x);
    uint32_t w = __float_as_uint(want), a1 = __float_as_uint(g1), a2 = __float_as_uint(g2);
    bool nanw = (w & 0x7FFFFFFF) > 0x7F800000;
    bool ok1 = nanw ? ((a1 & 0x7FFFFFFF) > 0x7F800000) : (a1 == w);
    bool ok2 = nanw ? ((a2 & 0x7FFFFFFF) > 0x7F800000) : (a2 == w);
    int c = category(b);
    if (!ok1) { unsigned long long n = atomicAdd(&bad1[c], 1ull); if (n < 8) ex1[c * 8 + n] = b; }
    if (!ok2) { unsigned long long n = atomicAdd(&bad2[c], 1ull); if (n < 8) ex2[c * 8 + n] = b; }
  }
}

int main() {
  unsigned long long *bad1, *bad2; uint32_t *ex1, *ex2;
  hipMalloc(&bad1, 5 * 8); hipMalloc(&bad2, 5 * 8); hipMalloc(&ex1, 40 * 4); hipMalloc(&ex2, 40 * 4);
  hipMemset(bad1, 0, 40); hipMemset(bad2, 0, 40); hipMemset(ex1, 0, 160); hipMemset(ex2, 0, 160);
  test<<<65536, 256>>>(bad1, bad2, ex1, ex2);
  unsigned long long h1[5], h2[5]; uint32_t e1[40], e2[40];
  hipMemcpy(h1, bad1, 40, hipMemcpyDeviceToHost); hipMemcpy(h2, bad2, 40, hipMemcpyDeviceToHost);
  hipMemcpy(e1, ex1, 160, hipMemcpyDeviceToHost); hipMemcpy(e2, ex2, 160, hipMemcpyDeviceToHost);
  const char *names[5] = {"zero/inf/nan", "denormal", "2^-126 <= |x| < 2^-125", "2^-125 <= |x| < 2^126", "|x| >= 2^126"};
  for (int c = 0; c < 5; c++) {
    printf("%-26s one step: %llu mismatches", names[c], h1[c]);
    for (int i = 0; i < 4 && i < (int)h1[c]; i++) printf(" %08x", e1[c * 8 + i]);
    printf("   two steps: %llu", h2[c]);
    for (int i = 0; i < 4 && i < (int)h2[c]; i++) printf(" %08x", e2[c * 8 + i]);
    printf("\n");
  }
  return 0;
}
