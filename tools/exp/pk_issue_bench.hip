// pk_issue_bench.hip -- what a stream of v_fma_f32 and of v_pk_fma_f32 costs on gfx950 at 1, 2 and 4 waves per SIMD
// (independent chains: 8 accumulators per lane for the plain form, 4 pairs for the packed one -- the same FMAs per lane).
//   hipcc --offload-arch=gfx950 -O3 pk_issue_bench.hip -o pk_issue_bench && ./pk_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
__global__ void k_fma(float *out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITERS; i++) {
    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_pk(float *out, float a, float b) {
  f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, aa = {a, a}, bb = {b, b};
  for (int i = 0; i < ITERS; i++) {
    asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                 "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(aa), "v"(bb));
  }
  f2 s = x0 + x1 + x2 + x3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
// mixed: what a leaf block looks like -- FMAs with a compare + select every few instructions
__global__ void k_mix(float *out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < ITERS; i++) {
    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                 "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_sub_f32 %6, %6, %9\n v_sub_f32 %7, %7, %9\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_mix_pk(float *out, float a, float b) {
  f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, aa = {a, a}, bb = {b, b};
  for (int i = 0; i < ITERS; i++) {
    asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_mul_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %5 neg_lo:[0,1] neg_hi:[0,1]\n"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(aa), "v"(bb));
  }
  f2 s = x0 + x1 + x2 + x3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <class K> static void run(const char *name, K kern, int waves_per_simd, double lane_fmas_per_iter, int insts_per_iter) {
  float *out; hipMalloc(&out, 256 * 4 * 64 * 8 * 4 * 4);
  int block = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;      // one workgroup per CU: 4 SIMDs x waves
  int blocks = 256 * (64 * 4 * waves_per_simd / block);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(block), 0, 0, out, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0); 
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(block), 0, 0, out, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double waves = (double)blocks * block / 64;
  double inst = waves * ITERS * insts_per_iter;                     // wave-instructions
  double cyc_per_inst_per_simd = ms * 1e-3 * 2.4e9 / (inst / 1024.0);
  double tflops = waves * 64 * ITERS * lane_fmas_per_iter * 2 / (ms * 1e-3) / 1e12;
  printf("%-10s %d waves/SIMD: %7.3f ms, %5.2f SIMD cycles per wave-instruction (at 2.4 GHz), %6.1f Tflop/s-equivalent\n", name, waves_per_simd, ms, cyc_per_inst_per_simd, tflops);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run("v_fma", k_fma, w, 8, 8);
    run("v_pk_fma", k_pk, w, 16, 8);
    run("mix", k_mix, w, 8, 8);
    run("mix_pk", k_mix_pk, w, 8, 4);
  }
  return 0;
}
