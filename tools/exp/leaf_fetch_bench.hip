// Micro-benchmark: what does it cost a CU to bring one 288-byte leaf tile per lane into registers, by access pattern?
// 256 workgroups x 1024 threads (16 waves per CU, as the path kernel), every lane picks a pseudo-random tile of an
// L2-resident pool per iteration.  Patterns:
//   0  per lane: 18 x dwordx4 from the lane's own tile (the path kernel's leaf block)          64 lines per instruction
//   1  pairs: lanes 2k, 2k+1 load the two 16-byte halves of a row of ONE tile (2 x 9 loads)     32 lines per instruction
//   2  quads: 4 lanes load 64 contiguous bytes of one tile, tile padded to 320 B (4 x 5 loads)  16 lines per instruction
//   3  all lanes of the wave the same tile (coherent camera rays)
//   4  per lane, 9 x dwordx4 only (half the bytes: is it bytes or instructions?)
//   5  per lane, 18 x dwordx4, tiles 512-byte aligned (each tile in 3 lines exactly -> does alignment matter?)
// hipcc --offload-arch=gfx950 -O3 leaf_fetch_bench.hip -o leaf_fetch_bench && ./leaf_fetch_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

template <int PATTERN>
__global__ __launch_bounds__(1024) void bench(const float4 *pool, int n_tiles, int tile_f4, int iters, float *out) {
  const int lane = threadIdx.x & 63;
  uint32_t rng = pcg(blockIdx.x * 1024u + threadIdx.x + 1u);
  float acc = 0.0f;
  for (int it = 0; it < iters; it++) {
    rng = pcg(rng);
    int g = (int)(rng % (uint32_t)n_tiles);
    if (PATTERN == 3) g = __builtin_amdgcn_readfirstlane(g);
    if (PATTERN == 0 || PATTERN == 3 || PATTERN == 5) {
      const float4 *t = pool + (size_t)g * tile_f4;
      float4 v[18];
#pragma unroll
      for (int i = 0; i < 18; i++) v[i] = t[i];
#pragma unroll
      for (int i = 0; i < 18; i++) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    } else if (PATTERN == 4) {
      const float4 *t = pool + (size_t)g * tile_f4;
      float4 v[9];
#pragma unroll
      for (int i = 0; i < 9; i++) v[i] = t[2 * i];
#pragma unroll
      for (int i = 0; i < 9; i++) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    } else if (PATTERN == 1) {
#pragma unroll
      for (int m = 0; m < 2; m++) {
        const int gm = __shfl(g, (lane & ~1) + m, 64);
        const float4 *t = pool + (size_t)gm * tile_f4 + (lane & 1);
        float4 v[9];
#pragma unroll
        for (int i = 0; i < 9; i++) v[i] = t[2 * i];
#pragma unroll
        for (int i = 0; i < 9; i++) acc += v[i].x + v[i].y + v[i].z + v[i].w;
      }
    } else if (PATTERN == 2) {
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const int gm = __shfl(g, (lane & ~3) + m, 64);
        const float4 *t = pool + (size_t)gm * tile_f4 + (lane & 3);
        float4 v[5];
#pragma unroll
        for (int i = 0; i < 5; i++) v[i] = t[4 * i];
#pragma unroll
        for (int i = 0; i < 5; i++) acc += v[i].x + v[i].y + v[i].z + v[i].w;
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int PATTERN>
static void run(const char *name, const float4 *pool, int n_tiles, int tile_f4, float *out) {
  const int iters = 2000;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(bench<PATTERN>, dim3(256), dim3(1024), 0, 0, pool, n_tiles, tile_f4, 200, out);
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(bench<PATTERN>, dim3(256), dim3(1024), 0, 0, pool, n_tiles, tile_f4, iters, out);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  // per CU: 16 waves x iters blocks
  double ns_per_block_cu = ms * 1e6 / ((double)iters * 16.0);
  printf("%-44s %8.3f ms   %7.1f ns per wave-block per CU  (%6.0f cycles @2.4GHz)   %6.1f G tiles/s chip\n", name, ms, ns_per_block_cu,
         ns_per_block_cu * 2.4, 256.0 * 1024 * iters / (ms * 1e6));
}

int main() {
  const int n_tiles = 4096;                      // helmet: 4096 leaf groups x 288 B = 1.2 MB, L2 resident
  std::vector<float> h((size_t)n_tiles * 128, 1.0f);
  float4 *pool; float *out;
  CHECK(hipMalloc(&pool, h.size() * 4));
  CHECK(hipMemcpy(pool, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&out, 64));
  run<0>("0 per lane, 18 x 16 B, 288-B tiles", pool, n_tiles, 18, out);
  run<1>("1 pairs (32 B contiguous), 2 x 9 loads", pool, n_tiles, 18, out);
  run<2>("2 quads (64 B contiguous), 4 x 5 loads, 320-B", pool, n_tiles, 20, out);
  run<3>("3 coherent: one tile per wave", pool, n_tiles, 18, out);
  run<4>("4 per lane, 9 x 16 B (half the bytes)", pool, n_tiles, 18, out);
  run<5>("5 per lane, 18 x 16 B, 512-B aligned tiles", pool, n_tiles, 32, out);
  run<0>("0 again", pool, n_tiles, 18, out);
  return 0;
}
