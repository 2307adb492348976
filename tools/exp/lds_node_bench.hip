// Micro-benchmark: LDS cost of the three node-read forms of the path kernel for 64 lanes on RANDOM nodes of a 585-node tree,
// plane-major (the kernel's image: 6 rows x 8 floats per node, 208-byte stride) against child-major (8 children x 6 floats,
// 24 bytes per child contiguous, 208-byte stride).  16 waves per CU as in the path kernel.
//   A  full block, plane-major: 12 x ds_read_b128 per lane (node_enter<NODE_LDS_ORDERED>)
//   B  one child, plane-major: 6 x ds_read_b32 at 32-byte stride (node_enter_few, per surviving child)
//   C  one child, near planes only, plane-major: 3 x ds_read_b32 (pop re-test)
//   D  full block, child-major: 8 x (ds_read_b128 + ds_read_b64)
//   E  one child, child-major: ds_read_b128 + ds_read_b64
//   F  one child, near planes, child-major: 3 x ds_read_b32 (not contiguous: the near plane depends on the ray's sign)
//   G  A with every lane on the SAME node (coherent camera rays: broadcast)
// hipcc --offload-arch=gfx950 -O3 lds_node_bench.hip -o lds_node_bench && ./lds_node_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define N_NODES 585
#define STRIDE_F 52          // 208 bytes

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}

template <int FORM>
__global__ __launch_bounds__(1024) void bench(int iters, float *out) {
  extern __shared__ float smem[];
  for (int i = threadIdx.x; i < N_NODES * STRIDE_F; i += 1024) smem[i] = 1.0f + (float)(i & 7);
  __syncthreads();
  uint32_t rng = pcg(blockIdx.x * 1024u + threadIdx.x + 1u);
  float acc = 0.0f;
  for (int it = 0; it < iters; it++) {
    rng = pcg(rng);
    int node = (int)(rng % N_NODES);
    if (FORM == 6) node = __builtin_amdgcn_readfirstlane(node);
    const int child = (int)((rng >> 12) & 7u);
    const int sx = (int)((rng >> 16) & 1u), sy = (int)((rng >> 17) & 1u), sz = (int)((rng >> 18) & 1u);
    const float *nb = smem + node * STRIDE_F;
    if (FORM == 0 || FORM == 6) {
      const float4 *q = reinterpret_cast<const float4 *>(nb);
#pragma unroll
      for (int k = 0; k < 12; k++) { float4 v = q[k]; acc += v.x + v.y + v.z + v.w; }
    } else if (FORM == 1) {
#pragma unroll
      for (int k = 0; k < 6; k++) acc += nb[k * 8 + child];
    } else if (FORM == 2) {
      acc += nb[(sx ? 24 : 0) + child] + nb[8 + (sy ? 24 : 0) + child] + nb[16 + (sz ? 24 : 0) + child];
    } else if (FORM == 3) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        float4 a = *reinterpret_cast<const float4 *>(nb + k * 6);          // (24-byte children: 8-byte aligned only -> two b64 + b64 in practice)
        float2 b = *reinterpret_cast<const float2 *>(nb + k * 6 + 4);
        acc += a.x + a.y + a.z + a.w + b.x + b.y;
      }
    } else if (FORM == 4) {
      float2 a = *reinterpret_cast<const float2 *>(nb + child * 6), b = *reinterpret_cast<const float2 *>(nb + child * 6 + 2),
             c = *reinterpret_cast<const float2 *>(nb + child * 6 + 4);
      acc += a.x + a.y + b.x + b.y + c.x + c.y;
    } else if (FORM == 5) {
      acc += nb[child * 6 + (sx ? 3 : 0)] + nb[child * 6 + 1 + (sy ? 3 : 0)] + nb[child * 6 + 2 + (sz ? 3 : 0)];
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int FORM>
static void run(const char *name, float *out) {
  const int iters = 20000;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&bench<FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int smem = N_NODES * STRIDE_F * 4;
  hipLaunchKernelGGL(bench<FORM>, dim3(256), dim3(1024), smem, 0, 100, out);
  CHECK(hipEventRecord(a));
  hipLaunchKernelGGL(bench<FORM>, dim3(256), dim3(1024), smem, 0, iters, out);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  double ns = ms * 1e6 / ((double)iters * 16.0);
  printf("%-58s %8.3f ms  %7.1f ns = %5.0f cycles per wave-read per CU\n", name, ms, ns, ns * 2.4);
}

int main() {
  float *out;
  CHECK(hipMalloc(&out, 64));
  run<0>("A full block, plane-major, 12 x b128, random nodes", out);
  run<1>("B one child, plane-major, 6 x b32", out);
  run<2>("C one child near planes, plane-major, 3 x b32", out);
  run<3>("D full block, child-major, 8 x (b128 + b64)", out);
  run<4>("E one child, child-major, 3 x b64", out);
  run<5>("F one child near planes, child-major, 3 x b32", out);
  run<6>("G full block, plane-major, all lanes one node", out);
  return 0;
}
