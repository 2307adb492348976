// Throw-away pricing kernel (not part of librt_hip.so): the 8-lanes-per-ray layout SURVEY.md section 2a names,
// against the one-ray-per-lane layout the render kernel uses, on the operation that dominates traversal: entering a
// BVH node = 8 slab tests (raytracer.c:190-230) + the near-first order of the candidates (raytracer.c:459-468).
//
//   A  one ray per lane: every lane tests the 8 children of its node and rank-sorts them in registers
//      (node_enter<true, NODE_LDS> of rt_kernels.hip, verbatim arithmetic; nodes from the workgroup's LDS copy).
//   B  eight lanes per ray: lane j of a group tests child j only; the order comes from cross-lane compares
//      (ds_bpermute broadcasts inside the group of 8) and the 32-bit order word from an OR-reduction over the group.
//
// Both produce the same order word for every (ray, node) pair (checked).  Printed: ns per (ray, node) pair at full
// lane utilisation -- the best case for B, whose 8 ray-groups per wave diverge 8 lanes at a time in a real traversal.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o lanes8 tools/exp/lanes8_node_test.hip && ./lanes8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rt_math.h"

struct Ray3 { rt_v3 o, d; float inv_x, inv_y, inv_z; };

__device__ __forceinline__ float fmin_hw(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_hw(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ int as_i(float f) { return __float_as_int(f); }

__device__ __forceinline__ float slab(const Ray3 &r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float t_max) {
  float t0x = (mnx - r.o.x) * r.inv_x, t1x = (mxx - r.o.x) * r.inv_x;
  float t0y = (mny - r.o.y) * r.inv_y, t1y = (mxy - r.o.y) * r.inv_y;
  float t0z = (mnz - r.o.z) * r.inv_z, t1z = (mxz - r.o.z) * r.inv_z;
  float sx = fmin_hw(t0x, t1x), sy = fmin_hw(t0y, t1y), sz = fmin_hw(t0z, t1z);
  float bx = fmax_hw(t0x, t1x), by = fmax_hw(t0y, t1y), bz = fmax_hw(t0z, t1z);
  float t_minv = fmax_hw(RT_EPS, fmax_hw(sx, fmax_hw(sy, sz)));
  float t_maxv = fmin_hw(t_max, fmin_hw(bx, fmin_hw(by, bz)));
  return (t_minv < t_maxv) ? t_minv : RT_INF;
}

__device__ __forceinline__ void load_ray(const float *rays, int i, Ray3 &r) {
  r.o = rt_v3_make(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]);
  r.d = rt_v3_make(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
  r.inv_x = 1.0f / r.d.x; r.inv_y = 1.0f / r.d.y; r.inv_z = 1.0f / r.d.z;
}

#define LDS_NODE_F4 13       // 12 float4 of data + 1 of padding per node, as in rt_kernels.hip

// the workgroup's LDS copy of the whole node table (the render kernel keeps the top of the BVH there)
__device__ __forceinline__ void stage_nodes(float4 *smem, const float *nodes, int n_nodes) {
  const float4 *g = reinterpret_cast<const float4 *>(nodes);
  for (int i = threadIdx.x; i < n_nodes * 12; i += blockDim.x) {
    int nd = i / 12, q = i - nd * 12;
    smem[nd * LDS_NODE_F4 + q] = g[i];
  }
  __syncthreads();
}

// A: one ray per lane
__global__ __launch_bounds__(1024) void node_test_lane(const float *nodes, int n_nodes, const float *rays, int n_rays, int reps,
                                                        uint32_t *out) {
  extern __shared__ float4 smem[];
  stage_nodes(smem, nodes, n_nodes);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rays; i += gridDim.x * blockDim.x) {
  Ray3 r;
  load_ray(rays, i, r);
  uint32_t acc = 0;
  int node = i % n_nodes;
  for (int it = 0; it < reps; it++) {
    const float4 *nb = smem + node * LDS_NODE_F4;
    int d[8];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      float4 mnx = nb[0 + h], mny = nb[2 + h], mnz = nb[4 + h], mxx = nb[6 + h], mxy = nb[8 + h], mxz = nb[10 + h];
      d[h * 4 + 0] = as_i(slab(r, mnx.x, mny.x, mnz.x, mxx.x, mxy.x, mxz.x, RT_INF));
      d[h * 4 + 1] = as_i(slab(r, mnx.y, mny.y, mnz.y, mxx.y, mxy.y, mxz.y, RT_INF));
      d[h * 4 + 2] = as_i(slab(r, mnx.z, mny.z, mnz.z, mxx.z, mxy.z, mxz.z, RT_INF));
      d[h * 4 + 3] = as_i(slab(r, mnx.w, mny.w, mnz.w, mxx.w, mxy.w, mxz.w, RT_INF));
    }
    int rank[8];
#pragma unroll
    for (int k = 0; k < 8; k++) rank[k] = k;
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
      for (int k = j + 1; k < 8; k++) {
        int kb = (int)((uint32_t)(d[k] - d[j]) >> 31);
        rank[j] += kb;
        rank[k] -= kb;
      }
    uint32_t w = 0, n_inf = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      w |= (uint32_t)j << (3 * rank[j]);
      n_inf += ((uint32_t)d[j] + 0x00800000u) >> 31;
    }
    w |= (8u - n_inf) << 24;
    acc = acc * 31u + w;
    node = (int)((uint32_t)(node * 7 + (int)(w & 7u) + 1) % (uint32_t)n_nodes);      // data-dependent next node, as in a traversal
  }
  out[i] = acc;
  }
}

// B: eight lanes per ray
__global__ __launch_bounds__(1024) void node_test_group(const float *nodes, int n_nodes, const float *rays, int n_rays, int reps,
                                                         uint32_t *out) {
  extern __shared__ float4 smem[];
  stage_nodes(smem, nodes, n_nodes);
  const float *lds = reinterpret_cast<const float *>(smem);
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; (t >> 3) < n_rays; t += gridDim.x * blockDim.x) {
  int i = t >> 3, j = t & 7;
  const int lane = threadIdx.x & 63, base = lane & ~7;
  Ray3 r;
  load_ray(rays, i, r);
  uint32_t acc = 0;
  int node = i % n_nodes;
  for (int it = 0; it < reps; it++) {
    const float *nb = lds + node * (LDS_NODE_F4 * 4) + j;   // child j: the six rows are 8 floats apart (one 32-byte row per group)
    int dj = as_i(slab(r, nb[0], nb[8], nb[16], nb[24], nb[32], nb[40], RT_INF));
    int rank = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {                            // rank of child j = children before it in (distance, index) order
      int dk = __shfl(dj, base + k, 64);
      rank += (dk < dj || (dk == dj && k < j)) ? 1 : 0;
    }
    uint32_t w = (uint32_t)j << (3 * rank);
    w |= ((uint32_t)dj + 0x00800000u) >> 31 ? 0u : (1u << 24);     // one candidate counted per finite distance
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {                  // combine the group's words: OR the slots, ADD the counts
      uint32_t o = (uint32_t)__shfl_xor((int)w, off, 64);
      w = ((w | o) & 0x00FFFFFFu) | (((w >> 24) + (o >> 24)) << 24);
    }
    acc = acc * 31u + w;
    node = (int)((uint32_t)(node * 7 + (int)(w & 7u) + 1) % (uint32_t)n_nodes);
  }
  if (j == 0) out[i] = acc;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int n_rays = 1 << 22, n_nodes = 585, reps = 64;
  std::vector<float> nodes((size_t)n_nodes * 48), rays((size_t)n_rays * 6);
  srand(7);
  auto rnd = [] { return (float)rand() / (float)RAND_MAX; };
  for (int n = 0; n < n_nodes; n++)
    for (int k = 0; k < 8; k++) {
      float c[3] = {rnd() * 4 - 2, rnd() * 4 - 2, rnd() * 4 - 2}, h = 0.2f + rnd() * ((k & 1) ? 1.5f : 0.4f);
      for (int a = 0; a < 3; a++) { nodes[(size_t)n * 48 + a * 8 + k] = c[a] - h; nodes[(size_t)n * 48 + (3 + a) * 8 + k] = c[a] + h; }
      if (k == 7 && (n & 3) == 0) for (int a = 0; a < 6; a++) nodes[(size_t)n * 48 + a * 8 + k] = 0.0f;      // unpopulated child
    }
  for (int i = 0; i < n_rays; i++) {
    float o[3] = {rnd() * 8 - 4, rnd() * 8 - 4, 6.0f}, d[3] = {rnd() - 0.5f - o[0] * 0.1f, rnd() - 0.5f - o[1] * 0.1f, -1.0f};
    float l = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int a = 0; a < 3; a++) { rays[(size_t)i * 6 + a] = o[a]; rays[(size_t)i * 6 + 3 + a] = d[a] * l; }
  }
  float *d_nodes, *d_rays;
  uint32_t *d_a, *d_b;
  CK(hipMalloc(&d_nodes, nodes.size() * 4)); CK(hipMalloc(&d_rays, rays.size() * 4));
  CK(hipMalloc(&d_a, (size_t)n_rays * 4)); CK(hipMalloc(&d_b, (size_t)n_rays * 4));
  CK(hipMemcpy(d_nodes, nodes.data(), nodes.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rays, rays.data(), rays.size() * 4, hipMemcpyHostToDevice));
  const int smem_bytes = n_nodes * LDS_NODE_F4 * 16;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&node_test_lane), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&node_test_group), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_a = 0, ms_b = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(node_test_lane, dim3(256), dim3(1024), smem_bytes, 0, d_nodes, n_nodes, d_rays, n_rays, reps, d_a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_a, e0, e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(node_test_group, dim3(256), dim3(1024), smem_bytes, 0, d_nodes, n_nodes, d_rays, n_rays, reps, d_b);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_b, e0, e1));
  }
  std::vector<uint32_t> a(n_rays), b(n_rays);
  CK(hipMemcpy(a.data(), d_a, (size_t)n_rays * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_b, (size_t)n_rays * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int i = 0; i < n_rays; i++) bad += a[i] != b[i];
  double pairs = (double)n_rays * reps;
  printf("node entries: %d rays x %d nodes each; order words differing between the layouts: %zu\n", n_rays, reps, bad);
  printf("A one ray per lane   : %8.3f ms  %6.3f ns per (ray, node)  %7.1f G node entries / s\n", ms_a, ms_a * 1e6 / pairs, pairs / ms_a / 1e6);
  printf("B eight lanes per ray: %8.3f ms  %6.3f ns per (ray, node)  %7.1f G node entries / s   (B / A = %.2f)\n", ms_b, ms_b * 1e6 / pairs,
         pairs / ms_b / 1e6, ms_b / ms_a);
  return bad ? 2 : 0;
}
