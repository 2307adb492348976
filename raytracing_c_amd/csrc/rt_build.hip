// rt_build.hip -- scene_init() on the GPU ("CPU first, then GPU").
//
// Produces, byte for byte, the Scene that scene_init() of rt_scene_build.c produces (reference scene.c:78-242,311-426
// with this build's documented choices: stable sort, early-leaf chain): the same triangles in the same slots, the same
// child boxes.  tests/test_gpu_builder.py compares the two on every asset and on random soups.
//
// The reference's split looks sequential -- per node up to seven binary cuts, each of them three stable sorts of the
// slice (x, y, z keys, every sort starting from the previous order), a surface-area comparison of the two halves
// and a fourth sort by the winning axis -- but WHERE the cuts fall depends on triangle COUNTS only
// (bvh_partition_triangles, scene.c:233-242): the whole tree of slices, their cut positions and the child slot each
// finished slice lands in are a pure function of n.  The host walks that function (plan_*), and the GPU executes it
// level by level, one "generation" of cuts at a time for ALL nodes of the level at once:
//
//   per generation:  for axis in x, y, z:  gather keys -> rocPRIM segmented stable radix sort -> per-slice bounds of the
//                    two halves -> surface areas;  pick the axis (last one among equals, scene.c:352);  sort once more
//                    by the picked axis (a no-op where that is z)
//   per level:       bounds of every finished slice -> the child box of its node (BVH_Node rows), unpopulated children
//                    stay all-zero
//   last row:        triangles_insert(): SoA coordinates + face normal / tangent frame per triangle (rt_math.h, the same
//                    fp32 operations as the CPU build)
//
// Sort keys are the reference's (x0 + x1 + x2 per axis, scene.c:206-221); -0 is folded onto +0 so that the radix order
// of the float bit patterns equals the `<` order the CPU's merge sort uses.  min / max reductions are exact, so any
// reduction order gives the CPU's boxes; the surface-area expression keeps the CPU's operation order.
//
// Built with the same flags as the kernels (-ffp-contract=off).  Not a performance target of the frame (the build is
// outside the timed region, driver.c:774-777): helmet 15 452 triangles ~ 200 small launches.

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "../../include/rt_scene.h"
#include "../../include/rt_math.h"

namespace {

struct Seg {            // a slice that is cut in this generation
  int begin, len, split;
};
struct Fin {            // a finished slice: child `slot` of node `node` (index in the implicit tree)
  int begin, len, node, slot;
};
struct NodeSeg {        // the triangles of one node of the current level
  int begin, len, node;
};

// scene.c:233-242
int partition_triangles(int n_triangles, int per_child) {
  int n = 0, left = n_triangles;
  while (n < n_triangles / 2 && left > per_child) { n += per_child; left -= per_child; }
  return n;
}

int n_leaf_nodes(int depth) { int n = 1; for (int i = 0; i < depth; i++) n *= 8; return n; }

// The slice tree of ONE node (rt_scene_build.c bvh_build, scene.c:333-380) on counts only: which slices are cut in
// which generation, and the child slot of every finished slice (the LIFO order of the reference's stack).
void plan_node(const NodeSeg &nd, int depth, std::vector<std::vector<Seg>> &gens, std::vector<Fin> &fins) {
  if (nd.len == 0) return;
  if (nd.len <= 8) {                      // early-leaf chain: everything goes down through child 0
    fins.push_back({nd.begin, nd.len, nd.node, 0});
    return;
  }
  const int per_child = n_leaf_nodes(depth);
  struct Item { int begin, len, gen; };
  std::vector<Item> stack;
  stack.push_back({nd.begin, nd.len, 0});
  int n_finished = 0;
  while (!stack.empty()) {
    Item it = stack.back();
    stack.pop_back();
    int split = partition_triangles(it.len, per_child);
    if ((int)gens.size() <= it.gen) gens.resize((size_t)it.gen + 1);
    gens[(size_t)it.gen].push_back({it.begin, it.len, split});
    Item left = {it.begin, split, it.gen + 1}, right = {it.begin + split, it.len - split, it.gen + 1};
    if (left.len > per_child) stack.push_back(left);
    else if (left.len) fins.push_back({left.begin, left.len, nd.node, n_finished++});
    if (right.len > per_child) stack.push_back(right);
    else if (right.len) fins.push_back({right.begin, right.len, nd.node, n_finished++});
  }
}

// ---------------------------------------------------------------------------------------------------------------
// kernels

// per input triangle: the three sort keys, its EPSILON-padded bounds, identity permutation
__global__ void prep_kernel(int n, const Triangle *tris, float *keys, float *lo, float *hi, uint32_t *idx) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const Triangle &T = tris[t];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    float p0 = T.positions[0].data[a], p1 = T.positions[1].data[a], p2 = T.positions[2].data[a];
    keys[(size_t)a * n + t] = (p0 + p1 + p2) + 0.0f;          // scene.c:206-221; -0 -> +0 (see header)
    float mn = p1 < p2 ? p1 : p2, mx = p1 > p2 ? p1 : p2;     // min3f / max3f of rt_scene_build.c
    mn = p0 < mn ? p0 : mn;
    mx = p0 > mx ? p0 : mx;
    lo[(size_t)a * n + t] = mn - RT_EPSILON;
    hi[(size_t)a * n + t] = mx + RT_EPSILON;
  }
  idx[t] = (uint32_t)t;
}

// one workgroup per slice: k[i] = key of axis (fixed, or axis_of[slice]) of the triangle at position i
__global__ void gather_keys_kernel(int n, const Seg *segs, const int *axis_of, int fixed_axis, const float *keys,
                                   const uint32_t *idx, float *k) {
  const Seg s = segs[blockIdx.x];
  const int axis = axis_of ? axis_of[blockIdx.x] : fixed_axis;
  const float *ka = keys + (size_t)axis * n;
  for (int i = threadIdx.x; i < s.len; i += blockDim.x) k[s.begin + i] = ka[idx[s.begin + i]];
}

// bounds of the triangles at positions [begin, begin + len) -> out[6] (lo xyz, hi xyz); all zero when len == 0
// (aabb_triangle_slice of rt_scene_build.c).  min / max are exact: the reduction order does not matter.
__device__ void slice_bounds(int n, const float *lo, const float *hi, const uint32_t *idx, int begin, int len, float out[6],
                             float *red /* [6][256] shared */) {
  float mn[3] = {RT_INF, RT_INF, RT_INF}, mx[3] = {-RT_INF, -RT_INF, -RT_INF};
  for (int i = threadIdx.x; i < len; i += blockDim.x) {
    uint32_t t = idx[begin + i];
#pragma unroll
    for (int a = 0; a < 3; a++) {
      float l = lo[(size_t)a * n + t], h = hi[(size_t)a * n + t];
      mn[a] = l < mn[a] ? l : mn[a];
      mx[a] = h > mx[a] ? h : mx[a];
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) { red[a * 256 + threadIdx.x] = mn[a]; red[(3 + a) * 256 + threadIdx.x] = mx[a]; }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        float x = red[a * 256 + threadIdx.x + s], y = red[(3 + a) * 256 + threadIdx.x + s];
        if (x < red[a * 256 + threadIdx.x]) red[a * 256 + threadIdx.x] = x;
        if (y > red[(3 + a) * 256 + threadIdx.x]) red[(3 + a) * 256 + threadIdx.x] = y;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 6; a++) out[a] = len > 0 ? red[a * 256] : 0.0f;
  __syncthreads();
}

// aabb_surface_area of rt_scene_build.c (scene.c:157-163), same operation order
__device__ float surface_area(const float b[6]) {
  float x = b[3] - b[0], y = b[4] - b[1], z = b[5] - b[2];
  return 2.0f * (x * y + y * z + z * x);
}

// one workgroup (256 threads) per slice: surface area of the left half + of the right half in the current order
__global__ __launch_bounds__(256) void slice_sa_kernel(int n, const Seg *segs, const float *lo, const float *hi, const uint32_t *idx,
                                                        float *sa /* [n_segs] */) {
  __shared__ float red[6 * 256];
  const Seg s = segs[blockIdx.x];
  float a[6], b[6];
  slice_bounds(n, lo, hi, idx, s.begin, s.split, a, red);
  slice_bounds(n, lo, hi, idx, s.begin + s.split, s.len - s.split, b, red);
  if (threadIdx.x == 0) sa[blockIdx.x] = surface_area(a) + surface_area(b);
}

// scene.c:345-357: the axis with the smallest sum, the LAST one among equals (`<=`)
__global__ void choose_axis_kernel(int n_segs, const float *sa0, const float *sa1, const float *sa2, int *axis) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_segs) return;
  float best = RT_INF;
  int   ax = 0;
  if (sa0[i] <= best) { best = sa0[i]; ax = 0; }
  if (sa1[i] <= best) { best = sa1[i]; ax = 1; }
  if (sa2[i] <= best) { best = sa2[i]; ax = 2; }
  axis[i] = ax;
}

// one workgroup per finished slice: its bounds are the box of child `slot` of node `node`
__global__ __launch_bounds__(256) void child_box_kernel(int n, const Fin *fins, const float *lo, const float *hi, const uint32_t *idx,
                                                         BVH_Node *nodes) {
  __shared__ float red[6 * 256];
  const Fin f = fins[blockIdx.x];
  float b[6];
  slice_bounds(n, lo, hi, idx, f.begin, f.len, b, red);
  if (threadIdx.x == 0) {
    BVH_Node *nd = &nodes[f.node];
    nd->min_x[f.slot] = b[0]; nd->min_y[f.slot] = b[1]; nd->min_z[f.slot] = b[2];
    nd->max_x[f.slot] = b[3]; nd->max_y[f.slot] = b[4]; nd->max_z[f.slot] = b[5];
  }
}

// triangles_insert of rt_scene_build.c (scene.c:105-155) for the leaf rows: position `pos` of the final order goes to
// slot `slot_of_pos[pos]` of the triangle block
__global__ void leaf_insert_kernel(int n, const Triangle *tris, const uint32_t *idx, const int *slot_of_pos, int len, float *block) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const Triangle &t = tris[idx[p]];
  const int slot = slot_of_pos[p];
  float *x[3], *y[3], *z[3];
  for (int k = 0; k < 3; k++) {
    x[k] = block + (size_t)len * (0 + k);
    y[k] = block + (size_t)len * (3 + k);
    z[k] = block + (size_t)len * (6 + k);
  }
  Triangle_AOS *aos = reinterpret_cast<Triangle_AOS *>(block + (size_t)len * 9) + slot;
  for (int k = 0; k < 3; k++) {
    x[k][slot] = t.positions[k].x;
    y[k][slot] = t.positions[k].y;
    z[k][slot] = t.positions[k].z;
  }
  rt_v3 p0 = rt_v3_make(t.positions[0].x, t.positions[0].y, t.positions[0].z);
  rt_v3 p1 = rt_v3_make(t.positions[1].x, t.positions[1].y, t.positions[1].z);
  rt_v3 p2 = rt_v3_make(t.positions[2].x, t.positions[2].y, t.positions[2].z);
  rt_v3 edge1 = rt_v3_sub(p1, p0), edge2 = rt_v3_sub(p2, p0);
  float du1 = t.tex_coords[1].x - t.tex_coords[0].x, dv1 = t.tex_coords[1].y - t.tex_coords[0].y;
  float du2 = t.tex_coords[2].x - t.tex_coords[0].x, dv2 = t.tex_coords[2].y - t.tex_coords[0].y;
  float d = du1 * dv2 - du2 * dv1;
  if (rt_absf(d) < 0.0001f) d = (d < 0) ? -0.0001f : 0.0001f;
  float inv_d = 1.0f / d;
  rt_v3 tangent = rt_v3_normalize_plain(rt_v3_scale(rt_v3_sub(rt_v3_scale(edge1, dv2), rt_v3_scale(edge2, dv1)), inv_d));
  rt_v3 bitangent = rt_v3_normalize_plain(rt_v3_scale(rt_v3_sub(rt_v3_scale(edge2, du1), rt_v3_scale(edge1, du2)), inv_d));
  rt_v3 fn = rt_v3_normalize_plain(rt_v3_cross_plain(edge1, edge2));
  aos->shader = t.shader;
  aos->normal.x = fn.x; aos->normal.y = fn.y; aos->normal.z = fn.z;
  aos->normal_a = t.normals[0];
  aos->normal_b = t.normals[1];
  aos->normal_c = t.normals[2];
  aos->tex_coords_a = t.tex_coords[0];
  aos->tex_coords_b = t.tex_coords[1];
  aos->tex_coords_c = t.tex_coords[2];
  aos->tangent.x = tangent.x; aos->tangent.y = tangent.y; aos->tangent.z = tangent.z;
  aos->bitangent.x = bitangent.x; aos->bitangent.y = bitangent.y; aos->bitangent.z = bitangent.z;
}

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T> T *as() const { return (T *)p; }
};

#define CK(expr)                                                                                       \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) { snprintf(err, (size_t)err_len, "%s failed: %s", #expr, hipGetErrorString(e_)); return -1; } \
  } while (0)

}  // namespace

// Builds into host memory that rt_scene_alloc() laid out: `nodes` (n_internal BVH_Node, zeroed) and `block` (the
// SoA + AoS triangle block of `block_len` slots, zeroed).  0 on success; on failure -1 and a message in err.
extern "C" int rt_gpu_build(const Triangle *h_tris, long n_in, long depth, BVH_Node *h_nodes, long n_internal, float *h_block,
                            long block_len, char *err, int err_len) {
  const int n = (int)n_in;
  if (n <= 0) return 0;
  const size_t block_bytes = (size_t)TRIANGLES_ALLOCATION_SIZE(block_len);

  DevBuf b_tris, b_keys, b_lo, b_hi, b_idx[2], b_k[2], b_nodes, b_block, b_segs, b_fins, b_sa, b_axis, b_off, b_slot, b_tmp;
  CK(b_tris.alloc((size_t)n * sizeof(Triangle)));
  CK(b_keys.alloc((size_t)n * 3 * 4)); CK(b_lo.alloc((size_t)n * 3 * 4)); CK(b_hi.alloc((size_t)n * 3 * 4));
  CK(b_idx[0].alloc((size_t)n * 4)); CK(b_idx[1].alloc((size_t)n * 4));
  CK(b_k[0].alloc((size_t)n * 4)); CK(b_k[1].alloc((size_t)n * 4));
  CK(b_nodes.alloc((size_t)(n_internal > 0 ? n_internal : 1) * sizeof(BVH_Node)));
  CK(b_block.alloc(block_bytes));
  CK(hipMemcpy(b_tris.p, h_tris, (size_t)n * sizeof(Triangle), hipMemcpyHostToDevice));
  CK(hipMemset(b_nodes.p, 0, (size_t)(n_internal > 0 ? n_internal : 1) * sizeof(BVH_Node)));
  CK(hipMemset(b_block.p, 0, block_bytes));
  const Triangle *d_tris = b_tris.as<Triangle>();
  float *keys = b_keys.as<float>(), *lo = b_lo.as<float>(), *hi = b_hi.as<float>();
  uint32_t *idx = b_idx[0].as<uint32_t>(), *idx_alt = b_idx[1].as<uint32_t>();
  float *k = b_k[0].as<float>(), *k_alt = b_k[1].as<float>();
  hipLaunchKernelGGL(prep_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_tris, keys, lo, hi, idx);

  // scratch sized for the largest generation: at most n slices / finished slices per level
  const size_t max_items = (size_t)n + 8;
  CK(b_segs.alloc(max_items * sizeof(Seg)));
  CK(b_fins.alloc(max_items * sizeof(Fin)));
  CK(b_sa.alloc(max_items * 3 * 4));
  CK(b_axis.alloc(max_items * 4));
  CK(b_off.alloc(max_items * 2 * 4));
  Seg *d_segs = b_segs.as<Seg>();
  Fin *d_fins = b_fins.as<Fin>();
  float *d_sa = b_sa.as<float>();
  int *d_axis = b_axis.as<int>();
  unsigned int *d_begin = b_off.as<unsigned int>(), *d_end = d_begin + max_items;

  size_t tmp_bytes = 0;
  void *tmp = nullptr;
  // one stable segmented sort of (k, idx) by k inside the given slices; everything outside the slices keeps its place
  auto sort_slices = [&](unsigned n_segs) -> hipError_t {
    hipError_t e = hipMemcpyAsync(idx_alt, idx, (size_t)n * 4, hipMemcpyDeviceToDevice, 0);
    if (e != hipSuccess) return e;
    size_t need = 0;
    e = rocprim::segmented_radix_sort_pairs(nullptr, need, k, k_alt, idx, idx_alt, (unsigned)n, n_segs, d_begin, d_end);
    if (e != hipSuccess) return e;
    if (need > tmp_bytes) {
      if (tmp) (void)hipFree(tmp);
      tmp = nullptr;
      b_tmp.p = nullptr;              // (b_tmp's destructor must not free the old block a second time if the hipMalloc below fails)
      tmp_bytes = 0;
      e = hipMalloc(&tmp, need);
      if (e != hipSuccess) return e;
      b_tmp.p = tmp;
      tmp_bytes = need;
    }
    size_t sz = tmp_bytes;
    e = rocprim::segmented_radix_sort_pairs(tmp, sz, k, k_alt, idx, idx_alt, (unsigned)n, n_segs, d_begin, d_end);
    if (e != hipSuccess) return e;
    uint32_t *t = idx; idx = idx_alt; idx_alt = t;
    return hipSuccess;
  };

  std::vector<NodeSeg> level_nodes, next_nodes;
  level_nodes.push_back({0, n, 0});
  for (long d = depth; d > 0; d--) {                 // d = internal levels at and below the nodes of this level
    std::vector<std::vector<Seg>> gens;
    std::vector<Fin> fins;
    for (const NodeSeg &nd : level_nodes) plan_node(nd, (int)d, gens, fins);
    for (const std::vector<Seg> &segs : gens) {
      const unsigned n_segs = (unsigned)segs.size();
      if (n_segs == 0) continue;
      std::vector<unsigned int> off((size_t)n_segs * 2);
      for (unsigned i = 0; i < n_segs; i++) { off[i] = (unsigned)segs[i].begin; off[n_segs + i] = (unsigned)(segs[i].begin + segs[i].len); }
      CK(hipMemcpy(d_segs, segs.data(), (size_t)n_segs * sizeof(Seg), hipMemcpyHostToDevice));
      CK(hipMemcpy(d_begin, off.data(), (size_t)n_segs * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(d_end, off.data() + n_segs, (size_t)n_segs * 4, hipMemcpyHostToDevice));
      for (int axis = 0; axis < 3; axis++) {          // scene.c:345-357: sort by x, then y, then z, each on the previous order
        hipLaunchKernelGGL(gather_keys_kernel, dim3(n_segs), dim3(256), 0, 0, n, d_segs, (const int *)nullptr, axis, keys, idx, k);
        CK(sort_slices(n_segs));
        hipLaunchKernelGGL(slice_sa_kernel, dim3(n_segs), dim3(256), 0, 0, n, d_segs, lo, hi, idx, d_sa + (size_t)axis * max_items);
      }
      hipLaunchKernelGGL(choose_axis_kernel, dim3((n_segs + 255) / 256), dim3(256), 0, 0, (int)n_segs, d_sa, d_sa + max_items,
                         d_sa + 2 * max_items, d_axis);
      // scene.c:358: `if (best_axis != 2) sort by best_axis` -- re-sorting a z-sorted slice by z changes nothing
      hipLaunchKernelGGL(gather_keys_kernel, dim3(n_segs), dim3(256), 0, 0, n, d_segs, (const int *)d_axis, 0, keys, idx, k);
      CK(sort_slices(n_segs));
    }
    if (!fins.empty()) {
      CK(hipMemcpy(d_fins, fins.data(), fins.size() * sizeof(Fin), hipMemcpyHostToDevice));
      hipLaunchKernelGGL(child_box_kernel, dim3((unsigned)fins.size()), dim3(256), 0, 0, n, d_fins, lo, hi, idx, b_nodes.as<BVH_Node>());
    }
    next_nodes.clear();
    for (const Fin &f : fins) next_nodes.push_back({f.begin, f.len, 8 * f.node + 1 + f.slot});
    level_nodes.swap(next_nodes);
  }
  // last row: level_nodes are leaf groups (or the single group of a depth-0 scene)
  {
    std::vector<int> slot_of_pos((size_t)n, 0);
    for (const NodeSeg &nd : level_nodes) {
      if (nd.len > 8) { snprintf(err, (size_t)err_len, "internal: leaf group with %d triangles", nd.len); return -1; }
      long group = (long)nd.node - n_internal;
      for (int i = 0; i < nd.len; i++) slot_of_pos[(size_t)nd.begin + i] = (int)(group * 8 + i);
    }
    CK(b_slot.alloc((size_t)n * 4));
    CK(hipMemcpy(b_slot.p, slot_of_pos.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(leaf_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_tris, idx, b_slot.as<int>(), (int)block_len,
                       b_block.as<float>());
  }
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  if (n_internal > 0) CK(hipMemcpy(h_nodes, b_nodes.p, (size_t)n_internal * sizeof(BVH_Node), hipMemcpyDeviceToHost));
  CK(hipMemcpy(h_block, b_block.p, block_bytes, hipMemcpyDeviceToHost));
  return 0;
}
