// rt_kernels.hip -- gfx950 device code of the render hot path.
//
// What runs here replaces, on the GPU, the reference's
//   render_thread_proc   raytracer.c:596-720   (pixel / sample loop)
//   cast_ray             raytracer.c:505-558   (bounce loop)
//   ray_bvh_node_hit     raytracer.c:443-483   (near-first 8-ary traversal)
//   ray_aabbs_hit_8      raytracer.c:190-230   (8-box slab test)
//   ray_triangles_hit_8  raytracer.c:84-188    (8-triangle Moeller-Trumbore)
//   disney_shader_proc & friends  driver.c:49-418 (textures, BSDF, background)
//
// Design (DESIGN.md has the long form):
//  * One persistent wave64 per scheduler slot.  A wave dequeues work items
//    (8x8 pixel tile x slab of samples) from a global head counter and keeps all
//    64 lanes busy by path regeneration: a lane whose path ended takes the next
//    (pixel, sample) of the item through a wave ballot / prefix count.
//  * One ray per lane.  Traversal keeps, per lane and per tree level, one
//    32-bit word in LDS holding the not-yet-visited children of the node on
//    that level in near-first order (3 bits each + count).  The entry distance
//    of a popped child is recomputed from the node (6 floats) only when the
//    closest hit changed since that node was entered; this reproduces the
//    reference's visiting order and its `dist < hit.distance` test exactly.
//  * Radiance is accumulated in 32.32 fixed point (rt_math.h), first in LDS
//    per tile, then with 64-bit integer atomics in HBM: exact and independent
//    of scheduling, so images are bit-identical to the CPU oracle.
//  * All arithmetic goes through include/rt_math.h and is compiled with
//    -ffp-contract=off: no fused multiply-add that the CPU would not do.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"
#include "../../include/rt_math.h"

#define RT_BLOCK_WAVES 4
#define RT_BLOCK_THREADS (RT_BLOCK_WAVES * 64)

// counters slots
#define CNT_PATHS 0
#define CNT_RAYS 1
#define CNT_NODES 2
#define CNT_LEAVES 3
#define CNT_SHADES 4
#define CNT_BG 5
#define CNT_TEXTURED 6

struct Ray3 {
  rt_v3 o, d;
  float inv_x, inv_y, inv_z;
  bool  fast;     // origin and reciprocal direction all finite: NaN-free slab arithmetic
};

struct HitRec {
  float t;
  int   tri;
  float u, v;
};

struct LaneCounters {
  uint32_t rays, nodes, leaves, shades, bgs, textured, paths;
};

__device__ __forceinline__ float4 ld4(const float *base, int idx4) {
  return reinterpret_cast<const float4 *>(base)[idx4];
}

// Scalar-cache reads: a pointer in the constant address space with a uniform (SGPR) address makes hipcc emit
// s_load_dwordx16 instead of one vector load per lane.  Used by the plain kernel (rt_path_kernel / trace_ray) for
// wave-uniform nodes and leaves; the scheduled kernel dropped these paths (LDS broadcast reads are faster and the
// 48 / 72 SGPRs per node / leaf tile cost it 34 spilled SGPRs).
typedef const float __attribute__((address_space(4))) cfloat;
__device__ __forceinline__ cfloat *as_scalar_ptr(const float *p) { return (cfloat *)(unsigned long long)p; }

__device__ __forceinline__ float as_f(int i) { return __int_as_float(i); }
__device__ __forceinline__ int   as_i(float f) { return __float_as_int(f); }

// ---------------------------------------------------------------------------------
// Slab tests.  Two code paths with identical results wherever both are defined:
//  * EXACT reproduces the operand order and the NaN behaviour of _mm256_min_ps /
//    _mm256_max_ps in raytracer.c:209-228 with compare+select;
//  * FAST uses v_min_f32 / v_max3_f32.  It is taken only for rays whose origin and
//    reciprocal direction are all finite (Ray3::fast): then no NaN can appear in
//    the slab arithmetic, and on NaN-free operands min/max are plain min/max, so
//    both paths return the same bits (sign of zero cannot matter: every distance
//    is clamped to >= EPSILON before it is used).  Axis-aligned rays (0 * inf)
//    take the EXACT path; tests/test_gpu_parity.py sends such rays.

__device__ __forceinline__ float fmin_hw(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_hw(float a, float b) { return __builtin_fmaxf(a, b); }

template <bool FAST>
__device__ __forceinline__ float slab_entry(const Ray3 &r, float mnx, float mny, float mnz,
                                            float mxx, float mxy, float mxz, float t_max) {
  float t0x = (mnx - r.o.x) * r.inv_x, t1x = (mxx - r.o.x) * r.inv_x;
  float t0y = (mny - r.o.y) * r.inv_y, t1y = (mxy - r.o.y) * r.inv_y;
  float t0z = (mnz - r.o.z) * r.inv_z, t1z = (mxz - r.o.z) * r.inv_z;
  if (FAST) {
    float sx = fmin_hw(t0x, t1x), sy = fmin_hw(t0y, t1y), sz = fmin_hw(t0z, t1z);
    float bx = fmax_hw(t0x, t1x), by = fmax_hw(t0y, t1y), bz = fmax_hw(t0z, t1z);
    float t_minv = fmax_hw(RT_EPS, fmax_hw(sx, fmax_hw(sy, sz)));
    float t_maxv = fmin_hw(t_max, fmin_hw(bx, fmin_hw(by, bz)));
    // t_maxv <= t_max, so "entry < exit" already implies the candidate test entry < t_max
    return (t_minv < t_maxv) ? t_minv : RT_INF;
  } else {
    float sx = rt_min_ps(t0x, t1x), sy = rt_min_ps(t0y, t1y), sz = rt_min_ps(t0z, t1z);
    float bx = rt_max_ps(t0x, t1x), by = rt_max_ps(t0y, t1y), bz = rt_max_ps(t0z, t1z);
    float t_minv = rt_max_ps(RT_EPS, rt_max_ps(sx, rt_max_ps(sy, sz)));
    float t_maxv = rt_min_ps(t_max, rt_min_ps(bx, rt_min_ps(by, bz)));
    float e = (t_minv >= t_maxv) ? RT_INF : t_minv;
    return (e < t_max) ? e : RT_INF;       // candidate test of raytracer.c:464
  }
}

// Entry distance of child j only; the miss test against t_max was already passed
// when the node was entered, so only t_minv is needed (see header comment).
template <bool FAST>
__device__ __forceinline__ float slab_entry_child(const float *n, const Ray3 &r) {
  // n -> element j of the node's first row; the six rows are 8 floats apart
  float mnx = n[0], mny = n[8], mnz = n[16], mxx = n[24], mxy = n[32], mxz = n[40];
  float t0x = (mnx - r.o.x) * r.inv_x, t1x = (mxx - r.o.x) * r.inv_x;
  float t0y = (mny - r.o.y) * r.inv_y, t1y = (mxy - r.o.y) * r.inv_y;
  float t0z = (mnz - r.o.z) * r.inv_z, t1z = (mxz - r.o.z) * r.inv_z;
  if (FAST) {
    float sx = fmin_hw(t0x, t1x), sy = fmin_hw(t0y, t1y), sz = fmin_hw(t0z, t1z);
    return fmax_hw(RT_EPS, fmax_hw(sx, fmax_hw(sy, sz)));
  }
  float sx = rt_min_ps(t0x, t1x), sy = rt_min_ps(t0y, t1y), sz = rt_min_ps(t0z, t1z);
  return rt_max_ps(RT_EPS, rt_max_ps(sx, rt_max_ps(sy, sz)));
}

// Tests the 8 children of `node` against the ray with t_max = hit_t and returns
// the near-first visiting order of the children that can still matter:
//   bits 0..23  child indices, nearest first (ties: lowest index first)
//   bits 24..27 how many of them are candidates (entry < hit_t)
// This is the selection loop of raytracer.c:459-468 done once, as a rank sort.
// Candidate distances are positive floats or +inf, so they order like their bit
// patterns: rank arithmetic runs on integers (sign bit of a difference), without
// compare/select pairs.  Non-candidates (+inf) rank behind every candidate, so
// the 8 ranks are a permutation and the word needs no per-child condition.
#define NODE_GLOBAL 0     // per-lane vector loads from HBM/L2/L1
#define NODE_SCALAR 1     // wave-uniform node: s_load through the scalar cache
#define NODE_LDS    2     // per-lane reads from the workgroup's LDS copy of the top of the tree
#define NODE_LDS_ORDERED 3   // NODE_LDS with the slab planes picked by address (FAST rays, boxes with min <= max)
#define RT_LDS_NODE_F4 13 // LDS node stride in float4 (12 data + 1 pad: 13 is odd, so random nodes spread over all 16-byte slots of a bank row)

// float4 index of LDS node `node`: a 24-bit multiply is full rate, v_mul_lo_u32 a quarter
__device__ __forceinline__ int lds_node_f4(int node) { return (int)__umul24((unsigned)node, (unsigned)RT_LDS_NODE_F4); }

template <bool FAST, int MODE>
__device__ __forceinline__ uint32_t node_enter(const RT_KParams &P, const Ray3 &r, int node, float hit_t,
                                               const float4 *lds_nodes) {
  int d[8];
  if (MODE == NODE_SCALAR) {               // `node` is wave-uniform: node data lives in SGPRs
    cfloat *nb = as_scalar_ptr(P.nodes) + (size_t)node * 48;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      d[k] = as_i(slab_entry<FAST>(r, nb[k], nb[8 + k], nb[16 + k], nb[24 + k], nb[32 + k], nb[40 + k], hit_t));
    }
  } else if (FAST && MODE == NODE_LDS_ORDERED) {
    // Near and far plane of every slab picked by ADDRESS from the sign of the reciprocal direction instead of by min / max
    // of the two products: with min <= max in every box (checked at upload, rt_api.cpp) and NaN-free operands,
    // (mn - o) * inv <= (mx - o) * inv for inv > 0 and >= for inv < 0 -- rounding is monotonic -- so the picked product IS
    // the minimum (maximum); for inv = 0 both are zero.  Six min / max fewer per child.
    const char *nbase = reinterpret_cast<const char *>(lds_nodes + lds_node_f4(node));
    const int nx = (as_i(r.inv_x) >> 31) & 96, ny = (as_i(r.inv_y) >> 31) & 96, nz = (as_i(r.inv_z) >> 31) & 96;   // bytes: min rows 0 / 32 / 64, max rows +96
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const float4 ax = *reinterpret_cast<const float4 *>(nbase + nx + h * 16), bx = *reinterpret_cast<const float4 *>(nbase + (96 - nx) + h * 16);
      const float4 ay = *reinterpret_cast<const float4 *>(nbase + 32 + ny + h * 16), by = *reinterpret_cast<const float4 *>(nbase + 32 + (96 - ny) + h * 16);
      const float4 az = *reinterpret_cast<const float4 *>(nbase + 64 + nz + h * 16), bz = *reinterpret_cast<const float4 *>(nbase + 64 + (96 - nz) + h * 16);
      const float nxs[4] = {ax.x, ax.y, ax.z, ax.w}, fxs[4] = {bx.x, bx.y, bx.z, bx.w};
      const float nys[4] = {ay.x, ay.y, ay.z, ay.w}, fys[4] = {by.x, by.y, by.z, by.w};
      const float nzs[4] = {az.x, az.y, az.z, az.w}, fzs[4] = {bz.x, bz.y, bz.z, bz.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float sx = (nxs[k] - r.o.x) * r.inv_x, bxx = (fxs[k] - r.o.x) * r.inv_x;
        const float sy = (nys[k] - r.o.y) * r.inv_y, byy = (fys[k] - r.o.y) * r.inv_y;
        const float sz = (nzs[k] - r.o.z) * r.inv_z, bzz = (fzs[k] - r.o.z) * r.inv_z;
        const float t_minv = fmax_hw(RT_EPS, fmax_hw(sx, fmax_hw(sy, sz)));
        const float t_maxv = fmin_hw(hit_t, fmin_hw(bxx, fmin_hw(byy, bzz)));
        d[h * 4 + k] = as_i((t_minv < t_maxv) ? t_minv : RT_INF);
      }
    }
  } else {
    const float4 *nb = (MODE == NODE_LDS || MODE == NODE_LDS_ORDERED) ? (lds_nodes + lds_node_f4(node))
                                          : (reinterpret_cast<const float4 *>(P.nodes) + (size_t)node * 12);
#pragma unroll
    for (int h = 0; h < 2; h++) {          // children 0-3, then 4-7: half the node in registers at a time
      float4 mnx = nb[0 + h], mny = nb[2 + h], mnz = nb[4 + h];
      float4 mxx = nb[6 + h], mxy = nb[8 + h], mxz = nb[10 + h];
      d[h * 4 + 0] = as_i(slab_entry<FAST>(r, mnx.x, mny.x, mnz.x, mxx.x, mxy.x, mxz.x, hit_t));
      d[h * 4 + 1] = as_i(slab_entry<FAST>(r, mnx.y, mny.y, mnz.y, mxx.y, mxy.y, mxz.y, hit_t));
      d[h * 4 + 2] = as_i(slab_entry<FAST>(r, mnx.z, mny.z, mnz.z, mxx.z, mxy.z, mxz.z, hit_t));
      d[h * 4 + 3] = as_i(slab_entry<FAST>(r, mnx.w, mny.w, mnz.w, mxx.w, mxy.w, mxz.w, hit_t));
    }
  }

  // rank[k] starts at k (the pairs (j,k), j<k, it loses by default) and moves by the sign bits
  int rank[8];
#pragma unroll
  for (int k = 0; k < 8; k++) rank[k] = k;
#pragma unroll
  for (int j = 0; j < 8; j++) {
#pragma unroll
    for (int k = j + 1; k < 8; k++) {
      int kb = (int)((uint32_t)(d[k] - d[j]) >> 31);      // 1 iff d[k] < d[j]
      rank[j] += kb;
      rank[k] -= kb;
    }
  }
  uint32_t w = 0, n_inf = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    int sh;       // 3 * rank as one full-rate v_lshl_add_u32 (hipcc turns r + 2r into a quarter-rate v_mul_lo_u32)
    asm("v_lshl_add_u32 %0, %1, 1, %1" : "=v"(sh) : "v"(rank[j]));
    w |= (uint32_t)j << sh;
    n_inf += ((uint32_t)d[j] + 0x00800000u) >> 31;          // 1 iff d[j] == +inf
  }
  return w | ((8u - n_inf) << 24);
}

// 1 / x in six instructions where hipcc's IEEE division takes eleven (one of them the double-length v_rcp_f32 in both):
// x is scaled by 2^24 (exact; brings every denormal into the normal range), v_rcp_f32 (1 ulp) is refined by one Newton
// step with the exact residual (two FMAs), v_div_fixup_f32 restores zero / infinity / NaN, and the quotient is scaled by
// 2^24 again (exact, or the same overflow to infinity as the division's).  The result EQUALS the correctly rounded 1.0f / x
// for EVERY x with |x| < 2^102 and for infinity and NaN: all those bit patterns are compared on the GPU against the IEEE
// sequence (rt_test_rcp_sweep, tests/test_gpu_parity.py).  For 2^102 <= |x| < infinity (quotient below 2^-102, 2^24 x
// beyond v_rcp_f32's range) it is NOT: callers must exclude such x (RT_SHORT_DIV_MAX_X) or divide.
#define RT_SHORT_DIV_MAX_X 0x1p102f
__device__ __forceinline__ float rcp_exact(float x) {
  float xs = x * 0x1p24f;
  float y = __builtin_amdgcn_rcpf(xs);
  float e = __builtin_fmaf(-xs, y, 1.0f);
  float z = __builtin_fmaf(e, y, y);
  return __builtin_amdgcn_div_fixupf(z, xs, 1.0f) * 0x1p24f;
}
__device__ __forceinline__ bool rcp_exact_outside(float x) {
  return __builtin_fabsf(x) >= RT_SHORT_DIV_MAX_X && __builtin_fabsf(x) < RT_INF;
}

// rt_v3_normalize() (rt_math.h: v * (1 / sqrt(v.v))) with the reciprocal from rcp_exact(): a square root lies in [0, 2^64] or
// is infinite / NaN, inside the domain on which rcp_exact() equals the division -- same bits, 5 instructions fewer.
__device__ __forceinline__ rt_v3 normalize_dev(rt_v3 v) { return rt_v3_scale(v, rcp_exact(rt_sqrtf(rt_v3_dot(v, v)))); }

// rt_accum_quantize() (rt_math.h: clamp to [0, 2^20], times 2^32 in double, truncate to u64) by shifts of the mantissa:
// mantissa << 29 is the value at exponent field 147 (2^20), one right shift brings it to its own exponent.  Same integer
// for every float (rt_test_quantize_sweep: all 2^32 bit patterns); the f64 conversions and multiplies cost twice as much.
__device__ __forceinline__ unsigned long long accum_quantize_dev(float c) {
  float v = (c > 0.0f) ? c : 0.0f;
  v = (v > RT_ACCUM_MAX) ? RT_ACCUM_MAX : v;
  const uint32_t b = __float_as_uint(v);
  const uint32_t e = b >> 23;                                          // 0 .. 147 after the clamp
  const unsigned long long m = (unsigned long long)((b & 0x007FFFFFu) | 0x00800000u) << 29;
  const uint32_t k = 147u - e;
  return m >> (k < 63u ? k : 63u);                                     // (exponent field 0: zero and denormals end as 0)
}

// 8-triangle test of leaf group g (raytracer.c:84-188 + min_f32x8 :15-32).
// One triangle: Moeller-Trumbore without determinant test; returns the sanitised distance.
__device__ __forceinline__ float tri_test(const Ray3 &r, float ax, float ay, float az, float e1x, float e1y, float e1z,
                                          float e2x, float e2y, float e2z, float &u_out, float &v_out) {
  // the leaf tile stores a, b-a, c-a: the two edge subtractions of raytracer.c:115-122 are done once
  // at upload (same fp32 subtraction, same bits) instead of once per visit
  rt_v3 a = rt_v3_make(ax, ay, az);
  rt_v3 edge1 = rt_v3_make(e1x, e1y, e1z);
  rt_v3 edge2 = rt_v3_make(e2x, e2y, e2z);
  rt_v3 rxe2 = rt_v3_cross(r.d, edge2);
  float det = rt_v3_dot(edge1, rxe2);
  float inv_det = 1.0f / det;
  rt_v3 s = rt_v3_sub(r.o, a);
  rt_v3 sxe1 = rt_v3_cross(s, edge1);
  float u = inv_det * rt_v3_dot(s, rxe2);
  float v = inv_det * rt_v3_dot(r.d, sxe1);
  float t = inv_det * rt_v3_dot(edge2, sxe1);
  bool miss = (u < -RT_EPS) || (u > 1.0f + RT_EPS) || (v < -RT_EPS) || (u + v > 1.0f + RT_EPS) || (t < RT_EPS);
  float dist = miss ? RT_INF : t;
  u_out = u;
  v_out = v;
  return (dist > 0.0f) ? dist : RT_INF;          // NaN -> +inf (min_f32x8)
}

// leaf_test<false>() with 1 / det by rcp_exact(): same bits as long as every |det| < 2^102, which the host guarantees
// from the scene's edge lengths and the camera matrix before it selects the kernel built on this (rt_api.cpp).
__device__ __forceinline__ bool leaf_test_short_div(const RT_KParams &P, const Ray3 &r, int g, HitRec &hit) {
  float best = RT_INF, bu = 0.0f, bv = 0.0f;
  int   bi = 0;
  const float *lb = P.leaves + (size_t)g * 72;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    float4 x0 = ld4(lb, 0 + h), x1 = ld4(lb, 2 + h), x2 = ld4(lb, 4 + h);
    float4 y0 = ld4(lb, 6 + h), y1 = ld4(lb, 8 + h), y2 = ld4(lb, 10 + h);
    float4 z0 = ld4(lb, 12 + h), z1 = ld4(lb, 14 + h), z2 = ld4(lb, 16 + h);
    float ax[4] = {x0.x, x0.y, x0.z, x0.w}, bx[4] = {x1.x, x1.y, x1.z, x1.w}, cx[4] = {x2.x, x2.y, x2.z, x2.w};
    float ay[4] = {y0.x, y0.y, y0.z, y0.w}, by[4] = {y1.x, y1.y, y1.z, y1.w}, cy[4] = {y2.x, y2.y, y2.z, y2.w};
    float az[4] = {z0.x, z0.y, z0.z, z0.w}, bz[4] = {z1.x, z1.y, z1.z, z1.w}, cz[4] = {z2.x, z2.y, z2.z, z2.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // tri_test() with the other reciprocal; same expression order
      rt_v3 edge1 = rt_v3_make(bx[k], by[k], bz[k]), edge2 = rt_v3_make(cx[k], cy[k], cz[k]);
      rt_v3 rxe2 = rt_v3_cross(r.d, edge2);
      float det = rt_v3_dot(edge1, rxe2);
      float inv_det = rcp_exact(det);
      rt_v3 s = rt_v3_sub(r.o, rt_v3_make(ax[k], ay[k], az[k]));
      rt_v3 sxe1 = rt_v3_cross(s, edge1);
      float u = inv_det * rt_v3_dot(s, rxe2);
      float v = inv_det * rt_v3_dot(r.d, sxe1);
      float t = inv_det * rt_v3_dot(edge2, sxe1);
      bool miss = (u < -RT_EPS) || (u > 1.0f + RT_EPS) || (v < -RT_EPS) || (u + v > 1.0f + RT_EPS) || (t < RT_EPS);
      float dist = miss ? RT_INF : t;
      dist = (dist > 0.0f) ? dist : RT_INF;          // NaN -> +inf (min_f32x8)
      if (dist < best) { best = dist; bi = h * 4 + k; bu = u; bv = v; }   // lowest lane wins ties
    }
  }
  if (best < hit.t) {
    hit.t = best;
    hit.tri = g * 8 + bi;
    hit.u = bu;
    hit.v = bv;
    return true;
  }
  return false;
}

template <bool SCALAR>
__device__ __forceinline__ bool leaf_test(const RT_KParams &P, const Ray3 &r, int g, HitRec &hit) {
  float best = RT_INF, bu = 0.0f, bv = 0.0f;
  int   bi = 0;
  if (SCALAR) {                            // `g` is wave-uniform: the 288-byte tile comes through SGPRs
    cfloat *lb = as_scalar_ptr(P.leaves) + (size_t)g * 72;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      float u, v;
      float dist = tri_test(r, lb[k], lb[24 + k], lb[48 + k], lb[8 + k], lb[32 + k], lb[56 + k],
                            lb[16 + k], lb[40 + k], lb[64 + k], u, v);
      if (dist < best) { best = dist; bi = k; bu = u; bv = v; }   // lowest lane wins ties
    }
  } else {
    const float *lb = P.leaves + (size_t)g * 72;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      float4 x0 = ld4(lb, 0 + h), x1 = ld4(lb, 2 + h), x2 = ld4(lb, 4 + h);
      float4 y0 = ld4(lb, 6 + h), y1 = ld4(lb, 8 + h), y2 = ld4(lb, 10 + h);
      float4 z0 = ld4(lb, 12 + h), z1 = ld4(lb, 14 + h), z2 = ld4(lb, 16 + h);
      float ax[4] = {x0.x, x0.y, x0.z, x0.w}, bx[4] = {x1.x, x1.y, x1.z, x1.w}, cx[4] = {x2.x, x2.y, x2.z, x2.w};
      float ay[4] = {y0.x, y0.y, y0.z, y0.w}, by[4] = {y1.x, y1.y, y1.z, y1.w}, cy[4] = {y2.x, y2.y, y2.z, y2.w};
      float az[4] = {z0.x, z0.y, z0.z, z0.w}, bz[4] = {z1.x, z1.y, z1.z, z1.w}, cz[4] = {z2.x, z2.y, z2.z, z2.w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float u, v;
        float dist = tri_test(r, ax[k], ay[k], az[k], bx[k], by[k], bz[k], cx[k], cy[k], cz[k], u, v);
        if (dist < best) { best = dist; bi = h * 4 + k; bu = u; bv = v; }   // lowest lane wins ties
      }
    }
  }
  if (best < hit.t) {
    hit.t = best;
    hit.tri = g * 8 + bi;
    hit.u = bu;
    hit.v = bv;
    return true;
  }
  return false;
}

// Closest hit along r (raytracer.c:497-503 -> :443-483).  perm = this wave's
// LDS perm-stack, indexed [level*64 + lane].
template <bool FAST>
__device__ __forceinline__ void trace_ray(const RT_KParams &P, const Ray3 &r, HitRec &hit,
                                          uint32_t *perm, int lane, LaneCounters &cn) {
  hit.t = RT_INF;
  hit.tri = -1;
  hit.u = 0.0f;
  hit.v = 0.0f;
  cn.rays += 1;
  if (P.depth <= 0) {          // one leaf group, no nodes (rt_scene.h, depth-0 rule)
    cn.leaves += 1;
    leaf_test<true>(P, r, 0, hit);
    return;
  }
  const int leaf_level = P.depth - 1;
  int      level = 0, node = 0;
  uint32_t dirty = 0;
  cn.nodes += 1;
  uint32_t cur = node_enter<FAST, FAST ? NODE_SCALAR : NODE_GLOBAL>(P, r, 0, hit.t, nullptr);     // the root is uniform by construction

  while (level >= 0) {
    uint32_t cnt = cur >> 24;
    bool do_leaf = false, do_enter = false;
    int  child = 0;
    if (cnt == 0) {
      level -= 1;
      node = (node - 1) >> 3;
      if (level >= 0) cur = perm[level * 64 + lane];
    } else {
      int j = (int)(cur & 7u);
      cur = ((cur >> 3) & 0x1FFFFFu) | ((cnt - 1u) << 24);
      bool go = true;
      if ((dirty >> level) & 1u) {
        float dj = slab_entry_child<FAST>(P.nodes + (size_t)node * 48 + j, r);
        if (!(dj < hit.t)) { cur = 0; go = false; }      // raytracer.c:470-472
      }
      if (go) {
        child = 8 * node + 1 + j;
        do_leaf = (level == leaf_level);
        do_enter = !do_leaf;
      }
    }
    if (do_leaf) {
      cn.leaves += 1;
      int  g = child - P.last_row_offset;
      int  g0 = __builtin_amdgcn_readfirstlane(g);
      bool got;
      if (FAST && __ballot(g != g0) == 0) got = leaf_test<true>(P, r, g0, hit);    // all lanes on one leaf
      else got = leaf_test<false>(P, r, g, hit);
      if (got) dirty = 0xFFFFFFFFu;
    }
    if (do_enter) {
      perm[level * 64 + lane] = cur;
      node = child;
      level += 1;
      cn.nodes += 1;
      int n0 = __builtin_amdgcn_readfirstlane(node);
      if (FAST && __ballot(node != n0) == 0) cur = node_enter<FAST, NODE_SCALAR>(P, r, n0, hit.t, nullptr);   // all lanes on one node
      else cur = node_enter<FAST, NODE_GLOBAL>(P, r, node, hit.t, nullptr);
      dirty &= ~(1u << level);
    }
  }
}

// ---------------------------------------------------------------------------------
// textures (driver.c:49-93); texels are RGBA8, alpha unused
// u8 / 255.999f (driver.c:70-87) as a multiplication: i * RN(1/255.999f) equals RN(i / 255.999f)
// for EVERY i in 0..255 (checked exhaustively in tests/test_oracle_kat.py), so this is the same
// value as the reference's division at a tenth of the instructions.
__device__ __forceinline__ rt_v3 texel_rgb(uint32_t t) {
  const float k = 1.0f / 255.999f;
  return rt_v3_make((float)(int)(t & 0xFFu) * k, (float)(int)((t >> 8) & 0xFFu) * k,
                    (float)(int)((t >> 16) & 0xFFu) * k);
}

template <class PT>
__device__ __forceinline__ rt_v3 tex_bilinear(const PT &P, int tex, float tx, float ty) {
  RT_DTexture T = P.textures[tex];
  if (tx < 0) tx += (float)(-(int)tx + 1);
  if (ty < 0) ty += (float)(-(int)ty + 1);
  tx = rt_fractf(tx);
  ty = rt_fractf(ty);
  float px = tx * (float)T.width;
  float py = ty * (float)T.height;
  int u = (int)px, v = (int)py;
  if (u > T.width - 1) u = T.width - 1;
  if (v > T.height - 1) v = T.height - 1;
  float a = px - (float)u;
  float b = py - (float)v;
  int u2 = (u + 1 < T.width) ? u + 1 : u;
  int v2 = (v + 1 < T.height) ? v + 1 : v;
  const uint32_t *tp = P.texels + T.offset;
  rt_v3 c00 = texel_rgb(tp[u + T.stride * v]);
  rt_v3 c10 = texel_rgb(tp[u2 + T.stride * v]);
  rt_v3 c01 = texel_rgb(tp[u + T.stride * v2]);
  rt_v3 c11 = texel_rgb(tp[u2 + T.stride * v2]);
  rt_v3 c0 = rt_v3_lerp(c00, c10, a);
  rt_v3 c1 = rt_v3_lerp(c01, c11, a);
  return rt_v3_lerp(c0, c1, b);
}

// rt_srgb_to_linear() of a bilinear texture sample: (x + 0.055f) / 1.055f as a multiplication by RN(1 / 1.055f) corrected
// with the exact residual (two FMAs) -- 3 instructions where the IEEE division takes 11.  The corrected quotient equals
// the division for every a = x + 0.055f with 2^-104 <= |a| < infinity and for NaN (tools/exp/div_test.hip, all 2^32 a);
// rt_test_srgb_sweep compares every x in [0, 2] -- 1.07 G bit patterns -- with rt_srgb_to_linear1().  A texture sample is a
// lerp of u8 / 255.999 values, 0 <= x <= 0.9961, or NaN for NaN texture coordinates.  Same rt_powf() afterwards.
__device__ __forceinline__ float srgb_to_linear_tex1(float x) {
#ifdef RT_EXP_SRGB_IEEE
  return rt_srgb_to_linear1(x);
#endif
  const float c = 1.0f / 1.055f;
  float a = x + 0.055f;
  float q = a * c;
  float r = __builtin_fmaf(-1.055f, q, a);
  return rt_powf(__builtin_fmaf(r, c, q), 2.4f);
}
__device__ __forceinline__ rt_v3 srgb_to_linear_tex(rt_v3 v) {
  return rt_v3_make(srgb_to_linear_tex1(v.x), srgb_to_linear_tex1(v.y), srgb_to_linear_tex1(v.z));
}

// driver.c:95-104
template <class PT>
__device__ __forceinline__ rt_v3 background_lookup(const PT &P, rt_v3 dir) {
  float inv_pi = 1.0f / RT_PI;
  float inv_two_pi = 1.0f / (2.0f * RT_PI);
  float u = 0.5f + rt_atan2f(dir.z, dir.x) * inv_two_pi;
  float v = 0.5f - rt_asinf(dir.y) * inv_pi;
  return srgb_to_linear_tex(tex_bilinear(P, P.bg_texture, u, v));
}

// ---------------------------------------------------------------------------------
// Disney-style BSDF, driver.c:118-348.  Expression order matches oracle/oracle.c.

__device__ __forceinline__ float pow5(float m) { return m * m * m * m * m; }
__device__ __forceinline__ float luminance(rt_v3 x) { return rt_v3_dot(x, rt_v3_make(0.2126f, 0.7152f, 0.0722f)); }

__device__ __forceinline__ float ggx_D(float roughness, float NoH) {          // driver.c:212-215, k = 2
  float a2 = roughness * roughness;
  float d = (NoH * NoH) * (a2 * a2 - 1.0f) + 1.0f;
  return a2 / (RT_PI * (d * d));
}

__device__ __forceinline__ float smith_G(float NDotV, float alpha2) {         // driver.c:217-221
  float a = alpha2 * alpha2;
  float b = NDotV * NDotV;
  return (2.0f * NDotV) / (NDotV + rt_sqrtf(a + b - a * b));
}

__device__ __forceinline__ rt_v3 cosine_hemisphere(uint32_t &rng) {           // driver.c:118-127
  float angle = rt_rand_f32(&rng) * 2.0f * RT_PI;
  float distance = rt_sqrtf(rt_rand_f32(&rng));
  float s, c;
  rt_sincosf(angle, &s, &c);
  return rt_v3_make(s * distance, c * distance, rt_sqrtf(1.0f - distance * distance));
}

__device__ __forceinline__ rt_v3 ggx_vndf(rt_v3 V, float ax, float ay, uint32_t &rng) {  // driver.c:230-250
  rt_v3 Vh = normalize_dev(rt_v3_make(ax * V.x, ay * V.y, V.z));
  float lensq = Vh.x * Vh.x + Vh.y * Vh.y;
  rt_v3 T1 = lensq > 0.0f ? rt_v3_scale(rt_v3_make(-Vh.y, Vh.x, 0.0f), rcp_exact(rt_sqrtf(lensq))) : rt_v3_make(1, 0, 0);
  rt_v3 T2 = rt_v3_cross(Vh, T1);
  float r = rt_sqrtf(rt_rand_f32(&rng));
  float phi = 2.0f * RT_PI * rt_rand_f32(&rng);
  float sn, cs;
  rt_sincosf(phi, &sn, &cs);
  float t1 = r * cs;
  float t2 = r * sn;
  float s = 0.5f * (1.0f + Vh.z);
  t2 = (1.0f - s) * rt_sqrtf(1.0f - t1 * t1) + s * t2;
  rt_v3 Nh = rt_v3_add(rt_v3_add(rt_v3_scale(T1, t1), rt_v3_scale(T2, t2)),
                       rt_v3_scale(Vh, rt_sqrtf(rt_max_ps(0.0f, 1.0f - t1 * t1 - t2 * t2))));
  return normalize_dev(rt_v3_make(ax * Nh.x, ay * Nh.y, rt_max_ps(0.0f, Nh.z)));
}

struct BrdfIn {
  float roughness, metalness, sheen, sheen_tint, aniso2;
  rt_v3 base_color;
};

// driver.c:287-348; returns the weight-pdf in brdf_a (<= 0: terminate)
__device__ __forceinline__ void sample_disney(const BrdfIn &m, rt_v3 in_dir, uint32_t &rng,
                                              rt_v3 &out_dir, rt_v3 &brdf_rgb, float &brdf_a) {
  float alpha_x = rt_lerpf(m.roughness * m.roughness, 1.0f, m.aniso2);
  float alpha_y = m.roughness * m.roughness;
  rt_v3 micro = ggx_vndf(in_dir, alpha_x, alpha_y, rng);

  rt_v3 f0 = rt_v3_lerp(rt_v3_make(0.04f, 0.04f, 0.04f), m.base_color, m.metalness);
  float f90 = rt_min_ps(1.0f, (1.0f / 0.04f) * luminance(f0));
  float theta = rt_v3_dot(in_dir, micro);
  rt_v3 fresnel = rt_v3_add(f0, rt_v3_scale(rt_v3_sub(rt_v3_make(f90, f90, f90), f0), pow5(1.0f - theta)));

  float dw = 1.0f - m.metalness;
  float sw = luminance(fresnel);
  float inv_w = 1.0f / (dw + sw);
  dw *= inv_w;
  sw *= inv_w;

  brdf_rgb = rt_v3_make(0, 0, 0);
  brdf_a = 0.0f;
  out_dir = rt_v3_make(0, 0, 0);
  if (rt_rand_f32(&rng) < dw) {
    out_dir = cosine_hemisphere(rng);
    micro = normalize_dev(rt_v3_add(out_dir, in_dir));
    float NoL = out_dir.z, NoV = in_dir.z;
    if (NoL <= 0.0f || NoV <= 0.0f) return;
    float LoH = rt_v3_dot(out_dir, micro);
    float pdf = NoL / RT_PI;
    float FD90 = 0.5f + 2.0f * m.roughness * LoH * LoH;
    float fa = 1.0f + (FD90 - 1.0f) * pow5(1.0f - NoL);
    float fb = 1.0f + (FD90 - 1.0f) * pow5(1.0f - NoV);
    rt_v3 diff = rt_v3_mul(rt_v3_scale(m.base_color, (fa * fb / RT_PI)), rt_v3_sub(rt_v3_make(1, 1, 1), fresnel));
    rt_v3 sheen = rt_v3_make(0, 0, 0);
    if (m.sheen > 0.0f) {                                                     // driver.c:166-183
      float lum = rt_v3_dot(rt_v3_make(0.3f, 0.6f, 1.0f), m.base_color);
      rt_v3 tint = (lum > 0.0f) ? rt_v3_scale(m.base_color, 1.0f / lum) : rt_v3_make(1, 1, 1);
      sheen = rt_v3_scale(rt_v3_lerp(rt_v3_make(1, 1, 1), tint, m.sheen_tint), m.sheen * pow5(1.0f - LoH));
    }
    diff = rt_v3_add(diff, sheen);
    brdf_rgb = rt_v3_make(diff.x * NoL, diff.y * NoL, diff.z * NoL);
    brdf_a = dw * pdf;
  } else {
    out_dir = rt_v3_reflect(rt_v3_scale(in_dir, -1.0f), micro);
    float NoL = out_dir.z, NoV = in_dir.z;
    if (NoL <= 0.0f || NoV <= 0.0f) return;
    NoL = rt_max_ps(NoL, 0.001f);
    NoV = rt_max_ps(NoV, 0.001f);
    float NoH = rt_min_ps(micro.z, 0.99f);
    float D = ggx_D(m.roughness, NoH);
    float G1 = smith_G(NoV, m.roughness * m.roughness);
    float pdf = (D * G1) / rt_max_ps(0.00001f, 4.0f * NoV);
    float a2 = m.roughness * m.roughness;
    float G = smith_G(NoV, a2) * smith_G(NoL, a2);
    rt_v3 spec = rt_v3_scale(fresnel, D * G / (4.0f * NoL * NoV));
    brdf_rgb = rt_v3_make(spec.x * NoL, spec.y * NoL, spec.z * NoL);
    brdf_a = sw * pdf;
  }
  out_dir = normalize_dev(out_dir);
}

struct ShadeIn {
  rt_v3 direction, normal, tangent, bitangent;
  float uvx, uvy;
};

// driver.c:129-153
template <class PT>
__device__ __forceinline__ rt_v3 normal_map(const PT &P, int tex, float strength, const ShadeIn &in) {
  rt_v3 normal = in.normal;
  if (tex >= 0) {
    rt_v3 v = tex_bilinear(P, tex, in.uvx, in.uvy);
    v = rt_v3_add(rt_v3_scale(v, 2.0f), rt_v3_make(-1.0f, -1.0f, -1.0f));
    v.y *= -1.0f;
    rt_v3 t = in.tangent, b = in.bitangent, n = in.normal;
    float s = strength;
    normal = normalize_dev(rt_v3_make(s * (v.x * t.x + v.y * b.x + v.z * n.x) + n.x * (1.0f - s),
                                        s * (v.x * t.y + v.y * b.y + v.z * n.y) + n.y * (1.0f - s),
                                        s * (v.x * t.z + v.y * b.z + v.z * n.z) + n.z * (1.0f - s)));
  }
  return normal;
}

// disney_shader_proc driver.c:350-409 / debug_shader_proc :411-418 on material `mat`
template <class PT>
__device__ __forceinline__ void shade(const PT &P, int mat, const ShadeIn &in, uint32_t &rng,
                                      rt_v3 &out_dir, rt_v3 &tint, rt_v3 &emission, bool &terminate,
                                      LaneCounters &cn) {
  const float *mb = P.mats + (size_t)mat * 20;
  float4 m0 = ld4(mb, 0), m1 = ld4(mb, 1), m2 = ld4(mb, 2), m3 = ld4(mb, 3), m4 = ld4(mb, 4);
  int tex_albedo = as_i(m3.x), tex_normal = as_i(m3.y), tex_mr = as_i(m3.z), tex_em = as_i(m3.w);
  int kind = as_i(m4.x);

  rt_v3 normal = normal_map(P, tex_normal, m2.x, in);
  terminate = false;
  tint = rt_v3_make(0, 0, 0);
  out_dir = rt_v3_make(0, 0, 0);

  if (kind == RT_MAT_DEBUG) {
    emission = rt_v3_add(rt_v3_scale(normal, 0.5f), rt_v3_make(0.5f, 0.5f, 0.5f));
    terminate = true;
    return;
  }

  if (tex_albedo >= 0 || tex_normal >= 0 || tex_mr >= 0 || tex_em >= 0) cn.textured += 1;

  rt_v3 base_color = rt_v3_make(m0.x, m0.y, m0.z);
  if (tex_albedo >= 0) base_color = rt_v3_mul(base_color, srgb_to_linear_tex(tex_bilinear(P, tex_albedo, in.uvx, in.uvy)));

  float roughness = m0.w, metalness = m1.w;
  if (tex_mr >= 0) {
    rt_v3 mr = tex_bilinear(P, tex_mr, in.uvx, in.uvy);
    roughness *= mr.y;
    metalness *= mr.z;
  }
  roughness = rt_clampf(roughness, 0.001f, 1.0f);
  if (metalness > 0.9f) metalness = 0.9f;
  metalness /= 0.9f;

  emission = rt_v3_make(m1.x, m1.y, m1.z);
  if (tex_em >= 0) emission = rt_v3_mul(emission, srgb_to_linear_tex(tex_bilinear(P, tex_em, in.uvx, in.uvy)));

  // basis(), driver.c:155-164
  rt_v3 t, b;
  if (rt_absf(rt_v3_dot(normal, in.direction)) < 0.9999f) {
    t = normalize_dev(rt_v3_cross(normal, in.direction));
  } else if (rt_absf(rt_v3_dot(normal, rt_v3_make(0, 1, 0))) < 0.9999f) {
    t = normalize_dev(rt_v3_cross(normal, rt_v3_make(0, 1, 0)));
  } else {
    t = normalize_dev(rt_v3_cross(normal, rt_v3_make(1, 0, 0)));
  }
  b = rt_v3_cross(normal, t);

  BrdfIn bi;
  bi.roughness = roughness;
  bi.metalness = metalness;
  bi.base_color = base_color;
  bi.sheen = m2.y;
  bi.sheen_tint = m2.z;
  bi.aniso2 = m2.w * m2.w;

  rt_v3 neg = rt_v3_scale(in.direction, -1.0f);
  rt_v3 in_dir = rt_v3_make(rt_v3_dot(t, neg), rt_v3_dot(b, neg), rt_v3_dot(normal, neg));
  rt_v3 o, rgb;
  float a;
  sample_disney(bi, in_dir, rng, o, rgb, a);

  out_dir = rt_v3_make(t.x * o.x + b.x * o.y + normal.x * o.z,
                       t.y * o.x + b.y * o.y + normal.y * o.z,
                       t.z * o.x + b.z * o.y + normal.z * o.z);
  if (a > 0.0f) {
    tint = rt_v3_make(rgb.x / a, rgb.y / a, rgb.z / a);
  } else {
    terminate = true;
  }
}

// ---------------------------------------------------------------------------------
// primary ray of (x, y, sample): raytracer.c:641-694 with exact 1/sqrt
template <class PT>
__device__ __forceinline__ void primary_ray(const PT &P, int x, int y, int sample, rt_v3 &o, rt_v3 &d) {
  // 1/width, 1/height, width/height (raytracer.c:615-617) are frame constants: the host computes the same
  // three fp32 divisions once (rt_api.cpp) instead of every lane for every path
  float inv_width = P.inv_width;
  float inv_height = P.inv_height;
  float aspect = P.aspect;
  float jitter = rt_hash12((float)x * 50.0f + (float)sample, (float)y);
  float uvx = ((float)x + jitter - 0.5f) * 2.0f * inv_width - 1.0f;
  float uvy = ((float)y + jitter - 0.5f) * 2.0f * inv_height - 1.0f;
  float dx = uvx * aspect, dy = -uvy, dz = -P.focal_length;
  // (rcp_exact: a square root lies in [0, 2^64] or is infinite / NaN -- inside the domain on which it equals the division)
  float inv_length = rcp_exact(rt_sqrtf(dx * dx + dy * dy + dz * dz));
  float rx = P.cam[0][0] * dx + P.cam[0][1] * dy + P.cam[0][2] * dz;
  float ry = P.cam[1][0] * dx + P.cam[1][1] * dy + P.cam[1][2] * dz;
  float rz = P.cam[2][0] * dx + P.cam[2][1] * dy + P.cam[2][2] * dz;
  o = rt_v3_make(P.cam[0][3], P.cam[1][3], P.cam[2][3]);
  d = rt_v3_make(rx * inv_length, ry * inv_length, rz * inv_length);
}

// SHORT_DIV: the reciprocals by rcp_exact() -- same bits as the division for |component| < 2^102, infinity and NaN.  The
// tile-stream kernel uses it where the host has bounded the camera matrix (rt_api.cpp): a camera direction is a unit
// vector through that matrix, every other direction comes out of shade() as t o.x + b o.y + n o.z of normalised vectors
// (components within +-3.1, or infinite / NaN when a normalisation met a zero or non-finite vector).
template <bool SHORT_DIV = false>
__device__ __forceinline__ void ray_setup(Ray3 &r, rt_v3 o, rt_v3 d) {
  r.o = o;
  r.d = d;
  if (SHORT_DIV) {
    r.inv_x = rcp_exact(d.x);
    r.inv_y = rcp_exact(d.y);
    r.inv_z = rcp_exact(d.z);
  } else {
    r.inv_x = 1.0f / d.x;       // raytracer.c:198-202
    r.inv_y = 1.0f / d.y;
    r.inv_z = 1.0f / d.z;
  }
  r.fast = (rt_absf(r.inv_x) < RT_INF) && (rt_absf(r.inv_y) < RT_INF) && (rt_absf(r.inv_z) < RT_INF) &&
           (rt_absf(o.x) < RT_INF) && (rt_absf(o.y) < RT_INF) && (rt_absf(o.z) < RT_INF);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// One accepted closest hit -> pass-through or material evaluation and the next ray of the path
// (raytracer.c:515-552).  Returns true when the path ended (radiance holds its value).
template <class PT>
__device__ __forceinline__ bool shade_hit(const PT &P, const HitRec &hit, rt_v3 &org, rt_v3 &dir,
                                          rt_v3 &tint, rt_v3 &emis, uint32_t &rng, int &bounce,
                                          LaneCounters &cn, rt_v3 &radiance) {
  bool done = false;
  const float *tb = P.tris + (size_t)hit.tri * 28;
  float4 q0 = ld4(tb, 0), q1 = ld4(tb, 1), q2 = ld4(tb, 2), q3 = ld4(tb, 3);
  float4 q4 = ld4(tb, 4), q5 = ld4(tb, 5), q6 = ld4(tb, 6);
  float t1 = hit.u, t2 = hit.v;
  float t0 = 1.0f - t1 - t2;
  rt_v3 point = rt_v3_add(org, rt_v3_scale(dir, hit.t));
  rt_v3 n_geo = rt_v3_make(q0.x, q0.y, q0.z);
  rt_v3 n_int = rt_v3_make(q1.x * t0 + q2.x * t1 + q3.x * t2,
                           q1.y * t0 + q2.y * t1 + q3.y * t2,
                           q1.z * t0 + q2.z * t1 + q3.z * t2);
  if (rt_v3_dot(n_geo, dir) > 0.0f || rt_v3_dot(n_int, dir) > 0.0f) {
    // back face: pass through, costs a bounce (raytracer.c:516-522)
    org = rt_v3_add(point, rt_v3_scale(dir, RT_EPS));
  } else {
    ShadeIn in;
    in.direction = dir;
    in.normal = normalize_dev(n_int);
    in.tangent = rt_v3_make(q4.x, q4.y, q4.z);
    in.bitangent = rt_v3_make(q5.x, q5.y, q5.z);
    in.uvx = q1.w * t0 + q3.w * t1 + q5.w * t2;
    in.uvy = q2.w * t0 + q4.w * t1 + q6.x * t2;
    rt_v3 out_dir, s_tint, s_emis;
    bool terminate;
    cn.shades += 1;
    shade(P, as_i(q0.w), in, rng, out_dir, s_tint, s_emis, terminate, cn);
    emis = rt_v3_add(emis, rt_v3_mul(s_emis, tint));
    if (terminate) {
      done = true;
      radiance = emis;
    } else {
      dir = out_dir;
      tint = rt_v3_mul(tint, s_tint);
      float below = (rt_v3_dot(n_geo, out_dir) < 0.0f) ? 1.0f : 0.0f;
      float bias = (0.5f - below) * 2.0f * RT_EPS;
      org = rt_v3_add(point, rt_v3_scale(n_geo, bias));
    }
  }
  if (!done) {
    bounce += 1;
    if (bounce >= P.max_bounces) {     // bounces exhausted: emission only (raytracer.c:557)
      done = true;
      radiance = emis;
    }
  }
  return done;
}

// ---------------------------------------------------------------------------------
// The path-tracing kernel.  Persistent: the grid is sized to the machine, each
// wave loops over work items until the head counter runs past n_work.
__global__ __launch_bounds__(RT_BLOCK_THREADS) void rt_path_kernel(RT_KParams P) {
  __shared__ uint32_t s_perm[RT_BLOCK_WAVES][RT_MAX_DEPTH * 64];
  __shared__ unsigned long long s_acc[RT_BLOCK_WAVES][RT_TILE_PIX * 3];

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  uint32_t *perm = s_perm[wave];
  unsigned long long *acc = s_acc[wave];

  LaneCounters cn;
  cn.rays = cn.nodes = cn.leaves = cn.shades = cn.bgs = cn.textured = cn.paths = 0;

  acc[lane] = 0ull;
  acc[lane + 64] = 0ull;
  acc[lane + 128] = 0ull;

  const int slab = 1 << P.slab_shift;
  const int item_paths = RT_TILE_PIX << P.slab_shift;

  for (;;) {
    // ---- dequeue one work item (wave-uniform) ----
    uint32_t w = 0;
    if (lane == 0) w = atomicAdd(P.work_head, 1u);
    w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
    if (w >= (uint32_t)P.n_work) break;

    // item -> (local chunk, 8x8 tile inside the chunk, slab of samples)
    const int slab_idx = (int)(w % (uint32_t)P.n_slabs);
    const int tile_idx = (int)(w / (uint32_t)P.n_slabs);
    const int lchunk = tile_idx >> 4, sub = tile_idx & 15;
    const int chunk = P.local_chunks[lchunk];
    const int tile_x0 = (chunk % P.chunks_x) * 32 + (sub & 3) * 8;
    const int tile_y0 = (chunk / P.chunks_x) * 32 + (sub >> 2) * 8;
    if (tile_x0 >= P.width || tile_y0 >= P.height) continue;   // tile entirely outside
    const int s_base = slab_idx << P.slab_shift;

    // ---- path state ----
    bool  alive = false;
    int   pix = 0, bounce = 0;
    uint32_t rng = 0;
    rt_v3 org = rt_v3_make(0, 0, 0), dir = rt_v3_make(0, 0, 1);
    rt_v3 tint = rt_v3_make(1, 1, 1), emis = rt_v3_make(0, 0, 0);
    int next_k = 0;      // wave-uniform

    for (;;) {
      // ---- regenerate: dead lanes take the next (pixel, sample) of the item ----
      if (next_k < item_paths) {
        unsigned long long need = __ballot(!alive);
        if (need) {
          int my_k = next_k + (int)__popcll(need & ((1ull << lane) - 1ull));
          next_k += (int)__popcll(need);
          if (!alive && my_k < item_paths) {
            int p = my_k >> P.slab_shift;                                 // pixel-major
            int s = P.sample_first + s_base + (my_k & (slab - 1));
            int x = tile_x0 + (p & 7), y = tile_y0 + (p >> 3);
            if (s < P.sample_end && x < P.width && y < P.height && P.max_bounces > 0) {
              alive = true;
              pix = p;
              bounce = 0;
              rng = rt_path_seed(P.seed, (uint32_t)(x + y * P.width), (uint32_t)s);
              primary_ray(P, x, y, s, org, dir);
              tint = rt_v3_make(1, 1, 1);
              emis = rt_v3_make(0, 0, 0);
              cn.paths += 1;
            } else if (s < P.sample_end && x < P.width && y < P.height) {
              cn.paths += 1;      // max_bounces == 0: the path exists and is black (the loop of raytracer.c:512 runs zero times)
            }
          }
        }
      }
      if (!__any(alive)) {
        if (next_k >= item_paths) break;
        continue;
      }

      // ---- extend: closest hit of every live path ----
      HitRec hit;
      hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;
      Ray3 ray;
      ray_setup(ray, org, dir);
      // wave-uniform choice of the slab code path (see "Slab tests" above)
      if (__all(!alive || ray.fast)) {
        if (alive) trace_ray<true>(P, ray, hit, perm, lane, cn);
      } else {
        if (alive) trace_ray<false>(P, ray, hit, perm, lane, cn);
      }

      // ---- shade / environment ----
      bool  done = false;
      rt_v3 radiance = rt_v3_make(0, 0, 0);
      if (alive) {
        if (hit.tri >= 0) {
          done = shade_hit(P, hit, org, dir, tint, emis, rng, bounce, cn, radiance);
        } else {
          cn.bgs += 1;
          rt_v3 bg = background_lookup(P, dir);
          radiance = rt_v3_add(rt_v3_mul(bg, tint), emis);
          done = true;
        }
      }
      if (done) {
        atomicAdd(&acc[pix * 3 + 0], (unsigned long long)rt_accum_quantize(radiance.x));
        atomicAdd(&acc[pix * 3 + 1], (unsigned long long)rt_accum_quantize(radiance.y));
        atomicAdd(&acc[pix * 3 + 2], (unsigned long long)rt_accum_quantize(radiance.z));
        alive = false;
      }
    }

    // ---- flush the tile: lane p owns pixel p ----
    {
      int x = tile_x0 + (lane & 7), y = tile_y0 + (lane >> 3);
      unsigned long long r = acc[lane * 3 + 0], g = acc[lane * 3 + 1], b = acc[lane * 3 + 2];
      acc[lane * 3 + 0] = 0ull;
      acc[lane * 3 + 1] = 0ull;
      acc[lane * 3 + 2] = 0ull;
      if (x < P.width && y < P.height) {
        unsigned long long *dst = P.accum + ((size_t)y * P.width + x) * 3;
        atomicAdd(dst + 0, r);
        atomicAdd(dst + 1, g);
        atomicAdd(dst + 2, b);
      }
    }
  }

  // ---- counters: one atomic per wave and counter ----
  uint32_t c0 = wave_sum(cn.paths), c1 = wave_sum(cn.rays), c2 = wave_sum(cn.nodes), c3 = wave_sum(cn.leaves);
  uint32_t c4 = wave_sum(cn.shades), c5 = wave_sum(cn.bgs), c6 = wave_sum(cn.textured);
  if (lane == 0) {
    atomicAdd(P.counters + CNT_PATHS, (unsigned long long)c0);
    atomicAdd(P.counters + CNT_RAYS, (unsigned long long)c1);
    atomicAdd(P.counters + CNT_NODES, (unsigned long long)c2);
    atomicAdd(P.counters + CNT_LEAVES, (unsigned long long)c3);
    atomicAdd(P.counters + CNT_SHADES, (unsigned long long)c4);
    atomicAdd(P.counters + CNT_BG, (unsigned long long)c5);
    atomicAdd(P.counters + CNT_TEXTURED, (unsigned long long)c6);
  }
}

// ---------------------------------------------------------------------------------
// The scheduled path kernel.  Same work items, same per-lane arithmetic as
// rt_path_kernel, different control: every lane carries a phase and the wave picks, per
// iteration, ONE block of code to run for all lanes that wait for it:
//
//   NODE  enter a BVH node (8 slab tests + rank sort)          \ the larger group of the two
//   LEAF  test the 8 triangles of a leaf group                 /  runs, the other one waits
//   S     shade hits, look up the environment for misses, start new camera paths --
//         run when at least `sched_thresh` lanes wait for it (or nothing else is runnable)
//
// so a traversal that takes long no longer parks the lanes that already finished (they are
// shaded / regenerated once enough of them wait), and node and leaf code each run on a dense
// set of lanes instead of splitting every iteration between them.  Traversal state (level,
// node, perm word, dirty mask, closest hit) simply persists in registers between blocks.
#define PH_NEED 0     // no path: wants a new (pixel, sample)
#define PH_POP  1     // traversal: take the next child of the current node (transient)
#define PH_NODE 2     // traversal: wants node_enter(child)
#define PH_LEAF 3     // traversal: wants leaf_test(child)
#define PH_HIT  4     // traversal finished with a hit: wants shading
#define PH_MISS 5     // traversal finished without a hit: wants the environment

template <int WAVES, bool LDSN, bool STATS, int MIN_WAVES_PER_SIMD = 1>
__global__ __launch_bounds__(WAVES * 64, MIN_WAVES_PER_SIMD) void rt_path_kernel_sched(RT_KParams P) {
  // dynamic LDS: [ top of the BVH, n_lds_nodes x 13 float4 (LDSN only) ][ per wave: perm stack, depth x 64 u32 |
  //               accumulator tile, 64 pixels x 3 x u64 ]
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_lds = LDSN ? P.n_lds_nodes : 0;
  const float4 *lds_nodes = smem;
  const int perm_f4 = (P.depth > 0 ? P.depth : 1) * 16;
  float4 *wave_base = smem + n_lds * RT_LDS_NODE_F4 + wave * (perm_f4 + 96);
  uint32_t *perm = reinterpret_cast<uint32_t *>(wave_base);
  unsigned long long *acc = reinterpret_cast<unsigned long long *>(wave_base + perm_f4);

  if (LDSN) {                 // the workgroup copies the first n_lds nodes (level order = top of the tree) once
    const float4 *g = reinterpret_cast<const float4 *>(P.nodes);
    for (int i = threadIdx.x; i < n_lds * 12; i += WAVES * 64) {
      int nd = i / 12, q = i - nd * 12;
      smem[nd * RT_LDS_NODE_F4 + q] = g[i];
    }
    __syncthreads();          // the only workgroup barrier of the kernel; waves are independent afterwards
  }

  LaneCounters cn;
  cn.rays = cn.nodes = cn.leaves = cn.shades = cn.bgs = cn.textured = cn.paths = 0;

  acc[lane] = 0ull;
  acc[lane + 64] = 0ull;
  acc[lane + 128] = 0ull;

  const int slab = 1 << P.slab_shift;
  const int item_paths = RT_TILE_PIX << P.slab_shift;
  const int leaf_level = P.depth - 1;
  const int thresh = P.sched_thresh;
  // diagnostic build only (STATS): how often each block ran and with how many lanes; wave-uniform
  const unsigned long long t_wave_start = STATS ? __builtin_amdgcn_s_memrealtime() : 0ull;   // 100 MHz wall clock
  uint32_t n_items_done = 0;
  uint32_t st[16];
#pragma unroll
  for (int i = 0; i < 16; i++) st[i] = 0;
#define STAT(slot, lanes) do { if (STATS) { st[2 * (slot)] += 1; st[2 * (slot) + 1] += (uint32_t)(lanes); } } while (0)
  // ... and the shader-clock cycles the wave spent in each kind of block (wall time of the wave, other waves' issue included)
  unsigned long long cyc[8];
#pragma unroll
  for (int i = 0; i < 8; i++) cyc[i] = 0ull;
  unsigned long long t_blk = 0ull;
  const unsigned long long t_loop0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
#define CYC_BEGIN() do { if (STATS) t_blk = __builtin_amdgcn_s_memtime(); } while (0)
#define CYC_END(slot) do { if (STATS) cyc[slot] += __builtin_amdgcn_s_memtime() - t_blk; } while (0)

  for (;;) {
    // ---- dequeue one work item (wave-uniform).  Items are small (8x8 pixels x 16 samples by default):
    //      measured, the frame time is set by how evenly the LAST items spread over the 4096 waves, not
    //      by the bubble at the end of each item (keeping two items in flight per wave bought nothing
    //      and cost 40 VGPRs) ----
    uint32_t w = 0;
    if (lane == 0) w = atomicAdd(P.work_head, 1u);
    w = (uint32_t)__builtin_amdgcn_readfirstlane((int)w);
    if (w >= (uint32_t)P.n_work) break;

    const int slab_idx = (int)(w % (uint32_t)P.n_slabs);
    // tiles are visited in the order the host prepared: most expensive first (cost = rays the tile needed in
    // the previous launch of this view), so that the last items of the launch are cheap ones
    const int tile_pos = (int)(w / (uint32_t)P.n_slabs);
    const int tile_idx = P.order ? (int)P.order[tile_pos] : tile_pos;
    const int lchunk = tile_idx >> 4, sub = tile_idx & 15;
    const int chunk = P.local_chunks[lchunk];
    const int tile_x0 = (chunk % P.chunks_x) * 32 + (sub & 3) * 8;
    const int tile_y0 = (chunk / P.chunks_x) * 32 + (sub >> 2) * 8;
    if (tile_x0 >= P.width || tile_y0 >= P.height) continue;
    const int s_base = slab_idx << P.slab_shift;
    const uint32_t rays_before = cn.rays;

    // ---- per-lane state ----
    int   phase = PH_NEED;
    int   pix = 0, bounce = 0;
    uint32_t rng = 0;
    Ray3  ray;
    ray_setup(ray, rt_v3_make(0, 0, 0), rt_v3_make(0, 0, 1));
    rt_v3 tint = rt_v3_make(1, 1, 1), emis = rt_v3_make(0, 0, 0);
    int   level = -1, node = 0, child = 0;
    uint32_t cur = 0, dirty = 0, live = 0;     // live: bit L set <=> the perm word stored for level L still has children
    HitRec hit;
    hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;
    int next_k = 0;      // wave-uniform

    for (;;) {
      const bool can_regen = next_k < item_paths;
      const int nN = (int)__popcll(__ballot(phase == PH_NODE));
      const int nL = (int)__popcll(__ballot(phase == PH_LEAF));
      const int nH = (int)__popcll(__ballot(phase == PH_HIT));
      const int nE = (int)__popcll(__ballot(phase == PH_MISS || (can_regen && phase == PH_NEED)));
      // (no lane is between blocks here: the pop loop at the end of an iteration runs until every lane has its next block)
      if (nN + nL + nH + nE == 0) break;          // every lane idle and the item has no paths left

      // Block choice: ONE combined block -- shade the hits, look up the environment for the misses, start new paths --
      // once `thresh` lanes wait for any of that, or when nothing is traversing.  (Separate thresholds for shading
      // and for environment + regeneration were measured and lost by 2-4 %.)
      const bool both = (nH + nE >= thresh) || (nN + nL == 0);
      const bool run_shade = both && nH > 0;
      const bool run_env = both && nE > 0;

      if (run_shade || run_env) {
        CYC_BEGIN();
        if (run_shade) STAT(0, nH);
        if (run_env) STAT(1, __popcll(__ballot(phase == PH_MISS)));
        if (run_env && can_regen) STAT(2, __popcll(__ballot(phase == PH_NEED)));
        bool  done = false, start = false;
        rt_v3 radiance = rt_v3_make(0, 0, 0);
        rt_v3 org = ray.o, dir = ray.d;
        if (run_shade && phase == PH_HIT) {
          // ================= SHADE: material evaluation of the closest hits =================
          done = shade_hit(P, hit, org, dir, tint, emis, rng, bounce, cn, radiance);
          start = !done;
        } else if (run_env && phase == PH_MISS) {
          // ================= ENV: environment for the misses =================
          cn.bgs += 1;
          rt_v3 bg = background_lookup(P, dir);
          radiance = rt_v3_add(rt_v3_mul(bg, tint), emis);
          done = true;
        }
        if (done) {
          atomicAdd(&acc[pix * 3 + 0], (unsigned long long)rt_accum_quantize(radiance.x));
          atomicAdd(&acc[pix * 3 + 1], (unsigned long long)rt_accum_quantize(radiance.y));
          atomicAdd(&acc[pix * 3 + 2], (unsigned long long)rt_accum_quantize(radiance.z));
          phase = PH_NEED;
        }
        if (run_env && can_regen) {
          // ================= REGEN: idle lanes take the next (pixel, sample) of the item =================
          unsigned long long need = __ballot(phase == PH_NEED);
          if (need) {
            int my_k = next_k + (int)__popcll(need & ((1ull << lane) - 1ull));
            next_k += (int)__popcll(need);
            if (phase == PH_NEED && my_k < item_paths) {
              // k -> (pixel of the tile, sample of the slab), pixel-major: the lanes of a wave stay on a few pixels
              // (sample-major, spreading them over the 64 pixels of the tile, was measured and is slower at every slab)
              int p = my_k >> P.slab_shift;
              int s = P.sample_first + s_base + (my_k & (slab - 1));
              int x = tile_x0 + (p & 7), y = tile_y0 + (p >> 3);
              if (s < P.sample_end && x < P.width && y < P.height && P.max_bounces > 0) {
                pix = p;
                bounce = 0;
                rng = rt_path_seed(P.seed, (uint32_t)(x + y * P.width), (uint32_t)s);
                primary_ray(P, x, y, s, org, dir);
                tint = rt_v3_make(1, 1, 1);
                emis = rt_v3_make(0, 0, 0);
                cn.paths += 1;
                start = true;
              } else if (s < P.sample_end && x < P.width && y < P.height) {
                cn.paths += 1;    // max_bounces == 0: the path exists and is black (the loop of raytracer.c:512 runs zero times)
              }
            }
          }
        }
        if (start) {                      // a new ray: traversal starts at the root (or at leaf group 0)
          ray_setup(ray, org, dir);
          hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;
          cn.rays += 1;
          dirty = 0;
          live = 0;
          cur = 0;
          level = -1;
          node = 0;
          child = (P.depth > 0) ? 0 : P.last_row_offset;
          phase = (P.depth > 0) ? PH_NODE : PH_LEAF;
        }
        CYC_END(run_shade ? 0 : 1);
        continue;
      }

    if (nN + nL == 0) {
        // only lanes between blocks: fall through to the pop loop
      } else if (nL >= nN) {
        // ================= LEAF =================
        CYC_BEGIN();
        if (phase == PH_LEAF) {
          cn.leaves += 1;
          int  g = child - P.last_row_offset;
          // per-lane vector loads also when all lanes are on one leaf (same-address loads are one cache line each):
          // measured 0.5 % faster than bringing the 288-byte tile through 72 SGPRs, and it keeps them free
          STAT(4, nL);
          bool got = leaf_test<false>(P, ray, g, hit);
          if (got) dirty = 0xFFFFFFFFu;
          phase = PH_POP;
        }
        CYC_END(3);
      } else {
        // ================= NODE =================
        CYC_BEGIN();
        const bool all_fast = __ballot(phase == PH_NODE && !ray.fast) == 0;
        if (phase == PH_NODE) {
          if (level >= 0) {
            perm[level * 64 + lane] = cur;
            live = (cur >> 24) ? (live | (1u << level)) : (live & ~(1u << level));
          }
          node = child;
          level += 1;
          cn.nodes += 1;
          if (all_fast) {
            // nodes of the LDS copy are read from LDS, also when all lanes want the same one (a broadcast read); nodes
            // outside the copy come through L1/L2.  (A scalar-cache path for wave-uniform nodes was measured: slower.)
            if (LDSN && __ballot(node >= n_lds) == 0) { STAT(6, nN); cur = node_enter<true, NODE_LDS>(P, ray, node, hit.t, lds_nodes); }
            else { STAT(5, nN); cur = node_enter<true, NODE_GLOBAL>(P, ray, node, hit.t, lds_nodes); }
          } else {
            cur = node_enter<false, NODE_GLOBAL>(P, ray, node, hit.t, lds_nodes);
          }
          dirty &= ~(1u << level);
          // the nearest child is taken right here, on the dense set of lanes of this block (its distance was
          // just compared with hit.t); only a node without candidates sends the lane to the pop loop
          if (cur >> 24) {
            child = 8 * node + 1 + (int)(cur & 7u);
            cur = ((cur >> 3) & 0x1FFFFFu) | (((cur >> 24) - 1u) << 24);
            phase = (level == leaf_level) ? PH_LEAF : PH_NODE;
          } else {
            phase = PH_POP;
          }
        }
        CYC_END(5);
      }

      // ---- pops: every lane that just finished a block takes its next child / goes up until it knows its next block
      //      (bounding the rounds per iteration and letting lanes wait in PH_POP was measured: 1 round 65.7 ms, 4 rounds
      //      57.1 ms, unbounded 56.8 ms) ----
      CYC_BEGIN();
      while (__any(phase == PH_POP)) {
        STAT(7, __popcll(__ballot(phase == PH_POP)));
        if (phase == PH_POP) {
          uint32_t cnt = cur >> 24;
          if (cnt == 0 || level < 0) {
            // go up to the nearest level that still has children to visit -- in one step: the k-th ancestor of
            // node n in the implicit 8-ary tree is (n - (8^k - 1)/7) >> 3k, and (8^k - 1)/7 is k ones 3 bits apart
            uint32_t above = (level > 0) ? (live & ((1u << level) - 1u)) : 0u;
            if (above == 0u) {
              level = -1;
              phase = (hit.tri >= 0) ? PH_HIT : PH_MISS;
            } else {
              int target = 31 - __clz((int)above);
              int k3 = 3 * (level - target);
              node = (int)(((uint32_t)node - (0x09249249u & ((1u << k3) - 1u))) >> k3);
              level = target;
              cur = perm[level * 64 + lane];
              cnt = cur >> 24;                  // > 0: the level is marked live
            }
          }
          if (phase == PH_POP) {                // same round: take the next child of the (possibly new) level
            int j = (int)(cur & 7u);
            cur = ((cur >> 3) & 0x1FFFFFu) | ((cnt - 1u) << 24);
            bool go = true;
            if ((dirty >> level) & 1u) {
              float dj;
              if (LDSN && node < n_lds) dj = slab_entry_child<false>(reinterpret_cast<const float *>(lds_nodes + lds_node_f4(node)) + j, ray);
              else dj = slab_entry_child<false>(P.nodes + (size_t)node * 48 + j, ray);
              if (!(dj < hit.t)) { cur = 0; go = false; }      // raytracer.c:470-472
            }
            if (go) {
              child = 8 * node + 1 + j;
              phase = (level == leaf_level) ? PH_LEAF : PH_NODE;
            }
          }
        }
      }
      CYC_END(7);
    }

    // ---- flush the tile: lane p owns pixel p ----
    {
      int x = tile_x0 + (lane & 7), y = tile_y0 + (lane >> 3);
      unsigned long long r = acc[lane * 3 + 0], g = acc[lane * 3 + 1], b = acc[lane * 3 + 2];
      acc[lane * 3 + 0] = 0ull;
      acc[lane * 3 + 1] = 0ull;
      acc[lane * 3 + 2] = 0ull;
      if (x < P.width && y < P.height) {
        unsigned long long *dst = P.accum + ((size_t)y * P.width + x) * 3;
        atomicAdd(dst + 0, r);
        atomicAdd(dst + 1, g);
        atomicAdd(dst + 2, b);
      }
    }
    if (STATS) n_items_done += 1;
    if (P.tile_cost) {          // rays this item needed: the next launch of the same view schedules by it
      uint32_t r = wave_sum(cn.rays - rays_before);
      if (lane == 0) atomicAdd(&P.tile_cost[tile_idx], r);
    }
  }

  uint32_t c0 = wave_sum(cn.paths), c1 = wave_sum(cn.rays), c2 = wave_sum(cn.nodes), c3 = wave_sum(cn.leaves);
  uint32_t c4 = wave_sum(cn.shades), c5 = wave_sum(cn.bgs), c6 = wave_sum(cn.textured);
  if (lane == 0) {
    atomicAdd(P.counters + CNT_PATHS, (unsigned long long)c0);
    atomicAdd(P.counters + CNT_RAYS, (unsigned long long)c1);
    atomicAdd(P.counters + CNT_NODES, (unsigned long long)c2);
    atomicAdd(P.counters + CNT_LEAVES, (unsigned long long)c3);
    atomicAdd(P.counters + CNT_SHADES, (unsigned long long)c4);
    atomicAdd(P.counters + CNT_BG, (unsigned long long)c5);
    atomicAdd(P.counters + CNT_TEXTURED, (unsigned long long)c6);
    if (STATS) {
#pragma unroll
      for (int i = 0; i < 16; i++) atomicAdd(P.counters + 8 + i, (unsigned long long)st[i]);
#pragma unroll
      for (int i = 0; i < 8; i++) atomicAdd(P.counters + 24 + i, cyc[i]);
      atomicAdd(P.counters + 32, __builtin_amdgcn_s_memtime() - t_loop0);
      if (P.wave_times) {
        int wid = blockIdx.x * WAVES + wave;
        P.wave_times[wid * 3 + 0] = t_wave_start;
        P.wave_times[wid * 3 + 1] = __builtin_amdgcn_s_memrealtime();
        P.wave_times[wid * 3 + 2] = n_items_done;
      }
    }
  }
#undef STAT
#undef CYC_BEGIN
#undef CYC_END
}

// ---------------------------------------------------------------------------------
// The tile-stream path kernel (default).  Same blocks and the same per-lane arithmetic as rt_path_kernel_sched;
// what changes is where the work comes from and how the loop is cut:
//
//  * A wave OWNS an 8x8-pixel tile (taken from the head counter, expensive tiles first) and pulls UNITS of it --
//    2 neighbouring pixels x 2^chunk_shift samples (128 paths at 64 samples, pixel-major, so the 64 lanes
//    sit on one or two pixels: coherent nodes, leaves and texels), one or two per atomic -- from the tile's own counter
//    `tile_next[tile]` until the tile is exhausted.  Lanes whose path ended are refilled across unit boundaries, so
//    there is no end-of-item drain (the scheduled kernel drains the wave at the end of every item, 40 us to 0.4 ms each);
//    the wave drains once per TILE.
//  * When the head counter runs dry a wave JOINS a tile that still has units (two-level scan with agent-scope loads:
//    `open_groups[g]` counts the open tiles of every group of 64) and pulls from the same counter: the tail of a
//    launch is balanced at unit granularity even when a rank of the 8-GPU partition has fewer tiles than the chip has
//    waves.  Every unit is handed out exactly once by an atomic; radiance sums are order-free integers, so results do
//    not depend on who traced what.  Owners always finish their tile, so a joiner may give up at any time: every wave
//    reaches an exit (bounded scans), there is no inter-wave dependency and no grid barrier.
//  * Camera rays of a tile whose pixel pyramid misses every child box of the root skip the root block (below); node blocks
//    whose lanes are (mostly) camera rays about to enter ONE node test only the child boxes that pyramid can touch
//    (pyramid_cull_mask, node_enter_few; the mask of a (tile, node) is cached in LDS).
//  * Hits are parked -- path state to memory, lane to a new path -- while a shade block would run sparse, and shaded
//    together when 48 lanes can be filled (`park`, S block).
//  * The traversal blocks run in an inner loop of their own; shading / environment / regeneration run in the outer
//    loop.  Path state (tint, emission, RNG, pixel) is untouched inside the inner loop, the block choice there is
//    two ballots, and the counters are wave-level scalars.
#define RT_PARK_FIELDS 18     // hit (t, triangle, u, v), ray origin and direction, tint, emission, rng, pixel of the tile | bounce << 6
#define RT_PARK_CAP 128       // parked hits per wave (fewer than RT_PARK_DENSE + 64 are ever parked)
#ifndef RT_PARK_DENSE
#define RT_PARK_DENSE 48      // lanes that make a shade block worth running while the tile still hands out paths
#endif
#ifndef RT_JOIN_CHOICES
#define RT_JOIN_CHOICES 4    // random open tiles a joining wave looks at; it takes the one with most units left
#endif
#ifndef RT_PYR_NUM
#define RT_PYR_NUM 3       // a pyramid-culled node block needs nG >= nN * RT_PYR_NUM / RT_PYR_DEN camera rays on one node
#define RT_PYR_DEN 4
#define RT_PYR_MIN 8
#endif
#define RT_STEAL_TRIES  16        // failed joins in a row before a wave retires

// Kernel arguments that are only needed outside the traversal loop (camera, frame and tile bookkeeping, material
// tables) are read from the kernarg segment WHERE they are used, through a pointer the compiler cannot see through:
// kept live across the traversal loop they cost ~50 scalar registers, and the spills of those (to VGPR lanes, then
// VGPRs to scratch) were measured at +3 % frame time.  A scalar load per use in the shade / regenerate block is free
// by comparison (that block runs once per ~4.6 traversal blocks and is several hundred instructions long).
// A wave-uniform LDS byte offset turned into a pointer WHERE it is used (the empty asm keeps the compiler from forming
// the address once, holding it in a VGPR across the loops and spilling it to scratch), and the lane index recomputed
// (v_mbcnt) instead of kept.
__device__ __forceinline__ float *lds_at(float4 *smem, int byte_off) {
  asm volatile("" : "+s"(byte_off));
  return reinterpret_cast<float *>(reinterpret_cast<char *>(smem) + byte_off);
}
__device__ __forceinline__ int lane_now() {
  unsigned ones = ~0u;
  asm volatile("" : "+s"(ones));
  return (int)__builtin_amdgcn_mbcnt_hi(ones, __builtin_amdgcn_mbcnt_lo(ones, 0u));
}

// ---- pyramid culling of node blocks (tile-stream kernel) ----
// `pyr` (LDS, per wave): outward normals of the four side planes of the tile's camera-ray pyramid at [4 q .. 4 q + 2],
// the common ray origin at [16 .. 18].  Lane l tests child (l & 7) of LDS node `node` against plane ((l >> 3) & 3);
// returns the 8-bit mask of the children that NO ray inside the pyramid can enter (outside one plane by a relative
// margin of 1e-3, or the all-zero box of an unpopulated child): ray_aabbs_hit_8 reports a miss for each of them
// (raytracer.c:190-230), whatever the ray's t_max.
__device__ __forceinline__ uint32_t pyramid_cull_mask(const float4 *lds_nodes, const float *pyr, int node) {
  const int lane = lane_now();
  const float *nb = reinterpret_cast<const float *>(lds_nodes + lds_node_f4(node)) + (lane & 7);
  const float *pl = pyr + ((lane >> 3) & 3) * 4;
  const float ox = pyr[16], oy = pyr[17], oz = pyr[18];
  const float mnx = nb[0], mny = nb[8], mnz = nb[16], mxx = nb[24], mxy = nb[32], mxz = nb[40];
  const float nx = pl[0], ny = pl[1], nz = pl[2];
  const bool empty = mnx == 0.0f && mny == 0.0f && mnz == 0.0f && mxx == 0.0f && mxy == 0.0f && mxz == 0.0f;
  const float lox = nx * (mnx - ox), hix = nx * (mxx - ox), loy = ny * (mny - oy), hiy = ny * (mxy - oy);
  const float loz = nz * (mnz - oz), hiz = nz * (mxz - oz);
  const float nearest = fminf(lox, hix) + fminf(loy, hiy) + fminf(loz, hiz);       // smallest n . (p - o) over the box
  const float extent = fmaxf(fabsf(lox), fabsf(hix)) + fmaxf(fabsf(loy), fabsf(hiy)) + fmaxf(fabsf(loz), fabsf(hiz));
  const bool outside = empty || nearest > 1e-3f * extent;                           // (NaN compares false: not outside)
  const uint32_t m = (uint32_t)__ballot(outside);                                  // lanes 0..31: 4 planes x 8 children
  return (m | (m >> 8) | (m >> 16) | (m >> 24)) & 0xFFu;
}

// slab_entry<true>() of child k of an LDS node with the planes picked by address (see NODE_LDS_ORDERED)
__device__ __forceinline__ float slab_entry_ordered(const Ray3 &r, const char *nbase, int k, int nx, int ny, int nz, float t_max) {
  const char *b = nbase + k * 4;
  const float sx = (*reinterpret_cast<const float *>(b + nx) - r.o.x) * r.inv_x;
  const float bx = (*reinterpret_cast<const float *>(b + (96 - nx)) - r.o.x) * r.inv_x;
  const float sy = (*reinterpret_cast<const float *>(b + 32 + ny) - r.o.y) * r.inv_y;
  const float by = (*reinterpret_cast<const float *>(b + 32 + (96 - ny)) - r.o.y) * r.inv_y;
  const float sz = (*reinterpret_cast<const float *>(b + 64 + nz) - r.o.z) * r.inv_z;
  const float bz = (*reinterpret_cast<const float *>(b + 64 + (96 - nz)) - r.o.z) * r.inv_z;
  const float t_minv = fmax_hw(RT_EPS, fmax_hw(sx, fmax_hw(sy, sz)));
  const float t_maxv = fmin_hw(t_max, fmin_hw(bx, fmin_hw(by, bz)));
  return (t_minv < t_maxv) ? t_minv : RT_INF;
}

// node_enter() for a node of which only the children in `surv` (1 to 4 of them, wave-uniform) can be entered: the
// same word -- the other children are misses, which rank behind every candidate and are never read.
__device__ __forceinline__ uint32_t node_enter_few(const Ray3 &r, const float4 *lds_nodes, int node, uint32_t surv,
                                                   float hit_t) {
  const char *nbase = reinterpret_cast<const char *>(lds_nodes + lds_node_f4(node));
  const int nx = (as_i(r.inv_x) >> 31) & 96, ny = (as_i(r.inv_y) >> 31) & 96, nz = (as_i(r.inv_z) >> 31) & 96;
  const int n = (int)__popc(surv);
  const int k0 = (int)__builtin_ctz(surv);
  const int e0 = as_i(slab_entry_ordered(r, nbase, k0, nx, ny, nz, hit_t));
  const uint32_t f0 = 1u - (((uint32_t)e0 + 0x00800000u) >> 31);                   // 1 iff e0 is finite (a candidate)
  if (n == 1) return (uint32_t)k0 | (f0 << 24);
  surv &= surv - 1u;
  const int k1 = (int)__builtin_ctz(surv);
  const int e1 = as_i(slab_entry_ordered(r, nbase, k1, nx, ny, nz, hit_t));
  const uint32_t f1 = 1u - (((uint32_t)e1 + 0x00800000u) >> 31);
  if (n == 2) {
    const bool swap = e1 < e0;                                                     // ties: lowest index first
    const uint32_t first = swap ? (uint32_t)k1 : (uint32_t)k0, second = swap ? (uint32_t)k0 : (uint32_t)k1;
    return first | (second << 3) | ((f0 + f1) << 24);
  }
  surv &= surv - 1u;
  const int k2 = (int)__builtin_ctz(surv);
  const int e2 = as_i(slab_entry_ordered(r, nbase, k2, nx, ny, nz, hit_t));
  const uint32_t f2 = 1u - (((uint32_t)e2 + 0x00800000u) >> 31);
  int e3 = 0x7F800000, k3 = 0;
  uint32_t f3 = 0;
  if (n == 4) {
    surv &= surv - 1u;
    k3 = (int)__builtin_ctz(surv);
      e3 = as_i(slab_entry_ordered(r, nbase, k3, nx, ny, nz, hit_t));
    f3 = 1u - (((uint32_t)e3 + 0x00800000u) >> 31);
  }
  const int e[4] = {e0, e1, e2, e3};
  const int kk[4] = {k0, k1, k2, k3};
  int rank[4] = {0, 1, 2, 3};
#pragma unroll
  for (int j = 0; j < 4; j++) {
#pragma unroll
    for (int k = j + 1; k < 4; k++) {
      int kb = (int)((uint32_t)(e[k] - e[j]) >> 31);      // 1 iff e[k] < e[j]
      rank[j] += kb;
      rank[k] -= kb;
    }
  }
  uint32_t w = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) w |= (uint32_t)kk[j] << (3 * rank[j]);
  return w | ((f0 + f1 + f2 + f3) << 24);
}

typedef const RT_KParams __attribute__((address_space(4))) *RT_KArgs;
__device__ __forceinline__ RT_KArgs cold_args() {
  RT_KArgs p = (RT_KArgs)__builtin_amdgcn_kernarg_segment_ptr();      // the RT_KParams block is the kernel's only argument
  asm volatile("" : "+s"(p));
  return p;
}

struct ShadeParams {            // what shade_hit / background_lookup read (same field names as RT_KParams)
  const float *tris, *mats;
  const RT_DTexture *textures;
  const uint32_t *texels;
  int32_t bg_texture, max_bounces;
};

struct PrimaryParams {          // what primary_ray reads
  float cam[3][4];
  float focal_length, inv_width, inv_height, aspect;
};


template <int WAVES, bool LDSN, int MIN_WAVES_PER_SIMD, bool SHORT_DIV>
__global__ __launch_bounds__(WAVES * 64, MIN_WAVES_PER_SIMD) void rt_path_kernel_stream(RT_KParams P) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_lds = LDSN ? P.n_lds_nodes : 0;
  const float4 *lds_nodes = smem;
  const int perm_f4 = (P.depth > 0 ? P.depth : 1) * 16;
  float4 *wave_base = smem + n_lds * RT_LDS_NODE_F4 + wave * (perm_f4 + 96);
  uint32_t *perm = reinterpret_cast<uint32_t *>(wave_base);
  unsigned long long *acc = reinterpret_cast<unsigned long long *>(wave_base + perm_f4);
  // the perm row of the deepest node level is never written (a leaf-level node has no node below it): it holds the
  // tile's camera-ray pyramid
  const int acc_off = (n_lds * RT_LDS_NODE_F4 + __builtin_amdgcn_readfirstlane(wave) * (perm_f4 + 96) + perm_f4) * 16;
  const int pyr_off = (n_lds * RT_LDS_NODE_F4 + __builtin_amdgcn_readfirstlane(wave) * (perm_f4 + 96) + perm_f4 - 16) * 16;
  // (+ 128 bytes: 32 cache entries, direct mapped by node: (node + 1) << 8 | cull mask)

  if (LDSN) {
    const float4 *g = reinterpret_cast<const float4 *>(P.nodes);
    for (int i = threadIdx.x; i < n_lds * 12; i += WAVES * 64) {
      int nd = i / 12, q = i - nd * 12;
      smem[nd * RT_LDS_NODE_F4 + q] = g[i];
    }
    __syncthreads();          // the only workgroup barrier of the kernel; waves are independent afterwards
  }

  acc[lane] = 0ull;
  acc[lane + 64] = 0ull;
  acc[lane + 128] = 0ull;

  // wave-level counters (scalar registers)
  uint32_t w_paths = 0, w_rays = 0, w_nodes = 0, w_leaves = 0, w_shades = 0, w_bgs = 0, w_tex = 0;
  const unsigned long long t_wave_start = cold_args()->wave_times ? __builtin_amdgcn_s_memrealtime() : 0ull;   // RT_WAVE_TIMES only
#ifdef RT_EXP_NODESTATS
  uint32_t xs[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  uint32_t n_tiles_done = 0;
  unsigned long long t_last_grab = 0ull;

  const int shift = P.chunk_shift;                 // samples per unit = 1 << shift
  const int unit_paths = 2 << shift;               // a unit = 2 neighbouring pixels x (1 << shift) samples
  const uint32_t n_chunks_tile = (uint32_t)P.n_chunks_tile;      // units per tile: 8 rows x sample blocks x 4 pixel pairs
  const int leaf_level = P.depth - 1;
  const int thresh = P.sched_thresh;
  const int drain_thresh = P.drain_thresh;
  const int pyr_nodes = LDSN ? P.pyr_nodes : 0;
  const int wave_id = (int)blockIdx.x * WAVES + wave;

  bool queue_open = true;
  int  steal_tries = 0;

  for (;;) {
    // ---------------- take a tile: own one from the queue, or join one that still has units ----------------
    int tile_idx = -1;
    if (queue_open) {
      RT_KArgs A = cold_args();
      uint32_t pos = 0;
      if (lane == 0) pos = atomicAdd(A->work_head, 1u);
      pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
      const uint32_t *order = A->order;
      if (pos < (uint32_t)A->n_tiles) tile_idx = order ? (int)order[pos] : (int)pos;
      else queue_open = false;
    }
    if (tile_idx < 0) {
      if (steal_tries >= RT_STEAL_TRIES) break;
      steal_tries += 1;
      // two-level scan with agent-scope loads: groups of 64 tiles that still have an open tile (open_groups[g] > 0),
      // then the tiles of one such group.  Start positions differ per wave so that joiners spread over the open tiles.
      RT_KArgs A = cold_args();
      const int n_tiles = A->n_tiles;
      const uint32_t *open_groups = A->open_groups, *tile_next = A->tile_next;
      const int n_groups = (n_tiles + 63) >> 6;
      const int g_rounds = (n_groups + 63) >> 6;
      const uint32_t hsh = ((uint32_t)wave_id * 2654435761u + (uint32_t)steal_tries * 40503u) >> 8;
      const int g_start = (int)(hsh % (uint32_t)g_rounds);
      for (int i = 0; i < g_rounds && tile_idx < 0; i++) {
        int r = g_start + i;
        if (r >= g_rounds) r -= g_rounds;
        int g = r * 64 + lane;
        uint32_t n_open = 0;
        if (g < n_groups) n_open = __hip_atomic_load(&open_groups[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long gm = __ballot(n_open != 0u);
        while (gm && tile_idx < 0) {
          int nth = (int)((hsh >> 6) % (uint32_t)__popcll(gm));
          unsigned long long m = gm;
          for (int k = 0; k < nth; k++) m &= m - 1ull;
          int gl = (int)__builtin_ctzll(m);
          gm &= ~(1ull << gl);
          int cand = (r * 64 + gl) * 64 + lane;
          uint32_t taken = 0xFFFFFFFFu;
          if (cand < n_tiles) taken = __hip_atomic_load(&tile_next[cand], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          unsigned long long open = __ballot(taken < n_chunks_tile);
          if (open) {
            // two random open tiles of the group, the one with more units left: fewer, longer joins (every join ends with
            // a drain of the paths in flight)
            const int n_open_tiles = (int)__popcll(open);
            int best = 0;
            uint32_t best_taken = 0xFFFFFFFFu;
#pragma unroll
            for (int c = 0; c < RT_JOIN_CHOICES; c++) {
              int nt = (int)(((hsh >> 12) * (uint32_t)(2 * c + 1) + (uint32_t)c * 7u) % (uint32_t)n_open_tiles);
              unsigned long long mm = open;
              for (int k = 0; k < nt; k++) mm &= mm - 1ull;
              const int pk_lane = (int)__builtin_ctzll(mm);
              const uint32_t tk = (uint32_t)__builtin_amdgcn_readlane((int)taken, pk_lane);
              if (tk < best_taken) { best_taken = tk; best = pk_lane; }
            }
            tile_idx = cand - lane + best;
          }
        }
      }
      if (tile_idx < 0) break;                      // nothing left to join
    }

    int tile_x0, tile_y0;
    {
      RT_KArgs A = cold_args();
      const int lchunk = tile_idx >> 4, sub = tile_idx & 15;
      const int chunk = A->local_chunks[lchunk];
      const int chunks_x = A->chunks_x;
      tile_x0 = (chunk % chunks_x) * 32 + (sub & 3) * 8;
      tile_y0 = (chunk / chunks_x) * 32 + (sub >> 2) * 8;
      if (tile_x0 >= A->width || tile_y0 >= A->height) {
        // tile entirely outside the image: mark it exhausted (once) so that no wave tries to join it
        if (lane == 0) {
          uint32_t old = atomicMax(&A->tile_next[tile_idx], n_chunks_tile);
          if (old < n_chunks_tile) atomicSub(&A->open_groups[tile_idx >> 6], 1u);
        }
        continue;
      }
    }
    // ---- can a camera ray of this tile touch the scene at all? ----
    // The primary rays of the tile share the origin and lie inside the pyramid through the corners of the tile's pixel
    // footprint.  If every populated child box of the ROOT lies outside one of the pyramid's four side planes -- by a
    // relative margin of 1e-3, a thousand times the rounding error of the slab test -- then ray_aabbs_hit_8 returns
    // "no candidate" for every one of them (raytracer.c:190-230, :459-472): the ray costs one node visit and goes to the
    // environment.  Such rays skip the root block here and are counted as that one visit.  (Lanes 0..7 test one child
    // box each; only rays with finite reciprocal direction take the shortcut, see ray_setup.)
    bool tile_root_miss = false;
    if (leaf_level >= 0) {
      RT_KArgs A = cold_args();
      const float m = 0.05f;                                   // footprint margin in pixels
      const float ux0 = ((float)tile_x0 - 0.5f - m) * 2.0f * A->inv_width - 1.0f;
      const float ux1 = ((float)tile_x0 + 7.5f + m) * 2.0f * A->inv_width - 1.0f;
      const float uy0 = ((float)tile_y0 - 0.5f - m) * 2.0f * A->inv_height - 1.0f;
      const float uy1 = ((float)tile_y0 + 7.5f + m) * 2.0f * A->inv_height - 1.0f;
      const float asp = A->aspect, fl = A->focal_length;
      rt_v3 c[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float cx = ((q == 1 || q == 2) ? ux1 : ux0) * asp, cy = -((q >= 2) ? uy1 : uy0), cz = -fl;
        c[q] = rt_v3_make(A->cam[0][0] * cx + A->cam[0][1] * cy + A->cam[0][2] * cz,
                          A->cam[1][0] * cx + A->cam[1][1] * cy + A->cam[1][2] * cz,
                          A->cam[2][0] * cx + A->cam[2][1] * cy + A->cam[2][2] * cz);
      }
      const rt_v3 o = rt_v3_make(A->cam[0][3], A->cam[1][3], A->cam[2][3]);
      rt_v3 pn[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        rt_v3 n = rt_v3_cross(c[q], c[(q + 1) & 3]);
        if (rt_v3_dot(n, c[(q + 2) & 3]) > 0.0f) n = rt_v3_scale(n, -1.0f);      // outward: the opposite corner is inside
        pn[q] = n;
      }
      float *pyr = lds_at(smem, pyr_off);
      if (lane == 0) {                                         // the pyramid of this tile, for the node blocks
#pragma unroll
        for (int q = 0; q < 4; q++) { pyr[q * 4 + 0] = pn[q].x; pyr[q * 4 + 1] = pn[q].y; pyr[q * 4 + 2] = pn[q].z; }
        pyr[16] = o.x; pyr[17] = o.y; pyr[18] = o.z;
      }
      if (lane < 32) reinterpret_cast<uint32_t *>(pyr)[32 + lane] = 0u;      // the cull masks found for this tile so far
      bool may_hit = false;
      if (lane < 8) {
        const float *nb = P.nodes + lane;                      // child `lane` of node 0: rows are 8 floats apart
        rt_v3 lo = rt_v3_make(nb[0] - o.x, nb[8] - o.y, nb[16] - o.z);
        rt_v3 hi = rt_v3_make(nb[24] - o.x, nb[32] - o.y, nb[40] - o.z);
        const bool empty = nb[0] == 0.0f && nb[8] == 0.0f && nb[16] == 0.0f && nb[24] == 0.0f && nb[32] == 0.0f && nb[40] == 0.0f;
        bool outside = empty;                                  // the all-zero box of an unpopulated child never hits
#pragma unroll
        for (int q = 0; q < 4; q++) {
          rt_v3 n = pn[q];
          float lox = n.x * lo.x, hix = n.x * hi.x, loy = n.y * lo.y, hiy = n.y * hi.y, loz = n.z * lo.z, hiz = n.z * hi.z;
          float nearest = fminf(lox, hix) + fminf(loy, hiy) + fminf(loz, hiz);     // smallest n . (p - o) over the box
          float extent = fmaxf(fabsf(lox), fabsf(hix)) + fmaxf(fabsf(loy), fabsf(hiy)) + fmaxf(fabsf(loz), fabsf(hiz));
          if (nearest > 1e-3f * extent) outside = true;        // (NaN compares false: not outside)
        }
        may_hit = !outside;
      }
      tile_root_miss = __ballot(may_hit) == 0ull;
    }
    const uint32_t rays_before = w_rays;

    // ---------------- per-lane state ----------------
    int   phase = PH_NEED;
    int   pix = 0, bounce = 0;
    uint32_t rng = 0;
    Ray3  ray;
    ray_setup(ray, rt_v3_make(0, 0, 0), rt_v3_make(0, 0, 1));
    rt_v3 tint = rt_v3_make(1, 1, 1), emis = rt_v3_make(0, 0, 0);
    int   level = -1, node = 0, child = 0;
    uint32_t cur = 0, dirty = 0, live = 0;
    HitRec hit;
    hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;

    // The tile hands out UNITS (2 pixels x one block of samples = 128 paths at 64 samples, lanes on one or two
    // pixels).  A wave grabs `grab_max` consecutive units per atomic while the tile has plenty left, fewer towards its
    // end: what a wave still holds when the launch runs dry is its tail.  Current unit (wave-uniform): paths
    // [c_next, c_end) at pixels (c_x0 .. c_x0 + 1, c_y), samples from c_s0.
    bool tile_open = true;
    int  c_next = 0, c_end = 0, c_x0 = 0, c_y = 0, c_pix0 = 0, c_s0 = 0;
    uint32_t u_cur = 0, u_end = 0, grab = queue_open ? (uint32_t)cold_args()->grab_max : 1u;
    bool took_any = false;
    int  n_parked = 0;                   // hits of this tile waiting in the wave's slice of `park`
    uint32_t *park = cold_args()->park;

    for (;;) {
      // ================= S: shade the hits, environment for the misses, start new paths =================
      {
        LaneCounters cn;
        cn.rays = cn.nodes = cn.leaves = cn.shades = cn.bgs = cn.textured = cn.paths = 0;
        bool  done = false, start = false, fresh = false;
        rt_v3 radiance = rt_v3_make(0, 0, 0);
        rt_v3 org = ray.o, dir = ray.d;
        RT_KArgs A = cold_args();
        ShadeParams SP;
        SP.tris = A->tris; SP.mats = A->mats; SP.textures = A->textures; SP.texels = A->texels;
        SP.bg_texture = A->bg_texture; SP.max_bounces = A->max_bounces;
        // ---- environment for the paths that left the scene ----
        if (phase == PH_MISS) {
          cn.bgs = 1;
          rt_v3 bg = background_lookup(SP, dir);
          radiance = rt_v3_add(rt_v3_mul(bg, tint), emis);
          done = true;
        }
        // ---- hits: shade them now, or park them until a dense shade block can be made of them ----
        // A shade block costs ~2 200 instructions whatever the number of lanes in it, and hits arrive ~34 at a time.  While
        // the tile still hands out paths, the hits of a sparse S block are PARKED (18 dwords of path state per hit into the
        // wave's slice of `park`, struct-of-arrays: every store is one 256-byte line) and their lanes start new camera
        // paths; once the hits at hand plus the parked ones that fit into idle lanes make RT_PARK_DENSE lanes, the parked
        // ones come back into idle lanes and all are shaded in one block.  A draining tile shades what it has.  Which
        // wave-iteration shades a path does not matter: its state (rng included) travels with it, sums are order-free.
        bool shade_now = true;
        if (park) {
          const unsigned long long mHit = __ballot(phase == PH_HIT);
          const unsigned long long mIdle = __ballot(phase == PH_NEED) | __ballot(phase == PH_MISS);
          const int h = (int)__popcll(mHit), f = (int)__popcll(mIdle);
          const int back = n_parked < f ? n_parked : f;                 // parked hits that fit into idle lanes
          uint32_t *pk = park + (size_t)__builtin_amdgcn_readfirstlane(wave_id) * (RT_PARK_FIELDS * RT_PARK_CAP);    // (scalar base)
          if (!tile_open || h + back >= RT_PARK_DENSE || n_parked + h > RT_PARK_CAP) {
            if (back > 0) {
              if (done) {                                               // (an environment lane is idle once its sample is added)
                unsigned long long *ap = reinterpret_cast<unsigned long long *>(lds_at(smem, acc_off) + pix * 6);
                atomicAdd(ap + 0, accum_quantize_dev(radiance.x));
                atomicAdd(ap + 1, accum_quantize_dev(radiance.y));
                atomicAdd(ap + 2, accum_quantize_dev(radiance.z));
                phase = PH_NEED;
                done = false;
              }
              const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mIdle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mIdle, 0u));
              if (phase == PH_NEED && rank < back) {
                const uint32_t *q = pk + (n_parked - 1 - rank);         // most recently parked first
                uint32_t v[RT_PARK_FIELDS];
#pragma unroll
                for (int i = 0; i < RT_PARK_FIELDS; i++)
                  v[i] = __hip_atomic_load(q + i * RT_PARK_CAP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hit.t = as_f((int)v[0]); hit.tri = (int)v[1]; hit.u = as_f((int)v[2]); hit.v = as_f((int)v[3]);
                org = rt_v3_make(as_f((int)v[4]), as_f((int)v[5]), as_f((int)v[6]));
                dir = rt_v3_make(as_f((int)v[7]), as_f((int)v[8]), as_f((int)v[9]));
                tint = rt_v3_make(as_f((int)v[10]), as_f((int)v[11]), as_f((int)v[12]));
                emis = rt_v3_make(as_f((int)v[13]), as_f((int)v[14]), as_f((int)v[15]));
                rng = v[16]; pix = (int)(v[17] & 63u); bounce = (int)(v[17] >> 6);
                phase = PH_HIT;
              }
              n_parked -= back;
            }
          } else if (h > 0) {
            shade_now = false;
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mHit >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mHit, 0u));
            if (phase == PH_HIT) {
              uint32_t *q = pk + (n_parked + rank);
              const uint32_t v[RT_PARK_FIELDS] = {(uint32_t)as_i(hit.t), (uint32_t)hit.tri, (uint32_t)as_i(hit.u), (uint32_t)as_i(hit.v),
                                                  (uint32_t)as_i(org.x), (uint32_t)as_i(org.y), (uint32_t)as_i(org.z),
                                                  (uint32_t)as_i(dir.x), (uint32_t)as_i(dir.y), (uint32_t)as_i(dir.z),
                                                  (uint32_t)as_i(tint.x), (uint32_t)as_i(tint.y), (uint32_t)as_i(tint.z),
                                                  (uint32_t)as_i(emis.x), (uint32_t)as_i(emis.y), (uint32_t)as_i(emis.z),
                                                  rng, (uint32_t)pix | ((uint32_t)bounce << 6)};
#pragma unroll
              for (int i = 0; i < RT_PARK_FIELDS; i++) q[i * RT_PARK_CAP] = v[i];
              phase = PH_NEED;
            }
            n_parked += h;
          }
        }
        if (phase == PH_HIT && shade_now) {
          done = shade_hit(SP, hit, org, dir, tint, emis, rng, bounce, cn, radiance);
          start = !done;
        }
        if (done) {
          // (32-bit address arithmetic from the wave's byte offset: `acc + pix * 3` is a 64-bit multiply-add on a pointer
          // that is kept in scratch)
          unsigned long long *ap = reinterpret_cast<unsigned long long *>(lds_at(smem, acc_off) + pix * 6);
          atomicAdd(ap + 0, accum_quantize_dev(radiance.x));
          atomicAdd(ap + 1, accum_quantize_dev(radiance.y));
          atomicAdd(ap + 2, accum_quantize_dev(radiance.z));
          phase = PH_NEED;
        }
        w_shades += (uint32_t)__popcll(__ballot(cn.shades != 0));
        w_tex += (uint32_t)__popcll(__ballot(cn.textured != 0));
        w_bgs += (uint32_t)__popcll(__ballot(cn.bgs != 0));

        // ---- regeneration: idle lanes take the next paths of the tile, across chunk boundaries ----
        if (tile_open) {
          unsigned long long need = __ballot(phase == PH_NEED);
          bool got = false;
          int  gx = 0, gy = 0, gs = 0, gp = 0;
          const int width = A->width, sample_end = A->sample_end;
          while (need) {
            if (c_next >= c_end) {
              if (u_cur + 1u < u_end) {
                u_cur += 1u;
              } else {
                uint32_t u0 = 0;
                if (lane == 0) u0 = atomicAdd(&A->tile_next[tile_idx], grab);
                u0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)u0);
                if (u0 >= n_chunks_tile) { tile_open = false; break; }
                u_cur = u0;
                u_end = u0 + grab < n_chunks_tile ? u0 + grab : n_chunks_tile;
                // exactly one wave receives the tile's last unit: it closes the tile in the group summary
                if (u_end == n_chunks_tile && lane == 0) atomicSub(&A->open_groups[tile_idx >> 6], 1u);
                if (t_wave_start) t_last_grab = __builtin_amdgcn_s_memrealtime();
                // next grab: `grab_max` units while the tile has plenty left, fewer towards its end.
                // (A launch-wide count of the remaining units would be the better guide, but a counter that every grab
                // updates -- one address or 64 shards of one line -- made the frame 2x slower: measured, removed.)
                const uint32_t left = n_chunks_tile - u_end;
                const uint32_t gmax = (uint32_t)A->grab_max;
                grab = left >= 8u * gmax ? gmax : (left >= 8u && gmax >= 2u ? 2u : 1u);
                took_any = true;
              }
              // unit u -> (row, sample block, pixel pair): all sample blocks of a row before the next row, so a
              // pixel's texture / geometry footprint is touched in one burst
              const uint32_t grp = u_cur >> 2, pair = u_cur & 3u;
              const uint32_t n_sb = (uint32_t)A->n_sample_blocks;
              const uint32_t row = grp / n_sb, sb = grp - row * n_sb;
              c_x0 = tile_x0 + (int)pair * 2;
              c_y = tile_y0 + (int)row;
              c_pix0 = (int)row * 8 + (int)pair * 2;
              c_s0 = A->sample_first + (int)(sb << shift);
              c_next = 0;
              c_end = (c_y < A->height && c_x0 < width) ? unit_paths : 0;      // units outside the image have no paths
              continue;
            }
            const int n_need = (int)__popcll(need);
            const int avail = c_end - c_next;
            const int take = n_need < avail ? n_need : avail;
            // set bits of `need` below this lane (v_mbcnt: no lane mask kept in registers)
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            bool valid = false;
            if (phase == PH_NEED && !got && rank < take) {
              // k -> (pixel of the row, sample of the chunk), pixel-major: the lanes of a wave stay on a few pixels
              int k = c_next + rank;
              int px = k >> shift;
              int sm = c_s0 + (k & ((1 << shift) - 1));
              int x = c_x0 + px;
              if (x < width && sm < sample_end) {
                valid = true;
                if (SP.max_bounces > 0) { got = true; gx = x; gy = c_y; gs = sm; gp = c_pix0 + px; }
                // max_bounces == 0: the path exists and is black (the loop of raytracer.c:512 runs zero times)
              }
            }
            w_paths += (uint32_t)__popcll(__ballot(valid));
            c_next += take;
            need = __ballot(phase == PH_NEED && !got);
          }
          if (got) {
            pix = gp;
            bounce = 0;
            rng = rt_path_seed(A->seed, (uint32_t)(gx + gy * width), (uint32_t)gs);
            PrimaryParams PP;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
              for (int j = 0; j < 4; j++) PP.cam[i][j] = A->cam[i][j];
            PP.focal_length = A->focal_length; PP.inv_width = A->inv_width; PP.inv_height = A->inv_height; PP.aspect = A->aspect;
            primary_ray(PP, gx, gy, gs, org, dir);
            tint = rt_v3_make(1, 1, 1);
            emis = rt_v3_make(0, 0, 0);
            start = true;
            fresh = true;
          }
        }
        bool skip_root = false;
        if (start) {                      // a new ray: traversal starts at the root (or at leaf group 0)
          ray_setup<SHORT_DIV>(ray, org, dir);
          hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;
          dirty = 0;
          live = 0;
          cur = 0;
          level = -1;
          node = 0;
          child = (leaf_level >= 0) ? 0 : P.last_row_offset;
          phase = (leaf_level >= 0) ? PH_NODE : PH_LEAF;
          // camera ray of a tile that cannot touch the scene: its one node visit finds no candidate (see tile_root_miss)
          skip_root = fresh && tile_root_miss && ray.fast;
          if (skip_root) phase = PH_MISS;
        }
        w_rays += (uint32_t)__popcll(__ballot(start));
        w_nodes += (uint32_t)__popcll(__ballot(skip_root));
      }

      const int n_trav0 = (int)__popcll(__ballot(phase == PH_NODE || phase == PH_LEAF));
      if (n_trav0 == 0) {
        if (__any(phase == PH_MISS)) continue;   // camera rays that skipped the root: straight to the environment
        if (n_parked > 0 || __any(phase == PH_HIT)) continue;      // parked hits come back in the next S block
        if (!tile_open) break;            // every path of the tile that this wave took has ended
        continue;                         // (nothing started, e.g. pixels outside the image: pull more)
      }

      // ================= traversal: NODE / LEAF blocks until `thresh` lanes wait for S =================
      for (;;) {
        const unsigned long long maskN = __ballot(phase == PH_NODE);
        const int nN = (int)__popcll(maskN);
        const int nL = (int)__popcll(__ballot(phase == PH_LEAF));
        // while the tile still hands out paths, wait until `thresh` lanes want the S block (dense shading); once it is
        // exhausted nothing refills the lanes, and what matters is the latency of the remaining paths' bounce chains:
        // shade as soon as `drain_thresh` lanes wait
        if (nN + nL == 0 || n_trav0 - (nN + nL) >= (tile_open ? thresh : drain_thresh)) break;

        if (nL >= nN) {
          // ----- LEAF -----
          w_leaves += (uint32_t)nL;
          if (phase == PH_LEAF) {
            int  g = child - P.last_row_offset;
            bool got = SHORT_DIV ? leaf_test_short_div(P, ray, g, hit) : leaf_test<false>(P, ray, g, hit);
            if (got) dirty = 0xFFFFFFFFu;
            phase = PH_POP;
          }
        } else {
          // ----- NODE -----
          w_nodes += (uint32_t)nN;
          const bool all_fast = (maskN & __ballot(!ray.fast)) == 0ull;
#ifdef RT_EXP_NODESTATS
          {
            unsigned long long act = __ballot(phase == PH_NODE), cam = __ballot(phase == PH_NODE && bounce == 0);
            xs[0] += 1; xs[1] += (uint32_t)nN; xs[2] += (uint32_t)__popcll(cam);
            if (cam) {
              int first = (int)__builtin_ctzll(cam);
              int c0 = __builtin_amdgcn_readlane(child, first), p0 = __builtin_amdgcn_readlane(pix >> 1, first);
              unsigned long long grp = __ballot(phase == PH_NODE && bounce == 0 && child == c0 && (pix >> 1) == p0);
              unsigned long long grt = __ballot(phase == PH_NODE && bounce == 0 && child == c0);
              xs[3] += (uint32_t)__popcll(grp); if (grp == act) xs[4] += 1;
              if (2 * (int)__popcll(grp) >= nN) xs[5] += 1;
              xs[7] += (uint32_t)__popcll(grt); if (grt == act) xs[8] += 1;
            } else xs[6] += 1;
          }
#endif
          // camera rays of this tile about to enter the same node: test only the children their pyramid can touch.  When
          // they are most of the block's lanes, the block runs for them alone; the others keep waiting for a node block.
          uint32_t surv = 0xFFFFu;
          bool in_blk = phase == PH_NODE;
          // (ballots of single comparisons combined with scalar ANDs: a ballot of a compound condition costs two more
          // vector instructions)
          const unsigned long long camN = maskN & __ballot(bounce == 0);
          if (LDSN && all_fast && camN != 0ull) {
            const int c0 = __builtin_amdgcn_readlane(child, (int)__builtin_ctzll(camN));
            const int nG = (int)__popcll(camN & __ballot(child == c0));
            if (c0 < pyr_nodes && nG * RT_PYR_DEN >= nN * RT_PYR_NUM && nG >= RT_PYR_MIN) {
              // the mask depends on (tile, node) only and the tile's camera rays keep coming back to the same nodes
              float *pyr = lds_at(smem, pyr_off);
              uint32_t *slot = reinterpret_cast<uint32_t *>(pyr) + 32 + (c0 & 31);
              const uint32_t ce = (uint32_t)__builtin_amdgcn_readfirstlane((int)*slot);
              if ((ce >> 8) == (uint32_t)c0 + 1u) {
                surv = 0xFFu & ~ce;
              } else {
                const uint32_t cull = pyramid_cull_mask(lds_nodes, pyr, c0);
                if (lane_now() == 0) *slot = (((uint32_t)c0 + 1u) << 8) | cull;
                surv = 0xFFu & ~cull;
              }
              if (__popc(surv) > 4) surv = 0xFFFFu;
              else { in_blk = phase == PH_NODE && bounce == 0 && child == c0; w_nodes -= (uint32_t)(nN - nG); }
#ifdef RT_EXP_NODESTATS
              xs[9 + (surv == 0xFFFFu ? 5 : (int)__popc(surv))] += 1;
#endif
            }
          }
          if (in_blk) {
            if (level >= 0) {
              perm[level * 64 + lane] = cur;
              live = (cur >> 24) ? (live | (1u << level)) : (live & ~(1u << level));
            }
            node = child;
            level += 1;
            if (surv <= 0xFFu) {
              cur = surv ? node_enter_few(ray, lds_nodes, node, surv, hit.t) : 0u;
            } else if (all_fast) {
              if (LDSN && __ballot(node >= n_lds) == 0) cur = node_enter<true, NODE_LDS_ORDERED>(P, ray, node, hit.t, lds_nodes);
              else cur = node_enter<true, NODE_GLOBAL>(P, ray, node, hit.t, lds_nodes);
            } else {
              cur = node_enter<false, NODE_GLOBAL>(P, ray, node, hit.t, lds_nodes);
            }
            dirty &= ~(1u << level);
            if (cur >> 24) {
              child = 8 * node + 1 + (int)(cur & 7u);
              cur = ((cur >> 3) & 0x1FFFFFu) | (((cur >> 24) - 1u) << 24);
              phase = (level == leaf_level) ? PH_LEAF : PH_NODE;
            } else {
              phase = PH_POP;
            }
          }
        }

        // ----- pops: every lane that just finished a block takes its next child / goes up -----
        while (__any(phase == PH_POP)) {
          if (phase == PH_POP) {
            uint32_t cnt = cur >> 24;
            if (cnt == 0 || level < 0) {
              uint32_t above = (level > 0) ? (live & ((1u << level) - 1u)) : 0u;
              if (above == 0u) {
                level = -1;
                phase = (hit.tri >= 0) ? PH_HIT : PH_MISS;
              } else {
                int target = 31 - __clz((int)above);
                int k3 = 3 * (level - target);
                node = (int)(((uint32_t)node - (0x09249249u & ((1u << k3) - 1u))) >> k3);
                level = target;
                cur = perm[level * 64 + lane];
                cnt = cur >> 24;
              }
            }
            if (phase == PH_POP) {
              int j = (int)(cur & 7u);
              cur = ((cur >> 3) & 0x1FFFFFu) | ((cnt - 1u) << 24);
              bool go = true;
              if ((dirty >> level) & 1u) {
                float dj;
                if (LDSN && node < n_lds) {
                  // entry distance from the three NEAR planes, picked by address (see NODE_LDS_ORDERED); the rays that are
                  // not NaN-free -- a lane in a blue moon -- redo it through the min / max form
                  const char *nb = reinterpret_cast<const char *>(lds_nodes + lds_node_f4(node)) + j * 4;
                  const float sx = (*reinterpret_cast<const float *>(nb + ((as_i(ray.inv_x) >> 31) & 96)) - ray.o.x) * ray.inv_x;
                  const float sy = (*reinterpret_cast<const float *>(nb + 32 + ((as_i(ray.inv_y) >> 31) & 96)) - ray.o.y) * ray.inv_y;
                  const float sz = (*reinterpret_cast<const float *>(nb + 64 + ((as_i(ray.inv_z) >> 31) & 96)) - ray.o.z) * ray.inv_z;
                  dj = fmax_hw(RT_EPS, fmax_hw(sx, fmax_hw(sy, sz)));
                  if (!ray.fast) dj = slab_entry_child<false>(reinterpret_cast<const float *>(lds_nodes + lds_node_f4(node)) + j, ray);
                } else {
                  dj = slab_entry_child<false>(P.nodes + (size_t)node * 48 + j, ray);
                }
                if (!(dj < hit.t)) { cur = 0; go = false; }      // raytracer.c:470-472
              }
              if (go) {
                child = 8 * node + 1 + j;
                phase = (level == leaf_level) ? PH_LEAF : PH_NODE;
              }
            }
          }
        }
      }
    }

    // ---------------- flush the wave's share of the tile: lane p owns pixel p ----------------
    if (took_any) {
      int x = tile_x0 + (lane & 7), y = tile_y0 + (lane >> 3);
      unsigned long long r = acc[lane * 3 + 0], g = acc[lane * 3 + 1], b = acc[lane * 3 + 2];
      acc[lane * 3 + 0] = 0ull;
      acc[lane * 3 + 1] = 0ull;
      acc[lane * 3 + 2] = 0ull;
      RT_KArgs A = cold_args();
      const int width = A->width;
      if (x < width && y < A->height && (r | g | b) != 0ull) {
        unsigned long long *dst = A->accum + ((size_t)y * width + x) * 3;
        atomicAdd(dst + 0, r);
        atomicAdd(dst + 1, g);
        atomicAdd(dst + 2, b);
      }
      uint32_t *tile_cost = A->tile_cost;
      if (tile_cost && lane == 0) atomicAdd(&tile_cost[tile_idx], w_rays - rays_before);
      steal_tries = 0;                    // joined (or owned) a tile that had work: keep looking for more
      n_tiles_done += 1;
    }
  }
  RT_KArgs A = cold_args();
  unsigned long long *wave_times = A->wave_times, *counters = A->counters;
  if (wave_times && lane == 0 && wave_id < 65536) {
    wave_times[wave_id * 3 + 0] = t_wave_start;
    wave_times[wave_id * 3 + 1] = __builtin_amdgcn_s_memrealtime();
    wave_times[wave_id * 3 + 2] = ((t_last_grab - t_wave_start) << 16) | (n_tiles_done & 0xFFFFu);
  }

  if (lane == 0) {
    atomicAdd(counters + CNT_PATHS, (unsigned long long)w_paths);
    atomicAdd(counters + CNT_RAYS, (unsigned long long)w_rays);
    atomicAdd(counters + CNT_NODES, (unsigned long long)w_nodes);
    atomicAdd(counters + CNT_LEAVES, (unsigned long long)w_leaves);
    atomicAdd(counters + CNT_SHADES, (unsigned long long)w_shades);
    atomicAdd(counters + CNT_BG, (unsigned long long)w_bgs);
    atomicAdd(counters + CNT_TEXTURED, (unsigned long long)w_tex);
#ifdef RT_EXP_NODESTATS
    for (int i = 0; i < 16; i++) atomicAdd(counters + 8 + i, (unsigned long long)xs[i]);
#endif
  }
}

// ---------------------------------------------------------------------------------
// accum -> mean -> clamp -> sRGB -> u8 (raytracer.c:700-716), one thread per pixel
// of this rank's chunks.
__global__ void rt_resolve_kernel(int width, int height, int samples, int chunks_x, const int32_t *local_chunks,
                                  int n_local_chunks, const unsigned long long *accum,
                                  uint8_t *tiles, uint8_t *image, float *linear) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_local_chunks * 1024) return;
  int lchunk = idx >> 10, p = idx & 1023;
  int chunk = local_chunks[lchunk];
  int x = (chunk % chunks_x) * 32 + (p & 31);
  int y = (chunk / chunks_x) * 32 + (p >> 5);
  uint8_t rgb[3] = {0, 0, 0};
  if (x < width && y < height) {
    size_t pix = (size_t)y * width + x;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      float lin = rt_accum_resolve(accum[pix * 3 + c], (uint32_t)samples);
      if (linear) linear[pix * 3 + c] = lin;
      rgb[c] = rt_encode_u8(lin);
      if (image) image[pix * 3 + c] = rgb[c];
    }
  }
  if (tiles) {
    tiles[(size_t)idx * 3 + 0] = rgb[0];
    tiles[(size_t)idx * 3 + 1] = rgb[1];
    tiles[(size_t)idx * 3 + 2] = rgb[2];
  }
}

// gathered compact tiles [world][max_local][1024*3] -> row-major image; owner_slot[chunk] = rank * max_local + slot
__global__ void rt_untile_kernel(int width, int height, int chunks_x, int n_chunks, const int32_t *owner_slot,
                                 const uint8_t *all_tiles, uint8_t *image) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_chunks * 1024) return;
  int chunk = idx >> 10, p = idx & 1023;
  int x = (chunk % chunks_x) * 32 + (p & 31);
  int y = (chunk / chunks_x) * 32 + (p >> 5);
  if (x >= width || y >= height) return;
  const uint8_t *src = all_tiles + ((size_t)owner_slot[chunk] * 1024 + p) * 3;
  uint8_t *dst = image + ((size_t)y * width + x) * 3;
  dst[0] = src[0];
  dst[1] = src[1];
  dst[2] = src[2];
}

// ---------------------------------------------------------------------------------
// lightmap_bake (raytracer.c:722-784): the second caller of the path loop.
// Pass 1 rasterises every triangle in UV space and records, per texel, the LAST triangle that covers it
// (the reference's sequential loop overwrites in triangle order); pass 2 bakes each owned texel:
// `samples` cosine-weighted paths of 8 bounces from the interpolated surface point.

// barycentric weights of texel (x, y) for the UV triangle (p0, p1, p2), raytracer.c:733-747
__device__ __forceinline__ bool lightmap_weights(float p0x, float p0y, float p1x, float p1y, float p2x, float p2y,
                                                 int x, int y, float &w0, float &w1, float &w2) {
  float denom = (p1y - p2y) * (p0x - p2x) + (p2x - p1x) * (p0y - p2y);
  float px = (float)x, py = (float)y;
  w0 = ((p1y - p2y) * (px - p2x) + (p2x - p1x) * (py - p2y)) / denom;
  w1 = ((p2y - p0y) * (px - p2x) + (p0x - p2x) * (py - p2y)) / denom;
  w2 = 1.0f - w0 - w1;
  return w0 >= -RT_EPS && w1 >= -RT_EPS && w2 >= -RT_EPS;
}

__device__ __forceinline__ float min3f(float a, float b, float c) { float m = b < c ? b : c; return a < m ? a : m; }
__device__ __forceinline__ float max3f(float a, float b, float c) { float m = b > c ? b : c; return a > m ? a : m; }

__global__ void rt_lightmap_owner_kernel(RT_KParams P, int n_tris, int lw, int lh, int *owner) {
  int i = blockIdx.x;
  if (i >= n_tris) return;
  const float *tb = P.tris + (size_t)i * 28;
  float uax = tb[7], uay = tb[11], ubx = tb[15], uby = tb[19], ucx = tb[23], ucy = tb[24];
  float flw = (float)lw, flh = (float)lh;
  int min_x = (int)(min3f(uax, ubx, ucx) * flw), max_x = (int)(max3f(uax, ubx, ucx) * flw);
  int min_y = (int)(min3f(uay, uby, ucy) * flh), max_y = (int)(max3f(uay, uby, ucy) * flh);
  float p0x = uax * flw, p0y = uay * flh, p1x = ubx * flw, p1y = uby * flh, p2x = ucx * flw, p2y = ucy * flh;
  // clip the loop to the image (texels outside are skipped, see oracle.h)
  int x0 = min_x < 0 ? 0 : min_x, x1 = max_x >= lw ? lw - 1 : max_x;
  int y0 = min_y < 0 ? 0 : min_y, y1 = max_y >= lh ? lh - 1 : max_y;
  int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
  if (bw <= 0 || bh <= 0) return;
  for (int t = threadIdx.x; t < bw * bh; t += blockDim.x) {
    int x = x0 + t % bw, y = y0 + t / bw;
    float w0, w1, w2;
    if (lightmap_weights(p0x, p0y, p1x, p1y, p2x, p2y, x, y, w0, w1, w2)) atomicMax(&owner[y * lw + x], i);
  }
}

// raytracer.c:505-558 for one ray, sequential per lane (used by the lightmap; the frame kernels schedule
// the same steps per phase instead)
__device__ __forceinline__ rt_v3 cast_ray_lane(const RT_KParams &P, rt_v3 org, rt_v3 dir, uint32_t &rng,
                                               uint32_t *perm, int lane, LaneCounters &cn) {
  rt_v3 tint = rt_v3_make(1, 1, 1), emis = rt_v3_make(0, 0, 0), radiance = rt_v3_make(0, 0, 0);
  int bounce = 0;
  bool done = P.max_bounces <= 0;
  while (!done) {
    Ray3 ray;
    ray_setup(ray, org, dir);
    HitRec hit;
    if (ray.fast) trace_ray<true>(P, ray, hit, perm, lane, cn);
    else trace_ray<false>(P, ray, hit, perm, lane, cn);
    if (hit.tri >= 0) {
      done = shade_hit(P, hit, org, dir, tint, emis, rng, bounce, cn, radiance);
    } else {
      cn.bgs += 1;
      radiance = rt_v3_add(rt_v3_mul(background_lookup(P, dir), tint), emis);
      done = true;
    }
  }
  return radiance;
}

// common.h:30-42
__device__ __forceinline__ rt_v3 rand_vec3_dev(uint32_t &rng) {
  for (;;) {
    rt_v3 p;
    p.x = rt_rand_f32(&rng) * (1.0f - -1.0f) + -1.0f;
    p.y = rt_rand_f32(&rng) * (1.0f - -1.0f) + -1.0f;
    p.z = rt_rand_f32(&rng) * (1.0f - -1.0f) + -1.0f;
    float lensq = rt_v3_dot(p, p);
    if (RT_EPS < lensq && lensq <= 1.0f) return rt_v3_scale(p, 1.0f / rt_sqrtf(lensq));
  }
}

__global__ __launch_bounds__(RT_BLOCK_THREADS) void rt_lightmap_bake_kernel(RT_KParams P, const float *verts, int lw, int lh,
                                                                            int stride, int comp, int samples,
                                                                            const int *owner, uint8_t *pixels) {
  __shared__ uint32_t s_perm[RT_BLOCK_WAVES][RT_MAX_DEPTH * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= lw * lh) return;
  int i = owner[idx];
  if (i < 0) return;
  int x = idx % lw, y = idx / lw;
  const float *tb = P.tris + (size_t)i * 28;
  float flw = (float)lw, flh = (float)lh;
  float w0, w1, w2;
  lightmap_weights(tb[7] * flw, tb[11] * flh, tb[15] * flw, tb[19] * flh, tb[23] * flw, tb[24] * flh, x, y, w0, w1, w2);
  const float *v = verts + (size_t)i * 9;       // x0 x1 x2 y0 y1 y2 z0 z1 z2 (original vertices, scene.h:53-60)
  rt_v3 position = rt_v3_make(v[0] * w0 + v[1] * w1 + v[2] * w2, v[3] * w0 + v[4] * w1 + v[5] * w2,
                              v[6] * w0 + v[7] * w1 + v[8] * w2);
  rt_v3 normal = rt_v3_make(tb[4] * w0 + tb[8] * w1 + tb[12] * w2, tb[5] * w0 + tb[9] * w1 + tb[13] * w2,
                            tb[6] * w0 + tb[10] * w1 + tb[14] * w2);
  rt_v3 org = rt_v3_add(position, rt_v3_scale(normal, RT_EPS));
  uint32_t rng = rt_path_seed(P.seed, (uint32_t)(x + y * lw), (uint32_t)i);
  LaneCounters cn;
  cn.rays = cn.nodes = cn.leaves = cn.shades = cn.bgs = cn.textured = cn.paths = 0;
  rt_v3 acc = rt_v3_make(0, 0, 0);
  for (int s = 0; s < samples; s++) {
    float cosv;
    rt_v3 d;
    int guard = 0;
    for (;;) {
      d = rand_vec3_dev(rng);
      cosv = rt_v3_dot(d, normal);
      if (cosv > 0.0f) break;
      if (++guard >= 64) { cosv = 0.0f; break; }
    }
    acc = rt_v3_add(acc, rt_v3_scale(cast_ray_lane(P, org, d, rng, s_perm[wave], lane, cn), cosv));
  }
  float out[3] = {acc.x / (float)samples, acc.y / (float)samples, acc.z / (float)samples};
#pragma unroll
  for (int c = 0; c < 3; c++) {
    float q = out[c] > 0.0f ? out[c] : 0.0f;
    q = q > 255.0f ? 255.0f : q;
    pixels[((size_t)x + (size_t)y * stride) * comp + c] = (uint8_t)q;
  }
}

extern "C" int rt_launch_lightmap(const RT_KParams *P, const float *verts, int n_tris, int lw, int lh, int stride, int comp,
                                  int samples, int *owner, uint8_t *pixels, hipStream_t stream) {
  hipLaunchKernelGGL(rt_lightmap_owner_kernel, dim3(n_tris), dim3(64), 0, stream, *P, n_tris, lw, lh, owner);
  int n = lw * lh;
  hipLaunchKernelGGL(rt_lightmap_bake_kernel, dim3((n + RT_BLOCK_THREADS - 1) / RT_BLOCK_THREADS), dim3(RT_BLOCK_THREADS), 0,
                     stream, *P, verts, lw, lh, stride, comp, samples, owner, pixels);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------
// Tile order for the next launch: counting sort of the tiles by descending cost bucket (4 buckets per
// power of two of last launch's ray count).  Order inside a bucket is whatever the atomics give: only the
// schedule depends on it, never a pixel (order-free accumulation).
#define RT_ORDER_BUCKETS 132
__device__ __forceinline__ int cost_bucket(uint32_t c) {
  if (c < 4u) return (int)c;                               // 0..3
  int e = 31 - __clz((int)c);                              // >= 2
  return 4 * (e - 1) + (int)((c >> (e - 2)) & 3u);         // 4..131, monotonic in c
}

// (most tiles of a frame fall into a handful of buckets: the counters are combined per workgroup in LDS first,
//  one global atomic per bucket and workgroup -- a global atomic per tile serialises on those few addresses)
__global__ void rt_order_hist_kernel(int n, const uint32_t *cost, uint32_t *hist) {
  __shared__ uint32_t local[RT_ORDER_BUCKETS];
  for (int b = threadIdx.x; b < RT_ORDER_BUCKETS; b += blockDim.x) local[b] = 0;
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&local[cost_bucket(cost[i])], 1u);
  __syncthreads();
  for (int b = threadIdx.x; b < RT_ORDER_BUCKETS; b += blockDim.x)
    if (local[b]) atomicAdd(&hist[b], local[b]);
}

__global__ void rt_order_scan_kernel(uint32_t *hist) {      // one thread: start offset of every bucket, expensive first
  uint32_t run = 0;
  for (int b = RT_ORDER_BUCKETS - 1; b >= 0; b--) {
    uint32_t c = hist[b];
    hist[b] = run;
    run += c;
  }
}

__global__ void rt_order_scatter_kernel(int n, const uint32_t *cost, uint32_t *cursor, uint32_t *order) {
  __shared__ uint32_t local[RT_ORDER_BUCKETS], base[RT_ORDER_BUCKETS];
  for (int b = threadIdx.x; b < RT_ORDER_BUCKETS; b += blockDim.x) local[b] = 0;
  __syncthreads();
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int bucket = 0;
  uint32_t pos = 0;
  if (i < n) {
    bucket = cost_bucket(cost[i]);
    pos = atomicAdd(&local[bucket], 1u);                   // rank inside this workgroup's share of the bucket
  }
  __syncthreads();
  for (int b = threadIdx.x; b < RT_ORDER_BUCKETS; b += blockDim.x)
    base[b] = local[b] ? atomicAdd(&cursor[b], local[b]) : 0u;   // reserve the workgroup's range once per bucket
  __syncthreads();
  if (i < n) order[base[bucket] + pos] = (uint32_t)i;
}

extern "C" int rt_launch_tile_order(int n_tiles, const uint32_t *cost, uint32_t *hist, uint32_t *order, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(hist, 0, RT_ORDER_BUCKETS * sizeof(uint32_t), stream);
  if (e != hipSuccess) return (int)e;
  int blocks = (n_tiles + 255) / 256;
  hipLaunchKernelGGL(rt_order_hist_kernel, dim3(blocks), dim3(256), 0, stream, n_tiles, cost, hist);
  hipLaunchKernelGGL(rt_order_scan_kernel, dim3(1), dim3(1), 0, stream, hist);
  hipLaunchKernelGGL(rt_order_scatter_kernel, dim3(blocks), dim3(256), 0, stream, n_tiles, cost, hist, order);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------
// unit-level kernels for parity tests

__global__ void rt_test_math_kernel(int op, int n, const float *x, const float *y, float *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = x[i], b = y ? y[i] : 0.0f, s, c;
  float r = 0.0f;
  switch (op) {
  case 0: r = rt_logf(a); break;
  case 1: r = rt_expf(a); break;
  case 2: r = rt_powf(a, b); break;
  case 3: rt_sincosf(a, &s, &c); r = s; break;
  case 4: rt_sincosf(a, &s, &c); r = c; break;
  case 5: r = rt_atan2f(a, b); break;
  case 6: r = rt_asinf(a); break;
  case 7: r = rt_srgb_to_linear1(a); break;
  case 8: r = rt_linear_to_srgb(a); break;
  case 9: r = rt_sqrtf(a); break;
  case 10: r = 1.0f / a; break;
  case 11: r = rcp_exact(a); break;
  case 12: r = rcp_exact_outside(a) ? 1.0f : 0.0f; break;
  case 13: r = srgb_to_linear_tex1(a); break;
  default: break;
  }
  out[i] = r;
}

// All 2^32 bit patterns x: rcp_exact(x) against the IEEE quotient 1.0f / x.  counts[0] = patterns inside the claimed
// domain (|x| < 2^102, infinity, NaN) that differ (NaN equals NaN), counts[1] = patterns outside it, counts[2] = of those,
// how many differ (why the domain ends there), counts[3] = first differing pattern inside the domain + 1.
__global__ void rt_test_rcp_sweep_kernel(unsigned long long *counts) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t bad_in = 0, n_out = 0, bad_out = 0, first = 0;
  for (uint32_t k = 0; k < 256u; k++) {
    const uint32_t b = tid * 256u + k;
    const float x = __uint_as_float(b);
    const uint32_t w = __float_as_uint(1.0f / x), g = __float_as_uint(rcp_exact(x));
    const bool w_nan = (w & 0x7FFFFFFFu) > 0x7F800000u, g_nan = (g & 0x7FFFFFFFu) > 0x7F800000u;
    const bool same = w_nan ? g_nan : (g == w);
    if (rcp_exact_outside(x)) { n_out += 1; bad_out += same ? 0u : 1u; }
    else if (!same) { bad_in += 1; if (!first) first = b + 1u; }
  }
  if (bad_in) atomicAdd(&counts[0], (unsigned long long)bad_in);
  if (n_out) atomicAdd(&counts[1], (unsigned long long)n_out);
  if (bad_out) atomicAdd(&counts[2], (unsigned long long)bad_out);
  if (first) atomicMax(&counts[3], (unsigned long long)first);
}

// accum_quantize_dev(x) against rt_accum_quantize(x) for all 2^32 bit patterns: counts[0] = patterns that differ,
// counts[1] = first differing pattern + 1.
__global__ void rt_test_quantize_sweep_kernel(unsigned long long *counts) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t bad = 0, first = 0;
  for (uint32_t k = 0; k < 256u; k++) {
    const uint32_t b = tid * 256u + k;
    const float x = __uint_as_float(b);
    if (accum_quantize_dev(x) != (unsigned long long)rt_accum_quantize(x)) { bad += 1; if (!first) first = b + 1u; }
  }
  if (bad) atomicAdd(&counts[0], (unsigned long long)bad);
  if (first) atomicMax(&counts[1], (unsigned long long)first);
}

// srgb_to_linear_tex1(x) against rt_srgb_to_linear1(x) for every float in [0, 2] and in [-0.046875, -0.03125]: counts[0] =
// patterns compared (2^30 + 2^22), counts[1] = patterns that differ, counts[2] = first differing pattern + 1.
__global__ void rt_test_srgb_sweep_kernel(unsigned long long *counts) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;       // 2^22 threads x 256 patterns = [0, 0x40000000)
  uint32_t n = 0, bad = 0, first = 0;
  for (uint32_t k = 0; k < 257u; k++) {
    uint32_t b = tid * 256u + k;
    if (k == 256u) b = tid == 0 ? 0x40000000u : 0xBD000000u + tid - 1u;        // 2.0, and 2^22 - 1 negative values from -0.03125 down
    const float x = __uint_as_float(b);
    const uint32_t w = __float_as_uint(rt_srgb_to_linear1(x)), g = __float_as_uint(srgb_to_linear_tex1(x));
    n += 1;
    if (w != g) { bad += 1; if (!first) first = b + 1u; }
  }
  atomicAdd(&counts[0], (unsigned long long)n);
  if (bad) atomicAdd(&counts[1], (unsigned long long)bad);
  if (first) atomicMax(&counts[2], (unsigned long long)first);
}

__global__ __launch_bounds__(RT_BLOCK_THREADS) void rt_test_trace_kernel(RT_KParams P, int n, const float *rays,
                                                                         float *out_t, int *out_tri, float *out_uv) {
  __shared__ uint32_t s_perm[RT_BLOCK_WAVES][RT_MAX_DEPTH * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  LaneCounters cn;
  cn.rays = cn.nodes = cn.leaves = cn.shades = cn.bgs = cn.textured = cn.paths = 0;
  if (i >= n) return;
  Ray3 r;
  ray_setup(r, rt_v3_make(rays[i * 6 + 0], rays[i * 6 + 1], rays[i * 6 + 2]),
            rt_v3_make(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]));
  HitRec hit;
  if (r.fast) trace_ray<true>(P, r, hit, s_perm[wave], lane, cn);
  else trace_ray<false>(P, r, hit, s_perm[wave], lane, cn);
  out_t[i] = hit.t;
  out_tri[i] = hit.tri;
  out_uv[i * 2 + 0] = hit.u;
  out_uv[i * 2 + 1] = hit.v;
}

__global__ void rt_test_texture_kernel(RT_KParams P, int tex, int n, const float *uv, float *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rt_v3 c = tex_bilinear(P, tex, uv[i * 2], uv[i * 2 + 1]);
  out[i * 3 + 0] = c.x;
  out[i * 3 + 1] = c.y;
  out[i * 3 + 2] = c.z;
}

// ---------------------------------------------------------------------------------
// launchers (called from rt_api.cpp)

// variant 1: plain while-while kernel; 2: phase-scheduled, 256-thread workgroups, nodes from L1/L2;
// 3: phase-scheduled, one 1024-thread workgroup per CU with the top of the BVH in LDS (default);
// 4: variant 3 plus block statistics (diagnostic).  (Occupancy experiments -- 5 or 6 waves per SIMD with register
// spills and no LDS node copy, two half-size LDS copies per CU -- lost to variant 3: numbers in DESIGN.md.)
// n_waves = total wavefronts wanted; smem_bytes = dynamic LDS per workgroup (variants >= 2).
template <int WAVES, bool LDSN, bool STATS, int MINW>
static int launch_sched(const RT_KParams *P, int n_waves, int smem_bytes, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set && smem_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rt_path_kernel_sched<WAVES, LDSN, STATS, MINW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((rt_path_kernel_sched<WAVES, LDSN, STATS, MINW>), dim3((n_waves + WAVES - 1) / WAVES), dim3(WAVES * 64),
                     smem_bytes, stream, *P);
  return (int)hipGetLastError();
}

// per launch: no chunk handed out yet; every group of 64 tiles is open
__global__ void rt_stream_init_kernel(int n_tiles, uint32_t *tile_next, uint32_t *open_groups) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_tiles) tile_next[i] = 0u;
  int n_groups = (n_tiles + 63) >> 6;
  if (i < n_groups) {
    int left = n_tiles - i * 64;
    open_groups[i] = (uint32_t)(left < 64 ? left : 64);
  }
}

extern "C" int rt_launch_stream_init(int n_tiles, uint32_t *tile_next, uint32_t *open_groups, hipStream_t stream) {
  hipLaunchKernelGGL(rt_stream_init_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, n_tiles, tile_next, open_groups);
  return (int)hipGetLastError();
}

template <int WAVES, bool LDSN, int MINW, bool SHORT_DIV>
static int launch_stream(const RT_KParams *P, int n_waves, int smem_bytes, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set && smem_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rt_path_kernel_stream<WAVES, LDSN, MINW, SHORT_DIV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((rt_path_kernel_stream<WAVES, LDSN, MINW, SHORT_DIV>), dim3((n_waves + WAVES - 1) / WAVES),
                     dim3(WAVES * 64), smem_bytes, stream, *P);
  return (int)hipGetLastError();
}

extern "C" int rt_launch_path_kernel(const RT_KParams *P, int n_waves, int variant, int smem_bytes, hipStream_t stream) {
  switch (variant) {
  case 5:
    return P->short_div ? launch_stream<16, true, 1, true>(P, n_waves, smem_bytes, stream)
                        : launch_stream<16, true, 1, false>(P, n_waves, smem_bytes, stream);
  case 1:
    hipLaunchKernelGGL(rt_path_kernel, dim3((n_waves + 3) / 4), dim3(RT_BLOCK_THREADS), 0, stream, *P);
    return (int)hipGetLastError();
  case 2: return launch_sched<4, false, false, 1>(P, n_waves, smem_bytes, stream);
  case 4: return launch_sched<16, true, true, 1>(P, n_waves, smem_bytes, stream);
  default: return launch_sched<16, true, false, 1>(P, n_waves, smem_bytes, stream);
  }
}

extern "C" int rt_launch_resolve(int width, int height, int samples, int chunks_x, const int32_t *local_chunks,
                                 int n_local_chunks, const unsigned long long *accum, uint8_t *tiles,
                                 uint8_t *image, float *linear, hipStream_t stream) {
  int n = n_local_chunks * 1024;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(rt_resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, width, height, samples,
                     chunks_x, local_chunks, n_local_chunks, accum, tiles, image, linear);
  return (int)hipGetLastError();
}

extern "C" int rt_launch_untile(int width, int height, int chunks_x, int n_chunks, const int32_t *owner_slot,
                                const uint8_t *all_tiles, uint8_t *image, hipStream_t stream) {
  int n = n_chunks * 1024;
  hipLaunchKernelGGL(rt_untile_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, width, height, chunks_x,
                     n_chunks, owner_slot, all_tiles, image);
  return (int)hipGetLastError();
}

// raw u8 image rows (stride pixels x comp bytes, comp >= 3) -> RGBA8 words, row-major width x height
__global__ void rt_pack_texture_kernel(const uint8_t *raw, int width, int height, int stride, int comp, uint32_t *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)width * height) return;
  int y = (int)(i / width), x = (int)(i % width);
  const uint8_t *p = raw + ((size_t)y * stride + x) * comp;
  out[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xFF000000u;
}

extern "C" int rt_launch_pack_texture(const uint8_t *raw, int width, int height, int stride, int comp, uint32_t *out,
                                      hipStream_t stream) {
  size_t n = (size_t)width * height;
  hipLaunchKernelGGL(rt_pack_texture_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, raw, width, height, stride,
                     comp, out);
  return (int)hipGetLastError();
}

extern "C" int rt_launch_test_math(int op, int n, const float *x, const float *y, float *out, hipStream_t stream) {
  hipLaunchKernelGGL(rt_test_math_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, op, n, x, y, out);
  return (int)hipGetLastError();
}

extern "C" int rt_launch_test_rcp_sweep(unsigned long long *counts, hipStream_t stream) {
  hipLaunchKernelGGL(rt_test_rcp_sweep_kernel, dim3(65536), dim3(256), 0, stream, counts);
  return (int)hipGetLastError();
}

extern "C" int rt_launch_test_quantize_sweep(unsigned long long *counts, hipStream_t stream) {
  hipLaunchKernelGGL(rt_test_quantize_sweep_kernel, dim3(65536), dim3(256), 0, stream, counts);
  return (int)hipGetLastError();
}

extern "C" int rt_launch_test_srgb_sweep(unsigned long long *counts, hipStream_t stream) {
  hipLaunchKernelGGL(rt_test_srgb_sweep_kernel, dim3(16384), dim3(256), 0, stream, counts);
  return (int)hipGetLastError();
}

extern "C" int rt_launch_test_trace(const RT_KParams *P, int n, const float *rays, float *out_t, int *out_tri,
                                    float *out_uv, hipStream_t stream) {
  hipLaunchKernelGGL(rt_test_trace_kernel, dim3((n + RT_BLOCK_THREADS - 1) / RT_BLOCK_THREADS),
                     dim3(RT_BLOCK_THREADS), 0, stream, *P, n, rays, out_t, out_tri, out_uv);
  return (int)hipGetLastError();
}

extern "C" int rt_launch_test_texture(const RT_KParams *P, int tex, int n, const float *uv, float *out,
                                      hipStream_t stream) {
  hipLaunchKernelGGL(rt_test_texture_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, *P, tex, n, uv, out);
  return (int)hipGetLastError();
}
