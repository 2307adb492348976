// rt_dev.hip.h -- device functions shared by the gfx950 kernels of the render hot path (rt_kernels.hip: the
// tile-stream path kernel, lightmap, unit-test kernels; rt_wavefront.hip: the split traversal / shading kernels;
// rt_kernels_diag.hip: the superseded kernel generations kept for bisecting).  Everything here is
// __device__ __forceinline__: each translation unit gets its own copy.
//
// Reference functions restated here (per-lane arithmetic identical to oracle/oracle.c):
//   ray_aabbs_hit_8      raytracer.c:190-230   slab_entry, node_enter, node_enter_few
//   ray_triangles_hit_8  raytracer.c:84-188    tri_test, leaf_test, leaf_test_short_div
//   ray_bvh_node_hit     raytracer.c:443-483   trace_ray (plain loop; the path kernels schedule the same steps per phase)
//   cast_ray             raytracer.c:505-558   shade_hit
//   disney_shader_proc & friends  driver.c:49-418
#ifndef RT_DEV_HIP_H
#define RT_DEV_HIP_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"
#include "../../include/rt_math.h"

#ifndef RT_SCALAR_TEX
#define RT_SCALAR_TEX 1
#endif
#ifndef RT_SCALAR_MATS
#define RT_SCALAR_MATS 1
#endif
#ifndef RT_LEAF_PAIRS
#define RT_LEAF_PAIRS 0      // 1: leaf tiles fetched by lane pairs (leaf_test_pair) -- bit-exact, measured slower (profiles/r03_experiments.md)
#endif

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
#define CNT_SKIPPED_ROOT 7   // of CNT_NODES: root visits of camera paths whose tile's pyramid misses the root -- counted, not executed

struct Ray3 {
  rt_v3 o, d;
  float inv_x, inv_y, inv_z;
  bool  fast;     // rt_slab_fast(): reciprocal direction and slab bias -(o * inv) all finite: NaN-free slab arithmetic
};

// The slab biases -(o * inv) of numeric contract v2 (rt_math.h): recomputed where a block needs them (three
// multiplications per block) instead of kept in three more registers per lane across the whole loop.
// (The products are loop invariants of the traversal loop: left to the compiler they are hoisted out of it and kept -- the
// three registers this function exists to save, 47 more spilled VGPRs when it was tried.  A volatile asm stays where it is;
// v_mul_f32 is the instruction the compiler emits for o * inv, the negation rides on the fma's source modifier.)
__device__ __forceinline__ rt_v3 slab_bias3(const Ray3 &r) {
#ifdef RT_MATH_NO_FMA
  return rt_v3_make(0.0f, 0.0f, 0.0f);           // contract v1: rt_slab_t_fast() does not read it
#elif defined(RT_BIAS_PLAIN)
  return rt_v3_make(rt_slab_bias(r.o.x, r.inv_x), rt_slab_bias(r.o.y, r.inv_y), rt_slab_bias(r.o.z, r.inv_z));
#else
  float px, py, pz;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(px) : "v"(r.o.x), "v"(r.inv_x));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(py) : "v"(r.o.y), "v"(r.inv_y));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(pz) : "v"(r.o.z), "v"(r.inv_z));
  return rt_v3_make(-px, -py, -pz);
#endif
}

struct HitRec {
  float t;
  int   tri;
  float u, v;
};

struct LaneCounters {
  uint32_t rays, nodes, leaves, shades, bgs, textured, paths;
};

// ---- block ledger (-DRT_LEDGER builds only; tools/exp_ledger.py, profiles/r04_blocks.md) ----------------------------------------
// Wave-level counters (scalar registers) of how often every block of the tile-stream kernel runs and with how many lanes, plus
// comment markers in the ISA (LGM) from which tools/ledger_static.py counts the instructions of each block.  The product
// build compiles none of this.
enum {
  // group 1: the S block
  LG_S_ITER = 0, LG_ENV_X, LG_ENV_L, LG_SHADE_X, LG_SHADE_L, LG_PSTORE_X, LG_PSTORE_L, LG_PLOAD_X, LG_PLOAD_L, LG_ACCUM_X, LG_ACCUM_L,
  // group 2: regeneration, ray start, tiles, loop rounds
  LG_REGEN_X, LG_REGEN_L, LG_START_X, LG_START_L, LG_PRIM_X, LG_PRIM_L, LG_TILE_X, LG_JOIN_X, LG_FLUSH_X, LG_GRAB_X, LG_ROUND_X, LG_TRAV_CALLS,
  // group 3: full node blocks
  LG_NFULL_X, LG_NFULL_L, LG_NFULL_CAM, LG_NGLOB_X, LG_NGLOB_L, LG_NEXACT_X, LG_NEXACT_L, LG_CULLMASK_X, LG_PYRCHK_X, LG_NODE_WAIT_L,
  // group 4: pyramid-culled node blocks by number of surviving children
  LG_NFEW0_X, LG_NFEW1_X, LG_NFEW2_X, LG_NFEW3_X, LG_NFEW4_X, LG_NFEW0_L, LG_NFEW1_L, LG_NFEW2_L, LG_NFEW3_L, LG_NFEW4_L,
  // group 5: leaf blocks and pop loops
  LG_LEAF_X, LG_LEAF_L, LG_LEAF_CAM, LG_POP_X, LG_POP_L, LG_POP_UP_L, LG_POP_RETEST_L, LG_POP_CAM, LG_POP_UP_X, LG_POP_RETEST_X, LG_POP_DONE_L,
  // group 6: shader-clock cycles (>> 4) per kind of block, summed over waves
  LG_CYC_S, LG_CYC_NODE, LG_CYC_LEAF, LG_CYC_POP, LG_CYC_WAVE, LG_CYC_TILE,
  // the loop of tiles whose pyramid misses the root (round 4): batches of up to 64 camera paths, primary ray -> environment
  LG_SKY_X, LG_SKY_L, LG_CYC_SKY,
  // small-launch ledger (round 5, tools/exp_small.py): the workgroup's copy of the tree into LDS, the join scans, and the DRAIN of a
  // tile -- from the moment its last unit is handed out to this wave until the wave's last path of it has ended (RT_LEDGER >= 2)
  LG_CYC_COPY, LG_CYC_JOIN, LG_CYC_DRAIN, LG_DRAIN_X, LG_DRAIN_L, LG_CYC_FLUSH,
  LG_N
};
#define RT_LEDGER_ROW 128      /* dwords per wave in g_ledger (LG_N <= 128 = RT_N_COUNTERS - 8) */
// The counters live in MEMORY, one row of 64 dwords per wave (g_ledger, rt_kernels.hip), bumped by lane 0 with atomics that
// return nothing: as scalar registers they did not fit beside the kernel's own (59 of them: 286 spilled VGPRs; even a
// dozen: 50-80), which would have measured a different kernel.  -DRT_LEDGER=1 counts blocks and lanes, =2 adds the
// shader-clock cycles per kind of block (s_memtime around the blocks: perturbs more).
#ifdef RT_LEDGER
__device__ __forceinline__ int lane_now();
__device__ __forceinline__ void lg_add(uint32_t *lg, int slot, uint32_t n) {
  if (lane_now() == 0 && n != 0u) __hip_atomic_fetch_add(lg + slot, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define LG(slot, n) lg_add(lg, (slot), (uint32_t)(n))
#define LG_ARG , lg
#else
#define LG(slot, n) ((void)0)
#define LG_ARG
#endif
#ifdef RT_LEDGER_MARKS          /* block boundaries as comments in the ISA of an otherwise unchanged product kernel (tools/ledger_static.py) */
#define LGM(name) asm volatile("; LEDGER_MARK " name)
#else
#define LGM(name) ((void)0)
#endif
#if defined(RT_LEDGER) && RT_LEDGER >= 2
#define LGT0() const unsigned long long lg_t0 = __builtin_amdgcn_s_memtime()
#define LGT1(slot) LG((slot), (uint32_t)((__builtin_amdgcn_s_memtime() - lg_t0) >> 4))
#define LGD0(n_live) do { if (!lg_drain_t0) { lg_drain_t0 = __builtin_amdgcn_s_memtime(); LG(LG_DRAIN_X, 1); LG(LG_DRAIN_L, (n_live)); } } while (0)
#define LGD1() do { if (lg_drain_t0) { LG(LG_CYC_DRAIN, (uint32_t)((__builtin_amdgcn_s_memtime() - lg_drain_t0) >> 4)); lg_drain_t0 = 0ull; } } while (0)
#else
#define LGT0() ((void)0)
#define LGT1(slot) ((void)0)
#define LGD0(n_live) ((void)0)
#define LGD1() ((void)0)
#endif

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
//  * FAST uses v_min_f32 / v_max3_f32, and under numeric contract v2 (rt_math.h) takes every plane
//    distance from one fused multiply-add, fma(plane, inv, -(o * inv)) -- the oracle does the same for
//    the same rays (rt_slab_fast / rt_slab_t_fast are shared).  It is taken only for rays whose
//    reciprocal direction and slab bias are all finite (Ray3::fast): then no NaN can appear in
//    the slab arithmetic, and on NaN-free operands min/max are plain min/max (sign of zero cannot
//    matter: every distance is clamped to >= EPSILON before it is used).  Axis-aligned rays (0 * inf)
//    take the EXACT path; tests/test_gpu_parity.py sends such rays.

__device__ __forceinline__ float fmin_hw(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float fmax_hw(float a, float b) { return __builtin_fmaxf(a, b); }

template <bool FAST>
__device__ __forceinline__ float slab_entry(const Ray3 &r, const rt_v3 &bs, float mnx, float mny, float mnz,
                                            float mxx, float mxy, float mxz, float t_max) {
  if (FAST) {
    float t0x = rt_slab_t_fast(mnx, r.o.x, r.inv_x, bs.x), t1x = rt_slab_t_fast(mxx, r.o.x, r.inv_x, bs.x);
    float t0y = rt_slab_t_fast(mny, r.o.y, r.inv_y, bs.y), t1y = rt_slab_t_fast(mxy, r.o.y, r.inv_y, bs.y);
    float t0z = rt_slab_t_fast(mnz, r.o.z, r.inv_z, bs.z), t1z = rt_slab_t_fast(mxz, r.o.z, r.inv_z, bs.z);
    float sx = fmin_hw(t0x, t1x), sy = fmin_hw(t0y, t1y), sz = fmin_hw(t0z, t1z);
    float bx = fmax_hw(t0x, t1x), by = fmax_hw(t0y, t1y), bz = fmax_hw(t0z, t1z);
    float t_minv = fmax_hw(RT_EPS, fmax_hw(sx, fmax_hw(sy, sz)));
    float t_maxv = fmin_hw(t_max, fmin_hw(bx, fmin_hw(by, bz)));
    // t_maxv <= t_max, so "entry < exit" already implies the candidate test entry < t_max
    return (t_minv < t_maxv) ? t_minv : RT_INF;
  } else {
    // ONLY for rays that are not NaN-free (Ray3::fast false): under contract v2 the two forms round differently, so a
    // NaN-free ray must never come through here (callers branch per lane: traversal_blocks, trace_ray)
    float t0x = rt_slab_t_exact(mnx, r.o.x, r.inv_x), t1x = rt_slab_t_exact(mxx, r.o.x, r.inv_x);
    float t0y = rt_slab_t_exact(mny, r.o.y, r.inv_y), t1y = rt_slab_t_exact(mxy, r.o.y, r.inv_y);
    float t0z = rt_slab_t_exact(mnz, r.o.z, r.inv_z), t1z = rt_slab_t_exact(mxz, r.o.z, r.inv_z);
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
// FAST: only for rays with Ray3::fast; the other instantiation only for rays without (the two forms round differently
// under contract v2).  slab_entry_child_any() branches per lane.
template <bool FAST>
__device__ __forceinline__ float slab_entry_child(const float *n, const Ray3 &r) {
  // n -> element j of the node's first row; the six rows are 8 floats apart
  float mnx = n[0], mny = n[8], mnz = n[16], mxx = n[24], mxy = n[32], mxz = n[40];
  if (FAST) {
    const rt_v3 bs = slab_bias3(r);
    float t0x = rt_slab_t_fast(mnx, r.o.x, r.inv_x, bs.x), t1x = rt_slab_t_fast(mxx, r.o.x, r.inv_x, bs.x);
    float t0y = rt_slab_t_fast(mny, r.o.y, r.inv_y, bs.y), t1y = rt_slab_t_fast(mxy, r.o.y, r.inv_y, bs.y);
    float t0z = rt_slab_t_fast(mnz, r.o.z, r.inv_z, bs.z), t1z = rt_slab_t_fast(mxz, r.o.z, r.inv_z, bs.z);
    float sx = fmin_hw(t0x, t1x), sy = fmin_hw(t0y, t1y), sz = fmin_hw(t0z, t1z);
    return fmax_hw(RT_EPS, fmax_hw(sx, fmax_hw(sy, sz)));
  }
  float t0x = rt_slab_t_exact(mnx, r.o.x, r.inv_x), t1x = rt_slab_t_exact(mxx, r.o.x, r.inv_x);
  float t0y = rt_slab_t_exact(mny, r.o.y, r.inv_y), t1y = rt_slab_t_exact(mxy, r.o.y, r.inv_y);
  float t0z = rt_slab_t_exact(mnz, r.o.z, r.inv_z), t1z = rt_slab_t_exact(mxz, r.o.z, r.inv_z);
  float sx = rt_min_ps(t0x, t1x), sy = rt_min_ps(t0y, t1y), sz = rt_min_ps(t0z, t1z);
  return rt_max_ps(RT_EPS, rt_max_ps(sx, rt_max_ps(sy, sz)));
}
__device__ __forceinline__ float slab_entry_child_any(const float *n, const Ray3 &r) {
  if (r.fast) return slab_entry_child<true>(n, r);
  return slab_entry_child<false>(n, r);
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
  const rt_v3 bs = FAST ? slab_bias3(r) : rt_v3_make(0.0f, 0.0f, 0.0f);
  if (MODE == NODE_SCALAR) {               // `node` is wave-uniform: node data lives in SGPRs
    cfloat *nb = as_scalar_ptr(P.nodes) + (size_t)node * 48;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      d[k] = as_i(slab_entry<FAST>(r, bs, nb[k], nb[8 + k], nb[16 + k], nb[24 + k], nb[32 + k], nb[40 + k], hit_t));
    }
  } else if (FAST && MODE == NODE_LDS_ORDERED) {
    // Near and far plane of every slab picked by ADDRESS from the sign of the reciprocal direction instead of by min / max
    // of the two distances: with min <= max in every box (checked at upload, rt_api.cpp) and NaN-free operands,
    // t(mn) <= t(mx) for inv > 0 and >= for inv < 0 -- rounding is monotonic, for (p - o) * inv and for fma(p, inv, bias)
    // alike -- so the picked distance IS the minimum (maximum); for inv = 0 both are equal.  Six min / max fewer per child.
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
        const float sx = rt_slab_t_fast(nxs[k], r.o.x, r.inv_x, bs.x), bxx = rt_slab_t_fast(fxs[k], r.o.x, r.inv_x, bs.x);
        const float sy = rt_slab_t_fast(nys[k], r.o.y, r.inv_y, bs.y), byy = rt_slab_t_fast(fys[k], r.o.y, r.inv_y, bs.y);
        const float sz = rt_slab_t_fast(nzs[k], r.o.z, r.inv_z, bs.z), bzz = rt_slab_t_fast(fzs[k], r.o.z, r.inv_z, bs.z);
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
      d[h * 4 + 0] = as_i(slab_entry<FAST>(r, bs, mnx.x, mny.x, mnz.x, mxx.x, mxy.x, mxz.x, hit_t));
      d[h * 4 + 1] = as_i(slab_entry<FAST>(r, bs, mnx.y, mny.y, mnz.y, mxx.y, mxy.y, mxz.y, hit_t));
      d[h * 4 + 2] = as_i(slab_entry<FAST>(r, bs, mnx.z, mny.z, mnz.z, mxx.z, mxy.z, mxz.z, hit_t));
      d[h * 4 + 3] = as_i(slab_entry<FAST>(r, bs, mnx.w, mny.w, mnz.w, mxx.w, mxy.w, mxz.w, hit_t));
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
// rcp_exact() without the v_div_fixup_f32, for the leaf blocks only: equal to 1.0f / x for every finite non-zero |x| < 2^102
// (the fix-up passes those through); NaN instead of +-infinity for x = +-0 and instead of +-0 for x = +-infinity.  A
// determinant of zero or infinity is no hit either way -- with inv_det = infinity u, v, t are infinite or NaN: an infinite
// u, v or t fails one of the five comparisons or is not below `best`, a NaN t is not below `best`; with inv_det = 0 the
// distance is 0 < epsilon; with NaN everything is NaN and t < best is false -- so ray_triangles_hit_8's outcome is the same.
__device__ __forceinline__ float rcp_leaf(float x) {
#ifdef RT_LEAF_SANITISE
  return rcp_exact(x);
#else
  float xs = x * 0x1p24f;
  float y = __builtin_amdgcn_rcpf(xs);
  float e = __builtin_fmaf(-xs, y, 1.0f);
  float z = __builtin_fmaf(e, y, y);
  return z * 0x1p24f;
#endif
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
      float inv_det = rcp_leaf(det);
      rt_v3 s = rt_v3_sub(r.o, rt_v3_make(ax[k], ay[k], az[k]));
      rt_v3 sxe1 = rt_v3_cross(s, edge1);
      float u = inv_det * rt_v3_dot(s, rxe2);
      float v = inv_det * rt_v3_dot(r.d, sxe1);
      float t = inv_det * rt_v3_dot(edge2, sxe1);
      bool miss = (u < -RT_EPS) || (u > 1.0f + RT_EPS) || (v < -RT_EPS) || (u + v > 1.0f + RT_EPS) || (t < RT_EPS);
#ifdef RT_LEAF_SANITISE
      float dist = miss ? RT_INF : t;
      dist = (dist > 0.0f) ? dist : RT_INF;          // NaN -> +inf (min_f32x8)
      if (dist < best) { best = dist; bi = h * 4 + k; bu = u; bv = v; }   // lowest lane wins ties
#else
      // min_f32x8's sanitising (lanes <= epsilon or NaN -> +inf, raytracer.c:15-32) folded into the comparison: a triangle
      // that is not a miss has t >= epsilon or t NaN, `best` is never NaN, and NaN < best is false like +inf < best --
      // the same triangle wins with the same t, three vector instructions fewer per triangle
      if (!miss && t < best) { best = t; bi = h * 4 + k; bu = u; bv = v; }   // lowest lane wins ties
#endif
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

__device__ __forceinline__ int lane_now();

// ---- leaf block with the tile fetched by lane PAIRS --------------------------------------------------------------
// What a vector load costs the CU's memory pipe is set by the number of distinct cache lines its 64 lanes touch -- about
// 1.24 cycles per line (tools/exp/leaf_fetch_bench.hip: 18 x dwordx4 from a different 288-byte tile per lane hold the
// pipe for 1 430 cycles, the 650 VALU instructions of the block hold the four SIMDs for 325) -- and every one of a lane's
// 18 loads touches its own line.  Here lanes 2k and 2k+1 share the work on BOTH their rays: each load instruction takes
// the two 16-byte halves of one tile row (32 contiguous bytes, one line) for the pair, so an instruction touches 32
// lines instead of 64 (measured 633 cycles per block); lane 2k tests triangles 0-3 of its own leaf and of its partner's,
// lane 2k+1 triangles 4-7 of both; the partner's result for my ray comes back through four DPP moves and is merged with
// the reference's tie rule (lowest triangle index wins, raytracer.c:27-29).  Same 18 loads and 8 triangle tests per
// lane, + 12 DPP moves and a merge; the arithmetic per triangle is tri_test()'s, bit for bit.
__device__ __forceinline__ int dpp_swap1_i(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true); }
__device__ __forceinline__ float dpp_swap1_f(float v) { return as_f(dpp_swap1_i(as_i(v))); }

// best of the 4 triangles of half `h` of leaf group g for the ray (o, d): sanitised distance (+inf: none), u, v, index in the group
template <bool SHORT_DIV>
__device__ __forceinline__ void leaf_half_test(const float *leaves, rt_v3 o, rt_v3 d, int g, int h, float &best, float &bu, float &bv, int &bi) {
  best = RT_INF; bu = 0.0f; bv = 0.0f; bi = 0;
  const float *lb = leaves + (size_t)g * 72 + h * 4;
  float4 x0 = ld4(lb, 0), x1 = ld4(lb, 2), x2 = ld4(lb, 4);
  float4 y0 = ld4(lb, 6), y1 = ld4(lb, 8), y2 = ld4(lb, 10);
  float4 z0 = ld4(lb, 12), z1 = ld4(lb, 14), z2 = ld4(lb, 16);
  float ax[4] = {x0.x, x0.y, x0.z, x0.w}, bx[4] = {x1.x, x1.y, x1.z, x1.w}, cx[4] = {x2.x, x2.y, x2.z, x2.w};
  float ay[4] = {y0.x, y0.y, y0.z, y0.w}, by[4] = {y1.x, y1.y, y1.z, y1.w}, cy[4] = {y2.x, y2.y, y2.z, y2.w};
  float az[4] = {z0.x, z0.y, z0.z, z0.w}, bz[4] = {z1.x, z1.y, z1.z, z1.w}, cz[4] = {z2.x, z2.y, z2.z, z2.w};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    rt_v3 edge1 = rt_v3_make(bx[k], by[k], bz[k]), edge2 = rt_v3_make(cx[k], cy[k], cz[k]);
    rt_v3 rxe2 = rt_v3_cross(d, edge2);
    float det = rt_v3_dot(edge1, rxe2);
    float inv_det = SHORT_DIV ? rcp_exact(det) : 1.0f / det;
    rt_v3 s = rt_v3_sub(o, rt_v3_make(ax[k], ay[k], az[k]));
    rt_v3 sxe1 = rt_v3_cross(s, edge1);
    float u = inv_det * rt_v3_dot(s, rxe2);
    float v = inv_det * rt_v3_dot(d, sxe1);
    float t = inv_det * rt_v3_dot(edge2, sxe1);
    bool miss = (u < -RT_EPS) || (u > 1.0f + RT_EPS) || (v < -RT_EPS) || (u + v > 1.0f + RT_EPS) || (t < RT_EPS);
    float dist = miss ? RT_INF : t;
    dist = (dist > 0.0f) ? dist : RT_INF;          // NaN -> +inf (min_f32x8)
    if (dist < best) { best = dist; bi = h * 4 + k; bu = u; bv = v; }   // lowest lane wins ties
  }
}

// Called by ALL lanes of the wave (uniform control flow); `in_leaf`: this lane's ray wants leaf group g tested.
template <bool SHORT_DIV>
__device__ __forceinline__ bool leaf_test_pair(const RT_KParams &P, const Ray3 &r, int g, bool in_leaf, HitRec &hit) {
  const int h = lane_now() & 1;
  // the partner's ray and leaf
  const bool p_act = dpp_swap1_i(in_leaf ? 1 : 0) != 0;
  const int  pg = dpp_swap1_i(g);
  const rt_v3 po = rt_v3_make(dpp_swap1_f(r.o.x), dpp_swap1_f(r.o.y), dpp_swap1_f(r.o.z));
  const rt_v3 pd = rt_v3_make(dpp_swap1_f(r.d.x), dpp_swap1_f(r.d.y), dpp_swap1_f(r.d.z));
  float bestA = RT_INF, uA = 0.0f, vA = 0.0f, bestB = RT_INF, uB = 0.0f, vB = 0.0f;
  int   iA = 0, iB = 0;
  if (in_leaf) leaf_half_test<SHORT_DIV>(P.leaves, r.o, r.d, g, h, bestA, uA, vA, iA);       // my half of my leaf
  if (p_act) leaf_half_test<SHORT_DIV>(P.leaves, po, pd, pg, h, bestB, uB, vB, iB);          // my half of the partner's leaf
  // the partner's half of MY leaf
  const float bestP = dpp_swap1_f(bestB), uP = dpp_swap1_f(uB), vP = dpp_swap1_f(vB);
  const int   iP = dpp_swap1_i(iB);
  bool got = false;
  if (in_leaf) {
    // triangles 0-3 (lane 2k's half) come before 4-7 in the reference's scan: on equal distance the lower half wins
    const bool take_p = h ? (bestP <= bestA) : (bestP < bestA);
    const float best = take_p ? bestP : bestA;
    if (best < hit.t) {
      hit.t = best;
      hit.tri = g * 8 + (take_p ? iP : iA);
      hit.u = take_p ? uP : uA;
      hit.v = take_p ? vP : vA;
      got = true;
    }
  }
  return got;
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

// sample_texture_bilinear (driver.c:49-93) in three steps, so that a caller with several textures can put ALL their texel
// loads in flight before it consumes the first one (shade(): 16 loads, one wait, instead of four dependent round trips):
//   tex_taps()    wrap rules, the four texel indices and the two weights
//   tex_fetch()   the four loads
//   tex_combine() u8 / 255.999, lerp x then y (same order, same bits as before)
struct TexTaps { int i00, i10, i01, i11; float a, b; };
struct TexQuad { uint32_t q00, q10, q01, q11; };

__device__ __forceinline__ TexTaps tex_taps(const RT_DTexture &T, float tx, float ty) {
  if (tx < 0) tx += (float)(-(int)tx + 1);
  if (ty < 0) ty += (float)(-(int)ty + 1);
  tx = rt_fractf(tx);
  ty = rt_fractf(ty);
  float px = tx * (float)T.width;
  float py = ty * (float)T.height;
  int u = (int)px, v = (int)py;
  if (u > T.width - 1) u = T.width - 1;
  if (v > T.height - 1) v = T.height - 1;
  TexTaps t;
  t.a = px - (float)u;
  t.b = py - (float)v;
#if RT_TEX_TILED
  // 4 x 4 tiles (rt_device.h): the neighbour to the right is the next texel of the tile, or the first of the tile's row in
  // the next tile (+ 16 - 3); the one below the next row of the tile, or the first row of the tile below; clamp to edge = 0
  const int du = (u + 1 < T.width) ? (((u & 3) == 3) ? 13 : 1) : 0;
  const int dv = (v + 1 < T.height) ? (((v & 3) == 3) ? T.stride * 16 - 12 : 4) : 0;
  t.i00 = RT_TEX_TILE_INDEX(u, v, T.stride);
  t.i10 = t.i00 + du;
  t.i01 = t.i00 + dv;
  t.i11 = t.i00 + du + dv;
#else
  int u2 = (u + 1 < T.width) ? u + 1 : u;
  int v2 = (v + 1 < T.height) ? v + 1 : v;
  t.i00 = u + T.stride * v;
  t.i10 = u2 + T.stride * v;
  t.i01 = u + T.stride * v2;
  t.i11 = u2 + T.stride * v2;
#endif
  return t;
}
__device__ __forceinline__ TexQuad tex_fetch(const uint32_t *tp, const TexTaps &t) {
  TexQuad q;
  q.q00 = tp[t.i00]; q.q10 = tp[t.i10]; q.q01 = tp[t.i01]; q.q11 = tp[t.i11];
  return q;
}
__device__ __forceinline__ rt_v3 tex_combine(const TexQuad &q, const TexTaps &t) {
  rt_v3 c0 = rt_v3_lerp(texel_rgb(q.q00), texel_rgb(q.q10), t.a);
  rt_v3 c1 = rt_v3_lerp(texel_rgb(q.q01), texel_rgb(q.q11), t.a);
  return rt_v3_lerp(c0, c1, t.b);
}

template <class PT>
__device__ __forceinline__ rt_v3 tex_bilinear(const PT &P, int tex, float tx, float ty) {
  RT_DTexture T;
#if RT_SCALAR_TEX
  {
    // the descriptor of a texture every lane of the block samples comes through the scalar cache (see shade())
    const int t0 = __builtin_amdgcn_readfirstlane(tex);
    if (__ballot(tex != t0) == 0ull) {
      typedef const RT_DTexture __attribute__((address_space(4))) CT;
      CT *tp = (CT *)(unsigned long long)(P.textures + t0);
      T.offset = tp->offset; T.width = tp->width; T.height = tp->height; T.stride = tp->stride;
    } else {
      T = P.textures[tex];
    }
  }
#else
  T = P.textures[tex];
#endif
  const TexTaps t = tex_taps(T, tx, ty);
  const TexQuad q = tex_fetch(P.texels + T.offset, t);
  return tex_combine(q, t);
}

// rt_srgb_to_linear() of a bilinear texture sample: (x + 0.055f) / 1.055f as a multiplication by RN(1 / 1.055f) corrected
// with the exact residual (two FMAs) -- 3 instructions where the IEEE division takes 11.  The corrected quotient equals
// the division for every a = x + 0.055f with 2^-104 <= |a| < infinity and for NaN (tools/exp/div_test.hip, all 2^32 a);
// rt_test_srgb_sweep compares every x in [0, 2] -- 1.07 G bit patterns -- with rt_srgb_to_linear1().  A texture sample is a
// lerp of u8 / 255.999 values, 0 <= x <= 0.9961, or NaN for NaN texture coordinates.  Same rt_powf() afterwards.
// POW24_LDS: the scale 2^(-2.4 n) of rt_pow24_core() read from a table in LDS by the exponent of b (RT_POW24_SCALES, rt_math.h;
// the kernel fills rt_pow24_lds before its first barrier: pow24_lds_init) -- a shift, a mask and a ds_read_b32 for five compares,
// five selects and four constant moves.  The same constants: the same product.  Asked for through the parameter struct of the
// caller (Pow24InLds<PT>): only a kernel that fills the table may.
static __shared__ float rt_pow24_lds[8];
template <class PT> struct Pow24InLds { static constexpr bool value = false; };
__device__ __forceinline__ void pow24_lds_init(int tid) {        // before a __syncthreads() of the kernel
  const float scales[8] = RT_POW24_SCALES;
  if (tid < 8) {
    float v = 0.0f;
#pragma unroll
    for (int k = 2; k < 8; k++) v = (tid == k) ? scales[k] : v;
    rt_pow24_lds[tid] = v;
  }
}

template <bool POW24_LDS = false>
__device__ __forceinline__ float srgb_to_linear_tex1(float x) {
#ifdef RT_EXP_SRGB_IEEE
  return rt_srgb_to_linear1(x);
#endif
  const float c = 1.0f / 1.055f;
  float a = x + 0.055f;
  float q = a * c;
  float r = __builtin_fmaf(-1.055f, q, a);
#if RT_MATH_POW24
  // contract v3 (rt_math.h): b^2.4 by rt_pow24_core() -- a degree-6 polynomial in the mantissa times one of six constants -- for
  // every b in [2^-5, 2), which is every b a texture sample produces; NaN (NaN texture coordinates) and values no u8 texture
  // yields take rt_powf() in a branch the wave skips.  Same bits as rt_srgb_to_linear1() for every x in [0, 2]: rt_test_srgb_sweep.
  const float b = __builtin_fmaf(r, c, q);
  float p;
  if (POW24_LDS) p = rt_pow24_poly(b) * *reinterpret_cast<const float *>(reinterpret_cast<const char *>(rt_pow24_lds) + (((uint32_t)as_i(b) >> 21) & 28u));
  else p = rt_pow24_core(b);
  if (!(b >= 0x1p-5f && b < 2.0f)) p = rt_powf(b, 2.4f);
  return p;
#elif defined(RT_EXP_POW_PLAIN)
  return rt_powf(__builtin_fmaf(r, c, q), 2.4f);
#else
  // rt_powf(b, 2.4f) (rt_math.h) straight-line: its clamp of 2.4 log b to [-87, 87] as ONE v_med3_f32 (two compares and two
  // selects as written; equal for every operand that is not NaN, and log b is not NaN for b > 0), its `b > 0 else 0` as a
  // select at the end instead of a branch around the body (rt_logf of b <= 0 or NaN is garbage that the select drops).
  // Same bits for every x: rt_test_srgb_sweep.
  const float b = __builtin_fmaf(r, c, q);
  const float t = __builtin_amdgcn_fmed3f(2.4f * rt_logf(b), -87.0f, 87.0f);
  const float p = rt_expf(t);
  return (b > 0.0f) ? p : 0.0f;
#endif
}
template <bool POW24_LDS = false>
__device__ __forceinline__ rt_v3 srgb_to_linear_tex(rt_v3 v) {
  return rt_v3_make(srgb_to_linear_tex1<POW24_LDS>(v.x), srgb_to_linear_tex1<POW24_LDS>(v.y), srgb_to_linear_tex1<POW24_LDS>(v.z));
}

// driver.c:95-104
template <class PT>
__device__ __forceinline__ rt_v3 background_lookup(const PT &P, rt_v3 dir) {
  float inv_pi = 1.0f / RT_PI;
  float inv_two_pi = 1.0f / (2.0f * RT_PI);
  float u = rt_madd(rt_atan2f(dir.z, dir.x), inv_two_pi, 0.5f);
  float v = rt_madd(-rt_asinf(dir.y), inv_pi, 0.5f);
  return srgb_to_linear_tex<Pow24InLds<PT>::value>(tex_bilinear(P, P.bg_texture, u, v));
}

// ---------------------------------------------------------------------------------
// Disney-style BSDF, driver.c:118-348.  Expression order matches oracle/oracle.c.

__device__ __forceinline__ float pow5(float m) { return m * m * m * m * m; }
__device__ __forceinline__ float luminance(rt_v3 x) { return rt_v3_dot(x, rt_v3_make(0.2126f, 0.7152f, 0.0722f)); }

__device__ __forceinline__ float ggx_D(float roughness, float NoH) {          // driver.c:212-215, k = 2
  float a2 = roughness * roughness;
  float d = rt_madd(NoH * NoH, rt_madd(a2, a2, -1.0f), 1.0f);
  return a2 / (RT_PI * (d * d));
}

__device__ __forceinline__ float smith_G(float NDotV, float alpha2) {         // driver.c:217-221
  float a = alpha2 * alpha2;
  float b = NDotV * NDotV;
  return (2.0f * NDotV) / (NDotV + rt_sqrtf(rt_madd(-a, b, a + b)));
}

__device__ __forceinline__ rt_v3 cosine_hemisphere(uint32_t &rng) {           // driver.c:118-127
  float angle = rt_rand_f32(&rng) * 2.0f * RT_PI;
  float distance = rt_sqrtf(rt_rand_f32(&rng));
  float s, c;
  rt_sincosf(angle, &s, &c);
  return rt_v3_make(s * distance, c * distance, rt_sqrtf(rt_madd(-distance, distance, 1.0f)));
}

__device__ __forceinline__ rt_v3 ggx_vndf(rt_v3 V, float ax, float ay, uint32_t &rng) {  // driver.c:230-250
  rt_v3 Vh = normalize_dev(rt_v3_make(ax * V.x, ay * V.y, V.z));
  float lensq = rt_dot2(Vh.x, Vh.x, Vh.y, Vh.y);
  rt_v3 T1 = lensq > 0.0f ? rt_v3_scale(rt_v3_make(-Vh.y, Vh.x, 0.0f), rcp_exact(rt_sqrtf(lensq))) : rt_v3_make(1, 0, 0);
  rt_v3 T2 = rt_v3_cross(Vh, T1);
  float r = rt_sqrtf(rt_rand_f32(&rng));
  float phi = 2.0f * RT_PI * rt_rand_f32(&rng);
  float sn, cs;
  rt_sincosf(phi, &sn, &cs);
  float t1 = r * cs;
  float t2 = r * sn;
  float s = rt_madd(0.5f, Vh.z, 0.5f);
  t2 = rt_madd(1.0f - s, rt_sqrtf(rt_madd(-t1, t1, 1.0f)), s * t2);
  rt_v3 Nh = rt_v3_comb3(T1, t1, T2, t2, Vh, rt_sqrtf(rt_max_ps(0.0f, rt_madd(-t2, t2, rt_madd(-t1, t1, 1.0f)))));
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
  rt_v3 fresnel = rt_v3_madd(rt_v3_sub(rt_v3_make(f90, f90, f90), f0), pow5(1.0f - theta), f0);

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
    float FD90 = rt_madd(2.0f * m.roughness * LoH, LoH, 0.5f);
    float fa = rt_madd(FD90 - 1.0f, pow5(1.0f - NoL), 1.0f);
    float fb = rt_madd(FD90 - 1.0f, pow5(1.0f - NoV), 1.0f);
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
__device__ __forceinline__ rt_v3 normal_map(bool has_map, rt_v3 v, float strength, const ShadeIn &in) {
  rt_v3 normal = in.normal;
  if (has_map) {
    v = rt_v3_madd(v, 2.0f, rt_v3_make(-1.0f, -1.0f, -1.0f));
    v.y *= -1.0f;
    rt_v3 t = in.tangent, b = in.bitangent, n = in.normal;
    float s = strength;
    normal = normalize_dev(rt_v3_make(rt_madd(s, rt_dot3(v.x, t.x, v.y, b.x, v.z, n.x), n.x * (1.0f - s)),
                                        rt_madd(s, rt_dot3(v.x, t.y, v.y, b.y, v.z, n.y), n.y * (1.0f - s)),
                                        rt_madd(s, rt_dot3(v.x, t.z, v.y, b.z, v.z, n.z), n.z * (1.0f - s))));
  }
  return normal;
}

// disney_shader_proc driver.c:350-409 / debug_shader_proc :411-418 on material `mat`
template <class PT>
__device__ __forceinline__ void shade(const PT &P, int mat, const ShadeIn &in, uint32_t &rng,
                                      rt_v3 &out_dir, rt_v3 &tint, rt_v3 &emission, bool &terminate,
                                      LaneCounters &cn) {
  float4 m0, m1, m2, m3, m4;
  RT_DTexture D[4];              // albedo, normal, metal_roughness, emission: embedded in the material record (rt_device.h)
#if RT_SCALAR_MATS
  // one material for every lane of this block (the helmet has one material, most blocks of any scene have one): the
  // record -- parameters AND the descriptors of its four textures -- comes through the scalar cache in ONE round trip
  // (~100 cycles instead of a vector load's several hundred, and no second dependent load for the descriptors)
  const int mat0 = __builtin_amdgcn_readfirstlane(mat);
  if (__ballot(mat != mat0) == 0ull) {
    cfloat *sb = as_scalar_ptr(P.mats) + (size_t)mat0 * RT_MAT_FLOATS;
    m0 = make_float4(sb[0], sb[1], sb[2], sb[3]);
    m1 = make_float4(sb[4], sb[5], sb[6], sb[7]);
    m2 = make_float4(sb[8], sb[9], sb[10], sb[11]);
    m3 = make_float4(sb[12], sb[13], sb[14], sb[15]);
    m4 = make_float4(sb[16], sb[17], sb[18], sb[19]);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      D[k].offset = (uint32_t)as_i(sb[20 + 4 * k]); D[k].width = as_i(sb[21 + 4 * k]); D[k].height = as_i(sb[22 + 4 * k]); D[k].stride = as_i(sb[23 + 4 * k]);
    }
  } else
#endif
  {
    const float *mb = P.mats + (size_t)mat * RT_MAT_FLOATS;
    m0 = ld4(mb, 0); m1 = ld4(mb, 1); m2 = ld4(mb, 2); m3 = ld4(mb, 3); m4 = ld4(mb, 4);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float4 dk = ld4(mb, 5 + k);
      D[k].offset = (uint32_t)as_i(dk.x); D[k].width = as_i(dk.y); D[k].height = as_i(dk.z); D[k].stride = as_i(dk.w);
    }
  }
  int tex_albedo = as_i(m3.x), tex_normal = as_i(m3.y), tex_mr = as_i(m3.z), tex_em = as_i(m3.w);
  int kind = as_i(m4.x);

  // RT_TEX_BATCH: how many of the material's four texture fetches have their loads in flight together.  1 = one after the
  // other (four dependent round trips of four texels, round 3), 2 = normal + albedo, then metal-roughness + emission,
  // 4 = all sixteen loads before the first is consumed.  Same arithmetic in every case.
#ifndef RT_TEX_BATCH
#define RT_TEX_BATCH 2
#endif
  const bool dbg = kind == RT_MAT_DEBUG;
  const bool ha = tex_albedo >= 0 && !dbg, hn = tex_normal >= 0, hm = tex_mr >= 0 && !dbg, he = tex_em >= 0 && !dbg;
  TexTaps tn, ta, tm, te;
  TexQuad qn, qa, qm, qe;
  if (hn) { tn = tex_taps(D[1], in.uvx, in.uvy); qn = tex_fetch(P.texels + D[1].offset, tn); }
  if (RT_TEX_BATCH >= 2 && ha) { ta = tex_taps(D[0], in.uvx, in.uvy); qa = tex_fetch(P.texels + D[0].offset, ta); }
  if (RT_TEX_BATCH >= 4 && hm) { tm = tex_taps(D[2], in.uvx, in.uvy); qm = tex_fetch(P.texels + D[2].offset, tm); }
  if (RT_TEX_BATCH >= 4 && he) { te = tex_taps(D[3], in.uvx, in.uvy); qe = tex_fetch(P.texels + D[3].offset, te); }

  rt_v3 normal = normal_map(hn, hn ? tex_combine(qn, tn) : rt_v3_make(0, 0, 0), m2.x, in);
  terminate = false;
  tint = rt_v3_make(0, 0, 0);
  out_dir = rt_v3_make(0, 0, 0);

  if (dbg) {
    emission = rt_v3_madd(normal, 0.5f, rt_v3_make(0.5f, 0.5f, 0.5f));
    terminate = true;
    return;
  }

  if (tex_albedo >= 0 || tex_normal >= 0 || tex_mr >= 0 || tex_em >= 0) cn.textured += 1;

  if (RT_TEX_BATCH < 2 && ha) { ta = tex_taps(D[0], in.uvx, in.uvy); qa = tex_fetch(P.texels + D[0].offset, ta); }
  if (RT_TEX_BATCH == 2 && hm) { tm = tex_taps(D[2], in.uvx, in.uvy); qm = tex_fetch(P.texels + D[2].offset, tm); }
  if (RT_TEX_BATCH == 2 && he) { te = tex_taps(D[3], in.uvx, in.uvy); qe = tex_fetch(P.texels + D[3].offset, te); }
  rt_v3 base_color = rt_v3_make(m0.x, m0.y, m0.z);
  if (ha) base_color = rt_v3_mul(base_color, srgb_to_linear_tex<Pow24InLds<PT>::value>(tex_combine(qa, ta)));

  float roughness = m0.w, metalness = m1.w;
  if (RT_TEX_BATCH < 2 && hm) { tm = tex_taps(D[2], in.uvx, in.uvy); qm = tex_fetch(P.texels + D[2].offset, tm); }
  if (hm) {
    rt_v3 mr = tex_combine(qm, tm);
    roughness *= mr.y;
    metalness *= mr.z;
  }
  roughness = rt_clampf(roughness, 0.001f, 1.0f);
  if (metalness > 0.9f) metalness = 0.9f;
  metalness /= 0.9f;

  emission = rt_v3_make(m1.x, m1.y, m1.z);
  if (RT_TEX_BATCH < 2 && he) { te = tex_taps(D[3], in.uvx, in.uvy); qe = tex_fetch(P.texels + D[3].offset, te); }
  if (he) emission = rt_v3_mul(emission, srgb_to_linear_tex<Pow24InLds<PT>::value>(tex_combine(qe, te)));

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

  out_dir = rt_v3_comb3(t, o.x, b, o.y, normal, o.z);
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
  float uvx = rt_madd(((float)x + jitter - 0.5f) * 2.0f, inv_width, -1.0f);
  float uvy = rt_madd(((float)y + jitter - 0.5f) * 2.0f, inv_height, -1.0f);
  float dx = uvx * aspect, dy = -uvy, dz = -P.focal_length;
  // (rcp_exact: a square root lies in [0, 2^64] or is infinite / NaN -- inside the domain on which it equals the division)
  float inv_length = rcp_exact(rt_sqrtf(rt_dot3(dx, dx, dy, dy, dz, dz)));
  float rx = rt_dot3(P.cam[0][0], dx, P.cam[0][1], dy, P.cam[0][2], dz);
  float ry = rt_dot3(P.cam[1][0], dx, P.cam[1][1], dy, P.cam[1][2], dz);
  float rz = rt_dot3(P.cam[2][0], dx, P.cam[2][1], dy, P.cam[2][2], dz);
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
  r.fast = rt_slab_fast(o.x, o.y, o.z, r.inv_x, r.inv_y, r.inv_z, rt_slab_bias(o.x, r.inv_x), rt_slab_bias(o.y, r.inv_y), rt_slab_bias(o.z, r.inv_z));
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
  rt_v3 point = rt_v3_madd(dir, hit.t, org);
  rt_v3 n_geo = rt_v3_make(q0.x, q0.y, q0.z);
  rt_v3 n_int = rt_v3_make(rt_dot3(q1.x, t0, q2.x, t1, q3.x, t2),
                           rt_dot3(q1.y, t0, q2.y, t1, q3.y, t2),
                           rt_dot3(q1.z, t0, q2.z, t1, q3.z, t2));
  if (rt_v3_dot(n_geo, dir) > 0.0f || rt_v3_dot(n_int, dir) > 0.0f) {
    // back face: pass through, costs a bounce (raytracer.c:516-522)
    org = rt_v3_madd(dir, RT_EPS, point);
  } else {
    ShadeIn in;
    in.direction = dir;
    in.normal = normalize_dev(n_int);
    in.tangent = rt_v3_make(q4.x, q4.y, q4.z);
    in.bitangent = rt_v3_make(q5.x, q5.y, q5.z);
    in.uvx = rt_dot3(q1.w, t0, q3.w, t1, q5.w, t2);
    in.uvy = rt_dot3(q2.w, t0, q4.w, t1, q6.x, t2);
    rt_v3 out_dir, s_tint, s_emis;
    bool terminate;
    cn.shades += 1;
    shade(P, as_i(q0.w), in, rng, out_dir, s_tint, s_emis, terminate, cn);
    emis = rt_v3_mul_add(s_emis, tint, emis);
    if (terminate) {
      done = true;
      radiance = emis;
    } else {
      dir = out_dir;
      tint = rt_v3_mul(tint, s_tint);
      float below = (rt_v3_dot(n_geo, out_dir) < 0.0f) ? 1.0f : 0.0f;
      float bias = (0.5f - below) * 2.0f * RT_EPS;
      org = rt_v3_madd(n_geo, bias, point);
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

#define PH_NEED 0     // no path: wants a new (pixel, sample)
#define PH_POP  1     // traversal: take the next child of the current node (transient)
#define PH_NODE 2     // traversal: wants node_enter(child)
#define PH_LEAF 3     // traversal: wants leaf_test(child)
#define PH_HIT  4     // traversal finished with a hit: wants shading
#define PH_MISS 5     // traversal finished without a hit: wants the environment

#define RT_PARK_FIELDS 18     // hit (t, triangle, u, v), ray origin and direction, tint, emission, rng, pixel of the tile | bounce << 6
#define RT_PARK_CAP 128       // parked hits per wave (fewer than RT_PARK_DENSE + 64 are ever parked)
static_assert(RT_PARK_FIELDS * RT_PARK_CAP == RT_PARK_RECORD_DWORDS, "park slice size (rt_device.h) out of step");
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
#ifndef RT_LEAF_TRIP
#define RT_LEAF_TRIP 1          // triangles whose scalar loads are in flight together in leaf_test_uniform (2: nine more SGPRs, which spill into the sky loop)
#endif
#ifndef RT_LEAF_CULL_MAX
#define RT_LEAF_CULL_MAX 4     // surviving triangles up to which a camera-ray leaf block takes the culled, triangle-by-triangle form
#endif
#ifndef RT_PARK_STOP_PATHS
#define RT_PARK_STOP_PATHS 0      // hits are shaded at once instead of parked when the tile has at most this many paths left to hand out (0: off)
#endif

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
  // contract v2: a fused slab distance fma(p, inv, -(o * inv)) is off by up to an ulp of max(|p|, |o|) * |inv|, i.e. it
  // places the plane within ~2^-23 max(|p|, |o|) of where it is; the margin carries 8 times that, in units of n . (p - o)
  const float coarse = fabsf(nx) * (fabsf(ox) + fmaxf(fabsf(mnx), fabsf(mxx))) + fabsf(ny) * (fabsf(oy) + fmaxf(fabsf(mny), fabsf(mxy))) +
                       fabsf(nz) * (fabsf(oz) + fmaxf(fabsf(mnz), fabsf(mxz)));
  const bool outside = empty || nearest > 1e-3f * extent + 1e-6f * coarse;          // (NaN compares false: not outside)
  const uint32_t m = (uint32_t)__ballot(outside);                                  // lanes 0..31: 4 planes x 8 children
  return (m | (m >> 8) | (m >> 16) | (m >> 24)) & 0xFFu;
}

// slab_entry<true>() of child k of an LDS node with the planes picked by address (see NODE_LDS_ORDERED)
__device__ __forceinline__ float slab_entry_ordered(const Ray3 &r, const rt_v3 &bs, const char *nbase, int k, int nx, int ny, int nz, float t_max) {
  const char *b = nbase + k * 4;
  const float sx = rt_slab_t_fast(*reinterpret_cast<const float *>(b + nx), r.o.x, r.inv_x, bs.x);
  const float bx = rt_slab_t_fast(*reinterpret_cast<const float *>(b + (96 - nx)), r.o.x, r.inv_x, bs.x);
  const float sy = rt_slab_t_fast(*reinterpret_cast<const float *>(b + 32 + ny), r.o.y, r.inv_y, bs.y);
  const float by = rt_slab_t_fast(*reinterpret_cast<const float *>(b + 32 + (96 - ny)), r.o.y, r.inv_y, bs.y);
  const float sz = rt_slab_t_fast(*reinterpret_cast<const float *>(b + 64 + nz), r.o.z, r.inv_z, bs.z);
  const float bz = rt_slab_t_fast(*reinterpret_cast<const float *>(b + 64 + (96 - nz)), r.o.z, r.inv_z, bs.z);
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
  const rt_v3 bs = slab_bias3(r);
  const int n = (int)__popc(surv);
  const int k0 = (int)__builtin_ctz(surv);
  const int e0 = as_i(slab_entry_ordered(r, bs, nbase, k0, nx, ny, nz, hit_t));
  const uint32_t f0 = 1u - (((uint32_t)e0 + 0x00800000u) >> 31);                   // 1 iff e0 is finite (a candidate)
  LGM("nfew_ret1");
  if (n == 1) return (uint32_t)k0 | (f0 << 24);
  surv &= surv - 1u;
  const int k1 = (int)__builtin_ctz(surv);
  const int e1 = as_i(slab_entry_ordered(r, bs, nbase, k1, nx, ny, nz, hit_t));
  const uint32_t f1 = 1u - (((uint32_t)e1 + 0x00800000u) >> 31);
  LGM("nfew_two");
  if (n == 2) {
    const bool swap = e1 < e0;                                                     // ties: lowest index first
    const uint32_t first = swap ? (uint32_t)k1 : (uint32_t)k0, second = swap ? (uint32_t)k0 : (uint32_t)k1;
    return first | (second << 3) | ((f0 + f1) << 24);
  }
  LGM("nfew_ret2");
  surv &= surv - 1u;
  const int k2 = (int)__builtin_ctz(surv);
  const int e2 = as_i(slab_entry_ordered(r, bs, nbase, k2, nx, ny, nz, hit_t));
  const uint32_t f2 = 1u - (((uint32_t)e2 + 0x00800000u) >> 31);
  int e3 = 0x7F800000, k3 = 0;
  uint32_t f3 = 0;
  LGM("nfew_four_begin");
  if (n == 4) {
    surv &= surv - 1u;
    k3 = (int)__builtin_ctz(surv);
      e3 = as_i(slab_entry_ordered(r, bs, nbase, k3, nx, ny, nz, hit_t));
    f3 = 1u - (((uint32_t)e3 + 0x00800000u) >> 31);
  }
  LGM("nfew_four_end");
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

// ---- pyramid culling of leaf blocks (tile-stream kernel) ----
// The camera rays of a wave sit on one or two pixels: most leaf blocks that hold camera rays hold (almost) nothing else, all
// of them on ONE leaf group (config #3: 6.4 M of 16.8 M leaf blocks, 53 lanes each).  Such a block tests only the triangles the
// tile's pyramid can touch.  Lane l tests triangle (l & 7) of leaf group g against plane ((l >> 3) & 3); returns the 8-bit mask
// of the triangles that lie entirely outside one plane -- all three vertices by the relative margin of pyramid_cull_mask() -- :
// no point of a ray inside the pyramid lies in such a triangle, so the point where the ray meets the triangle's plane has a
// barycentric coordinate outside [0, 1] by far more than the test's epsilon and ray_triangles_hit_8 reports a miss for it
// (raytracer.c:84-188), whatever the ray's t_max.  (Vertices b, c are a + (b - a), a + (c - a) here, an ulp off the reference's:
// the `coarse` term of the margin covers a thousand of those.)
__device__ __forceinline__ uint32_t pyramid_cull_tris(const float *leaves, const float *pyr, int g) {
  const int lane = lane_now();
  const float *tb = leaves + (size_t)g * 72 + (lane & 7);
  const float *pl = pyr + ((lane >> 3) & 3) * 4;
  const float ox = pyr[16], oy = pyr[17], oz = pyr[18];
  const float nx = pl[0], ny = pl[1], nz = pl[2];
  const float ax = tb[0], e1x = tb[8], e2x = tb[16], ay = tb[24], e1y = tb[32], e2y = tb[40], az = tb[48], e1z = tb[56], e2z = tb[64];
  bool outside = true;
#pragma unroll
  for (int v = 0; v < 3; v++) {
    const float vx = v == 0 ? ax : (v == 1 ? ax + e1x : ax + e2x);
    const float vy = v == 0 ? ay : (v == 1 ? ay + e1y : ay + e2y);
    const float vz = v == 0 ? az : (v == 1 ? az + e1z : az + e2z);
    const float px = nx * (vx - ox), py = ny * (vy - oy), pz = nz * (vz - oz);
    const float val = px + py + pz;                                                   // n . (vertex - o)
    const float extent = fabsf(px) + fabsf(py) + fabsf(pz);
    const float coarse = fabsf(nx) * (fabsf(ox) + fabsf(vx)) + fabsf(ny) * (fabsf(oy) + fabsf(vy)) + fabsf(nz) * (fabsf(oz) + fabsf(vz));
    outside = outside && (val > 1e-3f * extent + 1e-6f * coarse);                     // (NaN compares false: not outside)
  }
  const uint32_t m = (uint32_t)__ballot(outside);                                     // lanes 0..31: 4 planes x 8 triangles
  return (m | (m >> 8) | (m >> 16) | (m >> 24)) & 0xFFu;
}

// leaf_test_short_div() / leaf_test<false>() of leaf group g (WAVE-UNIFORM) restricted to the triangles in `surv` (wave-uniform):
// the triangle's nine floats come through the scalar cache, the arithmetic per triangle is tri_test()'s expression for
// expression, triangles in ascending order with a strict comparison (lowest index wins ties, raytracer.c:27-29).
template <bool SHORT_DIV>
__device__ __forceinline__ void tri_test_uniform(const Ray3 &r, rt_v3 a, rt_v3 edge1, rt_v3 edge2, int k, float &best, float &bu, float &bv, int &bi) {
  rt_v3 rxe2 = rt_v3_cross(r.d, edge2);
  float det = rt_v3_dot(edge1, rxe2);
  float inv_det = SHORT_DIV ? rcp_leaf(det) : 1.0f / det;
  rt_v3 s = rt_v3_sub(r.o, a);
  rt_v3 sxe1 = rt_v3_cross(s, edge1);
  float u = inv_det * rt_v3_dot(s, rxe2);
  float v = inv_det * rt_v3_dot(r.d, sxe1);
  float t = inv_det * rt_v3_dot(edge2, sxe1);
  bool miss = (u < -RT_EPS) || (u > 1.0f + RT_EPS) || (v < -RT_EPS) || (u + v > 1.0f + RT_EPS) || (t < RT_EPS);
  if (SHORT_DIV) {
    if (!miss && t < best) { best = t; bi = k; bu = u; bv = v; }          // (see leaf_test_short_div)
  } else {
    float dist = miss ? RT_INF : t;
    dist = (dist > 0.0f) ? dist : RT_INF;                                  // NaN -> +inf (min_f32x8)
    if (dist < best) { best = dist; bi = k; bu = u; bv = v; }
  }
}
template <bool SHORT_DIV>
__device__ __forceinline__ bool leaf_test_uniform(const RT_KParams &P, const Ray3 &r, int g, uint32_t surv, HitRec &hit) {
  float best = RT_INF, bu = 0.0f, bv = 0.0f;
  int   bi = 0;
  cfloat *lb = as_scalar_ptr(P.leaves) + (size_t)g * 72;
  // RT_LEAF_TRIP 2: two triangles per trip, their eighteen scalar loads in flight together (a tile's pyramid leaves two of a group's
  // eight triangles on average) -- equal on the helmet, but the nine extra SGPRs cost the sky loop four spill moves per batch (tower
  // +0.3 %): one per trip is what ships
  while (surv) {
    const int k0 = (int)__builtin_ctz(surv);
    surv &= surv - 1u;
#if RT_LEAF_TRIP == 1
    {
      cfloat *t0 = lb + k0;
      tri_test_uniform<SHORT_DIV>(r, rt_v3_make(t0[0], t0[24], t0[48]), rt_v3_make(t0[8], t0[32], t0[56]), rt_v3_make(t0[16], t0[40], t0[64]),
                                  k0, best, bu, bv, bi);
      continue;
    }
#endif
    const bool two = surv != 0u;
    const int k1 = two ? (int)__builtin_ctz(surv) : k0;
    surv &= surv - 1u;                                                     // (0 & anything = 0)
    cfloat *t0 = lb + k0, *t1 = lb + k1;
    const rt_v3 a0 = rt_v3_make(t0[0], t0[24], t0[48]), e10 = rt_v3_make(t0[8], t0[32], t0[56]), e20 = rt_v3_make(t0[16], t0[40], t0[64]);
    const rt_v3 a1 = rt_v3_make(t1[0], t1[24], t1[48]), e11 = rt_v3_make(t1[8], t1[32], t1[56]), e21 = rt_v3_make(t1[16], t1[40], t1[64]);
    tri_test_uniform<SHORT_DIV>(r, a0, e10, e20, k0, best, bu, bv, bi);
    if (two) tri_test_uniform<SHORT_DIV>(r, a1, e11, e21, k1, best, bu, bv, bi);
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

typedef const RT_KParams __attribute__((address_space(4))) *RT_KArgs;
__device__ __forceinline__ RT_KArgs cold_args() {
  RT_KArgs p = (RT_KArgs)__builtin_amdgcn_kernarg_segment_ptr();      // the RT_KParams block is the kernel's only argument
  asm volatile("" : "+s"(p));
  return p;
}

// ---- the traversal blocks of the path kernels -------------------------------------------------------------------------
// ray_bvh_node_hit (raytracer.c:443-483) for the 64 rays of a wave, phase-scheduled: per round the wave runs ONE block --
// LEAF (8-triangle test, raytracer.c:84-188) or NODE (8-box slab test + near-first order, raytracer.c:190-230, :459-468),
// whichever more lanes wait for -- then every lane that finished a block pops its next child (raytracer.c:459-482) until
// it knows its next block.  Returns when no lane traverses any more or when `exit_lanes` of the `n_trav0` lanes that
// traversed at entry have finished (phase PH_HIT / PH_MISS).  This ONE function is the traversal of the tile-stream path
// kernel (rt_kernels.hip), of the camera and trace kernels of the wavefront pipeline (rt_wavefront.hip) and of
// rt_test_trace_stream_kernel (unit-level parity against oracle_trace_rays).
//   PYRAMID: node blocks whose lanes are camera rays (`is_cam`) of the wave's tile about to enter ONE node test only the
//            child boxes the tile's pyramid (LDS, pyr_off) can touch (pyramid_cull_mask, node_enter_few).
struct TravState {
  int      phase, level, node, child;
  uint32_t cur, dirty, live;
  HitRec   hit;
};

template <bool LDSN, bool SHORT_DIV, bool PYRAMID>
__device__ __forceinline__ void traversal_blocks(const RT_KParams &P, float4 *smem, const float4 *lds_nodes, uint32_t *perm,
                                                 const int lane, const int n_lds, const int pyr_nodes, const int pyr_off,
                                                 const int leaf_level, const int exit_lanes, const int n_trav0, const Ray3 &ray,
                                                 const bool is_cam, int &phase, int &level, int &node, int &child, uint32_t &cur,
                                                 uint32_t &dirty, uint32_t &live, HitRec &hit, uint32_t &w_nodes, uint32_t &w_leaves,
                                                 uint32_t *lg = nullptr) {
  LG(LG_TRAV_CALLS, 1);
  for (;;) {
    LGM("round_begin");
    const unsigned long long maskN = __ballot(phase == PH_NODE);
    const int nN = (int)__popcll(maskN);
    const int nL = (int)__popcll(__ballot(phase == PH_LEAF));
    if (nN + nL == 0 || n_trav0 - (nN + nL) >= exit_lanes) break;
    LG(LG_ROUND_X, 1);

    if (nL >= nN) {
      // ----- LEAF -----
      LGT0();
      LGM("leaf_begin");
      LG(LG_LEAF_X, 1); LG(LG_LEAF_L, nL);
#ifdef RT_LEDGER
      LG(LG_LEAF_CAM, __popcll(__ballot(phase == PH_LEAF && is_cam)));
#endif
      w_leaves += (uint32_t)nL;
#if !RT_LEAF_PAIRS && !defined(RT_NO_LEAF_CULL)
      // camera rays of this tile about to test the same leaf group, (almost) alone in the block: only the triangles their
      // pyramid can touch; the other lanes keep waiting for a leaf block
      bool leaf_done = false;
      if (PYRAMID && pyr_nodes > 0) {
        const unsigned long long maskL = __ballot(phase == PH_LEAF);
        const unsigned long long camL = maskL & __ballot(is_cam);
        if (camL != 0ull) {
          const int c0 = __builtin_amdgcn_readlane(child, (int)__builtin_ctzll(camL));
          const int nG = (int)__popcll(camL & __ballot(child == c0));
          const int g0 = c0 - P.last_row_offset;
          if (nG * RT_PYR_DEN >= nL * RT_PYR_NUM && nG >= RT_PYR_MIN && (uint32_t)g0 < 0x800000u) {
            float *pyr = lds_at(smem, pyr_off);
            uint32_t *slot = reinterpret_cast<uint32_t *>(pyr) + 24 + (g0 & 7);      // 8 entries of their own, in front of the node masks'
            const uint32_t key = 0x800000u | (uint32_t)g0;
            const uint32_t ce = (uint32_t)__builtin_amdgcn_readfirstlane((int)*slot);
            uint32_t cull;
            if ((ce >> 8) == key) {
              cull = ce & 0xFFu;
            } else {
              cull = pyramid_cull_tris(P.leaves, pyr, g0);
              if (lane_now() == 0) *slot = (key << 8) | cull;
            }
            // (a group of which the pyramid leaves more than RT_LEAF_CULL_MAX triangles is cheaper in the full block, which has all
            //  eight in flight at once: tower at 1080p, whose triangles are many tiles wide, +0.8 % without this)
            if (__popc(0xFFu & ~cull) <= RT_LEAF_CULL_MAX) {
              w_leaves -= (uint32_t)(nL - nG);
              if (phase == PH_LEAF && is_cam && child == c0) {
                if (leaf_test_uniform<SHORT_DIV>(P, ray, g0, 0xFFu & ~cull, hit)) dirty = 0xFFFFFFFFu;
                phase = PH_POP;
              }
              leaf_done = true;
            }
          }
        }
      }
      if (!leaf_done)
#endif
#if RT_LEAF_PAIRS
      {
        const bool in_leaf = phase == PH_LEAF;
        const int  g = child - P.last_row_offset;
        if (leaf_test_pair<SHORT_DIV>(P, ray, g, in_leaf, hit)) dirty = 0xFFFFFFFFu;
        if (in_leaf) phase = PH_POP;
      }
#else
      if (phase == PH_LEAF) {
        int  g = child - P.last_row_offset;
        bool got = SHORT_DIV ? leaf_test_short_div(P, ray, g, hit) : leaf_test<false>(P, ray, g, hit);
        if (got) dirty = 0xFFFFFFFFu;
        phase = PH_POP;
      }
#endif
      LGM("leaf_end");
      LGT1(LG_CYC_LEAF);
    } else {
      // ----- NODE -----
      LGT0();
      LGM("node_begin");
      w_nodes += (uint32_t)nN;
      const bool all_fast = (maskN & __ballot(!ray.fast)) == 0ull;
      // camera rays of this tile about to enter the same node: test only the children their pyramid can touch.  When
      // they are most of the block's lanes, the block runs for them alone; the others keep waiting for a node block.
      uint32_t surv = 0xFFFFu;
      bool in_blk = phase == PH_NODE;
      if (PYRAMID) {
        // (ballots of single comparisons combined with scalar ANDs: a ballot of a compound condition costs two more
        // vector instructions)
        const unsigned long long camN = maskN & __ballot(is_cam);
        if (LDSN && all_fast && camN != 0ull) {
          const int c0 = __builtin_amdgcn_readlane(child, (int)__builtin_ctzll(camN));
          const int nG = (int)__popcll(camN & __ballot(child == c0));
          LG(LG_PYRCHK_X, 1);
          if (c0 < pyr_nodes && nG * RT_PYR_DEN >= nN * RT_PYR_NUM && nG >= RT_PYR_MIN) {
            // the mask depends on (tile, node) only and the tile's camera rays keep coming back to the same nodes
            float *pyr = lds_at(smem, pyr_off);
            uint32_t *slot = reinterpret_cast<uint32_t *>(pyr) + 32 + (c0 & 31);
            const uint32_t ce = (uint32_t)__builtin_amdgcn_readfirstlane((int)*slot);
            if ((ce >> 8) == (uint32_t)c0 + 1u) {
              surv = 0xFFu & ~ce;
            } else {
              LG(LG_CULLMASK_X, 1);
              LGM("cullmask_begin");
              const uint32_t cull = pyramid_cull_mask(lds_nodes, pyr, c0);
              if (lane_now() == 0) *slot = (((uint32_t)c0 + 1u) << 8) | cull;
              surv = 0xFFu & ~cull;
              LGM("cullmask_end");
            }
            if (__popc(surv) > 4) surv = 0xFFFFu;
            else {
              in_blk = phase == PH_NODE && is_cam && child == c0; w_nodes -= (uint32_t)(nN - nG);
#ifdef RT_LEDGER
              {
                const int ns = (int)__popc(surv);         // (constant slots: a dynamic index would put the ledger into scratch)
                LG(LG_NFEW0_X, ns == 0); LG(LG_NFEW1_X, ns == 1); LG(LG_NFEW2_X, ns == 2); LG(LG_NFEW3_X, ns == 3); LG(LG_NFEW4_X, ns == 4);
                LG(LG_NFEW0_L, ns == 0 ? nG : 0); LG(LG_NFEW1_L, ns == 1 ? nG : 0); LG(LG_NFEW2_L, ns == 2 ? nG : 0);
                LG(LG_NFEW3_L, ns == 3 ? nG : 0); LG(LG_NFEW4_L, ns == 4 ? nG : 0); LG(LG_NODE_WAIT_L, nN - nG);
              }
#endif
            }
          }
        }
      }
      if (in_blk) {
        if (level >= 0) {
          perm[level * 64 + lane] = cur;
          live = (cur >> 24) ? (live | (1u << level)) : (live & ~(1u << level));
        }
        node = child;
        level += 1;
        if (PYRAMID && surv <= 0xFFu) {
          LGM("nfew_begin");
          cur = surv ? node_enter_few(ray, lds_nodes, node, surv, hit.t) : 0u;
          LGM("nfew_end");
        } else if (all_fast) {
          if (LDSN && __ballot(node >= n_lds) == 0) {
            LG(LG_NFULL_X, 1); LG(LG_NFULL_L, nN);
#ifdef RT_LEDGER
            LG(LG_NFULL_CAM, __popcll(maskN & __ballot(is_cam)));
#endif
            LGM("nfull_begin");
            cur = node_enter<true, NODE_LDS_ORDERED>(P, ray, node, hit.t, lds_nodes);
            LGM("nfull_end");
          } else {
            LG(LG_NGLOB_X, 1); LG(LG_NGLOB_L, nN);
            LGM("nglob_begin");
            cur = node_enter<true, NODE_GLOBAL>(P, ray, node, hit.t, lds_nodes);
            LGM("nglob_end");
          }
        } else if (ray.fast) {
          LGM("nexact_begin");
          LG(LG_NEXACT_X, 1); LG(LG_NEXACT_L, nN);
          // a block that holds a ray that is not NaN-free (axis-aligned, 0 * inf): every lane takes the form ITS ray is
          // entitled to -- the two round differently under contract v2 -- in two passes over the block's lanes
          cur = node_enter<true, NODE_GLOBAL>(P, ray, node, hit.t, lds_nodes);
        } else {
          cur = node_enter<false, NODE_GLOBAL>(P, ray, node, hit.t, lds_nodes);
        }
        LGM("node_entered");
        dirty &= ~(1u << level);
        if (cur >> 24) {
          child = 8 * node + 1 + (int)(cur & 7u);
          cur = ((cur >> 3) & 0x1FFFFFu) | (((cur >> 24) - 1u) << 24);
          phase = (level == leaf_level) ? PH_LEAF : PH_NODE;
        } else {
          phase = PH_POP;
        }
      }
      LGM("node_end");
      LGT1(LG_CYC_NODE);
    }

    // ----- pops: every lane that just finished a block takes its next child / goes up -----
    LGT0();
    LGM("pop_begin");
    while (__any(phase == PH_POP)) {
#ifdef RT_LEDGER
      {
        const unsigned long long mp = __ballot(phase == PH_POP);
        LG(LG_POP_X, 1); LG(LG_POP_L, __popcll(mp)); LG(LG_POP_CAM, __popcll(mp & __ballot(is_cam)));
        const unsigned long long mu = mp & __ballot((cur >> 24) == 0 || level < 0);
        LG(LG_POP_UP_L, __popcll(mu)); LG(LG_POP_UP_X, mu != 0ull);
      }
      bool lg_retest = false;
#endif
      LGM("pop_iter_begin");
      if (phase == PH_POP) {
        uint32_t cnt = cur >> 24;
        LGM("pop_up_begin");
        if (cnt == 0 || level < 0) {
          // go up to the nearest level that still has children to visit -- in one step: the k-th ancestor of node n in the
          // implicit 8-ary tree is (n - (8^k - 1)/7) >> 3k, and (8^k - 1)/7 is k ones 3 bits apart
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
        LGM("pop_up_end");
        if (phase == PH_POP) {
          int j = (int)(cur & 7u);
          cur = ((cur >> 3) & 0x1FFFFFu) | ((cnt - 1u) << 24);
          bool go = true;
#ifdef RT_LEDGER
          lg_retest = ((dirty >> level) & 1u) != 0;
#endif
          LGM("pop_retest_begin");
          if ((dirty >> level) & 1u) {
            float dj;
            if (LDSN && node < n_lds) {
              // entry distance from the three NEAR planes, picked by address (see NODE_LDS_ORDERED); the rays that are
              // not NaN-free -- a lane in a blue moon -- redo it through the min / max form
              const char *nb = reinterpret_cast<const char *>(lds_nodes + lds_node_f4(node)) + j * 4;
              const rt_v3 bs = slab_bias3(ray);
              // the three planes in flight together (left to itself hipcc reuses one register for address and value and waits
              // for each read before it issues the next: three LDS round trips in the loop that has the fewest lanes to hide them)
              float px, py, pz;
              {
                const uint32_t ax = (uint32_t)(uintptr_t)(nb + ((as_i(ray.inv_x) >> 31) & 96));
                const uint32_t ay = (uint32_t)(uintptr_t)(nb + ((as_i(ray.inv_y) >> 31) & 96));
                const uint32_t az = (uint32_t)(uintptr_t)(nb + ((as_i(ray.inv_z) >> 31) & 96));
                asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4 offset:32\n\tds_read_b32 %2, %5 offset:64\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(px), "=&v"(py), "=&v"(pz) : "v"(ax), "v"(ay), "v"(az) : "memory");
              }
              const float sx = rt_slab_t_fast(px, ray.o.x, ray.inv_x, bs.x);
              const float sy = rt_slab_t_fast(py, ray.o.y, ray.inv_y, bs.y);
              const float sz = rt_slab_t_fast(pz, ray.o.z, ray.inv_z, bs.z);
              dj = fmax_hw(RT_EPS, fmax_hw(sx, fmax_hw(sy, sz)));
              LGM("pop_rare_begin");
              if (!ray.fast) dj = slab_entry_child<false>(reinterpret_cast<const float *>(lds_nodes + lds_node_f4(node)) + j, ray);
              LGM("pop_rare_end");
            } else {
              LGM("pop_glob_begin");
              dj = slab_entry_child_any(P.nodes + (size_t)node * 48 + j, ray);
              LGM("pop_glob_end");
            }
            if (!(dj < hit.t)) { cur = 0; go = false; }      // raytracer.c:470-472
          }
          LGM("pop_retest_end");
          if (go) {
            child = 8 * node + 1 + j;
            phase = (level == leaf_level) ? PH_LEAF : PH_NODE;
          }
        }
      }
      LGM("pop_iter_end");
#ifdef RT_LEDGER
      { const unsigned long long mr = __ballot(lg_retest); LG(LG_POP_RETEST_L, __popcll(mr)); LG(LG_POP_RETEST_X, mr != 0ull); }
#endif
    }
    LGM("pop_end");
    LGT1(LG_CYC_POP);
  }
}

struct ShadeParams {            // what shade_hit / background_lookup read (same field names as RT_KParams)
  const float *tris, *mats;
  const RT_DTexture *textures;
  const uint32_t *texels;
  int32_t bg_texture, max_bounces;
};

struct ShadeParamsLds : ShadeParams {};      // ... of a kernel that keeps the sRGB scale table in LDS (pow24_lds_init)
template <> struct Pow24InLds<ShadeParamsLds> { static constexpr bool value = RT_MATH_POW24 != 0; };

struct PrimaryParams {          // what primary_ray reads
  float cam[3][4];
  float focal_length, inv_width, inv_height, aspect;
};


#endif  // RT_DEV_HIP_H
