/* oracle.c -- CPU restatement of the reference render path.  TEST INFRASTRUCTURE,
 * see oracle.h for who may use it, the "parity unpinned" statement and the list
 * of deviations D1-D9.  Every function names the reference lines it follows.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, no fast-math: the results
 * must be bit-identical to the gfx950 kernels, which share include/rt_math.h).
 */
#include "oracle.h"
#include "../include/rt_math.h"

#include <immintrin.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* thread-local path state: the reference keeps the RNG state in a
 * thread_local (common.h:13); counters are this build's addition.            */

static _Thread_local u32             random_state;
static _Thread_local Oracle_Counters tl_counters;
static _Thread_local f32             tl_bary_u, tl_bary_v;   /* barycentrics of the last accepted hit */
/* ORACLE_LITERAL (oracle.h): the reference's literal semantics where the default oracle deviates (D1, D2, D6, D8) */
static _Thread_local bool            tl_literal;
#define PI_D 3.14159265358979323846      /* codin's PI is not in the reference tree; taken as a double constant */

static inline f32 rand_f32(void) { return rt_rand_f32(&random_state); }

static inline rt_v3 V(Vec3 v) { return rt_v3_make(v.x, v.y, v.z); }
static inline Vec3  U(rt_v3 v) { Vec3 r; r.x = v.x; r.y = v.y; r.z = v.z; return r; }

/* ------------------------------------------------------------------------- */
/* raytracer.c:15-32  min_f32x8: lanes that are not > epsilon (incl. NaN)
 * become +inf; horizontal min; index of the LOWEST lane equal to the min.     */
static f32 min_f32x8(f32 const vec[8], f32 epsilon, i32 *index) {
  f32 sanitized[8];
  for (int k = 0; k < 8; k++) sanitized[k] = (vec[k] > epsilon) ? vec[k] : RT_INF;
  f32 m = sanitized[0];
  for (int k = 1; k < 8; k++) m = rt_min_ps(m, sanitized[k]);
  *index = 0;
  for (int k = 0; k < 8; k++) {
    if (sanitized[k] == m) { *index = k; break; }
  }
  return m;
}

/* raytracer.c:84-188  Moeller-Trumbore against the 8 triangles of one leaf
 * group; no determinant test; epsilon-padded barycentric bounds; t >= eps.    */
static bool ray_triangles_hit_8_fill(Ray const *ray, Triangles const *triangles, isize offset, Hit *hit, i32 *lane_out,
                                     f32 const us[8], f32 const vs[8], f32 const distances[8]);
#if defined(__AVX2__) && (defined(__FMA__) || defined(RT_MATH_NO_FMA))
#define ORACLE_HAVE_AVX2 1
static int g_oracle_simd = 1;
static f32 min_f32x8_avx2(f32 const vec[8], f32 epsilon, i32 *index);
#else
#define ORACLE_HAVE_AVX2 0
static int g_oracle_simd = 0;
#endif

static void ray_triangles_8_scalar(Ray const *ray, Triangles const *triangles, isize offset, f32 us[8], f32 vs[8], f32 distances[8]) {
  rt_v3 dir = V(ray->direction), org = V(ray->position);

  for (int k = 0; k < 8; k++) {
    isize i = offset + k;
    rt_v3 a = rt_v3_make(triangles->x[0][i], triangles->y[0][i], triangles->z[0][i]);
    rt_v3 b = rt_v3_make(triangles->x[1][i], triangles->y[1][i], triangles->z[1][i]);
    rt_v3 c = rt_v3_make(triangles->x[2][i], triangles->y[2][i], triangles->z[2][i]);

    rt_v3 edge1 = rt_v3_sub(b, a);
    rt_v3 edge2 = rt_v3_sub(c, a);

    rt_v3 ray_cross_e2 = rt_v3_cross(dir, edge2);
    f32   det          = rt_v3_dot(edge1, ray_cross_e2);
    f32   inv_det      = 1.0f / det;

    rt_v3 s          = rt_v3_sub(org, a);
    rt_v3 s_cross_e1 = rt_v3_cross(s, edge1);

    f32 u = inv_det * rt_v3_dot(s, ray_cross_e2);
    f32 v = inv_det * rt_v3_dot(dir, s_cross_e1);
    f32 t = inv_det * rt_v3_dot(edge2, s_cross_e1);

    bool miss = (u < -RT_EPS) || (u > 1.0f + RT_EPS) ||
                (v < -RT_EPS) || (u + v > 1.0f + RT_EPS) ||
                (t < RT_EPS);
    us[k] = u;
    vs[k] = v;
    distances[k] = miss ? RT_INF : t;
  }
}

/* raytracer.c:157-187: horizontal minimum, closer-hit test and the fill of `Hit` (scalar in the reference as well) */
static bool ray_triangles_hit_8_fill(Ray const *ray, Triangles const *triangles, isize offset, Hit *hit, i32 *lane_out,
                                     f32 const us[8], f32 const vs[8], f32 const distances[8]) {
  rt_v3 dir = V(ray->direction), org = V(ray->position);
  i32 triangle_index;
  f32 min;
#if ORACLE_HAVE_AVX2
  if (g_oracle_simd) min = min_f32x8_avx2(distances, 0.0f, &triangle_index);
  else
#endif
    min = min_f32x8(distances, 0.0f, &triangle_index);

  if (min < hit->distance) {
    hit->distance = min;
    isize t = triangle_index + offset;
    Triangle_AOS const *aos = &triangles->aos[t];

    f32 t1 = us[triangle_index];
    f32 t2 = vs[triangle_index];
    f32 t0 = 1.0f - t1 - t2;
    tl_bary_u = t1;
    tl_bary_v = t2;

    hit->point = U(rt_v3_madd(dir, min, org));
    hit->normal.x = rt_dot3(aos->normal_a.x, t0, aos->normal_b.x, t1, aos->normal_c.x, t2);
    hit->normal.y = rt_dot3(aos->normal_a.y, t0, aos->normal_b.y, t1, aos->normal_c.y, t2);
    hit->normal.z = rt_dot3(aos->normal_a.z, t0, aos->normal_b.z, t1, aos->normal_c.z, t2);
    hit->tex_coords.x = rt_dot3(aos->tex_coords_a.x, t0, aos->tex_coords_b.x, t1, aos->tex_coords_c.x, t2);
    hit->tex_coords.y = rt_dot3(aos->tex_coords_a.y, t0, aos->tex_coords_b.y, t1, aos->tex_coords_c.y, t2);
    hit->shader     = aos->shader;
    hit->normal_geo = aos->normal;
    hit->tangent    = aos->tangent;
    hit->bitangent  = aos->bitangent;
    if (lane_out) *lane_out = (i32)t;
    return true;
  }
  return false;
}

/* ------------------------------------------------------------------------- */
/* The reference's 8-wide AVX2 forms (raytracer.c:15-32 min_f32x8, :84-188 ray_triangles_hit_8, :190-230
 * ray_aabbs_hit_8): __m256 arithmetic over the 8 triangles of a leaf group / the 8 child boxes of a node, same operand
 * order (the NaN rule of _mm256_min_ps / _mm256_max_ps is what rt_min_ps / rt_max_ps emulate in the scalar form), same
 * numeric contract (rt_math.h: one _mm256_fmadd_ps where the scalar form has one rt_madd).  Bit-identical to the scalar
 * form by construction and by test (tests/test_oracle_simd.py); the default when the compiler targets AVX2 (+ FMA under
 * contract v2) -- it is the reference's own SIMD path that bench.py times as `cpu_baseline` (kind "port-avx2"). */
#if ORACLE_HAVE_AVX2
static inline __m256 mm_madd(__m256 a, __m256 b, __m256 c) {
#ifdef RT_MATH_NO_FMA
  return _mm256_add_ps(_mm256_mul_ps(a, b), c);
#else
  return _mm256_fmadd_ps(a, b, c);
#endif
}
static inline __m256 mm_neg(__m256 a) { return _mm256_xor_ps(a, _mm256_set1_ps(-0.0f)); }
static inline __m256 mm_dot3(__m256 a0, __m256 b0, __m256 a1, __m256 b1, __m256 a2, __m256 b2) {      /* rt_dot3 */
  return mm_madd(a2, b2, mm_madd(a1, b1, _mm256_mul_ps(a0, b0)));
}
static inline __m256 mm_diff2(__m256 a, __m256 b, __m256 c, __m256 d) {                                /* rt_diff2 */
  return mm_madd(a, b, mm_neg(_mm256_mul_ps(c, d)));
}

/* raytracer.c:84-156 */
static void ray_triangles_8_avx2(Ray const *ray, Triangles const *triangles, isize offset, f32 us[8], f32 vs[8], f32 distances[8]) {
  __m256 dx = _mm256_set1_ps(ray->direction.x), dy = _mm256_set1_ps(ray->direction.y), dz = _mm256_set1_ps(ray->direction.z);
  __m256 ox = _mm256_set1_ps(ray->position.x), oy = _mm256_set1_ps(ray->position.y), oz = _mm256_set1_ps(ray->position.z);
  __m256 ax = _mm256_loadu_ps(triangles->x[0] + offset), ay = _mm256_loadu_ps(triangles->y[0] + offset), az = _mm256_loadu_ps(triangles->z[0] + offset);
  __m256 bx = _mm256_loadu_ps(triangles->x[1] + offset), by = _mm256_loadu_ps(triangles->y[1] + offset), bz = _mm256_loadu_ps(triangles->z[1] + offset);
  __m256 cx = _mm256_loadu_ps(triangles->x[2] + offset), cy = _mm256_loadu_ps(triangles->y[2] + offset), cz = _mm256_loadu_ps(triangles->z[2] + offset);
  __m256 e1x = _mm256_sub_ps(bx, ax), e1y = _mm256_sub_ps(by, ay), e1z = _mm256_sub_ps(bz, az);
  __m256 e2x = _mm256_sub_ps(cx, ax), e2y = _mm256_sub_ps(cy, ay), e2z = _mm256_sub_ps(cz, az);
  /* ray_cross_e2 = cross(dir, edge2) */
  __m256 rx = mm_diff2(dy, e2z, dz, e2y), ry = mm_diff2(dz, e2x, dx, e2z), rz = mm_diff2(dx, e2y, dy, e2x);
  __m256 det = mm_dot3(e1x, rx, e1y, ry, e1z, rz);
  __m256 inv_det = _mm256_div_ps(_mm256_set1_ps(1.0f), det);
  __m256 sx = _mm256_sub_ps(ox, ax), sy = _mm256_sub_ps(oy, ay), sz = _mm256_sub_ps(oz, az);
  /* s_cross_e1 = cross(s, edge1) */
  __m256 qx = mm_diff2(sy, e1z, sz, e1y), qy = mm_diff2(sz, e1x, sx, e1z), qz = mm_diff2(sx, e1y, sy, e1x);
  __m256 u = _mm256_mul_ps(inv_det, mm_dot3(sx, rx, sy, ry, sz, rz));
  __m256 v = _mm256_mul_ps(inv_det, mm_dot3(dx, qx, dy, qy, dz, qz));
  __m256 t = _mm256_mul_ps(inv_det, mm_dot3(e2x, qx, e2y, qy, e2z, qz));
  __m256 eps = _mm256_set1_ps(RT_EPS), neg_eps = _mm256_set1_ps(-RT_EPS), one_eps = _mm256_set1_ps(1.0f + RT_EPS);
  __m256 miss = _mm256_cmp_ps(u, neg_eps, _CMP_LT_OQ);
  miss = _mm256_or_ps(miss, _mm256_cmp_ps(u, one_eps, _CMP_GT_OQ));
  miss = _mm256_or_ps(miss, _mm256_cmp_ps(v, neg_eps, _CMP_LT_OQ));
  miss = _mm256_or_ps(miss, _mm256_cmp_ps(_mm256_add_ps(u, v), one_eps, _CMP_GT_OQ));
  miss = _mm256_or_ps(miss, _mm256_cmp_ps(t, eps, _CMP_LT_OQ));
  _mm256_storeu_ps(us, u);
  _mm256_storeu_ps(vs, v);
  _mm256_storeu_ps(distances, _mm256_blendv_ps(t, _mm256_set1_ps(RT_INF), miss));
}

/* raytracer.c:190-230 */
static void ray_aabbs_hit_8_avx2(Ray const *ray, f32 t_min, f32 t_max, BVH_Node const *node, f32 *distances) {
  f32 inv_x = 1.0f / ray->direction.x, inv_y = 1.0f / ray->direction.y, inv_z = 1.0f / ray->direction.z;
  f32 ox = ray->position.x, oy = ray->position.y, oz = ray->position.z;
  f32 bias_x = rt_slab_bias(ox, inv_x), bias_y = rt_slab_bias(oy, inv_y), bias_z = rt_slab_bias(oz, inv_z);
  bool fast = rt_slab_fast(ox, oy, oz, inv_x, inv_y, inv_z, bias_x, bias_y, bias_z);
  __m256 ix = _mm256_set1_ps(inv_x), iy = _mm256_set1_ps(inv_y), iz = _mm256_set1_ps(inv_z);
  __m256 mnx = _mm256_loadu_ps(node->min_x), mny = _mm256_loadu_ps(node->min_y), mnz = _mm256_loadu_ps(node->min_z);
  __m256 mxx = _mm256_loadu_ps(node->max_x), mxy = _mm256_loadu_ps(node->max_y), mxz = _mm256_loadu_ps(node->max_z);
  __m256 t0x, t0y, t0z, t1x, t1y, t1z;
#ifndef RT_MATH_NO_FMA
  if (fast) {                                   /* rt_slab_t_fast: fma(plane, inv, -(o * inv)) */
    __m256 bx = _mm256_set1_ps(bias_x), by = _mm256_set1_ps(bias_y), bz = _mm256_set1_ps(bias_z);
    t0x = _mm256_fmadd_ps(mnx, ix, bx); t0y = _mm256_fmadd_ps(mny, iy, by); t0z = _mm256_fmadd_ps(mnz, iz, bz);
    t1x = _mm256_fmadd_ps(mxx, ix, bx); t1y = _mm256_fmadd_ps(mxy, iy, by); t1z = _mm256_fmadd_ps(mxz, iz, bz);
  } else
#endif
  {                                             /* rt_slab_t_exact: (plane - o) * inv, raytracer.c:203-208 */
    (void)fast;
    __m256 vx = _mm256_set1_ps(ox), vy = _mm256_set1_ps(oy), vz = _mm256_set1_ps(oz);
    t0x = _mm256_mul_ps(_mm256_sub_ps(mnx, vx), ix); t0y = _mm256_mul_ps(_mm256_sub_ps(mny, vy), iy); t0z = _mm256_mul_ps(_mm256_sub_ps(mnz, vz), iz);
    t1x = _mm256_mul_ps(_mm256_sub_ps(mxx, vx), ix); t1y = _mm256_mul_ps(_mm256_sub_ps(mxy, vy), iy); t1z = _mm256_mul_ps(_mm256_sub_ps(mxz, vz), iz);
  }
  __m256 sx = _mm256_min_ps(t0x, t1x), sy = _mm256_min_ps(t0y, t1y), sz = _mm256_min_ps(t0z, t1z);
  __m256 bx2 = _mm256_max_ps(t0x, t1x), by2 = _mm256_max_ps(t0y, t1y), bz2 = _mm256_max_ps(t0z, t1z);
  __m256 t_minv = _mm256_max_ps(_mm256_set1_ps(t_min), _mm256_max_ps(sx, _mm256_max_ps(sy, sz)));
  __m256 t_maxv = _mm256_min_ps(_mm256_set1_ps(t_max), _mm256_min_ps(bx2, _mm256_min_ps(by2, bz2)));
  __m256 miss = _mm256_cmp_ps(t_minv, t_maxv, _CMP_GE_OQ);
  _mm256_storeu_ps(distances, _mm256_blendv_ps(t_minv, _mm256_set1_ps(RT_INF), miss));
}
#endif

#if ORACLE_HAVE_AVX2
/* raytracer.c:15-32: lanes that are not > epsilon (NaN included) become +inf; horizontal minimum by three
 * shuffle + min steps; index of the LOWEST lane equal to the minimum */
static f32 min_f32x8_avx2(f32 const vec[8], f32 epsilon, i32 *index) {
  __m256 v = _mm256_loadu_ps(vec);
  __m256 inf = _mm256_set1_ps(RT_INF);
  __m256 s = _mm256_blendv_ps(inf, v, _mm256_cmp_ps(v, _mm256_set1_ps(epsilon), _CMP_GT_OQ));
  __m256 m = _mm256_min_ps(s, _mm256_permute2f128_ps(s, s, 1));
  m = _mm256_min_ps(m, _mm256_shuffle_ps(m, m, _MM_SHUFFLE(1, 0, 3, 2)));
  m = _mm256_min_ps(m, _mm256_shuffle_ps(m, m, _MM_SHUFFLE(2, 3, 0, 1)));
  int mask = _mm256_movemask_ps(_mm256_cmp_ps(s, m, _CMP_EQ_OQ));
  *index = mask ? __builtin_ctz((unsigned)mask) : 0;
  return _mm256_cvtss_f32(m);
}
#endif

int oracle_have_avx2(void) { return ORACLE_HAVE_AVX2; }
/* 1 = the 8-wide AVX2 forms (default when compiled for AVX2), 0 = the scalar restatement; returns the mode now in force */
int oracle_set_simd(int on) {
  g_oracle_simd = (on && ORACLE_HAVE_AVX2) ? 1 : 0;
  return g_oracle_simd;
}

static bool ray_triangles_hit_8(Ray const *ray, Triangles const *triangles, isize offset, Hit *hit, i32 *lane_out) {
  f32 us[8], vs[8], distances[8];
#if ORACLE_HAVE_AVX2
  if (g_oracle_simd) ray_triangles_8_avx2(ray, triangles, offset, us, vs, distances);
  else
#endif
    ray_triangles_8_scalar(ray, triangles, offset, us, vs, distances);
  return ray_triangles_hit_8_fill(ray, triangles, offset, hit, lane_out, us, vs, distances);
}

static void ray_aabbs_hit_8_scalar(Ray const *ray, f32 t_min, f32 t_max, BVH_Node const *node, f32 *distances);
static void ray_aabbs_hit_8(Ray const *ray, f32 t_min, f32 t_max, BVH_Node const *node, f32 *distances) {
#if ORACLE_HAVE_AVX2
  if (g_oracle_simd) { ray_aabbs_hit_8_avx2(ray, t_min, t_max, node, distances); return; }
#endif
  ray_aabbs_hit_8_scalar(ray, t_min, t_max, node, distances);
}

/* raytracer.c:190-230  slab test of one ray against the 8 child boxes of a node (scalar form) */
static void ray_aabbs_hit_8_scalar(Ray const *ray, f32 t_min, f32 t_max, BVH_Node const *node, f32 *distances) {
  f32 inv_x = 1.0f / ray->direction.x;
  f32 inv_y = 1.0f / ray->direction.y;
  f32 inv_z = 1.0f / ray->direction.z;
  f32 ox = ray->position.x, oy = ray->position.y, oz = ray->position.z;
  /* numeric contract v2 (rt_math.h): NaN-free rays take every plane distance from ONE fused multiply-add,
   * fma(plane, inv, -(o * inv)); the others keep the reference's (plane - o) * inv (deviation D9) */
  f32 bias_x = rt_slab_bias(ox, inv_x), bias_y = rt_slab_bias(oy, inv_y), bias_z = rt_slab_bias(oz, inv_z);
  bool fast = rt_slab_fast(ox, oy, oz, inv_x, inv_y, inv_z, bias_x, bias_y, bias_z);

  for (int k = 0; k < 8; k++) {
    f32 t0x, t0y, t0z, t1x, t1y, t1z;
    if (fast) {
      t0x = rt_slab_t_fast(node->min_x[k], ox, inv_x, bias_x);
      t0y = rt_slab_t_fast(node->min_y[k], oy, inv_y, bias_y);
      t0z = rt_slab_t_fast(node->min_z[k], oz, inv_z, bias_z);
      t1x = rt_slab_t_fast(node->max_x[k], ox, inv_x, bias_x);
      t1y = rt_slab_t_fast(node->max_y[k], oy, inv_y, bias_y);
      t1z = rt_slab_t_fast(node->max_z[k], oz, inv_z, bias_z);
    } else {
      t0x = rt_slab_t_exact(node->min_x[k], ox, inv_x);
      t0y = rt_slab_t_exact(node->min_y[k], oy, inv_y);
      t0z = rt_slab_t_exact(node->min_z[k], oz, inv_z);
      t1x = rt_slab_t_exact(node->max_x[k], ox, inv_x);
      t1y = rt_slab_t_exact(node->max_y[k], oy, inv_y);
      t1z = rt_slab_t_exact(node->max_z[k], oz, inv_z);
    }

    f32 sx = rt_min_ps(t0x, t1x), sy = rt_min_ps(t0y, t1y), sz = rt_min_ps(t0z, t1z);
    f32 bx = rt_max_ps(t0x, t1x), by = rt_max_ps(t0y, t1y), bz = rt_max_ps(t0z, t1z);

    f32 t_minv = rt_max_ps(t_min, rt_max_ps(sx, rt_max_ps(sy, sz)));
    f32 t_maxv = rt_min_ps(t_max, rt_min_ps(bx, rt_min_ps(by, bz)));

    distances[k] = (t_minv >= t_maxv) ? RT_INF : t_minv;
  }
}

/* raytracer.c:443-483  recursive near-first traversal of the implicit 8-ary tree */
static void ray_bvh_node_hit(Ray const *ray, Scene const *scene, BVH_Index bvh_index, Hit *hit, isize depth, i32 *tri) {
  f32 distances[8];
  BVH_Node const *node = &scene->bvh.nodes.data[bvh_index];
  tl_counters.node_visits += 1;
  ray_aabbs_hit_8(ray, RT_EPS, hit->distance, node, distances);

  for (int i = 0; i < 8; i++) {
    f32 min_distance = hit->distance;
    i32 min_index    = -1;
    for (int j = 0; j < 8; j++) {
      if (distances[j] < min_distance) {
        min_distance = distances[j];
        min_index    = j;
      }
    }
    if (min_index == -1 || min_distance >= hit->distance) return;

    BVH_Index child = 8 * bvh_index + 1 + min_index;
    if (depth == 1) {
      tl_counters.leaf_visits += 1;
      ray_triangles_hit_8(ray, &scene->triangles, (child - scene->bvh.last_row_offset) * 8, hit, tri);
    } else {
      ray_bvh_node_hit(ray, scene, child, hit, depth - 1, tri);
    }
    distances[min_index] = RT_INF;
  }
}

/* raytracer.c:497-503, with deviation D3 for depth 0 */
static void ray_scene_hit(Ray const *ray, Scene const *scene, Hit *hit, i32 *tri) {
  tl_counters.rays += 1;
  if (scene->bvh.depth <= 0) {
    tl_counters.leaf_visits += 1;
    ray_triangles_hit_8(ray, &scene->triangles, 0, hit, tri);
    return;
  }
  ray_bvh_node_hit(ray, scene, 0, hit, scene->bvh.depth, tri);
}

/* ------------------------------------------------------------------------- */
/* driver.c:49-93  bilinear texture fetch with the reference's wrap rules     */
static rt_v3 sample_texture_bilinear(Image const *texture, f32 tx, f32 ty) {
  if (tx < 0) tx += (f32)(-(i32)tx + 1);
  if (ty < 0) ty += (f32)(-(i32)ty + 1);
  tx = rt_fractf(tx);
  ty = rt_fractf(ty);
  f32 px = tx * (f32)texture->width;
  f32 py = ty * (f32)texture->height;

  isize u = (isize)px;
  isize v = (isize)py;
  /* memory-safety clamp shared with the device code; never active for finite
   * coordinates (px < width is guaranteed by fract < 1) */
  if (u > texture->width - 1)  u = texture->width - 1;
  if (v > texture->height - 1) v = texture->height - 1;

  f32 a = px - (f32)u;
  f32 b = py - (f32)v;

  isize u2 = (u + 1 < texture->width)  ? u + 1 : u;
  isize v2 = (v + 1 < texture->height) ? v + 1 : v;

  byte const *p = texture->pixels.data;
  isize comp = texture->components, stride = texture->stride;
#define TEXEL(U_, V_) rt_v3_make(p[comp * ((U_) + stride * (V_)) + 0] / 255.999f, \
                                 p[comp * ((U_) + stride * (V_)) + 1] / 255.999f, \
                                 p[comp * ((U_) + stride * (V_)) + 2] / 255.999f)
  rt_v3 c00 = TEXEL(u,  v);
  rt_v3 c10 = TEXEL(u2, v);
  rt_v3 c01 = TEXEL(u,  v2);
  rt_v3 c11 = TEXEL(u2, v2);
#undef TEXEL
  rt_v3 c0 = rt_v3_lerp(c00, c10, a);
  rt_v3 c1 = rt_v3_lerp(c01, c11, a);
  return rt_v3_lerp(c0, c1, b);
}

/* driver.c:95-104  equirectangular environment lookup (deviation D4 inside rt_asinf) */
static rt_v3 sample_background_image(Image const *image, rt_v3 dir) {
  f32 inv_pi     = 1.0f / RT_PI;
  f32 inv_two_pi = 1.0f / (2.0f * RT_PI);
  if (tl_literal) {                        /* driver.c:96-97 with a double PI: one rounding each */
    inv_pi     = (f32)(1.0 / PI_D);
    inv_two_pi = (f32)(1.0 / (2.0 * PI_D));
  }
  f32 u = rt_madd(rt_atan2f(dir.z, dir.x), inv_two_pi, 0.5f);
  f32 v = rt_madd(-rt_asinf(dir.y), inv_pi, 0.5f);
  return rt_srgb_to_linear(sample_texture_bilinear(image, u, v));
}

/* driver.c:118-127 */
static rt_v3 sample_cosine_hemisphere(void) {
  f32 r1       = rand_f32();
  /* driver.c:119 `rand_f32() * 2 * PI`; literal: the float product times a double PI, rounded once */
  f32 angle    = tl_literal ? (f32)((double)(r1 * 2.0f) * PI_D) : r1 * 2.0f * RT_PI;
  f32 distance = rt_sqrtf(rand_f32());
  f32 s, c;
  rt_sincosf(angle, &s, &c);
  rt_v3 v = rt_v3_make(s * distance, c * distance, 0.0f);
  v.z = rt_sqrtf(rt_madd(-distance, distance, 1.0f));
  return v;
}

/* driver.c:129-153 */
static rt_v3 normal_map_apply(Image const *normal_map, f32 strength, Shader_Input const *input) {
  rt_v3 normal = V(input->normal);
  if (normal_map) {
    rt_v3 v = sample_texture_bilinear(normal_map, input->tex_coords.x, input->tex_coords.y);
    v = rt_v3_madd(v, 2.0f, rt_v3_make(-1.0f, -1.0f, -1.0f));
    v.y *= -1.0f;
    rt_v3 t = V(input->tangent), b = V(input->bitangent), n = V(input->normal);
    f32 s = strength;
    normal = rt_v3_normalize(rt_v3_make(
      rt_madd(s, rt_dot3(v.x, t.x, v.y, b.x, v.z, n.x), n.x * (1.0f - s)),
      rt_madd(s, rt_dot3(v.x, t.y, v.y, b.y, v.z, n.y), n.y * (1.0f - s)),
      rt_madd(s, rt_dot3(v.x, t.z, v.y, b.z, v.z, n.z), n.z * (1.0f - s))));
  }
  return normal;
}

/* driver.c:155-164 */
static void basis(rt_v3 view, rt_v3 normal, rt_v3 *tangent, rt_v3 *bitangent) {
  if (rt_absf(rt_v3_dot(normal, view)) < 0.9999f) {
    *tangent = rt_v3_normalize(rt_v3_cross(normal, view));
  } else if (rt_absf(rt_v3_dot(normal, rt_v3_make(0, 1, 0))) < 0.9999f) {
    *tangent = rt_v3_normalize(rt_v3_cross(normal, rt_v3_make(0, 1, 0)));
  } else {
    *tangent = rt_v3_normalize(rt_v3_cross(normal, rt_v3_make(1, 0, 0)));
  }
  *bitangent = rt_v3_cross(normal, *tangent);
}

/* driver.c:166-183 */
static f32 disney_fresnel_schlick_weight(f32 cos_theta) {
  f32 m = 1.0f - cos_theta;
  return m * m * m * m * m;
}

static rt_v3 disney_evaluate_sheen(f32 sheen, rt_v3 base_color, f32 sheen_tint, f32 h_dot_l) {
  if (sheen <= 0.0f) return rt_v3_make(0, 0, 0);
  f32 lum = rt_v3_dot(rt_v3_make(0.3f, 0.6f, 1.0f), base_color);
  rt_v3 tint = (lum > 0.0f) ? rt_v3_scale(base_color, 1.0f / lum) : rt_v3_make(1, 1, 1);
  return rt_v3_scale(rt_v3_lerp(rt_v3_make(1, 1, 1), tint, sheen_tint),
                     sheen * disney_fresnel_schlick_weight(h_dot_l));
}

/* driver.c:200-228; pow_f32(x, 5) and pow_f32(x, 2) are written as products (D5) */
static f32 luminance(rt_v3 x) { return rt_v3_dot(x, rt_v3_make(0.2126f, 0.7152f, 0.0722f)); }

static f32 pow5(f32 m) { return m * m * m * m * m; }

static f32 fresnel_schlick_f32(f32 f0, f32 f90, f32 theta) { return rt_madd(f90 - f0, pow5(1.0f - theta), f0); }

static rt_v3 fresnel_schlick_vec3(rt_v3 f0, f32 f90, f32 theta) {
  return rt_v3_madd(rt_v3_sub(rt_v3_make(f90, f90, f90), f0), pow5(1.0f - theta), f0);
}

static f32 distribution_GGX(f32 roughness, f32 NoH) {   /* k == 2 at every call site */
  f32 a2 = roughness * roughness;
  f32 d  = rt_madd(NoH * NoH, rt_madd(a2, a2, -1.0f), 1.0f);
  return a2 / (RT_PI * (d * d));
}

static f32 smith_G(f32 NDotV, f32 alpha2) {
  f32 a = alpha2 * alpha2;
  f32 b = NDotV * NDotV;
  /* driver.c:220 `(2.0 * NDotV) / (...)`: the double literal makes this a double division (D8) */
  if (tl_literal) return (f32)((2.0 * (double)NDotV) / (double)(NDotV + rt_sqrtf(a + b - a * b)));
  return (2.0f * NDotV) / (NDotV + rt_sqrtf(rt_madd(-a, b, a + b)));
}

static f32 geometry_term(f32 NoL, f32 NoV, f32 roughness) {
  f32 a2 = roughness * roughness;
  return smith_G(NoV, a2) * smith_G(NoL, a2);
}

/* driver.c:230-250 */
static rt_v3 sample_GGX_VNDF(rt_v3 Vv, f32 ax, f32 ay) {
  rt_v3 Vh = rt_v3_normalize(rt_v3_make(ax * Vv.x, ay * Vv.y, Vv.z));

  f32 lensq = rt_dot2(Vh.x, Vh.x, Vh.y, Vh.y);
  rt_v3 T1 = lensq > 0.0f ? rt_v3_scale(rt_v3_make(-Vh.y, Vh.x, 0.0f), 1.0f / rt_sqrtf(lensq)) : rt_v3_make(1, 0, 0);
  rt_v3 T2 = rt_v3_cross(Vh, T1);

  f32 r   = rt_sqrtf(rand_f32());
  f32 ru  = rand_f32();
  /* driver.c:238-246 carry double literals (D8): `2.0 * PI * rand`, `0.5 * (1.0 + Vh.z)`,
   * `(1.0 - s) * sqrt_f32(1.0 - t1 * t1) + s * t2`, `max(0.0, 1.0 - t1 * t1 - t2 * t2)` are evaluated in double and
   * rounded to f32 once where the default oracle rounds every step.  The literal form follows C's promotion rules. */
  f32 phi = tl_literal ? (f32)(2.0 * PI_D * (double)ru) : 2.0f * RT_PI * ru;
  f32 sn, cs;
  rt_sincosf(phi, &sn, &cs);
  f32 t1 = r * cs;
  f32 t2 = r * sn;
  f32 s  = rt_madd(0.5f, Vh.z, 0.5f);      /* 0.5 * (1 + Vh.z) in one rounding; the literal (double) form rounds once too */
  f32 tail;
  if (tl_literal) {
    t2   = (f32)((1.0 - (double)s) * (double)rt_sqrtf((f32)(1.0 - (double)(t1 * t1))) + (double)(s * t2));
    double rest = 1.0 - (double)(t1 * t1) - (double)(t2 * t2);
    tail = rt_sqrtf((f32)(rest > 0.0 ? rest : 0.0));
  } else {
    t2   = rt_madd(1.0f - s, rt_sqrtf(rt_madd(-t1, t1, 1.0f)), s * t2);
    tail = rt_sqrtf(rt_max_ps(0.0f, rt_madd(-t2, t2, rt_madd(-t1, t1, 1.0f))));
  }

  rt_v3 Nh = rt_v3_comb3(T1, t1, T2, t2, Vh, tail);

  return rt_v3_normalize(rt_v3_make(ax * Nh.x, ay * Nh.y, rt_max_ps(0.0f, Nh.z)));
}

/* driver.c:252-276 */
static f32 pdf_GGX_VNDF(f32 NoH, f32 NoV, f32 roughness) {
  f32 D  = distribution_GGX(roughness, NoH);
  f32 G1 = smith_G(NoV, roughness * roughness);
  return (D * G1) / rt_max_ps(0.00001f, 4.0f * NoV);
}

static rt_v3 disney_eval_diffuse(rt_v3 base_color, f32 NoL, f32 NoV, f32 LoH, f32 roughness) {
  f32 FD90 = rt_madd(2.0f * roughness * LoH, LoH, 0.5f);
  f32 a = fresnel_schlick_f32(1.0f, FD90, NoL);
  f32 b = fresnel_schlick_f32(1.0f, FD90, NoV);
  return rt_v3_scale(base_color, (a * b / RT_PI));
}

static rt_v3 disney_eval_specular(f32 roughness, rt_v3 F, f32 NoH, f32 NoV, f32 NoL) {
  f32 D = distribution_GGX(roughness, NoH);
  f32 G = geometry_term(NoL, NoV, roughness);
  return rt_v3_scale(F, D * G / (4.0f * NoL * NoV));
}

static f32 shadowed_f90(rt_v3 f0) {
  const f32 t = 1.0f / 0.04f;
  return rt_min_ps(1.0f, t * luminance(f0));
}

typedef struct {
  f32   roughness, metalness, sheen, sheen_tint, anisotropic_strength2;
  rt_v3 base_color;
} Disney_BRDF_Data;

/* driver.c:287-348; returns rgb in out[0..2] and the pdf-weight in out[3] */
static void sample_disney_BRDF(Disney_BRDF_Data const *data, rt_v3 in_dir, rt_v3 *out_dir, f32 brdf[4]) {
  f32 alpha_x = rt_lerpf(data->roughness * data->roughness, 1.0f, data->anisotropic_strength2);
  f32 alpha_y = data->roughness * data->roughness;
  rt_v3 micro_normal = sample_GGX_VNDF(in_dir, alpha_x, alpha_y);

  rt_v3 f0      = rt_v3_lerp(rt_v3_make(0.04f, 0.04f, 0.04f), data->base_color, data->metalness);
  rt_v3 fresnel = fresnel_schlick_vec3(f0, shadowed_f90(f0), rt_v3_dot(in_dir, micro_normal));

  f32 diffuse_weight  = 1.0f - data->metalness;
  f32 specular_weight = luminance(fresnel);
  f32 inverse_weight  = 1.0f / (diffuse_weight + specular_weight);
  diffuse_weight  *= inverse_weight;
  specular_weight *= inverse_weight;

  brdf[0] = brdf[1] = brdf[2] = brdf[3] = 0.0f;
  if (rand_f32() < diffuse_weight) {
    *out_dir     = sample_cosine_hemisphere();
    micro_normal = rt_v3_normalize(rt_v3_add(*out_dir, in_dir));

    f32 NoL = out_dir->z;
    f32 NoV = in_dir.z;
    if (NoL <= 0.0f || NoV <= 0.0f) return;
    f32 LoH = rt_v3_dot(*out_dir, micro_normal);
    f32 pdf = NoL / RT_PI;

    rt_v3 diff = rt_v3_mul(disney_eval_diffuse(data->base_color, NoL, NoV, LoH, data->roughness),
                           rt_v3_sub(rt_v3_make(1, 1, 1), fresnel));
    diff = rt_v3_add(diff, disney_evaluate_sheen(data->sheen, data->base_color, data->sheen_tint, LoH));
    brdf[0] = diff.x * NoL;
    brdf[1] = diff.y * NoL;
    brdf[2] = diff.z * NoL;
    brdf[3] = diffuse_weight * pdf;
  } else {
    *out_dir = rt_v3_reflect(rt_v3_scale(in_dir, -1.0f), micro_normal);

    f32 NoL = out_dir->z;
    f32 NoV = in_dir.z;
    if (NoL <= 0.0f || NoV <= 0.0f) return;
    NoL = rt_max_ps(NoL, 0.001f);
    NoV = rt_max_ps(NoV, 0.001f);
    f32 NoH = rt_min_ps(micro_normal.z, 0.99f);
    f32 pdf = pdf_GGX_VNDF(NoH, NoV, data->roughness);

    rt_v3 spec = disney_eval_specular(data->roughness, fresnel, NoH, NoV, NoL);
    brdf[0] = spec.x * NoL;
    brdf[1] = spec.y * NoL;
    brdf[2] = spec.z * NoL;
    brdf[3] = specular_weight * pdf;
  }
  *out_dir = rt_v3_normalize(*out_dir);
}

/* driver.c:350-409 */
static void oracle_disney_shader_proc(rawptr _data, Shader_Input const *input, Shader_Output *output) {
  PBR_Shader_Data const *data = (PBR_Shader_Data const *)_data;
  rt_v3 normal = normal_map_apply(data->texture_normal, data->normal_map_strength, input);

  if (data->texture_albedo || data->texture_normal || data->texture_metal_roughness || data->texture_emission) {
    tl_counters.textured += 1;
  }

  rt_v3 base_color = V(data->base_color);
  if (data->texture_albedo) {
    base_color = rt_v3_mul(base_color, rt_srgb_to_linear(
      sample_texture_bilinear(data->texture_albedo, input->tex_coords.x, input->tex_coords.y)));
  }

  f32 roughness = data->roughness;
  f32 metalness = data->metalness;
  if (data->texture_metal_roughness) {
    rt_v3 mr = sample_texture_bilinear(data->texture_metal_roughness, input->tex_coords.x, input->tex_coords.y);
    roughness *= mr.y;
    metalness *= mr.z;
  }

  roughness = rt_clampf(roughness, 0.001f, 1.0f);
  if (metalness > 0.9f) metalness = 0.9f;
  metalness /= 0.9f;

  rt_v3 emission = V(data->emission);
  if (data->texture_emission) {
    emission = rt_v3_mul(emission, rt_srgb_to_linear(
      sample_texture_bilinear(data->texture_emission, input->tex_coords.x, input->tex_coords.y)));
  }
  output->emission = U(emission);

  rt_v3 dir = V(input->direction);
  rt_v3 t, b;
  basis(dir, normal, &t, &b);

  Disney_BRDF_Data brdf_data;
  brdf_data.roughness             = roughness;
  brdf_data.metalness             = metalness;
  brdf_data.base_color            = base_color;
  brdf_data.sheen                 = data->sheen;
  brdf_data.sheen_tint            = data->sheen_tint;
  brdf_data.anisotropic_strength2 = data->anisotropic_strength * data->anisotropic_strength;

  /* world_to_tangent = transpose(from_basis(t, b, normal)) applied to -direction */
  rt_v3 neg    = rt_v3_scale(dir, -1.0f);
  rt_v3 in_dir = rt_v3_make(rt_v3_dot(t, neg), rt_v3_dot(b, neg), rt_v3_dot(normal, neg));
  rt_v3 o;
  f32   brdf[4];
  sample_disney_BRDF(&brdf_data, in_dir, &o, brdf);

  /* tangent_to_world: columns t, b, normal */
  output->direction = U(rt_v3_comb3(t, o.x, b, o.y, normal, o.z));

  if (brdf[3] > 0.0f) {
    output->tint.x = brdf[0] / brdf[3];
    output->tint.y = brdf[1] / brdf[3];
    output->tint.z = brdf[2] / brdf[3];
  } else {
    output->terminate = true;
  }
}

/* driver.c:411-418 */
static void oracle_debug_shader_proc(rawptr _data, Shader_Input const *input, Shader_Output *output) {
  PBR_Shader_Data const *data = (PBR_Shader_Data const *)_data;
  rt_v3 normal = normal_map_apply(data->texture_normal, data->normal_map_strength, input);
  output->emission  = U(rt_v3_madd(normal, 0.5f, rt_v3_make(0.5f, 0.5f, 0.5f)));
  output->terminate = true;
}

/* ------------------------------------------------------------------------- */
/* raytracer.c:505-558  one path                                              */
static rt_v3 cast_ray(Scene const *scene, Oracle_Config const *cfg, Ray ray, isize max_bounces) {
  rt_v3 accumulated_tint = rt_v3_make(1, 1, 1);
  rt_v3 emission         = rt_v3_make(0, 0, 0);

  for (isize i = 0; i < max_bounces; i++) {
    Hit hit;
    memset(&hit, 0, sizeof hit);
    hit.distance = RT_INF;
    ray_scene_hit(&ray, scene, &hit, NULL);
    if (hit.distance != RT_INF) {
      rt_v3 rd = V(ray.direction);
      if (rt_v3_dot(V(hit.normal_geo), rd) > 0.0f || rt_v3_dot(V(hit.normal), rd) > 0.0f) {
        ray.position = U(rt_v3_madd(rd, RT_EPS, V(hit.point)));
        continue;
      }

      Shader_Input shader_input;
      shader_input.direction  = ray.direction;
      shader_input.normal     = U(rt_v3_normalize(V(hit.normal)));
      shader_input.normal_geo = hit.normal_geo;
      shader_input.tangent    = hit.tangent;
      shader_input.bitangent  = hit.bitangent;
      shader_input.position   = hit.point;
      shader_input.tex_coords = hit.tex_coords;
      Shader_Output shader_output;
      memset(&shader_output, 0, sizeof shader_output);

      tl_counters.shades += 1;
      if (hit.shader.proc == cfg->disney_proc && cfg->disney_proc) {
        oracle_disney_shader_proc(hit.shader.data, &shader_input, &shader_output);
      } else if (hit.shader.proc == cfg->debug_proc && cfg->debug_proc) {
        oracle_debug_shader_proc(hit.shader.data, &shader_input, &shader_output);
      } else {
        hit.shader.proc(hit.shader.data, &shader_input, &shader_output);
      }

      emission = rt_v3_mul_add(V(shader_output.emission), accumulated_tint, emission);
      if (shader_output.terminate) break;

      ray.direction    = shader_output.direction;
      accumulated_tint = rt_v3_mul(accumulated_tint, V(shader_output.tint));

      f32 below = (rt_v3_dot(V(hit.normal_geo), V(shader_output.direction)) < 0.0f) ? 1.0f : 0.0f;
      f32 position_bias = (0.5f - below) * 2.0f * RT_EPS;
      ray.position = U(rt_v3_madd(V(hit.normal_geo), position_bias, V(hit.point)));
    } else {
      tl_counters.backgrounds += 1;
      rt_v3 bg;
      if (scene->background.proc == cfg->background_proc && cfg->background_proc) {
        bg = sample_background_image((Image const *)scene->background.data, V(ray.direction));
      } else {
        bg = V(scene->background.proc(scene->background.data, ray.direction));
      }
      return rt_v3_mul_add(bg, accumulated_tint, emission);
    }
  }
  return emission;
}

/* raytracer.c:641-694  primary ray of (x, y, sample); rand_a == rand_b is the
 * reference's own sampling pattern; deviation D2: exact 1/sqrt */
static Ray primary_ray(Camera const *camera, i32 width, i32 height, i32 x, i32 y, i32 sample) {
  f32 inv_width  = 1.0f / (f32)width;
  f32 inv_height = 1.0f / (f32)height;
  f32 aspect     = (f32)width / (f32)height;

  f32 jitter = rt_hash12((f32)x * 50.0f + (f32)sample, (f32)y);
  f32 rand_a = jitter, rand_b = jitter;

  f32 uvx = rt_madd(((f32)x + rand_a - 0.5f) * 2.0f, inv_width,  -1.0f);
  f32 uvy = rt_madd(((f32)y + rand_b - 0.5f) * 2.0f, inv_height, -1.0f);

  f32 dx = uvx * aspect;
  f32 dy = -uvy;
  f32 dz = -camera->focal_length;

  f32 inv_length = 1.0f / rt_sqrtf(rt_dot3(dx, dx, dy, dy, dz, dz));
  if (tl_literal) {                        /* raytracer.c:663: the ~12-bit hardware estimate, not 1/sqrt (D2) */
    inv_length = _mm_cvtss_f32(_mm256_castps256_ps128(_mm256_rsqrt_ps(_mm256_set1_ps(dx * dx + dy * dy + dz * dz))));
  }

  f32 const (*m)[4] = camera->view_matrix.rows;
  f32 rx = rt_dot3(m[0][0], dx, m[0][1], dy, m[0][2], dz);
  f32 ry = rt_dot3(m[1][0], dx, m[1][1], dy, m[1][2], dz);
  f32 rz = rt_dot3(m[2][0], dx, m[2][1], dy, m[2][2], dz);

  Ray r;
  /* camera_position = view_matrix * (0,0,0,1), raytracer.c:612 */
  r.position.x = m[0][3];
  r.position.y = m[1][3];
  r.position.z = m[2][3];
  r.direction.x = rx * inv_length;
  r.direction.y = ry * inv_length;
  r.direction.z = rz * inv_length;
  return r;
}

static rt_v3 trace_path(Scene const *scene, Oracle_Config const *cfg, i32 width, i32 height,
                        i32 x, i32 y, i32 sample, i32 max_bounces) {
  if (!tl_literal) random_state = rt_path_seed(cfg->seed, (u32)(x + y * width), (u32)sample);   /* D1 */
  tl_counters.paths += 1;
  Ray r = primary_ray(&scene->camera, width, height, x, y, sample);
  return cast_ray(scene, cfg, r, max_bounces);
}

void oracle_trace_path(Scene const *scene, Oracle_Config const *config, i32 width, i32 height,
                       i32 x, i32 y, i32 sample, i32 samples, i32 max_bounces, f32 rgb[3]) {
  (void)samples;
  rt_v3 c = trace_path(scene, config, width, height, x, y, sample, max_bounces);
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* ------------------------------------------------------------------------- */
/* raytracer.c:596-720  chunked pixel/sample loop, threads claim 32x32 chunks */

typedef struct {
  Scene const         *scene;
  Image const         *image;
  Oracle_Config        cfg;
  i32                  samples, max_bounces;
  i32                  width, height;
  f32                 *linear;
  u64                 *accum;
  atomic_int           current_chunk;
  pthread_mutex_t      lock;
  Oracle_Counters      total;
} Oracle_Job;

static void *oracle_worker(void *arg) {
  Oracle_Job *job = (Oracle_Job *)arg;
  memset(&tl_counters, 0, sizeof tl_counters);
  /* ORACLE_LITERAL: one RNG stream per thread, seeded once and running on across pixels and chunks
   * (raytracer.c:597 `random_state = time_now()`, with the frame seed in place of the clock) */
  tl_literal = job->cfg.literal != 0;
  if (tl_literal) random_state = job->cfg.seed;

  i32 width = job->width, height = job->height;
  i32 x0 = job->cfg.x0, y0 = job->cfg.y0, x1 = job->cfg.x1, y1 = job->cfg.y1;
  i32 chunks_x = (width + RT_CHUNK_SIZE - 1) / RT_CHUNK_SIZE;
  i32 chunks_y = (height + RT_CHUNK_SIZE - 1) / RT_CHUNK_SIZE;
  i32 n_chunks = chunks_x * chunks_y;
  i32 samples  = job->samples;
  i32 s_begin  = job->cfg.sample0;
  i32 s_end    = job->cfg.sample_count > 0 ? s_begin + job->cfg.sample_count : samples;
  if (s_end > samples) s_end = samples;
  f32 inv_samples = 1.0f / (f32)samples;

  for (;;) {
    i32 c = atomic_fetch_add(&job->current_chunk, 1);
    if (c >= n_chunks) break;
    i32 start_x = (c % chunks_x) * RT_CHUNK_SIZE;
    i32 start_y = (c / chunks_x) * RT_CHUNK_SIZE;
    for (i32 y = start_y; y < start_y + RT_CHUNK_SIZE && y < height; y++) {
      if (y < y0 || y >= y1) continue;
      for (i32 x = start_x; x < start_x + RT_CHUNK_SIZE && x < width; x++) {
        if (x < x0 || x >= x1) continue;

        rt_v3 color = rt_v3_make(0, 0, 0);
        u64   q[3]  = {0, 0, 0};
        for (i32 s = s_begin; s < s_end; s++) {
          rt_v3 c3 = trace_path(job->scene, &job->cfg, width, height, x, y, s, job->max_bounces);
          color = rt_v3_add(color, c3);
          q[0] += rt_accum_quantize(c3.x);
          q[1] += rt_accum_quantize(c3.y);
          q[2] += rt_accum_quantize(c3.z);
        }

        f32 lin[3];
        if (job->cfg.accum_mode == ORACLE_ACCUM_F32 || tl_literal) {
          color = rt_v3_scale(color, inv_samples);            /* raytracer.c:700 */
          lin[0] = color.x; lin[1] = color.y; lin[2] = color.z;
        } else {
          for (int k = 0; k < 3; k++) lin[k] = rt_accum_resolve(q[k], (u32)samples);
        }
        isize pix = (isize)x + (isize)y * width;
        if (job->accum)  for (int k = 0; k < 3; k++) job->accum[3 * pix + k] = q[k];
        if (job->linear) for (int k = 0; k < 3; k++) job->linear[3 * pix + k] = lin[k];
        if (job->image && job->image->pixels.data) {
          Image const *im = job->image;
          for (int k = 0; k < 3; k++) {
            im->pixels.data[im->components * (x + y * im->stride) + k] = rt_encode_u8(lin[k]);
          }
        }
      }
    }
  }

  pthread_mutex_lock(&job->lock);
  job->total.paths       += tl_counters.paths;
  job->total.rays        += tl_counters.rays;
  job->total.node_visits += tl_counters.node_visits;
  job->total.leaf_visits += tl_counters.leaf_visits;
  job->total.shades      += tl_counters.shades;
  job->total.backgrounds += tl_counters.backgrounds;
  job->total.textured    += tl_counters.textured;
  pthread_mutex_unlock(&job->lock);
  return NULL;
}

int oracle_render(Scene const *scene, Image const *image, isize samples, isize max_bounces,
                  Oracle_Config const *config, f32 *linear, u64 *accum, Oracle_Counters *counters) {
  if (!scene || !config || !image || samples <= 0 || max_bounces < 0) return -1;
  if (image->width <= 0 || image->height <= 0) return -1;
  if (image->pixels.data && image->components < 3) return -1;

  Oracle_Job job;
  memset(&job, 0, sizeof job);
  job.scene = scene;
  job.image = image;
  job.cfg = *config;
  job.samples = (i32)samples;
  job.max_bounces = (i32)max_bounces;
  job.width = (i32)image->width;
  job.height = (i32)image->height;
  job.linear = linear;
  job.accum = accum;
  if (job.cfg.x1 <= 0) job.cfg.x1 = job.width;
  if (job.cfg.y1 <= 0) job.cfg.y1 = job.height;
  atomic_init(&job.current_chunk, 0);
  pthread_mutex_init(&job.lock, NULL);

  i32 n = config->n_threads < 1 ? 1 : config->n_threads;
  if (n > 256) n = 256;
  if (config->literal) n = 1;              /* the literal stream is reproducible with one thread only */
  pthread_t threads[256];
  for (i32 i = 1; i < n; i++) pthread_create(&threads[i], NULL, oracle_worker, &job);
  oracle_worker(&job);
  for (i32 i = 1; i < n; i++) pthread_join(threads[i], NULL);
  pthread_mutex_destroy(&job.lock);
  tl_literal = false;                      /* unit-level entry points below always use the default semantics */

  if (counters) *counters = job.total;
  return 0;
}

/* ------------------------------------------------------------------------- */
/* unit-level entry points                                                     */

void oracle_rand_u32_seq(u32 state, i32 n, u32 *out) {
  for (i32 i = 0; i < n; i++) out[i] = rt_rand_u32(&state);
}

void oracle_rand_f32_seq(u32 state, i32 n, f32 *out) {
  for (i32 i = 0; i < n; i++) out[i] = rt_rand_f32(&state);
}

f32 oracle_hash12(f32 px, f32 py) { return rt_hash12(px, py); }

void oracle_ray_aabbs_hit_8(Ray const *ray, f32 t_min, f32 t_max, BVH_Node const *node, f32 *distances) {
  ray_aabbs_hit_8(ray, t_min, t_max, node, distances);
}

bool oracle_ray_triangles_hit_8(Ray const *ray, Triangles const *tris, isize offset, Hit *hit, i32 *lane) {
  return ray_triangles_hit_8(ray, tris, offset, hit, lane);
}

void oracle_ray_scene_hit(Ray const *ray, Scene const *scene, Hit *hit, i32 *triangle) {
  if (triangle) *triangle = -1;
  ray_scene_hit(ray, scene, hit, triangle);
}

void oracle_trace_rays(Scene const *scene, i32 n, f32 const *rays, f32 *out_t, i32 *out_tri, f32 *out_uv) {
  for (i32 i = 0; i < n; i++) {
    Ray r;
    r.position.x = rays[i * 6 + 0]; r.position.y = rays[i * 6 + 1]; r.position.z = rays[i * 6 + 2];
    r.direction.x = rays[i * 6 + 3]; r.direction.y = rays[i * 6 + 4]; r.direction.z = rays[i * 6 + 5];
    Hit hit;
    memset(&hit, 0, sizeof hit);
    hit.distance = RT_INF;
    i32 tri = -1;
    ray_scene_hit(&r, scene, &hit, &tri);
    out_t[i] = hit.distance;
    out_tri[i] = tri;
    out_uv[i * 2 + 0] = (tri >= 0) ? tl_bary_u : 0.0f;
    out_uv[i * 2 + 1] = (tri >= 0) ? tl_bary_v : 0.0f;
  }
}

/* oracle_trace_rays + the node / leaf visits those rays cost (ray_aabbs_hit_8 / ray_triangles_hit_8 calls, raytracer.c:452,476) */
void oracle_trace_rays_counted(Scene const *scene, i32 n, f32 const *rays, f32 *out_t, i32 *out_tri, f32 *out_uv, u64 visits[2]) {
  u64 n0 = tl_counters.node_visits, l0 = tl_counters.leaf_visits;
  oracle_trace_rays(scene, n, rays, out_t, out_tri, out_uv);
  visits[0] = tl_counters.node_visits - n0;
  visits[1] = tl_counters.leaf_visits - l0;
}

void oracle_sample_texture_bilinear(Image const *texture, f32 u, f32 v, f32 rgb[3]) {
  rt_v3 c = sample_texture_bilinear(texture, u, v);
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

void oracle_sample_background(Image const *image, f32 const dir[3], f32 rgb[3]) {
  rt_v3 c = sample_background_image(image, rt_v3_make(dir[0], dir[1], dir[2]));
  rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

void oracle_sample_disney_brdf(f32 roughness, f32 metalness, f32 sheen, f32 sheen_tint, f32 aniso2,
                               f32 const base_color[3], f32 const in_dir[3], u32 *state, f32 out_dir[3], f32 brdf[4]) {
  Disney_BRDF_Data d;
  d.roughness = roughness; d.metalness = metalness; d.sheen = sheen; d.sheen_tint = sheen_tint;
  d.anisotropic_strength2 = aniso2;
  d.base_color = rt_v3_make(base_color[0], base_color[1], base_color[2]);
  random_state = *state;
  rt_v3 o = rt_v3_make(0, 0, 0);
  sample_disney_BRDF(&d, rt_v3_make(in_dir[0], in_dir[1], in_dir[2]), &o, brdf);
  *state = random_state;
  out_dir[0] = o.x; out_dir[1] = o.y; out_dir[2] = o.z;
}

void oracle_disney_shade(PBR_Shader_Data const *data, Shader_Input const *in, u32 *state, Shader_Output *out) {
  random_state = *state;
  memset(out, 0, sizeof *out);
  oracle_disney_shader_proc((rawptr)data, in, out);
  *state = random_state;
}

void oracle_math(i32 op, i32 n, f32 const *x, f32 const *y, f32 *out) {
  for (i32 i = 0; i < n; i++) {
    f32 a = x[i], b = y ? y[i] : 0.0f, s, c;
    switch (op) {
    case 0: out[i] = rt_logf(a); break;
    case 1: out[i] = rt_expf(a); break;
    case 2: out[i] = rt_powf(a, b); break;
    case 3: rt_sincosf(a, &s, &c); out[i] = s; break;
    case 4: rt_sincosf(a, &s, &c); out[i] = c; break;
    case 5: out[i] = rt_atan2f(a, b); break;
    case 6: out[i] = rt_asinf(a); break;
    case 7: out[i] = rt_srgb_to_linear1(a); break;
    case 8: out[i] = rt_linear_to_srgb(a); break;
    case 9: out[i] = rt_sqrtf(a); break;
    case 10: out[i] = 1.0f / a; break;
    default: out[i] = 0.0f; break;
    }
  }
}

u8 oracle_encode_u8(f32 linear) { return rt_encode_u8(linear); }

/* ------------------------------------------------------------------------- */
/* denoiser.c:13-153  (SURVEY.md section 8f #3).  Single threaded: every output pixel depends only on
 * the source image, so the reference's chunked threading (denoiser.c:51-129) does not affect results. */

#define DENOISING_THRESHOLD  0.0125f       /* denoiser.c:13 */
#define NEIGHBOURHOOD_WEIGHT 5             /* denoiser.c:14 */

/* denoiser.c:20-30: clamp-to-edge fetch, u8 / 255.999f */
static void dn_sample(Image const *image, isize x, isize y, f32 rgb[3]) {
  if (x < 0) x = 0;
  if (y < 0) y = 0;
  if (x >= image->width)  x = image->width - 1;
  if (y >= image->height) y = image->height - 1;
  rgb[0] = rgb[1] = rgb[2] = 0.0f;
  isize nc = image->components < 3 ? image->components : 3;
  for (isize c = 0; c < nc; c++) {
    rgb[c] = image->pixels.data[(x + y * image->stride) * image->components + c] / 255.999f;
  }
}

void oracle_denoise_image(Image const *src, Image const *dst) {
  isize width = src->width, height = src->height;
  for (isize y = 0; y < height; y++) {
    for (isize x = 0; x < width; x++) {
      f32 colors[9][4];
      f32 original[4] = {0, 0, 0, 0};
      isize n_colors = 0;
      for (isize yo = -1; yo < 2; yo++) {
        for (isize xo = -1; xo < 2; xo++) {
          f32 color[4];
          dn_sample(src, x + xo, y + yo, color);
          color[3] = color[0] * 0.2126f + color[1] * 0.7152f + color[2] * 0.0722f;     /* denoiser.c:16-18 */
          if (xo == 0 && yo == 0) memcpy(original, color, sizeof color);
          /* denoiser.c:85-101: insertion before the first strictly brighter entry */
          bool found = false;
          for (isize i = 0; i < n_colors; i++) {
            if (colors[i][3] > color[3]) {
              found = true;
              for (isize j = n_colors; j > i; j--) memcpy(colors[j], colors[j - 1], sizeof color);
              memcpy(colors[i], color, sizeof color);
              break;
            }
          }
          if (!found) memcpy(colors[n_colors], color, sizeof color);
          n_colors += 1;
        }
      }
      f32 const *median = colors[4];                    /* denoiser.c:104 */
      f32 mean = 0;
      for (isize i = 1; i < 8; i++) mean += colors[i][3];     /* all but the darkest and the brightest */
      mean /= 7;
      f32 noisiness = rt_absf(median[3] - mean);
      f32 diff = rt_absf(median[3] - original[3]) - noisiness * NEIGHBOURHOOD_WEIGHT;
      diff = rt_clampf(diff, 0.0f, DENOISING_THRESHOLD) / DENOISING_THRESHOLD;
      isize nc = dst->components < 3 ? dst->components : 3;
      for (isize c = 0; c < nc; c++) {                    /* store_pixel, denoiser.c:32-41 */
        f32 v = rt_lerpf_plain(original[c], median[c], diff);
        dst->pixels.data[(x + y * dst->stride) * dst->components + c] = (u8)(v * 255.999f);
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* raytracer.c:722-784  lightmap_bake               */

/* common.h:26-42 */
static f32 rand_f32_range(f32 lo, f32 hi) { return rand_f32() * (hi - lo) + lo; }

static rt_v3 rand_vec3(void) {
  for (;;) {
    rt_v3 p;
    p.x = rand_f32_range(-1.0f, 1.0f);
    p.y = rand_f32_range(-1.0f, 1.0f);
    p.z = rand_f32_range(-1.0f, 1.0f);
    f32 lensq = rt_v3_dot(p, p);
    if (RT_EPS < lensq && lensq <= 1.0f) return rt_v3_scale(p, 1.0f / rt_sqrtf(lensq));
  }
}

static f32 min3(f32 a, f32 b, f32 c) { f32 m = b < c ? b : c; return a < m ? a : m; }    /* min(a, min(b, c)) */
static f32 max3(f32 a, f32 b, f32 c) { f32 m = b > c ? b : c; return a > m ? a : m; }

void oracle_lightmap_bake(Image const *lightmap, Scene const *scene, isize samples, Oracle_Config const *cfg) {
  memset(&tl_counters, 0, sizeof tl_counters);
  f32 lw = (f32)lightmap->width, lh = (f32)lightmap->height;
  Triangles const *T = &scene->triangles;
  for (isize i = 0; i < T->len; i++) {
    Triangle_AOS aos = T->aos[i];
    i32 min_x = (i32)(min3(aos.tex_coords_a.x, aos.tex_coords_b.x, aos.tex_coords_c.x) * lw);
    i32 max_x = (i32)(max3(aos.tex_coords_a.x, aos.tex_coords_b.x, aos.tex_coords_c.x) * lw);
    i32 min_y = (i32)(min3(aos.tex_coords_a.y, aos.tex_coords_b.y, aos.tex_coords_c.y) * lh);
    i32 max_y = (i32)(max3(aos.tex_coords_a.y, aos.tex_coords_b.y, aos.tex_coords_c.y) * lh);

    f32 p0x = aos.tex_coords_a.x * lw, p0y = aos.tex_coords_a.y * lh;
    f32 p1x = aos.tex_coords_b.x * lw, p1y = aos.tex_coords_b.y * lh;
    f32 p2x = aos.tex_coords_c.x * lw, p2y = aos.tex_coords_c.y * lh;
    f32 denom = (p1y - p2y) * (p0x - p2x) + (p2x - p1x) * (p0y - p2y);

    for (i32 y = min_y; y < max_y + 1; y++) {
      for (i32 x = min_x; x < max_x + 1; x++) {
        if (x < 0 || y < 0 || x >= lightmap->width || y >= lightmap->height) continue;   /* reference: out of bounds */
        f32 px = (f32)x, py = (f32)y;
        f32 w0 = ((p1y - p2y) * (px - p2x) + (p2x - p1x) * (py - p2y)) / denom;
        f32 w1 = ((p2y - p0y) * (px - p2x) + (p0x - p2x) * (py - p2y)) / denom;
        f32 w2 = 1.0f - w0 - w1;
        if (w0 >= -RT_EPS && w1 >= -RT_EPS && w2 >= -RT_EPS) {
          rt_v3 position = rt_v3_make(T->x[0][i] * w0 + T->x[1][i] * w1 + T->x[2][i] * w2,
                                      T->y[0][i] * w0 + T->y[1][i] * w1 + T->y[2][i] * w2,
                                      T->z[0][i] * w0 + T->z[1][i] * w1 + T->z[2][i] * w2);
          rt_v3 normal = rt_v3_make(aos.normal_a.x * w0 + aos.normal_b.x * w1 + aos.normal_c.x * w2,
                                    aos.normal_a.y * w0 + aos.normal_b.y * w1 + aos.normal_c.y * w2,
                                    aos.normal_a.z * w0 + aos.normal_b.z * w1 + aos.normal_c.z * w2);
          rt_v3 accumulated = rt_v3_make(0, 0, 0);
          Ray r;
          r.position = U(rt_v3_add(position, rt_v3_scale(normal, RT_EPS)));
          r.direction = U(rt_v3_make(0, 0, 0));
          random_state = rt_path_seed(cfg->seed, (u32)(x + y * (i32)lightmap->width), (u32)i);
          for (isize s = 0; s < samples; s++) {
            f32 cosv;
            i32 guard = 0;
            for (;;) {
              rt_v3 d = rand_vec3();
              cosv = rt_v3_dot(d, normal);
              if (cosv > 0.0f) { r.direction = U(d); break; }
              if (++guard >= 64) { cosv = 0.0f; r.direction = U(d); break; }   /* zero/NaN normal: the reference would spin forever */
            }
            accumulated = rt_v3_add(accumulated, rt_v3_scale(cast_ray(scene, cfg, r, 8), cosv));
          }
          f32 v[3] = { accumulated.x / (f32)samples, accumulated.y / (f32)samples, accumulated.z / (f32)samples };
          for (int c = 0; c < 3; c++) {
            f32 q = v[c] > 0.0f ? v[c] : 0.0f;          /* NaN and negatives -> 0 */
            q = q > 255.0f ? 255.0f : q;
            lightmap->pixels.data[(x + y * lightmap->stride) * lightmap->components + c] = (u8)q;
          }
        }
      }
    }
  }
}
