/* rt_math.h -- the numeric contract of the render path, single source for the
 * host C compiler and for hipcc (gfx950 device code).
 *
 * The reference leans on an absent third-party library for every scalar math
 * call on the path (codin `linalg.h`: vec3_normalize / vec3_lerp /
 * vec3_reflect; codin math wrappers: pow_f32, sin_f32, cos_f32, atan2_f32,
 * asin_f32, sqrt_f32 -- call sites common.h:84-91, driver.c:99-100,122-123,
 * 239-240, raytracer.c:526) and ships no tests, so their ulp-level behaviour is
 * unpinned.  This header defines it ONCE, using only
 * operations that IEEE-754 rounds identically on x86-64 and gfx950:
 *   + - * /  sqrt  floor  int<->float conversion  comparisons  bit casts
 * plus EXPLICIT fused multiply-adds (fmaf: one rounding, identical on x86-64 FMA
 * units and gfx950 v_fma_f32).  Both compilers MUST be run with
 * -ffp-contract=off (no IMPLICIT contraction anywhere) and without fast-math
 * (hipcc additionally keeps its default correctly rounded fp32 divide/sqrt);
 * then every function below returns bit-identical results on CPU and GPU, which
 * is what lets tests/ demand bit-exact images instead of a tolerance.
 *
 * Numeric contract v2 (round 4), "explicit FMA":
 *   * the polynomial kernels of the elementary functions use rt_fmaf Horner
 *     steps (they replace libm calls: there is no reference operation order);
 *   * every  a*b + c  of the path's vector algebra -- dots, crosses, lerps,
 *     reflections, the 3x3 products, point = origin + t * direction -- is ONE
 *     rt_madd(a, b, c): the multiply-add a CPU build of the reference gets from
 *     GNU C's default -ffp-contract=fast (raytracer.c:131-147 are
 *     `a*b + c*d + e*f` on vector types) and a single v_fma_f32 on gfx950.
 *     Sums of products are folded LEFT TO RIGHT:
 *         a0*b0 + a1*b1 + a2*b2  ==  madd(a2, b2, madd(a1, b1, a0*b0)),
 *     differences  a*b - c*d  as  madd(a, b, -(c*d)).
 *   * -DRT_MATH_NO_FMA (on BOTH compilers) restores contract v1: rt_madd(a,b,c)
 *     is a*b + c with two roundings, nothing else changes.  The two contracts are
 *     the same estimator (tests/test_oracle_contracts.py); bit parity between CPU
 *     and GPU holds under either.
 *   * functions with the suffix _plain never fuse: the BVH / tangent-frame
 *     builders (scene.c arithmetic, pinned by the survey's shape table and
 *     traversal statistics) and the denoiser (pinned by a numpy restatement)
 *     keep their arithmetic under both contracts.
 *
 * Numeric contract v3 (round 5) = v2 with a cheaper, more accurate power for the
 * sRGB decode (rt_pow24 below); -DRT_MATH_V2 keeps round 4's arithmetic.
 *
 * Polynomial coefficients are the classic single-precision minimax sets
 * (Cephes, S. Moshier) for log/exp/sin/cos/atan/asin.
 */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RT_FN __host__ __device__ static inline __attribute__((always_inline))
#else
#define RT_FN static inline __attribute__((always_inline))
#endif

#define RT_PI      3.14159265358979323846f
#define RT_EPS     0.0001f                       /* common.h:8 */
#define RT_INF     __builtin_inff()

typedef struct { float x, y, z; } rt_v3;
typedef struct { float x, y; } rt_v2;

/* ---- bit casts / basic selects --------------------------------------------- */

RT_FN uint32_t rt_f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
RT_FN float    rt_u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

/* _mm256_min_ps / _mm256_max_ps operand semantics (raytracer.c:212-225): the
 * SECOND operand is returned when either one is NaN. */
RT_FN float rt_min_ps(float a, float b) { return a < b ? a : b; }
RT_FN float rt_max_ps(float a, float b) { return a > b ? a : b; }

RT_FN float rt_clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
RT_FN float rt_absf(float x) { return rt_u2f(rt_f2u(x) & 0x7fffffffu); }
RT_FN float rt_sqrtf(float x) { return __builtin_sqrtf(x); }
RT_FN float rt_floorf(float x) { return __builtin_floorf(x); }
RT_FN float rt_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }   /* a*b+c, ONE rounding */
RT_FN float rt_fractf(float x) { return x - rt_floorf(x); }   /* raytracer.c:582 */

/* The multiply-add of the numeric contract (header comment): fused under contract v2, two roundings under
 * -DRT_MATH_NO_FMA (contract v1). */
#ifdef RT_MATH_NO_FMA
#define RT_MATH_CONTRACT 1
RT_FN float rt_madd(float a, float b, float c) { return a * b + c; }
#else
#ifdef RT_MATH_V2
#define RT_MATH_CONTRACT 2
#else
#define RT_MATH_CONTRACT 3
#endif
RT_FN float rt_madd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#endif
/* Contract v3 (round 5) = v2 + the sRGB decode's power through rt_pow24() below instead of exp(2.4 log x): a change of this
 * build's OWN libm stand-in (the reference calls codin's pow_f32, common.h:84), closer to the true power than the old one and
 * half its instructions.  -DRT_MATH_V2 (on both compilers) keeps round 4's decode; contract v1 implies it. */
#if RT_MATH_CONTRACT >= 3
#define RT_MATH_POW24 1
#else
#define RT_MATH_POW24 0
#endif
/* a0*b0 + a1*b1 + a2*b2, folded left to right */
RT_FN float rt_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  return rt_madd(a2, b2, rt_madd(a1, b1, a0 * b0));
}
/* a0*b0 + a1*b1 */
RT_FN float rt_dot2(float a0, float b0, float a1, float b1) { return rt_madd(a1, b1, a0 * b0); }
/* a*b - c*d */
RT_FN float rt_diff2(float a, float b, float c, float d) { return rt_madd(a, b, -(c * d)); }

/* ---- RNG (common.h:13-24) --------------------------------------------------- */

/* One step of the reference generator: returns the new state. */
RT_FN uint32_t rt_pcg(uint32_t v) {
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word  = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}

RT_FN uint32_t rt_rand_u32(uint32_t *state) {
  *state = rt_pcg(*state);
  return *state;
}

/* rand_u32() / (f32)U32_MAX; (f32)U32_MAX rounds to 2^32, the result lies in
 * [0, 1] INCLUSIVE (common.h:22-24). */
RT_FN float rt_rand_f32(uint32_t *state) {
  return (float)rt_rand_u32(state) / 4294967296.0f;
}

/* Per-path seeding rule of this build (replaces the wall-clock, per-thread
 * seeding of raytracer.c:597 -- SURVEY.md F4/H2): the stream of one path
 * depends only on (frame seed, pixel index, sample index). */
RT_FN uint32_t rt_path_seed(uint32_t seed, uint32_t pixel_index, uint32_t sample) {
  return rt_pcg(rt_pcg(seed ^ pixel_index) + sample);
}

/* ---- vectors ----------------------------------------------------------------- */

RT_FN rt_v3 rt_v3_make(float x, float y, float z) { rt_v3 v; v.x = x; v.y = y; v.z = z; return v; }
RT_FN rt_v3 rt_v3_add(rt_v3 a, rt_v3 b) { return rt_v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_FN rt_v3 rt_v3_sub(rt_v3 a, rt_v3 b) { return rt_v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_FN rt_v3 rt_v3_mul(rt_v3 a, rt_v3 b) { return rt_v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_FN rt_v3 rt_v3_scale(rt_v3 a, float s) { return rt_v3_make(a.x * s, a.y * s, a.z * s); }
RT_FN float rt_v3_dot(rt_v3 a, rt_v3 b) { return rt_dot3(a.x, b.x, a.y, b.y, a.z, b.z); }
/* v * s + w */
RT_FN rt_v3 rt_v3_madd(rt_v3 v, float s, rt_v3 w) { return rt_v3_make(rt_madd(v.x, s, w.x), rt_madd(v.y, s, w.y), rt_madd(v.z, s, w.z)); }
/* a * b + c, component-wise */
RT_FN rt_v3 rt_v3_mul_add(rt_v3 a, rt_v3 b, rt_v3 c) { return rt_v3_make(rt_madd(a.x, b.x, c.x), rt_madd(a.y, b.y, c.y), rt_madd(a.z, b.z, c.z)); }
/* a * s + b * t + c * u (tangent_to_world, driver.c:398; barycentric interpolation, raytracer.c:167-176) */
RT_FN rt_v3 rt_v3_comb3(rt_v3 a, float s, rt_v3 b, float t, rt_v3 c, float u) {
  return rt_v3_make(rt_dot3(a.x, s, b.x, t, c.x, u), rt_dot3(a.y, s, b.y, t, c.y, u), rt_dot3(a.z, s, b.z, t, c.z, u));
}

/* common.h:54-60 */
RT_FN rt_v3 rt_v3_cross(rt_v3 a, rt_v3 b) {
  return rt_v3_make(rt_diff2(a.y, b.z, a.z, b.y), rt_diff2(a.z, b.x, a.x, b.z), rt_diff2(a.x, b.y, a.y, b.x));
}

/* codin vec3_normalize: defined here as v * (1 / sqrt(v.v)). */
RT_FN rt_v3 rt_v3_normalize(rt_v3 v) {
  float inv = 1.0f / rt_sqrtf(rt_v3_dot(v, v));
  return rt_v3_scale(v, inv);
}

/* lerp_f32 of driver.c:283-285; codin vec3_lerp is defined component-wise the same. */
RT_FN float rt_lerpf(float x, float y, float t) { return rt_madd(y, t, x * (1.0f - t)); }
RT_FN rt_v3 rt_v3_lerp(rt_v3 a, rt_v3 b, float t) {
  return rt_v3_make(rt_lerpf(a.x, b.x, t), rt_lerpf(a.y, b.y, t), rt_lerpf(a.z, b.z, t));
}

/* codin vec3_reflect(v, n) = v - 2 (v.n) n (convention checked against
 * output.png by the survey probe). */
RT_FN rt_v3 rt_v3_reflect(rt_v3 v, rt_v3 n) {
  return rt_v3_madd(n, -(2.0f * rt_v3_dot(v, n)), v);
}

/* ---- never-fused forms (builders, denoiser: see the header comment) ------------ */
RT_FN float rt_v3_dot_plain(rt_v3 a, rt_v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_FN rt_v3 rt_v3_cross_plain(rt_v3 a, rt_v3 b) {
  return rt_v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_FN rt_v3 rt_v3_normalize_plain(rt_v3 v) {
  float inv = 1.0f / rt_sqrtf(rt_v3_dot_plain(v, v));
  return rt_v3_scale(v, inv);
}
RT_FN float rt_lerpf_plain(float x, float y, float t) { return x * (1.0f - t) + y * t; }

/* ---- slab arithmetic (ray_aabbs_hit_8, raytracer.c:198-210) ---------------------
 * Contract v1 and every ray that is not NaN-free:  t = (plane - o) * inv.
 * Contract v2, NaN-free rays:  t = fma(plane, inv, bias) with bias = -(o * inv), ONE instruction per plane instead
 * of two.  A ray is NaN-free ("fast") when, per axis, the reciprocal direction and the bias are finite: then the
 * exact product plane * inv is a finite real, bias is finite, and the fused result is finite or an overflowed
 * infinity -- never NaN -- and it is monotonic in `plane`, which the device code's pick-by-address relies on.
 *
 * DOMAIN of the fused form (round 5, deviation D9).  fma(plane, inv, bias) rounds twice: bias = RN(o * inv), off by at most
 * 2^-24 |o * inv|, and the result, off by at most 2^-24 |t|.  The first error does not shrink with the distance: it moves
 * the plane by up to 2^-24 |o| in SPACE, where the reference's (plane - o) * inv places a nearby plane (Sterbenz: the
 * difference of two floats within a factor of two is exact) to within 2^-23 of its DISTANCE.  scene.c pads every leaf box by
 * EPSILON = 1e-4 and the slab test clamps to EPSILON, so a grazing ray decides differently once 2^-24 |o| approaches that
 * padding (measured: primary-hit counts of the two forms differ from |o| ~ 1e5, tests/test_oracle_contracts.py).  A ray
 * therefore takes the fused form only while every origin component is below RT_SLAB_FUSED_MAX_ORIGIN = 256: placement
 * error < 2^-16 = 0.153 EPSILON.  Rays from farther out -- a camera or a scene far from the origin -- keep the reference's
 * form on the CPU and on the GPU alike (same shared predicate), at the cost of the slower exact node blocks. */
#define RT_SLAB_FUSED_MAX_ORIGIN 256.0f
RT_FN float rt_slab_bias(float o, float inv) { return -(o * inv); }
RT_FN int rt_slab_fast(float ox, float oy, float oz, float inv_x, float inv_y, float inv_z, float bias_x, float bias_y, float bias_z) {
  return (rt_absf(inv_x) < RT_INF) && (rt_absf(inv_y) < RT_INF) && (rt_absf(inv_z) < RT_INF) &&
         (rt_absf(bias_x) < RT_INF) && (rt_absf(bias_y) < RT_INF) && (rt_absf(bias_z) < RT_INF) &&
         (rt_absf(ox) < RT_SLAB_FUSED_MAX_ORIGIN) && (rt_absf(oy) < RT_SLAB_FUSED_MAX_ORIGIN) && (rt_absf(oz) < RT_SLAB_FUSED_MAX_ORIGIN);
}
RT_FN float rt_slab_t_exact(float plane, float o, float inv) { return (plane - o) * inv; }
#ifdef RT_MATH_NO_FMA
RT_FN float rt_slab_t_fast(float plane, float o, float inv, float bias) { (void)bias; return (plane - o) * inv; }
#else
RT_FN float rt_slab_t_fast(float plane, float o, float inv, float bias) { (void)o; return __builtin_fmaf(plane, inv, bias); }
#endif

/* ---- elementary functions ---------------------------------------------------- */

/* natural log, x > 0, normal numbers */
RT_FN float rt_logf(float x) {
  uint32_t ix = rt_f2u(x);
  int      e  = (int)(ix >> 23) - 126;
  float    m  = rt_u2f((ix & 0x007fffffu) | 0x3f000000u);   /* [0.5, 1) */
  if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
  float z = m * m;
  float y = 7.0376836292E-2f;
  y = rt_fmaf(y, m, -1.1514610310E-1f);
  y = rt_fmaf(y, m,  1.1676998740E-1f);
  y = rt_fmaf(y, m, -1.2420140846E-1f);
  y = rt_fmaf(y, m,  1.4249322787E-1f);
  y = rt_fmaf(y, m, -1.6668057665E-1f);
  y = rt_fmaf(y, m,  2.0000714765E-1f);
  y = rt_fmaf(y, m, -2.4999993993E-1f);
  y = rt_fmaf(y, m,  3.3333331174E-1f);
  y = y * m * z;
  float fe = (float)e;
  y = rt_fmaf(-2.12194440e-4f, fe, y);
  y = rt_fmaf(-0.5f, z, y);
  float r = m + y;
  r = rt_fmaf(0.693359375f, fe, r);
  return r;
}

/* e^x for |x| < 87 */
RT_FN float rt_expf(float x) {
  float n = rt_floorf(rt_fmaf(1.44269504088896341f, x, 0.5f));
  float r = rt_fmaf(n, -0.693359375f, x);
  r = rt_fmaf(n, 2.12194440e-4f, r);
  float z = r * r;
  float p = 1.9875691500E-4f;
  p = rt_fmaf(p, r, 1.3981999507E-3f);
  p = rt_fmaf(p, r, 8.3334519073E-3f);
  p = rt_fmaf(p, r, 4.1665795894E-2f);
  p = rt_fmaf(p, r, 1.6666665459E-1f);
  p = rt_fmaf(p, r, 5.0000001201E-1f);
  p = rt_fmaf(p, z, r) + 1.0f;
  int   in = (int)n;
  float sc = rt_u2f((uint32_t)(in + 127) << 23);
  return p * sc;
}

/* pow_f32(x, y) for the path's uses: x >= 0 (common.h:84-91).  x <= 0 -> 0. */
RT_FN float rt_powf(float x, float y) {
  if (!(x > 0.0f)) return 0.0f;
  float t = y * rt_logf(x);
  t = rt_clampf(t, -87.0f, 87.0f);
  return rt_expf(t);
}

/* sin and cos of x, 0 <= x <= ~8 (angles are rand*2pi, driver.c:119,238) */
RT_FN void rt_sincosf(float x, float *s, float *c) {
  int q = (int)(x * 1.27323954473516f);       /* x * 4/pi, truncated */
  q += q & 1;
  float y = (float)q;
  float r = rt_fmaf(y, -0.78515625f, x);
  r = rt_fmaf(y, -2.4187564849853515625e-4f, r);
  r = rt_fmaf(y, -3.77489497744594108e-8f, r);
  float z = r * r;
  float ps = rt_fmaf(rt_fmaf(-1.9515295891E-4f, z, 8.3321608736E-3f), z, -1.6666654611E-1f);
  ps = rt_fmaf(ps * z, r, r);
  float pc = rt_fmaf(rt_fmaf(2.443315711809948E-005f, z, -1.388731625493765E-003f), z, 4.166664568298827E-002f);
  pc = rt_fmaf(pc * z, z, rt_fmaf(-0.5f, z, 1.0f));
  int k = (q >> 1) & 3;                        /* x = k*pi/2 + r */
  float ss = (k & 1) ? pc : ps;
  float cc = (k & 1) ? ps : pc;
  if (k == 2 || k == 3) ss = -ss;
  if (k == 1 || k == 2) cc = -cc;
  *s = ss;
  *c = cc;
}

RT_FN float rt_atanf_pos(float a) {            /* a >= 0 */
  float yb = 0.0f;
  if (a > 2.414213562373095f) { yb = 1.5707963267948966192f; a = -(1.0f / a); }
  else if (a > 0.4142135623730950f) { yb = 0.7853981633974483096f; a = (a - 1.0f) / (a + 1.0f); }
  float z = a * a;
  float p = rt_fmaf(rt_fmaf(rt_fmaf(8.05374449538e-2f, z, -1.38776856032E-1f), z, 1.99777106478E-1f), z, -3.33329491539E-1f);
  return yb + rt_fmaf(p * z, a, a);
}

RT_FN float rt_atan2f(float y, float x) {
  if (x == 0.0f) {
    if (y > 0.0f) return 1.5707963267948966192f;
    if (y < 0.0f) return -1.5707963267948966192f;
    return 0.0f;
  }
  if (y == 0.0f) return x < 0.0f ? RT_PI : 0.0f;
  float q = y / x;
  float a = rt_atanf_pos(rt_absf(q));
  if (q < 0.0f) a = -a;
  float w = 0.0f;
  if (x < 0.0f) w = (y < 0.0f) ? -RT_PI : RT_PI;
  return w + a;
}

/* asin on [-1, 1]; arguments outside are clamped (the reference's
 * asin_f32(dir.y), driver.c:100, is NaN when |dir.y| > 1 by rounding). */
RT_FN float rt_asinf(float x) {
  float a = rt_absf(x);
  if (a > 1.0f) a = 1.0f;
  int   big = a > 0.5f;
  float z, w;
  if (big) { z = 0.5f * (1.0f - a); w = rt_sqrtf(z); } else { w = a; z = a * a; }
  float p = rt_fmaf(rt_fmaf(rt_fmaf(rt_fmaf(4.2163199048E-2f, z, 2.4181311049E-2f), z, 4.5470025998E-2f), z, 7.4953002686E-2f),
                    z, 1.6666752422E-1f);
  p = rt_fmaf(p * z, w, w);
  if (big) p = 1.5707963267948966192f - (p + p);
  return x < 0.0f ? -p : p;
}

/* b ^ 2.4 for the sRGB decode (common.h:84-91: pow_f32((x + 0.055) / 1.055, 2.4)), contract v3.
 * b = m 2^-n with m in [1, 2):  b^2.4 = P(m - 1.5) * 2^(-2.4 n),  P the degree-6 minimax polynomial of m^2.4 on [1, 2)
 * (relative fit error 2.2e-8) and 2^(-2.4 n) one of six constants.  The CORE is defined for
 * 2^-5 <= b < 2 -- every b a texture sample can produce: x in [0, 1] gives b in [0.0521, 1] -- where its result is within
 * 3.7 ulp (2.2e-7 relative) of the true power for EVERY float (all 6 x 2^23 of them compared with pow() in double: 3.63 ulp at
 * worst; tests/test_oracle_kat.py compares every fifth); exp(2.4 log b) of contract v2 was within 1.1e-6.  Outside (and for NaN) rt_pow24() is rt_powf():
 * 0 for b <= 0 or NaN as before.  7 fused multiply-adds / multiplies, 2 integer operations, 5 compare + select pairs. */
RT_FN float rt_pow24_poly(float b) {               /* P(m - 1.5), m = the mantissa of b as a float in [1, 2) */
  const float t = rt_u2f((rt_f2u(b) & 0x007fffffu) | 0x3f800000u) - 1.5f;
  float p = -1.257556141e-03f;
  p = rt_fmaf(p, t,  3.908345941e-03f);
  p = rt_fmaf(p, t, -1.751393452e-02f);
  p = rt_fmaf(p, t,  1.756089926e-01f);
  p = rt_fmaf(p, t,  1.975808740e+00f);
  p = rt_fmaf(p, t,  4.233884811e+00f);
  p = rt_fmaf(p, t,  2.646177769e+00f);
  return p;
}
/* 2^(-2.4 n), n = 0 .. 5, rounded to nearest, as a table over the three low bits of b's biased exponent (127 - n = 127 .. 122):
 * what the path kernel keeps in LDS (rt_dev.hip.h: 2 + 1 instructions instead of the 14 of the selects below) */
#define RT_POW24_SCALES {0.0f, 0.0f, 2.441406250e-04f, 1.288581989e-03f, 6.801176351e-03f, 3.589682281e-02f, 1.894645691e-01f, 1.0f}
RT_FN float rt_pow24_core(float b) {
  const float p = rt_pow24_poly(b);
  /* picked by comparisons with the binade boundaries (a chain of five selects: written over the bits of the exponent, hipcc
   * turned the select tree into divergent branches) */
  float s = 1.0f;
  s = (b < 1.0f)     ? 1.894645691e-01f : s;
  s = (b < 0.5f)     ? 3.589682281e-02f : s;
  s = (b < 0.25f)    ? 6.801176351e-03f : s;
  s = (b < 0.125f)   ? 1.288581989e-03f : s;
  s = (b < 0.0625f)  ? 2.441406250e-04f : s;
  return p * s;
}
RT_FN float rt_pow24(float b) {
  if (!(b >= 0x1p-5f && b < 2.0f)) return rt_powf(b, 2.4f);
  return rt_pow24_core(b);
}

/* ---- colour (common.h:82-92) -------------------------------------------------- */

/* NOTE: no linear toe segment, exactly like the reference. */
#if RT_MATH_POW24
RT_FN float rt_srgb_to_linear1(float x) { return rt_pow24((x + 0.055f) / 1.055f); }
#else
RT_FN float rt_srgb_to_linear1(float x) { return rt_powf((x + 0.055f) / 1.055f, 2.4f); }
#endif

RT_FN rt_v3 rt_srgb_to_linear(rt_v3 c) {
  return rt_v3_make(rt_srgb_to_linear1(c.x), rt_srgb_to_linear1(c.y), rt_srgb_to_linear1(c.z));
}

RT_FN float rt_linear_to_srgb(float c) {
  return (c <= 0.0031308f) ? (12.92f * c) : rt_madd(1.055f, rt_powf(c, 1.0f / 2.4f), -0.055f);
}

/* ---- primary-ray jitter (raytracer.c:584-594) ---------------------------------- */

RT_FN float rt_hash12(float px, float py) {
  float p3x = rt_fractf(px * 0.1031f);
  float p3y = rt_fractf(py * 0.1031f);
  float p3z = rt_fractf(px * 0.1031f);
  /* NOT fused under either contract: a hash amplifies every rounding, and the reference's AVX2 intrinsics
   * (_mm256_mul_ps / _mm256_add_ps, raytracer.c:584-594) are what tests/test_oracle_kat.py restates; once per path */
  float d = p3x * (p3y + 33.33f) + p3y * (p3z + 33.33f) + p3z * (p3x + 33.33f);
  return rt_fractf((p3x + p3y + d * 2.0f) * (p3z + d));
}

/* ---- exact radiance accumulation ------------------------------------------------
 * Per-sample radiance is summed per pixel in 64-bit fixed point (32 fractional
 * bits) so that the sum is independent of the order in which samples finish --
 * the GPU schedules paths dynamically, the oracle walks them in order, and both
 * get the same integer.  The reference sums in fp32 in sample order
 * (raytracer.c:695); the oracle can also do that (ORACLE_ACCUM_F32) and tests
 * bound the difference.  Negative and NaN samples count as 0, samples above
 * 2^20 are clamped (a pixel saturates at 1.0 long before). */
#define RT_ACCUM_FRAC_BITS 32
#define RT_ACCUM_MAX       1048576.0f

RT_FN uint64_t rt_accum_quantize(float c) {
  float v = (c > 0.0f) ? c : 0.0f;
  v = (v > RT_ACCUM_MAX) ? RT_ACCUM_MAX : v;
  return (uint64_t)((double)v * 4294967296.0);
}

/* sum of `samples` quantised values -> mean radiance, rounded once */
RT_FN float rt_accum_resolve(uint64_t sum, uint32_t samples) {
  double mean = (double)sum / ((double)samples * 4294967296.0);
  return (float)mean;
}

/* average -> clamp -> sRGB -> u8, raytracer.c:700-716 */
RT_FN uint8_t rt_encode_u8(float linear) {
  float c = rt_clampf(linear, 0.0f, 1.0f);
  c = rt_linear_to_srgb(c);
  c = c * 255.999f;
  return (uint8_t)c;
}

#endif /* RT_MATH_H */
