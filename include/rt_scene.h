/* rt_scene.h -- the scene data contract of the render hot path.
 *
 * Field-for-field layout of the structures a caller hands to
 * render_thread_proc(); each one names the reference declaration it replaces.
 * Sizes asserted at the bottom are the ones SURVEY.md section 8 quotes
 * (BVH_Node 192 B, Triangle 112 B, Triangle_AOS 112 B).
 *
 * Nothing here is GPU specific: a Scene is built on the host (scene_init,
 * rt_scene_build.c) and is flattened into HBM once per Scene* by the HIP layer
 * (rt_hip.h: rt_scene_upload).
 */
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include "rt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_BVH_WIDTH 8            /* reference raytracer.h:6  SIMD_WIDTH      */
#define RT_EPSILON   0.0001f      /* reference common.h:8     EPSILON         */
#define RT_CHUNK_SIZE 32          /* reference raytracer.c:601 CHUNK_SIZE     */

/* ---- materials are reached through a (data, proc) pair -------------------- */

/* reference scene.h:19-22 */
typedef struct {
  Vec3 direction, normal, normal_geo, tangent, bitangent, position;
  Vec2 tex_coords;
} Shader_Input;

/* reference scene.h:24-28 */
typedef struct {
  Vec3   direction;
  Color3 tint, emission;
  bool   terminate;
} Shader_Output;

/* reference scene.h:30-35 */
typedef void (*Shader_Proc)(rawptr, Shader_Input const *, Shader_Output *);

typedef struct {
  rawptr      data;
  Shader_Proc proc;
} Shader;

/* reference scene.h:65-70 */
typedef Color3 (*Background_Proc)(rawptr, Vec3 direction);

typedef struct {
  Background_Proc proc;
  rawptr          data;
} Background;

/* ---- geometry -------------------------------------------------------------- */

/* reference scene.h:10-12 */
typedef struct {
  Vec3 min, max;
} AABB;

/* reference scene.h:14-17: view_matrix maps camera space to world space
 * (rotation in rows[i][0..2], translation in rows[i][3]); focal_length is
 * 1/tan(fov/2), driver.c:765-767. */
typedef struct {
  Matrix_4x4 view_matrix;
  f32        fov, focal_length;
} Camera;

/* reference scene.h:37-42: builder input, one per triangle. */
typedef struct {
  Vec3   positions [3];
  Vec3   normals   [3];
  Vec2   tex_coords[3];
  Shader shader;
} Triangle;

typedef Slice(Triangle) Triangle_Slice;

/* reference scene.h:46-51: per-triangle shading record, read once per accepted
 * hit (raytracer.c:162-182). */
typedef struct {
  Vec3   normal, normal_a, normal_b, normal_c;
  Vec3   tangent, bitangent;
  Vec2   tex_coords_a, tex_coords_b, tex_coords_c;
  Shader shader;
} Triangle_AOS;

/* reference scene.h:53-63: nine f32 arrays of `len` entries (vertex k of
 * triangle i is x[k][i], y[k][i], z[k][i]) followed by aos[len], all in ONE
 * allocation that starts at x[0] (scene.c:84-98).  len is a multiple of 8;
 * leaf group g owns triangles [8g, 8g+8). */
typedef struct {
  f32          *x[3];
  f32          *y[3];
  f32          *z[3];
  Triangle_AOS *aos;
  i32           len;
} Triangles;

#define TRIANGLES_ALLOCATION_SIZE(N) \
  ((isize)(N) * (isize)(sizeof(f32) * 9 + sizeof(Triangle_AOS)))

/* reference scene.h:72-76 with Vec3x8 (common.h:50-52) written out: the eight
 * child boxes of one node, structure-of-arrays.  Unpopulated children are
 * all-zero boxes, which the slab test always rejects. */
typedef struct {
  f32 min_x[RT_BVH_WIDTH], min_y[RT_BVH_WIDTH], min_z[RT_BVH_WIDTH];
  f32 max_x[RT_BVH_WIDTH], max_y[RT_BVH_WIDTH], max_z[RT_BVH_WIDTH];
} BVH_Node;

typedef int BVH_Index;

/* reference scene.h:86-90: implicit complete 8-ary tree.  Node i has children
 * 8i+1 .. 8i+8; `depth` levels of internal nodes; the children of the last
 * internal level are leaf groups, child c -> triangles (c-last_row_offset)*8. */
typedef struct {
  Slice(BVH_Node) nodes;
  isize           depth;
  isize           last_row_offset;
} BVH;

/* reference scene.h:92-97 */
typedef struct {
  BVH        bvh;
  Camera     camera;
  Triangles  triangles;
  Background background;
} Scene;

/* 8^depth, reference scene.h:103-109 */
static inline isize bvh_n_leaf_nodes(isize depth) {
  isize n = 1;
  while (depth-- > 0) n *= RT_BVH_WIDTH;
  return n;
}

/* sum_{i<depth} 8^i, reference scene.h:111-119 */
static inline isize bvh_n_internal_nodes(isize depth) {
  isize level = 1, total = 0;
  while (depth-- > 0) { total += level; level *= RT_BVH_WIDTH; }
  return total;
}

/* Host-side scene construction, reference scene.h:101 / scene.c:416-426.
 * Implemented in raytracing_c_amd/csrc/rt_scene_build.c (CPU; the GPU never
 * builds).  Deviations from the reference, both defect fixes (SURVEY.md F6,H6):
 *   - a subtree that receives <= 8 triangles above the leaf row is carried down
 *     a chain of single-child nodes instead of being written at a negative
 *     triangle offset (scene.c:318-321);
 *   - depth 0 (<= 8 triangles) keeps the reference's shape (no nodes, one leaf
 *     group) and the render path tests that one group directly instead of
 *     reading nodes[0] out of bounds (raytracer.c:451). */
extern void scene_init(Scene *scene, Triangle_Slice src_triangles, Allocator allocator);

/* The allocation half of scene_init: sets bvh.depth / last_row_offset / nodes (zeroed) and the zeroed triangle block for
 * `n_triangles` input triangles.  false when the allocator fails. */
extern bool rt_scene_alloc(Scene *scene, isize n_triangles, Allocator allocator);

/* Opt-in quality builder (the reference has none): surface-area-heuristic build into the
 * SAME implicit 8-ary layout -- same depth, node format, leaf-group addressing -- so everything that consumes a
 * Scene (render_thread_proc, lightmap_bake, the .scene file, the CPU oracle) works on it unchanged.  Fewer
 * box and triangle tests per ray; the image can differ from a scene_init() scene only where two triangles
 * are hit at exactly the same distance (the tie goes to the lower slot, raytracer.c:27-29,159). */
extern void scene_init_sah(Scene *scene, Triangle_Slice src_triangles, Allocator allocator);

/* Releases what scene_init / scene_init_sah allocated with the default allocator (blocks from a caller's
 * Allocator and scenes that alias a file buffer are left alone) and drops the device copy of the scene. */
extern void rt_scene_free(Scene *scene);

/* The `.scene` cache file, reference scene.h:99-100 / scene.c:13-76: a 96-byte header
 * {i32 version, n_nodes, n_triangles, bvh_depth; Camera}, the BVH nodes, then the triangle allocation (nine f32
 * arrays + Triangle_AOS records), all raw.  scene_load_bytes has the reference's signature and semantics: the
 * scene ALIASES `data` (32-byte aligned, kept alive by the caller); false on a short / inconsistent file.
 * scene_save_bytes replaces scene_save_writer (the codin `Writer` is not part of the reference tree): it writes
 * the same bytes into a caller buffer of scene_file_size() bytes and returns the count, or -1 when too small. */
extern bool  scene_load_bytes(Byte_Slice data, Scene *scene);
extern isize scene_save_bytes(Scene const *scene, byte *dst, isize capacity);
extern isize scene_file_size(Scene const *scene);

#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
#define RT_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define RT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif
RT_STATIC_ASSERT(sizeof(BVH_Node) == 192, "BVH_Node must be 192 bytes");
RT_STATIC_ASSERT(sizeof(Triangle) == 112, "Triangle must be 112 bytes");
RT_STATIC_ASSERT(sizeof(Triangle_AOS) == 112, "Triangle_AOS must be 112 bytes");

#endif /* RT_SCENE_H */
