/* rt_scene_build.c -- host-side construction of the implicit 8-ary BVH and the
 * SoA+AoS triangle block that the render path consumes.
 *
 * Replaces reference scene.c:78-242,311-426 (scene_init and what it calls).
 * This stays on the CPU: the GPU only ever reads the finished layout
 * (SURVEY.md section 8f #2).  Same layout, same split rule, with two defect
 * fixes that include/rt_scene.h documents (early-leaf chain, depth 0) and two
 * choices the reference leaves open: the sort is a STABLE merge sort (codin
 * sort_slice_by is unspecified); the subtrees below the root are built by one
 * thread each (the reference's 12 builder threads likewise write disjoint node
 * slots, so the result does not depend on threading).
 */
#include "../../include/rt_scene.h"
#include "../../include/rt_math.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* The device layer (rt_api.cpp) keeps an HBM copy per Scene*: building into or freeing a Scene drops it.  Weak, so
 * that this file also links on its own (host-only tools, the sanitizer build of tests/c/). */
extern void rt_scene_invalidate(Scene const *scene) __attribute__((weak));

/* Blocks handed out by the DEFAULT allocator (Allocator.proc == NULL): rt_scene_free() releases exactly these.
 * Memory from a caller's Allocator, and a scene that aliases a file buffer (scene_load_bytes), is the caller's. */
static pthread_mutex_t g_owned_mutex = PTHREAD_MUTEX_INITIALIZER;
static rawptr         *g_owned = NULL;
static isize           g_owned_len = 0, g_owned_cap = 0;

static void owned_add(rawptr p) {
  pthread_mutex_lock(&g_owned_mutex);
  if (g_owned_len == g_owned_cap) {
    isize cap = g_owned_cap ? g_owned_cap * 2 : 16;
    rawptr *grown = (rawptr *)realloc(g_owned, (size_t)cap * sizeof *grown);
    if (grown) { g_owned = grown; g_owned_cap = cap; }
  }
  if (g_owned_len < g_owned_cap) g_owned[g_owned_len++] = p;     /* (on realloc failure the block is simply never freed) */
  pthread_mutex_unlock(&g_owned_mutex);
}

static bool owned_take(rawptr p) {
  bool found = false;
  pthread_mutex_lock(&g_owned_mutex);
  for (isize i = 0; i < g_owned_len; i++) {
    if (g_owned[i] == p) { g_owned[i] = g_owned[--g_owned_len]; found = true; break; }
  }
  if (g_owned_len == 0) { free(g_owned); g_owned = NULL; g_owned_cap = 0; }
  pthread_mutex_unlock(&g_owned_mutex);
  return found;
}

static rawptr rt_alloc_zeroed(Allocator a, isize size, isize align) {
  if (size <= 0) size = align;
  rawptr p;
  if (a.proc) {
    p = a.proc(a.user, size, align);
  } else {
    isize rounded = (size + align - 1) / align * align;
    p = aligned_alloc((size_t)align, (size_t)rounded);
    if (p) owned_add(p);
  }
  if (p) memset(p, 0, (size_t)size);
  return p;
}

/* scene.c:78-99: one allocation, nine coordinate arrays then the AoS records */
static bool triangles_init(Triangles *triangles, isize len, Allocator allocator) {
  while (len % RT_BVH_WIDTH) len += 1;
  triangles->len = (i32)len;
  f32 *data = (f32 *)rt_alloc_zeroed(allocator, TRIANGLES_ALLOCATION_SIZE(len), 64);
  if (!data) return false;
  for (int k = 0; k < 3; k++) {
    triangles->x[k] = data + len * (0 + k);
    triangles->y[k] = data + len * (3 + k);
    triangles->z[k] = data + len * (6 + k);
  }
  triangles->aos = (Triangle_AOS *)(data + len * 9);
  return true;
}

static rt_v3 P(Vec3 v) { return rt_v3_make(v.x, v.y, v.z); }
static Vec3 Q(rt_v3 v) { Vec3 r; r.x = v.x; r.y = v.y; r.z = v.z; return r; }

/* scene.c:105-155: copy positions to the SoA arrays and precompute the face
 * normal and the UV-aligned tangent frame of every triangle */
static void triangles_insert(Triangles *triangles, Triangle const *v, isize count, isize offset) {
  for (isize i = 0; i < count; i++) {
    Triangle const *t = &v[i];
    for (int k = 0; k < 3; k++) {
      triangles->x[k][offset + i] = t->positions[k].x;
      triangles->y[k][offset + i] = t->positions[k].y;
      triangles->z[k][offset + i] = t->positions[k].z;
    }

    rt_v3 edge1 = rt_v3_sub(P(t->positions[1]), P(t->positions[0]));
    rt_v3 edge2 = rt_v3_sub(P(t->positions[2]), P(t->positions[0]));

    f32 du1 = t->tex_coords[1].x - t->tex_coords[0].x, dv1 = t->tex_coords[1].y - t->tex_coords[0].y;
    f32 du2 = t->tex_coords[2].x - t->tex_coords[0].x, dv2 = t->tex_coords[2].y - t->tex_coords[0].y;

    f32 d = du1 * dv2 - du2 * dv1;
    if (rt_absf(d) < 0.0001f) d = (d < 0) ? -0.0001f : 0.0001f;
    f32 inv_d = 1.0f / d;

    rt_v3 tangent   = rt_v3_normalize_plain(rt_v3_scale(rt_v3_sub(rt_v3_scale(edge1, dv2), rt_v3_scale(edge2, dv1)), inv_d));
    rt_v3 bitangent = rt_v3_normalize_plain(rt_v3_scale(rt_v3_sub(rt_v3_scale(edge2, du1), rt_v3_scale(edge1, du2)), inv_d));

    Triangle_AOS *aos = &triangles->aos[offset + i];
    aos->shader       = t->shader;
    aos->normal       = Q(rt_v3_normalize_plain(rt_v3_cross_plain(edge1, edge2)));
    aos->normal_a     = t->normals[0];
    aos->normal_b     = t->normals[1];
    aos->normal_c     = t->normals[2];
    aos->tex_coords_a = t->tex_coords[0];
    aos->tex_coords_b = t->tex_coords[1];
    aos->tex_coords_c = t->tex_coords[2];
    aos->tangent      = Q(tangent);
    aos->bitangent    = Q(bitangent);
  }
}

/* scene.c:157-201 */
static f32 aabb_surface_area(AABB const *aabb) {
  f32 x = aabb->max.x - aabb->min.x;
  f32 y = aabb->max.y - aabb->min.y;
  f32 z = aabb->max.z - aabb->min.z;
  return 2.0f * (x * y + y * z + z * x);
}

static f32 min3f(f32 a, f32 b, f32 c) { f32 m = b < c ? b : c; return a < m ? a : m; }
static f32 max3f(f32 a, f32 b, f32 c) { f32 m = b > c ? b : c; return a > m ? a : m; }

static void aabb_triangle(Triangle const *t, AABB *aabb) {
  for (int ax = 0; ax < 3; ax++) {
    aabb->min.data[ax] = min3f(t->positions[0].data[ax], t->positions[1].data[ax], t->positions[2].data[ax]) - RT_EPSILON;
    aabb->max.data[ax] = max3f(t->positions[0].data[ax], t->positions[1].data[ax], t->positions[2].data[ax]) + RT_EPSILON;
  }
}

static void aabb_triangle_slice(Triangle const *tris, isize count, AABB *aabb) {
  memset(aabb, 0, sizeof *aabb);
  for (isize i = 0; i < count; i++) {
    AABB t;
    aabb_triangle(&tris[i], &t);
    if (i == 0) *aabb = t;
    for (int ax = 0; ax < 3; ax++) {
      if (t.min.data[ax] < aabb->min.data[ax]) aabb->min.data[ax] = t.min.data[ax];
      if (t.max.data[ax] > aabb->max.data[ax]) aabb->max.data[ax] = t.max.data[ax];
    }
  }
}

/* scene.c:203-222: ascending by the sum of the three vertex coordinates on
 * `axis`; stable merge sort on (key, position) */
typedef struct { f32 key; i32 idx; } Sort_Key;

static void merge_sort_keys(Sort_Key *a, Sort_Key *tmp, isize n) {
  if (n < 2) return;
  isize h = n / 2;
  merge_sort_keys(a, tmp, h);
  merge_sort_keys(a + h, tmp, n - h);
  isize i = 0, j = h, k = 0;
  while (i < h && j < n) tmp[k++] = (a[j].key < a[i].key) ? a[j++] : a[i++];
  while (i < h) tmp[k++] = a[i++];
  while (j < n) tmp[k++] = a[j++];
  memcpy(a, tmp, (size_t)n * sizeof *a);
}

typedef struct {
  Sort_Key *keys, *tmp;
  Triangle *scratch;
} Sort_Buffers;

static void sort_triangle_slice(Triangle *tris, isize count, int axis, Sort_Buffers *sb) {
  for (isize i = 0; i < count; i++) {
    sb->keys[i].key = tris[i].positions[0].data[axis] + tris[i].positions[1].data[axis] + tris[i].positions[2].data[axis];
    sb->keys[i].idx = (i32)i;
  }
  merge_sort_keys(sb->keys, sb->tmp, count);
  for (isize i = 0; i < count; i++) sb->scratch[i] = tris[sb->keys[i].idx];
  memcpy(tris, sb->scratch, (size_t)count * sizeof *tris);
}

/* scene.c:224-242 */
static isize bvh_required_depth(isize n_triangles) {
  n_triangles = (n_triangles + RT_BVH_WIDTH - 1) / RT_BVH_WIDTH;
  isize n = 1, i = 0;
  while (n < n_triangles) { n *= RT_BVH_WIDTH; i += 1; }
  return i;
}

static isize bvh_partition_triangles(isize n_triangles, isize per_child) {
  isize n = 0, left = n_triangles;
  while (n < n_triangles / 2 && left > per_child) { n += per_child; left -= per_child; }
  return n;
}

typedef struct { Triangle *data; isize len; } Tri_Span;

typedef struct {
  Scene       *scene;
  Triangle    *tris;
  isize        count, depth;
  BVH_Index    index;
  Sort_Buffers sb;
} Build_Job;
static void *build_job_run(void *arg);

/* scene.c:311-414: fixed-capacity split of `tris` into at most 8 children of
 * capacity 8^depth triangles each; `depth` = internal levels at and below
 * `index` (0 = `index` is a leaf group) */
static void bvh_build(Scene *scene, Triangle *tris, isize count, isize depth, BVH_Index index, Sort_Buffers *sb) {
  if (count <= RT_BVH_WIDTH) {
    if (depth == 0) {
      triangles_insert(&scene->triangles, tris, count, ((isize)index - scene->bvh.last_row_offset) * RT_BVH_WIDTH);
      return;
    }
    /* early-leaf fix: hand the small set down through child 0 (reference
     * scene.c:318-321 would write at a negative triangle offset here) */
    if (count == 0) return;
    AABB aabb;
    aabb_triangle_slice(tris, count, &aabb);
    BVH_Node *node = &scene->bvh.nodes.data[index];
    node->min_x[0] = aabb.min.x; node->min_y[0] = aabb.min.y; node->min_z[0] = aabb.min.z;
    node->max_x[0] = aabb.max.x; node->max_y[0] = aabb.max.y; node->max_z[0] = aabb.max.z;
    bvh_build(scene, tris, count, depth - 1, index * RT_BVH_WIDTH + 1, sb);
    return;
  }

  isize per_child = bvh_n_leaf_nodes(depth);

  Tri_Span slices[RT_BVH_WIDTH];
  Tri_Span finished[RT_BVH_WIDTH];
  isize n_slices = 1, n_finished = 0;
  slices[0].data = tris;
  slices[0].len  = count;

  while (n_slices != 0) {
    n_slices -= 1;
    Tri_Span slice = slices[n_slices];
    isize split = bvh_partition_triangles(slice.len, per_child);
    Tri_Span left  = { slice.data, split };
    Tri_Span right = { slice.data + split, slice.len - split };

    f32 min_surface_area = RT_INF;
    int best_axis = 0;
    for (int axis = 0; axis < 3; axis++) {
      sort_triangle_slice(slice.data, slice.len, axis, sb);
      AABB a, b;
      aabb_triangle_slice(left.data, left.len, &a);
      aabb_triangle_slice(right.data, right.len, &b);
      f32 surface_area = aabb_surface_area(&a) + aabb_surface_area(&b);
      if (surface_area <= min_surface_area) {
        min_surface_area = surface_area;
        best_axis = axis;
      }
    }
    if (best_axis != 2) sort_triangle_slice(slice.data, slice.len, best_axis, sb);

    if (left.len > per_child) slices[n_slices++] = left;
    else if (left.len)        finished[n_finished++] = left;
    if (right.len > per_child) slices[n_slices++] = right;
    else if (right.len)        finished[n_finished++] = right;
  }

  BVH_Node node;
  memset(&node, 0, sizeof node);
  for (isize i = 0; i < n_finished; i++) {
    AABB aabb;
    aabb_triangle_slice(finished[i].data, finished[i].len, &aabb);
    node.min_x[i] = aabb.min.x; node.min_y[i] = aabb.min.y; node.min_z[i] = aabb.min.z;
    node.max_x[i] = aabb.max.x; node.max_y[i] = aabb.max.y; node.max_z[i] = aabb.max.z;
  }
  scene->bvh.nodes.data[index] = node;

  /* The children own disjoint triangle ranges, node slots and leaf groups, so below the root they are built by one
   * thread each (the reference runs 12 builder threads, scene.c:244-309); the result does not depend on threading. */
  if (index == 0 && depth >= 2 && count >= 2048) {
    pthread_t th[RT_BVH_WIDTH];
    Build_Job jobs[RT_BVH_WIDTH];
    bool      started[RT_BVH_WIDTH];
    for (isize i = 0; i < n_finished; i++) {
      Build_Job *j = &jobs[i];
      j->scene = scene; j->tris = finished[i].data; j->count = finished[i].len; j->depth = depth - 1;
      j->index = index * RT_BVH_WIDTH + 1 + (BVH_Index)i;
      j->sb.keys    = (Sort_Key *)malloc((size_t)j->count * sizeof(Sort_Key));
      j->sb.tmp     = (Sort_Key *)malloc((size_t)j->count * sizeof(Sort_Key));
      j->sb.scratch = (Triangle *)malloc((size_t)j->count * sizeof(Triangle));
      started[i] = j->sb.keys && j->sb.tmp && j->sb.scratch && pthread_create(&th[i], NULL, build_job_run, j) == 0;
      if (!started[i]) bvh_build(scene, j->tris, j->count, j->depth, j->index, sb);      /* fall back to this thread */
    }
    for (isize i = 0; i < n_finished; i++) {
      if (started[i]) pthread_join(th[i], NULL);
      free(jobs[i].sb.keys); free(jobs[i].sb.tmp); free(jobs[i].sb.scratch);
    }
    return;
  }
  for (isize i = 0; i < n_finished; i++)
    bvh_build(scene, finished[i].data, finished[i].len, depth - 1, index * RT_BVH_WIDTH + 1 + (BVH_Index)i, sb);
}

static void *build_job_run(void *arg) {
  Build_Job *j = (Build_Job *)arg;
  bvh_build(j->scene, j->tris, j->count, j->depth, j->index, &j->sb);
  return NULL;
}

/* scene.c:416-426.  Sorts a private copy of the input (the reference sorts the
 * caller's slice in place). */
/* The allocation half of scene_init (scene.c:416-424): depth and node count from the triangle count, zeroed node array,
 * zeroed SoA + AoS triangle block.  false (and an empty scene, which the upload refuses) when the allocator fails.
 * Shared by scene_init, scene_init_sah and the GPU builder scene_init_gpu (rt_api.cpp). */
bool rt_scene_alloc(Scene *scene, isize n_triangles, Allocator allocator) {
  if (rt_scene_invalidate) rt_scene_invalidate(scene);      /* a device copy of what this Scene held before is stale */
  isize depth      = bvh_required_depth(n_triangles);
  isize n_internal = bvh_n_internal_nodes(depth);
  scene->bvh.depth           = depth;
  scene->bvh.last_row_offset = n_internal;
  scene->bvh.nodes.len       = n_internal;
  scene->bvh.nodes.data      = (BVH_Node *)rt_alloc_zeroed(allocator, n_internal * (isize)sizeof(BVH_Node), 64);
  memset(&scene->triangles, 0, sizeof scene->triangles);
  if (!scene->bvh.nodes.data) {          /* allocation failed: an empty scene (len 0), which the upload refuses */
    scene->bvh.nodes.len = 0;
    return false;
  }
  if (!triangles_init(&scene->triangles, bvh_n_leaf_nodes(depth) * RT_BVH_WIDTH, allocator)) {
    memset(&scene->triangles, 0, sizeof scene->triangles);
    return false;
  }
  return true;
}

void scene_init(Scene *scene, Triangle_Slice src, Allocator allocator) {
  if (!rt_scene_alloc(scene, src.len, allocator)) return;
  isize depth = scene->bvh.depth;
  if (src.len <= 0) return;

  isize n = src.len;
  Triangle *work = (Triangle *)malloc((size_t)n * sizeof *work);
  Sort_Buffers sb;
  sb.keys    = (Sort_Key *)malloc((size_t)n * sizeof *sb.keys);
  sb.tmp     = (Sort_Key *)malloc((size_t)n * sizeof *sb.tmp);
  sb.scratch = (Triangle *)malloc((size_t)n * sizeof *sb.scratch);
  if (work && sb.keys && sb.tmp && sb.scratch) {
    memcpy(work, src.data, (size_t)n * sizeof *work);
    bvh_build(scene, work, n, depth, 0, &sb);
  }
  free(work); free(sb.keys); free(sb.tmp); free(sb.scratch);
}

/* ---------------------------------------------------------------------------------------------------
 * scene_init_sah() -- opt-in quality builder: a surface-area-heuristic build that
 * emits the SAME layout as scene_init() (implicit complete 8-ary tree of bvh_required_depth(n) levels,
 * BVH_Node boxes, leaf groups of 8 triangles in the last row, unused children all-zero), so the render path
 * -- CPU oracle and GPU kernels alike -- traverses it unchanged.  The reference's split (scene.c:311-414)
 * packs every leaf group full and cuts each node's triangles at multiples of the child capacity, wherever
 * that falls in space; half of the leaf slots of the complete tree stay empty (helmet: 1 932 of 4 096).
 * This builder spends that slack on tighter boxes:
 *   - a node's triangles are cut into up to 8 children by repeated binary splits; the cut of a slice is the
 *     position (sweep over the centroid order of each axis) that minimises
 *         area(left) * (n_left + 8) + area(right) * (n_right + 8)
 *     -- the triangles below a child plus the 8-wide test the child itself costs, whether it holds 1 or 8
 *     triangles (raytracer.c:84-188 always tests 8);
 *   - a cut is only taken if every child still fits its subtree (8^depth triangles) and the node keeps at
 *     most 8 children: sum over slices of ceil(len / capacity) <= 8 is the invariant;
 *   - the slice with the largest area * (len + 8) is cut next; slices over capacity are cut first; a cut that
 *     does not lower the cost is not taken.
 * Deterministic (stable sorts, no threads). */
typedef struct { Triangle *data; isize len; AABB box; f32 area; bool final; } Sah_Slice;

typedef struct {
  Sort_Buffers sb;
  AABB        *suffix;      /* suffix[i] = box of triangles [i, len) of the slice being swept */
} Sah_Buffers;

static void aabb_grow(AABB *a, AABB const *t) {
  for (int ax = 0; ax < 3; ax++) {
    if (t->min.data[ax] < a->min.data[ax]) a->min.data[ax] = t->min.data[ax];
    if (t->max.data[ax] > a->max.data[ax]) a->max.data[ax] = t->max.data[ax];
  }
}

static isize ceil_div(isize a, isize b) { return (a + b - 1) / b; }
/* cost weight of a child holding n triangles: the triangles below it plus one 8-wide test of its own.  Measured
 * against the alternatives on the three asset scenes with the oracle's counters (node / leaf visits per ray, helmet):
 * ceil(n/8) 3.276 / 0.992, n 3.215 / 1.256, n + 8 3.205 / 0.962 (reference split: 3.378 / 1.232). */
static f32 sah_w(isize n) { return (f32)n + (f32)RT_BVH_WIDTH; }

/* best cut of `s` on the axis it is currently sorted by; returns the cost, writes the position */
static f32 sah_sweep(Sah_Slice const *s, isize capacity, isize other_slots, Sah_Buffers *sb, isize *pos_out) {
  isize n = s->len;
  AABB run;
  for (isize i = n - 1; i >= 0; i--) {
    AABB t;
    aabb_triangle(&s->data[i], &t);
    if (i == n - 1) run = t; else aabb_grow(&run, &t);
    sb->suffix[i] = run;
  }
  f32 best = RT_INF;
  *pos_out = -1;
  for (isize i = 1; i < n; i++) {
    AABB t;
    aabb_triangle(&s->data[i - 1], &t);
    if (i == 1) run = t; else aabb_grow(&run, &t);
    if (other_slots + ceil_div(i, capacity) + ceil_div(n - i, capacity) > RT_BVH_WIDTH) continue;
    f32 cost = aabb_surface_area(&run) * sah_w(i) +
               aabb_surface_area(&sb->suffix[i]) * sah_w(n - i);
    if (cost < best) { best = cost; *pos_out = i; }
  }
  return best;
}

static void sah_build(Scene *scene, Triangle *tris, isize count, isize depth, BVH_Index index, Sah_Buffers *sb) {
  if (count == 0) return;
  if (depth == 0) {
    triangles_insert(&scene->triangles, tris, count, ((isize)index - scene->bvh.last_row_offset) * RT_BVH_WIDTH);
    return;
  }
  isize capacity = bvh_n_leaf_nodes(depth);          /* triangles one child subtree can hold */

  Sah_Slice slices[RT_BVH_WIDTH];
  isize n_slices = 1;
  slices[0].data = tris;
  slices[0].len = count;
  slices[0].final = false;
  aabb_triangle_slice(tris, count, &slices[0].box);
  slices[0].area = aabb_surface_area(&slices[0].box);

  for (;;) {
    /* next slice to cut: over-capacity ones first (they must be cut), then the most expensive one */
    isize pick = -1;
    f32 pick_key = -1.0f;
    bool forced = false;
    for (isize i = 0; i < n_slices; i++) {
      Sah_Slice *s = &slices[i];
      if (s->len > capacity) {
        if (!forced || s->len > slices[pick].len) { pick = i; forced = true; }
      } else if (!forced && !s->final && s->len >= 2 && n_slices < RT_BVH_WIDTH) {
        f32 key = s->area * sah_w(s->len);
        if (key > pick_key) { pick_key = key; pick = i; }
      }
    }
    if (pick < 0) break;
    Sah_Slice *s = &slices[pick];
    isize other_slots = 0;
    for (isize i = 0; i < n_slices; i++) if (i != pick) other_slots += ceil_div(slices[i].len, capacity);

    f32 best_cost = RT_INF;
    int best_axis = -1;
    isize best_pos = -1;
    for (int axis = 0; axis < 3; axis++) {
      sort_triangle_slice(s->data, s->len, axis, &sb->sb);
      isize pos;
      f32 cost = sah_sweep(s, capacity, other_slots, sb, &pos);
      if (pos > 0 && cost < best_cost) { best_cost = cost; best_axis = axis; best_pos = pos; }
    }
    f32 whole = s->area * sah_w(s->len);
    if (forced && best_axis < 0) {
      /* A slice that MUST be cut has a feasible position (the invariant sum ceil(len / capacity) <= 8 holds), but the sweep
       * accepts only finite costs: a vertex at +-inf, or coordinates around 1e20 whose surface area overflows, leave every
       * candidate at inf or NaN.  Cut by count then, as the reference's split does (scene.c:333-380): a multiple of the child
       * capacity keeps the slot budget, and no triangle is dropped. */
      best_axis = 2;                                   /* (the order the last sort left) */
      best_pos = capacity * (ceil_div(s->len, capacity) / 2);
    } else if (best_axis < 0 || (!forced && !(best_cost < whole))) {
      s->final = true;
      continue;
    }
    if (best_axis != 2) sort_triangle_slice(s->data, s->len, best_axis, &sb->sb);
    Sah_Slice left, right;
    left.data = s->data;             left.len = best_pos;           left.final = false;
    right.data = s->data + best_pos; right.len = s->len - best_pos; right.final = false;
    aabb_triangle_slice(left.data, left.len, &left.box);
    aabb_triangle_slice(right.data, right.len, &right.box);
    left.area = aabb_surface_area(&left.box);
    right.area = aabb_surface_area(&right.box);
    slices[pick] = left;
    slices[n_slices++] = right;
  }

  BVH_Node node;
  memset(&node, 0, sizeof node);
  for (isize i = 0; i < n_slices; i++) {
    node.min_x[i] = slices[i].box.min.x; node.min_y[i] = slices[i].box.min.y; node.min_z[i] = slices[i].box.min.z;
    node.max_x[i] = slices[i].box.max.x; node.max_y[i] = slices[i].box.max.y; node.max_z[i] = slices[i].box.max.z;
  }
  scene->bvh.nodes.data[index] = node;
  for (isize i = 0; i < n_slices; i++)
    sah_build(scene, slices[i].data, slices[i].len, depth - 1, index * RT_BVH_WIDTH + 1 + (BVH_Index)i, sb);
}

void scene_init_sah(Scene *scene, Triangle_Slice src, Allocator allocator) {
  if (!rt_scene_alloc(scene, src.len, allocator)) return;
  isize depth = scene->bvh.depth;
  if (src.len <= 0) return;

  isize n = src.len;
  Triangle *work = (Triangle *)malloc((size_t)n * sizeof *work);
  Sah_Buffers sb;
  sb.sb.keys    = (Sort_Key *)malloc((size_t)n * sizeof *sb.sb.keys);
  sb.sb.tmp     = (Sort_Key *)malloc((size_t)n * sizeof *sb.sb.tmp);
  sb.sb.scratch = (Triangle *)malloc((size_t)n * sizeof *sb.sb.scratch);
  sb.suffix     = (AABB *)malloc((size_t)n * sizeof *sb.suffix);
  if (work && sb.sb.keys && sb.sb.tmp && sb.sb.scratch && sb.suffix) {
    memcpy(work, src.data, (size_t)n * sizeof *work);
    sah_build(scene, work, n, depth, 0, &sb);
  }
  free(work); free(sb.sb.keys); free(sb.sb.tmp); free(sb.sb.scratch); free(sb.suffix);
}

/* Releases what scene_init() allocated with the DEFAULT allocator and drops the device copy.  Blocks that came from
 * a caller's Allocator, or that alias a file buffer (scene_load_bytes), are not touched: they are the caller's. */
void rt_scene_free(Scene *scene) {
  if (!scene) return;
  if (rt_scene_invalidate) rt_scene_invalidate(scene);
  if (scene->bvh.nodes.data && owned_take(scene->bvh.nodes.data)) free(scene->bvh.nodes.data);
  if (scene->triangles.x[0] && owned_take(scene->triangles.x[0])) free(scene->triangles.x[0]);
  scene->bvh.nodes.data = NULL;
  scene->bvh.nodes.len  = 0;
  memset(&scene->triangles, 0, sizeof scene->triangles);
}

/* ---- the `.scene` cache file, reference scene.c:13-76 ---------------------------------------------
 * Byte layout (little endian, the reference writes its structs raw):
 *   header  { i32 version = 0, n_nodes, n_triangles, bvh_depth; Camera camera; } padded to 32-byte alignment (96 B)
 *   nodes   n_nodes x BVH_Node (192 B)
 *   tris    9 x n_triangles f32 (x[0..2], y[0..2], z[0..2]) then n_triangles x Triangle_AOS (112 B)
 * The Shader pointers inside Triangle_AOS are the writer's host addresses, as in the reference: a loaded scene
 * is good for traversal and for the BVH visualiser, and needs its shaders re-bound before it is rendered. */
typedef struct __attribute__((aligned(32))) {
  i32    version, n_nodes, n_triangles, bvh_depth;
  Camera camera;
} Scene_File_Header;

isize scene_file_size(Scene const *scene) {
  return (isize)sizeof(Scene_File_Header) + scene->bvh.nodes.len * (isize)sizeof(BVH_Node) +
         TRIANGLES_ALLOCATION_SIZE(scene->triangles.len);
}

/* scene.c:18-34 with the codin Writer replaced by a caller buffer: returns the bytes written, or -1 when
 * `capacity` is too small (nothing is written then). */
isize scene_save_bytes(Scene const *scene, byte *dst, isize capacity) {
  isize total = scene_file_size(scene);
  if (!dst || capacity < total) return -1;
  Scene_File_Header header;
  memset(&header, 0, sizeof header);
  header.version     = 0;
  header.n_nodes     = (i32)scene->bvh.nodes.len;
  header.n_triangles = scene->triangles.len;
  header.bvh_depth   = (i32)scene->bvh.depth;
  header.camera      = scene->camera;
  memcpy(dst, &header, sizeof header);
  dst += sizeof header;
  isize node_bytes = scene->bvh.nodes.len * (isize)sizeof(BVH_Node);
  if (node_bytes > 0) memcpy(dst, scene->bvh.nodes.data, (size_t)node_bytes);
  dst += node_bytes;
  isize tri_bytes = TRIANGLES_ALLOCATION_SIZE(scene->triangles.len);
  if (tri_bytes > 0) memcpy(dst, scene->triangles.x[0], (size_t)tri_bytes);
  return total;
}

/* scene.c:36-76.  The scene ALIASES `data` (no copy), so `data` must stay alive and be 32-byte aligned; returns
 * false on a short or inconsistent file (the reference asserts on misalignment, this returns false).  Sets
 * last_row_offset, which the reference's loader leaves unset (scene.c:420 sets it only in scene_init). */
bool scene_load_bytes(Byte_Slice data, Scene *scene) {
  Scene_File_Header header;
  if (!data.data || !scene || ((uintptr_t)data.data % 32) != 0) return false;
  if (data.len < (isize)sizeof header) return false;
  memcpy(&header, data.data, sizeof header);
  if (header.version != 0 || header.n_nodes < 0 || header.n_triangles < 0 || header.bvh_depth < 0) return false;
  if (data.len != (isize)sizeof header + (isize)header.n_nodes * (isize)sizeof(BVH_Node) +
                      TRIANGLES_ALLOCATION_SIZE(header.n_triangles))
    return false;
  if (header.bvh_depth > 10 || header.n_nodes != bvh_n_internal_nodes(header.bvh_depth)) return false;
  if ((header.n_triangles % RT_BVH_WIDTH) != 0) return false;

  scene->camera               = header.camera;
  scene->bvh.depth            = header.bvh_depth;
  scene->bvh.last_row_offset  = header.n_nodes;
  scene->bvh.nodes.len        = header.n_nodes;
  scene->bvh.nodes.data       = (BVH_Node *)(data.data + sizeof header);
  f32 *tris = (f32 *)(data.data + sizeof header + (size_t)header.n_nodes * sizeof(BVH_Node));
  isize n = header.n_triangles;
  for (int k = 0; k < 3; k++) {
    scene->triangles.x[k] = tris + n * (0 + k);
    scene->triangles.y[k] = tris + n * (3 + k);
    scene->triangles.z[k] = tris + n * (6 + k);
  }
  scene->triangles.aos = (Triangle_AOS *)(tris + n * 9);
  scene->triangles.len = header.n_triangles;
  return true;
}
