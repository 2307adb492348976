/* rt_device.h -- layout of the flattened scene in HBM and the kernel argument
 * block.  Shared by rt_api.cpp (host, fills it) and rt_kernels.hip (reads it).
 *
 * HBM layout (all read-only during a frame, 16-byte aligned):
 *   nodes   : BVH_Node as is, 12 float4 per node                 192 B / node
 *   leaves  : per leaf group g, 9 rows of 8 f32 (ax e1x e2x ay e1y e2y az e1z e2z,
 *             e1 = b-a, e2 = c-a): the reference's nine SoA arrays (scene.h:53-63)
 *             re-tiled so one group is 288 contiguous bytes, with the two edge
 *             subtractions of raytracer.c:115-122 done at upload  288 B / leaf
 *   tris    : Triangle_AOS without the Shader pair, plus a material id
 *                                                                 112 B / tri
 *   mats    : PBR_Shader_Data with texture pointers replaced by indices
 *                                                                  80 B / mat
 *   textures: descriptor table + one RGBA8 texel pool
 */
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <stdint.h>

#define RT_NODE_F4   12      /* float4 per node           */
#define RT_LEAF_F4   18      /* float4 per leaf group     */
#define RT_TRI_F4     7      /* float4 per triangle record*/
#define RT_MAT_F4     9      /* float4 per material       */
#define RT_MAT_FLOATS 36
#define RT_MAX_DEPTH  8      /* perm-stack levels in LDS  */

#define RT_MAT_DISNEY 0
#define RT_MAT_DEBUG  1

#define RT_TILE      8       /* work item = 8x8 pixel tile x sample slab */
#define RT_TILE_PIX  64

#define RT_PARK_RECORD_DWORDS (18 * 128)   /* a wave's slice of RT_KParams.park: RT_PARK_FIELDS x RT_PARK_CAP (rt_dev.hip.h) */
#define RT_N_COUNTERS 136    /* 0..6 ray counters; 8..23 block statistics of the diagnostic kernel, 24..39 its cycle sums; or 8..135 the
                              * block ledger of the tile-stream kernel (-DRT_LEDGER builds, LG_* slots in rt_dev.hip.h) */

/* triangle record, 28 floats:
 *  [0..2] face normal      [3]  material id (int bits)
 *  [4..6] normal_a         [7]  uv_a.x
 *  [8..10] normal_b        [11] uv_a.y
 *  [12..14] normal_c       [15] uv_b.x
 *  [16..18] tangent        [19] uv_b.y
 *  [20..22] bitangent      [23] uv_c.x
 *  [24] uv_c.y             [25..27] pad
 */

/* material record, 20 floats:
 *  [0..2] base_color  [3] roughness
 *  [4..6] emission    [7] metalness
 *  [8] normal_map_strength [9] sheen [10] sheen_tint [11] anisotropic_strength
 *  [12] tex_albedo [13] tex_normal [14] tex_metal_roughness [15] tex_emission (int bits, -1 none)
 *  [16] kind (int bits)  [17..19] pad
 *  [20..35] the descriptors of the four textures, (offset, width, height, stride) as int bits each, in the order albedo,
 *           normal, metal_roughness, emission: a copy of textures[tex] so that a shade block needs no third dependent
 *           load between the material record and the texels (zeros for an absent texture)
 */

/* Texel layout in the pool.  RT_TEX_TILED = 1: 4 x 4-texel tiles of 64 bytes (one cache line), tiles row-major,
 * `stride` = tiles per row: the 2 x 2 footprint of a bilinear fetch lies in ONE line 9 times in 16 instead of never (two
 * rows of a row-major image are two lines).  0: row-major, `stride` = texels per row.  Addressing only: same texels. */
#ifndef RT_TEX_TILED
#define RT_TEX_TILED 1
#endif
typedef struct {
  uint32_t offset;    /* first texel in the pool */
  int32_t  width, height, stride;
} RT_DTexture;
#define RT_TEX_TILE_INDEX(x, y, tpr) ((((y) >> 2) * (tpr) + ((x) >> 2)) * 16 + (((y) & 3) << 2) + ((x) & 3))

typedef struct {
  /* scene */
  const float    *nodes;
  const float    *leaves;
  const float    *tris;
  const float    *mats;
  const RT_DTexture *textures;
  const uint32_t *texels;      /* RGBA8 */
  int32_t depth;               /* bvh.depth                */
  int32_t last_row_offset;     /* bvh.last_row_offset      */
  int32_t bg_texture;          /* index into textures      */
  int32_t n_nodes;
  /* camera: rows 0..2 of view_matrix (rotation | translation), focal length */
  float cam[3][4];
  float focal_length;
  float inv_width, inv_height, aspect;   /* 1/width, 1/height, width/height in fp32 (raytracer.c:615-617) */
  /* frame */
  int32_t width, height, samples, max_bounces;
  int32_t sample_first, sample_end;   /* this launch traces samples [first, end) of every pixel */
  uint32_t seed;
  int32_t chunks_x, n_chunks;
  int32_t rank, world, n_local_chunks;
  int32_t slab_shift;          /* samples per work item = 1 << slab_shift */
  int32_t n_slabs;             /* ceil((sample_end - sample_first) / slab) */
  int32_t n_work;              /* n_local_chunks * 16 * n_slabs           */
  int32_t sched_thresh;        /* lanes waiting for shade / environment / regeneration that trigger that block */
  int32_t n_lds_nodes;         /* BVH nodes [0, n) are also in the workgroup's LDS   */
  /* outputs */
  unsigned long long *accum;   /* [height*width*3] 32.32 fixed point      */
  unsigned long long *counters;/* RT_N_COUNTERS                           */
  uint32_t *work_head;         /* dequeue counter                         */
  const int32_t *local_chunks; /* global chunk index of this rank's l-th chunk (partition table) */
  const uint32_t *order;       /* tile visiting order (NULL = identity)   */
  uint32_t *tile_cost;         /* rays per tile of THIS launch (NULL = off)*/
  unsigned long long *wave_times; /* diagnostic kernel: per wave start, end (100 MHz), items */
  /* tile-stream kernel (variant 5): a tile hands out units = 2 pixels x (1 << chunk_shift) samples */
  uint32_t *tile_next;         /* [n_tiles] chunks handed out so far (zero at launch)                       */
  uint32_t *open_groups;       /* [ceil(n_tiles / 64)] tiles of the group that still have chunks            */
  int32_t n_tiles;             /* n_local_chunks * 16                                                       */
  int32_t chunk_shift;         /* log2 samples per chunk                                                    */
  int32_t n_chunks_tile;       /* units per tile: 8 rows x n_sample_blocks x 4 pixel pairs                  */
  int32_t n_sample_blocks;     /* ceil(samples of this launch / samples per unit)                           */
  int32_t drain_thresh;        /* lanes waiting for S that trigger it once the wave's tile is exhausted     */
  int32_t grab_max;            /* units a wave takes per atomic while its tile has plenty left (1, 2 or 4)   */
  uint32_t *park;              /* tile-stream kernel: [waves][18][128] dwords, hits parked until a dense shade block (nullptr = off) */
  int32_t short_div;           /* 1: leaf blocks take 1 / det from rcp_exact() (host-checked determinant bound), 0: IEEE division */
  int32_t pyr_nodes;           /* node blocks of camera rays on nodes [0, n) test only the children the tile's pyramid can touch; 0 = off */
  /* wavefront pipeline (rt_wavefront.hip): camera / trace / shade kernels joined by record queues in HBM.  A queue is
   * an array of chunks of WF_CHUNK records, struct-of-arrays inside a chunk ([field][WF_CHUNK] dwords), plus the number
   * of records in every chunk; chunks are handed to producing waves by an atomic counter (wf_ctl). */
  uint32_t *wf_hit0;           /* camera-ray hits: WF_HIT0_FIELDS dwords per record                                     */
  uint32_t *wf_hit;            /* hits of continuation rays: WF_HIT_FIELDS                                              */
  uint32_t *wf_ray[2];         /* continuation rays, written by the shade kernel of bounce b into [b & 1]: WF_RAY_FIELDS */
  uint32_t *wf_cnt_hit0, *wf_cnt_hit, *wf_cnt_ray[2];   /* records per chunk                                            */
  uint32_t *wf_ctl;            /* WF_CTL_* words, WF_CTL_STRIDE dwords apart                                            */
  int32_t wf_soft_chunks;      /* camera kernel: a wave that is handed chunk >= this of wf_hit0 stops taking units      */
  int32_t wf_bounce;           /* shade / trace kernels: bounce index of the records they read                          */
  int32_t wf_n_waves;          /* waves of this launch (the last one to finish resets the control words it consumed)    */
} RT_KParams;

#define WF_CHUNK        256
#define WF_HIT0_FIELDS  9      /* direction (3), t, triangle, u, v, pixel (y << 16 | x), sample                         */
#define WF_HIT_FIELDS   5      /* index of the ray record (chunk * WF_CHUNK + slot), t, triangle, u, v                  */
#define WF_RAY_FIELDS   15     /* origin (3), direction (3), tint (3), emission (3), rng, pixel (y << 16 | x), bounce   */
#define WF_CTL_STRIDE   16     /* one control word per 64-byte line                                                     */
enum {
  WF_HIT0_ALLOC = 0, WF_HIT0_HEAD, WF_HIT_ALLOC, WF_HIT_HEAD, WF_RAY0_ALLOC, WF_RAY0_HEAD, WF_RAY1_ALLOC, WF_RAY1_HEAD,
  WF_STOPPED,      /* set by a camera-kernel wave that ran into wf_soft_chunks: units are left, another pass is needed */
  WF_DONE_WAVES,   /* waves of the running kernel that have finished                                                   */
  WF_N_CTL
};

#endif /* RT_DEVICE_H */
