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


#include "rt_dev.hip.h"

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

#ifdef RT_LEDGER
#define RT_LEDGER_WAVES 8192
__device__ uint32_t g_ledger[RT_LEDGER_WAVES * RT_LEDGER_ROW];      // block ledger: one row of LG_* slots per wave (rt_dev.hip.h)
__global__ void rt_ledger_reduce_kernel(unsigned long long *counters) {
  const int slot = threadIdx.x;
  if (slot >= LG_N || slot >= RT_LEDGER_ROW) return;
  unsigned long long sum = 0ull;
  for (int w = 0; w < RT_LEDGER_WAVES; w++) sum += g_ledger[w * RT_LEDGER_ROW + slot];
  counters[8 + slot] = sum;
}
#endif

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
  // (the pyramid: planes and origin in floats 0..18, 8 cache entries of leaf masks at 24..31 (0x800000 | group) << 8 | cull mask, then 128
  //  bytes: 32 cache entries of node masks, direct mapped by node: (node + 1) << 8 | cull mask)

#ifdef RT_LEDGER
  uint32_t *lg = g_ledger + (size_t)__builtin_amdgcn_readfirstlane((int)blockIdx.x * WAVES + wave) * RT_LEDGER_ROW;
  const unsigned long long lg_wave_t0 = __builtin_amdgcn_s_memtime();
#if RT_LEDGER >= 2
  unsigned long long lg_drain_t0 = 0ull;
#endif
#endif
  pow24_lds_init((int)threadIdx.x);
  if (LDSN) {
    LGT0();
    const float4 *g = reinterpret_cast<const float4 *>(P.nodes);
    for (int i = threadIdx.x; i < n_lds * 12; i += WAVES * 64) {
      int nd = i / 12, q = i - nd * 12;
      smem[nd * RT_LDS_NODE_F4 + q] = g[i];
    }
    __syncthreads();          // the only workgroup barrier of the kernel; waves are independent afterwards
    LGT1(LG_CYC_COPY);
  } else {
    __syncthreads();          // (the sRGB scale table)
  }

  acc[lane] = 0ull;
  acc[lane + 64] = 0ull;
  acc[lane + 128] = 0ull;

  // wave-level counters (scalar registers)
  uint32_t w_paths = 0, w_rays = 0, w_nodes = 0, w_leaves = 0, w_shades = 0, w_bgs = 0, w_tex = 0;
  const unsigned long long t_wave_start = cold_args()->wave_times ? __builtin_amdgcn_s_memrealtime() : 0ull;   // RT_WAVE_TIMES only
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
    LGT0();
    LGM("tile_begin");
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
      LG(LG_JOIN_X, 1);
      LGT0();
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
      LGT1(LG_CYC_JOIN);
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
      if (lane < 40) reinterpret_cast<uint32_t *>(pyr)[24 + lane] = 0u;      // the cull masks found for this tile so far
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
          // (+ the placement error of a fused slab distance, see pyramid_cull_mask)
          float coarse = fabsf(n.x) * (fabsf(o.x) + fmaxf(fabsf(nb[0]), fabsf(nb[24]))) + fabsf(n.y) * (fabsf(o.y) + fmaxf(fabsf(nb[8]), fabsf(nb[32]))) +
                         fabsf(n.z) * (fabsf(o.z) + fmaxf(fabsf(nb[16]), fabsf(nb[40])));
          if (nearest > 1e-3f * extent + 1e-6f * coarse) outside = true;        // (NaN compares false: not outside)
        }
        may_hit = !outside;
      }
      tile_root_miss = __ballot(may_hit) == 0ull;
    }
    const uint32_t rays_before = w_rays;
    LG(LG_TILE_X, 1);
    LGM("tile_end");
    LGT1(LG_CYC_TILE);

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

#ifndef RT_NO_SKY_LOOP
    // ================= tiles whose pyramid misses every child of the root: a loop of their own =================
    // Every camera ray of such a tile (56 % of the camera paths of config #3: 299 M of 531 M) costs one node visit that finds no candidate and
    // goes to the environment -- no traversal state, no phases, no parking, no RNG draw.  The loop below does exactly that for
    // batches of up to 64 paths: primary ray, environment lookup, sample into the LDS tile; same arithmetic, same counters as
    // the general loop, which takes over at once -- from the same unit, nothing consumed -- should a ray turn up that is not
    // NaN-free (such a ray does not take the shortcut: its root visit must be computed, see tile_root_miss).
    if (tile_root_miss && cold_args()->max_bounces > 0) {
      RT_KArgs A = cold_args();
      const int width = A->width, height = A->height, sample_first = A->sample_first, sample_end = A->sample_end;
      const uint32_t n_sb = (uint32_t)A->n_sample_blocks, gmax = (uint32_t)A->grab_max;
      PrimaryParams PP;
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) PP.cam[i][j] = A->cam[i][j];
      PP.focal_length = A->focal_length; PP.inv_width = A->inv_width; PP.inv_height = A->inv_height; PP.aspect = A->aspect;
      ShadeParamsLds SP;
      SP.tris = nullptr; SP.mats = nullptr; SP.textures = A->textures; SP.texels = A->texels;
      SP.bg_texture = A->bg_texture; SP.max_bounces = A->max_bounces;
      uint32_t *tile_next = A->tile_next, *open_groups = A->open_groups;
      uint32_t n_sky = 0;                 // paths served here: their root visit is COUNTED (like the oracle's) but not executed
      for (;;) {
        if (c_next >= c_end) {
          if (u_cur + 1u < u_end) {
            u_cur += 1u;
          } else {
            uint32_t u0 = 0;
            if (lane == 0) u0 = atomicAdd(&tile_next[tile_idx], grab);
            u0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)u0);
            if (u0 >= n_chunks_tile) { tile_open = false; break; }
            u_cur = u0;
            u_end = u0 + grab < n_chunks_tile ? u0 + grab : n_chunks_tile;
            if (u_end == n_chunks_tile && lane == 0) atomicSub(&open_groups[tile_idx >> 6], 1u);
            if (t_wave_start) t_last_grab = __builtin_amdgcn_s_memrealtime();
            LG(LG_GRAB_X, 1);
            const uint32_t left = n_chunks_tile - u_end;
            grab = left >= 8u * gmax ? gmax : (left >= 8u && gmax >= 2u ? 2u : 1u);
            took_any = true;
          }
          const uint32_t grp = u_cur >> 2, pair = u_cur & 3u;
          const uint32_t row = grp / n_sb, sb = grp - row * n_sb;
          c_x0 = tile_x0 + (int)pair * 2;
          c_y = tile_y0 + (int)row;
          c_pix0 = (int)row * 8 + (int)pair * 2;
          c_s0 = sample_first + (int)(sb << shift);
          c_next = 0;
          c_end = (c_y < height && c_x0 < width) ? unit_paths : 0;
          continue;
        }
        // RT_SKY_PATHS paths per lane and iteration.  Two or three -- independent chains of primary ray -> environment lookup for
        // the scheduler to interleave, in a loop that carries no other state -- measure the same as one (32.50 / 32.65 vs 32.55 ms:
        // profiles/r04o_sky_ab.log): what the loop saves is the phase machine's instructions, scalar and vector, not latency.
#ifndef RT_SKY_PATHS
#define RT_SKY_PATHS 1
#endif
        LGM("sky_begin");
        LGT0();
        const int avail = c_end - c_next;
        const int take = avail < 64 * RT_SKY_PATHS ? avail : 64 * RT_SKY_PATHS;
        const int l = lane_now();
        bool valid[RT_SKY_PATHS];
        int  pxs[RT_SKY_PATHS];
        rt_v3 dirs[RT_SKY_PATHS];
        bool slow = false;
#pragma unroll
        for (int q = 0; q < RT_SKY_PATHS; q++) {
          const int kq = l + 64 * q;
          const int k = c_next + kq;
          pxs[q] = k >> shift;
          const int sm = c_s0 + (k & ((1 << shift) - 1));
          const int x = c_x0 + pxs[q];
          valid[q] = kq < take && x < width && sm < sample_end;
          rt_v3 o = rt_v3_make(0, 0, 0);
          dirs[q] = rt_v3_make(0, 0, 1);
          Ray3 r;
          if (valid[q]) primary_ray(PP, x, c_y, sm, o, dirs[q]);
          ray_setup<SHORT_DIV>(r, o, dirs[q]);
          slow = slow || (valid[q] && !r.fast);
        }
        if (__ballot(slow) != 0ull) break;          // the general loop redoes this batch, and the rest of the tile
        uint32_t nv = 0;
#pragma unroll
        for (int q = 0; q < RT_SKY_PATHS; q++) {
          if (valid[q]) {
            const rt_v3 bg = background_lookup(SP, dirs[q]);
            const rt_v3 radiance = rt_v3_mul_add(bg, rt_v3_make(1, 1, 1), rt_v3_make(0, 0, 0));      // tint 1, emission 0 (raytracer.c:554)
            unsigned long long *ap = reinterpret_cast<unsigned long long *>(lds_at(smem, acc_off) + (c_pix0 + pxs[q]) * 6);
            atomicAdd(ap + 0, accum_quantize_dev(radiance.x));
            atomicAdd(ap + 1, accum_quantize_dev(radiance.y));
            atomicAdd(ap + 2, accum_quantize_dev(radiance.z));
          }
          nv += (uint32_t)__popcll(__ballot(valid[q]));
        }
        w_paths += nv; w_rays += nv; w_nodes += nv; w_bgs += nv;      // (the skipped root visit counts, as in the general loop)
        n_sky += nv;
        LG(LG_SKY_X, 1); LG(LG_SKY_L, nv);
        c_next += take;
        LGT1(LG_CYC_SKY);
        LGM("sky_end");
      }
      // counters[CNT_SKIPPED_ROOT]: node visits that are counted but not executed (bench.py's roofline footnote)
      if (n_sky != 0u && lane == 0) atomicAdd(cold_args()->counters + CNT_SKIPPED_ROOT, (unsigned long long)n_sky);
    }
#endif

    for (;;) {
      // ================= S: shade the hits, environment for the misses, start new paths =================
      {
        LGT0();
        LGM("s_begin");
        LG(LG_S_ITER, 1);
        LaneCounters cn;
        cn.rays = cn.nodes = cn.leaves = cn.shades = cn.bgs = cn.textured = cn.paths = 0;
        bool  done = false, start = false, fresh = false;
        rt_v3 radiance = rt_v3_make(0, 0, 0);
        rt_v3 org = ray.o, dir = ray.d;
        RT_KArgs A = cold_args();
        ShadeParamsLds SP;
        SP.tris = A->tris; SP.mats = A->mats; SP.textures = A->textures; SP.texels = A->texels;
        SP.bg_texture = A->bg_texture; SP.max_bounces = A->max_bounces;
        // ---- environment for the paths that left the scene ----
#ifdef RT_LEDGER
        { const int n_env = (int)__popcll(__ballot(phase == PH_MISS)); LG(LG_ENV_X, n_env != 0); LG(LG_ENV_L, n_env); }
#endif
        LGM("env_begin");
        if (phase == PH_MISS) {
          cn.bgs = 1;
          rt_v3 bg = background_lookup(SP, dir);
          radiance = rt_v3_mul_add(bg, tint, emis);
          done = true;
        }
        LGM("env_end");
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
          // (the last paths of a tile are not parked: what is parked when the tile closes runs its bounce chain AFTER the chains of
          // the paths in flight -- the drain of the tile, and of the launch, gets a second chain long; RT_PARK_STOP_PATHS)
          const bool tile_ending = RT_PARK_STOP_PATHS > 0 &&
                                   (int)(n_chunks_tile - u_end) * unit_paths + (int)(u_end - u_cur) * unit_paths <= RT_PARK_STOP_PATHS;
          if (!tile_open || tile_ending || h + back >= RT_PARK_DENSE || n_parked + h > RT_PARK_CAP) {
            if (back > 0) {
              LG(LG_PLOAD_X, 1); LG(LG_PLOAD_L, back);
              LGM("pload_begin");
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
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
              LGM("pload_end");
            }
          } else if (h > 0) {
            LG(LG_PSTORE_X, 1); LG(LG_PSTORE_L, h);
            LGM("pstore_begin");
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
              // the record is read back by ANOTHER lane of this wave (agent-scope loads below): the release / acquire pair at
              // wavefront scope states that order to the compiler; the hardware returns one wave's vector memory operations in order
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              phase = PH_NEED;
            }
            n_parked += h;
            LGM("pstore_end");
          }
        }
#ifdef RT_LEDGER
        { const int n_sh = (int)__popcll(__ballot(phase == PH_HIT && shade_now)); LG(LG_SHADE_X, n_sh != 0); LG(LG_SHADE_L, n_sh); }
#endif
        LGM("shade_begin");
        if (phase == PH_HIT && shade_now) {
          done = shade_hit(SP, hit, org, dir, tint, emis, rng, bounce, cn, radiance);
          start = !done;
        }
        LGM("shade_end");
#ifdef RT_LEDGER
        { const int n_acc = (int)__popcll(__ballot(done)); LG(LG_ACCUM_X, n_acc != 0); LG(LG_ACCUM_L, n_acc); }
#endif
        LGM("accum_begin");
        if (done) {
          // (32-bit address arithmetic from the wave's byte offset: `acc + pix * 3` is a 64-bit multiply-add on a pointer
          // that is kept in scratch)
          unsigned long long *ap = reinterpret_cast<unsigned long long *>(lds_at(smem, acc_off) + pix * 6);
          atomicAdd(ap + 0, accum_quantize_dev(radiance.x));
          atomicAdd(ap + 1, accum_quantize_dev(radiance.y));
          atomicAdd(ap + 2, accum_quantize_dev(radiance.z));
          phase = PH_NEED;
        }
        LGM("accum_end");
        w_shades += (uint32_t)__popcll(__ballot(cn.shades != 0));
        w_tex += (uint32_t)__popcll(__ballot(cn.textured != 0));
        w_bgs += (uint32_t)__popcll(__ballot(cn.bgs != 0));

        // ---- regeneration: idle lanes take the next paths of the tile, across chunk boundaries ----
        if (tile_open) {
          unsigned long long need = __ballot(phase == PH_NEED);
          bool got = false;
          int  gx = 0, gy = 0, gs = 0, gp = 0;
          const int width = A->width, sample_end = A->sample_end;
          LGM("regen_begin");
          while (need) {
            LG(LG_REGEN_X, 1);
            if (c_next >= c_end) {
              if (u_cur + 1u < u_end) {
                u_cur += 1u;
              } else {
                uint32_t u0 = 0;
                if (lane == 0) u0 = atomicAdd(&A->tile_next[tile_idx], grab);
                u0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)u0);
                if (u0 >= n_chunks_tile) {
                  tile_open = false;
                  LGD0((int)__popcll(__ballot(phase != PH_NEED || got)) + n_parked);      // (ledger builds: the tile's drain starts here)
                  break;
                }
                u_cur = u0;
                u_end = u0 + grab < n_chunks_tile ? u0 + grab : n_chunks_tile;
                // exactly one wave receives the tile's last unit: it closes the tile in the group summary
                if (u_end == n_chunks_tile && lane == 0) atomicSub(&A->open_groups[tile_idx >> 6], 1u);
                LG(LG_GRAB_X, 1);
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
          LGM("regen_end");
#ifdef RT_LEDGER
          { const int n_pr = (int)__popcll(__ballot(got)); LG(LG_PRIM_X, n_pr != 0); LG(LG_PRIM_L, n_pr); LG(LG_REGEN_L, n_pr); }
#endif
          LGM("prim_begin");
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
          LGM("prim_end");
        }
        bool skip_root = false;
#ifdef RT_LEDGER
        { const int n_st = (int)__popcll(__ballot(start)); LG(LG_START_X, n_st != 0); LG(LG_START_L, n_st); }
#endif
        LGM("start_begin");
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
        LGM("start_end");
        w_rays += (uint32_t)__popcll(__ballot(start));
        {
          const uint32_t n_skip = (uint32_t)__popcll(__ballot(skip_root));
          w_nodes += n_skip;
          if (n_skip != 0u && lane == 0) atomicAdd(cold_args()->counters + CNT_SKIPPED_ROOT, (unsigned long long)n_skip);   // (rare: a sky tile outside the sky loop)
        }
        LGM("s_end");
        LGT1(LG_CYC_S);
      }

      const int n_trav0 = (int)__popcll(__ballot(phase == PH_NODE || phase == PH_LEAF));
      if (n_trav0 == 0) {
        if (__any(phase == PH_MISS)) continue;   // camera rays that skipped the root: straight to the environment
        if (n_parked > 0 || __any(phase == PH_HIT)) continue;      // parked hits come back in the next S block
        if (!tile_open) break;            // every path of the tile that this wave took has ended
        continue;                         // (nothing started, e.g. pixels outside the image: pull more)
      }

      // ================= traversal: NODE / LEAF blocks until `thresh` lanes wait for S =================
      // while the tile still hands out paths, wait until `thresh` lanes want the S block (dense shading); once it is
      // exhausted nothing refills the lanes, and what matters is the latency of the remaining paths' bounce chains:
      // shade as soon as `drain_thresh` lanes wait
      traversal_blocks<LDSN, SHORT_DIV, true>(P, smem, lds_nodes, perm, lane, n_lds, pyr_nodes, pyr_off, leaf_level,
                                              tile_open ? thresh : drain_thresh, n_trav0, ray, bounce == 0, phase, level, node,
                                              child, cur, dirty, live, hit, w_nodes, w_leaves LG_ARG);
    }

    // ---------------- flush the wave's share of the tile: lane p owns pixel p ----------------
    LGD1();
    LGM("flush_begin");
    if (took_any) {
      LGT0();
      LG(LG_FLUSH_X, 1);
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
      LGT1(LG_CYC_FLUSH);
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
#ifdef RT_LEDGER
    LG(LG_CYC_WAVE, (uint32_t)((__builtin_amdgcn_s_memtime() - lg_wave_t0) >> 4));
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
      radiance = rt_v3_mul_add(background_lookup(P, dir), tint, emis);
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
// Per-launch preparation in ONE launch of one workgroup: zero the ray counters and the work head, reset every tile's unit
// counter and the open-tile count of every group of 64 tiles, zero the cost buffer this launch will fill, and -- when the
// previous launch of the same view left its costs -- the tile order of this one: counting sort of the tiles by descending
// cost bucket (4 buckets per power of two of that launch's ray count; order inside a bucket is whatever the LDS atomics give:
// only the schedule depends on it, never a pixel).  Round 2 spent three memsets and five small kernels on this, ~25 us
// of launch gaps per frame -- a fifth of a frame at the reference's default size (1024 x 1024, 16 spp).
#define RT_ORDER_BUCKETS 132
__device__ __forceinline__ int cost_bucket(uint32_t c) {
  if (c < 4u) return (int)c;                               // 0..3
  int e = 31 - __clz((int)c);                              // >= 2
  return 4 * (e - 1) + (int)((c >> (e - 2)) & 3u);         // 4..131, monotonic in c
}

__global__ __launch_bounds__(1024) void rt_prepare_kernel(int n_tiles, uint32_t *tile_next, uint32_t *open_groups,
                                                          unsigned long long *counters, uint32_t *work_head, uint32_t *cost_cur,
                                                          const uint32_t *cost_prev, uint32_t *order) {
  // counting sort: every wave counts its own share of the tiles per bucket in its own LDS row, one scan turns the 16 x 132
  // counts into per-(wave, bucket) start positions, and every wave scatters its tiles from its own cursors.  (One shared
  // row of LDS atomics took 100-170 us for the 32 400 tiles of 1080p: most tiles fall into a handful of buckets.)
  __shared__ uint32_t hist[16][RT_ORDER_BUCKETS];
  __shared__ uint32_t bucket_start[RT_ORDER_BUCKETS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < RT_N_COUNTERS) counters[tid] = 0ull;
  if (tid < 16) work_head[tid] = 0u;
  for (int i = tid; i < n_tiles; i += 1024) {
    if (tile_next) tile_next[i] = 0u;
    if (cost_cur) cost_cur[i] = 0u;
  }
  if (open_groups) {
    const int n_groups = (n_tiles + 63) >> 6;
    for (int g = tid; g < n_groups; g += 1024) {
      int left = n_tiles - g * 64;
      open_groups[g] = (uint32_t)(left < 64 ? left : 64);
    }
  }
  if (!cost_prev || !order) return;
  // a wave owns the tiles [wave * per_wave, (wave + 1) * per_wave) and its own row of counters in LDS.  Per batch of 64
  // tiles the (up to) three most common buckets are counted by ballot -- neighbouring tiles cost about the same, most of a
  // batch falls into one or two buckets, and 64 LDS atomics on one word serialise -- the stragglers by LDS atomics.
  for (int b = lane; b < RT_ORDER_BUCKETS; b += 64) hist[wave][b] = 0u;
  const int per_wave = ((n_tiles + 15) / 16 + 63) & ~63;
  const int first = wave * per_wave, last = first + per_wave < n_tiles ? first + per_wave : n_tiles;
  // (16 batches' costs are loaded together: one dependent load per batch would cost its full latency -- the cost buffer was
  // written by the previous launch's atomics and sits at the memory side -- 32 times per wave and phase)
  for (int group = first; group < last; group += 64 * 16) {
    int bk[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = group + 64 * k + lane;
      // (an unsigned cost of 2^31 or more must not read as "no tile" -- the tile would drop out of the count AND of the scatter and
      //  leave a stale entry in the order: it saturates; the bucket function is monotonic)
      bk[k] = i < last ? (int)(cost_prev[i] > 0x7FFFFFFFu ? 0x7FFFFFFFu : cost_prev[i]) : -1;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int bkt = bk[k] >= 0 ? cost_bucket((uint32_t)bk[k]) : -1;
      unsigned long long todo = __ballot(bkt >= 0);
#pragma unroll
      for (int round = 0; round < 3; round++) {
        if (!todo) break;
        const int b0 = __builtin_amdgcn_readlane(bkt, (int)__builtin_ctzll(todo));
        const unsigned long long m = __ballot(bkt == b0);
        if (lane == 0) atomicAdd(&hist[wave][b0], (uint32_t)__popcll(m));
        todo &= ~m;
      }
      if ((todo >> lane) & 1ull) atomicAdd(&hist[wave][bkt], 1u);
    }
  }
  __syncthreads();
  if (tid < RT_ORDER_BUCKETS) {                            // per bucket: total, and the waves' shares as running offsets
    uint32_t run = 0;
    for (int w = 0; w < 16; w++) { uint32_t c = hist[w][tid]; hist[w][tid] = run; run += c; }
    bucket_start[tid] = run;                               // (count for now)
  }
  __syncthreads();
  if (tid == 0) {                                          // start offset of every bucket, expensive first
    uint32_t run = 0;
    for (int b = RT_ORDER_BUCKETS - 1; b >= 0; b--) { uint32_t c = bucket_start[b]; bucket_start[b] = run; run += c; }
  }
  __syncthreads();
  for (int b = lane; b < RT_ORDER_BUCKETS; b += 64) hist[wave][b] += bucket_start[b];     // this wave's cursors
  for (int group = first; group < last; group += 64 * 16) {
    int bk[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = group + 64 * k + lane;
      // (an unsigned cost of 2^31 or more must not read as "no tile" -- the tile would drop out of the count AND of the scatter and
      //  leave a stale entry in the order: it saturates; the bucket function is monotonic)
      bk[k] = i < last ? (int)(cost_prev[i] > 0x7FFFFFFFu ? 0x7FFFFFFFu : cost_prev[i]) : -1;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int i = group + 64 * k + lane;
      const int bkt = bk[k] >= 0 ? cost_bucket((uint32_t)bk[k]) : -1;
      unsigned long long todo = __ballot(bkt >= 0);
#pragma unroll
      for (int round = 0; round < 3; round++) {
        if (!todo) break;
        const int b0 = __builtin_amdgcn_readlane(bkt, (int)__builtin_ctzll(todo));
        const unsigned long long m = __ballot(bkt == b0);
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&hist[wave][b0], (uint32_t)__popcll(m));
        pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (bkt == b0) order[pos + (uint32_t)rank] = (uint32_t)i;
        todo &= ~m;
      }
      if ((todo >> lane) & 1ull) order[atomicAdd(&hist[wave][bkt], 1u)] = (uint32_t)i;
    }
  }
}

extern "C" int rt_launch_prepare(int n_tiles, uint32_t *tile_next, uint32_t *open_groups, unsigned long long *counters,
                                 uint32_t *work_head, uint32_t *cost_cur, const uint32_t *cost_prev, uint32_t *order,
                                 hipStream_t stream) {
  hipLaunchKernelGGL(rt_prepare_kernel, dim3(1), dim3(1024), 0, stream, n_tiles, tile_next, open_groups, counters, work_head, cost_cur,
                     cost_prev, order);
  return (int)hipGetLastError();
}

#ifdef RT_DIAG_VARIANTS
// ---------------------------------------------------------------------------------
// unit-level kernels for parity tests: compiled into the DIAGNOSTIC library only (librt_hip_diag.so, include/rt_hip_diag.h).
// They instantiate the same device functions as the product's kernels (rt_dev.hip.h; traversal_blocks() is the path
// kernel's own traversal); the product library carries no test entry point.

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
// rcp_leaf(x), the leaf blocks' form without the fix-up: counts[4] = finite non-zero patterns with |x| < 2^102 that differ from
// 1.0f / x, counts[5] = patterns x = +-0, +-infinity, NaN for which it is NOT NaN (what the leaf blocks' argument rests on).
__global__ void rt_test_rcp_sweep_kernel(unsigned long long *counts) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t bad_in = 0, n_out = 0, bad_out = 0, first = 0, leaf_bad = 0, leaf_special = 0;
  for (uint32_t k = 0; k < 256u; k++) {
    const uint32_t b = tid * 256u + k;
    const float x = __uint_as_float(b);
    const uint32_t w = __float_as_uint(1.0f / x), g = __float_as_uint(rcp_exact(x));
    const bool w_nan = (w & 0x7FFFFFFFu) > 0x7F800000u, g_nan = (g & 0x7FFFFFFFu) > 0x7F800000u;
    const bool same = w_nan ? g_nan : (g == w);
    if (rcp_exact_outside(x)) { n_out += 1; bad_out += same ? 0u : 1u; }
    else if (!same) { bad_in += 1; if (!first) first = b + 1u; }
    const uint32_t l = __float_as_uint(rcp_leaf(x)), mag = b & 0x7FFFFFFFu;
    const bool l_nan = (l & 0x7FFFFFFFu) > 0x7F800000u;
    if (mag == 0u || mag >= 0x7F800000u) leaf_special += l_nan ? 0u : 1u;
    else if (!rcp_exact_outside(x)) leaf_bad += (l == w) ? 0u : 1u;
  }
  if (leaf_bad) atomicAdd(&counts[4], (unsigned long long)leaf_bad);
  if (leaf_special) atomicAdd(&counts[5], (unsigned long long)leaf_special);
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

// Arbitrary rays through the PRODUCTION traversal: traversal_blocks() -- the NODE / LEAF / pop code of the path kernels --
// in the path kernel's workgroup geometry (16 waves, tree in LDS, per-wave perm stacks), lanes refilled from the ray list
// as they finish, blocks mixed exactly as a frame mixes them.  With a pyramid (`pyr`: 4 outward plane normals at
// [4 q .. 4 q + 2], the common ray origin at [16 .. 18]) every ray counts as a camera ray of one tile: node blocks take
// the culled form (pyramid_cull_mask, node_enter_few) whenever the path kernel would.  visits[0 / 1] += node / leaf
// visits (raytracer.c:452 / :476 calls).  Compared with oracle_trace_rays_counted() by tests/test_gpu_trace_stream.py.
template <bool SHORT_DIV, bool PYRAMID>
__global__ __launch_bounds__(16 * 64, 1) void rt_test_trace_stream_kernel(RT_KParams P, int n, const float *rays, const float *pyr_in,
                                                                         int exit_lanes, float *out_t, int *out_tri, float *out_uv,
                                                                         unsigned long long *visits) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_lds = P.n_lds_nodes;
  const float4 *lds_nodes = smem;
  const int perm_f4 = (P.depth > 0 ? P.depth : 1) * 16;
  uint32_t *perm = reinterpret_cast<uint32_t *>(smem + n_lds * RT_LDS_NODE_F4 + wave * (perm_f4 + 96));
  const int pyr_off = (n_lds * RT_LDS_NODE_F4 + __builtin_amdgcn_readfirstlane(wave) * (perm_f4 + 96) + perm_f4 - 16) * 16;
  {
    const float4 *g = reinterpret_cast<const float4 *>(P.nodes);
    for (int i = threadIdx.x; i < n_lds * 12; i += 16 * 64) {
      int nd = i / 12, q = i - nd * 12;
      smem[nd * RT_LDS_NODE_F4 + q] = g[i];
    }
    __syncthreads();
  }
  if (PYRAMID) {
    float *pyr = lds_at(smem, pyr_off);
    if (lane < 19) pyr[lane] = pyr_in[lane];
    if (lane < 40) reinterpret_cast<uint32_t *>(pyr)[24 + lane] = 0u;
  }
  const int leaf_level = P.depth - 1;
  const int n_waves = (int)gridDim.x * 16, wave_id = (int)blockIdx.x * 16 + wave;
  const int per_wave = (n + n_waves - 1) / n_waves;
  int next = wave_id * per_wave;                                   // this wave's slice of the ray list
  const int end = next + per_wave < n ? next + per_wave : n;

  int   phase = PH_NEED, idx = 0;
  Ray3  ray;
  ray_setup(ray, rt_v3_make(0, 0, 0), rt_v3_make(0, 0, 1));
  int   level = -1, node = 0, child = 0;
  uint32_t cur = 0, dirty = 0, live = 0, w_nodes = 0, w_leaves = 0;
  HitRec hit;
  hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;
  for (;;) {
    if (phase == PH_HIT || phase == PH_MISS) {
      out_t[idx] = hit.t;
      out_tri[idx] = hit.tri;
      out_uv[idx * 2 + 0] = hit.u;
      out_uv[idx * 2 + 1] = hit.v;
      phase = PH_NEED;
    }
    if (next < end) {
      const unsigned long long need = __ballot(phase == PH_NEED);
      const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
      if (phase == PH_NEED && next + rank < end) {
        idx = next + rank;
        const float *r = rays + (size_t)idx * 6;
        ray_setup<SHORT_DIV>(ray, rt_v3_make(r[0], r[1], r[2]), rt_v3_make(r[3], r[4], r[5]));
        hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;
        dirty = 0; live = 0; cur = 0; level = -1; node = 0;
        child = (leaf_level >= 0) ? 0 : P.last_row_offset;
        phase = (leaf_level >= 0) ? PH_NODE : PH_LEAF;
      }
      next += (int)__popcll(need);
    }
    const int n_trav0 = (int)__popcll(__ballot(phase == PH_NODE || phase == PH_LEAF));
    if (n_trav0 == 0) {
      if (next >= end) break;
      continue;
    }
    traversal_blocks<true, SHORT_DIV, PYRAMID>(P, smem, lds_nodes, perm, lane, n_lds, PYRAMID ? P.pyr_nodes : 0, pyr_off, leaf_level,
                                               next < end ? exit_lanes : 1, n_trav0, ray, true, phase, level, node, child, cur, dirty,
                                               live, hit, w_nodes, w_leaves);
  }
  if (lane == 0) {
    atomicAdd(visits + 0, (unsigned long long)w_nodes);
    atomicAdd(visits + 1, (unsigned long long)w_leaves);
  }
}

extern "C" int rt_launch_test_trace_stream(const RT_KParams *P, int n, const float *rays, const float *pyr, int exit_lanes, int n_blocks,
                                           int smem_bytes, float *out_t, int *out_tri, float *out_uv, unsigned long long *visits,
                                           hipStream_t stream) {
#define RT_TTS(SD, PY)                                                                                                          \
  do {                                                                                                                          \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rt_test_trace_stream_kernel<SD, PY>),                    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                 \
    if (e != hipSuccess) return (int)e;                                                                                         \
    hipLaunchKernelGGL((rt_test_trace_stream_kernel<SD, PY>), dim3(n_blocks), dim3(16 * 64), smem_bytes, stream, *P, n, rays,  \
                       pyr, exit_lanes, out_t, out_tri, out_uv, visits);                                                        \
  } while (0)
  if (P->short_div) { if (pyr) RT_TTS(true, true); else RT_TTS(true, false); }
  else { if (pyr) RT_TTS(false, true); else RT_TTS(false, false); }
#undef RT_TTS
  return (int)hipGetLastError();
}
#endif  // RT_DIAG_VARIANTS (unit-level kernels)

// ---------------------------------------------------------------------------------
// launchers (called from rt_api.cpp)

template <int WAVES, bool LDSN, int MINW, bool SHORT_DIV>
static int launch_stream(const RT_KParams *P, int n_waves, int smem_bytes, hipStream_t stream) {
  // (the attribute belongs to the kernel ON ONE DEVICE: a frame spread over N GPUs launches from N devices)
  static unsigned attr_devices = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (smem_bytes > 48 * 1024 && (dev >= 32 || !(__atomic_load_n(&attr_devices, __ATOMIC_RELAXED) & (1u << dev)))) {
    // (dynamic + the kernel's 32 static bytes, rt_pow24_lds, must stay within the 160 KB of a CU)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rt_path_kernel_stream<WAVES, LDSN, MINW, SHORT_DIV>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    if (e != hipSuccess) return (int)e;
    if (dev < 32) __atomic_fetch_or(&attr_devices, 1u << dev, __ATOMIC_RELAXED);
  }
#ifdef RT_LEDGER
  {
    void *sym = nullptr;
    if (n_waves > RT_LEDGER_WAVES) return (int)hipErrorInvalidValue;
    hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(g_ledger));
    if (e == hipSuccess) e = hipMemsetAsync(sym, 0, sizeof(uint32_t) * RT_LEDGER_WAVES * RT_LEDGER_ROW, stream);
    if (e != hipSuccess) return (int)e;
  }
#endif
  hipLaunchKernelGGL((rt_path_kernel_stream<WAVES, LDSN, MINW, SHORT_DIV>), dim3((n_waves + WAVES - 1) / WAVES),
                     dim3(WAVES * 64), smem_bytes, stream, *P);
#ifdef RT_LEDGER
  hipLaunchKernelGGL(rt_ledger_reduce_kernel, dim3(1), dim3(RT_LEDGER_ROW), 0, stream, P->counters);
#endif
  return (int)hipGetLastError();
}

#ifdef RT_DIAG_VARIANTS
extern "C" int rt_launch_path_kernel_diag(const RT_KParams *P, int n_waves, int variant, int smem_bytes, hipStream_t stream);
#endif

#ifndef RT_STREAM_MINW
#define RT_STREAM_MINW 1     // (experiment builds: 5 caps the kernel at 96 VGPRs, 6 at 80 -- profiles/r03_experiments.md)
#endif
extern "C" int rt_math_contract(void) { return RT_MATH_CONTRACT; }      // include/rt_math.h: 2 = explicit FMA, 1 = -DRT_MATH_NO_FMA

// variant 5 = the tile-stream kernel, the only path kernel of the product library; 1-4 exist in the diagnostic build only
// `wg_waves` = waves per workgroup, 8 / 12 / 16 (one workgroup per CU: 2 / 3 / 4 waves per SIMD), chosen by rt_api.cpp from the size
// of the launch; the same kernel source, three instances of its launch geometry.
extern "C" int rt_launch_path_kernel(const RT_KParams *P, int n_waves, int variant, int smem_bytes, int wg_waves, hipStream_t stream) {
#ifdef RT_DIAG_VARIANTS
  if (variant >= 1 && variant <= 4) return rt_launch_path_kernel_diag(P, n_waves, variant, smem_bytes, stream);
#endif
  (void)variant;
  if (wg_waves == 8)
    return P->short_div ? launch_stream<8, true, RT_STREAM_MINW, true>(P, n_waves, smem_bytes, stream)
                        : launch_stream<8, true, RT_STREAM_MINW, false>(P, n_waves, smem_bytes, stream);
  if (wg_waves == 12)
    return P->short_div ? launch_stream<12, true, RT_STREAM_MINW, true>(P, n_waves, smem_bytes, stream)
                        : launch_stream<12, true, RT_STREAM_MINW, false>(P, n_waves, smem_bytes, stream);
  return P->short_div ? launch_stream<16, true, RT_STREAM_MINW, true>(P, n_waves, smem_bytes, stream)
                      : launch_stream<16, true, RT_STREAM_MINW, false>(P, n_waves, smem_bytes, stream);
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

// raw u8 image rows (stride pixels x comp bytes, comp >= 3) -> RGBA8 words in the pool layout of rt_device.h (4 x 4 tiles, or
// row-major).  `rows` rows starting at row y0 of a texture `width` wide whose first texel is out[0] (rt_scene_touch packs a
// few rows of a resident texture again).
__global__ void rt_pack_texture_kernel(const uint8_t *raw, int width, int rows, int y0, int stride, int comp, uint32_t *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)width * rows) return;
  int yl = (int)(i / width), x = (int)(i % width);
  const uint8_t *p = raw + ((size_t)yl * stride + x) * comp;
  const int y = y0 + yl;
#if RT_TEX_TILED
  const size_t o = (size_t)RT_TEX_TILE_INDEX(x, y, (width + 3) >> 2);
#else
  const size_t o = (size_t)y * width + x;
#endif
  out[o] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xFF000000u;
}

extern "C" int rt_launch_pack_texture(const uint8_t *raw, int width, int rows, int y0, int stride, int comp, uint32_t *out,
                                      hipStream_t stream) {
  size_t n = (size_t)width * rows;
  hipLaunchKernelGGL(rt_pack_texture_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, raw, width, rows, y0, stride,
                     comp, out);
  return (int)hipGetLastError();
}

#ifdef RT_DIAG_VARIANTS
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
#endif  // RT_DIAG_VARIANTS (unit-level launchers)
