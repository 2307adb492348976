// rt_kernels_diag.hip -- the superseded generations of the path kernel (variants 1-4), kept as bisect tools.
// Compiled ONLY into librt_hip_diag.so (make diag, -DRT_DIAG_VARIANTS); the product library librt_hip.so carries one
// path kernel (rt_path_kernel_stream, rt_kernels.hip) and has no switch between generations.
//   variant 1: rt_path_kernel            plain "regenerate -> trace every live ray to the end -> shade" loop
//   variant 2: rt_path_kernel_sched<4,false>   phase scheduled, 256-thread workgroups, nodes from L1/L2
//   variant 3: rt_path_kernel_sched<16,true>   phase scheduled, top of the BVH in LDS (round 1's kernel)
//   variant 4: variant 3 + block statistics
#include "rt_dev.hip.h"

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
          radiance = rt_v3_mul_add(bg, tint, emis);
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
          radiance = rt_v3_mul_add(bg, tint, emis);
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
          } else if (ray.fast) {       // (mixed block: every lane in the form its own ray is entitled to, see traversal_blocks)
            cur = node_enter<true, NODE_GLOBAL>(P, ray, node, hit.t, lds_nodes);
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
              if (LDSN && node < n_lds) dj = slab_entry_child_any(reinterpret_cast<const float *>(lds_nodes + lds_node_f4(node)) + j, ray);
              else dj = slab_entry_child_any(P.nodes + (size_t)node * 48 + j, ray);
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

// variant 1: plain while-while kernel; 2: phase-scheduled, 256-thread workgroups, nodes from L1/L2;
// 3: phase-scheduled, one 1024-thread workgroup per CU with the top of the BVH in LDS (default);
// 4: variant 3 plus block statistics (diagnostic).  (Occupancy experiments -- 5 or 6 waves per SIMD with register
// spills and no LDS node copy, two half-size LDS copies per CU -- lost to variant 3: numbers in DESIGN.md.)
// n_waves = total wavefronts wanted; smem_bytes = dynamic LDS per workgroup (variants >= 2).
template <int WAVES, bool LDSN, bool STATS, int MINW>
static int launch_sched(const RT_KParams *P, int n_waves, int smem_bytes, hipStream_t stream) {
  if (smem_bytes > 48 * 1024) {          // (per device; the diagnostic generations set it at every launch)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rt_path_kernel_sched<WAVES, LDSN, STATS, MINW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((rt_path_kernel_sched<WAVES, LDSN, STATS, MINW>), dim3((n_waves + WAVES - 1) / WAVES), dim3(WAVES * 64),
                     smem_bytes, stream, *P);
  return (int)hipGetLastError();
}

extern "C" int rt_launch_path_kernel_diag(const RT_KParams *P, int n_waves, int variant, int smem_bytes, hipStream_t stream) {
  switch (variant) {
  case 1:
    hipLaunchKernelGGL(rt_path_kernel, dim3((n_waves + 3) / 4), dim3(RT_BLOCK_THREADS), 0, stream, *P);
    return (int)hipGetLastError();
  case 2: return launch_sched<4, false, false, 1>(P, n_waves, smem_bytes, stream);
  case 4: return launch_sched<16, true, true, 1>(P, n_waves, smem_bytes, stream);
  default: return launch_sched<16, true, false, 1>(P, n_waves, smem_bytes, stream);
  }
}
