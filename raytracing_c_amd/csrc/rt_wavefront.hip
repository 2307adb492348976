// rt_wavefront.hip -- the render hot path as THREE kernels joined by record queues in HBM, instead of one kernel that
// carries shading and traversal in the same 128 registers (rt_path_kernel_stream, rt_kernels.hip):
//
//   rt_wf_camera_kernel   render_thread_proc's pixel / sample loop (raytracer.c:596-720) + the first ray_scene_hit of
//                         every path (raytracer.c:513-514) + the environment for camera rays that leave the scene
//                         (raytracer.c:554).  Persistent waves own 8x8 tiles and pull units of them exactly like the
//                         tile-stream kernel (same counters, same joining); ALL lanes are camera rays of one tile, so
//                         every node block can be culled by the tile's pyramid.  A hit is not shaded here: 9 dwords
//                         (direction, t, triangle, u, v, pixel, sample) go to the hit queue and the lane takes the
//                         next camera path.  No path state (tint, emission, rng, bounce) lives in registers.
//   rt_wf_shade_kernel    the body of cast_ray's loop for one accepted hit (raytracer.c:515-552 -> disney_shader_proc,
//                         driver.c:350-409): reads 64 hit records per block -- every lane busy, whatever the bounce --
//                         and writes the continuation ray (15 dwords) to the ray queue, or adds the finished path's
//                         radiance to the frame with 64-bit atomics.
//   rt_wf_trace_kernel    ray_scene_hit for continuation rays (raytracer.c:497-503 -> :443-483): persistent waves
//                         refill idle lanes from the ray queue; a lane carries its ray, its traversal state and the
//                         INDEX of its ray record -- tint / emission / rng stay in memory until the path ends (miss:
//                         environment * tint + emission -> atomics) or is shaded again (hit: 5 dwords to the hit queue).
//
// One frame = camera kernel, then (shade, trace) per bounce; launches of one stream, so a kernel boundary is the only
// synchronisation between a producer and its consumer: no polling, no inter-kernel flags, nothing that can deadlock.
// The queues are sized by the host (rt_api.cpp); when the camera kernel runs into the end of its hit queue its waves
// stop taking units and the host runs the bounces and calls the camera kernel again -- the tile / unit counters keep the
// position.  Per-lane arithmetic is that of rt_dev.hip.h in every kernel, radiance sums are order-free integers
// (rt_math.h), so images and counters equal the tile-stream kernel's and the CPU oracle's bit for bit.

#include "rt_dev.hip.h"

#define WF_NONE 0xFFFFFFFFu

// ---- wave-level queue output ---------------------------------------------------------------------------------------
// A wave fills chunks it obtained from the queue's allocation counter; `chunk` / `pos` are wave-uniform.  A chunk is
// closed (its record count stored) when the next append does not fit; at most 63 slots of a chunk stay unused.
struct WfOut {
  uint32_t chunk;
  int      pos;
};

template <int FIELDS>
__device__ __forceinline__ bool wf_append(uint32_t *data, uint32_t *cnt, uint32_t *alloc, WfOut &o, bool has,
                                          const uint32_t (&v)[FIELDS], uint32_t soft_chunks) {
  const unsigned long long m = __ballot(has);
  bool over = false;
  if (m == 0ull) return over;
  const int n = (int)__popcll(m);
  if (o.pos + n > WF_CHUNK) {
    const int lane = lane_now();
    if (o.chunk != WF_NONE && lane == 0) cnt[o.chunk] = (uint32_t)o.pos;
    uint32_t c = 0;
    if (lane == 0) c = atomicAdd(alloc, 1u);
    o.chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    o.pos = 0;
    over = o.chunk >= soft_chunks;
  }
  const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  if (has) {
    uint32_t *dst = data + (size_t)o.chunk * (FIELDS * WF_CHUNK) + (uint32_t)(o.pos + rank);
#pragma unroll
    for (int i = 0; i < FIELDS; i++) dst[i * WF_CHUNK] = v[i];
  }
  o.pos += n;
  return over;
}

__device__ __forceinline__ void wf_close(uint32_t *cnt, const WfOut &o) {
  if (o.chunk != WF_NONE && lane_now() == 0) cnt[o.chunk] = (uint32_t)o.pos;
}

// A consumer kernel's last wave resets the control words of the queue it read, so that the next producer starts from
// zero without a launch in between (every other wave has finished with them: it added to WF_DONE_WAVES after its last use).
__device__ __forceinline__ void wf_finish_consumer(uint32_t *ctl, int n_waves, int w_alloc, int w_head) {
  if (lane_now() == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t done = atomicAdd(&ctl[WF_DONE_WAVES * WF_CTL_STRIDE], 1u);
    if (done + 1u == (uint32_t)n_waves) {
      __hip_atomic_store(&ctl[w_alloc * WF_CTL_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctl[w_head * WF_CTL_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&ctl[WF_DONE_WAVES * WF_CTL_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// =====================================================================================================================
// camera kernel
// =====================================================================================================================
template <int WAVES, bool LDSN, int MIN_WAVES_PER_SIMD, bool SHORT_DIV>
__global__ __launch_bounds__(WAVES * 64, MIN_WAVES_PER_SIMD) void rt_wf_camera_kernel(RT_KParams P) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_lds = LDSN ? P.n_lds_nodes : 0;
  const float4 *lds_nodes = smem;
  const int perm_f4 = (P.depth > 0 ? P.depth : 1) * 16;
  float4 *wave_base = smem + n_lds * RT_LDS_NODE_F4 + wave * (perm_f4 + 96);
  uint32_t *perm = reinterpret_cast<uint32_t *>(wave_base);
  unsigned long long *acc = reinterpret_cast<unsigned long long *>(wave_base + perm_f4);
  // (the perm row of the deepest node level is never written: it holds the tile's camera-ray pyramid and the cull-mask cache)
  const int acc_off = (n_lds * RT_LDS_NODE_F4 + __builtin_amdgcn_readfirstlane(wave) * (perm_f4 + 96) + perm_f4) * 16;
  const int pyr_off = (n_lds * RT_LDS_NODE_F4 + __builtin_amdgcn_readfirstlane(wave) * (perm_f4 + 96) + perm_f4 - 16) * 16;

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

  uint32_t w_paths = 0, w_rays = 0, w_nodes = 0, w_leaves = 0, w_bgs = 0;
  const int shift = P.chunk_shift;
  const int unit_paths = 2 << shift;
  const uint32_t n_chunks_tile = (uint32_t)P.n_chunks_tile;
  const int leaf_level = P.depth - 1;
  const int thresh = P.sched_thresh;
  const int pyr_nodes = LDSN ? P.pyr_nodes : 0;
  const int wave_id = (int)blockIdx.x * WAVES + wave;

  bool queue_open = true, stopped = false;
  int  steal_tries = 0;
  WfOut out;
  out.chunk = WF_NONE;
  out.pos = WF_CHUNK;

  for (;;) {
    // ---------------- take a tile: own one from the queue, or join one that still has units ----------------
    if (stopped) break;
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
        if (lane == 0) {
          uint32_t old = atomicMax(&A->tile_next[tile_idx], n_chunks_tile);
          if (old < n_chunks_tile) atomicSub(&A->open_groups[tile_idx >> 6], 1u);
        }
        continue;
      }
    }
    // ---- the tile's camera-ray pyramid (tile-stream kernel: same construction, same margin) ----
    bool tile_root_miss = false;
    rt_v3 cam_o;
    {
      RT_KArgs A = cold_args();
      cam_o = rt_v3_make(A->cam[0][3], A->cam[1][3], A->cam[2][3]);
    }
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
      const rt_v3 o = cam_o;
      rt_v3 pn[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        rt_v3 n = rt_v3_cross(c[q], c[(q + 1) & 3]);
        if (rt_v3_dot(n, c[(q + 2) & 3]) > 0.0f) n = rt_v3_scale(n, -1.0f);      // outward: the opposite corner is inside
        pn[q] = n;
      }
      float *pyr = lds_at(smem, pyr_off);
      if (lane == 0) {
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
        bool outside = empty;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          rt_v3 n = pn[q];
          float lox = n.x * lo.x, hix = n.x * hi.x, loy = n.y * lo.y, hiy = n.y * hi.y, loz = n.z * lo.z, hiz = n.z * hi.z;
          float nearest = fminf(lox, hix) + fminf(loy, hiy) + fminf(loz, hiz);
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

    // ---------------- per-lane state: a camera ray and its traversal; no path state ----------------
    int   phase = PH_NEED;
    int   pix = 0, smp = 0;
    Ray3  ray;
    ray_setup(ray, cam_o, rt_v3_make(0, 0, 1));
    int   level = -1, node = 0, child = 0;
    uint32_t cur = 0, dirty = 0, live = 0;
    HitRec hit;
    hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;

    bool tile_open = true;
    int  c_next = 0, c_end = 0, c_x0 = 0, c_y = 0, c_pix0 = 0, c_s0 = 0;
    uint32_t u_cur = 0, u_end = 0, grab = queue_open ? (uint32_t)cold_args()->grab_max : 1u;
    bool took_any = false;

    for (;;) {
      // ================= S: environment for the misses, hits to the queue, new camera paths =================
      {
        RT_KArgs A = cold_args();
        bool  done = false, start = false;
        rt_v3 radiance = rt_v3_make(0, 0, 0);
        if (phase == PH_MISS) {
          ShadeParams SP;
          SP.tris = nullptr; SP.mats = nullptr; SP.textures = A->textures; SP.texels = A->texels;
          SP.bg_texture = A->bg_texture; SP.max_bounces = 0;
          // cast_ray's `background * tint + emission` (raytracer.c:554) with tint = 1, emission = 0: x * 1 + 0 is x for every
          // x the quantisation can tell from zero
          radiance = background_lookup(SP, ray.d);
          done = true;
        }
        w_bgs += (uint32_t)__popcll(__ballot(done));
        {
          const uint32_t rec[WF_HIT0_FIELDS] = {(uint32_t)as_i(ray.d.x), (uint32_t)as_i(ray.d.y), (uint32_t)as_i(ray.d.z),
                                                (uint32_t)as_i(hit.t), (uint32_t)hit.tri, (uint32_t)as_i(hit.u), (uint32_t)as_i(hit.v),
                                                (uint32_t)(tile_x0 + (pix & 7)) | ((uint32_t)(tile_y0 + (pix >> 3)) << 16), (uint32_t)smp};
          const bool is_hit = phase == PH_HIT;
          if (wf_append<WF_HIT0_FIELDS>(A->wf_hit0, A->wf_cnt_hit0, A->wf_ctl + WF_HIT0_ALLOC * WF_CTL_STRIDE, out, is_hit, rec,
                                        (uint32_t)A->wf_soft_chunks)) {
            // the hit queue is (nearly) full: finish the unit at hand, take no more; the host runs the bounces and
            // launches this kernel again -- tile_next / work_head keep the position
            if (!stopped && lane == 0) __hip_atomic_store(A->wf_ctl + WF_STOPPED * WF_CTL_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stopped = true;
            queue_open = false;
          }
          if (is_hit) phase = PH_NEED;
        }
        if (done) {
          unsigned long long *ap = reinterpret_cast<unsigned long long *>(lds_at(smem, acc_off) + pix * 6);
          atomicAdd(ap + 0, accum_quantize_dev(radiance.x));
          atomicAdd(ap + 1, accum_quantize_dev(radiance.y));
          atomicAdd(ap + 2, accum_quantize_dev(radiance.z));
          phase = PH_NEED;
        }

        // ---- regeneration: idle lanes take the next paths of the tile, across unit boundaries ----
        rt_v3 dir = ray.d;
        if (tile_open) {
          unsigned long long need = __ballot(phase == PH_NEED);
          bool got = false;
          int  gx = 0, gy = 0, gs = 0, gp = 0;
          const int width = A->width, sample_end = A->sample_end;
          const int max_bounces = A->max_bounces;
          while (need) {
            if (c_next >= c_end) {
              if (u_cur + 1u < u_end) {
                u_cur += 1u;
              } else {
                if (stopped) { tile_open = false; break; }
                uint32_t u0 = 0;
                if (lane == 0) u0 = atomicAdd(&A->tile_next[tile_idx], grab);
                u0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)u0);
                if (u0 >= n_chunks_tile) { tile_open = false; break; }
                u_cur = u0;
                u_end = u0 + grab < n_chunks_tile ? u0 + grab : n_chunks_tile;
                if (u_end == n_chunks_tile && lane == 0) atomicSub(&A->open_groups[tile_idx >> 6], 1u);
                const uint32_t left = n_chunks_tile - u_end;
                const uint32_t gmax = (uint32_t)A->grab_max;
                grab = left >= 8u * gmax ? gmax : (left >= 8u && gmax >= 2u ? 2u : 1u);
                took_any = true;
              }
              const uint32_t grp = u_cur >> 2, pair = u_cur & 3u;
              const uint32_t n_sb = (uint32_t)A->n_sample_blocks;
              const uint32_t row = grp / n_sb, sb = grp - row * n_sb;
              c_x0 = tile_x0 + (int)pair * 2;
              c_y = tile_y0 + (int)row;
              c_pix0 = (int)row * 8 + (int)pair * 2;
              c_s0 = A->sample_first + (int)(sb << shift);
              c_next = 0;
              c_end = (c_y < A->height && c_x0 < width) ? unit_paths : 0;
              continue;
            }
            const int n_need = (int)__popcll(need);
            const int avail = c_end - c_next;
            const int take = n_need < avail ? n_need : avail;
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            bool valid = false;
            if (phase == PH_NEED && !got && rank < take) {
              int k = c_next + rank;
              int px = k >> shift;
              int sm = c_s0 + (k & ((1 << shift) - 1));
              int x = c_x0 + px;
              if (x < width && sm < sample_end) {
                valid = true;
                if (max_bounces > 0) { got = true; gx = x; gy = c_y; gs = sm; gp = c_pix0 + px; }
                // max_bounces == 0: the path exists and is black (the loop of raytracer.c:512 runs zero times)
              }
            }
            w_paths += (uint32_t)__popcll(__ballot(valid));
            c_next += take;
            need = __ballot(phase == PH_NEED && !got);
          }
          if (got) {
            pix = gp;
            smp = gs;
            PrimaryParams PP;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
              for (int j = 0; j < 4; j++) PP.cam[i][j] = A->cam[i][j];
            PP.focal_length = A->focal_length; PP.inv_width = A->inv_width; PP.inv_height = A->inv_height; PP.aspect = A->aspect;
            rt_v3 org;
            primary_ray(PP, gx, gy, gs, org, dir);
            start = true;
          }
        }
        bool skip_root = false;
        if (start) {
          ray_setup<SHORT_DIV>(ray, cam_o, dir);
          hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;
          dirty = 0;
          live = 0;
          cur = 0;
          level = -1;
          node = 0;
          child = (leaf_level >= 0) ? 0 : P.last_row_offset;
          phase = (leaf_level >= 0) ? PH_NODE : PH_LEAF;
          skip_root = tile_root_miss && ray.fast;       // its one node visit finds no candidate
          if (skip_root) phase = PH_MISS;
        }
        w_rays += (uint32_t)__popcll(__ballot(start));
        w_nodes += (uint32_t)__popcll(__ballot(skip_root));
      }

      const int n_trav0 = (int)__popcll(__ballot(phase == PH_NODE || phase == PH_LEAF));
      if (n_trav0 == 0) {
        if (__any(phase == PH_MISS)) continue;
        if (!tile_open) break;
        continue;
      }

      // ================= traversal: NODE / LEAF blocks until `thresh` lanes wait for S =================
      traversal_blocks<LDSN, SHORT_DIV, true>(P, smem, lds_nodes, perm, lane, n_lds, pyr_nodes, pyr_off, leaf_level, thresh,
                                              n_trav0, ray, true, phase, level, node, child, cur, dirty, live, hit, w_nodes, w_leaves);
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
      steal_tries = 0;
    }
  }
  RT_KArgs A = cold_args();
  wf_close(A->wf_cnt_hit0, out);
  unsigned long long *counters = A->counters;
  if (lane == 0) {
    atomicAdd(counters + CNT_PATHS, (unsigned long long)w_paths);
    atomicAdd(counters + CNT_RAYS, (unsigned long long)w_rays);
    atomicAdd(counters + CNT_NODES, (unsigned long long)w_nodes);
    atomicAdd(counters + CNT_LEAVES, (unsigned long long)w_leaves);
    atomicAdd(counters + CNT_BG, (unsigned long long)w_bgs);
  }
}

// =====================================================================================================================
// trace kernel: closest hits of the continuation rays of one bounce
// =====================================================================================================================
template <int WAVES, bool LDSN, int MIN_WAVES_PER_SIMD, bool SHORT_DIV>
__global__ __launch_bounds__(WAVES * 64, MIN_WAVES_PER_SIMD) void rt_wf_trace_kernel(RT_KParams P) {
  extern __shared__ float4 smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_lds = LDSN ? P.n_lds_nodes : 0;
  const float4 *lds_nodes = smem;
  const int perm_f4 = (P.depth > 0 ? P.depth : 1) * 16;
  float4 *wave_base = smem + n_lds * RT_LDS_NODE_F4 + wave * (perm_f4 + 96);
  uint32_t *perm = reinterpret_cast<uint32_t *>(wave_base);
  unsigned long long *acc = reinterpret_cast<unsigned long long *>(wave_base + perm_f4);
  const int acc_off = (n_lds * RT_LDS_NODE_F4 + __builtin_amdgcn_readfirstlane(wave) * (perm_f4 + 96) + perm_f4) * 16;

  if (LDSN) {
    const float4 *g = reinterpret_cast<const float4 *>(P.nodes);
    for (int i = threadIdx.x; i < n_lds * 12; i += WAVES * 64) {
      int nd = i / 12, q = i - nd * 12;
      smem[nd * RT_LDS_NODE_F4 + q] = g[i];
    }
    __syncthreads();
  }
  // Finished paths add to an 8x8-pixel accumulator tile in LDS, flushed to the frame when the wave moves on to rays of
  // another tile: a chunk of the ray queue descends from the hits ONE camera wave wrote, i.e. from one or two tiles.
  // (64-bit atomics on the frame execute at the memory side -- per lane to scattered addresses they were measured at
  // half of this kernel's time -- and stay in the in-order vmcnt queue in front of every later load.)
  acc[lane] = 0ull;
  acc[lane + 64] = 0ull;
  acc[lane + 128] = 0ull;
  uint32_t acc_key = 0xFFFFFFFFu;          // (tile_y << 13) | tile_x of the pixels the LDS tile stands for

  uint32_t w_rays = 0, w_nodes = 0, w_leaves = 0, w_bgs = 0;
  const int leaf_level = P.depth - 1;
  const int thresh = P.sched_thresh;
  const int qi = (P.wf_bounce - 1) & 1;                      // rays of bounce b were written by the shade kernel of bounce b - 1
  const int w_in_alloc = WF_RAY0_ALLOC + 2 * qi, w_in_head = WF_RAY0_HEAD + 2 * qi;

  uint32_t n_in;                                             // chunks of the input queue (all closed: the producer has finished)
  {
    RT_KArgs A = cold_args();
    n_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)A->wf_ctl[w_in_alloc * WF_CTL_STRIDE]);
  }
  bool input_open = true;
  uint32_t rc = 0;
  int rpos = 0, rcnt = 0;
  WfOut out;
  out.chunk = WF_NONE;
  out.pos = WF_CHUNK;

  int   phase = PH_NEED;
  uint32_t ray_idx = 0;
  Ray3  ray;
  ray_setup(ray, rt_v3_make(0, 0, 0), rt_v3_make(0, 0, 1));
  int   level = -1, node = 0, child = 0;
  uint32_t cur = 0, dirty = 0, live = 0;
  HitRec hit;
  hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;

  for (;;) {
    // ================= S: environment for the misses, hits to the queue, next rays =================
    {
      RT_KArgs A = cold_args();
      const uint32_t *rq = A->wf_ray[qi];
      const unsigned long long mMiss = __ballot(phase == PH_MISS);
      if (mMiss) {
        rt_v3 radiance = rt_v3_make(0, 0, 0);
        uint32_t pixel = 0;
        if (phase == PH_MISS) {
          const uint32_t *rr = rq + (size_t)(ray_idx >> 8) * (WF_RAY_FIELDS * WF_CHUNK) + (ray_idx & 255u);
          const rt_v3 tint = rt_v3_make(as_f((int)rr[6 * WF_CHUNK]), as_f((int)rr[7 * WF_CHUNK]), as_f((int)rr[8 * WF_CHUNK]));
          const rt_v3 emis = rt_v3_make(as_f((int)rr[9 * WF_CHUNK]), as_f((int)rr[10 * WF_CHUNK]), as_f((int)rr[11 * WF_CHUNK]));
          pixel = rr[13 * WF_CHUNK];
          ShadeParams SP;
          SP.tris = nullptr; SP.mats = nullptr; SP.textures = A->textures; SP.texels = A->texels;
          SP.bg_texture = A->bg_texture; SP.max_bounces = 0;
          const rt_v3 bg = background_lookup(SP, ray.d);
          radiance = rt_v3_mul_add(bg, tint, emis);         // raytracer.c:554
        }
        unsigned long long *frame = A->accum;
        if (frame) {
          const uint32_t key = ((pixel >> 19) << 13) | ((pixel & 0xFFFFu) >> 3);
          const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)__builtin_ctzll(mMiss));
          if (k0 != acc_key) {                                      // (wave-uniform) the tile in LDS goes to the frame
            if (acc_key != 0xFFFFFFFFu) {
              const int l = lane_now();
              const int x = (int)(acc_key & 8191u) * 8 + (l & 7), y = (int)(acc_key >> 13) * 8 + (l >> 3);
              unsigned long long *ap = reinterpret_cast<unsigned long long *>(lds_at(smem, acc_off) + l * 6);
              const unsigned long long r = ap[0], g = ap[1], b = ap[2];
              ap[0] = 0ull; ap[1] = 0ull; ap[2] = 0ull;
              const int width = A->width;
              if (x < width && y < A->height && (r | g | b) != 0ull) {
                unsigned long long *dst = frame + ((size_t)y * width + x) * 3;
                atomicAdd(dst + 0, r);
                atomicAdd(dst + 1, g);
                atomicAdd(dst + 2, b);
              }
            }
            acc_key = k0;
          }
          if (phase == PH_MISS) {
            const unsigned long long qr = accum_quantize_dev(radiance.x), qg = accum_quantize_dev(radiance.y), qb = accum_quantize_dev(radiance.z);
            if (key == acc_key) {
              const int pin = (int)(((pixel >> 16) & 7u) * 8u + (pixel & 7u));
              unsigned long long *ap = reinterpret_cast<unsigned long long *>(lds_at(smem, acc_off) + pin * 6);
              atomicAdd(ap + 0, qr);
              atomicAdd(ap + 1, qg);
              atomicAdd(ap + 2, qb);
            } else {                                                // a ray of another tile in the same block: straight to the frame
              unsigned long long *dst = frame + ((size_t)(pixel >> 16) * A->width + (pixel & 0xFFFFu)) * 3;
              if (qr) atomicAdd(dst + 0, qr);
              if (qg) atomicAdd(dst + 1, qg);
              if (qb) atomicAdd(dst + 2, qb);
            }
          }
        }
      }
      w_bgs += (uint32_t)__popcll(__ballot(phase == PH_MISS));
      {
        const uint32_t rec[WF_HIT_FIELDS] = {ray_idx, (uint32_t)as_i(hit.t), (uint32_t)hit.tri, (uint32_t)as_i(hit.u), (uint32_t)as_i(hit.v)};
        wf_append<WF_HIT_FIELDS>(A->wf_hit, A->wf_cnt_hit, A->wf_ctl + WF_HIT_ALLOC * WF_CTL_STRIDE, out, phase == PH_HIT, rec, 0xFFFFFFFFu);
      }
      if (phase == PH_MISS || phase == PH_HIT) phase = PH_NEED;

      bool start = false;
      rt_v3 org = ray.o, dir = ray.d;
      if (input_open) {
        unsigned long long need = __ballot(phase == PH_NEED);
        while (need) {
          if (rpos >= rcnt) {
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(A->wf_ctl + w_in_head * WF_CTL_STRIDE, 1u);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if (c >= n_in) { input_open = false; break; }
            rc = c;
            rcnt = (int)__builtin_amdgcn_readfirstlane((int)A->wf_cnt_ray[qi][c]);
            rpos = 0;
            continue;
          }
          const int n_need = (int)__popcll(need);
          const int avail = rcnt - rpos;
          const int take = n_need < avail ? n_need : avail;
          const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
          if (phase == PH_NEED && !start && rank < take) {
            ray_idx = rc * WF_CHUNK + (uint32_t)(rpos + rank);
            const uint32_t *rr = rq + (size_t)rc * (WF_RAY_FIELDS * WF_CHUNK) + (uint32_t)(rpos + rank);
            org = rt_v3_make(as_f((int)rr[0]), as_f((int)rr[1 * WF_CHUNK]), as_f((int)rr[2 * WF_CHUNK]));
            dir = rt_v3_make(as_f((int)rr[3 * WF_CHUNK]), as_f((int)rr[4 * WF_CHUNK]), as_f((int)rr[5 * WF_CHUNK]));
            start = true;
          }
          rpos += take;
          need = __ballot(phase == PH_NEED && !start);
        }
      }
      if (start) {
        ray_setup<SHORT_DIV>(ray, org, dir);
        hit.t = RT_INF; hit.tri = -1; hit.u = 0; hit.v = 0;
        dirty = 0;
        live = 0;
        cur = 0;
        level = -1;
        node = 0;
        child = (leaf_level >= 0) ? 0 : P.last_row_offset;
        phase = (leaf_level >= 0) ? PH_NODE : PH_LEAF;
      }
      w_rays += (uint32_t)__popcll(__ballot(start));
    }

    const int n_trav0 = (int)__popcll(__ballot(phase == PH_NODE || phase == PH_LEAF));
    if (n_trav0 == 0) {
      if (!input_open) break;
      continue;
    }

    traversal_blocks<LDSN, SHORT_DIV, false>(P, smem, lds_nodes, perm, lane, n_lds, 0, 0, leaf_level, thresh, n_trav0, ray, false,
                                             phase, level, node, child, cur, dirty, live, hit, w_nodes, w_leaves);
  }

  RT_KArgs A = cold_args();
  if (acc_key != 0xFFFFFFFFu && A->accum) {
    const int x = (int)(acc_key & 8191u) * 8 + (lane & 7), y = (int)(acc_key >> 13) * 8 + (lane >> 3);
    const unsigned long long r = acc[lane * 3 + 0], g = acc[lane * 3 + 1], b = acc[lane * 3 + 2];
    const int width = A->width;
    if (x < width && y < A->height && (r | g | b) != 0ull) {
      unsigned long long *dst = A->accum + ((size_t)y * width + x) * 3;
      atomicAdd(dst + 0, r);
      atomicAdd(dst + 1, g);
      atomicAdd(dst + 2, b);
    }
  }
  wf_close(A->wf_cnt_hit, out);
  unsigned long long *counters = A->counters;
  if (lane == 0) {
    atomicAdd(counters + CNT_RAYS, (unsigned long long)w_rays);
    atomicAdd(counters + CNT_NODES, (unsigned long long)w_nodes);
    atomicAdd(counters + CNT_LEAVES, (unsigned long long)w_leaves);
    atomicAdd(counters + CNT_BG, (unsigned long long)w_bgs);
  }
  wf_finish_consumer(A->wf_ctl, A->wf_n_waves, w_in_alloc, w_in_head);
}

// =====================================================================================================================
// shade kernel: one dense block of 64 accepted hits at a time
// =====================================================================================================================
template <bool FIRST>
__global__ __launch_bounds__(256, 4) void rt_wf_shade_kernel(RT_KParams P) {
  const int lane = threadIdx.x & 63;
  uint32_t *ctl = P.wf_ctl;
  const int w_in_alloc = FIRST ? WF_HIT0_ALLOC : WF_HIT_ALLOC, w_in_head = FIRST ? WF_HIT0_HEAD : WF_HIT_HEAD;
  const uint32_t *hq = FIRST ? P.wf_hit0 : P.wf_hit;
  const uint32_t *hcnt = FIRST ? P.wf_cnt_hit0 : P.wf_cnt_hit;
  const int qi = P.wf_bounce & 1;                            // output ray queue; hits of bounce b >= 1 point into queue (b - 1) & 1
  const uint32_t *rq_in = P.wf_ray[qi ^ 1];
  uint32_t *rq_out = P.wf_ray[qi];
  const uint32_t n_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctl[w_in_alloc * WF_CTL_STRIDE]);

  uint32_t w_shades = 0, w_tex = 0;
  uint32_t hc = 0;
  int hpos = 0, hcount = 0;
  bool input_open = true;
  WfOut out;
  out.chunk = WF_NONE;
  out.pos = WF_CHUNK;

  ShadeParams SP;
  SP.tris = P.tris; SP.mats = P.mats; SP.textures = P.textures; SP.texels = P.texels;
  SP.bg_texture = P.bg_texture; SP.max_bounces = P.max_bounces;

  for (;;) {
    // ---- 64 records, across chunk boundaries ----
    int have = 0;
    bool valid = false;
    uint32_t my_chunk = 0, my_slot = 0;
    while (have < 64) {
      if (hpos >= hcount) {
        if (!input_open) break;
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(&ctl[w_in_head * WF_CTL_STRIDE], 1u);
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        if (c >= n_in) { input_open = false; break; }
        hc = c;
        hcount = (int)__builtin_amdgcn_readfirstlane((int)hcnt[c]);
        hpos = 0;
        continue;
      }
      const int room = 64 - have, avail = hcount - hpos;
      const int take = room < avail ? room : avail;
      if (lane >= have && lane < have + take) {
        valid = true;
        my_chunk = hc;
        my_slot = (uint32_t)(hpos + (lane - have));
      }
      hpos += take;
      have += take;
    }
    if (have == 0) break;

    bool done = false, cont = false;
    rt_v3 org = rt_v3_make(0, 0, 0), dir = rt_v3_make(0, 0, 1), tint = rt_v3_make(1, 1, 1), emis = rt_v3_make(0, 0, 0);
    rt_v3 radiance = rt_v3_make(0, 0, 0);
    uint32_t rng = 0, pixel = 0;
    int bounce = 0;
    LaneCounters cn;
    cn.rays = cn.nodes = cn.leaves = cn.shades = cn.bgs = cn.textured = cn.paths = 0;
    if (valid) {
      HitRec hit;
      if (FIRST) {
        const uint32_t *hr = hq + (size_t)my_chunk * (WF_HIT0_FIELDS * WF_CHUNK) + my_slot;
        dir = rt_v3_make(as_f((int)hr[0]), as_f((int)hr[1 * WF_CHUNK]), as_f((int)hr[2 * WF_CHUNK]));
        hit.t = as_f((int)hr[3 * WF_CHUNK]); hit.tri = (int)hr[4 * WF_CHUNK];
        hit.u = as_f((int)hr[5 * WF_CHUNK]); hit.v = as_f((int)hr[6 * WF_CHUNK]);
        pixel = hr[7 * WF_CHUNK];
        const uint32_t sample = hr[8 * WF_CHUNK];
        org = rt_v3_make(P.cam[0][3], P.cam[1][3], P.cam[2][3]);
        rng = rt_path_seed(P.seed, (pixel & 0xFFFFu) + (pixel >> 16) * (uint32_t)P.width, sample);
      } else {
        const uint32_t *hr = hq + (size_t)my_chunk * (WF_HIT_FIELDS * WF_CHUNK) + my_slot;
        const uint32_t ri = hr[0];
        hit.t = as_f((int)hr[1 * WF_CHUNK]); hit.tri = (int)hr[2 * WF_CHUNK];
        hit.u = as_f((int)hr[3 * WF_CHUNK]); hit.v = as_f((int)hr[4 * WF_CHUNK]);
        const uint32_t *rr = rq_in + (size_t)(ri >> 8) * (WF_RAY_FIELDS * WF_CHUNK) + (ri & 255u);
        org = rt_v3_make(as_f((int)rr[0]), as_f((int)rr[1 * WF_CHUNK]), as_f((int)rr[2 * WF_CHUNK]));
        dir = rt_v3_make(as_f((int)rr[3 * WF_CHUNK]), as_f((int)rr[4 * WF_CHUNK]), as_f((int)rr[5 * WF_CHUNK]));
        tint = rt_v3_make(as_f((int)rr[6 * WF_CHUNK]), as_f((int)rr[7 * WF_CHUNK]), as_f((int)rr[8 * WF_CHUNK]));
        emis = rt_v3_make(as_f((int)rr[9 * WF_CHUNK]), as_f((int)rr[10 * WF_CHUNK]), as_f((int)rr[11 * WF_CHUNK]));
        rng = rr[12 * WF_CHUNK];
        pixel = rr[13 * WF_CHUNK];
        bounce = (int)rr[14 * WF_CHUNK];
      }
      done = shade_hit(SP, hit, org, dir, tint, emis, rng, bounce, cn, radiance);
      cont = !done;
    }
    w_shades += (uint32_t)__popcll(__ballot(cn.shades != 0));
    w_tex += (uint32_t)__popcll(__ballot(cn.textured != 0));
    if (done && P.accum) {
      unsigned long long *dst = P.accum + ((size_t)(pixel >> 16) * P.width + (pixel & 0xFFFFu)) * 3;
      const unsigned long long qr = accum_quantize_dev(radiance.x), qg = accum_quantize_dev(radiance.y), qb = accum_quantize_dev(radiance.z);
      if (qr) atomicAdd(dst + 0, qr);
      if (qg) atomicAdd(dst + 1, qg);
      if (qb) atomicAdd(dst + 2, qb);
    }
    const uint32_t rec[WF_RAY_FIELDS] = {(uint32_t)as_i(org.x), (uint32_t)as_i(org.y), (uint32_t)as_i(org.z),
                                         (uint32_t)as_i(dir.x), (uint32_t)as_i(dir.y), (uint32_t)as_i(dir.z),
                                         (uint32_t)as_i(tint.x), (uint32_t)as_i(tint.y), (uint32_t)as_i(tint.z),
                                         (uint32_t)as_i(emis.x), (uint32_t)as_i(emis.y), (uint32_t)as_i(emis.z),
                                         rng, pixel, (uint32_t)bounce};
    wf_append<WF_RAY_FIELDS>(rq_out, P.wf_cnt_ray[qi], ctl + (WF_RAY0_ALLOC + 2 * qi) * WF_CTL_STRIDE, out, cont, rec, 0xFFFFFFFFu);
  }

  wf_close(P.wf_cnt_ray[qi], out);
  if (lane == 0) {
    atomicAdd(P.counters + CNT_SHADES, (unsigned long long)w_shades);
    atomicAdd(P.counters + CNT_TEXTURED, (unsigned long long)w_tex);
  }
  wf_finish_consumer(ctl, P.wf_n_waves, w_in_alloc, w_in_head);
}

// =====================================================================================================================
// launchers
// =====================================================================================================================
template <typename K>
static int wf_set_lds(K kernel, int smem_bytes) {
  if (smem_bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

// geometry: 0 = one 16-wave workgroup per CU (4 waves per SIMD, the whole tree in LDS when it fits); 1 = two 12-wave
// workgroups per CU (6 per SIMD at <= 80 VGPRs); 2 = two 10-wave workgroups (5 per SIMD at <= 96 VGPRs).  With two
// workgroups per CU each holds as much of the top of the tree as half the LDS allows.
#define WF_LAUNCH(KERNEL, W, M)                                                                                          \
  do {                                                                                                                   \
    if (P->short_div) {                                                                                                  \
      if ((rc = wf_set_lds(KERNEL<W, true, M, true>, smem_bytes))) return rc;                                            \
      hipLaunchKernelGGL((KERNEL<W, true, M, true>), dim3(n_blocks), dim3(W * 64), smem_bytes, stream, *P);              \
    } else {                                                                                                             \
      if ((rc = wf_set_lds(KERNEL<W, true, M, false>, smem_bytes))) return rc;                                           \
      hipLaunchKernelGGL((KERNEL<W, true, M, false>), dim3(n_blocks), dim3(W * 64), smem_bytes, stream, *P);             \
    }                                                                                                                    \
  } while (0)

// (geometries 1 and 2 were measured slower, profiles/r03_experiments.md: they exist in the diagnostic build only)
extern "C" int rt_wf_launch_camera(const RT_KParams *P, int n_blocks, int geometry, int smem_bytes, hipStream_t stream) {
  int rc;
#ifdef RT_DIAG_VARIANTS
  if (geometry == 1) WF_LAUNCH(rt_wf_camera_kernel, 12, 6);
  else if (geometry == 2) WF_LAUNCH(rt_wf_camera_kernel, 10, 5);
  else
#endif
  WF_LAUNCH(rt_wf_camera_kernel, 16, 4);
  (void)geometry;
  return (int)hipGetLastError();
}

extern "C" int rt_wf_launch_trace(const RT_KParams *P, int n_blocks, int geometry, int smem_bytes, hipStream_t stream) {
  int rc;
#ifdef RT_DIAG_VARIANTS
  if (geometry == 1) WF_LAUNCH(rt_wf_trace_kernel, 12, 6);
  else if (geometry == 2) WF_LAUNCH(rt_wf_trace_kernel, 10, 5);
  else
#endif
  WF_LAUNCH(rt_wf_trace_kernel, 16, 4);
  (void)geometry;
  return (int)hipGetLastError();
}

extern "C" int rt_wf_launch_shade(const RT_KParams *P, int n_blocks, int first, hipStream_t stream) {
  if (first) hipLaunchKernelGGL(rt_wf_shade_kernel<true>, dim3(n_blocks), dim3(256), 0, stream, *P);
  else hipLaunchKernelGGL(rt_wf_shade_kernel<false>, dim3(n_blocks), dim3(256), 0, stream, *P);
  return (int)hipGetLastError();
}
