/* rt_hip_diag.h -- entry points of the DIAGNOSTIC library librt_hip_diag.so (make -C raytracing_c_amd/csrc diag,
 * -DRT_DIAG_VARIANTS) that the product library librt_hip.so does NOT export: unit-level device entry points for the parity
 * tests, the wavefront pipeline (built and measured in round 3, slower than the tile-stream kernel on every BASELINE
 * configuration: profiles/r03_experiments.md), per-block statistics of the diagnostic kernel generations.  The diagnostic
 * library is built from the same sources and exports everything rt_hip.h declares as well; tests load it beside the product
 * library (raytracing_c_amd.native.diag).  Nothing here is part of the drop-in boundary.
 */
#ifndef RT_HIP_DIAG_H
#define RT_HIP_DIAG_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Make this library recognise ANOTHER library's material / background tokens (rt_materials.h) in Shader.proc /
 * Background.proc: a test process that has the product library mapped as well builds its scenes once, with the product's
 * token addresses.  NULL keeps the current one. */
extern void rt_diag_set_tokens(void *disney_proc, void *debug_proc, void *background_proc);

/* Fault injection for the multi-device frame (tests/test_gpu_multi_device.py): no_peer != 0 = treat every device as if peer
 * access to the primary GPU had been refused (its tiles are staged through pinned host memory); failing_slot >= 0 = that
 * device slot fails its part of the next frames (-1 = none). */
extern void rt_diag_multi_fault(i32 no_peer, i32 failing_slot);

/* 0 (default): the tile-stream path kernel.  1: the wavefront pipeline (camera / shade / trace kernels joined by record
 * queues in HBM, csrc/rt_wavefront.hip): the same images and counters bit for bit, measured slower;
 * selectable for measurements.  rt_set_wavefront_capacity(): camera-ray hits its queues hold per pass (default 96 M). */
extern int  rt_set_pipeline(i32 pipeline);
extern i32  rt_get_pipeline(void);
extern void rt_set_wavefront_capacity(i64 records);

/* Diagnostic kernel only (env RT_KERNEL=4): out[0..15] = 8 pairs (times a block ran, lanes it ran with) for
 * shade, environment, regenerate, leaf (uniform), leaf (per lane), node (uniform), node (per lane), pop;
 * out[16..23] = shader-clock cycles the waves spent in S blocks with shading (16), S blocks without (17), leaf
 * blocks (19), node blocks (21), pop loops (23), summed over waves; out[24] = cycles of the whole wave loops. */
extern int rt_get_sched_stats(u64 out[32]);
/* Block ledger of a -DRT_LEDGER build of the tile-stream kernel (tools/exp_ledger.py): out[0 .. n) = the LG_* slots of
 * csrc/rt_dev.hip.h of the last launch (all zero in other builds). */
extern int rt_get_ledger(u64 *out, i32 n);
/* ... and per wave (start tick, end tick, items) of the last diagnostic launch; ticks are 10 ns.  Returns the wave count. */
extern int rt_get_wave_times(u64 *out, i32 max_waves);


/* ---- unit-level device entry points (parity tests call the same device
 * functions the render kernel uses) ------------------------------------------ */

/* rt_math.h on the GPU; op codes as oracle_math() (oracle/oracle.h).  Host
 * pointers. */
extern int rt_test_math(i32 op, i32 n, f32 const *x, f32 const *y, f32 *out);
/* the leaf blocks' four-instruction reciprocal against IEEE 1.0f / x over all 2^32 bit patterns: out[0] differing patterns
 * inside its domain (0 expected), out[1] patterns outside the domain, out[2] differing ones among those, out[3] first
 * differing pattern inside the domain + 1; the form without the fix-up that the leaf blocks use (rcp_leaf): out[4] finite non-zero
 * patterns below 2^102 that differ (0 expected), out[5] patterns 0 / infinity / NaN whose result is not NaN (0 expected) */
extern int rt_test_rcp_sweep(u64 out[6]);
/* the kernels' sRGB decode of a texture sample against rt_srgb_to_linear1() for every float in [0, 2] (and 4 M negative ones):
 * out[0] patterns compared, out[1] differing (0 expected), out[2] first differing pattern + 1 */
extern int rt_test_srgb_sweep(u64 out[3]);
/* the tile-stream kernel's fixed-point conversion of a sample against rt_accum_quantize() over all 2^32 bit patterns:
 * out[0] differing patterns (0 expected), out[1] first differing pattern + 1 */
extern int rt_test_quantize_sweep(u64 out[2]);

/* Closest hit of n rays (host arrays, 6 f32 per ray: origin, direction) against
 * an uploaded scene: out_t[n], out_tri[n] (-1 = miss), out_uv[2n]. */
extern int rt_test_trace(RT_Device_Scene *dscene, i32 n, f32 const *rays,
                         f32 *out_t, i32 *out_tri, f32 *out_uv);

/* The same through the PRODUCTION traversal: traversal_blocks() (csrc/rt_dev.hip.h) -- the NODE / LEAF / pop code the path
 * kernels run -- in the path kernel's launch geometry (16-wave workgroups, tree in LDS), lanes refilled from the ray list as
 * they finish so that blocks mix rays at different depths as a frame does.
 *   pyramid    NULL, or 19 floats: 4 outward plane normals at [4 q .. 4 q + 2], the rays' common origin at [16 .. 18]; every
 *              ray then counts as a camera ray of one tile and node blocks take the pyramid-culled form (node_enter_few)
 *              where the path kernel would.  The caller guarantees origin and planes hold for every ray.
 *   exit_lanes 1 .. 64: finished lanes that end a round of blocks (the path kernel uses 48)
 *   mode       0 = the kernel instance a frame of this scene would use, 1 = IEEE division in the leaf blocks,
 *              2 = nodes through L1 / L2 instead of the LDS copy
 *   visits     [0] += 8-box tests (raytracer.c:452), [1] += 8-triangle tests (raytracer.c:476) */
extern int rt_test_trace_stream(RT_Device_Scene *dscene, i32 n, f32 const *rays, f32 const *pyramid, i32 exit_lanes, i32 mode,
                                f32 *out_t, i32 *out_tri, f32 *out_uv, u64 visits[2]);

/* The visiting order the per-launch preparation kernel derives from per-tile costs (rays a tile needed in the previous
 * launch of the same view): order[0 .. n_tiles) = a permutation of the tiles, most expensive cost bucket first. */
extern int rt_test_tile_order(i32 n_tiles, u32 const *cost, u32 *order);

/* Bilinear fetch (driver.c:49-93) of n (u,v) pairs on texture `tex` of the
 * uploaded scene (index in upload order; -1 = background image). */
extern int rt_test_texture(RT_Device_Scene *dscene, i32 tex, i32 n, f32 const *uv, f32 *out_rgb);

#ifdef __cplusplus
}
#endif

#endif /* RT_HIP_DIAG_H */
