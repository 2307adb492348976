/* rt_hip.h -- C-ABI of the MI355X render layer beyond the reference's own
 * entry points (rt_raytracer.h).  Plain pointers and sizes only.
 *
 * The reference has no device layer; these calls are what its driver would use
 * to control one (seed, device choice, error text, explicit scene residency)
 * and what a multi-process launcher needs to render a subset of the 32x32
 * chunks of raytracer.c:601-627 on each GPU.  INTEGRATION.md shows the
 * reference-side stubs.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include "rt_raytracer.h"
#include "rt_materials.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Text of the most recent failure in this process ("" when none).  The render
 * entry points are void in the reference (raytracer.h:51-56), so errors are
 * reported here and on stderr. */
extern char const *rt_last_error(void);
extern void        rt_clear_error(void);

/* Selects the HIP device for this process (default 0) and reads the library's configuration -- the environment
 * variables RT_DEVICES / RT_DEVICES_REHEARSE below, nothing else -- once.  0 on success. */
extern int rt_init(int device);

/* A frame behind render_thread_proc() / render() / rt_render_frame() is spread over the first `n_devices` GPUs of the
 * node (primary device, then the following ones): every GPU renders the chunks rt_chunk_owner() gives its rank from its
 * own copy of the scene, the compact u8 tiles go to the primary GPU with one peer copy per device, which untiles and
 * fills the caller's pixels (driver.c:793-818 needs no change: its `-T n` threads enter as before, one of them drives
 * all GPUs).  Default 1, or the environment variable RT_DEVICES read by rt_init().  `rehearse` != 0 (RT_DEVICES_REHEARSE=1)
 * maps all n logical devices onto the primary GPU -- the N-device code path on a one-GPU machine, for tests.
 * rt_device_count() = how many GPUs the next frame will use (min(n_devices, GPUs present)). */
extern int rt_set_devices(i32 n_devices, i32 rehearse);
extern i32 rt_device_count(void);


/* Frame seed of the per-path RNG rule rt_path_seed() (rt_math.h); replaces
 * `random_state = time_now()` of raytracer.c:597.  Default 0x1234ABCD. */
extern void rt_set_seed(u32 seed);
extern u32  rt_get_seed(void);

/* ---- scene residency ------------------------------------------------------- */

typedef struct RT_Device_Scene RT_Device_Scene;

/* Flattens a host Scene into HBM: BVH nodes, per-leaf SoA tiles, trimmed AoS
 * records with material ids, the material table, and every Image referenced by
 * a material or by the background (RGB8/RGBA8 -> RGBA8).  Fails (NULL +
 * rt_last_error) on a shader/background proc that is not one of the exported
 * tokens of rt_materials.h.  render_thread_proc() calls this itself and caches
 * the result per Scene*; rt_scene_invalidate() drops that cache entry after the
 * host Scene was modified. */
extern RT_Device_Scene *rt_scene_upload(Scene const *scene);
extern void             rt_scene_release(RT_Device_Scene *dscene);
extern void             rt_scene_invalidate(Scene const *scene);
/* The reference reads the live Scene every frame (raytracer.c:596-612); a frame here renders from the cached device copy.
 * What keeps the two equal, per frame behind render_thread_proc() / render() / rt_render_frame():
 *  1. BEFORE the frame is enqueued (tens of microseconds): dimensions and base pointers of the host Scene, its material
 *     records and the descriptors of their Images in full, a bounded sample of every block of geometry and texel bytes;
 *  2. WHILE the GPU renders, on the calling thread: every byte of the BVH, the coordinate arrays, the AoS records and the
 *     materials, the texels of images above 64 KB one word in 61.  If that differs from what was uploaded, the frame is
 *     discarded, the scene uploaded again and the frame rendered again: an in-place edit of vertices, boxes or materials is
 *     seen by the next frame like in the reference.  A frame of an unchanged scene waits for max(kernel, check), not for
 *     their sum (RT_Frame_Timing.verify_ms; helmet: ~0.5 ms of host time behind a kernel of 0.6 ms or more).
 *     rt_scene_set_static(scene, 1) switches (2) off for a host that never edits in place (or tells: rt_scene_touch).
 *  3. rt_scene_touch(scene, begin, bytes): "I wrote these bytes" -- nodes, coordinates, AoS records, a material record or
 *     texels of a texture / the background.  The block they belong to is patched on every device (a few texture rows, a few
 *     leaf tiles) instead of the whole scene being uploaded again, and the stamps are refreshed: also the way to get a
 *     single-texel edit seen that the 1-in-61 sampling of (2) can miss.  Returns 0 = patched in place, 1 = not a patchable
 *     range (the copies were dropped, the next frame uploads), -1 = error.
 *     A copy is patched only if the blocks that differ from what it was made from are the one(s) the range lies in; a block that
 *     changed without a word drops the copy instead (1) -- the unreported edit is not absorbed into the new reference.
 * Every device keeps the fingerprint of ITS OWN copy: a frame over N devices is checked device by device, so a copy that one
 * device made before an edit cannot pass because another device has been brought up to date since.
 * rt_scene_invalidate() drops the cached copies on every device (and a rt_scene_set_static() opt-out: a Scene rebuilt at the same
 * address starts checked); rt_scene_verify() is the comparison of (2) on demand, for every device: 1 = every cached copy still
 * matches the host scene, 0 = one did not (the stale ones are dropped; the next frame uploads), -1 = nothing cached on the
 * primary device for this Scene. */
extern int              rt_scene_verify(Scene const *scene);
extern int              rt_scene_touch(Scene const *scene, void const *begin, size_t bytes);
extern void             rt_scene_set_static(Scene const *scene, i32 is_static);
extern i64              rt_scene_device_bytes(RT_Device_Scene const *dscene);

/* Camera used by rt_render_accumulate() for an explicitly uploaded scene;
 * rt_scene_upload() captures scene->camera, this replaces it. */
extern int rt_set_camera(RT_Device_Scene *dscene, Camera const *camera);

/* scene_init() (rt_scene.h, reference scene.c:416-426) with the build done by GPU kernels: fills `scene` with the SAME
 * bytes scene_init() would -- same triangles in the same slots, same child boxes (the cut positions of the reference's
 * split depend on counts only, so the host plans them and the GPU runs the sorts, bounds and inserts level by level;
 * csrc/rt_build.hip).  Host memory comes from `allocator` as in scene_init.  0 on success. */
extern int scene_init_gpu(Scene *scene, Triangle_Slice src_triangles, Allocator allocator);

/* ---- rendering -------------------------------------------------------------- */

typedef struct {
  u64 paths;        /* camera paths started                               */
  u64 rays;         /* ray_scene_hit equivalents (raytracer.c:514)        */
  u64 node_visits;  /* 8-box slab tests (raytracer.c:452)                 */
  u64 leaf_visits;  /* 8-triangle tests (raytracer.c:476)                 */
  u64 shades;       /* material evaluations (raytracer.c:535)             */
  u64 backgrounds;  /* environment lookups (raytracer.c:554)              */
  u64 textured;     /* shades on a material with at least one texture     */
} RT_Counters;

typedef struct {
  i32 width, height;     /* image size                                          */
  i32 samples;           /* samples per pixel (Rendering_Context.samples)       */
  i32 max_bounces;       /* Rendering_Context.max_bounces                       */
  u32 seed;              /* frame seed                                          */
  i32 rank, world;       /* this process renders the chunks rt_chunk_owner() gives `rank` */
  i32 slab;              /* samples per work item (0 = library default)         */
  i32 flags;             /* RT_FLAG_*                                           */
  i32 sample_first;      /* this call traces samples [sample_first,             */
  i32 sample_count;      /*   sample_first + sample_count) of every pixel; 0 = all.
                            `samples` stays the pixel's total: the accumulation
                            buffer can be filled by several calls (progressive)
                            and resolved once                                   */
} RT_Render_Params;

enum {
  RT_FLAG_NONE = 0,
};

/* Image partition across the GPUs of a node.  The frame is cut into the reference's 32x32 chunks
 * (raytracer.c:601-610; chunk c = cx + cy * ceil(width / 32)); chunk (cx, cy) belongs to rank
 * (cx + B * cy) mod world, B = the integer coprime to `world` nearest to 0.618 * world, so that every
 * chunk column and row is spread over all ranks.  A rank numbers its chunks in ascending global order.
 * Pure host arithmetic (no GPU needed):
 *   rt_chunk_count           chunks of the image
 *   rt_chunk_owner           rank that owns `chunk`, -1 if out of range
 *   rt_local_chunk_count     chunks `rank` owns
 *   rt_max_local_chunk_count the largest of those over all ranks (= slots per rank in the gathered tile buffer)
 *   rt_local_chunk_list      writes up to `capacity` of rank's chunk indices to out, returns the count */
extern i32 rt_chunk_count(i32 width, i32 height);
extern i32 rt_chunk_owner(i32 width, i32 height, i32 world, i32 chunk);
extern i32 rt_local_chunk_count(i32 width, i32 height, i32 rank, i32 world);
extern i32 rt_max_local_chunk_count(i32 width, i32 height, i32 world);
extern i32 rt_local_chunk_list(i32 width, i32 height, i32 rank, i32 world, i32 *out, i32 capacity);

/* Renders this rank's chunks.  All pointers are DEVICE pointers owned by the
 * caller, `stream` is a hipStream_t (NULL = default stream); the call only
 * enqueues work.
 *   d_accum  : u64[height*width*3] fixed-point radiance sums (rt_math.h), must
 *              be zero on entry for the chunks this rank owns
 * 0 on success. */
extern int rt_render_accumulate(RT_Device_Scene *dscene, RT_Render_Params const *params,
                                void *d_accum, void *stream);

/* accum -> mean radiance -> clamp -> sRGB -> u8 (raytracer.c:700-716) for this
 * rank's chunks.
 *   d_tiles  : u8[n_local_chunks*32*32*3], chunk-major compact tiles, or NULL
 *   d_image  : u8[height*width*3] row-major image, or NULL
 *   d_linear : f32[height*width*3] mean radiance before clamp, or NULL */
extern int rt_resolve(RT_Render_Params const *params, void const *d_accum,
                      void *d_tiles, void *d_image, void *d_linear, void *stream);

/* Scatters gathered compact tiles of ALL ranks (rank-major:
 * [world][rt_max_local_chunk_count()][32*32*3]) into a row-major u8 image on the device. */
extern int rt_untile(i32 width, i32 height, i32 world, void const *d_all_tiles,
                     void *d_image, void *stream);

/* denoise_image() (rt_raytracer.h) on DEVICE buffers: u8[height*width*3] row-major -> same layout. */
extern int rt_denoise(i32 width, i32 height, void const *d_src, void *d_dst, void *stream);

/* Whole frame from host memory to host memory on one GPU: upload/cached scene,
 * accumulate, resolve, copy back.  pixels: u8[height*stride*components] as the
 * reference lays out Image; linear (optional): f32[height*width*3];
 * accum (optional): u64[height*width*3].  0 on success. */
extern int rt_render_frame(Scene const *scene, Image const *image, isize samples, isize max_bounces,
                           f32 *linear, u64 *accum);

/* Frames in flight.  A launch of the path kernel ends with 0.6 - 1.0 ms of thinning bounce chains whatever its size
 * (27 % of the reference driver's default frame, driver.c:733-742); a blocking render_thread_proc() / render() has nothing to
 * fill that with, a host that has a NEXT frame does: rt_frame_begin() enqueues the frame -- scene check, accumulator clear,
 * path kernel, resolve, on one of two internal lanes with a stream, buffers and launch state of its own -- and returns a
 * ticket (>= 0; -1 + rt_last_error() on failure, also when two frames are in flight already); rt_frame_end(ticket) waits
 * for that frame and fills the pixels of the Image given at begin (the header is copied at begin, the pixels must stay valid
 * until end).  begin(A), begin(B), end(A), begin(C), end(B) ... keeps two frames on the GPU: the second one's workgroups take
 * the CUs as the first one's leave them.  Same pixels as render() / rt_render_frame(), bit for bit (per-path seeds, order-free
 * sums); seed and camera are read at begin; rt_get_counters() / rt_get_frame_timing() after an end describe THAT frame.
 * The per-frame scene check is the blocking path's: the sampled stamp at begin, the full content check inside rt_frame_end()
 * while the GPU renders -- a host scene that no longer equals the copy the frame was rendered from is rendered again there,
 * synchronously.  Do not edit (or rt_scene_touch) a scene between the begin and the end of a frame that renders it.
 * rt_frame_end() waits WITHOUT the library's device lock: two host threads can each keep a frame in flight (a ticket is ended
 * by one thread, once).
 * With rt_set_devices(n > 1) the frame is rendered inside rt_frame_begin() over the n devices (that pipeline has its own
 * overlap) and rt_frame_end() returns its status. */
extern int rt_frame_begin(Scene const *scene, Image const *image, isize samples, isize max_bounces);
extern int rt_frame_end(int ticket);

/* Counters of the last rt_render_accumulate / rt_render_frame on this process
 * (read back synchronously; summed over the devices of a multi-device frame). */
extern int rt_get_counters(RT_Counters *out);
/* Of node_visits of the last rt_render_accumulate launch (single device): the visits that were counted but not executed -- the
 * ONE root visit of every camera path whose 8x8 tile's pixel pyramid misses every child box of the root (raytracer.c:459-472
 * finds no candidate for any of them; the kernel proves that per tile and skips the block).  The oracle counts them too. */
extern int rt_get_skipped_root_visits(u64 *out);

/* Where the time of the last frame behind render_thread_proc / render / rt_render_frame went, in milliseconds.
 * Host clock: stamp = the per-frame scene check, upload = the scene upload when one was needed, enqueue = launching the
 * frame, total = the whole call.  HIP events on the frame's stream: gpu_prep = accumulator clear + the preparation kernel,
 * gpu_path = the path kernel, gpu_resolve, gpu_copy = device-to-host copy of the image.  A multi-device frame reports the
 * SLOWEST device's prep / path / resolve times, gpu_copy = that device's tile copy to the primary GPU, the largest stamp /
 * upload / enqueue time of any device, and gather_ms. */
typedef struct {
  f32 stamp_ms, upload_ms, enqueue_ms, gpu_prep_ms, gpu_path_ms, gpu_resolve_ms, gpu_copy_ms, total_ms;
  f32 verify_ms;       /* the full content check of the host scene, on the calling thread WHILE the GPU renders (rt_scene_touch) */
  f32 gather_ms;       /* multi-device frames: waiting for the other devices' tiles + untile + copy to the caller's pixels      */
  i32 n_devices;       /* GPUs the frame was spread over                                                                         */
  i32 slowest_device;  /* multi-device frames: the slot whose prep / path / resolve / copy times are reported above              */
} RT_Frame_Timing;
extern int rt_get_frame_timing(RT_Frame_Timing *out);

/* GPU time of the most recent path-tracing kernel launch in milliseconds
 * (HIP events on the launch stream); negative if none.  Synchronises. */
extern f32 rt_last_kernel_ms(void);

/* Mean GPU time per path-kernel launch over the launches since
 * rt_kernel_timing_reset() (at most the last 256); *n_launches (optional)
 * receives how many were averaged.  Synchronises. */
extern void rt_kernel_timing_reset(void);
extern f32  rt_kernel_timing_mean_ms(i32 *n_launches);

/* Numeric contract the library's kernels were built with (include/rt_math.h): 3 = explicit fused multiply-add + rt_pow24() in
 * the sRGB decode (the product), 2 = -DRT_MATH_V2 (round 4's arithmetic, librt_hip_v2.so), 1 = -DRT_MATH_NO_FMA (round 3's,
 * librt_hip_v1.so) -- the A/B partners.  The CPU checker must be built for the same one. */
extern int rt_math_contract(void);

/* Unit-level device entry points (rt_test_*), the wavefront pipeline switch and the kernel-generation knobs live in the
 * DIAGNOSTIC library only: include/rt_hip_diag.h, librt_hip_diag.so. */

#ifdef __cplusplus
}
#endif

#endif /* RT_HIP_H */
