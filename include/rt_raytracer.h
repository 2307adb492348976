/* rt_raytracer.h -- the drop-in boundary of the render hot path.
 *
 * These are the entry points the reference declares in raytracer.h:51-56 and
 * that its driver binds (driver.c:793-818).  librt_hip.so exports them with the
 * same names, argument meaning and completion protocol; the work behind them
 * runs on the MI355X (raytracing_c_amd/csrc/rt_kernels.hip).  There is no CPU
 * implementation behind these symbols: when no HIP device or no device code is
 * available they report through rt_last_error() and leave the image untouched.
 */
#ifndef RT_RAYTRACER_H
#define RT_RAYTRACER_H

#include "rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference raytracer.h:23-26 */
typedef struct {
  Vec3 position;
  Vec3 direction;
} Ray;

/* reference raytracer.h:28-33 (88 bytes) */
typedef struct {
  f32    distance;
  Vec3   normal, normal_geo, point, tangent, bitangent;
  Vec2   tex_coords;
  Shader shader;
} Hit;

/* `_Atomic i32` in the reference; C++ translation units see a plain i32 of the
 * same size and alignment and touch it only through __atomic builtins. */
#ifdef __cplusplus
typedef i32 rt_atomic_i32;
#else
typedef _Atomic i32 rt_atomic_i32;
#endif

/* reference raytracer.h:44-49.  `image` is embedded by value and shares the
 * caller's pixel storage (u8, components >= 3); the caller presets n_threads to
 * the number of threads it will start on render_thread_proc (driver.c:793-803)
 * and zero-initialises _current_chunk. */
typedef struct {
  Image         image;
  Scene        *scene;
  isize         samples, max_bounces;
  rt_atomic_i32 n_threads, _current_chunk;
} Rendering_Context;

/* reference raytracer.c:596-720.  May be entered by any number of threads on
 * the same context.  The entrant whose claim of _current_chunk returns 0 owns
 * the frame: it uploads (or re-uses) the device copy of ctx->scene, renders all
 * chunks on the GPU, writes ctx->image.pixels, then stores n_chunks into
 * _current_chunk.  Every entrant decrements n_threads on its way out and the
 * owner does so last, so n_threads == 0 implies a complete image. */
extern void render_thread_proc(Rendering_Context *context);

/* reference raytracer.c:786-788 */
extern bool rendering_context_is_finished(Rendering_Context *context);

/* reference raytracer.c:790-794 */
extern void rendering_context_finish(Rendering_Context *context);

/* reference raytracer.c:722-784: UV-space light baking, the second caller of the
 * path loop (the reference driver never invokes it).  Rasterises every triangle in
 * UV space on the GPU and bakes `samples` cosine-weighted 8-bounce paths per texel
 * into the u8 lightmap (host memory).  Texels covered by several triangles keep the
 * last triangle's value, texels outside the image are skipped; errors: rt_last_error(). */
extern void lightmap_bake(Image const *lightmap, Scene const *scene, isize samples);

/* reference denoiser.h / denoiser.c:131-153: 3x3 luminance-sorted median,
 * blended by neighbourhood noisiness, on u8 images in HOST memory (what driver.c:827-837 calls after
 * the render).  Runs as one HIP kernel; n_threads is accepted and ignored.  Errors: rt_last_error(). */
extern void denoise_image(Image const *src, Image const *dst, isize n_threads);

/* Blocking convenience wrapper named by BASELINE.json's north_star; not part of
 * the reference.  Returns 0 on success, -1 on error. */
extern int render(Scene *scene, Image *image, isize samples, isize max_bounces);

#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
static_assert(sizeof(Hit) == 88, "Hit must be 88 bytes");
#else
_Static_assert(sizeof(Hit) == 88, "Hit must be 88 bytes");
#endif

#endif /* RT_RAYTRACER_H */
