/* rt_materials.h -- the material table side of the boundary.
 *
 * In the reference, shading is a host callback: cast_ray calls
 * hit.shader.proc(hit.shader.data, &in, &out) (raytracer.c:535) and
 * scene->background.proc(data, dir) (raytracer.c:554); the procs and the
 * PBR_Shader_Data layout are private to driver.c (driver.c:191-198,350-418,95).
 * A GPU kernel cannot call host function pointers, so this library makes the
 * three procs and the data layout public and recognises them BY ADDRESS while
 * flattening a Scene (rt_hip.h: rt_scene_upload):
 *
 *   shader.proc == disney_shader_proc -> device Disney BSDF on PBR_Shader_Data
 *   shader.proc == debug_shader_proc  -> device normal visualiser
 *   background.proc == (Background_Proc)sample_background
 *                                      -> device equirect lookup on an Image
 *
 * Any other proc makes the upload fail (rt_last_error()), never silently
 * render something else.  The exported functions themselves are address tokens:
 * calling them on the host only records an error.
 */
#ifndef RT_MATERIALS_H
#define RT_MATERIALS_H

#include "rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* reference driver.c:191-198 (80 bytes).  Image pointers may be NULL. */
typedef struct {
  Vec3   base_color, emission;
  f32    roughness, metalness, normal_map_strength, sheen, sheen_tint, anisotropic_strength;
  Image *texture_albedo;
  Image *texture_normal;
  Image *texture_metal_roughness;
  Image *texture_emission;
} PBR_Shader_Data;

/* reference driver.c:350-409 */
extern void disney_shader_proc(rawptr data, Shader_Input const *input, Shader_Output *output);

/* reference driver.c:411-418 */
extern void debug_shader_proc(rawptr data, Shader_Input const *input, Shader_Output *output);

/* reference driver.c:95-104; the driver stores it as
 * (Background_Proc)sample_background with data = Image* (driver.c:760-763). */
extern Color3 sample_background(Image const *image, Vec3 direction);

#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
static_assert(sizeof(PBR_Shader_Data) == 80, "PBR_Shader_Data must be 80 bytes");
#else
_Static_assert(sizeof(PBR_Shader_Data) == 80, "PBR_Shader_Data must be 80 bytes");
#endif

#endif /* RT_MATERIALS_H */
