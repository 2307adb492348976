/* rt_model.h -- C model loading for a host of librt_hip.so: what driver.c:510-728 does with codin's obj.h / gltf.h /
 * stb_image (none of which are in the reference tree).  `.obj` (+ `.mtl`) and `.glb` / `.gltf` files become the
 * Triangle[] + PBR_Shader_Data[] + Image[] (+ Camera) that scene_init() and render_thread_proc() consume.
 *
 * Texels of image k of a model: the side file `<model path>.image<k>.rgb8` when it exists (16-byte header "RT8I", i32
 * width, height, components, then the rows; tools/extract_textures.py writes them from any codec PIL reads), else the
 * stream the glTF file embeds or names, decoded by rt_jpeg.c when it is a baseline or progressive JPEG (helmet.glb's four images:
 * libjpeg's default arithmetic restated, so both routes give the same bytes) or by rt_png.c when it is a PNG.  A
 * material that references an image neither route can produce fails the load with a message.  The map_Kd / map_Ke /
 * norm / map_Pm files of a .mtl are decoded the same way (a file that does not exist is no map, as in loaders.py).
 *
 * Semantics follow raytracing_c_amd/loaders.py (the loader the benchmark configs use) field for field: material
 * defaults of driver.c:549-568 (OBJ) and :628-639 (glTF, with the glTF-spec defaults metallic = roughness = 1 where
 * the file omits a factor), node TRS / matrix transforms applied to positions and normals, first perspective camera
 * node (driver.c:599-612), fan triangulation of OBJ polygons, face normals where a file has none.
 */
#ifndef RT_MODEL_H
#define RT_MODEL_H

#include <stddef.h>

#include "rt_raytracer.h"
#include "rt_materials.h"

typedef struct {
  Triangle        *triangles;      /* shader = { &materials[k], disney_shader_proc } (driver.c:574-577, 670-673) */
  isize            n_triangles;
  PBR_Shader_Data *materials;
  isize            n_materials;
  Image           *images;         /* RGB8, stride == width */
  isize            n_images;
  bool             has_camera;     /* glTF: first perspective camera node; OBJ: never */
  Camera           camera;         /* view_matrix (node transform), fov = yfov, focal_length = 1 / tan(fov / 2) */
} RT_Model;

/* Dispatch on the file suffix as driver.c:685-728.  false + message in err on any failure. */
bool rt_model_load(char const *path, RT_Model *out, char *err, size_t err_len);
void rt_model_free(RT_Model *model);

/* driver.c:765-767: T = (0, 0, 3), R = I, fov = 70 degrees */
Camera rt_model_default_camera(void);

/* matrix_4x4_translation_rotation_scale of driver.c:765 (quaternion x, y, z, w; unit scale) + field of view in radians */
Camera rt_model_camera(f32 const translation[3], f32 const rotation[4], f32 yfov);

/* one RT8I side file (what image k of a model is read from) -> Image; used for the environment map too */
bool rt_model_load_rgb8(char const *path, Image *out, char *err, size_t err_len);

/* rt_jpeg.c: a baseline or progressive JPEG stream (what helmet.glb embeds) -> RGB8 Image with libjpeg's default arithmetic (islow IDCT,
 * fancy upsampling), i.e. the texels the Python loader gets from PIL; pixels.data is malloc'ed.  false + message otherwise. */
bool rt_jpeg_decode(const unsigned char *data, size_t n, Image *out, char *err, size_t err_len);
/* rt_png.c: a non-interlaced PNG stream -> RGB8 Image the way PIL's .convert("RGB") maps it (alpha dropped, palette applied,
 * low bit depths scaled, the high byte of 16-bit samples). */
bool rt_png_decode(const unsigned char *data, size_t n, Image *out, char *err, size_t err_len);

#endif /* RT_MODEL_H */
