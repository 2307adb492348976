/* driver_min.c -- a plain C host that uses librt_hip.so exactly the way the reference's driver uses
 * raytracer.c (driver.c:747-837): build a Scene with scene_init, fill a Rendering_Context, start T threads on
 * render_thread_proc, poll rendering_context_is_finished, optionally denoise, write the image.
 *
 *   make -C examples        (cc -std=gnu11 ... driver_min.c rt_model.c -lrt_hip -lpthread -lm)
 *   examples/driver_min MODEL W H SAMPLES BOUNCES THREADS out.ppm [-D] [--background bg.rgb8] [--camera "tx ty tz qx qy qz qw fov"]
 *
 * MODEL is a model file as in driver.c:685-728 -- `.obj`, `.glb`, `.gltf`, loaded by rt_model.c (textures and the
 * environment map from RT8I side files, tools/extract_textures.py) -- or a `.rtscene` dump of what the loaders produce
 * (raytracing_c_amd/scene_dump.py).  This file is the reference-side binding of INTEGRATION.md in compilable form.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "rt_hip.h"
#include "rt_model.h"

typedef struct { i32 magic, version, n_triangles, n_materials, n_images, background; } Dump_Header;
typedef struct { f32 pos[9], nrm[9], uv[6]; i32 material; } Dump_Triangle;
typedef struct { f32 base_color[3], emission[3], roughness, metalness, normal_map_strength, sheen, sheen_tint, aniso;
                 i32 tex_albedo, tex_normal, tex_mr, tex_emission; } Dump_Material;
typedef struct { i32 width, height, components; } Dump_Image;

static void *thread_main(void *arg) {
  render_thread_proc((Rendering_Context *)arg);
  return NULL;
}

static int die(char const *msg) {
  fprintf(stderr, "driver_min: %s\n", msg);
  return 1;
}

static int render_and_write(Scene *scene, int width, int height, int samples, int bounces, int n_threads, int denoise, char const *out_path);

/* driver.c:685-728 + :758-775: model file -> triangles, materials, images, camera; environment map; scene_init */
static int main_model(int argc, char **argv, int width, int height, int samples, int bounces, int n_threads) {
  int denoise = 0;
  char const *bg_path = NULL, *cam_text = NULL;
  for (int i = 8; i < argc; i++) {
    if (!strcmp(argv[i], "-D")) denoise = 1;
    else if (!strcmp(argv[i], "--background") && i + 1 < argc) bg_path = argv[++i];
    else if (!strcmp(argv[i], "--camera") && i + 1 < argc) cam_text = argv[++i];
  }
  char err[512] = "";
  RT_Model model;
  if (!rt_model_load(argv[1], &model, err, sizeof err)) { fprintf(stderr, "driver_min: %s\n", err); return 1; }
  static Image background;
  static byte grey[6] = {128, 128, 128, 128, 128, 128};
  if (bg_path) {
    if (!rt_model_load_rgb8(bg_path, &background, err, sizeof err)) { fprintf(stderr, "driver_min: %s\n", err); return 1; }
  } else {                                             /* background.png is a missing blob of the reference */
    background = (Image){ .components = 3, .pixel_type = PT_u8, .width = 2, .stride = 2, .height = 1, .pixels = { grey, 6 } };
  }
  Scene scene;
  memset(&scene, 0, sizeof scene);
  scene.background = (Background){ .proc = (Background_Proc)sample_background, .data = &background };            /* driver.c:760-763 */
  scene.camera = model.has_camera ? model.camera : rt_model_default_camera();                                     /* driver.c:765-767 */
  if (cam_text) {
    f32 v[8];
    if (sscanf(cam_text, "%f %f %f %f %f %f %f %f", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7]) != 8)
      return die("--camera wants \"tx ty tz qx qy qz qw fov\"");
    scene.camera = rt_model_camera(v, v + 3, v[7]);
  }
  scene_init(&scene, (Triangle_Slice){ model.triangles, model.n_triangles }, (Allocator){ 0 });                  /* driver.c:775 */
  printf("driver_min: %s: %ld triangles, %ld materials, %ld images, BVH depth %ld\n", argv[1], (long)model.n_triangles,
         (long)model.n_materials, (long)model.n_images, (long)scene.bvh.depth);
  return render_and_write(&scene, width, height, samples, bounces, n_threads, denoise, argv[7]);
}

static int has_suffix(char const *s, char const *suf) {
  size_t n = strlen(s), m = strlen(suf);
  return n >= m && strcmp(s + n - m, suf) == 0;
}

int main(int argc, char **argv) {
  if (argc < 8) return die("usage: driver_min MODEL W H SAMPLES BOUNCES THREADS out.ppm [-D] [--background bg.rgb8] [--camera \"tx ty tz qx qy qz qw fov\"]");
  int width = atoi(argv[2]), height = atoi(argv[3]), samples = atoi(argv[4]), bounces = atoi(argv[5]);
  int n_threads = atoi(argv[6]);
  int denoise = argc > 8 && strcmp(argv[8], "-D") == 0;
  if (n_threads < 1 || n_threads > 64) return die("THREADS must be 1..64");
  if (!has_suffix(argv[1], ".rtscene")) return main_model(argc, argv, width, height, samples, bounces, n_threads);

  FILE *f = fopen(argv[1], "rb");
  if (!f) return die("cannot open scene file");
  Dump_Header hd;
  if (fread(&hd, sizeof hd, 1, f) != 1 || hd.magic != 0x43535452 || hd.version != 1) return die("bad scene file");
  Camera camera;
  if (fread(&camera, sizeof camera, 1, f) != 1) return die("short read (camera)");
  Dump_Triangle *dt = malloc(sizeof *dt * (size_t)hd.n_triangles);
  Dump_Material *dm = malloc(sizeof *dm * (size_t)hd.n_materials);
  if (fread(dt, sizeof *dt, (size_t)hd.n_triangles, f) != (size_t)hd.n_triangles) return die("short read (triangles)");
  if (fread(dm, sizeof *dm, (size_t)hd.n_materials, f) != (size_t)hd.n_materials) return die("short read (materials)");
  Image *images = calloc((size_t)hd.n_images, sizeof *images);
  for (int i = 0; i < hd.n_images; i++) {
    Dump_Image di;
    if (fread(&di, sizeof di, 1, f) != 1) return die("short read (image header)");
    size_t n = (size_t)di.width * di.height * di.components;
    images[i] = (Image){ .components = di.components, .pixel_type = PT_u8, .width = di.width, .stride = di.width,
                         .height = di.height, .pixels = { malloc(n), (isize)n } };
    if (fread(images[i].pixels.data, 1, n, f) != n) return die("short read (image)");
  }
  fclose(f);

  /* driver.c:628-660: one PBR_Shader_Data per material, textures by pointer */
  PBR_Shader_Data *mats = calloc((size_t)hd.n_materials, sizeof *mats);
  for (int i = 0; i < hd.n_materials; i++) {
    Dump_Material const *m = &dm[i];
    mats[i] = (PBR_Shader_Data){
      .base_color = {{ m->base_color[0], m->base_color[1], m->base_color[2] }},
      .emission = {{ m->emission[0], m->emission[1], m->emission[2] }},
      .roughness = m->roughness, .metalness = m->metalness, .normal_map_strength = m->normal_map_strength,
      .sheen = m->sheen, .sheen_tint = m->sheen_tint, .anisotropic_strength = m->aniso,
      .texture_albedo = m->tex_albedo >= 0 ? &images[m->tex_albedo] : NULL,
      .texture_normal = m->tex_normal >= 0 ? &images[m->tex_normal] : NULL,
      .texture_metal_roughness = m->tex_mr >= 0 ? &images[m->tex_mr] : NULL,
      .texture_emission = m->tex_emission >= 0 ? &images[m->tex_emission] : NULL,
    };
  }
  /* driver.c:668-681: triangles with shader = { &material, disney_shader_proc } */
  Triangle *tris = calloc((size_t)hd.n_triangles, sizeof *tris);
  for (int i = 0; i < hd.n_triangles; i++) {
    for (int j = 0; j < 3; j++) {
      tris[i].positions[j] = (Vec3){{ dt[i].pos[3 * j], dt[i].pos[3 * j + 1], dt[i].pos[3 * j + 2] }};
      tris[i].normals[j] = (Vec3){{ dt[i].nrm[3 * j], dt[i].nrm[3 * j + 1], dt[i].nrm[3 * j + 2] }};
      tris[i].tex_coords[j] = (Vec2){{ dt[i].uv[2 * j], dt[i].uv[2 * j + 1] }};
    }
    tris[i].shader = (Shader){ .data = &mats[dt[i].material], .proc = disney_shader_proc };
  }

  Scene scene;
  memset(&scene, 0, sizeof scene);
  scene.background = (Background){ .proc = (Background_Proc)sample_background, .data = &images[hd.background] };   /* driver.c:760-763 */
  scene.camera = camera;
  scene_init(&scene, (Triangle_Slice){ tris, hd.n_triangles }, (Allocator){ 0 });                                  /* driver.c:775 */
  return render_and_write(&scene, width, height, samples, bounces, n_threads, denoise, argv[7]);
}

static int render_and_write(Scene *scene_p, int width, int height, int samples, int bounces, int n_threads, int denoise, char const *out_path) {
  Scene scene = *scene_p;
  Image image = { .components = 3, .pixel_type = PT_u8, .width = width, .stride = width, .height = height };       /* driver.c:747-754 */
  image.pixels.len = (isize)width * height * 3;
  image.pixels.data = aligned_alloc(64, ((size_t)image.pixels.len + 63) / 64 * 64);
  memset(image.pixels.data, 0, (size_t)image.pixels.len);

  rt_set_seed(0x1234ABCD);
  Rendering_Context ctx = { .image = image, .scene = &scene, .max_bounces = bounces, .n_threads = n_threads,       /* driver.c:793-799 */
                            .samples = samples };
  pthread_t th[64];
  /* DRIVER_MIN_FRAMES=n renders the frame n times (the context re-armed each time) and prints where the time of every frame
   * went -- the reference's `-V` prints render ms and samples/second (driver.c:821-825); rt_get_frame_timing() splits it. */
  int frames = getenv("DRIVER_MIN_FRAMES") ? atoi(getenv("DRIVER_MIN_FRAMES")) : 1;
  if (frames < 1) frames = 1;
  for (int f = 0; f < frames; f++) {
    ctx.n_threads = n_threads;
    ctx._current_chunk = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, thread_main, &ctx);                            /* driver.c:801-803 */
    while (!rendering_context_is_finished(&ctx)) usleep(frames > 1 ? 20 : 1000);                                   /* driver.c:810-818 */
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
    if (rt_last_error()[0]) { fprintf(stderr, "driver_min: render failed: %s\n", rt_last_error()); return 2; }
    if (frames > 1) {
      RT_Frame_Timing ft;
      rt_get_frame_timing(&ft);
      double wall = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
      printf("driver_min: frame %d: host wall %.3f ms (%.1f Msample/s) | library total %.3f = stamp %.3f + upload %.3f + enqueue %.3f ... | "
             "gpu: clear+prepare %.3f, path kernel %.3f, resolve %.3f, copy to host %.3f\n", f, wall,
             (double)width * height * samples / wall / 1e3, ft.total_ms, ft.stamp_ms, ft.upload_ms, ft.enqueue_ms, ft.gpu_prep_ms,
             ft.gpu_path_ms, ft.gpu_resolve_ms, ft.gpu_copy_ms);
    }
  }

  if (denoise) {                                                                                                   /* driver.c:827-837 */
    Image denoised = image;
    denoised.pixels.data = malloc((size_t)image.pixels.len);
    denoise_image(&image, &denoised, n_threads);
    if (rt_last_error()[0]) { fprintf(stderr, "driver_min: denoise failed: %s\n", rt_last_error()); return 2; }
    image = denoised;
  }

  FILE *o = fopen(out_path, "wb");
  if (!o) return die("cannot open output");
  fprintf(o, "P6\n%d %d\n255\n", width, height);
  fwrite(image.pixels.data, 1, (size_t)image.pixels.len, o);
  fclose(o);
  printf("driver_min: %dx%d, %d spp, %d bounces, %d thread(s), chunk counter %d -> %s\n", width, height, samples, bounces,
         n_threads, (int)ctx._current_chunk, out_path);
  return 0;
}
