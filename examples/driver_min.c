/* driver_min.c -- a plain C host that uses librt_hip.so exactly the way the reference's driver uses
 * raytracer.c (driver.c:747-837): build a Scene with scene_init, fill a Rendering_Context, start T threads on
 * render_thread_proc, poll rendering_context_is_finished, optionally denoise, write the image.
 *
 *   make -C examples        (cc -std=gnu11 ... driver_min.c rt_model.c rt_jpeg.c rt_png.c -lrt_hip -lpthread -lm)
 *   examples/driver_min MODEL W H SAMPLES BOUNCES THREADS out.ppm|out.png|out.qoi [-D] [--background bg.rgb8] [--camera "tx ty tz qx qy qz qw fov"]
 *
 * MODEL is a model file as in driver.c:685-728 -- `.obj`, `.glb`, `.gltf`, loaded by rt_model.c (textures from the file's own baseline
 * JPEGs / PNGs via rt_jpeg.c / rt_png.c, or from RT8I side files of tools/extract_textures.py, which also writes the environment map) -- or a `.rtscene` dump of what the loaders produce
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

/* ---- image writers, chosen by the suffix of the output path as in driver.c:839-877 (png_save_writer / qoi_save_writer /
 * ppm_save_writer are codin functions there; these are self-contained) ------------------------------------------------ */
static unsigned crc32_update(unsigned c, unsigned char const *p, size_t n) {
  static unsigned table[256];
  if (!table[1]) for (unsigned i = 0; i < 256; i++) { unsigned k = i; for (int j = 0; j < 8; j++) k = (k & 1) ? 0xEDB88320u ^ (k >> 1) : k >> 1; table[i] = k; }
  c = ~c;
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return ~c;
}
static void put_be32(unsigned char *p, unsigned v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
static int png_chunk(FILE *o, char const *type, unsigned char const *data, size_t n) {
  unsigned char hd[8], tail[4];
  put_be32(hd, (unsigned)n);
  memcpy(hd + 4, type, 4);
  unsigned c = crc32_update(0, hd + 4, 4);
  if (n) c = crc32_update(c, data, n);
  put_be32(tail, c);
  return fwrite(hd, 1, 8, o) == 8 && (!n || fwrite(data, 1, n, o) == n) && fwrite(tail, 1, 4, o) == 4;
}
/* 8-bit RGB PNG, filter 0 on every row, zlib stream of STORED deflate blocks (valid PNG, no compression) */
static int write_png(FILE *o, unsigned char const *rgb, int w, int h) {
  static unsigned char const sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  unsigned char ihdr[13];
  put_be32(ihdr, (unsigned)w); put_be32(ihdr + 4, (unsigned)h);
  ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  size_t row = (size_t)w * 3 + 1, raw = row * (size_t)h, n_blocks = (raw + 65534) / 65535;
  size_t zlen = 2 + raw + 5 * n_blocks + 4;
  unsigned char *z = malloc(zlen), *q = z;
  if (!z) return 0;
  *q++ = 0x78; *q++ = 0x01;
  unsigned a = 1, b = 0;                                   /* Adler-32 of the filtered scanlines */
  size_t pos = 0;
  for (size_t blk = 0; blk < n_blocks; blk++) {
    size_t len = raw - pos < 65535 ? raw - pos : 65535;
    *q++ = blk + 1 == n_blocks; *q++ = len & 0xFF; *q++ = len >> 8; *q++ = ~len & 0xFF; *q++ = (~len >> 8) & 0xFF;
    for (size_t i = 0; i < len; i++, pos++) {
      size_t y = pos / row, x = pos % row;
      unsigned char v = x ? rgb[y * (size_t)w * 3 + x - 1] : 0;
      *q++ = v;
      a = (a + v) % 65521; b = (b + a) % 65521;
    }
  }
  put_be32(q, (b << 16) | a);
  int ok = fwrite(sig, 1, 8, o) == 8 && png_chunk(o, "IHDR", ihdr, 13) && png_chunk(o, "IDAT", z, zlen) && png_chunk(o, "IEND", NULL, 0);
  free(z);
  return ok;
}
/* QOI (qoiformat.org), 3 channels, sRGB */
static int write_qoi(FILE *o, unsigned char const *rgb, int w, int h) {
  unsigned char hd[14] = {'q', 'o', 'i', 'f'};
  put_be32(hd + 4, (unsigned)w); put_be32(hd + 8, (unsigned)h); hd[12] = 3; hd[13] = 0;
  size_t n = (size_t)w * h;
  unsigned char *buf = malloc(n * 4 + 8), *q = buf;
  if (!buf) return 0;
  unsigned char index[64][4];
  memset(index, 0, sizeof index);
  unsigned char pr = 0, pg = 0, pb = 0;
  int run = 0;
  for (size_t i = 0; i < n; i++) {
    unsigned char r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
    if (r == pr && g == pg && b == pb) {
      if (++run == 62 || i + 1 == n) { *q++ = 0xC0 | (run - 1); run = 0; }
      continue;
    }
    if (run) { *q++ = 0xC0 | (run - 1); run = 0; }
    int hpos = (r * 3 + g * 5 + b * 7 + 255 * 11) % 64;
    if (index[hpos][0] == r && index[hpos][1] == g && index[hpos][2] == b && index[hpos][3] == 255) {
      *q++ = (unsigned char)hpos;
    } else {
      index[hpos][0] = r; index[hpos][1] = g; index[hpos][2] = b; index[hpos][3] = 255;
      signed char dr = (signed char)(r - pr), dg = (signed char)(g - pg), db = (signed char)(b - pb);
      signed char dr_dg = (signed char)(dr - dg), db_dg = (signed char)(db - dg);
      if (dr > -3 && dr < 2 && dg > -3 && dg < 2 && db > -3 && db < 2) {
        *q++ = 0x40 | ((dr + 2) << 4) | ((dg + 2) << 2) | (db + 2);
      } else if (dr_dg > -9 && dr_dg < 8 && dg > -33 && dg < 32 && db_dg > -9 && db_dg < 8) {
        *q++ = 0x80 | (dg + 32);
        *q++ = ((dr_dg + 8) << 4) | (db_dg + 8);
      } else {
        *q++ = 0xFE; *q++ = r; *q++ = g; *q++ = b;
      }
    }
    pr = r; pg = g; pb = b;
  }
  static unsigned char const end[8] = {0, 0, 0, 0, 0, 0, 0, 1};
  memcpy(q, end, 8); q += 8;
  int ok = fwrite(hd, 1, 14, o) == 14 && fwrite(buf, 1, (size_t)(q - buf), o) == (size_t)(q - buf);
  free(buf);
  return ok;
}
static int write_image(char const *path, unsigned char const *rgb, int w, int h) {
  FILE *o = fopen(path, "wb");
  if (!o) return 0;
  size_t len = strlen(path);
  int ok;
  if (len > 4 && !strcmp(path + len - 4, ".png")) ok = write_png(o, rgb, w, h);
  else if (len > 4 && !strcmp(path + len - 4, ".qoi")) ok = write_qoi(o, rgb, w, h);
  else ok = fprintf(o, "P6\n%d %d\n255\n", w, h) > 0 && fwrite(rgb, 1, (size_t)w * h * 3, o) == (size_t)w * h * 3;   /* .ppm */
  return fclose(o) == 0 && ok;
}

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
             "gpu: clear+prepare %.3f, path kernel %.3f, resolve %.3f, copy to host %.3f | verify %.3f (host, under the kernel) | "
             "devices %d, slowest %d, gather %.3f\n", f, wall,
             (double)width * height * samples / wall / 1e3, ft.total_ms, ft.stamp_ms, ft.upload_ms, ft.enqueue_ms, ft.gpu_prep_ms,
             ft.gpu_path_ms, ft.gpu_resolve_ms, ft.gpu_copy_ms, ft.verify_ms, (int)ft.n_devices, (int)ft.slowest_device, ft.gather_ms);
    }
  }

  if (denoise) {                                                                                                   /* driver.c:827-837 */
    Image denoised = image;
    denoised.pixels.data = malloc((size_t)image.pixels.len);
    denoise_image(&image, &denoised, n_threads);
    if (rt_last_error()[0]) { fprintf(stderr, "driver_min: denoise failed: %s\n", rt_last_error()); return 2; }
    image = denoised;
  }

  if (!write_image(out_path, image.pixels.data, width, height)) return die("cannot write output");
  printf("driver_min: %dx%d, %d spp, %d bounces, %d thread(s), chunk counter %d -> %s\n", width, height, samples, bounces,
         n_threads, (int)ctx._current_chunk, out_path);
  return 0;
}
