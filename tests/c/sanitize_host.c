/* Host-side C code under AddressSanitizer + UBSan (CPU only; tests/test_sanitizers.py builds and runs this):
 * scene_init (threaded builder), the .scene file round trip, and the oracle's render / denoise / lightmap loops on
 * the scenes it produces -- random triangle soups of sizes that hit depth 0, the early-leaf chain and 3 levels. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/rt_materials.h"
#include "../../include/rt_raytracer.h"
#include "../../oracle/oracle.h"

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s; }
static float frand(uint32_t *s) { return (float)(lcg(s) >> 8) / 16777216.0f; }

/* address tokens for the scene: the oracle compares shader.proc against the addresses given in Oracle_Config */
static void token_disney(rawptr d, Shader_Input const *i, Shader_Output *o) { (void)d; (void)i; (void)o; }
static Color3 token_background(rawptr d, Vec3 dir) { (void)d; (void)dir; Color3 c = {{0, 0, 0}}; return c; }

static int run_case(int n_tris, uint32_t seed) {
  uint32_t s = seed;
  PBR_Shader_Data mat;
  memset(&mat, 0, sizeof mat);
  mat.base_color.x = 0.8f; mat.base_color.y = 0.6f; mat.base_color.z = 0.4f;
  mat.roughness = 0.5f;
  Triangle *tris = (Triangle *)calloc((size_t)(n_tris > 0 ? n_tris : 1), sizeof *tris);
  for (int i = 0; i < n_tris; i++) {
    float cx = frand(&s) * 4 - 2, cy = frand(&s) * 4 - 2, cz = frand(&s) * 4 - 6;
    for (int k = 0; k < 3; k++) {
      tris[i].positions[k].x = cx + frand(&s) * 0.5f;
      tris[i].positions[k].y = cy + frand(&s) * 0.5f;
      tris[i].positions[k].z = cz + frand(&s) * 0.5f;
      tris[i].normals[k].x = 0; tris[i].normals[k].y = 0; tris[i].normals[k].z = 1;
      tris[i].tex_coords[k].x = frand(&s); tris[i].tex_coords[k].y = frand(&s);
    }
    tris[i].shader.data = &mat;
    tris[i].shader.proc = token_disney;
  }
  Scene scene;
  memset(&scene, 0, sizeof scene);
  Triangle_Slice sl = { tris, n_tris };
  Allocator none = { 0, 0 };
  scene_init(&scene, sl, none);
  if (!scene.triangles.x[0]) { fprintf(stderr, "scene_init failed\n"); return 1; }

  /* camera at the origin looking down -z, 60 degrees */
  memset(&scene.camera, 0, sizeof scene.camera);
  for (int i = 0; i < 4; i++) scene.camera.view_matrix.rows[i][i] = 1.0f;
  scene.camera.fov = 1.0472f;
  scene.camera.focal_length = 1.7320508f;
  static uint8_t bg_px[16 * 8 * 3];
  for (size_t i = 0; i < sizeof bg_px; i++) bg_px[i] = (uint8_t)(40 + i % 150);
  Image bg;
  memset(&bg, 0, sizeof bg);
  bg.components = 3; bg.width = 16; bg.stride = 16; bg.height = 8; bg.pixels.data = bg_px; bg.pixels.len = sizeof bg_px;
  scene.background.proc = token_background;
  scene.background.data = &bg;

  /* .scene round trip into a 32-byte aligned buffer */
  isize bytes = scene_file_size(&scene);
  byte *file = (byte *)aligned_alloc(32, (size_t)((bytes + 31) / 32 * 32));
  if (scene_save_bytes(&scene, file, bytes) != bytes) { fprintf(stderr, "save failed\n"); return 1; }
  Scene loaded;
  memset(&loaded, 0, sizeof loaded);
  Byte_Slice data = { file, bytes };
  if (!scene_load_bytes(data, &loaded)) { fprintf(stderr, "load failed\n"); return 1; }
  if (loaded.triangles.len != scene.triangles.len || loaded.bvh.depth != scene.bvh.depth) return 1;
  loaded.background = scene.background;
  Byte_Slice cut = { file, bytes - 32 };
  Scene junk;
  if (scene_load_bytes(cut, &junk)) { fprintf(stderr, "truncated file accepted\n"); return 1; }

  /* oracle render of the LOADED scene (aliases `file`), 3 threads, then denoise, then a lightmap */
  enum { W = 37, H = 21 };
  static uint8_t px[W * H * 3], px2[W * H * 3];
  Image img;
  memset(&img, 0, sizeof img);
  img.components = 3; img.width = W; img.stride = W; img.height = H; img.pixels.data = px; img.pixels.len = sizeof px;
  Image img2 = img;
  img2.pixels.data = px2;
  Oracle_Config cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.disney_proc = token_disney;
  cfg.background_proc = token_background;
  cfg.seed = 0x1234ABCDu;
  cfg.n_threads = 3;
  Oracle_Counters cnt;
  static u64 accum[W * H * 3];
  static f32 linear[W * H * 3];
  if (oracle_render(&loaded, &img, 3, 5, &cfg, linear, accum, &cnt) != 0) { fprintf(stderr, "render failed\n"); return 1; }
  if (cnt.paths != (u64)W * H * 3) { fprintf(stderr, "paths %llu\n", (unsigned long long)cnt.paths); return 1; }
  oracle_denoise_image(&img, &img2);
  static uint8_t lm_px[24 * 24 * 3];
  Image lm = img;
  lm.width = 24; lm.stride = 24; lm.height = 24; lm.pixels.data = lm_px; lm.pixels.len = sizeof lm_px;
  cfg.n_threads = 1;
  oracle_lightmap_bake(&lm, &loaded, 2, &cfg);

  printf("n_tris %d depth %ld slots %d rays %llu ok\n", n_tris, (long)scene.bvh.depth, (int)scene.triangles.len,
         (unsigned long long)cnt.rays);
  free(file);
  rt_scene_free(&scene);
  free(tris);
  return 0;
}

int main(void) {
  int sizes[] = {0, 1, 8, 9, 64, 65, 700, 5000};
  for (unsigned i = 0; i < sizeof sizes / sizeof sizes[0]; i++)
    if (run_case(sizes[i], 1000u + i) != 0) return 1;
  return 0;
}
