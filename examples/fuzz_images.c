/* fuzz_images.c -- the C host's readers under AddressSanitizer / UBSan on the CPU (tests/test_c_loader.py builds and runs it):
 *   fuzz_images ROUNDS FILE...          decodes every image FILE (rt_jpeg.c / rt_png.c), then ROUNDS mutated copies of it
 *   fuzz_images -m DIR ROUNDS MODEL...  loads every MODEL (.obj / .gltf / .glb, rt_model.c), then ROUNDS mutated copies written to DIR
 * (byte flips, truncations, spliced blocks; xorshift, seeded by the file's length).  A damaged file may load or fail with a
 * message -- it must not read or write out of bounds, leak or overflow.  Prints the number of files accepted / refused. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_model.h"

static unsigned long long rng_state;
static unsigned rnd(void) {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (unsigned)(rng_state >> 11);
}

static int decode(const unsigned char *b, size_t n, int *ok, int *bad) {
  Image img;
  char err[256] = "";
  bool r = (n >= 2 && b[0] == 0xFF) ? rt_jpeg_decode(b, n, &img, err, sizeof err) : rt_png_decode(b, n, &img, err, sizeof err);
  if (r) {
    if (!img.pixels.data || img.width <= 0 || img.height <= 0 || img.pixels.len != (isize)img.width * img.height * 3) return 1;
    unsigned long long sum = 0;                       /* touch every byte of the result */
    for (isize i = 0; i < img.pixels.len; i++) sum += ((unsigned char *)img.pixels.data)[i];
    if (sum == ~0ull) puts("");
    free(img.pixels.data);
    (*ok)++;
  } else {
    if (!err[0] || img.pixels.data) return 1;         /* a refusal carries a message and no buffer */
    (*bad)++;
  }
  return 0;
}

/* rt_model.c names the library's shader entry points in the triangles it builds; nothing here calls them */
void disney_shader_proc(rawptr data, Shader_Input const *input, Shader_Output *output) { (void)data; (void)input; (void)output; }
void debug_shader_proc(rawptr data, Shader_Input const *input, Shader_Output *output) { (void)data; (void)input; (void)output; }

static int load_model(const char *path, int *ok, int *bad) {
  RT_Model m;
  char err[512] = "";
  if (rt_model_load(path, &m, err, sizeof err)) {
    unsigned long long sum = 0;
    for (isize i = 0; i < m.n_triangles; i++) sum += ((unsigned char *)&m.triangles[i])[0] + (m.triangles[i].shader.data != NULL);
    for (isize i = 0; i < m.n_images; i++)
      if (m.images[i].pixels.data) sum += ((unsigned char *)m.images[i].pixels.data)[m.images[i].pixels.len - 1];
    if (sum == ~0ull) puts("");
    rt_model_free(&m);
    (*ok)++;
  } else {
    if (!err[0]) return 1;
    (*bad)++;
  }
  return 0;
}

static void mutate(unsigned char *m, const unsigned char *orig, size_t n, size_t *len) {
  memcpy(m, orig, n);
  *len = n;
  switch (rnd() % 4) {
    case 0: for (int k = 1 + (int)(rnd() % 4); k > 0; k--) m[rnd() % n] ^= (unsigned char)(1u << (rnd() % 8)); break;
    case 1: for (int k = 1 + (int)(rnd() % 8); k > 0; k--) m[rnd() % n] = (unsigned char)rnd(); break;
    case 2: *len = rnd() % n; break;
    default: { size_t at = rnd() % n, from = rnd() % n, cnt = rnd() % 64; for (size_t k = 0; k < cnt && at + k < n && from + k < n; k++) m[at + k] = orig[from + k]; }
  }
}

static int fuzz_models(int argc, char **argv) {
  const char *dir = argv[2];
  int rounds = atoi(argv[3]), ok = 0, bad = 0;
  for (int a = 4; a < argc; a++) {
    FILE *f = fopen(argv[a], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    size_t n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *orig = malloc(n + 1), *m = malloc(n + 1);
    if (fread(orig, 1, n, f) != n) return 2;
    fclose(f);
    if (load_model(argv[a], &ok, &bad)) return 3;
    const char *ext = strrchr(argv[a], '.');
    char path[4096];
    snprintf(path, sizeof path, "%s/mutant%s", dir, ext ? ext : "");
    rng_state = 0x9E3779B97F4A7C15ull ^ n;
    for (int r = 0; r < rounds; r++) {
      size_t len;
      mutate(m, orig, n, &len);
      FILE *o = fopen(path, "wb");
      if (!o || fwrite(m, 1, len, o) != len) return 2;
      fclose(o);
      if (load_model(path, &ok, &bad)) { fprintf(stderr, "round %d of %s: refused without a message (mutant kept at %s)\n", r, argv[a], path); return 3; }
    }
    free(orig);
    free(m);
  }
  printf("%d loaded, %d refused\n", ok, bad);
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 5 && !strcmp(argv[1], "-m")) return fuzz_models(argc, argv);
  if (argc < 3) return 2;
  int rounds = atoi(argv[1]), ok = 0, bad = 0;
  for (int a = 2; a < argc; a++) {
    FILE *f = fopen(argv[a], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    size_t n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *orig = malloc(n + 1), *m = malloc(n + 1);
    if (fread(orig, 1, n, f) != n) return 2;
    fclose(f);
    if (decode(orig, n, &ok, &bad)) return 3;
    rng_state = 0x9E3779B97F4A7C15ull ^ n;
    for (int r = 0; r < rounds; r++) {
      size_t len;
      mutate(m, orig, n, &len);
      if (decode(m, len, &ok, &bad)) return 3;
    }
    free(orig);
    free(m);
  }
  printf("%d decoded, %d refused\n", ok, bad);
  return 0;
}
