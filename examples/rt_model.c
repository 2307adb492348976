/* rt_model.c -- see rt_model.h.  Plain C11, no dependencies beyond libc / libm.
 * Build with -ffp-contract=off: the transform arithmetic is written to round like raytracing_c_amd/loaders.py. */
#include "rt_model.h"

#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------- */
/* small utilities                                                                                                 */

typedef struct { char *err; size_t err_len; } Err;

static bool fail(Err *e, char const *fmt, ...) {
  if (e && e->err && e->err_len) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(e->err, e->err_len, fmt, ap);
    va_end(ap);
  }
  return false;
}

static byte *read_file(char const *path, size_t *len) {
  FILE *f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  byte *b = n >= 0 ? malloc((size_t)n + 1) : NULL;
  if (b && fread(b, 1, (size_t)n, f) != (size_t)n) { free(b); b = NULL; }
  fclose(f);
  if (b) { b[n] = 0; *len = (size_t)n; }
  return b;
}

static void dir_of(char const *path, char *out, size_t n) {
  char const *s = strrchr(path, '/');
  size_t k = s ? (size_t)(s - path) + 1 : 0;
  if (k >= n) k = n - 1;
  memcpy(out, path, k);
  out[k] = 0;
}

typedef struct { void *data; size_t len, cap, elem; } Vec;
static void *vec_push(Vec *v, size_t elem) {
  if (!v->elem) v->elem = elem;
  if (v->len == v->cap) {
    size_t cap = v->cap ? v->cap * 2 : 64;
    void *p = realloc(v->data, cap * v->elem);
    if (!p) return NULL;
    v->data = p;
    v->cap = cap;
  }
  return (char *)v->data + (v->len++) * v->elem;
}

static bool load_rgb8(char const *p, Image *img, Err *e);

bool rt_model_load_rgb8(char const *path, Image *out, char *err, size_t err_len) {
  Err e = { err, err_len };
  return load_rgb8(path, out, &e);
}

static bool load_rgb8(char const *p, Image *img, Err *e) {
  size_t n;
  byte *b = read_file(p, &n);
  if (!b) return fail(e, "texture side file '%s' is missing (tools/extract_textures.py writes it)", p);
  i32 hd[3];
  if (n < 16 || memcmp(b, "RT8I", 4) != 0) { free(b); return fail(e, "'%s' is not an RT8I file", p); }
  memcpy(hd, b + 4, 12);
  size_t want = (size_t)hd[0] * (size_t)hd[1] * (size_t)hd[2];
  if (hd[0] <= 0 || hd[1] <= 0 || hd[2] < 3 || n != 16 + want) { free(b); return fail(e, "'%s' has a bad header", p); }
  memset(img, 0, sizeof *img);
  img->components = hd[2];
  img->pixel_type = PT_u8;
  img->width = hd[0];
  img->stride = hd[0];
  img->height = hd[1];
  img->pixels.data = malloc(want);
  img->pixels.len = (isize)want;
  if (!img->pixels.data) { free(b); return fail(e, "out of memory"); }
  memcpy(img->pixels.data, b + 16, want);
  free(b);
  return true;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* cameras (loaders.py: quat_to_matrix, camera_from_trs, default_camera)                                          */

static void trs_matrix(f32 const t[3], f32 const q[4], f32 const s[3], f32 m[4][4]) {
  f32 x = q[0], y = q[1], z = q[2], w = q[3];
  f32 const one = 1.0f, two = 2.0f;
  f32 r[3][3] = {
    { one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w) },
    { two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w) },
    { two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y) } };
  memset(m, 0, sizeof(f32) * 16);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) m[i][j] = r[i][j] * s[j];
    m[i][3] = t[i];
  }
  m[3][3] = 1.0f;
}

static void set_camera(Camera *cam, f32 const m[4][4], f32 yfov) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) cam->view_matrix.rows[i][j] = m[i][j];
  cam->fov = yfov;
  cam->focal_length = 1.0f / tanf(yfov * 0.5f);               /* driver.c:607,767 */
}

Camera rt_model_camera(f32 const translation[3], f32 const rotation[4], f32 yfov) {
  f32 s[3] = {1, 1, 1}, m[4][4];
  trs_matrix(translation, rotation, s, m);
  Camera cam;
  memset(&cam, 0, sizeof cam);
  set_camera(&cam, m, yfov);
  return cam;
}

Camera rt_model_default_camera(void) {
  f32 t[3] = {0, 0, 3}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1}, m[4][4];
  trs_matrix(t, q, s, m);
  Camera cam;
  memset(&cam, 0, sizeof cam);
  f32 fov = (70.0f / 360.0f) * 3.14159265358979323846f * 2.0f;       /* driver.c:766 */
  set_camera(&cam, m, fov);
  return cam;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* materials                                                                                                       */

static PBR_Shader_Data material_default(f32 r, f32 g, f32 b, f32 roughness, f32 metalness) {
  PBR_Shader_Data m;
  memset(&m, 0, sizeof m);
  m.base_color.x = r; m.base_color.y = g; m.base_color.z = b;
  m.roughness = roughness;
  m.metalness = metalness;
  return m;
}

/* an encoded image (what stb_image_load_bytes takes in driver.c:106-116) -> RGB8: baseline JPEG (rt_jpeg.c) or PNG (rt_png.c) */
static bool decode_image(byte const *bytes, size_t n, Image *img, char *msg, size_t msg_len) {
  if (n >= 2 && bytes[0] == 0xFF && bytes[1] == 0xD8) return rt_jpeg_decode(bytes, n, img, msg, msg_len);
  if (n >= 4 && bytes[0] == 0x89 && bytes[1] == 'P' && bytes[2] == 'N' && bytes[3] == 'G') return rt_png_decode(bytes, n, img, msg, msg_len);
  snprintf(msg, msg_len, "neither a JPEG nor a PNG stream");
  return false;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* OBJ / MTL (loaders.py: _parse_mtl, load_obj; driver.c:510-587)                                                  */

typedef struct {
  char name[256];
  f32  kd[3], ke[3];
  bool pbr, has_pr, has_pm, has_ps, has_aniso;
  f32  pr, pm, ps, aniso;
  char maps[4][1024];              /* files of map_Kd, map_Ke, norm, map_Pm ("" = none), relative to the .mtl */
  char base[1024];
} Mtl;

static int split_ws(char *line, char **tok, int max_tok) {
  int n = 0;
  char *p = line;
  while (*p && n < max_tok) {
    while (*p && isspace((unsigned char)*p)) p++;
    if (!*p) break;
    tok[n++] = p;
    while (*p && !isspace((unsigned char)*p)) p++;
    if (*p) *p++ = 0;
  }
  return n;
}

static void join_tokens(char **tok, int from, int n, char *out, size_t cap) {
  out[0] = 0;
  for (int i = from; i < n; i++) {
    if (i > from) strncat(out, " ", cap - strlen(out) - 1);
    strncat(out, tok[i], cap - strlen(out) - 1);
  }
}

static void parse_mtl(char const *path, Vec *mtls) {
  size_t n;
  char *text = (char *)read_file(path, &n);
  if (!text) return;                                  /* a missing .mtl is not an error (tower.obj) */
  Mtl *cur = NULL;
  char *save = NULL;
  for (char *line = strtok_r(text, "\n", &save); line; line = strtok_r(NULL, "\n", &save)) {
    char *tok[16];
    int nt = split_ws(line, tok, 16);
    if (!nt || tok[0][0] == '#') continue;
    if (!strcmp(tok[0], "newmtl")) {
      cur = vec_push(mtls, sizeof(Mtl));
      if (!cur) break;
      memset(cur, 0, sizeof *cur);
      join_tokens(tok, 1, nt, cur->name, sizeof cur->name);
      cur->kd[0] = cur->kd[1] = cur->kd[2] = (f32)0.8;
      dir_of(path, cur->base, sizeof cur->base);
      continue;
    }
    if (!cur) continue;
    if ((!strcmp(tok[0], "Kd") || !strcmp(tok[0], "Ke")) && nt >= 4) {
      f32 *dst = tok[0][1] == 'd' ? cur->kd : cur->ke;
      for (int k = 0; k < 3; k++) dst[k] = (f32)strtod(tok[1 + k], NULL);
    } else if (nt >= 2 && (!strcmp(tok[0], "Pr") || !strcmp(tok[0], "Pm") || !strcmp(tok[0], "Ps") || !strcmp(tok[0], "Pc") ||
                           !strcmp(tok[0], "Pcr") || !strcmp(tok[0], "aniso") || !strcmp(tok[0], "anisor"))) {
      f32 v = (f32)strtod(tok[1], NULL);
      cur->pbr = true;
      if (!strcmp(tok[0], "Pr")) { cur->pr = v; cur->has_pr = true; }
      else if (!strcmp(tok[0], "Pm")) { cur->pm = v; cur->has_pm = true; }
      else if (!strcmp(tok[0], "Ps")) { cur->ps = v; cur->has_ps = true; }
      else if (!strcmp(tok[0], "aniso")) { cur->aniso = v; cur->has_aniso = true; }
    }
    else if (nt >= 2) {                                 /* texture maps: the last token is the file (options before it are ignored) */
      static char const *const keys[4] = { "map_Kd", "map_Ke", "norm", "map_Pm" };
      for (int k = 0; k < 4; k++)
        if (!strcmp(tok[0], keys[k])) snprintf(cur->maps[k], sizeof cur->maps[k], "%s", tok[nt - 1]);
    }
  }
  free(text);
}

typedef struct { int v, t, n; } ObjIdx;

static bool load_obj(char const *path, RT_Model *out, Err *e) {
  size_t len;
  char *text = (char *)read_file(path, &len);
  if (!text) return fail(e, "cannot read '%s'", path);
  char base[4096];
  dir_of(path, base, sizeof base);
  Vec V = {0}, VT = {0}, VN = {0}, F = {0}, FM = {0}, mtls = {0};
  int cur_mat = -1;                                   /* index into mtls, -1 = none / unknown name */
  bool ok = true;
  char *save = NULL;
  for (char *line = strtok_r(text, "\n", &save); line && ok; line = strtok_r(NULL, "\n", &save)) {
    char *tok[256];
    int nt = split_ws(line, tok, 256);
    if (!nt) continue;
    if (!strcmp(tok[0], "v") && nt >= 4) {
      f32 *p = vec_push(&V, sizeof(f32) * 3);
      if (!p) { ok = false; break; }
      for (int k = 0; k < 3; k++) p[k] = (f32)strtod(tok[1 + k], NULL);
    } else if (!strcmp(tok[0], "vt") && nt >= 2) {
      f32 *p = vec_push(&VT, sizeof(f32) * 2);
      if (!p) { ok = false; break; }
      p[0] = (f32)strtod(tok[1], NULL);
      p[1] = nt > 2 ? (f32)strtod(tok[2], NULL) : 0.0f;
    } else if (!strcmp(tok[0], "vn") && nt >= 4) {
      f32 *p = vec_push(&VN, sizeof(f32) * 3);
      if (!p) { ok = false; break; }
      for (int k = 0; k < 3; k++) p[k] = (f32)strtod(tok[1 + k], NULL);
    } else if (!strcmp(tok[0], "mtllib") && nt >= 2) {
      char name[1024], full[4096 + 1024];
      join_tokens(tok, 1, nt, name, sizeof name);
      snprintf(full, sizeof full, "%s%s", base, name);
      parse_mtl(full, &mtls);
    } else if (!strcmp(tok[0], "usemtl")) {
      char name[256];
      join_tokens(tok, 1, nt, name, sizeof name);
      cur_mat = -1;
      for (size_t k = 0; k < mtls.len; k++)
        if (!strcmp(((Mtl *)mtls.data)[k].name, name)) { cur_mat = (int)k; break; }
    } else if (!strcmp(tok[0], "f") && nt >= 4) {
      ObjIdx idx[255];
      int ni = 0;
      for (int k = 1; k < nt && ni < 255; k++) {
        ObjIdx a = {0, 0, 0};
        char *s = tok[k];
        a.v = (int)strtol(s, &s, 10);
        if (*s == '/') { s++; if (*s != '/') a.t = (int)strtol(s, &s, 10); if (*s == '/') { s++; a.n = (int)strtol(s, &s, 10); } }
        idx[ni++] = a;
      }
      for (int j = 1; j + 1 < ni; j++) {              /* fan triangulation */
        ObjIdx *f = vec_push(&F, sizeof(ObjIdx) * 3);
        int *m = vec_push(&FM, sizeof(int));
        if (!f || !m) { ok = false; break; }
        f[0] = idx[0]; f[1] = idx[j]; f[2] = idx[j + 1];
        *m = cur_mat;
      }
    }
  }
  free(text);
  if (!ok) { free(V.data); free(VT.data); free(VN.data); free(F.data); free(FM.data); free(mtls.data); return fail(e, "out of memory"); }

  isize n = (isize)F.len;
  /* materials: one per MTL entry (driver.c:549-568); faces without a known material share one default */
  bool need_default = false, mat_fail = false;
  for (isize i = 0; i < n; i++) need_default |= ((int *)FM.data)[i] < 0;
  out->n_materials = (isize)mtls.len + (need_default ? 1 : 0);
  out->materials = calloc((size_t)(out->n_materials > 0 ? out->n_materials : 1), sizeof *out->materials);
  out->images = calloc(4 * mtls.len + 1, sizeof *out->images);      /* one per map that exists, in the order loaders.py appends them */
  out->n_images = 0;
  for (size_t k = 0; k < mtls.len && ok; k++) {
    Mtl const *m = &((Mtl *)mtls.data)[k];
    PBR_Shader_Data d = material_default(m->kd[0], m->kd[1], m->kd[2], 0.5f, 0.0f);
    d.emission.x = m->ke[0]; d.emission.y = m->ke[1]; d.emission.z = m->ke[2];
    if (m->pbr) {
      d.anisotropic_strength = m->has_aniso ? m->aniso : 0.0f;
      d.metalness = m->has_pm ? m->pm : 0.0f;
      d.roughness = m->has_pr ? m->pr : 0.0f;
      d.sheen = m->has_ps ? m->ps : 0.0f;
    }
    Image **slots[4] = { &d.texture_albedo, &d.texture_emission, &d.texture_normal, &d.texture_metal_roughness };
    for (int t = 0; t < 4 && ok; t++) {
      if (!m->maps[t][0] || (t >= 2 && !m->pbr)) continue;          /* norm / map_Pm only count in a PBR material */
      char full[2100], msg[200] = "";
      snprintf(full, sizeof full, "%s%s", m->base, m->maps[t]);
      size_t bn;
      byte *bytes = read_file(full, &bn);
      if (!bytes) continue;                                          /* a map that is not there is no map (loaders.py: add_image) */
      Image *img = &out->images[out->n_images];
      if (!decode_image(bytes, bn, img, msg, sizeof msg)) { free(bytes); ok = fail(e, "'%s': %s", full, msg); mat_fail = true; break; }
      free(bytes);
      out->n_images++;
      *slots[t] = img;
    }
    out->materials[k] = d;
  }
  if (mat_fail) { free(V.data); free(VT.data); free(VN.data); free(F.data); free(FM.data); free(mtls.data); return false; }
  if (need_default) out->materials[mtls.len] = material_default((f32)0.8, (f32)0.8, (f32)0.8, 0.5f, 0.0f);

  out->n_triangles = n;
  out->triangles = calloc((size_t)(n > 0 ? n : 1), sizeof *out->triangles);
  f32 const *pv = V.data, *pt = VT.data, *pn = VN.data;
  for (isize i = 0; i < n && ok; i++) {
    ObjIdx const *f = (ObjIdx const *)F.data + 3 * i;
    Triangle *t = &out->triangles[i];
    for (int k = 0; k < 3; k++) {
      int vi = f[k].v < 0 ? f[k].v + (int)V.len + 1 : f[k].v;            /* negative = relative to the end */
      int ti = f[k].t < 0 ? f[k].t + (int)VT.len + 1 : f[k].t;
      if (vi < 1 || vi > (int)V.len) { ok = false; break; }
      for (int a = 0; a < 3; a++) t->positions[k].data[a] = pv[(size_t)(vi - 1) * 3 + a];
      if (ti > 0 && ti <= (int)VT.len) { t->tex_coords[k].x = pt[(size_t)(ti - 1) * 2]; t->tex_coords[k].y = pt[(size_t)(ti - 1) * 2 + 1]; }
    }
    if (!ok) break;
    /* face normal (fp32, as numpy computes it on float32 arrays) for corners without a `vn` */
    f32 e1[3], e2[3], fn[3];
    for (int a = 0; a < 3; a++) { e1[a] = t->positions[1].data[a] - t->positions[0].data[a]; e2[a] = t->positions[2].data[a] - t->positions[0].data[a]; }
    fn[0] = e1[1] * e2[2] - e1[2] * e2[1];
    fn[1] = e1[2] * e2[0] - e1[0] * e2[2];
    fn[2] = e1[0] * e2[1] - e1[1] * e2[0];
    f32 norm = sqrtf(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]);
    if (!(norm > 1e-30f)) norm = 1e-30f;
    for (int a = 0; a < 3; a++) fn[a] = fn[a] / norm;
    for (int k = 0; k < 3; k++) {
      int ni = f[k].n < 0 ? f[k].n + (int)VN.len + 1 : f[k].n;
      for (int a = 0; a < 3; a++) t->normals[k].data[a] = (ni > 0 && ni <= (int)VN.len) ? pn[(size_t)(ni - 1) * 3 + a] : fn[a];
    }
    int m = ((int *)FM.data)[i];
    t->shader.data = &out->materials[m >= 0 ? (size_t)m : mtls.len];
    t->shader.proc = disney_shader_proc;
  }
  free(V.data); free(VT.data); free(VN.data); free(F.data); free(FM.data); free(mtls.data);
  if (!ok) return fail(e, "'%s': a face refers to a vertex that does not exist", path);
  out->has_camera = false;
  return true;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* a small JSON reader (objects, arrays, strings, numbers, true / false / null) for the glTF chunk                  */

typedef enum { J_NULL, J_BOOL, J_NUM, J_STR, J_ARR, J_OBJ } J_Kind;
typedef struct J_Value {
  J_Kind kind;
  double num;
  char  *str;                     /* J_STR: value; object members: `key` */
  char  *key;
  struct J_Value *first, *next;   /* children of arrays / objects */
} J_Value;

typedef struct { char const *p, *end; bool ok; } J_Parser;

static void j_ws(J_Parser *P) { while (P->p < P->end && isspace((unsigned char)*P->p)) P->p++; }

static char *j_string(J_Parser *P) {
  if (P->p >= P->end || *P->p != '"') { P->ok = false; return NULL; }
  P->p++;
  char const *s = P->p;
  while (P->p < P->end && *P->p != '"') { if (*P->p == '\\') P->p++; P->p++; }
  if (P->p >= P->end) { P->ok = false; return NULL; }
  size_t n = (size_t)(P->p - s);
  char *out = malloc(n + 1), *o = out;
  for (char const *c = s; c < s + n; c++) {
    if (*c == '\\' && c + 1 < s + n) { c++; *o++ = (*c == 'n') ? '\n' : (*c == 't') ? '\t' : *c; }   /* (\uXXXX is kept verbatim) */
    else *o++ = *c;
  }
  *o = 0;
  P->p++;
  return out;
}

static J_Value *j_value(J_Parser *P);

static J_Value *j_container(J_Parser *P, bool object) {
  J_Value *v = calloc(1, sizeof *v), **tail = &v->first;
  v->kind = object ? J_OBJ : J_ARR;
  char close = object ? '}' : ']';
  P->p++;
  j_ws(P);
  if (P->p < P->end && *P->p == close) { P->p++; return v; }
  while (P->ok && P->p < P->end) {
    char *key = NULL;
    j_ws(P);
    if (object) {
      key = j_string(P);
      j_ws(P);
      if (!P->ok || P->p >= P->end || *P->p != ':') { P->ok = false; free(key); break; }
      P->p++;
    }
    J_Value *c = j_value(P);
    if (!c) { free(key); break; }
    c->key = key;
    *tail = c;
    tail = &c->next;
    j_ws(P);
    if (P->p < P->end && *P->p == ',') { P->p++; continue; }
    if (P->p < P->end && *P->p == close) { P->p++; return v; }
    P->ok = false;
  }
  P->ok = false;
  return v;
}

static J_Value *j_value(J_Parser *P) {
  j_ws(P);
  if (P->p >= P->end) { P->ok = false; return NULL; }
  char c = *P->p;
  if (c == '{' || c == '[') return j_container(P, c == '{');
  J_Value *v = calloc(1, sizeof *v);
  if (c == '"') { v->kind = J_STR; v->str = j_string(P); return v; }
  if (!strncmp(P->p, "true", 4)) { v->kind = J_BOOL; v->num = 1; P->p += 4; return v; }
  if (!strncmp(P->p, "false", 5)) { v->kind = J_BOOL; v->num = 0; P->p += 5; return v; }
  if (!strncmp(P->p, "null", 4)) { v->kind = J_NULL; P->p += 4; return v; }
  char *endp = NULL;
  v->kind = J_NUM;
  v->num = strtod(P->p, &endp);
  if (endp == P->p) { P->ok = false; free(v); return NULL; }
  P->p = endp;
  return v;
}

static void j_free(J_Value *v) {
  while (v) {
    J_Value *n = v->next;
    j_free(v->first);
    free(v->str);
    free(v->key);
    free(v);
    v = n;
  }
}

static J_Value *j_get(J_Value const *o, char const *key) {
  if (!o || o->kind != J_OBJ) return NULL;
  for (J_Value *c = o->first; c; c = c->next)
    if (c->key && !strcmp(c->key, key)) return c;
  return NULL;
}
static J_Value *j_at(J_Value const *a, int i) {
  if (!a || a->kind != J_ARR) return NULL;
  J_Value *c = a->first;
  while (c && i-- > 0) c = c->next;
  return c;
}
static int j_len(J_Value const *a) {
  int n = 0;
  if (a && (a->kind == J_ARR || a->kind == J_OBJ)) for (J_Value *c = a->first; c; c = c->next) n++;
  return n;
}
static double j_num(J_Value const *v, double dflt) { return (v && (v->kind == J_NUM || v->kind == J_BOOL)) ? v->num : dflt; }
static int    j_int(J_Value const *v, int dflt) { return (v && v->kind == J_NUM) ? (int)v->num : dflt; }

/* ------------------------------------------------------------------------------------------------------------- */
/* glTF 2.0 (loaders.py: _read_gltf, _accessor, _node_local, load_gltf; driver.c:589-683)                           */

typedef struct { byte *data; size_t len; } Blob;

typedef struct {
  J_Value *js;
  Blob    *buffers;
  bool    *owned;                 /* buffer i was read from its own file (not a chunk of the .glb) */
  int      n_buffers;
} Gltf;

static bool gltf_open(char const *path, Gltf *g, byte **file_data, Err *e) {
  size_t len;
  byte *data = read_file(path, &len);
  if (!data) return fail(e, "cannot read '%s'", path);
  *file_data = data;
  char base[4096];
  dir_of(path, base, sizeof base);
  char const *json = NULL;
  size_t json_len = 0;
  Blob bins[8];
  int n_bins = 0;
  if (len >= 12 && !memcmp(data, "glTF", 4)) {
    u32 total;
    memcpy(&total, data + 8, 4);
    size_t off = 12;
    while (off + 8 <= len && off < total) {
      u32 clen, ctype;
      memcpy(&clen, data + off, 4);
      memcpy(&ctype, data + off + 4, 4);
      if (off + 8 + clen > len) return fail(e, "'%s': truncated chunk", path);
      if (ctype == 0x4E4F534Au) { json = (char const *)data + off + 8; json_len = clen; }
      else if (ctype == 0x004E4942u && n_bins < 8) { bins[n_bins].data = data + off + 8; bins[n_bins].len = clen; n_bins++; }
      off += 8 + (size_t)clen;
    }
  } else {
    json = (char const *)data;
    json_len = len;
  }
  if (!json) return fail(e, "'%s': no JSON chunk", path);
  J_Parser P = { json, json + json_len, true };
  g->js = j_value(&P);
  if (!g->js || !P.ok) return fail(e, "'%s': JSON does not parse", path);
  J_Value *bufs = j_get(g->js, "buffers");
  g->n_buffers = j_len(bufs);
  g->buffers = calloc((size_t)(g->n_buffers > 0 ? g->n_buffers : 1), sizeof *g->buffers);
  g->owned = calloc((size_t)(g->n_buffers > 0 ? g->n_buffers : 1), sizeof(bool));
  for (int i = 0; i < g->n_buffers; i++) {
    J_Value *uri = j_get(j_at(bufs, i), "uri");
    if (uri && uri->kind == J_STR) {
      char full[4096 + 1024];
      snprintf(full, sizeof full, "%s%s", base, uri->str);
      g->buffers[i].data = read_file(full, &g->buffers[i].len);
      if (!g->buffers[i].data) return fail(e, "cannot read buffer '%s'", full);
      g->owned[i] = true;
    } else if (i < n_bins) {
      g->buffers[i] = bins[i];
    } else {
      return fail(e, "'%s': buffer %d has no data", path, i);
    }
  }
  return true;
}

static int comp_size(int ct) { return ct == 5120 || ct == 5121 ? 1 : ct == 5122 || ct == 5123 ? 2 : 4; }
static int type_comps(char const *t) {
  return !strcmp(t, "SCALAR") ? 1 : !strcmp(t, "VEC2") ? 2 : !strcmp(t, "VEC3") ? 3 : !strcmp(t, "VEC4") ? 4 : !strcmp(t, "MAT4") ? 16 : 0;
}

/* element (i, c) of accessor `idx` as double; *count / *ncomp on request */
typedef struct { byte const *base; int stride, ct, nc, count; } Accessor;
static bool accessor_open(Gltf const *g, int idx, Accessor *a, Err *e) {
  J_Value *acc = j_at(j_get(g->js, "accessors"), idx);
  J_Value *bv = acc ? j_at(j_get(g->js, "bufferViews"), j_int(j_get(acc, "bufferView"), -1)) : NULL;
  J_Value *type = j_get(acc, "type");
  if (!acc || !bv || !type || type->kind != J_STR) return fail(e, "accessor %d is not usable (sparse accessors are not supported)", idx);
  int buf = j_int(j_get(bv, "buffer"), -1);
  if (buf < 0 || buf >= g->n_buffers) return fail(e, "accessor %d: bad buffer", idx);
  a->ct = j_int(j_get(acc, "componentType"), 0);
  a->nc = type_comps(type->str);
  a->count = j_int(j_get(acc, "count"), 0);
  size_t start = (size_t)j_num(j_get(bv, "byteOffset"), 0) + (size_t)j_num(j_get(acc, "byteOffset"), 0);
  int elem = comp_size(a->ct) * a->nc;
  a->stride = j_int(j_get(bv, "byteStride"), 0);
  if (!a->stride) a->stride = elem;
  if (!a->nc || a->count < 0 || (a->count > 0 && start + (size_t)(a->count - 1) * (size_t)a->stride + (size_t)elem > g->buffers[buf].len))
    return fail(e, "accessor %d reaches outside its buffer", idx);
  a->base = g->buffers[buf].data + start;
  return true;
}
static f32 accessor_f32(Accessor const *a, int i, int c) {
  byte const *p = a->base + (size_t)i * (size_t)a->stride + (size_t)c * (size_t)comp_size(a->ct);
  switch (a->ct) {
  case 5126: { f32 v; memcpy(&v, p, 4); return v; }
  case 5125: { u32 v; memcpy(&v, p, 4); return (f32)v; }
  case 5123: { uint16_t v; memcpy(&v, p, 2); return (f32)v; }
  case 5122: { int16_t v; memcpy(&v, p, 2); return (f32)v; }
  case 5121: return (f32)*p;
  default: return (f32)*(int8_t const *)p;
  }
}
static i64 accessor_index(Accessor const *a, int i) {
  byte const *p = a->base + (size_t)i * (size_t)a->stride;
  switch (a->ct) {
  case 5125: { u32 v; memcpy(&v, p, 4); return v; }
  case 5123: { uint16_t v; memcpy(&v, p, 2); return v; }
  case 5121: return *p;
  default: return 0;
  }
}

static void node_local(J_Value const *node, f32 m[4][4]) {
  J_Value *mat = j_get(node, "matrix");
  if (mat && j_len(mat) == 16) {
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) m[i][j] = (f32)j_num(j_at(mat, j * 4 + i), 0);        /* JSON is column-major */
    return;
  }
  f32 t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
  J_Value *jt = j_get(node, "translation"), *jq = j_get(node, "rotation"), *js = j_get(node, "scale");
  for (int k = 0; k < 3; k++) { if (jt) t[k] = (f32)j_num(j_at(jt, k), 0); if (js) s[k] = (f32)j_num(j_at(js, k), 1); }
  for (int k = 0; k < 4; k++) if (jq) q[k] = (f32)j_num(j_at(jq, k), k == 3);
  trs_matrix(t, q, s, m);
}

static void visit_node(J_Value const *nodes, int i, f32 const parent[4][4], f32 (*globals)[4][4], bool *seen) {
  J_Value const *node = j_at(nodes, i);
  if (!node || seen[i]) return;
  f32 local[4][4];
  node_local(node, local);
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) {                      /* fp32 product, k ascending */
      f32 acc = parent[r][0] * local[0][c];
      for (int k = 1; k < 4; k++) acc = acc + parent[r][k] * local[k][c];
      globals[i][r][c] = acc;
    }
  seen[i] = true;
  J_Value *ch = j_get(node, "children");
  for (int k = 0; k < j_len(ch); k++) visit_node(nodes, j_int(j_at(ch, k), -1), (f32 const (*)[4])globals[i], globals, seen);
}

static int texture_source(Gltf const *g, J_Value const *ref) {        /* {"index": t} -> textures[t].source, -1 if absent */
  if (!ref) return -1;
  J_Value *tex = j_at(j_get(g->js, "textures"), j_int(j_get(ref, "index"), -1));
  return j_int(j_get(tex, "source"), -1);
}

/* image `im` of a glTF file: the RT8I side file when there is one (any codec, written by tools/extract_textures.py), else the
 * embedded or referenced stream itself: baseline JPEG (rt_jpeg.c) or PNG (rt_png.c), the bytes PIL hands the Python loader */
static bool load_gltf_image(Gltf const *g, char const *path, int im, Image *img, Err *e) {
  char p[4096];
  snprintf(p, sizeof p, "%s.image%d.rgb8", path, im);
  FILE *side = fopen(p, "rb");
  if (side) { fclose(side); return load_rgb8(p, img, e); }
  J_Value *ji = j_at(j_get(g->js, "images"), im);
  J_Value *bvr = j_get(ji, "bufferView"), *uri = j_get(ji, "uri");
  byte const *bytes = NULL;
  byte *owned = NULL;
  size_t n = 0;
  if (bvr) {
    J_Value *bv = j_at(j_get(g->js, "bufferViews"), j_int(bvr, -1));
    int b = j_int(j_get(bv, "buffer"), -1);
    size_t off = (size_t)j_num(j_get(bv, "byteOffset"), 0), len = (size_t)j_num(j_get(bv, "byteLength"), 0);
    if (!bv || b < 0 || b >= g->n_buffers || off + len > g->buffers[b].len) return fail(e, "image %d: bad bufferView", im);
    bytes = g->buffers[b].data + off;
    n = len;
  } else if (uri && uri->kind == J_STR && strncmp(uri->str, "data:", 5) != 0) {
    char base[4096], full[8300];
    dir_of(path, base, sizeof base);
    snprintf(full, sizeof full, "%s%s", base, uri->str);
    owned = read_file(full, &n);
    if (!owned) return fail(e, "image %d: cannot read '%s'", im, full);
    bytes = owned;
  } else {
    return fail(e, "image %d: no side file '%s' and no bufferView / file uri to decode", im, p);
  }
  char msg[200] = "";
  bool ok = decode_image(bytes, n, img, msg, sizeof msg);
  free(owned);
  if (!ok) return fail(e, "image %d: %s; write '%s' with tools/extract_textures.py instead", im, msg, p);
  return true;
}

static bool load_gltf(char const *path, RT_Model *out, Err *e) {
  Gltf g;
  memset(&g, 0, sizeof g);
  byte *file_data = NULL;
  bool ok = gltf_open(path, &g, &file_data, e);
  Vec T = {0};
  f32 (*globals)[4][4] = NULL;
  bool *seen = NULL;
  if (!ok) goto done;

  /* images, in the order of the "images" array; loaded when a material uses them (load_gltf_image) */
  out->n_images = j_len(j_get(g.js, "images"));
  out->images = calloc((size_t)(out->n_images > 0 ? out->n_images : 1), sizeof *out->images);
  bool *image_loaded = calloc((size_t)(out->n_images > 0 ? out->n_images : 1), sizeof(bool));

  /* materials (driver.c:628-660) */
  J_Value *jm = j_get(g.js, "materials");
  int n_mat = j_len(jm);
  out->materials = calloc((size_t)n_mat + 1, sizeof *out->materials);        /* + the default one, used on demand */
  out->n_materials = n_mat;
  int tex_of[4];
  for (int i = 0; i < n_mat && ok; i++) {
    J_Value *m = j_at(jm, i), *pbr = j_get(m, "pbrMetallicRoughness");
    J_Value *bc = j_get(pbr, "baseColorFactor"), *em = j_get(m, "emissiveFactor");
    PBR_Shader_Data d = material_default((f32)j_num(j_at(bc, 0), 1), (f32)j_num(j_at(bc, 1), 1), (f32)j_num(j_at(bc, 2), 1),
                                         (f32)j_num(j_get(pbr, "roughnessFactor"), 1.0), (f32)j_num(j_get(pbr, "metallicFactor"), 1.0));
    d.emission.x = (f32)j_num(j_at(em, 0), 0); d.emission.y = (f32)j_num(j_at(em, 1), 0); d.emission.z = (f32)j_num(j_at(em, 2), 0);
    J_Value *sheen = j_get(j_get(j_get(m, "extensions"), "KHR_materials_sheen"), "sheenColorFactor");
    f32 s0 = (f32)j_num(j_at(sheen, 0), 0), s1 = (f32)j_num(j_at(sheen, 1), 0), s2 = (f32)j_num(j_at(sheen, 2), 0);
    d.sheen = 0.2126f * s0 + 0.7152f * s1 + 0.0722f * s2;
    J_Value *nt = j_get(m, "normalTexture");
    if (nt) d.normal_map_strength = (f32)j_num(j_get(nt, "scale"), 1.0);
    tex_of[0] = texture_source(&g, j_get(pbr, "baseColorTexture"));
    tex_of[1] = texture_source(&g, nt);
    tex_of[2] = texture_source(&g, j_get(pbr, "metallicRoughnessTexture"));
    tex_of[3] = texture_source(&g, j_get(m, "emissiveTexture"));
    Image **slots[4] = { &d.texture_albedo, &d.texture_normal, &d.texture_metal_roughness, &d.texture_emission };
    for (int k = 0; k < 4 && ok; k++) {
      int im = tex_of[k];
      if (im < 0) continue;
      if (im >= out->n_images) { ok = fail(e, "material %d refers to image %d of %d", i, im, (int)out->n_images); break; }
      if (!image_loaded[im]) { ok = load_gltf_image(&g, path, im, &out->images[im], e); image_loaded[im] = ok; }
      *slots[k] = &out->images[im];
    }
    out->materials[i] = d;
  }
  free(image_loaded);
  if (!ok) goto done;

  /* node transforms */
  J_Value *nodes = j_get(g.js, "nodes");
  int n_nodes = j_len(nodes);
  globals = calloc((size_t)(n_nodes > 0 ? n_nodes : 1), sizeof *globals);
  seen = calloc((size_t)(n_nodes > 0 ? n_nodes : 1), sizeof(bool));
  f32 ident[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  J_Value *scenes = j_get(g.js, "scenes");
  if (scenes) {
    J_Value *roots = j_get(j_at(scenes, j_int(j_get(g.js, "scene"), 0)), "nodes");
    for (int k = 0; k < j_len(roots); k++) visit_node(nodes, j_int(j_at(roots, k), -1), ident, globals, seen);
  } else {
    for (int k = 0; k < n_nodes; k++) visit_node(nodes, k, ident, globals, seen);
  }

  /* first perspective camera node (driver.c:599-612) */
  out->has_camera = false;
  for (int i = 0; i < n_nodes && !out->has_camera; i++) {
    J_Value *node = j_at(nodes, i), *cref = j_get(node, "camera");
    if (!cref || !seen[i]) continue;
    J_Value *cam = j_at(j_get(g.js, "cameras"), j_int(cref, -1)), *type = j_get(cam, "type");
    if (!type || type->kind != J_STR || strcmp(type->str, "perspective")) continue;
    memset(&out->camera, 0, sizeof out->camera);
    set_camera(&out->camera, (f32 const (*)[4])globals[i], (f32)j_num(j_get(j_get(cam, "perspective"), "yfov"), 0));
    out->has_camera = true;
  }

  /* meshes -> world-space triangles */
  int default_mat = -1;
  J_Value *meshes = j_get(g.js, "meshes");
  for (int i = 0; i < n_nodes && ok; i++) {
    J_Value *node = j_at(nodes, i), *mref = j_get(node, "mesh");
    if (!mref || !seen[i]) continue;
    double m3[3][3], t3[3], inv[3][3];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) m3[r][c] = (double)globals[i][r][c]; t3[r] = (double)globals[i][r][3]; }
    {   /* inverse of the 3x3 (normals transform by its transpose); cofactors / determinant in double */
      double a = m3[0][0], b = m3[0][1], c = m3[0][2], d = m3[1][0], ee = m3[1][1], f = m3[1][2], gg = m3[2][0], h = m3[2][1], k = m3[2][2];
      double det = a * (ee * k - f * h) - b * (d * k - f * gg) + c * (d * h - ee * gg);
      double id = 1.0 / det;
      inv[0][0] = (ee * k - f * h) * id; inv[0][1] = (c * h - b * k) * id; inv[0][2] = (b * f - c * ee) * id;
      inv[1][0] = (f * gg - d * k) * id; inv[1][1] = (a * k - c * gg) * id; inv[1][2] = (c * d - a * f) * id;
      inv[2][0] = (d * h - ee * gg) * id; inv[2][1] = (b * gg - a * h) * id; inv[2][2] = (a * ee - b * d) * id;
    }
    J_Value *prims = j_get(j_at(meshes, j_int(mref, -1)), "primitives");
    for (int p = 0; p < j_len(prims) && ok; p++) {
      J_Value *prim = j_at(prims, p), *att = j_get(prim, "attributes");
      if (j_int(j_get(prim, "mode"), 4) != 4) continue;
      Accessor pos, nrm, uv, ind;
      bool has_n = j_get(att, "NORMAL") != NULL, has_uv = j_get(att, "TEXCOORD_0") != NULL, has_i = j_get(prim, "indices") != NULL;
      ok = j_get(att, "POSITION") ? accessor_open(&g, j_int(j_get(att, "POSITION"), -1), &pos, e)
                                  : fail(e, "'%s': a primitive without a POSITION attribute", path);
      if (ok && has_n) ok = accessor_open(&g, j_int(j_get(att, "NORMAL"), -1), &nrm, e);
      if (ok && has_uv) ok = accessor_open(&g, j_int(j_get(att, "TEXCOORD_0"), -1), &uv, e);
      if (ok && has_i) ok = accessor_open(&g, j_int(j_get(prim, "indices"), -1), &ind, e);
      if (!ok) break;
      int mat = j_int(j_get(prim, "material"), -1);
      if (mat < 0 || mat >= n_mat) {
        if (default_mat < 0) {                          /* appended once: base (1,1,1), roughness 1, metalness 1 */
          default_mat = n_mat;
          out->materials[n_mat] = material_default(1, 1, 1, 1.0f, 1.0f);
          out->n_materials = n_mat + 1;
        }
        mat = default_mat;
      }
      int n_idx = has_i ? ind.count : pos.count;
      for (int f0 = 0; f0 + 3 <= n_idx; f0 += 3) {
        Triangle *t = vec_push(&T, sizeof(Triangle));
        if (!t) { ok = fail(e, "out of memory"); break; }
        memset(t, 0, sizeof *t);
        for (int k = 0; k < 3; k++) {
          i64 vi = has_i ? accessor_index(&ind, f0 + k) : f0 + k;
          if (vi < 0 || vi >= pos.count) { ok = fail(e, "index %lld outside %d vertices", (long long)vi, pos.count); break; }
          double px = accessor_f32(&pos, (int)vi, 0), py = accessor_f32(&pos, (int)vi, 1), pz = accessor_f32(&pos, (int)vi, 2);
          for (int r = 0; r < 3; r++) t->positions[k].data[r] = (f32)(px * m3[r][0] + py * m3[r][1] + pz * m3[r][2] + t3[r]);
          if (has_n && vi < nrm.count) {
            double nx = accessor_f32(&nrm, (int)vi, 0), ny = accessor_f32(&nrm, (int)vi, 1), nz = accessor_f32(&nrm, (int)vi, 2);
            double w[3];
            for (int c = 0; c < 3; c++) w[c] = nx * inv[0][c] + ny * inv[1][c] + nz * inv[2][c];      /* n . inverse(m3) */
            double len = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            if (!(len > 1e-30)) len = 1e-30;
            for (int c = 0; c < 3; c++) t->normals[k].data[c] = (f32)(w[c] / len);
          }
          if (has_uv && vi < uv.count) { t->tex_coords[k].x = accessor_f32(&uv, (int)vi, 0); t->tex_coords[k].y = accessor_f32(&uv, (int)vi, 1); }
        }
        if (!ok) break;
        if (!has_n) {                                    /* face normal of the world-space triangle, fp32 */
          f32 e1[3], e2[3], fn[3];
          for (int a = 0; a < 3; a++) { e1[a] = t->positions[1].data[a] - t->positions[0].data[a]; e2[a] = t->positions[2].data[a] - t->positions[0].data[a]; }
          fn[0] = e1[1] * e2[2] - e1[2] * e2[1];
          fn[1] = e1[2] * e2[0] - e1[0] * e2[2];
          fn[2] = e1[0] * e2[1] - e1[1] * e2[0];
          f32 norm = sqrtf(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]);
          if (!(norm > 1e-30f)) norm = 1e-30f;
          for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) t->normals[k].data[a] = fn[a] / norm;
        }
        t->shader.data = (rawptr)(size_t)mat;            /* index for now: the material array is final below */
        t->shader.proc = disney_shader_proc;
      }
    }
  }
  if (ok && T.len == 0) ok = fail(e, "'%s': no triangles", path);
  if (ok) {
    out->triangles = T.data;
    out->n_triangles = (isize)T.len;
    T.data = NULL;
    for (isize i = 0; i < out->n_triangles; i++) out->triangles[i].shader.data = &out->materials[(size_t)out->triangles[i].shader.data];
  }
done:
  free(T.data);
  free(globals);
  free(seen);
  for (int i = 0; i < g.n_buffers && g.buffers && g.owned; i++)
    if (g.owned[i]) free(g.buffers[i].data);        /* (chunks of the .glb live inside file_data) */
  free(g.buffers);
  free(g.owned);
  j_free(g.js);
  free(file_data);
  return ok;
}

/* ------------------------------------------------------------------------------------------------------------- */

static bool has_suffix(char const *s, char const *suf) {
  size_t n = strlen(s), m = strlen(suf);
  if (n < m) return false;
  for (size_t i = 0; i < m; i++)
    if (tolower((unsigned char)s[n - m + i]) != suf[i]) return false;
  return true;
}

bool rt_model_load(char const *path, RT_Model *out, char *err, size_t err_len) {
  Err e = { err, err_len };
  if (!path || !out) return fail(&e, "rt_model_load: NULL argument");
  memset(out, 0, sizeof *out);
  bool ok;
  if (has_suffix(path, ".obj")) ok = load_obj(path, out, &e);                       /* driver.c:685-728 */
  else if (has_suffix(path, ".glb") || has_suffix(path, ".gltf")) ok = load_gltf(path, out, &e);
  else ok = fail(&e, "Unrecognized file type: '%s'", path);
  if (!ok) rt_model_free(out);
  return ok;
}

void rt_model_free(RT_Model *m) {
  if (!m) return;
  for (isize i = 0; i < m->n_images; i++) free(m->images ? m->images[i].pixels.data : NULL);
  free(m->images);
  free(m->materials);
  free(m->triangles);
  memset(m, 0, sizeof *m);
}
