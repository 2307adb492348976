/* rt_jpeg.c -- a baseline JPEG decoder for the C host (examples/rt_model.c), so that a .glb with embedded JPEG textures
 * (helmet.glb: four 2048 x 2048 baseline 4:2:0 images) loads without a preparation step.  The reference decodes through
 * codin's stb_image_load_bytes (driver.c:106-116, 621), which is not in the reference tree; the benchmark's Python loader
 * decodes with PIL, i.e. libjpeg(-turbo) with its defaults.  This decoder restates THOSE defaults so that the C host and the
 * Python loader hand the renderer the same texels, byte for byte (tests/test_c_loader.py compares them):
 *
 *   * sequential baseline DCT (SOF0), 8 bits, 1 or 3 components, Huffman, optional restart intervals;
 *   * the "islow" inverse DCT of the IJG library (jidctint.c: Loeffler-Ligtenberg-Moshovitz, 13-bit constants, two passes
 *     with PASS1_BITS = 2);
 *   * "fancy" chroma upsampling (jdsample.c: the triangle filter, 3/4 + 1/4 per direction, with the IJG's alternating
 *     rounding biases; rows above the first / below the last are the edge rows themselves) for 2x2 and 2x1 subsampling of
 *     planes more than two samples wide, replication for anything else (libjpeg-turbo also filters 1x2, which no encoder
 *     here writes and no test can pin: replicated);
 *   * YCbCr -> RGB by the IJG's 16-bit fixed-point tables (jdcolor.c).
 *
 * Not a general JPEG library: no progressive, arithmetic, 12-bit, CMYK or 16-bit-table support -- such files fail with a
 * message and the host falls back to the RT8I side files of tools/extract_textures.py.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_model.h"

typedef struct {
  uint8_t  bits[17];
  uint8_t  vals[256];
  int      maxcode[18];      /* largest code of length k (-1 if none) */
  int      valptr[17];
  int      mincode[17];
  int      present;
} Huff;

typedef struct {
  int id, h, v, tq, td, ta;
  int blocks_w, blocks_h;    /* blocks per row / column of the padded plane */
  int down_w, down_h;        /* ceil(width * h / hmax), ceil(height * v / vmax): the samples that count */
  int pred;
  uint8_t *plane;            /* blocks_w * 8 x blocks_h * 8 samples */
} Comp;

typedef struct {
  const uint8_t *p, *end;
  uint32_t bitbuf;
  int      bitcnt;
  int      marker;           /* a marker met inside the entropy-coded data */
} Bits;

static const uint8_t ZIGZAG[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                   15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static bool jfail(char *err, size_t n, const char *msg) {
  if (err && n) snprintf(err, n, "rt_jpeg: %s", msg);
  return false;
}

static void huff_build(Huff *h) {
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    h->valptr[l] = k;
    h->mincode[l] = code;
    code += h->bits[l];
    k += h->bits[l];
    h->maxcode[l] = h->bits[l] ? code - 1 : -1;
    code <<= 1;
  }
  h->maxcode[17] = 0x7fffffff;
  h->present = 1;
}

static void fill_bits(Bits *b) {
  while (b->bitcnt <= 24) {
    int c = 0;
    if (b->marker == 0 && b->p < b->end) {
      c = *b->p++;
      if (c == 0xFF) {
        int c2 = b->p < b->end ? *b->p : 0;
        if (c2 == 0) b->p++;                       /* stuffed zero */
        else { b->marker = c2; b->p++; c = 0; }    /* a marker: feed zeros from here on */
      }
    }
    b->bitbuf |= (uint32_t)c << (24 - b->bitcnt);
    b->bitcnt += 8;
  }
}

static int get_bits(Bits *b, int n) {
  if (n == 0) return 0;
  if (b->bitcnt < n) fill_bits(b);
  int v = (int)(b->bitbuf >> (32 - n));
  b->bitbuf <<= n;
  b->bitcnt -= n;
  return v;
}

static int huff_decode(Bits *b, const Huff *h) {
  int code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | get_bits(b, 1);
    if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l]) return h->vals[h->valptr[l] + code - h->mincode[l]];
  }
  return -1;
}

static int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

/* jidctint.c (IJG release 6b): dequantise + inverse DCT of one block into out[8 rows][stride].  64-bit temporaries: the
 * library's 32-bit ones give the same values on every valid stream and overflow (undefined in C) on damaged ones. */
typedef int64_t idct_t;
#define CONST_BITS 13
#define PASS1_BITS 2
#define DESCALE(x, n) (((x) + ((idct_t)1 << ((n) - 1))) >> (n))
static void idct_islow(const int16_t *coef, const uint16_t *q, uint8_t *out, int stride) {
  idct_t ws[64];
  for (int c = 0; c < 8; c++) {
    const int16_t *in = coef + c;
    const uint16_t *qq = q + c;
    idct_t z2 = in[16] * qq[16], z3 = in[48] * qq[48];
    idct_t z1 = (z2 + z3) * 4433;
    idct_t tmp2 = z1 + z3 * -15137, tmp3 = z1 + z2 * 6270;
    z2 = in[0] * qq[0];
    z3 = in[32] * qq[32];
    idct_t tmp0 = (z2 + z3) * (1 << CONST_BITS), tmp1 = (z2 - z3) * (1 << CONST_BITS);
    idct_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[56] * qq[56];
    tmp1 = in[40] * qq[40];
    tmp2 = in[24] * qq[24];
    tmp3 = in[8] * qq[8];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    idct_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    idct_t *w = ws + c;
    w[0] = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
    w[56] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
    w[8] = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
    w[48] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
    w[16] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
    w[40] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
    w[24] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
    w[32] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
  }
  for (int r = 0; r < 8; r++) {
    const idct_t *w = ws + r * 8;
    idct_t z2 = w[2], z3 = w[6];
    idct_t z1 = (z2 + z3) * 4433;
    idct_t tmp2 = z1 + z3 * -15137, tmp3 = z1 + z2 * 6270;
    idct_t tmp0 = (w[0] + w[4]) * (1 << CONST_BITS), tmp1 = (w[0] - w[4]) * (1 << CONST_BITS);
    idct_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    idct_t z4 = tmp1 + tmp3, z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const idct_t v[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
    uint8_t *o = out + (size_t)r * stride;
    for (int k = 0; k < 8; k++) {
      idct_t s = DESCALE(v[k], CONST_BITS + PASS1_BITS + 3) + 128;
      o[k] = (uint8_t)(s < 0 ? 0 : s > 255 ? 255 : s);
    }
  }
}

/* jdsample.c h2v2_fancy_upsample: plane (dw x dh samples that count, row stride `stride`) -> out (2 dw x 2 dh) */
static void upsample_h2v2(const uint8_t *plane, int stride, int dw, int dh, uint8_t *out, int ostride) {
  for (int y = 0; y < dh; y++) {
    const uint8_t *in0 = plane + (size_t)y * stride;
    for (int v = 0; v < 2; v++) {
      const int yn = v == 0 ? (y > 0 ? y - 1 : 0) : (y + 1 < dh ? y + 1 : dh - 1);       /* nearer neighbour row; edges: themselves */
      const uint8_t *in1 = plane + (size_t)yn * stride;
      uint8_t *o = out + (size_t)(2 * y + v) * ostride;
      int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
      o[0] = (uint8_t)((thiscol * 4 + 8) >> 4);
      o[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
      lastcol = thiscol;
      thiscol = nextcol;
      for (int x = 1; x < dw - 1; x++) {
        nextcol = in0[x + 1] * 3 + in1[x + 1];
        o[2 * x] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        o[2 * x + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol;
        thiscol = nextcol;
      }
      o[2 * (dw - 1)] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
      o[2 * (dw - 1) + 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
    }
  }
}

/* jdsample.c h2v1_fancy_upsample */
static void upsample_h2v1(const uint8_t *plane, int stride, int dw, int dh, uint8_t *out, int ostride) {
  for (int y = 0; y < dh; y++) {
    const uint8_t *in = plane + (size_t)y * stride;
    uint8_t *o = out + (size_t)y * ostride;
    o[0] = in[0];
    o[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
    for (int x = 1; x < dw - 1; x++) {
      o[2 * x] = (uint8_t)((in[x] * 3 + in[x - 1] + 1) >> 2);
      o[2 * x + 1] = (uint8_t)((in[x] * 3 + in[x + 1] + 2) >> 2);
    }
    o[2 * (dw - 1)] = (uint8_t)((in[dw - 1] * 3 + in[dw - 2] + 1) >> 2);
    o[2 * (dw - 1) + 1] = in[dw - 1];
  }
}

static void upsample_replicate(const uint8_t *plane, int stride, int hs, int vs, int ow, int oh, uint8_t *out, int ostride) {
  for (int y = 0; y < oh; y++)
    for (int x = 0; x < ow; x++) out[(size_t)y * ostride + x] = plane[(size_t)(y / vs) * stride + x / hs];
}

bool rt_jpeg_decode(const unsigned char *data, size_t n, Image *out, char *err, size_t err_len) {
  uint16_t qt[4][64];
  int      qt_present[4] = {0, 0, 0, 0};
  Huff     dc[4], ac[4];
  Comp     comp[3];
  int      ncomp = 0, width = 0, height = 0, hmax = 1, vmax = 1, restart = 0;
  bool     ok = false;
  memset(dc, 0, sizeof dc);
  memset(ac, 0, sizeof ac);
  memset(comp, 0, sizeof comp);
  memset(out, 0, sizeof *out);
  if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) return jfail(err, err_len, "not a JPEG stream");
  size_t i = 2;
  const uint8_t *scan = NULL;
  while (i + 4 <= n) {
    if (data[i] != 0xFF) return jfail(err, err_len, "marker expected");
    int m = data[i + 1];
    if (m == 0xFF) { i++; continue; }
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
    size_t len = ((size_t)data[i + 2] << 8) | data[i + 3];
    if (len < 2 || i + 2 + len > n) return jfail(err, err_len, "truncated segment");
    const uint8_t *s = data + i + 4;
    size_t sl = len - 2;
    if (m == 0xDB) {
      while (sl >= 65) {
        int pq = s[0] >> 4, tq = s[0] & 15;
        if (pq != 0 || tq > 3) return jfail(err, err_len, "16-bit quantisation tables are not supported");
        for (int k = 0; k < 64; k++) qt[tq][ZIGZAG[k]] = s[1 + k];
        qt_present[tq] = 1;
        s += 65;
        sl -= 65;
      }
    } else if (m == 0xC4) {
      while (sl >= 17) {
        int tc = s[0] >> 4, th = s[0] & 15;
        if (tc > 1 || th > 3) return jfail(err, err_len, "bad Huffman table id");
        Huff *h = tc ? &ac[th] : &dc[th];
        int total = 0;
        h->bits[0] = 0;
        for (int k = 1; k <= 16; k++) { h->bits[k] = s[k]; total += s[k]; }
        if (total > 256 || sl < (size_t)(17 + total)) return jfail(err, err_len, "bad Huffman table");
        memcpy(h->vals, s + 17, (size_t)total);
        huff_build(h);
        s += 17 + total;
        sl -= (size_t)(17 + total);
      }
    } else if (m == 0xC0 || m == 0xC1) {
      if (sl < 6 || s[0] != 8) return jfail(err, err_len, "only 8-bit baseline JPEG is supported");
      height = (s[1] << 8) | s[2];
      width = (s[3] << 8) | s[4];
      ncomp = s[5];
      if ((ncomp != 1 && ncomp != 3) || width <= 0 || height <= 0 || sl < (size_t)(6 + 3 * ncomp)) return jfail(err, err_len, "unsupported frame header");
      for (int k = 0; k < ncomp; k++) {
        comp[k].id = s[6 + 3 * k];
        comp[k].h = s[7 + 3 * k] >> 4;
        comp[k].v = s[7 + 3 * k] & 15;
        comp[k].tq = s[8 + 3 * k];
        if (comp[k].h < 1 || comp[k].h > 4 || comp[k].v < 1 || comp[k].v > 4 || comp[k].tq > 3) return jfail(err, err_len, "bad sampling factors");
        if (comp[k].h > hmax) hmax = comp[k].h;
        if (comp[k].v > vmax) vmax = comp[k].v;
      }
      for (int k = 0; k < ncomp; k++)
        if (hmax % comp[k].h || vmax % comp[k].v) return jfail(err, err_len, "fractional sampling ratios are not supported");
      if ((size_t)width * (size_t)height > ((size_t)1 << 28)) return jfail(err, err_len, "image larger than 2^28 pixels");
    } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return jfail(err, err_len, "progressive / lossless / arithmetic JPEG is not supported (use the RT8I side files)");
    } else if (m == 0xDD) {
      if (sl >= 2) restart = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {
      if (ncomp == 0 || sl < (size_t)(1 + 2 * s[0] + 3) || s[0] != ncomp) return jfail(err, err_len, "unsupported scan header (one interleaved scan expected)");
      for (int k = 0; k < ncomp; k++) {
        int cid = s[1 + 2 * k], found = -1;
        for (int c = 0; c < ncomp; c++) if (comp[c].id == cid) found = c;
        if (found < 0) return jfail(err, err_len, "scan names an unknown component");
        comp[found].td = s[2 + 2 * k] >> 4;
        comp[found].ta = s[2 + 2 * k] & 15;
        if (comp[found].td > 3 || comp[found].ta > 3) return jfail(err, err_len, "bad Huffman table id in the scan header");
      }
      scan = data + i + 2 + len;
      break;
    }
    i += 2 + len;
  }
  if (!scan) return jfail(err, err_len, "no scan");
  const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
  const int mcus_x = (width + mcu_w - 1) / mcu_w, mcus_y = (height + mcu_h - 1) / mcu_h;
  for (int k = 0; k < ncomp; k++) {
    Comp *c = &comp[k];
    if (!qt_present[c->tq] || !dc[c->td].present || !ac[c->ta].present) { jfail(err, err_len, "a table the scan needs is missing"); goto done; }
    c->blocks_w = mcus_x * c->h;
    c->blocks_h = mcus_y * c->v;
    c->down_w = (width * c->h + hmax - 1) / hmax;
    c->down_h = (height * c->v + vmax - 1) / vmax;
    c->plane = (uint8_t *)malloc((size_t)c->blocks_w * 8 * (size_t)c->blocks_h * 8);
    if (!c->plane) { jfail(err, err_len, "out of memory"); goto done; }
  }
  {
    Bits b;
    memset(&b, 0, sizeof b);
    b.p = scan;
    b.end = data + n;
    int16_t coef[64];
    int     until_restart = restart, next_rst = 0;
    for (int my = 0; my < mcus_y; my++)
      for (int mx = 0; mx < mcus_x; mx++) {
        if (restart && until_restart == 0) {
          /* byte-align, expect RSTn */
          b.bitbuf = 0;
          b.bitcnt = 0;
          if (b.marker == 0) {
            while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++;
            if (b.p + 1 < b.end) b.p += 2;
          } else if (b.marker != 0xD0 + next_rst) { jfail(err, err_len, "restart marker out of sequence"); goto done; }
          b.marker = 0;
          next_rst = (next_rst + 1) & 7;
          until_restart = restart;
          for (int k = 0; k < ncomp; k++) comp[k].pred = 0;
        }
        for (int k = 0; k < ncomp; k++) {
          Comp *c = &comp[k];
          for (int by = 0; by < c->v; by++)
            for (int bx = 0; bx < c->h; bx++) {
              memset(coef, 0, sizeof coef);
              int t = huff_decode(&b, &dc[c->td]);
              if (t < 0 || t > 11) { jfail(err, err_len, "bad DC code"); goto done; }
              int diff = t ? extend(get_bits(&b, t), t) : 0;
              c->pred += diff;
              coef[0] = (int16_t)c->pred;
              for (int kk = 1; kk < 64;) {
                int rs = huff_decode(&b, &ac[c->ta]);
                if (rs < 0) { jfail(err, err_len, "bad AC code"); goto done; }
                int r = rs >> 4, sz = rs & 15;
                if (sz == 0) {
                  if (r != 15) break;
                  kk += 16;
                  continue;
                }
                kk += r;
                if (kk > 63) { jfail(err, err_len, "AC coefficient out of range"); goto done; }
                coef[ZIGZAG[kk]] = (int16_t)extend(get_bits(&b, sz), sz);
                kk++;
              }
              const int px = (mx * c->h + bx) * 8, py = (my * c->v + by) * 8;
              idct_islow(coef, qt[c->tq], c->plane + (size_t)py * c->blocks_w * 8 + px, c->blocks_w * 8);
            }
        }
        if (restart) until_restart--;
      }
  }
  {
    uint8_t *rgb = (uint8_t *)malloc((size_t)width * height * 3);
    uint8_t *full[3] = {NULL, NULL, NULL};
    int      fstride[3] = {0, 0, 0};
    if (!rgb) { jfail(err, err_len, "out of memory"); goto done; }
    bool up_ok = true;
    for (int k = 0; k < ncomp && up_ok; k++) {
      Comp *c = &comp[k];
      const int hs = hmax / c->h, vs = vmax / c->v;
      if (hs == 1 && vs == 1) { full[k] = c->plane; fstride[k] = c->blocks_w * 8; continue; }
      const int ow = c->down_w * hs, oh = c->down_h * vs;
      uint8_t *u = (uint8_t *)malloc((size_t)ow * oh);
      if (!u) { up_ok = false; break; }
      /* jdsample.c jinit_upsampler: the triangle filters only for planes more than two samples wide */
      const bool fancy = c->down_w > 2;
      if (fancy && hs == 2 && vs == 2) upsample_h2v2(c->plane, c->blocks_w * 8, c->down_w, c->down_h, u, ow);
      else if (fancy && hs == 2 && vs == 1) upsample_h2v1(c->plane, c->blocks_w * 8, c->down_w, c->down_h, u, ow);
      else upsample_replicate(c->plane, c->blocks_w * 8, hs, vs, ow, oh, u, ow);
      full[k] = u;
      fstride[k] = ow;
    }
    if (!up_ok) { jfail(err, err_len, "out of memory"); free(rgb); for (int k = 0; k < 3; k++) if (full[k] && full[k] != comp[k].plane) free(full[k]); goto done; }
    if (ncomp == 1) {
      for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
          uint8_t g = full[0][(size_t)y * fstride[0] + x];
          uint8_t *o = rgb + ((size_t)y * width + x) * 3;
          o[0] = o[1] = o[2] = g;
        }
    } else {
      /* jdcolor.c: SCALEBITS 16, FIX(x) = (int)(x * 65536 + 0.5) */
      static int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
      for (int v = 0; v < 256; v++) {
        int32_t x = v - 128;
        cr_r[v] = (91881 * x + 32768) >> 16;
        cb_b[v] = (116130 * x + 32768) >> 16;
        cr_g[v] = -46802 * x;
        cb_g[v] = -22554 * x + 32768;
      }
      for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
          int Y = full[0][(size_t)y * fstride[0] + x], cb = full[1][(size_t)y * fstride[1] + x], cr = full[2][(size_t)y * fstride[2] + x];
          int r = Y + cr_r[cr], g = Y + ((cb_g[cb] + cr_g[cr]) >> 16), bl = Y + cb_b[cb];
          uint8_t *o = rgb + ((size_t)y * width + x) * 3;
          o[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
          o[1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g);
          o[2] = (uint8_t)(bl < 0 ? 0 : bl > 255 ? 255 : bl);
        }
    }
    for (int k = 0; k < 3; k++) if (full[k] && full[k] != comp[k].plane) free(full[k]);
    out->components = 3;
    out->pixel_type = PT_u8;
    out->width = width;
    out->stride = width;
    out->height = height;
    out->pixels.data = rgb;
    out->pixels.len = (isize)width * height * 3;
    ok = true;
  }
done:
  for (int k = 0; k < 3; k++) free(comp[k].plane);
  return ok;
}
