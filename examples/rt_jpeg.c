/* rt_jpeg.c -- a baseline JPEG decoder for the C host (examples/rt_model.c), so that a .glb with embedded JPEG textures
 * (helmet.glb: four 2048 x 2048 baseline 4:2:0 images) loads without a preparation step.  The reference decodes through
 * codin's stb_image_load_bytes (driver.c:106-116, 621), which is not in the reference tree; the benchmark's Python loader
 * decodes with PIL, i.e. libjpeg(-turbo) with its defaults.  This decoder restates THOSE defaults so that the C host and the
 * Python loader hand the renderer the same texels, byte for byte (tests/test_c_loader.py compares them):
 *
 *   * baseline / extended sequential (SOF0 / SOF1) and progressive (SOF2) Huffman streams, 8 bits, 1 or 3 components, any scan
 *     layout (interleaved or one component per scan), optional restart intervals: every scan goes into coefficient buffers of
 *     the whole frame (ITU T.81 F.2.2, G.1.2, G.2: DC / AC first passes, the successive-approximation refinements with their
 *     end-of-band runs), the pixels are made once the last scan is in;
 *   * the "islow" inverse DCT of the IJG library (jidctint.c: Loeffler-Ligtenberg-Moshovitz, 13-bit constants, two passes
 *     with PASS1_BITS = 2);
 *   * "fancy" chroma upsampling (jdsample.c: the triangle filter, 3/4 + 1/4 per direction, with the IJG's alternating
 *     rounding biases; rows above the first / below the last are the edge rows themselves) for 2x2 and 2x1 subsampling of
 *     planes more than two samples wide, replication for anything else (libjpeg-turbo also filters 1x2, which no encoder
 *     here writes and no test can pin: replicated);
 *   * YCbCr -> RGB by the IJG's 16-bit fixed-point tables (jdcolor.c).
 *
 * Not a general JPEG library: no arithmetic coding, 12-bit samples, CMYK or 16-bit tables -- such files fail with a
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
  int      starved;          /* bytes of zero padding fed after the input ended WITHOUT a marker: a truncated stream */
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
    } else if (b->marker == 0) {
      b->starved++;                                /* the data ended in the middle of a scan */
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

/* ---- scans: coefficients of every block first (one scan of a baseline stream, up to dozens of a progressive one), pixels after ---- */

typedef struct {
  int      width, height, ncomp, hmax, vmax, restart, progressive;
  Comp     comp[3];
  int16_t *coef[3];            /* blocks_w * blocks_h * 64 per component, natural order */
  uint16_t qt[4][64];
  int      qt_present[4];
  Huff     dc[4], ac[4];
} Dec;

static int get_bit(Bits *b) { return get_bits(b, 1); }

/* one block of a scan (ITU T.81 F.2.2 for sequential streams, G.1.2 / G.2 for progressive ones) */
static bool scan_block(Dec *d, Bits *b, Comp *c, int16_t *blk, int ss, int se, int ah, int al, int *eobrun) {
  if (!d->progressive) {
    int t = huff_decode(b, &d->dc[c->td]);
    if (t < 0 || t > 11) return false;
    c->pred += t ? extend(get_bits(b, t), t) : 0;
    blk[0] = (int16_t)c->pred;
    for (int k = 1; k < 64;) {
      int rs = huff_decode(b, &d->ac[c->ta]);
      if (rs < 0) return false;
      int r = rs >> 4, sz = rs & 15;
      if (sz == 0) {
        if (r != 15) break;
        k += 16;
        continue;
      }
      k += r;
      if (k > 63) return false;
      blk[ZIGZAG[k]] = (int16_t)extend(get_bits(b, sz), sz);
      k++;
    }
    return true;
  }
  if (ss == 0) {                                   /* DC scan: first pass or one more bit */
    if (ah == 0) {
      int t = huff_decode(b, &d->dc[c->td]);
      if (t < 0 || t > 11) return false;
      c->pred += t ? extend(get_bits(b, t), t) : 0;
      blk[0] = (int16_t)(c->pred * (1 << al));
    } else if (get_bit(b)) {
      blk[0] |= (int16_t)(1 << al);
    }
    return true;
  }
  const int p1 = 1 << al, m1 = -(1 << al);
  if (ah == 0) {                                   /* AC scan, first pass of the band ss .. se */
    if (*eobrun > 0) { (*eobrun)--; return true; }
    for (int k = ss; k <= se; k++) {
      int rs = huff_decode(b, &d->ac[c->ta]);
      if (rs < 0) return false;
      int r = rs >> 4, sz = rs & 15;
      if (sz == 0) {
        if (r < 15) {
          *eobrun = (1 << r) - 1;
          if (r) *eobrun += get_bits(b, r);
          break;
        }
        k += 15;
      } else {
        k += r;
        if (k > se) return false;
        blk[ZIGZAG[k]] = (int16_t)(extend(get_bits(b, sz), sz) * (1 << al));
      }
    }
    return true;
  }
  int k = ss;                                      /* AC scan, one more bit for the band */
  if (*eobrun == 0) {
    for (; k <= se; k++) {
      int rs = huff_decode(b, &d->ac[c->ta]);
      if (rs < 0) return false;
      int r = rs >> 4, sz = rs & 15, val = 0;
      if (sz) {
        if (sz != 1) return false;
        val = get_bit(b) ? p1 : m1;
      } else if (r != 15) {
        *eobrun = 1 << r;
        if (r) *eobrun += get_bits(b, r);
        break;
      }
      do {                                         /* pass r coefficients that are still zero; the non-zero ones on the way get their bit */
        int16_t *q = &blk[ZIGZAG[k]];
        if (*q != 0) {
          if (get_bit(b) && (*q & p1) == 0) *q = (int16_t)(*q + (*q >= 0 ? p1 : m1));
        } else if (--r < 0) {
          break;
        }
        k++;
      } while (k <= se);
      if (val) {
        if (k > se) return false;
        blk[ZIGZAG[k]] = (int16_t)val;
      }
    }
  }
  if (*eobrun > 0) {
    for (; k <= se; k++) {
      int16_t *q = &blk[ZIGZAG[k]];
      if (*q != 0 && get_bit(b) && (*q & p1) == 0) *q = (int16_t)(*q + (*q >= 0 ? p1 : m1));
    }
    (*eobrun)--;
  }
  return true;
}

/* the entropy-coded segment after an SOS header; returns the offset of the marker that ends it (0: failure) */
static size_t decode_scan(Dec *d, const uint8_t *data, size_t n, size_t at, int n_sc, const int *sc, int ss, int se, int ah, int al, char *err, size_t err_len) {
  Bits b;
  memset(&b, 0, sizeof b);
  b.p = data + at;
  b.end = data + n;
  for (int k = 0; k < n_sc; k++) d->comp[sc[k]].pred = 0;
  int eobrun = 0, until_restart = d->restart, next_rst = 0;
  /* one component: its blocks in raster order, as many as cover the image; several: MCU by MCU */
  Comp *c0 = &d->comp[sc[0]];
  const int mcus_x = n_sc == 1 ? (c0->down_w + 7) / 8 : (d->width + 8 * d->hmax - 1) / (8 * d->hmax);
  const int mcus_y = n_sc == 1 ? (c0->down_h + 7) / 8 : (d->height + 8 * d->vmax - 1) / (8 * d->vmax);
  for (int my = 0; my < mcus_y; my++)
    for (int mx = 0; mx < mcus_x; mx++) {
      if (d->restart && until_restart == 0) {
        b.bitbuf = 0;
        b.bitcnt = 0;
        if (b.marker == 0) {
          while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++;
          if (b.p + 1 < b.end) b.p += 2;
        } else if (b.marker != 0xD0 + next_rst) { jfail(err, err_len, "restart marker out of sequence"); return 0; }
        b.marker = 0;
        next_rst = (next_rst + 1) & 7;
        until_restart = d->restart;
        eobrun = 0;
        for (int k = 0; k < n_sc; k++) d->comp[sc[k]].pred = 0;
      }
      for (int k = 0; k < n_sc; k++) {
        Comp *c = &d->comp[sc[k]];
        const int nh = n_sc == 1 ? 1 : c->h, nv = n_sc == 1 ? 1 : c->v;
        for (int by = 0; by < nv; by++)
          for (int bx = 0; bx < nh; bx++) {
            const size_t col = (size_t)mx * nh + bx, row = (size_t)my * nv + by;
            if (col >= (size_t)c->blocks_w || row >= (size_t)c->blocks_h) { jfail(err, err_len, "block outside the frame"); return 0; }
            if (!scan_block(d, &b, c, d->coef[sc[k]] + (row * c->blocks_w + col) * 64, ss, se, ah, al, &eobrun)) { jfail(err, err_len, "bad entropy-coded data"); return 0; }
          }
      }
      /* (the bit reader looks 4 bytes ahead: only padding that was actually CONSUMED means the scan ran past the end) */
      if (b.starved > 4) { jfail(err, err_len, "truncated entropy-coded data"); return 0; }
      if (d->restart) until_restart--;
    }
  /* the marker that ends the scan: already met by the bit reader, or ahead of it */
  if (b.marker) return (size_t)(b.p - data) - 2;
  const uint8_t *q = b.p;
  while (q + 1 < b.end && !(q[0] == 0xFF && q[1] != 0 && !(q[1] >= 0xD0 && q[1] <= 0xD7))) q++;
  return (size_t)(q - data);
}

bool rt_jpeg_decode(const unsigned char *data, size_t n, Image *out, char *err, size_t err_len) {
  Dec  *d = (Dec *)calloc(1, sizeof *d);
  bool  ok = false, frame = false, scanned = false;
  memset(out, 0, sizeof *out);
  if (!d) return jfail(err, err_len, "out of memory");
  d->hmax = d->vmax = 1;
  if (n < 4 || data[0] != 0xFF || data[1] != 0xD8) { jfail(err, err_len, "not a JPEG stream"); goto done; }
  for (size_t i = 2; i + 4 <= n;) {
    if (data[i] != 0xFF) { jfail(err, err_len, "marker expected"); goto done; }
    int m = data[i + 1];
    if (m == 0xFF) { i++; continue; }
    if (m == 0xD9) break;                                                     /* EOI */
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
    size_t len = ((size_t)data[i + 2] << 8) | data[i + 3];
    if (len < 2 || i + 2 + len > n) { jfail(err, err_len, "truncated segment"); goto done; }
    const uint8_t *s = data + i + 4;
    size_t sl = len - 2;
    if (m == 0xDB) {
      while (sl >= 65) {
        int pq = s[0] >> 4, tq = s[0] & 15;
        if (pq != 0 || tq > 3) { jfail(err, err_len, "16-bit quantisation tables are not supported"); goto done; }
        for (int k = 0; k < 64; k++) d->qt[tq][ZIGZAG[k]] = s[1 + k];
        d->qt_present[tq] = 1;
        s += 65;
        sl -= 65;
      }
    } else if (m == 0xC4) {
      while (sl >= 17) {
        int tc = s[0] >> 4, th = s[0] & 15;
        if (tc > 1 || th > 3) { jfail(err, err_len, "bad Huffman table id"); goto done; }
        Huff *h = tc ? &d->ac[th] : &d->dc[th];
        int total = 0;
        h->bits[0] = 0;
        for (int k = 1; k <= 16; k++) { h->bits[k] = s[k]; total += s[k]; }
        if (total > 256 || sl < (size_t)(17 + total)) { jfail(err, err_len, "bad Huffman table"); goto done; }
        memcpy(h->vals, s + 17, (size_t)total);
        huff_build(h);
        s += 17 + total;
        sl -= (size_t)(17 + total);
      }
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      if (frame) { jfail(err, err_len, "two frame headers"); goto done; }
      if (sl < 6 || s[0] != 8) { jfail(err, err_len, "only 8-bit JPEG is supported"); goto done; }
      d->progressive = m == 0xC2;
      d->height = (s[1] << 8) | s[2];
      d->width = (s[3] << 8) | s[4];
      d->ncomp = s[5];
      if ((d->ncomp != 1 && d->ncomp != 3) || d->width <= 0 || d->height <= 0 || sl < (size_t)(6 + 3 * d->ncomp)) { jfail(err, err_len, "unsupported frame header"); goto done; }
      if ((size_t)d->width * (size_t)d->height > ((size_t)1 << 28)) { jfail(err, err_len, "image larger than 2^28 pixels"); goto done; }
      /* a few hundred bytes must not command gigabytes of coefficient buffers: an MCU block costs at least ~1 bit per scan */
      if ((size_t)d->width * (size_t)d->height / 64 > n * 64 + 4096) { jfail(err, err_len, "frame header claims more pixels than the stream can hold"); goto done; }
      for (int k = 0; k < d->ncomp; k++) {
        Comp *c = &d->comp[k];
        c->id = s[6 + 3 * k];
        c->h = s[7 + 3 * k] >> 4;
        c->v = s[7 + 3 * k] & 15;
        c->tq = s[8 + 3 * k];
        if (c->h < 1 || c->h > 4 || c->v < 1 || c->v > 4 || c->tq > 3) { jfail(err, err_len, "bad sampling factors"); goto done; }
        if (c->h > d->hmax) d->hmax = c->h;
        if (c->v > d->vmax) d->vmax = c->v;
      }
      if (d->ncomp == 1) d->comp[0].h = d->comp[0].v = d->hmax = d->vmax = 1;   /* one component: its sampling factors mean nothing */
      const int mcus_x = (d->width + 8 * d->hmax - 1) / (8 * d->hmax), mcus_y = (d->height + 8 * d->vmax - 1) / (8 * d->vmax);
      for (int k = 0; k < d->ncomp; k++) {
        Comp *c = &d->comp[k];
        if (d->hmax % c->h || d->vmax % c->v) { jfail(err, err_len, "fractional sampling ratios are not supported"); goto done; }
        c->blocks_w = mcus_x * c->h;
        c->blocks_h = mcus_y * c->v;
        c->down_w = (d->width * c->h + d->hmax - 1) / d->hmax;
        c->down_h = (d->height * c->v + d->vmax - 1) / d->vmax;
        d->coef[k] = (int16_t *)calloc((size_t)c->blocks_w * c->blocks_h * 64, sizeof(int16_t));
        if (!d->coef[k]) { jfail(err, err_len, "out of memory"); goto done; }
      }
      frame = true;
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      jfail(err, err_len, "lossless / hierarchical / arithmetic-coded JPEG is not supported (use the RT8I side files)");
      goto done;
    } else if (m == 0xDD) {
      if (sl >= 2) d->restart = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {
      if (!frame || sl < 1 || s[0] < 1 || s[0] > d->ncomp || sl < (size_t)(1 + 2 * s[0] + 3)) { jfail(err, err_len, "bad scan header"); goto done; }
      int n_sc = s[0], sc[3];
      for (int k = 0; k < n_sc; k++) {
        int cid = s[1 + 2 * k], found = -1;
        for (int c = 0; c < d->ncomp; c++) if (d->comp[c].id == cid) found = c;
        if (found < 0) { jfail(err, err_len, "scan names an unknown component"); goto done; }
        for (int j = 0; j < k; j++) if (sc[j] == found) { jfail(err, err_len, "scan names a component twice"); goto done; }
        sc[k] = found;
        d->comp[found].td = s[2 + 2 * k] >> 4;
        d->comp[found].ta = s[2 + 2 * k] & 15;
        if (d->comp[found].td > 3 || d->comp[found].ta > 3) { jfail(err, err_len, "bad Huffman table id in the scan header"); goto done; }
      }
      const int ss = s[1 + 2 * n_sc], se = s[2 + 2 * n_sc], ah = s[3 + 2 * n_sc] >> 4, al = s[3 + 2 * n_sc] & 15;
      if (d->progressive ? (ss > se || se > 63 || al > 13 || ah > 13 || (ss == 0 && se != 0) || (ss > 0 && n_sc != 1)) : (ss != 0 || se != 63 || ah || al)) {
        jfail(err, err_len, "bad spectral selection / successive approximation");
        goto done;
      }
      for (int k = 0; k < n_sc; k++) {
        Comp *c = &d->comp[sc[k]];
        const bool need_dc = !d->progressive || (ss == 0 && ah == 0), need_ac = !d->progressive || ss > 0;
        if ((need_dc && !d->dc[c->td].present) || (need_ac && !d->ac[c->ta].present)) { jfail(err, err_len, "a Huffman table the scan needs is missing"); goto done; }
      }
      size_t next = decode_scan(d, data, n, i + 2 + len, n_sc, sc, ss, se, ah, al, err, err_len);
      if (!next) goto done;
      scanned = true;
      i = next;
      continue;
    }
    i += 2 + len;
  }
  if (!frame || !scanned) { jfail(err, err_len, "no scan"); goto done; }
  for (int k = 0; k < d->ncomp; k++) {
    Comp *c = &d->comp[k];
    if (!d->qt_present[c->tq]) { jfail(err, err_len, "a quantisation table the frame needs is missing"); goto done; }
    c->plane = (uint8_t *)malloc((size_t)c->blocks_w * 8 * (size_t)c->blocks_h * 8);
    if (!c->plane) { jfail(err, err_len, "out of memory"); goto done; }
    for (int by = 0; by < c->blocks_h; by++)
      for (int bx = 0; bx < c->blocks_w; bx++)
        idct_islow(d->coef[k] + ((size_t)by * c->blocks_w + bx) * 64, d->qt[c->tq], c->plane + (size_t)by * 8 * c->blocks_w * 8 + (size_t)bx * 8, c->blocks_w * 8);
  }
  {
    const int width = d->width, height = d->height, ncomp = d->ncomp, hmax = d->hmax, vmax = d->vmax;
    Comp *comp = d->comp;
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
  for (int k = 0; k < 3; k++) { free(d->comp[k].plane); free(d->coef[k]); }
  free(d);
  return ok;
}
