/* rt_png.c -- a PNG decoder for the C host (examples/rt_model.c): the other codec glTF allows for textures, and what the
 * map_* keys of a .mtl usually name.  The reference decodes through codin's stb_image_load_bytes (driver.c:106-116, 621),
 * which is not in the reference tree; the benchmark's Python loader decodes with PIL and `.convert("RGB")`.  PNG is
 * lossless, so "the same texels" only needs the same mapping to RGB8, restated here from PIL's:
 *
 *   gray 1 / 2 / 4 / 8 bits   -> scaled to 0 .. 255 (x 255, x 85, x 17, x 1), replicated into R, G, B
 *   gray + alpha, RGB, RGBA   -> alpha dropped; 16-bit samples: the high byte
 *   palette 1 / 2 / 4 / 8     -> PLTE entry (black beyond the end of the table); tRNS ignored
 *   16-bit gray               -> refused (PIL clips it instead of scaling: not something to imitate)
 *   Adam7 interlace           -> the seven passes scattered into place (pinned by hand-written streams that PIL reads)
 *
 * zlib: stored, fixed and dynamic blocks (RFC 1950 / 1951), no preset dictionary; the Adler-32 and the chunk CRCs are not
 * checked (a damaged stream fails in the Huffman decoder or on its length instead).  Filters 0-4 of the PNG specification.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_model.h"

static bool pfail(char *err, size_t n, const char *msg) {
  if (err && n) snprintf(err, n, "rt_png: %s", msg);
  return false;
}

/* ---- inflate ------------------------------------------------------------------------------------------------------- */

typedef struct {
  const uint8_t *in;
  size_t   in_len, in_pos;
  uint32_t bitbuf;
  int      bitcnt;
  uint8_t *out;
  size_t   out_len, out_pos;
  bool     bad;
} Inflate;

typedef struct { uint16_t count[16], symbol[288]; } HuffTable;

static int in_bits(Inflate *s, int need) {
  while (s->bitcnt < need) {
    if (s->in_pos >= s->in_len) { s->bad = true; return 0; }
    s->bitbuf |= (uint32_t)s->in[s->in_pos++] << s->bitcnt;
    s->bitcnt += 8;
  }
  int v = (int)(s->bitbuf & ((1u << need) - 1u));
  s->bitbuf >>= need;
  s->bitcnt -= need;
  return v;
}

static bool huff_make(HuffTable *h, const uint8_t *lengths, int n) {
  int offs[16];
  memset(h->count, 0, sizeof h->count);
  for (int i = 0; i < n; i++) h->count[lengths[i]]++;
  int left = 1;
  for (int len = 1; len < 16; len++) {
    left <<= 1;
    left -= h->count[len];
    if (left < 0) return false;                    /* over-subscribed */
  }
  offs[1] = 0;
  for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + h->count[len];
  for (int i = 0; i < n; i++)
    if (lengths[i]) h->symbol[offs[lengths[i]]++] = (uint16_t)i;
  return true;
}

static int huff_sym(Inflate *s, const HuffTable *h) {
  int code = 0, first = 0, index = 0;
  for (int len = 1; len < 16; len++) {
    code |= in_bits(s, 1);
    if (s->bad) return -1;
    int count = h->count[len];
    if (code - count < first) return h->symbol[index + (code - first)];
    index += count;
    first += count;
    first <<= 1;
    code <<= 1;
  }
  return -1;
}

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t  LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                       4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t  DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static bool inflate_codes(Inflate *s, const HuffTable *lit, const HuffTable *dist) {
  for (;;) {
    int sym = huff_sym(s, lit);
    if (sym < 0) return false;
    if (sym < 256) {
      if (s->out_pos >= s->out_len) return false;
      s->out[s->out_pos++] = (uint8_t)sym;
    } else if (sym == 256) {
      return true;
    } else {
      sym -= 257;
      if (sym >= 29) return false;
      int len = LEN_BASE[sym] + in_bits(s, LEN_EXTRA[sym]);
      int ds = huff_sym(s, dist);
      if (ds < 0 || ds >= 30) return false;
      size_t d = (size_t)DIST_BASE[ds] + (size_t)in_bits(s, DIST_EXTRA[ds]);
      if (s->bad || d > s->out_pos || s->out_pos + (size_t)len > s->out_len) return false;
      for (int k = 0; k < len; k++, s->out_pos++) s->out[s->out_pos] = s->out[s->out_pos - d];
    }
  }
}

static bool inflate_zlib(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
  if (in_len < 2 || (in[0] & 15) != 8 || ((in[0] << 8) | in[1]) % 31 != 0 || (in[1] & 0x20)) return false;
  Inflate s;
  memset(&s, 0, sizeof s);
  s.in = in + 2;
  s.in_len = in_len - 2;
  s.out = out;
  s.out_len = out_len;
  static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  HuffTable lit, dist;
  int last;
  do {
    last = in_bits(&s, 1);
    int type = in_bits(&s, 2);
    if (s.bad) return false;
    if (type == 0) {
      s.bitbuf = 0;
      s.bitcnt = 0;
      if (s.in_pos + 4 > s.in_len) return false;
      size_t len = s.in[s.in_pos] | ((size_t)s.in[s.in_pos + 1] << 8), nlen = s.in[s.in_pos + 2] | ((size_t)s.in[s.in_pos + 3] << 8);
      s.in_pos += 4;
      if ((len ^ 0xFFFFu) != nlen || s.in_pos + len > s.in_len || s.out_pos + len > s.out_len) return false;
      memcpy(s.out + s.out_pos, s.in + s.in_pos, len);
      s.in_pos += len;
      s.out_pos += len;
    } else if (type == 1) {
      uint8_t l[288];
      for (int i = 0; i < 144; i++) l[i] = 8;
      for (int i = 144; i < 256; i++) l[i] = 9;
      for (int i = 256; i < 280; i++) l[i] = 7;
      for (int i = 280; i < 288; i++) l[i] = 8;
      huff_make(&lit, l, 288);
      for (int i = 0; i < 30; i++) l[i] = 5;
      huff_make(&dist, l, 30);
      if (!inflate_codes(&s, &lit, &dist)) return false;
    } else if (type == 2) {
      int nlen = in_bits(&s, 5) + 257, ndist = in_bits(&s, 5) + 1, ncode = in_bits(&s, 4) + 4;
      if (s.bad || nlen > 286 || ndist > 30) return false;
      uint8_t l[320];
      memset(l, 0, sizeof l);
      for (int i = 0; i < ncode; i++) l[ORDER[i]] = (uint8_t)in_bits(&s, 3);
      HuffTable cl;
      if (s.bad || !huff_make(&cl, l, 19)) return false;
      memset(l, 0, sizeof l);
      for (int i = 0; i < nlen + ndist;) {
        int sym = huff_sym(&s, &cl);
        if (sym < 0) return false;
        if (sym < 16) { l[i++] = (uint8_t)sym; continue; }
        int prev = 0, rep;
        if (sym == 16) { if (i == 0) return false; prev = l[i - 1]; rep = 3 + in_bits(&s, 2); }
        else if (sym == 17) rep = 3 + in_bits(&s, 3);
        else rep = 11 + in_bits(&s, 7);
        if (s.bad || i + rep > nlen + ndist) return false;
        while (rep--) l[i++] = (uint8_t)prev;
      }
      if (l[256] == 0 || !huff_make(&lit, l, nlen) || !huff_make(&dist, l + nlen, ndist)) return false;
      if (!inflate_codes(&s, &lit, &dist)) return false;
    } else {
      return false;
    }
  } while (!last);
  return s.out_pos == s.out_len;
}

/* ---- PNG ----------------------------------------------------------------------------------------------------------- */

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c) {
  int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool rt_png_decode(const unsigned char *data, size_t n, Image *out, char *err, size_t err_len) {
  static const uint8_t SIG[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
  memset(out, 0, sizeof *out);
  if (n < 8 || memcmp(data, SIG, 8) != 0) return pfail(err, err_len, "not a PNG stream");
  uint32_t w = 0, h = 0;
  int      depth = 0, ctype = -1, n_pal = 0, interlace = 0;
  uint8_t  pal[256][3];
  memset(pal, 0, sizeof pal);
  uint8_t *z = (uint8_t *)malloc(n), *raw = NULL, *rgb = NULL;
  size_t   zn = 0;
  bool     ok = false, ended = false;
  if (!z) return pfail(err, err_len, "out of memory");
  for (size_t i = 8; i + 12 <= n && !ended;) {
    uint32_t len = be32(data + i);
    const uint8_t *type = data + i + 4, *body = data + i + 8;
    if ((size_t)len > n - i - 12) { pfail(err, err_len, "truncated chunk"); goto done; }
    if (!memcmp(type, "IHDR", 4)) {
      if (len < 13) { pfail(err, err_len, "bad IHDR"); goto done; }
      w = be32(body);
      h = be32(body + 4);
      depth = body[8];
      ctype = body[9];
      if (body[10] != 0 || body[11] != 0) { pfail(err, err_len, "unknown compression / filter method"); goto done; }
      if (body[12] > 1) { pfail(err, err_len, "unknown interlace method"); goto done; }
      interlace = body[12];
    } else if (!memcmp(type, "PLTE", 4)) {
      n_pal = (int)(len / 3 > 256 ? 256 : len / 3);
      memcpy(pal, body, (size_t)n_pal * 3);
    } else if (!memcmp(type, "IDAT", 4)) {
      memcpy(z + zn, body, len);
      zn += len;
    } else if (!memcmp(type, "IEND", 4)) {
      ended = true;
    }
    i += 12 + (size_t)len;
  }
  int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8)
                : (depth == 8 || depth == 16);
  if (ctype == 0 && depth == 16) { pfail(err, err_len, "16-bit grayscale PNG is not supported (use the RT8I side files)"); goto done; }
  if (!channels || !depth_ok || w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24)) { pfail(err, err_len, "unsupported header"); goto done; }
  if ((size_t)w * (size_t)h > ((size_t)1 << 28)) { pfail(err, err_len, "image larger than 2^28 pixels"); goto done; }
  {
    /* one pass (the whole image) or the seven of Adam7: pass p holds the pixels (x0 + i dx, y0 + j dy), as an image of its own */
    static const uint8_t X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1}, DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
    const size_t bits = (size_t)channels * (size_t)depth, bpp = bits >= 8 ? bits / 8 : 1;
    const int n_pass = interlace ? 7 : 1;
    size_t total = 0;
    for (int p = 0; p < n_pass; p++) {
      const size_t pw = interlace ? ((size_t)w + DX[p] - 1 - X0[p]) / DX[p] : w, ph = interlace ? ((size_t)h + DY[p] - 1 - Y0[p]) / DY[p] : h;
      if (pw && ph) total += ((pw * bits + 7) / 8 + 1) * ph;
    }
    /* deflate expands at most ~1032 : 1 -- a header whose pixels the IDAT bytes cannot possibly fill is refused before it
     * commands the buffers */
    if (total / 1032 > zn + 64) { pfail(err, err_len, "header claims more pixels than the IDAT data can hold"); goto done; }
    raw = (uint8_t *)malloc(total ? total : 1);
    rgb = (uint8_t *)malloc((size_t)w * h * 3);
    if (!raw || !rgb) { pfail(err, err_len, "out of memory"); goto done; }
    if (!inflate_zlib(z, zn, raw, total)) { pfail(err, err_len, "bad zlib stream (or not the size the header announces)"); goto done; }
    uint8_t *at = raw;
    for (int p = 0; p < n_pass; p++) {
      const size_t x0 = interlace ? X0[p] : 0, y0 = interlace ? Y0[p] : 0, dx = interlace ? DX[p] : 1, dy = interlace ? DY[p] : 1;
      const size_t pw = ((size_t)w + dx - 1 - x0) / dx, ph = ((size_t)h + dy - 1 - y0) / dy;
      if (!pw || !ph) continue;
      const size_t row = (pw * bits + 7) / 8;
      for (size_t y = 0; y < ph; y++) {
        uint8_t *cur = at + y * (row + 1) + 1;
        const uint8_t *up = y ? cur - (row + 1) : NULL;
        const int f = cur[-1];
        if (f > 4) { pfail(err, err_len, "unknown filter type"); goto done; }
        for (size_t x = 0; x < row; x++) {
          const int a = x >= bpp ? cur[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
          const int add = f == 0 ? 0 : f == 1 ? a : f == 2 ? b : f == 3 ? ((a + b) >> 1) : paeth(a, b, c);
          cur[x] = (uint8_t)(cur[x] + add);
        }
        const int step = depth == 16 ? 2 : 1;                              /* 16-bit samples: the high byte comes first */
        for (size_t x = 0; x < pw; x++) {
          uint8_t *o = rgb + ((y0 + y * dy) * (size_t)w + (x0 + x * dx)) * 3;
          if (ctype == 2 || ctype == 6) {
            const uint8_t *q = cur + x * (size_t)channels * step;
            o[0] = q[0]; o[1] = q[step]; o[2] = q[2 * step];
          } else if (ctype == 4) {
            o[0] = o[1] = o[2] = cur[x * 2 * step];
          } else {
            const int per = 8 / depth, v = (cur[x / per] >> ((per - 1 - (int)(x % per)) * depth)) & ((1 << depth) - 1);
            if (ctype == 3) { o[0] = pal[v][0]; o[1] = pal[v][1]; o[2] = pal[v][2]; }
            else o[0] = o[1] = o[2] = (uint8_t)(v * 255 / ((1 << depth) - 1));
          }
        }
      }
      at += (row + 1) * ph;
    }
  }
  (void)n_pal;
  out->components = 3;
  out->pixel_type = PT_u8;
  out->width = (int)w;
  out->stride = (int)w;
  out->height = (int)h;
  out->pixels.data = rgb;
  out->pixels.len = (isize)w * h * 3;
  rgb = NULL;
  ok = true;
done:
  free(z);
  free(raw);
  free(rgb);
  return ok;
}
