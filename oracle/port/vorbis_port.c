/* oracle/port/vorbis_port.c -- TEST INFRASTRUCTURE ONLY.  Never linked, imported or
 * executed by the product (vorbis_amd/); only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may use it, and only as the checker.
 *
 * A from-scratch, single-threaded, plain-C restatement of libvorbis' per-block
 * encode analysis -- the VBR path of mapping0_forward (reference
 * lib/mapping0.c:230-696) down to _vp_couple_quantize_normalize -- that consumes
 * the same setup blob as the GPU library (include/vamd_setup.h).  It keeps the
 * reference's *sequential* formulation (running pointers, Bresenham stepping,
 * monotone stack walk, per-partition sort), deliberately unlike the wave-parallel
 * restructuring in vorbis_amd/csrc, so the two can check each other.
 *
 * Parity status: PINNED.  tests/test_oracle.py compares every function here
 * bit-for-bit with (a) oracle/_ref/libvorbis_ref.so = the unmodified reference
 * sources compiled in place, on seeded random and stream-cut blocks at q=-0.1..1.0,
 * and (b) the golden fixtures tests/golden/*.npz that tools/make_golden.py
 * generated from that same reference build.  The reference's own test-suite holds
 * no numeric vectors for this path.
 *
 * Build: strict IEEE -- gcc -O2 -fno-fast-math -ffp-contract=off (oracle/Makefile).
 * Each function cites the reference lines it follows.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "vamd_setup.h"

#define NEG_INF (-9999.f)

typedef struct port_enc {
  unsigned char *blob;
  vamd_setup_header h;
} port_enc;

typedef struct port_taps {
  float *windowed, *mdct_raw, *fft_packed, *logfft, *logmdct, *noise, *tone, *logmask, *mdct;
  int *posts, *post_valid, *ilogmask, *iwork, *nonzero;
  float *local_ampmax, *ampmax_out;
  /* residue back-end: classes per partition and the codebook entries in emission order */
  int *res_class;
  long res_class_cap, res_partvals;
  unsigned short *res_entries;
  long res_entries_cap, res_count;
} port_taps;

/* per candidate packet of a bitrate-managed block (lib/mapping0.c:507-573,596-646) */
typedef struct port_mtaps {
  int *posts;      /* [15][ch][VAMD_POSIT] */
  int *post_valid; /* [15][ch] */
  int *ilogmask;   /* [15][ch][n/2] */
  int *iwork;      /* [15][ch][n/2] */
  int *nonzero;    /* [15][ch] */
} port_mtaps;

static const float *tabf(const port_enc *e, uint32_t off) { return (const float *)(e->blob + off); }
static const int32_t *tabi(const port_enc *e, uint32_t off) { return (const int32_t *)(e->blob + off); }

port_enc *port_open(const void *blob, size_t bytes) {
  port_enc *e;
  vamd_setup_header h;
  if (!blob || bytes < sizeof(h)) return NULL;
  memcpy(&h, blob, sizeof(h));
  if (h.magic != VAMD_SETUP_MAGIC || h.version != VAMD_SETUP_VERSION || h.total_bytes > bytes) return NULL;
  e = (port_enc *)calloc(1, sizeof(*e));
  e->blob = (unsigned char *)malloc(h.total_bytes);
  memcpy(e->blob, blob, h.total_bytes);
  e->h = h;
  return e;
}

void port_close(port_enc *e) {
  if (!e) return;
  free(e->blob);
  free(e);
}

int port_channels(const port_enc *e) { return e->h.channels; }
int port_blocksize(const port_enc *e, int W) { return e->h.blocksizes[W]; }
int port_floor_posts(const port_enc *e, int W) { return e->h.mode[W].floor[0].posts; }

/* ---- scalar helpers --------------------------------------------------------- */

/* todB, lib/scales.h:43-51: integer-bit-trick log of |x| */
static float to_dB(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u &= 0x7fffffffu;
  return (float)(u * 7.17711438e-7f - 764.6161886f);
}

/* unitnorm, lib/scales.h:32-40 */
static float unit_norm(float x) {
  uint32_t u;
  float r;
  memcpy(&u, &x, 4);
  u = (u & 0x80000000u) | 0x3f800000u;
  memcpy(&r, &u, 4);
  return r;
}

/* vorbis_dBquant, lib/floor1.c:273-278 */
static int dB_quant(float x) {
  int i = (int)(x * 7.3142857f + 1023.5f);
  if (i > 1023) return 1023;
  if (i < 0) return 0;
  return i;
}

/* floor1_inverse_dB_table (Vorbis I spec 10.1; lib/floor1.c:280-345): bit patterns
 * shared with the product's generated header */
#include "../../vorbis_amd/csrc/floor1_db_table.h"
static const uint32_t inverse_dB_bits[256] = {VAMD_FLOOR1_DB_TABLE_BITS};
static float inverse_dB(int i) {
  float f;
  memcpy(&f, &inverse_dB_bits[i & 255], 4);
  return f;
}

/* ---- window, lib/window.c:2102-2135 ----------------------------------------- */
void port_apply_window(const port_enc *e, float *d, int lW, int W, int nW) {
  long n, ln, rn, lb, le, rb, re, i, p;
  const float *wl, *wr;
  lW = W ? lW : 0;
  nW = W ? nW : 0;
  n = e->h.blocksizes[W];
  ln = e->h.blocksizes[lW];
  rn = e->h.blocksizes[nW];
  wl = tabf(e, e->h.xform[lW].off_window);
  wr = tabf(e, e->h.xform[nW].off_window);
  lb = n / 4 - ln / 4;
  le = lb + ln / 2;
  rb = n / 2 + n / 4 - rn / 4;
  re = rb + rn / 2;
  for (i = 0; i < lb; i++) d[i] = 0.f;
  for (p = 0; i < le; i++, p++) d[i] *= wl[p];
  for (i = rb, p = rn / 2 - 1; i < re; i++, p--) d[i] *= wr[p];
  for (; i < n; i++) d[i] = 0.f;
}

/* ---- forward MDCT, lib/mdct.c:93-394,492-562 ---------------------------------- */
#define PI3_8 .38268343236508977175F
#define PI2_8 .70710678118654752441F
#define PI1_8 .92387953251128675613F

static void bf8(float *x) { /* lib/mdct.c:93-114 */
  float a = x[6] + x[2], b = x[6] - x[2], c = x[4] + x[0], d = x[4] - x[0], t, u;
  x[6] = a + c;
  x[4] = a - c;
  t = x[5] - x[1];
  u = x[7] - x[3];
  x[0] = b + t;
  x[2] = b - t;
  t = x[5] + x[1];
  b = x[7] + x[3];
  x[3] = u + d;
  x[1] = u - d;
  x[7] = b + t;
  x[5] = b - t;
}

static void bf16(float *x) { /* lib/mdct.c:117-149 */
  float s = x[1] - x[9], t = x[0] - x[8];
  x[8] += x[0];
  x[9] += x[1];
  x[0] = (s + t) * PI2_8;
  x[1] = (s - t) * PI2_8;
  s = x[3] - x[11];
  t = x[10] - x[2];
  x[10] += x[2];
  x[11] += x[3];
  x[2] = s;
  x[3] = t;
  s = x[12] - x[4];
  t = x[13] - x[5];
  x[12] += x[4];
  x[13] += x[5];
  x[4] = (s - t) * PI2_8;
  x[5] = (s + t) * PI2_8;
  s = x[14] - x[6];
  t = x[15] - x[7];
  x[14] += x[6];
  x[15] += x[7];
  x[6] = s;
  x[7] = t;
  bf8(x);
  bf8(x + 8);
}

static void bf32(float *x) { /* lib/mdct.c:152-213 */
  float s = x[30] - x[14], t = x[31] - x[15];
  x[30] += x[14];
  x[31] += x[15];
  x[14] = s;
  x[15] = t;
  s = x[28] - x[12];
  t = x[29] - x[13];
  x[28] += x[12];
  x[29] += x[13];
  x[12] = s * PI1_8 - t * PI3_8;
  x[13] = s * PI3_8 + t * PI1_8;
  s = x[26] - x[10];
  t = x[27] - x[11];
  x[26] += x[10];
  x[27] += x[11];
  x[10] = (s - t) * PI2_8;
  x[11] = (s + t) * PI2_8;
  s = x[24] - x[8];
  t = x[25] - x[9];
  x[24] += x[8];
  x[25] += x[9];
  x[8] = s * PI3_8 - t * PI1_8;
  x[9] = t * PI3_8 + s * PI1_8;
  s = x[22] - x[6];
  t = x[7] - x[23];
  x[22] += x[6];
  x[23] += x[7];
  x[6] = t;
  x[7] = s;
  s = x[4] - x[20];
  t = x[5] - x[21];
  x[20] += x[4];
  x[21] += x[5];
  x[4] = t * PI1_8 + s * PI3_8;
  x[5] = t * PI3_8 - s * PI1_8;
  s = x[2] - x[18];
  t = x[3] - x[19];
  x[18] += x[2];
  x[19] += x[3];
  x[2] = (t + s) * PI2_8;
  x[3] = (t - s) * PI2_8;
  s = x[0] - x[16];
  t = x[1] - x[17];
  x[16] += x[0];
  x[17] += x[1];
  x[0] = t * PI3_8 + s * PI1_8;
  x[1] = t * PI1_8 - s * PI3_8;
  bf16(x);
  bf16(x + 16);
}

/* mdct_butterfly_first / _generic, lib/mdct.c:216-314: one pass over `points`
 * values walking down from the top, trig stride `step` */
static void bf_pass(const float *T, float *x, int points, int step) {
  float *hi = x + points, *lo = x + (points >> 1);
  while (lo > x) {
    float s, t;
    hi -= 2;
    lo -= 2;
    s = hi[0] - lo[0];
    t = hi[1] - lo[1];
    hi[0] += lo[0];
    hi[1] += lo[1];
    lo[0] = t * T[1] + s * T[0];
    lo[1] = t * T[0] - s * T[1];
    T += step;
  }
}

static void mdct_fwd(int n, int log2n, float scale, const float *trig, const int32_t *rev, const float *in,
                     float *out);
void port_mdct_forward(const port_enc *e, int W, const float *in, float *out) {
  const vamd_xform_tab *x = &e->h.xform[W];
  mdct_fwd(x->n, x->log2n, x->mdct_scale, tabf(e, x->off_mdct_trig), tabi(e, x->off_mdct_bitrev), in, out);
}

/* mdct_forward, lib/mdct.c:492-562, for any lookup (the block transforms and the detector's) */
static void mdct_fwd(int n, int log2n, float scale, const float *trig, const int32_t *rev, const float *in,
                     float *out) {
  const int n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
  float *w = (float *)malloc(sizeof(float) * n), *w2 = w + n2;
  const float *T = trig + n2;
  const float *a = in + n2 + n4, *b = a + 1;
  int i, s, stages;

  /* fold + pre-twiddle, lib/mdct.c:506-544 */
  for (i = 0; i < n8; i += 2) {
    float r0, r1;
    a -= 4;
    T -= 2;
    r0 = a[2] + b[0];
    r1 = a[0] + b[2];
    w2[i] = r1 * T[1] + r0 * T[0];
    w2[i + 1] = r1 * T[0] - r0 * T[1];
    b += 4;
  }
  b = in + 1;
  for (; i < n2 - n8; i += 2) {
    float r0, r1;
    T -= 2;
    a -= 4;
    r0 = a[2] - b[0];
    r1 = a[0] - b[2];
    w2[i] = r1 * T[1] + r0 * T[0];
    w2[i + 1] = r1 * T[0] - r0 * T[1];
    b += 4;
  }
  a = in + n;
  for (; i < n2; i += 2) {
    float r0, r1;
    T -= 2;
    a -= 4;
    r0 = -a[2] - b[0];
    r1 = -a[0] - b[2];
    w2[i] = r1 * T[1] + r0 * T[0];
    w2[i + 1] = r1 * T[0] - r0 * T[1];
    b += 4;
  }

  /* mdct_butterflies, lib/mdct.c:316-336 */
  stages = log2n - 5;
  if (--stages > 0) bf_pass(trig, w2, n2, 4);
  for (s = 1; --stages > 0; s++) {
    int j;
    for (j = 0; j < (1 << s); j++) bf_pass(trig, w2 + (n2 >> s) * j, n2 >> s, 4 << s);
  }
  for (i = 0; i < n2; i += 32) bf32(w2 + i);

  /* mdct_bitreverse, lib/mdct.c:346-394 */
  {
    float *w0 = w, *w1 = w2;
    const int32_t *bit = rev;
    T = trig + n;
    do {
      const float *p = w2 + bit[0], *q = w2 + bit[1];
      float r0 = p[1] - q[1], r1 = p[0] + q[0];
      float r2 = r1 * T[0] + r0 * T[1], r3 = r1 * T[1] - r0 * T[0];
      w1 -= 4;
      r0 = (p[1] + q[1]) * .5f;
      r1 = (p[0] - q[0]) * .5f;
      w0[0] = r0 + r2;
      w1[2] = r0 - r2;
      w0[1] = r1 + r3;
      w1[3] = r3 - r1;
      p = w2 + bit[2];
      q = w2 + bit[3];
      r0 = p[1] - q[1];
      r1 = p[0] + q[0];
      r2 = r1 * T[2] + r0 * T[3];
      r3 = r1 * T[3] - r0 * T[2];
      r0 = (p[1] + q[1]) * .5f;
      r1 = (p[0] - q[0]) * .5f;
      w0[2] = r0 + r2;
      w1[0] = r0 - r2;
      w0[3] = r1 + r3;
      w1[1] = r3 - r1;
      T += 4;
      bit += 4;
      w0 += 4;
    } while (w0 < w1);
  }

  /* final rotation and scale, lib/mdct.c:552-561 */
  T = trig + n2;
  for (i = 0; i < n4; i++) {
    out[i] = (w[2 * i] * T[0] + w[2 * i + 1] * T[1]) * scale;
    out[n2 - 1 - i] = (w[2 * i] * T[1] - w[2 * i + 1] * T[0]) * scale;
    T += 2;
  }
  free(w);
}

/* ---- real FFT, lib/smallft.c:113-268,572-631 ---------------------------------- */
static void rf2(int ido, int l1, const float *cc, float *ch, const float *wa1) { /* dradf2 */
  const int t0 = l1 * ido;
  int k, i;
  for (k = 0; k < l1; k++) {
    ch[2 * k * ido] = cc[k * ido] + cc[t0 + k * ido];
    ch[2 * k * ido + 2 * ido - 1] = cc[k * ido] - cc[t0 + k * ido];
  }
  if (ido < 2) return;
  if (ido != 2) {
    for (k = 0; k < l1; k++)
      for (i = 2; i < ido; i += 2) {
        const int c1 = t0 + k * ido + i, c0 = k * ido + i;
        const int up = 2 * k * ido + i, dn = 2 * k * ido + 2 * ido - i;
        const float tr2 = wa1[i - 2] * cc[c1 - 1] + wa1[i - 1] * cc[c1];
        const float ti2 = wa1[i - 2] * cc[c1] - wa1[i - 1] * cc[c1 - 1];
        ch[up] = cc[c0] + ti2;
        ch[dn] = ti2 - cc[c0];
        ch[up - 1] = cc[c0 - 1] + tr2;
        ch[dn - 1] = cc[c0 - 1] - tr2;
      }
    if (ido % 2 == 1) return;
  }
  for (k = 0; k < l1; k++) {
    ch[ido + 2 * k * ido] = -cc[ido - 1 + t0 + k * ido];
    ch[ido + 2 * k * ido - 1] = cc[ido - 1 + k * ido];
  }
}

static void rf4(int ido, int l1, const float *cc, float *ch, const float *wa1, const float *wa2,
                const float *wa3) { /* dradf4 */
  static const float hsqt2 = .70710678118654752f;
  const int t0 = l1 * ido;
  int k, i;
  for (k = 0; k < l1; k++) {
    const int q0 = k * ido, q1 = t0 + k * ido, q2 = 2 * t0 + k * ido, q3 = 3 * t0 + k * ido;
    const float tr1 = cc[q1] + cc[q3], tr2 = cc[q0] + cc[q2];
    const int o = 4 * q0;
    ch[o] = tr1 + tr2;
    ch[o + 4 * ido - 1] = tr2 - tr1;
    ch[o + 2 * ido - 1] = cc[q0] - cc[q2];
    ch[o + 2 * ido] = cc[q3] - cc[q1];
  }
  if (ido < 2) return;
  if (ido != 2) {
    for (k = 0; k < l1; k++)
      for (i = 2; i < ido; i += 2) {
        const int q0 = k * ido + i;
        const int o4 = 4 * k * ido + i, o5 = 4 * k * ido + 2 * ido - i, o6 = 2 * ido;
        int q = q0 + t0;
        float cr2, ci2, cr3, ci3, cr4, ci4, tr1, tr2, tr3, tr4, ti1, ti2, ti3, ti4;
        cr2 = wa1[i - 2] * cc[q - 1] + wa1[i - 1] * cc[q];
        ci2 = wa1[i - 2] * cc[q] - wa1[i - 1] * cc[q - 1];
        q += t0;
        cr3 = wa2[i - 2] * cc[q - 1] + wa2[i - 1] * cc[q];
        ci3 = wa2[i - 2] * cc[q] - wa2[i - 1] * cc[q - 1];
        q += t0;
        cr4 = wa3[i - 2] * cc[q - 1] + wa3[i - 1] * cc[q];
        ci4 = wa3[i - 2] * cc[q] - wa3[i - 1] * cc[q - 1];
        tr1 = cr2 + cr4;
        tr4 = cr4 - cr2;
        ti1 = ci2 + ci4;
        ti4 = ci2 - ci4;
        ti2 = cc[q0] + ci3;
        ti3 = cc[q0] - ci3;
        tr2 = cc[q0 - 1] + cr3;
        tr3 = cc[q0 - 1] - cr3;
        ch[o4 - 1] = tr1 + tr2;
        ch[o4] = ti1 + ti2;
        ch[o5 - 1] = tr3 - ti4;
        ch[o5] = tr4 - ti3;
        ch[o4 + o6 - 1] = ti4 + tr3;
        ch[o4 + o6] = tr4 + ti3;
        ch[o5 + o6 - 1] = tr2 - tr1;
        ch[o5 + o6] = ti1 - ti2;
      }
    if (ido & 1) return;
  }
  for (k = 0; k < l1; k++) {
    const int q1 = t0 + ido - 1 + k * ido, q2 = q1 + 2 * t0;
    const int o = ido + 4 * k * ido, q6 = ido + k * ido;
    const float ti1 = -hsqt2 * (cc[q1] + cc[q2]);
    const float tr1 = hsqt2 * (cc[q1] - cc[q2]);
    ch[o - 1] = tr1 + cc[q6 - 1];
    ch[o + 2 * ido - 1] = cc[q6 - 1] - tr1;
    ch[o] = ti1 - cc[q1 + t0];
    ch[o + 2 * ido] = ti1 + cc[q1 + t0];
  }
}

void port_drft_forward(const port_enc *e, int W, float *c) { /* drftf1, lib/smallft.c:572-631 */
  const vamd_xform_tab *x = &e->h.xform[W];
  const int n = x->n, nf = x->fft_nf;
  const float *wa = tabf(e, x->off_fft_wa);
  float *ch = (float *)malloc(sizeof(float) * n);
  int na = 1, l2 = n, iw = n, k1, i;
  for (k1 = 0; k1 < nf; k1++) {
    const int ip = x->fft_fac[nf - k1 - 1], l1 = l2 / ip, ido = n / l2;
    iw -= (ip - 1) * ido;
    na = 1 - na;
    if (ip == 4) {
      if (na)
        rf4(ido, l1, ch, c, wa + iw - 1, wa + iw + ido - 1, wa + iw + 2 * ido - 1);
      else
        rf4(ido, l1, c, ch, wa + iw - 1, wa + iw + ido - 1, wa + iw + 2 * ido - 1);
    } else {
      if (na)
        rf2(ido, l1, ch, c, wa + iw - 1);
      else
        rf2(ido, l1, c, ch, wa + iw - 1);
    }
    l2 = l1;
  }
  if (na != 1)
    for (i = 0; i < n; i++) c[i] = ch[i];
  free(ch);
}

/* ---- noise masking, lib/psy.c:547-752 ------------------------------------------ */
static void bark_noise(int n, const int32_t *b, const float *f, float *noise, const float offset,
                       const int fixed) { /* bark_noise_hybridmp, lib/psy.c:547-704 */
  float *N = (float *)malloc(5 * n * sizeof(float)), *X = N + n, *XX = X + n, *Y = XX + n, *XY = Y + n;
  float tN = 0.f, tX = 0.f, tXX = 0.f, tY = 0.f, tXY = 0.f;
  float R = 0.f, A = 0.f, B = 0.f, D = 1.f, w, x, y;
  int i, lo, hi;

  y = f[0] + offset;
  if (y < 1.f) y = 1.f;
  w = y * y * .5;
  tN += w;
  tX += w;
  tY += w * y;
  N[0] = tN;
  X[0] = tX;
  XX[0] = tXX;
  Y[0] = tY;
  XY[0] = tXY;
  for (i = 1, x = 1.f; i < n; i++, x += 1.f) {
    y = f[i] + offset;
    if (y < 1.f) y = 1.f;
    w = y * y;
    tN += w;
    tX += w * x;
    tXX += w * x * x;
    tY += w * y;
    tXY += w * x * y;
    N[i] = tN;
    X[i] = tX;
    XX[i] = tXX;
    Y[i] = tY;
    XY[i] = tXY;
  }

  for (i = 0, x = 0.f; i < n; i++, x += 1.f) {
    lo = b[i] >> 16;
    hi = b[i] & 0xffff;
    if (lo >= 0 || -lo >= n) break;
    if (hi >= n) break;
    tN = N[hi] + N[-lo];
    tX = X[hi] - X[-lo];
    tXX = XX[hi] + XX[-lo];
    tY = Y[hi] + Y[-lo];
    tXY = XY[hi] - XY[-lo];
    A = tY * tXX - tX * tXY;
    B = tN * tXY - tX * tY;
    D = tN * tXX - tX * tX;
    R = (A + x * B) / D;
    if (R < 0.f) R = 0.f;
    noise[i] = R - offset;
  }
  for (; i < n; i++, x += 1.f) {
    lo = b[i] >> 16;
    hi = b[i] & 0xffff;
    if (lo < 0 || lo >= n) break;
    if (hi >= n) break;
    tN = N[hi] - N[lo];
    tX = X[hi] - X[lo];
    tXX = XX[hi] - XX[lo];
    tY = Y[hi] - Y[lo];
    tXY = XY[hi] - XY[lo];
    A = tY * tXX - tX * tXY;
    B = tN * tXY - tX * tY;
    D = tN * tXX - tX * tX;
    R = (A + x * B) / D;
    if (R < 0.f) R = 0.f;
    noise[i] = R - offset;
  }
  for (; i < n; i++, x += 1.f) {
    R = (A + x * B) / D;
    if (R < 0.f) R = 0.f;
    noise[i] = R - offset;
  }

  if (fixed > 0) {
    for (i = 0, x = 0.f; i < n; i++, x += 1.f) {
      hi = i + fixed / 2;
      lo = hi - fixed;
      if (hi >= n) break;
      if (lo >= 0) break;
      tN = N[hi] + N[-lo];
      tX = X[hi] - X[-lo];
      tXX = XX[hi] + XX[-lo];
      tY = Y[hi] + Y[-lo];
      tXY = XY[hi] - XY[-lo];
      A = tY * tXX - tX * tXY;
      B = tN * tXY - tX * tY;
      D = tN * tXX - tX * tX;
      R = (A + x * B) / D;
      if (R - offset < noise[i]) noise[i] = R - offset;
    }
    for (; i < n; i++, x += 1.f) {
      hi = i + fixed / 2;
      lo = hi - fixed;
      if (hi >= n) break;
      if (lo < 0) break;
      tN = N[hi] - N[lo];
      tX = X[hi] - X[lo];
      tXX = XX[hi] - XX[lo];
      tY = Y[hi] - Y[lo];
      tXY = XY[hi] - XY[lo];
      A = tY * tXX - tX * tXY;
      B = tN * tXY - tX * tY;
      D = tN * tXX - tX * tX;
      R = (A + x * B) / D;
      if (R - offset < noise[i]) noise[i] = R - offset;
    }
    for (; i < n; i++, x += 1.f) {
      R = (A + x * B) / D;
      if (R - offset < noise[i]) noise[i] = R - offset;
    }
  }
  free(N);
}

void port_noisemask(const port_enc *e, int psy, const float *logmdct, float *logmask) { /* lib/psy.c:706-752 */
  const vamd_psy_tab *p = &e->h.psy[psy];
  const int n = p->n;
  const int32_t *bark = tabi(e, p->off_bark);
  float *work = (float *)malloc(n * sizeof(float));
  int i;
  bark_noise(n, bark, logmdct, logmask, 140., -1);
  for (i = 0; i < n; i++) work[i] = logmdct[i] - logmask[i];
  bark_noise(n, bark, work, logmask, 0., p->noisewindowfixed);
  for (i = 0; i < n; i++) work[i] = logmdct[i] - work[i];
  for (i = 0; i < n; i++) {
    int dB = logmask[i] + .5;
    if (dB >= VAMD_NOISE_COMPAND_LEVELS) dB = VAMD_NOISE_COMPAND_LEVELS - 1;
    if (dB < 0) dB = 0;
    logmask[i] = work[i] + p->noisecompand[dB];
  }
  free(work);
}

/* ---- tone masking, lib/psy.c:390-545,754-777 ------------------------------------- */
static void seed_curve(float *seed, const float *curves /*[8][58]*/, float amp, int oc, int n, int linesper,
                       float dBoffset) { /* lib/psy.c:390-415 */
  int choice = (int)((amp + dBoffset - 30.) * .1f);
  const float *posts, *curve;
  int i, post1, seedptr;
  if (choice < 0) choice = 0;
  if (choice > VAMD_P_LEVELS - 1) choice = VAMD_P_LEVELS - 1;
  posts = curves + choice * (VAMD_EHMER_MAX + 2);
  curve = posts + 2;
  post1 = (int)posts[1];
  seedptr = oc + (posts[0] - VAMD_EHMER_OFFSET) * linesper - (linesper >> 1);
  for (i = posts[0]; i < post1; i++) {
    if (seedptr > 0) {
      float lin = amp + curve[i];
      if (seed[seedptr] < lin) seed[seedptr] = lin;
    }
    seedptr += linesper;
    if (seedptr >= n) break;
  }
}

static void seed_chase(float *seeds, int linesper, long n) { /* lib/psy.c:454-508 */
  long *posstack = (long *)malloc(n * sizeof(long));
  float *ampstack = (float *)malloc(n * sizeof(float));
  long stack = 0, pos = 0, i;
  for (i = 0; i < n; i++) {
    if (stack < 2) {
      posstack[stack] = i;
      ampstack[stack++] = seeds[i];
      continue;
    }
    for (;;) {
      if (seeds[i] < ampstack[stack - 1]) break; /* push below */
      if (i < posstack[stack - 1] + linesper && stack > 1 && ampstack[stack - 1] <= ampstack[stack - 2] &&
          i < posstack[stack - 2] + linesper) {
        stack--; /* the top entry is completely overlapped */
        continue;
      }
      break;
    }
    posstack[stack] = i;
    ampstack[stack++] = seeds[i];
  }
  for (i = 0; i < stack; i++) {
    long endpos;
    if (i < stack - 1 && ampstack[i + 1] > ampstack[i])
      endpos = posstack[i + 1];
    else
      endpos = posstack[i] + linesper + 1;
    if (endpos > n) endpos = n;
    for (; pos < endpos; pos++) seeds[pos] = ampstack[i];
  }
  free(posstack);
  free(ampstack);
}

void port_tonemask(const port_enc *e, int psy, const float *logfft, float *logmask, float global_specmax,
                   float local_specmax) { /* lib/psy.c:754-777 */
  const vamd_psy_tab *p = &e->h.psy[psy];
  const int n = p->n, nl = p->total_octave_lines, linesper = p->eighth_octave_lines;
  const float *ath = tabf(e, p->off_ath);
  const int32_t *octave = tabi(e, p->off_octave);
  const float *curves = tabf(e, p->off_tonecurves);
  float *seed = (float *)malloc(nl * sizeof(float));
  float att = local_specmax + p->ath_adjatt;
  long i;
  for (i = 0; i < nl; i++) seed[i] = NEG_INF;
  if (att < p->ath_maxatt) att = p->ath_maxatt;
  for (i = 0; i < n; i++) logmask[i] = ath[i] + att;

  { /* seed_loop, lib/psy.c:417-452 */
    const float dBoffset = p->max_curve_dB - global_specmax;
    for (i = 0; i < n; i++) {
      float max = logfft[i];
      long oc = octave[i];
      while (i + 1 < n && octave[i + 1] == oc) {
        i++;
        if (logfft[i] > max) max = logfft[i];
      }
      if (max + 6.f > logmask[i]) {
        oc = oc >> p->shiftoc;
        if (oc >= VAMD_P_BANDS) oc = VAMD_P_BANDS - 1;
        if (oc < 0) oc = 0;
        seed_curve(seed, curves + oc * VAMD_P_LEVELS * (VAMD_EHMER_MAX + 2), max, octave[i] - p->firstoc, nl,
                   linesper, dBoffset);
      }
    }
  }

  { /* max_seeds, lib/psy.c:512-545 */
    long linpos = 0, pos;
    seed_chase(seed, linesper, nl);
    pos = octave[0] - p->firstoc - (linesper >> 1);
    while (linpos + 1 < n) {
      float minV = seed[pos];
      long end = ((octave[linpos] + octave[linpos + 1]) >> 1) - p->firstoc;
      if (minV > p->tone_abs_limit) minV = p->tone_abs_limit;
      while (pos + 1 <= end) {
        pos++;
        if ((seed[pos] > NEG_INF && seed[pos] < minV) || minV == NEG_INF) minV = seed[pos];
      }
      end = pos + p->firstoc;
      for (; linpos < n && octave[linpos] <= end; linpos++)
        if (logmask[linpos] < minV) logmask[linpos] = minV;
    }
    {
      float minV = seed[nl - 1];
      for (; linpos < n; linpos++)
        if (logmask[linpos] < minV) logmask[linpos] = minV;
    }
  }
  free(seed);
}

/* ---- _vp_offset_and_mix (offset_select 1), lib/psy.c:779-835 --------------------- */
static void offset_and_mix(const port_enc *e, int psy, const float *noise, const float *tone, int offset_select,
                           float *logmask, float *mdct, const float *logmdct) {
  const vamd_psy_tab *p = &e->h.psy[psy];
  const int n = p->n;
  const float *noff = tabf(e, p->off_noiseoffset) + (size_t)offset_select * n; /* noiseoffset[offset_select] */
  const float toneatt = p->tone_masteratt[offset_select], cx = p->m_val;
  float de, coeffi;
  int i;
  for (i = 0; i < n; i++) {
    float val = noise[i] + noff[i];
    float t;
    if (val > p->noisemaxsupp) val = p->noisemaxsupp;
    t = tone[i] + toneatt;
    logmask[i] = (val < t) ? t : val;
    if (offset_select == 1) { /* AoTuV M1 touches the spectrum for the middle curve only, lib/psy.c:807 */
      coeffi = -17.2;
      val = val - logmdct[i];
      if (val > coeffi) {
        de = 1.0 - ((val - coeffi) * 0.005 * cx);
        if (de < 0) de = 0.0001;
      } else
        de = 1.0 - ((val - coeffi) * 0.0003 * cx);
      mdct[i] *= de;
    }
  }
}

/* ---- floor 1 fit, lib/floor1.c:406-729 -------------------------------------------- */
typedef struct {
  int x0, x1;
  int xa, ya, x2a, y2a, xya, an;
  int xb, yb, x2b, y2b, xyb, bn;
} fit_acc;

static int acc_fit(const float *flr, const float *mdct, int x0, int x1, fit_acc *a, int n, float atten) {
  long i;
  memset(a, 0, sizeof(*a));
  a->x0 = x0;
  a->x1 = x1;
  if (x1 >= n) x1 = n - 1;
  for (i = x0; i <= x1; i++) {
    int q = dB_quant(flr[i]);
    if (!q) continue;
    if (mdct[i] + atten >= flr[i]) {
      a->xa += i;
      a->ya += q;
      a->x2a += i * i;
      a->y2a += q * q;
      a->xya += i * q;
      a->an++;
    } else {
      a->xb += i;
      a->yb += q;
      a->x2b += i * i;
      a->y2b += q * q;
      a->xyb += i * q;
      a->bn++;
    }
  }
  return a->an;
}

static int line_fit(const fit_acc *a, int fits, int *y0, int *y1, float twofitweight) { /* fit_line */
  double xb = 0, yb = 0, x2b = 0, y2b = 0, xyb = 0, bn = 0;
  const int x0 = a[0].x0, x1 = a[fits - 1].x1;
  int i;
  for (i = 0; i < fits; i++) {
    double weight = (a[i].bn + a[i].an) * twofitweight / (a[i].an + 1) + 1.;
    xb += a[i].xb + a[i].xa * weight;
    yb += a[i].yb + a[i].ya * weight;
    x2b += a[i].x2b + a[i].x2a * weight;
    y2b += a[i].y2b + a[i].y2a * weight;
    xyb += a[i].xyb + a[i].xya * weight;
    bn += a[i].bn + a[i].an * weight;
  }
  if (*y0 >= 0) {
    xb += x0;
    yb += *y0;
    x2b += x0 * x0;
    y2b += *y0 * *y0;
    xyb += *y0 * x0;
    bn++;
  }
  if (*y1 >= 0) {
    xb += x1;
    yb += *y1;
    x2b += x1 * x1;
    y2b += *y1 * *y1;
    xyb += *y1 * x1;
    bn++;
  }
  (void)y2b;
  {
    double denom = (bn * x2b - xb * xb);
    if (denom > 0.) {
      double aa = (yb * x2b - xyb * xb) / denom;
      double bb = (bn * xyb - xb * yb) / denom;
      *y0 = rint(aa + bb * x0);
      *y1 = rint(aa + bb * x1);
      if (*y0 > 1023) *y0 = 1023;
      if (*y1 > 1023) *y1 = 1023;
      if (*y0 < 0) *y0 = 0;
      if (*y1 < 0) *y1 = 0;
      return 0;
    }
    *y0 = 0;
    *y1 = 0;
    return 1;
  }
}

static int inspect_err(int x0, int x1, int y0, int y1, const float *mask, const float *mdct,
                       const vamd_floor1_tab *f) { /* inspect_error, lib/floor1.c:516-565 */
  int dy = y1 - y0, adx = x1 - x0, ady = abs(dy), base = dy / adx;
  int sy = (dy < 0 ? base - 1 : base + 1), x = x0, y = y0, err = 0;
  int val = dB_quant(mask[x]), mse, n = 0;
  ady -= abs(base * adx);
  mse = (y - val);
  mse *= mse;
  n++;
  if (mdct[x] + f->twofitatten >= mask[x]) {
    if (y + f->maxover < val) return 1;
    if (y - f->maxunder > val) return 1;
  }
  while (++x < x1) {
    err += ady;
    if (err >= adx) {
      err -= adx;
      y += sy;
    } else
      y += base;
    val = dB_quant(mask[x]);
    mse += (y - val) * (y - val);
    n++;
    if (mdct[x] + f->twofitatten >= mask[x] && val) {
      if (y + f->maxover < val) return 1;
      if (y - f->maxunder > val) return 1;
    }
  }
  if (f->maxover * f->maxover / n > f->maxerr) return 0;
  if (f->maxunder * f->maxunder / n > f->maxerr) return 0;
  if (mse / n > f->maxerr) return 1;
  return 0;
}

static int mid_Y(const int *A, const int *B, int pos) { /* post_Y */
  if (A[pos] < 0) return B[pos];
  if (B[pos] < 0) return A[pos];
  return (A[pos] + B[pos]) >> 1;
}

static int point_on_line(int x0, int x1, int y0, int y1, int x) { /* render_point, lib/floor1.c:257-271 */
  int dy, adx, ady, off;
  y0 &= 0x7fff;
  y1 &= 0x7fff;
  dy = y1 - y0;
  adx = x1 - x0;
  ady = abs(dy);
  off = ady * (x - x0) / adx;
  return dy < 0 ? y0 - off : y0 + off;
}

/* floor1_fit; returns 1 and fills out[posts], or 0 for the all-zero floor (NULL) */
static int floor_fit(const vamd_floor1_tab *f, const float *logmdct, const float *logmask, int *out) {
  const long n = f->look_n, posts = f->posts;
  fit_acc fits[VAMD_POSIT];
  int fitA[VAMD_POSIT], fitB[VAMD_POSIT], lon[VAMD_POSIT], hin[VAMD_POSIT], memo[VAMD_POSIT];
  long i, j, nonzero = 0;
  for (i = 0; i < posts; i++) {
    fitA[i] = fitB[i] = -200;
    lon[i] = 0;
    hin[i] = 1;
    memo[i] = -1;
  }
  for (i = 0; i < posts - 1; i++)
    nonzero += acc_fit(logmask, logmdct, f->sorted_index[i], f->sorted_index[i + 1], fits + i, n, f->twofitatten);
  if (!nonzero) return 0;
  {
    int y0 = -200, y1 = -200;
    line_fit(fits, posts - 1, &y0, &y1, f->twofitweight);
    fitA[0] = fitB[0] = y0;
    fitA[1] = fitB[1] = y1;
  }
  for (i = 2; i < posts; i++) {
    const int sortpos = f->reverse_index[i], ln = lon[sortpos], hn = hin[sortpos];
    int lsortpos, hsortpos, lx, hx, ly, hy;
    if (memo[ln] == hn) continue;
    lsortpos = f->reverse_index[ln];
    hsortpos = f->reverse_index[hn];
    memo[ln] = hn;
    lx = f->postlist[ln];
    hx = f->postlist[hn];
    ly = mid_Y(fitA, fitB, ln);
    hy = mid_Y(fitA, fitB, hn);
    if (inspect_err(lx, hx, ly, hy, logmask, logmdct, f)) {
      int ly0 = -200, ly1 = -200, hy0 = -200, hy1 = -200;
      const int ret0 = line_fit(fits + lsortpos, sortpos - lsortpos, &ly0, &ly1, f->twofitweight);
      const int ret1 = line_fit(fits + sortpos, hsortpos - sortpos, &hy0, &hy1, f->twofitweight);
      if (ret0) {
        ly0 = ly;
        ly1 = hy0;
      }
      if (ret1) {
        hy0 = ly1;
        hy1 = hy;
      }
      if (ret0 && ret1) {
        fitA[i] = fitB[i] = -200;
      } else {
        fitB[ln] = ly0;
        if (ln == 0) fitA[ln] = ly0;
        fitA[i] = ly1;
        fitB[i] = hy0;
        fitA[hn] = hy1;
        if (hn == 1) fitB[hn] = hy1;
        if (ly1 >= 0 || hy0 >= 0) {
          for (j = sortpos - 1; j >= 0 && hin[j] == hn; j--) hin[j] = i;
          for (j = sortpos + 1; j < posts && lon[j] == ln; j++) lon[j] = i;
        }
      }
    } else {
      fitA[i] = fitB[i] = -200;
    }
  }
  out[0] = mid_Y(fitA, fitB, 0);
  out[1] = mid_Y(fitA, fitB, 1);
  for (i = 2; i < posts; i++) {
    const int ln = f->loneighbor[i - 2], hn = f->hineighbor[i - 2];
    const int predicted = point_on_line(f->postlist[ln], f->postlist[hn], out[ln], out[hn], f->postlist[i]);
    const int vx = mid_Y(fitA, fitB, i);
    out[i] = (vx >= 0 && predicted != vx) ? vx : (predicted | 0x8000);
  }
  return 1;
}

/* the value half of floor1_encode: lib/floor1.c:766-831 (quantise, predict, settle
 * flags) and :923-946 (render_line0); `post` is a private copy */
static int floor_curve(const vamd_floor1_tab *f, const int *fit, int have_fit, int n2, int *ilogmask) {
  int post[VAMD_POSIT];
  long i, j;
  if (!have_fit) {
    memset(ilogmask, 0, n2 * sizeof(int));
    return 0;
  }
  for (i = 0; i < f->posts; i++) {
    int val = fit[i] & 0x7fff;
    switch (f->mult) {
      case 1: val >>= 2; break;
      case 2: val >>= 3; break;
      case 3: val /= 12; break;
      case 4: val >>= 4; break;
    }
    post[i] = val | (fit[i] & 0x8000);
  }
  for (i = 2; i < f->posts; i++) {
    const int ln = f->loneighbor[i - 2], hn = f->hineighbor[i - 2];
    const int predicted = point_on_line(f->postlist[ln], f->postlist[hn], post[ln], post[hn], f->postlist[i]);
    if ((post[i] & 0x8000) || predicted == post[i]) {
      post[i] = predicted | 0x8000;
    } else {
      post[ln] &= 0x7fff;
      post[hn] &= 0x7fff;
    }
  }
  {
    int hx = 0, lx = 0, ly = post[0] * f->mult;
    for (j = 1; j < f->posts; j++) {
      const int cur = f->forward_index[j];
      int hy = post[cur] & 0x7fff;
      if (hy != post[cur]) continue;
      hy *= f->mult;
      hx = f->postlist[cur];
      { /* render_line0, lib/floor1.c:376-403 */
        int dy = hy - ly, adx = hx - lx, ady = abs(dy), base = dy / adx;
        int sy = (dy < 0 ? base - 1 : base + 1), x = lx, y = ly, err = 0, lim = n2;
        ady -= abs(base * adx);
        if (lim > hx) lim = hx;
        if (x < lim) ilogmask[x] = y;
        while (++x < lim) {
          err += ady;
          if (err >= adx) {
            err -= adx;
            y += sy;
          } else
            y += base;
          ilogmask[x] = y;
        }
      }
      lx = hx;
      ly = hy;
    }
    for (j = hx; j < n2; j++) ilogmask[j] = ly;
  }
  return 1;
}

/* ---- couple / quantise / normalise, lib/psy.c:918-1213 ------------------------------ */
static const double stereo_thr[] = {0.0, .5, 1.0, 1.5, 2.5, 4.5, 8.5, 16.5, 9e10};
static const double stereo_thr_limited[] = {0.0, .5, 1.0, 1.5, 2.0, 2.5, 4.5, 8.5, 9e10};

/* stable descending order by *key -- what glibc's qsort (a merge sort at these
 * sizes) does with apsort, lib/psy.c:918-922 */
static void sort_desc(float **v, int count) {
  int i, j;
  for (i = 1; i < count; i++) {
    float *t = v[i];
    for (j = i; j > 0 && *v[j - 1] < *t; j--) v[j] = v[j - 1];
    v[j] = t;
  }
}

static float noise_norm(const vamd_psy_tab *p, int limit, float *r, float *q, float *f, int *flags, float acc,
                        int i, int n, int *out) { /* noise_normalize, lib/psy.c:941-1010 */
  float *sort[64];
  int j, count = 0;
  int start = (p->normal_p ? p->normal_start - i : n);
  if (start > n) start = n;
  acc = 0.f;
  for (j = 0; j < start; j++) {
    if (!flags || !flags[j]) {
      float ve = q[j] / f[j];
      if (r[j] < 0)
        out[j] = -rint(sqrt(ve));
      else
        out[j] = rint(sqrt(ve));
    }
  }
  for (; j < n; j++) {
    if (!flags || !flags[j]) {
      float ve = q[j] / f[j];
      if (ve < .25f && (!flags || j >= limit - i)) {
        acc += ve;
        sort[count++] = q + j;
      } else {
        if (r[j] < 0)
          out[j] = -rint(sqrt(ve));
        else
          out[j] = rint(sqrt(ve));
        q[j] = out[j] * out[j] * f[j];
      }
    }
  }
  if (count) {
    sort_desc(sort, count);
    for (j = 0; j < count; j++) {
      int k = sort[j] - q;
      if (acc >= p->normal_thresh) {
        out[k] = unit_norm(r[k]);
        acc -= 1.f;
        q[k] = f[k];
      } else {
        out[k] = 0;
        q[k] = 0.f;
      }
    }
  }
  return acc;
}

static void couple_quantize(const port_enc *e, int psy, int W, int blob, float **mdct, int **iwork, int *nonzero) {
  const vamd_psy_tab *p = &e->h.psy[psy];
  const vamd_mode_tab *m = &e->h.mode[W];
  const vamd_psy_global_tab *g = &e->h.psy_g;
  const int ch = e->h.channels;
  const int n = p->n;
  const int partition = (p->normal_p ? p->normal_partition : 16);
  const int limit = g->coupling_pointlimit[p->blockflag][blob];
  const int sliding_lowpass = g->sliding_lowpass[W][blob];
  float prepoint = stereo_thr[g->coupling_prepointamp[blob]];
  float postpoint = stereo_thr[g->coupling_postpointamp[blob]];
  float raw[VAMD_MAX_CH][64], quant[VAMD_MAX_CH][64], flo[VAMD_MAX_CH][64];
  int flag[VAMD_MAX_CH][64], nz[VAMD_MAX_CH];
  float acc[VAMD_MAX_CH + VAMD_MAX_COUPLING];
  int step;
  int i, j, k;
  if (n > 1000) postpoint = stereo_thr_limited[g->coupling_postpointamp[blob]];
  if (partition > 64) return; /* scratch rows hold 64 bins; libvorbisenc never exceeds 32 */
  for (i = 0; i < ch + m->coupling_steps; i++) acc[i] = 0.f;

  for (i = 0; i < n; i += partition) {
    const int jn = partition > n - i ? n - i : partition;
    int track = 0;
    memcpy(nz, nonzero, sizeof(*nz) * ch);
    memset(flag, 0, sizeof(flag));
    for (k = 0; k < ch; k++) {
      int *iout = &iwork[k][i];
      if (nz[k]) {
        for (j = 0; j < jn; j++) flo[k][j] = inverse_dB(iout[j]);
        for (j = 0; j < jn; j++) { /* flag_lossless, lib/psy.c:924-935 */
          float point = j >= limit - i ? postpoint : prepoint;
          float r = fabs(mdct[k][i + j]) / flo[k][j];
          flag[k][j] = r < point ? 0 : 1;
        }
        for (j = 0; j < jn; j++) {
          quant[k][j] = raw[k][j] = mdct[k][i + j] * mdct[k][i + j];
          if (mdct[k][i + j] < 0.f) raw[k][j] *= -1.f;
          flo[k][j] *= flo[k][j];
        }
        acc[track] = noise_norm(p, limit, raw[k], quant[k], flo[k], NULL, acc[track], i, jn, iout);
      } else {
        for (j = 0; j < jn; j++) {
          flo[k][j] = 1e-10f;
          raw[k][j] = 0.f;
          quant[k][j] = 0.f;
          flag[k][j] = 0;
          iout[j] = 0;
        }
        acc[track] = 0.f;
      }
      track++;
    }
    for (step = 0; step < m->coupling_steps; step++) { /* lib/psy.c:1111-1201 */
      const int Mi = m->coupling_mag[step], Ai = m->coupling_ang[step];
      int *iM = &iwork[Mi][i], *iA = &iwork[Ai][i];
      float *reM = raw[Mi], *reA = raw[Ai], *qeM = quant[Mi], *qeA = quant[Ai];
      float *floorM = flo[Mi], *floorA = flo[Ai];
      int *fM = flag[Mi], *fA = flag[Ai];
      if (nz[Mi] || nz[Ai]) {
        nz[Mi] = nz[Ai] = 1;
        for (j = 0; j < jn; j++) {
          if (j < sliding_lowpass - i) {
            if (fM[j] || fA[j]) {
              int A, B;
              reM[j] = fabs(reM[j]) + fabs(reA[j]);
              qeM[j] = qeM[j] + qeA[j];
              fM[j] = fA[j] = 1;
              A = iM[j];
              B = iA[j];
              if (abs(A) > abs(B)) {
                iA[j] = (A > 0 ? A - B : B - A);
              } else {
                iA[j] = (B > 0 ? A - B : B - A);
                iM[j] = B;
              }
              if (iA[j] >= abs(iM[j]) * 2) {
                iA[j] = -iA[j];
                iM[j] = -iM[j];
              }
            } else {
              if (j < limit - i) {
                reM[j] += reA[j];
                qeM[j] = fabs(reM[j]);
              } else {
                if (reM[j] + reA[j] < 0)
                  reM[j] = -(qeM[j] = fabs(reM[j]) + fabs(reA[j]));
                else
                  reM[j] = (qeM[j] = fabs(reM[j]) + fabs(reA[j]));
              }
              reA[j] = qeA[j] = 0.f;
              fA[j] = 1;
              iA[j] = 0;
            }
          }
          floorM[j] = floorA[j] = floorM[j] + floorA[j];
        }
        acc[track] = noise_norm(p, limit, raw[Mi], quant[Mi], flo[Mi], flag[Mi], acc[track], i, jn, iM);
        track++;
      }
    }
  }
  for (step = 0; step < m->coupling_steps; step++) /* lib/psy.c:1204-1212 */
    if (nonzero[m->coupling_mag[step]] || nonzero[m->coupling_ang[step]]) {
      nonzero[m->coupling_mag[step]] = 1;
      nonzero[m->coupling_ang[step]] = 1;
    }
}

/* ---- residue back-end, type 2 (lib/res0.c:322-382,479-532,534-640,766-809) ----------------
 * What res2_class decides and which codebook entries res2_forward emits, in emission order
 * (the Huffman/bit-packing of those entries is host code and not restated).  Sequential like
 * the reference: one running work vector, stages outermost.  Integer arithmetic throughout. */
static int book_besterror(const port_enc *e, const vamd_book_tab *bk, int *a) { /* local_book_besterror */
  const signed char *len = (const signed char *)(e->blob + bk->off_lengths);
  const int dim = bk->dim, minval = bk->minval, del = bk->delta, qv = bk->quantvals, ze = qv >> 1;
  int i, j, o, index = 0;
  int p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (del != 1) {
    for (i = 0, o = dim; i < dim; i++) {
      int v = (a[--o] - minval + (del >> 1)) / del;
      int m = (v < ze ? ((ze - v) << 1) - 1 : ((v - ze) << 1));
      index = index * qv + (m < 0 ? 0 : (m >= qv ? qv - 1 : m));
      p[o] = v * del + minval;
    }
  } else {
    for (i = 0, o = dim; i < dim; i++) {
      int v = a[--o] - minval;
      int m = (v < ze ? ((ze - v) << 1) - 1 : ((v - ze) << 1));
      index = index * qv + (m < 0 ? 0 : (m >= qv ? qv - 1 : m));
      p[o] = v * del + minval;
    }
  }
  if (len[index] <= 0) { /* not a populated entry: exhaustive search over the lattice, :349-376 */
    int best = -1;
    int ev[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int maxval = minval + del * (qv - 1);
    for (i = 0; i < bk->entries; i++) {
      if (len[i] > 0) {
        int this = 0;
        for (j = 0; j < dim; j++) {
          int val = ev[j] - a[j];
          this += val * val;
        }
        if (best == -1 || this < best) {
          memcpy(p, ev, sizeof(p));
          best = this;
          index = i;
        }
      }
      j = 0;
      while (ev[j] >= maxval) ev[j++] = 0;
      if (ev[j] >= 0) ev[j] += del;
      ev[j] = -ev[j];
    }
  }
  if (index > -1)
    for (i = 0; i < dim; i++) *a++ -= p[i];
  return index;
}

/* One submap's residue (lib/mapping0.c:660-684 hands the back-end the submap's bundle of channels):
 * type 2 classifies and codes the bundle interleaved as one vector (res2_class / res2_forward,
 * :766-809); type 1 every channel with a non-trivial floor on its own (res1_class / res1_forward,
 * :729-762, _01class :412-470) -- _01forward then walks stage, partition, channel (:585-636; the phrase
 * words it interleaves are host / k_pack business and not restated).  Classes are appended to the taps
 * partition-major, channel-minor; entries in emission order.  Returns 0, or -1 for residue type 0. */
static int residue_submap(const port_enc *e, int W, int sm, int **in, const int *nonzero, int ch, port_taps *t) {
  const vamd_residue_tab *r = &e->h.res[W][sm];
  const vamd_book_tab *books = (const vamd_book_tab *)(e->blob + e->h.off_books);
  const int n2 = e->h.blocksizes[W] / 2;
  const int spp = r->grouping, nparts = r->partitions, n = r->end - r->begin, partvals = n / spp;
  int *work, *cls, *src[VAMD_MAX_CH];
  long i, j, k, l, s, q, used = 0, streams;
  if (r->type != 1 && r->type != 2) return -1;
  for (i = 0; i < ch; i++)
    if (nonzero[i]) src[used++] = in[i];
  if (!used) return 0; /* res*_class returns NULL, res*_forward writes nothing */
  streams = r->type == 2 ? 1 : used;
  cls = (int *)malloc(sizeof(int) * (partvals * streams + 1));
  if (r->type == 1) { /* _01class over the coded channels, :436-453 */
    const float scale = 100. / spp;
    for (i = 0; i < partvals; i++) {
      const int offset = (int)(i * spp + r->begin);
      for (q = 0; q < used; q++) {
        int max = 0, ent = 0;
        for (k = 0; k < spp; k++) {
          if (abs(src[q][offset + k]) > max) max = abs(src[q][offset + k]);
          ent += abs(src[q][offset + k]);
        }
        ent *= scale;
        for (k = 0; k < nparts - 1; k++)
          if (max <= r->classmetric1[k] && (r->classmetric2[k] < 0 || ent < r->classmetric2[k])) break;
        cls[i * streams + q] = (int)k;
      }
    }
  } else
  for (i = 0, l = r->begin / ch; i < partvals; i++) { /* _2class, :501-518 (all channels of the bundle, coded or not) */
    int magmax = 0, angmax = 0;
    for (j = 0; j < spp; j += ch) {
      if (abs(in[0][l]) > magmax) magmax = abs(in[0][l]);
      for (k = 1; k < ch; k++)
        if (abs(in[k][l]) > angmax) angmax = abs(in[k][l]);
      l++;
    }
    for (j = 0; j < nparts - 1; j++)
      if (magmax <= r->classmetric1[j] && angmax <= r->classmetric2[j]) break;
    cls[i] = (int)j;
  }
  for (i = 0; i < partvals * streams; i++) {
    if (t->res_class && t->res_partvals < t->res_class_cap) t->res_class[t->res_partvals] = cls[i];
    t->res_partvals++;
  }
  work = (int *)malloc(sizeof(int) * ch * n2);
  if (r->type == 2) { /* res2_forward's interleaved working vector, :791-797 */
    for (i = 0; i < ch; i++)
      for (j = 0, k = i; j < n2; j++, k += ch) work[k] = in[i][j];
  } else {
    for (q = 0; q < used; q++) memcpy(work + q * n2, src[q], sizeof(int) * n2);
  }
  for (s = 0; s < r->stages; s++) /* _01forward, :585-636 */
    for (i = 0; i < partvals; i++) {
      const long offset = i * spp + r->begin;
      for (q = 0; q < streams; q++) {
        const int c = cls[i * streams + q];
        if (r->secondstages[c] & (1 << s)) {
          const int bn = r->partbooks[c][s];
          if (bn >= 0) {
            const vamd_book_tab *bk = books + bn;
            const int step = spp / bk->dim;
            for (k = 0; k < step; k++) { /* _encodepart, :396-410 */
              int entry = book_besterror(e, bk, work + (r->type == 2 ? 0 : q * n2) + offset + k * bk->dim);
              if (t->res_entries && t->res_count < t->res_entries_cap) t->res_entries[t->res_count] = (unsigned short)entry;
              t->res_count++;
            }
          }
        }
      }
    }
  free(work);
  free(cls);
  return 0;
}

/* every submap in order, lib/mapping0.c:660-684 */
static int residue_all(const port_enc *e, int W, int **iwork, const int *nonzero, port_taps *t) {
  const vamd_mode_tab *m = &e->h.mode[W];
  int sm, c;
  t->res_partvals = 0;
  t->res_count = 0;
  for (sm = 0; sm < m->submaps; sm++) {
    int *bundle[VAMD_MAX_CH], zb[VAMD_MAX_CH], nb = 0;
    for (c = 0; c < e->h.channels; c++)
      if (m->chmuxlist[c] == sm) {
        zb[nb] = nonzero[c] ? 1 : 0;
        bundle[nb++] = iwork[c];
      }
    if (residue_submap(e, W, sm, bundle, zb, nb, t)) return -1;
  }
  return 0;
}

/* ---- the block: mapping0_forward's VBR path, lib/mapping0.c:254-646 ------------------ */
/* floor1_interpolate_fit, lib/floor1.c:731-750; returns whether a curve exists */
static int interpolate_fit(const vamd_floor1_tab *f, const int *A, int haveA, const int *B, int haveB, int del,
                           int *out) {
  int i;
  memset(out, 0, sizeof(int) * VAMD_POSIT);
  if (!(haveA && haveB)) return 0;
  for (i = 0; i < f->posts; i++) {
    out[i] = ((65536 - del) * (A[i] & 0x7fff) + del * (B[i] & 0x7fff) + 32768) >> 16;
    if (A[i] & 0x8000 && B[i] & 0x8000) out[i] |= 0x8000;
  }
  return 1;
}

static int tap_block(const port_enc *e, const float *pcm_in, int lW, int W, int nW, int blocktype, float ampmax_in,
                     port_taps *t, port_mtaps *m);

int port_tap_block(const port_enc *e, const float *pcm_in, int lW, int W, int nW, int blocktype, float ampmax_in,
                   port_taps *t) {
  return tap_block(e, pcm_in, lW, W, nW, blocktype, ampmax_in, t, NULL);
}

/* the same block as a bitrate-managed encoder analyses it: all 15 candidate packets' floors and residues */
int port_tap_block_managed(const port_enc *e, const float *pcm_in, int lW, int W, int nW, int blocktype,
                           float ampmax_in, port_taps *t, port_mtaps *m) {
  return tap_block(e, pcm_in, lW, W, nW, blocktype, ampmax_in, t, m);
}

static int tap_block(const port_enc *e, const float *pcm_in, int lW, int W, int nW, int blocktype, float ampmax_in,
                     port_taps *t, port_mtaps *m) {
  const int ch = e->h.channels, n = e->h.blocksizes[W], n2 = n / 2;
  const int psy = blocktype + (W ? 2 : 0);
  const vamd_floor1_tab *fl;
  float *pcm = (float *)malloc(sizeof(float) * ch * n);
  float *gm = (float *)malloc(sizeof(float) * ch * n2);
  int *iw = (int *)malloc(sizeof(int) * ch * n2);
  float *noise = (float *)malloc(sizeof(float) * n2), *tone = (float *)malloc(sizeof(float) * n2);
  float global_ampmax = ampmax_in, local_ampmax[VAMD_MAX_CH];
  int nonzero[VAMD_MAX_CH], fit[VAMD_MAX_CH][VAMD_POSIT], have[VAMD_MAX_CH];
  int mfit[VAMD_MAX_CH][VAMD_PACKETBLOBS][VAMD_POSIT], mhave[VAMD_MAX_CH][VAMD_PACKETBLOBS];
  float *gmp[VAMD_MAX_CH];
  int *iwp[VAMD_MAX_CH];
  int i, j;
  memcpy(pcm, pcm_in, sizeof(float) * ch * n);

  for (i = 0; i < ch; i++) {
    float scale = 4.f / n, scale_dB, *p = pcm + (size_t)i * n, *logfft = p;
    gmp[i] = gm + (size_t)i * n2;
    iwp[i] = iw + (size_t)i * n2;
    scale_dB = to_dB(scale) + .345;
    port_apply_window(e, p, lW, W, nW);
    if (t->windowed) memcpy(t->windowed + (size_t)i * n, p, n * sizeof(float));
    port_mdct_forward(e, W, p, gmp[i]);
    if (t->mdct_raw) memcpy(t->mdct_raw + (size_t)i * n2, gmp[i], n2 * sizeof(float));
    port_drft_forward(e, W, p);
    if (t->fft_packed) memcpy(t->fft_packed + (size_t)i * n, p, n * sizeof(float));
    logfft[0] = scale_dB + to_dB(p[0]) + .345;
    local_ampmax[i] = logfft[0];
    for (j = 1; j < n - 1; j += 2) {
      float temp = p[j] * p[j] + p[j + 1] * p[j + 1];
      temp = logfft[(j + 1) >> 1] = scale_dB + .5f * to_dB(temp) + .345;
      if (temp > local_ampmax[i]) local_ampmax[i] = temp;
    }
    if (local_ampmax[i] > 0.f) local_ampmax[i] = 0.f;
    if (local_ampmax[i] > global_ampmax) global_ampmax = local_ampmax[i];
    if (t->logfft) memcpy(t->logfft + (size_t)i * n2, logfft, n2 * sizeof(float));
    if (t->local_ampmax) t->local_ampmax[i] = local_ampmax[i];
  }
  for (i = 0; i < ch; i++) {
    float *mdct = gmp[i], *logfft = pcm + (size_t)i * n, *logmdct = logfft + n2, *logmask = logfft;
    fl = &e->h.mode[W].floor[e->h.mode[W].chmuxlist[i]]; /* floorsubmap[chmuxlist[i]], lib/mapping0.c:497 */
    for (j = 0; j < n2; j++) logmdct[j] = to_dB(mdct[j]) + .345;
    if (t->logmdct) memcpy(t->logmdct + (size_t)i * n2, logmdct, n2 * sizeof(float));
    port_noisemask(e, psy, logmdct, noise);
    if (t->noise) memcpy(t->noise + (size_t)i * n2, noise, n2 * sizeof(float));
    port_tonemask(e, psy, logfft, tone, global_ampmax, local_ampmax[i]);
    if (t->tone) memcpy(t->tone + (size_t)i * n2, tone, n2 * sizeof(float));
    offset_and_mix(e, psy, noise, tone, 1, logmask, mdct, logmdct);
    if (t->logmask) memcpy(t->logmask + (size_t)i * n2, logmask, n2 * sizeof(float));
    if (t->mdct) memcpy(t->mdct + (size_t)i * n2, mdct, n2 * sizeof(float));
    memset(fit[i], 0, sizeof(fit[i]));
    have[i] = floor_fit(fl, logmdct, logmask, fit[i]);
    if (t->post_valid) t->post_valid[i] = have[i];
    if (t->posts) memcpy(t->posts + (size_t)i * VAMD_POSIT, fit[i], VAMD_POSIT * sizeof(int));
    if (m) { /* the hi / lo fits and the interpolated curves, lib/mapping0.c:507-573 */
      const int mid = VAMD_PACKETBLOBS / 2, last = VAMD_PACKETBLOBS - 1;
      int k;
      memset(mfit[i], 0, sizeof(mfit[i]));
      memset(mhave[i], 0, sizeof(mhave[i]));
      memcpy(mfit[i][mid], fit[i], sizeof(fit[i]));
      mhave[i][mid] = have[i];
      if (have[i]) {
        offset_and_mix(e, psy, noise, tone, 2, logmask, mdct, logmdct);
        mhave[i][last] = floor_fit(fl, logmdct, logmask, mfit[i][last]);
        offset_and_mix(e, psy, noise, tone, 0, logmask, mdct, logmdct);
        mhave[i][0] = floor_fit(fl, logmdct, logmask, mfit[i][0]);
        for (k = 1; k < mid; k++)
          mhave[i][k] = interpolate_fit(fl, mfit[i][0], mhave[i][0], mfit[i][mid], mhave[i][mid], k * 65536 / mid, mfit[i][k]);
        for (k = mid + 1; k < last; k++)
          mhave[i][k] = interpolate_fit(fl, mfit[i][mid], mhave[i][mid], mfit[i][last], mhave[i][last],
                                        (k - mid) * 65536 / mid, mfit[i][k]);
      }
      for (k = 0; k < VAMD_PACKETBLOBS; k++) {
        if (!mhave[i][k]) memset(mfit[i][k], 0, sizeof(mfit[i][k]));
        if (m->post_valid) m->post_valid[k * ch + i] = mhave[i][k];
        if (m->posts) memcpy(m->posts + ((size_t)k * ch + i) * VAMD_POSIT, mfit[i][k], VAMD_POSIT * sizeof(int));
      }
    }
  }
  if (t->ampmax_out) *t->ampmax_out = global_ampmax;
  if (m) { /* lib/mapping0.c:596-646 for every candidate packet */
    int k;
    for (k = 0; k < VAMD_PACKETBLOBS; k++) {
      for (i = 0; i < ch; i++) {
        nonzero[i] = floor_curve(&e->h.mode[W].floor[e->h.mode[W].chmuxlist[i]], mfit[i][k], mhave[i][k], n2, iwp[i]);
        if (m->ilogmask) memcpy(m->ilogmask + ((size_t)k * ch + i) * n2, iwp[i], n2 * sizeof(int));
      }
      couple_quantize(e, psy, W, k, gmp, iwp, nonzero);
      for (i = 0; i < ch; i++) {
        if (m->iwork) memcpy(m->iwork + ((size_t)k * ch + i) * n2, iwp[i], n2 * sizeof(int));
        if (m->nonzero) m->nonzero[k * ch + i] = nonzero[i];
      }
    }
    free(pcm);
    free(gm);
    free(iw);
    free(noise);
    free(tone);
    return 0;
  }
  for (i = 0; i < ch; i++) {
    nonzero[i] = floor_curve(&e->h.mode[W].floor[e->h.mode[W].chmuxlist[i]], fit[i], have[i], n2, iwp[i]);
    if (t->ilogmask) memcpy(t->ilogmask + (size_t)i * n2, iwp[i], n2 * sizeof(int));
  }
  couple_quantize(e, psy, W, VAMD_PACKETBLOBS / 2, gmp, iwp, nonzero);
  for (i = 0; i < ch; i++) {
    if (t->iwork) memcpy(t->iwork + (size_t)i * n2, iwp[i], n2 * sizeof(int));
    if (t->nonzero) t->nonzero[i] = nonzero[i];
  }
  residue_all(e, W, iwp, nonzero, t);
  free(pcm);
  free(gm);
  free(iw);
  free(noise);
  free(tone);
  return 0;
}

/* ---- the block-switching detector, lib/envelope.c:89-262 ------------------------------
 * Kept in the reference's running-state form (rings, refreshed accumulator, ve->stretch),
 * deliberately unlike the replay / all-stretch-values formulation of vorbis_amd/csrc/k_envelope.h.
 * Pinned against the reference's own _ve_envelope_search by tests/test_envelope.py. */
typedef struct port_env_filter { /* envelope_filter_state, lib/envelope.h:34-45 */
  float ampbuf[VAMD_VE_AMP];
  int ampptr;
  float nearDC[VAMD_VE_NEARDC];
  float nearDC_acc, nearDC_partialacc;
  int nearptr;
} port_env_filter;

typedef struct port_env_state {
  int stretch; /* ve->stretch */
  port_env_filter f[VAMD_MAX_CH][VAMD_VE_BANDS];
} port_env_state;

/* _ve_amp, lib/envelope.c:89-215 */
static int env_amp(const port_enc *e, const float *data, port_env_filter *filters, int ve_stretch) {
  const vamd_envelope_tab *t = &e->h.env;
  const int n = t->winlength;
  float vec[1024];
  const float *win = tabf(e, t->off_window);
  int ret = 0, i, j;
  float decay, minV = t->minenergy;
  int stretch = VAMD_VE_MINSTRETCH > ve_stretch / 2 ? VAMD_VE_MINSTRETCH : ve_stretch / 2;
  float penalty = t->stretch_penalty - (ve_stretch / 2 - VAMD_VE_MINSTRETCH);
  if (penalty < 0.f) penalty = 0.f;
  if (penalty > t->stretch_penalty) penalty = t->stretch_penalty;

  for (i = 0; i < n; i++) vec[i] = data[i] * win[i];
  mdct_fwd(n, t->log2n, t->mdct_scale, tabf(e, t->off_mdct_trig), tabi(e, t->off_mdct_bitrev), vec, vec);

  { /* near-DC spreading, :120-144 */
    float temp = vec[0] * vec[0] + .7 * vec[1] * vec[1] + .2 * vec[2] * vec[2];
    int ptr = filters->nearptr;
    if (ptr == 0) {
      decay = filters->nearDC_acc = filters->nearDC_partialacc + temp;
      filters->nearDC_partialacc = temp;
    } else {
      decay = filters->nearDC_acc += temp;
      filters->nearDC_partialacc += temp;
    }
    filters->nearDC_acc -= filters->nearDC[ptr];
    filters->nearDC[ptr] = temp;
    decay *= (1. / (VAMD_VE_NEARDC + 1));
    filters->nearptr++;
    if (filters->nearptr >= VAMD_VE_NEARDC) filters->nearptr = 0;
    decay = to_dB(decay) * .5 - 15.f;
  }

  for (i = 0; i < n / 2; i += 2) { /* :149-156 */
    float val = vec[i] * vec[i] + vec[i + 1] * vec[i + 1];
    val = to_dB(val) * .5f;
    if (val < decay) val = decay;
    if (val < minV) val = minV;
    vec[i >> 1] = val;
    decay -= 8.;
  }

  for (j = 0; j < VAMD_VE_BANDS; j++) { /* :161-205 */
    float acc = 0.;
    float valmax, valmin;
    for (i = 0; i < t->band_end[j]; i++) acc += vec[i + t->band_begin[j]] * t->band_window[j][i];
    acc *= t->band_total[j];
    {
      int p, this = filters[j].ampptr;
      float postmax, postmin, premax = -99999.f, premin = 99999.f;
      p = this;
      p--;
      if (p < 0) p += VAMD_VE_AMP;
      postmax = acc > filters[j].ampbuf[p] ? acc : filters[j].ampbuf[p];
      postmin = acc < filters[j].ampbuf[p] ? acc : filters[j].ampbuf[p];
      for (i = 0; i < stretch; i++) {
        p--;
        if (p < 0) p += VAMD_VE_AMP;
        premax = premax > filters[j].ampbuf[p] ? premax : filters[j].ampbuf[p];
        premin = premin < filters[j].ampbuf[p] ? premin : filters[j].ampbuf[p];
      }
      valmin = postmin - premin;
      valmax = postmax - premax;
      filters[j].ampbuf[this] = acc;
      filters[j].ampptr++;
      if (filters[j].ampptr >= VAMD_VE_AMP) filters[j].ampptr = 0;
    }
    if (valmax > t->preecho_thresh[j] + penalty) {
      ret |= 1;
      ret |= 4;
    }
    if (valmin < t->postecho_thresh[j] - penalty) ret |= 2;
  }
  return ret;
}

/* the step loop of _ve_envelope_search, lib/envelope.c:234-259 (flags out; the mark[] update
 * is the caller's).  pcm[ch][len], step j reads pcm[c][j*searchstep ..+winlength). */
int port_envelope_steps(const port_enc *e, port_env_state *st, const float *pcm, long len, long nsteps,
                        unsigned char *ret_out) {
  const vamd_envelope_tab *t = &e->h.env;
  long j;
  int i;
  if (t->winlength > 1024 || (nsteps - 1) * t->searchstep + t->winlength > len) return -1;
  for (j = 0; j < nsteps; j++) {
    int ret = 0;
    st->stretch++;
    if (st->stretch > VAMD_VE_MAXSTRETCH * 2) st->stretch = VAMD_VE_MAXSTRETCH * 2;
    for (i = 0; i < e->h.channels; i++) ret |= env_amp(e, pcm + (size_t)i * len + t->searchstep * j, st->f[i], st->stretch);
    ret_out[j] = (unsigned char)ret;
    if (ret & 4) st->stretch = -1;
  }
  return 0;
}

/* _vp_ampmax_decay, lib/psy.c:837-848 */
float port_ampmax_decay(const port_enc *e, float amp, int W) {
  int n = e->h.blocksizes[W] / 2;
  float secs = (float)n / e->h.rate;
  amp += secs * e->h.psy_g.ampmax_att_per_sec;
  if (amp < -9999) amp = -9999;
  return amp;
}

/* wall-clock seconds for `reps` passes over pcm[nblocks][ch][n] long blocks, batch convention
 * (1,1,1), LONG, ampmax_in=-9999 */
double port_time_dsp(const port_enc *e, const float *pcm, long nblocks, int reps) {
  struct timespec t0, t1;
  const int n = e->h.blocksizes[1], ch = e->h.channels;
  port_taps t;
  long k;
  int r;
  memset(&t, 0, sizeof(t));
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (r = 0; r < reps; r++)
    for (k = 0; k < nblocks; k++) port_tap_block(e, pcm + (size_t)k * ch * n, 1, 1, 1, 1, -9999.f, &t);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
}
