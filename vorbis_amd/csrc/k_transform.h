// k_transform.h -- stage 1 of the per-block analysis, one wavefront per
// channel-block: window -> forward MDCT -> real FFT -> logfft / logmdct / local
// ampmax.  Covers SURVEY.md 8a rows a1, a3, a5, a6, a7
// (reference lib/mapping0.c:254-360,384-385).
//
// Exactness: every butterfly is the reference's expression tree evaluated in
// fp32 with no contraction (build flag -ffp-contract=off); all trig/twiddle/
// window values come from the host-built tables in the setup blob.  Only the
// *schedule* differs: each stage's independent butterflies are spread across
// the 64 lanes with the work vectors in LDS.
//
// LDS: A[n+4] (PCM -> windowed -> FFT buffer "c"), B[n + n/32] (MDCT work "w" with its
// padded butterfly half, then FFT buffer "ch"; n/32 >= 4 covers the offset layout).
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"

namespace vamd {

// A block of PCM held in registers (lane l owns quads l, l+64, ...): fetched from HBM one
// block ahead of its use so that the load latency hides behind the previous block's
// transforms (persistent kernels), then windowed on its way into LDS.
struct PcmTile {
  float v[VAMD_QPL2][4];
};

VAMD_DEV void pcm_fetch(PcmTile &t, const float *__restrict__ pcm, int n) {
  LANE_QUADS2(kq, q, n >> 2) f4_get(((const F4 *)pcm)[q], t.v[kq]);
}

// _vorbis_apply_window, lib/window.c:2102-2135, applied while the tile is written to LDS.
VAMD_DEV void window_store(const XformP &P, int W, int lW, int nW, const PcmTile &t, float *A, bool apply_window) {
  const int n = P.n;
  if (!apply_window) {
    LANE_QUADS2(kq, q, n >> 2)((F4 *)A)[q] = f4_make(t.v[kq]);
    return;
  }
  lW = W ? lW : 0;
  nW = W ? nW : 0;
  const int ln = lW ? P.bs1 : P.bs0;
  const int rn = nW ? P.bs1 : P.bs0;
  const float *winL = lW ? P.win_long : P.win_short;
  const float *winR = nW ? P.win_long : P.win_short;
  const int leftbegin = n / 4 - ln / 4, leftend = leftbegin + ln / 2;
  const int rightbegin = n / 2 + n / 4 - rn / 4, rightend = rightbegin + rn / 2;
  // every boundary is a multiple of 4 (block sizes are powers of two >= 64), so a
  // 16-byte quad never straddles two regions
  LANE_QUADS2(kq, q, n >> 2) {
    const int i = q << 2;
    float v[4] = {t.v[kq][0], t.v[kq][1], t.v[kq][2], t.v[kq][3]};
    if (i < leftbegin || i >= rightend) {
      v[0] = v[1] = v[2] = v[3] = 0.f;
    } else if (i < leftend) {
      float w[4];
      f4_get(*(const F4 *)(winL + (i - leftbegin)), w);
      v[0] *= w[0]; v[1] *= w[1]; v[2] *= w[2]; v[3] *= w[3];
    } else if (i >= rightbegin) {
      float w[4];  // falling slope = the rising half-window read backwards
      f4_get(*(const F4 *)(winR + (rn / 2 - 4 - (i - rightbegin))), w);
      v[0] *= w[3]; v[1] *= w[2]; v[2] *= w[1]; v[3] *= w[0];
    }
    ((F4 *)A)[q] = f4_make(v);
  }
}

// The same window for blocks the register tile does not hold (n > 2048): HBM -> LDS directly.
VAMD_DEV void window_store_hbm(const XformP &P, int W, int lW, int nW, const float *__restrict__ pcm, float *A) {
  const int n = P.n;
  lW = W ? lW : 0;
  nW = W ? nW : 0;
  const int ln = lW ? P.bs1 : P.bs0;
  const int rn = nW ? P.bs1 : P.bs0;
  const float *winL = lW ? P.win_long : P.win_short;
  const float *winR = nW ? P.win_long : P.win_short;
  const int leftbegin = n / 4 - ln / 4, leftend = leftbegin + ln / 2;
  const int rightbegin = n / 2 + n / 4 - rn / 4, rightend = rightbegin + rn / 2;
  WAVE_FOR(q, n >> 2) {
    const int i = q << 2;
    float v[4];
    f4_get(((const F4 *)pcm)[q], v);
    if (i < leftbegin || i >= rightend) {
      v[0] = v[1] = v[2] = v[3] = 0.f;
    } else if (i < leftend) {
      float w[4];
      f4_get(*(const F4 *)(winL + (i - leftbegin)), w);
      v[0] *= w[0]; v[1] *= w[1]; v[2] *= w[2]; v[3] *= w[3];
    } else if (i >= rightbegin) {
      float w[4];
      f4_get(*(const F4 *)(winR + (rn / 2 - 4 - (i - rightbegin))), w);
      v[0] *= w[3]; v[1] *= w[2]; v[2] *= w[1]; v[3] *= w[0];
    }
    ((F4 *)A)[q] = f4_make(v);
  }
}

// cPI*_8 of lib/mdct.h:43-45
#define VAMD_C1 .92387953251128675613F
#define VAMD_C2 .70710678118654752441F
#define VAMD_C3 .38268343236508977175F

// mdct_butterfly_8, lib/mdct.c:93-114 (in registers)
VAMD_DEV void bfly8(float *x) {
  float r0 = x[6] + x[2], r1 = x[6] - x[2], r2 = x[4] + x[0], r3 = x[4] - x[0];
  x[6] = r0 + r2;
  x[4] = r0 - r2;
  r0 = x[5] - x[1];
  r2 = x[7] - x[3];
  x[0] = r1 + r0;
  x[2] = r1 - r0;
  r0 = x[5] + x[1];
  r1 = x[7] + x[3];
  x[3] = r2 + r3;
  x[1] = r2 - r3;
  x[7] = r1 + r0;
  x[5] = r1 - r0;
}

// mdct_butterfly_16, lib/mdct.c:117-149
VAMD_DEV void bfly16(float *x) {
  float r0 = x[1] - x[9], r1 = x[0] - x[8];
  x[8] += x[0];
  x[9] += x[1];
  x[0] = (r0 + r1) * VAMD_C2;
  x[1] = (r0 - r1) * VAMD_C2;
  r0 = x[3] - x[11];
  r1 = x[10] - x[2];
  x[10] += x[2];
  x[11] += x[3];
  x[2] = r0;
  x[3] = r1;
  r0 = x[12] - x[4];
  r1 = x[13] - x[5];
  x[12] += x[4];
  x[13] += x[5];
  x[4] = (r0 - r1) * VAMD_C2;
  x[5] = (r0 + r1) * VAMD_C2;
  r0 = x[14] - x[6];
  r1 = x[15] - x[7];
  x[14] += x[6];
  x[15] += x[7];
  x[6] = r0;
  x[7] = r1;
  bfly8(x);
  bfly8(x + 8);
}

// mdct_butterfly_32, lib/mdct.c:152-213
VAMD_DEV void bfly32(float *x) {
  float r0 = x[30] - x[14], r1 = x[31] - x[15];
  x[30] += x[14];
  x[31] += x[15];
  x[14] = r0;
  x[15] = r1;
  r0 = x[28] - x[12];
  r1 = x[29] - x[13];
  x[28] += x[12];
  x[29] += x[13];
  x[12] = r0 * VAMD_C1 - r1 * VAMD_C3;
  x[13] = r0 * VAMD_C3 + r1 * VAMD_C1;
  r0 = x[26] - x[10];
  r1 = x[27] - x[11];
  x[26] += x[10];
  x[27] += x[11];
  x[10] = (r0 - r1) * VAMD_C2;
  x[11] = (r0 + r1) * VAMD_C2;
  r0 = x[24] - x[8];
  r1 = x[25] - x[9];
  x[24] += x[8];
  x[25] += x[9];
  x[8] = r0 * VAMD_C3 - r1 * VAMD_C1;
  x[9] = r1 * VAMD_C3 + r0 * VAMD_C1;
  r0 = x[22] - x[6];
  r1 = x[7] - x[23];
  x[22] += x[6];
  x[23] += x[7];
  x[6] = r1;
  x[7] = r0;
  r0 = x[4] - x[20];
  r1 = x[5] - x[21];
  x[20] += x[4];
  x[21] += x[5];
  x[4] = r1 * VAMD_C1 + r0 * VAMD_C3;
  x[5] = r1 * VAMD_C3 - r0 * VAMD_C1;
  r0 = x[2] - x[18];
  r1 = x[3] - x[19];
  x[18] += x[2];
  x[19] += x[3];
  x[2] = (r1 + r0) * VAMD_C2;
  x[3] = (r1 - r0) * VAMD_C2;
  r0 = x[0] - x[16];
  r1 = x[1] - x[17];
  x[16] += x[0];
  x[17] += x[1];
  x[0] = r1 * VAMD_C3 + r0 * VAMD_C1;
  x[1] = r1 * VAMD_C1 - r0 * VAMD_C3;
  bfly16(x);
  bfly16(x + 16);
}

// The butterfly work vector lives in LDS with two floats of padding after every 32:
// logical index p sits at PW(p).  Pairs (even p, p+1) stay adjacent and 8-byte
// aligned, and the 32-point groups that one lane each pull into registers start 34
// floats apart, which spreads the 64 lanes over all LDS banks (at a plain stride of
// 32 every lane would hit the same bank).
#define VAMD_PW(p) ((p) + (((p) >> 5) << 1))
#define VAMD_PW_SIZE(n2) ((n2) + ((n2) >> 4))

// mdct_forward, lib/mdct.c:492-562.  `in` = A (windowed block, LDS, n floats);
// w = work buffer: w[0..n2) plain + padded butterfly vector at w + n2
// (VAMD_PW_SIZE(n2) floats).  The n/2 spectrum is written to out_lds[0..n2).
// LOGS > 0 runs 2^LOGS independent transforms of the same size side by side (transform t at
// in + t*in_stride, w + t*w_stride, out_lds + t*out_stride): every loop then ranges over
// (transform, item) so that small transforms -- the 128-point one of the block-switching
// detector has only two 32-point groups -- still fill the wave.
// LOGN > 0 fixes the transform size at compile time (n = 2^LOGN): loop counts, strides and every index
// expression derived from n then fold into constants and immediate offsets -- the stage is bound by
// instruction issue, and a good part of its instructions is address arithmetic.  0 = take n from P.
template <int LOGS = 0, int LOGN = 0>
VAMD_DEV void mdct_forward_wave(const XformP &P, const float *in0, float *w0, float *out0, PhaseClock &pc,
                                int in_stride = 0, int w_stride = 0, int out_stride = 0) {
  const int n = LOGN ? (1 << LOGN) : P.n, n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
  const int log2n = LOGN ? LOGN : P.log2n;
  const float *__restrict__ trig = P.trig;
// item index -> (transform t, item g) for a loop of `1 << lcount` items per transform
#define VAMD_MDCT_SPLIT(gg, lcount)                            \
  const int t_ = LOGS ? (gg) >> (lcount) : 0;                  \
  const int g_ = LOGS ? (gg) & ((1 << (lcount)) - 1) : (gg);   \
  const float *in = in0 + t_ * in_stride;                      \
  float *w = w0 + t_ * w_stride;                               \
  float *w2 = w + n2; /* padded: use VAMD_PW() */              \
  float *out_lds = out0 + t_ * out_stride;                     \
  (void)in; (void)w; (void)w2; (void)out_lds;

  // fold + pre-twiddle ("window + rotate + step 1"), lib/mdct.c:506-544.
  // Pair p writes w2[2p], w2[2p+1]; the three loops differ in which input
  // quarter is folded with which sign.  x0[0],x0[2] / x1[0],x1[2] of the reference
  // are the .x,.z / .y,.w lanes of two aligned quads of the input.
  WAVE_FOR(pp, n4 << LOGS) {
    VAMD_MDCT_SPLIT(pp, log2n - 2)
    const int p = g_;
    const F2 T = *(const F2 *)(trig + n2 - 2 * (p + 1));
    float r0, r1;
    if (2 * p < n8) {
      const F4 x0 = *(const F4 *)(in + n2 + n4 - 4 * (p + 1));
      const F4 x1 = *(const F4 *)(in + n2 + n4 + 4 * p);
      r0 = x0.z + x1.y;
      r1 = x0.x + x1.w;
    } else if (2 * p < n2 - n8) {
      const F4 x0 = *(const F4 *)(in + n2 + n4 - 4 * (p + 1));
      const F4 x1 = *(const F4 *)(in + 4 * (p - n8 / 2));
      r0 = x0.z - x1.y;
      r1 = x0.x - x1.w;
    } else {
      const F4 x0 = *(const F4 *)(in + n - 4 * (p - (n2 - n8) / 2 + 1));
      const F4 x1 = *(const F4 *)(in + 4 * (p - n8 / 2));
      r0 = -x0.z - x1.y;
      r1 = -x0.x - x1.w;
    }
    F2 o;
    o.x = r1 * T.y + r0 * T.x;
    o.y = r1 * T.x - r0 * T.y;
    *(F2 *)(w2 + VAMD_PW(2 * p)) = o;
  }
  WAVE_SYNC();
  pc.mark(1);

  // mdct_butterflies, lib/mdct.c:316-336, on x = w2, points = n2.
  // Stage s (s = 0 is mdct_butterfly_first, s >= 1 the generic passes) splits
  // x into 2^s sub-blocks of n2>>s points; butterfly q of a sub-block pairs
  // x[pts-2-2q] with x[pts/2-2-2q] and uses T[(4<<s)*q].  n2/4 = n/8 butterflies per
  // stage in total, all independent.
  const int nstages = log2n - 6;  // first + (log2n-7) generic passes
  for (int s = 0; s < nstages; s++) {
    const int pts = n2 >> s, lper = log2n - 3 - s, tstride = 4 << s;  // per = pts/4 = 1 << lper
    WAVE_FOR(gg, n8 << LOGS) {
      VAMD_MDCT_SPLIT(gg, log2n - 3)
      const int g = g_;
      const int j = g >> lper, q = g & ((1 << lper) - 1);
      const int ia = pts * j + pts - 2 - 2 * q, ib = pts * j + (pts >> 1) - 2 - 2 * q;
      F2 *pa = (F2 *)(w2 + VAMD_PW(ia)), *pb = (F2 *)(w2 + VAMD_PW(ib));
      const F2 T = *(const F2 *)(trig + tstride * q);
      F2 a = *pa, b = *pb;
      const float r0 = a.x - b.x, r1 = a.y - b.y;
      a.x += b.x;
      a.y += b.y;
      b.x = r1 * T.y + r0 * T.x;
      b.y = r1 * T.x - r0 * T.y;
      *pa = a;
      *pb = b;
    }
    WAVE_SYNC();
  }
  pc.mark(2);
  // 32-point butterflies, one group per lane, in registers
  WAVE_FOR(gg, (n2 / 32) << LOGS) {
    VAMD_MDCT_SPLIT(gg, log2n - 6)
    const int g = g_;
    float v[32];
    F2 *pg = (F2 *)(w2 + 34 * g);  // == VAMD_PW(32 g)
#if VAMD_GPU
#pragma unroll
#endif
    for (int k = 0; k < 16; k++) {
      const F2 t = pg[k];
      v[2 * k] = t.x;
      v[2 * k + 1] = t.y;
    }
    bfly32(v);
#if VAMD_GPU
#pragma unroll
#endif
    for (int k = 0; k < 16; k++) {
      F2 t;
      t.x = v[2 * k];
      t.y = v[2 * k + 1];
      pg[k] = t;
    }
  }
  WAVE_SYNC();
  pc.mark(3);

  // mdct_bitreverse, lib/mdct.c:346-394: reads x = w2 (upper half), writes the
  // lower half w[0..n2).  Unit u produces w[2u], w[2u+1], w[n2-2u-2], w[n2-2u-1].
  const int *__restrict__ bit = P.bitrev;
  WAVE_FOR(uu, n8 << LOGS) {
    VAMD_MDCT_SPLIT(uu, log2n - 3)
    const int u = g_;
    const I2 bi = *(const I2 *)(bit + 2 * u);
    const F2 x0 = *(const F2 *)(w2 + VAMD_PW(bi.x));
    const F2 x1 = *(const F2 *)(w2 + VAMD_PW(bi.y));
    const F2 T = *(const F2 *)(trig + n + 2 * u);
    float r0 = x0.y - x1.y;
    float r1 = x0.x + x1.x;
    const float r2 = r1 * T.x + r0 * T.y;
    const float r3 = r1 * T.y - r0 * T.x;
    r0 = (x0.y + x1.y) * .5f;
    r1 = (x0.x - x1.x) * .5f;
    F2 lo, hi;
    lo.x = r0 + r2;
    lo.y = r1 + r3;
    hi.x = r0 - r2;
    hi.y = r3 - r1;
    *(F2 *)(w + 2 * u) = lo;
    *(F2 *)(w + n2 - 2 * u - 2) = hi;
  }
  WAVE_SYNC();
  pc.mark(4);

  // final rotate * scale, lib/mdct.c:552-561 -> out[n2]
  WAVE_FOR(ii, n4 << LOGS) {
    VAMD_MDCT_SPLIT(ii, log2n - 2)
    const int i = g_;
    const F2 T = *(const F2 *)(trig + n2 + 2 * i);
    const F2 ab = *(const F2 *)(w + 2 * i);
    out_lds[i] = (ab.x * T.x + ab.y * T.y) * P.mdct_scale;
    out_lds[n2 - 1 - i] = (ab.x * T.y - ab.y * T.x) * P.mdct_scale;
  }
  WAVE_SYNC();
}
#undef VAMD_MDCT_SPLIT

// FFT buffers and the packed layout.  FFTPACK's half-complex packing puts every
// (real, imag) pair at indices (2m-1, 2m): odd first.  Stored plainly that is never
// 8-byte aligned, so each pair costs two LDS accesses.  Every pass therefore WRITES
// its output one float into its buffer ("offset layout": logical index t lives at
// base[t+1]); from the second pass on, sources and destinations are both offset and
// every pair moves as one aligned 64-bit access.  Only the first pass reads the plain
// windowed block (it has ido = 1 or touches single values only).
template <bool AL>
VAMD_DEV F2 ld_pair(const float *p, int t) {  // (p[t-1], p[t]) for even t
  if (AL) return *(const F2 *)(p + t - 1);
  F2 r;
  r.x = p[t - 1];
  r.y = p[t];
  return r;
}
VAMD_DEV void st_pair(float *p, int t, float a, float b) {  // p[t-1] = a, p[t] = b, offset layout
  F2 r;
  r.x = a;
  r.y = b;
  *(F2 *)(p + t - 1) = r;
}

// dradf4, lib/smallft.c:168-268: one radix-4 pass cc -> ch.  wa1/2/3 are the
// reference's 1-based-offset twiddle pointers (wa+iw-1 etc.).  The reference runs three
// loops per pass: over k (the i = 0 column), over (k, i = 2,4,..) and over k again (the
// i = ido column).  Here one flat loop over g = (k, m) with m in [0, ido/2) covers all
// three: m = 0 does both k-only columns, m >= 1 the (k, i = 2m) butterfly.  ido is a power
// of two (>= 4) for every pass but the first, so k and m are a shift and a mask.
template <bool AL>
VAMD_DEV void radf4_wave(int ido, int l1, const float *__restrict__ cc, float *__restrict__ ch, const float *__restrict__ wa1,
                         const float *__restrict__ wa2, const float *__restrict__ wa3) {
  const float hsqt2 = .70710678118654752f;
  const int t0 = l1 * ido;
  if (ido == 1) {
    // first pass: four contiguous outputs per k -> one 16-byte store (offset layout: index 4k at +1)
    WAVE_FOR(k, l1) {
      const float c1 = cc[t0 + k], c2 = cc[3 * t0 + k], c3 = cc[k], c4 = cc[2 * t0 + k];
      const float tr1 = c1 + c2, tr2 = c3 + c4;
      float *o = ch + 4 * k;
      o[0] = tr1 + tr2;
      o[1] = c3 - c4;
      o[2] = c2 - c1;
      o[3] = tr2 - tr1;
    }
    return;
  }
  const int lh = 31 - __builtin_clz((unsigned)(ido >> 1));  // log2(ido/2)
  // the two k-only columns (i = 0 and i = ido), one lane per k
  WAVE_FOR(k, l1) {
    {
      {
        const int t1 = t0 + k * ido, t2 = 3 * t0 + k * ido, t3 = k * ido, t4 = 2 * t0 + k * ido;
        const float tr1 = cc[t1] + cc[t2];
        const float tr2 = cc[t3] + cc[t4];
        int t5 = t3 << 2;
        ch[t5] = tr1 + tr2;
        ch[(ido << 2) + t5 - 1] = tr2 - tr1;
        t5 += ido << 1;
        ch[t5 - 1] = cc[t3] - cc[t4];
        ch[t5] = cc[t2] - cc[t1];
      }
      {
        const int t1 = t0 + ido - 1 + k * ido, t2 = t1 + (t0 << 1);
        const int t4 = ido + k * (ido << 2), t5 = ido << 1, t6 = ido + k * ido;
        const float ti1 = -hsqt2 * (cc[t1] + cc[t2]);
        const float tr1 = hsqt2 * (cc[t1] - cc[t2]);
        ch[t4 - 1] = tr1 + cc[t6 - 1];
        ch[t4 + t5 - 1] = cc[t6 - 1] - tr1;
        ch[t4] = ti1 - cc[t1 + t0];
        ch[t4 + t5] = ti1 + cc[t1 + t0];
      }
    }
  }
  // the (k, i = 2m) butterflies, m >= 1 (the m = 0 slot of each k idles: k and m stay a shift and a mask)
  WAVE_FOR(g, l1 << lh) {
    const int k = g >> lh, m = g & ((1 << lh) - 1);
    if (m != 0) {
      const int i = 2 * m;
      const int t1 = k * ido;
      const int t2 = t1 + i;
      const int t4 = (t1 << 2) + i;
      const int t6 = ido << 1;
      const int t5 = t6 + (t1 << 2) - i;
      const F2 w1 = *(const F2 *)(wa1 + i - 2), w2 = *(const F2 *)(wa2 + i - 2), w3 = *(const F2 *)(wa3 + i - 2);
      const F2 c0 = ld_pair<AL>(cc, t2), c1 = ld_pair<AL>(cc, t2 + t0), c2 = ld_pair<AL>(cc, t2 + 2 * t0),
               c3 = ld_pair<AL>(cc, t2 + 3 * t0);
      const float cr2 = w1.x * c1.x + w1.y * c1.y;
      const float ci2 = w1.x * c1.y - w1.y * c1.x;
      const float cr3 = w2.x * c2.x + w2.y * c2.y;
      const float ci3 = w2.x * c2.y - w2.y * c2.x;
      const float cr4 = w3.x * c3.x + w3.y * c3.y;
      const float ci4 = w3.x * c3.y - w3.y * c3.x;
      const float tr1 = cr2 + cr4, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
      const float ti2 = c0.y + ci3, ti3 = c0.y - ci3;
      const float tr2 = c0.x + cr3, tr3 = c0.x - cr3;
      st_pair(ch, t4, tr1 + tr2, ti1 + ti2);
      st_pair(ch, t5, tr3 - ti4, tr4 - ti3);
      st_pair(ch, t4 + t6, ti4 + tr3, tr4 + ti3);
      st_pair(ch, t5 + t6, tr2 - tr1, ti1 - ti2);
    }
  }
}

// dradf2, lib/smallft.c:113-166, same flattening
template <bool AL>
VAMD_DEV void radf2_wave(int ido, int l1, const float *__restrict__ cc, float *__restrict__ ch, const float *__restrict__ wa1) {
  const int t0 = l1 * ido;
  if (ido == 1) {
    WAVE_FOR(k, l1) {
      ch[2 * k] = cc[k] + cc[t0 + k];
      ch[2 * k + 1] = cc[k] - cc[t0 + k];
    }
    return;
  }
  const int lh = 31 - __builtin_clz((unsigned)(ido >> 1));
  WAVE_FOR(g, l1 << lh) {
    const int k = g >> lh, m = g & ((1 << lh) - 1);
    if (m == 0) {
      {
        const int t1 = k * ido, t2 = t0 + k * ido;
        ch[t1 << 1] = cc[t1] + cc[t2];
        ch[(t1 << 1) + (ido << 1) - 1] = cc[t1] - cc[t2];
      }
      {
        const int t1 = ido + k * (ido << 1), t2 = ido - 1 + t0 + k * ido, t3 = ido - 1 + k * ido;
        ch[t1] = -cc[t2];
        ch[t1 - 1] = cc[t3];
      }
    } else {
      const int i = 2 * m;
      const int t1 = k * ido, t2 = t0 + k * ido;
      const int t3 = t2 + i, t4 = (t1 << 1) + (ido << 1) - i, t5 = t1 + i, t6 = (t1 << 1) + i;
      const F2 w1 = *(const F2 *)(wa1 + i - 2);
      const F2 c3 = ld_pair<AL>(cc, t3), c5 = ld_pair<AL>(cc, t5);
      const float tr2 = w1.x * c3.x + w1.y * c3.y;
      const float ti2 = w1.x * c3.y - w1.y * c3.x;
      st_pair(ch, t6, c5.x + tr2, c5.y + ti2);
      st_pair(ch, t4, c5.x - tr2, ti2 - c5.y);
    }
  }
}

// drftf1, lib/smallft.c:572-631: unnormalised real FFT with the reference's pass
// order and c<->ch ping-pong.  `c` holds the windowed block (plain layout, n+4
// floats available), `ch` is scratch (n+4 floats).  Returns the buffer (offset layout:
// element t at ret[t]) that holds the packed spectrum R0,R1,I1,...,R(n/2).
// LOGN > 0: n = 2^LOGN, whose FFTPACK factorisation is radix 4 throughout with one radix-2 pass at the end
// when LOGN is odd (drfti1 tries 4 first and moves a leftover 2 to the front of ifac[], which drftf1
// walks backwards: lib/smallft.c:35-70,572-631); vamd_create() checks the blob's factors against that.
template <int LOGN = 0>
VAMD_DEV const float *drft_forward_wave(const XformP &P, float *c, float *ch) {
  const int n = LOGN ? (1 << LOGN) : P.n, nf = LOGN ? (LOGN >> 1) + (LOGN & 1) : P.fft_nf;
  const float *__restrict__ wa = P.wa;
  float *bufc = c + 1, *bufh = ch + 1;  // offset layouts of the two buffers
  int na = 1, l2 = n, iw = n;
#if VAMD_GPU
#pragma unroll
#endif
  for (int k1 = 0; k1 < nf; k1++) {
    const int ip = LOGN ? (k1 < (LOGN >> 1) ? 4 : 2) : P.fft_fac[nf - k1 - 1];
    const int l1 = l2 / ip, ido = n / l2;
    iw -= (ip - 1) * ido;
    na = 1 - na;
    float *dst = na ? bufc : bufh;
    if (k1 == 0) {
      // the first pass reads the plain (un-offset) block; ld_pair<false> covers the case
      // where it has pairs to read (ido > 2)
      if (ip == 4)
        radf4_wave<false>(ido, l1, c, dst, wa + iw - 1, wa + iw + ido - 1, wa + iw + 2 * ido - 1);
      else
        radf2_wave<false>(ido, l1, c, dst, wa + iw - 1);
    } else {
      const float *src = na ? bufh : bufc;
      if (ip == 4)
        radf4_wave<true>(ido, l1, src, dst, wa + iw - 1, wa + iw + ido - 1, wa + iw + 2 * ido - 1);
      else
        radf2_wave<true>(ido, l1, src, dst, wa + iw - 1);
    }
    WAVE_SYNC();
    l2 = l1;
  }
  // the reference copies ch back into c when the last pass landed in ch; the caller
  // just reads whichever buffer holds the result
  return na ? bufc : bufh;
}

// The whole stage for one channel-block.
//   pcm      HBM [n]           un-windowed block (vb->pcm[i])
//   A, B     LDS [n] each
//   outputs  HBM, each may be null
// Returns the channel's local_ampmax (all lanes).
// `tile` holds the un-windowed block (pcm_fetch); it is consumed before anything else, so the
// caller may refill it for the next block as soon as this returns from its first phase.
VAMD_DEV void transform_window(const XformP &P, int W, int lW, int nW, const PcmTile &tile, float *A,
                               PhaseClock &pc) {
  window_store(P, W, lW, nW, tile, A, true);
  WAVE_SYNC();
  pc.mark(0);
}

template <int LOGN = 0>
VAMD_DEV float transform_block(const XformP &P, float *A, float *B, float *__restrict__ mdct_out,
                               float *__restrict__ logmdct_out, float *__restrict__ logfft_out, PhaseClock &pc) {
  const int n = LOGN ? (1 << LOGN) : P.n, n2 = n >> 1;

  // MDCT: spectrum lands in B[n2..n) (LDS), then goes out with its dB twin
  mdct_forward_wave<0, LOGN>(P, A, B, B + n2, pc);
  WAVE_FOR(j, n2) {
    const float m = B[n2 + j];
    if (mdct_out) mdct_out[j] = m;
    if (logmdct_out) logmdct_out[j] = todB_345(m);  // lib/mapping0.c:384-385
  }
  WAVE_SYNC();

  pc.mark(5);
  // FFT of the same windowed block (A), ping-ponging with B
  const float *spec = drft_forward_wave<LOGN>(P, A, B);
  pc.mark(6);

  // logfft + local ampmax, lib/mapping0.c:255-346
  const float scale = 4.f / n;
  const float scale_dB = todB_345(scale);
  float amp = -1e30f;
  WAVE_FOR(k, n2) {
    float v;
    if (k == 0) {
      v = (float)((double)(scale_dB + todB(spec[0])) + .345);
    } else {
      const F2 z = *(const F2 *)(spec + 2 * k - 1);  // (Re_k, Im_k), aligned in the offset layout
      const float temp = z.x * z.x + z.y * z.y;
      v = (float)((double)(scale_dB + .5f * todB(temp)) + .345);
    }
    if (logfft_out) logfft_out[k] = v;
    amp = fmaxf(amp, v);
  }
  amp = wave_max(amp);
  if (amp > 0.f) amp = 0.f;
  WAVE_SYNC();
  pc.mark(7);
  return amp;
}

}  // namespace vamd
