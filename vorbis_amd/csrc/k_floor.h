// k_floor.h -- _vp_offset_and_mix (reference lib/psy.c:779-835), floor1_fit
// (lib/floor1.c:576-729) and the post-quantise + render_line0 half of
// floor1_encode (lib/floor1.c:766-831,923-946); SURVEY.md 8a rows a11-a13.  One
// wavefront per channel-block.
//
// Parallel form: the mix is per bin.  accumulate_fit's integer sums take one
// lane per post interval.  The greedy split loop is inherently ordered (each
// decision moves the neighbours of later posts); every lane runs it uniformly
// with the small state in LDS, while inspect_error -- the only part that
// touches O(n) bins -- is evaluated by all 64 lanes at once: the Bresenham line
// has the closed form y(x) = y0 + k*base + sgn*floor(k*ady'/adx), the squared
// error is an integer wave sum and the early-outs are an any().  fit_line's
// fp64 sums run in the reference's term order on every lane.
//
// LDS: qc[n] 16-bit words -- everything the fit reads per bin: the mask quantised by vorbis_dBquant
// (bits 0-9; the fit never looks at the float mask) and the "mdct + twofitatten >= mask" class
// (bit 15; the only thing the fit reads logmdct for) --, FloorScratch (interval accumulators + the
// rendered segment list).
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"
#include "k_tone.h"

namespace vamd {

#define VAMD_MAXPOSTS 32

struct FitAcc {  // lsfit_acc, lib/floor1.c:32-49 (x0/x1 come from sorted_index; y2a/y2b, which the reference
  int xa, ya, x2a, xya, an;  // sums and never reads, are not kept)
  int xb, yb, x2b, xyb, bn;
};

// fit_line's per-interval contribution (lib/floor1.c:463-472): depends only on the
// interval's accumulators, so it is formed once (one lane per interval) and the
// ordered fp64 summation over a range of intervals just adds these up
struct FitTerm {
  double xb, yb, x2b, xyb, bn;
};

// LDS scratch.  Everything the ordered sections chase serially (fit values, neighbour
// maps, the floor's static index tables) is NOT here: it lives one entry per lane in
// registers (LaneInts) and is read with v_readlane.
struct FloorScratch {
  FitAcc acc[VAMD_MAXPOSTS];  // 40 B each; later reused as [intervals][5] doubles (40 B each)
  double pair_sums[16];
};

// _vp_offset_and_mix with offset_select == 1 (the only select the VBR path uses), for one quad of bins:
//   nz / tn / md  noise curve, tone curve, spectrum (md is scaled in place: AoTuV M1);  mk  the mask out
// Returns what the fit reads of the four bins: 16 bits each (see offset_and_mix_wave).
VAMD_DEV I2 offset_and_mix_quad(const PsyP &P, int q, const float *nz, const float *tn, float *md, float *mk, float twofitatten) {
  const float toneatt = P.tone_masteratt1;
  const float cx = P.m_val;
  const float coeffi = -17.2f;  // float coeffi = -17.2 (lib/psy.c:808)
  float no[4], lmv[4];
  f4_get(((const F4 *)P.noiseoffset1)[q], no);
  // logmdct (lib/mapping0.c:384-385) is a function of the spectrum that is read here anyway: recomputed, not
  // fetched -- the transform stage need not write it, nor this one read it (16 KB per stereo block)
  for (int c = 0; c < 4; c++) lmv[c] = todB_345(md[c]);
#pragma unroll
  for (int c = 0; c < 4; c++) {
    float val = nz[c] + no[c];
    if (val > P.noisemaxsupp) val = P.noisemaxsupp;
    const float t = tn[c] + toneatt;
    mk[c] = (val < t) ? t : val;  // max(val, tone+toneatt), lib/psy.c:795 with os.h:78 max()
    // AoTuV M1, lib/psy.c:807-832: double-promoted by the 1.0 / 0.005 / 0.0003 literals
    val = val - lmv[c];
    // (the two arms differ in one literal: the literal is selected, not the arm -- a wave's lanes take both)
    const bool above = val > coeffi;
    float de = (float)(1.0 - ((double)(val - coeffi) * (above ? 0.005 : 0.0003) * (double)cx));
    if (above && de < 0) de = 0.0001f;
    md[c] *= de;
  }
  // accumulate_fit / inspect_error read the mask only through vorbis_dBquant and split bins by the
  // class test (lib/floor1.c:421-427,530-536): 2 bytes per bin instead of two floats
  uint32_t w[2] = {0, 0};
  for (int c = 0; c < 4; c++) {
    const uint32_t v = (uint32_t)dBquant(mk[c]) | (lmv[c] + twofitatten >= mk[c] ? 0x8000u : 0u);
    w[c >> 1] |= v << (16 * (c & 1));
  }
  I2 pk;
  pk.x = (int)w[0];
  pk.y = (int)w[1];
  return pk;
}

VAMD_DEV void offset_and_mix_wave(const PsyP &P, const float *__restrict__ noise, const float *__restrict__ tone,
                                  const float *__restrict__ mdct_io_src, float *__restrict__ mdct_out, float *__restrict__ logmask_out /* HBM or null */,
                                  unsigned short *qc, float twofitatten, PhaseClock &pc) {
  const int n = P.n;
  WAVE_FOR(q, n >> 2) {
    float nz[4], tn[4], md[4], mk[4];
    f4_get(((const F4 *)noise)[q], nz);
    f4_get(((const F4 *)tone)[q], tn);
    f4_get(((const F4 *)mdct_io_src)[q], md);
    const I2 pk = offset_and_mix_quad(P, q, nz, tn, md, mk, twofitatten);
    if (logmask_out) ((F4 *)logmask_out)[q] = f4_make(mk);
    ((I2 *)qc)[q] = pk;
    ((F4 *)mdct_out)[q] = f4_make(md);
  }
  WAVE_SYNC();
  pc.mark(0);
}

// The same with the tone curve formed on the spot (k_tone.h: tone_fold_prepare has left the painted seed lines and the
// groups' minima in LDS): a lane folds its quad and mixes it, the curve never exists in memory unless `tone_out` asks
// for the tap.  The fit's 16-bit state takes the place of the seed lines in LDS, so it waits in registers (two per quad)
// until every lane is through with them.  Blocks of up to 4 * 64 * VAMD_QPL bins.
VAMD_DEV void fold_and_mix_wave(const PsyP &P, float att, const float *seed, const float *gmin,
                                const float *__restrict__ noise, float *__restrict__ tone_out /* HBM or null */,
                                const float *__restrict__ mdct_io_src, float *__restrict__ mdct_out,
                                float *__restrict__ logmask_out /* HBM or null */, unsigned short *qc,
                                float twofitatten, PhaseClock &pc) {
  const int n = P.n;
  I2 keep[VAMD_QPL];
  LANE_QUADS(kq, q, n >> 2) {
    float nz[4], tn[4], md[4], mk[4];
    f4_get(((const F4 *)noise)[q], nz);
    f4_get(((const F4 *)mdct_io_src)[q], md);
    tone_fold_quad(P, att, seed, gmin, q, tn);
    if (tone_out) ((F4 *)tone_out)[q] = f4_make(tn);
    keep[kq] = offset_and_mix_quad(P, q, nz, tn, md, mk, twofitatten);
    if (logmask_out) ((F4 *)logmask_out)[q] = f4_make(mk);
    ((F4 *)mdct_out)[q] = f4_make(md);
  }
  WAVE_SYNC();  // nobody reads seed lines any more
  LANE_QUADS(kq, q, n >> 2)((I2 *)qc)[q] = keep[kq];
  WAVE_SYNC();
  pc.mark(0);
}

// _vp_offset_and_mix's mask for offset_select 0 or 2 (the lo / hi curves of a bitrate-managed block,
// lib/mapping0.c:507-545), reduced to what the fit reads: those selects leave the spectrum alone
// (lib/psy.c:807) and nobody keeps their float mask.
VAMD_DEV void mask_quantise_wave(const PsyP &P, int offset_select, const float *__restrict__ noise,
                                 const float *__restrict__ tone, const float *__restrict__ mdct_raw_in,
                                 unsigned short *qc, float twofitatten) {
  const int n = P.n;
  const float toneatt = offset_select ? P.tone_masteratt2 : P.tone_masteratt0;
  const float *__restrict__ noff = offset_select ? P.noiseoffset2 : P.noiseoffset0;
  WAVE_FOR(q, n >> 2) {
    float nz[4], no[4], tn[4], lmv[4];
    f4_get(((const F4 *)noise)[q], nz);
    f4_get(((const F4 *)noff)[q], no);
    f4_get(((const F4 *)tone)[q], tn);
    f4_get(((const F4 *)mdct_raw_in)[q], lmv);
    for (int c = 0; c < 4; c++) lmv[c] = todB_345(lmv[c]);  // logmdct, lib/mapping0.c:384-385
    uint32_t w[2] = {0, 0};
    for (int c = 0; c < 4; c++) {
      float val = nz[c] + no[c];
      if (val > P.noisemaxsupp) val = P.noisemaxsupp;
      const float t = tn[c] + toneatt;
      const float mk = (val < t) ? t : val;
      const uint32_t v = (uint32_t)dBquant(mk) | (lmv[c] + twofitatten >= mk ? 0x8000u : 0u);
      w[c >> 1] |= v << (16 * (c & 1));
    }
    I2 pk;
    pk.x = (int)w[0];
    pk.y = (int)w[1];
    ((I2 *)qc)[q] = pk;
  }
  WAVE_SYNC();
}

// One record of the fit work list (derive_fit_segments): the lane sums its chunk's share of one interval and adds
// it to that interval's accumulators.  Returns the class-a count (accumulate_fit's return value, summed by the
// caller).  The sums are taken relative to the chunk's first bin and packed several to a register -- for a bin at
// offset c (0..15) holding q (0..1023, 0 = skipped by the reference):
//   [q != 0] * (1 | c << 8 | c*c << 16)   -> count (<= 16), sum c (<= 120), sum c*c (<= 1240)
//   q * (1 | c << 14)                     -> sum q (<= 16368 < 2^14), sum c*q (<= 122760 < 2^17)
// once for all bins and once for class a (mdct + twofitatten >= mask, bit 15); class b is the difference.  With
// i = base + c:  sum i = base*n + sum c,  sum i*i = base*base*n + 2*base*sum c + sum c*c,  sum i*q = base*sum q +
// sum c*q -- integer identities, so the totals are the reference's (lib/floor1.c:416-436).
VAMD_DEV int accumulate_segment(const unsigned int *rec, const unsigned short *qc, FitAcc *acc) {
  const I4 h = ((const I4 *)rec)[0], m0 = ((const I4 *)rec)[1], m1 = ((const I4 *)rec)[2];
  const int chunk = h.x;
  const I4 qa = ((const I4 *)qc)[2 * chunk], qb = ((const I4 *)qc)[2 * chunk + 1];
  const unsigned int w[8] = {(unsigned)(qa.x & m0.x), (unsigned)(qa.y & m0.y), (unsigned)(qa.z & m0.z),
                             (unsigned)(qa.w & m0.w), (unsigned)(qb.x & m1.x), (unsigned)(qb.y & m1.y),
                             (unsigned)(qb.z & m1.z), (unsigned)(qb.w & m1.w)};
  unsigned int cnt_all = 0, qs_all = 0, cnt_a = 0, qs_a = 0;
#pragma unroll
  for (int c = 0; c < 16; c++) {
    const unsigned int hw = (w[c >> 1] >> (16 * (c & 1))) & 0xffffu;
    const unsigned int q = hw & 0x7fffu, a01 = hw >> 15;
    const unsigned int v01 = (q + 1023u) >> 10;  // q <= 1023: 1 where q != 0
    const unsigned int kc = 1u | ((unsigned)c << 8) | ((unsigned)(c * c) << 16), lc = 1u | ((unsigned)c << 14);
    cnt_all += v01 * kc;
    qs_all += q * lc;
    cnt_a += (v01 & a01) * kc;
    qs_a += (q * a01) * lc;
  }
  const int base = chunk << 4, j = h.y;
  FitAcc t;
  {
    const int n = (int)(cnt_a & 0xff), sc = (int)((cnt_a >> 8) & 0xff), sc2 = (int)(cnt_a >> 16);
    const int sq = (int)(qs_a & 0x3fff), scq = (int)(qs_a >> 14);
    t.an = n; t.xa = base * n + sc; t.x2a = base * base * n + 2 * base * sc + sc2; t.ya = sq; t.xya = base * sq + scq;
  }
  {
    const int n = (int)(cnt_all & 0xff), sc = (int)((cnt_all >> 8) & 0xff), sc2 = (int)(cnt_all >> 16);
    const int sq = (int)(qs_all & 0x3fff), scq = (int)(qs_all >> 14);
    t.bn = n - t.an; t.xb = base * n + sc - t.xa; t.x2b = base * base * n + 2 * base * sc + sc2 - t.x2a;
    t.yb = sq - t.ya; t.xyb = base * sq + scq - t.xya;
  }
  FitAcc *dst = acc + j;
  if (t.an) {
    lds_atomic_add(&dst->xa, t.xa); lds_atomic_add(&dst->ya, t.ya); lds_atomic_add(&dst->x2a, t.x2a);
    lds_atomic_add(&dst->xya, t.xya); lds_atomic_add(&dst->an, t.an);
  }
  if (t.bn) {
    lds_atomic_add(&dst->xb, t.xb); lds_atomic_add(&dst->yb, t.yb); lds_atomic_add(&dst->x2b, t.x2b);
    lds_atomic_add(&dst->xyb, t.xyb); lds_atomic_add(&dst->bn, t.bn);
  }
  return t.an;
}

// fit_line, lib/floor1.c:456-514.  a[0..fits) are consecutive intervals whose
// outer x range is [x0, x1] (sorted_index of the first / one past the last).
VAMD_DEV FitTerm fit_term(const FitAcc &a, float twofitweight) {
  const double weight = (double)((float)(a.bn + a.an) * twofitweight / (float)(a.an + 1)) + 1.;
  FitTerm t;
  t.xb = a.xb + a.xa * weight;
  t.yb = a.yb + a.ya * weight;
  t.x2b = a.x2b + a.x2a * weight;
  t.xyb = a.xyb + a.xya * weight;
  t.bn = a.bn + a.an * weight;
  return t;  // (the reference also sums y2b, which nothing reads)
}


// The line walk of inspect_error / render_line0 (lib/floor1.c:516-527,923-946) in closed form.  The reference steps
// y by base = dy / adx and by one more whenever the running remainder of ady' = |dy| - |base| * adx overflows adx:
// after k steps y = y0 + k * base + sgn * floor(k * ady' / adx) = y0 + sgn * floor(k * |dy| / adx), since
// k * |base| is whole and |dy| = |base| * adx + ady'.
struct LineStep {
  int ady, sgn;        // |dy|, sign of dy
  unsigned int magic;  // div_magic()'s multiplier for adx
};
VAMD_DEV LineStep line_step(int x0, int x1, int y0, int y1, const unsigned int *magic) {
  LineStep s;
  const int dy = y1 - y0;
  s.ady = dy < 0 ? -dy : dy;
  s.sgn = dy < 0 ? -1 : 1;
  s.magic = magic[x1 - x0];
  return s;
}
VAMD_DEV int line_y(const LineStep &s, int y0, int k) { return y0 + s.sgn * div_magic(mad24(k, s.ady, 0), s.magic); }

// inspect_error, lib/floor1.c:516-565, wave-parallel over x in [x0, x1)
VAMD_DEV int inspect_error_wave(int x0, int x1, int y0, int y1, const unsigned short *qc, const FloorP &F) {
  LineStep s;  // (line_step with wave-uniform operands)
  s.ady = y1 < y0 ? y0 - y1 : y1 - y0;
  s.sgn = y1 < y0 ? -1 : 1;
  s.magic = load_uniform_u32(F.div_magic, x1 - x0);
  const int cnt = (x1 - x0) > 1 ? (x1 - x0) : 1;  // points visited: x0, then x0+1 .. x1-1
  int mse = 0;
  bool bad = false;
  if (F.int_tests) {
    // The two tests in integers (floor_derive_tests), as one range check on d = val - y: the point is bad when d is
    // outside (-under_i, over_i).  They apply to a point of class a (bit 15 of qc: mdct + twofitatten >= mask) that
    // is non-zero, and to the first point even when it is zero (lib/floor1.c:536-539): "qc > 0x8000", with bit 0
    // forced for the first point.
    const int bias = F.under_i - 1;
    const unsigned int span = (unsigned int)(F.over_i + F.under_i - 1);
    unsigned int first = LANE == 0 ? 1u : 0u;
    int nsgn = -s.sgn;
    keep_opaque(nsgn);  // (a known +-1 would turn the multiply-add into negate + select)
    for (int k = LANE; k < cnt; k += NLANES) {
      const int q = div_magic(mad24(k, s.ady, 0), s.magic);
      const unsigned int qv = qc[x0 + k];
      const int d = mad24(q, nsgn, (int)(qv & 0x7fffu) - y0);  // val - y
      mse = mad24(d, d, mse);
      bad = bad || (((qv | first) > 0x8000u) && ((unsigned int)(d + bias) >= span));
      first = 0;
    }
  } else {
    WAVE_FOR(k, cnt) {
      const int x = x0 + k;
      const int y = line_y(s, y0, k);
      const int qv = qc[x];
      const int val = qv & 0x7fff;
      mse += (y - val) * (y - val);
      if ((qv & 0x8000) && (k == 0 || val)) {
        if ((float)y + F.maxover < (float)val) bad = true;
        if ((float)y - F.maxunder > (float)val) bad = true;
      }
    }
  }
  if (wave_any(bad)) return 1;
  // maxover^2 / cnt > maxerr, maxunder^2 / cnt > maxerr (:556-557), as thresholds on cnt (floor_derive_tests)
  if (cnt <= F.cnt_over || cnt <= F.cnt_under) return 0;
  mse = wave_sum(mse);
  // (float)(mse / cnt) > maxerr, without the integer divide: the quotient q is an integer, so for
  // maxerr >= 0 the test is q >= floor(maxerr) + 1, i.e. mse >= (floor(maxerr) + 1) * cnt.  (q >= 2^24,
  // where the float conversion would round, is far above any maxerr and true on both sides.)
  if (F.maxerr >= 0.f && F.maxerr < 1048576.f) return mse >= ((int)F.maxerr + 1) * cnt;
  if ((float)(mse / cnt) > F.maxerr) return 1;
  return 0;
}

VAMD_DEV int post_Y(const LaneInts &A, const LaneInts &B, int pos) {
  const int a = A.get(pos), b = B.get(pos);
  if (a < 0) return b;
  if (b < 0) return a;
  return (a + b) >> 1;
}

// render_point, lib/floor1.c:257-271
//   k = x - x0, magic = div_magic()'s multiplier for x1 - x0: fixed per post by the look (PostSteps below)
VAMD_DEV int render_point(int y0, int y1, int k, unsigned int magic) {
  y0 &= 0x7fff;
  y1 &= 0x7fff;
  const int dy = y1 - y0;
  const int ady = dy < 0 ? -dy : dy;
  const int off = div_magic(mad24(ady, k, 0), magic);
  return dy < 0 ? y0 - off : y0 + off;
}

// What render_point needs of a post's place between its two neighbours, one post per lane (posts 0 and 1 have no
// neighbours: zeros)
struct PostSteps {
  LaneInts k, magic;
  VAMD_MEM void load(const FloorP &F, const LaneInts &postlist, const LaneInts &lo2, const LaneInts &hi2) {
    k.fill(0);
    magic.fill(0);
    WAVE_FOR(i, F.posts) {
      const int x0 = postlist.gather(lo2.at(i)), x1 = postlist.gather(hi2.at(i));
      if (i >= 2) {
        k.put(i, postlist.at(i) - x0);
        magic.put(i, (int)F.div_magic[x1 - x0]);
      }
    }
  }
};

// ---- the two wave-parallel pieces of the fit and of the curve.  (The one-lane test build, which has no lanes to
// deal the work to, takes serial forms of the same two functions from tests/emul/k_floor_host.h.)
#if VAMD_GPU
// The two fit_line calls of a split (lib/floor1.c:648-651: left and right of the new post, both
// unconstrained) as ONE pass over the wave: lanes 0-4 sum the five quantities of the left range,
// lanes 8-12 those of the right range, each in interval order out of LDS (one 8-byte read and one
// dependent fp64 add per interval instead of ten v_readlane and five adds), then every lane of a
// group evaluates the closed form once.  Same operand order as fit_line above.
//   term   LDS [intervals][5] doubles: xb, yb, x2b, xyb, bn of each interval (fit_term)
//   sums   LDS [2][8] doubles scratch
VAMD_DEV void fit_line_pair(const double *term, double *sums, int firstL, int fitsL, int x0L, int x1L, int firstR,
                            int fitsR, int x0R, int x1R, int *ret0, int *ly0, int *ly1, int *ret1, int *hy0,
                            int *hy1) {
  const int grp = (LANE >> 3) & 1, q = LANE & 7;
  const int first = grp ? firstR : firstL, fits = grp ? fitsR : fitsL;
  const int most = fitsL > fitsR ? fitsL : fitsR;
  double acc = 0.;
  if (LANE < 16 && q < 5) {
    const double *t = term + first * 5 + q;
    for (int i = 0; i < most; i++)
      if (i < fits) acc += t[i * 5];
    sums[grp * 8 + q] = acc;
  }
  WAVE_SYNC();
  const double xb = sums[grp * 8], yb = sums[grp * 8 + 1], x2b = sums[grp * 8 + 2], xyb = sums[grp * 8 + 3],
               bn = sums[grp * 8 + 4];
  const int x0 = grp ? x0R : x0L, x1 = grp ? x1R : x1L;
  const double denom = (bn * x2b - xb * xb);
  int r = 1, y0 = 0, y1 = 0;
  if (denom > 0.) {
    const double aa = (yb * x2b - xyb * xb) / denom;
    const double bb = (bn * xyb - xb * yb) / denom;
    y0 = (int)rint(aa + bb * x0);
    y1 = (int)rint(aa + bb * x1);
    if (y0 > 1023) y0 = 1023;
    if (y1 > 1023) y1 = 1023;
    if (y0 < 0) y0 = 0;
    if (y1 < 0) y1 = 0;
    r = 0;
  }
  *ret0 = __builtin_amdgcn_readlane(r, 0);
  *ly0 = __builtin_amdgcn_readlane(y0, 0);
  *ly1 = __builtin_amdgcn_readlane(y1, 0);
  *ret1 = __builtin_amdgcn_readlane(r, 8);
  *hy0 = __builtin_amdgcn_readlane(y0, 8);
  *hy1 = __builtin_amdgcn_readlane(y1, 8);
  WAVE_SYNC();  // sums[] is rewritten by the next split
}

// The integer curve of floor1_encode / render_line0 (lib/floor1.c:923-946), from the quantised posts.
VAMD_DEV void floor_render_curve(const FloorP &F, int posts, int n2, const LaneInts &forward_index, const LaneInts &post,
                                 const LaneInts &postlist, FloorScratch *sc, ilog_t *__restrict__ ilogmask, PhaseClock &pc) {
  // Lane j looks at the j-th post in x order.  The curve over [x_j, x_j+1) is the line from the last USED post at or
  // before j to the first used one after it (render_line0, lib/floor1.c:923-946; held flat past the last used post,
  // :941-943): both are bit scans of the ballot of used posts, their x / y come over from those lanes, and lane j
  // leaves the line's constants in row j.  A bin then needs no search at all: bin_interval[x] (static) IS its j.
  // The rows overlay the fit's accumulators, which are dead by now.
  struct SegRow {
    int x0, y0, ady, sgn;
    unsigned int magic;
    int pad[3];
  };
  SegRow *rows = (SegRow *)sc->acc;
  {
    const int j = LANE;
    const int cur = forward_index.at(j);
    const int src = j < posts ? cur : 0;
    const int pv = post.gather(src), px = postlist.gather(src);  // (gathers need every lane active)
    const bool used = j < posts && (j == 0 || (pv & 0x8000) == 0);
    const unsigned long long um = __ballot(used);
    const int myx = j == 0 ? 0 : px, myy = (pv & 0x7fff) * F.mult;
    const unsigned long long upto = j >= 63 ? ~0ull : ((2ull << j) - 1ull);
    const int sidx = 63 - __builtin_clzll(um & upto);  // (bit 0 is always set)
    const unsigned long long above = um & ~upto;
    const int eidx = above ? __builtin_ctzll(above) : sidx;
    const int xs = __shfl(myx, sidx, 64), ys = __shfl(myy, sidx, 64);
    const int xe = __shfl(myx, eidx, 64), ye = __shfl(myy, eidx, 64);
    if (j < posts) {
      SegRow r;
      r.x0 = xs, r.y0 = ys;
      r.ady = 0, r.sgn = 1, r.magic = 0, r.pad[0] = r.pad[1] = r.pad[2] = 0;
      if (above) {
        const LineStep st = line_step(xs, xe, ys, ye, F.div_magic);
        r.ady = st.ady, r.sgn = st.sgn, r.magic = st.magic;
      }
      rows[j] = r;
    }
  }
  WAVE_SYNC();
  pc.mark(3);
  if (ilogmask) {
    WAVE_FOR(q, n2 >> 2) {
      const unsigned int jq = ((const unsigned int *)F.bin_interval)[q];
      int v[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int jb = (int)((jq >> (8 * c)) & 0xff);
        const SegRow r = rows[jb == 255 ? posts - 1 : (jb & 0x7f)];
        const int k = 4 * q + c - r.x0;
        v[c] = mad24(div_magic(mad24(k, r.ady, 0), r.magic), r.sgn, r.y0);
      }
      ((unsigned int *)ilogmask)[q] = (unsigned)v[0] | ((unsigned)v[1] << 8) | ((unsigned)v[2] << 16) | ((unsigned)v[3] << 24);  // (ilog_t)
    }
  }
}
#else
#include "k_floor_host.h"
#endif

// floor1_fit for one channel-block (lib/floor1.c:576-729).
//   qc    LDS [n2]   quantised mask + class bit, see offset_and_mix_wave
//   outp  floor1_fit's return, one post per lane (bit 15 = unused flag); untouched when it returns 0
// Returns 1, or 0 where the reference returns NULL (nothing above the fit's floor).
VAMD_DEV int floor_fit_posts(const FloorP &F, const unsigned short *qc, FloorScratch *sc, LaneInts &outp,
                             PhaseClock &pc) {
  const int posts = F.posts, n = F.look_n;

  LaneInts postlist, sorted_index, reverse_index, hineighbor, loneighbor;
  postlist.load(F.postlist, posts);
  sorted_index.load(F.sorted_index, posts);
  reverse_index.load(F.reverse_index, posts);
  hineighbor.load(F.hineighbor, posts);
  loneighbor.load(F.loneighbor, posts);
  LaneInts fitA, fitB, lon, hin, memo;
  fitA.fill(-200);
  fitB.fill(-200);
  lon.fill(0);
  hin.fill(1);
  memo.fill(-1);
  outp.fill(0);
  WAVE_FOR(i, (posts - 1) * 10)((int *)sc->acc)[i] = 0;
  WAVE_SYNC();
  // accumulate_fit for all post intervals at once, a lane per record of the floor's work list (integer adds
  // commute, so the totals equal the reference's sequential sums)
  int nz = 0;
  WAVE_FOR(sg, F.fit_nseg) nz += accumulate_segment(F.fit_segs + VAMD_FITSEG_WORDS * sg, qc, sc->acc);
  nz = wave_sum(nz);
  WAVE_SYNC();
  // lane i forms interval i's fit_line contribution; the terms go to LDS as rows for fit_line_pair, each over its own
  // interval's accumulators (40 bytes either way), which are dead from here on
  WAVE_FOR(i, posts - 1) {
    // (the row overlays the very accumulators it is formed from, ints then doubles through one address: the ints are
    // copied out and a compiler fence stands between their last read and the first double store)
    const FitAcc mine = sc->acc[i];
    const FitTerm ft = fit_term(mine, F.twofitweight);
    WAVE_SYNC();
    double row[5] = {ft.xb, ft.yb, ft.x2b, ft.xyb, ft.bn};
    memcpy((char *)sc->acc + (size_t)i * sizeof(row), row, sizeof(row));
  }
  WAVE_SYNC();
  pc.mark(1);

  if (!nz) return 0;  // floor1_fit returns NULL

  // ---- greedy progressive split, lib/floor1.c:610-698.  Wave-uniform: every lane
  // walks the same decisions; the state is in lane registers.
  {
    int y0 = -200, y1 = -200;
    int r0, r1, u0, u1;  // the whole-range fit rides in the left half of the pair routine
    fit_line_pair((const double *)sc->acc, sc->pair_sums, 0, posts - 1, sorted_index.get(0), sorted_index.get(posts - 1),
                  0, 0, 0, 0, &r0, &y0, &y1, &r1, &u0, &u1);
    fitA.set(0, y0);
    fitB.set(0, y0);
    fitB.set(1, y1);
    fitA.set(1, y1);
  }
  for (int i = 2; i < posts; i++) {
    const int sortpos = reverse_index.get(i);
    const int ln = lon.get(sortpos);
    const int hn = hin.get(sortpos);
    if (memo.get(ln) != hn) {
      const int lsortpos = reverse_index.get(ln);
      const int hsortpos = reverse_index.get(hn);
      memo.set(ln, hn);
      const int lx = postlist.get(ln), hx = postlist.get(hn);
      const int ly = post_Y(fitA, fitB, ln);
      const int hy = post_Y(fitA, fitB, hn);
      // (ly == -1 || hy == -1 => exit(1) in the reference: unreachable, fits are >= 0 or -200)
      if (inspect_error_wave(lx, hx, ly, hy, qc, F)) {
        int ly0 = -200, ly1 = -200, hy0 = -200, hy1 = -200;
        int ret0, ret1;
        fit_line_pair((const double *)sc->acc, sc->pair_sums, lsortpos, sortpos - lsortpos, sorted_index.get(lsortpos),
                      sorted_index.get(sortpos), sortpos, hsortpos - sortpos, sorted_index.get(sortpos),
                      sorted_index.get(hsortpos), &ret0, &ly0, &ly1, &ret1, &hy0, &hy1);
        if (ret0) {
          ly0 = ly;
          ly1 = hy0;
        }
        if (ret1) {
          hy0 = ly1;
          hy1 = hy;
        }
        if (ret0 && ret1) {
          fitA.set(i, -200);
          fitB.set(i, -200);
        } else {
          fitB.set(ln, ly0);
          if (ln == 0) fitA.set(ln, ly0);
          fitA.set(i, ly1);
          fitB.set(i, hy0);
          fitA.set(hn, hy1);
          if (hn == 1) fitB.set(hn, hy1);
          if (ly1 >= 0 || hy0 >= 0) {
            hin.replace_run_down(sortpos, hn, i);
            lon.replace_run_up(sortpos + 1, posts, ln, i);
          }
        }
      } else {
        fitA.set(i, -200);
        fitB.set(i, -200);
      }
    }
  }

  pc.mark(2);
  // ---- posts out, lib/floor1.c:700-724.  Post i is settled from its two fixed neighbours, so
  // the list order of the reference can be replaced by dependency levels: one lane per post.
  LaneInts lo2, hi2, level;
  lo2.load_shifted(F.loneighbor, 2, posts);
  hi2.load_shifted(F.hineighbor, 2, posts);
  level.load(F.level, posts);
  PostSteps ps;
  ps.load(F, postlist, lo2, hi2);
  LaneInts fitted;  // the post's own fit (the mean of its two sides), settled before the levels
  fitted.fill(0);
  WAVE_FOR(i, posts) {
    const int a = fitA.at(i), b = fitB.at(i);
    const int vx = a < 0 ? b : (b < 0 ? a : (a + b) >> 1);
    fitted.put(i, vx);
    if (i < 2) outp.put(i, vx);
  }
  for (int L = 1; L <= F.nlevels; L++) {
    WAVE_FOR(i, posts) {
      const int y0 = outp.gather(lo2.at(i)), y1 = outp.gather(hi2.at(i));
      if (i >= 2 && level.at(i) == L) {
        const int predicted = render_point(y0, y1, ps.k.at(i), (unsigned int)ps.magic.at(i));
        const int vx = fitted.at(i);
        outp.put(i, (vx >= 0 && predicted != vx) ? vx : (predicted | 0x8000));
      }
    }
  }
  return 1;
}

// floor1_encode, value half: quantise by mult, predict, settle the "unused" flags
// (lib/floor1.c:766-831).  A post keeps its flag iff it is itself trivial (flagged by the fit, or
// equal to its prediction) and no non-trivial post names it as a neighbour; values by dependency
// level (a post's neighbours always sit on lower levels).
//   outp     the posts as fitted, one per lane
//   post     <- quantised values, bit 15 = unused (then the value is the prediction)
//   wrapped  <- (optional) out[]: what floor1_encode writes for each post -- posts 0/1 verbatim,
//               the others' deviation from the prediction folded into [0, range) (:805-824)
VAMD_DEV void floor_quantise_predict(const FloorP &F, const LaneInts &outp, const LaneInts &postlist, LaneInts &post,
                                     LaneInts *wrapped) {
  const int posts = F.posts;
  LaneInts lo2, hi2, level;
  lo2.load_shifted(F.loneighbor, 2, posts);
  hi2.load_shifted(F.hineighbor, 2, posts);
  level.load(F.level, posts);
  post.fill(0);
  WAVE_FOR(i, posts) {
    const int o = outp.at(i);
    int val = o & 0x7fff;
    switch (F.mult) {
      case 1: val >>= 2; break;
      case 2: val >>= 3; break;
      case 3: val /= 12; break;
      case 4: val >>= 4; break;
    }
    post.put(i, val | (o & 0x8000));
    if (wrapped) wrapped->put(i, i < 2 ? val | (o & 0x8000) : 0);
  }
  PostSteps ps;
  ps.load(F, postlist, lo2, hi2);
  unsigned long long needed = 3ull;  // posts 0 and 1 are always used
  for (int L = 1; L <= F.nlevels; L++) {
    WAVE_FOR(i, posts) {
      const int ln = lo2.at(i), hn = hi2.at(i);
      const int y0 = post.gather(ln), y1 = post.gather(hn);
      if (i >= 2 && level.at(i) == L) {
        const int pi = post.at(i);
        const int predicted = render_point(y0, y1, ps.k.at(i), (unsigned int)ps.magic.at(i));
        if ((pi & 0x8000) || predicted == pi) {
          post.put(i, predicted | 0x8000);
        } else {
          needed |= (1ull << i) | (1ull << ln) | (1ull << hn);
          if (wrapped) {
            const int room = F.quant_q - predicted < predicted ? F.quant_q - predicted : predicted;
            int val = pi - predicted;
            if (val < 0)
              val = val < -room ? room - val - 1 : -1 - (val * 2);
            else
              val = val >= room ? val + room : val << 1;
            wrapped->put(i, val);
          }
        }
      }
    }
  }
  needed = wave_or64(needed);
  WAVE_FOR(i, posts) {
    if ((needed >> i) & 1) post.put(i, post.at(i) & 0x7fff);
  }
}

// The curve half of floor1_encode for one set of posts (lib/floor1.c:766-831,923-952): quantise,
// predict, settle the unused flags, render the integer curve.
//   outp / valid  a floor1_fit result (floor_fit_posts) or an interpolation of two
//   posts_out HBM [VAMD_POSTS_STRIDE] the posts as fitted (what the host hands floor1_encode)
//   ilogmask  HBM [n2]
// Returns floor1_encode's nonzero flag (1 = non-trivial floor).
//   wrapped_out HBM [posts] or null: floor1_encode's out[] (what the packet stage writes for each post), for k_pack
VAMD_DEV int floor_encode_render(const FloorP &F, int n2, const LaneInts &outp, int valid, FloorScratch *sc,
                                 int *__restrict__ posts_out, int *__restrict__ post_valid,
                                 ilog_t *__restrict__ ilogmask, PhaseClock &pc, int *__restrict__ wrapped_out = nullptr) {
  const int posts = F.posts;
  if (!valid) {
    // no fit: floor1_encode writes a zero curve (lib/floor1.c:948-952)
    WAVE_FOR(i, VAMD_POSTS_STRIDE) if (posts_out) posts_out[i] = 0;
    if (post_valid && LANE == 0) *post_valid = 0;
    WAVE_FOR(i, n2) if (ilogmask) ilogmask[i] = 0;
    WAVE_SYNC();
    return 0;
  }
  LaneInts postlist, forward_index, post;
  postlist.load(F.postlist, posts);
  forward_index.load(F.forward_index, posts);
  WAVE_FOR(i, VAMD_POSTS_STRIDE) if (posts_out) posts_out[i] = i < posts ? outp.at(i) : 0;
  if (post_valid && LANE == 0) *post_valid = 1;
  if (wrapped_out) {
    LaneInts wrapped;
    wrapped.fill(0);
    floor_quantise_predict(F, outp, postlist, post, &wrapped);
    WAVE_FOR(i, posts) wrapped_out[i] = wrapped.at(i);
  } else {
    floor_quantise_predict(F, outp, postlist, post, nullptr);
  }

  // ---- render the integer curve, lib/floor1.c:923-946
  floor_render_curve(F, posts, n2, forward_index, post, postlist, sc, ilogmask, pc);
  WAVE_SYNC();
  pc.mark(4);
  return 1;
}

// floor1_fit + the curve half of floor1_encode for one channel-block (the VBR path: one curve)
VAMD_DEV int floor_fit_render_block(const FloorP &F, int n2, const unsigned short *qc, FloorScratch *sc,
                                    int *__restrict__ posts_out, int *__restrict__ post_valid,
                                    ilog_t *__restrict__ ilogmask, PhaseClock &pc, int *__restrict__ wrapped_out = nullptr) {
  LaneInts outp;
  const int valid = floor_fit_posts(F, qc, sc, outp, pc);
  return floor_encode_render(F, n2, outp, valid, sc, posts_out, post_valid, ilogmask, pc, wrapped_out);
}

// floor1_interpolate_fit, lib/floor1.c:731-750, one post per lane
VAMD_DEV void floor_interpolate(const LaneInts &A, int haveA, const LaneInts &B, int haveB, int del, LaneInts &out,
                                int *have) {
  *have = haveA && haveB;
  out.fill(0);
  if (!*have) return;
  WAVE_FOR(i, 64) {
    const int a = A.at(i), b = B.at(i);
    int v = ((65536 - del) * (a & 0x7fff) + del * (b & 0x7fff) + 32768) >> 16;
    if ((a & 0x8000) && (b & 0x8000)) v |= 0x8000;
    out.put(i, v);
  }
}

// A bitrate-managed block's floors for one channel (lib/mapping0.c:499-573 + the floor half of
// :613-646 for every candidate packet k): three fits (the middle one was just prepared in qc by
// offset_and_mix_wave), twelve interpolations, fifteen encode/render passes.
//   posts_out [15][VAMD_POSTS_STRIDE], post_valid [15], ilogmask [15][n2], nonzero [15], each with
//   the given element stride between consecutive k
VAMD_DEV void floor_managed_block(const PsyP &P, const FloorP &F, int n2, const float *__restrict__ noise,
                                  const float *__restrict__ tone, const float *__restrict__ mdct_raw,
                                  unsigned short *qc, FloorScratch *sc, int *__restrict__ posts_out, long posts_stride,
                                  int *__restrict__ post_valid, long valid_stride, ilog_t *__restrict__ ilogmask,
                                  long ilog_stride, int *__restrict__ nonzero, long nz_stride, PhaseClock &pc) {
  const int mid = VAMD_PACKETBLOBS / 2, last = VAMD_PACKETBLOBS - 1;
  LaneInts fmid, flo, fhi;
  fmid.fill(0);
  flo.fill(0);
  fhi.fill(0);
  const int hmid = floor_fit_posts(F, qc, sc, fmid, pc);
  int hlo = 0, hhi = 0;
  if (hmid) {
    WAVE_SYNC();
    mask_quantise_wave(P, 2, noise, tone, mdct_raw, qc, F.twofitatten);
    hhi = floor_fit_posts(F, qc, sc, fhi, pc);
    WAVE_SYNC();
    mask_quantise_wave(P, 0, noise, tone, mdct_raw, qc, F.twofitatten);
    hlo = floor_fit_posts(F, qc, sc, flo, pc);
    WAVE_SYNC();
  }
  for (int k = 0; k < VAMD_PACKETBLOBS; k++) {
    LaneInts cur;
    int have;
    if (k == mid) {
      cur = fmid;
      have = hmid;
    } else if (!hmid) {  // the managed branch is skipped: every other curve stays NULL
      cur.fill(0);
      have = 0;
    } else if (k == 0) {
      cur = flo;
      have = hlo;
    } else if (k == last) {
      cur = fhi;
      have = hhi;
    } else if (k < mid) {
      floor_interpolate(flo, hlo, fmid, hmid, k * 65536 / mid, cur, &have);
    } else {
      floor_interpolate(fmid, hmid, fhi, hhi, (k - mid) * 65536 / mid, cur, &have);
    }
    const int nzf = floor_encode_render(F, n2, cur, have, sc, posts_out + k * posts_stride,
                                        post_valid + k * valid_stride, ilogmask + k * ilog_stride, pc);
    if (LANE == 0) nonzero[k * nz_stride] = nzf;
    WAVE_SYNC();
  }
}

}  // namespace vamd
