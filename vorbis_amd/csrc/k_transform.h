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
// LDS: A[VAMD_XF_A_FLOATS(n)] (PCM -> windowed -> FFT buffer "c"), B[VAMD_XF_B_FLOATS(n)] (MDCT work "w" with its
// padded butterfly half, then FFT buffer "ch"); the FFT's early passes leave their output in padded layouts
// (VAMD_F2_POS / VAMD_F3_POS below), which is what the sizes allow for.
//
// Who runs a block: a TEAM.  Every phase below is a loop over independent items (butterflies, pairs,
// quads) followed by a team-wide sync; the items are dealt round the team's threads.  The kernels run one
// wave per transform (WaveTeam) -- measured round 2: four waves per block (a workgroup-wide team) bring nothing, 2.9
// against 2.3 ms, because the stage is bound by the CU's LDS pipe (92 % busy, half of it bank conflicts) and
// not by any wave's latency -- and the test build runs everything in a single lane.
#pragma once
#include <type_traits>
#include "vamd_wave.h"
#include "vamd_params.h"

namespace vamd {

// the fold of the size-specialised transform reads whole quads, one trip in flight (mdct_forward_wave); GPU only: the
// one-lane test build keeps the plain loop
#ifndef VAMD_XF_FOLD_QUADS
#define VAMD_XF_FOLD_QUADS 0  // measured round 5 on this build: 1.513 against 1.482 ms (profiles/r05_xf_variants.txt); kept as a variant
#endif

#define VAMD_XF_A_FLOATS(n) (((n) + ((n) >> 5) + 4 + 3) & ~3)
#define VAMD_XF_B_FLOATS(n) (((n) + ((n) >> 4) + 4 + 3) & ~3)

struct WaveTeam {  // the 64 lanes of one wavefront (one lane in the test build)
  VAMD_MEM int tid() const { return LANE; }
  VAMD_MEM int size() const { return NLANES; }
  VAMD_MEM void sync() const { WAVE_SYNC(); }
};

// A block of PCM held in registers (thread t of the team owns quads t, t+T, ...): fetched from HBM one
// block ahead of its use so that the load latency hides behind the previous block's
// transforms (persistent kernels), then windowed on its way into LDS.  QPT = quads per thread.
template <int QPT>
struct PcmTile {
  float v[QPT][4];
};

template <int QPT, class Team>
VAMD_DEV void pcm_fetch(PcmTile<QPT> &t, const float *__restrict__ pcm, int n, const Team &tm) {
  TEAM_QUADS(kq, q, n >> 2, QPT, tm) f4_get(((const F4 *)pcm)[q], t.v[kq]);
}

// _vorbis_apply_window, lib/window.c:2102-2135, applied while the tile is written to LDS.
template <int QPT, class Team>
VAMD_DEV void window_store(const XformP &P, int W, int lW, int nW, const PcmTile<QPT> &t, float *A, bool apply_window,
                           const Team &tm) {
  const int n = P.n;
  if (!apply_window) {
    TEAM_QUADS(kq, q, n >> 2, QPT, tm)((F4 *)A)[q] = f4_make(t.v[kq]);
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
  TEAM_QUADS(kq, q, n >> 2, QPT, tm) {
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

// cPI*_8 of lib/mdct.h:43-45
#define VAMD_C1 .92387953251128675613F
#define VAMD_C2 .70710678118654752441F
#define VAMD_C3 .38268343236508977175F

// The last three butterfly levels of mdct_butterflies (lib/mdct.c:93-213: the 32-, 16- and 8-point
// networks the reference finishes each group of 32 with), stated as what they are:
//   level 32: eight pair-operations per group, each on the pairs (x[2a], x[2a+1]) and (x[16+2a], x[17+2a]);
//   level 16: four per 16-point half, on (x[2b], x[2b+1]) and (x[8+2b], x[9+2b]);
//   level  8: per 8 points a two-layer add/subtract network.
// In each pair-operation the upper pair takes the sums, the lower pair a rotation of the differences by a
// multiple of pi/8 -- the reference spells each of the eight (four) out with its own constants and signs, and
// the expression trees are kept: which differences, which products, and whether a sum is multiplied or
// products are summed.  (Negating an operand or swapping the operands of an add changes no bit.)
// x = the padded vector (VAMD_PW); g = group of 32.
struct PairOp {
  F2 lo, hi;
};
VAMD_DEV PairOp bfly_level32(int a, F2 lo, F2 hi) {
  PairOp r;
  r.hi.x = hi.x + lo.x;
  r.hi.y = hi.y + lo.y;
  // differences as the reference takes them: upper minus lower for a >= 3 (the y of a == 3 the other way),
  // lower minus upper for a <= 2
  const float dx = a >= 3 ? hi.x - lo.x : lo.x - hi.x;
  const float dy = (a >= 4) ? hi.y - lo.y : lo.y - hi.y;
  switch (a) {
    case 7: r.lo.x = dx; r.lo.y = dy; break;
    case 6: r.lo.x = dx * VAMD_C1 - dy * VAMD_C3; r.lo.y = dx * VAMD_C3 + dy * VAMD_C1; break;
    case 5: r.lo.x = (dx - dy) * VAMD_C2; r.lo.y = (dx + dy) * VAMD_C2; break;
    case 4: r.lo.x = dx * VAMD_C3 - dy * VAMD_C1; r.lo.y = dy * VAMD_C3 + dx * VAMD_C1; break;
    case 3: r.lo.x = dy; r.lo.y = dx; break;
    case 2: r.lo.x = dy * VAMD_C1 + dx * VAMD_C3; r.lo.y = dy * VAMD_C3 - dx * VAMD_C1; break;
    case 1: r.lo.x = (dy + dx) * VAMD_C2; r.lo.y = (dy - dx) * VAMD_C2; break;
    default: r.lo.x = dy * VAMD_C3 + dx * VAMD_C1; r.lo.y = dy * VAMD_C1 - dx * VAMD_C3; break;
  }
  return r;
}
VAMD_DEV PairOp bfly_level16(int b, F2 lo, F2 hi) {
  PairOp r;
  r.hi.x = hi.x + lo.x;
  r.hi.y = hi.y + lo.y;
  if (b == 0) {
    const float d0 = lo.y - hi.y, d1 = lo.x - hi.x;
    r.lo.x = (d0 + d1) * VAMD_C2;
    r.lo.y = (d0 - d1) * VAMD_C2;
  } else if (b == 1) {
    r.lo.x = lo.y - hi.y;
    r.lo.y = hi.x - lo.x;
  } else if (b == 2) {
    const float d0 = hi.x - lo.x, d1 = hi.y - lo.y;
    r.lo.x = (d0 - d1) * VAMD_C2;
    r.lo.y = (d0 + d1) * VAMD_C2;
  } else {
    r.lo.x = hi.x - lo.x;
    r.lo.y = hi.y - lo.y;
  }
  return r;
}
// eight points e[0..7] in place: first the four sums and four differences of the points four apart, then one
// more add/subtract between them
VAMD_DEV void bfly_level8(float *e) {
  const float s0 = e[6] + e[2], t0 = e[6] - e[2], s1 = e[4] + e[0], t1 = e[4] - e[0];
  const float t2 = e[5] - e[1], t3 = e[7] - e[3], s2 = e[5] + e[1], s3 = e[7] + e[3];
  e[6] = s0 + s1;
  e[4] = s0 - s1;
  e[0] = t0 + t2;
  e[2] = t0 - t2;
  e[3] = t3 + t1;
  e[1] = t3 - t1;
  e[7] = s3 + s2;
  e[5] = s3 - s2;
}

// The butterfly work vector lives in LDS with two floats of padding after every 32:
// logical index p sits at PW(p).  Pairs (even p, p+1) stay adjacent and 8-byte
// aligned, and the 32-point groups that one lane each pull into registers start 34
// floats apart, which spreads the 64 lanes over all LDS banks (at a plain stride of
// 32 every lane would hit the same bank).
#define VAMD_PW(p) ((p) + (((p) >> 5) << 1))
#define VAMD_PW_SIZE(n2) ((n2) + ((n2) >> 4))

// The trig pairs of the generic butterfly stages s = 1 .. log2n-7 (stage s reads trig[(4 << s) q], q < n/8 >> s),
// stage after stage: n/8 - 16 pairs, stage s starting at pair n/8 - (n/8 >> (s-1)).  Thread `first` of `step`.
#define VAMD_TPACK_FLOATS(n) ((n) / 4 - 32)
VAMD_DEV void mdct_tpack_fill(float *tpack, const float *__restrict__ trig, int n, int first, int step) {
  for (int i = first; i < n / 8 - 16; i += step) {
    int s = 1, q = i, cnt = n / 16;
    while (q >= cnt) {
      q -= cnt;
      cnt >>= 1;
      s++;
    }
    tpack[2 * i] = trig[(4 << s) * q];
    tpack[2 * i + 1] = trig[(4 << s) * q + 1];
  }
}

// mdct_forward, lib/mdct.c:492-562.  `in` = A (windowed block, LDS, n floats);
// w = work buffer: w[0..n2) plain + padded butterfly vector at w + n2
// (VAMD_PW_SIZE(n2) floats).  The n/2 spectrum is written to out_lds[0..n2), which may be w itself (the plain half is
// not used otherwise) but must not overlap the padded half.
// LOGS > 0 runs 2^LOGS independent transforms of the same size side by side (transform t at
// in + t*in_stride, w + t*w_stride, out_lds + t*out_stride): every loop then ranges over
// (transform, item) so that small transforms -- the 128-point one of the block-switching
// detector has only two 32-point groups -- still fill the wave.
// LOGN > 0 fixes the transform size at compile time (n = 2^LOGN): loop counts, strides and every index
// expression derived from n then fold into constants and immediate offsets -- the stage is bound by
// instruction issue, and a good part of its instructions is address arithmetic.  0 = take n from P.
// PACKED: P.tpack holds the generic butterfly stages' trig pairs one stage after the other, each at stride 1
// (mdct_tpack_fill) -- read out of the plain table at its stride of 4 << s floats, the 32 lanes of a load share
// one or two LDS banks from the third stage on (16-way conflicts: a tenth of the stage's LDS time) -- and the
// bit-reverse indices are computed, not fetched (P.bitrev_std).
// The fold's operands of one lane for a block read straight out of HBM (k_mdct_only): x0.z, x0.x, x1.y, x1.w of the
// reference's quads (lib/mdct.c:506-544) for each of the lane's trips over the n/4 pairs.  fold_fetch only issues the
// loads: all of a frame's thirty-two words are in flight before the first is used.  (Measured round 4 and not kept:
// holding the NEXT frame's operands across the butterflies -- 162 registers, twelve waves per CU instead of sixteen:
// 312 against 328 M frames/s at 65 536 frames, 355 against 360 at 262 144; whole quads instead of the two words a pair
// needs of each: no change.  The kernel sits at 4.0-4.4 TB/s, 65-70 % of what a copy reaches.)
template <int LOGN>
struct FoldOps {
  static constexpr int NT = LOGN >= 8 ? (1 << LOGN) / 4 / 64 : 1;
  float xa[NT], xb[NT], ya[NT], yb[NT];
};
template <int LOGN>
VAMD_DEV void fold_fetch(FoldOps<LOGN> &o, const float *in) {
  constexpr int n = 1 << LOGN, n2 = n >> 1, n4 = n >> 2, n8 = n >> 3;
#pragma unroll
  for (int k = 0; k < FoldOps<LOGN>::NT; k++) {
    const int p = LANE + 64 * k;
    const float *q0, *q1;
    if (2 * (64 * k) < n8) {
      q0 = in + n2 + n4 - 4 * (p + 1), q1 = in + n2 + n4 + 4 * p;
    } else if (2 * (64 * k) < n2 - n8) {
      q0 = in + n2 + n4 - 4 * (p + 1), q1 = in + 4 * (p - n8 / 2);
    } else {
      q0 = in + n - 4 * (p - (n2 - n8) / 2 + 1), q1 = in + 4 * (p - n8 / 2);
    }
    o.xa[k] = q0[2], o.xb[k] = q0[0], o.ya[k] = q1[1], o.yb[k] = q1[3];
  }
}

// FOLD_AHEAD: the block is read straight out of HBM (k_mdct_only) -- every fold operand of the lane is fetched before the
// first is used.  Written as one loop the fold waits for its four words eight times over (n = 2048: a frame spent 10 of
// its 13 us there, and the kernel sat at the bytes its sixteen waves per CU keep in flight: 3.8 TB/s); the three regimes
// change at the pairs n/16 and 3n/16, multiples of a wave's stride of 64 for n >= 1024, so which quarters a trip folds is
// known when it is compiled.  (Round 5: the form used to be taken for n = 512 too, where the regimes change in the
// middle of a trip -- found by the whole-quad variant below failing the 22 kHz short blocks; no shipped path reached
// k_mdct_only at 512, and tests/test_gpu_parity.py::test_mdct_forward_every_size now does.)
// ---- round 6: the head of the 2048-sample transform without LDS ------------------------------
// The stage is bound by the CU's LDS pipe; of the MDCT's six trips through it, two move data that never needs to leave
// the wave's registers:
//   * fold -> first butterfly trip.  The fold's item p = LANE + 64 k produces the pair (w2[2p], w2[2p+1]); the trip that
//     fuses stages 0-2 has its item q gather the eight pairs 64 k + 63 - q, k = 0..7.  Run item q = 63 - LANE and those
//     are the very pairs the lane has just formed: nothing to exchange.
//   * first trip -> second trip (stages 3, 4).  Item (j, q) of the second trip -- sub-block j of 64 pairs, q < 16 --
//     wants the pairs 64 j + (15 - q) + 16 m, m = 0..3: register E[j] of the four lanes (15 - q) + 16 m.  Give lane L the
//     items with 15 - q = L & 15 and j = (L >> 4) + 4 i, i = 0, 1: what it needs sits in its own COLUMN of sixteenth-rows,
//     and the exchange is a 4 x 4 transpose between the lane's bits 5:4 and the register index's bits 1:0 -- two rounds of
//     gfx950's half-exchanges, v_permlane32_swap (lane bit 5 <-> register bit 1) and v_permlane16_swap (lane bit 4 <->
//     register bit 0), sixteen VALU instructions on a vector unit that is half idle in this stage, in place of eight
//     64-bit LDS stores and eight loads per lane on the pipe that binds it.
// Same butterflies on the same operands in the same order (the expression trees are the reference's, lib/mdct.c:216-336);
// only who holds a pair changes.  The second trip stores to LDS as before: the 32-point groups want sixteen consecutive
// pairs in one lane, a transpose inside rows of sixteen lanes, which the DPP can only do a bit at a time.
#ifndef VAMD_XF_HEAD_REGS
#define VAMD_XF_HEAD_REGS 1
#endif
#if VAMD_GPU
VAMD_DEV void swap_halves32(float &a, float &b) {  // a: [a.lo, b.lo], b: [a.hi, b.hi]  (lo = lanes 0-31)
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
VAMD_DEV void swap_rows16(float &a, float &b) {  // a's odd rows of sixteen lanes <-> b's even rows
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
#endif

// WIN: `in` holds the samples as they came, and the fold multiplies each by its window value `win[i]` on the way in (the
// detector, k_env_spectrum: its steps overlap by half, so transform t's samples start in_stride = 64 floats after its
// predecessor's and a windowed copy of every step would be twice the LDS).  x * win[i] is rounded as the separate
// windowing pass rounds it; the sums the fold takes of such products are the same sums.
template <int LOGS = 0, int LOGN = 0, class Team = WaveTeam, bool PACKED = false, bool FOLD_AHEAD = false, bool WIN = false>
VAMD_DEV void mdct_forward_wave(const XformP &P, const float *in0, float *w0, float *out0, PhaseClock &pc,
                                int in_stride = 0, int w_stride = 0, int out_stride = 0, const Team &tm = Team(),
                                FoldOps<LOGN> *ahead = nullptr, const float *in_next = nullptr, const float *win = nullptr) {
  static_assert(!WIN || (!FOLD_AHEAD && !(VAMD_XF_FOLD_QUADS && LOGN >= 10 && LOGS == 0)), "the window in the fold: the generic fold only");
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

  // one butterfly: the upper pair takes the sum, the lower the difference rotated by T (lib/mdct.c:222-257)
  auto bfly = [](F2 &a, F2 &b, const F2 T) {
    const float r0 = a.x - b.x, r1 = a.y - b.y;
    a.x += b.x;
    a.y += b.y;
    b.x = r1 * T.y + r0 * T.x;
    b.y = r1 * T.x - r0 * T.y;
  };
  auto stage_trig = [&](int s, int q) {
    return PACKED && s > 0 ? *(const F2 *)(P.tpack + (n4 - (n4 >> (s - 1))) + 2 * q) : *(const F2 *)(trig + (4 << s) * q);
  };
  int s = 0;
#if VAMD_GPU
  constexpr bool head_regs = VAMD_XF_HEAD_REGS && LOGN == 11 && LOGS == 0 && std::is_same<Team, WaveTeam>::value;
#else
  constexpr bool head_regs = false;
#endif
  if constexpr (head_regs) {
#if VAMD_GPU
    float *w2 = w0 + n2;
    F2 E[8];  // pair 64 k + LANE
    {
      FoldOps<LOGN> mine;
      if constexpr (FOLD_AHEAD) {
        if (!ahead) {
          fold_fetch<LOGN>(mine, in0);
          ahead = &mine;
        }
      } else {
        fold_fetch<LOGN>(mine, in0);  // (out of LDS: the windowed block)
        ahead = &mine;
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int p = LANE + 64 * k;
        const F2 T = *(const F2 *)(trig + n2 - 2 * (p + 1));
        float r0, r1;
        if (2 * (64 * k) < n8) {
          r0 = ahead->xa[k] + ahead->ya[k], r1 = ahead->xb[k] + ahead->yb[k];
        } else if (2 * (64 * k) < n2 - n8) {
          r0 = ahead->xa[k] - ahead->ya[k], r1 = ahead->xb[k] - ahead->yb[k];
        } else {
          r0 = -ahead->xa[k] - ahead->ya[k], r1 = -ahead->xb[k] - ahead->yb[k];
        }
        E[k].x = r1 * T.y + r0 * T.x;
        E[k].y = r1 * T.x - r0 * T.y;
      }
      if constexpr (FOLD_AHEAD) {
        if (in_next) fold_fetch<LOGN>(*ahead, in_next);
      }
    }
    pc.mark(1);
    {  // stages 0-2 on the lane's own eight pairs: item q = 63 - LANE of the three-stage trip below (pts = n2, j = 0)
      const int q = 63 - LANE, h = n2 >> 4;
      bfly(E[7], E[3], stage_trig(0, q));
      bfly(E[6], E[2], stage_trig(0, q + h));
      bfly(E[5], E[1], stage_trig(0, q + 2 * h));
      bfly(E[4], E[0], stage_trig(0, q + 3 * h));
      const F2 T1a = stage_trig(1, q), T1b = stage_trig(1, q + h);
      bfly(E[7], E[5], T1a);
      bfly(E[6], E[4], T1b);
      bfly(E[3], E[1], T1a);
      bfly(E[2], E[0], T1b);
      const F2 T2 = stage_trig(2, q);
      bfly(E[7], E[6], T2);
      bfly(E[5], E[4], T2);
      bfly(E[3], E[2], T2);
      bfly(E[1], E[0], T2);
    }
    // the 4 x 4 transpose: lane bits 5:4 <-> register bits 1:0 (register bit 2 = which of the lane's two items)
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
      for (int jl = 0; jl < 2; jl++) {
        swap_halves32(E[4 * i + jl].x, E[4 * i + jl + 2].x);
        swap_halves32(E[4 * i + jl].y, E[4 * i + jl + 2].y);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
      for (int jh = 0; jh < 2; jh++) {
        swap_rows16(E[4 * i + 2 * jh].x, E[4 * i + 2 * jh + 1].x);
        swap_rows16(E[4 * i + 2 * jh].y, E[4 * i + 2 * jh + 1].y);
      }
    }
    {  // stages 3, 4: the lane's two items (j = (LANE >> 4) + 4 i, q = 15 - (LANE & 15)); E[4 i + m] = pair 64 j + (15 - q) + 16 m
      constexpr int pts = n2 >> 3;
      const int q = 15 - (LANE & 15);
      const F2 Tab = stage_trig(3, q), Tcd = stage_trig(3, q + (pts >> 3)), T1 = stage_trig(4, q);
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int j = (LANE >> 4) + 4 * i;
        F2 &A = E[4 * i + 3], &B = E[4 * i + 1], &C = E[4 * i + 2], &D = E[4 * i + 0];
        bfly(A, B, Tab);
        bfly(C, D, Tcd);
        bfly(A, C, T1);
        bfly(B, D, T1);
        const int base = pts * j - 2 - 2 * q;
        *(F2 *)(w2 + VAMD_PW(base + pts)) = A;
        *(F2 *)(w2 + VAMD_PW(base + (pts >> 1))) = B;
        *(F2 *)(w2 + VAMD_PW(base + 3 * (pts >> 2))) = C;
        *(F2 *)(w2 + VAMD_PW(base + (pts >> 2))) = D;
      }
    }
    tm.sync();
    s = 5;  // == nstages for n = 2048: the loops below have nothing left
#endif
  } else {
  // fold + pre-twiddle ("window + rotate + step 1"), lib/mdct.c:506-544.
  // Pair p writes w2[2p], w2[2p+1]; the three loops differ in which input
  // quarter is folded with which sign.  x0[0],x0[2] / x1[0],x1[2] of the reference
  // are the .x,.z / .y,.w lanes of two aligned quads of the input.
  if constexpr (FOLD_AHEAD && LOGN >= 10 && LOGN <= 11 && LOGS == 0) {  // (n = 4096: 64 operands per lane would spill; n = 512: below)
    // every operand of the lane first, then the arithmetic (`ahead`: a caller that fetched them earlier; `in_next`: the
    // block whose operands it wants requested as soon as these are consumed)
    constexpr int NT = FoldOps<LOGN>::NT;
    float *w2 = w0 + n2;
    FoldOps<LOGN> mine;
    if (!ahead) {
      fold_fetch<LOGN>(mine, in0);
      ahead = &mine;
    }
#pragma unroll
    for (int k = 0; k < NT; k++) {
      const int p = LANE + 64 * k;
      const F2 T = *(const F2 *)(trig + n2 - 2 * (p + 1));
      float r0, r1;
      if (2 * (64 * k) < n8) {
        r0 = ahead->xa[k] + ahead->ya[k], r1 = ahead->xb[k] + ahead->yb[k];
      } else if (2 * (64 * k) < n2 - n8) {
        r0 = ahead->xa[k] - ahead->ya[k], r1 = ahead->xb[k] - ahead->yb[k];
      } else {
        r0 = -ahead->xa[k] - ahead->ya[k], r1 = -ahead->xb[k] - ahead->yb[k];
      }
      F2 o;
      o.x = r1 * T.y + r0 * T.x;
      o.y = r1 * T.x - r0 * T.y;
      *(F2 *)(w2 + VAMD_PW(2 * p)) = o;
    }
    if (in_next) fold_fetch<LOGN>(*ahead, in_next);
  } else if constexpr (VAMD_XF_FOLD_QUADS && LOGN >= 10 && LOGS == 0) {  // (n = 512: the regimes change inside a wave's trip)
    // The same fold out of LDS with WHOLE quads (round 5; measured round 4, profiles/r04_xf_phases.txt): a pair needs two
    // words of each of its two quads, and the compiler narrows such a read to ds_read2_b32 -- lanes 16 bytes apart on a
    // 4-byte access are a four-way bank conflict (two thirds of the fold's LDS cycles).  All four words through an
    // opaque use make it a ds_read_b128, which the LDS serves 16 lanes at a time from all banks; one trip in flight
    // (four would want 32 registers the kernel does not have: 7 spills, +2 %), the regimes known per trip as in the
    // FOLD_AHEAD form.  Conflicts 30.6 -> 23.1 % of the kernel's LDS cycles; time 1.482 -> 1.513 ms (round 5) -- not the default.
    constexpr int NT = (1 << LOGN) / 4 / 64;
    float *w2 = w0 + n2;
#pragma unroll 1
    for (int k = 0; k < NT; k++) {
      const int p = LANE + 64 * k;
      const F2 T = *(const F2 *)(trig + n2 - 2 * (p + 1));
      const float *q0, *q1;
      const int regime = 2 * (64 * k) < n8 ? 0 : (2 * (64 * k) < n2 - n8 ? 1 : 2);  // (wave-uniform: n8 is a multiple of 128 for n >= 1024)
      if (regime == 0) {
        q0 = in0 + n2 + n4 - 4 * (p + 1), q1 = in0 + n2 + n4 + 4 * p;
      } else if (regime == 1) {
        q0 = in0 + n2 + n4 - 4 * (p + 1), q1 = in0 + 4 * (p - n8 / 2);
      } else {
        q0 = in0 + n - 4 * (p - (n2 - n8) / 2 + 1), q1 = in0 + 4 * (p - n8 / 2);
      }
      F4 x0 = *(const F4 *)q0, x1 = *(const F4 *)q1;
      asm volatile("" : "+v"(x0.x), "+v"(x0.y), "+v"(x0.z), "+v"(x0.w), "+v"(x1.x), "+v"(x1.y), "+v"(x1.z), "+v"(x1.w));
      float r0, r1;
      if (regime == 0) {
        r0 = x0.z + x1.y, r1 = x0.x + x1.w;
      } else if (regime == 1) {
        r0 = x0.z - x1.y, r1 = x0.x - x1.w;
      } else {
        r0 = -x0.z - x1.y, r1 = -x0.x - x1.w;
      }
      F2 o;
      o.x = r1 * T.y + r0 * T.x;
      o.y = r1 * T.x - r0 * T.y;
      *(F2 *)(w2 + VAMD_PW(2 * p)) = o;
    }
  } else
  TEAM_EACH(pp, n4 << LOGS, tm) {
    VAMD_MDCT_SPLIT(pp, log2n - 2)
    const int p = g_;
    const F2 T = *(const F2 *)(trig + n2 - 2 * (p + 1));
    float r0, r1;
    // (the two quads of the input a pair folds, windowed on the way in where the caller asked for that)
    auto quad = [&](int at) {
      F4 x = *(const F4 *)(in + at);
      if constexpr (WIN) {
        const F4 wv = *(const F4 *)(win + at);
        x.x *= wv.x, x.y *= wv.y, x.z *= wv.z, x.w *= wv.w;
      }
      return x;
    };
    if (2 * p < n8) {
      const F4 x0 = quad(n2 + n4 - 4 * (p + 1));
      const F4 x1 = quad(n2 + n4 + 4 * p);
      r0 = x0.z + x1.y;
      r1 = x0.x + x1.w;
    } else if (2 * p < n2 - n8) {
      const F4 x0 = quad(n2 + n4 - 4 * (p + 1));
      const F4 x1 = quad(4 * (p - n8 / 2));
      r0 = x0.z - x1.y;
      r1 = x0.x - x1.w;
    } else {
      const F4 x0 = quad(n - 4 * (p - (n2 - n8) / 2 + 1));
      const F4 x1 = quad(4 * (p - n8 / 2));
      r0 = -x0.z - x1.y;
      r1 = -x0.x - x1.w;
    }
    F2 o;
    o.x = r1 * T.y + r0 * T.x;
    o.y = r1 * T.x - r0 * T.y;
    *(F2 *)(w2 + VAMD_PW(2 * p)) = o;
  }
  tm.sync();
  pc.mark(1);
  }

  // mdct_butterflies, lib/mdct.c:316-336, on x = w2, points = n2.
  // Stage s (s = 0 is mdct_butterfly_first, s >= 1 the generic passes) splits
  // x into 2^s sub-blocks of n2>>s points; butterfly q of a sub-block pairs
  // x[pts-2-2q] with x[pts/2-2-2q] and uses T[(4<<s)*q].  n2/4 = n/8 butterflies per
  // stage in total, all independent.
  const int nstages = log2n - 6;  // first + (log2n-7) generic passes
  // Two stages per trip through LDS where there are two to take: the butterflies (A, B) and (C, D) of stage s -- A, C
  // in the upper half of a sub-block, a quarter apart, B, D below them -- feed exactly the butterflies (A, C) and
  // (B, D) of stage s+1, whose sub-blocks are those halves.  Same operations on the same operands; half the loads
  // and stores (the stage is bound by the LDS pipe, stores above all).
  // ... and three where there are three: eight pairs E1..E8 an eighth of a sub-block apart (E8 on top) close under
  // the butterflies (E8,E4) (E7,E3) (E6,E2) (E5,E1) of stage s, (E8,E6) (E7,E5) (E4,E2) (E3,E1) of stage s+1 and
  // (E8,E7) (E6,E5) (E4,E3) (E2,E1) of stage s+2.
  for (; s + 2 < nstages && (nstages - s) != 4; s += 3) {
    const int pts = n2 >> s, lq = log2n - 5 - s;  // units per sub-block: pts/16 = 1 << lq
    const int e8 = pts >> 3;
    TEAM_EACH(gg, (n8 >> 2) << LOGS, tm) {
      VAMD_MDCT_SPLIT(gg, log2n - 5)
      const int g = g_;
      const int j = g >> lq, q = g & ((1 << lq) - 1);
      const int base = pts * j - 2 - 2 * q;
      F2 E[8];
#pragma unroll
      for (int k = 0; k < 8; k++) E[k] = *(const F2 *)(w2 + VAMD_PW(base + (k + 1) * e8));
      const int h = pts >> 4;  // q advances by a sixteenth of the sub-block from one eighth to the next
      bfly(E[7], E[3], stage_trig(s, q));
      bfly(E[6], E[2], stage_trig(s, q + h));
      bfly(E[5], E[1], stage_trig(s, q + 2 * h));
      bfly(E[4], E[0], stage_trig(s, q + 3 * h));
      const F2 T1a = stage_trig(s + 1, q), T1b = stage_trig(s + 1, q + h);
      bfly(E[7], E[5], T1a);
      bfly(E[6], E[4], T1b);
      bfly(E[3], E[1], T1a);
      bfly(E[2], E[0], T1b);
      const F2 T2 = stage_trig(s + 2, q);
      bfly(E[7], E[6], T2);
      bfly(E[5], E[4], T2);
      bfly(E[3], E[2], T2);
      bfly(E[1], E[0], T2);
#pragma unroll
      for (int k = 0; k < 8; k++) *(F2 *)(w2 + VAMD_PW(base + (k + 1) * e8)) = E[k];
    }
    tm.sync();
  }
  for (; s + 1 < nstages; s += 2) {
    const int pts = n2 >> s, lq = log2n - 4 - s;  // units per sub-block: pts/8 = 1 << lq
    TEAM_EACH(gg, (n8 >> 1) << LOGS, tm) {
      VAMD_MDCT_SPLIT(gg, log2n - 4)
      const int g = g_;
      const int j = g >> lq, q = g & ((1 << lq) - 1);
      const int base = pts * j - 2 - 2 * q;
      F2 *pA = (F2 *)(w2 + VAMD_PW(base + pts)), *pB = (F2 *)(w2 + VAMD_PW(base + (pts >> 1)));
      F2 *pC = (F2 *)(w2 + VAMD_PW(base + 3 * (pts >> 2))), *pD = (F2 *)(w2 + VAMD_PW(base + (pts >> 2)));
      const F2 Tab = stage_trig(s, q), Tcd = stage_trig(s, q + (pts >> 3)), T1 = stage_trig(s + 1, q);
      F2 A = *pA, B = *pB, C = *pC, D = *pD;
      bfly(A, B, Tab);
      bfly(C, D, Tcd);
      bfly(A, C, T1);
      bfly(B, D, T1);
      *pA = A;
      *pB = B;
      *pC = C;
      *pD = D;
    }
    tm.sync();
  }
  for (; s < nstages; s++) {
    const int pts = n2 >> s, lper = log2n - 3 - s;  // per = pts/4 = 1 << lper
    TEAM_EACH(gg, n8 << LOGS, tm) {
      VAMD_MDCT_SPLIT(gg, log2n - 3)
      const int g = g_;
      const int j = g >> lper, q = g & ((1 << lper) - 1);
      const int ia = pts * j + pts - 2 - 2 * q, ib = pts * j + (pts >> 1) - 2 - 2 * q;
      F2 *pa = (F2 *)(w2 + VAMD_PW(ia)), *pb = (F2 *)(w2 + VAMD_PW(ib));
      F2 a = *pa, b = *pb;
      bfly(a, b, stage_trig(s, q));
      *pa = a;
      *pb = b;
    }
    tm.sync();
  }
  pc.mark(2);
  // the 32-, 16- and 8-point levels (see bfly_level32 above): one thread finishes a group of 32 in registers --
  // one trip through LDS for all three levels (the stage is bound by the LDS pipe, not by idle lanes).  Group g
  // starts at w2 + 34 g (== VAMD_PW(32 g)); inside a group the padded layout is plain.
  TEAM_EACH(gg, (n2 / 32) << LOGS, tm) {
    VAMD_MDCT_SPLIT(gg, log2n - 6)
    F2 *pg = (F2 *)(w2 + 34 * g_);
    F2 e[16];
#pragma unroll
    for (int k = 0; k < 16; k++) e[k] = pg[k];
#pragma unroll
    for (int a = 0; a < 8; a++) {
      const PairOp r = bfly_level32(a, e[a], e[8 + a]);
      e[a] = r.lo;
      e[8 + a] = r.hi;
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int b2 = 0; b2 < 4; b2++) {
        const PairOp r = bfly_level16(b2, e[8 * h + b2], e[8 * h + 4 + b2]);
        e[8 * h + b2] = r.lo;
        e[8 * h + 4 + b2] = r.hi;
      }
    }
#pragma unroll
    for (int o = 0; o < 4; o++) {
      float v[8] = {e[4 * o].x, e[4 * o].y, e[4 * o + 1].x, e[4 * o + 1].y, e[4 * o + 2].x, e[4 * o + 2].y, e[4 * o + 3].x, e[4 * o + 3].y};
      bfly_level8(v);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        F2 t;
        t.x = v[2 * k];
        t.y = v[2 * k + 1];
        pg[4 * o + k] = t;
      }
    }
  }
  tm.sync();
  pc.mark(3);

  // mdct_bitreverse, lib/mdct.c:346-394: reads x = w2 (upper half), writes the
  // lower half w[0..n2).  Unit u produces w[2u], w[2u+1], w[n2-2u-2], w[n2-2u-1].
  const int *__restrict__ bit = P.bitrev;
  TEAM_EACH(uu, n8 << LOGS, tm) {
    VAMD_MDCT_SPLIT(uu, log2n - 3)
    const int u = g_;
    I2 bi;
    if (PACKED) {  // lib/mdct.c:77-88
      bi.y = (int)(brev32((unsigned)u) >> (32 - (log2n - 1)));
      bi.x = ((~bi.y) & ((1 << (log2n - 1)) - 1)) - 1;
    } else {
      bi = *(const I2 *)(bit + 2 * u);
    }
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
    // lo, hi are w[2u], w[2u+1] and w[n2-2u-2], w[n2-2u-1] of the reference's work vector: exactly the pairs that
    // items i = u and i = n4-1-u of the final rotate * scale (lib/mdct.c:552-561) consume, so they never go to
    // LDS.  (out_lds must not overlap the butterfly vector w2, which other threads are still gathering from: the
    // callers hand over w itself, or a buffer of their own.)
    {
      const F2 Tl = *(const F2 *)(trig + n2 + 2 * u);
      out_lds[u] = (lo.x * Tl.x + lo.y * Tl.y) * P.mdct_scale;
      out_lds[n2 - 1 - u] = (lo.x * Tl.y - lo.y * Tl.x) * P.mdct_scale;
      const int i2 = n4 - 1 - u;
      const F2 Th = *(const F2 *)(trig + n2 + 2 * i2);
      out_lds[i2] = (hi.x * Th.x + hi.y * Th.y) * P.mdct_scale;
      out_lds[n2 - 1 - i2] = (hi.x * Th.y - hi.y * Th.x) * P.mdct_scale;
    }
  }
  tm.sync();
  pc.mark(4);
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
// SRC3: cc is in the VAMD_F3_POS layout (the pass with ido = 64 that follows fft_pass3_wave): a k-block of 64
// starts 66 floats after the previous one; the four input quarters stay a constant distance apart.
template <bool AL, bool SRC3 = false, class Team>
VAMD_DEV void radf4_wave(int ido, int l1, const float *__restrict__ cc, float *__restrict__ ch, const float *__restrict__ wa1,
                         const float *__restrict__ wa2, const float *__restrict__ wa3, const Team &tm) {
  const float hsqt2 = .70710678118654752f;
  const int t0 = l1 * ido;
  const int ks = SRC3 ? ido + 2 : ido;              // source stride between k-blocks
  const int q0 = SRC3 ? t0 + (t0 >> 5) : t0;        // ... and between input quarters
  if (ido == 1) {
    // first pass: four contiguous outputs per k -> one 16-byte store (offset layout: index 4k at +1)
    TEAM_EACH(k, l1, tm) {
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
  TEAM_EACH(k, l1, tm) {
    {
      {
        const int t1 = q0 + k * ks, t2 = 3 * q0 + k * ks, t3 = k * ks, t4 = 2 * q0 + k * ks;
        const float tr1 = cc[t1] + cc[t2];
        const float tr2 = cc[t3] + cc[t4];
        int t5 = (k * ido) << 2;
        ch[t5] = tr1 + tr2;
        ch[(ido << 2) + t5 - 1] = tr2 - tr1;
        t5 += ido << 1;
        ch[t5 - 1] = cc[t3] - cc[t4];
        ch[t5] = cc[t2] - cc[t1];
      }
      {
        const int t1 = q0 + ido - 1 + k * ks, t2 = t1 + (q0 << 1);
        const int t4 = ido + k * (ido << 2), t5 = ido << 1, t6 = ido + k * ks;
        const float ti1 = -hsqt2 * (cc[t1] + cc[t2]);
        const float tr1 = hsqt2 * (cc[t1] - cc[t2]);
        ch[t4 - 1] = tr1 + cc[t6 - 1];
        ch[t4 + t5 - 1] = cc[t6 - 1] - tr1;
        ch[t4] = ti1 - cc[t1 + q0];
        ch[t4 + t5] = ti1 + cc[t1 + q0];
      }
    }
  }
  // the (k, i = 2m) butterflies, m >= 1 (the m = 0 slot of each k idles: k and m stay a shift and a mask)
  TEAM_EACH(g, l1 << lh, tm) {
    const int k = g >> lh, m = g & ((1 << lh) - 1);
    if (m != 0) {
      const int i = 2 * m;
      const int t1 = k * ido;
      const int t2 = k * ks + i;
      const int t4 = (t1 << 2) + i;
      const int t6 = ido << 1;
      const int t5 = t6 + (t1 << 2) - i;
      const F2 w1 = *(const F2 *)(wa1 + i - 2), w2 = *(const F2 *)(wa2 + i - 2), w3 = *(const F2 *)(wa3 + i - 2);
      const F2 c0 = ld_pair<AL>(cc, t2), c1 = ld_pair<AL>(cc, t2 + q0), c2 = ld_pair<AL>(cc, t2 + 2 * q0),
               c3 = ld_pair<AL>(cc, t2 + 3 * q0);
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

// ---- the first three passes (ido = 1, 4, 16) when n >= 256 --------------------------------------------------
// As radf4_wave runs them, the passes with a small ido scatter their output: a thread's stores land 4*ido floats from
// its neighbour's, i.e. on the same few LDS banks (16-way conflicts at ido = 4, 32-way for the two k-only columns at
// ido = 16) -- half of the stage's LDS time (SQ_LDS_BANK_CONFLICT).  Two changes, same butterflies:
//  * passes 1 and 2 run in ONE trip: the sixteen values in[k + (n/16) m] that the ido = 4 butterfly group k consumes
//    come from four ido = 1 butterflies that need nothing else, so a thread loads the sixteen, does both passes in
//    registers and stores sixteen consecutive outputs.  The ido = 4 twiddles are the same for every thread.
//  * the outputs go out in padded layouts: after pass 2 two floats of padding per 32 (VAMD_F2_POS, the butterfly
//    vector's layout) so that threads 16 floats apart store to distinct banks; after pass 3 two per 64
//    (VAMD_F3_POS), with pass 3 dealing sixteen DIFFERENT k to neighbouring lanes (their stores are then 66 floats
//    apart: distinct banks; and their loads 16 or 18 apart: distinct too).  The ido = 64 pass reads that layout
//    (radf4_wave<.., SRC3>) with lanes along i, where a plain layout is conflict-free, and writes plainly.
#define VAMD_F2_POS(t) ((t) + (((t) >> 5) << 1))
#define VAMD_F3_POS(t) ((t) + (((t) >> 6) << 1))

// passes 1 + 2: c plain [n] -> dst (offset layout, VAMD_F2_POS).  w = the ido = 4 pass's three twiddle pairs.
template <int LOGN, class Team>
VAMD_DEV void fft_pass12_wave(const float *__restrict__ c, float *__restrict__ dst, const float *__restrict__ wa1,
                              const float *__restrict__ wa2, const float *__restrict__ wa3, const Team &tm) {
  constexpr int n = 1 << LOGN, u = n / 16;
  const float hsqt2 = .70710678118654752f;
  const F2 w1 = *(const F2 *)wa1, w2 = *(const F2 *)wa2, w3 = *(const F2 *)wa3;
  TEAM_EACH(k, u, tm) {
    float x[16];
#pragma unroll
    for (int m = 0; m < 16; m++) x[m] = c[k + u * m];
    // dradf4 with ido = 1 (lib/smallft.c:176-193), butterfly k + u*t: its cc[t0+k], cc[3t0+k], cc[k], cc[2t0+k]
    // are x[t+4], x[t+12], x[t], x[t+8]; y[t][0..3] = its four outputs = the ido = 4 pass's cc[4k + i + (n/4) t]
    float y[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const float c1 = x[t + 4], c2 = x[t + 12], c3 = x[t], c4 = x[t + 8];
      const float tr1 = c1 + c2, tr2 = c3 + c4;
      y[t][0] = tr1 + tr2;
      y[t][1] = c3 - c4;
      y[t][2] = c2 - c1;
      y[t][3] = tr2 - tr1;
    }
    float o[16];
    {  // ido = 4, the i = 0 column (lib/smallft.c:176-193): cc[t1], cc[t2], cc[t3], cc[t4] = y[1][0], y[3][0], y[0][0], y[2][0]
      const float tr1 = y[1][0] + y[3][0];
      const float tr2 = y[0][0] + y[2][0];
      o[0] = tr1 + tr2;
      o[15] = tr2 - tr1;
      o[7] = y[0][0] - y[2][0];
      o[8] = y[3][0] - y[1][0];
    }
    {  // the i = ido column (:246-268): cc[t1], cc[t2], cc[t6-1], cc[t1+t0] = y[1][3], y[3][3], y[0][3], y[2][3]
      const float ti1 = -hsqt2 * (y[1][3] + y[3][3]);
      const float tr1 = hsqt2 * (y[1][3] - y[3][3]);
      o[3] = tr1 + y[0][3];
      o[11] = y[0][3] - tr1;
      o[4] = ti1 - y[2][3];
      o[12] = ti1 + y[2][3];
    }
    {  // the i = 2 butterfly (:197-243) on the pairs (y[t][1], y[t][2])
      const float cr2 = w1.x * y[1][1] + w1.y * y[1][2];
      const float ci2 = w1.x * y[1][2] - w1.y * y[1][1];
      const float cr3 = w2.x * y[2][1] + w2.y * y[2][2];
      const float ci3 = w2.x * y[2][2] - w2.y * y[2][1];
      const float cr4 = w3.x * y[3][1] + w3.y * y[3][2];
      const float ci4 = w3.x * y[3][2] - w3.y * y[3][1];
      const float tr1 = cr2 + cr4, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
      const float ti2 = y[0][2] + ci3, ti3 = y[0][2] - ci3;
      const float tr2 = y[0][1] + cr3, tr3 = y[0][1] - cr3;
      o[1] = tr1 + tr2;
      o[2] = ti1 + ti2;
      o[5] = tr3 - ti4;
      o[6] = tr4 - ti3;
      o[9] = ti4 + tr3;
      o[10] = tr4 + ti3;
      o[13] = tr2 - tr1;
      o[14] = ti1 - ti2;
    }
    float *d = dst + VAMD_F2_POS(16 * k);  // the sixteen share a group of 32: one pad for all
    d[0] = o[0];
#pragma unroll
    for (int m = 1; m < 8; m++) st_pair(d, 2 * m, o[2 * m - 1], o[2 * m]);
    d[15] = o[15];
  }
}

// pass 3 (ido = 16, l1 = n/64): src offset layout in VAMD_F2_POS, dst offset layout in VAMD_F3_POS.
template <int LOGN, class Team>
VAMD_DEV void fft_pass3_wave(const float *__restrict__ cc, float *__restrict__ ch, const float *__restrict__ wa1,
                             const float *__restrict__ wa2, const float *__restrict__ wa3, const Team &tm) {
  constexpr int n = 1 << LOGN, ido = 16, l1 = n / 64, t0 = n / 4;
  constexpr int LL1 = LOGN - 6;
  constexpr int q0 = t0 + (t0 >> 4);  // VAMD_F2_POS of a quarter's start (t0 is a multiple of 32)
  const float hsqt2 = .70710678118654752f;
  // the two k-only columns
  TEAM_EACH(k, l1, tm) {
    const float *s = cc + VAMD_F2_POS(16 * k);
    float *d = ch + 66 * k;
    {
      const float a1 = s[q0], a2 = s[3 * q0], a3 = s[0], a4 = s[2 * q0];
      const float tr1 = a1 + a2;
      const float tr2 = a3 + a4;
      d[0] = tr1 + tr2;
      d[63] = tr2 - tr1;
      d[31] = a3 - a4;
      d[32] = a2 - a1;
    }
    {
      const float b1 = s[q0 + 15], b2 = s[3 * q0 + 15], b6 = s[15], b3 = s[2 * q0 + 15];
      const float ti1 = -hsqt2 * (b1 + b2);
      const float tr1 = hsqt2 * (b1 - b2);
      d[15] = tr1 + b6;
      d[47] = b6 - tr1;
      d[16] = ti1 - b3;
      d[48] = ti1 + b3;
    }
  }
  // the (k, i = 2m) butterflies, m = 1..7: sixteen k side by side in the wave, then m
  TEAM_EACH(g, l1 * 8, tm) {
    int k, m;
    if (LL1 >= 4) {
      k = (g & 15) | (((g >> 6) & ((1 << (LL1 - 4)) - 1)) << 4);
      m = ((g >> 4) & 3) | ((g >> (6 + (LL1 >= 4 ? LL1 - 4 : 0))) << 2);
    } else {
      k = g & (l1 - 1);
      m = g >> LL1;
    }
    if (m != 0) {
      const int i = 2 * m;
      const float *s = cc + VAMD_F2_POS(16 * k) + i;
      float *d = ch + 66 * k;
      const F2 w1 = *(const F2 *)(wa1 + i - 2), w2 = *(const F2 *)(wa2 + i - 2), w3 = *(const F2 *)(wa3 + i - 2);
      const F2 c0 = ld_pair<true>(s, 0), c1 = ld_pair<true>(s, q0), c2 = ld_pair<true>(s, 2 * q0), c3 = ld_pair<true>(s, 3 * q0);
      const float cr2 = w1.x * c1.x + w1.y * c1.y;
      const float ci2 = w1.x * c1.y - w1.y * c1.x;
      const float cr3 = w2.x * c2.x + w2.y * c2.y;
      const float ci3 = w2.x * c2.y - w2.y * c2.x;
      const float cr4 = w3.x * c3.x + w3.y * c3.y;
      const float ci4 = w3.x * c3.y - w3.y * c3.x;
      const float tr1 = cr2 + cr4, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
      const float ti2 = c0.y + ci3, ti3 = c0.y - ci3;
      const float tr2 = c0.x + cr3, tr3 = c0.x - cr3;
      st_pair(d, i, tr1 + tr2, ti1 + ti2);
      st_pair(d, 32 - i, tr3 - ti4, tr4 - ti3);
      st_pair(d, 32 + i, ti4 + tr3, tr4 + ti3);
      st_pair(d, 64 - i, tr2 - tr1, ti1 - ti2);
    }
  }
}


// dradf2, lib/smallft.c:113-166, same flattening
template <bool AL, class Team>
VAMD_DEV void radf2_wave(int ido, int l1, const float *__restrict__ cc, float *__restrict__ ch, const float *__restrict__ wa1,
                         const Team &tm) {
  const int t0 = l1 * ido;
  if (ido == 1) {
    TEAM_EACH(k, l1, tm) {
      ch[2 * k] = cc[k] + cc[t0 + k];
      ch[2 * k + 1] = cc[k] - cc[t0 + k];
    }
    return;
  }
  const int lh = 31 - __builtin_clz((unsigned)(ido >> 1));
  TEAM_EACH(g, l1 << lh, tm) {
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

// The last radix-4 pass (l1 = 2, ido = n/8) and the radix-2 pass that ends an odd log2 n (ido = n/2) in ONE trip:
// the radix-2 butterfly at index i takes the pair at i of the radix-4 pass's first output block and the pair at i of
// its second, and the radix-4 butterflies (k = 0, i) and (k = 1, i) produce exactly the pairs at i, 2ido-i, 2ido+i
// and 4ido-i of those two blocks -- so a thread runs both, then the four radix-2 butterflies on what it holds.
// Unit 0 takes the two k-only columns of both blocks (values at 0, ido-1, ido, ... 4ido-1) and the radix-2 work
// they feed (its i = 0 column and the butterflies at ido, 2ido, 3ido).  src, dst: offset layout, plain.
template <int LOGN, class Team>
VAMD_DEV void fft_tail42_wave(const float *__restrict__ cc, float *__restrict__ ch, const float *__restrict__ wa,
                              const Team &tm) {
  constexpr int n = 1 << LOGN, ido = n / 8, t0 = n / 4, n2 = n / 2;
  const float hsqt2 = .70710678118654752f;
  const float *__restrict__ wa1 = wa + n2, *__restrict__ wa2 = wa1 + ido, *__restrict__ wa3 = wa2 + ido;  // iw = n/2 + 1
  // dradf2's butterfly at i (lib/smallft.c:139-163) on the pairs c5 (first block) and c3 (second block)
  auto radf2_bfly = [&](int i, const F2 c5, const F2 c3) {
    const F2 w1 = *(const F2 *)(wa + i - 2);  // iw = 1
    const float tr2 = w1.x * c3.x + w1.y * c3.y;
    const float ti2 = w1.x * c3.y - w1.y * c3.x;
    st_pair(ch, i, c5.x + tr2, c5.y + ti2);
    st_pair(ch, n - i, c5.x - tr2, ti2 - c5.y);
  };
  TEAM_EACH(g, ido / 2, tm) {
    if (g == 0) {
      float V[2][8];  // block k's values at 0, ido-1, ido, 2ido-1, 2ido, 3ido-1, 3ido, 4ido-1
#pragma unroll
      for (int k = 0; k < 2; k++) {
        {
          const int t1 = t0 + k * ido, t2 = 3 * t0 + k * ido, t3 = k * ido, t4 = 2 * t0 + k * ido;
          const float tr1 = cc[t1] + cc[t2];
          const float tr2 = cc[t3] + cc[t4];
          V[k][0] = tr1 + tr2;
          V[k][7] = tr2 - tr1;
          V[k][3] = cc[t3] - cc[t4];
          V[k][4] = cc[t2] - cc[t1];
        }
        {
          const int t1 = t0 + ido - 1 + k * ido, t2 = t1 + (t0 << 1), t6 = ido + k * ido;
          const float ti1 = -hsqt2 * (cc[t1] + cc[t2]);
          const float tr1 = hsqt2 * (cc[t1] - cc[t2]);
          V[k][1] = tr1 + cc[t6 - 1];
          V[k][5] = cc[t6 - 1] - tr1;
          V[k][2] = ti1 - cc[t1 + t0];
          V[k][6] = ti1 + cc[t1 + t0];
        }
      }
      // dradf2's i = 0 column (lib/smallft.c:121-136): cc[0], cc[t0] and the two values at ido-1 of its own pass
      ch[0] = V[0][0] + V[1][0];
      ch[n - 1] = V[0][0] - V[1][0];
      ch[n2] = -V[1][7];
      ch[n2 - 1] = V[0][7];
#pragma unroll
      for (int j = 1; j < 4; j++) {
        F2 c5, c3;
        c5.x = V[0][2 * j - 1], c5.y = V[0][2 * j];
        c3.x = V[1][2 * j - 1], c3.y = V[1][2 * j];
        radf2_bfly(j * ido, c5, c3);
      }
    } else {
      const int i = 2 * g;
      const F2 w1 = *(const F2 *)(wa1 + i - 2), w2 = *(const F2 *)(wa2 + i - 2), w3 = *(const F2 *)(wa3 + i - 2);
      F2 Pk[2][4];  // block k's pairs at i, 2ido-i, 2ido+i, 4ido-i
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int t2 = k * ido + i;
        const F2 c0 = ld_pair<true>(cc, t2), c1 = ld_pair<true>(cc, t2 + t0), c2 = ld_pair<true>(cc, t2 + 2 * t0),
                 c3 = ld_pair<true>(cc, t2 + 3 * t0);
        const float cr2 = w1.x * c1.x + w1.y * c1.y;
        const float ci2 = w1.x * c1.y - w1.y * c1.x;
        const float cr3 = w2.x * c2.x + w2.y * c2.y;
        const float ci3 = w2.x * c2.y - w2.y * c2.x;
        const float cr4 = w3.x * c3.x + w3.y * c3.y;
        const float ci4 = w3.x * c3.y - w3.y * c3.x;
        const float tr1 = cr2 + cr4, tr4 = cr4 - cr2, ti1 = ci2 + ci4, ti4 = ci2 - ci4;
        const float ti2 = c0.y + ci3, ti3 = c0.y - ci3;
        const float tr2 = c0.x + cr3, tr3 = c0.x - cr3;
        Pk[k][0].x = tr1 + tr2, Pk[k][0].y = ti1 + ti2;
        Pk[k][1].x = tr3 - ti4, Pk[k][1].y = tr4 - ti3;
        Pk[k][2].x = ti4 + tr3, Pk[k][2].y = tr4 + ti3;
        Pk[k][3].x = tr2 - tr1, Pk[k][3].y = ti1 - ti2;
      }
      radf2_bfly(i, Pk[0][0], Pk[1][0]);
      radf2_bfly(2 * ido - i, Pk[0][1], Pk[1][1]);
      radf2_bfly(2 * ido + i, Pk[0][2], Pk[1][2]);
      radf2_bfly(4 * ido - i, Pk[0][3], Pk[1][3]);
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
template <int LOGN = 0, class Team = WaveTeam>
VAMD_DEV const float *drft_forward_wave(const XformP &P, float *c, float *ch, const Team &tm = Team()) {
  const int n = LOGN ? (1 << LOGN) : P.n, nf = LOGN ? (LOGN >> 1) + (LOGN & 1) : P.fft_nf;
  const float *__restrict__ wa = P.wa;
  float *bufc = c + 1, *bufh = ch + 1;  // offset layouts of the two buffers
  int na = 1, l2 = n, iw = n, kfirst = 0;
  if (LOGN >= 8) {
    // ido = 1 and 4 in one trip (c -> ch), ido = 16 (ch -> c); twiddles as drftf1 would hand them to dradf4:
    // iw = n - 3 after the first pass, n - 15 for the second, n - 63 for the third
    fft_pass12_wave<LOGN ? LOGN : 8>(c, bufh, wa + n - 16, wa + n - 12, wa + n - 8, tm);
    tm.sync();
    fft_pass3_wave<LOGN ? LOGN : 8>(bufh, bufc, wa + n - 64, wa + n - 48, wa + n - 32, tm);
    tm.sync();
    na = 1;  // the data is in c: the next pass writes ch
    l2 = n >> 6;
    iw = n - 63;
    kfirst = 3;
  }
  constexpr bool TAIL42 = LOGN >= 11 && (LOGN & 1);  // (at log2 n = 9 the last radix-4 pass reads the padded layout)
  const int nloop = TAIL42 ? nf - 2 : nf;
#pragma unroll
  for (int k1 = kfirst; k1 < nloop; k1++) {
    const int ip = LOGN ? (k1 < (LOGN >> 1) ? 4 : 2) : P.fft_fac[nf - k1 - 1];
    const int l1 = l2 / ip, ido = n / l2;
    iw -= (ip - 1) * ido;
    na = 1 - na;
    float *dst = na ? bufc : bufh;
    if (k1 == 0) {
      // the first pass reads the plain (un-offset) block; ld_pair<false> covers the case
      // where it has pairs to read (ido > 2)
      if (ip == 4)
        radf4_wave<false, false>(ido, l1, c, dst, wa + iw - 1, wa + iw + ido - 1, wa + iw + 2 * ido - 1, tm);
      else
        radf2_wave<false>(ido, l1, c, dst, wa + iw - 1, tm);
    } else {
      const float *src = na ? bufh : bufc;
      if (ip == 4 && LOGN >= 8 && k1 == 3)
        radf4_wave<true, true>(ido, l1, src, dst, wa + iw - 1, wa + iw + ido - 1, wa + iw + 2 * ido - 1, tm);
      else if (ip == 4)
        radf4_wave<true, false>(ido, l1, src, dst, wa + iw - 1, wa + iw + ido - 1, wa + iw + 2 * ido - 1, tm);
      else
        radf2_wave<true>(ido, l1, src, dst, wa + iw - 1, tm);
    }
    tm.sync();
    l2 = l1;
  }
  if (TAIL42) {
    na = 1 - na;
    fft_tail42_wave<TAIL42 ? LOGN : 11>(na ? bufh : bufc, na ? bufc : bufh, wa, tm);
    tm.sync();
  }
  // the reference copies ch back into c when the last pass landed in ch; the caller
  // just reads whichever buffer holds the result
  return na ? bufc : bufh;
}

// The whole stage for one channel-block.
//   A, B     LDS [VAMD_XF_A_FLOATS(n)], [VAMD_XF_B_FLOATS(n)]
//   outputs  HBM, each may be null
// `tile` holds the un-windowed block (pcm_fetch); it is consumed by transform_window, so the caller may refill
// it for the next block as soon as that returns.
template <int QPT, class Team>
VAMD_DEV void transform_window(const XformP &P, int W, int lW, int nW, const PcmTile<QPT> &tile, float *A, PhaseClock &pc,
                               const Team &tm) {
  window_store(P, W, lW, nW, tile, A, true, tm);
  tm.sync();
  pc.mark(0);
}

// MDCT (spectrum and its dB twin out to HBM) and the FFT of the same windowed block.  Returns the buffer holding the
// packed FFT output (offset layout), which is A + 1 or B + 1 depending on the number of passes: when it is B's, A
// is free again and the caller may already put the next block into it.
template <int LOGN = 0, class Team = WaveTeam>
VAMD_DEV const float *transform_spectra(const XformP &P, float *A, float *B, float *__restrict__ mdct_out,
                                        float *__restrict__ logmdct_out, PhaseClock &pc, const Team &tm = Team()) {
  const int n = LOGN ? (1 << LOGN) : P.n, n2 = n >> 1;
  // MDCT: spectrum lands in B[0..n2) (LDS), then goes out with its dB twin, 16 bytes per thread and tensor
  mdct_forward_wave<0, LOGN, Team, LOGN != 0>(P, A, B, B, pc, 0, 0, 0, tm);
  TEAM_EACH(q, n2 >> 2, tm) {
    float m[4], l[4];
    f4_get(((const F4 *)B)[q], m);
    for (int c = 0; c < 4; c++) l[c] = todB_345(m[c]);  // lib/mapping0.c:384-385
    if (mdct_out) ((F4 *)mdct_out)[q] = f4_make(m);
    if (logmdct_out) ((F4 *)logmdct_out)[q] = f4_make(l);
  }
  tm.sync();
  pc.mark(5);
  // FFT of the same windowed block (A), ping-ponging with B
  const float *spec = drft_forward_wave<LOGN, Team>(P, A, B, tm);
  pc.mark(6);
  return spec;
}

// quads of an n/2-bin spectrum per thread of a one-wave team (the generic size: up to 4096 samples)
#define VAMD_XF_QPS(LOGN) ((LOGN) ? (((1 << (LOGN)) / 8 + 63) / 64) : 4096 / 8 / 64)
// the run ids transform_logfft wants, for this thread's quads (fetched once by a persistent kernel)
template <int LOGN, class Team>
VAMD_DEV void xf_run_ids(const XformP &P, const unsigned short *__restrict__ run_of_bin, I2 *rid, const Team &tm) {
  const int n2 = (LOGN ? (1 << LOGN) : P.n) >> 1;
  if (!LOGN) return;  // (the generic size reads the table as it goes)
  TEAM_QUADS(kq, q, n2 >> 2, VAMD_XF_QPS(LOGN), tm) rid[kq] = ((const I2 *)run_of_bin)[q];
}

// logfft + local ampmax, lib/mapping0.c:255-346; four bins per thread.  Returns the local_ampmax contribution of
// THIS WAVE (every lane holds it): the caller combines the team's waves.
//   peaks_out  (optional) HBM [nruns]: the maximum of logfft over each run of bins that share an octave line -- all the
//              tone stage reads of logfft (seed_loop's inner maximum, lib/psy.c:429-440; run_peak states it serially).
//              run_of_bin [n/2] = the run of each bin (PsyP::run_of_bin); rid[kq] = the same for the four bins of this
//              thread's quad number kq, 16 bits each, where the size is fixed at compile time (the persistent kernel
//              fetches them once, they are the same for every block: xf_run_ids); peaks_lds = nruns floats of LDS that
//              `spec` does not overlap.  Every bin's value is folded into its run's slot with an LDS float maximum as it is
//              formed: no pass of its own, nothing that waits -- this stage has two waves per SIMD to hide a wait behind.
//              (The order of a maximum matters for nothing here: the values are finite, and a tie between +0 and -0
//              feeds additions only.)
template <int LOGN = 0, class Team = WaveTeam>
VAMD_DEV float transform_logfft(const XformP &P, const float *spec, float *__restrict__ logfft_out, PhaseClock &pc,
                                const Team &tm = Team(), float *raw_max = nullptr, float *peaks_lds = nullptr,
                                const I2 *rid = nullptr, const unsigned short *__restrict__ run_of_bin = nullptr,
                                int nruns = 0, float *__restrict__ peaks_out = nullptr) {
  const int n = LOGN ? (1 << LOGN) : P.n, n2 = n >> 1;
  const float scale = 4.f / n;
  const float scale_dB = todB_345(scale);
  float amp = -1e30f;
  if (peaks_out) {
    TEAM_EACH(r, nruns, tm) peaks_lds[r] = f_from_bits(0xff800000u);  // -inf: every run has a bin, and every bin a finite value
    tm.sync();
  }
  int kq = 0;  // (the size-specialised kernels unroll this loop entirely: rid[kq] is a register)
  TEAM_EACH(q, n2 >> 2, tm) {
    float v[4];
    for (int c = 0; c < 4; c++) {
      const int k = 4 * q + c;
      if (k == 0) {
        v[c] = (float)((double)(scale_dB + todB(spec[0])) + .345);
      } else {
        const F2 z = *(const F2 *)(spec + 2 * k - 1);  // (Re_k, Im_k), aligned in the offset layout
        const float temp = z.x * z.x + z.y * z.y;
        v[c] = (float)((double)(scale_dB + .5f * todB(temp)) + .345);
      }
      amp = fmaxf(amp, v[c]);
    }
    if (logfft_out) ((F4 *)logfft_out)[q] = f4_make(v);
    if (peaks_out) {
      const I2 r = LOGN ? rid[kq] : ((const I2 *)run_of_bin)[q];
      lds_atomic_max(peaks_lds + (r.x & 0xffff), v[0]);
      lds_atomic_max(peaks_lds + (int)((unsigned)r.x >> 16), v[1]);
      lds_atomic_max(peaks_lds + (r.y & 0xffff), v[2]);
      lds_atomic_max(peaks_lds + (int)((unsigned)r.y >> 16), v[3]);
    }
    kq++;
  }
  if (peaks_out) {
    tm.sync();
    TEAM_EACH(r, nruns, tm) peaks_out[r] = peaks_lds[r];
  }
  // (every v[c] is finite whatever the samples were -- todB() converts the BITS of its argument, so a NaN or Inf
  // spectrum value becomes a large finite dB figure -- hence fmaxf here and in wave_max is the reference's
  // `if(temp>local_ampmax[i])` compare, lib/mapping0.c:343, on every input)
  amp = wave_max(amp);
  if (raw_max) *raw_max = amp;  // before the clamp: the input-domain test (VAMD_INPUT_LIMIT_DB)
  if (amp > 0.f) amp = 0.f;  // lib/mapping0.c:345 (the clamp commutes with the maximum over the team's waves)
  tm.sync();
  pc.mark(7);
  return amp;
}

template <int LOGN = 0, class Team = WaveTeam>
VAMD_DEV float transform_block(const XformP &P, float *A, float *B, float *__restrict__ mdct_out,
                               float *__restrict__ logmdct_out, float *__restrict__ logfft_out, PhaseClock &pc,
                               const Team &tm = Team(), float *raw_max = nullptr, const I2 *rid = nullptr,
                               const unsigned short *__restrict__ run_of_bin = nullptr, int nruns = 0,
                               float *__restrict__ peaks_out = nullptr) {
  const float *spec = transform_spectra<LOGN, Team>(P, A, B, mdct_out, logmdct_out, pc, tm);
  // (the spectrum sits in one of the two buffers: the other one is free by now)
  float *lds_free = (spec >= A && spec < A + VAMD_XF_A_FLOATS(P.n)) ? B : A;
  return transform_logfft<LOGN, Team>(P, spec, logfft_out, pc, tm, raw_max, lds_free, rid, run_of_bin, nruns, peaks_out);
}

// log2 n when the size-specialised transforms apply -- a power of two in [256, 4096] whose FFT factors are the ones
// they assume (radix 4 throughout, one radix-2 pass last for an odd log2 n) -- else 0 (the general code).
inline int fixed_logn(const XformP &P) {
  const int l = P.log2n;
  if (l < 8 || l > 12 || (1 << l) != P.n || P.fft_nf != (l >> 1) + (l & 1) || !P.bitrev_std) return 0;
  for (int k1 = 0; k1 < P.fft_nf; k1++)
    if (P.fft_fac[P.fft_nf - k1 - 1] != (k1 < (l >> 1) ? 4 : 2)) return 0;
  return l;
}

}  // namespace vamd
