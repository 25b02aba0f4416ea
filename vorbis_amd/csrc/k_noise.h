// k_noise.h -- _vp_noisemask (reference lib/psy.c:706-752) with its two
// bark_noise_hybridmp passes (lib/psy.c:547-704); SURVEY.md 8a row a9.  A team
// of up to four wavefronts per channel-block.
//
// What is parallel and what is not: the per-bin terms y, w, w*x, w*x*x, w*y,
// w*x*y and the windowed line evaluation are independent per bin: lane l of
// wave w takes the bins i0(w) + l + 64k, so that every LDS access of a wave is a
// run of consecutive words (no bank conflicts).  The five running sums N, X,
// XX, Y, XY are fp32 accumulations *in index order* in the reference; windowed
// differences of them feed a division, so any re-association moves the answer
// by up to 5e-3.  They are walked in order by the team's
// first wave, eight lanes per chain (running_sum_rounds below).  The regime
// boundaries of the window loops depend only on the static bark[] table and are
// precomputed by vamd_create().
//
// LDS: S[5][VAMD_NZ_STRIDE(n)] running sums only; the noise curve and work vector stay in registers.
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"

namespace vamd {

// The five running-sum arrays of a block in LDS.  Element e of an array lives at word nz_swz(e): bit 6 of
// the index is folded into bit 2, which swaps neighbouring 16-byte quads in every other run of 64.  The
// per-bin phases touch consecutive e from consecutive lanes and stay conflict-free under it (a wave's 32
// lanes of an LDS cycle still cover 32 different banks); the ordered walk (running_sum_rounds), where each
// lane of a chain's eight moves the 64 bytes of its own run, needs it: without the fold lanes j and j+4 of
// a chain would queue on the same four banks.  The arrays are exactly n floats apart: at 1024 bins a block's
// five arrays are 20 480 bytes and EIGHT blocks fill a CU's 160 KB to the byte -- one more block in flight is
// worth more than the few LDS cycles the walk's 16-byte accesses lose to the chains sharing their banks.
#define VAMD_NZ_STRIDE(n) (n)
VAMD_DEV int nz_swz(int e) { return e ^ ((e >> 4) & 4); }

struct LineFit {
  float A, B, D;
};

// A, B, D of one windowed least-squares line, from (tN,tX,tXX,tY,tXY); the
// expression order is the reference's (lib/psy.c:619-621)
VAMD_DEV LineFit fit_from_sums(float tN, float tX, float tXX, float tY, float tXY) {
  LineFit r;
  r.A = tY * tXX - tX * tXY;
  r.B = tN * tXY - tX * tY;
  r.D = tN * tXX - tX * tX;
  return r;
}

// In-place running sum p[i] = p[0] + ... + p[i], strictly left to right in fp32
// (the reference's tN/tX/... accumulators, lib/psy.c:576-603), one lane per array.
VAMD_DEV void running_sum_inplace(float *p, int n) {
  float acc = 0.f;
  for (int i = 0; i < n; i++) {
    acc += p[nz_swz(i)];
    p[nz_swz(i)] = acc;
  }
}

#if VAMD_GPU
// The running sums of up to eight chains by ONE wave (tools/micro/scan_e.hip).  A lone wave issues an
// instruction every ~5 cycles and a dependent v_add_f32 every ~7.2, so what a walk costs is the number of
// instructions it puts between two elements of a chain.  Here L = 8 lanes share a chain; per step each lane
// loads E consecutive elements (its "run"), and the runs are summed in L rounds under an exec mask: in
// round r only lane r of every chain executes, E back-to-back dependent adds starting from the total that
// lane r-1 handed over (DPP row_shr:1).  A round is E adds + ~5 bookkeeping instructions, so with E = 16
// the wave walks at 8.5 cycles per element of every chain it carries (E = 4, the hand-off after every
// quad that round 1 shipped, measured 14; one lane per chain out of a chain-interleaved layout 20.6).
// Chain c lives at S + c*stride; n is a multiple of 8*E.
VAMD_DEV float dpp_from_lane_below(float v) {  // row_shr:1 -- lane i receives lane i-1's value
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}
template <int L>
VAMD_DEV float dpp_from_last_lane(float v) {  // row_shl:(L-1) -- lane i receives lane i+L-1's value
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + (L - 1), 0xf, 0xf, true));
}
// L = 8 or 16 lanes per chain (64/L chains per wave; a chain must not straddle a row of 16 lanes: DPP)
template <int L, int E>
VAMD_DEV void running_sum_rounds(float *S, int stride, int nchains, int n) {
  constexpr int Q = E / 4;
  const int g = LANE / L, j = LANE % L;
  const bool act = g < nchains;
  // this lane's run of step s: elements [s*L*E + j*E, +E); its quad k sits at quad k ^ m of the run (nz_swz:
  // bit 6 of the element index, constant per lane for E >= 8; E = 4: such blocks are too short to matter)
  const int m = E >= 8 ? ((j * E) >> 6) & 1 : 0;
  float *p = S + (act ? g : 0) * stride + j * E;
  const int steps = n / (L * E);
  float c0 = 0.f;  // lane j == 0: the chain's total through the previous step
  for (int s = 0; s < steps; s++) {
    float v[E];
#pragma unroll
    for (int k = 0; k < Q; k++) f4_get(*(const F4 *)(p + s * L * E + 4 * (k ^ m)), v + 4 * k);
    float cin = c0;
#pragma unroll
    for (int r = 0; r < L; r++) {
      if (j == r) {
        v[0] = cin + v[0];
#pragma unroll
        for (int k = 1; k < E; k++) v[k] = v[k - 1] + v[k];
      }
      cin = dpp_from_lane_below(v[E - 1]);  // lane r+1 now holds lane r's total (the others do not care)
    }
    c0 = dpp_from_last_lane<L>(v[E - 1]);
    if (act) {
#pragma unroll
      for (int k = 0; k < Q; k++) *(F4 *)(p + s * L * E + 4 * (k ^ m)) = f4_make(v + 4 * k);
    }
  }
}

// workgroup barrier for data exchanged through LDS: the wave's own LDS traffic has landed, then s_barrier.
// (__syncthreads() would also drain the HBM loads and stores in flight.)
VAMD_DEV void team_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Who runs the five running sums of a block: the team's first wave, between two barriers.
struct ScanTeam {
  VAMD_MEM void before_terms() const { team_lds_barrier(); }  // nobody still evaluates lines from the previous sums
  VAMD_MEM void scan(float *S, int n) const {
    team_lds_barrier();
    if (threadIdx.x < 64) {
      // the walk is one dependent add after the other: every issue slot it loses to a neighbour's per-bin
      // phase on the same SIMD lengthens it, so it goes first
      __builtin_amdgcn_s_setprio(3);
      if ((n & 127) == 0) running_sum_rounds<8, 16>(S, VAMD_NZ_STRIDE(n), 5, n);
      else if ((n & 63) == 0) running_sum_rounds<8, 8>(S, VAMD_NZ_STRIDE(n), 5, n);
      else if ((n & 31) == 0) running_sum_rounds<8, 4>(S, VAMD_NZ_STRIDE(n), 5, n);
      else if (LANE < 5) running_sum_inplace(S + LANE * VAMD_NZ_STRIDE(n), n);
      __builtin_amdgcn_s_setprio(0);
    }
    team_lds_barrier();
  }
};
#endif

// One window of running sums -> (A, B, D).  `hi` is the upper edge; `e2` the lower edge, or the mirrored
// lower edge -lo when `mir` (lo < 0 in the reference: lib/psy.c:613-617,666-670 against :635-639,687-691);
// both already as LDS word positions (nz_swz).
// MIR = false: the caller knows that none of the wave's bins is in the mirrored regime.
template <bool MIR>
VAMD_DEV LineFit fit_window(const float *S, int n, int hi, int e2, bool mir) {
  const float *N = S, *X = S + VAMD_NZ_STRIDE(n), *XX = S + 2 * VAMD_NZ_STRIDE(n), *Y = S + 3 * VAMD_NZ_STRIDE(n), *XY = S + 4 * VAMD_NZ_STRIDE(n);
  const float n1 = N[hi], x1 = X[hi], xx1 = XX[hi], y1 = Y[hi], xy1 = XY[hi];
  const float n2 = N[e2], x2 = X[e2], xx2 = XX[e2], y2 = Y[e2], xy2 = XY[e2];
  if (MIR && mir) return fit_from_sums(n1 + n2, x1 - x2, xx1 + xx2, y1 + y2, xy1 - xy2);
  return fit_from_sums(n1 - n2, x1 - x2, xx1 - xx2, y1 - y2, xy1 - xy2);
}

// bark_noise_hybridmp(n, bark, f, noise, offset, fixed).  f and noise are per-lane register
// arrays: this lane owns the bins i0 + LANE + 64k, k < KPL (LANE_BINS).
//   bk[k]   the window of bin min(i, bark_i2 - 1) -- the bins past the last fitted line extend it
//           (lib/psy.c:649-656), which is what evaluating that line's window again yields -- as LDS word
//           positions: upper edge | lower (or mirrored lower) edge << 16 (noise_bark_edges).  Prepared by
//           the caller once per block type (noise_bark_fetch + noise_bark_edges): a property of the lane's bins,
//           not of the block.
// LOGN > 0: the bin count is the compile-time constant 2^LOGN (array strides and window-edge addresses
// then fold into immediate offsets); 0 = P.n.
template <class Scan, int KPL, int LOGN>
VAMD_DEV void bark_noise_bins(const PsyP &P, const float *f, const int *bk, float *noise, const float offset,
                              const int fixed, float *S, const Scan &scan, PhaseClock &pc, int slot, int i0) {
  const int n = LOGN ? (1 << LOGN) : P.n;
  float *N = S, *X = S + VAMD_NZ_STRIDE(n), *XX = S + 2 * VAMD_NZ_STRIDE(n), *Y = S + 3 * VAMD_NZ_STRIDE(n), *XY = S + 4 * VAMD_NZ_STRIDE(n);

  // per-bin terms (lib/psy.c:571-597)
  scan.before_terms();
  LANE_BINS(k, i, i0, KPL, n) {
    float y = f[k] + offset;
    if (y < 1.f) y = 1.f;
    const float x = (float)i;
    float w = y * y;
    if (i == 0) w = (float)((double)w * .5);
    const int e = nz_swz(i);
    N[e] = w;
    X[e] = i == 0 ? w : w * x;  // sic: the reference adds w, not w*x, at bin 0 (lib/psy.c:577)
    XX[e] = w * x * x;
    Y[e] = w * y;
    XY[e] = w * x * y;
  }
  pc.mark(slot);

  // the five running sums, in index order (lib/psy.c:576-603)
  scan.scan(S, n);
  pc.mark(slot + 1);

  // line evaluation; three regimes split at the static indices i1 <= i2 (lib/psy.c:606-656): mirrored
  // lower edge, plain window, and beyond i2 the last fitted line extended.  Written without branches over
  // the lane's bins so that all their LDS reads are in flight together.
  {
    const bool any_mir = i0 < P.bark_i1 || P.bark_i2 - 1 < P.bark_i1, none = P.bark_i2 <= 0;  // wave-uniform
    LineFit L[KPL];
    if (any_mir) {
      LANE_BINS(k, i, i0, KPL, n) {
        const bool mir = (i < P.bark_i2 ? i : P.bark_i2 - 1) < P.bark_i1;
        L[k] = fit_window<true>(S, n, bk[k] & 0xffff, (int)((unsigned)bk[k] >> 16), mir);
      }
    } else {
      LANE_BINS(k, i, i0, KPL, n) L[k] = fit_window<false>(S, n, bk[k] & 0xffff, (int)((unsigned)bk[k] >> 16), false);
    }
    LANE_BINS(k, i, i0, KPL, n) {
      if (none) L[k].A = 0.f, L[k].B = 0.f, L[k].D = 1.f;
      const float x = (float)i;
      float R = (L[k].A + x * L[k].B) / L[k].D;
      if (R < 0.f) R = 0.f;
      noise[k] = R - offset;
    }
  }
  if (fixed > 0) {
    // fixed-width window pass: keep the lower of the two curves (lib/psy.c:660-703)
    const bool any_mir = i0 < P.fix_i1 || P.fix_i2 - 1 < P.fix_i1, none = P.fix_i2 <= 0;
    LineFit L[KPL];
    LANE_BINS(k, i, i0, KPL, n) {
      const int ii = i < P.fix_i2 ? i : P.fix_i2 - 1;
      const int hi = ii + fixed / 2, lo = hi - fixed;
      const bool mir = any_mir && ii < P.fix_i1;
      if (any_mir)
        L[k] = fit_window<true>(S, n, none ? 0 : nz_swz(hi), none ? 0 : nz_swz(mir ? -lo : lo), mir);
      else
        L[k] = fit_window<false>(S, n, none ? 0 : nz_swz(hi), none ? 0 : nz_swz(lo), false);
    }
    LANE_BINS(k, i, i0, KPL, n) {
      if (none) L[k].A = 0.f, L[k].B = 0.f, L[k].D = 1.f;
      const float x = (float)i;
      const float R = (L[k].A + x * L[k].B) / L[k].D;
      if (R - offset < noise[k]) noise[k] = R - offset;
    }
  }
  pc.mark(slot + 2);
}

// bark[] of this lane's bins for bark_noise_bins (see there): the raw table words, fetched early
template <int KPL, int LOGN>
VAMD_DEV void noise_bark_fetch(const PsyP &P, int *braw, int i0) {
  const int n = LOGN ? (1 << LOGN) : P.n;
  LANE_BINS(k, i, i0, KPL, n) {
    const int ii = i < P.bark_i2 ? i : P.bark_i2 - 1;
    braw[k] = P.bark[ii < 0 ? 0 : ii];
  }
}
// ... and turned into LDS word positions once the words have arrived
template <int KPL, int LOGN>
VAMD_DEV void noise_bark_edges(const PsyP &P, const int *braw, int *bk, int i0) {
  const int n = LOGN ? (1 << LOGN) : P.n;
  LANE_BINS(k, i, i0, KPL, n) {
    const int ii = i < P.bark_i2 ? i : P.bark_i2 - 1;
    const int lo = braw[k] >> 16, hi = braw[k] & 0xffff;
    bk[k] = nz_swz(hi) | (nz_swz(ii < P.bark_i1 ? -lo : lo) << 16);
  }
}

// _vp_noisemask on a block whose logmdct is already in the lanes' registers (lm[k] = bin i0 + LANE + 64k)
//   compand  noisecompand[] as a callable (level) -> value: the table itself in the test build; on the GPU a
//            cross-lane read of the copy the wave keeps one entry per lane (the index is data-dependent, the
//            team's LDS is full to the byte, and a trip to L1 at the very end of a block is exposed latency)
template <class Scan, int KPL, int LOGN, class Compand>
VAMD_DEV void noisemask_bins(const PsyP &P, const float *lm, const int *bk, float *o, float *S, const Compand &compand,
                             const Scan &scan, PhaseClock &pc, int i0) {
  const int n = LOGN ? (1 << LOGN) : P.n;
  float nz[KPL], wk[KPL];
  bark_noise_bins<Scan, KPL, LOGN>(P, lm, bk, nz, 140.f, -1, S, scan, pc, 0, i0);
  LANE_BINS(k, i, i0, KPL, n) wk[k] = lm[k] - nz[k];
  pc.mark(3);
  bark_noise_bins<Scan, KPL, LOGN>(P, wk, bk, nz, 0.f, P.noisewindowfixed, S, scan, pc, 4, i0);
  LANE_BINS(k, i, i0, KPL, n) {
    const float w = lm[k] - wk[k];
    int dB = (int)((double)nz[k] + .5);
    if (dB >= VAMD_NOISE_COMPAND_LEVELS) dB = VAMD_NOISE_COMPAND_LEVELS - 1;
    if (dB < 0) dB = 0;
    o[k] = w + compand(dB);
  }
  pc.mark(7);
}

}  // namespace vamd
