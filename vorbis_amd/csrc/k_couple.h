// k_couple.h -- _vp_couple_quantize_normalize (reference lib/psy.c:1014-1213)
// with flag_lossless (:924-935) and noise_normalize (:941-1010); SURVEY.md 8a
// row a14.  One wavefront per (stereo or mono) block.
//
// Parallel form: partitions never talk to each other (noise_normalize zeroes
// its accumulator on entry, lib/psy.c:950), and inside a partition every step is
// per bin EXCEPT noise normalisation's sort (bins >= normal_start, active only
// below q 0.4 at 44.1 kHz).  So: pass A quantises each channel per bin, pass B
// couples and re-normalises the magnitude per bin, and wherever a partition has
// noise-norm candidates one lane per partition replays the reference's ordered
// accumulate / sort / threshold walk on just those candidates.
//
// fp64 appears exactly where the reference promotes: fabs()/floor in
// flag_lossless, rint(sqrt(ve)).
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"
#include "floor1_db_table.h"

namespace vamd {

VAMD_CONST_TABLE uint32_t g_floor1_db_bits[256] = {VAMD_FLOOR1_DB_TABLE_BITS};
VAMD_DEV float floor1_fromdB(int i) { return f_from_bits(g_floor1_db_bits[i & 255]); }

// +-rint(sqrt(ve)) as the reference writes it (lib/psy.c:958-962: sqrt and rint in fp64).  The answer is the integer
// k with (k - 1/2)^2 <= ve <= (k + 1/2)^2, the even one where ve sits on a boundary; fp64 is only how the reference
// gets there.  For k < 2^22 a single-precision square root lands within one of k, k +- 1/2 are exact floats, and
// fma(h, h, -ve) has the sign of h*h - ve exactly (one rounding, and zero only for an exact zero), so two fused
// multiply-adds settle the candidate; beyond that the fp64 route is kept.
VAMD_DEV int quant_energy_f64(float ve, float r) {
  const double m = rint(sqrt((double)ve));
  return r < 0 ? (int)(-m) : (int)m;
}
// (kf: a whole number within one of the answer, below 2^22)
VAMD_DEV int quant_energy_from(float ve, float r, float kf) {
  const float hi = kf + 0.5f, lo = kf - 0.5f;
  const float th = __builtin_fmaf(hi, hi, -ve), tl = __builtin_fmaf(lo, lo, -ve);
  int k = (int)kf;
  const bool odd = (k & 1) != 0;
  if (th < 0.f || (th == 0.f && odd))
    k++;
  else if (k > 0 && (tl > 0.f || (tl == 0.f && odd)))
    k--;
  return r < 0 ? -k : k;
}
VAMD_DEV int quant_energy(float ve, float r) {
  if (!(ve < 1.7e13f)) return quant_energy_f64(ve, r);  // k >= 2^22, or not a number
  return quant_energy_from(ve, r, rintf(approx_sqrtf(ve)));  // (v_sqrt_f32: within one ulp)
}

struct CoupleLds {
  // noise-norm candidates of the current call: ve (or -1 when the bin is not a candidate)
  float *cand;  // [n2]
  float *key;   // [n2] sort key q[] = |r| (or the coupled energy)
  float *sgn;   // [n2] r (sign source for unitnorm)
  float *accp;  // [partitions] the partitions' energy budgets
};

// The ordered part of noise_normalize (lib/psy.c:976-1007) for every partition of the block at once.
// Per partition the reference accumulates the candidates' ve in index order, sorts them by key
// descending (stable: glibc's qsort is a merge sort at these sizes) and promotes them to +-1 in that
// order while the energy budget lasts: "if(acc>=thresh){out=unitnorm; acc-=1.f}else out=0".
//  * the budget: one lane per partition adds its candidates in index order (fp32, as the reference);
//  * the order: a candidate's place in the sorted list is the number of candidates with a larger key,
//    plus those with an equal key and a smaller index -- counted by the candidate's own lane;
//  * the walk: once a candidate fails the test the budget stops moving and every later one fails
//    too, so the candidate at place r is promoted iff acc0 - r >= thresh; acc0 - r in one
//    subtraction equals r subtractions of 1.f, all of which are exact (the operand is a float no
//    larger than jn/4, the result is smaller and a multiple of the operand's ulp).
//   accp  LDS [nparts] scratch
//   L.cand  ve of a candidate, -1 otherwise;  L.key  the sort key q[] of a candidate, -1.f otherwise (read as
//           integers: non-negative floats order like their bit patterns, and -1.f is below all of them)
//   accp    LDS [nparts] scratch
VAMD_DEV void noise_norm_wave(const PsyP &P, const CoupleLds &L, float *accp, int n2, int partition, int nparts, int nstart,
                              int *out) {
  const int p_first = nstart / partition;  // partitions before normal_start hold no candidates
  const bool quads = (partition & 3) == 0;
  WAVE_FOR(pp, nparts - p_first) {
    const int p = pp + p_first;
    const int b0 = p * partition, jn = partition > n2 - b0 ? n2 - b0 : partition;
    float acc = 0.f;
    if (quads && jn == partition) {
      for (int j = 0; j < jn; j += 4) {
        const F4 v = *(const F4 *)(L.cand + b0 + j);
        acc += v.x >= 0.f ? v.x : 0.f;  // (adding +0 is exact: the sum is the candidates' alone, in index order)
        acc += v.y >= 0.f ? v.y : 0.f;
        acc += v.z >= 0.f ? v.z : 0.f;
        acc += v.w >= 0.f ? v.w : 0.f;
      }
    } else {
      for (int j = 0; j < jn; j++)
        if (L.cand[b0 + j] >= 0.f) acc += L.cand[b0 + j];
    }
    accp[p] = acc;
  }
  WAVE_SYNC();
  const int *sk = (const int *)L.key;
  const int lo = p_first * partition;
  WAVE_FOR(bb, n2 - lo) {
    const int b = bb + lo;
    if (L.cand[b] >= 0.f) {
      const int p = b / partition, b0 = p * partition, jn = partition > n2 - b0 ? n2 - b0 : partition;
      const int kb = sk[b];
      // place in the stable descending sort: keys above mine, and equal keys ahead of me (s >= k <=> s + 1 > k)
      int rank = 0;
      if (quads && jn == partition) {
        for (int j = 0; j < jn; j += 4) {
          const I4 v = *(const I4 *)(sk + b0 + j);
          const int o = b0 + j;
          rank += (v.x + (o < b ? 1 : 0)) > kb ? 1 : 0;
          rank += (v.y + (o + 1 < b ? 1 : 0)) > kb ? 1 : 0;
          rank += (v.z + (o + 2 < b ? 1 : 0)) > kb ? 1 : 0;
          rank += (v.w + (o + 3 < b ? 1 : 0)) > kb ? 1 : 0;
        }
      } else {
        for (int j = 0; j < jn; j++) rank += (sk[b0 + j] + (b0 + j < b ? 1 : 0)) > kb ? 1 : 0;
      }
      const float left = accp[p] - (float)rank;
      out[b] = (double)left >= P.normal_thresh ? (int)unitnorm(L.sgn[b]) : 0;
    }
  }
}

// per-bin state of one channel before coupling (lib/psy.c:1081-1109)
struct ChanBin {
  float re, qe, fl2;  // signed energy, energy, squared floor
  int fg;             // flag_lossless
  int out;            // first quantisation (valid unless `cand` >= 0)
  float cand;         // noise-norm candidate energy, or -1
};

VAMD_DEV ChanBin chan_bin(int nzk, float m, int ilog, int b, int nstart, const CoupleP &C) {
  ChanBin r;
  r.out = 0;
  r.cand = -1.f;
  if (nzk) {
    const float f = floor1_fromdB(ilog);
    const float point = b >= C.pointlimit ? C.postpoint : C.prepoint;
    // flag_lossless, lib/psy.c:928-933: `float r = fabs(mdct)/floor` divides in fp64 and rounds to
    // fp32; for fp32 operands that double rounding is innocuous (53 >= 2*24+2), i.e. it IS the
    // correctly rounded fp32 quotient
    const float rr = fabsf(m) / f;
    r.fg = rr < point ? 0 : 1;
    r.re = m * m;
    r.qe = r.re;
    if (m < 0.f) r.re *= -1.f;
    r.fl2 = f * f;
    const float ve = r.qe / r.fl2;
    if (b < nstart || !(ve < .25f))
      r.out = quant_energy(ve, r.re);
    else
      r.cand = ve;  // flags == NULL: every small bin past normal_start is a candidate
  } else {
    r.fl2 = 1e-10f;
    r.re = 0.f;
    r.qe = 0.f;
    r.fg = 0;
  }
  return r;
}

// ---- the same bin without the divisions, where that is provably the same.  What the stage wants of |m| / f is a
// flag (the quotient against the coupling point) and an integer (the rounded root of the energy ratio): both are
// step functions of rho = |m| / f, and away from their steps a 2-instruction estimate of rho decides them.
//   reference: flag from RN32(|m| / f) = rho * (1 + d'), |d'| <= 2^-24;  k = rint(sqrt_f64(RN32(RN32(m*m) / RN32(f*f))))
//              = rint(rho * (1 + d)), |d| <= (3 * 2^-24) / 2 + O(2^-47) (three fp32 roundings under a square root)
//   here:      rr = RN32(|m| * rcp(f)), v_rcp_f32 within one ulp: rr = rho * (1 + e), |e| <= 2^-23 + 2^-24 + O(2^-47)
// so rr is within 2.25 * 2^-23 * rho of either, and wherever rr is further than `band` * rr (2^-21 = 4 * 2^-23)
// from the point and from every half-integer, the flag and rint(rr) ARE the reference's.  (Overflowing or vanishing
// intermediates: an infinite or huge rr is never sure; a vanishing one gives k = 0 on both sides.)  A lane that is not sure says so,
// and the wave then takes the whole quad through the exact forms above (one wave's quads in a hundred or so at |k| ~ 10).
// (`band` is a kernel argument so that a test can widen it until the exact path runs for most quads or for all.)
VAMD_DEV float approx_ratio_root(float num, float den) { return approx_sqrtf(num * approx_rcpf(den)); }
VAMD_DEV bool off_the_steps(float v, float kf, float band) { return fabsf(v - kf) < 0.5f - v * band; }  // (false for NaN)
// (written without branches: the or-ed conditions as integers, both coupling arms computed and one selected)
VAMD_DEV ChanBin chan_bin_sure(int nzk, float m, int ilog, int b, const CoupleP &C, float band, bool &unsure) {
  ChanBin r;
  r.out = 0;
  r.cand = -1.f;
  if (nzk) {
    const float f = floor1_fromdB(ilog);
    const float point = b >= C.pointlimit ? C.postpoint : C.prepoint;
    const float rr = fabsf(m) * approx_rcpf(f);
    r.fg = rr < point ? 0 : 1;
    const float kf = rintf(rr);
    const int at_point = !(fabsf(rr - point) > point * band), at_step = !off_the_steps(rr, kf, band);
    unsure = (int)unsure | at_point | at_step;
    r.re = m * fabsf(m);  // m*m, negated for m < 0 (a zero's sign differs for m = -0: nothing below tells them apart)
    r.qe = fabsf(r.re);
    r.fl2 = f * f;
    r.out = (int)copysignf(kf, m);  // "r < 0 ? -k : k": k != 0 only where m*m is not zero either
  } else {
    r.fl2 = 1e-10f;
    r.re = 0.f;
    r.qe = 0.f;
    r.fg = 0;
  }
  return r;
}

// out[j]*out[j] as the reference's x86-64 build computes it (lib/psy.c:985: an int product, converted to float
// afterwards): past |out| = 46340 -- spectra 93 dB over full scale -- the product wraps modulo 2^32, the "energy" can
// come out negative and the bin then counts as a noise-normalisation candidate.  Signed overflow is undefined in C,
// so the wrap is spelt out in unsigned arithmetic here (a compiler may otherwise assume the square is non-negative);
// the soak holds such blocks (tests/soak_lib.py kind 10).
VAMD_DEV float int_square_as_float(int v) { return (float)(int)((unsigned int)v * (unsigned int)v); }

// one bin of the coupling step (lib/psy.c:1129-1196) followed by the magnitude's
// re-normalisation (noise_normalize with flags); M/A are updated in place, iM/iA
// are the integers quantised so far.  Returns the magnitude's noise-norm
// candidate energy or -1.
VAMD_DEV void couple_bin_mix(ChanBin &M, ChanBin &A, int &iM, int &iA, int b, int nstart, const CoupleP &C) {
  if (b < C.sliding_lowpass) {
    if (M.fg || A.fg) {
      // lossless: square-polar coupling of the already quantised integers
      // (float)(fabs((double)M.re) + fabs((double)A.re)): the sum of two floats rounded through fp64 is the fp32 sum
      // (53 >= 2 * 24 + 2, as for the quotient in chan_bin)
      M.re = fabsf(M.re) + fabsf(A.re);
      M.qe = M.qe + A.qe;
      M.fg = A.fg = 1;
      const int a = iM, bb = iA;
      const int aA = a < 0 ? -a : a, aB = bb < 0 ? -bb : bb;
      if (aA > aB) {
        iA = (a > 0 ? a - bb : bb - a);
      } else {
        iA = (bb > 0 ? a - bb : bb - a);
        iM = bb;
      }
      if (iA >= (iM < 0 ? -iM : iM) * 2) {
        iA = -iA;
        iM = -iM;
      }
    } else {
      // lossy point coupling
      if (b < C.pointlimit) {
        M.re += A.re;
        M.qe = (float)fabs((double)M.re);
      } else {
        const float e = fabsf(M.re) + fabsf(A.re);  // (as above)
        M.qe = e;
        M.re = (M.re + A.re < 0) ? -e : e;
      }
      A.re = A.qe = 0.f;
      A.fg = 1;
      iA = 0;
    }
  }
  else if (b >= nstart) {
    // past the sliding lowpass nothing is coupled, and the magnitude is renormalised from what the
    // per-channel noise_normalize left in quant[]: out*out*floor for a final value, floor for a
    // promoted candidate, 0 for a dropped one (lib/psy.c:985,998-1003) -- all three are iM*iM*floor.
    // (Only bitrate-managed candidates 0..6 put the lowpass inside the block.)
    M.qe = int_square_as_float(iM) * M.fl2;
  }
  M.fl2 = A.fl2 = M.fl2 + A.fl2;
}
VAMD_DEV float couple_bin(ChanBin &M, ChanBin &A, int &iM, int &iA, int b, int nstart, const CoupleP &C) {
  couple_bin_mix(M, A, iM, iA, b, nstart, C);
  float cand = -1.f;
  if (!M.fg) {
    const float ve = M.qe / M.fl2;
    if (b < nstart || !(ve < .25f && b >= C.pointlimit))
      iM = quant_energy(ve, M.re);
    else
      cand = ve;
  }
  return cand;
}

// couple_bin where no bin is a noise-normalisation candidate, the re-quantisation by estimate (see chan_bin_sure:
// s = sqrt(qe * rcp(fl2)) is within (2^-23 + 2^-24) / 2 + 2^-23 of the root of the true ratio, the reference's fp64
// root of the rounded quotient within 2^-25 of it)
VAMD_DEV void couple_bin_sure(ChanBin &M, ChanBin &A, int &iM, int &iA, int b, const CoupleP &C, float band, bool &unsure) {
  const float fl2 = M.fl2 + A.fl2;
  M.fl2 = A.fl2 = fl2;
  float re = M.re, qe = M.qe;
  int fg = M.fg;
  if (b < C.sliding_lowpass) {
    // lossless arm (lib/psy.c:1141-1166): the integers only -- a flagged magnitude is not re-quantised, so nothing
    // reads its energies again
    const int a = iM, bb = iA;
    const int aA = a < 0 ? -a : a, aB = bb < 0 ? -bb : bb, d1 = a - bb, d2 = bb - a;
    const bool gt = aA > aB;
    int lA = (gt ? a : bb) > 0 ? d1 : d2;
    int lM = gt ? a : bb;
    const bool flip = lA >= (gt ? aA : aB) * 2;
    lA = flip ? -lA : lA;
    lM = flip ? -lM : lM;
    // lossy arm (:1167-1190)
    const float sum = M.re + A.re, e = fabsf(M.re) + fabsf(A.re);
    const bool below = b < C.pointlimit;
    re = below ? sum : (sum < 0 ? -e : e);
    qe = below ? fabsf(sum) : e;
    fg = M.fg | A.fg;
    iM = fg ? lM : iM;
    iA = fg ? lA : 0;
  }
  const float s = approx_ratio_root(qe, fl2);
  const float kf = rintf(s);
  const int at_step = !off_the_steps(s, kf, band);
  unsure = (int)unsure | (at_step & (fg ^ 1));
  iM = fg ? iM : (int)copysignf(kf, re);  // (k != 0 only where re is not a zero)
  M.re = re, M.qe = qe, M.fg = fg;
}

// The input domain's integer edge, first half (include/vorbis_amd.h, vamd_params.h): every value this stage writes is a
// float -> int conversion (lib/psy.c:958-962: defined below 2^31, VAMD_QUANT_LIMIT_INT), and where noise normalisation is
// at work -- from bin `nstart` on -- it is squared in an int (:985: defined up to VAMD_QUANT_LIMIT_SQUARE).  Held against
// the FINAL values: in a coupling step the magnitude keeps the larger of its two inputs and the angle their difference
// (:1141-1166), so no value of a block's first quantisation exceeds the largest final one, and a block none of whose
// channels is flagged has met neither hazard.  (An out-of-range float converts to INT_MIN / INT_MAX here, which the
// bound catches.)  The second half -- the residue search's own bound on the positions it codes -- is k_residue's.
// A lane's running extremes for one channel: one three-operand minimum and one maximum per pair of values.
struct QuantSpan {
  int lo = 0, hi = 0;
  VAMD_MEM void take(int v) {
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  VAMD_MEM void take4(const int *v) {  // (the compiler pairs these into v_min3_i32 / v_max3_i32)
    for (int c = 0; c < 4; c++) take(v[c]);
  }
  VAMD_MEM bool beyond(int lim) const { return hi > lim || lo < -lim; }
};
// The ordered paths (noise normalisation's sort, layouts beyond stereo) leave their final values in HBM after a
// wave-wide sync: one more pass over them, a lane's share of every channel (their time is the ordered walks').
VAMD_DEV unsigned quant_span_reread(int ch, int n2, int *const *iwork, int nstart) {
  unsigned over = 0;
  for (int k = 0; k < ch; k++) {
    QuantSpan below, from;
    WAVE_FOR(b, n2) {
      const int v = iwork[k][b];
      if (b < nstart) below.take(v); else from.take(v);
    }
    if (below.beyond(VAMD_QUANT_LIMIT_INT) || from.beyond(VAMD_QUANT_LIMIT_SQUARE)) over |= 1u << k;
  }
  return over;
}

// The stage where nothing is ordered (noise normalisation inactive in this block size, e.g. q >= 0.4 at 44.1 kHz):
// each lane takes quads of bins straight through quantise -> couple -> re-normalise with one 16-byte load per input
// tensor, by estimate first (chan_bin_sure) and exactly where some lane of the wave is not sure.
//   ALL   compile-time promise that both channels exist, have a floor and are coupled (the flags are then not read)
template <bool ALL>
VAMD_DEV void couple_quads(const CoupleP &C, int n2, const float *__restrict__ mdctM, const float *__restrict__ mdctA,
                           const ilog_t *__restrict__ ilogM, const ilog_t *__restrict__ ilogA, int *__restrict__ iworkM,
                           int *__restrict__ iworkA, int nzM_in, int nzA_in, bool two_in, bool coupled_in, int nstart,
                           float band, QuantSpan &spM, QuantSpan &spA) {
  const int nzM = ALL ? 1 : nzM_in, nzA = ALL ? 1 : nzA_in;
  const bool two = ALL ? true : two_in, coupled = ALL ? true : coupled_in;
  // (two quads in flight, not WAVE_FOR's four: at four the kernel needs 110 VGPRs and the SIMD holds four waves)
  // (the quads are dealt over the whole TEAM: for a handful of blocks the launch gives a block four waves, which
  // brings a lone block's 18 us down to 6)
#pragma unroll 1
  TEAM_FOR(q, n2 >> 2) {
    float m0[4], m1[4];
    int l0[4], l1[4], o0[4], o1[4];
    f4_get(((const F4 *)mdctM)[q], m0);
    const unsigned int t0 = ((const unsigned int *)ilogM)[q];  // (ilog_t: a byte per bin)
    l0[0] = t0 & 0xff; l0[1] = (t0 >> 8) & 0xff; l0[2] = (t0 >> 16) & 0xff; l0[3] = t0 >> 24;
    if (two) {
      f4_get(((const F4 *)mdctA)[q], m1);
      const unsigned int t1 = ((const unsigned int *)ilogA)[q];
      l1[0] = t1 & 0xff; l1[1] = (t1 >> 8) & 0xff; l1[2] = (t1 >> 16) & 0xff; l1[3] = t1 >> 24;
    }
    bool unsure = false;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int b = (q << 2) + c;
      ChanBin M = chan_bin_sure(nzM, m0[c], l0[c], b, C, band, unsure);
      int iM = M.out, iA = 0;
      if (two) {
        ChanBin A = chan_bin_sure(nzA, m1[c], l1[c], b, C, band, unsure);
        iA = A.out;
        if (coupled) couple_bin_sure(M, A, iM, iA, b, C, band, unsure);
      }
      o0[c] = iM;
      o1[c] = iA;
    }
    I4 w0, w1;
    w0.x = o0[0]; w0.y = o0[1]; w0.z = o0[2]; w0.w = o0[3];
    ((I4 *)iworkM)[q] = w0;
    spM.take4(o0);
    if (two) {
      w1.x = o1[0]; w1.y = o1[1]; w1.z = o1[2]; w1.w = o1[3];
      ((I4 *)iworkA)[q] = w1;
      spA.take4(o1);
    }
    if (wave_any(unsure)) {
      // some bin of the wave's 64 quads sits on a step: the reference's own arithmetic, a bin at a time from the
      // tensors again and over what was just written -- the rare path holds no registers of the usual one that way
#pragma unroll 1
      for (int c = 0; c < 4; c++) {
        const int b = (q << 2) + c;
        ChanBin M = chan_bin(nzM, mdctM[b], ilogM[b], b, nstart, C);
        int iM = M.out, iA = 0;
        if (two) {
          ChanBin A = chan_bin(nzA, mdctA[b], ilogA[b], b, nstart, C);
          iA = A.out;
          if (coupled) couple_bin(M, A, iM, iA, b, nstart, C);
          iworkA[b] = iA;
          spA.take(iA);
        }
        iworkM[b] = iM;
        spM.take(iM);
      }
    }
  }
}

// mdct[k]      HBM [n2]  post-M1 spectrum of channel k
// ilogmask[k]  HBM [n2]  integer floor curve (floor1_encode's output)
// iwork[k]     HBM [n2]  out: quantised (and coupled) residue
// nonzero      [ch] in: floor1_encode's return per channel; out: after the coupling fix-up
// (one or two channels, at most one coupling step: every stereo and mono setup)
// NORM = false: the caller knows noise normalisation is inactive for this size class (the launch picks the
// instantiation): the ordered general path below is then not even compiled in, which halves the registers.
//   over  the input domain's integer edge (QuantSpan): bit k of the returned `over` is set in a lane that wrote a value
//         beyond its bound (QuantSpan above) for channel k (the caller ORs the lanes)
template <bool NORM = true>
VAMD_DEV void couple_block(const CoupleP &C_set, const PsyP &P, int n2, const float *const *mdct,
                           const ilog_t *const *ilogmask, int *const *iwork, int *nonzero, const CoupleLds &L,
                           PhaseClock &pc, float band = VAMD_COUPLE_BAND, unsigned *over = nullptr) {
  const CoupleP C = C_set;  // (by value: the fields in scalar registers, not behind the set's run-time index)
  const int ch = C.ch;
  const int partition = P.normal_p ? P.normal_partition : 16;
  const int nstart = P.normal_p ? P.normal_start : 0x7fffffff;  // first bin subject to noise norm
  const bool norm_active = NORM && nstart < n2;
  const int nparts = (n2 + partition - 1) / partition;
  int nz[VAMD_MAX_CH];
  for (int k = 0; k < ch; k++) nz[k] = nonzero[k];
  const bool coupled = C.coupling_steps == 1 && (nz[C.mag[0]] || nz[C.ang[0]]);

  if (!norm_active) {
    // Common case: nothing is ordered (couple_quads above)
    const int Mi = C.coupling_steps == 1 ? C.mag[0] : 0, Ai = C.coupling_steps == 1 ? C.ang[0] : (ch > 1 ? 1 : 0);
    // (the usual block -- two channels, both with a floor, coupled -- gets a loop compiled for exactly that: the
    // wave-uniform tests on nonzero[] and on the channel count otherwise stand between every two bins)
    QuantSpan spM, spA;
    if (ch > 1 && coupled && nz[Mi] && nz[Ai])
      couple_quads<true>(C, n2, mdct[Mi], mdct[Ai], ilogmask[Mi], ilogmask[Ai], iwork[Mi], iwork[Ai], 1, 1, true, true,
                         nstart, band, spM, spA);
    else
      couple_quads<false>(C, n2, mdct[Mi], mdct[Ai], ilogmask[Mi], ilogmask[Ai], iwork[Mi], iwork[Ai], nz[Mi], nz[Ai],
                          ch > 1, coupled, nstart, band, spM, spA);
    // (this path runs where noise normalisation is not at work: no value is squared)
    if (over) *over = (spM.beyond(VAMD_QUANT_LIMIT_INT) ? 1u << Mi : 0u) | (ch > 1 && spA.beyond(VAMD_QUANT_LIMIT_INT) ? 1u << Ai : 0u);
    pc.mark(0);
    if (coupled) nz[C.mag[0]] = nz[C.ang[0]] = 1;  // lib/psy.c:1204-1212
    for (int k = 0; k < ch; k++) nonzero[k] = nz[k];
    pc.mark(1);
    return;
  }
  if (!NORM) return;

  // ---- general path: noise normalisation's ordered sort may touch any partition.
  // per channel: floor lookup, lossless flags, energies, first quantisation
  for (int k = 0; k < ch; k++) {
    WAVE_FOR(b, n2) {
      const ChanBin B = chan_bin(nz[k], nz[k] ? mdct[k][b] : 0.f, nz[k] ? ilogmask[k][b] : 0, b, nstart, C);
      iwork[k][b] = B.out;
      L.cand[b] = B.cand;
      L.key[b] = B.cand >= 0.f ? B.qe : -1.f;
      L.sgn[b] = B.re;
    }
    WAVE_SYNC();
    if (nz[k]) noise_norm_wave(P, L, L.accp, n2, partition, nparts, nstart, iwork[k]);
    WAVE_SYNC_GLOBAL();  // iwork[] changes hands between lanes through HBM
  }
  pc.mark(0);
  // ---- coupling (one step: magnitude Mi, angle Ai), lib/psy.c:1111-1201
  if (coupled) {
    const int Mi = C.mag[0], Ai = C.ang[0];
    WAVE_FOR(b, n2) {
      // rebuild the per-bin state the first pass had (cheaper than keeping it in LDS)
      ChanBin M = chan_bin(nz[Mi], nz[Mi] ? mdct[Mi][b] : 0.f, nz[Mi] ? ilogmask[Mi][b] : 0, b, nstart, C);
      ChanBin A = chan_bin(nz[Ai], nz[Ai] ? mdct[Ai][b] : 0.f, nz[Ai] ? ilogmask[Ai][b] : 0, b, nstart, C);
      int iM = iwork[Mi][b], iA = iwork[Ai][b];
      const float cand = couple_bin(M, A, iM, iA, b, nstart, C);
      iwork[Mi][b] = iM;
      iwork[Ai][b] = iA;
      L.cand[b] = cand;
      L.key[b] = cand >= 0.f ? M.qe : -1.f;
      L.sgn[b] = M.re;
    }
    WAVE_SYNC();
    noise_norm_wave(P, L, L.accp, n2, partition, nparts, nstart, iwork[Mi]);
    WAVE_SYNC_GLOBAL();
    nz[Mi] = nz[Ai] = 1;  // lib/psy.c:1204-1212
  }
  if (over) *over = quant_span_reread(ch, n2, iwork, nstart);
  for (int k = 0; k < ch; k++) nonzero[k] = nz[k];
  pc.mark(1);
}

// ---- any channel count, any number of coupling steps (the 5.1 layout: six channels, four steps, the
// left channel the magnitude of three of them; lib/psy.c:1111-1201 "depth>1 coupling").
// Per bin the steps are a fixed sequence of couple_bin() calls on the channels' running state (signed
// energy, energy, squared floor, lossless flag, the integer quantised so far); between steps that
// state rests in a per-unit workspace in HBM, one row per channel, so a step touches only its
// magnitude and angle rows, addressed by the step's channel numbers.  The only thing a bin needs
// from its neighbours is noise normalisation's outcome for the candidates of its partition
// (noise_norm_wave); what the reference then leaves in quant[] past normal_start -- out*out*floor,
// or floor / 0 for a promoted / dropped candidate, all three out*out*floor (lib/psy.c:985,998-1003) --
// is applied when the channel is next read (`pend`).
//   state  HBM [4][ch][n2]: re, qe, fl2 (floats) and fg (ints)
struct CoupleState {
  float *re, *qe, *fl2;
  int *fg;
};

VAMD_DEV void couple_block_general(const CoupleP &C, const PsyP &P, int n2, const float *const *mdct,
                                   const ilog_t *const *ilogmask, int *const *iwork, int *nonzero, const CoupleLds &L,
                                   const CoupleState &S, PhaseClock &pc, unsigned *over = nullptr) {
  const int ch = C.ch, steps = C.coupling_steps;
  const int partition = P.normal_p ? P.normal_partition : 16;
  const int nstart = P.normal_p ? P.normal_start : 0x7fffffff;
  const bool norm_active = nstart < n2;
  const int nparts = (n2 + partition - 1) / partition;
  int nz[VAMD_MAX_CH], nzl[VAMD_MAX_CH], pend[VAMD_MAX_CH];
  for (int k = 0; k < ch; k++) nz[k] = nzl[k] = nonzero[k];

  // prefill: per channel, first quantisation (and its ordered part)
  for (int k = 0; k < ch; k++) {
    WAVE_FOR(b, n2) {
      const ChanBin B = chan_bin(nz[k], nz[k] ? mdct[k][b] : 0.f, nz[k] ? ilogmask[k][b] : 0, b, nstart, C);
      iwork[k][b] = B.out;
      S.re[k * n2 + b] = B.re;
      S.qe[k * n2 + b] = B.qe;
      S.fl2[k * n2 + b] = B.fl2;
      S.fg[k * n2 + b] = B.fg;
      if (norm_active) {
        L.cand[b] = B.cand;
        L.key[b] = B.cand >= 0.f ? B.qe : -1.f;
        L.sgn[b] = B.re;
      }
    }
    if (norm_active) {
      WAVE_SYNC();
      if (nz[k]) noise_norm_wave(P, L, L.accp, n2, partition, nparts, nstart, iwork[k]);
    }
    WAVE_SYNC_GLOBAL();
    pend[k] = nz[k] ? 1 : 0;  // quant[] = out*out*floor past normal_start, flags or not (flags == NULL)
  }
  pc.mark(0);
  for (int t = 0; t < steps; t++) {
    const int Mi = C.mag[t], Ai = C.ang[t];
    if (!(nzl[Mi] || nzl[Ai])) continue;  // lib/psy.c:1125
    nzl[Mi] = nzl[Ai] = 1;
    const int pm = pend[Mi], pa = pend[Ai];
    WAVE_FOR(b, n2) {
      ChanBin M, A;
      M.re = S.re[Mi * n2 + b], M.qe = S.qe[Mi * n2 + b], M.fl2 = S.fl2[Mi * n2 + b], M.fg = S.fg[Mi * n2 + b];
      A.re = S.re[Ai * n2 + b], A.qe = S.qe[Ai * n2 + b], A.fl2 = S.fl2[Ai * n2 + b], A.fg = S.fg[Ai * n2 + b];
      int iM = iwork[Mi][b], iA = iwork[Ai][b];
      if (b >= nstart) {
        if (pm == 1 || (pm == 2 && !M.fg)) M.qe = int_square_as_float(iM) * M.fl2;
        if (pa == 1 || (pa == 2 && !A.fg)) A.qe = int_square_as_float(iA) * A.fl2;
      }
      const float cand = couple_bin(M, A, iM, iA, b, nstart, C);
      S.re[Mi * n2 + b] = M.re, S.qe[Mi * n2 + b] = M.qe, S.fl2[Mi * n2 + b] = M.fl2, S.fg[Mi * n2 + b] = M.fg;
      S.re[Ai * n2 + b] = A.re, S.qe[Ai * n2 + b] = A.qe, S.fl2[Ai * n2 + b] = A.fl2, S.fg[Ai * n2 + b] = A.fg;
      iwork[Mi][b] = iM;
      iwork[Ai][b] = iA;
      if (norm_active) {
        L.cand[b] = cand;
        L.key[b] = cand >= 0.f ? M.qe : -1.f;
        L.sgn[b] = M.re;
      }
    }
    if (norm_active) {
      WAVE_SYNC();
      noise_norm_wave(P, L, L.accp, n2, partition, nparts, nstart, iwork[Mi]);
    }
    WAVE_SYNC_GLOBAL();
    pend[Mi] = 2;  // noise_normalize with flags: unflagged bins only
    pend[Ai] = 0;
  }
  pc.mark(1);
  if (over) *over = quant_span_reread(ch, n2, iwork, nstart);
  for (int t = 0; t < steps; t++)  // lib/psy.c:1204-1212, in step order
    if (nz[C.mag[t]] || nz[C.ang[t]]) nz[C.mag[t]] = nz[C.ang[t]] = 1;
  for (int k = 0; k < ch; k++) nonzero[k] = nz[k];
}

}  // namespace vamd
