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

#if VAMD_GPU
__constant__ uint32_t g_floor1_db_bits[256] = {VAMD_FLOOR1_DB_TABLE_BITS};
#else
static const uint32_t g_floor1_db_bits[256] = {VAMD_FLOOR1_DB_TABLE_BITS};
#endif
VAMD_DEV float floor1_fromdB(int i) { return f_from_bits(g_floor1_db_bits[i & 255]); }

// +-rint(sqrt(ve)) as the reference writes it (lib/psy.c:958-962): sqrt and rint in fp64
VAMD_DEV int quant_energy(float ve, float r) {
  const double m = rint(sqrt((double)ve));
  return r < 0 ? (int)(-m) : (int)m;
}

struct CoupleLds {
  // noise-norm candidates of the current call: ve (or -1 when the bin is not a candidate)
  float *cand;  // [n2]
  float *key;   // [n2] sort key q[] = |r| (or the coupled energy)
  float *sgn;   // [n2] r (sign source for unitnorm)
};

// The ordered part of noise_normalize for one partition [b0, b0+jn): accumulate
// candidates' ve in index order, sort them by key descending (stable: glibc's
// qsort is a merge sort at these sizes), then promote to +-1 while the energy
// budget lasts (lib/psy.c:993-1007).
VAMD_DEV void noise_norm_partition(const PsyP &P, const CoupleLds &L, int b0, int jn, int *out) {
  float acc = 0.f;
  int count = 0;
  for (int j = 0; j < jn; j++)
    if (L.cand[b0 + j] >= 0.f) {
      acc += L.cand[b0 + j];
      count++;
    }
  // selection in sorted order without materialising the permutation: repeatedly
  // take the largest remaining key, earliest index first among equals
  for (int t = 0; t < count; t++) {
    int best = -1;
    float bk = 0.f;
    for (int j = 0; j < jn; j++) {
      const int b = b0 + j;
      if (L.cand[b] >= 0.f && (best < 0 || L.key[b] > bk)) {
        best = b;
        bk = L.key[b];
      }
    }
    if ((double)acc >= P.normal_thresh) {
      out[best] = (int)unitnorm(L.sgn[best]);
      acc -= 1.f;
    } else {
      out[best] = 0;
    }
    L.cand[best] = -1.f;  // consumed
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

// one bin of the coupling step (lib/psy.c:1129-1196) followed by the magnitude's
// re-normalisation (noise_normalize with flags); M/A are updated in place, iM/iA
// are the integers quantised so far.  Returns the magnitude's noise-norm
// candidate energy or -1.
VAMD_DEV float couple_bin(ChanBin &M, ChanBin &A, int &iM, int &iA, int b, int nstart, const CoupleP &C) {
  if (b < C.sliding_lowpass) {
    if (M.fg || A.fg) {
      // lossless: square-polar coupling of the already quantised integers
      M.re = (float)(fabs((double)M.re) + fabs((double)A.re));
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
        const float e = (float)(fabs((double)M.re) + fabs((double)A.re));
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
    M.qe = (float)(iM * iM) * M.fl2;
  }
  M.fl2 = A.fl2 = M.fl2 + A.fl2;
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

// mdct[k]      HBM [n2]  post-M1 spectrum of channel k
// ilogmask[k]  HBM [n2]  integer floor curve (floor1_encode's output)
// iwork[k]     HBM [n2]  out: quantised (and coupled) residue
// nonzero      [ch] in: floor1_encode's return per channel; out: after the coupling fix-up
// (one or two channels, at most one coupling step: every stereo and mono setup)
VAMD_DEV void couple_block(const CoupleP &C, const PsyP &P, int n2, const float *const *mdct,
                           const int *const *ilogmask, int *const *iwork, int *nonzero, const CoupleLds &L,
                           PhaseClock &pc) {
  const int ch = C.ch;
  const int partition = P.normal_p ? P.normal_partition : 16;
  const int nstart = P.normal_p ? P.normal_start : 0x7fffffff;  // first bin subject to noise norm
  const bool norm_active = nstart < n2;
  const int nparts = (n2 + partition - 1) / partition;
  int nz[VAMD_MAX_CH];
  for (int k = 0; k < ch; k++) nz[k] = nonzero[k];
  const bool coupled = C.coupling_steps == 1 && (nz[C.mag[0]] || nz[C.ang[0]]);

  if (!norm_active) {
    // Common case (noise normalisation inactive in this block size, e.g. q >= 0.4 at
    // 44.1 kHz): nothing is ordered, so each lane takes quads of bins straight through
    // quantise -> couple -> re-normalise with one 16-byte load per input tensor.
    const int Mi = C.coupling_steps == 1 ? C.mag[0] : 0, Ai = C.coupling_steps == 1 ? C.ang[0] : (ch > 1 ? 1 : 0);
    WAVE_FOR(q, n2 >> 2) {
      float m0[4], m1[4];
      int l0[4], l1[4], o0[4], o1[4];
      f4_get(((const F4 *)mdct[Mi])[q], m0);
      const I4 t0 = ((const I4 *)ilogmask[Mi])[q];
      l0[0] = t0.x; l0[1] = t0.y; l0[2] = t0.z; l0[3] = t0.w;
      if (ch > 1) {
        f4_get(((const F4 *)mdct[Ai])[q], m1);
        const I4 t1 = ((const I4 *)ilogmask[Ai])[q];
        l1[0] = t1.x; l1[1] = t1.y; l1[2] = t1.z; l1[3] = t1.w;
      }
#if VAMD_GPU
#pragma unroll
#endif
      for (int c = 0; c < 4; c++) {
        const int b = (q << 2) + c;
        ChanBin M = chan_bin(nz[Mi], m0[c], l0[c], b, nstart, C);
        int iM = M.out, iA = 0;
        if (ch > 1) {
          ChanBin A = chan_bin(nz[Ai], m1[c], l1[c], b, nstart, C);
          iA = A.out;
          if (coupled) couple_bin(M, A, iM, iA, b, nstart, C);
        }
        o0[c] = iM;
        o1[c] = iA;
      }
      I4 w0, w1;
      w0.x = o0[0]; w0.y = o0[1]; w0.z = o0[2]; w0.w = o0[3];
      ((I4 *)iwork[Mi])[q] = w0;
      if (ch > 1) {
        w1.x = o1[0]; w1.y = o1[1]; w1.z = o1[2]; w1.w = o1[3];
        ((I4 *)iwork[Ai])[q] = w1;
      }
    }
    pc.mark(0);
    if (coupled) nz[C.mag[0]] = nz[C.ang[0]] = 1;  // lib/psy.c:1204-1212
    for (int k = 0; k < ch; k++) nonzero[k] = nz[k];
    pc.mark(1);
    return;
  }

  // ---- general path: noise normalisation's ordered sort may touch any partition.
  // per channel: floor lookup, lossless flags, energies, first quantisation
  for (int k = 0; k < ch; k++) {
    WAVE_FOR(b, n2) {
      const ChanBin B = chan_bin(nz[k], nz[k] ? mdct[k][b] : 0.f, nz[k] ? ilogmask[k][b] : 0, b, nstart, C);
      iwork[k][b] = B.out;
      L.cand[b] = B.cand;
      L.key[b] = B.qe;
      L.sgn[b] = B.re;
    }
    if (nz[k]) {
      WAVE_SYNC_GLOBAL();  // iwork[] changes hands between lanes through HBM
      WAVE_FOR(p, nparts) {
        const int b0 = p * partition;
        const int jn = partition > n2 - b0 ? n2 - b0 : partition;
        noise_norm_partition(P, L, b0, jn, iwork[k]);
      }
      WAVE_SYNC_GLOBAL();
    }
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
      L.key[b] = M.qe;
      L.sgn[b] = M.re;
    }
    WAVE_SYNC_GLOBAL();
    WAVE_FOR(p, nparts) {
      const int b0 = p * partition;
      const int jn = partition > n2 - b0 ? n2 - b0 : partition;
      noise_norm_partition(P, L, b0, jn, iwork[Mi]);
    }
    WAVE_SYNC_GLOBAL();
    nz[Mi] = nz[Ai] = 1;  // lib/psy.c:1204-1212
  }
  for (int k = 0; k < ch; k++) nonzero[k] = nz[k];
  pc.mark(1);
}

// ---- any channel count, any number of coupling steps (the 5.1 layout: six channels, four steps, the
// left channel the magnitude of three of them; lib/psy.c:1111-1201 "depth>1 coupling").
// Per bin the steps are a fixed sequence of couple_bin() calls on the channels' running state; the only
// thing one bin needs from its neighbours is the outcome of noise normalisation's sort for bins that
// were candidates in an earlier step.  So every step is: per bin, REPLAY the prefill and the earlier
// steps (taking candidates' outcomes from a snapshot of what the sort left), run this step, hand the
// new candidates to the per-partition sort, snapshot.  Without noise normalisation in the block there
// are no candidates and one replay of all steps does everything.
//   pre   [ch][n2]   the channels' integers after the prefill (and its sort)
//   snap  [steps][n2] the magnitude channel's integers after each step's sort
struct CoupleGeneralLds {
  CoupleLds L;
  int *pre, *snap;
};

// state of every channel of bin b after the prefill and coupling steps [0, upto]; returns step upto's
// noise-norm candidate energy for its magnitude channel (or -1)
VAMD_DEV float couple_replay(const CoupleP &C, int ch, const int *nz0, int b, int n2, int nstart,
                             const float *const *mdct, const int *const *ilogmask, const int *pre, const int *snap,
                             int upto, ChanBin *st, int *io) {
  int nzl[VAMD_MAX_CH];
  for (int k = 0; k < ch; k++) {
    nzl[k] = nz0[k];
    st[k] = chan_bin(nz0[k], nz0[k] ? mdct[k][b] : 0.f, nz0[k] ? ilogmask[k][b] : 0, b, nstart, C);
    io[k] = pre ? pre[k * n2 + b] : st[k].out;
    // what noise_normalize leaves in quant[] past normal_start: out*out*floor, or floor / 0 for a promoted /
    // dropped candidate -- all three are out*out*floor (lib/psy.c:985,998-1003)
    if (nz0[k] && b >= nstart) st[k].qe = (float)(io[k] * io[k]) * st[k].fl2;
  }
  float cand = -1.f;
  for (int t = 0; t <= upto; t++) {
    const int Mi = C.mag[t], Ai = C.ang[t];
    if (!(nzl[Mi] || nzl[Ai])) continue;  // lib/psy.c:1125
    nzl[Mi] = nzl[Ai] = 1;
    cand = couple_bin(st[Mi], st[Ai], io[Mi], io[Ai], b, nstart, C);
    if (t < upto) {
      if (cand >= 0.f) io[Mi] = snap[t * n2 + b];
      if (b >= nstart && !st[Mi].fg) st[Mi].qe = (float)(io[Mi] * io[Mi]) * st[Mi].fl2;
    }
  }
  return cand;
}

VAMD_DEV void couple_block_general(const CoupleP &C, const PsyP &P, int n2, const float *const *mdct,
                                   const int *const *ilogmask, int *const *iwork, int *nonzero,
                                   const CoupleGeneralLds &G, PhaseClock &pc) {
  const int ch = C.ch, steps = C.coupling_steps;
  const CoupleLds &L = G.L;
  const int partition = P.normal_p ? P.normal_partition : 16;
  const int nstart = P.normal_p ? P.normal_start : 0x7fffffff;
  const bool norm_active = nstart < n2;
  const int nparts = (n2 + partition - 1) / partition;
  int nz[VAMD_MAX_CH];
  for (int k = 0; k < ch; k++) nz[k] = nonzero[k];

  if (!norm_active) {
    WAVE_FOR(b, n2) {
      ChanBin st[VAMD_MAX_CH];
      int io[VAMD_MAX_CH];
      couple_replay(C, ch, nz, b, n2, nstart, mdct, ilogmask, nullptr, nullptr, steps - 1, st, io);
      for (int k = 0; k < ch; k++) iwork[k][b] = io[k];
    }
  } else {
    // prefill: per channel, first quantisation and its sort
    for (int k = 0; k < ch; k++) {
      WAVE_FOR(b, n2) {
        const ChanBin B = chan_bin(nz[k], nz[k] ? mdct[k][b] : 0.f, nz[k] ? ilogmask[k][b] : 0, b, nstart, C);
        iwork[k][b] = B.out;
        L.cand[b] = B.cand;
        L.key[b] = B.qe;
        L.sgn[b] = B.re;
      }
      WAVE_SYNC_GLOBAL();
      if (nz[k]) {
        WAVE_FOR(p, nparts) {
          const int b0 = p * partition;
          noise_norm_partition(P, L, b0, partition > n2 - b0 ? n2 - b0 : partition, iwork[k]);
        }
        WAVE_SYNC_GLOBAL();
      }
      WAVE_FOR(b, n2) G.pre[k * n2 + b] = iwork[k][b];
      WAVE_SYNC();
    }
    pc.mark(0);
    int nzl[VAMD_MAX_CH];
    for (int k = 0; k < ch; k++) nzl[k] = nz[k];
    for (int t = 0; t < steps; t++) {
      const int Mi = C.mag[t], Ai = C.ang[t];
      if (!(nzl[Mi] || nzl[Ai])) continue;
      nzl[Mi] = nzl[Ai] = 1;
      WAVE_FOR(b, n2) {
        ChanBin st[VAMD_MAX_CH];
        int io[VAMD_MAX_CH];
        const float cand = couple_replay(C, ch, nz, b, n2, nstart, mdct, ilogmask, G.pre, G.snap, t, st, io);
        iwork[Mi][b] = io[Mi];
        iwork[Ai][b] = io[Ai];
        L.cand[b] = cand;
        L.key[b] = st[Mi].qe;
        L.sgn[b] = st[Mi].re;
      }
      WAVE_SYNC_GLOBAL();
      WAVE_FOR(p, nparts) {
        const int b0 = p * partition;
        noise_norm_partition(P, L, b0, partition > n2 - b0 ? n2 - b0 : partition, iwork[Mi]);
      }
      WAVE_SYNC_GLOBAL();
      WAVE_FOR(b, n2) G.snap[t * n2 + b] = iwork[Mi][b];
      WAVE_SYNC();
    }
  }
  pc.mark(1);
  for (int t = 0; t < steps; t++)  // lib/psy.c:1204-1212, in step order
    if (nz[C.mag[t]] || nz[C.ang[t]]) nz[C.mag[t]] = nz[C.ang[t]] = 1;
  for (int k = 0; k < ch; k++) nonzero[k] = nz[k];
}

}  // namespace vamd
