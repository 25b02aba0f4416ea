// k_noise.h -- _vp_noisemask (reference lib/psy.c:706-752) with its two
// bark_noise_hybridmp passes (lib/psy.c:547-704); SURVEY.md 8a row a9.  One
// wavefront per channel-block.
//
// What is parallel and what is not: the per-bin terms y, w, w*x, w*x*x, w*y,
// w*x*y and the windowed line evaluation are independent per bin and are spread
// over the 64 lanes.  The five running sums N, X, XX, Y, XY are fp32
// accumulations *in index order* in the reference; windowed differences of them
// feed a division, so any re-association moves the answer by up to 5e-3
// (SURVEY.md Appendix A).  They are therefore accumulated serially, one lane
// per array, out of LDS -- five lanes busy for n steps.  The regime boundaries
// of the window loops depend only on the static bark[] table and are
// precomputed by vamd_create().
//
// LDS: S[5][VAMD_NZ_STRIDE(n)] running sums only; the noise curve and work vector stay in registers.
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"

namespace vamd {

// floats between a block's five running-sum arrays in LDS: n plus a stagger of 16, which puts the four
// chains a 16-lane LDS phase of the four-lanes-per-chain scan touches on disjoint bank groups
#define VAMD_NZ_STRIDE(n) ((n) + 16)

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

// window sums with a mirrored low edge (lo < 0 in the reference: lib/psy.c:613-617,666-670)
VAMD_DEV LineFit fit_mirrored(const float *S, int n, int hi, int mlo /* = -lo */) {
  const float *N = S, *X = S + VAMD_NZ_STRIDE(n), *XX = S + 2 * VAMD_NZ_STRIDE(n), *Y = S + 3 * VAMD_NZ_STRIDE(n), *XY = S + 4 * VAMD_NZ_STRIDE(n);
  return fit_from_sums(N[hi] + N[mlo], X[hi] - X[mlo], XX[hi] + XX[mlo], Y[hi] + Y[mlo], XY[hi] - XY[mlo]);
}
// plain window differences (lib/psy.c:635-639,687-691)
VAMD_DEV LineFit fit_plain(const float *S, int n, int hi, int lo) {
  const float *N = S, *X = S + VAMD_NZ_STRIDE(n), *XX = S + 2 * VAMD_NZ_STRIDE(n), *Y = S + 3 * VAMD_NZ_STRIDE(n), *XY = S + 4 * VAMD_NZ_STRIDE(n);
  return fit_from_sums(N[hi] - N[lo], X[hi] - X[lo], XX[hi] - XX[lo], Y[hi] - Y[lo], XY[hi] - XY[lo]);
}

VAMD_DEV LineFit bark_fit_from(const PsyP &P, int n, const float *S, int i, int b /* = bark[i] */) {
  const int lo = b >> 16, hi = b & 0xffff;
  return (i < P.bark_i1) ? fit_mirrored(S, n, hi, -lo) : fit_plain(S, n, hi, lo);
}
VAMD_DEV LineFit fixed_fit_at(const PsyP &P, int n, const float *S, int i, int fixed) {
  const int hi = i + fixed / 2, lo = hi - fixed;
  return (i < P.fix_i1) ? fit_mirrored(S, n, hi, -lo) : fit_plain(S, n, hi, lo);
}

// In-place running sum p[i] = p[0] + ... + p[i], strictly left to right in fp32
// (the reference's tN/tX/... accumulators, lib/psy.c:576-603).  One lane; n is a
// multiple of 32 (block sizes are powers of two >= 64).  The next 16 values are
// fetched from LDS while the current 16 are being added, so the dependent add
// chain -- the irreducible part -- is the only thing on the critical path.
#define VAMD_SCAN4(acc, v) \
  acc += v.x; v.x = acc; acc += v.y; v.y = acc; acc += v.z; v.z = acc; acc += v.w; v.w = acc;

VAMD_DEV void running_sum_inplace(float *p, int n) {
  F4 *q = (F4 *)p;
  float acc = 0.f;
  const int nblk = n >> 4;  // 16 values (4 quads) per block; nblk is even for n >= 32
  // two register sets (a*, b*) alternate so that no value is ever copied: while one set
  // is being summed the other set's loads are in flight and the previous stores drain
  F4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
  F4 b0 = q[4], b1 = q[5], b2 = q[6], b3 = q[7];
  for (int b = 0; b < nblk; b += 2) {
    VAMD_SCAN4(acc, a0) VAMD_SCAN4(acc, a1) VAMD_SCAN4(acc, a2) VAMD_SCAN4(acc, a3)
    q[4 * b] = a0;
    q[4 * b + 1] = a1;
    q[4 * b + 2] = a2;
    q[4 * b + 3] = a3;
    if (b + 2 < nblk) {
      a0 = q[4 * b + 8];
      a1 = q[4 * b + 9];
      a2 = q[4 * b + 10];
      a3 = q[4 * b + 11];
    }
    VAMD_SCAN4(acc, b0) VAMD_SCAN4(acc, b1) VAMD_SCAN4(acc, b2) VAMD_SCAN4(acc, b3)
    q[4 * b + 4] = b0;
    q[4 * b + 5] = b1;
    q[4 * b + 6] = b2;
    q[4 * b + 7] = b3;
    if (b + 3 < nblk) {
      b0 = q[4 * b + 12];
      b1 = q[4 * b + 13];
      b2 = q[4 * b + 14];
      b3 = q[4 * b + 15];
    }
  }
}

// Who runs the five running sums of a block.
//  ScanSolo : the wave that owns the block (5 lanes busy) -- single-wave workgroups, tests.
//  ScanGroup: the waves of a workgroup meet at a barrier and the first few of them walk the chains
//    of every block of the round, eight chains per wave and eight lanes per chain (below).
struct ScanSolo {
  VAMD_MEM void before_terms() const {}  // the wave's own WAVE_SYNC after the previous evaluation suffices
  VAMD_MEM void operator()(float *S, int n) const {
    WAVE_SYNC();
    WAVE_FOR(a, 5) running_sum_inplace(S + a * VAMD_NZ_STRIDE(n), n);
    WAVE_SYNC();
  }
};
#if VAMD_GPU
// The same running sums with L = 4 or 8 lanes per chain.  A wave's LDS instruction moves 64 lanes'
// worth of bytes whether or not the lanes carry data, and a VALU instruction occupies the SIMD for
// its four cycles whatever the lane mask, so a wave walks 64/L chains, 4L values of each per
// load/store pair: lane j of a chain's L owns quad L*s+j of step s.  The order of the adds is kept
// by handing the running total from lane to lane (DPP row_shr:1, fused into the next add): every
// lane runs all L rounds into temporaries -- only lane r's round-r result is meaningful, and it is
// exactly what lane r+1 receives for round r+1 -- so nothing on the dependent chain is predicated;
// a select tree off the chain then keeps each lane's own round.  tools/micro/scan_quad.hip, cycles
// per element of a chain: one lane per chain 18, L = 4: 12.2, L = 8: 10.4 -- elapsed time falls, but
// the SIMD time per chain-element does not (more instructions serve fewer chains), so this pays
// only where the other SIMDs would idle: small groups (<= 16 chains, one wave, L = 4), and the
// stand-alone 7-block group (L = 8, one wave per SIMD).
// Chains [first, first+count) of S_all, count <= 64/L; n a multiple of 8L.
VAMD_DEV float dpp_from_lane_below(float v) {  // row_shr:1 -- lane i receives lane i-1's value
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}
template <int L>
VAMD_DEV float dpp_from_last_lane(float v) {  // row_shl:(L-1) -- lane i receives lane i+L-1's value
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + (L - 1), 0xf, 0xf, true));
}
template <int L>
VAMD_DEV void running_sum_lanes(float *S_all, int first, int count, int n) {
  const int g = LANE / L, j = LANE % L;
  const bool act = g < count;
  F4 *q = (F4 *)(S_all + (first + (act ? g : 0)) * VAMD_NZ_STRIDE(n)) + j;
  const int steps = n / (4 * L);  // even
  float carry = 0.f;              // lane j == 0: the chain's total through the previous step
  F4 a = q[0], b = q[L];
  for (int s = 0; s < steps; s += 2) {
#pragma unroll
    for (int half = 0; half < 2; half++) {
      F4 &v = half ? b : a;
      F4 t[L];
      float cin = carry;
#pragma unroll
      for (int r = 0; r < L; r++) {
        t[r].x = cin + v.x;
        t[r].y = t[r].x + v.y;
        t[r].z = t[r].y + v.z;
        t[r].w = t[r].z + v.w;
        cin = dpp_from_lane_below(t[r].w);
      }
      carry = dpp_from_last_lane<L>(t[L - 1].w);
      F4 o = t[0];
#pragma unroll
      for (int r = 1; r < L; r++) {
        const bool m = j == r;
        o.x = m ? t[r].x : o.x;
        o.y = m ? t[r].y : o.y;
        o.z = m ? t[r].z : o.z;
        o.w = m ? t[r].w : o.w;
      }
      if (act) q[L * (s + half)] = o;
      if (s + half + 2 < steps) v = q[L * (s + half + 2)];
    }
  }
}

struct ScanGroup {
  float *S_all;  // the workgroup's LDS: chain c lives at S_all + c*VAMD_NZ_STRIDE(n)
  int nchains;   // 5 x blocks
  // a block's arrays are shared by the waves that split its bins: nobody may overwrite them
  // with new terms while a sibling is still evaluating lines from the previous sums
  VAMD_MEM void before_terms() const { __syncthreads(); }
  VAMD_MEM void operator()(float *, int n) const {
    __syncthreads();
    if (nchains <= 16) {  // one wave, four lanes per chain
      if ((threadIdx.x >> 6) == 0) running_sum_lanes<4>(S_all, 0, nchains, n);
    } else {              // eight lanes per chain, eight chains per wave
      const int first = (threadIdx.x >> 6) * 8;
      if (first < nchains) running_sum_lanes<8>(S_all, first, nchains - first < 8 ? nchains - first : 8, n);
    }
    __syncthreads();
  }
};
#endif

// bark_noise_hybridmp(n, bark, f, noise, offset, fixed).  f and noise are per-lane
// register tiles: this wave owns the quads [q0, q1) of the block, lane l the quads q0+l,
// q0+l+64, ... (SLICE_QUADS, QPS per lane), four bins each.  With one wave per block the slice is
// the whole block; the persistent kernel gives a block to two waves so that the per-bin phases --
// bound by each wave's dependent chains, not by issue slots -- take half as long.
// LOGN > 0: the bin count is the compile-time constant 2^LOGN (array strides and window-edge addresses
// then fold into immediate offsets); 0 = P.n.
template <class Scan, int QPS, int LOGN = 0>
VAMD_DEV void bark_noise_wave(const PsyP &P, const float (*f)[4], float (*noise)[4], const float offset,
                              const int fixed, float *S, const Scan &scan, PhaseClock &pc, int slot, int q0, int q1) {
  const int n = LOGN ? (1 << LOGN) : P.n;
  float *N = S, *X = S + VAMD_NZ_STRIDE(n), *XX = S + 2 * VAMD_NZ_STRIDE(n), *Y = S + 3 * VAMD_NZ_STRIDE(n), *XY = S + 4 * VAMD_NZ_STRIDE(n);

  // per-bin terms (lib/psy.c:571-597)
  scan.before_terms();
  SLICE_QUADS(kq, q, q0, q1, QPS) {
    float tn[4], tx[4], txx[4], ty[4], txy[4];
#if VAMD_GPU
#pragma unroll
#endif
    for (int c = 0; c < 4; c++) {
      const int i = (q << 2) + c;
      float y = f[kq][c] + offset;
      if (y < 1.f) y = 1.f;
      if (i == 0) {
        const float w = (float)((double)(y * y) * .5);
        tn[c] = w;
        tx[c] = w;  // sic: the reference adds w, not w*x, at bin 0 (lib/psy.c:577)
        txx[c] = 0.f;
        ty[c] = w * y;
        txy[c] = 0.f;
      } else {
        const float x = (float)i;
        const float w = y * y;
        tn[c] = w;
        tx[c] = w * x;
        txx[c] = w * x * x;
        ty[c] = w * y;
        txy[c] = w * x * y;
      }
    }
    ((F4 *)N)[q] = f4_make(tn);
    ((F4 *)X)[q] = f4_make(tx);
    ((F4 *)XX)[q] = f4_make(txx);
    ((F4 *)Y)[q] = f4_make(ty);
    ((F4 *)XY)[q] = f4_make(txy);
  }
  pc.mark(slot);

  // the five running sums, in index order, one lane each (lib/psy.c:576-603)
  scan(S, n);
  pc.mark(slot + 1);

  // line evaluation; three regimes split at the static indices i1 <= i2
  // (lib/psy.c:606-656).  Beyond i2 the last fitted line is extended.
  LineFit last;
  last.A = 0.f;
  last.B = 0.f;
  last.D = 1.f;
  if (P.bark_i2 > 0) last = bark_fit_from(P, n, S, P.bark_i2 - 1, P.bark[P.bark_i2 - 1]);
  SLICE_QUADS(kq, q, q0, q1, QPS) {
    const I4 bq = ((const I4 *)P.bark)[q];
    const int bk[4] = {bq.x, bq.y, bq.z, bq.w};
#if VAMD_GPU
#pragma unroll
#endif
    for (int c = 0; c < 4; c++) {
      const int i = (q << 2) + c;
      const LineFit L = (i < P.bark_i2) ? bark_fit_from(P, n, S, i, bk[c]) : last;
      const float x = (float)i;
      float R = (L.A + x * L.B) / L.D;
      if (R < 0.f) R = 0.f;
      noise[kq][c] = R - offset;
    }
  }
  if (fixed > 0) {
    // fixed-width window pass: keep the lower of the two curves (lib/psy.c:660-703)
    if (P.fix_i2 > 0) last = fixed_fit_at(P, n, S, P.fix_i2 - 1, fixed);
    SLICE_QUADS(kq, q, q0, q1, QPS) {
#if VAMD_GPU
#pragma unroll
#endif
      for (int c = 0; c < 4; c++) {
        const int i = (q << 2) + c;
        const LineFit L = (i < P.fix_i2) ? fixed_fit_at(P, n, S, i, fixed) : last;
        const float x = (float)i;
        const float R = (L.A + x * L.B) / L.D;
        if (R - offset < noise[kq][c]) noise[kq][c] = R - offset;
      }
    }
  }
  WAVE_SYNC();  // S is rewritten by the next pass
  pc.mark(slot + 2);
}

// _vp_noisemask on a block whose logmdct is already in a register tile
template <class Scan, int QPS, int LOGN = 0>
VAMD_DEV void noisemask_tile(const PsyP &P, const float (*lm)[4], float (*o)[4], float *S, const Scan &scan,
                             PhaseClock &pc, int q0, int q1) {
  float nz[QPS][4], wk[QPS][4];
  bark_noise_wave<Scan, QPS, LOGN>(P, lm, nz, 140.f, -1, S, scan, pc, 0, q0, q1);
  SLICE_QUADS(kq, q, q0, q1, QPS) {
    for (int c = 0; c < 4; c++) wk[kq][c] = lm[kq][c] - nz[kq][c];
  }
  pc.mark(3);
  bark_noise_wave<Scan, QPS, LOGN>(P, wk, nz, 0.f, P.noisewindowfixed, S, scan, pc, 4, q0, q1);
  SLICE_QUADS(kq, q, q0, q1, QPS) {
#if VAMD_GPU
#pragma unroll
#endif
    for (int c = 0; c < 4; c++) {
      const float w = lm[kq][c] - wk[kq][c];
      int dB = (int)((double)nz[kq][c] + .5);
      if (dB >= VAMD_NOISE_COMPAND_LEVELS) dB = VAMD_NOISE_COMPAND_LEVELS - 1;
      if (dB < 0) dB = 0;
      o[kq][c] = w + P.noisecompand[dB];
    }
  }
  pc.mark(7);
}

// _vp_noisemask(p, logmdct, logmask) for one block (single-wave form)
//   logmdct  [n] input (HBM), out [n] (HBM); S = LDS [5][VAMD_NZ_STRIDE(n)]
// The noise curve and the work vector never change hands between lanes, so they live in
// registers (VAMD_QPL quads per lane: block sizes up to 2048 on the GPU).
VAMD_DEV void noisemask_block(const PsyP &P, const float *__restrict__ logmdct, float *__restrict__ out, float *S,
                              PhaseClock &pc) {
  const int nq = P.n >> 2;
  float lm[VAMD_QPL][4], o[VAMD_QPL][4];
  LANE_QUADS(kq, q, nq) f4_get(((const F4 *)logmdct)[q], lm[kq]);
  noisemask_tile<ScanSolo, VAMD_QPL>(P, lm, o, S, ScanSolo(), pc, 0, nq);
  LANE_QUADS(kq, q, nq)((F4 *)out)[q] = f4_make(o[kq]);
}

}  // namespace vamd
