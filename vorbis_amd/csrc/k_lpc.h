// k_lpc.h -- the two extrapolations either end of a stream that vorbis_analysis_wrote() performs on the host
// (reference lib/block.c:417-458 _preextrapolate_helper, :474-512 the end-of-stream padding), with lib/lpc.c's
// vorbis_lpc_from_data (:61-131) and vorbis_lpc_predict (:135-160) behind them -- device-resident, so that a stream
// that arrives from host memory as 16-bit samples (vamd_feed) is cut into exactly the blocks the reference cuts.
//
// One wave per (stream, channel).  What the reference computes, and how it is laid out here:
//   * autocorrelation: aut[j] = sum_{i=j}^{n-1} (double)data[i]*data[i-j], j = 0..m, each a serial fp64 sum in index
//     order (the order is part of the result for float input).  Lane j owns lag j: m+1 <= 33 lanes walk the data --
//     staged in LDS -- side by side; data[i] is a broadcast read, data[i-j] consecutive addresses.
//   * Levinson-Durbin in fp64 (lib/lpc.c:80-114) and the damping (:119-126): a few hundred dependent operations on
//     one lane, the arrays in LDS.
//   * the predictor: y_i = -(sum_j work[i+j]*coeff[m-1-j]) with every product rounded to float and subtracted in
//     order j = 0..m-1 (:150-158).  Output i's LAST term is output i-1, the term before it output i-2 ...: the chain
//     from one output to the next is one multiply and one subtract, everything else of an output's sum only needs
//     older outputs.  So the sum is evaluated as a systolic line: lane j holds coeff[m-1-j]; at step t every lane
//     multiplies the SAME sample work[t] (the newest output, v_readlane of lane m-1) and subtracts it from the partial
//     sum its lower neighbour formed the step before (DPP wave_shr:1).  Output i leaves lane m-1 at step i+m-1: one
//     output per step, each of its m subtractions in the reference's order.  (A lane walking the m terms of each
//     output serially: 6144 x 32 dependent subtractions for a stream's tail, 0.6 ms; the line: 6176 steps, ~0.07 ms.)
#pragma once
#include "vamd_wave.h"

namespace vamd {

#define VAMD_LPC_MAX_ORDER 32

// lib/lpc.c:61-131 on data[0..n) in LDS (or host memory); coeff[m] out.  Every lane returns the same coefficients
// in coeff[] (LDS / host array of m floats).  `aut` is scratch for 2m+1 doubles (the lags, then the fp64 coefficients).
VAMD_DEV void lpc_from_data(const float *data, int n, int m, double *aut, float *coeff) {
#if VAMD_GPU
  if (LANE <= m) {
    const int j = LANE;
    double d = 0.;
    int i = j;
    for (; i + 8 <= n; i += 8) {  // (eight terms' reads in flight; the sum itself stays in index order)
      float a[8], b[8];
#pragma unroll
      for (int k = 0; k < 8; k++) a[k] = data[i + k], b[k] = data[i + k - j];
#pragma unroll
      for (int k = 0; k < 8; k++) d += (double)a[k] * (double)b[k];
    }
    for (; i < n; i++) d += (double)data[i] * (double)data[i - j];
    aut[j] = d;
  }
  WAVE_SYNC();
#else
  for (int j = m; j >= 0; j--) {
    double d = 0.;
    for (int i = j; i < n; i++) d += (double)data[i] * data[i - j];
    aut[j] = d;
  }
#endif
  double *lpc = aut + m + 1;
  if (LANE == 0) {  // (a few hundred dependent fp64 operations: one lane, the arrays in LDS)
    double error = aut[0] * (1. + 1e-10);
    const double epsilon = 1e-9 * aut[0] + 1e-10;
    for (int i = 0; i < m; i++) {
      double r = -aut[i + 1];
      if (error < epsilon) {
        for (int k = i; k < m; k++) lpc[k] = 0.;
        break;
      }
      for (int j = 0; j < i; j++) r -= lpc[j] * aut[i - j];
      r /= error;
      lpc[i] = r;
      int j;
      for (j = 0; j < i / 2; j++) {
        const double tmp = lpc[j];
        lpc[j] += r * lpc[i - 1 - j];
        lpc[i - 1 - j] += r * tmp;
      }
      if (i & 1) lpc[j] += lpc[j] * r;
      error *= 1. - r * r;
    }
    const double g = .99;
    double damp = g;
    for (int j = 0; j < m; j++) {
      lpc[j] *= damp;
      damp *= g;
    }
    for (int j = 0; j < m; j++) coeff[j] = (float)lpc[j];
  }
  WAVE_SYNC();
}

// lib/lpc.c:135-160: prime[0..m) -> out[0..n).  coeff, prime, out: LDS (or host) arrays; out may not overlap prime.
VAMD_DEV void lpc_predict(const float *coeff, const float *prime, int m, float *out, int n) {
#if VAMD_GPU
  const float c = LANE < m ? coeff[m - 1 - LANE] : 0.f;
  const int primed = LANE < m ? __float_as_int(prime[LANE]) : 0;  // work[t], t < m, read with v_readlane
  float S = 0.f, newest = 0.f, keep = 0.f;
  // the steps that consume the primer (no output is complete before step m-1)
  for (int t = 0; t < m; t++) {
    const float wt = __int_as_float(wave_read(primed, t));
    const float below = __int_as_float(wave_shift_up1(__float_as_int(S), 0));  // lane 0 starts an output: y = 0
    const float p = wt * c;
    S = below - p;
  }
  newest = __int_as_float(wave_read(__float_as_int(S), m - 1));  // output 0
  if (LANE == 0) keep = newest;
  // from here on every step consumes the newest output and completes the next one; outputs gather one per lane and
  // leave for `out` sixty-four at a time (no LDS traffic inside the chain)
  for (int i = 1; i < n; i++) {
    const float below = __int_as_float(wave_shift_up1(__float_as_int(S), 0));
    const float p = newest * c;
    S = below - p;
    newest = __int_as_float(wave_read(__float_as_int(S), m - 1));  // output i
    if ((i & 63) == LANE) keep = newest;
    if ((i & 63) == 63) out[i - 63 + LANE] = keep;
  }
  if ((n & 63) && LANE < (n & 63)) out[(n & ~63) + LANE] = keep;
  WAVE_SYNC();
#else
  float work[VAMD_LPC_MAX_ORDER];
  for (int i = 0; i < m; i++) work[i] = prime[i];
  for (int i = 0; i < n; i++) {
    float y = 0.f;
    for (int j = 0; j < m; j++) {
      const float p = work[j] * coeff[m - 1 - j];
      y -= p;
    }
    for (int j = 0; j + 1 < m; j++) work[j] = work[j + 1];
    work[m - 1] = y;
    out[i] = y;
  }
#endif
}

}  // namespace vamd
