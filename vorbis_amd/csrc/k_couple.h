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

// mdct[k]      HBM [n2]  post-M1 spectrum of channel k
// ilogmask[k]  HBM [n2]  integer floor curve (floor1_encode's output)
// iwork[k]     HBM [n2]  out: quantised (and coupled) residue
// nonzero      [ch] in: floor1_encode's return per channel; out: after the coupling fix-up
VAMD_DEV void couple_block(const CoupleP &C, const PsyP &P, int n2, const float *const *mdct,
                           const int *const *ilogmask, int *const *iwork, int *nonzero, const CoupleLds &L, PhaseClock &pc) {
  const int ch = C.ch;
  const int partition = P.normal_p ? P.normal_partition : 16;
  const int nstart = P.normal_p ? P.normal_start : 0x7fffffff;  // first bin subject to noise norm
  const bool norm_active = nstart < n2;
  const int nparts = (n2 + partition - 1) / partition;
  int nz[VAMD_MAX_CH];
  for (int k = 0; k < ch; k++) nz[k] = nonzero[k];

  // ---- per channel: floor lookup, lossless flags, energies, first quantisation
  for (int k = 0; k < ch; k++) {
    WAVE_FOR(b, n2) {
      int out = 0;
      float cand = -1.f, key = 0.f, sg = 0.f;
      if (nz[k]) {
        const float m = mdct[k][b];
        const float fl = floor1_fromdB(ilogmask[k][b]);
        float raw = m * m;
        const float quant = raw;
        if (m < 0.f) raw *= -1.f;
        const float fl2 = fl * fl;
        const float ve = quant / fl2;
        if (b < nstart || !(ve < .25f)) {
          out = quant_energy(ve, raw);
        } else {
          cand = ve;  // flags == NULL: every small bin past normal_start is a candidate
          key = quant;
          sg = raw;
        }
      }
      iwork[k][b] = out;
      if (norm_active) {
        L.cand[b] = cand;
        L.key[b] = key;
        L.sgn[b] = sg;
      }
    }
    if (norm_active && nz[k]) {
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
  if (C.coupling_steps == 1 && (nz[C.mag] || nz[C.ang])) {
    const int Mi = C.mag, Ai = C.ang;
    WAVE_FOR(b, n2) {
      // rebuild the per-bin state pass A had (cheaper than keeping it in LDS)
      float re[2], qe[2], fl[2];
      int fg[2];
      for (int s = 0; s < 2; s++) {
        const int k = s ? Ai : Mi;
        if (nz[k]) {
          const float m = mdct[k][b];
          const float f = floor1_fromdB(ilogmask[k][b]);
          const float point = b >= C.pointlimit ? C.postpoint : C.prepoint;
          const float r = (float)(fabs((double)m) / (double)f);  // flag_lossless, lib/psy.c:928-933
          fg[s] = r < point ? 0 : 1;
          re[s] = m * m;
          qe[s] = re[s];
          if (m < 0.f) re[s] *= -1.f;
          fl[s] = f * f;
        } else {
          fl[s] = 1e-10f;
          re[s] = 0.f;
          qe[s] = 0.f;
          fg[s] = 0;
        }
      }
      int iM = iwork[Mi][b], iA = iwork[Ai][b];
      if (b < C.sliding_lowpass) {
        if (fg[0] || fg[1]) {
          // lossless: square-polar coupling of the already quantised integers
          re[0] = (float)(fabs((double)re[0]) + fabs((double)re[1]));
          qe[0] = qe[0] + qe[1];
          fg[0] = fg[1] = 1;
          const int A = iM, B = iA;
          const int aA = A < 0 ? -A : A, aB = B < 0 ? -B : B;
          if (aA > aB) {
            iA = (A > 0 ? A - B : B - A);
          } else {
            iA = (B > 0 ? A - B : B - A);
            iM = B;
          }
          if (iA >= (iM < 0 ? -iM : iM) * 2) {
            iA = -iA;
            iM = -iM;
          }
        } else {
          // lossy point coupling
          if (b < C.pointlimit) {
            re[0] += re[1];
            qe[0] = (float)fabs((double)re[0]);
          } else {
            const float e = (float)(fabs((double)re[0]) + fabs((double)re[1]));
            qe[0] = e;
            re[0] = (re[0] + re[1] < 0) ? -e : e;
          }
          re[1] = qe[1] = 0.f;
          fg[1] = 1;
          iA = 0;
        }
      }
      fl[0] = fl[1] = fl[0] + fl[1];
      // normalise the magnitude vector (noise_normalize with flags = fM)
      float cand = -1.f;
      if (!fg[0]) {
        const float ve = qe[0] / fl[0];
        if (b < nstart || !(ve < .25f && b >= C.pointlimit)) {
          iM = quant_energy(ve, re[0]);
        } else {
          cand = ve;
        }
      }
      iwork[Mi][b] = iM;
      iwork[Ai][b] = iA;
      if (norm_active) {
        L.cand[b] = cand;
        L.key[b] = qe[0];
        L.sgn[b] = re[0];
      }
    }
    if (norm_active) {
      WAVE_SYNC_GLOBAL();
      WAVE_FOR(p, nparts) {
        const int b0 = p * partition;
        const int jn = partition > n2 - b0 ? n2 - b0 : partition;
        noise_norm_partition(P, L, b0, jn, iwork[Mi]);
      }
      WAVE_SYNC_GLOBAL();
    }
    // lib/psy.c:1204-1212
    nz[Mi] = nz[Ai] = 1;
  }
  for (int k = 0; k < ch; k++) nonzero[k] = nz[k];
  pc.mark(1);
}

}  // namespace vamd
