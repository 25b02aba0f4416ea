// k_tone.h -- _vp_tonemask (reference lib/psy.c:754-777): ATH floor, seed_loop /
// seed_curve scatter (lib/psy.c:390-452), seed_chase (:454-508) and max_seeds
// (:512-545); SURVEY.md 8a row a10.  One wavefront per channel-block.
//
// Parallel form:
//  * runs of bins sharing an octave[] value are a property of the static table;
//    vamd_create() lists them, lanes take one run each, find the run's peak and
//    scatter the chosen tone curve into seed[] with an LDS float max (max is
//    order-free, so the result equals the reference's sequential update).
//  * seed_chase is NOT a sliding-window max (SURVEY.md 0.8 iv): its stack
//    discipline is restated literally and walked by one lane.
//  * max_seeds' pointer walk over (octave line, bin) is static too: per bin the
//    span of seed lines it folds is precomputed, every lane folds its own bins.
//
// LDS: seed[total_octave_lines], stack_pos[total], stack_amp[total].
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"

namespace vamd {

#define VAMD_NEGINF (-9999.f)

// seed_curve, lib/psy.c:390-415
VAMD_DEV void seed_curve_scatter(float *seed, const float *__restrict__ curves /*[8][58] of one band*/, float amp,
                                 int oc, int nlines, int linesper, float dBoffset) {
  int choice = (int)(((double)(amp + dBoffset) - 30.) * (double).1f);
  choice = choice < 0 ? 0 : choice;
  choice = choice > VAMD_P_LEVELS - 1 ? VAMD_P_LEVELS - 1 : choice;
  const float *posts = curves + choice * (VAMD_EHMER_MAX + 2);
  const float *curve = posts + 2;
  const int post1 = (int)posts[1];
  int seedptr = (int)((float)oc + (posts[0] - (float)VAMD_EHMER_OFFSET) * (float)linesper - (float)(linesper >> 1));
  for (int i = (int)posts[0]; i < post1; i++) {
    if (seedptr > 0) {
      const float lin = amp + curve[i];
      lds_atomic_max(seed + seedptr, lin);
    }
    seedptr += linesper;
    if (seedptr >= nlines) break;
  }
}

// seed_chase, lib/psy.c:454-508 -- literal, single lane
VAMD_DEV void seed_chase_serial(float *seeds, int linesper, int n, int *posstack, float *ampstack) {
  int stack = 0;
  for (int i = 0; i < n; i++) {
    const float s = seeds[i];
    if (stack < 2) {
      posstack[stack] = i;
      ampstack[stack++] = s;
    } else {
      while (1) {
        if (s < ampstack[stack - 1]) {
          posstack[stack] = i;
          ampstack[stack++] = s;
          break;
        } else {
          if (i < posstack[stack - 1] + linesper) {
            if (stack > 1 && ampstack[stack - 1] <= ampstack[stack - 2] && i < posstack[stack - 2] + linesper) {
              stack--;  // fully overlapped: stack-1 is irrelevant
              continue;
            }
          }
          posstack[stack] = i;
          ampstack[stack++] = s;
          break;
        }
      }
    }
  }
  int pos = 0;
  for (int i = 0; i < stack; i++) {
    int endpos;
    if (i < stack - 1 && ampstack[i + 1] > ampstack[i])
      endpos = posstack[i + 1];
    else
      endpos = posstack[i] + linesper + 1;
    if (endpos > n) endpos = n;
    const float a = ampstack[i];
    for (; pos < endpos; pos++) seeds[pos] = a;
  }
}

// _vp_tonemask(p, logfft, logmask, global_specmax, local_specmax)
VAMD_DEV void tonemask_block(const PsyP &P, const float *__restrict__ logfft, float *__restrict__ out,
                             float global_ampmax, float local_ampmax, float *seed, int *posstack, float *ampstack,
                             float *flr /* LDS [n] */) {
  const int n = P.n, nlines = P.total_octave_lines;
  float att = local_ampmax + P.ath_adjatt;
  if (att < P.ath_maxatt) att = P.ath_maxatt;

  WAVE_FOR(i, nlines) seed[i] = VAMD_NEGINF;
  WAVE_FOR(i, n) flr[i] = P.ath[i] + att;
  WAVE_SYNC();

  // seed_loop, lib/psy.c:417-452
  const float dBoffset = P.max_curve_dB - global_ampmax;
  WAVE_FOR(r, P.nruns) {
    const int s = P.run_start[r], e = P.run_start[r + 1];  // bins [s, e)
    float mx = logfft[s];
    for (int i = s + 1; i < e; i++)
      if (logfft[i] > mx) mx = logfft[i];
    if (mx + 6.f > flr[e - 1]) {
      const int ocv = P.octave[s];
      int band = ocv >> P.shiftoc;
      if (band >= VAMD_P_BANDS) band = VAMD_P_BANDS - 1;
      if (band < 0) band = 0;
      seed_curve_scatter(seed, P.tonecurves + band * (VAMD_P_LEVELS * (VAMD_EHMER_MAX + 2)), mx, ocv - P.firstoc,
                         nlines, P.eighth_octave_lines, dBoffset);
    }
  }
  WAVE_SYNC();

  WAVE_FOR(z, 1) seed_chase_serial(seed, P.eighth_octave_lines, nlines, posstack, ampstack);
  WAVE_SYNC();

  // max_seeds' fold, lib/psy.c:522-543, per bin over its precomputed line span
  WAVE_FOR(i, n) {
    float minV;
    if (i >= P.tail_linpos) {
      minV = seed[nlines - 1];
    } else {
      const int p0 = P.seed_span[2 * i], p1 = P.seed_span[2 * i + 1];
      minV = seed[p0];
      if (minV > P.tone_abs_limit) minV = P.tone_abs_limit;
      for (int p = p0 + 1; p <= p1; p++) {
        const float s = seed[p];
        if ((s > VAMD_NEGINF && s < minV) || minV == VAMD_NEGINF) minV = s;
      }
    }
    float v = flr[i];
    if (v < minV) v = minV;
    out[i] = v;
  }
  WAVE_SYNC();
}

}  // namespace vamd
