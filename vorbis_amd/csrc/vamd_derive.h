// vamd_derive.h -- host-side, once per context: static index tables that the
// reference re-discovers with data-independent loops on every block.  They
// depend only on the setup blob (octave[], bark[], n, window widths), never on
// audio, so vamd_create() computes them once and ships them to HBM next to the
// blob.  Pure integer walks; each cites the loop it freezes.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "vamd_setup.h"
#include "vamd_params.h"

namespace vamd {

// one run of bins sharing an octave[] value, everything seed_loop needs about it that
// does not depend on the audio (lib/psy.c:429-449), 16 bytes
struct RunRec {
  int32_t start_end;  // start | end << 16   (bins [start, end))
  int32_t ocpos;      // octave[start] - firstoc
  int32_t band;       // clamp(octave[start] >> shiftoc, 0, P_BANDS-1)
  float ath_last;     // ath[end-1]
};

struct PsyDerived {
  int bark_i1, bark_i2;  // lib/psy.c:606-656
  int fix_i1, fix_i2;    // lib/psy.c:660-703
  std::vector<int32_t> run_start;  // nruns+1 entries, lib/psy.c:429-435
  std::vector<RunRec> runs;        // nruns
  std::vector<float> curves64;     // tone curve points re-strided [17][8][64] (rows 256-byte aligned)
  std::vector<int32_t> seed_span;  // [n][2], lib/psy.c:522-537
  int tail_linpos;                 // lib/psy.c:539-543
  // the same walk, organised for a line-parallel fold: every iteration of max_seeds' outer
  // loop is a "group" g that starts from seed[p0_g] and then scans lines (p0_g, p1_g]
  std::vector<int32_t> bin_fold;     // [n]   p0 | group << 16  (bins below tail_linpos)
  std::vector<uint16_t> line_group;  // [nl padded to 16] group whose scan covers the line, 0xffff = none
  int ngroups;
};

inline PsyDerived derive_psy(const vamd_psy_tab &t, const unsigned char *blob) {
  PsyDerived d;
  const int n = t.n;
  const int32_t *bark = (const int32_t *)(blob + t.off_bark);
  const int32_t *octave = (const int32_t *)(blob + t.off_octave);

  // bark_noise_hybridmp pass 1: first loop runs while lo<0 && -lo<n && hi<n,
  // second while 0<=lo<n && hi<n, the rest extends the last line.
  int i = 0;
  for (; i < n; i++) {
    const int lo = bark[i] >> 16, hi = bark[i] & 0xffff;
    if (lo >= 0 || -lo >= n) break;
    if (hi >= n) break;
  }
  d.bark_i1 = i;
  for (; i < n; i++) {
    const int lo = bark[i] >> 16, hi = bark[i] & 0xffff;
    if (lo < 0 || lo >= n) break;
    if (hi >= n) break;
  }
  d.bark_i2 = i;

  // fixed-width pass: hi = i + fixed/2, lo = hi - fixed
  const int fixed = t.noisewindowfixed;
  d.fix_i1 = d.fix_i2 = 0;
  if (fixed > 0) {
    for (i = 0; i < n; i++) {
      const int hi = i + fixed / 2, lo = hi - fixed;
      if (hi >= n) break;
      if (lo >= 0) break;
    }
    d.fix_i1 = i;
    for (; i < n; i++) {
      const int hi = i + fixed / 2, lo = hi - fixed;
      if (hi >= n) break;
      if (lo < 0) break;
    }
    d.fix_i2 = i;
  }

  // seed_loop: maximal runs of equal octave[] values
  for (i = 0; i < n;) {
    d.run_start.push_back(i);
    int j = i;
    while (j + 1 < n && octave[j + 1] == octave[i]) j++;
    i = j + 1;
  }
  d.run_start.push_back(n);
  {
    const float *ath = (const float *)(blob + t.off_ath);
    for (size_t r = 0; r + 1 < d.run_start.size(); r++) {
      RunRec rr;
      const int s = d.run_start[r], e = d.run_start[r + 1];
      rr.start_end = s | (e << 16);
      rr.ocpos = octave[s] - t.firstoc;
      int band = octave[s] >> t.shiftoc;
      if (band >= VAMD_P_BANDS) band = VAMD_P_BANDS - 1;
      if (band < 0) band = 0;
      rr.band = band;
      rr.ath_last = ath[e - 1];
      d.runs.push_back(rr);
    }
    const float *tc = (const float *)(blob + t.off_tonecurves);
    // row = the 56 curve points, -inf outside the fence posts [tc[0], tc[1]) that seed_curve
    // (lib/psy.c:396-398) limits the walk to, so the scatter needs no range test
    uint32_t ninf_bits = 0xff800000u;
    float ninf;
    memcpy(&ninf, &ninf_bits, 4);
    d.curves64.assign((size_t)VAMD_P_BANDS * VAMD_P_LEVELS * 64, ninf);
    for (int bc = 0; bc < VAMD_P_BANDS * VAMD_P_LEVELS; bc++) {
      const float *row = tc + (size_t)bc * (VAMD_EHMER_MAX + 2);
      const int i0 = (int)row[0], i1 = (int)row[1];
      for (int k = 0; k < VAMD_EHMER_MAX; k++)
        if (k >= i0 && k < i1) d.curves64[(size_t)bc * 64 + k] = row[2 + k];
      // the fence posts ride in the row's padding: the scatter skips the point groups no lane needs
      d.curves64[(size_t)bc * 64 + VAMD_EHMER_MAX] = (float)i0;
      d.curves64[(size_t)bc * 64 + VAMD_EHMER_MAX + 1] = (float)i1;
    }
  }

  // max_seeds: replay the (pos, linpos) walk; record per bin the seed-line span
  // [p0, p1] whose fold gives that bin's minV.
  d.seed_span.assign((size_t)2 * n, 0);
  d.bin_fold.assign((size_t)n, 0);
  d.line_group.assign((size_t)((t.total_octave_lines + 15) & ~15), 0xffff);
  d.ngroups = 0;
  {
    const int linesper = t.eighth_octave_lines;
    long linpos = 0;
    long pos = octave[0] - t.firstoc - (linesper >> 1);
    while (linpos + 1 < n) {
      const long p0 = pos;
      long end = ((octave[linpos] + octave[linpos + 1]) >> 1) - t.firstoc;
      while (pos + 1 <= end) pos++;
      end = pos + t.firstoc;
      for (long p = p0 + 1; p <= pos; p++) d.line_group[(size_t)p] = (uint16_t)d.ngroups;
      for (; linpos < n && octave[linpos] <= end; linpos++) {
        d.seed_span[2 * linpos] = (int32_t)p0;
        d.seed_span[2 * linpos + 1] = (int32_t)pos;
        d.bin_fold[(size_t)linpos] = (int32_t)(p0 | ((long)d.ngroups << 16));
      }
      d.ngroups++;
    }
    d.tail_linpos = (int)linpos;
  }
  return d;
}

// accumulate_fit (lib/floor1.c:406-454) is called once per pair of neighbouring
// posts with the inclusive bin range [sorted_index[j], sorted_index[j+1]] clipped to
// look_n-1.  bin_interval[i] = the interval j whose range starts at or before bin i
// (the last such j); a bin sitting exactly on an interior post also belongs to j-1,
// which is flagged by bit 7.  255 = the bin belongs to no interval.
inline std::vector<unsigned char> derive_bin_interval(const vamd_floor1_tab &f, int n2) {
  std::vector<unsigned char> t((size_t)((n2 + 15) & ~15), 255);
  for (int j = 0; j + 1 < f.posts; j++) {
    int x0 = f.sorted_index[j], x1 = f.sorted_index[j + 1];
    if (x1 >= f.look_n) x1 = f.look_n - 1;
    for (int i = x0; i <= x1 && i < n2; i++) t[i] = (unsigned char)(j | ((j > 0 && i == x0) ? 0x80 : 0));
  }
  return t;
}

// The same ranges as a work list for a wave: interval j's bins cut at the 16-bin chunks of the quantised mask
// (one 32-byte LDS read), so that a lane sums one chunk's share of ONE interval and adds it to that interval once.
// A bin on an interior post sits in two records, as it sits in two of the reference's calls.  Record = 12 words:
// [0] chunk, [1] interval, [2..3] 0, [4..11] keep-masks for the chunk's eight words (two 16-bit bins each).
inline std::vector<uint32_t> derive_fit_segments(const vamd_floor1_tab &f, int n2, int *nseg) {
  std::vector<uint32_t> t;
  *nseg = 0;
  for (int j = 0; j + 1 < f.posts; j++) {
    int x0 = f.sorted_index[j], x1 = f.sorted_index[j + 1];
    if (x1 >= f.look_n) x1 = f.look_n - 1;
    if (x1 >= n2) x1 = n2 - 1;
    for (int c = x0 >> 4; x0 <= x1 && c <= (x1 >> 4); c++) {
      uint32_t r[VAMD_FITSEG_WORDS] = {(uint32_t)c, (uint32_t)j};
      for (int b = 0; b < 16; b++) {
        const int i = 16 * c + b;
        if (i >= x0 && i <= x1) r[4 + (b >> 1)] |= 0xffffu << (16 * (b & 1));
      }
      t.insert(t.end(), r, r + VAMD_FITSEG_WORDS);
      (*nseg)++;
    }
  }
  return t;
}

// div_magic()'s multipliers: ceil(2^32 / den) for den = 2 .. VAMD_DIV_MAGIC_MAX ([0] and [1] only ever meet num == 0)
inline std::vector<uint32_t> derive_div_magic() {
  std::vector<uint32_t> t(VAMD_DIV_MAGIC_MAX + 1, 0xffffffffu);
  for (uint64_t d = 2; d <= VAMD_DIV_MAGIC_MAX; d++) t[d] = (uint32_t)(((1ull << 32) + d - 1) / d);
  return t;
}

// floor1_fit / floor1_encode settle the posts in list order, each from its two neighbours
// (lib/floor1.c:708-724,790-831).  The neighbours are fixed by the look, so posts can be
// settled level by level: level[i] = 1 + max(level[lo], level[hi]), posts 0 and 1 at level 0.
inline std::vector<int32_t> derive_post_levels(const vamd_floor1_tab &f, int *nlevels) {
  std::vector<int32_t> lv(64, 0);
  int mx = 0;
  for (int i = 2; i < f.posts; i++) {
    const int a = lv[f.loneighbor[i - 2]], b = lv[f.hineighbor[i - 2]];
    lv[i] = 1 + (a > b ? a : b);
    if (lv[i] > mx) mx = lv[i];
  }
  *nlevels = mx;
  return lv;
}

// stereo_threshholds / _limited, lib/psy.c:32-33
inline float stereo_threshold(int idx, bool limited) {
  static const double a[] = {0.0, .5, 1.0, 1.5, 2.5, 4.5, 8.5, 16.5, 9e10};
  static const double b[] = {0.0, .5, 1.0, 1.5, 2.0, 2.5, 4.5, 8.5, 9e10};
  return (float)(limited ? b[idx] : a[idx]);
}

}  // namespace vamd
