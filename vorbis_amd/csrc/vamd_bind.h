// vamd_bind.h -- host-side: validate a setup blob, append the derived index
// tables (vamd_derive.h) and bind the kernel parameter structs to a base
// address (the HBM copy for the product, the host copy for tests/emul).
#pragma once
#include <math.h>
#include <stdlib.h>
#include <stddef.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "vamd_derive.h"
#include "vamd_params.h"
#include "vorbis_amd.h"

namespace vamd {

#define VAMD_QPL_GPU 4  // must equal VAMD_QPL of the HIP build (vamd_wave.h)

struct Bound {
  int channels, rate;
  int bs[2];
  XformP xf[2];
  PsyP psy[4];
  FloorP floor[2][VAMD_MAX_SUBMAPS];
  ChMap chmap[2];
  CoupleP couple[2];
  CoupleSet couple_all[2];
  EnvP env;
  ResP res[2][VAMD_MAX_SUBMAPS];
  PackP pack[2];
  int res_cap[2];         // entries per block over all submaps (row length of res_entries); 0 = not covered
  int res_lds_ints[2];    // LDS ints k_residue needs for the largest submap of the mode
  int res_off_ints[2];    // largest stages*slots + 1 over the mode's submaps (k_pack's offset arrays)
  float ampmax_att_per_sec;
  QLimitP qlimit[2];      // the input domain's integer edge per size class and channel (derive_quant_limit)
};

// Checks the blob and produces `image` = blob + derived tables (what gets copied to HBM).
// Returns VAMD_OK or an OV_*-valued error with `err` set.
inline int build_image(const void *blob_v, size_t bytes, std::vector<unsigned char> *image,
                       std::vector<uint32_t> *derived_off /* per psy: run_start, seed_span */,
                       std::vector<PsyDerived> *derived, std::string *err) {
  const unsigned char *blob = (const unsigned char *)blob_v;
  if (!blob || bytes < sizeof(vamd_setup_header)) {
    *err = "setup blob missing or truncated";
    return VAMD_EINVAL;
  }
  vamd_setup_header h;
  memcpy(&h, blob, sizeof(h));
  if (h.magic != VAMD_SETUP_MAGIC) {
    *err = "setup blob: bad magic";
    return VAMD_EINVAL;
  }
  if (h.version != VAMD_SETUP_VERSION) {
    *err = "setup blob: version mismatch";
    return VAMD_EVERSION;
  }
  if (h.total_bytes > bytes || h.total_bytes < sizeof(vamd_setup_header)) {
    *err = "setup blob: total_bytes exceeds the buffer or is shorter than the header";
    return VAMD_EINVAL;
  }
  if (h.channels < 1 || h.channels > VAMD_MAX_CH) {
    *err = "channel count not covered (1 or 2)";
    return VAMD_EIMPL;
  }
  for (int W = 0; W < 2; W++) {
    const vamd_xform_tab &x = h.xform[W];
    if (x.n != h.blocksizes[W] || x.n < 64 || x.n > 8192 || (x.n & (x.n - 1))) {
      *err = "block size must be a power of two in [64, 8192]";
      return VAMD_EINVAL;
    }
    if (x.n > 4096) {
      *err = "block sizes above 4096 are not covered";
      return VAMD_EIMPL;
    }
    if ((1 << x.log2n) != x.n) {
      *err = "transform table: log2n does not match n";
      return VAMD_EINVAL;
    }
    {
      const uint64_t need[] = {(uint64_t)x.off_mdct_trig + 4ull * (x.n + x.n / 4), (uint64_t)x.off_mdct_bitrev + 4ull * (x.n / 4),
                               (uint64_t)x.off_fft_wa + 8ull * x.n, (uint64_t)x.off_window + 4ull * (x.n / 2)};
      for (uint64_t e : need)
        if (e > h.total_bytes) {
          *err = "setup blob: transform table offset out of range";
          return VAMD_EINVAL;
        }
      if ((x.off_mdct_trig | x.off_mdct_bitrev | x.off_fft_wa | x.off_window) & 3) {
        *err = "setup blob: transform table misaligned";
        return VAMD_EINVAL;
      }
      // mdct_bitreverse gathers through this table (lib/mdct.c:346-394): entries index the n/2 work vector
      const int32_t *br = (const int32_t *)(blob + x.off_mdct_bitrev);
      for (int i = 0; i < x.n / 4; i++)
        if (br[i] < 0 || br[i] > x.n / 2 - 2 || (br[i] & 1)) {
          *err = "setup blob: MDCT bit-reverse entry out of range";
          return VAMD_EINVAL;
        }
    }
    if (x.fft_nf < 1 || x.fft_nf > 8) {
      *err = "unsupported FFT factorisation";
      return VAMD_EIMPL;
    }
    {
      long prod = 1;
      for (int i = 0; i < x.fft_nf; i++) prod *= x.fft_fac[i];
      if (prod != x.n) {
        *err = "FFT factors do not multiply to the block size";
        return VAMD_EINVAL;
      }
    }
    for (int i = 0; i < x.fft_nf; i++)
      if (x.fft_fac[i] != 2 && x.fft_fac[i] != 4) {
        *err = "FFT radix other than 2/4 (dradfg) is not covered";
        return VAMD_EIMPL;
      }
    const vamd_mode_tab &m = h.mode[W];
    if (m.submaps < 1 || m.submaps > VAMD_MAX_SUBMAPS || m.coupling_steps < 0 || m.coupling_steps > VAMD_MAX_COUPLING) {
      *err = "mapping: submaps / coupling steps out of the covered range";
      return VAMD_EIMPL;
    }
    for (int i = 0; i < m.coupling_steps; i++)
      if (m.coupling_mag[i] < 0 || m.coupling_mag[i] >= h.channels || m.coupling_ang[i] < 0 ||
          m.coupling_ang[i] >= h.channels || m.coupling_mag[i] == m.coupling_ang[i]) {
        *err = "mapping: coupling channel out of range";
        return VAMD_EINVAL;
      }
    for (int c = 0; c < h.channels; c++)
      if (m.chmuxlist[c] < 0 || m.chmuxlist[c] >= m.submaps) {
        *err = "mapping: channel multiplex out of range";
        return VAMD_EINVAL;
      }
    for (int sm = 0; sm < m.submaps; sm++) {
      if (m.floor[sm].posts < 2 || m.floor[sm].posts > VAMD_POSTS_STRIDE) {
        *err = "floor1 post count out of range";
        return VAMD_EIMPL;
      }
      if (m.floor[sm].look_n > x.n / 2 || m.floor[sm].look_n < 1) {
        *err = "floor1 range exceeds the block";
        return VAMD_EINVAL;
      }
      // the fit walks these as indices (lib/floor1.c:576-729): posts inside the range, index maps inside the posts
      const vamd_floor1_tab &f = m.floor[sm];
      bool ok = f.mult >= 1 && f.mult <= 4 && f.quant_q >= 1 && f.quant_q <= 256;
      for (int i = 0; ok && i < f.posts; i++)
        ok = f.postlist[i] >= 0 && f.postlist[i] <= f.look_n && f.sorted_index[i] >= 0 && f.sorted_index[i] <= f.look_n &&
             f.forward_index[i] >= 0 && f.forward_index[i] < f.posts && f.reverse_index[i] >= 0 && f.reverse_index[i] < f.posts;
      for (int i = 0; ok && i + 1 < f.posts; i++) ok = f.sorted_index[i] < f.sorted_index[i + 1];
      for (int i = 0; ok && i + 2 < f.posts; i++)  // neighbours of post i+2: earlier posts only (what makes the level order valid)
        ok = f.hineighbor[i] >= 0 && f.hineighbor[i] < i + 2 && f.loneighbor[i] >= 0 && f.loneighbor[i] < i + 2 &&
             f.postlist[f.loneighbor[i]] < f.postlist[f.hineighbor[i]];
      if (!ok) {
        *err = "setup blob: floor1 post / neighbour tables out of range";
        return VAMD_EINVAL;
      }
    }
  }
  for (int p = 0; p < 4; p++) {
    const vamd_psy_tab &t = h.psy[p];
    if (t.n != h.blocksizes[p >> 1] / 2) {
      *err = "psy look size does not match its block size";
      return VAMD_EINVAL;
    }
    const uint64_t need[] = {(uint64_t)t.off_ath + 4ull * t.n, (uint64_t)t.off_octave + 4ull * t.n,
                             (uint64_t)t.off_bark + 4ull * t.n, (uint64_t)t.off_noiseoffset + 12ull * t.n,
                             (uint64_t)t.off_tonecurves + 4ull * 17 * 8 * 58};
    for (uint64_t e : need)
      if (e > h.total_bytes) {
        *err = "setup blob: table offset out of range";
        return VAMD_EINVAL;
      }
    if (t.total_octave_lines != h.psy[p & ~1].total_octave_lines ||
        t.eighth_octave_lines != h.psy[p & ~1].eighth_octave_lines) {
      *err = "psy looks of one size class must share their octave-line geometry";
      return VAMD_EIMPL;
    }
    if (t.total_octave_lines < 1 || t.total_octave_lines > 4096) {
      *err = "total_octave_lines out of range";
      return VAMD_EINVAL;
    }
    if ((t.off_ath | t.off_octave | t.off_bark | t.off_noiseoffset | t.off_tonecurves) & 3) {
      *err = "setup blob: psy table misaligned";
      return VAMD_EINVAL;
    }
    if (t.eighth_octave_lines < 1 || t.eighth_octave_lines > 64 || t.shiftoc < 0 || t.shiftoc > 16) {
      *err = "psy octave geometry out of range";
      return VAMD_EINVAL;
    }
    // seed_chase's ring of VAMD_RING stack entries and its "an entry a ring-length below the top is final" rule hold
    // for linesper <= VAMD_RING (k_tone.h); libvorbisenc sets 8 everywhere
    if (t.eighth_octave_lines > 16) {
      *err = "eighth_octave_lines above 16 is outside the covered path (seed_chase's ring holds 16 entries)";
      return VAMD_EIMPL;
    }
    // tables the kernels use as indices: octave[] is a non-decreasing line position inside the seed vector
    // (seed_loop / max_seeds, lib/psy.c:417-545); bark[] packs window edges inside the block or its mirror
    // (bark_noise_hybridmp, lib/psy.c:606-656)
    {
      const int32_t *oc = (const int32_t *)(blob + t.off_octave), *bk = (const int32_t *)(blob + t.off_bark);
      for (int i = 0; i < t.n; i++) {
        const long line = (long)oc[i] - t.firstoc;
        if (line < 0 || line >= t.total_octave_lines || (i && oc[i] < oc[i - 1])) {
          *err = "setup blob: octave[] entry outside the seed vector";
          return VAMD_EINVAL;
        }
        const int lo = bk[i] >> 16, hi = bk[i] & 0xffff;
        if (lo <= -32768 + 1 || lo > t.n || hi > 2 * t.n) {
          *err = "setup blob: bark[] window edge out of range";
          return VAMD_EINVAL;
        }
      }
      const float *tc = (const float *)(blob + t.off_tonecurves);
      for (int c = 0; c < VAMD_P_BANDS * VAMD_P_LEVELS; c++) {
        const float p0 = tc[(size_t)c * (VAMD_EHMER_MAX + 2)], p1 = tc[(size_t)c * (VAMD_EHMER_MAX + 2) + 1];
        if (!(p0 >= 0.f && p0 <= (float)VAMD_EHMER_MAX && p1 >= 0.f && p1 <= (float)VAMD_EHMER_MAX)) {
          *err = "setup blob: tone curve fence posts out of range";
          return VAMD_EINVAL;
        }
      }
    }
    if (t.normal_p && (t.normal_partition < 1 || t.normal_partition > t.n || t.normal_start < 0)) {
      *err = "setup blob: noise normalisation partition out of range";
      return VAMD_EINVAL;
    }
    if (t.noisewindowfixed > (1 << 20)) {  // (<= 0: the pass is off; wider than the block: every bin extends nothing)
      *err = "setup blob: fixed noise window out of range";
      return VAMD_EINVAL;
    }
  }

  {
    const vamd_envelope_tab &e = h.env;
    const int n = e.winlength;
    if (n != 128 || (1 << e.log2n) != n || e.searchstep != 64) {  // (k_env_spectrum stages an item's samples in 64-sample chunks)
      *err = "envelope detector: only the 128-sample window and the 64-sample step of lib/envelope.c:35-37 are covered";
      return VAMD_EIMPL;
    }
    const uint64_t need[] = {(uint64_t)e.off_mdct_trig + 4ull * (n + n / 4), (uint64_t)e.off_mdct_bitrev + 4ull * (n / 4),
                             (uint64_t)e.off_window + 4ull * n};
    for (uint64_t x : need)
      if (x > h.total_bytes) {
        *err = "setup blob: envelope table offset out of range";
        return VAMD_EINVAL;
      }
    for (int i = 0; i < VAMD_VE_BANDS; i++)
      if (e.band_begin[i] < 0 || e.band_end[i] < 0 || e.band_end[i] > VAMD_VE_BANDWIN ||
          e.band_begin[i] + e.band_end[i] > n / 4) {
        *err = "envelope detector: band out of range";
        return VAMD_EINVAL;
      }
  }

  if (h.nbooks < 0 || h.nbooks > 256 || (uint64_t)h.off_books + (uint64_t)h.nbooks * sizeof(vamd_book_tab) > h.total_bytes) {
    *err = "setup blob: codebook table out of range";
    return VAMD_EINVAL;
  }
  for (int i = 0; i < h.nbooks; i++) {
    vamd_book_tab bk;
    memcpy(&bk, blob + h.off_books + (size_t)i * sizeof(bk), sizeof(bk));
    if (bk.dim < 1 || bk.entries < 0 || (uint64_t)bk.off_lengths + (uint64_t)bk.entries > h.total_bytes ||
        (bk.off_codes & 3) || (uint64_t)bk.off_codes + 4 * (uint64_t)bk.entries > h.total_bytes) {
      *err = "setup blob: codebook out of range";
      return VAMD_EINVAL;
    }
  }
  if (h.modebits < 0 || h.modebits > 8) {
    *err = "setup blob: mode number width out of range";
    return VAMD_EINVAL;
  }
  for (int W = 0; W < 2; W++)
   for (int sm = 0; sm < h.mode[W].submaps; sm++) {
    const vamd_floor1_tab &f = h.mode[W].floor[sm];
    int covered = 2;
    bool ok = f.partitions >= 0 && f.partitions <= VAMD_FLOOR_PARTS;
    for (int i = 0; ok && i < f.partitions; i++) {
      const int c = f.partitionclass[i];
      ok = c >= 0 && c < VAMD_FLOOR_CLASSES && f.class_dim[c] >= 1 && f.class_dim[c] <= 8 && f.class_subs[c] >= 0 &&
           f.class_subs[c] <= 3 && f.class_book[c] < h.nbooks && (f.class_subs[c] == 0 || f.class_book[c] >= 0);
      for (int k = 0; ok && k < 8; k++) ok = f.class_subbook[c][k] < h.nbooks;
      if (ok) covered += f.class_dim[c];
    }
    if (!ok || covered != f.posts) {
      *err = "setup blob: floor1 partition tables out of range";
      return VAMD_EINVAL;
    }
  }
  for (int W = 0; W < 2; W++)
   for (int sm = 0; sm < h.mode[W].submaps; sm++) {
    const vamd_residue_tab &r = h.res[W][sm];
    if (r.groupbook < 0 || r.groupbook >= h.nbooks || r.groupbook_dim < 1) {
      *err = "setup blob: residue phrase book out of range";
      return VAMD_EINVAL;
    }
    if (r.partitions < 1 || r.partitions > VAMD_RES_MAXCLASS || r.stages < 0 || r.stages > VAMD_RES_MAXSTAGE ||
        r.grouping < 1 || r.begin < 0 || r.end < r.begin) {
      *err = "setup blob: residue table out of range";
      return VAMD_EINVAL;
    }
    {
      // the coded range lies inside the submap's bundle (type 2: the channels interleaved; types 0/1: one channel)
      int bundle = 0;
      for (int c = 0; c < h.channels; c++) bundle += h.mode[W].chmuxlist[c] == sm;
      const long span = (long)(h.blocksizes[W] / 2) * (r.type == 2 ? (bundle > 0 ? bundle : 1) : 1);
      if (r.end > span) {
        *err = "setup blob: residue range exceeds its bundle";
        return VAMD_EINVAL;
      }
    }
    for (int c = 0; c < r.partitions; c++)
      for (int s = 0; s < VAMD_RES_MAXSTAGE; s++)
        if (r.partbooks[c][s] >= h.nbooks) {
          *err = "setup blob: residue book number out of range";
          return VAMD_EINVAL;
        }
  }

  image->assign(blob, blob + h.total_bytes);
  derived->clear();
  derived_off->clear();
  for (int p = 0; p < 4; p++) {
    PsyDerived d = derive_psy(h.psy[p], blob);
    while (image->size() & 15) image->push_back(0);
    derived_off->push_back((uint32_t)image->size());
    const unsigned char *a = (const unsigned char *)d.run_start.data();
    image->insert(image->end(), a, a + 4 * d.run_start.size());
    while (image->size() & 15) image->push_back(0);
    derived_off->push_back((uint32_t)image->size());
    const unsigned char *b = (const unsigned char *)d.seed_span.data();
    image->insert(image->end(), b, b + 4 * d.seed_span.size());
    derived->push_back(d);
  }
  // the transform stage hands the tone stage logfft's peak per run of bins (k_transform.h, run_peak) before it knows a
  // block's type: the runs follow from the size and the rate alone (octave[], lib/psy.c:327-329), so the two looks of a
  // size class share them -- checked, not assumed
  for (int W = 0; W < 2; W++)
    if ((*derived)[2 * W].run_start != (*derived)[2 * W + 1].run_start) {
      *err = "the two psy looks of a block size disagree on octave[] (outside the covered path)";
      return VAMD_EIMPL;
    }
  for (int p = 0; p < 4; p++) {  // slots 8..15: per psy RunRec[] and the re-strided tone curves
    const PsyDerived &d = (*derived)[p];
    while (image->size() & 255) image->push_back(0);
    derived_off->push_back((uint32_t)image->size());
    const unsigned char *a = (const unsigned char *)d.runs.data();
    image->insert(image->end(), a, a + sizeof(RunRec) * d.runs.size());
    while (image->size() & 255) image->push_back(0);
    derived_off->push_back((uint32_t)image->size());
    const unsigned char *b = (const unsigned char *)d.curves64.data();
    image->insert(image->end(), b, b + 4 * d.curves64.size());
  }
  for (int p = 0; p < 4; p++) {  // slots 16..23: per psy bin_fold[] and line_group[]
    const PsyDerived &d = (*derived)[p];
    while (image->size() & 15) image->push_back(0);
    derived_off->push_back((uint32_t)image->size());
    const unsigned char *a = (const unsigned char *)d.bin_fold.data();
    image->insert(image->end(), a, a + 4 * d.bin_fold.size());
    while (image->size() & 15) image->push_back(0);
    derived_off->push_back((uint32_t)image->size());
    const unsigned char *b = (const unsigned char *)d.line_group.data();
    image->insert(image->end(), b, b + 2 * d.line_group.size());
  }
  for (int W = 0; W < 2; W++)
    for (int sm = 0; sm < VAMD_MAX_SUBMAPS; sm++) {  // slots 24 + 2W + sm: bin -> fit interval, per floor
      const vamd_floor1_tab &f = h.mode[W].floor[sm < h.mode[W].submaps ? sm : 0];
      std::vector<unsigned char> bi = derive_bin_interval(f, h.blocksizes[W] / 2);
      while (image->size() & 15) image->push_back(0);
      derived_off->push_back((uint32_t)image->size());
      image->insert(image->end(), bi.begin(), bi.end());
    }
  for (int W = 0; W < 2; W++)
    for (int sm = 0; sm < VAMD_MAX_SUBMAPS; sm++) {  // slots 28 + 2W + sm: post levels
      int nl = 0;
      std::vector<int32_t> lv = derive_post_levels(h.mode[W].floor[sm < h.mode[W].submaps ? sm : 0], &nl);
      while (image->size() & 15) image->push_back(0);
      derived_off->push_back((uint32_t)image->size());
      const unsigned char *a = (const unsigned char *)lv.data();
      image->insert(image->end(), a, a + 4 * lv.size());
    }
  for (int W = 0; W < 2; W++)
    for (int sm = 0; sm < VAMD_MAX_SUBMAPS; sm++) {  // slots 32 + 2W + sm: accumulate_fit work list
      int ns = 0;
      std::vector<uint32_t> sg =
          derive_fit_segments(h.mode[W].floor[sm < h.mode[W].submaps ? sm : 0], h.blocksizes[W] / 2, &ns);
      while (image->size() & 15) image->push_back(0);
      derived_off->push_back((uint32_t)image->size());
      const unsigned char *a = (const unsigned char *)sg.data();
      image->insert(image->end(), a, a + 4 * sg.size());
    }
  {  // slot 36: div_magic's multipliers
    std::vector<uint32_t> mg = derive_div_magic();
    while (image->size() & 15) image->push_back(0);
    derived_off->push_back((uint32_t)image->size());
    const unsigned char *a = (const unsigned char *)mg.data();
    image->insert(image->end(), a, a + 4 * mg.size());
  }
  for (int W = 0; W < 2; W++) {  // slots 37 + W: bin -> run of bins of one octave line (run_start's inverse), per size class
    const std::vector<int32_t> &rs = (*derived)[2 * W].run_start;
    std::vector<uint16_t> rob((size_t)((h.blocksizes[W] / 2 + 7) & ~7), 0);
    for (size_t r = 0; r + 1 < rs.size(); r++)
      for (int i = rs[r]; i < rs[r + 1]; i++) rob[(size_t)i] = (uint16_t)r;
    while (image->size() & 15) image->push_back(0);
    derived_off->push_back((uint32_t)image->size());
    const unsigned char *a = (const unsigned char *)rob.data();
    image->insert(image->end(), a, a + 2 * rob.size());
  }
  for (int W = 0; W < 2; W++)
    for (int sm = 0; sm < VAMD_MAX_SUBMAPS; sm++) {
      // slots 39 + 2 W + sm: what the residue search needs of its tables, packed for one trip into LDS (residue_block_chunks,
      // k_residue.h): classmetric1 [partitions], classmetric2 [partitions], then per (class, stage) a ResStage
      const vamd_residue_tab &r = h.res[W][sm < h.mode[W].submaps ? sm : 0];
      std::vector<int32_t> tab;
      const int np = r.partitions >= 0 && r.partitions <= VAMD_RES_MAXCLASS ? r.partitions : 0;
      const int nst = r.stages >= 0 && r.stages <= VAMD_RES_MAXSTAGE ? r.stages : 0;
      for (int c = 0; c < np; c++) tab.push_back(r.classmetric1[c]);
      for (int c = 0; c < np; c++) tab.push_back(r.classmetric2[c]);
      while (tab.size() & 3) tab.push_back(0);
      const vamd_book_tab *bks = (const vamd_book_tab *)(blob + h.off_books);
      for (int c = 0; c < np; c++)
        for (int st = 0; st < nst; st++) {
          ResStage e;
          memset(&e, 0, sizeof(e));
          e.bn = -1;
          const int bn = ((r.secondstages[c] >> st) & 1) ? r.partbooks[c][st] : -1;
          if (bn >= 0 && bn < h.nbooks && bks[bn].dim >= 1 && (size_t)bks[bn].off_lengths + (size_t)bks[bn].entries <= h.total_bytes) {
            const vamd_book_tab &bk = bks[bn];
            e.bn = bn, e.dim = bk.dim, e.minval = bk.minval, e.delta = bk.delta, e.quantvals = bk.quantvals;
            e.off_lengths = (int32_t)bk.off_lengths, e.entries = bk.entries;
            e.nv = r.grouping / bk.dim;
            const signed char *len = (const signed char *)(blob + bk.off_lengths);
            e.full = 1;  // every entry populated: local_book_besterror never takes its exhaustive search
            for (int i = 0; i < bk.entries; i++)
              if (len[i] <= 0) e.full = 0;
          }
          const int32_t *w = (const int32_t *)&e;
          tab.insert(tab.end(), w, w + sizeof(e) / 4);
        }
      while (image->size() & 15) image->push_back(0);
      derived_off->push_back((uint32_t)image->size());
      const unsigned char *a = (const unsigned char *)tab.data();
      image->insert(image->end(), a, a + 4 * tab.size());
    }
  while (image->size() & 15) image->push_back(0);
  return VAMD_OK;
}

// inspect_error's comparisons whose operands do not depend on the audio (lib/floor1.c:536-563), settled here with
// the reference's own float expressions:
//  * "maxover*maxover/n > maxerr" for a point count n: the quotient falls as n grows, so the test is n <= cnt_over;
//  * "y + maxover < val" / "y - maxunder > val" with y, val integers in [0, 1023]: when maxover is a multiple of
//    2^-13 below 1024 the float sum is exact and the test is the integer val - y >= floor(maxover) + 1 (likewise
//    y - val >= floor(maxunder) + 1); otherwise the kernels keep the float form (int_tests = 0).
inline void floor_derive_tests(FloorP *F) {
  F->cnt_over = F->cnt_under = 0;
  for (int n = 1; n <= 65536; n++) {
    if (F->maxover * F->maxover / (float)n > F->maxerr) F->cnt_over = n;
    if (F->maxunder * F->maxunder / (float)n > F->maxerr) F->cnt_under = n;
  }
  auto nice = [](float v) { return fabsf(v) < 1024.f && v * 8192.f == floorf(v * 8192.f); };
  F->int_tests = nice(F->maxover) && nice(F->maxunder) && F->maxover >= 0.f && F->maxunder >= 0.f;
  F->over_i = F->int_tests ? (int)floorf(F->maxover) + 1 : 0;
  F->under_i = F->int_tests ? (int)floorf(F->maxunder) + 1 : 0;
}

// The input domain's integer edge (include/vorbis_amd.h): the largest |quantised value| Q of a block for which every
// integer the reference forms downstream of _vp_couple_quantize_normalize is defined by C, for THIS setup.
//  * lib/psy.c:958-962: out = rint(sqrt(ve)) converted to int -- below 2^31 (VAMD_QUANT_LIMIT_INT) -- and, where noise
//    normalisation is at work (normal_p, bins from normal_start on), squared in an int (:985) -- |out| <= 46340
//    (VAMD_QUANT_LIMIT_SQUARE).  (A value the coupling stage leaves is never smaller than what it was first quantised
//    to: a coupling step's magnitude keeps the larger of its two inputs, :1141-1166.)  Both are k_couple's to test.
//  * lib/res0.c:322-382, local_book_besterror on a vector a of dim <= 8 values, stage after stage (:585-640).  With
//    |a_j| <= A:  (a - minval + del/2) / del, v * del + minval and the index arithmetic stay below 2^31 for any A < 2^29;
//    the exhaustive search (:349-376) sums (e_j - a_j)^2 over the dim coordinates for EVERY populated entry e, whose
//    coordinates lie in [minval, minval + del * (quantvals - 1)], |e_j| <= E(book): the sum is at most
//    dim * (A + E)^2, defined while A + E <= R(dim) = floor(sqrt((2^31 - 1) / dim)).  The stage then leaves a - p with
//    p the (unclamped) nearest lattice point, |a_j - p_j| <= del <= max(A, E), or the best entry, |a_j - p_j| <= A + E:
//    after the stages t < s of a class's cascade |a_j| <= Q + sum E(book_t).  So the cascade of class c is defined
//    for Q <= min over its stages s of R(dim_s) - sum_{t <= s} E(book_t).  Which class a partition lands in depends on
//    its values: every class but the last takes a partition only while its largest magnitudes are within the class's
//    metrics (_2class :509-512: magmax <= classmetric1 && angmax <= classmetric2; _01class :447-450: max <=
//    classmetric1), so such a class constrains Q only if values up to its own metrics could break its cascade (never,
//    for libvorbisenc's books: checked here all the same); the last class takes whatever is left, and its cascade
//    sets the bound.  The search only ever sees the bins [begin, end) its residue codes (a 5.1 setup's LFE submap:
//    bins 0..11 -- a full-scale sine above them quantises to ~24 000 against a floor of -140 dB, coded by nobody, and
//    is the reference's defined result): elsewhere only the first bullet applies (QLimitP).
//  * _01class / _2class (:412-532) add up to `grouping` absolute values: far below 2^31 at these magnitudes.
// For the libvorbisenc setups the last class's cascade starts on a two-dimensional book reaching a few thousand:
// Q = 32 767 - ~2 000 ... 23 170 - ~10, spectra ~ +87 ... +90 dB over full scale.
inline int derive_quant_limit(const vamd_setup_header &h, const vamd_book_tab *hb, int W, int sm) {
  long q = VAMD_QUANT_LIMIT_INT;
  {
    const vamd_residue_tab &r = h.res[W][sm];
    for (int c = 0; c < r.partitions && c < VAMD_RES_MAXCLASS; c++) {
      long reach = 0, qc = VAMD_QUANT_LIMIT_INT;
      for (int s = 0; s < r.stages && s < VAMD_RES_MAXSTAGE; s++) {
        if (!((r.secondstages[c] >> s) & 1) || r.partbooks[c][s] < 0 || r.partbooks[c][s] >= h.nbooks) continue;
        const vamd_book_tab &bk = hb[r.partbooks[c][s]];
        const long lo = bk.minval, hi = (long)bk.minval + (long)bk.delta * (bk.quantvals - 1);
        long e = std::max(std::labs(lo), std::labs(hi));
        e = std::max(e, std::labs((long)bk.delta));
        reach += e;
        const int dim = bk.dim < 1 ? 1 : (bk.dim > 8 ? 8 : bk.dim);
        long R = (long)floor(sqrt(2147483647.0 / dim));
        while ((R + 1) * (R + 1) * dim <= 2147483647L) R++;
        while (R * R * dim > 2147483647L) R--;
        qc = std::min(qc, R - reach);
      }
      // the largest magnitude a partition of this class can hold; unbounded for the last class (and for a metric the
      // format does not bound: a negative classmetric2 switches the second test off only for type 1, where it is a mean)
      const bool last = c == r.partitions - 1;
      long held = std::max(0, (int)r.classmetric1[c]);
      if (r.type == 2) held = std::max(held, (long)std::max(0, (int)r.classmetric2[c]));
      if (last || held > qc) q = std::min(q, qc);
    }
  }
  return (int)std::max(q, 0L);
}

// Bind parameter structs to `base` (address of the image in the memory space
// the kernels will read: device pointer for HIP, host pointer for tests/emul).
inline void bind_params(const std::vector<unsigned char> &image, const std::vector<uint32_t> &derived_off,
                        const std::vector<PsyDerived> &derived, const unsigned char *base, Bound *B) {
  vamd_setup_header h;
  memcpy(&h, image.data(), sizeof(h));
  B->channels = h.channels;
  B->rate = h.rate;
  B->bs[0] = h.blocksizes[0];
  B->bs[1] = h.blocksizes[1];
  B->ampmax_att_per_sec = h.psy_g.ampmax_att_per_sec;
  for (int W = 0; W < 2; W++) {
    const vamd_xform_tab &x = h.xform[W];
    XformP &X = B->xf[W];
    X.n = x.n;
    X.log2n = x.log2n;
    X.mdct_scale = x.mdct_scale;
    X.trig = (const float *)(base + x.off_mdct_trig);
    X.bitrev = (const int *)(base + x.off_mdct_bitrev);
    X.wa = (const float *)(base + x.off_fft_wa);
    X.win_long = (const float *)(base + h.xform[1].off_window);
    X.win_short = (const float *)(base + h.xform[0].off_window);
    X.bs0 = h.blocksizes[0];
    X.bs1 = h.blocksizes[1];
    X.fft_nf = x.fft_nf;
    for (int i = 0; i < 8; i++) X.fft_fac[i] = i < x.fft_nf ? x.fft_fac[i] : 0;
    X.tpack = nullptr;
    {  // is the bit-reverse table the one mdct_init builds (lib/mdct.c:77-88)?  Then kernels need not fetch it.
      const int32_t *br = (const int32_t *)(image.data() + x.off_mdct_bitrev);
      const int lb = x.log2n - 1, mask = (1 << lb) - 1;
      X.bitrev_std = x.log2n >= 4 && x.log2n <= 16 && (1 << x.log2n) == x.n;
      for (int i = 0; X.bitrev_std && i < x.n / 8; i++) {
        int acc = 0;
        for (int j = 0; j < lb; j++)
          if ((i >> (lb - 1 - j)) & 1) acc |= 1 << j;
        if (br[2 * i] != ((~acc) & mask) - 1 || br[2 * i + 1] != acc) X.bitrev_std = 0;
      }
    }

    B->chmap[W].submaps = h.mode[W].submaps;
    for (int c = 0; c < VAMD_MAX_CH; c++) B->chmap[W].sub[c] = c < h.channels ? (unsigned char)h.mode[W].chmuxlist[c] : 0;
    for (int sm = 0; sm < VAMD_MAX_SUBMAPS; sm++) {
      const int src = sm < h.mode[W].submaps ? sm : 0;  // unused slots mirror submap 0
      const vamd_floor1_tab &f = h.mode[W].floor[src];
      FloorP &F = B->floor[W][sm];
      F.posts = f.posts;
      F.look_n = f.look_n;
      F.quant_q = f.quant_q;
      F.mult = f.mult;
      F.maxover = f.maxover;
      F.maxunder = f.maxunder;
      F.maxerr = f.maxerr;
      F.twofitweight = f.twofitweight;
      F.twofitatten = f.twofitatten;
      floor_derive_tests(&F);
      const size_t fo = offsetof(vamd_setup_header, mode) + sizeof(vamd_mode_tab) * W + offsetof(vamd_mode_tab, floor) +
                        sizeof(vamd_floor1_tab) * src;
      F.postlist = (const int *)(base + fo + offsetof(vamd_floor1_tab, postlist));
      F.sorted_index = (const int *)(base + fo + offsetof(vamd_floor1_tab, sorted_index));
      F.forward_index = (const int *)(base + fo + offsetof(vamd_floor1_tab, forward_index));
      F.reverse_index = (const int *)(base + fo + offsetof(vamd_floor1_tab, reverse_index));
      F.hineighbor = (const int *)(base + fo + offsetof(vamd_floor1_tab, hineighbor));
      F.loneighbor = (const int *)(base + fo + offsetof(vamd_floor1_tab, loneighbor));
      F.bin_interval = base + derived_off[24 + 2 * W + src];
      F.level = (const int *)(base + derived_off[28 + 2 * W + src]);
      int nl = 0;
      derive_post_levels(f, &nl);
      F.nlevels = nl;
      F.fit_segs = (const unsigned int *)(base + derived_off[32 + 2 * W + src]);
      derive_fit_segments(f, h.blocksizes[W] / 2, &F.fit_nseg);
      F.div_magic = (const unsigned int *)(base + derived_off[36]);
    }

    CoupleP &C = B->couple[W];
    const int blob_k = VAMD_PACKETBLOBS / 2;
    C.ch = h.channels;
    C.coupling_steps = h.mode[W].coupling_steps;
    for (int i = 0; i < VAMD_MAX_COUPLING; i++) {
      C.mag[i] = (signed char)(i < C.coupling_steps ? h.mode[W].coupling_mag[i] : 0);
      C.ang[i] = (signed char)(i < C.coupling_steps ? h.mode[W].coupling_ang[i] : 0);
    }
    // lib/psy.c:1027-1029,1056-1057; blockflag of the psy looks of this W is W
    C.pointlimit = h.psy_g.coupling_pointlimit[W][blob_k];
    C.prepoint = stereo_threshold(h.psy_g.coupling_prepointamp[blob_k], false);
    C.postpoint = stereo_threshold(h.psy_g.coupling_postpointamp[blob_k], (x.n / 2) > 1000);
    C.sliding_lowpass = h.psy_g.sliding_lowpass[W][blob_k];
    for (int k = 0; k < VAMD_PACKETBLOBS; k++) {  // lib/psy.c:1027-1029,1056-1057 per blobno
      CoupleP &K = B->couple_all[W].c[k];
      K = C;
      K.pointlimit = h.psy_g.coupling_pointlimit[W][k];
      K.prepoint = stereo_threshold(h.psy_g.coupling_prepointamp[k], false);
      K.postpoint = stereo_threshold(h.psy_g.coupling_postpointamp[k], (x.n / 2) > 1000);
      K.sliding_lowpass = h.psy_g.sliding_lowpass[W][k];
    }
  }
  {
    const vamd_envelope_tab &e = h.env;
    EnvP &E = B->env;
    memset(&E, 0, sizeof(E));
    E.mdct.n = e.winlength;
    E.mdct.log2n = e.log2n;
    E.mdct.mdct_scale = e.mdct_scale;
    E.mdct.trig = (const float *)(base + e.off_mdct_trig);
    E.mdct.bitrev = (const int *)(base + e.off_mdct_bitrev);
    E.mdct.bitrev_std = 0;
    E.mdct.tpack = nullptr;
    E.win = (const float *)(base + e.off_window);
    E.searchstep = e.searchstep;
    E.minenergy = e.minenergy;
    E.stretch_penalty = e.stretch_penalty;
    for (int i = 0; i < VAMD_VE_BANDS; i++) {
      E.preecho_thresh[i] = e.preecho_thresh[i];
      E.postecho_thresh[i] = e.postecho_thresh[i];
      E.band_begin[i] = e.band_begin[i];
      E.band_end[i] = e.band_end[i];
      E.band_total[i] = e.band_total[i];
      for (int j = 0; j < VAMD_VE_BANDWIN; j++) E.band_window[i][j] = e.band_window[i][j];
    }
  }
  const vamd_book_tab *hb = (const vamd_book_tab *)(image.data() + h.off_books);
  for (int W = 0; W < 2; W++) {
    // a channel's values meet the residue search only at the bins its submap's residue codes: [begin, end) of the
    // channel's own bins for type 1, of the bundle's interleaved samples for type 2 (lib/res0.c:738-746,791-797)
    QLimitP &Q = B->qlimit[W];
    const vamd_mode_tab &md = h.mode[W];
    // noise normalisation (lib/psy.c:941-1010) squares the values it leaves from normal_start on, when it is switched on
    // at all; the two block types of a size class share the setting in every libvorbisenc setup (the earlier start
    // otherwise)
    Q.sq = h.blocksizes[W] / 2;
    for (int bt = 0; bt < 2; bt++) {
      const vamd_psy_tab &t = h.psy[2 * W + bt];
      if (t.normal_p) Q.sq = std::max(0, std::min(Q.sq, (int)t.normal_start));
    }
    for (int c = 0; c < VAMD_MAX_CH; c++) {
      Q.q[c] = VAMD_QUANT_LIMIT_INT, Q.lo[c] = Q.hi[c] = 0;
      if (c >= h.channels) continue;
      const int sm = md.chmuxlist[c] < md.submaps && md.chmuxlist[c] < VAMD_MAX_SUBMAPS ? md.chmuxlist[c] : 0;
      const vamd_residue_tab &r = h.res[W][sm];
      int bundle = 0;
      for (int k = 0; k < h.channels; k++) bundle += md.chmuxlist[k] == sm;
      const int per = r.type == 2 && bundle > 0 ? bundle : 1, n2 = h.blocksizes[W] / 2;
      Q.q[c] = derive_quant_limit(h, hb, W, sm);
      // channel c is the ci-th of its bundle: its bin j is the residue's position j * per + ci (type 2), or j (type 1)
      int ci = 0;
      for (int k = 0; k < c; k++) ci += md.chmuxlist[k] == md.chmuxlist[c];
      const int off = per > 1 ? ci : 0;
      auto ceil_div = [](int a, int b) { return a <= 0 ? 0 : (a + b - 1) / b; };
      Q.lo[c] = (short)std::min(n2, ceil_div(r.begin - off, per));
      Q.hi[c] = (short)std::min(n2, ceil_div(r.end - off, per));
    }
  }
  auto longest = [&](int bn) {  // longest codeword of a book
    int m = 0;
    if (bn < 0) return 0;
    const signed char *len = (const signed char *)(image.data() + hb[bn].off_lengths);
    for (int e = 0; e < hb[bn].entries; e++)
      if (len[e] > m) m = len[e];
    return m;
  };
  for (int W = 0; W < 2; W++) {
    const vamd_mode_tab &m = h.mode[W];
    const int n2 = h.blocksizes[W] / 2;
    PackP &K = B->pack[W];
    K.books = (const vamd_book_tab *)(base + h.off_books);
    K.nbooks = h.nbooks;
    K.base = base;
    K.modebits = h.modebits;
    long bits = 1 + K.modebits + 2;  // the longest packet this size class can produce, field by field
    long head = bits;                // ... and the longest header + floors part of one
    bool all_ok = true;
    int ent_base = 0, lds = 0, offi = 0;
    for (int sm = 0; sm < VAMD_MAX_SUBMAPS; sm++) {
      const int src = sm < m.submaps ? sm : 0;
      const vamd_residue_tab &r = h.res[W][src];
      ResP &Rp = B->res[W][sm];
      Rp.tab = (const vamd_residue_tab *)(base + offsetof(vamd_setup_header, res) +
                                          sizeof(vamd_residue_tab) * (W * VAMD_MAX_SUBMAPS + src));
      Rp.books = K.books;
      Rp.base = base;
      int bundle = 0;
      for (int c = 0; c < h.channels; c++) bundle += m.chmuxlist[c] == src;
      const int partvals = (r.end - r.begin) / r.grouping;
      // covered: the interleaved (type 2) residue of a bundle and the per-channel type-1 residue, holding
      // vectors of <= 8 dimensions that tile the partitions
      const int streams = r.type == 2 ? 1 : bundle;
      bool ok = sm < m.submaps && (r.type == 2 || r.type == 1) && bundle >= 1 &&
                r.end <= (r.type == 2 ? bundle : 1) * n2 && (r.type != 2 || (r.begin % bundle) == 0) &&
                partvals >= 1 && partvals * streams <= VAMD_RES_CLASS_STRIDE &&
                (r.type != 2 || r.grouping % bundle == 0);
      int worst = 0;
      long worst_bits = 0;
      // ... and, for residue_block_chunks: two channels interleaved, partitions of 8, 16 or 32 values starting on a
      // multiple of eight, every book's dimension a divisor of eight
      bool tiles8 = r.type == 2 && bundle == 2 && (r.grouping == 8 || r.grouping == 16 || r.grouping == 32) && (r.begin % 8) == 0;
      for (int c = 0; c < r.partitions && ok; c++) {
        int per = 0;
        long perb = 0;
        for (int s = 0; s < r.stages; s++)
          if (((r.secondstages[c] >> s) & 1) && r.partbooks[c][s] >= 0) {
            const vamd_book_tab &bk = hb[r.partbooks[c][s]];
            if (bk.dim < 1 || 8 % bk.dim) tiles8 = false;
            if (bk.dim > 8 || bk.dim > r.grouping || bk.entries > 65535 || r.grouping % bk.dim) ok = false;
            else per += r.grouping / bk.dim, perb += (long)(r.grouping / bk.dim) * longest(r.partbooks[c][s]);
          }
        if (per > worst) worst = per;
        if (perb > worst_bits) worst_bits = perb;
      }
      Rp.covered = ok ? 1 : 0;
      Rp.bundle = bundle;
      Rp.partvals = partvals;
      Rp.slots = partvals * streams;
      Rp.cap = ok ? worst * Rp.slots : 0;
      Rp.cls_base = sm * VAMD_RES_CLASS_STRIDE;
      Rp.ent_base = ent_base;
      Rp.lds_ints = bundle * n2 + VAMD_RES_CLASS_STRIDE + 2 * r.stages * Rp.slots + 1;
      Rp.qmax = derive_quant_limit(h, hb, W, src);
      Rp.chunked = ok && tiles8 ? 1 : 0;
      Rp.tab_grouping = r.grouping;
      Rp.begin = r.begin, Rp.nparts = r.partitions, Rp.nstages = r.stages;
      Rp.groupbook = r.groupbook, Rp.groupbook_dim = r.groupbook_dim;
      Rp.fast = (const int *)(base + derived_off[39 + 2 * W + sm]);
      Rp.fast_ints = ((2 * r.partitions + 3) & ~3) + r.partitions * r.stages * (int)(sizeof(ResStage) / 4);
      if (sm < m.submaps) {
        all_ok = all_ok && ok;
        ent_base += Rp.cap;
        lds = std::max(lds, bundle * n2 + VAMD_RES_CLASS_STRIDE + 2 * r.stages * Rp.slots + 1);
        offi = std::max(offi, r.stages * Rp.slots + 1);
        bits += worst_bits * Rp.slots + (long)((partvals + r.groupbook_dim - 1) / r.groupbook_dim) * streams * longest(r.groupbook);
      }
      // the floor of this submap's channels
      const vamd_floor1_tab &f = m.floor[src];
      K.ftab[sm] = (const vamd_floor1_tab *)(base + offsetof(vamd_setup_header, mode) + sizeof(vamd_mode_tab) * W +
                                             offsetof(vamd_mode_tab, floor) + sizeof(vamd_floor1_tab) * src);
      K.qbits[sm] = 0;
      for (unsigned v = f.quant_q > 0 ? (unsigned)(f.quant_q - 1) : 0; v; v >>= 1) K.qbits[sm]++;  // ov_ilog
      long fl = 1 + 2 * K.qbits[sm];
      for (int i = 0; i < f.partitions; i++) {
        const int c = f.partitionclass[i];
        int sub = 0;
        for (int k = 0; k < (1 << f.class_subs[c]); k++) sub = std::max(sub, longest(f.class_subbook[c][k]));
        fl += (f.class_subs[c] ? longest(f.class_book[c]) : 0) + f.class_dim[c] * sub;
      }
      if (sm < m.submaps) bits += fl * bundle, head += fl * bundle;
    }
    B->res_cap[W] = all_ok ? ent_base : 0;
    B->res_lds_ints[W] = lds;
    B->res_off_ints[W] = offi;
    if (!all_ok)
      for (int sm = 0; sm < VAMD_MAX_SUBMAPS; sm++) B->res[W][sm].covered = 0;
    // (k_pack_pair assembles the residue part a whole number of words past the longest possible head before it moves it
    // down to the real one: two words of slack)
    K.head_words = (int)((head + 31) / 32) + 1;
    K.capacity = all_ok ? (int)(((bits + 31) / 32) * 4) + 8 : 0;
  }
  for (int p = 0; p < 4; p++) {
    const vamd_psy_tab &t = h.psy[p];
    PsyP &P = B->psy[p];
    P.n = t.n;
    P.firstoc = t.firstoc;
    P.shiftoc = t.shiftoc;
    P.eighth_octave_lines = t.eighth_octave_lines;
    P.total_octave_lines = t.total_octave_lines;
    P.m_val = t.m_val;
    P.ath_adjatt = t.ath_adjatt;
    P.ath_maxatt = t.ath_maxatt;
    P.tone_masteratt1 = t.tone_masteratt[1];
    P.tone_masteratt0 = t.tone_masteratt[0];
    P.tone_masteratt2 = t.tone_masteratt[2];
    P.tone_abs_limit = t.tone_abs_limit;
    P.noisemaxsupp = t.noisemaxsupp;
    P.noisewindowfixed = t.noisewindowfixed;
    P.max_curve_dB = t.max_curve_dB;
    P.ath = (const float *)(base + t.off_ath);
    P.octave = (const int *)(base + t.off_octave);
    P.bark = (const int *)(base + t.off_bark);
    P.noiseoffset1 = (const float *)(base + t.off_noiseoffset) + t.n;
    P.noiseoffset0 = (const float *)(base + t.off_noiseoffset);
    P.noiseoffset2 = (const float *)(base + t.off_noiseoffset) + 2 * t.n;
    P.tonecurves = (const float *)(base + t.off_tonecurves);
    P.noisecompand = (const float *)(base + offsetof(vamd_setup_header, psy) + sizeof(vamd_psy_tab) * p +
                                     offsetof(vamd_psy_tab, noisecompand));
    const PsyDerived &d = derived[p];
    P.bark_i1 = d.bark_i1;
    P.bark_i2 = d.bark_i2;
    P.fix_i1 = d.fix_i1;
    P.fix_i2 = d.fix_i2;
    P.run_start = (const int *)(base + derived_off[2 * p]);
    P.nruns = (int)d.run_start.size() - 1;
    P.run_of_bin = (const unsigned short *)(base + derived_off[37 + (p >> 1)]);
    P.seed_span = (const int *)(base + derived_off[2 * p + 1]);
    P.runs = (const int *)(base + derived_off[8 + 2 * p]);
    P.curves64 = (const float *)(base + derived_off[8 + 2 * p + 1]);
    P.curve_stride = 64;
    P.bin_fold = (const int *)(base + derived_off[16 + 2 * p]);
    P.line_group = (const unsigned short *)(base + derived_off[16 + 2 * p + 1]);
    P.ngroups = d.ngroups;
    P.tail_linpos = d.tail_linpos;
    P.normal_p = t.normal_p;
    P.normal_start = t.normal_start;
    P.normal_partition = t.normal_partition;
    P.normal_thresh = t.normal_thresh;
  }
}

}  // namespace vamd
