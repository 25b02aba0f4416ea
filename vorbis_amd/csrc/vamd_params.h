// vamd_params.h -- by-value kernel parameter structs: pointers into the setup
// blob resident in HBM (include/vamd_setup.h) plus the scalars each stage reads.
#pragma once
#include <stdint.h>
#include "vamd_setup.h"

namespace vamd {

// The integer floor curve between the floor and the coupling stage (floor1_encode's render, lib/floor1.c:923-952):
// values are a post's quantised height times mult -- the fit's 0..1023 scale cut to 256 / mult steps (lib/floor1.c:
// 772-781), i.e. at most 255 whatever mult is -- and the lines between two of them, so they travel as bytes: the
// coupling stage, which moves its data at what HBM gives, reads a quarter of what an int curve would cost it, and
// the table look-up it feeds (floor1_fromdB) has 256 entries.  The int32 `ilogmask` tap of the C ABI is widened from
// it on request.
typedef unsigned char ilog_t;

// window + MDCT + FFT tables for one size class W
struct XformP {
  int n;                 // block size
  int log2n;
  float mdct_scale;      // 4/n
  const float *trig;     // [n + n/4]   mdct_lookup.trig   (lib/mdct.c:64-73)
  const int *bitrev;     // [n/4]       mdct_lookup.bitrev (lib/mdct.c:77-88)
  const float *wa;       // [2n]        drft_lookup.trigcache + n
  const float *win_long; // [blocksizes[1]/2] rising half window of the long size
  const float *win_short;// [blocksizes[0]/2]
  int bs0, bs1;          // blocksizes
  int fft_nf;
  int fft_fac[8];
  int bitrev_std;        // bitrev[] is lib/mdct.c:77-88's table for this n (vamd_create checks): kernels may compute it
  const float *tpack;    // the butterfly stages' trig pairs repacked per stage (mdct_tpack_fill), or null
};

// one vorbis_look_psy (+ the vorbis_info_psy scalars)
struct PsyP {
  int n;
  int firstoc, shiftoc, eighth_octave_lines, total_octave_lines;
  float m_val;
  float ath_adjatt, ath_maxatt;
  float tone_masteratt1;     // tone_masteratt[1]: VBR uses offset_select 1 only
  float tone_masteratt0, tone_masteratt2;  // the lo / hi curves of a bitrate-managed block
  float tone_abs_limit;
  float noisemaxsupp;
  int noisewindowfixed;
  float max_curve_dB;
  const float *ath;          // [n]
  const int *octave;         // [n]
  const int *bark;           // [n]
  const float *noiseoffset1; // [n]  noiseoffset[1]
  const float *noiseoffset0, *noiseoffset2;
  const float *tonecurves;   // [17][8][58]
  const float *noisecompand; // [40] (inside the blob header copy in HBM)
  // derived at vamd_create() from the static tables (host, once):
  int bark_i1, bark_i2;      // regime boundaries of bark_noise_hybridmp pass 1 (lib/psy.c:606-656)
  int fix_i1, fix_i2;        // same for the fixed-window pass (lib/psy.c:660-703)
  const int *run_start;      // [nruns+1] starts of runs of equal octave[] (lib/psy.c:429-435)
  int nruns;
  const unsigned short *run_of_bin;  // [n] the run a bin belongs to (the size class's: both block types share the runs)
  const int *runs;           // [nruns][4] RunRec (vamd_derive.h)
  const float *curves64;     // [17][8] tone-curve rows of 56 points, `curve_stride` floats apart
  int curve_stride;          // 64 in HBM (256-byte rows); 60 for an LDS copy (rows staggered over the banks)
  const int *seed_span;      // [n][2] octave-line span (pos0,pos1) each bin folds in max_seeds (lib/psy.c:524-537)
  const int *bin_fold;       // [n] p0 | group << 16
  const unsigned short *line_group;  // [nl16] group of each octave line, 0xffff = none
  int ngroups;
  int tail_linpos;           // first bin handled by max_seeds' tail loop (lib/psy.c:539-543)
  int normal_p, normal_start, normal_partition;
  double normal_thresh;
};

// the block-switching detector (_ve_amp / _ve_envelope_search, lib/envelope.c)
struct EnvP {
  XformP mdct;             // n = 128 MDCT tables only (no FFT, no vwin windows)
  const float *win;        // [n] mdct_win
  int searchstep;
  float minenergy, stretch_penalty;
  float preecho_thresh[VAMD_VE_BANDS], postecho_thresh[VAMD_VE_BANDS];
  int band_begin[VAMD_VE_BANDS], band_end[VAMD_VE_BANDS];
  float band_total[VAMD_VE_BANDS];
  float band_window[VAMD_VE_BANDS][VAMD_VE_BANDWIN];
};

// residue back-end of one mode and submap: tables stay in the HBM image
// one (class, stage) of a residue as residue_block_chunks wants it: the book and what local_book_besterror reads of it
struct ResStage {
  int32_t bn;           // book number, -1: the class has no book at this stage
  int32_t dim, minval, delta, quantvals;
  int32_t off_lengths;  // the book's codeword lengths, relative to the image
  int32_t full;         // every entry populated (no exhaustive search, the lengths are not looked at)
  int32_t nv;           // vectors of a partition: grouping / dim
  int32_t entries;
  int32_t pad[3];
};
struct ResP {
  const vamd_residue_tab *tab;
  const vamd_book_tab *books;
  const unsigned char *base;  // image base: books[].off_lengths are relative to it
  int cap;                    // entries a block can emit at most (sizes the output rows)
  int covered;                // 1 when the GPU handles this residue
  int bundle;                 // channels in this submap's bundle
  int partvals;               // (end - begin) / grouping
  int slots;                  // classified (partition, stream) pairs at most: partvals x streams
  int cls_base, ent_base;     // where this submap's rows start inside a block's res_class / res_entries rows
  int lds_ints;               // LDS ints k_residue needs for this submap
  int qmax;                   // largest |value| in [begin, end) for which the search's integers are defined (derive_quant_limit)
  int chunked;                // 1: a stereo type-2 residue whose vectors tile runs of eight values (residue_block_chunks, k_residue.h)
  int tab_grouping;           // tab->grouping, for the host (tab points into the HBM image)
  int begin, nparts, nstages; // tab->begin / partitions / stages, as kernel arguments
  int groupbook, groupbook_dim;  // tab->groupbook / groupbook_dim
  const int *fast;            // classmetric1 [partitions], classmetric2 [partitions] (padded to 4), ResStage [partitions][stages] (vamd_bind.h)
  int fast_ints;
};

// packet assembly (k_pack.h): the floor's class tables and the codebooks' codewords, in the HBM image
struct PackP {
  const vamd_floor1_tab *ftab[VAMD_MAX_SUBMAPS];
  const vamd_book_tab *books;
  int nbooks;
  const unsigned char *base;
  int modebits;                    // width of the mode number
  int qbits[VAMD_MAX_SUBMAPS];     // ilog(quant_q - 1): width of the two end posts
  int capacity;   // bytes the largest packet of this size class can take (multiple of 4); 0 = not assembled here
  int head_words; // words that hold the longest header + floors part of a packet, plus one
};

// which submap (floor, residue) each channel belongs to: vorbis_info_mapping0.chmuxlist
struct ChMap {
  int submaps;
  unsigned char sub[VAMD_MAX_CH];
};

#define VAMD_FITSEG_WORDS 12  // one record of FloorP::fit_segs
#define VAMD_DIV_MAGIC_MAX 2048  // the longest line of a floor: half the largest block (blocksizes <= 4096)

struct FloorP {
  int posts, look_n, quant_q, mult;
  float maxover, maxunder, maxerr, twofitweight, twofitatten;
  const int *postlist, *sorted_index, *forward_index, *reverse_index, *hineighbor, *loneighbor;
  const unsigned char *bin_interval;  // [n2] derived: accumulate_fit interval of each bin (255 = none)
  const int *level;                   // [64] derived: dependency level of each post
  int nlevels;
  const unsigned int *fit_segs;       // [fit_nseg][12] derived: accumulate_fit work list (derive_fit_segments)
  int fit_nseg;
  const unsigned int *div_magic;      // [VAMD_DIV_MAGIC_MAX + 1] derived: div_magic()'s multiplier per divisor
  // inspect_error's tests (lib/floor1.c:516-565) with the float work done once, at vamd_create (floor_derive_tests):
  int cnt_over, cnt_under;  // maxover^2 / n > maxerr holds exactly for the point counts n <= cnt_over (same for under)
  int int_tests;            // maxover / maxunder are multiples of 2^-13 below 1024: "y + maxover < val" is exact in
  int over_i, under_i;      //   integers: val - y >= over_i, resp. y - val >= under_i
};

// k_couple's estimate-then-verify margin (k_couple.h, chan_bin_sure): relative distance from a decision step inside
// which a bin is sent through the reference's own divisions.  The estimate and the reference's own roundings are
// together within 2.25 * 2^-23 of each other (derivation there); the steps are a whole number apart, so the share of
// bins sent to the exact path grows with the band AND with the bin's magnitude -- hence no wider than this.
#define VAMD_COUPLE_BAND 0x1p-21f

struct CoupleP {
  int ch;
  int coupling_steps;
  signed char mag[VAMD_MAX_COUPLING], ang[VAMD_MAX_COUPLING];  // applied in order (lib/psy.c:1111-1201)
  int pointlimit;        // coupling_pointlimit[blockflag][PACKETBLOBS/2]
  float prepoint, postpoint;
  int sliding_lowpass;   // sliding_lowpass[W][PACKETBLOBS/2]
};

// the coupling parameters of every candidate packet (blob) of a size class; VBR uses [PACKETBLOBS/2]
struct CoupleSet {
  CoupleP c[VAMD_PACKETBLOBS];
};

// The input domain's integer edge (include/vorbis_amd.h "Input domain" (2); derive_quant_limit, vamd_bind.h), channel by
// channel: a quantised value of channel c at a bin in [lo[c], hi[c]) -- the bins its submap's residue codes -- must stay
// within q[c] (lib/res0.c:361-364); one at a bin from sq on -- where noise normalisation is at work, sq = n/2 where it
// is not -- within VAMD_QUANT_LIMIT_SQUARE (lib/psy.c:985 squares it in an int); and every one within VAMD_QUANT_LIMIT_INT
// (:958-962 convert a float to int: the largest float below 2^31).
#define VAMD_QUANT_LIMIT_SQUARE 46340
#define VAMD_QUANT_LIMIT_INT 0x7fffff80
struct QLimitP {
  int q[VAMD_MAX_CH];
  short lo[VAMD_MAX_CH], hi[VAMD_MAX_CH];
  int sq;
};

// per-block descriptor source (arrays may be null -> uniform value)
struct DescP {
  const int *lW, *nW, *blocktype;
  const float *ampmax_in;
  int u_lW, u_nW, u_blocktype;
  float u_ampmax_in;
  unsigned long long *dbg;  // phase stopwatch slots (null = off), 16 per stage kernel
  unsigned long long *clk;  // vamd_clock_probe's accumulator (null = off): {shader ticks, 100 MHz ticks, samples}
  // input-domain report (include/vorbis_amd.h, "Input domain"): status[channel-block] is a bit set --
  // VAMD_STATUS_NONFINITE where the block's spectral peak, before the 0 dB clamp of lib/mapping0.c:345, is above
  // VAMD_NONFINITE_DB (k_transform: a NaN / Inf sample, or finite ones so large that the reference's own fp32 spectrum
  // overflows), VAMD_STATUS_RANGE where a quantised value of the channel is beyond the setup's proven integer bound
  // (k_couple: Bound::qlimit).  bad[0] counts flagged channel-blocks since vamd_input_status() last looked, bad[1]
  // detector steps, bad[2] the non-finite ones among bad[0]
  unsigned char *status;
  unsigned int *bad;
  // blocks read in place (vamd_batch_io::pcm_src): block b, channel c at pcm + src[b] + c * cstride; null = packed
  const long long *src;
  long cstride;
};

// The input domain (include/vorbis_amd.h) has two edges.
//  * NON-FINITE ARITHMETIC.  Any NaN or Inf sample puts every FFT bin's todB() above +330 dB (todB reads the float's
//    BITS, lib/scales.h:43-51, so the dB value of a NaN is a large finite number -- which is also why no NaN ever
//    reaches the maximum in transform_logfft); finite samples get there only from ~3e16 x full scale, where the
//    reference's own fp32 power spectrum re*re + im*im overflows to Inf.  One compare per channel-block in k_transform.
//  * THE REFERENCE'S INTEGERS.  Its quantised values are float -> int conversions (lib/psy.c:958-962), squared in an int
//    where noise normalisation is at work (:985), and searched against codebooks with sums of squared differences in an
//    int (lib/res0.c:361-364): defined by C while the values stay within the three bounds of QLimitP above (the last one
//    13 000 - 32 000 for the libvorbisenc setups, i.e. spectra ~ +85 ... +90 dB over full scale).  k_couple holds every
//    value it writes against the first two, k_residue every value it loads from a coded position against the third.
#define VAMD_NONFINITE_DB 330.f
#define VAMD_STATUS_RANGE 1      // bits of status[]
#define VAMD_STATUS_NONFINITE 2
// the same test in the block-switching detector, on its unscaled 128-point spectra (todB(re^2+im^2)*.5): finite
// samples inside the limit above stay below ~ +150 dB there, a NaN or Inf lands above +380
#define VAMD_ENV_LIMIT_DB 300.f

}  // namespace vamd
