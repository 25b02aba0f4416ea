// k_tone.h -- _vp_tonemask (reference lib/psy.c:754-777): ATH floor, seed_loop /
// seed_curve scatter (lib/psy.c:390-452), seed_chase (:454-508) and max_seeds
// (:512-545); SURVEY.md 8a row a10.  One wavefront per channel-block.
//
// Parallel form:
//  * runs of bins sharing an octave[] value are a property of the static table;
//    vamd_create() lists them, lanes take one run each, find the run's peak and
//    scatter the chosen tone curve into seed[] with an LDS float max (max is
//    order-free, so the result equals the reference's sequential update).
//  * seed_chase is NOT a sliding-window max (SURVEY.md 0.8 iv).  Its first half
//    (the stack discipline) is restated literally and walked by one lane, with
//    the two top-of-stack entries held in registers so the common path never
//    waits on LDS; its second half (painting each surviving entry over its span)
//    is a prefix-max over the entries' end positions and is done by all lanes.
//  * max_seeds' pointer walk over (octave line, bin) is static too: per bin the
//    span of seed lines it folds is precomputed, every lane folds its own bins.
//
// The stage is three kernels so that the ordered stack walk -- one useful lane if a
// wave owned a single block -- can instead run with one *lane* per channel-block
// (64 walks per wave instruction):
//   tone_seed_block   wave per block   : scatter -> seed[] to HBM
//   tone_chase_thread thread per block : stack walk -> list of surviving lines
//   tone_fold_block   wave per block   : paint + max_seeds fold -> tone curve
// tonemask_block() composes the same three pieces for one block (test build).
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"

namespace vamd {

#define VAMD_NEGINF (-9999.f)
// a block's row of seed lines / survivors in HBM: whole 128-byte lines (tone_chase_thread)
#define VAMD_LINES_PAD(nl) (((nl) + 31) & ~31)

// seed_curve, lib/psy.c:390-415.  The reference walks i = posts[0] .. post1-1 with
// seedptr advancing by linesper and stops once seedptr >= n; point i therefore
// lands on line oc + (i-16)*linesper - linesper/2 and is applied iff that line is
// in (0, n).  Written branch-free: the curve rows (re-strided to 64 floats, 256-byte
// aligned, vamd_derive.h) hold -inf outside [posts[0], posts[1]) so "amp + c" cannot
// win there, and seed[] carries seed_pad_lo() floats in front and seed_pad_hi() behind so
// that out-of-range lines land in padding nobody reads (line 0, which the reference
// also skips, is reset by the caller).  Per point that leaves add, add, ds_max_f32.
VAMD_HOSTDEV int seed_pad_lo(int linesper) { return (VAMD_EHMER_OFFSET * linesper + (linesper >> 1) + 3) & ~3; }
VAMD_HOSTDEV int seed_pad_hi(int linesper) { return ((VAMD_EHMER_MAX - VAMD_EHMER_OFFSET) * linesper + 3) & ~3; }
// LP > 0 fixes linesper (eighth_octave_lines: 8 in every libvorbisenc setup) at compile time, which turns
// the 56 scatter addresses into one base register plus immediate offsets.
template <int LP = 0>
VAMD_DEV void seed_curve_scatter(float *seed, const float *__restrict__ band_rows /*[8] rows of one band*/,
                                 int stride, float amp, int oc, int linesper_rt, float dBoffset, bool active) {
  const int linesper = LP ? LP : linesper_rt;
  int choice = (int)(((double)(amp + dBoffset) - 30.) * (double).1f);
  choice = choice < 0 ? 0 : choice;
  choice = choice > VAMD_P_LEVELS - 1 ? VAMD_P_LEVELS - 1 : choice;
  const F4 *__restrict__ row = (const F4 *)(band_rows + choice * stride);
  float *p = seed + (oc - VAMD_EHMER_OFFSET * linesper - (linesper >> 1));
  // The rows are finite only between their fence posts -- 8 to 50 of the 56 points, 12 to 25 for most of the
  // spectrum -- and what a point costs is an LDS float atomic plus its share of the row's trip from L1 whether
  // the value is -inf or not.  So the points go in seven groups of eight, and a group that holds no finite
  // point for any lane of the wave is skipped (one wave-uniform branch per group: per-POINT skipping was
  // measured slower than no skipping at all).  Inside a group everything stays straight-line with immediate
  // offsets.  The posts live in the row's padding: curves64 rows are 64 floats, [56] = first, [57] = last+1.
  int p0 = VAMD_EHMER_MAX, p1 = 0;
  if (active) {
    const F2 posts = *(const F2 *)((const float *)row + VAMD_EHMER_MAX);
    p0 = (int)posts.x;
    p1 = (int)posts.y;
  }
#pragma unroll
  for (int g = 0; g < VAMD_EHMER_MAX / 8; g++) {
    if (!wave_any(active && p0 < 8 * g + 8 && p1 > 8 * g)) continue;
    if (active) {
      float c[8];
      f4_get(row[2 * g], c);
      f4_get(row[2 * g + 1], c + 4);
#pragma unroll
      for (int i = 0; i < 8; i++) lds_atomic_max(p + (8 * g + i) * linesper, amp + c[i]);
    }
  }
}

// seed_chase part 2, lib/psy.c:489-503: entry k paints [start_k, end_k) where
// end_k = next.pos if the next entry is louder else pos_k + linesper + 1 (clipped
// to n) and start_k = max(end_0 .. end_{k-1}) because the reference's write
// pointer only moves forward.  Spans are disjoint, so all lanes paint at once.
//   seeds     LDS, painted in place
//   src       where an entry's amplitude (the unpainted value of its line) is read.  On the GPU that is `seeds` itself:
//             chunk c+1's amplitudes are fetched before chunk c paints, and what chunks <= c-1 painted ends at most
//             linesper lines past their last entry -- short of chunk c+1's first entry, which is 65 or more entries and
//             therefore lines further on (linesper <= 16).  The one-lane test build, whose chunks are single entries,
//             hands over a copy of the unpainted lines instead.
//   posstack  the survivor list (HBM);  head  its first two chunks as fetched ahead by surv_head_load (optional)
#include "k_tone_fold.inc"  // SurvHead, surv_head_load, seed_chase_paint, tone_fold_prepare, tone_ath_att, tone_fold_quad

// Thread-per-block form of seed_chase part 1.  Only the top ~9 stack entries can
// ever be popped or inspected (an entry more than `linesper` lines behind the
// current line fails both position tests for good), so the stack lives in a
// 16-slot ring per lane; an entry pushed out of the ring is final and its line
// index is appended to the survivor list.  Ring layout [slot][lane] keeps the 64
// lanes on distinct LDS banks.
//   seeds   this block's seed[] (HBM, read 32 lines -- a cache line -- at a time)
//   surv    out: surviving line indices, ascending (uint16), returns their count
#define VAMD_RING 16
VAMD_DEV int tone_chase_thread(const float *__restrict__ seeds, int linesper, int n, float *ring_amp, int *ring_pos,
                               int rstride, int rlane, unsigned short *__restrict__ surv) {
  int stack = 0, hmax = 0;  // hmax = highest stack index ever written
  // survivors leave in index order, 0, 1, 2, ...: eight at a time as one 16-byte store (a lane per block means every
  // lane stores into a cache line of its own: with 2-byte stores the stage took 0.75 ms, with these 0.5)
  unsigned long long wlo = 0, whi = 0;  // the last eight, oldest in the low bits of wlo
  int nw = 0;
  auto emit = [&](int pos) {
    wlo = (wlo >> 16) | (whi << 48);
    whi = (whi >> 16) | ((unsigned long long)(unsigned)pos << 48);
    nw++;
    if ((nw & 7) == 0) {
      unsigned long long *dst = (unsigned long long *)(surv + nw - 8);
      dst[0] = wlo;
      dst[1] = whi;
    }
  };
  float a1 = 0.f, a2 = 0.f;
  int p1 = 0, p2 = 0;
  // A lane reads its row a whole 128-byte line at a time (rows are padded to VAMD_LINES_PAD and start on a line): with 64
  // bytes per trip every line was asked for twice, a few hundred cycles apart, by one lane of 65 536 whose lines fill the
  // L2s exactly -- half of them had left by the second request (11.6 KB fetched per stereo block for 6.2 KB of lines).
  const F4 *q = (const F4 *)seeds;
  const int nblk = (n + 31) >> 5;
  F4 w[8];
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = q[k];  // the next 32 lines load while these are walked
  for (int b = 0; b < nblk; b++) {
    float blk[32];
#pragma unroll
    for (int k = 0; k < 8; k++) blk[4 * k] = w[k].x, blk[4 * k + 1] = w[k].y, blk[4 * k + 2] = w[k].z, blk[4 * k + 3] = w[k].w;
    if (b + 1 < nblk) {
#pragma unroll
      for (int k = 0; k < 8; k++) w[k] = q[8 * b + 8 + k];
    }
#pragma unroll
    for (int j = 0; j < 32; j++) {
      const int i = (b << 5) + j;
      if (i < n) {
        const float s = blk[j];
        if (stack >= 2) {
          while (!(s < a1) && i < p1 + linesper && a1 <= a2 && i < p2 + linesper) {
            stack--;
            a1 = a2;
            p1 = p2;
            if (stack < 2) break;
            const int slot = (stack - 2) & (VAMD_RING - 1);
            a2 = ring_amp[slot * rstride + rlane];
            p2 = ring_pos[slot * rstride + rlane];
          }
        }
        const int slot = stack & (VAMD_RING - 1);
        if (stack == hmax) {
          // first write of this index: the entry one ring-length below is final, emit it
          // before its slot is reused (re-pushes of an index already seen emit nothing)
          if (stack >= VAMD_RING) emit(ring_pos[slot * rstride + rlane]);
          hmax++;
        }
        ring_amp[slot * rstride + rlane] = s;
        ring_pos[slot * rstride + rlane] = i;
        stack++;
        a2 = a1;
        p2 = p1;
        a1 = s;
        p1 = i;
      }
    }
  }
  for (int k = hmax > VAMD_RING ? hmax - VAMD_RING : 0; k < stack; k++) emit(ring_pos[(k & (VAMD_RING - 1)) * rstride + rlane]);
  for (int k = nw & ~7; k < nw; k++) {  // the last one to seven
    const int sh = 16 * (8 - (nw & 7) + (k & 7));
    surv[k] = (unsigned short)(sh < 64 ? wlo >> sh : whi >> (sh - 64));
  }
  return stack;
}

// ---- seed_chase part 1 for ONE block spread over a wave (small batches, the per-block entry points) -------------
// One lane per block is the right shape for a batch of thousands; for a handful of blocks it leaves a single lane
// walking ~800 lines while the caller waits (most of a block's latency on the GPU).  The walk can be cut into
// chunks because it forgets: an entry can be popped, or looked at, only while the current line is within
// `linesper` of it, so what steps >= t do depends on the past only through WHICH of the lines t-linesper+1 .. t-1
// are still on the stack (their amplitudes are the seed values; anything older fails the position tests, and the
// stack holds two entries or more from line 2 on).  So chunk c is walked from a cold start `warm` lines early, and the
// result is accepted iff every chunk's state on entry -- that alive mask -- equals its predecessor's state on exit
// (chunk 0, and any chunk whose warm-up reaches back to line 0, start from the true state: induction does the rest).
// A chunk whose entry state differs from its predecessor's exit is walked again, this time started exactly in that
// exit state (a few rounds; runs of equal values -- stretches no curve reached -- carry a phase from their beginning
// and need them); a block still inconsistent after VAMD_CHASE_ROUNDS takes the serial walk, so the outcome is the
// reference's in every case (tests/test_kernel_bodies_cpu.py counts rounds and fallbacks).
// A chunk also walks linesper-1 lines past its end: that is where its own last lines can still be popped.
struct ChaseChunk {
  uint32_t popped;   // bit k: line s+k was popped
  uint32_t sig_in;   // alive mask of the window before line s, as the cold start left it
  uint32_t sig_out;  // alive mask of the window before line e (valid when e < n)
  int exact;         // the walk started at line 0: its state is the true one
};
//   warm >= 0: cold start `warm` lines before s.  warm < 0: start AT s from the state `enter` (an alive mask of the
//   window before s, e.g. the predecessor's sig_out): the stack is rebuilt as one entry far out of reach -- standing
//   for everything older, which can neither be popped nor pass a position test -- plus the window's alive lines.
VAMD_DEV ChaseChunk chase_chunk(const float *seeds, int linesper, int n, int s, int e, int warm, uint32_t enter,
                                float *ring_amp, int *ring_pos, int rstride, int rlane) {
  ChaseChunk out;
  out.popped = out.sig_in = out.sig_out = 0;
  const int ws = warm < 0 ? s : (s - warm > 0 ? s - warm : 0);
  out.exact = ws == 0;
  const int stop = e + linesper - 1 < n ? e + linesper - 1 : n;
  int stack = 0;
  float a1 = 0.f, a2 = 0.f;
  int p1 = 0, p2 = 0;
  if (warm < 0 && s > 0) {
    auto push = [&](float a, int pos) {
      const int slot = stack & (VAMD_RING - 1);
      ring_amp[slot * rstride + rlane] = a;
      ring_pos[slot * rstride + rlane] = pos;
      stack++;
      a2 = a1;
      p2 = p1;
      a1 = a;
      p1 = pos;
    };
    push(0.f, -(1 << 24));
    const int lo = s - linesper + 1;
    for (int k = 0; k < linesper - 1; k++)
      if (lo + k >= 0 && ((enter >> k) & 1u)) push(seeds[lo + k], lo + k);
  }
  // alive mask of lines t-linesper+1 .. t-1 (bit = line - (t-linesper+1)): the stack's top entries, newest first
  auto window = [&](int t) {
    uint32_t m = 0;
    const int lo = t - linesper + 1;
    for (int k = stack - 1; k >= 0 && k >= stack - linesper; k--) {
      const int pos = k == stack - 1 ? p1 : (k == stack - 2 ? p2 : ring_pos[(k & (VAMD_RING - 1)) * rstride + rlane]);
      if (pos < lo) break;
      m |= 1u << (pos - lo);
    }
    return m;
  };
  for (int i = ws; i < stop; i++) {
    if (i == s) out.sig_in = window(s);
    if (i == e) out.sig_out = window(e);
    const float sv = seeds[i];
    if (stack >= 2) {
      while (!(sv < a1) && i < p1 + linesper && a1 <= a2 && i < p2 + linesper) {
        if (p1 >= s && p1 < e) out.popped |= 1u << (p1 - s);
        stack--;
        a1 = a2;
        p1 = p2;
        if (stack < 2) break;
        const int slot = (stack - 2) & (VAMD_RING - 1);
        a2 = ring_amp[slot * rstride + rlane];
        p2 = ring_pos[slot * rstride + rlane];
      }
    }
    const int slot = stack & (VAMD_RING - 1);
    ring_amp[slot * rstride + rlane] = sv;
    ring_pos[slot * rstride + rlane] = i;
    stack++;
    a2 = a1;
    p2 = p1;
    a1 = sv;
    p1 = i;
  }
  return out;
}

// The same walk with its state in registers.  What a step can pop or look at are the lines within linesper of it, so the
// stack as far as the walk can tell is a bit mask of which of the last LP-1 lines are still on it and their values,
// newest first; older entries fail both position tests (lib/psy.c:470-474) whatever they hold, and chase_chunk's cold
// start knows none either.  A push and a pop are LP-2 register moves: no ring in LDS, no trip to it per pop (a lone
// wave waited some 20 us of a block's 25 there), and the alive masks chase_chunk gathers by walking its stack are the
// state itself.  Masks here are newest-first (bit k = line t-1-k), the reverse of chase_chunk's: they are only ever
// compared with, or handed to, this function's own.
//   cs    the length every chunk of the block was cut to (the lanes of a wave step together: warm + cs + LP-1 steps,
//         those before line 0 or past the chunk's last look doing nothing)
template <int LP>
VAMD_DEV ChaseChunk chase_chunk_regs(const float *seeds, int n, int s, int e, int cs, int warm, uint32_t enter) {
  constexpr int WN = LP - 1;
  constexpr uint32_t WMASK = (1u << WN) - 1u;
  ChaseChunk out;
  out.popped = out.sig_in = out.sig_out = 0;
  const int ws = warm < 0 ? s : s - warm;
  out.exact = ws <= 0;
  const int stop = e + LP - 1 < n ? e + LP - 1 : n;
  const int steps = (warm < 0 ? 0 : warm) + cs + LP - 1;
  uint32_t m = 0;
  float st[WN];
#pragma unroll
  for (int j = 0; j < WN; j++) st[j] = 0.f;
  if (warm < 0 && s > 0) {
    m = enter & WMASK;
    uint32_t mm = m;
#pragma unroll
    for (int j = 0; j < WN; j++)
      if (mm) {
        st[j] = seeds[s - 1 - __builtin_ctz(mm)];
        mm &= mm - 1;
      }
  }
  float nxt = ws >= 0 && ws < n ? seeds[ws] : 0.f;  // a step's line is asked for a step ahead
  for (int t = 0; t < steps; t++) {
    const int i = ws + t;
    const float sv = nxt;
    nxt = i + 1 >= 0 && i + 1 < n ? seeds[i + 1] : 0.f;
    if (i == s) out.sig_in = m;
    if (i == e && i < stop) out.sig_out = m;
    if (i >= 0 && i < stop) {
      while ((m & (m - 1)) != 0 && !(sv < st[0]) && st[0] <= st[1]) {
        const int p1 = i - 1 - __builtin_ctz(m);
        if (p1 >= s && p1 < e) out.popped |= 1u << (p1 - s);
        m &= m - 1;
#pragma unroll
        for (int j = 0; j + 1 < WN; j++) st[j] = st[j + 1];
      }
      m = ((m << 1) | 1u) & WMASK;
#pragma unroll
      for (int j = WN - 1; j > 0; j--) st[j] = st[j - 1];
      st[0] = sv;
    }
  }
  return out;
}

#define VAMD_CHASE_CHUNKS 64  // chunks per block = lanes of the wave
#ifndef VAMD_CHASE_WARM
#define VAMD_CHASE_WARM 2    // cold start this many windows (linesper) before a chunk: one needs a repair round every time,
                             // two in one block out of thirty, three or four never (measured on the host; 16 + 20 lines walked)
#endif
// What the chunks of one block add up to: accepted iff every chunk entered in its predecessor's exit state
// (k_tone_chase_wave runs chase_chunk one per lane and combines with wave operations; tests/emul/emul.cpp walks the same
// chunks one after the other and compares with the serial walk).
// Repair rounds before a block is handed to the serial walk.  A run of equal values (a stretch no curve reached)
// carries a phase from its beginning, so the true state crosses it one chunk per round: worth it up to about two dozen
// chunks (a round is ~20 lines per lane, the serial walk ~800 for one); a block with a longer run goes serial at once
// (chase_flat_chunk / VAMD_CHASE_FLAT_MAX).
#define VAMD_CHASE_ROUNDS 26
#define VAMD_CHASE_FLAT_MAX 24
// does chunk [s, e) merely continue a run of equal values (every line equals line s-1)?
VAMD_DEV int chase_flat_chunk(const float *seeds, int s, int e) {
  if (s == 0 || s >= e) return 0;
  const float v = seeds[s - 1];
  int flat = 1;
  for (int i = s; i < e; i++) flat &= seeds[i] == v;
  return flat;
}

// scatter half: seed[] for one channel-block (LDS), lib/psy.c:417-452,762-771
//   peaks  HBM [nruns]: the maximum of logfft over each run of bins that share an octave line (run_peak, formed by
//          the transform stage while the block's logfft was in LDS): a float per run is all this stage needs of it
//
// Round 5.  A wave spent its life waiting: per trip of 64 runs it asked for the runs' records and peaks, waited, asked
// for the chosen rows' fence posts, waited, and then for each of up to seven groups of points asked for the group and
// waited before its eight maxima went out -- up to nine dependent trips to L1/L2 per 64 runs, forty-five per block, in a
// kernel of 600 vector instructions (12 800 cycles per wave, 4 % of them issuing).  Now: the fence posts of all 136
// rows sit in three registers per lane (fetched once, beside the first records; a row's pair comes out of the lanes
// with ds_bpermute), the next trip's records and peaks are asked for before the current trip's rows, and the groups a
// trip needs are asked for four and three at a time, all of a batch in flight together: two trips to memory per 64
// runs instead of nine.
#if VAMD_GPU
VAMD_DEV int seed_level(float amp, float dBoffset) {
  int choice = (int)(((double)(amp + dBoffset) - 30.) * (double).1f);
  choice = choice < 0 ? 0 : choice;
  return choice > VAMD_P_LEVELS - 1 ? VAMD_P_LEVELS - 1 : choice;
}
template <int LP = 0>
VAMD_DEV void tone_seed_block(const PsyP &P, const float *__restrict__ peaks, float global_ampmax,
                              float local_ampmax, float *seed, PhaseClock &pc) {
  const int nlines = P.total_octave_lines, nruns = P.nruns;
  const int linesper = LP ? LP : P.eighth_octave_lines;
  const int stride = P.curve_stride;
  float att = local_ampmax + P.ath_adjatt;
  if (att < P.ath_maxatt) att = P.ath_maxatt;
  // first trip's records and peaks, and the fence posts of every row: one trip to memory for all of it
  I4 rec_n = {0, 0, 0, 0};
  float mx_n = 0.f;
  if (LANE < nruns) {
    rec_n = ((const I4 *)P.runs)[LANE];
    mx_n = peaks[LANE];
  }
  static_assert(VAMD_P_BANDS * VAMD_P_LEVELS <= 3 * 64, "the rows' fence posts are kept three per lane");
  int posts_of[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int e = LANE + 64 * k;
    posts_of[k] = VAMD_EHMER_MAX;  // (first = 56, last + 1 = 0: nothing)
    if (e < VAMD_P_BANDS * VAMD_P_LEVELS) {
      const F2 pp = *(const F2 *)(P.curves64 + (size_t)e * stride + VAMD_EHMER_MAX);
      posts_of[k] = (int)pp.x | ((int)pp.y << 8);
    }
  }
  WAVE_FOR(i, nlines) seed[i] = VAMD_NEGINF;  // (the padding either side is write-only)
  WAVE_SYNC();
  pc.mark(0);
  const float dBoffset = P.max_curve_dB - global_ampmax;
  for (int r0 = 0; r0 < nruns; r0 += NLANES) {  // (wave-uniform: every lane makes every trip)
    const I4 rec = rec_n;
    const float mx = mx_n;
    const bool valid = r0 + LANE < nruns;
    if (r0 + NLANES + LANE < nruns) {
      rec_n = ((const I4 *)P.runs)[r0 + NLANES + LANE];
      mx_n = peaks[r0 + NLANES + LANE];
    }
    const bool active = valid && mx + 6.f > f_from_bits((uint32_t)rec.w) + att;
    const int e = rec.z * VAMD_P_LEVELS + seed_level(mx, dBoffset);
    const int g0 = wave_gather(posts_of[0], e & 63), g1 = wave_gather(posts_of[1], e & 63), g2 = wave_gather(posts_of[2], e & 63);
    const int pk = e < 64 ? g0 : (e < 128 ? g1 : g2);
    const int p0 = active ? (pk & 0xff) : VAMD_EHMER_MAX, p1 = active ? (pk >> 8) : 0;
    const F4 *__restrict__ row = (const F4 *)(P.curves64 + (size_t)e * stride);
    float *p = seed + (rec.y - VAMD_EHMER_OFFSET * linesper - (linesper >> 1));
    // The rows are finite only between their fence posts -- 8 to 50 of the 56 points, 12 to 25 for most of the
    // spectrum -- so the points go in seven groups of eight, and a group that holds no finite point for any lane of
    // the wave is skipped (one wave-uniform branch per group: per-POINT skipping was measured slower than no skipping
    // at all).  Inside a group everything stays straight-line with immediate offsets.
    bool need[VAMD_EHMER_MAX / 8];
#pragma unroll
    for (int g = 0; g < VAMD_EHMER_MAX / 8; g++) need[g] = wave_any(p0 < 8 * g + 8 && p1 > 8 * g);
#pragma unroll
    for (int b0 = 0; b0 < VAMD_EHMER_MAX / 8; b0 += 4) {
      F4 c[4][2];
#pragma unroll
      for (int g = b0; g < b0 + 4 && g < VAMD_EHMER_MAX / 8; g++)
        if (need[g] && active) {
          c[g - b0][0] = row[2 * g];
          c[g - b0][1] = row[2 * g + 1];
        }
#pragma unroll
      for (int g = b0; g < b0 + 4 && g < VAMD_EHMER_MAX / 8; g++)
        if (need[g] && active) {
          float v[8];
          f4_get(c[g - b0][0], v);
          f4_get(c[g - b0][1], v + 4);
#pragma unroll
          for (int i = 0; i < 8; i++) lds_atomic_max(p + (8 * g + i) * linesper, mx + v[i]);
        }
    }
  }
  WAVE_SYNC();
  if (LANE == 0) seed[0] = VAMD_NEGINF;  // seedptr > 0, lib/psy.c:406
  WAVE_SYNC();
  pc.mark(1);
}
#else
template <int LP = 0>
VAMD_DEV void tone_seed_block(const PsyP &P, const float *__restrict__ peaks, float global_ampmax,
                              float local_ampmax, float *seed, PhaseClock &pc) {
  const int nlines = P.total_octave_lines;
  float att = local_ampmax + P.ath_adjatt;
  if (att < P.ath_maxatt) att = P.ath_maxatt;
  WAVE_FOR(i, nlines) seed[i] = VAMD_NEGINF;  // (the padding either side is write-only)
  WAVE_SYNC();
  pc.mark(0);
  const float dBoffset = P.max_curve_dB - global_ampmax;
  WAVE_FOR(r, P.nruns) {
    const I4 rec = ((const I4 *)P.runs)[r];
    const float mx = peaks[r];
    seed_curve_scatter<LP>(seed, P.curves64 + rec.z * (VAMD_P_LEVELS * P.curve_stride), P.curve_stride, mx, rec.y,
                           P.eighth_octave_lines, dBoffset, mx + 6.f > f_from_bits((uint32_t)rec.w) + att);
  }
  WAVE_SYNC();
  if (LANE == 0) seed[0] = VAMD_NEGINF;  // seedptr > 0, lib/psy.c:406
  WAVE_SYNC();
  pc.mark(1);
}
#endif

VAMD_DEV void tone_fold_block(const PsyP &P, float local_ampmax, float *seed, const float *seed_src,
                              const unsigned short *__restrict__ surv, int nsurv, float *gmin /* LDS [ngroups] */,
                              float *__restrict__ out, PhaseClock &pc) {
  const float att = tone_ath_att(P, local_ampmax);
  tone_fold_prepare(P, seed, seed_src, surv, nsurv, gmin, pc);
  WAVE_FOR(q, P.n >> 2) {
    float o[4];
    tone_fold_quad(P, att, seed, gmin, q, o);
    ((F4 *)out)[q] = f4_make(o);
  }
  WAVE_SYNC();
  pc.mark(4);
}

// _vp_tonemask(p, logfft, logmask, global_specmax, local_specmax), one block end to
// end (the test build; the GPU runs the three pieces as separate launches)
//   seed LDS [seed_pad_lo | nlines padded to 16 | seed_pad_hi] (pointer at line 0), fft LDS [n],
//   seed_copy [nlines] scratch for the unpainted values, posstack/ampstack LDS [nlines],
//   ring_amp/ring_pos [VAMD_RING], surv [nlines]
VAMD_DEV void tonemask_block(const PsyP &P, const float *__restrict__ logfft, float *__restrict__ out,
                             float global_ampmax, float local_ampmax, float *seed, float *seed_copy,
                             float *fft, float *ring_amp, int *ring_pos, unsigned short *surv, PhaseClock &pc) {
  for (int r = 0; r < P.nruns; r++) fft[r] = run_peak(logfft, P.run_start[r], P.run_start[r + 1]);  // (what k_transform hands over)
  tone_seed_block(P, fft, global_ampmax, local_ampmax, seed, pc);
  const int nsurv = tone_chase_thread(seed, P.eighth_octave_lines, P.total_octave_lines, ring_amp, ring_pos, 1, 0, surv);
  for (int i = 0; i < P.total_octave_lines; i++) seed_copy[i] = seed[i];  // (single-lane test build only)
  tone_fold_block(P, local_ampmax, seed, seed_copy, surv, nsurv, fft /* reused as gmin */, out, pc);
}

}  // namespace vamd
