// k_blockout.h -- the block-size decisions of vorbis_analysis_blockout() (reference lib/block.c:534-693) and
// the integer half of _ve_envelope_search() / _ve_envelope_mark() (lib/envelope.c:241-353), device-resident:
// one WAVE per stream walks its whole mark sequence (sixty-four marks per look) and emits the stream's block list, so
// that a set of streams goes from PCM to analysed blocks without a host round trip per block.
//
// The reference keeps a sliding PCM buffer and shifts every position by the block advance after each block
// (lib/block.c:649-685, _ve_envelope_shift); here every position is an ABSOLUTE sample index into the stream's
// buffer as the encoder would hold it had nothing been shifted out (what vorbis_analysis_buffer/_wrote
// accumulate, including the start-of-stream pre-extrapolation of lib/block.c:398-458, which is host code in
// the reference and stays the caller's).  Shifting subtracts the same amount from both sides of every
// comparison, so the decisions are the same.  The one asymmetry -- ve->curmark stops being shifted once it
// is negative, lib/envelope.c:372 -- only ever makes an already out-of-range mark stay out of range.
//
// What a plan covers: with BlockoutP::eof == 0 the blocks the reference would hand out while the data given suffices,
// i.e. with v->eofflag == 0 throughout.  With eof > 0 the stream ENDS at that sample (v->eofflag of
// vorbis_analysis_wrote(v, 0), lib/block.c:474-488, as an absolute index) and the buffer holds the reference's
// three long blocks of LPC padding behind it (k_lpc.h): a cursor walk that runs out of steps then forces a short
// next block instead of waiting (lib/block.c:558-563), and the block whose centre lies at or beyond eof is the
// stream's last (:664-670).
#pragma once
#include "vamd_wave.h"

namespace vamd {

#define VAMD_VE_WIN 4   // lib/envelope.h:23
#define VAMD_VE_POST 2  // lib/envelope.h:24

struct BlockoutP {
  int bs[2];        // blocksizes
  int searchstep;   // 64
  long nsamples;    // samples per channel in each stream's buffer
  long nsteps;      // detector steps available per stream (flags[s][0 .. nsteps))
  int maxblocks;    // capacity of a stream's row in `blocks`
  long eof;         // 0: the stream goes on (more data may come); > 0: v->eofflag as an absolute sample index
  int lstep;        // log2(searchstep) where that is a power of two (libvorbis: 64), else -1 (blockout_set_step)
};
// positions / searchstep for positions >= 0 (every one the walk divides): a shift where the step allows it -- a 64-bit
// division by a run-time value is some eighty vector instructions, and the walk made three or four per block
VAMD_DEV long blockout_div_step(const BlockoutP &B, long x) { return B.lstep >= 0 ? x >> B.lstep : x / B.searchstep; }
VAMD_HOSTDEV void blockout_set_step(BlockoutP &B, int searchstep) {
  B.searchstep = searchstep;
  B.lstep = -1;
  for (int l = 0; l < 30; l++)
    if ((1 << l) == searchstep) B.lstep = l;
}

// one planned block: W | lW << 1 | nW << 2 | blocktype << 3 in `kind`, and where its window starts
struct PlannedBlock {
  int kind;
  int begin;  // first sample of the block's window in the stream's buffer (centerW - blocksize/2)
};

// ve->mark[p] once every step that can touch it has run (lib/envelope.c:241-258): step j clears mark[j+2],
// then a pre-echo flag (1) marks j and j+1, a post-echo flag (2) marks j and j-1.  The clear of p happens at
// step p-2, before any of its setters (steps p-1, p, p+1), so the final value is the OR of those.
VAMD_DEV int mark_at(const unsigned char *__restrict__ flags, long nsteps, long p) {
  int m = 0;
  if (p < 0) return 0;
  if (p >= 1 && p - 1 < nsteps) m |= flags[p - 1] & 1;
  if (p < nsteps) m |= flags[p] & 3;
  if (p + 1 < nsteps) m |= flags[p + 1] & 2;
  return m != 0;
}

VAMD_DEV long blockout_steps(const BlockoutP &B) {
  long last = blockout_div_step(B, B.nsamples) - VAMD_VE_WIN;
  if (last > B.nsteps) last = B.nsteps;
  return last < 0 ? 0 : last;
}

// The walk for one stream.  Returns the number of blocks planned (<= maxblocks) and counts per size class.
//   marks  ve->mark[] of the steps [0, last) as bytes (mark_at applied by the caller's lanes; entries at and beyond
//          `last` are 0: steps not taken yet)
//   pending_center  (optional) centerW of the block the walk stopped in front of: where the reference's buffer would
//          begin (centerW - blocksizes[1]/2) when vorbis_analysis_wrote(v, 0) pads the stream
VAMD_DEV int plan_stream(const BlockoutP &B, const unsigned char *marks, PlannedBlock *__restrict__ out,
                         int *count_short, int *count_long, long *pending_center = nullptr) {
  const long step = B.searchstep;
  // vorbis_analysis_init / _ve_envelope_init: lib/block.c:211-213, lib/envelope.c:41
  int W = 0, lW = 0;
  long centerW = B.bs[1] / 2, cursor = B.bs[1] / 2, curmark = 0;
  // what _ve_envelope_search has marked: steps [0, last), last = pcm_current/searchstep - VE_WIN (:223-224)
  const long last = blockout_steps(B);
  const long current = last * step;
  int n = 0, n0 = 0, n1 = 0;
  while (n < B.maxblocks) {
    // ---- _ve_envelope_search's cursor walk, lib/envelope.c:262-325
    const long testW = centerW + B.bs[W] / 4 + B.bs[1] / 2 + B.bs[0] / 4;
    int bp = -1;
#if VAMD_GPU
    // The same walk sixty-four steps at a time (every lane of the wave runs plan_stream; all its values are
    // wave-uniform): lane l looks at j = base + l * step.  The walk stops at the first j that is past the horizon
    // (j >= testW: long block, cursor stays on the last j visited before it) or marked beyond the block's centre (short
    // block, cursor and curmark on it); a j at or past current - step is not visited at all, and a walk that runs out
    // of them leaves the cursor on the last one it did visit.
    for (long base = cursor;; base += 64 * step) {
      const long j = base + (long)LANE * step;
      const bool visited = j < current - step;  // (the visited lanes are a prefix of the wave)
      const bool past = visited && j >= testW;
      const bool marked = visited && !past && marks[blockout_div_step(B, j)] && j > centerW;
      const unsigned long long stop = __ballot(past || marked);
      if (stop) {
        const int l = __builtin_ctzll(stop);
        if ((__ballot(past) >> l) & 1) {
          bp = 1;
          if (l > 0) cursor = base + (long)(l - 1) * step;  // (else: where the previous trip, or the caller, left it)
        } else {
          bp = 0;
          cursor = curmark = base + (long)l * step;
        }
        break;
      }
      const int nvis = __builtin_popcountll(__ballot(visited));
      if (nvis > 0) cursor = base + (long)(nvis - 1) * step;
      if (nvis < 64) break;  // ran out of steps: bp stays -1
    }
#else
    for (long j = cursor; j < current - step; j += step) {
      if (j >= testW) {
        bp = 1;
        break;
      }
      cursor = j;
      if (marks[blockout_div_step(B, j)] && j > centerW) {
        curmark = j;
        bp = j >= testW ? 1 : 0;
        break;
      }
    }
#endif
    if (bp < 0 && !B.eof) break;  // "not enough data currently to search for a full long block", lib/block.c:558-560
    const int nW = (bp < 0 || B.bs[0] == B.bs[1]) ? 0 : bp;  // (at the end of a stream: nW = 0, :561)
    const long centerNext = centerW + B.bs[W] / 4 + B.bs[nW] / 4;
    if (B.nsamples < centerNext + B.bs[nW] / 2) break;  // lib/block.c:574-583
    // ---- the block, lib/block.c:589-611
    int blocktype;
    if (W) {
      blocktype = (!lW || !nW) ? 0 /* BLOCKTYPE_TRANSITION */ : 1 /* BLOCKTYPE_LONG */;
    } else {
      // _ve_envelope_mark, lib/envelope.c:329-353 (W == 0: both neighbours count as short)
      const long beginW = centerW - B.bs[0] / 4 - B.bs[0] / 4, endW = centerW + B.bs[0] / 4 + B.bs[0] / 4;
      int hit = curmark >= beginW && curmark < endW;
#if VAMD_GPU
      const long i_begin = blockout_div_step(B, beginW), i_end = blockout_div_step(B, endW);  // (beginW >= 0: centerW >= blocksizes[1] / 2)
      for (long i0 = i_begin; !hit && i0 < i_end; i0 += 64) {  // (a short block's span is a handful of steps: one trip)
        const long i = i0 + LANE;
        hit = __ballot(i < i_end && i >= 0 && i < last && marks[i]) != 0;
      }
#else
      for (long i = blockout_div_step(B, beginW); !hit && i < blockout_div_step(B, endW); i++) hit = i >= 0 && i < last && marks[i];
#endif
      blocktype = hit ? 0 /* BLOCKTYPE_IMPULSE */ : 1 /* BLOCKTYPE_PADDING */;
    }
    if (LANE == 0 && out) {
      out[n].kind = W | (lW << 1) | (nW << 2) | (blocktype << 3);
      out[n].begin = (int)(centerW - B.bs[W] / 2);
    }
    n++;
    if (W) n1++; else n0++;
    if (B.eof && centerW >= B.eof) {  // the stream's last block (vb->eofflag = 1), lib/block.c:664-670
      centerW = -1;
      break;
    }
    // ---- advance, lib/block.c:649-685 (positions stay absolute: nothing to shift)
    lW = W;
    W = nW;
    centerW = centerNext;
  }
  *count_short = n0;
  *count_long = n1;
  if (pending_center) *pending_center = centerW;  // (-1: the stream is over)
  return n;
}

}  // namespace vamd
