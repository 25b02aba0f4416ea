// vamd_hip.hip -- libvorbis_amd.so: the gfx950 kernels (thin __global__ shells
// around the wave-level bodies in k_*.h) and the C ABI of include/vorbis_amd.h.
//
// Launch geometry: one 64-lane wavefront per workgroup, one workgroup per
// channel-block (per block for the coupling stage).  A 65 536-block stereo batch
// is 131 072 workgroups per stage -- ~500 per CU -- so the chip is filled many
// times over and the per-wave latency of the ordered sections (running sums,
// seed_chase, the greedy floor split) is hidden by the other resident waves.
// Intermediates between stages live in an HBM workspace owned by the context;
// a tensor the caller asked for is written straight to the caller's buffer.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (mandatory: the
// reference's results depend on separately rounded mul/add).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <new>
#include <string>
#include <vector>

#include "vorbis_amd.h"
#include "vamd_bind.h"
#include "vamd_knobs.h"
#include "k_transform.h"
#include "k_noise.h"
#include "k_tone.h"
#include "k_floor.h"
#include "vamd_wave_pair.h"
#include "k_couple.h"
#include "k_envelope.h"
#include "k_residue.h"
#include "k_pack.h"
#include "k_blockout.h"

using namespace vamd;

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) unsigned char vamd_smem[];

__device__ __forceinline__ int d_lW(const DescP &d, long b) { return d.lW ? d.lW[b] : d.u_lW; }
__device__ __forceinline__ int d_nW(const DescP &d, long b) { return d.nW ? d.nW[b] : d.u_nW; }
__device__ __forceinline__ int d_bt(const DescP &d, long b) { return d.blocktype ? d.blocktype[b] : d.u_blocktype; }
__device__ __forceinline__ float d_amp(const DescP &d, long b) { return d.ampmax_in ? d.ampmax_in[b] : d.u_ampmax_in; }
// the input domain's integer edge (k_couple.h: QuantSpan): channel-block i holds a quantised value beyond the setup's
// bound.  One lane per channel-block calls this; a channel-block k_transform has already flagged is not counted again.
// (The fifteen candidate packets of a bitrate-managed block are fifteen units that may flag the same channel-block:
// the byte is right either way, the count may then run up to fourteen high.)
__device__ __forceinline__ void flag_range(const DescP &d, long i) {
  const unsigned char old = d.status[i];
  if (!(old & VAMD_STATUS_RANGE)) {
    d.status[i] = old | VAMD_STATUS_RANGE;
    if (!old) atomicAdd(d.bad, 1u);
  }
}

// ---- transform kernels: persistent workgroups with the tables staged in LDS -------
// One workgroup per CU, VAMD_XF_WAVES independent waves each owning one channel-block
// at a time and looping over the batch.  The window, MDCT trig/bit-reverse and FFT
// twiddle tables (24.5 KB at n = 2048) are copied into LDS once per workgroup and every
// butterfly of every block then reads them at LDS latency instead of going to L2.
// Waves never synchronise with each other after the staging barrier (WAVE_SYNC is
// wave-local), so they drift apart and overlap each other's memory phases.
#define VAMD_XF_WAVES 8
// channel-blocks from which the floor stage takes the two channels of a stereo block in one wave (k_floor_pair), per size
// class.  Measured round 5 (profiles/r05_floor_pair.txt, tools/floor_pair_ab.sh): SHORT blocks gain -- their 128 bins and
// 13 / 19 posts leave half of a wave's lanes idle in every phase of k_floor: C5's floor 3.80 -> 3.53 ms, the step
// 11.38 -> 11.09 ms, at six waves per SIMD (77 registers) -- from a batch that fills the chip; LONG blocks lose at every
// occupancy (1.96 -> 2.42 ms at best: 12 % fewer vector instructions per channel-block, but 9.6 KB of LDS per wave
// hold the CU to sixteen waves, and the half-uniform reads of the ordered sections go through the LDS pipe where
// v_readlane did not: 64 % of the issue slots used against 100 %) -- never.
#ifndef VAMD_FLOOR_PAIR_MIN_LONG
#define VAMD_FLOOR_PAIR_MIN_LONG 0x7fffffffL
#endif
#ifndef VAMD_FLOOR_PAIR_MIN_SHORT
#define VAMD_FLOOR_PAIR_MIN_SHORT 16384L
#endif

struct XformLds {
  XformP P;       // table pointers rebound to the LDS copies
  float *A, *B;   // this wave's work buffers
};

template <int LOGN>
__device__ __forceinline__ XformLds stage_transform_tables(const XformP &G) {
  const int n = G.n;
  float *trig = (float *)vamd_smem;          // [n + n/4]
  float *wa = trig + n + n / 4;              // [n]   (the twiddles the passes touch: wa[0 .. n-1))
  float *winL = wa + n;                      // [bs1/2]
  float *winS = winL + G.bs1 / 2;            // [bs0/2]
  // [n/4]: the bit-reverse table, or (size-specialised kernels, which compute those indices) the butterfly
  // stages' repacked trig pairs
  int *bitrev = (int *)(winS + G.bs0 / 2);
  float *work = (float *)(bitrev + n / 4);
  for (int i = threadIdx.x; i < n + n / 4; i += blockDim.x) trig[i] = G.trig[i];
  for (int i = threadIdx.x; i < n; i += blockDim.x) wa[i] = G.wa[i];
  for (int i = threadIdx.x; i < G.bs1 / 2; i += blockDim.x) winL[i] = G.win_long[i];
  for (int i = threadIdx.x; i < G.bs0 / 2; i += blockDim.x) winS[i] = G.win_short[i];
  if (LOGN)
    mdct_tpack_fill((float *)bitrev, G.trig, n, threadIdx.x, blockDim.x);
  else
    for (int i = threadIdx.x; i < n / 4; i += blockDim.x) bitrev[i] = G.bitrev[i];
  __syncthreads();
  XformLds L;
  L.P = G;
  L.P.trig = trig;
  L.P.wa = wa;
  L.P.win_long = winL;
  L.P.win_short = winS;
  L.P.bitrev = LOGN ? nullptr : bitrev;
  L.P.tpack = LOGN ? (const float *)bitrev : nullptr;
  const int wave = threadIdx.x >> 6;
  const int per_wave = VAMD_XF_A_FLOATS(n) + VAMD_XF_B_FLOATS(n);
  L.A = work + wave * per_wave;
  L.B = L.A + VAMD_XF_A_FLOATS(n);
  return L;
}

static size_t transform_lds_bytes(const XformP &P, int waves) {
  const size_t tables = (size_t)(P.n + P.n / 4) + P.n + P.bs1 / 2 + P.bs0 / 2 + P.n / 4;
  return (tables + (size_t)waves * (VAMD_XF_A_FLOATS(P.n) + VAMD_XF_B_FLOATS(P.n))) * 4;
}

// mdct_forward only (BASELINE config 2): in[nframes][n] -> out[nframes][n/2].  The one HBM-bound
// kernel of the path.  The fold reads each input value exactly once, so it reads the frame straight
// from HBM (no LDS copy of it): a wave then needs only the butterfly buffer (8.4 KB at n = 2048) and
// sixteen waves fit on a CU beside the trig / bit-reverse tables.
#define VAMD_MD_WAVES 16
// log2 n when the transform kernels have an instantiation for this size and the blob's FFT factors are the
// ones that instantiation assumes (radix 4 throughout, one radix-2 pass last for an odd log2 n); else 0
static size_t mdct_only_lds_bytes(const XformP &P, int waves) {
  const size_t n2 = P.n / 2;
  return ((size_t)(P.n + P.n / 4) + P.n / 4 + (size_t)waves * (n2 + VAMD_PW_SIZE(n2))) * 4;
}
template <int LOGN>
__global__ __launch_bounds__(64 * VAMD_MD_WAVES) void k_mdct_only(XformP G, int W, long nframes,
                                                                 const float *__restrict__ in,
                                                                 float *__restrict__ out) {
  const int n = LOGN ? (1 << LOGN) : G.n, n2 = n >> 1, nw = blockDim.x >> 6;
  float *trig = (float *)vamd_smem;          // [n + n/4]
  int *bitrev = (int *)(trig + n + n / 4);   // [n/4]
  float *work = (float *)(bitrev + n / 4);
  for (int i = threadIdx.x; i < n + n / 4; i += blockDim.x) trig[i] = G.trig[i];
  if (LOGN)  // the slot holds the butterfly stages' repacked trig instead (mdct_forward_wave<.., PACKED>)
    mdct_tpack_fill((float *)bitrev, G.trig, n, threadIdx.x, blockDim.x);
  else
    for (int i = threadIdx.x; i < n / 4; i += blockDim.x) bitrev[i] = G.bitrev[i];
  __syncthreads();
  XformP P = G;
  P.trig = trig;
  P.bitrev = LOGN ? nullptr : bitrev;
  P.tpack = LOGN ? (const float *)bitrev : nullptr;
  float *B = work + (size_t)(threadIdx.x >> 6) * (n2 + VAMD_PW_SIZE(n2));
  PhaseClock pc;
  pc.start(nullptr);
  for (long f = (long)blockIdx.x * nw + (threadIdx.x >> 6); f < nframes; f += (long)gridDim.x * nw) {
    mdct_forward_wave<0, LOGN, WaveTeam, LOGN != 0, true>(P, in + f * n, B, B, pc);
    WAVE_FOR(q, n2 >> 2)((F4 *)(out + f * n2))[q] = ((const F4 *)B)[q];
    WAVE_SYNC();
  }
}

// stage 1: window + MDCT + FFT + logs, one wave per channel-block.  Instantiated per block size (LOGN =
// log2 n; 0 = any size, read from the parameters).
#ifndef VAMD_XF_VGPRS
#define VAMD_XF_VGPRS 256
#endif
template <int LOGN>
#ifndef VAMD_XF_BOUND_WAVES  // (scratch builds: the register budget of a workgroup of this many waves, whatever is launched)
#define VAMD_XF_BOUND_WAVES VAMD_XF_WAVES
#endif
__global__ __launch_bounds__(64 * VAMD_XF_BOUND_WAVES) __attribute__((amdgpu_num_vgpr(VAMD_XF_VGPRS))) void k_transform(XformP G, int W, DescP d, int ch, long ncb,
                                                                 const float *__restrict__ pcm,
                                                                 float *__restrict__ mdct_raw,
                                                                 float *__restrict__ logmdct,
                                                                 float *__restrict__ logfft,
                                                                 float *__restrict__ local_ampmax,
                                                                 const unsigned short *__restrict__ run_of_bin, int nruns,
                                                                 int nrp, float *__restrict__ peaks) {
  const XformLds L = stage_transform_tables<LOGN>(G);
  const XformP &P = L.P;
  const int n = LOGN ? (1 << LOGN) : P.n, n2 = n >> 1, nw = blockDim.x >> 6;
  PhaseClock pc;
  pc.start(d.dbg);
  // cb = channel-block index = block*ch + channel
  const long cstride = (long)gridDim.x * nw;
  long cb = (long)blockIdx.x * nw + (threadIdx.x >> 6);
  constexpr int QPT = LOGN ? ((1 << LOGN) / 4 + 63) / 64 : 4096 / 4 / 64;  // quads of a block per lane
  const WaveTeam tm;
  PcmTile<QPT> tile;
  // A block's samples AND its window flags are fetched one block ahead: a load issued at the top of the loop -- even a
  // conditional one that is not taken -- makes the wait there a wait for everything outstanding, the stores of the
  // previous block's spectra included.
  int lW = 0, nW = 0;
  I2 rid[VAMD_XF_QPS(LOGN)];  // which run of bins each of this lane's bins belongs to: the same for every block
  if (peaks) xf_run_ids<LOGN>(P, run_of_bin, rid, tm);
  // where a channel-block's samples start: packed [block][channel][n], or in place (a stream plan's offsets)
  auto samples = [&](long cbi, long blk) -> const float * {
    return d.src ? pcm + d.src[blk] + (cbi - blk * ch) * d.cstride : pcm + cbi * n;
  };
  if (cb < ncb) {
    const long blk = (long)((unsigned)cb / (unsigned)ch);
    lW = d_lW(d, blk), nW = d_nW(d, blk);
    pcm_fetch(tile, samples(cb, blk), n, tm);
  }
  for (; cb < ncb; cb += cstride) {
#ifdef VAMD_XF_NO_PREFETCH  // (scratch builds, profiles/r05_xf_variants.txt: what the next block's samples in registers are worth)
    {
      const long blk0 = (long)((unsigned)cb / (unsigned)ch);
      lW = d_lW(d, blk0), nW = d_nW(d, blk0);
      pcm_fetch(tile, samples(cb, blk0), n, tm);
    }
#endif
    transform_window(P, W, lW, nW, tile, L.A, pc, tm);
#ifndef VAMD_XF_NO_PREFETCH
    if (cb + cstride < ncb) {  // next block, one ahead
      const long blk = (long)((unsigned)(cb + cstride) / (unsigned)ch);
      lW = d_lW(d, blk), nW = d_nW(d, blk);
      pcm_fetch(tile, samples(cb + cstride, blk), n, tm);
    }
#endif
    float raw;
    // (logfft goes out as what the tone stage reads of it -- its peak over each run of bins of one octave line, nrp
    // floats per channel-block -- and in full only where a caller taps it)
    const float amp = transform_block<LOGN>(P, L.A, L.B, mdct_raw + cb * n2, logmdct ? logmdct + cb * n2 : nullptr,
                                            logfft ? logfft + cb * n2 : nullptr, pc, tm, &raw, rid, run_of_bin, nruns,
                                            peaks ? peaks + cb * nrp : nullptr);
    if (LANE == 0) {
      local_ampmax[cb] = amp;
      const bool bad = raw > VAMD_NONFINITE_DB;  // outside the input domain: the block's arithmetic is not finite
      d.status[cb] = bad ? VAMD_STATUS_NONFINITE : 0;
      if (bad) {
        atomicAdd(d.bad, 1u);
        atomicAdd(d.bad + 2, 1u);
      }
    }
  }
  pc.flush();
}

// stage 2: _vp_noisemask.  One workgroup ("team") of up to four waves per channel-block: every wave takes a
// quarter of the bins for the per-bin phases, the first wave walks the five ordered running sums
// (ScanTeam, k_noise.h).  LDS per team: the five sums (20.3 KB at 1024 bins), so a CU holds seven teams;
// what hides the ~17k cycles a block spends in its two ordered walks is the other six teams.
// LOGN2 = log2 of the bin count n/2.
template <int LOGN2>
struct NoiseGeom {
  static constexpr int n2 = 1 << LOGN2;
  static constexpr int NW = n2 >= 256 ? 4 : (n2 >= 64 ? n2 / 64 : 1);  // waves per team
  static constexpr int KPL = n2 / (64 * NW) > 0 ? n2 / (64 * NW) : 1;  // bins per lane
};
// the stage for the teams `first`, first + nteams, ... of the batch (k_noise: the whole grid; k_noise_tone: its first part)
template <int LOGN2>
__global__ __launch_bounds__(64 * NoiseGeom<LOGN2>::NW, NoiseGeom<LOGN2>::KPL <= 4 ? 8 : 4) void k_noise(PsyP P0, PsyP P1, DescP d, int ch, long ncb,
                                                                     const float *__restrict__ mdct_raw,
                                                                     float *__restrict__ noise) {
  constexpr int n2 = NoiseGeom<LOGN2>::n2, KPL = NoiseGeom<LOGN2>::KPL;
  float *S = (float *)vamd_smem;  // the five running sums and nothing else: see VAMD_NZ_STRIDE
  const int i0 = (threadIdx.x >> 6) * 64 * KPL;  // this wave's first bin
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 16 : nullptr);
  // persistent.  The next block's spectrum is fetched at the end of this one, not a block ahead: four registers held across
  // a whole block cost more (the stage sits on the 64-register line) than the fetch does beside five other teams
  float lm[KPL];
  int braw[KPL], bk[KPL], bt_have = -1;
  float compand_lane = 0.f;  // noisecompand[LANE]
  long cb = blockIdx.x;
  // (the spectrum in dB, lib/mapping0.c:384-385, is formed here from the spectrum itself: nobody writes it to HBM)
  if (cb < ncb) LANE_BINS(k, i, i0, KPL, n2) lm[k] = mdct_raw[cb * n2 + i];
  for (; cb < ncb; cb += gridDim.x) {
    const int bt = d_bt(d, (long)((unsigned)cb / (unsigned)ch));  // (cb < 2^31: check_desc)
    const PsyP &P = bt ? P1 : P0;
    float o[KPL];
    const long nb = cb + gridDim.x < ncb ? cb + gridDim.x : cb;
    if (bt != bt_have) {  // the window edges of this lane's bins and noisecompand[]: properties of the block type, kept across blocks
      noise_bark_fetch<KPL, LOGN2>(P, braw, i0);
      noise_bark_edges<KPL, LOGN2>(P, braw, bk, i0);
      compand_lane = LANE < VAMD_NOISE_COMPAND_LEVELS ? P.noisecompand[LANE] : 0.f;
      bt_have = bt;
    }
    LANE_BINS(k, i, i0, KPL, n2) lm[k] = todB_345(lm[k]);
    noisemask_bins<ScanTeam, KPL, LOGN2>(
        P, lm, bk, o, S,
        [&](int dB) { return __int_as_float(__builtin_amdgcn_ds_bpermute(dB << 2, __float_as_int(compand_lane))); }, ScanTeam(), pc,
        i0);
    LANE_BINS(k, i, i0, KPL, n2) noise[cb * n2 + i] = o[k];
    LANE_BINS(k, i, i0, KPL, n2) lm[k] = mdct_raw[nb * n2 + i];
  }
  pc.flush();
}

// The same stage for the one block of workgroup blockIdx.x (k_noise_tone).  A restatement of k_noise's body, not a function
// the two share: the batch kernel sits exactly on its 64-register line, and as a caller of a shared
// body it came out with six registers spilt.
template <int LOGN2>
__device__ __forceinline__ void noise_teams_once(const PsyP &P0, const PsyP &P1, const DescP &d, int ch, const float *__restrict__ mdct_raw,
                                                 float *__restrict__ noise) {
  constexpr int n2 = NoiseGeom<LOGN2>::n2, KPL = NoiseGeom<LOGN2>::KPL;
  float *S = (float *)vamd_smem;  // the five running sums and nothing else: see VAMD_NZ_STRIDE
  const int i0 = (threadIdx.x >> 6) * 64 * KPL;  // this wave's first bin
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 16 : nullptr);
  const long cb = blockIdx.x;
  float lm[KPL], o[KPL];
  int braw[KPL], bk[KPL];
  LANE_BINS(k, i, i0, KPL, n2) lm[k] = mdct_raw[cb * n2 + i];
  const PsyP &P = d_bt(d, (long)((unsigned)cb / (unsigned)ch)) ? P1 : P0;
  noise_bark_fetch<KPL, LOGN2>(P, braw, i0);
  noise_bark_edges<KPL, LOGN2>(P, braw, bk, i0);
  const float compand_lane = LANE < VAMD_NOISE_COMPAND_LEVELS ? P.noisecompand[LANE] : 0.f;
  LANE_BINS(k, i, i0, KPL, n2) lm[k] = todB_345(lm[k]);  // (the spectrum in dB, lib/mapping0.c:384-385)
  noisemask_bins<ScanTeam, KPL, LOGN2>(
      P, lm, bk, o, S,
      [&](int dB) { return __int_as_float(__builtin_amdgcn_ds_bpermute(dB << 2, __float_as_int(compand_lane))); }, ScanTeam(), pc,
      i0);
  LANE_BINS(k, i, i0, KPL, n2) noise[cb * n2 + i] = o[k];
  pc.flush();
}

// block-level ampmax: global = max(ampmax_in, local[0..ch)); one thread per block
__global__ void k_ampmax(DescP d, int ch, long nblocks, const float *__restrict__ local_ampmax,
                         float *__restrict__ ampmax_glob) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  float g = d_amp(d, b);
  for (int c = 0; c < ch; c++) {
    const float l = local_ampmax[b * ch + c];
    if (l > g) g = l;  // lib/mapping0.c:346
  }
  ampmax_glob[b] = g;  // becomes vbi->ampmax, lib/mapping0.c:576
}

// stream mode: the ampmax recurrence across blocks, serial in fp32:
// in_k = max(out_{k-1} + secs*att, -9999), out_k = max(in_k, locals_k)
__global__ void k_ampmax_stream(int ch, long nblocks, float secs, float att, float state,
                                const float *__restrict__ local_ampmax, float *__restrict__ ampmax_in,
                                float *__restrict__ ampmax_glob) {
  if (blockIdx.x || threadIdx.x) return;
  float amp = state;
  for (long b = 0; b < nblocks; b++) {
    amp += secs * att;  // _vp_ampmax_decay, lib/psy.c:837-848
    if (amp < -9999) amp = -9999;
    ampmax_in[b] = amp;
    for (int c = 0; c < ch; c++) {
      const float l = local_ampmax[b * ch + c];
      if (l > amp) amp = l;
    }
    ampmax_glob[b] = amp;
  }
}

// a stream that mixes both size classes: order[k] = W << 30 | index inside W's batch
__global__ void k_ampmax_stream_mixed(int ch, long ntotal, const int *__restrict__ order, float secs0, float secs1,
                                      float att, float state, const float *__restrict__ local0,
                                      const float *__restrict__ local1, float *__restrict__ in0,
                                      float *__restrict__ in1, float *__restrict__ glob0, float *__restrict__ glob1,
                                      float *__restrict__ state_out, int first_given) {
  if (blockIdx.x || threadIdx.x) return;
  float amp = state;
  for (long k = 0; k < ntotal; k++) {
    const int o = order[k], W = (o >> 30) & 1;
    const long b = o & 0x3fffffff;
    if (!(first_given && k == 0)) {  // (first_given: `state` is what block 0 receives, already decayed by the caller's blockout)
      amp += (W ? secs1 : secs0) * att;  // _vp_ampmax_decay with vd->W = this block's size class
      if (amp < -9999) amp = -9999;
    }
    (W ? in1 : in0)[b] = amp;
    const float *loc = W ? local1 : local0;
    for (int c = 0; c < ch; c++) {
      const float l = loc[b * ch + c];
      if (l > amp) amp = l;
    }
    (W ? glob1 : glob0)[b] = amp;
  }
  *state_out = amp;
}

// many streams at once: thread s walks order[start[s] .. start[s+1]) with its own running state
__global__ void k_ampmax_streams_mixed(int ch, long nstreams, const long long *__restrict__ start,
                                       const int *__restrict__ order, float secs0, float secs1, float att,
                                       float *__restrict__ states, const float *__restrict__ local0,
                                       const float *__restrict__ local1, float *__restrict__ in0,
                                       float *__restrict__ in1, float *__restrict__ glob0, float *__restrict__ glob1) {
  const long sidx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (sidx >= nstreams) return;
  float amp = states[sidx];
  for (long long k = start[sidx]; k < start[sidx + 1]; k++) {
    const int o = order[k], W = (o >> 30) & 1;
    const long b = o & 0x3fffffff;
    amp += (W ? secs1 : secs0) * att;
    if (amp < -9999) amp = -9999;
    (W ? in1 : in0)[b] = amp;
    const float *loc = W ? local1 : local0;
    for (int c = 0; c < ch; c++) {
      const float l = loc[b * ch + c];
      if (l > amp) amp = l;
    }
    (W ? glob1 : glob0)[b] = amp;
  }
  states[sidx] = amp;
}

// stage 3: _vp_tonemask, in three launches (k_tone.h).  nlp = octave lines padded to 32 (VAMD_LINES_PAD).
template <int LP>
__global__ __launch_bounds__(64) void k_tone_seed(PsyP P0, PsyP P1, DescP d, int ch, int nlp, int nrp,
                                                  const float *__restrict__ peaks,
                                                  const float *__restrict__ local_ampmax,
                                                  const float *__restrict__ ampmax_glob, float *__restrict__ ampmax_make,
                                                  float *__restrict__ seed_g) {
  const long cb = blockIdx.x;
  const long blk = cb / ch;
  const PsyP &P = d_bt(d, blk) ? P1 : P0;
  // the block's ampmax (lib/mapping0.c:346,576): read where a stream's chain has already formed it; otherwise formed here
  // -- max of the incoming value and the channels' spectral peaks, what k_ampmax would have launched for -- and written
  // once per block by its first channel's wave
  float g_amp;
  if (ampmax_make) {
    g_amp = d_amp(d, blk);
    for (int c = 0; c < ch; c++) {
      const float l = local_ampmax[blk * ch + c];
      if (l > g_amp) g_amp = l;
    }
    if (LANE == 0 && cb == blk * ch) ampmax_make[blk] = g_amp;
  } else {
    g_amp = ampmax_glob[blk];
  }
  const int n2 = P.n, nl = P.total_octave_lines;
  float *seed = (float *)vamd_smem + seed_pad_lo(P.eighth_octave_lines);  // padded either side, see seed_curve_scatter
  (void)n2;
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 32 : nullptr);
  tone_seed_block<LP>(P, peaks + cb * nrp, g_amp, local_ampmax[cb], seed, pc);
  WAVE_FOR(i, nlp) seed_g[cb * nlp + i] = i < nl ? seed[i] : VAMD_NEGINF;
  pc.flush();
}

// seed + chase in one launch for a small batch: one WAVE per channel-block scatters the curves into the lines in LDS
// and walks them there, cut into one chunk per lane with the walk's state in registers (chase_chunk_regs, k_tone.h) --
// the lines go out for the fold but do not come back in, and the chain is a launch shorter (a lone block: k_tone_seed
// 9 us + k_tone_chase_wave 25 us -> this kernel's 14).  Every libvorbisenc setup has eight lines per window; others
// take the two kernels.  LDS: the padded seed lines, then a 16-slot ring for the serial walk's fallback.
template <int LP>
__device__ __forceinline__ void tone_seed_chase_run(const PsyP &P0, const PsyP &P1, const DescP &d, int ch, int nlp, int nrp,
                                                    const float *__restrict__ peaks, const float *__restrict__ local_ampmax,
                                                    const float *__restrict__ ampmax_glob, float *__restrict__ ampmax_make,
                                                    float *__restrict__ seed_g, unsigned short *__restrict__ surv,
                                                    int *__restrict__ nsurv, long cb) {
  const long blk = cb / ch;
  const PsyP &P = d_bt(d, blk) ? P1 : P0;
  float g_amp;  // (the block's ampmax: as k_tone_seed)
  if (ampmax_make) {
    g_amp = d_amp(d, blk);
    for (int c = 0; c < ch; c++) {
      const float l = local_ampmax[blk * ch + c];
      if (l > g_amp) g_amp = l;
    }
    if (LANE == 0 && cb == blk * ch) ampmax_make[blk] = g_amp;
  } else {
    g_amp = ampmax_glob[blk];
  }
  const int nl = P.total_octave_lines;
  float *seed = (float *)vamd_smem + seed_pad_lo(LP);
  float *ring_amp = seed + nlp + seed_pad_hi(LP);
  int *ring_pos = (int *)(ring_amp + VAMD_RING);
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 32 : nullptr);
  tone_seed_block<LP>(P, peaks + cb * nrp, g_amp, local_ampmax[cb], seed, pc);
  WAVE_FOR(i, nlp) {
    if (i >= nl) seed[i] = VAMD_NEGINF;  // (the row's padding, as it goes out: the serial walk reads whole lines of it)
    seed_g[cb * nlp + i] = seed[i];
  }
  WAVE_SYNC();
  const int cs = (nl + 63) / 64;
  const int s0 = LANE * cs < nl ? LANE * cs : nl, e0 = s0 + cs < nl ? s0 + cs : nl;
  unsigned short *out = surv + cb * nlp;
  bool accepted = false;
  ChaseChunk r;
  r.popped = r.sig_in = r.sig_out = 0;
  r.exact = 1;
  // a long run of equal values (a stretch no curve reached) would take a repair round per chunk: serial at once
  unsigned long long fm = __ballot(s0 < nl && chase_flat_chunk(seed, s0, e0));
  int longest = 0;
  for (; fm && longest <= VAMD_CHASE_FLAT_MAX; longest++) fm &= fm << 1;
  if (longest <= VAMD_CHASE_FLAT_MAX) {
    r = chase_chunk_regs<LP>(seed, nl, s0, e0, cs, VAMD_CHASE_WARM * LP, 0);  // (a lane past the last line walks nothing)
    uint32_t used = r.sig_in;
    for (int rd = 0; rd <= VAMD_CHASE_ROUNDS; rd++) {
      const uint32_t prev_out = (uint32_t)wave_shift_up1((int)r.sig_out, 0);
      const bool need = s0 < nl && !r.exact && used != prev_out;
      if (!__any(need)) {
        accepted = true;
        break;
      }
      if (rd == VAMD_CHASE_ROUNDS) break;
      if (need) {  // walked again, started exactly in the predecessor's exit state
        const ChaseChunk t = chase_chunk_regs<LP>(seed, nl, s0, e0, cs, -1, prev_out);
        used = prev_out;
        r.popped = t.popped;
        r.sig_out = t.sig_out;
      }
    }
  }
  if (accepted) {
    const uint32_t alive = ~r.popped & (e0 - s0 >= 32 ? ~0u : ((1u << (e0 - s0)) - 1u));
    const int cnt = __builtin_popcount(alive);
    const int incl = wave_scan_sum(cnt);
    int at = incl - cnt;
    for (uint32_t m = alive; m; m &= m - 1) out[at++] = (unsigned short)(s0 + __builtin_ctz(m));
    if (LANE == 63) nsurv[cb] = incl;
  } else if (LANE == 0) {
    nsurv[cb] = tone_chase_thread(seed, LP, nl, ring_amp, ring_pos, 1, 0, out);
  }
  pc.mark(2);
  pc.flush();
}
template <int LP>
__global__ __launch_bounds__(64) void k_tone_seed_chase(PsyP P0, PsyP P1, DescP d, int ch, int nlp, int nrp,
                                                        const float *__restrict__ peaks,
                                                        const float *__restrict__ local_ampmax,
                                                        const float *__restrict__ ampmax_glob, float *__restrict__ ampmax_make,
                                                        float *__restrict__ seed_g, unsigned short *__restrict__ surv,
                                                        int *__restrict__ nsurv) {
  tone_seed_chase_run<LP>(P0, P1, d, ch, nlp, nrp, peaks, local_ampmax, ampmax_glob, ampmax_make, seed_g, surv, nsurv, blockIdx.x);
}

// Both masks of a handful of blocks in ONE launch: workgroups [0, ncb) are the noise stage's teams, [ncb, 2 ncb) the tone
// chain's waves (the first wave of the workgroup; the others leave).  The two stages need nothing of each other, and a
// second stream with its event pair costs a lone block as much as it saves -- below 64 channel-blocks they used to run one
// after the other (a stereo block: 13 + 24 us of its latency; here 24).
template <int LOGN2, int LP>
__global__ __launch_bounds__(64 * NoiseGeom<LOGN2>::NW) void k_noise_tone(PsyP P0, PsyP P1, DescP d, int ch, long ncb,
                                                                          const float *__restrict__ mdct_raw, float *__restrict__ noise,
                                                                          int nlp, int nrp, const float *__restrict__ peaks,
                                                                          const float *__restrict__ local_ampmax,
                                                                          const float *__restrict__ ampmax_glob, float *__restrict__ ampmax_make,
                                                                          float *__restrict__ seed_g, unsigned short *__restrict__ surv,
                                                                          int *__restrict__ nsurv) {
  if ((long)blockIdx.x < ncb) {
    noise_teams_once<LOGN2>(P0, P1, d, ch, mdct_raw, noise);
  } else if (threadIdx.x < 64) {
    tone_seed_chase_run<LP>(P0, P1, d, ch, nlp, nrp, peaks, local_ampmax, ampmax_glob, ampmax_make, seed_g, surv, nsurv,
                            (long)blockIdx.x - ncb);
  }
}

// one THREAD per channel-block: the ordered stack walk of seed_chase, VAMD_CHASE_LANES walks per wave.  (Measured
// round 2: half-filled waves -- twice as many waves for the SIMDs to interleave -- are slower, 1.01 against 0.82 ms per
// 131 072 stereo blocks; a walk whose stack is a register bit mask fed through coalesced LDS tiles executes three
// times the instructions once 64 divergent walks share them, 2.96 ms.  tools/pmc_quick.sh has the counters.)
#define VAMD_CHASE_LANES 64
__global__ __launch_bounds__(64) void k_tone_chase(int linesper, int nl, int nlp, long ncb, DescP d,
                                                   const float *__restrict__ seed_g,
                                                   unsigned short *__restrict__ surv, int *__restrict__ nsurv) {
  float *ring_amp = (float *)vamd_smem;
  int *ring_pos = (int *)(ring_amp + VAMD_RING * VAMD_CHASE_LANES);
  const long cb = (long)blockIdx.x * VAMD_CHASE_LANES + threadIdx.x;
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 32 : nullptr);
  // Measurement aid (vamd_clock_probe): the shader clock while the chip is busy.  The first wave of this launch --
  // which runs beside the noise mask, the path's longest stage -- adds the shader ticks (s_memtime: the counter the
  // issue costs of tools/micro/chip_rate.hip are priced in) and the ticks of the chip-wide 100 MHz clock
  // (s_memrealtime) of its own life to the caller's accumulator.  No launch of its own: a probe kernel on a third stream
  // sat in front of this very chain often enough to show (tone tail 0.42 -> 0.46-0.51 ms).
  const bool probe = d.clk && blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long w0 = 0;
  long long t0 = 0;
  if (probe) w0 = wall_clock64(), t0 = clock64();
  if (cb < ncb)
    nsurv[cb] = tone_chase_thread(seed_g + cb * nlp, linesper, nl, ring_amp, ring_pos, VAMD_CHASE_LANES, threadIdx.x, surv + cb * nlp);
  if (probe) {
    atomicAdd(d.clk, (unsigned long long)(clock64() - t0));
    atomicAdd(d.clk + 1, wall_clock64() - w0);
    atomicAdd(d.clk + 2, 1ull);
  }
  pc.mark(2);
  pc.flush();
}

// the same for a small batch: one WAVE per channel-block, the walk cut into one chunk per lane (chase_chunk, k_tone.h)
// LDS: the block's seed lines [nlp], then the rings [VAMD_RING][64] x 2
__global__ __launch_bounds__(64) void k_tone_chase_wave(int linesper, int nl, int nlp, DescP d,
                                                        const float *__restrict__ seed_g,
                                                        unsigned short *__restrict__ surv, int *__restrict__ nsurv) {
  float *seed = (float *)vamd_smem;
  float *ring_amp = seed + nlp;
  int *ring_pos = (int *)(ring_amp + VAMD_RING * 64);
  const long cb = blockIdx.x;
  WAVE_FOR(q, nlp >> 2)((F4 *)seed)[q] = ((const F4 *)(seed_g + cb * nlp))[q];
  WAVE_SYNC();
  const int cs = (nl + 63) / 64;
  const int s0 = LANE * cs < nl ? LANE * cs : nl, e0 = s0 + cs < nl ? s0 + cs : nl;
  unsigned short *out = surv + cb * nlp;
  bool accepted = false;
  ChaseChunk r;
  r.popped = r.sig_in = r.sig_out = 0;
  r.exact = 1;
  // a long run of equal values (a stretch no curve reached) would take a repair round per chunk: serial at once
  unsigned long long fm = __ballot(s0 < nl && chase_flat_chunk(seed, s0, e0));
  int longest = 0;
  for (; fm && longest <= VAMD_CHASE_FLAT_MAX; longest++) fm &= fm << 1;
  if (longest <= VAMD_CHASE_FLAT_MAX) {
    if (s0 < nl) r = chase_chunk(seed, linesper, nl, s0, e0, VAMD_CHASE_WARM * linesper, 0, ring_amp, ring_pos, 64, LANE);
    uint32_t used = r.sig_in;
    for (int rd = 0; rd <= VAMD_CHASE_ROUNDS; rd++) {
      const uint32_t prev_out = (uint32_t)wave_shift_up1((int)r.sig_out, 0);
      const bool need = s0 < nl && !r.exact && used != prev_out;
      if (!__any(need)) {
        accepted = true;
        break;
      }
      if (rd == VAMD_CHASE_ROUNDS) break;
      if (need) {  // walked again, started exactly in the predecessor's exit state
        const ChaseChunk t = chase_chunk(seed, linesper, nl, s0, e0, -1, prev_out, ring_amp, ring_pos, 64, LANE);
        used = prev_out;
        r.popped = t.popped;
        r.sig_out = t.sig_out;
      }
    }
  }
  if (accepted) {
    const uint32_t alive = ~r.popped & (e0 - s0 >= 32 ? ~0u : ((1u << (e0 - s0)) - 1u));
    const int cnt = __builtin_popcount(alive);
    const int incl = wave_scan_sum(cnt);
    int at = incl - cnt;
    for (uint32_t m = alive; m; m &= m - 1) out[at++] = (unsigned short)(s0 + __builtin_ctz(m));
    if (LANE == 63) nsurv[cb] = incl;
  } else if (LANE == 0) {
    nsurv[cb] = tone_chase_thread(seed, linesper, nl, ring_amp, ring_pos, 64, 0, out);
  }
}

__global__ __launch_bounds__(64) void k_tone_fold(PsyP P0, PsyP P1, DescP d, int ch, int nlp,
                                                  const float *__restrict__ seed_g,
                                                  const unsigned short *__restrict__ surv,
                                                  const int *__restrict__ nsurv,
                                                  const float *__restrict__ local_ampmax, float *__restrict__ tone) {
  const long cb = blockIdx.x;
  const long blk = cb / ch;
  const PsyP &P = d_bt(d, blk) ? P1 : P0;
  const int n2 = P.n;
  float *seed = (float *)vamd_smem;  // [nlp]
  float *gmin = seed + nlp;          // [ngroups]
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 32 : nullptr);
  WAVE_FOR(q, nlp >> 2)((F4 *)seed)[q] = ((const F4 *)(seed_g + cb * nlp))[q];
  WAVE_SYNC();
  tone_fold_block(P, local_ampmax[cb], seed, seed, surv + cb * nlp, nsurv[cb], gmin, tone + cb * n2, pc);
  pc.flush();
}

// stage 4: offset_and_mix + floor1_fit + floor curve
// (eight waves per SIMD, i.e. 64 registers: measured against the 73 the compiler would take and six or seven waves --
// the stage is latency-bound, its time follows the blocks in flight: tools/floor_occ.sh -- 2.14 against 2.24 ms)
// With `seed_g` the stage begins with the tone chain's last step (tone_fold_block: paint the chase's survivors,
// max_seeds' fold) for its own channel-block: the tone curve then goes out and comes straight back through L2 inside
// one wave instead of through a launch boundary, and the fold's waits sit among thirty-one other waves' floor fits.
// The fold's LDS (seed lines + group minima, 3.7 KB) is the fit's own, used before it.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_floor(const Bound *__restrict__ Bd, int W, DescP d, int ch,
                                              const float *__restrict__ noise, float *__restrict__ tone,
                                              const float *__restrict__ seed_g, const unsigned short *__restrict__ surv,
                                              const int *__restrict__ nsurv, const float *__restrict__ local_ampmax, int nlp,
                                              const float *__restrict__ mdct_raw,
                                              float *__restrict__ mdct, float *__restrict__ logmask_out,
                                              int *__restrict__ posts, int *__restrict__ post_valid,
                                              ilog_t *__restrict__ ilogmask, int *__restrict__ nonzero,
                                              int *__restrict__ wrapped /* [cb][VAMD_POSTS_STRIDE] for k_pack, or null */) {
  const long cb = blockIdx.x;
  const long blk = cb / ch;
  // (the parameter structs stay in HBM and are read field by field through the scalar cache: four of them by value
  // are more SGPRs than the stage has)
  const PsyP &P = Bd->psy[2 * W + (d_bt(d, blk) ? 1 : 0)];
  const FloorP &F = Bd->floor[W][Bd->chmap[W].sub[cb - blk * ch]];  // the floor of this channel's submap
  const int n2 = P.n;
  unsigned short *qc = (unsigned short *)vamd_smem;  // [n2 rounded up to 16]
  FloorScratch *sc = (FloorScratch *)(qc + ((n2 + 15) & ~15));
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 48 : nullptr);
#ifdef VAMD_STOP_AFTER
  pc.stoppable = true;
#endif
  if (seed_g) {
    float *seed = (float *)vamd_smem;  // [nlp], then the group minima
    // (one trip to memory for both: the head of the survivor list is asked for before the lines, and the survivors'
    // amplitudes then come out of the lines' LDS copy)
    const bool ahead = nlp >= 2 * 64 + 2;  // (the row holds the entries surv_head_load reads)
    SurvHead head;
    if (ahead) head = surv_head_load(surv + cb * nlp);
    WAVE_FOR(q, nlp >> 2)((F4 *)seed)[q] = ((const F4 *)(seed_g + cb * nlp))[q];
    WAVE_SYNC();
    tone_fold_prepare(P, seed, seed, surv + cb * nlp, nsurv[cb], seed + nlp, pc, 5, ahead ? &head : nullptr);
    fold_and_mix_wave(P, tone_ath_att(P, local_ampmax[cb]), seed, seed + nlp, noise + cb * n2, tone ? tone + cb * n2 : nullptr,
                      mdct_raw + cb * n2, mdct + cb * n2, logmask_out ? logmask_out + cb * n2 : nullptr, qc, F.twofitatten, pc);
  } else {
    offset_and_mix_wave(P, noise + cb * n2, tone + cb * n2, mdct_raw + cb * n2, mdct + cb * n2,
                        logmask_out ? logmask_out + cb * n2 : nullptr, qc, F.twofitatten, pc);
  }
  const int nzf = floor_fit_render_block(F, n2, qc, sc, posts + cb * VAMD_POSTS_STRIDE, post_valid + cb,
                                         ilogmask + cb * n2, pc, wrapped ? wrapped + cb * VAMD_POSTS_STRIDE : nullptr);
  if (LANE == 0) nonzero[cb] = nzf;
  pc.flush();
}

// The same stage with the two channels of a stereo block in ONE wave, 32 lanes each (vamd_wave_pair.h: the bodies of
// k_floor.inc compiled against the half-wave vocabulary).  One pass through the ordered sections -- the greedy split loop,
// the level loops -- serves both channels, and a short block's 128 bins fill a half where they left half a wave idle.
// Launched for stereo setups whose two channels share a floor (launch_rest); everything per block (psy look, floor,
// sizes) is wave-uniform as before, everything per channel lives in the half's lanes.  LDS: `half_bytes` per half.
#ifndef VAMD_FLOOR_PAIR_WAVES
#define VAMD_FLOOR_PAIR_WAVES 6  // waves per SIMD: 77 registers, nothing spilt (4 / 5 / 6 / 8 measured)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(VAMD_FLOOR_PAIR_WAVES, VAMD_FLOOR_PAIR_WAVES))) void k_floor_pair(const Bound *__restrict__ Bd, int W, DescP d, int half_bytes,
                                                   const float *__restrict__ noise, float *__restrict__ tone,
                                                   const float *__restrict__ seed_g, const unsigned short *__restrict__ surv,
                                                   const int *__restrict__ nsurv, const float *__restrict__ local_ampmax, int nlp,
                                                   const float *__restrict__ mdct_raw,
                                                   float *__restrict__ mdct, float *__restrict__ logmask_out,
                                                   int *__restrict__ posts, int *__restrict__ post_valid,
                                                   ilog_t *__restrict__ ilogmask, int *__restrict__ nonzero,
                                                   int *__restrict__ wrapped /* [cb][VAMD_POSTS_STRIDE] for k_pack, or null */) {
  typedef vamd::pair::Bodies vp;
  const long blk = blockIdx.x;
  const long cb = blk * 2 + VAMD_PAIR_HALF;
  const PsyP &P = Bd->psy[2 * W + (d_bt(d, blk) ? 1 : 0)];
  const FloorP &F = Bd->floor[W][Bd->chmap[W].sub[0]];  // (both channels' floor: the launch checked)
  const int n2 = P.n;
  unsigned char *mine = vamd_smem + (size_t)VAMD_PAIR_HALF * half_bytes;
  unsigned short *qc = (unsigned short *)mine;  // [n2 rounded up to 16]
  vp::FloorScratch *sc = (vp::FloorScratch *)(qc + ((n2 + 15) & ~15));
  PhaseClock pc;
  pc.start(nullptr);
  if (seed_g) {
    float *seed = (float *)mine;  // [nlp], then the group minima
    const bool ahead = nlp >= 2 * 32 + 2;
    vp::SurvHead head;
    if (ahead) head = vp::surv_head_load(surv + cb * nlp);
    for (int q = (int)(threadIdx.x & 31); q < (nlp >> 2); q += 32) ((F4 *)seed)[q] = ((const F4 *)(seed_g + cb * nlp))[q];
    WAVE_SYNC();
    vp::tone_fold_prepare(P, seed, seed, surv + cb * nlp, nsurv[cb], seed + nlp, pc, 5, ahead ? &head : nullptr);
    vp::fold_and_mix_wave(P, vp::tone_ath_att(P, local_ampmax[cb]), seed, seed + nlp, noise + cb * n2, tone ? tone + cb * n2 : nullptr,
                          mdct_raw + cb * n2, mdct + cb * n2, logmask_out ? logmask_out + cb * n2 : nullptr, qc, F.twofitatten, pc);
  } else {
    vp::offset_and_mix_wave(P, noise + cb * n2, tone + cb * n2, mdct_raw + cb * n2, mdct + cb * n2,
                            logmask_out ? logmask_out + cb * n2 : nullptr, qc, F.twofitatten, pc);
  }
  const int nzf = vp::floor_fit_render_block(F, n2, qc, sc, posts + cb * VAMD_POSTS_STRIDE, post_valid + cb,
                                             ilogmask + cb * n2, pc, wrapped ? wrapped + cb * VAMD_POSTS_STRIDE : nullptr);
  if ((threadIdx.x & 31) == 0) nonzero[cb] = nzf;
}

// the int32 `ilogmask` tap of the C ABI from the 16-bit curve the stages exchange
__global__ void k_widen_ilog(long n, const ilog_t *__restrict__ in, int *__restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}

// stage 5: couple / quantise / normalise, one wave per block (all channels)
// A unit is one (block, candidate packet): VBR has one packet per block (blob_base = PACKETBLOBS/2,
// nblobs = 1), a bitrate-managed block all fifteen, each with its own coupling parameters over the
// same spectrum.  ilogmask / iwork / nonzero are indexed by unit, mdct by block.
// NORM = false may be launched with several waves per unit (small batches: couple_block deals its quads over the team)
template <bool NORM>
__device__ __forceinline__ void couple_unit(const PsyP &P0, const PsyP &P1, const CoupleSet &CS, int blob_base, int nblobs, const DescP &d,
                                            const float *__restrict__ mdct, const ilog_t *__restrict__ ilogmask,
                                            int *__restrict__ iwork, int *__restrict__ nonzero, float band) {
  const long unit = blockIdx.x, mblk = unit / nblobs;
  const CoupleP &C = CS.c[blob_base + (int)(unit - mblk * nblobs)];
  const PsyP &P = d_bt(d, mblk) ? P1 : P0;
  const int n2 = P.n, ch = C.ch;
  const long blk = unit;
  CoupleLds L;
  L.cand = (float *)vamd_smem;
  L.key = L.cand + n2;
  L.sgn = L.key + n2;
  L.accp = L.sgn + n2;
  const float *mp[VAMD_MAX_CH];
  const ilog_t *ip[VAMD_MAX_CH];
  int *op[VAMD_MAX_CH];
  int nz[VAMD_MAX_CH];
  for (int c = 0; c < ch; c++) {
    mp[c] = mdct + (mblk * ch + c) * n2;
    ip[c] = ilogmask + (blk * ch + c) * n2;
    op[c] = iwork + (blk * ch + c) * n2;
    nz[c] = nonzero[blk * ch + c];
  }
  WAVE_SYNC_GLOBAL();  // every lane has read nonzero[] before lane 0 rewrites it
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 64 : nullptr);
  unsigned over = 0;
  couple_block<NORM>(C, P, n2, mp, ip, op, nz, L, pc, band, &over);
  if (TEAM_LEADER)
    for (int c = 0; c < ch; c++) nonzero[blk * ch + c] = nz[c];
  for (int c = 0; c < ch; c++) {  // (the team's lanes each saw their own quads)
    const int any = __syncthreads_or((int)((over >> c) & 1u));
    if (TEAM_LEADER && any) flag_range(d, mblk * ch + c);
  }
  pc.flush();
}
// Two kernels, one body: the usual one (no ordered path compiled in) and the one with noise normalisation's sort.
// (The input domain's watch over the written values, round 5, is one pair of running extremes per channel: 78 / 127
// registers, six / four waves per SIMD as before.  A watch that also knew the residue's coded bins cost a wave per SIMD
// and 10 % of the stage's time; that half of the test went to k_residue, which reads those values anyway.)
__global__ __launch_bounds__(256) void k_couple(PsyP P0, PsyP P1, CoupleSet CS, int blob_base, int nblobs, DescP d,
                                               const float *__restrict__ mdct, const ilog_t *__restrict__ ilogmask,
                                               int *__restrict__ iwork, int *__restrict__ nonzero, float band) {
  couple_unit<false>(P0, P1, CS, blob_base, nblobs, d, mdct, ilogmask, iwork, nonzero, band);
}
__global__ __launch_bounds__(64) void k_couple_norm(PsyP P0, PsyP P1, CoupleSet CS, int blob_base, int nblobs, DescP d,
                                               const float *__restrict__ mdct, const ilog_t *__restrict__ ilogmask,
                                               int *__restrict__ iwork, int *__restrict__ nonzero, float band) {
  couple_unit<true>(P0, P1, CS, blob_base, nblobs, d, mdct, ilogmask, iwork, nonzero, band);
}

// the same stage for layouts beyond stereo (more than two channels or more than one coupling step:
// couple_block_general, k_couple.h).  LDS: cand/key/sgn [n2] each + the partitions' budgets; the channels'
// running state lives in `state` [unit][4][ch][n2].
__global__ __launch_bounds__(64) void k_couple_general(PsyP P0, PsyP P1, CoupleSet CS, int blob_base, int nblobs, DescP d,
                                                       const float *__restrict__ mdct, const ilog_t *__restrict__ ilogmask,
                                                       int *__restrict__ iwork, int *__restrict__ nonzero,
                                                       float *__restrict__ state) {
  const long unit = blockIdx.x, mblk = unit / nblobs;
  const CoupleP &C = CS.c[blob_base + (int)(unit - mblk * nblobs)];
  const PsyP &P = d_bt(d, mblk) ? P1 : P0;
  const int n2 = P.n, ch = C.ch;
  CoupleLds L;
  L.cand = (float *)vamd_smem;
  L.key = L.cand + n2;
  L.sgn = L.key + n2;
  L.accp = L.sgn + n2;
  CoupleState S;
  S.re = state + unit * 4 * ch * n2;
  S.qe = S.re + ch * n2;
  S.fl2 = S.qe + ch * n2;
  S.fg = (int *)(S.fl2 + ch * n2);
  const float *mp[VAMD_MAX_CH];
  const ilog_t *ip[VAMD_MAX_CH];
  int *op[VAMD_MAX_CH];
  int nz[VAMD_MAX_CH];
  for (int c = 0; c < ch; c++) {
    mp[c] = mdct + (mblk * ch + c) * n2;
    ip[c] = ilogmask + (unit * ch + c) * n2;
    op[c] = iwork + (unit * ch + c) * n2;
    nz[c] = nonzero[unit * ch + c];
  }
  WAVE_SYNC_GLOBAL();  // every lane has read nonzero[] before lane 0 rewrites it
  PhaseClock pc;
  pc.start(nullptr);
  unsigned over = 0;
  couple_block_general(C, P, n2, mp, ip, op, nz, L, S, pc, &over);
  if (LANE == 0)
    for (int c = 0; c < ch; c++) nonzero[unit * ch + c] = nz[c];
  for (int c = 0; c < ch; c++)
    if (wave_any((int)((over >> c) & 1u)) && LANE == 0) flag_range(d, mblk * ch + c);
}

// stage 4 of a bitrate-managed batch: the same offset_and_mix, then three fits, twelve interpolated
// curves and fifteen rendered floors per channel (floor_managed_block, k_floor.h).  Outputs are laid
// out [block][candidate packet][channel][...].
__global__ __launch_bounds__(64) void k_floor_managed(PsyP P0, PsyP P1, FloorP F0, FloorP F1, ChMap cm, DescP d, int ch,
                                                      const float *__restrict__ noise, const float *__restrict__ tone,
                                                      const float *__restrict__ mdct_raw, float *__restrict__ mdct,
                                                      float *__restrict__ logmask_out, int *__restrict__ posts,
                                                      int *__restrict__ post_valid, ilog_t *__restrict__ ilogmask,
                                                      int *__restrict__ nonzero) {
  const long cb = blockIdx.x;
  const long blk = cb / ch;
  const int c = (int)(cb - blk * ch);
  const PsyP &P = d_bt(d, blk) ? P1 : P0;
  const FloorP &F = cm.sub[c] ? F1 : F0;
  const int n2 = P.n;
  unsigned short *qc = (unsigned short *)vamd_smem;
  FloorScratch *sc = (FloorScratch *)(qc + ((n2 + 15) & ~15));
  PhaseClock pc;
  pc.start(nullptr);
  offset_and_mix_wave(P, noise + cb * n2, tone + cb * n2, mdct_raw + cb * n2, mdct + cb * n2,
                      logmask_out ? logmask_out + cb * n2 : nullptr, qc, F.twofitatten, pc);
  const long u0 = blk * VAMD_PACKETBLOBS * ch + c;  // unit (blk, k = 0), channel c
  floor_managed_block(P, F, n2, noise + cb * n2, tone + cb * n2, mdct_raw + cb * n2, qc, sc,
                      posts + u0 * VAMD_POSTS_STRIDE, (long)ch * VAMD_POSTS_STRIDE, post_valid + u0, ch,
                      ilogmask + u0 * n2, (long)ch * n2, nonzero + u0, ch, pc);
}

// stage 6 (optional): residue classification + lattice-VQ search, one wave per unit and submap
// (k_residue.h).  Output rows of a unit: res_class [submaps][VAMD_RES_CLASS_STRIDE], res_entries [ent_row],
// res_count [submaps][2]; this launch fills submap `sm`'s part.
#define VAMD_RES_WAVES 4  // waves per unit: they share one LDS copy of the work vector
__global__ __launch_bounds__(64 * VAMD_RES_WAVES) void k_residue(ResP R, ChMap cm, int sm, int ent_row, int nblobs, DescP d, int ch, int n2,
                                                const int *__restrict__ iwork, const int *__restrict__ nonzero,
                                                int *__restrict__ res_class, unsigned short *__restrict__ res_entries,
                                                int *__restrict__ res_count, unsigned char *__restrict__ res_books) {
  const long u = blockIdx.x;
  int *work = (int *)vamd_smem;                 // [bundle*n2]
  int *cls = work + R.bundle * n2;              // [VAMD_RES_CLASS_STRIDE]
  int *off = cls + VAMD_RES_CLASS_STRIDE;       // [stages*slots + 1], then info [stages*slots]
  int *info = off + (R.tab->stages * R.slots + 1);
  const int *ip[VAMD_MAX_CH];
  int nz[VAMD_MAX_CH], chan[VAMD_MAX_CH];
  int nb = 0;
  for (int c = 0; c < ch; c++)
    if (cm.sub[c] == sm) {
      ip[nb] = iwork + (u * ch + c) * n2;
      nz[nb] = nonzero[u * ch + c];
      chan[nb] = c;
      nb++;
    }
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 72 : nullptr);
  unsigned over = 0;
  residue_block(R, n2, ip, nz, work, cls, off, info, res_class + u * (cm.submaps * VAMD_RES_CLASS_STRIDE) + R.cls_base,
                res_entries + u * (long)ent_row + R.ent_base, res_count + (u * cm.submaps + sm) * 2, pc,
                res_books ? res_books + u * (long)ent_row + R.ent_base : nullptr, &over);
  // the input domain's integer edge, the search's half (k_residue.h): the team's lanes each loaded their own values
  const long blk = u / nblobs;
  for (int k = 0; k < nb; k++) {
    const int any = __syncthreads_or((int)((over >> k) & 1u));
    if (TEAM_LEADER && any) flag_range(d, blk * ch + chan[k]);
  }
  pc.flush();
}

// stage 7 (optional): packet assembly, one wave per packet (k_pack.h).  unit = block * nblobs + candidate
__global__ __launch_bounds__(64) void k_pack(PackP K, FloorP F0, FloorP F1, ResP R0, ResP R1, ChMap cm, int ent_row, int lds_ints,
                                             DescP d, int ch, int W, int nblobs, const int *__restrict__ posts,
                                             const int *__restrict__ wrapped /* k_floor's, or null */,
                                             const int *__restrict__ post_valid, const int *__restrict__ res_class,
                                             const unsigned short *__restrict__ res_entries,
                                             const unsigned char *__restrict__ res_books,
                                             const int *__restrict__ res_count, unsigned *__restrict__ packets,
                                             int stride_words, int *__restrict__ packet_bits) {
  const long u = blockIdx.x, blk = u / nblobs;
  int *ring = (int *)vamd_smem;                  // [VAMD_PK_RING]
  int *outv = ring + VAMD_PK_RING;               // [VAMD_POSTS_STRIDE]
  int *cls = outv + VAMD_POSTS_STRIDE;           // [VAMD_RES_CLASS_STRIDE]
  int *off = cls + VAMD_RES_CLASS_STRIDE;        // [stages*slots + 1], then info [stages*slots], sized for the larger submap
  int *info = off + lds_ints;
  int *tabs = info + lds_ints;                   // [VAMD_PK_FTAB_INTS + 3 * nbooks], PackTabs
  PhaseClock pc;  // (marks 2..7 of the residue stage's slot set: k_residue uses 0 and 1)
  pc.start(d.dbg ? d.dbg + 72 : nullptr);
  pack_block(K, F0, F1, R0, R1, cm, ch, W, d_lW(d, blk), d_nW(d, blk), posts + u * ch * VAMD_POSTS_STRIDE,
             wrapped ? wrapped + u * ch * VAMD_POSTS_STRIDE : nullptr, post_valid + u * ch,
             res_class + u * (cm.submaps * VAMD_RES_CLASS_STRIDE), res_entries + u * (long)ent_row,
             res_books ? res_books + u * (long)ent_row : nullptr, res_count + u * cm.submaps * 2, ring, outv, cls, off, info, tabs, packets + u * (long)stride_words, stride_words,
             packet_bits + u, pc);
  pc.flush();
}

// The same for a handful of packets, TWO waves each: the header + floors part and the residue part of a packet are
// looked up and packed at the same time -- wave 1 writes the former where it belongs; wave 0 assembles the latter
// K.head_words into the row, past the longest head there can be, and when both are done the two waves move it down to
// where the head really ended (a shift by a whole number of words and `sh` bits, low words first: a destination never
// lies above its source).  A lone wave's packet is two strings of dependent lookups one after the other (k_pack: 10 +
// 11 us of a block's latency); here they overlap.  LDS: two rings, then as k_pack.
__global__ __launch_bounds__(128) void k_pack_pair(PackP K, FloorP F0, FloorP F1, ResP R0, ResP R1, ChMap cm, int ent_row, int lds_ints,
                                                   DescP d, int ch, int W, int nblobs, const int *__restrict__ posts,
                                                   const int *__restrict__ wrapped, const int *__restrict__ post_valid,
                                                   const int *__restrict__ res_class,
                                                   const unsigned short *__restrict__ res_entries,
                                                   const unsigned char *__restrict__ res_books,
                                                   const int *__restrict__ res_count, unsigned *__restrict__ packets,
                                                   int stride_words, int *__restrict__ packet_bits) {
  const long u = blockIdx.x, blk = u / nblobs;
  const int wave = threadIdx.x >> 6;
  int *ring = (int *)vamd_smem + wave * VAMD_PK_RING;  // [2][VAMD_PK_RING]
  int *outv = (int *)vamd_smem + 2 * VAMD_PK_RING;     // [VAMD_POSTS_STRIDE]   (wave 1)
  int *cls = outv + VAMD_POSTS_STRIDE;                 // [VAMD_RES_CLASS_STRIDE], off, info (wave 0)
  int *off = cls + VAMD_RES_CLASS_STRIDE;
  int *info = off + lds_ints;
  int *tabs = info + lds_ints;
  int *share = tabs + VAMD_PK_FTAB_INTS + 3 * K.nbooks;  // [4]: head bits, head's last (partial) word, residue bits
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 72 : nullptr);
  PackTabs T;
  T.at(tabs);
  for (int b = threadIdx.x; b < K.nbooks; b += blockDim.x) {  // (pack_book_table, both waves)
    const vamd_book_tab &bk = K.books[b];
    T.books[3 * b] = bk.entries;
    T.books[3 * b + 1] = (int)bk.off_lengths;
    T.books[3 * b + 2] = (int)bk.off_codes;
  }
  WAVE_FOR(i, VAMD_PK_RING) ring[i] = 0;
  __syncthreads();
  unsigned *row = packets + u * (long)stride_words;
  BitRing r;
  r.ring = ring;
  r.bitpos = 0;
  r.flushed = 0;
  if (wave == 1) {
    r.out = row;
    r.out_words = stride_words;
    {  // lib/mapping0.c:598-604, as pack_block
      const int lW = d_lW(d, blk), nW = d_nW(d, blk);
      unsigned hdr = (unsigned)W << 1;
      int len = 1 + K.modebits;
      if (W) {
        hdr |= (unsigned)(lW ? 1 : 0) << len;
        hdr |= (unsigned)(nW ? 1 : 0) << (len + 1);
        len += 2;
      }
      ring_put(r, hdr, LANE == 0 ? len : 0);
    }
    for (int c = 0; c < ch; c++) {
      const int sm = cm.sub[c];
      pack_floor(K, T, sm, sm ? F1 : F0, posts + (u * ch + c) * VAMD_POSTS_STRIDE,
                 wrapped ? wrapped + (u * ch + c) * VAMD_POSTS_STRIDE : nullptr, post_valid[u * ch + c], outv, r, pc);
    }
    ring_flush(r, r.bitpos >> 5);  // whole words out; the last, partial one goes to wave 0's first
    if (LANE == 0) {
      share[0] = (int)r.bitpos;
      share[1] = (r.bitpos & 31) ? ring[(r.bitpos >> 5) & (VAMD_PK_RING - 1)] : 0;
    }
  } else {
    r.out = row + K.head_words;
    r.out_words = stride_words - K.head_words;
    for (int sm = 0; sm < cm.submaps; sm++) {
      const ResP &R = sm ? R1 : R0;
      pack_residue(K, T, R, res_class + u * (cm.submaps * VAMD_RES_CLASS_STRIDE) + R.cls_base,
                   res_entries + u * (long)ent_row + R.ent_base, res_books ? res_books + u * (long)ent_row + R.ent_base : nullptr,
                   res_count + (u * cm.submaps + sm) * 2, cls, off, info, r, pc);
    }
    ring_flush(r, (r.bitpos + 31) >> 5);
    if (LANE == 0) share[2] = (int)r.bitpos;
  }
  __syncthreads();  // (workgroup scope: wave 0's words in the row are visible to wave 1's lanes and the other way round)
  const int headbits = share[0], resbits = share[2];
  const unsigned headword = (unsigned)share[1];
  const int total = headbits + resbits;
  const int w0 = headbits >> 5, wend = (total + 31) >> 5;
  const int delta = K.head_words * 32 - headbits;  // > 0: bits the residue part moves down by
  const int dw = delta >> 5, sh = delta & 31;
  const int src_end = K.head_words + ((resbits + 31) >> 5);  // the residue part's words are row[head_words, src_end)
  for (int base = w0; base < src_end; base += (int)blockDim.x) {
    const int w = base + (int)threadIdx.x;
    unsigned val = 0;
    if (w < wend) {
      const int s0 = w + dw;
      const unsigned lo = s0 >= K.head_words && s0 < src_end && s0 < stride_words ? row[s0] : 0u;
      const unsigned hi = s0 + 1 >= K.head_words && s0 + 1 < src_end && s0 + 1 < stride_words ? row[s0 + 1] : 0u;
      val = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
      if (w == w0) val |= headword;
    }
    __syncthreads();  // every source of this trip is read before any of its destinations is written
    if (w < src_end && w < stride_words) row[w] = val;  // (past wend: what the move left behind, zeroed)
    __syncthreads();
  }
  if (threadIdx.x == 0) packet_bits[u] = total;
  pc.mark(6);
  pc.flush();
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------

// ---- the block-switching detector (k_envelope.h).  Series layouts, s = stream, c = channel:
//   near [s][c][VAMD_VE_NEAR_HIST + nsteps]        near-DC terms behind their history prefix
//   raw  [s][c][nsteps][32]                        unlimited dB pairs
//   amp  [s][c][VAMD_VE_AMP_HIST + nsteps][8]      band amplitudes behind their history prefix
//   bits [s][nsteps]                               trigger bits for the 13 values of stretch/2
__global__ void k_env_prolog(int ch, long nstreams, long nsteps, const vamd_envelope_state *__restrict__ st,
                             float *__restrict__ near, float *__restrict__ amp) {
  const long per = (long)ch * (VAMD_VE_NEAR_HIST + VAMD_VE_AMP_HIST * 8);
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nstreams * per) return;
  const long s = t / per;
  long r = t - s * per;
  const int c = (int)(r / (VAMD_VE_NEAR_HIST + VAMD_VE_AMP_HIST * 8));
  r -= (long)c * (VAMD_VE_NEAR_HIST + VAMD_VE_AMP_HIST * 8);
  if (r < VAMD_VE_NEAR_HIST)
    near[(s * ch + c) * (VAMD_VE_NEAR_HIST + nsteps) + r] = st[s].near_hist[c][r];
  else {
    r -= VAMD_VE_NEAR_HIST;
    amp[(s * ch + c) * (VAMD_VE_AMP_HIST + nsteps) * 8 + r] = st[s].amp_hist[c][r >> 3][r & 7];
  }
}

// a wave takes VAMD_ENV_STEPS consecutive steps of one (stream, channel) at a time
#define VAMD_ENV_LOGS 2
#define VAMD_ENV_STEPS (1 << VAMD_ENV_LOGS)
#define VAMD_ENV_WAVES 4
__global__ __launch_bounds__(64 * VAMD_ENV_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_env_spectrum(EnvP E, int ch, long nstreams, long nsteps,
                                                                      const float *__restrict__ pcm, long stream_stride,
                                                                      long channel_stride, float *__restrict__ near,
                                                                      float *__restrict__ raw, unsigned int *bad) {
  const int n = E.mdct.n, n2 = n >> 1, wave = threadIdx.x >> 6;
  const int per_step = n + n2 + VAMD_PW_SIZE(n2) + n2;
  float *A = (float *)vamd_smem + (size_t)wave * per_step * VAMD_ENV_STEPS;
  float *Wk = A + n * VAMD_ENV_STEPS, *spec = Wk + (n2 + VAMD_PW_SIZE(n2)) * VAMD_ENV_STEPS;
  PhaseClock pc;
  pc.start(nullptr);
  const long groups = (nsteps + VAMD_ENV_STEPS - 1) / VAMD_ENV_STEPS, items = nstreams * ch * groups;
  // an item's samples are requested while the previous item is in its transform (a wave lives for ~130 items and has
  // three neighbours on its SIMD: the trip to memory at the head of every item was a fifth of its time)
  auto where = [&](long it, long &sc, long &j, int &count) -> const float * {
    sc = it / groups;
    j = (it - sc * groups) * VAMD_ENV_STEPS;
    const long s = sc / ch;
    const int c = (int)(sc - s * ch);
    count = nsteps - j < VAMD_ENV_STEPS ? (int)(nsteps - j) : VAMD_ENV_STEPS;
    return pcm + s * stream_stride + c * channel_stride + j * E.searchstep;
  };
  const long stride = (long)gridDim.x * VAMD_ENV_WAVES;
  long it = (long)blockIdx.x * VAMD_ENV_WAVES + wave;
  EnvSamples<VAMD_ENV_LOGS> cur, nxt;
  long sc, j;
  int count;
  const float *src = nullptr;
  if (it < items) {
    src = where(it, sc, j, count);
    env_fetch<VAMD_ENV_LOGS>(cur, src, count, E.searchstep);
  }
  for (; it < items; it += stride) {
    long sc2 = 0, j2 = 0;
    int count2 = 0;
    const float *src2 = nullptr;
    if (it + stride < items) {
      src2 = where(it + stride, sc2, j2, count2);
      env_fetch<VAMD_ENV_LOGS>(nxt, src2, count2, E.searchstep);
    }
    env_spectrum_wave<VAMD_ENV_LOGS>(E, src, count, A, Wk, spec, near + sc * (VAMD_VE_NEAR_HIST + nsteps) + VAMD_VE_NEAR_HIST + j,
                                     raw + (sc * nsteps + j) * VAMD_VE_SPREAD, pc, bad, &cur);
    cur = nxt, sc = sc2, j = j2, count = count2, src = src2;
  }
}

__global__ void k_env_amp(EnvP E, long nsc /* streams x channels */, long nsteps,
                                  const vamd_envelope_state *__restrict__ st, int ch, const float *__restrict__ near, const float *__restrict__ raw,
                          float *__restrict__ amp) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = (int)(t & 7);
  const long it = t >> 3;
  if (it >= nsc * nsteps) return;
  const long sc = it / nsteps, j = it - sc * nsteps;
  if (b >= VAMD_VE_BANDS) {  // the pad lane of the 8-float rows: keep the state deterministic
    amp[(sc * (VAMD_VE_AMP_HIST + nsteps) + VAMD_VE_AMP_HIST + j) * 8 + b] = 0.f;
    return;
  }
  const float decay = env_decay(near + sc * (VAMD_VE_NEAR_HIST + nsteps) + VAMD_VE_NEAR_HIST + j, (long)st[sc / ch].steps + j);
  amp[(sc * (VAMD_VE_AMP_HIST + nsteps) + VAMD_VE_AMP_HIST + j) * 8 + b] = env_band_amp(E, raw + it * VAMD_VE_SPREAD, decay, b);
}

// The same for big batches, a tile of VAMD_ENV_TJ steps of one (stream, channel) per workgroup: the tile's near-DC terms
// (with the 29 before it that the replay reaches back to) and its raw dB pairs are staged in LDS once, the decay of a
// step is replayed once (a lane per step) instead of once per band, and the band lanes read both out of LDS.  (The
// thread-per-band form above has every step's eight lanes fetch the same 44 terms and replay the same 44 adds, and
// its 32 raw values come seven overlapping times out of L2: 0.71 ms for 4 M channel-steps, twice its issue time.)
#define VAMD_ENV_TJ 32
#define VAMD_ENV_BACK (2 * VAMD_VE_NEARDC - 1)
__global__ __launch_bounds__(8 * VAMD_ENV_TJ) void k_env_amp_tiled(EnvP E, long nsc, long nsteps,
                                                                   const vamd_envelope_state *__restrict__ st, int ch,
                                                                   const float *__restrict__ near, const float *__restrict__ raw,
                                                                   float *__restrict__ amp) {
  __shared__ float s_near[VAMD_ENV_TJ + VAMD_ENV_BACK + 3];
  __shared__ float s_decay[VAMD_ENV_TJ];
  __shared__ __attribute__((aligned(16))) float s_raw[VAMD_ENV_TJ * VAMD_VE_SPREAD];
  const long tiles = (nsteps + VAMD_ENV_TJ - 1) / VAMD_ENV_TJ;
  const long sc = blockIdx.x / tiles, j0 = (blockIdx.x - sc * tiles) * VAMD_ENV_TJ;
  const int cnt = nsteps - j0 < VAMD_ENV_TJ ? (int)(nsteps - j0) : VAMD_ENV_TJ;
  const float *nearp = near + sc * (VAMD_VE_NEAR_HIST + nsteps) + VAMD_VE_NEAR_HIST + j0;  // this tile's first term
  for (int i = threadIdx.x; i < cnt + VAMD_ENV_BACK; i += blockDim.x) s_near[i] = nearp[i - VAMD_ENV_BACK];
  {
    const F4 *src = (const F4 *)(raw + (sc * nsteps + j0) * VAMD_VE_SPREAD);
    for (int i = threadIdx.x; i < cnt * (VAMD_VE_SPREAD / 4); i += blockDim.x) ((F4 *)s_raw)[i] = src[i];
  }
  __syncthreads();
  if ((int)threadIdx.x < cnt) s_decay[threadIdx.x] = env_decay(s_near + VAMD_ENV_BACK + threadIdx.x, (long)st[sc / ch].steps + j0 + threadIdx.x);
  __syncthreads();
  const int jj = threadIdx.x >> 3, b = threadIdx.x & 7;
  if (jj >= cnt) return;
  float *out = amp + (sc * (VAMD_VE_AMP_HIST + nsteps) + VAMD_VE_AMP_HIST + j0 + jj) * 8 + b;
  *out = b >= VAMD_VE_BANDS ? 0.f : env_band_amp(E, s_raw + jj * VAMD_VE_SPREAD, s_decay[jj], b);
}

// sixteen lanes per (stream, step): the (channel, band) pairs are dealt round them and their trigger bits OR-ed
// together (a thread per step walked 14 pairs x 12 dependent loads: 47 us for the sixteen steps of one blockout call)
__global__ void k_env_bits(EnvP E, int ch, long nstreams, long nsteps, const float *__restrict__ amp,
                           uint32_t *__restrict__ bits) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long t = gid >> 4;
  const int sub = (int)(gid & 15);
  const bool live = t < nstreams * nsteps;
  uint32_t my = 0;
  if (live) {
    const long s = t / nsteps, j = t - s * nsteps;
    for (int cb = sub; cb < ch * VAMD_VE_BANDS; cb += 16) {
      const int c = cb / VAMD_VE_BANDS, b = cb - c * VAMD_VE_BANDS;
      my |= env_trigger_bits_one(E, amp + ((s * ch + c) * (VAMD_VE_AMP_HIST + nsteps) + VAMD_VE_AMP_HIST + j) * 8 + b, 8, b);
    }
  }
  my |= (uint32_t)__shfl_xor((int)my, 1, 64);
  my |= (uint32_t)__shfl_xor((int)my, 2, 64);
  my |= (uint32_t)__shfl_xor((int)my, 4, 64);
  my |= (uint32_t)__shfl_xor((int)my, 8, 64);
  if (live && sub == 0) bits[t] = my;
}

// ... and a thread per (stream, step) for big batches, where threads are plentiful and sixteen of them fetching the
// same histories only multiply the loads (0.86 against 0.70 ms for 2 M steps)
__global__ void k_env_bits_batch(EnvP E, int ch, long nstreams, long nsteps, const float *__restrict__ amp,
                                 uint32_t *__restrict__ bits) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nstreams * nsteps) return;
  const long s = t / nsteps, j = t - s * nsteps;
  const float *a[VAMD_MAX_CH];
  for (int c = 0; c < ch; c++) a[c] = amp + ((s * ch + c) * (VAMD_VE_AMP_HIST + nsteps) + VAMD_VE_AMP_HIST + j) * 8;
  bits[t] = env_trigger_bits(E, a, ch, 8);
}

// ... and with the amplitudes staged: a wave takes 64 consecutive steps of one stream, copies the 64 + 13 rows of every
// channel they and their histories cover into LDS once (rows padded to nine floats: lanes a step apart read a row apart),
// and every lane forms its step's bits out of them.  (A thread per step fetched its 14 (channel, band) histories -- 196
// words, 13 of every 14 of them its neighbour's too -- out of L2: 0.67 ms for 2 M steps, five times its issue time.)
#define VAMD_ENV_BROWS (64 + VAMD_VE_MAXSTRETCH + 1)
__global__ __launch_bounds__(256) void k_env_bits_tiled(EnvP E, int ch, long nstreams, long nsteps, const float *__restrict__ amp,
                                                        uint32_t *__restrict__ bits) {
  float *tile = (float *)vamd_smem + (size_t)(threadIdx.x >> 6) * ch * VAMD_ENV_BROWS * 9;  // [ch][BROWS][9]
  const long tiles = (nsteps + 63) / 64;
  const long item = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (item >= nstreams * tiles) return;  // (waves are independent: no workgroup barrier below)
  const long s = item / tiles, j0 = (item - s * tiles) * 64;
  const int cnt = nsteps - j0 < 64 ? (int)(nsteps - j0) : 64;
  const int back = VAMD_VE_MAXSTRETCH + 1;  // rows a step reaches back to: 13 <= VAMD_VE_AMP_HIST
  for (int c = 0; c < ch; c++) {
    const F4 *src = (const F4 *)(amp + ((s * ch + c) * (VAMD_VE_AMP_HIST + nsteps) + VAMD_VE_AMP_HIST + j0 - back) * 8);
    float *dst = tile + (size_t)c * VAMD_ENV_BROWS * 9;
    for (int i = LANE; i < (cnt + back) * 2; i += 64) {
      const F4 v = src[i];
      float *d = dst + (i >> 1) * 9 + (i & 1) * 4;
      d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    }
  }
  WAVE_SYNC();
  if (LANE < cnt) {
    uint32_t my = 0;
    for (int c = 0; c < ch; c++)
      for (int b = 0; b < VAMD_VE_BANDS; b++)
        my |= env_trigger_bits_one(E, tile + (size_t)c * VAMD_ENV_BROWS * 9 + (back + LANE) * 9 + b, 9, b);
    bits[s * nsteps + j0 + LANE] = my;
  }
}

// the stretch recurrence, one wave per stream; then the state's histories roll forward
__global__ __launch_bounds__(64) void k_env_walk(int ch, long nstreams, long nsteps, const uint32_t *__restrict__ bits,
                                                 const float *__restrict__ near, const float *__restrict__ amp,
                                                 vamd_envelope_state *__restrict__ st,
                                                 unsigned char *__restrict__ ret) {
  const long s = blockIdx.x;
  const int stretch = env_walk_wave(bits + s * nsteps, nsteps, st[s].stretch, ret + s * nsteps);
  if (LANE == 0) {
    st[s].stretch = stretch;
    st[s].steps += nsteps;
  }
  for (int c = 0; c < ch; c++) {
    const float *nt = near + (s * ch + c) * (VAMD_VE_NEAR_HIST + nsteps) + nsteps;  // the last NEAR_HIST entries
    WAVE_FOR(i, VAMD_VE_NEAR_HIST) st[s].near_hist[c][i] = nt[i];
    const float *at = amp + ((s * ch + c) * (VAMD_VE_AMP_HIST + nsteps) + nsteps) * 8;
    WAVE_FOR(i, VAMD_VE_AMP_HIST * 8) st[s].amp_hist[c][i >> 3][i & 7] = at[i];
  }
}

// ---- device-resident stream control (k_blockout.h) ----------------------------------------------------
// one wave per stream: the lanes turn the stream's flags into its mark bytes in LDS (coalesced reads, ve->mark[] as
// mark_at defines it), then one lane does the walk out of LDS -- a dependent chain of a few thousand steps that would
// otherwise pay a trip to HBM at each of them
__global__ __launch_bounds__(64) void k_plan_streams(BlockoutP B, long nstreams, const unsigned char *__restrict__ flags,
                                                     PlannedBlock *__restrict__ blocks, int *__restrict__ counts) {
  unsigned char *marks = (unsigned char *)vamd_smem;  // [nsteps + 4]
  const long s = blockIdx.x;
  const long last = blockout_steps(B);
  const unsigned char *f = flags + s * B.nsteps;
  for (long p = threadIdx.x; p < B.nsteps + 4; p += 64) marks[p] = p < last ? (unsigned char)mark_at(f, last, p) : 0;
  __syncthreads();
  int n0 = 0, n1 = 0;
  plan_stream(B, marks, blocks + s * B.maxblocks, &n0, &n1);  // (the whole wave: it looks at 64 marks at a time)
  if (threadIdx.x == 0) {
    counts[2 * s] = n0;
    counts[2 * s + 1] = n1;
  }
}

// base[2s + W] = index of stream s's first block inside size class W's batch; start[s] = into order[]
struct PlanOut {
  int *lW[2], *nW[2], *bt[2];
  long long *src[2];
  int *order;
};
// a wave per stream: lane l takes the stream's blocks l, l + 64, ...; a block's place in its size class's batch is the
// class's base plus the blocks of that class ahead of it in the stream -- a count over the lower lanes' ballot bits
// (a thread per stream walking its ~140 blocks took 0.16 ms for a thousand streams: sixteen waves on the whole chip)
__global__ __launch_bounds__(64) void k_plan_emit(BlockoutP B, long nstreams, long stream_stride, const PlannedBlock *__restrict__ blocks,
                                                  const int *__restrict__ counts, const long long *__restrict__ base,
                                                  const long long *__restrict__ start, PlanOut O) {
  const long s = blockIdx.x;
  if (s >= nstreams) return;
  const int n = counts[2 * s] + counts[2 * s + 1];
  long long at[2] = {base[2 * s], base[2 * s + 1]};
  const unsigned long long below = (1ull << LANE) - 1ull;
  for (int k0 = 0; k0 < n; k0 += 64) {
    const int k = k0 + LANE;
    const bool live = k < n;
    PlannedBlock b;
    b.kind = 0, b.begin = 0;
    if (live) b = blocks[s * B.maxblocks + k];
    const int W = b.kind & 1;
    const unsigned long long is_long = __ballot(live && W), is_short = __ballot(live && !W);
    if (live) {
      const long long i = at[W] + __builtin_popcountll((W ? is_long : is_short) & below);
      O.lW[W][i] = (b.kind >> 1) & 1;
      O.nW[W][i] = (b.kind >> 2) & 1;
      O.bt[W][i] = (b.kind >> 3) & 1;
      O.src[W][i] = (long long)s * stream_stride + b.begin;
      O.order[start[s] + k] = (W << 30) | (int)i;
    }
    at[0] += __builtin_popcountll(is_short);
    at[1] += __builtin_popcountll(is_long);
  }
}

// out[b][c][0 .. n) = pcm[src[b] + c*channel_stride ..): one 16-byte piece per thread
__global__ void k_gather_blocks(int ch, int n, long nb, const long long *__restrict__ src, long channel_stride,
                                const float *__restrict__ pcm, float *__restrict__ out) {
  const long nq = n >> 2, total = nb * ch * nq;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long b = t / (ch * nq), r = t - b * ch * nq;
    const int c = (int)(r / nq);
    const long q = r - c * nq;
    ((F4 *)out)[t] = ((const F4 *)(pcm + src[b] + (long)c * channel_stride))[q];
  }
}

// calibration copy for counter passes (vamd_calib_copy): exactly 16 bytes in and 16 bytes out per lane-trip
__global__ __launch_bounds__(256) void k_calib_copy(const F4 *__restrict__ src, F4 *__restrict__ dst, long n16) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
};

struct vamd_ctx {
  int device = 0;
  int num_cus = 256;
  size_t lds_per_block = 160 * 1024;
  hipStream_t stream = nullptr;
  // noise masking and tone masking read different inputs and write different outputs; the tone
  // kernels run on this library-owned side stream, forked from / joined back into `stream`
  hipStream_t side = nullptr;
  unsigned long long *d_clk = nullptr;  // vamd_clock_probe's accumulator (the caller's, device memory), or null
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;  // (ev_join2: the short size class of a mixed run)
  bool overlap = true;
  float couple_band = VAMD_COUPLE_BAND;  // k_couple.h, chan_bin_sure
  Bound B;                 // parameter structs bound to the HBM image
  unsigned char *d_image = nullptr;
  Bound *d_bound = nullptr;  // c->B in HBM: kernels that would otherwise carry several parameter structs in SGPRs read it
  unsigned int *d_bad = nullptr;  // [0] channel-blocks, [1] detector steps outside the input domain since vamd_input_status(), [2] the non-finite ones among [0] (behind d_bound)
  size_t image_bytes = 0;
  std::string err;
  Knobs K;           // the environment knobs, read once at vamd_create (vamd_knobs.h)
  char config[1024]; // ... and as text (vamd_config_string)
  // workspace, grown on demand (vamd_reserve to pre-size)
  enum { WS_MDCT_RAW, WS_LOGMDCT, WS_LOGFFT, WS_NOISE, WS_TONE, WS_MDCT, WS_ILOGMASK, WS_IWORK, WS_POSTS, WS_POSTVALID,
         WS_NONZERO, WS_LOCAL, WS_AMPIN, WS_AMPGLOB, WS_PCM, WS_SEED, WS_SURV, WS_NSURV, WS_MISC,
         WS_ENV_NEAR, WS_ENV_RAW, WS_ENV_AMP, WS_ENV_BITS, WS_ENV_STAGE, WS_M_ILOGMASK, WS_M_STAGE,
         WS_RES_CLASS, WS_RES_ENTRIES, WS_RES_COUNT, WS_COUPLE_STATE,
         WS_PLAN_FLAGS, WS_PLAN_BLOCKS, WS_PLAN_COUNTS, WS_PLAN_BASE, WS_PLAN_DESC, WS_PLAN_ORDER, WS_STATUS, WS_WRAPPED, WS_RES_BOOKS, WS_COUNT };
  DevBuf ws[2][WS_COUNT];  // per size class (a mixed stream keeps both batches in flight)
  // pinned staging for the per-block host API
  void *h_stage = nullptr;
  size_t h_stage_bytes = 0;
  void *h_plan = nullptr;  // pinned: a stream plan's per-stream bases on their way up (vamd_plan_streams)
  size_t h_plan_bytes = 0;
  // optional per-stage timing (vamd_profile): one event before each stage + one after the last
  unsigned long long *d_dbg = nullptr;  // 80 phase-stopwatch slots when armed
  bool profile = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<int> ev_stage;  // per recorded event: the stage whose interval it closes (VAMD_ST_BEGIN = none)
  int prof_runs = 0;          // batches recorded since the last vamd_stage_ms()
};

// stage ids of vamd_stage_ms(); a mark closes the interval of the stage it names (VAMD_ST_BEGIN: opens one)
enum { VAMD_ST_BEGIN = -1, VAMD_ST_TRANSFORM = 0, VAMD_ST_AMPMAX, VAMD_ST_NOISE, VAMD_ST_TONE, VAMD_ST_FLOOR, VAMD_ST_COUPLE,
       VAMD_ST_RESIDUE, VAMD_ST_PACK, VAMD_ST_COUNT };
static void prof_mark(vamd_ctx *c, int stage) {
  if (!c->profile) return;
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    c->ev_pool.push_back(e);
  }
  (void)hipEventRecord(c->ev_pool[c->ev_used], c->stream);
  if (c->ev_stage.size() <= c->ev_used) c->ev_stage.resize(c->ev_used + 1);
  c->ev_stage[c->ev_used++] = stage;
}

// waves per persistent transform workgroup: as many as fit beside the staged tables
static int xf_waves(const vamd_ctx *c, const XformP &P) {
  const int cap = c->K.xf_waves_cap;  // (measurement aid, a test knob: vamd_knobs.h)
  int w = cap > 0 && cap < VAMD_XF_WAVES ? cap : VAMD_XF_WAVES;
  while (w > 1 && transform_lds_bytes(P, w) > c->lds_per_block) w--;
  return w;
}

static int fail(vamd_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return code;
}

// what a host-pointer call makes of its block's status bytes (include/vorbis_amd.h, "Input domain")
static int status_verdict(vamd_ctx *c, const unsigned char *st, size_t ch) {
  unsigned any = 0;
  for (size_t i = 0; i < ch; i++) any |= st[i];
  if (any & VAMD_STATUS_NONFINITE)
    return fail(c, VAMD_ENONFINITE, "input outside the domain: a NaN / Inf sample (or finite ones beyond ~3e16 x full scale, where the fp32 spectrum overflows)");
  if (any & VAMD_STATUS_RANGE)
    return fail(c, VAMD_EDOMAIN, "input outside the domain: a quantised value beyond the bound up to which the reference's integer arithmetic is defined (vamd_quant_limit)");
  return VAMD_OK;
}

#define HIP_TRY(c, expr)                                              \
  do {                                                                \
    hipError_t e__ = (expr);                                          \
    if (e__ != hipSuccess) return fail((c), VAMD_EFAULT, #expr, e__); \
  } while (0)


// A context is bound to ONE device (vamd_create).  Every public entry point runs with that device current --
// workspace allocations, pinned staging, launches and the side stream all belong to it -- and puts the caller's
// device back on the way out, so a context on GPU 1 works while the caller (or torch) sits on GPU 0.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(const vamd_ctx *c) {
    if (!c) return;
    if (hipGetDevice(&prev) == hipSuccess && prev != c->device) switched = hipSetDevice(c->device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

static int ws_get(vamd_ctx *c, int W, int which, size_t bytes, void **out) {
  DevBuf &b = c->ws[W][which];
  if (b.bytes < bytes) {
    if (b.p) HIP_TRY(c, hipFree(b.p));
    b.p = nullptr;
    b.bytes = 0;
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
  }
  *out = b.p;
  return VAMD_OK;
}

extern "C" {

const char *vamd_config_string(const vamd_ctx *c) { return c ? c->config : ""; }

int vamd_create_abi(vamd_ctx **out, const void *setup_blob, size_t blob_bytes, int device, int caller_abi_version) {
  if (!out) return VAMD_EINVAL;
  *out = nullptr;
  if (caller_abi_version != VAMD_ABI_VERSION) {
    fprintf(stderr, "vamd_create: the caller was built against ABI %d of include/vorbis_amd.h, this library is ABI %d\n",
            caller_abi_version, VAMD_ABI_VERSION);
    return VAMD_EVERSION;
  }
  vamd_ctx *c = new (std::nothrow) vamd_ctx;
  if (!c) return VAMD_EFAULT;
  c->K = read_knobs();
  knobs_string(c->K, c->config, sizeof(c->config));
  std::vector<unsigned char> image;
  std::vector<uint32_t> doff;
  std::vector<PsyDerived> derived;
  int r = build_image(setup_blob, blob_bytes, &image, &doff, &derived, &c->err);
  if (r != VAMD_OK) {
    fprintf(stderr, "vamd_create: %s\n", c->err.c_str());
    delete c;
    return r;
  }
  hipError_t e = hipSuccess;
  int caller_device = -1;
  (void)hipGetDevice(&caller_device);  // put back before returning: creating a context must not move the caller
  if (device >= 0) e = hipSetDevice(device);
  if (e == hipSuccess) e = hipGetDevice(&c->device);
  if (e == hipSuccess) {
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, c->device);
    if (e == hipSuccess) {
      c->num_cus = prop.multiProcessorCount;
      c->lds_per_block = prop.sharedMemPerBlock;
      // opt in to the full LDS for the persistent transform kernels (a no-op where the
      // runtime does not require it)
#define VAMD_OPT_IN(LOGN)                                                                                      \
  (void)hipFuncSetAttribute((const void *)k_transform<LOGN>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                            (int)c->lds_per_block);                                                            \
  (void)hipFuncSetAttribute((const void *)k_mdct_only<LOGN>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                            (int)c->lds_per_block);
      VAMD_OPT_IN(0) VAMD_OPT_IN(8) VAMD_OPT_IN(9) VAMD_OPT_IN(10) VAMD_OPT_IN(11) VAMD_OPT_IN(12)
#undef VAMD_OPT_IN
      (void)hipGetLastError();
      if (c->K.verbose)
        fprintf(stderr, "vamd_create: %d CUs, %zu B LDS per workgroup\n", c->num_cus, c->lds_per_block);
    }
  }
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming);
  c->overlap = !c->K.no_overlap;
  // (test aid: k_couple's estimate-then-verify margin as a power of two; 1 sends every quad through the exact path)
  c->couple_band = c->K.couple_band_set ? ldexpf(1.f, c->K.couple_band_log2) : VAMD_COUPLE_BAND;
  if (e == hipSuccess) e = hipMalloc((void **)&c->d_image, image.size());
  if (e == hipSuccess) e = hipMemcpy(c->d_image, image.data(), image.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    fprintf(stderr, "vamd_create: HIP failure: %s\n", hipGetErrorString(e));
    if (c->d_image) (void)hipFree(c->d_image);
  if (c->d_bound) (void)hipFree(c->d_bound);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    delete c;
    return VAMD_EFAULT;
  }
  c->image_bytes = image.size();
  bind_params(image, doff, derived, c->d_image, &c->B);
  const size_t bound_bytes = (sizeof(Bound) + 15) & ~(size_t)15;
  if (hipMalloc((void **)&c->d_bound, bound_bytes + 16) != hipSuccess ||
      hipMemset(c->d_bound, 0, bound_bytes + 16) != hipSuccess ||
      hipMemcpy(c->d_bound, &c->B, sizeof(Bound), hipMemcpyHostToDevice) != hipSuccess) {
    fprintf(stderr, "vamd_create: HIP failure uploading the parameter block\n");
    if (c->d_bound) (void)hipFree(c->d_bound);
    (void)hipFree(c->d_image);
    (void)hipEventDestroy(c->ev_fork);
    (void)hipEventDestroy(c->ev_join);
    (void)hipEventDestroy(c->ev_join2);
    (void)hipStreamDestroy(c->side);
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    delete c;
    return VAMD_EFAULT;
  }
  c->d_bad = (unsigned int *)((unsigned char *)c->d_bound + bound_bytes);
  if (caller_device >= 0 && caller_device != c->device) (void)hipSetDevice(caller_device);
  *out = c;
  return VAMD_OK;
}

void vamd_destroy(vamd_ctx *c) {
  DeviceGuard dev_guard(c);
  if (!c) return;
  for (int W = 0; W < 2; W++)
    for (int i = 0; i < vamd_ctx::WS_COUNT; i++)
      if (c->ws[W][i].p) (void)hipFree(c->ws[W][i].p);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->h_plan) (void)hipHostFree(c->h_plan);
  if (c->d_dbg) (void)hipFree(c->d_dbg);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_join2) (void)hipEventDestroy(c->ev_join2);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->d_image) (void)hipFree(c->d_image);
  if (c->d_bound) (void)hipFree(c->d_bound);
  delete c;
}

int vamd_abi_version(void) { return VAMD_ABI_VERSION; }

const char *vamd_last_error(const vamd_ctx *c) { return c ? c->err.c_str() : "null context"; }

int vamd_set_stream(vamd_ctx *c, void *s) {
  if (!c) return VAMD_EINVAL;
  c->stream = (hipStream_t)s;
  return VAMD_OK;
}

// Did any block (or detector step) issued on this context since the last call fall outside the input domain?
int vamd_input_status(vamd_ctx *c, long *bad_channel_blocks, long *bad_detector_steps) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  unsigned int h[3] = {0, 0, 0};
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(h, c->d_bad, sizeof(h), hipMemcpyDeviceToHost));
  if (h[0] | h[1] | h[2]) HIP_TRY(c, hipMemset(c->d_bad, 0, sizeof(h)));
  if (bad_channel_blocks) *bad_channel_blocks = (long)h[0];
  if (bad_detector_steps) *bad_detector_steps = (long)h[1];
  if (h[1] | h[2]) return fail(c, VAMD_ENONFINITE, "input outside the domain: NaN / Inf samples");
  if (h[0]) return fail(c, VAMD_EDOMAIN, "input outside the domain: quantised values beyond the bound up to which the reference's integer arithmetic is defined (vamd_quant_limit)");
  return VAMD_OK;
}

int vamd_quant_limit(const vamd_ctx *c, int W, int channel, int *first_bin, int *end_bin, int *square_bin) {
  if (!c || (W != 0 && W != 1) || channel < 0 || channel >= c->B.channels) return VAMD_EINVAL;
  if (first_bin) *first_bin = c->B.qlimit[W].lo[channel];
  if (end_bin) *end_bin = c->B.qlimit[W].hi[channel];
  if (square_bin) *square_bin = c->B.qlimit[W].sq;
  return c->B.qlimit[W].q[channel];
}

int vamd_profile(vamd_ctx *c, int enable) {
  if (!c) return VAMD_EINVAL;
  c->profile = enable != 0;
  c->ev_used = 0;
  c->prof_runs = 0;
  return VAMD_OK;
}

int vamd_stage_ms(vamd_ctx *c, float *ms, int nstages, int *runs) {
  DeviceGuard dev_guard(c);
  if (!c || !ms || nstages < 1) return VAMD_EINVAL;
  for (int i = 0; i < nstages; i++) ms[i] = 0.f;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (size_t i = 1; i < c->ev_used; i++) {
    const int st = c->ev_stage[i];
    if (st < 0 || st >= nstages) continue;
    float t = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&t, c->ev_pool[i - 1], c->ev_pool[i]));
    ms[st] += t;
  }
  if (runs) *runs = c->prof_runs;
  c->ev_used = 0;
  c->prof_runs = 0;
  return VAMD_OK;
}

int vamd_calib_copy(vamd_ctx *c, void *dst, const void *src, size_t bytes) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (!dst || !src || (bytes & 15) || (((uintptr_t)dst | (uintptr_t)src) & 15)) return fail(c, VAMD_EINVAL, "calibration copy: 16-byte aligned buffers and size");
  if (bytes == 0) return VAMD_OK;
  hipLaunchKernelGGL(k_calib_copy, dim3((unsigned)(c->num_cus * 16)), dim3(256), 0, c->stream, (const F4 *)src, (F4 *)dst, (long)(bytes / 16));
  HIP_TRY(c, hipGetLastError());
  return VAMD_OK;
}

int vamd_clock_probe(vamd_ctx *c, unsigned long long *acc3) {
  if (!c) return VAMD_EINVAL;
  c->d_clk = acc3;
  return VAMD_OK;
}

int vamd_debug_cycles(vamd_ctx *c, int enable, unsigned long long *out80) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (out80 && c->d_dbg) {
    std::vector<unsigned long long> all(64 * 80);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(all.data(), c->d_dbg, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < 80; k++) {
      out80[k] = 0;
      for (int r = 0; r < 64; r++) out80[k] += all[(size_t)r * 80 + k];
    }
  }
  if (enable) {
    if (!c->d_dbg) HIP_TRY(c, hipMalloc((void **)&c->d_dbg, 64 * 80 * sizeof(unsigned long long)));
    HIP_TRY(c, hipMemset(c->d_dbg, 0, 64 * 80 * sizeof(unsigned long long)));
  } else if (c->d_dbg) {
    HIP_TRY(c, hipFree(c->d_dbg));
    c->d_dbg = nullptr;
  }
  return VAMD_OK;
}

int vamd_channels(const vamd_ctx *c) { return c ? c->B.channels : VAMD_EINVAL; }
int vamd_blocksize(const vamd_ctx *c, int W) { return (c && (W == 0 || W == 1)) ? c->B.bs[W] : VAMD_EINVAL; }
int vamd_posts(const vamd_ctx *c, int W) { return (c && (W == 0 || W == 1)) ? c->B.floor[W][0].posts : VAMD_EINVAL; }

struct WsPlan {
  float *mdct_raw, *logmdct, *logfft, *noise, *tone, *mdct, *local, *ampin, *ampglob, *seed;
  float *peaks;  // [channel-blocks][run_peaks_stride]: logfft's peak per run of bins, or null below the psy level
  unsigned short *surv;
  int32_t *nsurv;
  ilog_t *ilogmask;  // a byte per bin, workspace only (the int32 tap is widened from it: k_widen_ilog)
  int32_t *iwork, *posts, *post_valid, *nonzero;
  int32_t *wrapped;  // [channel-blocks][VAMD_POSTS_STRIDE] floor1_encode's out[], k_floor -> k_pack; null unless packets are assembled
  unsigned char *status;
};

// floats per channel-block of the run-peak hand-over (k_transform -> k_tone_seed), rows 16-byte aligned
static int run_peaks_stride(const PsyP &P) { return (P.nruns + 3) & ~3; }

// Resolve every inter-stage tensor: the caller's buffer when given, otherwise workspace.
static int plan(vamd_ctx *c, int W, long nb, const vamd_batch_io *io, int level, WsPlan *p) {
  const size_t ch = c->B.channels, n2 = c->B.bs[W] / 2;
  const size_t per = (size_t)nb * ch * n2 * 4;
  p->wrapped = nullptr;
  void *v;
#define PICK(field, user, slot, bytes)                     \
  do {                                                     \
    if (user) {                                            \
      p->field = user;                                     \
    } else {                                               \
      int r__ = ws_get(c, W, vamd_ctx::slot, (bytes), &v); \
      if (r__) return r__;                                 \
      p->field = (decltype(p->field))v;                    \
    }                                                      \
  } while (0)
  PICK(mdct_raw, io ? io->mdct_raw : nullptr, WS_MDCT_RAW, per);
  p->logmdct = io ? io->logmdct : nullptr;  // a tap only: the later stages form it from mdct_raw
  p->logfft = io ? io->logfft : nullptr;  // a tap only: the tone stage reads the run peaks
  p->peaks = nullptr;
  if (level >= VAMD_LEVEL_PSY) {
    PICK(peaks, (float *)nullptr, WS_LOGFFT, (size_t)nb * ch * run_peaks_stride(c->B.psy[2 * W]) * 4);
  }
  PICK(local, io ? io->local_ampmax : nullptr, WS_LOCAL, (size_t)nb * ch * 4);
  PICK(ampglob, io ? io->ampmax_out : nullptr, WS_AMPGLOB, (size_t)nb * 4);
  PICK(ampin, (float *)nullptr, WS_AMPIN, (size_t)nb * 4);
  PICK(status, io ? io->status : nullptr, WS_STATUS, (size_t)nb * ch);
  if (level >= VAMD_LEVEL_PSY) {
    PICK(noise, io ? io->noise : nullptr, WS_NOISE, per);
    PICK(tone, io ? io->tone : nullptr, WS_TONE, per);
    const size_t nlp = (size_t)VAMD_LINES_PAD(c->B.psy[2 * W].total_octave_lines);
    PICK(seed, (float *)nullptr, WS_SEED, (size_t)nb * ch * nlp * 4);
    PICK(surv, (unsigned short *)nullptr, WS_SURV, (size_t)nb * ch * nlp * 2);
    PICK(nsurv, (int32_t *)nullptr, WS_NSURV, (size_t)nb * ch * 4);
  }
  if (level >= VAMD_LEVEL_FULL) {
    PICK(mdct, io ? io->mdct : nullptr, WS_MDCT, per);
    PICK(ilogmask, (ilog_t *)nullptr, WS_ILOGMASK, per / 4 * sizeof(ilog_t));
    PICK(iwork, io ? io->iwork : nullptr, WS_IWORK, per);
    PICK(posts, io ? io->posts : nullptr, WS_POSTS, (size_t)nb * ch * VAMD_POSTS_STRIDE * 4);
    PICK(post_valid, io ? io->post_valid : nullptr, WS_POSTVALID, (size_t)nb * ch * 4);
    PICK(nonzero, io ? io->nonzero : nullptr, WS_NONZERO, (size_t)nb * ch * 4);
    if (io && io->packets) PICK(wrapped, (int32_t *)nullptr, WS_WRAPPED, (size_t)nb * ch * VAMD_POSTS_STRIDE * 4);
  }
#undef PICK
  return VAMD_OK;
}

int vamd_reserve(vamd_ctx *c, int W, long max_blocks) {
  DeviceGuard dev_guard(c);
  if (!c || (W != 0 && W != 1) || max_blocks < 1) return VAMD_EINVAL;
  WsPlan p;
  return plan(c, W, max_blocks, nullptr, VAMD_LEVEL_FULL, &p);
}

int vamd_mdct_forward_batch(vamd_ctx *c, int W, const float *in, float *out, long nframes) {
  DeviceGuard dev_guard(c);
  if (!c || (W != 0 && W != 1) || nframes < 0) return VAMD_EINVAL;
  if (nframes == 0) return VAMD_OK;
  if (!in || !out) return fail(c, VAMD_EINVAL, "null frame buffer");
  if (nframes > 0x7fffffffL) return fail(c, VAMD_EINVAL, "too many frames for one launch");
  const XformP &P = c->B.xf[W];
  int waves = VAMD_MD_WAVES;
  while (waves > 1 && mdct_only_lds_bytes(P, waves) > c->lds_per_block) waves--;
  const long groups = (nframes + waves - 1) / waves;
  const unsigned grid = (unsigned)(groups < c->num_cus ? groups : c->num_cus);
#define VAMD_GO(LOGN)                                                                                                     \
  hipLaunchKernelGGL(k_mdct_only<LOGN>, dim3(grid), dim3(64 * waves), mdct_only_lds_bytes(P, waves), c->stream, P, W, nframes, \
                     in, out)
  switch (fixed_logn(P)) {
    case 8: VAMD_GO(8); break;
    case 9: VAMD_GO(9); break;
    case 10: VAMD_GO(10); break;
    case 11: VAMD_GO(11); break;
    case 12: VAMD_GO(12); break;
    default: VAMD_GO(0);
  }
#undef VAMD_GO
  HIP_TRY(c, hipGetLastError());
  return VAMD_OK;
}

static int check_desc(vamd_ctx *c, const vamd_batch_desc *d, const vamd_batch_io *io) {
  if (!c) return VAMD_EINVAL;
  if (!d || !io || !io->pcm) return fail(c, VAMD_EINVAL, "null descriptor / io / pcm");
  if (d->W != 0 && d->W != 1) return fail(c, VAMD_EINVAL, "W must be 0 or 1");
  if (d->nblocks < 0 || d->nblocks * (long)c->B.channels > 0x7fffffffL)
    return fail(c, VAMD_EINVAL, "nblocks out of range");
  if (!d->blocktype && (d->uniform_blocktype != 0 && d->uniform_blocktype != 1))
    return fail(c, VAMD_EINVAL, "blocktype must be 0 or 1");
  if (!d->lW && (d->uniform_lW & ~1)) return fail(c, VAMD_EINVAL, "lW must be 0 or 1");
  if (!d->nW && (d->uniform_nW & ~1)) return fail(c, VAMD_EINVAL, "nW must be 0 or 1");
  return VAMD_OK;
}

// ---- the launch sequence ---------------------------------------------------------
// the residue search's outputs: the caller's buffers, or workspace when only the packets are wanted
struct ResBufs {
  int32_t *cls;
  uint16_t *entries;
  int32_t *count;
  uint8_t *books;  // [units][res_cap] the book of every entry, k_residue -> k_pack (workspace only)
};
struct BatchRun {
  int W;
  long nb;
  WsPlan p;
  DescP d;
  const vamd_batch_io *io;
  ResBufs rb;
  float *couple_state;  // [units][4][ch][n2] or null (alloc_couple_state)
  bool make_ampmax;     // the block ampmax is formed by k_tone_seed (independent blocks at the psy level or above: no k_ampmax launch)
};

static int check_packets(vamd_ctx *c, int W, int level, const void *packets, const void *bits, int64_t stride) {
  if (!(packets && bits)) return fail(c, VAMD_EINVAL, "packets / packet_bits go together");
  if (level < VAMD_LEVEL_FULL) return fail(c, VAMD_EINVAL, "packet outputs need level FULL");
  if (stride < 4 || (stride & 3) || stride > 0x7fffffffL) return fail(c, VAMD_EINVAL, "packet_stride must be a positive multiple of 4");
  if ((W != 0 && W != 1) || c->B.pack[W].capacity == 0)
    return fail(c, VAMD_EIMPL, "this mode's packets are not assembled on the GPU (its residue back-end is not covered)");
  return VAMD_OK;
}

static int res_bufs(vamd_ctx *c, int W, long units, int32_t *cls, uint16_t *entries, int32_t *count, ResBufs *o) {
  o->cls = cls, o->entries = entries, o->count = count;
  void *v;
  int r;
  if ((r = ws_get(c, W, vamd_ctx::WS_RES_BOOKS, (size_t)units * c->B.res_cap[W], &v))) return r;
  o->books = (uint8_t *)v;
  if (entries) return VAMD_OK;
  if ((r = ws_get(c, W, vamd_ctx::WS_RES_CLASS, (size_t)units * c->B.chmap[W].submaps * VAMD_RES_CLASS_STRIDE * 4, &v))) return r;
  o->cls = (int32_t *)v;
  if ((r = ws_get(c, W, vamd_ctx::WS_RES_ENTRIES, (size_t)units * c->B.res_cap[W] * 2, &v))) return r;
  o->entries = (uint16_t *)v;
  if ((r = ws_get(c, W, vamd_ctx::WS_RES_COUNT, (size_t)units * c->B.chmap[W].submaps * 8, &v))) return r;
  o->count = (int32_t *)v;
  return VAMD_OK;
}

static int prepare_run(vamd_ctx *c, const vamd_batch_desc *desc, const vamd_batch_io *io, int level, BatchRun *R) {
  memset(R, 0, sizeof(*R));
  if (io && (io->res_class || io->res_entries || io->res_count)) {
    if (!(io->res_class && io->res_entries && io->res_count)) return fail(c, VAMD_EINVAL, "res_class / res_entries / res_count go together");
    if (level < VAMD_LEVEL_FULL) return fail(c, VAMD_EINVAL, "residue outputs need level FULL");
    if ((desc->W != 0 && desc->W != 1) || !c->B.res_cap[desc->W])
      return fail(c, VAMD_EIMPL, "this mode's residue back-end is not covered on the GPU (residue types 1 and 2 are)");
  }
  if (io && (io->packets || io->packet_bits)) {
    int r = check_packets(c, desc->W, level, io->packets, io->packet_bits, io->packet_stride);
    if (r) return r;
  }
  R->W = desc->W;
  R->nb = desc->nblocks;
  R->io = io;
  if (R->nb == 0) return VAMD_OK;
  int r = plan(c, R->W, R->nb, io, level, &R->p);
  if (r) return r;
  if (io && level >= VAMD_LEVEL_FULL && (io->res_entries || io->packets) &&
      (r = res_bufs(c, R->W, R->nb, io->res_class, io->res_entries, io->res_count, &R->rb)))
    return r;
  DescP &d = R->d;
  d.lW = desc->lW;
  d.nW = desc->nW;
  d.blocktype = desc->blocktype;
  d.ampmax_in = desc->ampmax_in;
  d.u_lW = desc->uniform_lW;
  d.u_nW = desc->uniform_nW;
  d.u_blocktype = desc->uniform_blocktype;
  d.u_ampmax_in = desc->uniform_ampmax_in;
  d.dbg = c->d_dbg;
  d.clk = c->d_clk;
  d.status = R->p.status;
  d.bad = c->d_bad;
  d.src = nullptr;
  d.cstride = 0;
  if (io && io->pcm_src) {
    if ((io->pcm_channel_stride & 3) || ((uintptr_t)io->pcm & 15)) return fail(c, VAMD_EINVAL, "pcm_src: pcm 16-byte aligned, pcm_channel_stride a multiple of 4");
    d.src = (const long long *)io->pcm_src;
    d.cstride = (long)io->pcm_channel_stride;
  }
  return VAMD_OK;
}

// stage 1 (window, MDCT, FFT, logs, local ampmax)
static void launch_transform(vamd_ctx *c, BatchRun *R) {
  if (R->nb == 0) return;
  const int ch = c->B.channels;
  const XformP &X = c->B.xf[R->W];
  const unsigned gcb = (unsigned)(R->nb * ch);
  const int waves = xf_waves(c, X);
  const long groups = ((long)gcb + waves - 1) / waves;
  const unsigned grid = (unsigned)(groups < c->num_cus ? groups : c->num_cus);
  const PsyP &PS = c->B.psy[2 * R->W];  // (the runs are the size class's: vamd_bind checks both block types share them)
  prof_mark(c, VAMD_ST_BEGIN);
#define VAMD_GO(LOGN)                                                                                                      \
  hipLaunchKernelGGL(k_transform<LOGN>, dim3(grid), dim3(64 * waves), transform_lds_bytes(X, waves), c->stream, X, R->W, R->d, \
                     ch, (long)gcb, R->io->pcm, R->p.mdct_raw, R->p.logmdct, R->p.logfft, R->p.local, PS.run_of_bin, PS.nruns,    \
                     run_peaks_stride(PS), R->p.peaks)
  switch (fixed_logn(X)) {
    case 8: VAMD_GO(8); break;
    case 9: VAMD_GO(9); break;
    case 10: VAMD_GO(10); break;
    case 11: VAMD_GO(11); break;
    case 12: VAMD_GO(12); break;
    default: VAMD_GO(0);
  }
#undef VAMD_GO
  prof_mark(c, VAMD_ST_TRANSFORM);
}

// stages 2..5 (masking, floor, couple); R->d.ampmax_in / p.ampglob must be final
// stage 6 for every submap of the mode, then (optionally) stage 7; a unit is a (block, candidate packet)
static void launch_residue_pack(vamd_ctx *c, BatchRun *R, hipStream_t s, long units, int nblobs, const int *posts,
                                const int *wrapped /* k_floor's out[] per post, or null */, const int *post_valid, const int *iwork, const int *nonzero, const ResBufs &rb,
                                void *packets, int64_t packet_stride, int32_t *packet_bits) {
  const int W = R->W, ch = c->B.channels, n2 = c->B.xf[W].n / 2;
  const ChMap &cm = c->B.chmap[W];
  const long res_team_max = c->K.res_team_max;
  for (int sm = 0; sm < cm.submaps; sm++)
    // a stereo bundle's search keeps two waves busy, the five-channel bundle of the 5.1 layout four; a handful of units
    // takes four either way (nothing else wants the CU, and a lone unit's latency is the caller's)
    hipLaunchKernelGGL(k_residue, dim3((unsigned)units), dim3(64 * (c->B.res[W][sm].bundle * n2 > 4096 || units <= res_team_max ? VAMD_RES_WAVES : 2)),
                       (size_t)c->B.res[W][sm].lds_ints * 4, s, c->B.res[W][sm], cm, sm,
                       c->B.res_cap[W], nblobs, R->d, ch, n2, iwork, nonzero, rb.cls, rb.entries, rb.count, packets ? rb.books : nullptr);
  prof_mark(c, VAMD_ST_RESIDUE);
  if (packets) {
    const size_t lds = ((size_t)VAMD_PK_RING + VAMD_POSTS_STRIDE + VAMD_RES_CLASS_STRIDE + 2 * (size_t)c->B.res_off_ints[W] +
                        VAMD_PK_FTAB_INTS + 3 * (size_t)c->B.pack[W].nbooks) * 4;
    const long pair_max = c->K.pack_pair_max;
    // a handful of packets: two waves each -- where the rows hold any packet (the residue part is assembled past the
    // longest possible head and then moved down: in a shorter row the end of a cut-off packet would be lost on the way)
    if (units <= pair_max && packet_stride >= c->B.pack[W].capacity)
      hipLaunchKernelGGL(k_pack_pair, dim3((unsigned)units), dim3(128), lds + ((size_t)VAMD_PK_RING + 4) * 4, s, c->B.pack[W],
                         c->B.floor[W][0], c->B.floor[W][1], c->B.res[W][0], c->B.res[W][1], cm, c->B.res_cap[W], c->B.res_off_ints[W],
                         R->d, ch, W, nblobs, posts, wrapped, post_valid, rb.cls, rb.entries, rb.books, rb.count, (unsigned *)packets,
                         (int)(packet_stride / 4), packet_bits);
    else
    hipLaunchKernelGGL(k_pack, dim3((unsigned)units), dim3(64), lds, s, c->B.pack[W], c->B.floor[W][0], c->B.floor[W][1],
                       c->B.res[W][0], c->B.res[W][1], cm, c->B.res_cap[W], c->B.res_off_ints[W], R->d, ch, W, nblobs, posts,
                       wrapped, post_valid, rb.cls, rb.entries, rb.books, rb.count, (unsigned *)packets, (int)(packet_stride / 4), packet_bits);
    prof_mark(c, VAMD_ST_PACK);
  }
}

// layouts beyond stereo keep the channels' running state of the coupling stage in HBM (k_couple.h)
static bool needs_general_couple(const vamd_ctx *c, int W) {
  return c->B.channels > 2 || c->B.couple[W].coupling_steps > 1;
}
static int alloc_couple_state(vamd_ctx *c, BatchRun *R, long units) {
  R->couple_state = nullptr;
  if (R->nb == 0 || !needs_general_couple(c, R->W)) return VAMD_OK;
  void *v;
  int r = ws_get(c, R->W, vamd_ctx::WS_COUPLE_STATE, (size_t)units * 4 * c->B.channels * (c->B.bs[R->W] / 2) * 4, &v);
  if (r) return r;
  R->couple_state = (float *)v;
  return VAMD_OK;
}

// couple / quantise / normalise for `units` (block, candidate) pairs
static void launch_couple(vamd_ctx *c, BatchRun *R, hipStream_t s, long units, int blob_base, int nblobs, const float *mdct,
                          const ilog_t *ilogmask, int *iwork, int *nonzero) {
  const int W = R->W, ch = c->B.channels;
  const PsyP &P0 = c->B.psy[2 * W], &P1 = c->B.psy[2 * W + 1];
  const int n2 = c->B.xf[W].n / 2;
  if (needs_general_couple(c, W)) {
    hipLaunchKernelGGL(k_couple_general, dim3((unsigned)units), dim3(64), (size_t)n2 * 12 + 1024, s, P0, P1, c->B.couple_all[W],
                       blob_base, nblobs, R->d, mdct, ilogmask, iwork, nonzero, R->couple_state);
    return;
  }
  // the LDS arrays serve noise normalisation's sort only (lib/psy.c:941-1010); without it the
  // stage is register-only and the CU holds twice as many of its waves
  const bool norm0 = P0.normal_p && P0.normal_start < n2, norm1 = P1.normal_p && P1.normal_start < n2;
  if (norm0 || norm1)
    hipLaunchKernelGGL(k_couple_norm, dim3((unsigned)units), dim3(64), (size_t)n2 * 12 + 1024, s, P0, P1, c->B.couple_all[W], blob_base,
                       nblobs, R->d, mdct, ilogmask, iwork, nonzero, c->couple_band);
  else  // (a handful of blocks: four waves each)
    hipLaunchKernelGGL(k_couple, dim3((unsigned)units), dim3(units <= 2048 && n2 >= 512 ? 256 : 64), 0, s, P0, P1, c->B.couple_all[W],
                       blob_base, nblobs, R->d, mdct, ilogmask, iwork, nonzero, c->couple_band);
}

//   part: 1 = the masks only (noise on the main stream, the tone chain beside it, their join left open), 2 = the rest
//         (join, floor, couple, ...), 3 = both.  A mixed run issues both size classes' masks before either's rest, so that
//         the short blocks' tone chain -- as long as the long blocks', beside a noise mask a fifth as long -- has the long
//         blocks' noise mask and floor fits to run beside (C5: visible tone tail 1.08 -> see DESIGN section 6).
//   forked: the side stream already waits for everything the tone chain needs (run_streams_mixed's ampmax chain)
static void launch_rest(vamd_ctx *c, BatchRun *R, int level, const vamd_managed_io *M = nullptr, ilog_t *m_ilogmask = nullptr,
                        int part = 3, bool forked = false) {
  if (R->nb == 0) return;
  const ResBufs &rb = R->rb;
  const int W = R->W, ch = c->B.channels;
  const WsPlan &p = R->p;
  const DescP &d = R->d;
  const PsyP &P0 = c->B.psy[2 * W], &P1 = c->B.psy[2 * W + 1];
  const int n2 = c->B.xf[W].n / 2, nl = P0.total_octave_lines;
  const unsigned gcb = (unsigned)(R->nb * ch), gb = (unsigned)R->nb;
  hipStream_t s = c->stream;
  // (a handful of blocks: the fork / join through events costs more than running the tone chain beside the noise mask
  // saves -- one stereo block 192 us with it, 181 without)
  const bool overlap = c->overlap && gcb > 64;
  // the VBR path's floor stage takes the tone chain's last step with it (k_floor)
  const bool fold_env = !c->K.fold_separate;
  const int nlp_all = VAMD_LINES_PAD(nl);
  const size_t fold_lds = (size_t)(nlp_all + (P0.ngroups > P1.ngroups ? P0.ngroups : P1.ngroups)) * 4;
  const bool fold_in_floor = fold_env && level >= VAMD_LEVEL_FULL && !M && n2 <= 64 * 4 * VAMD_QPL;
  hipEvent_t ev_join = W ? c->ev_join : c->ev_join2;
  if (level >= VAMD_LEVEL_PSY && (part & 1)) {
    if (overlap && !forked) {  // fork: the tone chain needs only what is already queued on `stream`
      (void)hipEventRecord(c->ev_fork, c->stream);
      (void)hipStreamWaitEvent(c->side, c->ev_fork, 0);
    }
    const int nlp = VAMD_LINES_PAD(nl);
    const size_t seed_lds = (size_t)(seed_pad_lo(P0.eighth_octave_lines) + nlp + seed_pad_hi(P0.eighth_octave_lines)) * 4;
    // a lane per block for batches, a wave per block (the walk in 64 chunks) where that would leave the GPU to a
    // handful of lanes walking ~800 lines each: the per-block entry points, the batcher's small batches
    const long wave_max_cb = c->K.chase_wave_max;
    const bool by_wave = (long)gcb <= wave_max_cb && P0.eighth_octave_lines <= 16 && nl <= 2048;
    const bool lp8 = P0.eighth_octave_lines == 8 && P1.eighth_octave_lines == 8;
    // a handful of blocks, no second stream: both masks in one launch, side by side (k_noise_tone)
    const bool merge_env = !c->K.masks_separate;
    const bool merged = merge_env && !overlap && by_wave && lp8;
    if (merged) {
      const size_t nlds = (size_t)5 * VAMD_NZ_STRIDE(n2) * 4, tlds = seed_lds + (size_t)VAMD_RING * 8;
#define VAMD_GO(L)                                                                                                          \
  hipLaunchKernelGGL((k_noise_tone<L, 8>), dim3(2 * gcb), dim3(64 * NoiseGeom<L>::NW), nlds > tlds ? nlds : tlds, s, P0, P1, d, ch, \
                     (long)gcb, p.mdct_raw, p.noise, nlp, run_peaks_stride(P0), p.peaks, p.local, p.ampglob,                  \
                     R->make_ampmax ? p.ampglob : nullptr, p.seed, p.surv, p.nsurv)
      switch (n2) {
        case 32: VAMD_GO(5); break;
        case 64: VAMD_GO(6); break;
        case 128: VAMD_GO(7); break;
        case 256: VAMD_GO(8); break;
        case 512: VAMD_GO(9); break;
        case 1024: VAMD_GO(10); break;
        default: VAMD_GO(11); break;
      }
#undef VAMD_GO
      prof_mark(c, VAMD_ST_NOISE);
    } else {
      // persistent teams.  A CU's LDS and 32 wave slots hold 8 of them at 1024 bins (both exactly full) -- but then the
      // tone chain on the side stream finds no room until they retire and runs behind them.  Six teams (three quarters of
      // the wave slots) keep the vector units as busy -- the stage is issue-bound -- and leave eight slots and 40 KB in
      // which the tone kernels, which wait on LDS atomics and memory, run BESIDE them: per 131 072 stereo blocks
      // noise + tone tail 2.24 + 1.16 ms with eight teams, 2.42 + 0.82 with seven, 2.62 + 0.50 with six, 2.89 + 0.29
      // with five, 3.30 + 0.01 with four.
      const size_t lds = (size_t)5 * VAMD_NZ_STRIDE(n2) * 4;
      const int nw = n2 >= 256 ? 4 : (n2 >= 64 ? n2 / 64 : 1);
      long per_cu = (long)(c->lds_per_block / lds);
      if (per_cu > 32 / nw) per_cu = 32 / nw;
      const int noise_cap = c->K.noise_teams;  // (measurement aid)
      if (noise_cap > 0) {
        if (per_cu > noise_cap) per_cu = noise_cap;
      } else if (overlap) {
        // ... and where the tone chain carries its own last step (the fold as a launch of its own: the masks-only level,
        // bitrate-managed blocks) it needs half the CU to finish beside the noise mask: four teams.  65 536 stereo blocks
        // at the masks-only level: noise + visible tone tail 1.17 + 0.80 ms with seven teams, 1.25 + 0.64 with six,
        // 1.38 + 0.60 with five, 1.63 + 0.08 with four.
        const long cap = (fold_in_floor ? 24 : 16) / nw;
        if (per_cu > cap) per_cu = cap > 0 ? cap : 1;
      }
      if (per_cu < 1) per_cu = 1;
      const unsigned grid = (unsigned)((long)gcb < per_cu * c->num_cus ? (long)gcb : per_cu * c->num_cus);
#define VAMD_GO(L)                                                                                                    \
  hipLaunchKernelGGL(k_noise<L>, dim3(grid), dim3(64 * NoiseGeom<L>::NW), lds, s, P0, P1, d, ch, (long)gcb, p.mdct_raw, \
                     p.noise)
      switch (n2) {
        case 32: VAMD_GO(5); break;
        case 64: VAMD_GO(6); break;
        case 128: VAMD_GO(7); break;
        case 256: VAMD_GO(8); break;
        case 512: VAMD_GO(9); break;
        case 1024: VAMD_GO(10); break;
        default: VAMD_GO(11); break;  // 2048 bins: the largest block size the context accepts
      }
#undef VAMD_GO
      prof_mark(c, VAMD_ST_NOISE);
    }
    if (overlap) s = c->side;
    {
      if (merged) {
        // (launched with the noise stage)
      } else if (by_wave && lp8) {  // ... and seed + chase in one launch (k_tone_seed_chase)
        hipLaunchKernelGGL(k_tone_seed_chase<8>, dim3(gcb), dim3(64), seed_lds + (size_t)VAMD_RING * 8, s, P0, P1, d, ch, nlp,
                           run_peaks_stride(P0), p.peaks, p.local, p.ampglob, R->make_ampmax ? p.ampglob : nullptr, p.seed, p.surv,
                           p.nsurv);
      } else {
        if (lp8)
          hipLaunchKernelGGL(k_tone_seed<8>, dim3(gcb), dim3(64), seed_lds, s, P0, P1, d, ch, nlp, run_peaks_stride(P0), p.peaks, p.local,
                             p.ampglob, R->make_ampmax ? p.ampglob : nullptr, p.seed);
        else
          hipLaunchKernelGGL(k_tone_seed<0>, dim3(gcb), dim3(64), seed_lds, s, P0, P1, d, ch, nlp, run_peaks_stride(P0), p.peaks, p.local,
                             p.ampglob, R->make_ampmax ? p.ampglob : nullptr, p.seed);
        if (by_wave)
          hipLaunchKernelGGL(k_tone_chase_wave, dim3(gcb), dim3(64), (size_t)nlp * 4 + (size_t)VAMD_RING * 64 * 8, s,
                             P0.eighth_octave_lines, nl, nlp, d, p.seed, p.surv, p.nsurv);
        else
          hipLaunchKernelGGL(k_tone_chase, dim3((gcb + VAMD_CHASE_LANES - 1) / VAMD_CHASE_LANES), dim3(VAMD_CHASE_LANES),
                             (size_t)VAMD_RING * VAMD_CHASE_LANES * 8, s,
                             P0.eighth_octave_lines, nl, nlp, (long)gcb, d, p.seed, p.surv, p.nsurv);
      }
      if (!fold_in_floor)
        hipLaunchKernelGGL(k_tone_fold, dim3(gcb), dim3(64), (size_t)(nlp + (P0.ngroups > P1.ngroups ? P0.ngroups : P1.ngroups)) * 4, s, P0, P1, d, ch, nlp, p.seed, p.surv,
                           p.nsurv, p.local, p.tone);
    }
    if (overlap) (void)hipEventRecord(ev_join, c->side);
    s = c->stream;
  }
  if (!(part & 2)) return;
  if (level >= VAMD_LEVEL_PSY) {
    if (overlap) (void)hipStreamWaitEvent(s, ev_join, 0);  // join
    prof_mark(c, VAMD_ST_TONE);
  }
  if (level >= VAMD_LEVEL_FULL && M) {
    // bitrate-managed: fifteen candidate packets per block
    const size_t flds = (size_t)((n2 + 15) & ~15) * 2 + sizeof(FloorScratch);
    hipLaunchKernelGGL(k_floor_managed, dim3(gcb), dim3(64), flds, s, P0, P1, c->B.floor[W][0], c->B.floor[W][1], c->B.chmap[W], d, ch, p.noise, p.tone,
                       p.mdct_raw, p.mdct, R->io->logmask, M->posts, M->post_valid, m_ilogmask, M->nonzero);
    prof_mark(c, VAMD_ST_FLOOR);
    launch_couple(c, R, s, (long)gb * VAMD_PACKETBLOBS, 0, VAMD_PACKETBLOBS, p.mdct, m_ilogmask, M->iwork, M->nonzero);
    prof_mark(c, VAMD_ST_COUPLE);
    if (M->res_entries || M->packets)
      launch_residue_pack(c, R, s, (long)gb * VAMD_PACKETBLOBS, VAMD_PACKETBLOBS, M->posts, nullptr, M->post_valid, M->iwork, M->nonzero, rb,
                          M->packets, M->packet_stride, M->packet_bits);
  } else if (level >= VAMD_LEVEL_FULL) {
    const size_t floor_pad = (size_t)c->K.floor_lds_pad;  // (experiment: occupancy)
    size_t floor_lds = (size_t)((n2 + 15) & ~15) * 2 + sizeof(FloorScratch) + floor_pad;
    if (fold_in_floor && fold_lds > floor_lds) floor_lds = fold_lds;
    // two channels per wave (k_floor_pair) for stereo setups whose channels share a floor of at most 32 posts, from
    // `floor_pair_min` channel-blocks up (a test knob; the default is set by what was measured: DESIGN section 6)
    const FloorP &F0 = c->B.floor[W][c->B.chmap[W].sub[0]];
    const long pair_min = c->K.floor_pair_min >= 0 ? c->K.floor_pair_min : (W ? VAMD_FLOOR_PAIR_MIN_LONG : VAMD_FLOOR_PAIR_MIN_SHORT);
    const bool paired = ch == 2 && c->B.chmap[W].sub[0] == c->B.chmap[W].sub[1] && F0.posts <= 32 && (long)gcb >= pair_min && pair_min >= 0 &&
                        ((c->K.floor_pair_w >> W) & 1) &&
                        n2 <= 32 * 4 * 8 && 2 * floor_lds <= c->lds_per_block;
    if (paired)
      hipLaunchKernelGGL(k_floor_pair, dim3(gb), dim3(64), 2 * floor_lds, s,
                         (const Bound *)c->d_bound, W, d, (int)floor_lds, p.noise, fold_in_floor ? R->io->tone : p.tone, fold_in_floor ? p.seed : nullptr, p.surv, p.nsurv, p.local,
                         nlp_all, p.mdct_raw, p.mdct,
                         R->io->logmask, p.posts, p.post_valid, p.ilogmask, p.nonzero, p.wrapped);
    else
    hipLaunchKernelGGL(k_floor, dim3(gcb), dim3(64), floor_lds, s,
                       (const Bound *)c->d_bound, W, d, ch, p.noise, fold_in_floor ? R->io->tone : p.tone, fold_in_floor ? p.seed : nullptr, p.surv, p.nsurv, p.local,
                       nlp_all, p.mdct_raw, p.mdct,
                       R->io->logmask, p.posts, p.post_valid, p.ilogmask, p.nonzero, p.wrapped);
    if (R->io->ilogmask)  // (a tap: tests and callers with their own quantiser)
      hipLaunchKernelGGL(k_widen_ilog, dim3(1024), dim3(256), 0, s, (long)gcb * n2, (const ilog_t *)p.ilogmask, R->io->ilogmask);
    prof_mark(c, VAMD_ST_FLOOR);
    launch_couple(c, R, s, gb, VAMD_PACKETBLOBS / 2, 1, p.mdct, p.ilogmask, p.iwork, p.nonzero);
    prof_mark(c, VAMD_ST_COUPLE);
    if (R->io && (R->io->res_entries || R->io->packets))
      launch_residue_pack(c, R, s, gb, 1, p.posts, p.wrapped, p.post_valid, p.iwork, p.nonzero, rb, R->io->packets, R->io->packet_stride,
                          R->io->packet_bits);
  }
}

static int run_batch(vamd_ctx *c, const vamd_batch_desc *desc, const vamd_batch_io *io, int level, bool stream_mode,
                     float *ampmax_state, const vamd_managed_io *M = nullptr) {
  BatchRun R;
  int r = prepare_run(c, desc, io, level, &R);
  if (r) return r;
  if (R.nb == 0) return VAMD_OK;
  ilog_t *m_ilogmask = nullptr;
  if (M) {  // the fifteen integer floor curves live in workspace only
    void *v;
    r = ws_get(c, R.W, vamd_ctx::WS_M_ILOGMASK,
               (size_t)R.nb * VAMD_PACKETBLOBS * c->B.channels * (c->B.bs[R.W] / 2) * sizeof(ilog_t), &v);
    if (r) return r;
    m_ilogmask = (ilog_t *)v;
    if ((M->res_entries || M->packets) &&
        (r = res_bufs(c, R.W, R.nb * VAMD_PACKETBLOBS, M->res_class, M->res_entries, M->res_count, &R.rb)))
      return r;
  }
  if (level >= VAMD_LEVEL_FULL && (r = alloc_couple_state(c, &R, M ? R.nb * VAMD_PACKETBLOBS : R.nb))) return r;
  const int ch = c->B.channels;
  hipStream_t s = c->stream;
  launch_transform(c, &R);
  if (stream_mode) {
    const float secs = (float)(c->B.xf[R.W].n / 2) / (float)c->B.rate;  // lib/psy.c:842-843
    hipLaunchKernelGGL(k_ampmax_stream, dim3(1), dim3(1), 0, s, ch, R.nb, secs, c->B.ampmax_att_per_sec, *ampmax_state,
                       R.p.local, R.p.ampin, R.p.ampglob);
    R.d.ampmax_in = R.p.ampin;
  } else if (level >= VAMD_LEVEL_PSY) {
    R.make_ampmax = true;  // (one launch less: 4 us of a single block's 180)
  } else {
    hipLaunchKernelGGL(k_ampmax, dim3((unsigned)((R.nb + 255) / 256)), dim3(256), 0, s, R.d, ch, R.nb, R.p.local,
                       R.p.ampglob);
  }
  prof_mark(c, VAMD_ST_AMPMAX);
  launch_rest(c, &R, level, M, m_ilogmask);
  if (c->profile) c->prof_runs++;
  HIP_TRY(c, hipGetLastError());
  if (stream_mode) {
    // new state = ampmax_out of the last block
    HIP_TRY(c, hipMemcpyAsync(ampmax_state, R.p.ampglob + (R.nb - 1), sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
  }
  return VAMD_OK;
}

int vamd_analyze_batch(vamd_ctx *c, const vamd_batch_desc *desc, const vamd_batch_io *io, int level) {
  DeviceGuard dev_guard(c);
  int r = check_desc(c, desc, io);
  if (r) return r;
  if (level < VAMD_LEVEL_TRANSFORM || level > VAMD_LEVEL_FULL) return fail(c, VAMD_EINVAL, "bad level");
  return run_batch(c, desc, io, level, false, nullptr);
}

int vamd_analyze_batch_managed(vamd_ctx *c, const vamd_batch_desc *desc, const vamd_batch_io *io,
                               const vamd_managed_io *m) {
  DeviceGuard dev_guard(c);
  int r = check_desc(c, desc, io);
  if (r) return r;
  if (!m || !m->posts || !m->post_valid || !m->iwork || !m->nonzero)
    return fail(c, VAMD_EINVAL, "managed outputs posts / post_valid / iwork / nonzero are required");
  if (m->res_class || m->res_entries || m->res_count) {
    if (!(m->res_class && m->res_entries && m->res_count))
      return fail(c, VAMD_EINVAL, "res_class / res_entries / res_count go together");
    if (!c->B.res_cap[desc->W])
      return fail(c, VAMD_EIMPL, "this mode's residue back-end is not covered on the GPU (residue types 1 and 2 are)");
  }
  if ((m->packets || m->packet_bits) &&
      (r = check_packets(c, desc->W, VAMD_LEVEL_FULL, m->packets, m->packet_bits, m->packet_stride)))
    return r;
  vamd_batch_io shared = *io;  // per-candidate fields of the VBR io do not apply
  shared.packets = nullptr;
  shared.packet_bits = nullptr;
  shared.posts = shared.post_valid = shared.ilogmask = shared.iwork = shared.nonzero = nullptr;
  shared.res_class = nullptr;
  shared.res_entries = nullptr;
  shared.res_count = nullptr;
  return run_batch(c, desc, &shared, VAMD_LEVEL_FULL, false, nullptr, m);
}

int vamd_analyze_block_managed(vamd_ctx *c, const float *const *pcm, int lW, int W, int nW, int blocktype,
                               float ampmax_in, float *mdct, float *ampmax_out, int32_t *posts,
                               int32_t *post_valid, int32_t *iwork, int32_t *nonzero, int32_t *res_class,
                               uint16_t *res_entries, int32_t *res_count) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (!pcm || (W != 0 && W != 1)) return fail(c, VAMD_EINVAL, "bad pcm / W");
  const bool want_res = res_class || res_entries || res_count;
  if (want_res && !c->B.res_cap[W])
    return fail(c, VAMD_EIMPL, "this mode's residue back-end is not covered on the GPU (residue types 1 and 2 are)");
  const size_t rcap = want_res ? (size_t)c->B.res_cap[W] : 0;
  const size_t S = (size_t)c->B.chmap[W].submaps;
  const size_t ch = c->B.channels, n = c->B.bs[W], n2 = n / 2, K = VAMD_PACKETBLOBS;
  auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const size_t o_pcm = 0, o_mdct = al(o_pcm + ch * n * 4), o_amp = al(o_mdct + ch * n2 * 4), o_posts = o_amp + 16,
               o_valid = al(o_posts + K * ch * VAMD_POSTS_STRIDE * 4), o_nz = al(o_valid + K * ch * 4),
               o_iwork = al(o_nz + K * ch * 4), o_rcls = al(o_iwork + K * ch * n2 * 4),
               o_rcnt = al(o_rcls + (want_res ? K * S * VAMD_RES_CLASS_STRIDE * 4 : 0)),
               o_rent = al(o_rcnt + (want_res ? K * S * 2 * 4 : 0)), total = al(o_rent + K * rcap * 2);
  if (c->h_stage_bytes < total) {
    if (c->h_stage) HIP_TRY(c, hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->h_stage_bytes = 0;
    HIP_TRY(c, hipHostMalloc(&c->h_stage, total, hipHostMallocDefault));
    c->h_stage_bytes = total;
  }
  void *dv;
  int r = ws_get(c, W, vamd_ctx::WS_M_STAGE, total, &dv);
  if (r) return r;
  unsigned char *hs = (unsigned char *)c->h_stage, *ds = (unsigned char *)dv;
  for (size_t i = 0; i < ch; i++) {
    if (!pcm[i]) return fail(c, VAMD_EINVAL, "null channel pointer");
    memcpy(hs + o_pcm + i * n * 4, pcm[i], n * 4);
  }
  hipStream_t s = c->stream;
  HIP_TRY(c, hipMemcpyAsync(ds + o_pcm, hs + o_pcm, ch * n * 4, hipMemcpyHostToDevice, s));
  vamd_batch_desc d;
  memset(&d, 0, sizeof(d));
  d.W = W;
  d.nblocks = 1;
  d.uniform_lW = lW;
  d.uniform_nW = nW;
  d.uniform_blocktype = blocktype;
  d.uniform_ampmax_in = ampmax_in;
  vamd_batch_io io;
  memset(&io, 0, sizeof(io));
  io.pcm = (const float *)(ds + o_pcm);
  io.mdct = (float *)(ds + o_mdct);
  io.ampmax_out = (float *)(ds + o_amp);
  io.status = ds + o_amp + 4;
  vamd_managed_io m;
  memset(&m, 0, sizeof(m));
  m.posts = (int32_t *)(ds + o_posts);
  m.post_valid = (int32_t *)(ds + o_valid);
  m.nonzero = (int32_t *)(ds + o_nz);
  m.iwork = (int32_t *)(ds + o_iwork);
  if (want_res) {
    m.res_class = (int32_t *)(ds + o_rcls);
    m.res_count = (int32_t *)(ds + o_rcnt);
    m.res_entries = (uint16_t *)(ds + o_rent);
  }
  r = vamd_analyze_batch_managed(c, &d, &io, &m);
  if (r) return r;
  HIP_TRY(c, hipMemcpyAsync(hs + o_mdct, ds + o_mdct, total - o_mdct, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  if (ampmax_out) memcpy(ampmax_out, hs + o_amp, 4);  // (the block's ampmax comes out of its FFT: delivered with a domain error too)
  if ((r = status_verdict(c, hs + o_amp + 4, ch))) return r;
  if (mdct) memcpy(mdct, hs + o_mdct, ch * n2 * 4);
  if (posts) memcpy(posts, hs + o_posts, K * ch * VAMD_POSTS_STRIDE * 4);
  if (post_valid) memcpy(post_valid, hs + o_valid, K * ch * 4);
  if (nonzero) memcpy(nonzero, hs + o_nz, K * ch * 4);
  if (iwork) memcpy(iwork, hs + o_iwork, K * ch * n2 * 4);
  if (want_res) {
    if (res_class) memcpy(res_class, hs + o_rcls, K * S * VAMD_RES_CLASS_STRIDE * 4);
    if (res_count) memcpy(res_count, hs + o_rcnt, K * S * 2 * 4);
    if (res_entries) memcpy(res_entries, hs + o_rent, K * rcap * 2);
  }
  return VAMD_OK;
}

int vamd_analyze_stream(vamd_ctx *c, const vamd_batch_desc *desc, const vamd_batch_io *io, float *ampmax_state) {
  DeviceGuard dev_guard(c);
  int r = check_desc(c, desc, io);
  if (r) return r;
  if (!ampmax_state) return fail(c, VAMD_EINVAL, "null ampmax_state");
  return run_batch(c, desc, io, VAMD_LEVEL_FULL, true, ampmax_state);
}

// the two-size-class stream run; nstreams == 0: one stream whose state is the host float *ampmax_state,
// otherwise `stream_start` [nstreams+1] and `states` [nstreams] are device arrays
static int run_streams_mixed(vamd_ctx *c, const vamd_batch_desc *desc_short, const vamd_batch_io *io_short,
                             const vamd_batch_desc *desc_long, const vamd_batch_io *io_long, const int32_t *order,
                             long nblocks_total, float *ampmax_state, const int64_t *stream_start, long nstreams,
                             float *states, bool first_given = false) {
  if (desc_short->W != 0 || desc_long->W != 1) return fail(c, VAMD_EINVAL, "desc_short->W must be 0, desc_long->W 1");
  if (nblocks_total != desc_short->nblocks + desc_long->nblocks || (nblocks_total && !order))
    return fail(c, VAMD_EINVAL, "order[] must name every block of both batches exactly once");
  int r;
  if (desc_short->nblocks && (r = check_desc(c, desc_short, io_short))) return r;
  if (desc_long->nblocks && (r = check_desc(c, desc_long, io_long))) return r;
  if (nblocks_total == 0) return VAMD_OK;
  BatchRun R[2];
  if ((r = prepare_run(c, desc_short, io_short, VAMD_LEVEL_FULL, &R[0]))) return r;
  if ((r = prepare_run(c, desc_long, io_long, VAMD_LEVEL_FULL, &R[1]))) return r;
  if ((r = alloc_couple_state(c, &R[0], R[0].nb)) || (r = alloc_couple_state(c, &R[1], R[1].nb))) return r;
  // scratch for the chained state; an empty size class still needs valid (unused) pointers
  void *misc = nullptr;
  if ((r = ws_get(c, 0, vamd_ctx::WS_MISC, 256, &misc))) return r;
  float *d_state = (float *)misc;
  for (int W = 0; W < 2; W++)
    if (R[W].nb == 0) R[W].p.ampin = R[W].p.ampglob = R[W].p.local = (float *)misc + 16;
  hipStream_t s = c->stream;
  launch_transform(c, &R[0]);
  launch_transform(c, &R[1]);
  const float secs0 = (float)(c->B.bs[0] / 2) / (float)c->B.rate, secs1 = (float)(c->B.bs[1] / 2) / (float)c->B.rate;
  // The chains' walk (a thread per stream, ~0.3 ms for a thousand streams of 130 blocks: latency, not load) feeds the tone
  // seeds and nothing else of the masking stage, so where the tone chain runs on the side stream the walk goes there
  // too, ahead of it, and the noise masks start at once on the main stream.
  const bool chain_on_side = nstreams && c->overlap && (R[0].nb == 0 || R[0].nb * c->B.channels > 64) &&
                             (R[1].nb == 0 || R[1].nb * c->B.channels > 64);
  if (chain_on_side) {
    (void)hipEventRecord(c->ev_fork, c->stream);
    (void)hipStreamWaitEvent(c->side, c->ev_fork, 0);
    s = c->side;
  }
  if (nstreams)
    hipLaunchKernelGGL(k_ampmax_streams_mixed, dim3((unsigned)((nstreams + 63) / 64)), dim3(64), 0, s, c->B.channels, nstreams,
                       (const long long *)stream_start, (const int *)order, secs0, secs1, c->B.ampmax_att_per_sec, states,
                       R[0].p.local, R[1].p.local, R[0].p.ampin, R[1].p.ampin, R[0].p.ampglob, R[1].p.ampglob);
  else
    hipLaunchKernelGGL(k_ampmax_stream_mixed, dim3(1), dim3(1), 0, s, c->B.channels, nblocks_total, (const int *)order, secs0,
                       secs1, c->B.ampmax_att_per_sec, *ampmax_state, R[0].p.local, R[1].p.local, R[0].p.ampin,
                       R[1].p.ampin, R[0].p.ampglob, R[1].p.ampglob, d_state, first_given ? 1 : 0);
  s = c->stream;
  prof_mark(c, VAMD_ST_AMPMAX);
  R[0].d.ampmax_in = R[0].p.ampin;
  R[1].d.ampmax_in = R[1].p.ampin;
  if (chain_on_side) {  // both classes' masks first, the long blocks' leading
    launch_rest(c, &R[1], VAMD_LEVEL_FULL, nullptr, nullptr, 1, true);
    launch_rest(c, &R[0], VAMD_LEVEL_FULL, nullptr, nullptr, 1, true);
    launch_rest(c, &R[1], VAMD_LEVEL_FULL, nullptr, nullptr, 2, true);
    launch_rest(c, &R[0], VAMD_LEVEL_FULL, nullptr, nullptr, 2, true);
  } else {
    launch_rest(c, &R[0], VAMD_LEVEL_FULL);
    launch_rest(c, &R[1], VAMD_LEVEL_FULL);
  }
  if (chain_on_side && R[0].nb == 0 && R[1].nb == 0) {  // (cannot happen -- nblocks_total > 0 -- but nothing may be left unjoined)
    (void)hipEventRecord(c->ev_join, c->side);
    (void)hipStreamWaitEvent(c->stream, c->ev_join, 0);
  }
  if (c->profile) c->prof_runs++;
  HIP_TRY(c, hipGetLastError());
  if (!nstreams) {
    HIP_TRY(c, hipMemcpyAsync(ampmax_state, d_state, sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
  }
  return VAMD_OK;
}

int vamd_analyze_stream_mixed(vamd_ctx *c, const vamd_batch_desc *desc_short, const vamd_batch_io *io_short,
                              const vamd_batch_desc *desc_long, const vamd_batch_io *io_long, const int32_t *order,
                              long nblocks_total, float *ampmax_state) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (!desc_short || !desc_long || !ampmax_state) return fail(c, VAMD_EINVAL, "null argument");
  return run_streams_mixed(c, desc_short, io_short, desc_long, io_long, order, nblocks_total, ampmax_state, nullptr, 0, nullptr);
}

int vamd_analyze_streams_mixed(vamd_ctx *c, const vamd_batch_desc *desc_short, const vamd_batch_io *io_short,
                               const vamd_batch_desc *desc_long, const vamd_batch_io *io_long, const int32_t *order,
                               const int64_t *stream_start, long nstreams, long nblocks_total, float *ampmax_states) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (!desc_short || !desc_long) return fail(c, VAMD_EINVAL, "null argument");
  if (nstreams < 1 || !stream_start || !ampmax_states) return fail(c, VAMD_EINVAL, "stream_start / ampmax_states / nstreams");
  return run_streams_mixed(c, desc_short, io_short, desc_long, io_long, order, nblocks_total, nullptr, stream_start, nstreams,
                           ampmax_states);
}

int vamd_analyze_block(vamd_ctx *c, const float *const *pcm, int lW, int W, int nW, int blocktype, float ampmax_in,
                       float *mdct, float *logmask, int32_t *posts, int32_t *post_valid, int32_t *iwork,
                       int32_t *nonzero, float *ampmax_out) {
  DeviceGuard dev_guard(c);
  return vamd_analyze_block_res(c, pcm, lW, W, nW, blocktype, ampmax_in, mdct, logmask, posts, post_valid, iwork,
                                nonzero, ampmax_out, nullptr, nullptr, nullptr);
}

int vamd_analyze_block_res(vamd_ctx *c, const float *const *pcm, int lW, int W, int nW, int blocktype,
                           float ampmax_in, float *mdct, float *logmask, int32_t *posts, int32_t *post_valid,
                           int32_t *iwork, int32_t *nonzero, float *ampmax_out, int32_t *res_class,
                           uint16_t *res_entries, int32_t *res_count) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (!pcm || (W != 0 && W != 1)) return fail(c, VAMD_EINVAL, "bad pcm / W");
  const bool want_res = res_class || res_entries || res_count;
  if (want_res && !c->B.res_cap[W])
    return fail(c, VAMD_EIMPL, "this mode's residue back-end is not covered on the GPU (residue types 1 and 2 are)");
  const size_t rcap = want_res ? (size_t)c->B.res_cap[W] : 0;
  const size_t S = (size_t)c->B.chmap[W].submaps;
  const int ch = c->B.channels, n = c->B.bs[W], n2 = n / 2;
  // one pinned + one device arena: [pcm | mdct | logmask | iwork | posts | post_valid | nonzero | ampmax]
  const size_t o_pcm = 0, o_mdct = o_pcm + (size_t)ch * n * 4, o_mask = o_mdct + (size_t)ch * n2 * 4,
               o_iwork = o_mask + (size_t)ch * n2 * 4, o_posts = o_iwork + (size_t)ch * n2 * 4,
               o_valid = o_posts + (size_t)ch * VAMD_POSTS_STRIDE * 4, o_nz = o_valid + (size_t)ch * 4,
               o_amp = ((o_nz + (size_t)ch * 4 + 15) & ~(size_t)15), o_rcls = o_amp + 16, o_rcnt = o_rcls + S * VAMD_RES_CLASS_STRIDE * 4,
               o_rent = o_rcnt + 16, total = o_rent + ((rcap * 2 + 15) & ~(size_t)15);
  if (c->h_stage_bytes < total) {
    if (c->h_stage) HIP_TRY(c, hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->h_stage_bytes = 0;
    HIP_TRY(c, hipHostMalloc(&c->h_stage, total, hipHostMallocDefault));
    c->h_stage_bytes = total;
  }
  void *dv;
  int r = ws_get(c, W, vamd_ctx::WS_PCM, total, &dv);
  if (r) return r;
  unsigned char *hs = (unsigned char *)c->h_stage, *ds = (unsigned char *)dv;
  for (int i = 0; i < ch; i++) {
    if (!pcm[i]) return fail(c, VAMD_EINVAL, "null channel pointer");
    memcpy(hs + o_pcm + (size_t)i * n * 4, pcm[i], (size_t)n * 4);
  }
  hipStream_t s = c->stream;
  HIP_TRY(c, hipMemcpyAsync(ds + o_pcm, hs + o_pcm, (size_t)ch * n * 4, hipMemcpyHostToDevice, s));
  vamd_batch_desc d;
  memset(&d, 0, sizeof(d));
  d.W = W;
  d.nblocks = 1;
  d.uniform_lW = lW;
  d.uniform_nW = nW;
  d.uniform_blocktype = blocktype;
  d.uniform_ampmax_in = ampmax_in;
  vamd_batch_io io;
  memset(&io, 0, sizeof(io));
  io.pcm = (const float *)(ds + o_pcm);
  io.mdct = (float *)(ds + o_mdct);
  io.logmask = (float *)(ds + o_mask);
  io.iwork = (int32_t *)(ds + o_iwork);
  io.posts = (int32_t *)(ds + o_posts);
  io.post_valid = (int32_t *)(ds + o_valid);
  io.nonzero = (int32_t *)(ds + o_nz);
  io.ampmax_out = (float *)(ds + o_amp);
  io.status = ds + o_amp + 4;  // ch <= 8 bytes behind the float, inside its 16-byte slot
  if (want_res) {
    io.res_class = (int32_t *)(ds + o_rcls);
    io.res_count = (int32_t *)(ds + o_rcnt);
    io.res_entries = (uint16_t *)(ds + o_rent);
  }
  r = vamd_analyze_batch(c, &d, &io, VAMD_LEVEL_FULL);
  if (r) return r;
  HIP_TRY(c, hipMemcpyAsync(hs + o_mdct, ds + o_mdct, total - o_mdct, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  if (ampmax_out) memcpy(ampmax_out, hs + o_amp, 4);  // (the block's ampmax comes out of its FFT: delivered with a domain error too)
  if ((r = status_verdict(c, hs + o_amp + 4, (size_t)ch))) return r;
  if (mdct) memcpy(mdct, hs + o_mdct, (size_t)ch * n2 * 4);
  if (logmask) memcpy(logmask, hs + o_mask, (size_t)ch * n2 * 4);
  if (iwork) memcpy(iwork, hs + o_iwork, (size_t)ch * n2 * 4);
  if (posts) memcpy(posts, hs + o_posts, (size_t)ch * VAMD_POSTS_STRIDE * 4);
  if (post_valid) memcpy(post_valid, hs + o_valid, (size_t)ch * 4);
  if (nonzero) memcpy(nonzero, hs + o_nz, (size_t)ch * 4);
  if (want_res) {
    if (res_count) memcpy(res_count, hs + o_rcnt, S * 8);
    if (res_class) memcpy(res_class, hs + o_rcls, S * VAMD_RES_CLASS_STRIDE * 4);
    if (res_entries) memcpy(res_entries, hs + o_rent, rcap * 2);
  }
  return VAMD_OK;
}

int vamd_packet_capacity(const vamd_ctx *c, int W) {
  if (!c || (W != 0 && W != 1)) return 0;
  return c->B.pack[W].capacity;
}

int vamd_encode_block(vamd_ctx *c, const float *const *pcm, int lW, int W, int nW, int blocktype, float ampmax_in,
                      int managed, float *ampmax_out, uint8_t *packets, long packet_stride, int32_t *packet_bits) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (!pcm || (W != 0 && W != 1)) return fail(c, VAMD_EINVAL, "bad pcm / W");
  if (!packets || !packet_bits) return fail(c, VAMD_EINVAL, "null packets / packet_bits");
  const size_t cap = (size_t)c->B.pack[W].capacity;
  if (cap == 0)
    return fail(c, VAMD_EIMPL, "this mode's packets are not assembled on the GPU (its residue back-end is not covered)");
  if (packet_stride < 4) return fail(c, VAMD_EINVAL, "packet_stride too small");
  if (c->K.fail_encode_after >= 0) {  // (test knob: a GPU failure under a block, for the binding's error path)
    static std::atomic<long> calls{0};
    if (calls.fetch_add(1) >= c->K.fail_encode_after) return fail(c, VAMD_EFAULT, "injected failure (VAMD_FAIL_ENCODE_AFTER)");
  }
  const size_t ch = c->B.channels, n = c->B.bs[W], n2 = n / 2, K = managed ? VAMD_PACKETBLOBS : 1;
  const size_t row = cap < (size_t)packet_stride ? cap : ((size_t)packet_stride & ~(size_t)3);  // device row length
  auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
  // one pinned + one device arena: [pcm | ampmax | bits | packets || the managed candidates' intermediates]
  const size_t o_pcm = 0, o_amp = al(ch * n * 4), o_bits = o_amp + 16, o_pk = al(o_bits + K * 4), o_back = al(o_pk + K * row),
               o_posts = o_back, o_valid = al(o_posts + K * ch * VAMD_POSTS_STRIDE * 4), o_nz = al(o_valid + K * ch * 4),
               o_iwork = al(o_nz + K * ch * 4), total = managed ? al(o_iwork + K * ch * n2 * 4) : o_back;
  if (c->h_stage_bytes < o_back) {
    if (c->h_stage) HIP_TRY(c, hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->h_stage_bytes = 0;
    HIP_TRY(c, hipHostMalloc(&c->h_stage, o_back, hipHostMallocDefault));
    c->h_stage_bytes = o_back;
  }
  void *dv;
  int r = ws_get(c, W, managed ? vamd_ctx::WS_M_STAGE : vamd_ctx::WS_PCM, total, &dv);
  if (r) return r;
  unsigned char *hs = (unsigned char *)c->h_stage, *ds = (unsigned char *)dv;
  for (size_t i = 0; i < ch; i++) {
    if (!pcm[i]) return fail(c, VAMD_EINVAL, "null channel pointer");
    memcpy(hs + o_pcm + i * n * 4, pcm[i], n * 4);
  }
  hipStream_t s = c->stream;
  // The kernels read the samples out of, and write the packet into, the pinned arena itself (it is mapped into the
  // device's address space): 16 KB in and a few hundred bytes out per block cross the link inside the first and the
  // last kernel instead of as two copy commands either side of them.  VAMD_STAGE_COPIES=1 brings the copies back
  // (measurement aid).
  const bool staged_copies = c->K.stage_copies;
  unsigned char *io_base = ds;
  if (!staged_copies) {
    void *mapped = nullptr;
    HIP_TRY(c, hipHostGetDevicePointer(&mapped, hs, 0));
    io_base = (unsigned char *)mapped;
  } else {
    HIP_TRY(c, hipMemcpyAsync(ds + o_pcm, hs + o_pcm, ch * n * 4, hipMemcpyHostToDevice, s));
  }
  vamd_batch_desc d;
  memset(&d, 0, sizeof(d));
  d.W = W;
  d.nblocks = 1;
  d.uniform_lW = lW;
  d.uniform_nW = nW;
  d.uniform_blocktype = blocktype;
  d.uniform_ampmax_in = ampmax_in;
  vamd_batch_io io;
  memset(&io, 0, sizeof(io));
  io.pcm = (const float *)(io_base + o_pcm);
  io.ampmax_out = (float *)(io_base + o_amp);
  io.status = io_base + o_amp + 4;  // ch <= 8 bytes behind the float, inside its 16-byte slot
  if (managed) {
    vamd_managed_io m;
    memset(&m, 0, sizeof(m));
    m.posts = (int32_t *)(ds + o_posts);
    m.post_valid = (int32_t *)(ds + o_valid);
    m.nonzero = (int32_t *)(ds + o_nz);
    m.iwork = (int32_t *)(ds + o_iwork);
    m.packets = io_base + o_pk;
    m.packet_bits = (int32_t *)(io_base + o_bits);
    m.packet_stride = (int64_t)row;
    r = vamd_analyze_batch_managed(c, &d, &io, &m);
  } else {
    io.packets = io_base + o_pk;
    io.packet_bits = (int32_t *)(io_base + o_bits);
    io.packet_stride = (int64_t)row;
    r = vamd_analyze_batch(c, &d, &io, VAMD_LEVEL_FULL);
  }
  if (r) return r;
  if (staged_copies) HIP_TRY(c, hipMemcpyAsync(hs + o_amp, ds + o_amp, o_back - o_amp, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  if (ampmax_out) memcpy(ampmax_out, hs + o_amp, 4);  // (the block's ampmax comes out of its FFT: delivered with a domain error too)
  if ((r = status_verdict(c, hs + o_amp + 4, ch))) return r;
  memcpy(packet_bits, hs + o_bits, K * 4);
  for (size_t k = 0; k < K; k++) {
    size_t bytes = ((size_t)(packet_bits[k] > 0 ? packet_bits[k] : 0) + 7) / 8;
    if (bytes > row) bytes = row;  // (cut off: packet_bits says so)
    memcpy(packets + k * (size_t)packet_stride, hs + o_pk + k * row, bytes);
  }
  return VAMD_OK;
}

// N consecutive blocks of ONE stream from host memory to their packets in one launch sequence (the binding's look-ahead,
// integration/mapping0_vamd.c): what vamd_encode_block does for one block, with the ampmax chain between them on the device.
int vamd_encode_blocks(vamd_ctx *c, long nblocks, const float *const *pcm, const int32_t *lW, const int32_t *W,
                       const int32_t *nW, const int32_t *blocktype, float ampmax_in_first, float *ampmax_in, float *ampmax_out,
                       uint8_t *packets, long packet_stride, int32_t *packet_bits, int32_t *verdict) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (nblocks < 0 || nblocks > 0x3fffffffL) return fail(c, VAMD_EINVAL, "nblocks out of range");
  if (nblocks == 0) return VAMD_OK;
  if (!pcm || !lW || !W || !nW || !blocktype || !packets || !packet_bits || !verdict) return fail(c, VAMD_EINVAL, "null argument");
  if (packet_stride < 4) return fail(c, VAMD_EINVAL, "packet_stride too small");
  const size_t ch = c->B.channels;
  if (c->B.pack[0].capacity == 0 || c->B.pack[1].capacity == 0)
    return fail(c, VAMD_EIMPL, "this mode's packets are not assembled on the GPU (its residue back-end is not covered)");
  if (c->K.fail_encode_after >= 0) {  // (test knob, as in vamd_encode_block)
    static std::atomic<long> calls{0};
    if (calls.fetch_add(1) >= c->K.fail_encode_after) return fail(c, VAMD_EFAULT, "injected failure (VAMD_FAIL_ENCODE_AFTER)");
  }
  long nb[2] = {0, 0};
  for (long b = 0; b < nblocks; b++) {
    if (W[b] != 0 && W[b] != 1) return fail(c, VAMD_EINVAL, "W must be 0 or 1");
    if ((lW[b] & ~1) || (nW[b] & ~1) || (blocktype[b] & ~1)) return fail(c, VAMD_EINVAL, "lW / nW / blocktype must be 0 or 1");
    for (size_t k = 0; k < ch; k++)
      if (!pcm[b * ch + k]) return fail(c, VAMD_EINVAL, "null channel pointer");
    nb[W[b]]++;
  }
  auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
  // one pinned arena, read and written in place by the kernels (mapped): per size class [pcm | lW | nW | blocktype |
  // ampmax_out | bits | status | packets], then the stream order
  size_t o_pcm[2], o_lW[2], o_nW[2], o_bt[2], o_amp[2], o_bits[2], o_st[2], o_pk[2], row[2], at = 0;
  for (int w = 0; w < 2; w++) {
    const size_t n = c->B.bs[w], cap = (size_t)c->B.pack[w].capacity;
    row[w] = cap < (size_t)packet_stride ? cap : ((size_t)packet_stride & ~(size_t)3);
    o_pcm[w] = at, at = al(at + (size_t)nb[w] * ch * n * 4);
    o_lW[w] = at, at = al(at + (size_t)nb[w] * 4);
    o_nW[w] = at, at = al(at + (size_t)nb[w] * 4);
    o_bt[w] = at, at = al(at + (size_t)nb[w] * 4);
    o_amp[w] = at, at = al(at + (size_t)nb[w] * 4);
    o_bits[w] = at, at = al(at + (size_t)nb[w] * 4);
    o_st[w] = at, at = al(at + (size_t)nb[w] * ch);
    o_pk[w] = at, at = al(at + (size_t)nb[w] * row[w]);
  }
  const size_t o_order = at, total = al(o_order + (size_t)nblocks * 4);
  if (c->h_stage_bytes < total) {
    if (c->h_stage) HIP_TRY(c, hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->h_stage_bytes = 0;
    HIP_TRY(c, hipHostMalloc(&c->h_stage, total + total / 2, hipHostMallocDefault));
    c->h_stage_bytes = total + total / 2;
  }
  unsigned char *hs = (unsigned char *)c->h_stage;
  void *mapped = nullptr;
  HIP_TRY(c, hipHostGetDevicePointer(&mapped, hs, 0));
  unsigned char *ds = (unsigned char *)mapped;
  std::vector<long> slot((size_t)nblocks);  // block b's index inside its size class
  long seen[2] = {0, 0};
  for (long b = 0; b < nblocks; b++) {
    const int w = W[b];
    const long i = seen[w]++;
    const size_t n = c->B.bs[w];
    slot[(size_t)b] = i;
    for (size_t k = 0; k < ch; k++) memcpy(hs + o_pcm[w] + ((size_t)i * ch + k) * n * 4, pcm[b * ch + k], n * 4);
    ((int32_t *)(hs + o_lW[w]))[i] = lW[b];
    ((int32_t *)(hs + o_nW[w]))[i] = nW[b];
    ((int32_t *)(hs + o_bt[w]))[i] = blocktype[b];
    ((int32_t *)(hs + o_order))[b] = (int32_t)((w << 30) | (int)i);
  }
  vamd_batch_desc d[2];
  vamd_batch_io io[2];
  memset(d, 0, sizeof(d));
  memset(io, 0, sizeof(io));
  for (int w = 0; w < 2; w++) {
    d[w].W = w;
    d[w].nblocks = nb[w];
    d[w].lW = (const int32_t *)(ds + o_lW[w]);
    d[w].nW = (const int32_t *)(ds + o_nW[w]);
    d[w].blocktype = (const int32_t *)(ds + o_bt[w]);
    io[w].pcm = (const float *)(ds + o_pcm[w]);
    io[w].ampmax_out = (float *)(ds + o_amp[w]);
    io[w].status = ds + o_st[w];
    io[w].packets = ds + o_pk[w];
    io[w].packet_bits = (int32_t *)(ds + o_bits[w]);
    io[w].packet_stride = (int64_t)row[w];
  }
  float state = ampmax_in_first;
  int r = run_streams_mixed(c, &d[0], &io[0], &d[1], &io[1], (const int32_t *)(ds + o_order), nblocks, &state, nullptr, 0, nullptr,
                            true);  // (synchronises: the chain's final state comes back)
  if (r) return r;
  const float att = c->B.ampmax_att_per_sec;
  float prev_out = 0.f;
  for (long b = 0; b < nblocks; b++) {
    const int w = W[b];
    const long i = slot[(size_t)b];
    const float out = ((const float *)(hs + o_amp[w]))[i];
    if (ampmax_in) {  // what the block received: the caller's figure, then _vp_ampmax_decay of its predecessor's (lib/psy.c:837-848)
      float a = ampmax_in_first;
      if (b > 0) {
        a = prev_out + ((float)(c->B.bs[w] / 2) / (float)c->B.rate) * att;
        if (a < -9999) a = -9999;
      }
      ampmax_in[b] = a;
    }
    prev_out = out;
    if (ampmax_out) ampmax_out[b] = out;
    unsigned any = 0;
    for (size_t k = 0; k < ch; k++) any |= hs[o_st[w] + (size_t)i * ch + k];
    verdict[b] = (any & VAMD_STATUS_NONFINITE) ? VAMD_ENONFINITE : ((any & VAMD_STATUS_RANGE) ? VAMD_EDOMAIN : VAMD_OK);
    const int32_t bits = ((const int32_t *)(hs + o_bits[w]))[i];
    packet_bits[b] = bits;
    size_t bytes = ((size_t)(bits > 0 ? bits : 0) + 7) / 8;
    if (bytes > row[w]) bytes = row[w];  // (cut off: packet_bits says so)
    if (verdict[b] == VAMD_OK) memcpy(packets + (size_t)b * (size_t)packet_stride, hs + o_pk[w] + (size_t)i * row[w], bytes);
  }
  return VAMD_OK;
}

int vamd_residue_capacity(const vamd_ctx *c, int W) {
  if (!c || (W != 0 && W != 1)) return 0;
  return c->B.res_cap[W];
}

int vamd_submaps(const vamd_ctx *c, int W) { return (c && (W == 0 || W == 1)) ? c->B.chmap[W].submaps : VAMD_EINVAL; }

int vamd_residue_offset(const vamd_ctx *c, int W, int submap) {
  if (!c || (W != 0 && W != 1) || submap < 0 || submap >= c->B.chmap[W].submaps) return VAMD_EINVAL;
  return c->B.res[W][submap].ent_base;
}

int vamd_envelope_geometry(const vamd_ctx *c, int *winlength, int *searchstep) {
  if (!c) return VAMD_EINVAL;
  if (winlength) *winlength = c->B.env.mdct.n;
  if (searchstep) *searchstep = c->B.env.searchstep;
  return VAMD_OK;
}

// `bad`: the word (device) that counts detector steps outside the input domain
static int envelope_search_batch(vamd_ctx *c, const float *pcm, long stream_stride, long channel_stride, long nstreams,
                                 long nsteps, vamd_envelope_state *states, unsigned char *ret, unsigned int *bad) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (nstreams < 0 || nsteps < 0) return fail(c, VAMD_EINVAL, "negative stream / step count");
  if (nstreams == 0 || nsteps == 0) return VAMD_OK;
  if (!pcm || !states || !ret) return fail(c, VAMD_EINVAL, "null pcm / states / ret");
  const EnvP &E = c->B.env;
  const int ch = c->B.channels, n = E.mdct.n, n2 = n / 2;
  const long nsc = nstreams * ch;
  void *v_near, *v_raw, *v_amp, *v_bits;
  int r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_ENV_NEAR, (size_t)nsc * (VAMD_VE_NEAR_HIST + nsteps) * 4, &v_near))) return r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_ENV_RAW, (size_t)nsc * nsteps * VAMD_VE_SPREAD * 4, &v_raw))) return r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_ENV_AMP, (size_t)nsc * (VAMD_VE_AMP_HIST + nsteps) * 8 * 4, &v_amp))) return r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_ENV_BITS, (size_t)nstreams * nsteps * 4, &v_bits))) return r;
  float *near = (float *)v_near, *raw = (float *)v_raw, *amp = (float *)v_amp;
  uint32_t *bits = (uint32_t *)v_bits;
  hipStream_t s = c->stream;
  {
    const long t = nsc * (VAMD_VE_NEAR_HIST + VAMD_VE_AMP_HIST * 8);
    hipLaunchKernelGGL(k_env_prolog, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, ch, nstreams, nsteps, states,
                       near, amp);
  }
  {
    const long items = nsc * ((nsteps + VAMD_ENV_STEPS - 1) / VAMD_ENV_STEPS);
    const long groups = (items + VAMD_ENV_WAVES - 1) / VAMD_ENV_WAVES;
    const long cap = (long)c->num_cus * 8;
    const size_t lds = (size_t)VAMD_ENV_WAVES * VAMD_ENV_STEPS * (n + n2 + VAMD_PW_SIZE(n2) + n2) * 4;
    hipLaunchKernelGGL(k_env_spectrum, dim3((unsigned)(groups < cap ? groups : cap)), dim3(64 * VAMD_ENV_WAVES), lds, s, E,
                       ch, nstreams, nsteps, pcm, stream_stride, channel_stride, near, raw, bad);
  }
  const bool env_untiled = c->K.env_untiled;  // (measurement aid: the thread-per-item forms)
  const bool big = nstreams * nsteps > 65536 && !env_untiled;
  if (big) {
    const long tiles = (nsteps + VAMD_ENV_TJ - 1) / VAMD_ENV_TJ;
    hipLaunchKernelGGL(k_env_amp_tiled, dim3((unsigned)(nsc * tiles)), dim3(8 * VAMD_ENV_TJ), 0, s, E, nsc, nsteps, states, ch,
                       near, raw, amp);
  } else {
    const long t = nsc * nsteps * 8;
    hipLaunchKernelGGL(k_env_amp, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, E, nsc, nsteps, states, ch,
                       near, raw, amp);
  }
  if (nstreams * nsteps <= 65536)
    hipLaunchKernelGGL(k_env_bits, dim3((unsigned)((nstreams * nsteps * 16 + 255) / 256)), dim3(256), 0, s, E, ch, nstreams, nsteps,
                       amp, bits);
  else if (!env_untiled) {
    const long items = nstreams * ((nsteps + 63) / 64);
    hipLaunchKernelGGL(k_env_bits_tiled, dim3((unsigned)((items + 3) / 4)), dim3(256), (size_t)4 * ch * VAMD_ENV_BROWS * 9 * 4, s, E, ch,
                       nstreams, nsteps, amp, bits);
  } else
    hipLaunchKernelGGL(k_env_bits_batch, dim3((unsigned)((nstreams * nsteps + 255) / 256)), dim3(256), 0, s, E, ch, nstreams,
                       nsteps, amp, bits);
  hipLaunchKernelGGL(k_env_walk, dim3((unsigned)nstreams), dim3(64), 0, s, ch, nstreams, nsteps, bits, near, amp,
                     states, ret);
  HIP_TRY(c, hipGetLastError());
  return VAMD_OK;
}

int vamd_envelope_search_batch(vamd_ctx *c, const float *pcm, long stream_stride, long channel_stride, long nstreams,
                               long nsteps, vamd_envelope_state *states, unsigned char *ret) {
  return envelope_search_batch(c, pcm, stream_stride, channel_stride, nstreams, nsteps, states, ret, c ? c->d_bad + 1 : nullptr);
}

int vamd_envelope_search(vamd_ctx *c, const float *const *pcm, long nsteps, vamd_envelope_state *state,
                         unsigned char *ret) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (nsteps < 0) return fail(c, VAMD_EINVAL, "negative step count");
  if (nsteps == 0) return VAMD_OK;
  if (!pcm || !state || !ret) return fail(c, VAMD_EINVAL, "null pcm / state / ret");
  if (c->K.fail_envelope_after >= 0) {  // (test knob: a GPU failure under the detector, for the binding's error path)
    static std::atomic<long> calls{0};
    if (calls.fetch_add(1) >= c->K.fail_envelope_after) return fail(c, VAMD_EFAULT, "injected failure (VAMD_FAIL_ENVELOPE_AFTER)");
  }
  const int ch = c->B.channels, n = c->B.env.mdct.n, step = c->B.env.searchstep;
  const long len = (nsteps - 1) * step + n;  // samples per channel the steps read
  // [pcm | state | bad (one word, zero on the way up) | ret]
  const size_t o_pcm = 0, o_state = ((size_t)ch * len * 4 + 15) & ~(size_t)15,
               o_bad = o_state + ((sizeof(vamd_envelope_state) + 15) & ~(size_t)15), o_ret = o_bad + 16,
               total = o_ret + (((size_t)nsteps + 15) & ~(size_t)15);
  if (c->h_stage_bytes < total) {
    if (c->h_stage) HIP_TRY(c, hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->h_stage_bytes = 0;
    HIP_TRY(c, hipHostMalloc(&c->h_stage, total, hipHostMallocDefault));
    c->h_stage_bytes = total;
  }
  void *dv;
  int r = ws_get(c, 0, vamd_ctx::WS_ENV_STAGE, total, &dv);
  if (r) return r;
  unsigned char *hs = (unsigned char *)c->h_stage, *ds = (unsigned char *)dv;
  for (int i = 0; i < ch; i++) {
    if (!pcm[i]) return fail(c, VAMD_EINVAL, "null channel pointer");
    memcpy(hs + o_pcm + (size_t)i * len * 4, pcm[i], (size_t)len * 4);
  }
  memcpy(hs + o_state, state, sizeof(*state));
  memset(hs + o_bad, 0, 16);
  hipStream_t s = c->stream;
  HIP_TRY(c, hipMemcpyAsync(ds, hs, o_ret, hipMemcpyHostToDevice, s));
  r = envelope_search_batch(c, (const float *)(ds + o_pcm), (long)ch * len, len, 1, nsteps,
                            (vamd_envelope_state *)(ds + o_state), ds + o_ret, (unsigned int *)(ds + o_bad));
  if (r) return r;
  HIP_TRY(c, hipMemcpyAsync(hs + o_state, ds + o_state, total - o_state, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  if (*(const unsigned int *)(hs + o_bad))  // (the state is left as it was: the stream is over for this caller)
    return fail(c, VAMD_ENONFINITE, "input outside the domain: a non-finite sample (include/vorbis_amd.h, Input domain)");
  memcpy(state, hs + o_state, sizeof(*state));
  memcpy(ret, hs + o_ret, (size_t)nsteps);
  return VAMD_OK;
}

int vamd_plan_streams(vamd_ctx *c, const float *pcm, long stream_stride, long channel_stride, long nstreams, long nsamples,
                      vamd_envelope_state *states, vamd_stream_plan *plan) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (!plan) return fail(c, VAMD_EINVAL, "null plan");
  memset(plan, 0, sizeof(*plan));
  if (nstreams < 0 || nsamples < 0) return fail(c, VAMD_EINVAL, "negative stream / sample count");
  if (nstreams == 0) return VAMD_OK;
  if (!pcm || !states) return fail(c, VAMD_EINVAL, "null pcm / states");
  if ((stream_stride | channel_stride) & 3) return fail(c, VAMD_EINVAL, "stream / channel strides must be multiples of 4 samples");
  if (nstreams > 0x3fffffffL || nsamples > 0x3fffffffL) return fail(c, VAMD_EINVAL, "too many streams / samples for one plan");
  const EnvP &E = c->B.env;
  BlockoutP B;
  B.bs[0] = c->B.bs[0];
  B.bs[1] = c->B.bs[1];
  B.searchstep = E.searchstep;
  B.nsamples = nsamples;
  B.nsteps = nsamples / E.searchstep - VAMD_VE_WIN;  // the steps _ve_envelope_search takes with this much data (lib/envelope.c:223-224)
  if (B.nsteps < 0) B.nsteps = 0;
  B.maxblocks = (int)(nsamples / (B.bs[0] / 2)) + 2;  // a block advances the stream by at least blocksizes[0]/2
  plan->nstreams = nstreams;
  void *v_flags, *v_blocks, *v_counts, *v_base;
  int r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_FLAGS, (size_t)nstreams * (B.nsteps ? B.nsteps : 1), &v_flags))) return r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_BLOCKS, (size_t)nstreams * B.maxblocks * sizeof(PlannedBlock), &v_blocks))) return r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_COUNTS, (size_t)nstreams * 2 * sizeof(int), &v_counts))) return r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_BASE, (size_t)(3 * nstreams + 1) * sizeof(long long), &v_base))) return r;
  if (B.nsteps && (r = vamd_envelope_search_batch(c, pcm, stream_stride, channel_stride, nstreams, B.nsteps, states, (unsigned char *)v_flags)))
    return r;
  hipStream_t s = c->stream;
  const size_t plan_lds = (size_t)((B.nsteps + 4 + 15) & ~15L);
  if (plan_lds > c->lds_per_block) return fail(c, VAMD_EINVAL, "streams too long for one plan (their marks must fit a workgroup's LDS)");
  // (above the default 64 KB of dynamic LDS the launch needs the opt-in, and a launch that fails leaves counts[] --
  // which sizes everything below -- uninitialised: hence the checks straight after it)
  HIP_TRY(c, hipFuncSetAttribute((const void *)k_plan_streams, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_per_block));
  HIP_TRY(c, hipMemsetAsync(v_counts, 0, (size_t)nstreams * 2 * sizeof(int), s));
  hipLaunchKernelGGL(k_plan_streams, dim3((unsigned)nstreams), dim3(64), plan_lds, s, B, nstreams,
                     (const unsigned char *)v_flags, (PlannedBlock *)v_blocks, (int *)v_counts);
  HIP_TRY(c, hipGetLastError());
  std::vector<int> counts((size_t)nstreams * 2);
  HIP_TRY(c, hipMemcpyAsync(counts.data(), v_counts, counts.size() * sizeof(int), hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  for (long i = 0; i < nstreams; i++)
    if (counts[2 * i] < 0 || counts[2 * i + 1] < 0 || (long)counts[2 * i] + counts[2 * i + 1] > B.maxblocks)
      return fail(c, VAMD_EFAULT, "stream plan: a block count outside its bound (the planning kernel did not run to completion)");
  // [2s + W] then start[nstreams + 1]; pinned and the context's own, so that its upload needs no wait: the next plan on this
  // context cannot write it before its own count read-back, which is queued behind the upload, has come home
  if (c->h_plan_bytes < ((size_t)3 * nstreams + 1) * sizeof(long long)) {
    if (c->h_plan) HIP_TRY(c, hipHostFree(c->h_plan));
    c->h_plan = nullptr;
    c->h_plan_bytes = 0;
    HIP_TRY(c, hipHostMalloc(&c->h_plan, ((size_t)3 * nstreams + 1) * sizeof(long long), hipHostMallocDefault));
    c->h_plan_bytes = ((size_t)3 * nstreams + 1) * sizeof(long long);
  }
  long long *base = (long long *)c->h_plan;
  long long tot[2] = {0, 0}, all = 0;
  for (long i = 0; i < nstreams; i++) {
    base[2 * i] = tot[0];
    base[2 * i + 1] = tot[1];
    base[2 * nstreams + i] = all;
    tot[0] += counts[2 * i];
    tot[1] += counts[2 * i + 1];
    all += counts[2 * i] + counts[2 * i + 1];
  }
  base[3 * nstreams] = all;
  if (tot[0] > 0x3fffffffLL || tot[1] > 0x3fffffffLL) return fail(c, VAMD_EINVAL, "plan too large: order[] holds 30-bit indices");
  HIP_TRY(c, hipMemcpyAsync(v_base, base, ((size_t)3 * nstreams + 1) * sizeof(long long), hipMemcpyHostToDevice, s));
  // descriptor arrays: per class lW, nW, blocktype (int32) and src (int64); then order
  void *v_desc, *v_order;
  const size_t per[2] = {(size_t)tot[0], (size_t)tot[1]};
  const size_t desc_bytes = (per[0] + per[1]) * (3 * sizeof(int) + sizeof(long long)) + 64;
  if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_DESC, desc_bytes, &v_desc))) return r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_ORDER, (size_t)(all ? all : 1) * sizeof(int), &v_order))) return r;
  PlanOut O;
  long long *p64 = (long long *)v_desc;  // the 8-byte arrays first (alignment)
  O.src[0] = p64;
  O.src[1] = p64 + per[0];
  int *p32 = (int *)(p64 + per[0] + per[1]);
  for (int W = 0; W < 2; W++) {
    O.lW[W] = p32, p32 += per[W];
    O.nW[W] = p32, p32 += per[W];
    O.bt[W] = p32, p32 += per[W];
  }
  O.order = (int *)v_order;
  hipLaunchKernelGGL(k_plan_emit, dim3((unsigned)nstreams), dim3(64), 0, s, B, nstreams, stream_stride,
                     (const PlannedBlock *)v_blocks, (const int *)v_counts, (const long long *)v_base,
                     (const long long *)v_base + 2 * nstreams, O);
  HIP_TRY(c, hipGetLastError());
  for (int W = 0; W < 2; W++) {
    plan->nblocks[W] = tot[W];
    plan->lW[W] = O.lW[W];
    plan->nW[W] = O.nW[W];
    plan->blocktype[W] = O.bt[W];
    plan->src[W] = (const int64_t *)O.src[W];
  }
  plan->order = O.order;
  plan->stream_start = (const int64_t *)((const long long *)v_base + 2 * nstreams);
  return VAMD_OK;
}

int vamd_gather_blocks(vamd_ctx *c, const vamd_stream_plan *plan, int W, const float *pcm, long channel_stride,
                       float *pcm_blocks) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (!plan || (W != 0 && W != 1)) return fail(c, VAMD_EINVAL, "null plan / bad size class");
  const long nb = plan->nblocks[W];
  if (nb == 0) return VAMD_OK;
  if (!pcm || !pcm_blocks) return fail(c, VAMD_EINVAL, "null pcm / pcm_blocks");
  if (channel_stride & 3) return fail(c, VAMD_EINVAL, "channel stride must be a multiple of 4 samples");
  if (((uintptr_t)pcm | (uintptr_t)pcm_blocks) & 15) return fail(c, VAMD_EINVAL, "pcm / pcm_blocks must be 16-byte aligned");
  const int ch = c->B.channels, n = c->B.bs[W];
  const long total = nb * ch * (n / 4);
  const long blocks = (total + 255) / 256;
  const long cap = (long)c->num_cus * 16;
  hipLaunchKernelGGL(k_gather_blocks, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, c->stream, ch, n, nb,
                     (const long long *)plan->src[W], channel_stride, pcm, pcm_blocks);
  HIP_TRY(c, hipGetLastError());
  return VAMD_OK;
}

int vamd_plan_fetch(vamd_ctx *c, const vamd_stream_plan *plan, int32_t *const lW[2], int32_t *const nW[2],
                    int32_t *const blocktype[2], int64_t *const src[2], int32_t *order, int64_t *stream_start) {
  DeviceGuard dev_guard(c);
  if (!c) return VAMD_EINVAL;
  if (!plan) return fail(c, VAMD_EINVAL, "null plan");
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int W = 0; W < 2; W++) {
    const size_t n = (size_t)plan->nblocks[W];
    if (!n) continue;
    if (lW && lW[W]) HIP_TRY(c, hipMemcpy(lW[W], plan->lW[W], n * 4, hipMemcpyDeviceToHost));
    if (nW && nW[W]) HIP_TRY(c, hipMemcpy(nW[W], plan->nW[W], n * 4, hipMemcpyDeviceToHost));
    if (blocktype && blocktype[W]) HIP_TRY(c, hipMemcpy(blocktype[W], plan->blocktype[W], n * 4, hipMemcpyDeviceToHost));
    if (src && src[W]) HIP_TRY(c, hipMemcpy(src[W], plan->src[W], n * 8, hipMemcpyDeviceToHost));
  }
  const size_t all = (size_t)(plan->nblocks[0] + plan->nblocks[1]);
  if (order && all) HIP_TRY(c, hipMemcpy(order, plan->order, all * 4, hipMemcpyDeviceToHost));
  if (stream_start && plan->nstreams) HIP_TRY(c, hipMemcpy(stream_start, plan->stream_start, (size_t)(plan->nstreams + 1) * 8, hipMemcpyDeviceToHost));
  return VAMD_OK;
}

}  // extern "C"
