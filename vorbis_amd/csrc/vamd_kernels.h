// vamd_kernels.h -- the __global__ shells of libvorbis_amd.so: thin kernels around the wave-level bodies in k_*.h (one
// section per stage, in pipeline order; then the block-switching detector, stream control, the calibration copy).  Part of
// the library's single translation unit: included by vamd_hip.hip, once, after the body headers.
#pragma once
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
// A stream's ampmax chain (lib/psy.c:837-848 _vp_ampmax_decay, lib/mapping0.c:346,576): block k receives the running
// value decayed by its own half-length, and hands on the larger of that and its channels' spectral peaks.  The float
// additions make the order part of the result, so the chain is walked in order -- but by a WAVE: 64 links' block
// indices, decays and peaks are fetched at once (one lane each: the thread-per-stream form paid two dependent trips to
// memory per link, 1.5 us each, 0.39 ms for a stream of 250 blocks), the walk itself runs on registers (v_readlane with
// the unrolled link number), and each lane keeps and stores its own link's two values.
// order == nullptr: a stream of one size class, link k is block k.  first_given: `amp` is what link k0 receives.
VAMD_DEV float ampmax_chain_wave(int ch, long long k0, long long k1, const int *__restrict__ order, float secs0, float secs1,
                                 float att, float amp, const float *__restrict__ local0, const float *__restrict__ local1,
                                 float *__restrict__ in0, float *__restrict__ in1, float *__restrict__ glob0,
                                 float *__restrict__ glob1, int first_given) {
  for (long long base = k0; base < k1; base += 64) {
    const int cnt = (int)((k1 - base) < 64 ? (k1 - base) : 64);
    int W = 0;
    long b = 0;
    float dec = 0.f, peak = VAMD_NEGINF;
    if (LANE < cnt) {
      const long long k = base + LANE;
      if (order) {
        const int o = order[k];
        W = (o >> 30) & 1;
        b = o & 0x3fffffff;
      } else {
        b = (long)k;
      }
      dec = (W ? secs1 : secs0) * att;
      const float *loc = (W ? local1 : local0) + b * ch;
      for (int c = 0; c < ch; c++) {  // (the larger of `amp` and every peak, whatever the order; a NaN peak never wins)
        const float l = loc[c];
        if (l > peak) peak = l;
      }
    }
    const bool keep0 = first_given && base == k0;
    float my_in = 0.f, my_out = 0.f;
#pragma unroll
    for (int j = 0; j < 64; j++) {
      if (j >= cnt) break;
      const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dec), j));
      const float lj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, peak), j));
      if (!(j == 0 && keep0)) {
        amp += dj;
        if (amp < -9999) amp = -9999;
      }
      const float a_in = amp;
      if (lj > amp) amp = lj;
      if (LANE == j) {
        my_in = a_in;
        my_out = amp;
      }
    }
    if (LANE < cnt) {
      (W ? in1 : in0)[b] = my_in;
      (W ? glob1 : glob0)[b] = my_out;
    }
  }
  return amp;
}

__global__ __launch_bounds__(64) void k_ampmax_stream(int ch, long nblocks, float secs, float att, float state,
                                                      const float *__restrict__ local_ampmax,
                                                      float *__restrict__ ampmax_in, float *__restrict__ ampmax_glob) {
  (void)ampmax_chain_wave(ch, 0, nblocks, nullptr, secs, secs, att, state, local_ampmax, local_ampmax, ampmax_in, ampmax_in,
                          ampmax_glob, ampmax_glob, 0);
}

// a stream that mixes both size classes: order[k] = W << 30 | index inside W's batch
__global__ __launch_bounds__(64) void k_ampmax_stream_mixed(int ch, long ntotal, const int *__restrict__ order, float secs0,
                                                            float secs1, float att, float state,
                                                            const float *__restrict__ local0,
                                                            const float *__restrict__ local1, float *__restrict__ in0,
                                                            float *__restrict__ in1, float *__restrict__ glob0,
                                                            float *__restrict__ glob1, float *__restrict__ state_out,
                                                            int first_given) {
  const float amp = ampmax_chain_wave(ch, 0, ntotal, order, secs0, secs1, att, state, local0, local1, in0, in1, glob0, glob1,
                                      first_given);
  if (LANE == 0) *state_out = amp;
}

// many streams at once: wave s walks order[start[s] .. start[s+1]) with its own running state
__global__ __launch_bounds__(64) void k_ampmax_streams_mixed(int ch, long nstreams, const long long *__restrict__ start,
                                                             const int *__restrict__ order, float secs0, float secs1,
                                                             float att, float *__restrict__ states,
                                                             const float *__restrict__ local0,
                                                             const float *__restrict__ local1, float *__restrict__ in0,
                                                             float *__restrict__ in1, float *__restrict__ glob0,
                                                             float *__restrict__ glob1) {
  const long sidx = blockIdx.x;
  if (sidx >= nstreams) return;
  const float amp = ampmax_chain_wave(ch, start[sidx], start[sidx + 1], order, secs0, secs1, att, states[sidx], local0, local1,
                                      in0, in1, glob0, glob1, 0);
  if (LANE == 0) states[sidx] = amp;
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
// 9 us + k_tone_chase_wave 25 us -> this kernel's 24, measured: profiles/r04_block_path.txt; what is left is the
// lock step -- 36 steps at the slowest lane's pop count).  Every libvorbisenc setup has eight lines per window; others
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
//   chunked: a stereo type-2 residue whose vectors tile runs of eight values (ResP::chunked), at most a run a thread: the
//            search out of registers (residue_team_chunks), the search's tables where the work vector would be
__global__ __launch_bounds__(64 * VAMD_RES_WAVES) void k_residue(ResP R, ChMap cm, int sm, int ent_row, int nblobs, DescP d, int ch, int n2,
                                                const int *__restrict__ iwork, const int *__restrict__ nonzero,
                                                int *__restrict__ res_class, unsigned short *__restrict__ res_entries,
                                                int *__restrict__ res_count, unsigned char *__restrict__ res_books, int chunked) {
  const long u = blockIdx.x;
  int *work = (int *)vamd_smem;                 // [bundle*n2] (chunked: the tables, [R.fast_ints])
  int *cls = work + (chunked ? R.fast_ints : R.bundle * n2);  // [VAMD_RES_CLASS_STRIDE]
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
  if (chunked)
    over = residue_team_chunks(R, ip[0], ip[1], nz[0] | nz[1], work, cls, off, res_class + u * (cm.submaps * VAMD_RES_CLASS_STRIDE) + R.cls_base,
                               res_entries + u * (long)ent_row + R.ent_base, res_count + (u * cm.submaps + sm) * 2, pc,
                               res_books ? res_books + u * (long)ent_row + R.ent_base : nullptr);
  else
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

// The same stage for a stereo type-2 residue whose vectors tile runs of eight values (ResP::chunked): persistent waves, a
// unit each at a time, the search out of registers (residue_wave_chunks, k_residue.h).  LDS: the search's tables once
// per workgroup, then per wave cls [partvals] and off [stages * partvals + 1].
#define VAMD_RESC_WAVES 4
__global__ __launch_bounds__(64 * VAMD_RESC_WAVES, 8) void k_residue_chunks(ResP R, ChMap cm, int sm, int ent_row, int nblobs, DescP d, int ch, int n2,
                                                       long units, const int *__restrict__ iwork, const int *__restrict__ nonzero,
                                                       int *__restrict__ res_class, unsigned short *__restrict__ res_entries,
                                                       int *__restrict__ res_count, unsigned char *__restrict__ res_books) {
  int *tab = (int *)vamd_smem;
  for (int i = threadIdx.x; i < (R.fast_ints >> 2); i += blockDim.x) ((I4 *)tab)[i] = ((const I4 *)R.fast)[i];
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int per_wave = (R.partvals + R.nstages * R.partvals + 1 + 3) & ~3;
  int *cls = tab + R.fast_ints + wave * per_wave, *off = cls + R.partvals;
  int c0 = -1, c1 = -1;  // the bundle's two channels
  for (int c = 0; c < ch; c++)
    if (cm.sub[c] == sm) {
      if (c0 < 0) c0 = c;
      else if (c1 < 0) c1 = c;
    }
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 72 : nullptr);
  for (long u = (long)blockIdx.x * VAMD_RESC_WAVES + wave; u < units; u += (long)gridDim.x * VAMD_RESC_WAVES) {
    const int nz = nonzero[u * ch + c0] | nonzero[u * ch + c1];
    const unsigned over = residue_wave_chunks(R, iwork + (u * ch + c0) * n2, iwork + (u * ch + c1) * n2, nz, tab, cls, off,
                                                      res_class + u * (cm.submaps * VAMD_RES_CLASS_STRIDE) + R.cls_base,
                                                      res_entries + u * (long)ent_row + R.ent_base, res_count + (u * cm.submaps + sm) * 2, pc,
                                                      res_books ? res_books + u * (long)ent_row + R.ent_base : nullptr);
    // the input domain's integer edge, the search's half (k_residue.h)
    const long blk = u / nblobs;
    const int any0 = wave_any((int)(over & 1u)), any1 = wave_any((int)(over & 2u));
    if (LANE == 0 && any0) flag_range(d, blk * ch + c0);
    if (LANE == 0 && any1) flag_range(d, blk * ch + c1);
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

// The same for a batch of a mode with ONE submap (mono, stereo): persistent waves, a packet each at a time.  k_pack sets a
// packet's tables up per packet -- the books' sizes and offsets, the floor's class tables, the residue's (class, stage)
// rows: 14 k of a packet's 64 k cycles, and its 9.4 KB of LDS hold a CU to seventeen one-wave workgroups.  Here a
// workgroup's four waves share one copy made once, every wave keeps its own ring (all zero between packets: ring_flush
// hands the slots back) and offsets arrays, and the emission offsets come out of the LDS rows (pack_residue's rtab).
//   lds_ints: ints of cls + off + info per wave (slots + 2 * stages * slots + 1, rounded up to 4)
#define VAMD_PKW_WAVES 4
__global__ __launch_bounds__(64 * VAMD_PKW_WAVES, 6) void k_pack_waves(PackP K, FloorP F0, ResP R0, ChMap cm, int ent_row, int lds_ints,
                                                                  DescP d, int ch, int W, int nblobs, long units,
                                                                  const int *__restrict__ posts, const int *__restrict__ wrapped,
                                                                  const int *__restrict__ post_valid, const int *__restrict__ res_class,
                                                                  const unsigned short *__restrict__ res_entries,
                                                                  const unsigned char *__restrict__ res_books,
                                                                  const int *__restrict__ res_count, unsigned *__restrict__ packets,
                                                                  int stride_words, int *__restrict__ packet_bits) {
  int *tabs = (int *)vamd_smem;                          // [VAMD_PK_FTAB_INTS + 3 * nbooks], PackTabs
  int *rtab = tabs + ((VAMD_PK_FTAB_INTS + 3 * K.nbooks + 3) & ~3);  // [R0.fast_ints]
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int *mine = rtab + R0.fast_ints + wave * (VAMD_PK_RING + VAMD_POSTS_STRIDE + lds_ints);
  int *ring = mine;                                      // [VAMD_PK_RING]
  int *outv = ring + VAMD_PK_RING;                       // [VAMD_POSTS_STRIDE]
  int *cls = outv + VAMD_POSTS_STRIDE;                   // [slots], then off [stages*slots + 1], info [stages*slots]
  int *off = cls + R0.slots;
  int *info = off + R0.nstages * R0.slots + 1;
  PackTabs T;
  T.at(tabs);
  for (int i = threadIdx.x; i < (R0.fast_ints >> 2); i += blockDim.x) ((I4 *)rtab)[i] = ((const I4 *)R0.fast)[i];
  WAVE_FOR(i, VAMD_PK_RING) ring[i] = 0;
  if (wave == 0) {
    pack_book_table(K, T);
    pack_floor_table(*K.ftab[0], 0, T);
  }
  __syncthreads();
  T.floor_of = 0;  // (every wave: the shared copy holds submap 0's floor)
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 72 : nullptr);
  for (long u = (long)blockIdx.x * VAMD_PKW_WAVES + wave; u < units; u += (long)gridDim.x * VAMD_PKW_WAVES) {
    const long blk = u / nblobs;
    pack_block_body(K, T, F0, F0, R0, R0, cm, ch, W, d_lW(d, blk), d_nW(d, blk), posts + u * ch * VAMD_POSTS_STRIDE,
                    wrapped ? wrapped + u * ch * VAMD_POSTS_STRIDE : nullptr, post_valid + u * ch,
                    res_class + u * (cm.submaps * VAMD_RES_CLASS_STRIDE), res_entries + u * (long)ent_row,
                    res_books ? res_books + u * (long)ent_row : nullptr, res_count + u * cm.submaps * 2, ring, outv, cls, off, info,
                    packets + u * (long)stride_words, stride_words, packet_bits + u, pc, rtab);
  }
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
  int *rtab = share + 4;                                 // [R0.fast_ints]: submap 0's (class, stage) rows (pack_residue's rtab)
  PhaseClock pc;
  pc.start(d.dbg ? d.dbg + 72 : nullptr);
  PackTabs T;
  T.at(tabs);
  for (int i = threadIdx.x; i < R0.fast_ints; i += blockDim.x) rtab[i] = R0.fast[i];
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
                   res_count + (u * cm.submaps + sm) * 2, cls, off, info, r, pc, sm == 0 ? rtab : nullptr);
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

// a wave takes VAMD_ENV_STEPS consecutive steps of one (stream, channel) at a time.  Eight (round 6): the 32-point groups of
// the 128-point MDCT -- a lane a group: the reference spells each of their butterflies out with its own constants -- then
// fill sixteen lanes instead of eight, and an item's bookkeeping is shared by twice the steps (profiles/r06_env_phases.txt:
// 2 875 cycles per four steps behind the fetch instead of 5 200).  What made that possible is where the next item's
// samples wait: not in sixteen registers per lane held across the item (the kernel then wanted more than 128), but in
// LDS -- fetched there by global_load_lds, which needs no register for the data: the steps of an item overlap by half, so
// its (steps + 1) * 64 samples are one contiguous run, a 256-byte chunk per instruction, lane l's float to word l.
#define VAMD_ENV_LOGS 3
#define VAMD_ENV_STEPS (1 << VAMD_ENV_LOGS)
#define VAMD_ENV_WAVES 4
#define VAMD_ENV_STAGE_FLOATS ((VAMD_ENV_STEPS + 1) * 64)  // an item's samples: (steps - 1) * searchstep + 128, searchstep 64
__global__ __launch_bounds__(64 * VAMD_ENV_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_env_spectrum(EnvP E, int ch, long nstreams, long nsteps,
                                                                      const float *__restrict__ pcm, long stream_stride,
                                                                      long channel_stride, float *__restrict__ near,
                                                                      float *__restrict__ raw, unsigned int *bad,
                                                                      const long long *__restrict__ first_of, unsigned long long *dbg) {
  // (the wave number in a scalar register: an item's place -- stream, channel, first step -- is then scalar arithmetic)
  const int n = E.mdct.n, n2 = n >> 1, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // per wave: two staging buffers (the item at hand, the next one's on their way), then per step the transform's work buffer,
  // whose plain half takes the spectrum.  No windowed copy of the samples: the fold windows them on the way in.
  const int per_step = n2 + VAMD_PW_SIZE(n2);
  float *stage0 = (float *)vamd_smem + (size_t)wave * (2 * VAMD_ENV_STAGE_FLOATS + per_step * VAMD_ENV_STEPS);
  float *Wk = stage0 + 2 * VAMD_ENV_STAGE_FLOATS, *spec = Wk;
  PhaseClock pc;
  pc.start(dbg);  // (the transform's slot set: tools/env_profile.py)
  // the transform's tables out of LDS, staged once per workgroup (a wave lives for ~100 items): every twiddle of every
  // item used to be a trip to L1 with a 64-bit address formed in vector registers
  {
    float *ttrig = (float *)vamd_smem + (size_t)VAMD_ENV_WAVES * (2 * VAMD_ENV_STAGE_FLOATS + per_step * VAMD_ENV_STEPS);  // [n + n/4], then win [n], bitrev [n/4]
    float *twin = ttrig + n + n / 4;
    int *tbit = (int *)(twin + n);
    for (int i = threadIdx.x; i < n + n / 4; i += blockDim.x) ttrig[i] = E.mdct.trig[i];
    for (int i = threadIdx.x; i < n; i += blockDim.x) twin[i] = E.win[i];
    for (int i = threadIdx.x; i < n / 4; i += blockDim.x) tbit[i] = E.mdct.bitrev[i];
    __syncthreads();
    E.mdct.trig = ttrig;
    E.win = twin;
    E.mdct.bitrev = tbit;
  }
  const long groups = (nsteps + VAMD_ENV_STEPS - 1) / VAMD_ENV_STEPS, items = nstreams * ch * groups;
  auto where = [&](long it, long &sc, long &j, int &count) -> const float * {
    // (32-bit quotients: the launch checks items < 2^31 -- a 64-bit division by a run-time value is some eighty vector
    // instructions, two of them per item were a third of the kernel's)
    sc = (long)((unsigned)it / (unsigned)groups);
    j = (it - sc * groups) * VAMD_ENV_STEPS;
    const long s = (long)((unsigned)sc / (unsigned)ch);
    const int c = (int)(sc - s * ch);
    count = nsteps - j < VAMD_ENV_STEPS ? (int)(nsteps - j) : VAMD_ENV_STEPS;
    // (first_of: streams whose steps start at different samples -- the end-of-stream pass of streams of unequal length)
    return pcm + s * stream_stride + c * channel_stride + (first_of ? first_of[s] : 0) + j * E.searchstep;
  };
  // an item's samples into the staging buffer: chunk c = samples [64 c, 64 c + 64), count + 1 of them
  auto send_for = [&](const float *src, int count, float *stage) {
    typedef const __attribute__((address_space(1))) void *gptr;
    typedef __attribute__((address_space(3))) void *lptr;
#pragma unroll
    for (int c = 0; c <= VAMD_ENV_STEPS; c++)
      if (c <= count) __builtin_amdgcn_global_load_lds((gptr)(src + 64 * c + LANE), (lptr)(stage + 64 * c), 4, 0, 0);
  };
  const long stride = (long)gridDim.x * VAMD_ENV_WAVES;
  long it = (long)blockIdx.x * VAMD_ENV_WAVES + wave;
  long sc, j;
  int count;
  const float *src = nullptr;
  int cur = 0;  // which staging buffer holds the item at hand
  if (it < items) {
    src = where(it, sc, j, count);
    send_for(src, count, stage0);
  }
  for (; it < items; it += stride) {
    long sc2 = 0, j2 = 0;
    int count2 = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this item's samples have landed (and nothing else is in flight)
    WAVE_SYNC();
    if (it + stride < items) {  // the next item's into the other buffer, on their way while this one is transformed
      const float *src2 = where(it + stride, sc2, j2, count2);
      send_for(src2, count2, stage0 + (cur ^ 1) * VAMD_ENV_STAGE_FLOATS);
    }
    env_spectrum_wave<VAMD_ENV_LOGS, true>(E, stage0 + cur * VAMD_ENV_STAGE_FLOATS, count, nullptr, Wk, spec,
                                           near + sc * (VAMD_VE_NEAR_HIST + nsteps) + VAMD_VE_NEAR_HIST + j,
                                           raw + (sc * nsteps + j) * VAMD_VE_SPREAD, pc, bad, (const EnvSamples<VAMD_ENV_LOGS> *)nullptr,
                                           n2 + VAMD_PW_SIZE(n2));
    sc = sc2, j = j2, count = count2;
    cur ^= 1;
  }
  pc.flush();
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
// (count_of: stream s takes only its first count_of[s] <= nsteps steps -- streams of unequal length in one launch; the
// state it leaves is the state after exactly those)
__global__ __launch_bounds__(64) void k_env_walk(int ch, long nstreams, long nsteps, const uint32_t *__restrict__ bits,
                                                 const float *__restrict__ near, const float *__restrict__ amp,
                                                 vamd_envelope_state *__restrict__ st,
                                                 unsigned char *__restrict__ ret, const int *__restrict__ count_of) {
  const long s = blockIdx.x;
  const long mine = count_of ? (long)count_of[s] : nsteps;
  const int stretch = env_walk_wave(bits + s * nsteps, mine, st[s].stretch, ret + s * nsteps);
  if (LANE == 0) {
    st[s].stretch = stretch;
    st[s].steps += mine;
  }
  for (int c = 0; c < ch; c++) {
    const float *nt = near + (s * ch + c) * (VAMD_VE_NEAR_HIST + nsteps) + mine;  // the last NEAR_HIST entries
    WAVE_FOR(i, VAMD_VE_NEAR_HIST) st[s].near_hist[c][i] = nt[i];
    const float *at = amp + ((s * ch + c) * (VAMD_VE_AMP_HIST + nsteps) + mine) * 8;
    WAVE_FOR(i, VAMD_VE_AMP_HIST * 8) st[s].amp_hist[c][i >> 3][i & 7] = at[i];
  }
}

// ---- device-resident stream control (k_blockout.h) ----------------------------------------------------
// one wave per stream: the lanes turn the stream's flags into its mark bytes in LDS (coalesced reads, ve->mark[] as
// mark_at defines it), then one lane does the walk out of LDS -- a dependent chain of a few thousand steps that would
// otherwise pay a trip to HBM at each of them
// flags: row s of `flags` holds the steps [0, split) at stride stride1, row s of `flags2` the steps from split on at stride2 (the
// steps of a stream's end-of-stream padding are taken in a second detector pass, vamd_plan_streams_whole; split == B.nsteps:
// one array).  pending != null: a dry run -- nothing is emitted, pending[s] = centerW of the block the walk stopped in front of.
// geo != null: streams of unequal length -- geo[s] overrides B.nsamples, B.eof, B.nsteps and split for stream s.
struct PlanGeo {
  long long nsamples, eof;
  int nsteps, split;
};
__global__ __launch_bounds__(64) void k_plan_streams(BlockoutP B, long nstreams, const unsigned char *__restrict__ flags, long stride1,
                                                     long split, const unsigned char *__restrict__ flags2, long stride2,
                                                     PlannedBlock *__restrict__ blocks, int *__restrict__ counts,
                                                     long long *__restrict__ pending, const PlanGeo *__restrict__ geo,
                                                     int with_eof) {
  unsigned char *marks = (unsigned char *)vamd_smem;  // [nsteps + 4]
  const long s = blockIdx.x;
  const long lds_steps = B.nsteps;  // (the launch's LDS holds this many marks + 4)
  if (geo) {
    B.nsamples = geo[s].nsamples;
    B.eof = geo[s].eof;
    B.nsteps = geo[s].nsteps;
    split = geo[s].split;
    if (!with_eof) {  // the dry run: as far as the real samples go
      B.nsamples = geo[s].eof;
      B.eof = 0;
      B.nsteps = split;
    }
  }
  const long last = blockout_steps(B);
  const unsigned char *f = flags + s * stride1, *f2 = flags2 + s * stride2;
  auto flag = [&](long p) -> int { return p < split ? f[p] : f2[p - split]; };
  for (long p = threadIdx.x; p < lds_steps + 4; p += 64) {
    int m = 0;  // mark_at(), over the two pieces
    if (p < last) {
      if (p >= 1) m |= flag(p - 1) & 1;
      m |= flag(p) & 3;
      if (p + 1 < last) m |= flag(p + 1) & 2;
    }
    marks[p] = (unsigned char)(m != 0);
  }
  __syncthreads();
  int n0 = 0, n1 = 0;
  long pc = 0;
  plan_stream(B, marks, pending ? nullptr : blocks + s * B.maxblocks, &n0, &n1, &pc);  // (the whole wave: it looks at 64 marks at a time)
  if (threadIdx.x == 0) {
    if (pending) pending[s] = pc;
    else {
      counts[2 * s] = n0;
      counts[2 * s + 1] = n1;
    }
  }
}

// ---- the two ends of a stream (k_lpc.h): what vorbis_analysis_wrote() extrapolates on the host in the reference ----
// a wave per (stream, channel).  x = the channel's buffer: x[0, head) the (zero) space in front of the first sample,
// x[head, head + n) the first n real samples.  lib/block.c:417-458.
__global__ __launch_bounds__(64) void k_lpc_head(int ch, long nstreams, float *__restrict__ pcm, long stream_stride,
                                                 long channel_stride, int head, int n, const PlanGeo *__restrict__ geo) {
  const long sc = blockIdx.x, s = sc / ch;
  if (geo) {  // streams of unequal length: as many of the first samples as this stream has, up to n
    const long long frames = geo[s].eof - head;
    if (frames < n) n = (int)frames;
  }
  if (n <= 32) return;  // "if(v->pcm_current-v->centerW>order*2)", lib/block.c:427
  const int c = (int)(sc - s * ch);
  float *x = pcm + s * stream_stride + (long)c * channel_stride;
  double *aut = (double *)vamd_smem;                      // [2 * 16 + 1], padded to 80
  float *coeff = (float *)(aut + 80);                     // [32]
  float *work = coeff + VAMD_LPC_MAX_ORDER;               // [n + head]: the stream reversed, then what precedes it
  const int order = 16;
  WAVE_FOR(j, n) work[j] = x[head + n - 1 - j];
  WAVE_SYNC();
  lpc_from_data(work, n, order, aut, coeff);
  lpc_predict(coeff, work + n - order, order, work + n, head);
  WAVE_FOR(i, head) x[head - 1 - i] = work[n + i];
}
// the end: x[eof, eof + pad) from the last min(eof - start, bs1) samples before eof, start = where the reference's
// buffer begins when the stream is closed (pending centre - bs1/2).  lib/block.c:474-512.
__global__ __launch_bounds__(64) void k_lpc_tail(int ch, long nstreams, float *__restrict__ pcm, long stream_stride,
                                                 long channel_stride, long eof, int bs1, int pad,
                                                 const long long *__restrict__ pending, const PlanGeo *__restrict__ geo) {
  const long sc = blockIdx.x, s = sc / ch;
  if (geo) eof = (long)geo[s].eof;
  const int c = (int)(sc - s * ch);
  float *x = pcm + s * stream_stride + (long)c * channel_stride;
  double *aut = (double *)vamd_smem;                      // [2 * 32 + 1], padded to 80
  float *coeff = (float *)(aut + 80);                     // [32]
  float *data = coeff + VAMD_LPC_MAX_ORDER;               // [bs1]
  float *out = data + bs1;                                // [pad]
  const int order = 32;
  long start = (long)pending[s] - bs1 / 2;
  if (start < 0) start = 0;
  const long have = eof - start;                          // v->eofflag in the reference's (shifted) coordinates
  if (have > order * 2) {
    const int n = have < bs1 ? (int)have : bs1;
    WAVE_FOR(i, n) data[i] = x[eof - n + i];
    WAVE_SYNC();
    lpc_from_data(data, n, order, aut, coeff);
    lpc_predict(coeff, data + n - order, order, out, pad);
    WAVE_FOR(i, pad) x[eof + i] = out[i];
  } else {
    WAVE_FOR(i, pad) x[eof + i] = 0.f;                     // "not enough data to extrapolate ... zeroes will do"
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
