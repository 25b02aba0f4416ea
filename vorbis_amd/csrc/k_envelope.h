// k_envelope.h -- the block-switching detector: _ve_amp and the step loop of
// _ve_envelope_search (reference lib/envelope.c:89-215, :217-262); SURVEY.md 8f rank 1.
//
// The reference runs one detector step every `searchstep` (64) samples per channel: a
// sin^2-windowed 128-point MDCT, a near-DC smoother, a 32-value dB spread, seven band
// amplitudes, and pre-/post-echo triggers against the band's recent history.  It keeps
// all of that as running state, so it looks serial -- but only one integer is:
//
//  * the near-DC accumulator is "regularly refreshed from scratch" (lib/envelope.c:127-137):
//    its value at step J depends on the terms of steps J-29 .. J only, in a fixed order, so
//    every step can replay its own <= 44 adds (env_decay);
//  * the band history is a plain ring of the last 17 amplitudes: step J reads steps
//    J-1 .. J-13 (lib/envelope.c:171-187), which are outputs of the parallel stage;
//  * ve->stretch (how far back the pre-echo window reaches, and the penalty) is the true
//    recurrence: it is reset by a trigger and otherwise counts up.  It takes 13 distinct
//    values of stretch/2, so the trigger bits are computed for all 13 in parallel and a
//    single thread per stream then walks the steps picking 2 bits per step (env_walk).
//
// Stages (one launch each):
//   env_spectrum_wave  wave   per (stream, channel, step): window, MDCT, near-DC term, raw dB pairs
//   env_band_amp       thread per (stream, channel, step, band): decay replay, spread, band amplitude
//   env_trigger_bits   thread per (stream, step): 13 x {pre-echo, post-echo} bits over channels and bands
//   env_walk_wave      wave   per stream: the stretch recurrence -> ret flags (1|4 pre-echo, 2 post-echo)
// Series carry a history prefix (the previous call's tail, zeros at stream start == the
// reference's calloc'ed filter state) so a stream can be processed in calls of any size.
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"
#include "vorbis_amd.h"
#include "k_transform.h"

namespace vamd {

// VAMD_VE_NEAR_HIST (30: two refresh periods) and VAMD_VE_AMP_HIST (16 >= 1 + VE_MAXSTRETCH) are
// part of the state layout: include/vorbis_amd.h
#define VAMD_VE_SPREAD 32     // winlength/4 spread values (winlength == 128 is enforced at bind time)
#define VAMD_VE_HSTATES 13    // values of stretch/2: 0 .. VE_MAXSTRETCH

// window + MDCT + the two per-step series that need the spectrum (lib/envelope.c:110-124,148-152),
// for 2^LOGS consecutive steps of one channel at a time (their transforms run side by side, see
// mdct_forward_wave): a 128-point transform alone would leave most of the wave idle.
//   pcm    HBM, the first step's samples; step t starts t*searchstep further on
//   count  steps actually wanted (<= 2^LOGS; the rest of the slots compute on zeros, unstored)
//   A LDS [n] per step, Wk LDS [n/2 + VAMD_PW_SIZE(n/2)] per step, spec LDS [n/2] per step
//   near_out [count] the near-DC terms `temp`;  raw_out [count][n/4]  todB(re^2+im^2)*.5f before limiting
// a lane's samples of 2^LOGS consecutive steps (lane l holds k = l + 64 m: step k >> 7, sample k & 127), fetched by the
// caller one item ahead of the transform that consumes them (k_env_spectrum); steps past `count` read as zeros
template <int LOGS>
struct EnvSamples {
  float v[(128 << LOGS) / 64];
};
template <int LOGS>
VAMD_DEV void env_fetch(EnvSamples<LOGS> &x, const float *__restrict__ pcm, int count, int searchstep) {
#pragma unroll
  for (int m = 0; m < (128 << LOGS) / 64; m++) {
    const int k = LANE + 64 * m, t = k >> 7, i = k & 127;
    x.v[m] = t < count ? pcm[t * searchstep + i] : 0.f;
  }
}

//   STAGED: `pcm` is the wave's staging buffer in LDS holding the item's samples as they came (k_env_spectrum: step t's
//           start t * searchstep floats in); no windowed copy is made, the fold windows on the way in (mdct_forward_wave's WIN),
//           and `A` is not used.  Steps past `count` then transform whatever the buffer holds; nothing of them is stored.
template <int LOGS, bool STAGED = false>
VAMD_DEV void env_spectrum_wave(const EnvP &E, const float *__restrict__ pcm, int count, float *A, float *Wk,
                                float *spec, float *__restrict__ near_out, float *__restrict__ raw_out,
                                PhaseClock &pc, unsigned int *bad = nullptr, const EnvSamples<LOGS> *pre = nullptr,
                                int spec_stride = 64) {
  // (the detector's transform is 128 points whatever the setup -- vamd_bind refuses anything else -- so its size is a
  // compile-time constant here as the block transforms' are in k_transform: loop counts, strides and index arithmetic
  // fold away; the stage is bound by vector issue)
  constexpr int ln = 7, n = 1 << ln, n2 = n >> 1;
  if constexpr (STAGED) {
    pc.mark(0);
    mdct_forward_wave<LOGS, ln, WaveTeam, false, false, true>(E.mdct, pcm, Wk, spec, pc, E.searchstep, n2 + VAMD_PW_SIZE(n2), spec_stride,
                                                             WaveTeam(), nullptr, nullptr, E.win);
  } else {
#if VAMD_GPU
  if (pre) {  // the samples are already in registers; a lane's window values are two (i = LANE, LANE + 64)
    const float w0 = E.win[LANE], w1 = E.win[LANE + 64];
#pragma unroll
    for (int m = 0; m < (n << LOGS) / 64; m++) A[LANE + 64 * m] = pre->v[m] * (m & 1 ? w1 : w0);
  } else
#endif
  WAVE_FOR(k, n << LOGS) {
    const int t = k >> ln, i = k & (n - 1);
    A[k] = t < count ? pcm[t * E.searchstep + i] * E.win[i] : 0.f;
  }
  WAVE_SYNC();
  pc.mark(0);
  // (spec_stride: floats between the steps' spectra -- n2 for a buffer of their own, the work buffers' stride where the
  // spectrum is left in a work buffer's plain half: spec == Wk)
  mdct_forward_wave<LOGS, ln>(E.mdct, A, Wk, spec, pc, n, n2 + VAMD_PW_SIZE(n2), spec_stride);
  }
  pc.mark(5);
  WAVE_FOR(t, count) {
    // float temp=vec[0]*vec[0]+.7*vec[1]*vec[1]+.2*vec[2]*vec[2];  the literals make it fp64
    const float v0 = spec[t * spec_stride], v1 = spec[t * spec_stride + 1], v2 = spec[t * spec_stride + 2];
    near_out[t] = (float)((double)(v0 * v0) + (.7 * (double)v1) * (double)v1 + (.2 * (double)v2) * (double)v2);
  }
  float top = -1e30f;  // the input-domain test (VAMD_ENV_LIMIT_DB, vamd_params.h): dB values are finite whatever the samples were
  WAVE_FOR(k, (n >> 2) << LOGS) {
    const int t = k >> (ln - 2), kk = k & ((n >> 2) - 1);
    if (t < count) {
      const F2 z = *(const F2 *)(spec + t * spec_stride + 2 * kk);
      const float val = z.x * z.x + z.y * z.y;
      const float dB = todB(val) * .5f;
      raw_out[t * (n >> 2) + kk] = dB;
      top = dB > top ? dB : top;
    }
  }
  if (bad && wave_any(top > VAMD_ENV_LIMIT_DB) && LANE == 0) lds_or_global_count(bad);
  WAVE_SYNC();
  pc.mark(6);
}

// `decay` as _ve_amp leaves it at lib/envelope.c:143 for global step J.
//   t  points at this step's near-DC term; t[-k] is the term k steps earlier (zeros before the stream)
VAMD_DEV float env_decay(const float *__restrict__ t, long J) {
  const int p = (int)(J % VAMD_VE_NEARDC);  // filters->nearptr at this step
  const float *r = t - p;                   // the last refresh step (nearptr == 0)
  float part = r[-VAMD_VE_NEARDC];          // nearDC_partialacc restarted one period before it...
  for (int i = -VAMD_VE_NEARDC + 1; i <= -1; i++) part += r[i];  // ...and has summed that period
  float acc = part + r[0];                  // :128  decay = nearDC_acc = nearDC_partialacc + temp
  float decay = acc;
  acc -= r[-VAMD_VE_NEARDC];                // :134  nearDC[ptr] still holds the term 15 steps back
  for (int i = 1; i <= p; i++) {
    acc += r[i];                            // :131  decay = nearDC_acc += temp
    decay = acc;
    acc -= r[i - VAMD_VE_NEARDC];
  }
  decay = (float)((double)decay * (1. / (VAMD_VE_NEARDC + 1)));
  return (float)((double)todB(decay) * .5 - (double)15.f);
}

// One band's amplitude for one step (lib/envelope.c:148-165): limit the raw dB pairs the band
// covers by the falling decay line and the energy floor, then the windowed mean.
VAMD_DEV float env_band_amp(const EnvP &E, const float *__restrict__ raw /*[32] of this step*/, float decay, int b) {
  const int begin = E.band_begin[b], end = E.band_end[b];
  for (int k = 0; k < begin; k++) decay = (float)((double)decay - 8.);  // decay-=8. once per pair, in order
  float acc = 0.f;
  for (int i = 0; i < end; i++) {
    float val = raw[begin + i];
    if (val < decay) val = decay;
    if (val < E.minenergy) val = E.minenergy;
    acc += val * E.band_window[b][i];
    decay = (float)((double)decay - 8.);
  }
  return acc * E.band_total[b];
}

// Trigger bits of one step for every value h of stretch/2 (lib/envelope.c:99-104,167-205):
// bit 2h = pre-echo (ret |= 1|4), bit 2h+1 = post-echo (ret |= 2), OR-ed over channels and bands.
//   amp[c]  points at this step's [VAMD_VE_BANDS+1] amplitudes of channel c; earlier steps sit
//           `amp_stride` floats lower each
// one (channel, band): `a` points at the step's band amplitude, earlier steps amp_stride floats apart.  The twelve
// history values are fetched before any is used (one trip to memory, not twelve).
VAMD_DEV uint32_t env_trigger_bits_one(const EnvP &E, const float *__restrict__ a, long amp_stride, int b) {
  uint32_t bits = 0;
  const float acc = a[0], prev = a[-amp_stride];
  float hist[VAMD_VE_MAXSTRETCH];
#pragma unroll
  for (int i = 0; i < VAMD_VE_MAXSTRETCH; i++) hist[i] = a[-(2 + i) * amp_stride];
  const float postmax = acc > prev ? acc : prev;
  const float postmin = acc < prev ? acc : prev;
  float premax = -99999.f, premin = 99999.f;
#pragma unroll
  for (int i = 0; i < VAMD_VE_MAXSTRETCH; i++) {
    const float v = hist[i];
    premax = premax > v ? premax : v;
    premin = premin < v ? premin : v;
    const int w = i + 1;  // window length reached: the `stretch` of h = w (and of h = 0,1 when w == 2)
    if (w < VAMD_VE_MINSTRETCH) continue;
    const float valmin = postmin - premin, valmax = postmax - premax;
    for (int h = (w == VAMD_VE_MINSTRETCH ? 0 : w); h <= w; h++) {
      float penalty = E.stretch_penalty - (float)(h - VAMD_VE_MINSTRETCH);
      if (penalty < 0.f) penalty = 0.f;
      if (penalty > E.stretch_penalty) penalty = E.stretch_penalty;
      if (valmax > E.preecho_thresh[b] + penalty) bits |= 1u << (2 * h);
      if (valmin < E.postecho_thresh[b] - penalty) bits |= 2u << (2 * h);
    }
  }
  return bits;
}

VAMD_DEV uint32_t env_trigger_bits(const EnvP &E, const float *const *amp, int ch, long amp_stride) {
  uint32_t bits = 0;
  for (int c = 0; c < ch; c++)
    for (int b = 0; b < VAMD_VE_BANDS; b++) bits |= env_trigger_bits_one(E, amp[c] + b, amp_stride, b);
  return bits;
}

// The recurrence itself (lib/envelope.c:234-239,258): returns the updated ve->stretch.  One wave
// per stream: the lanes fetch 64 steps' bits together and write 64 flags together; in between
// the chain runs on wave-uniform values (readlane + scalar integer ops), a handful of
// instructions per step with no memory access on the dependent path.
VAMD_DEV int env_walk_wave(const uint32_t *__restrict__ bits, long nsteps, int stretch, unsigned char *__restrict__ ret) {
  for (long base = 0; base < nsteps; base += NLANES) {
    const int cnt = nsteps - base < NLANES ? (int)(nsteps - base) : NLANES;
    LaneInts v;
    v.load((const int *)bits + base, cnt);
    int mine = 0;
    for (int i = 0; i < cnt; i++) {
      stretch++;
      if (stretch > VAMD_VE_MAXSTRETCH * 2) stretch = VAMD_VE_MAXSTRETCH * 2;
      const uint32_t two = ((uint32_t)v.get(i) >> (2 * (stretch / 2))) & 3u;
      const int r = (two & 1u ? 5 : 0) | (two & 2u ? 2 : 0);
      if (i == LANE) mine = r;
      if (two & 1u) stretch = -1;
    }
    if (LANE < cnt) ret[base + LANE] = (unsigned char)mine;
  }
  return stretch;
}

}  // namespace vamd
