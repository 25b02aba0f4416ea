// vamd_hip.hip -- libvorbis_amd.so: the C ABI of include/vorbis_amd.h -- entry points, workspace planning and the launch
// sequence -- over the gfx950 kernels (vamd_kernels.h: thin __global__ shells around the wave-level bodies in k_*.h) and the
// context (vamd_ctx.h).  One translation unit, three files (round 5: it was one file of 2 700 lines).
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
#include "k_lpc.h"

using namespace vamd;

#include "vamd_kernels.h"  // every __global__ shell
#include "vamd_ctx.h"      // struct vamd_ctx and the helpers of the entry points

// ---------------------------------------------------------------------------
// the C ABI (include/vorbis_amd.h): entry points, workspace planning, the launch sequence
// ---------------------------------------------------------------------------
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
  if (c->h_geo) (void)hipHostFree(c->h_geo);
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

int vamd_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : VAMD_EFAULT;
}

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
  for (int sm = 0; sm < cm.submaps; sm++) {
    const ResP &Rs = c->B.res[W][sm];
    // round 6: a stereo type-2 residue whose vectors tile runs of eight values is searched out of registers, a lane per
    // run, a wave per block (k_residue_chunks: persistent waves)
    const int chunks = Rs.chunked && !c->K.res_in_lds && ((uintptr_t)iwork & 15) == 0 && (n2 & 3) == 0
                           ? Rs.partvals * (Rs.tab_grouping >> 3) : 0;
    if (chunks > 0 && units <= res_team_max && chunks <= 64 * VAMD_RES_WAVES) {
      // a handful of units: a workgroup a unit, a thread a run (residue_team_chunks) -- a lone block's search in half the time
      hipLaunchKernelGGL(k_residue, dim3((unsigned)units), dim3((chunks + 63) & ~63),
                         (size_t)(Rs.lds_ints - Rs.bundle * n2 + Rs.fast_ints) * 4, s, Rs, cm, sm,
                         c->B.res_cap[W], nblobs, R->d, ch, n2, iwork, nonzero, rb.cls, rb.entries, rb.count, packets ? rb.books : nullptr, 1);
      continue;
    }
    if (chunks > 0 && units > res_team_max) {
      const size_t per_wave = (size_t)((Rs.partvals + Rs.nstages * Rs.partvals + 1 + 3) & ~3);
      const size_t lds = ((size_t)Rs.fast_ints + VAMD_RESC_WAVES * per_wave) * 4;
      int resident = 0;  // (persistent: as many workgroups as are resident at once)
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void *)k_residue_chunks, 64 * VAMD_RESC_WAVES, lds) != hipSuccess || resident < 1) {
        (void)hipGetLastError();
        resident = 1;
      }
      const long want = (units + VAMD_RESC_WAVES - 1) / VAMD_RESC_WAVES, fill = (long)c->num_cus * resident;
      const unsigned grid = (unsigned)(want < fill ? want : fill);
      hipLaunchKernelGGL(k_residue_chunks, dim3(grid), dim3(64 * VAMD_RESC_WAVES), lds, s, Rs, cm, sm, c->B.res_cap[W], nblobs, R->d, ch,
                         n2, units, iwork, nonzero, rb.cls, rb.entries, rb.count, packets ? rb.books : nullptr);
      continue;
    }
    // a stereo bundle's search keeps two waves busy, the five-channel bundle of the 5.1 layout four; a handful of units
    // takes four either way (nothing else wants the CU, and a lone unit's latency is the caller's)
    hipLaunchKernelGGL(k_residue, dim3((unsigned)units), dim3(64 * (c->B.res[W][sm].bundle * n2 > 4096 || units <= res_team_max ? VAMD_RES_WAVES : 2)),
                       (size_t)c->B.res[W][sm].lds_ints * 4, s, c->B.res[W][sm], cm, sm,
                       c->B.res_cap[W], nblobs, R->d, ch, n2, iwork, nonzero, rb.cls, rb.entries, rb.count, packets ? rb.books : nullptr, 0);
  }
  prof_mark(c, VAMD_ST_RESIDUE);
  if (packets) {
    const size_t lds = ((size_t)VAMD_PK_RING + VAMD_POSTS_STRIDE + VAMD_RES_CLASS_STRIDE + 2 * (size_t)c->B.res_off_ints[W] +
                        VAMD_PK_FTAB_INTS + 3 * (size_t)c->B.pack[W].nbooks) * 4;
    const long pair_max = c->K.pack_pair_max;
    // a handful of packets: two waves each -- where the rows hold any packet (the residue part is assembled past the
    // longest possible head and then moved down: in a shorter row the end of a cut-off packet would be lost on the way)
    if (units <= pair_max && packet_stride >= c->B.pack[W].capacity)
      hipLaunchKernelGGL(k_pack_pair, dim3((unsigned)units), dim3(128), lds + ((size_t)VAMD_PK_RING + 4 + c->B.res[W][0].fast_ints) * 4, s, c->B.pack[W],
                         c->B.floor[W][0], c->B.floor[W][1], c->B.res[W][0], c->B.res[W][1], cm, c->B.res_cap[W], c->B.res_off_ints[W],
                         R->d, ch, W, nblobs, posts, wrapped, post_valid, rb.cls, rb.entries, rb.books, rb.count, (unsigned *)packets,
                         (int)(packet_stride / 4), packet_bits);
    else if (cm.submaps == 1 && units >= 4 * (long)c->num_cus && !c->K.pack_per_packet) {
      // a batch of a one-submap mode: persistent waves over the packets, the tables staged once per workgroup (k_pack_waves)
      const ResP &R0 = c->B.res[W][0];
      const int per_wave_ints = (R0.slots + 2 * R0.nstages * R0.slots + 1 + 3) & ~3;
      const size_t ldsw = ((size_t)((VAMD_PK_FTAB_INTS + 3 * c->B.pack[W].nbooks + 3) & ~3) + (size_t)R0.fast_ints +
                           (size_t)VAMD_PKW_WAVES * ((size_t)VAMD_PK_RING + VAMD_POSTS_STRIDE + per_wave_ints)) * 4;
      // (persistent: as many workgroups as are resident at once -- registers, not LDS, set that here)
      int resident = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void *)k_pack_waves, 64 * VAMD_PKW_WAVES, ldsw) != hipSuccess || resident < 1) {
        (void)hipGetLastError();
        resident = 1;
      }
      const long per_cu = resident;
      const long want = (units + VAMD_PKW_WAVES - 1) / VAMD_PKW_WAVES, fill = per_cu * c->num_cus;
      hipLaunchKernelGGL(k_pack_waves, dim3((unsigned)(want < fill ? want : fill)), dim3(64 * VAMD_PKW_WAVES), ldsw, s, c->B.pack[W],
                         c->B.floor[W][0], R0, cm, c->B.res_cap[W], per_wave_ints, R->d, ch, W, nblobs, units, posts, wrapped, post_valid,
                         rb.cls, rb.entries, rb.books, rb.count, (unsigned *)packets, (int)(packet_stride / 4), packet_bits);
    } else
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
//   alone: no other size class's masks and floors in this run for the tone chain to run beside
static void launch_rest(vamd_ctx *c, BatchRun *R, int level, const vamd_managed_io *M = nullptr, ilog_t *m_ilogmask = nullptr,
                        int part = 3, bool forked = false, bool alone = true) {
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
        // (round 6, with round 5's faster seeding: a run of ONE size class does better with five teams -- noise mask +
        // visible tone tail per 131 072 stereo blocks 2.77 + 0.01 ms against 2.51 + 0.32 with six, 2.34 + 0.54 with seven,
        // 2.16 + 0.93 with eight; a mixed run, whose chains also have the other class's masks and floor fits to run
        // beside, keeps six: C5 11.25 ms against 11.34 with five.  profiles/r06_noise_teams.txt)
        const int beside = c->K.noise_waves > 0 ? c->K.noise_waves : (alone ? 20 : 24);
        const long cap = (fold_in_floor ? beside : 16) / nw;
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
    hipLaunchKernelGGL(k_ampmax_stream, dim3(1), dim3(64), 0, s, ch, R.nb, secs, c->B.ampmax_att_per_sec, *ampmax_state,
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
                             float *states, bool first_given = false, const vamd_managed_io *M0 = nullptr,
                             const vamd_managed_io *M1 = nullptr) {
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
  // bitrate-managed blocks (vamd_encode_blocks): fifteen candidate packets per block, as run_batch sets them up
  const vamd_managed_io *MM[2] = {M0, M1};
  ilog_t *m_ilog[2] = {nullptr, nullptr};
  for (int W = 0; W < 2; W++) {
    if (!MM[W] || R[W].nb == 0) continue;
    void *v;
    if ((r = ws_get(c, W, vamd_ctx::WS_M_ILOGMASK, (size_t)R[W].nb * VAMD_PACKETBLOBS * c->B.channels * (c->B.bs[W] / 2) * sizeof(ilog_t), &v)))
      return r;
    m_ilog[W] = (ilog_t *)v;
    if ((MM[W]->res_entries || MM[W]->packets) &&
        (r = res_bufs(c, W, R[W].nb * VAMD_PACKETBLOBS, MM[W]->res_class, MM[W]->res_entries, MM[W]->res_count, &R[W].rb)))
      return r;
  }
  if ((r = alloc_couple_state(c, &R[0], MM[0] ? R[0].nb * VAMD_PACKETBLOBS : R[0].nb)) ||
      (r = alloc_couple_state(c, &R[1], MM[1] ? R[1].nb * VAMD_PACKETBLOBS : R[1].nb)))
    return r;
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
  // The chains' walk (a wave per stream) feeds the tone
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
    hipLaunchKernelGGL(k_ampmax_streams_mixed, dim3((unsigned)nstreams), dim3(64), 0, s, c->B.channels, nstreams,
                       (const long long *)stream_start, (const int *)order, secs0, secs1, c->B.ampmax_att_per_sec, states,
                       R[0].p.local, R[1].p.local, R[0].p.ampin, R[1].p.ampin, R[0].p.ampglob, R[1].p.ampglob);
  else
    hipLaunchKernelGGL(k_ampmax_stream_mixed, dim3(1), dim3(64), 0, s, c->B.channels, nblocks_total, (const int *)order, secs0,
                       secs1, c->B.ampmax_att_per_sec, *ampmax_state, R[0].p.local, R[1].p.local, R[0].p.ampin,
                       R[1].p.ampin, R[0].p.ampglob, R[1].p.ampglob, d_state, first_given ? 1 : 0);
  s = c->stream;
  prof_mark(c, VAMD_ST_AMPMAX);
  R[0].d.ampmax_in = R[0].p.ampin;
  R[1].d.ampmax_in = R[1].p.ampin;
  if (chain_on_side) {  // both classes' masks first, the long blocks' leading
    launch_rest(c, &R[1], VAMD_LEVEL_FULL, MM[1], m_ilog[1], 1, true, R[0].nb == 0);
    launch_rest(c, &R[0], VAMD_LEVEL_FULL, MM[0], m_ilog[0], 1, true, R[1].nb == 0);
    launch_rest(c, &R[1], VAMD_LEVEL_FULL, MM[1], m_ilog[1], 2, true, R[0].nb == 0);
    launch_rest(c, &R[0], VAMD_LEVEL_FULL, MM[0], m_ilog[0], 2, true, R[1].nb == 0);
  } else {
    launch_rest(c, &R[0], VAMD_LEVEL_FULL, MM[0], m_ilog[0], 3, false, R[1].nb == 0);
    launch_rest(c, &R[1], VAMD_LEVEL_FULL, MM[1], m_ilog[1], 3, false, R[0].nb == 0);
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
                       const int32_t *nW, const int32_t *blocktype, float ampmax_in_first, int managed, float *ampmax_in,
                       float *ampmax_out, uint8_t *packets, long packet_stride, int32_t *packet_bits, int32_t *verdict) {
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
  const size_t K = managed ? VAMD_PACKETBLOBS : 1;  // packets per block: one, or a bitrate-managed block's fifteen candidates
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
    o_bits[w] = at, at = al(at + (size_t)nb[w] * K * 4);
    o_st[w] = at, at = al(at + (size_t)nb[w] * ch);
    o_pk[w] = at, at = al(at + (size_t)nb[w] * K * row[w]);
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
  vamd_managed_io m[2];
  memset(d, 0, sizeof(d));
  memset(io, 0, sizeof(io));
  memset(m, 0, sizeof(m));
  int r;
  for (int w = 0; w < 2; w++) {
    d[w].W = w;
    d[w].nblocks = nb[w];
    d[w].lW = (const int32_t *)(ds + o_lW[w]);
    d[w].nW = (const int32_t *)(ds + o_nW[w]);
    d[w].blocktype = (const int32_t *)(ds + o_bt[w]);
    io[w].pcm = (const float *)(ds + o_pcm[w]);
    io[w].ampmax_out = (float *)(ds + o_amp[w]);
    io[w].status = ds + o_st[w];
    if (!managed) {
      io[w].packets = ds + o_pk[w];
      io[w].packet_bits = (int32_t *)(ds + o_bits[w]);
      io[w].packet_stride = (int64_t)row[w];
    } else if (nb[w]) {
      // the candidates' intermediates stay on the device (workspace); their packets go to the arena
      const size_t n2 = (size_t)c->B.bs[w] / 2, units = (size_t)nb[w] * K;
      const size_t q_posts = 0, q_valid = al(q_posts + units * ch * VAMD_POSTS_STRIDE * 4), q_nz = al(q_valid + units * ch * 4),
                   q_iwork = al(q_nz + units * ch * 4), q_total = al(q_iwork + units * ch * n2 * 4);
      void *dv;
      if ((r = ws_get(c, w, vamd_ctx::WS_M_STAGE, q_total, &dv))) return r;
      unsigned char *dm = (unsigned char *)dv;
      m[w].posts = (int32_t *)(dm + q_posts);
      m[w].post_valid = (int32_t *)(dm + q_valid);
      m[w].nonzero = (int32_t *)(dm + q_nz);
      m[w].iwork = (int32_t *)(dm + q_iwork);
      m[w].packets = ds + o_pk[w];
      m[w].packet_bits = (int32_t *)(ds + o_bits[w]);
      m[w].packet_stride = (int64_t)row[w];
    }
  }
  float state = ampmax_in_first;
  r = run_streams_mixed(c, &d[0], &io[0], &d[1], &io[1], (const int32_t *)(ds + o_order), nblocks, &state, nullptr, 0, nullptr,
                        true, managed && nb[0] ? &m[0] : nullptr, managed && nb[1] ? &m[1] : nullptr);  // (synchronises: the chain's final state comes back)
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
    for (size_t k = 0; k < K; k++) {
      const int32_t bits = ((const int32_t *)(hs + o_bits[w]))[(size_t)i * K + k];
      packet_bits[(size_t)b * K + k] = bits;
      size_t bytes = ((size_t)(bits > 0 ? bits : 0) + 7) / 8;
      if (bytes > row[w]) bytes = row[w];  // (cut off: packet_bits says so)
      if (verdict[b] == VAMD_OK)
        memcpy(packets + ((size_t)b * K + k) * (size_t)packet_stride, hs + o_pk[w] + ((size_t)i * K + k) * row[w], bytes);
    }
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
// count_of / first_of (device, optional): streams of unequal length in one launch -- stream s takes its first count_of[s] steps
// only (its state is left after exactly those), and its first step starts first_of[s] samples into its buffer
static int envelope_search_batch(vamd_ctx *c, const float *pcm, long stream_stride, long channel_stride, long nstreams,
                                 long nsteps, vamd_envelope_state *states, unsigned char *ret, unsigned int *bad,
                                 const int *count_of = nullptr, const long long *first_of = nullptr) {
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
    if (items > 0x7fffffffL) return fail(c, VAMD_EINVAL, "detector: more than 2^31 groups of steps in one call");
    const long groups = (items + VAMD_ENV_WAVES - 1) / VAMD_ENV_WAVES;
    const size_t lds = ((size_t)VAMD_ENV_WAVES * (2 * VAMD_ENV_STAGE_FLOATS + VAMD_ENV_STEPS * (n2 + VAMD_PW_SIZE(n2))) + (n + n / 4) + n + n / 4) * 4;  // + the transform's tables
    int resident = 0;  // (persistent: as many workgroups as are resident at once)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void *)k_env_spectrum, 64 * VAMD_ENV_WAVES, lds) != hipSuccess || resident < 1) {
      (void)hipGetLastError();
      resident = 1;
    }
    const long cap = (long)c->num_cus * resident;
    hipLaunchKernelGGL(k_env_spectrum, dim3((unsigned)(groups < cap ? groups : cap)), dim3(64 * VAMD_ENV_WAVES), lds, s, E,
                       ch, nstreams, nsteps, pcm, stream_stride, channel_stride, near, raw, bad, first_of, c->d_dbg);
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
  else if (!env_untiled && (size_t)4 * ch * VAMD_ENV_BROWS * 9 * 4 <= c->lds_per_block) {  // (the tile of 7.1 wants 88.7 KB: a part
    // with 64 KB of LDS per workgroup takes the thread-per-step form below instead of failing the launch)
    const long items = nstreams * ((nsteps + 63) / 64);
    hipLaunchKernelGGL(k_env_bits_tiled, dim3((unsigned)((items + 3) / 4)), dim3(256), (size_t)4 * ch * VAMD_ENV_BROWS * 9 * 4, s, E, ch,
                       nstreams, nsteps, amp, bits);
  } else
    hipLaunchKernelGGL(k_env_bits_batch, dim3((unsigned)((nstreams * nsteps + 255) / 256)), dim3(256), 0, s, E, ch, nstreams,
                       nsteps, amp, bits);
  hipLaunchKernelGGL(k_env_walk, dim3((unsigned)nstreams), dim3(64), 0, s, ch, nstreams, nsteps, bits, near, amp,
                     states, ret, count_of);
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

// whole != 0: the streams are complete (vamd_plan_streams_whole) -- `nsamples` counts the space in front of the first
// sample and the real samples; the buffers have room for the end-of-stream padding behind them
// frames_of (host, whole streams only): the streams' own lengths, each <= nsamples - blocksizes[1]/2
static int plan_streams(vamd_ctx *c, float *pcm, long stream_stride, long channel_stride, long nstreams, long nsamples,
                        vamd_envelope_state *states, vamd_stream_plan *plan, int whole, const int64_t *frames_of = nullptr) {
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
  const int ch = c->B.channels, head = c->B.bs[1] / 2, pad = whole ? 3 * c->B.bs[1] : 0;
  if (whole && (nsamples < head || channel_stride < nsamples + pad))
    return fail(c, VAMD_EINVAL, "whole streams: a channel needs blocksizes[1]/2 samples of room in front and 3 * blocksizes[1] behind its samples");
  BlockoutP B;
  B.bs[0] = c->B.bs[0];
  B.bs[1] = c->B.bs[1];
  blockout_set_step(B, E.searchstep);
  B.nsamples = nsamples;
  B.eof = 0;
  // the steps _ve_envelope_search takes with this much data (lib/envelope.c:223-224); a whole stream's padding adds
  // pad / searchstep more, taken in a second pass once the padding exists
  long steps1 = nsamples / E.searchstep - VAMD_VE_WIN;
  if (steps1 < 0) steps1 = 0;
  long steps_all = (nsamples + pad) / E.searchstep - VAMD_VE_WIN;
  if (steps_all < 0) steps_all = 0;
  B.nsteps = steps1;
  B.maxblocks = (int)((nsamples + pad) / (B.bs[0] / 2)) + 2;  // a block advances the stream by at least blocksizes[0]/2
  plan->nstreams = nstreams;
  void *v_flags, *v_blocks, *v_counts, *v_base;
  int r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_FLAGS, (size_t)nstreams * (steps_all ? steps_all : 1), &v_flags))) return r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_BLOCKS, (size_t)nstreams * B.maxblocks * sizeof(PlannedBlock), &v_blocks))) return r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_COUNTS, (size_t)nstreams * 2 * sizeof(int), &v_counts))) return r;
  if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_BASE, (size_t)(3 * nstreams + 1) * sizeof(long long), &v_base))) return r;
  hipStream_t s = c->stream;
  const size_t plan_lds = (size_t)((steps_all + 4 + 15) & ~15L);
  if (plan_lds > c->lds_per_block) return fail(c, VAMD_EINVAL, "streams too long for one plan (their marks must fit a workgroup's LDS)");
  // (above the default 64 KB of dynamic LDS the launch needs the opt-in, and a launch that fails leaves counts[] --
  // which sizes everything below -- uninitialised: hence the checks straight after it)
  HIP_TRY(c, hipFuncSetAttribute((const void *)k_plan_streams, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_per_block));
  unsigned char *flags1 = (unsigned char *)v_flags, *flags2 = flags1 + (size_t)nstreams * steps1;
  const PlanGeo *geo = nullptr;      // streams of unequal length: their own sample counts, step counts and first padding steps
  const int *count1 = nullptr, *count2 = nullptr;
  const long long *first2 = nullptr;
  long steps2 = steps_all - steps1;  // steps of the second detector pass (the launch's: the longest stream's)
  if (whole && frames_of) {
    // [geo | count1 | count2 | first2] built on the host (a few words per stream) in a pinned buffer of the context's, one upload
    const size_t o_c1 = (size_t)nstreams * sizeof(PlanGeo), o_c2 = o_c1 + (size_t)nstreams * 4, o_f2 = (o_c2 + (size_t)nstreams * 4 + 7) & ~(size_t)7,
                 total = o_f2 + (size_t)nstreams * 8;
    if (c->h_geo_bytes < total) {
      if (c->h_geo) HIP_TRY(c, hipHostFree(c->h_geo));
      c->h_geo = nullptr, c->h_geo_bytes = 0;
      HIP_TRY(c, hipHostMalloc(&c->h_geo, total + total / 2, hipHostMallocDefault));
      c->h_geo_bytes = total + total / 2;
    }
    void *v_geo;
    if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_GEO, total, &v_geo))) return r;
    // (the previous plan's upload out of this buffer has long been consumed: every plan ends with a stream synchronisation)
    unsigned char *hg = (unsigned char *)c->h_geo;
    PlanGeo *g = (PlanGeo *)hg;
    int *c1 = (int *)(hg + o_c1), *c2 = (int *)(hg + o_c2);
    long long *f2 = (long long *)(hg + o_f2);
    steps2 = 0;
    for (long i = 0; i < nstreams; i++) {
      const long fr = (long)frames_of[i];
      if (fr < 1 || head + fr > nsamples) return fail(c, VAMD_EINVAL, "whole streams: a stream's length must be 1 .. the launch's frame count");
      long s1 = (head + fr) / E.searchstep - VAMD_VE_WIN, sa = (head + fr + pad) / E.searchstep - VAMD_VE_WIN;
      if (s1 < 0) s1 = 0;
      if (sa < s1) sa = s1;
      g[i].nsamples = head + fr + pad, g[i].eof = head + fr, g[i].nsteps = (int)sa, g[i].split = (int)s1;
      c1[i] = (int)s1, c2[i] = (int)(sa - s1), f2[i] = (long long)s1 * E.searchstep;
      if (sa - s1 > steps2) steps2 = sa - s1;
    }
    HIP_TRY(c, hipMemcpyAsync(v_geo, hg, total, hipMemcpyHostToDevice, s));
    geo = (const PlanGeo *)v_geo;
    count1 = (const int *)((unsigned char *)v_geo + o_c1), count2 = (const int *)((unsigned char *)v_geo + o_c2);
    first2 = (const long long *)((unsigned char *)v_geo + o_f2);
    // flags2's rows are steps2 long; the flag buffer was sized for steps_all per stream: steps1 + steps2 may exceed it by VE_WIN
    if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_FLAGS, (size_t)nstreams * (steps1 + steps2 + 1), &v_flags))) return r;
    flags1 = (unsigned char *)v_flags, flags2 = flags1 + (size_t)nstreams * steps1;
  }
  if (whole) {
    // the start of a stream as the example's 1024-sample writes make it (lib/block.c:524-528: the helper runs after the
    // first write that leaves more than blocksizes[1] samples beyond the centre, or when the stream is closed)
    const long frames = nsamples - head;
    long n_head = ((long)c->B.bs[1] / 1024 + 1) * 1024;
    if (frames < n_head) n_head = frames;
    const size_t lpc_lds = 80 * 8 + VAMD_LPC_MAX_ORDER * 4 + (size_t)(n_head + head > c->B.bs[1] + pad ? n_head + head : c->B.bs[1] + pad) * 4;
    if (lpc_lds > c->lds_per_block) return fail(c, VAMD_EIMPL, "block size too large for the stream-end extrapolation");
    HIP_TRY(c, hipFuncSetAttribute((const void *)k_lpc_head, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_per_block));
    HIP_TRY(c, hipFuncSetAttribute((const void *)k_lpc_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_per_block));
    if (n_head > 32)
      hipLaunchKernelGGL(k_lpc_head, dim3((unsigned)(nstreams * ch)), dim3(64), lpc_lds, s, ch, nstreams, pcm, stream_stride,
                         channel_stride, head, (int)n_head, geo);
    if (steps1 && (r = envelope_search_batch(c, pcm, stream_stride, channel_stride, nstreams, steps1, states, flags1, c->d_bad + 1, count1)))
      return r;
    // where every stream's walk stands when the data runs out: the reference's buffer begins blocksizes[1]/2 before it
    void *v_pending;
    if ((r = ws_get(c, 0, vamd_ctx::WS_PLAN_PENDING, (size_t)nstreams * sizeof(long long), &v_pending))) return r;
    hipLaunchKernelGGL(k_plan_streams, dim3((unsigned)nstreams), dim3(64), plan_lds, s, B, nstreams, flags1, steps1, steps1, flags2, steps2,
                       (PlannedBlock *)nullptr, (int *)nullptr, (long long *)v_pending, geo, 0);
    hipLaunchKernelGGL(k_lpc_tail, dim3((unsigned)(nstreams * ch)), dim3(64), lpc_lds, s, ch, nstreams, pcm, stream_stride,
                       channel_stride, nsamples, c->B.bs[1], pad, (const long long *)v_pending, geo);
    HIP_TRY(c, hipGetLastError());
    if (steps2 > 0 &&
        (r = envelope_search_batch(c, geo ? pcm : pcm + steps1 * E.searchstep, stream_stride, channel_stride, nstreams, steps2, states, flags2,
                                   c->d_bad + 1, count2, first2)))
      return r;
    B.eof = nsamples;
    B.nsamples = nsamples + pad;
    B.nsteps = steps_all;
  } else if (steps1 && (r = vamd_envelope_search_batch(c, pcm, stream_stride, channel_stride, nstreams, steps1, states, flags1)))
    return r;
  HIP_TRY(c, hipMemsetAsync(v_counts, 0, (size_t)nstreams * 2 * sizeof(int), s));
  hipLaunchKernelGGL(k_plan_streams, dim3((unsigned)nstreams), dim3(64), plan_lds, s, B, nstreams, flags1, steps1, steps1, flags2, steps2,
                     (PlannedBlock *)v_blocks, (int *)v_counts, (long long *)nullptr, geo, 1);
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

int vamd_plan_streams(vamd_ctx *c, const float *pcm, long stream_stride, long channel_stride, long nstreams, long nsamples,
                      vamd_envelope_state *states, vamd_stream_plan *plan) {
  return plan_streams(c, (float *)pcm, stream_stride, channel_stride, nstreams, nsamples, states, plan, 0);
}

int vamd_plan_streams_whole(vamd_ctx *c, float *pcm, long stream_stride, long channel_stride, long nstreams, long nframes,
                            vamd_envelope_state *states, vamd_stream_plan *plan) {
  if (c && nframes < 0) return fail(c, VAMD_EINVAL, "negative frame count");
  return plan_streams(c, pcm, stream_stride, channel_stride, nstreams, c ? c->B.bs[1] / 2 + nframes : 0, states, plan, 1);
}

int vamd_plan_streams_whole_v(vamd_ctx *c, float *pcm, long stream_stride, long channel_stride, long nstreams, long max_frames,
                              const int64_t *nframes, vamd_envelope_state *states, vamd_stream_plan *plan) {
  if (c && (max_frames < 0 || !nframes)) return fail(c, VAMD_EINVAL, "negative frame count / null lengths");
  return plan_streams(c, pcm, stream_stride, channel_stride, nstreams, c ? c->B.bs[1] / 2 + max_frames : 0, states, plan, 1, nframes);
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
