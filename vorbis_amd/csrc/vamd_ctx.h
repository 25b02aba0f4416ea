// vamd_ctx.h -- the context behind a vamd_ctx handle: the parameter structs bound to the HBM image, streams and events,
// the workspace, staging, the profiling hooks, the environment knobs -- and the small helpers every entry point uses
// (fail / HIP_TRY, DeviceGuard, ws_get).  Part of the library's single translation unit: included by vamd_hip.hip, once,
// after vamd_kernels.h.
#pragma once
// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
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
         WS_PLAN_FLAGS, WS_PLAN_BLOCKS, WS_PLAN_COUNTS, WS_PLAN_BASE, WS_PLAN_DESC, WS_PLAN_ORDER, WS_STATUS, WS_WRAPPED, WS_RES_BOOKS, WS_PLAN_PENDING, WS_PLAN_GEO, WS_COUNT };
  DevBuf ws[2][WS_COUNT];  // per size class (a mixed stream keeps both batches in flight)
  // pinned staging for the per-block host API
  void *h_stage = nullptr;
  size_t h_stage_bytes = 0;
  void *h_plan = nullptr;  // pinned: a stream plan's per-stream bases on their way up (vamd_plan_streams)
  size_t h_plan_bytes = 0;
  void *h_geo = nullptr;   // pinned: per-stream geometry of whole streams of unequal length (vamd_plan_streams_whole_v)
  size_t h_geo_bytes = 0;
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
