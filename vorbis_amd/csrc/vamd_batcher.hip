// vamd_batcher.hip -- the host shim for encoders that call vorbis_analysis() one block at a time from MANY
// threads (the other half of device-resident stream control: vamd_plan_streams serves callers that
// can hand over whole streams, this serves callers that cannot).
//
// libvorbis' own unit of work is one block of one stream (mapping0_forward, reference lib/mapping0.c:233-687); on
// the GPU one block is a handful of wavefronts walking serial phases (~0.4 ms), a batch of a thousand costs little
// more.  A batcher owns ONE context and turns concurrent vamd_batcher_encode_block() calls -- same contract as
// vamd_encode_block(), from any number of threads, one stream per thread as libvorbis requires -- into batched
// launches: whoever finds no batch under way becomes its leader, waits until every attached stream has a block
// pending (or `max_batch` have, or `max_wait_us` passed), takes the pending blocks of one size class, runs them as
// ONE vamd_analyze_batch() with packet output, hands the packets back, wakes THEIR owners (each request has its
// own condition variable: a finished batch must not wake hundreds of sleepers to find out it was not for them)
// and appoints the owner of the oldest pending block to lead the next batch.  No thread of its own; blocks of a
// stream stay in order because a stream has at most one pending.
//
// A batcher has a few LANES (contexts with their own stream and staging; VAMD_BATCH_LANES, default 2): one leader
// gathers at a time, but while its batch is on the GPU the next leader already gathers and launches on another lane --
// a batch of a few dozen blocks is latency on the GPU, not load, and the gaps between batches were most of the time.
//
// Built on the public C ABI only (a context is used by one thread at a time: the leader that holds its lane).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>
#include "vorbis_amd.h"

namespace {

struct Request {
  const float *const *pcm;
  int lW, W, nW, blocktype;
  float ampmax_in;
  float *ampmax_out;
  uint8_t *packet;
  long packet_cap;
  int32_t *packet_bits;
  int status = 0;
  bool done = false;
  bool lead = false;  // appointed to lead the next batch
  std::condition_variable cv;  // its owner sleeps here: a finished batch wakes its own blocks' owners, nobody else
};

}  // namespace

// one context with its stream and staging arenas: a batch runs on one lane
struct Lane {
  vamd_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  void *h_stage = nullptr, *d_stage = nullptr;
  size_t stage_bytes = 0;
  bool in_use = false;
};

struct vamd_batcher {
  std::vector<Lane> lanes;  // VAMD_BATCH_LANES (default 2): while one batch is on the GPU the next is gathered and launched
  int device = 0, ch = 0;
  int bs[2] = {0, 0};
  long pkcap[2] = {0, 0};
  int max_batch = 0, max_wait_us = 0;
  std::mutex m;
  std::condition_variable cv_lead;  // the collecting leader sleeps here
  int gatherers = 0;                // leaders gathering right now (a re-leading or appointed leader can start while another still waits)
  long in_flight = 0;               // blocks taken into batches that are on the GPU: their streams cannot submit meanwhile
  int leaders = 0;                  // leaders at work, gathering or running: <= lanes.size()
  std::vector<Request *> pending[2];
  int attached = 0;       // streams that announced themselves (vamd_batcher_attach)
  long nbatches = 0, nblocks = 0;
  double run_seconds = 0.;  // spent inside the batched GPU calls (staging copies included)
  // where a batch's time goes (vamd_batcher_report): seconds summed over batches
  double t_gather = 0., t_stage = 0., t_gpu = 0., t_unpack = 0., t_wake = 0.;
  long end_full = 0, end_all = 0, end_timeout = 0;  // why gathers ended: max_batch reached / every free stream in / max_wait_us passed
  long size_hist[12] = {0};                          // batches of 1, 2-3, 4-7, ... blocks
  std::string err;
};

static size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }

// one batch of blocks of size class W on lane L; called by a leader with the mutex NOT held
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// phase[0..2]: seconds spent staging the inputs, on the GPU (upload, kernels, download, sync), handing the packets out
static int run_batch(vamd_batcher *b, Lane &L, int W, Request *const *reqs, size_t nb, std::string *err, double *phase) {
  const double t_in = now_s();
  const size_t ch = (size_t)b->ch, n = (size_t)b->bs[W], row = (size_t)b->pkcap[W];
  // arena: [pcm | lW | nW | blocktype | ampmax_in || ampmax_out | bits | input status | packets]
  const size_t o_pcm = 0, o_lW = al16(nb * ch * n * 4), o_nW = al16(o_lW + nb * 4), o_bt = al16(o_nW + nb * 4),
               o_ain = al16(o_bt + nb * 4), o_out = al16(o_ain + nb * 4), o_aout = o_out, o_bits = al16(o_aout + nb * 4),
               o_st = al16(o_bits + nb * 4), o_pk = al16(o_st + nb * ch), total = al16(o_pk + nb * row);
  // the leader is an application thread inside vorbis_analysis(): its current device is put back on every way out
  struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { (void)hipGetDevice(&prev); }
    ~DeviceRestore() {
      if (prev >= 0) (void)hipSetDevice(prev);
    }
  } device_restore;
  hipError_t e = hipSetDevice(b->device);
  if (e == hipSuccess && L.stage_bytes < total) {
    if (L.h_stage) (void)hipHostFree(L.h_stage);
    if (L.d_stage) (void)hipFree(L.d_stage);
    L.h_stage = L.d_stage = nullptr;
    L.stage_bytes = 0;
    const size_t want = total + total / 2;
    e = hipHostMalloc(&L.h_stage, want, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&L.d_stage, want);
    if (e == hipSuccess) L.stage_bytes = want;
  }
  if (e != hipSuccess) {
    *err = std::string("batcher staging: ") + hipGetErrorString(e);
    return VAMD_EFAULT;
  }
  unsigned char *hs = (unsigned char *)L.h_stage, *ds = (unsigned char *)L.d_stage;
  for (size_t k = 0; k < nb; k++) {
    const Request &r = *reqs[k];
    for (size_t c = 0; c < ch; c++) memcpy(hs + o_pcm + (k * ch + c) * n * 4, r.pcm[c], n * 4);
    ((int32_t *)(hs + o_lW))[k] = r.lW;
    ((int32_t *)(hs + o_nW))[k] = r.nW;
    ((int32_t *)(hs + o_bt))[k] = r.blocktype;
    ((float *)(hs + o_ain))[k] = r.ampmax_in;
  }
  const double t_staged = now_s();
  e = hipMemcpyAsync(ds, hs, o_out, hipMemcpyHostToDevice, L.stream);
  if (e != hipSuccess) {
    *err = std::string("batcher upload: ") + hipGetErrorString(e);
    return VAMD_EFAULT;
  }
  vamd_batch_desc d;
  memset(&d, 0, sizeof(d));
  d.W = W;
  d.nblocks = (long)nb;
  d.lW = (const int32_t *)(ds + o_lW);
  d.nW = (const int32_t *)(ds + o_nW);
  d.blocktype = (const int32_t *)(ds + o_bt);
  d.ampmax_in = (const float *)(ds + o_ain);
  vamd_batch_io io;
  memset(&io, 0, sizeof(io));
  io.pcm = (const float *)(ds + o_pcm);
  io.ampmax_out = (float *)(ds + o_aout);
  io.packets = ds + o_pk;
  io.packet_bits = (int32_t *)(ds + o_bits);
  io.packet_stride = (int64_t)row;
  io.status = ds + o_st;
  int r = vamd_analyze_batch(L.ctx, &d, &io, VAMD_LEVEL_FULL);
  if (r) {
    *err = std::string("batcher analyze: ") + vamd_last_error(L.ctx);
    return r;
  }
  e = hipMemcpyAsync(hs + o_out, ds + o_out, total - o_out, hipMemcpyDeviceToHost, L.stream);
  if (e == hipSuccess) e = hipStreamSynchronize(L.stream);
  if (e != hipSuccess) {
    *err = std::string("batcher download: ") + hipGetErrorString(e);
    return VAMD_EFAULT;
  }
  const double t_back = now_s();
  for (size_t k = 0; k < nb; k++) {
    Request &q = *reqs[k];
    bool outside = false;  // this block's input was outside the domain (include/vorbis_amd.h): its own error, nobody else's
    for (size_t c = 0; c < ch; c++) outside |= hs[o_st + k * ch + c] != 0;
    if (outside) {
      q.status = VAMD_EINVAL;
      continue;
    }
    const int32_t bits = ((const int32_t *)(hs + o_bits))[k];
    const size_t bytes = ((size_t)bits + 7) / 8;
    if (bits < 0 || bytes > row || bytes > (size_t)q.packet_cap) {
      q.status = VAMD_EINVAL;  // the caller's buffer is shorter than vamd_packet_capacity()
      continue;
    }
    memcpy(q.packet, hs + o_pk + k * row, bytes);
    *q.packet_bits = bits;
    if (q.ampmax_out) *q.ampmax_out = ((const float *)(hs + o_aout))[k];
    q.status = VAMD_OK;
  }
  phase[0] = t_staged - t_in, phase[1] = t_back - t_staged, phase[2] = now_s() - t_back;
  return VAMD_OK;
}

extern "C" {

static void free_lanes(vamd_batcher *b) {
  for (Lane &L : b->lanes) {
    if (L.h_stage) (void)hipHostFree(L.h_stage);
    if (L.d_stage) (void)hipFree(L.d_stage);
    if (L.ctx) vamd_destroy(L.ctx);
    if (L.stream) (void)hipStreamDestroy(L.stream);
  }
  b->lanes.clear();
}

int vamd_batcher_create(vamd_batcher **out, const void *setup_blob, size_t blob_bytes, int device, int max_batch,
                        int max_wait_us) {
  if (!out) return VAMD_EINVAL;
  *out = nullptr;
  if (max_batch < 1 || max_wait_us < 0) return VAMD_EINVAL;
  int nlanes = getenv("VAMD_BATCH_LANES") ? atoi(getenv("VAMD_BATCH_LANES")) : 2;
  if (nlanes < 1) nlanes = 1;
  if (nlanes > 8) nlanes = 8;
  vamd_batcher *b = new vamd_batcher;
  int cur = 0;
  (void)hipGetDevice(&cur);
  b->device = device >= 0 ? device : cur;
  b->lanes.resize((size_t)nlanes);
  int r = VAMD_OK;
  for (Lane &L : b->lanes) {
    r = vamd_create(&L.ctx, setup_blob, blob_bytes, device);
    if (r) break;
    hipError_t e = hipSetDevice(b->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
    if (e == hipSuccess && vamd_set_stream(L.ctx, L.stream) != VAMD_OK) e = hipErrorUnknown;
    if (e != hipSuccess) {
      r = VAMD_EFAULT;
      break;
    }
  }
  if (!r) {
    vamd_ctx *c0 = b->lanes[0].ctx;
    b->ch = vamd_channels(c0);
    for (int W = 0; W < 2; W++) {
      b->bs[W] = vamd_blocksize(c0, W);
      b->pkcap[W] = vamd_packet_capacity(c0, W);
    }
    // (a mode whose packets the GPU does not assemble has nothing to batch here)
    if (b->pkcap[0] <= 0 || b->pkcap[1] <= 0) r = VAMD_EIMPL;
  }
  b->max_batch = max_batch;
  b->max_wait_us = max_wait_us;
  if (r) {
    (void)hipSetDevice(b->device);
    free_lanes(b);
    (void)hipSetDevice(cur);
    delete b;
    return r;
  }
  if (cur != b->device) (void)hipSetDevice(cur);
  *out = b;
  return VAMD_OK;
}

void vamd_batcher_destroy(vamd_batcher *b) {
  if (!b) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  (void)hipSetDevice(b->device);
  free_lanes(b);
  (void)hipSetDevice(cur);
  delete b;
}

void vamd_batcher_attach(vamd_batcher *b) {
  if (!b) return;
  std::lock_guard<std::mutex> g(b->m);
  b->attached++;
}

void vamd_batcher_detach(vamd_batcher *b) {
  if (!b) return;
  {
    std::lock_guard<std::mutex> g(b->m);
    if (b->attached > 0) b->attached--;
  }
  b->cv_lead.notify_one();  // a leader waiting for this stream's block need not wait any longer
}

int vamd_batcher_encode_block(vamd_batcher *b, const float *const *pcm, int lW, int W, int nW, int blocktype,
                              float ampmax_in, float *ampmax_out, uint8_t *packet, long packet_cap,
                              int32_t *packet_bits) {
  if (!b || !pcm || !packet || !packet_bits || (W != 0 && W != 1)) return VAMD_EINVAL;
  for (int c = 0; c < b->ch; c++)
    if (!pcm[c]) return VAMD_EINVAL;
  Request rq;
  rq.pcm = pcm, rq.lW = lW, rq.W = W, rq.nW = nW, rq.blocktype = blocktype, rq.ampmax_in = ampmax_in;
  rq.ampmax_out = ampmax_out, rq.packet = packet, rq.packet_cap = packet_cap, rq.packet_bits = packet_bits;
  std::unique_lock<std::mutex> lk(b->m);
  b->pending[W].push_back(&rq);
  if (b->gatherers) {
    b->cv_lead.notify_all();  // the gathering leaders count it
  } else if (b->leaders < (int)b->lanes.size()) {
    b->leaders++;  // nobody is gathering and a lane is free: we lead
    rq.lead = true;
  }
  for (;;) {
    // sleep until our block comes back, or until we are appointed to lead a batch
    while (!rq.done && !rq.lead) rq.cv.wait(lk);
    if (rq.done) break;
    // ---- we lead: gather (one leader at a time), then run the batch on a free lane, then hand the results back
    b->gatherers++;
    const double t_g0 = now_s();
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(b->max_wait_us);
    for (;;) {
      // everybody who can still submit has: the attached streams less those whose block is in a running batch
      const size_t have = b->pending[0].size() + b->pending[1].size();
      const long free_streams = (long)b->attached - b->in_flight;
      if (have >= (size_t)b->max_batch) {
        b->end_full++;
        break;
      }
      if (b->attached > 0 && have > 0 && (long)have >= free_streams) {
        b->end_all++;
        break;
      }
      if (b->cv_lead.wait_until(lk, deadline) == std::cv_status::timeout) {
        b->end_timeout++;
        break;
      }
    }
    b->t_gather += now_s() - t_g0;
    // the size class with more blocks waiting goes first (ours, if it is a tie)
    const int Wb = b->pending[W].size() >= b->pending[1 - W].size() ? W : 1 - W;
    std::vector<Request *> take;
    {
      std::vector<Request *> &q = b->pending[Wb];
      const size_t nb = q.size() < (size_t)b->max_batch ? q.size() : (size_t)b->max_batch;
      take.assign(q.begin(), q.begin() + (long)nb);
      q.erase(q.begin(), q.begin() + (long)nb);
    }
    Lane *lane = nullptr;
    for (Lane &L : b->lanes)
      if (!L.in_use) {
        lane = &L;
        break;
      }
    b->gatherers--;
    if (take.empty() || !lane) {
      // nothing left to run (an appointed leader that woke up late finds everything, its own block included, in
      // another leader's batch): step down and wait for the block like everybody else
      for (Request *t : take) t->status = VAMD_EFAULT, t->done = true, t->cv.notify_one();  // (no lane: cannot happen)
      rq.lead = false;
      b->leaders--;
      continue;
    }
    lane->in_use = true;  // (there is one: leaders <= lanes, and every other leader holds at most one)
    b->in_flight += (long)take.size();
    if (b->gatherers) b->cv_lead.notify_all();  // fewer streams are left to wait for
    // what is still pending (the other size class, latecomers) gets its own leader at once if a lane is free
    if (b->leaders < (int)b->lanes.size()) {
      Request *next = nullptr;
      for (int w = 0; w < 2 && !next; w++)
        for (Request *t : b->pending[w])
          if (!t->lead) {
            next = t;
            break;
          }
      if (next) {
        b->leaders++;
        next->lead = true;
        next->cv.notify_one();
      }
    }
    lk.unlock();
    std::string err;
    const auto t0 = std::chrono::steady_clock::now();
    double phase[3] = {0., 0., 0.};
    const int r = run_batch(b, *lane, Wb, take.data(), take.size(), &err, phase);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    lk.lock();
    const double t_w0 = now_s();
    b->t_stage += phase[0], b->t_gpu += phase[1], b->t_unpack += phase[2];
    {
      int h = 0;
      for (size_t v = take.size(); v > 1 && h < 11; v >>= 1) h++;
      b->size_hist[h]++;
    }
    lane->in_use = false;
    b->in_flight -= (long)take.size();
    if (r) b->err = err;
    b->run_seconds += dt;
    b->nbatches++;
    b->nblocks += (long)take.size();
    for (Request *t : take) {
      if (r) t->status = r;
      t->done = true;
      if (t != &rq) t->cv.notify_one();
    }
    b->t_wake += now_s() - t_w0;
  }
  // ---- our block is back: pass the lead to the owner of the oldest pending block that has none, if any
  if (rq.lead) {
    Request *next = nullptr;
    if (!b->gatherers)
      for (int w = 0; w < 2 && !next; w++)
        for (Request *t : b->pending[w])
          if (!t->lead) {
            next = t;
            break;
          }
    if (next) {
      next->lead = true;
      next->cv.notify_one();
    } else {
      b->leaders--;
    }
  }
  return rq.status;
}

const char *vamd_batcher_last_error(const vamd_batcher *b) { return b ? b->err.c_str() : "null batcher"; }

void vamd_batcher_stats(vamd_batcher *b, long *batches, long *blocks, double *run_seconds) {
  if (!b) return;
  std::lock_guard<std::mutex> g(b->m);
  if (batches) *batches = b->nbatches;
  if (blocks) *blocks = b->nblocks;
  if (run_seconds) *run_seconds = b->run_seconds;
}

long vamd_batcher_report(vamd_batcher *b, char *buf, long cap) {
  if (!b || !buf || cap < 1) return 0;
  std::lock_guard<std::mutex> g(b->m);
  const double nbt = b->nbatches > 0 ? (double)b->nbatches : 1.;
  int n = snprintf(buf, (size_t)cap,
                   "batches %ld (%ld lanes), blocks %ld; a batch on average: gathered for %.0f us (ended: %ld full, %ld every free stream in, %ld "
                   "timed out), staged in %.0f us, on the GPU %.0f us, packets handed out in %.0f us, owners woken in %.0f us\nbatch sizes 1 / 2-3 / 4-7 / ... :",
                   b->nbatches, (long)b->lanes.size(), b->nblocks, 1e6 * b->t_gather / nbt, b->end_full, b->end_all, b->end_timeout,
                   1e6 * b->t_stage / nbt, 1e6 * b->t_gpu / nbt, 1e6 * b->t_unpack / nbt, 1e6 * b->t_wake / nbt);
  for (int h = 0; h < 12 && n > 0 && n < cap; h++) n += snprintf(buf + n, (size_t)(cap - n), " %ld", b->size_hist[h]);
  return n;
}

vamd_ctx *vamd_batcher_context(vamd_batcher *b) { return b && !b->lanes.empty() ? b->lanes[0].ctx : nullptr; }

}  // extern "C"
