// vamd_feed.hip -- the host-fed farm: whole streams in from host memory (16-bit interleaved, as
// examples/encoder_example.c:179-202 reads them), finished packets back to host memory, over one or several GPUs
// (include/vorbis_amd.h, "the host-fed farm"; SURVEY.md 8d "H2D/D2H-inclusive", 8e).
//
// Every figure the library reported up to round 5 was for samples already resident in HBM.  A caller's samples are
// in host memory, and the link is the narrowest pipe on the way: a long stereo block advances its stream by 1024
// frames = 4 KB of 16-bit samples, so a 64 GB/s link feeds at most ~14 M blocks/s -- IF it is busy all the time and
// carries nothing but samples up and packet bytes down.  Hence the shape:
//   * LANES.  A lane is a context, a HIP stream, a thread of the library's, a pinned input arena, a pinned output
//     arena and the HBM buffers of one group of streams.  A group's life on its lane: one copy command up (the pinned
//     arena is what the copy engine reads: no staging copy on the host) -> k_feed_ingest (16-bit -> float, planar, with
//     the room either end that vamd_plan_streams_whole fills) -> the plan (LPC ends, detector, block walk; its block
//     counts are the lane thread's one wait in mid-flight) -> vamd_analyze_streams_mixed with packet output, blocks read
//     where they lie (50 % overlap never copied) -> three small kernels that lay the packets end to end: per-stream
//     sizes (a wave per stream), a scan over the streams, and a wave per packet that copies its words STRAIGHT INTO the
//     pinned output arena (mapped into the device's address space: the packets cross the link inside that kernel, no
//     copy command, no second wait).  Lanes are independent: while one computes, another's upload is on the wire.
//   * the call sequence is libvorbis' own, for a group: vamd_feed_buffer / _wrote / _packets / _release.
// Built on the public C ABI only (a context is used by one thread: its lane's), like vamd_batcher.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "vorbis_amd.h"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
size_t al(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- kernels ------------------------------------------------------------------------------------------------

// in [s][frame][c] (int16 or float) -> pcm[s * ss + c * cs + head + frame]; the room in front (head samples) and behind
// (pad samples) zeroed, as the reference's calloc'ed / not yet written buffer is.  A thread takes four frames of every
// channel: one 8 ch-byte (16-bit) or 16 ch-byte read, one 16-byte store per channel.
// frames_of / first_of (optional): streams of unequal length laid back to back -- stream s has frames_of[s] <= frames frames
// starting at frame first_of[s] of the arena; the rest of its buffer (laid out for `frames`) is zeroed.
template <typename T>
__global__ void k_feed_ingest(const T *__restrict__ in, int ch, long nstreams, long frames, int head, int pad,
                              float *__restrict__ pcm, long ss, long cs, float *__restrict__ amp,
                              vamd_envelope_state *__restrict__ states, const long long *__restrict__ frames_of,
                              const long long *__restrict__ first_of) {
  const long quads = (frames + 3) >> 2, hq = head >> 2, pq = pad >> 2, per = hq + quads + pq, total = nstreams * per;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long s = t / per, q = t - s * per;
    float *row = pcm + s * ss;
    const long mine = frames_of ? (long)frames_of[s] : frames, first = first_of ? (long)first_of[s] : s * frames;
    if (q < hq) {
      for (int c = 0; c < ch; c++) ((float4 *)(row + (long)c * cs))[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (q == 0) amp[s] = VAMD_AMPMAX_FLOOR;
    } else if (q < hq + quads) {
      const long f0 = (q - hq) << 2;
      const T *src = in + (first + f0) * ch;
      const int live = mine - f0 < 4 ? (mine > f0 ? (int)(mine - f0) : 0) : 4;
      for (int c = 0; c < ch; c++) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          float x = 0.f;
          if (k < live) {
            if (sizeof(T) == 2) x = (float)(int)src[k * ch + c] / 32768.f;  // examples/encoder_example.c:197-202
            else x = (float)src[k * ch + c];
          }
          v[k] = x;
        }
        ((float4 *)(row + (long)c * cs + head))[q - hq] = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
      const long f0 = (quads << 2) + ((q - hq - quads) << 2);
      for (int c = 0; c < ch; c++) ((float4 *)(row + (long)c * cs + head + f0))[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // a fresh detector state per stream (all-zero == a stream's start, include/vorbis_amd.h)
  const long words = nstreams * (long)(sizeof(vamd_envelope_state) / 4);
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < words; t += (long)gridDim.x * blockDim.x) ((uint32_t *)states)[t] = 0u;
}

struct FeedPlan {  // what the packing kernels need of a vamd_stream_plan and of the analysis' outputs
  const int32_t *order;
  const int64_t *stream_start;
  const int64_t *src[2];
  const int32_t *bits[2];
  const uint8_t *status[2];
  const uint8_t *packets[2];
  int64_t stride[2];
  int bs[2];
  int ch;
  int64_t stream_stride, eof;  // eof: first sample past the stream's real ones, in its buffer's coordinates
  const long long *frames_of;  // streams of unequal length: eof = head + frames_of[s]
  int head;
};

// a wave per stream: rel[k] = bytes (each packet rounded up to 4) of the stream's packets before packet k
__global__ __launch_bounds__(64) void k_feed_sizes(FeedPlan P, long nstreams, int64_t *__restrict__ rel, int64_t *__restrict__ stream_bytes) {
  const long s = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t k0 = P.stream_start[s], k1 = P.stream_start[s + 1];
  int64_t run = 0;
  for (int64_t base = k0; base < k1; base += 64) {
    const int64_t k = base + lane;
    int bytes = 0;
    if (k < k1) {
      const int o = P.order[k], W = (o >> 30) & 1, i = o & 0x3fffffff;
      unsigned st = 0;
      for (int c = 0; c < P.ch; c++) st |= P.status[W][(int64_t)i * P.ch + c];
      bytes = st ? 0 : (((P.bits[W][i] + 7) >> 3) + 3) & ~3;
    }
    int incl = bytes;  // inclusive scan over the wave
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (k < k1) rel[k] = run + incl - bytes;
    run += __shfl(incl, 63, 64);
  }
  if (lane == 0) stream_bytes[s] = run;
}

// one workgroup: stream_off[s] = bytes of all streams before s; stream_off[nstreams] = the total
__global__ __launch_bounds__(1024) void k_feed_scan(long nstreams, const int64_t *__restrict__ stream_bytes, int64_t *__restrict__ stream_off) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const long per = (nstreams + 1023) / 1024, lo = (long)t * per, hi = lo + per < nstreams ? lo + per : nstreams;
  int64_t sum = 0;
  for (long s = lo; s < hi; s++) sum += stream_bytes[s];
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int64_t v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int64_t run = part[t] - sum;
  for (long s = lo; s < hi; s++) {
    stream_off[s] = run;
    run += stream_bytes[s];
  }
  if (t == 1023) stream_off[nstreams] = part[1023];
}

// a wave per packet: its words into the output arena (host memory, mapped), its record beside them
struct FeedOut {
  int64_t *stream_start, *offset, *granulepos, *total;
  int32_t *bits;
  uint8_t *info, *bytes;
  int64_t cap;  // bytes the arena holds
};
__global__ __launch_bounds__(256) void k_feed_copy(FeedPlan P, long nstreams, long nblocks, const int64_t *__restrict__ rel,
                                                   const int64_t *__restrict__ stream_off, const int32_t *__restrict__ sid,
                                                   FeedOut O) {
  const long k = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k == 0 && lane == 0) *O.total = stream_off[nstreams];
  if (k <= nstreams && lane == 1) O.stream_start[k] = P.stream_start[k];  // (nstreams + 1 <= nblocks + 1 entries; see the launch)
  if (k >= nblocks) return;
  const int s = sid[k];
  const int o = P.order[k], W = (o >> 30) & 1, i = o & 0x3fffffff;
  unsigned st = 0;
  for (int c = 0; c < P.ch; c++) st |= P.status[W][(int64_t)i * P.ch + c];
  const int bits = P.bits[W][i], words = st ? 0 : (((bits + 7) >> 3) + 3) >> 2;
  const int64_t off = stream_off[s] + rel[k];
  const bool fits = off + 4 * (int64_t)words <= O.cap;
  if (fits) {
    const uint32_t *src = (const uint32_t *)(P.packets[W] + (int64_t)i * P.stride[W]);
    uint32_t *dst = (uint32_t *)(O.bytes + off);
    for (int w = lane; w < words; w += 64) dst[w] = src[w];
  }
  if (lane == 0) {
    const int64_t begin = P.src[W][i] - (int64_t)s * P.stream_stride, center = begin + P.bs[W] / 2;
    const bool last = k + 1 == P.stream_start[s + 1];
    O.offset[k] = off;
    O.bits[k] = st ? -1 : bits;
    const int64_t eof = P.frames_of ? (int64_t)P.head + P.frames_of[s] : P.eof;
    O.granulepos[k] = (center < eof ? center : eof) - P.bs[1] / 2;
    O.info[k] = (uint8_t)(W | (last ? 2 : 0) | ((st & 3) << 2));
  }
}

// sid[k] = the stream packet k belongs to (a wave per stream)
__global__ __launch_bounds__(64) void k_feed_sid(const int64_t *__restrict__ stream_start, int32_t *__restrict__ sid) {
  const long s = blockIdx.x;
  for (int64_t k = stream_start[s] + threadIdx.x; k < stream_start[s + 1]; k += 64) sid[k] = (int32_t)s;
}

struct Buf {
  void *p = nullptr;
  size_t bytes = 0;
  bool host = false;
  hipError_t need(size_t n) {
    if (bytes >= n) return hipSuccess;
    if (p) (void)(host ? hipHostFree(p) : hipFree(p));
    p = nullptr, bytes = 0;
    const hipError_t e = host ? hipHostMalloc(&p, n, hipHostMallocDefault) : hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  void drop() {
    if (p) (void)(host ? hipHostFree(p) : hipFree(p));
    p = nullptr, bytes = 0;
  }
};

}  // namespace

enum { LANE_FREE = 0, LANE_FILLING, LANE_QUEUED, LANE_DONE };

struct FeedLane {
  int device = 0;
  vamd_ctx *ctx = nullptr;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev_up = nullptr, ev_end = nullptr;
  Buf h_in, h_out, h_rec;                      // pinned: the group's samples; its packets; their records
  Buf d_in, d_pcm, d_states, d_amp;            // HBM: the samples as they came; as floats, planar; detector states; ampmax chains
  Buf d_pk[2], d_bits[2], d_status[2];         // the analysis' packet rows per size class
  Buf d_rel, d_sid, d_sbytes, d_soff, d_len, h_len;  // (d_len / h_len: [frames_of | first_of] of a group of unequal streams)
  std::thread worker;
  std::mutex *upload_turn = nullptr;  // its device's (vamd_feed::upload_turns)
  // the job (guarded by vamd_feed::m)
  int state = LANE_FREE;
  long nstreams = 0, frames = 0;  // frames: the group's longest stream
  std::vector<int64_t> frames_of;  // empty: every stream is `frames` long
  int format = 0;
  int status = 0;
  std::string err;
  vamd_feed_result result;
  double t_wrote = 0.;
  long served = 0;  // groups this lane has carried (the free lane that has waited longest goes out first)
};

struct vamd_feed {
  std::vector<FeedLane> lanes;
  int ch = 0, bs[2] = {0, 0};
  long pkcap[2] = {0, 0};
  long max_streams = 0, max_frames = 0;
  int format = VAMD_FEED_S16;
  std::mutex m;
  std::vector<std::unique_ptr<std::mutex>> upload_turns;  // one per device
  std::condition_variable cv_work, cv_done;
  bool stop = false;
  long turn = 0;
  std::string err;
};

#define FEED_TRY(expr)                                                              \
  do {                                                                              \
    const hipError_t e__ = (expr);                                                  \
    if (e__ != hipSuccess) {                                                        \
      L.err = std::string(#expr) + ": " + hipGetErrorString(e__);                   \
      return VAMD_EFAULT;                                                           \
    }                                                                               \
  } while (0)
#define FEED_CALL(expr)                                                             \
  do {                                                                              \
    const int r__ = (expr);                                                         \
    if (r__) {                                                                      \
      L.err = std::string(#expr) + ": " + vamd_last_error(L.ctx);                   \
      return r__;                                                                   \
    }                                                                               \
  } while (0)

// one group through its lane (the lane's own thread; its device is current)
static int run_group(vamd_feed *f, FeedLane &L) {
  const long ns = L.nstreams, frames = L.frames;
  const int ch = f->ch, head = f->bs[1] / 2, pad = 3 * f->bs[1];
  const size_t sample = L.format == VAMD_FEED_S16 ? 2 : 4;
  const bool uneven = !L.frames_of.empty();
  size_t in_frames = (size_t)ns * frames;
  if (uneven) {
    in_frames = 0;
    for (long i = 0; i < ns; i++) in_frames += (size_t)L.frames_of[(size_t)i];
  }
  const size_t in_bytes = in_frames * ch * sample;
  const long cs = (long)al((size_t)head + ((frames + 3) & ~3L) + pad, 64), ss = cs * ch;
  hipStream_t st = L.stream;
  FEED_TRY(L.d_in.need(in_bytes ? in_bytes : 16));
  FEED_TRY(L.d_pcm.need((size_t)ns * ss * 4));
  FEED_TRY(L.d_states.need((size_t)ns * sizeof(vamd_envelope_state)));
  FEED_TRY(L.d_amp.need((size_t)ns * 4));
  const long long *d_frames_of = nullptr, *d_first_of = nullptr;
  if (uneven) {
    FEED_TRY(L.h_len.need((size_t)ns * 16));
    FEED_TRY(L.d_len.need((size_t)ns * 16));
    long long *h = (long long *)L.h_len.p, at = 0;
    for (long i = 0; i < ns; i++) {
      h[i] = L.frames_of[(size_t)i];
      h[ns + i] = at;
      at += h[i];
    }
    FEED_TRY(hipMemcpyAsync(L.d_len.p, h, (size_t)ns * 16, hipMemcpyHostToDevice, st));
    d_frames_of = (const long long *)L.d_len.p;
    d_first_of = d_frames_of + ns;
  }
  {
    // ONE upload at a time per device.  The link is a single resource: lanes that upload side by side each get a share
    // of it and all finish late together -- and then all compute together while the link idles (measured: three lanes
    // in lockstep, 2.3 ms of every 13 without a single kernel on the chip).  Taking turns, a lane has the whole link,
    // starts its kernels the moment its samples are up, and the next lane's upload runs beside them: the lanes stagger
    // themselves.
    std::lock_guard<std::mutex> turn(*L.upload_turn);
    FEED_TRY(hipEventRecord(L.ev0, st));
    if (in_bytes) FEED_TRY(hipMemcpyAsync(L.d_in.p, L.h_in.p, in_bytes, hipMemcpyHostToDevice, st));
    FEED_TRY(hipEventRecord(L.ev_up, st));
    FEED_TRY(hipEventSynchronize(L.ev_up));
  }
  {
    const long total = ns * ((long)(head >> 2) + ((frames + 3) >> 2) + (pad >> 2));
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 32) blocks = 256L * 32;
    if (blocks < 1) blocks = 1;
    if (L.format == VAMD_FEED_S16)
      hipLaunchKernelGGL(k_feed_ingest<int16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const int16_t *)L.d_in.p, ch, ns, frames, head, pad,
                         (float *)L.d_pcm.p, ss, cs, (float *)L.d_amp.p, (vamd_envelope_state *)L.d_states.p, d_frames_of, d_first_of);
    else
      hipLaunchKernelGGL(k_feed_ingest<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float *)L.d_in.p, ch, ns, frames, head, pad,
                         (float *)L.d_pcm.p, ss, cs, (float *)L.d_amp.p, (vamd_envelope_state *)L.d_states.p, d_frames_of, d_first_of);
    FEED_TRY(hipGetLastError());
  }
  vamd_stream_plan plan;
  if (uneven)
    FEED_CALL(vamd_plan_streams_whole_v(L.ctx, (float *)L.d_pcm.p, ss, cs, ns, frames, L.frames_of.data(), (vamd_envelope_state *)L.d_states.p, &plan));
  else
    FEED_CALL(vamd_plan_streams_whole(L.ctx, (float *)L.d_pcm.p, ss, cs, ns, frames, (vamd_envelope_state *)L.d_states.p, &plan));
  const long nb = (long)(plan.nblocks[0] + plan.nblocks[1]);
  vamd_batch_desc desc[2];
  vamd_batch_io io[2];
  for (int W = 0; W < 2; W++) {
    const size_t n = (size_t)plan.nblocks[W];
    FEED_TRY(L.d_pk[W].need((n ? n : 1) * (size_t)f->pkcap[W]));
    FEED_TRY(L.d_bits[W].need((n ? n : 1) * 4));
    FEED_TRY(L.d_status[W].need((n ? n : 1) * (size_t)ch));
    memset(&desc[W], 0, sizeof(desc[W]));
    memset(&io[W], 0, sizeof(io[W]));
    desc[W].W = W;
    desc[W].nblocks = (long)n;
    desc[W].lW = plan.lW[W], desc[W].nW = plan.nW[W], desc[W].blocktype = plan.blocktype[W];
    if (!n) continue;
    io[W].pcm = (const float *)L.d_pcm.p;
    io[W].pcm_src = plan.src[W];
    io[W].pcm_channel_stride = cs;
    io[W].packets = (uint8_t *)L.d_pk[W].p;
    io[W].packet_bits = (int32_t *)L.d_bits[W].p;
    io[W].packet_stride = f->pkcap[W];
    io[W].status = (uint8_t *)L.d_status[W].p;
  }
  if (nb)
    FEED_CALL(vamd_analyze_streams_mixed(L.ctx, &desc[0], &io[0], &desc[1], &io[1], plan.order, plan.stream_start, ns, nb,
                                         (float *)L.d_amp.p));
  // the packets end to end, into the pinned arena
  FEED_TRY(L.d_rel.need((size_t)(nb ? nb : 1) * 8));
  FEED_TRY(L.d_sid.need((size_t)(nb ? nb : 1) * 4));
  FEED_TRY(L.d_sbytes.need((size_t)ns * 8));
  FEED_TRY(L.d_soff.need((size_t)(ns + 1) * 8));
  // records: [total | stream_start (ns + 1) | offset (nb) | granulepos (nb) | bits (nb) | info (nb)]
  const size_t o_start = 8, o_off = o_start + (size_t)(ns + 1) * 8, o_gp = o_off + (size_t)nb * 8, o_bits = o_gp + (size_t)nb * 8,
               o_info = o_bits + (size_t)nb * 4, rec_bytes = al(o_info + (size_t)nb, 16);
  FEED_TRY(L.h_rec.need(rec_bytes + rec_bytes / 4));
  FeedPlan P;
  P.order = plan.order, P.stream_start = plan.stream_start;
  for (int W = 0; W < 2; W++) {
    P.src[W] = plan.src[W], P.bits[W] = (const int32_t *)L.d_bits[W].p, P.status[W] = (const uint8_t *)L.d_status[W].p;
    P.packets[W] = (const uint8_t *)L.d_pk[W].p, P.stride[W] = f->pkcap[W], P.bs[W] = f->bs[W];
  }
  P.ch = ch, P.stream_stride = ss, P.eof = head + frames, P.frames_of = d_frames_of, P.head = head;
  for (int attempt = 0;; attempt++) {
    uint8_t *hrec = (uint8_t *)L.h_rec.p;
    void *drec = nullptr, *dbytes = nullptr;
    FEED_TRY(hipHostGetDevicePointer(&drec, hrec, 0));
    FEED_TRY(hipHostGetDevicePointer(&dbytes, L.h_out.p, 0));
    FeedOut O;
    uint8_t *dr = (uint8_t *)drec;
    O.total = (int64_t *)dr, O.stream_start = (int64_t *)(dr + o_start), O.offset = (int64_t *)(dr + o_off);
    O.granulepos = (int64_t *)(dr + o_gp), O.bits = (int32_t *)(dr + o_bits), O.info = dr + o_info;
    O.bytes = (uint8_t *)dbytes, O.cap = (int64_t)L.h_out.bytes;
    hipLaunchKernelGGL(k_feed_sid, dim3((unsigned)ns), dim3(64), 0, st, plan.stream_start, (int32_t *)L.d_sid.p);
    hipLaunchKernelGGL(k_feed_sizes, dim3((unsigned)ns), dim3(64), 0, st, P, ns, (int64_t *)L.d_rel.p, (int64_t *)L.d_sbytes.p);
    hipLaunchKernelGGL(k_feed_scan, dim3(1), dim3(1024), 0, st, ns, (const int64_t *)L.d_sbytes.p, (int64_t *)L.d_soff.p);
    const long waves = (nb > ns + 1 ? nb : ns + 1);
    hipLaunchKernelGGL(k_feed_copy, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, P, ns, nb, (const int64_t *)L.d_rel.p,
                       (const int64_t *)L.d_soff.p, (const int32_t *)L.d_sid.p, O);
    FEED_TRY(hipGetLastError());
    FEED_TRY(hipEventRecord(L.ev_end, st));
    FEED_TRY(hipEventSynchronize(L.ev_end));
    const int64_t total = *(const int64_t *)hrec;
    if (total <= (int64_t)L.h_out.bytes) {
      vamd_feed_result &R = L.result;
      R.nstreams = ns, R.nblocks = nb;
      R.stream_start = (const int64_t *)(hrec + o_start), R.offset = (const int64_t *)(hrec + o_off);
      R.granulepos = (const int64_t *)(hrec + o_gp), R.bits = (const int32_t *)(hrec + o_bits), R.info = hrec + o_info;
      R.bytes = (const uint8_t *)L.h_out.p, R.total_bytes = total;
      break;
    }
    if (attempt) {
      L.err = "packet arena still too small after growing it";
      return VAMD_EFAULT;
    }
    FEED_TRY(L.h_out.need((size_t)total + (size_t)total / 8));  // the packets are still in HBM: lay them out again
  }
  float up = 0.f, dev = 0.f;
  (void)hipEventElapsedTime(&up, L.ev0, L.ev_up);
  (void)hipEventElapsedTime(&dev, L.ev0, L.ev_end);
  L.result.upload_ms = up, L.result.device_ms = dev;
  return VAMD_OK;
}

static void feed_lane_main(vamd_feed *f, FeedLane *lane) {
  FeedLane &L = *lane;
  (void)hipSetDevice(L.device);
  std::unique_lock<std::mutex> g(f->m);
  for (;;) {
    f->cv_work.wait(g, [&] { return f->stop || L.state == LANE_QUEUED; });
    if (f->stop) return;
    g.unlock();
    const int r = run_group(f, L);
    const double t = now_s();
    g.lock();
    L.status = r;
    L.result.total_ms = (t - L.t_wrote) * 1e3;
    L.state = LANE_DONE;
    if (r) f->err = L.err;
    f->cv_done.notify_all();
  }
}

static void feed_free(vamd_feed *f) {
  {
    std::lock_guard<std::mutex> g(f->m);
    f->stop = true;
  }
  f->cv_work.notify_all();
  f->cv_done.notify_all();
  for (FeedLane &L : f->lanes)
    if (L.worker.joinable()) L.worker.join();
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (FeedLane &L : f->lanes) {
    (void)hipSetDevice(L.device);
    if (L.stream) (void)hipStreamSynchronize(L.stream);
    if (L.ctx) vamd_destroy(L.ctx);
    Buf *all[] = {&L.d_len, &L.h_len, &L.h_in, &L.h_out, &L.h_rec, &L.d_in, &L.d_pcm, &L.d_states, &L.d_amp, &L.d_pk[0], &L.d_pk[1], &L.d_bits[0],
                  &L.d_bits[1], &L.d_status[0], &L.d_status[1], &L.d_rel, &L.d_sid, &L.d_sbytes, &L.d_soff};
    for (Buf *b : all) b->drop();
    if (L.ev0) (void)hipEventDestroy(L.ev0);
    if (L.ev_up) (void)hipEventDestroy(L.ev_up);
    if (L.ev_end) (void)hipEventDestroy(L.ev_end);
    if (L.stream) (void)hipStreamDestroy(L.stream);
  }
  (void)hipSetDevice(cur);
  f->lanes.clear();
}

extern "C" {

int vamd_feed_create(vamd_feed **out, const void *setup_blob, size_t blob_bytes, const int *devices, int ndevices,
                     int lanes_per_device, long max_streams, long max_frames, int format) {
  if (!out) return VAMD_EINVAL;
  *out = nullptr;
  if (!setup_blob || lanes_per_device < 1 || lanes_per_device > 8 || max_streams < 1 || max_frames < 1 || ndevices < 0 ||
      ndevices > 64 || (ndevices > 0 && !devices) || (format != VAMD_FEED_S16 && format != VAMD_FEED_F32))
    return VAMD_EINVAL;
  int cur = 0;
  if (hipGetDevice(&cur) != hipSuccess) return VAMD_EFAULT;
  std::vector<int> devs;
  if (ndevices == 0) devs.push_back(cur);
  for (int i = 0; i < ndevices; i++) devs.push_back(devices[i] >= 0 ? devices[i] : cur);
  vamd_feed *f = new vamd_feed;
  f->max_streams = max_streams, f->max_frames = max_frames, f->format = format;
  f->lanes.resize(devs.size() * (size_t)lanes_per_device);
  for (size_t d = 0; d < devs.size(); d++) f->upload_turns.emplace_back(new std::mutex);
  int r = VAMD_OK;
  // lane l runs on device l % ndevices: consecutive groups go to different devices first, to a device's next lane after
  for (size_t l = 0; l < f->lanes.size() && !r; l++) {
    FeedLane &L = f->lanes[l];
    L.device = devs[l % devs.size()];
    L.upload_turn = f->upload_turns[l % devs.size()].get();
    L.h_in.host = L.h_out.host = L.h_rec.host = L.h_len.host = true;
    r = vamd_create(&L.ctx, setup_blob, blob_bytes, L.device);
    if (r) break;
    hipError_t e = hipSetDevice(L.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&L.ev0, hipEventDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&L.ev_up, hipEventBlockingSync);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&L.ev_end, hipEventBlockingSync);
    if (e == hipSuccess && vamd_set_stream(L.ctx, L.stream) != VAMD_OK) e = hipErrorUnknown;
    if (e == hipSuccess && l == 0) {
      f->ch = vamd_channels(L.ctx);
      for (int W = 0; W < 2; W++) f->bs[W] = vamd_blocksize(L.ctx, W), f->pkcap[W] = vamd_packet_capacity(L.ctx, W);
      if (f->pkcap[0] <= 0 || f->pkcap[1] <= 0) r = VAMD_EIMPL;  // packets of this mode are not assembled on the GPU
    }
    // the arenas: the group's samples; packets: half the samples' size AS 16-BIT to start with (a q 0.4 stream is a
    // tenth of that, q 1.0 on noise a third; run_group grows the arena when a group needs more)
    const size_t in_cap = (size_t)max_streams * max_frames * f->ch * (format == VAMD_FEED_S16 ? 2 : 4);
    if (e == hipSuccess && !r) e = L.h_in.need(in_cap);
    if (e == hipSuccess && !r) e = L.h_out.need(al((size_t)max_streams * max_frames * f->ch + (size_t)max_streams * 65536, 4096));
    if (e != hipSuccess) r = VAMD_EFAULT;
  }
  (void)hipSetDevice(cur);
  if (!r) {
    try {
      for (FeedLane &L : f->lanes) L.worker = std::thread(feed_lane_main, f, &L);
    } catch (...) {
      r = VAMD_EFAULT;
    }
  }
  if (r) {
    feed_free(f);
    delete f;
    return r;
  }
  *out = f;
  return VAMD_OK;
}

void vamd_feed_destroy(vamd_feed *f) {
  if (!f) return;
  feed_free(f);
  delete f;
}

int vamd_feed_lanes(const vamd_feed *f) { return f ? (int)f->lanes.size() : VAMD_EINVAL; }

int vamd_feed_device(const vamd_feed *f, int slot) {
  return (f && slot >= 0 && slot < (int)f->lanes.size()) ? f->lanes[(size_t)slot].device : VAMD_EINVAL;
}

int vamd_feed_buffer(vamd_feed *f, void **pcm) {
  if (!f || !pcm) return VAMD_EINVAL;
  std::unique_lock<std::mutex> g(f->m);
  for (;;) {
    if (f->stop) return VAMD_EFAULT;
    int best = -1;
    for (size_t l = 0; l < f->lanes.size(); l++)
      if (f->lanes[l].state == LANE_FREE && (best < 0 || f->lanes[l].served < f->lanes[(size_t)best].served)) best = (int)l;
    if (best >= 0) {
      FeedLane &L = f->lanes[(size_t)best];
      L.state = LANE_FILLING;
      L.served = ++f->turn;
      *pcm = L.h_in.p;
      return best;
    }
    // every lane is out: wait for a release -- unless nothing can release one (all handed out and none queued or done
    // would be the caller waiting for itself)
    bool hope = false;
    for (const FeedLane &L : f->lanes) hope |= L.state == LANE_QUEUED || L.state == LANE_DONE;
    if (!hope) return VAMD_EINVAL;
    f->cv_done.wait(g);
  }
}

int vamd_feed_wrote(vamd_feed *f, int slot, long nstreams, long frames) {
  if (!f || slot < 0 || slot >= (int)f->lanes.size()) return VAMD_EINVAL;
  if (nstreams < 1 || nstreams > f->max_streams || frames < 1 || frames > f->max_frames) return VAMD_EINVAL;
  std::lock_guard<std::mutex> g(f->m);
  FeedLane &L = f->lanes[(size_t)slot];
  if (L.state != LANE_FILLING) return VAMD_EINVAL;
  L.nstreams = nstreams, L.frames = frames, L.format = f->format;
  L.frames_of.clear();
  L.status = 0;
  memset(&L.result, 0, sizeof(L.result));
  L.t_wrote = now_s();
  L.state = LANE_QUEUED;
  f->cv_work.notify_all();
  return VAMD_OK;
}

int vamd_feed_wrote_v(vamd_feed *f, int slot, long nstreams, const int64_t *frames) {
  if (!f || !frames || slot < 0 || slot >= (int)f->lanes.size()) return VAMD_EINVAL;
  if (nstreams < 1 || nstreams > f->max_streams) return VAMD_EINVAL;
  long longest = 0;
  long long total = 0;
  for (long i = 0; i < nstreams; i++) {
    if (frames[i] < 1 || frames[i] > f->max_frames) return VAMD_EINVAL;
    if (frames[i] > longest) longest = (long)frames[i];
    total += frames[i];
  }
  if (total > (long long)f->max_streams * f->max_frames) return VAMD_EINVAL;
  std::lock_guard<std::mutex> g(f->m);
  FeedLane &L = f->lanes[(size_t)slot];
  if (L.state != LANE_FILLING) return VAMD_EINVAL;
  L.nstreams = nstreams, L.frames = longest, L.format = f->format;
  L.frames_of.assign(frames, frames + nstreams);
  L.status = 0;
  memset(&L.result, 0, sizeof(L.result));
  L.t_wrote = now_s();
  L.state = LANE_QUEUED;
  f->cv_work.notify_all();
  return VAMD_OK;
}

int vamd_feed_packets(vamd_feed *f, int slot, vamd_feed_result *out) {
  if (!f || !out || slot < 0 || slot >= (int)f->lanes.size()) return VAMD_EINVAL;
  std::unique_lock<std::mutex> g(f->m);
  FeedLane &L = f->lanes[(size_t)slot];
  if (L.state != LANE_QUEUED && L.state != LANE_DONE) return VAMD_EINVAL;
  f->cv_done.wait(g, [&] { return f->stop || L.state == LANE_DONE; });
  if (L.state != LANE_DONE) return VAMD_EFAULT;
  *out = L.result;
  return L.status;
}

int vamd_feed_release(vamd_feed *f, int slot) {
  if (!f || slot < 0 || slot >= (int)f->lanes.size()) return VAMD_EINVAL;
  std::lock_guard<std::mutex> g(f->m);
  FeedLane &L = f->lanes[(size_t)slot];
  if (L.state != LANE_DONE && L.state != LANE_FILLING) return VAMD_EINVAL;
  L.state = LANE_FREE;
  f->cv_done.notify_all();
  return VAMD_OK;
}

const char *vamd_feed_last_error(const vamd_feed *f) { return f ? f->err.c_str() : "null feed"; }

}  // extern "C"
