// vamd_wave.h -- the execution vocabulary every kernel body is written in.
//
// One 64-lane wavefront owns one channel-block (or one stereo block).  A kernel
// body is a sequence of *phases*; inside a phase the lanes stride over
// independent work items (WAVE_FOR), and WAVE_SYNC() separates phases that
// communicate through LDS.
//
// This header defines that vocabulary for gfx950 -- the ONLY implementation the product contains: LANE =
// threadIdx.x & 63, WAVE_SYNC = a compiler fence (the LDS unit executes a wave's DS instructions in order),
// reductions and scans on the VALU's DPP path, small per-block arrays held one entry per lane and read with
// v_readlane.  The test suite also compiles the kernel bodies with the host compiler as ONE lane (tests/emul), to
// check their arithmetic against the oracle on a machine without a GPU; the vocabulary for that build lives with
// the tests (tests/emul/vamd_wave_host.h, found only through the test build's include path) -- it is not a
// fallback and is never part of the library.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define VAMD_GPU 1
#include <hip/hip_runtime.h>
#define VAMD_DEV __device__ __forceinline__
#define VAMD_HOSTDEV __host__ __device__ __forceinline__
#define VAMD_MEM __device__ __forceinline__
#define VAMD_DEV_NOINLINE __device__ __noinline__
#define VAMD_CONST_TABLE __constant__
#define LANE ((int)(threadIdx.x & 63))  // a workgroup may hold several independent waves
#define NLANES 64
// Phase boundary for data exchanged through LDS.  The workgroup IS one wavefront, and
// the LDS unit executes a wave's DS instructions in issue order, so a later ds_read
// already sees every lane's earlier ds_write: all that is needed is that the
// compiler keeps the two sides apart.  Unlike __syncthreads() this does not drain
// outstanding HBM loads/stores (s_waitcnt vmcnt(0)) at every phase.
#define WAVE_SYNC()                                         \
  do {                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  \
    __builtin_amdgcn_wave_barrier();                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  \
  } while (0)
// Phase boundary for data exchanged through HBM between lanes of the wave.
#define WAVE_SYNC_GLOBAL() __syncthreads()

// Register tiles: a lane may keep "its" quads of a block in registers across phases.
// Lane l owns quads l, l+64, ... (four consecutive bins each); VAMD_QPL bounds how
// many (block sizes up to 2048 -> 1024 bins -> 4 quads per lane).
#define VAMD_QPL 4
#define LANE_QUADS(kq, q, nq) _Pragma("unroll") for (int kq = 0, q = LANE; kq < VAMD_QPL; kq++, q += NLANES) if (q < (nq))
// a slice [q0, q1) of the quads, QPS per lane: several waves can share one block's bins
#define SLICE_QUADS(kq, q, q0, q1, QPS) \
  _Pragma("unroll") for (int kq = 0, q = (q0) + LANE; kq < (QPS); kq++, q += NLANES) if (q < (q1))
// the same for a whole n-sample block (2048 samples -> 8 quads per lane)
#define VAMD_QPL2 8
#define LANE_QUADS2(kq, q, nq) _Pragma("unroll") for (int kq = 0, q = LANE; kq < VAMD_QPL2; kq++, q += NLANES) if (q < (nq))
// lanes stride over [0, count), unrolled x4 so that the independent HBM/L2 loads of four iterations are in
// flight together
#define WAVE_FOR(i, count) _Pragma("unroll 4") for (int i = LANE; i < (count); i += NLANES)
// a lane's bins i0 + LANE + 64 k, k < KPL (k_noise)
#define LANE_BINS(k, i, i0, KPL, n) _Pragma("unroll") for (int k = 0, i = (i0) + LANE; k < (KPL); k++, i += NLANES) if (i < (n))
// A "team": all the waves of a workgroup working on one unit (k_residue: the search of a block's
// vectors is wide enough for several waves, and sharing one LDS copy of the work vector between them
// keeps more units resident per CU).  With a 64-thread workgroup a team is a wave.
#define TEAM_FOR(i, count) for (int i = (int)threadIdx.x; i < (count); i += (int)blockDim.x)
#define TEAM_RANGE(i, lo, hi) for (int i = (lo) + (int)threadIdx.x; i < (hi); i += (int)blockDim.x)
#define TEAM_SYNC() __syncthreads()
#define TEAM_FIRST_WAVE (threadIdx.x < 64)
#define TEAM_LEADER (threadIdx.x == 0)
// items / register quads dealt over a team policy object (k_transform.h: WaveTeam); the item loop is
// unrolled x4 so that the independent LDS reads of four items are in flight together
#define TEAM_EACH(i, count, tm) _Pragma("unroll 4") for (int i = (tm).tid(); i < (count); i += (tm).size())
#define TEAM_QUADS(kq, q, nq, QPT, tm) _Pragma("unroll") for (int kq = 0, q = (tm).tid(); kq < (QPT); kq++, q += (tm).size()) if (q < (nq))

namespace vamd {

VAMD_DEV unsigned brev32(unsigned x) { return __builtin_bitreverse32(x); }  // bit 0 <-> bit 31

// Full-wave reductions and scans on the VALU's DPP path.  (The __shfl family compiles to
// ds_bpermute_b32, which occupies the CU's LDS pipe like any other LDS instruction; the ordered
// phases call these dozens of times per block.)  Every lane must be active; results of the
// reductions are wave-uniform.  Steps: Hillis-Steele inside each row of 16 (row_shr 1,2,4,8),
// then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3 -- an inclusive scan whose
// last lane holds the total.  Lanes without a source keep `ident`.
#define VAMD_DPP_SCAN(v, ident, OP)                                                  \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));          \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));          \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false));          \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false));          \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false));          \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x143, 0xc, 0xf, false));
// idempotent operators (max, or): a lane without a source combines with itself, no identity needed
#define VAMD_DPP_SCAN_SELF(v, OP)                                                    \
  { int t_; \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); v = OP(v, t_);     \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false); v = OP(v, t_);     \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false); v = OP(v, t_);     \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false); v = OP(v, t_);     \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false); v = OP(v, t_);     \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false); v = OP(v, t_); }
#define VAMD_OP_ADD(a, b) ((a) + (b))
#define VAMD_OP_OR(a, b) ((a) | (b))
#define VAMD_OP_IMAX(a, b) ((a) > (b) ? (a) : (b))
#define VAMD_OP_FMAXBITS(a, b) __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b)))
VAMD_DEV float wave_max(float x) {
  int v = __float_as_int(x);
  VAMD_DPP_SCAN_SELF(v, VAMD_OP_FMAXBITS)
  return __int_as_float(__builtin_amdgcn_readlane(v, 63));
}
VAMD_DEV int wave_sum(int v) {
  VAMD_DPP_SCAN(v, 0, VAMD_OP_ADD)
  return __builtin_amdgcn_readlane(v, 63);
}
VAMD_DEV int wave_scan_sum(int v) {  // inclusive prefix sum over the lanes
  VAMD_DPP_SCAN(v, 0, VAMD_OP_ADD)
  return v;
}
VAMD_DEV int wave_any(int pred) { return __any(pred); }
// the lanes for which `pred` holds, bit l = lane l; a value of lane `lane` (wave-uniform index: one v_readlane); a value
// of a lane of each lane's own choosing (ds_bpermute)
VAMD_DEV unsigned long long wave_ballot(bool pred) { return __ballot(pred); }
VAMD_DEV int wave_read(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
VAMD_DEV int wave_gather(int v, int lane) { return __shfl(v, lane, 64); }
VAMD_DEV unsigned long long wave_or64(unsigned long long x) {
  int lo = (int)(unsigned int)x, hi = (int)(unsigned int)(x >> 32);
  VAMD_DPP_SCAN_SELF(lo, VAMD_OP_OR)
  VAMD_DPP_SCAN_SELF(hi, VAMD_OP_OR)
  return ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(hi, 63) << 32) |
         (unsigned int)__builtin_amdgcn_readlane(lo, 63);
}
// inclusive prefix max over the lanes of the wave
VAMD_DEV int wave_scan_max(int v) {
  VAMD_DPP_SCAN_SELF(v, VAMD_OP_IMAX)
  return v;
}
VAMD_DEV int wave_shift_up1(int v, int fill) {  // lane l gets lane l-1's value, lane 0 gets `fill`
  return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
}
VAMD_DEV int wave_last(int v) { return __builtin_amdgcn_readlane(v, 63); }
VAMD_DEV int wave_first(int v) { return __builtin_amdgcn_readfirstlane(v); }
VAMD_DEV float f_from_bits(uint32_t u) { return __uint_as_float(u); }
VAMD_DEV uint32_t f_bits(float f) { return __float_as_uint(f); }
// order-free float max / min into LDS (seed scatter, fold minima): ds_max_f32 / ds_min_f32.
// No NaNs reach these (dB values), so the result is the plain maximum whatever the order.
VAMD_DEV void lds_atomic_max(float *p, float v) {
  (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
VAMD_DEV void lds_atomic_min(float *p, float v) {
  (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
VAMD_DEV void lds_atomic_add(int *p, int v) { atomicAdd(p, v); }
VAMD_DEV void lds_atomic_or(int *p, int v) { atomicOr(p, v); }
VAMD_DEV void lds_or_global_count(unsigned int *p) { atomicAdd(p, 1u); }  // an event counter in HBM (rare: error reports)

// Optional in-kernel stopwatch (measurement aid, off unless vamd_debug_cycles() armed it):
// lane 0 of every wave adds the shader-clock ticks spent since the previous mark to a slot.
struct PhaseClock {
  // ticks are summed in registers and flushed once per wave into one of 64 replicated
  // slot sets (spread by workgroup id), so the stopwatch itself costs a handful of
  // atomics per wave instead of one contended atomic per phase
  unsigned long long *slots;
  long long t;
  unsigned int acc[8];
  bool stoppable = false;
  VAMD_DEV void start(unsigned long long *s) {
    slots = s;
#pragma unroll
    for (int k = 0; k < 8; k++) acc[k] = 0;
    if (slots) t = clock64();
  }
  VAMD_DEV void mark(int k) {
#ifdef VAMD_STOP_AFTER  // scratch builds for counting a stage's instructions phase by phase (tools/floor_phases_pmc.sh)
    if (stoppable && k == VAMD_STOP_AFTER) __builtin_amdgcn_endpgm();
#endif
    if (slots) {
      const long long now = clock64();
      acc[k & 7] += (unsigned int)(now - t);
      t = now;
    }
  }
  VAMD_DEV void flush() {
    if (slots && LANE == 0) {
      unsigned long long *dst = slots + (size_t)(blockIdx.x & 63) * 80;
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (acc[k]) atomicAdd(dst + k, (unsigned long long)acc[k]);
    }
  }
};

// A small array (<= 64 entries) kept one entry per lane in a VGPR.  Reads with a
// wave-uniform index are a single v_readlane (no LDS round trip), which is what the
// ordered, wave-uniform sections (floor split loop, post settling) are bound by.
struct LaneInts {
  int v;
  VAMD_MEM int get(int i) const { return __builtin_amdgcn_readlane(v, i); }
  VAMD_MEM void set(int i, int x) { v = (LANE == i) ? x : v; }
  VAMD_MEM void fill(int x) { v = x; }
  VAMD_MEM void load(const int *__restrict__ p, int count) { v = LANE < count ? p[LANE] : 0; }
  VAMD_MEM int mine() const { return v; }
  // lane-per-entry access: at/put touch the calling lane's own entry i (== LANE), gather
  // reads any entry with a per-lane index (ds_bpermute)
  VAMD_MEM int at(int) const { return v; }
  VAMD_MEM void put(int, int x) { v = x; }
  VAMD_MEM int gather(int idx) const { return __shfl(v, idx, 64); }
  VAMD_MEM void load_shifted(const int *__restrict__ p, int shift, int count) {
    v = (LANE >= shift && LANE < count) ? p[LANE - shift] : 0;
  }
  // entries j = from-1, from-2, ... while == oldv become newv (the reference's "for(j=..;j>=0;j--)
  // if(a[j]==old)a[j]=new; else break;")
  VAMD_MEM void replace_run_down(int from, int oldv, int newv) {
    const unsigned long long eq = __ballot(v == oldv);
    const unsigned long long below = from >= 64 ? ~0ull : ((1ull << from) - 1ull);
    const unsigned long long stop = ~eq & below;  // entries below `from` that end the run
    const int first = stop ? 64 - __builtin_clzll(stop) : 0;
    if (LANE >= first && LANE < from) v = newv;
  }
  // entries j = from, from+1, ... < count while == oldv become newv
  VAMD_MEM void replace_run_up(int from, int count, int oldv, int newv) {
    const unsigned long long eq = __ballot(v == oldv);
    const unsigned long long range = (count >= 64 ? ~0ull : ((1ull << count) - 1ull)) & ~((1ull << from) - 1ull);
    const unsigned long long stop = ~eq & range;
    const int last = stop ? __builtin_ctzll(stop) : count;  // first entry that ends the run
    if (LANE >= from && LANE < last) v = newv;
  }
};



// For quotients below 2^12 (num < 2^24): the hardware's one-instruction reciprocal (1 ulp) leaves the product within
// a thousandth of the true quotient, far inside div_small's +-1 fix-up; the result is the exact floor either way.
VAMD_DEV float div_rcp_fast(int den) { return __builtin_amdgcn_rcpf((float)den); }
// products of operands that fit 24 signed bits (bin indices, line heights, their small quotients): one full-rate
// instruction
VAMD_DEV int mad24(int a, int b, int c) { return __mul24(a, b) + c; }
// floor(num / den) in one multiply for the line walks of floor 1, where den = x1 - x0 <= 2048 and
// num = k * |dy| with k < den and |dy| <= 1023: magic = ceil(2^32 / den) (derive_div_magic, den >= 2) makes
// (num * magic) >> 32 exact while num * (magic * den - 2^32) < 2^32, i.e. for num * (den - 1) < 2^32 -- here
// num * den < 2047 * 1023 * 2048 < 2^32.  den == 1 only ever divides num == 0 (k < den).
VAMD_DEV int div_magic(int num, unsigned int magic) { return (int)__umulhi((unsigned int)num, magic); }
// p[i] for a wave-uniform i out of a table in HBM: through the scalar cache (the compiler cannot tell that a pointer it
// read from a parameter struct points at constant memory, and would send all 64 lanes to fetch the same word)
VAMD_DEV unsigned int load_uniform_u32(const unsigned int *p, int i) {
  typedef const unsigned int __attribute__((address_space(4))) *cptr;
  return ((cptr)p)[__builtin_amdgcn_readfirstlane(i)];
}
// v_sqrt_f32: within one ulp of the root (k_couple's quant_energy settles the candidate exactly)
VAMD_DEV float approx_sqrtf(float x) { return __builtin_amdgcn_sqrtf(x); }
// v_rcp_f32: within one ulp of the reciprocal
VAMD_DEV float approx_rcpf(float x) { return __builtin_amdgcn_rcpf(x); }
// keeps a wave-uniform value opaque to the optimiser (k_floor: a known +-1 would turn a multiply-add into negate + select)
VAMD_DEV void keep_opaque(int &v) { asm volatile("" : "+s"(v)); }

}  // namespace vamd
#else
#define VAMD_GPU 0
#include "vamd_wave_host.h"  // the one-lane test vocabulary: tests/emul only, see the note at the top
#endif

namespace vamd {

struct alignas(16) F4 {
  float x, y, z, w;
};
struct alignas(16) I4 {
  int x, y, z, w;
};
struct alignas(8) F2 {
  float x, y;
};
struct alignas(8) I2 {
  int x, y;
};

// Exact floor(num/den) for 0 <= num < 2^24, 0 < den < 2^12 without the integer-divide
// expansion: one fp32 multiply by a precomputed reciprocal, then a +-1 fix-up.
VAMD_DEV float div_rcp(int den) { return 1.0f / (float)den; }
VAMD_DEV int div_small(int num, int den, float rcp) {
  int q = (int)((float)num * rcp);
  const int r = num - q * den;
  if (r < 0) q--;
  if (r >= den) q++;
  return q;
}

VAMD_DEV void f4_get(const F4 &v, float *a) { a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }
VAMD_DEV F4 f4_make(const float *a) { F4 v; v.x = a[0]; v.y = a[1]; v.z = a[2]; v.w = a[3]; return v; }

// ---- scalar helpers shared by all stages ------------------------------------

// todB(): reference lib/scales.h:43-51.  An integer-bit-trick log, not log10:
// the float's magnitude bits, converted to float, scaled and offset.  fp32 ops.
VAMD_DEV float todB(float x) {
  uint32_t i = f_bits(x) & 0x7fffffffu;
  return (float)i * 7.17711438e-7f - 764.6161886f;
}
// "todB(x) + .345" as the reference writes it: the .345 literal is a double
// (lib/mapping0.c:264,310,327,385), so the add is done in fp64 then rounded.
VAMD_DEV float todB_345(float x) { return (float)((double)todB(x) + .345); }

// unitnorm(): lib/scales.h:32-40 (+-1 with the sign of x)
VAMD_DEV float unitnorm(float x) { return f_from_bits((f_bits(x) & 0x80000000u) | 0x3f800000u); }

// vorbis_dBquant(): lib/floor1.c:273-278
VAMD_DEV int dBquant(float x) {
  int i = (int)(x * 7.3142857f + 1023.5f);
  if (i > 1023) return 1023;
  if (i < 0) return 0;
  return i;
}

// The peak of logfft over one run of bins [s, e) that share an octave line (seed_loop's inner maximum,
// lib/psy.c:429-440), as the reference compares: `if (f[i] > max) max = f[i]`.  Formed by the transform stage while
// the block's logfft is still in LDS, and handed to the tone stage a float per run instead of a float per bin.
VAMD_DEV float run_peak(const float *fft, int s, int e) {
  float mx = fft[s];
  for (int i = s + 1; i < e; i++)
    if (fft[i] > mx) mx = fft[i];
  return mx;
}

}  // namespace vamd
