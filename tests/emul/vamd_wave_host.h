// tests/emul/vamd_wave_host.h -- TEST BUILD ONLY.
//
// The vocabulary of vorbis_amd/csrc/vamd_wave.h for the host compiler: ONE lane (LANE = 0, NLANES = 1), so a
// kernel body's phases run as ordinary serial loops, reductions are identities and the per-lane arrays are plain
// arrays.  tests/emul compiles the product's kernel bodies against it to check their arithmetic and their
// re-formulations bit for bit against the oracle on a machine without a GPU.  Nothing under vorbis_amd/ can see this
// file (it is found through the test build's -I tests/emul only); it is not a fallback.
#pragma once
#include <math.h>
#include <string.h>
#define VAMD_DEV static inline
#define VAMD_HOSTDEV static inline
#define VAMD_MEM inline
#define VAMD_DEV_NOINLINE static
#define VAMD_CONST_TABLE static const
#define LANE 0
#define NLANES 1
#define WAVE_SYNC() ((void)0)
#define WAVE_SYNC_GLOBAL() ((void)0)
#define VAMD_QPL 1024
#define LANE_QUADS(kq, q, nq) for (int kq = 0, q = LANE; kq < VAMD_QPL && q < (nq); kq++, q += NLANES)
#define SLICE_QUADS(kq, q, q0, q1, QPS) for (int kq = 0, q = (q0) + LANE; kq < (QPS) && q < (q1); kq++, q += NLANES)
#define VAMD_QPL2 2048
#define LANE_QUADS2(kq, q, nq) for (int kq = 0, q = LANE; kq < VAMD_QPL2 && q < (nq); kq++, q += NLANES)
#define WAVE_FOR(i, count) for (int i = LANE; i < (count); i += NLANES)
#define LANE_BINS(k, i, i0, KPL, n) for (int k = 0, i = (i0); k < (KPL) && i < (n); k++, i++)
#define TEAM_FOR(i, count) for (int i = 0; i < (count); i++)
#define TEAM_RANGE(i, lo, hi) for (int i = (lo); i < (hi); i++)
#define TEAM_SYNC() ((void)0)
#define TEAM_FIRST_WAVE 1
#define TEAM_LEADER 1
#define TEAM_EACH(i, count, tm) for (int i = (tm).tid(); i < (count); i += (tm).size())
#define TEAM_QUADS(kq, q, nq, QPT, tm) for (int kq = 0, q = (tm).tid(); kq < (QPT) && q < (nq); kq++, q += (tm).size())

namespace vamd {

VAMD_DEV unsigned brev32(unsigned x) {  // bit 0 <-> bit 31
  unsigned r = 0;
  for (int b = 0; b < 32; b++) r |= ((x >> b) & 1u) << (31 - b);
  return r;
}
VAMD_DEV float wave_max(float v) { return v; }
VAMD_DEV int wave_sum(int v) { return v; }
VAMD_DEV int wave_any(int pred) { return pred != 0; }
VAMD_DEV unsigned long long wave_or64(unsigned long long v) { return v; }
VAMD_DEV int wave_scan_max(int v) { return v; }
VAMD_DEV int wave_scan_sum(int v) { return v; }
VAMD_DEV int wave_shift_up1(int v, int fill) { (void)v; return fill; }
VAMD_DEV int wave_last(int v) { return v; }
VAMD_DEV int wave_first(int v) { return v; }
VAMD_DEV float f_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
VAMD_DEV uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
VAMD_DEV void lds_atomic_max(float *p, float v) { if (*p < v) *p = v; }
VAMD_DEV void lds_atomic_min(float *p, float v) { if (v < *p) *p = v; }
VAMD_DEV void lds_atomic_add(int *p, int v) { *p += v; }
VAMD_DEV void lds_atomic_or(int *p, int v) { *p |= v; }
VAMD_DEV void lds_or_global_count(unsigned int *p) { *p += 1u; }
// Optional in-kernel stopwatch (measurement aid, off unless vamd_debug_cycles() armed it):
// lane 0 of every wave adds the shader-clock ticks spent since the previous mark to a slot.
struct PhaseClock {
  VAMD_DEV void start(unsigned long long *) {}
  VAMD_DEV void mark(int) {}
  VAMD_DEV void flush() {}
};

struct LaneInts {
  int a[64];
  VAMD_MEM int get(int i) const { return a[i]; }
  VAMD_MEM void set(int i, int x) { a[i] = x; }
  VAMD_MEM void fill(int x) { for (int i = 0; i < 64; i++) a[i] = x; }
  VAMD_MEM void load(const int *p, int count) { for (int i = 0; i < 64; i++) a[i] = i < count ? p[i] : 0; }
  VAMD_MEM int at(int i) const { return a[i]; }
  VAMD_MEM void put(int i, int x) { a[i] = x; }
  VAMD_MEM int gather(int idx) const { return a[idx]; }
  VAMD_MEM void load_shifted(const int *p, int shift, int count) {
    for (int i = 0; i < 64; i++) a[i] = (i >= shift && i < count) ? p[i - shift] : 0;
  }
  VAMD_MEM void replace_run_down(int from, int oldv, int newv) {
    for (int j = from - 1; j >= 0; j--) {
      if (a[j] != oldv) break;
      a[j] = newv;
    }
  }
  VAMD_MEM void replace_run_up(int from, int count, int oldv, int newv) {
    for (int j = from; j < count; j++) {
      if (a[j] != oldv) break;
      a[j] = newv;
    }
  }
};



VAMD_DEV float div_rcp_fast(int den) { return 1.0f / (float)den; }
VAMD_DEV int mad24(int a, int b, int c) { return a * b + c; }
VAMD_DEV int div_magic(int num, unsigned int magic) { return (int)(((unsigned long long)(unsigned int)num * magic) >> 32); }
VAMD_DEV unsigned int load_uniform_u32(const unsigned int *p, int i) { return p[i]; }
// The hardware's estimates (v_sqrt_f32, v_rcp_f32) are within one ulp and nothing more is promised: the test build
// returns the correctly rounded value moved by -1, 0 or +1 ulp, picked by a hash of the argument's bits (or as emul_couple_estimate_check says), so that code
// which leans on more than "within one ulp" fails here.
static int g_estimate_nudge = 2;  // -1, 0, +1: every estimate moved that way; 2: by the hash
VAMD_DEV float nudge_one_ulp(float exact, float arg) {
  unsigned int a, e;
  memcpy(&a, &arg, 4);
  memcpy(&e, &exact, 4);
  if (!(exact == exact) || (e & 0x7f800000u) == 0x7f800000u || (e & 0x7fffffffu) < 0x00800001u) return exact;
  a = (a ^ (a >> 15)) * 0x2c1b3c6du;
  a ^= a >> 12;
  e += g_estimate_nudge == 2 ? (int)(a % 3u) - 1 : g_estimate_nudge;
  float r;
  memcpy(&r, &e, 4);
  return r;
}
VAMD_DEV float approx_sqrtf(float x) { return nudge_one_ulp(sqrtf(x), x); }
VAMD_DEV float approx_rcpf(float x) { return nudge_one_ulp(1.0f / x, x); }
VAMD_DEV void keep_opaque(int &) {}

}  // namespace vamd
