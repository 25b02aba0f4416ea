// vamd_wave_pair.h -- the execution vocabulary of vamd_wave.h for TWO channel-blocks per wavefront: lanes 0-31 own one,
// lanes 32-63 the other (the two channels of one stereo block: same block type, window flags, floor and psy tables, so
// everything "per block" stays wave-uniform and only what is per channel differs between the halves).
//
// Why.  The floor stage is bound by vector-instruction issue, and a third of its instructions
// are the ordered sections -- the greedy split loop, the level loops of post settling and quantise/predict, the
// bookkeeping between them -- in which 64 lanes carry one channel's decisions (and, on a short block's 128 bins, half the
// lanes of every per-bin phase have nothing to do).  Here one pass through those sections serves two channel-blocks:
// the loop orders are static (lib/floor1.c:625-697 walks i = 2 .. posts-1 whatever the data), only the branches are
// data-dependent, and a half that does not take a branch sits it out under the exec mask.  posts <= 32.
//
// How.  k_floor.inc and k_tone_fold.inc are written against names, not against the wave: LANE, NLANES, WAVE_FOR,
// wave_sum / wave_any / wave_ballot / wave_read / ..., LaneInts.  This header declares namespace vamd::pair, defines
// those names for a half (LANE = lane within the half, NLANES = 32; reductions and scans stay on the DPP path, confined
// to the half; a "wave-uniform" index is now half-uniform, so v_readlane becomes ds_bpermute), and includes the two
// bodies a second time inside it.  Every collective below is called with whole halves active: control flow in the
// bodies diverges only on per-channel conditions.
#pragma once
#include "vamd_wave.h"
#include "vamd_params.h"

#if VAMD_GPU
namespace vamd {
namespace pair {

#define VAMD_PAIR_HALF ((int)((threadIdx.x >> 5) & 1))
#define VAMD_PAIR_BASE ((int)(threadIdx.x & 32))

// inclusive scan inside each half: Hillis-Steele inside the rows of 16, then row_bcast:15 into the odd rows
#define VAMD_DPP_SCAN32(v, ident, OP)                                                \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x111, 0xf, 0xf, false));          \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x112, 0xf, 0xf, false));          \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x114, 0xf, 0xf, false));          \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x118, 0xf, 0xf, false));          \
  v = OP(v, __builtin_amdgcn_update_dpp(ident, v, 0x142, 0xa, 0xf, false));
#define VAMD_DPP_SCAN32_SELF(v, OP)                                                  \
  { int t_; \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false); v = OP(v, t_);     \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false); v = OP(v, t_);     \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false); v = OP(v, t_);     \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false); v = OP(v, t_); \
  t_ = __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false); v = OP(v, t_); }

// the value the LAST lane of this lane's half holds (two v_readlane and a select: no trip to the LDS pipe)
VAMD_DEV int half_last(int v) {
  const int a = __builtin_amdgcn_readlane(v, 31), b = __builtin_amdgcn_readlane(v, 63);
  return VAMD_PAIR_HALF ? b : a;
}
VAMD_DEV int wave_sum(int v) {
  VAMD_DPP_SCAN32(v, 0, VAMD_OP_ADD)
  return half_last(v);
}
VAMD_DEV int wave_scan_sum(int v) {
  VAMD_DPP_SCAN32(v, 0, VAMD_OP_ADD)
  return v;
}
VAMD_DEV int wave_scan_max(int v) {
  VAMD_DPP_SCAN32_SELF(v, VAMD_OP_IMAX)
  return v;
}
VAMD_DEV int wave_last(int v) { return half_last(v); }
VAMD_DEV int wave_first(int v) {
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 32);
  return VAMD_PAIR_HALF ? b : a;
}
VAMD_DEV unsigned long long wave_ballot(bool pred) {  // bit l = lane l of this lane's half
  const unsigned long long b = __ballot(pred);
  return VAMD_PAIR_HALF ? (b >> 32) : (b & 0xffffffffull);
}
VAMD_DEV int wave_any(int pred) { return wave_ballot(pred != 0) != 0ull; }
VAMD_DEV unsigned long long wave_or64(unsigned long long x) {  // (posts <= 32: the low word holds everything the floor ors)
  int lo = (int)(unsigned int)x, hi = (int)(unsigned int)(x >> 32);
  VAMD_DPP_SCAN32_SELF(lo, VAMD_OP_OR)
  VAMD_DPP_SCAN32_SELF(hi, VAMD_OP_OR)
  return ((unsigned long long)(unsigned int)half_last(hi) << 32) | (unsigned int)half_last(lo);
}
VAMD_DEV int wave_shift_up1(int v, int fill) {  // lane l of a half gets lane l-1's value, its lane 0 gets `fill`
  const int r = __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);  // wave_shr:1 (lane 32 would get lane 31's)
  return (threadIdx.x & 31) == 0 ? fill : r;
}
// a value of lane `lane` of this lane's half; the index is the same for the lanes of a half, not for the wave.
// Two forms: through the LDS pipe (ds_bpermute: one instruction, a round trip of ~100 cycles on the ordered chain), or on
// the vector unit alone -- the two halves' indices to scalar registers, two v_readlane, one select: five instructions,
// no trip (VAMD_PAIR_READLANE; which one is measured in profiles/r05_floor_pair.txt)
#ifndef VAMD_PAIR_READLANE
#define VAMD_PAIR_READLANE 0
#endif
VAMD_DEV int half_read(int v, int lane) {
#if VAMD_PAIR_READLANE
  const int i0 = __builtin_amdgcn_readlane(lane, 0) & 31, i1 = __builtin_amdgcn_readlane(lane, 32) & 31;
  const int a = __builtin_amdgcn_readlane(v, i0), b = __builtin_amdgcn_readlane(v, 32 + i1);
  return VAMD_PAIR_HALF ? b : a;
#else
  return __shfl(v, VAMD_PAIR_BASE + lane, 64);
#endif
}
VAMD_DEV int wave_read(int v, int lane) { return half_read(v, lane); }
VAMD_DEV int wave_gather(int v, int lane) { return __shfl(v, VAMD_PAIR_BASE + lane, 64); }
VAMD_DEV unsigned int load_uniform_u32(const unsigned int *p, int i) { return p[i]; }  // (half-uniform index: a vector load)
VAMD_DEV void keep_opaque(int &v) { asm volatile("" : "+v"(v)); }

// A small array (<= 32 entries) kept one entry per lane of the half
struct LaneInts {
  int v;
  VAMD_MEM int get(int i) const { return half_read(v, i); }
  VAMD_MEM void set(int i, int x) { v = ((int)(threadIdx.x & 31) == i) ? x : v; }
  VAMD_MEM void fill(int x) { v = x; }
  VAMD_MEM void load(const int *__restrict__ p, int count) { v = (int)(threadIdx.x & 31) < count ? p[threadIdx.x & 31] : 0; }
  VAMD_MEM int mine() const { return v; }
  VAMD_MEM int at(int) const { return v; }
  VAMD_MEM void put(int, int x) { v = x; }
  VAMD_MEM int gather(int idx) const { return __shfl(v, VAMD_PAIR_BASE + idx, 64); }
  VAMD_MEM void load_shifted(const int *__restrict__ p, int shift, int count) {
    const int l = (int)(threadIdx.x & 31);
    v = (l >= shift && l < count) ? p[l - shift] : 0;
  }
  VAMD_MEM void replace_run_down(int from, int oldv, int newv) {
    const unsigned int eq = (unsigned int)wave_ballot(v == oldv);
    const unsigned int below = from >= 32 ? ~0u : ((1u << from) - 1u);
    const unsigned int stop = ~eq & below;  // entries below `from` that end the run
    const int first = stop ? 32 - __builtin_clz(stop) : 0;
    const int l = (int)(threadIdx.x & 31);
    if (l >= first && l < from) v = newv;
  }
  VAMD_MEM void replace_run_up(int from, int count, int oldv, int newv) {
    const unsigned int eq = (unsigned int)wave_ballot(v == oldv);
    const unsigned int range = (count >= 32 ? ~0u : ((1u << count) - 1u)) & ~((1u << from) - 1u);
    const unsigned int stop = ~eq & range;
    const int last = stop ? __builtin_ctz(stop) : count;  // first entry that ends the run
    const int l = (int)(threadIdx.x & 31);
    if (l >= from && l < last) v = newv;
  }
};

// ---- the bodies once more, against the names above -- as static members of a struct: a call from one member to another
// then stops at class scope, where argument-dependent lookup would otherwise also offer the wave form of the same name
// (their parameter types live in namespace vamd)
struct Bodies {
#undef VAMD_DEV
#define VAMD_DEV static __device__ __forceinline__
#undef LANE
#undef NLANES
#undef VAMD_QPL
#define LANE ((int)(threadIdx.x & 31))
#define NLANES 32
#define VAMD_QPL 8  // quads of a block per lane: 1024 bins -> 256 quads over 32 lanes
#include "k_tone_fold.inc"
#include "k_floor.inc"
#undef LANE
#undef NLANES
#undef VAMD_QPL
#define LANE ((int)(threadIdx.x & 63))
#define NLANES 64
#define VAMD_QPL 4
#undef VAMD_DEV
#define VAMD_DEV __device__ __forceinline__
};

}  // namespace pair
}  // namespace vamd
#endif
