// Whole-chip VALU issue rate with every workgroup's life on record.
// tools/micro/valu_rate.hip's chip rows divide a kernel's wall time by the instructions issued; a wave's own
// s_memtime view of the same kernel is 40 % shorter.  Here every workgroup writes where it ran (XCC, SE, CU),
// when it started and stopped on the chip-wide 100 MHz clock (s_memrealtime) and how many shader ticks
// (s_memtime) it lived, so that the host can lay the lives out on one time axis: resident workgroups over the
// kernel, workgroups per CU, the ramp at either end, and the rate INSIDE the fully occupied stretch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/chip_rate tools/micro/chip_rate.hip
//   tools/micro/chip_rate            (the standard table)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum { OP_ADD, OP_FMA, OP_PKFMA, OP_FMA64, OP_ADD_5LANES, OP_ADD_1LANE, OP_MIX, OP_COUNT };
static const char *names[] = {"v_add_f32", "v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_add_f32, 5 lanes live", "v_add_f32, 1 lane live",
                              "v_add_f32 / s_add_u32 alternating"};

struct Life {
  unsigned long long rt0, rt1;  // s_memrealtime, 100 MHz, one counter for the chip
  unsigned long long ticks;     // s_memtime over the loop
  unsigned hw, xcc;
};

template <int OP>
__global__ void k_life(int iters, Life *out, float *sink) {
  float r[16];
  v2f p[16];
  double d[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    r[i] = 1.0f + 1e-3f * (threadIdx.x + i);
    p[i] = v2f{r[i], r[i] * 0.5f};
    d[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  }
  float c = 1.0000001f;
  v2f pc = {c, c};
  double dc = 1.0000001;
  unsigned sacc = 0;
  __syncthreads();
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long full = __builtin_amdgcn_read_exec();
  if (OP == OP_ADD_5LANES) asm volatile("s_mov_b64 exec, %0" ::"s"(0x0101010101ull));  // lanes 0, 8, 16, 24, 32
  if (OP == OP_ADD_1LANE) asm volatile("s_mov_b64 exec, 1");
  for (int it = 0; it < iters; it++) {
    if (OP == OP_ADD || OP == OP_ADD_5LANES || OP == OP_ADD_1LANE) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
      REP16(X)
#undef X
    } else if (OP == OP_FMA64) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dc));
      REP16(X)
#undef X
    } else if (OP == OP_MIX) {  // 16 vector + 16 scalar instructions: does a scalar instruction take a vector slot?
#define X(i) asm volatile("v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, 3" : "+v"(r[i]), "+s"(sacc) : "v"(c));
      REP16(X)
#undef X
    }
  }
  asm volatile("s_mov_b64 exec, %0" ::"s"(full));
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    Life l;
    l.rt0 = w0, l.rt1 = w1, l.ticks = t1 - t0, l.hw = hw, l.xcc = xcc & 0xf;
    out[blockIdx.x] = l;
  }
  float s = (float)sacc;
  constexpr bool PK = OP == OP_PKFMA;
  constexpr bool F64 = OP == OP_FMA64;
#pragma unroll
  for (int i = 0; i < 16; i++) s += PK ? p[i].x + p[i].y : (F64 ? (float)d[i] : r[i]);
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

// ---- the price list: one instruction kind per row, 8192 one-wave workgroups (eight per SIMD), first-in..last-out ----------------
#define OPLIST(F)                                                                                          \
  F(0, "v_add_f32", "v_add_f32 %0, %0, %1")                                                                 \
  F(1, "v_mul_f32", "v_mul_f32 %0, %0, %1")                                                                 \
  F(2, "v_fma_f32", "v_fma_f32 %0, %0, %1, %1")                                                             \
  F(3, "v_fmac_f32", "v_fmac_f32 %0, %1, %1")                                                               \
  F(4, "v_max_f32", "v_max_f32 %0, %0, %1")                                                                 \
  F(5, "v_max3_f32", "v_max3_f32 %0, %0, %1, %1")                                                           \
  F(6, "v_add_u32", "v_add_u32 %0, %0, %1")                                                                 \
  F(7, "v_and_b32", "v_and_b32 %0, %0, %1")                                                                 \
  F(8, "v_lshlrev_b32", "v_lshlrev_b32 %0, 1, %0")                                                          \
  F(9, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 1, %1")                                                    \
  F(10, "v_add3_u32", "v_add3_u32 %0, %0, %1, %1")                                                          \
  F(11, "v_bfe_u32", "v_bfe_u32 %0, %0, 3, 9")                                                              \
  F(12, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1")                                                          \
  F(13, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %1")                                                          \
  F(14, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1")                                                        \
  F(15, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %1")                                                    \
  F(16, "v_mad_u64_u32 (pair)", "v_mad_u64_u32 %2, vcc, %0, %1, %2")                                        \
  F(17, "v_cmp_lt_f32 -> vcc", "v_cmp_lt_f32 vcc, %0, %1")                                                  \
  F(18, "v_cmp_lt_f32 -> sgpr pair", "v_cmp_lt_f32 s[20:21], %0, %1")                                       \
  F(19, "v_cndmask_b32 (sgpr pair)", "v_cndmask_b32 %0, %0, %1, s[20:21]")                                  \
  F(20, "v_cmp + v_cndmask (vcc)", "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc")             \
  F(21, "v_mov_b32", "v_mov_b32 %0, %1")                                                                    \
  F(22, "v_mov_b32 dpp row_shr:1", "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
  F(23, "v_add_f32 dpp row_shr:1", "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") \
  F(24, "v_readlane_b32", "v_readlane_b32 s20, %0, 3")                                                      \
  F(25, "v_readfirstlane_b32", "v_readfirstlane_b32 s20, %0")                                               \
  F(26, "v_writelane_b32", "v_writelane_b32 %0, s20, 5")                                                    \
  F(27, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0")                                                            \
  F(28, "v_cvt_i32_f32", "v_cvt_i32_f32 %0, %0")                                                            \
  F(29, "v_rcp_f32", "v_rcp_f32 %0, %0")                                                                    \
  F(30, "v_sqrt_f32", "v_sqrt_f32 %0, %0")                                                                  \
  F(31, "v_log_f32", "v_log_f32 %0, %0")                                                                    \
  F(32, "v_div_scale_f32", "v_div_scale_f32 %0, vcc, %0, %1, %0")                                           \
  F(33, "v_div_fmas_f32", "v_div_fmas_f32 %0, %0, %1, %1")                                                  \
  F(34, "v_div_fixup_f32", "v_div_fixup_f32 %0, %0, %1, %1")                                                \
  F(35, "v_add_f64", "v_add_f64 %2, %2, %3")                                                                \
  F(36, "v_mul_f64", "v_mul_f64 %2, %2, %3")                                                                \
  F(37, "v_fma_f64", "v_fma_f64 %2, %2, %3, %3")                                                            \
  F(38, "v_cvt_f64_f32", "v_cvt_f64_f32 %2, %0")                                                            \
  F(39, "v_cvt_f32_f64", "v_cvt_f32_f64 %0, %2")                                                            \
  F(40, "v_rcp_f64", "v_rcp_f64 %2, %2")                                                                    \
  F(41, "v_sqrt_f64", "v_sqrt_f64 %2, %2")                                                                  \
  F(42, "v_pk_add_f32", "v_pk_add_f32 %2, %2, %3")                                                          \
  F(43, "v_pk_mul_f32", "v_pk_mul_f32 %2, %2, %3")                                                          \
  F(44, "v_pk_fma_f32", "v_pk_fma_f32 %2, %2, %3, %3")                                                      \
  F(45, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %1")                                                            \
  F(46, "v_perm_b32", "v_perm_b32 %0, %0, %1, %1")                                                          \
  F(47, "v_alignbit_b32", "v_alignbit_b32 %0, %0, %1, 7")                                                   \
  F(48, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %1, %0")                                              \
  F(49, "s_add_u32 (scalar only)", "s_add_u32 s20, s20, 3")                                                 \
  F(50, "v_add_f32 + s_add_u32", "v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 3")                           \
  F(51, "v_add_f32 + 2 x s_add_u32", "v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 3\n s_and_b32 s21, s20, 7") \
  F(52, "v_add_f32 + s_nop 0", "v_add_f32 %0, %0, %1\n s_nop 0")                                           \
  F(53, "v_add_f32, literal operand", "v_add_f32 %0, 0x3f800001, %0")                                       \
  F(54, "v_add_f32, sgpr operand", "v_add_f32 %0, s22, %0")                                                 \
  F(55, "v_add_co_u32 + v_addc_co_u32", "v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc") \
  F(56, "v_sub_f32 + v_mul_f32 (dependent pair)", "v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1")           \
  F(57, "v_trunc_f32", "v_trunc_f32 %0, %0")                                                                \
  F(58, "v_rndne_f32", "v_rndne_f32 %0, %0")                                                                \
  F(59, "v_med3_f32", "v_med3_f32 %0, %0, %1, %1")                                                          \
  F(60, "v_xor_b32", "v_xor_b32 %0, %0, %1")                                                                \
  F(61, "v_bfi_b32", "v_bfi_b32 %0, %1, %0, %1")                                                            \
  F(62, "v_ashrrev_i32", "v_ashrrev_i32 %0, 1, %0")                                                         \
  F(63, "v_min_u32", "v_min_u32 %0, %0, %1")

static const char *price_names[] = {
#define F(i, n, a) n,
    OPLIST(F)
#undef F
};
static const int price_insts[] = {  // vector instructions per asm statement
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1};

template <int OP>
__global__ void k_price(int iters, Life *out, float *sink) {
  float r[16];
  double d[16];
#pragma unroll
  for (int i = 0; i < 16; i++) r[i] = 1.0f + 1e-3f * (threadIdx.x + i), d[i] = 1.0 + 1e-3 * (threadIdx.x + i);
  float c = 1.0000001f;
  double dc = 1.0000001;
  asm volatile("s_mov_b32 s22, 0x3f800001\n s_mov_b32 s20, 0\n s_mov_b32 s21, 0" ::: "s20", "s21", "s22");
  const unsigned long long w0 = wall_clock64();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#define F(i, n, a)                                                                                                       \
  if (OP == i) {                                                                                                         \
    _Pragma("unroll") for (int k = 0; k < 16; k++) asm volatile(a : "+v"(r[k]) : "v"(c), "v"(d[k]), "v"(dc) : "vcc", "s20", "s21"); \
  }
    OPLIST(F)
#undef F
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  if (threadIdx.x == 0) {
    Life l;
    l.rt0 = w0, l.rt1 = w1, l.ticks = t1 - t0, l.hw = 0, l.xcc = 0;
    out[blockIdx.x] = l;
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += r[i] + (float)d[i];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int OP>
static void price(Life *dl, float *sink) {
  const int grid = 8192, iters = 4000;
  hipLaunchKernelGGL((k_price<OP>), dim3(grid), dim3(64), 0, 0, iters, dl, sink);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k_price<OP>), dim3(grid), dim3(64), 0, 0, iters, dl, sink);
  hipDeviceSynchronize();
  std::vector<Life> L(grid);
  hipMemcpy(L.data(), dl, sizeof(Life) * grid, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  double ticks = 0, life = 0, tmin = 1e30;
  for (auto &l : L) {
    lo = std::min(lo, l.rt0), hi = std::max(hi, l.rt1);
    ticks += (double)l.ticks, life += (l.rt1 - l.rt0) / 100.0;
    tmin = std::min(tmin, (double)l.ticks);
  }
  const double span_us = (hi - lo) / 100.0, mhz = ticks / life;
  const double stmts = (double)iters * 16;
  // per SIMD: eight waves' statements over the span
  printf("%-40s %6.2f cycles per statement per SIMD (%d vector instruction%s in it)   lone-wave pace %6.2f   clock %4.0f MHz\n", price_names[OP],
         span_us * mhz / (8 * stmts), price_insts[OP], price_insts[OP] == 1 ? "" : "s", tmin / stmts, mhz);
}

template <int OP>
static void price_all(Life *dl, float *sink) {
  price<OP>(dl, sink);
  if constexpr (OP + 1 < 64) price_all<OP + 1>(dl, sink);
}

template <int OP>
static void run(Life *dl, float *sink, int grid, int threads, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_life<OP>), dim3(grid), dim3(threads), 0, 0, iters, dl, sink);  // warm
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_life<OP>), dim3(grid), dim3(threads), 0, 0, iters, dl, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<Life> L(grid);
  hipMemcpy(L.data(), dl, sizeof(Life) * grid, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (auto &l : L) lo = std::min(lo, l.rt0), hi = std::max(hi, l.rt1);
  const double span_us = (hi - lo) / 100.0;
  // resident workgroups over time (10 ns steps), and the fully resident stretch: where >= 95 % of the maximum is resident
  if (hi - lo > 100000000ull) {  // (a stale or torn record: say so instead of sizing a vector by it)
    printf("%-34s grid %5d x %4d: implausible span, first-in %llu last-out %llu\n", names[OP], grid, threads, lo, hi);
    return;
  }
  const int T = (int)(hi - lo) + 2;
  std::vector<int> res(T, 0);
  for (auto &l : L) {
    res[l.rt0 - lo]++;
    res[l.rt1 - lo + 1]--;
  }
  int cur = 0, mx = 0;
  for (int t = 0; t < T; t++) cur += res[t], res[t] = cur, mx = std::max(mx, cur);
  int full_lo = -1, full_hi = -1;
  double area = 0;
  for (int t = 0; t < T; t++) {
    area += res[t];
    if (res[t] * 100 >= mx * 95) {
      if (full_lo < 0) full_lo = t;
      full_hi = t;
    }
  }
  std::map<unsigned, int> per_cu;
  for (auto &l : L) per_cu[(l.xcc << 16) | ((l.hw >> 8) & 0xfff)]++;  // CU_ID[11:8] SH_ID[12] SE_ID[15:13] (+ xcc)
  int cmin = 1 << 30, cmax = 0;
  for (auto &kv : per_cu) cmin = std::min(cmin, kv.second), cmax = std::max(cmax, kv.second);
  double life_us = 0, ticks = 0, life_min = 1e30, life_max = 0;
  for (auto &l : L) {
    const double u = (l.rt1 - l.rt0) / 100.0;
    life_us += u, ticks += (double)l.ticks, life_min = std::min(life_min, u), life_max = std::max(life_max, u);
  }
  life_us /= grid, ticks /= grid;
  const int waves = threads / 64;
  const double wave_insts = (double)iters * 16;                 // vector instructions of one wave
  const double insts = wave_insts * waves * grid;               // wave-instructions of the launch
  const double simds = 1024;
  printf("%-34s grid %5d x %4d  event %8.3f ms  first-in..last-out %8.1f us | life of a workgroup: mean %8.1f us (min %.1f max %.1f), %6.2f ticks per own "
         "instruction, tick rate %4.0f MHz | resident workgroups: max %d, mean over the span %.1f, >=95%% of max for %.1f us | %zu CUs seen, "
         "workgroups per CU %d..%d\n",
         names[OP], grid, threads, ms, span_us, life_us, life_min, life_max, ticks / wave_insts, ticks / (life_us * 1e-6) / 1e6, mx,
         area / T, (full_hi - full_lo + 1) / 100.0, per_cu.size(), cmin, cmax);
  // The rate: every instruction of the launch over the first-in..last-out span at the measured tick rate.  A wave's OWN ticks per
  // instruction say nothing about the SIMD's rate: arbitration is oldest-first, so the first workgroups run at a lone wave's pace and
  // leave, the youngest wait -- lives are spread linearly between `min` and the whole span (decile table below).
  const double mhz = ticks / (life_us * 1e-6) / 1e6;
  const double cyc_span = span_us * mhz * simds / insts;
  std::vector<double> ends;
  for (auto &l : L) ends.push_back((l.rt1 - lo) / 100.0);
  std::sort(ends.begin(), ends.end());
  printf("    -> %.2f shader cycles per wave-instruction per SIMD (all %.3g instructions / 1024 SIMDs over the span at %.0f MHz; %.2f from the "
         "event time at a nominal 2.4 GHz); the first workgroup to leave ran at %.2f ticks per own instruction; workgroups gone after "
         "10..100 %% of the span:",
         cyc_span, insts, mhz, 2.4e9 * simds / (insts / (ms * 1e-3)), life_min * mhz / wave_insts);
  for (int k = 1; k <= 10; k++) printf(" %d", (int)(std::upper_bound(ends.begin(), ends.end(), span_us * k / 10.0) - ends.begin()));
  printf("\n");
}

int main(int argc, char **argv) {
  Life *d;
  float *sink;
  const int maxgrid = 1 << 16;
  hipMalloc(&d, sizeof(Life) * maxgrid);
  hipMalloc(&sink, 1 << 20);
  if (argc > 1 && !strcmp(argv[1], "prices")) {
    printf("# price list: 8192 one-wave workgroups (eight waves per SIMD, every SIMD of the chip), 4000 rounds of 16 independent statements each;\n"
           "# cycles per statement per SIMD = first-in..last-out span x measured tick rate / (8 waves x statements)\n");
    price_all<0>(d, sink);
    return 0;
  }
  // (e) few lanes live: what k_noise's ordered walks issue
  run<OP_ADD_5LANES>(d, sink, 8192, 64, 20000);
  run<OP_ADD_1LANE>(d, sink, 8192, 64, 20000);
  // (a) r03's geometry: 512 workgroups of 16 waves, 20 000 rounds
  run<OP_ADD>(d, sink, 512, 1024, 20000);
  run<OP_FMA>(d, sink, 512, 1024, 20000);
  run<OP_PKFMA>(d, sink, 512, 1024, 20000);
  run<OP_FMA64>(d, sink, 512, 1024, 20000);
  // (b) the same work as one-wave workgroups, eight per SIMD at once: k_floor's geometry
  run<OP_ADD>(d, sink, 8192, 64, 20000);
  run<OP_FMA>(d, sink, 8192, 64, 20000);
  // (c) four and two waves per SIMD (is the ceiling the SIMD's or the chip's?)
  run<OP_FMA>(d, sink, 4096, 64, 20000);
  run<OP_FMA>(d, sink, 2048, 64, 20000);
  run<OP_FMA>(d, sink, 1024, 64, 20000);
  // (d) half the CUs busy (128 workgroups of 16 waves, then 256): does a CU run faster when its neighbours idle?
  run<OP_FMA>(d, sink, 128, 1024, 20000);
  run<OP_FMA>(d, sink, 256, 1024, 20000);
  // (g) many short workgroups: 262 144 one-wave workgroups of 40 rounds (a k_floor-sized launch): dispatch included
  run<OP_FMA>(d, sink, 65536, 64, 2500);
  return 0;
}
