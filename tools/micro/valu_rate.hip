// What does a wave64 VALU instruction cost on gfx950?  (the guide says CDNA4 SIMDs
// are 32 lanes wide = 2 cycles per wave64 instruction; SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU read ~1
// quad-cycle = 4.)  One workgroup on one CU, w waves per SIMD (blockDim = 256 w), every wave runs
// `iters` rounds of 16 instructions of one kind on 16 independent registers (or ONE register for the
// dependent-latency rows).  Reported: shader cycles (s_memtime) per instruction per SIMD, i.e.
// elapsed cycles / (instructions issued by the waves of one SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_rate tools/micro/valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum { OP_ADD, OP_ADD_E64, OP_FMAC, OP_MUL, OP_FMA, OP_PKMUL, OP_PKADD, OP_PKFMA, OP_ADD64, OP_MUL64, OP_FMA64, OP_RCP, OP_SQRT64,
       OP_RCP64, OP_CND, OP_MULLO, OP_ADDDPP, OP_MOVDPP, OP_READLANE, OP_CVT, OP_DEP_ADD, OP_DEP_ADDDPP, OP_DEP_MOVDPP_ADD,
       OP_DEP_FMA64, OP_MAX3, OP_COUNT };
static const char *names[] = {"v_add_f32", "v_add_f32_e64 (8-byte encoding)", "v_fmac_f32_e32 (4-byte fma)", "v_mul_f32", "v_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_pk_fma_f32",
                              "v_add_f64", "v_mul_f64", "v_fma_f64", "v_rcp_f32", "v_sqrt_f64", "v_rcp_f64", "v_cndmask_b32",
                              "v_mul_lo_u32", "v_add_f32 dpp row_shr:1", "v_mov_b32 dpp row_shr:1", "v_readlane_b32",
                              "v_cvt_f32_i32", "DEPENDENT v_add_f32", "DEPENDENT v_add_f32 dpp", "DEPENDENT v_mov dpp + v_add",
                              "DEPENDENT v_fma_f64", "v_max3_f32"};

template <int OP>
__global__ void k_rate(int iters, unsigned long long *out, float *sink) {
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
  int sg = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    if (OP == OP_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_ADD_E64) {
#define X(i) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_FMAC) {
#define X(i) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
      REP16(X)
#undef X
    } else if (OP == OP_PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
      REP16(X)
#undef X
    } else if (OP == OP_PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
      REP16(X)
#undef X
    } else if (OP == OP_ADD64) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
      REP16(X)
#undef X
    } else if (OP == OP_MUL64) {
#define X(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
      REP16(X)
#undef X
    } else if (OP == OP_FMA64) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dc));
      REP16(X)
#undef X
    } else if (OP == OP_RCP) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
      REP16(X)
#undef X
    } else if (OP == OP_SQRT64) {
#define X(i) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[i]));
      REP16(X)
#undef X
    } else if (OP == OP_RCP64) {
#define X(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
      REP16(X)
#undef X
    } else if (OP == OP_CND) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c) : "vcc");
      REP16(X)
#undef X
    } else if (OP == OP_MULLO) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_ADDDPP) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_MOVDPP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_READLANE) {
#define X(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(r[i]));
      REP16(X)
#undef X
    } else if (OP == OP_CVT) {
#define X(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(r[i]));
      REP16(X)
#undef X
    } else if (OP == OP_MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_DEP_ADD) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[0]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_DEP_ADDDPP) {
#define X(i) asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[0]) : "v"(c));
      REP16(X)
#undef X
    } else if (OP == OP_DEP_MOVDPP_ADD) {  // 8 x (add, then hand the value to the next lane): 16 instructions
#define X(i) asm volatile("v_add_f32 %0, %0, %1\n s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[0]) : "v"(c));
      X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#undef X
    } else if (OP == OP_DEP_FMA64) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[0]) : "v"(dc));
      REP16(X)
#undef X
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long w1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    out[2 * (threadIdx.x >> 6)] = t1 - t0;
    out[2 * (threadIdx.x >> 6) + 1] = w1 - w0;
  }
  if (threadIdx.x == 0) {  // where this workgroup ran: HW_ID (cu/sh/se) and the XCC id
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[40 + (blockIdx.x & 7)] = ((unsigned long long)(xcc & 0xf) << 32) | ((hw >> 8) & 0xff);
  }
  // (only the registers the instruction under test uses stay alive: with all 80 of them a 1024-thread workgroup
  // would be alone on its CU and the "whole chip" rows would measure four waves per SIMD, not eight)
  float s = (float)sg;
  constexpr bool PK = OP == OP_PKMUL || OP == OP_PKADD || OP == OP_PKFMA;
  constexpr bool F64 = OP == OP_ADD64 || OP == OP_MUL64 || OP == OP_FMA64 || OP == OP_SQRT64 || OP == OP_RCP64 || OP == OP_DEP_FMA64;
#pragma unroll
  for (int i = 0; i < 16; i++) s += PK ? p[i].x + p[i].y : (F64 ? (float)d[i] : r[i]);
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

// one CU: the stream is masked down to CU 0 of the first shader engine, so that two 1024-thread workgroups (w = 8)
// land on the same CU as one does
static hipStream_t one_cu_stream() {
  static hipStream_t s = nullptr;
  if (!s) {
    uint32_t mask[8] = {1u, 0, 0, 0, 0, 0, 0, 0};
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) s = nullptr;
  }
  return s;
}
template <int OP>
void run(unsigned long long *dbuf, float *sink, int grid) {
  const int iters = 2000;
  hipStream_t st = one_cu_stream();
  printf("%-30s", names[OP]);
  for (int w : {1, 2, 4, 8}) {
    unsigned long long h[64];
    const int wgs = w > 4 ? 2 : 1, threads = 256 * (w > 4 ? 4 : w);
    if (w > 4 && !st) continue;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k_rate<OP>), dim3(wgs), dim3(threads), 0, st, iters, dbuf, sink);
    hipStreamSynchronize(st);
    hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);
    const double cyc = (double)h[0], wall = (double)h[1];
    // instructions issued per SIMD = iters * 16 * w  (w = 8: both workgroups run side by side; each times itself)
    const bool same_cu = wgs == 1 || h[40] == h[41];
    printf("  w=%d: %5.2f (clk %.0f)%s", w, cyc / (iters * 16.0 * w), cyc / wall * 100.0, same_cu ? "" : " [two CUs!]");
  }
  printf("\n");
}

// whole-chip rate: grid x 1024 threads, wall time by events
template <int OP>
void run_chip(unsigned long long *dbuf, float *sink, double flops_per_lane_inst) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000, grid = 256 * 2;
  hipLaunchKernelGGL((k_rate<OP>), dim3(grid), dim3(1024), 0, 0, iters, dbuf, sink);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k_rate<OP>), dim3(grid), dim3(1024), 0, 0, iters, dbuf, sink);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)grid * 16 * iters * 16;  // wave-instructions
  unsigned long long h[32];
  hipMemcpy(h, dbuf, sizeof(h), hipMemcpyDeviceToHost);  // (all workgroups write the same slots: whoever came last)
  // one wave's own view: its s_memtime ticks per instruction and the tick rate against the 100 MHz wall clock
  printf("chip %-26s %8.3f ms  %7.2f G wave-inst/s  = %6.1f TFLOP/s  (%.2f cyc/inst/SIMD at 2.4 GHz); a wave saw %.2f ticks per own instruction, tick rate %.0f MHz\n",
         names[OP], ms, insts / ms / 1e6, insts * 64 * flops_per_lane_inst / ms / 1e9, 2.4e9 * 1024 / (insts / (ms * 1e-3)),
         (double)h[0] / (iters * 16.0), (double)h[0] / (double)h[1] * 100.0);
}

int main() {
  unsigned long long *d;
  float *sink;
  hipMalloc(&d, 4096);
  hipMalloc(&sink, 1 << 20);
  printf("one CU (CU-masked stream), w waves per SIMD: s_memtime ticks per instruction per SIMD (tick rate in MHz against the 100 MHz wall clock)\n");
  run<OP_ADD>(d, sink, 1);
  run<OP_ADD_E64>(d, sink, 1);
  run<OP_FMAC>(d, sink, 1);
  run<OP_MUL>(d, sink, 1);
  run<OP_FMA>(d, sink, 1);
  run<OP_PKMUL>(d, sink, 1);
  run<OP_PKADD>(d, sink, 1);
  run<OP_PKFMA>(d, sink, 1);
  run<OP_ADD64>(d, sink, 1);
  run<OP_MUL64>(d, sink, 1);
  run<OP_FMA64>(d, sink, 1);
  run<OP_RCP>(d, sink, 1);
  run<OP_SQRT64>(d, sink, 1);
  run<OP_RCP64>(d, sink, 1);
  run<OP_CND>(d, sink, 1);
  run<OP_MULLO>(d, sink, 1);
  run<OP_ADDDPP>(d, sink, 1);
  run<OP_MOVDPP>(d, sink, 1);
  run<OP_READLANE>(d, sink, 1);
  run<OP_CVT>(d, sink, 1);
  run<OP_MAX3>(d, sink, 1);
  run<OP_DEP_ADD>(d, sink, 1);
  run<OP_DEP_ADDDPP>(d, sink, 1);
  run<OP_DEP_MOVDPP_ADD>(d, sink, 1);
  run<OP_DEP_FMA64>(d, sink, 1);
  printf("whole chip, 512 workgroups x 16 waves, wall clock by events\n");
  run_chip<OP_DEP_ADD>(d, sink, 1);
  run_chip<OP_ADD>(d, sink, 1);
  run_chip<OP_ADD_E64>(d, sink, 1);
  run_chip<OP_FMAC>(d, sink, 2);
  run_chip<OP_FMA>(d, sink, 2);
  run_chip<OP_PKFMA>(d, sink, 4);
  run_chip<OP_PKMUL>(d, sink, 2);
  run_chip<OP_FMA64>(d, sink, 2);
  run_chip<OP_CND>(d, sink, 0);
  return 0;
}
