// Ordered fp32 running sums on the MATRIX pipe.  v_mfma_f32_16x16x4_f32 is, bit for bit, a k-ordered chain of
// f32 fused multiply-adds, D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C)))) -- with B = 1 that is
// ((C + a0) + a1) + a2) + a3, four strictly ordered fp32 additions per instruction and row, sixteen rows at once,
// and not one VALU issue slot.  This benchmark walks five chains of n elements (rows 0, 4, 8, 12, 1 of the tile)
// out of LDS, writes every fourth running total back in place, fills the three totals in between with
// plain adds (one lane per quad), and compares the result bit for bit with a one-lane-per-chain serial walk.
// Prints cycles per element for: the serial lane-per-chain walk, the exec-masked VALU round scan (scan_e's), and the
// MFMA walk with and without the fill-in.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/micro/mfma_scan tools/micro/mfma_scan.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
extern __shared__ __attribute__((aligned(16))) float sm[];
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct alignas(16) F4 { float x, y, z, w; };

#define NCH 5
// chain c lives in tile row ROW(c): rows 0, 4, 8, 12 are register 0 of the lanes 0, 16, 32, 48 (column 0); row 1 is
// register 1 of lane 0  (C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg)
__device__ __forceinline__ int row_chain(int row) { return row == 1 ? 4 : ((row & 3) == 0 ? row >> 2 : -1); }

// walk: S[c * stride + e], e < n (n a multiple of 4); afterwards element 4g+3 of every chain holds its running total
template <bool STORE>
__device__ __forceinline__ void mfma_walk(float *S, int stride, int n) {
  const int lane = threadIdx.x & 63, row = lane & 15, k = lane >> 4;
  const int c = row_chain(row);
  const float *src = S + (c < 0 ? 0 : c) * stride + k;   // A[row][k] of group g: element 4g + k
  // who stores what: lanes 0,16,32,48 store reg 0 (chains 0..3), lane 0 also reg 1 (chain 4)
  float *dst0 = S + (lane >> 4) * stride + 3;
  float *dst1 = S + 4 * stride + 3;
  const bool st0 = STORE && (lane & 15) == 0, st1 = STORE && lane == 0;
  // Eight groups per trip: their A operands are fetched a trip ahead, and two accumulators alternate so that the
  // totals of group g leave for LDS under the shadow of group g+1's instruction (the matrix pipe hands a dependent
  // accumulator over after 40 cycles; nothing else needs to sit on that chain).
  constexpr int U = 8;
  const int groups = n >> 2;
  f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB;
  float a[U], an[U];
#pragma unroll
  for (int u = 0; u < U; u++) a[u] = c < 0 ? 0.f : src[4 * u];
  for (int g0 = 0; g0 < groups; g0 += U) {
    if (g0 + U < groups) {
#pragma unroll
      for (int u = 0; u < U; u++) an[u] = c < 0 ? 0.f : src[4 * (g0 + U + u)];
    }
#pragma unroll
    for (int u = 0; u < U; u += 2) {
      accB = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], 1.0f, accA, 0, 0, 0);
      if (u || g0) {  // the previous group's totals (in accA)
        if (st0) dst0[4 * (g0 + u - 1)] = accA[0];
        if (st1) dst1[4 * (g0 + u - 1)] = accA[1];
      }
      accA = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u + 1], 1.0f, accB, 0, 0, 0);
      if (st0) dst0[4 * (g0 + u)] = accB[0];
      if (st1) dst1[4 * (g0 + u)] = accB[1];
    }
#pragma unroll
    for (int u = 0; u < U; u++) a[u] = an[u];
  }
  if ((lane & 15) == 0) dst0[4 * (groups - 1)] = accA[0];
  if (lane == 0) dst1[4 * (groups - 1)] = accA[1];
}
// The same walk one element per instruction: v_mfma_f32_4x4x1_16b_f32 (two passes) computes, per block of four lanes,
// D[i][j] = C[i][j] + A[i] * B[j]; with A = 1 in every lane and B = the lane's own next element, all four registers of
// a lane advance the lane's own chain by one element -- a lane per chain, sixty-four chains, EVERY running total (no
// fill-in), and still no vector issue slot.  Four accumulators rotate so that a quad's totals leave with four b32
// stores while the chain moves on (a b128 store would want the four totals in consecutive registers: they are the
// first registers of four different tuples).
template <bool STORE>
__device__ __forceinline__ void mfma_walk41(float *S, int stride, int n) {
  const int lane = threadIdx.x & 63;
  float *p = S + (lane < NCH ? lane : 0) * stride;
  const bool st = STORE;
  constexpr int U = 4;  // quads fetched ahead
  const int quads = n >> 2;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  F4 v[U], vn[U];
#pragma unroll
  for (int u = 0; u < U; u++) v[u] = *(const F4 *)(p + 4 * u);
  for (int q0 = 0; q0 < quads; q0 += U) {
    if (q0 + U < quads) {
#pragma unroll
      for (int u = 0; u < U; u++) vn[u] = *(const F4 *)(p + 4 * (q0 + U + u));
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const f32x4 a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, v[u].x, acc, 0, 0, 0);
      const f32x4 a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, v[u].y, a1, 0, 0, 0);
      const f32x4 a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, v[u].z, a2, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, v[u].w, a3, 0, 0, 0);
      if (st) {
        float *d = p + 4 * (q0 + u);
        __builtin_nontemporal_store(a1[0], d);  // (kept as four b32 stores: see above)
        __builtin_nontemporal_store(a2[0], d + 1);
        __builtin_nontemporal_store(a3[0], d + 2);
        __builtin_nontemporal_store(acc[0], d + 3);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = vn[u];
  }
  if (!STORE && lane < NCH) p[n - 1] = acc[0];
}
// fill-in: one lane per quad
__device__ __forceinline__ void fill_in(float *S, int stride, int n) {
  const int groups = n >> 2;
  for (int t = threadIdx.x; t < groups; t += blockDim.x)
    for (int c = 0; c < NCH; c++) {
      float *q = S + c * stride + 4 * t;
      const float T = t ? q[-1] : 0.f;
      F4 v = *(F4 *)q;
      v.x = T + v.x;
      v.y = v.x + v.y;
      v.z = v.y + v.z;
      *(F4 *)q = v;
    }
}
__device__ __forceinline__ void serial_walk(float *S, int stride, int n) {
  const int lane = threadIdx.x;
  if (lane < NCH) {
    float acc = 0.f;
    for (int i = 0; i < n; i++) {
      acc += S[lane * stride + i];
      S[lane * stride + i] = acc;
    }
  }
}

__global__ void k(int mode, int n, int stride, const float *in, float *out, unsigned long long *ticks, int reps) {
  for (int i = threadIdx.x; i < NCH * stride; i += blockDim.x) sm[i] = in[i];
  __syncthreads();
  unsigned long long t0 = 0, acc = 0;
  for (int r = 0; r < reps; r++) {
    if (r) {
      for (int i = threadIdx.x; i < NCH * stride; i += blockDim.x) sm[i] = in[i];
      __syncthreads();
    }
    t0 = __builtin_readcyclecounter();
    if (mode == 0) serial_walk(sm, stride, n);
    if ((mode == 1 || mode == 2) && threadIdx.x < 64) mfma_walk<true>(sm, stride, n);
    if (mode == 3 && threadIdx.x < 64) mfma_walk<false>(sm, stride, n);
    if (mode == 4 && threadIdx.x < NCH) mfma_walk41<true>(sm, stride, n);   // (the matrix instruction ignores EXEC;
    if (mode == 5 && threadIdx.x < NCH) mfma_walk41<false>(sm, stride, n);  // the loads and stores need only these lanes)
    __syncthreads();
    if (mode == 2) fill_in(sm, stride, n);
    __syncthreads();
    acc += __builtin_readcyclecounter() - t0;
  }
  if (threadIdx.x == 0) ticks[blockIdx.x] = acc;
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < NCH * stride; i += blockDim.x) out[i] = sm[i];
}

int main() {
  const int n = 1024, stride = 1024, reps = 20;
  std::vector<float> h(NCH * stride);
  unsigned s = 12345;
  for (auto &v : h) {
    s = s * 1664525u + 1013904223u;
    v = 1.0f + (float)(s >> 8) * (1.0f / 16777216.0f) * 40000.f;  // y*y-like magnitudes: roundings at every add
  }
  // a few hard cases: denormal-free zeros, tiny and huge terms
  h[7] = 0.f; h[stride + 100] = 1e-30f; h[2 * stride + 555] = 3e11f;
  float *d_in, *d_out;
  unsigned long long *d_t;
  hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, h.size() * 4); hipMalloc(&d_t, 4096 * 8);
  hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> ref(h.size()), got(h.size());
  for (int c = 0; c < NCH; c++) {
    float acc = 0.f;
    for (int i = 0; i < n; i++) { acc += h[c * stride + i]; ref[c * stride + i] = acc; }
  }
  const char *names[6] = {"serial lane-per-chain", "mfma walk (totals only)", "mfma walk + fill-in", "mfma chain, no stores",
                          "4x4x1 walk, every total", "4x4x1 chain, no stores"};
  for (int mode = 0; mode < 6; mode++)
    for (int grid : {1, 256 * 8}) {
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), NCH * stride * 4, 0, mode, n, stride, d_in, d_out, d_t, reps);
      hipDeviceSynchronize();
      std::vector<unsigned long long> t(grid);
      hipMemcpy(t.data(), d_t, grid * 8, hipMemcpyDeviceToHost);
      hipMemcpy(got.data(), d_out, got.size() * 4, hipMemcpyDeviceToHost);
      double avg = 0;
      for (auto v : t) avg += (double)v;
      avg /= grid * (double)reps;
      long bad = 0;
      for (int c = 0; c < NCH; c++)
        for (int i = 0; i < n; i++) {
          if (mode == 1 && (i & 3) != 3) continue;
          if ((mode == 3 || mode == 5) && i != n - 1) continue;
          bad += memcmp(&ref[c * stride + i], &got[c * stride + i], 4) != 0;
        }
      printf("%-26s grid %5d: %8.0f ticks per walk = %5.2f per element; mismatches vs host serial sum: %ld\n", names[mode], grid,
             avg, avg / n, bad);
    }
  return 0;
}
