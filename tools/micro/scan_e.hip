// Ordered fp32 running sums, L lanes per chain, E consecutive elements per lane per step, rounds under an
// exec mask: in round r only lane r of every chain runs its E dependent adds, then hands its total to lane
// r+1 (DPP row_shr:1).  The instruction stream per round is E adds + a few bookkeeping instructions, so for
// large E a wave walks at close to the dependent-add latency (7.2 cycles/element measured, tools/micro/scan_t)
// however many chains (64/L) ride in it.  Chains live in LDS, chain c at sm + c*stride, plain order.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/micro/scan_e tools/micro/scan_e.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string.h>
extern __shared__ __attribute__((aligned(16))) float sm[];
struct alignas(16) F4 { float x, y, z, w; };

__device__ __forceinline__ float shr1(float v) {  // lane i receives lane i-1's value (within a row of 16)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ float shlN(float v) {  // lane i receives lane i+N's value (within a row of 16)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true));
}

// L in {4, 8, 16}; E in {8, 16}.  chains of one wave: lanes [L*c, L*c + L)
template <int L, int E>
__device__ __forceinline__ void scan_chains(float *base, int stride, int nchains, int n) {
  const int lane = threadIdx.x & 63, g = lane / L, j = lane % L;
  const bool act = g < nchains;
  float *p = base + (act ? g : 0) * stride + j * E;
  const int steps = n / (L * E);
  float c0 = 0.f;  // lane j == 0: the chain's total through the previous step
  for (int s = 0; s < steps; s++) {
    float v[E];
#pragma unroll
    for (int k = 0; k < E / 4; k++) {
      const F4 t = ((const F4 *)(p + s * L * E))[k];
      v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
    }
    float cin = c0;
#pragma unroll
    for (int r = 0; r < L; r++) {
      if (j == r) {
        v[0] = cin + v[0];
#pragma unroll
        for (int k = 1; k < E; k++) v[k] = v[k - 1] + v[k];
      }
      cin = shr1(v[E - 1]);  // lane r+1 now holds lane r's total (other lanes: don't care)
    }
    c0 = shlN<L - 1>(v[E - 1]);  // lane j == 0 receives lane j == L-1's total
    if (act) {
#pragma unroll
      for (int k = 0; k < E / 4; k++) {
        F4 t;
        t.x = v[4 * k]; t.y = v[4 * k + 1]; t.z = v[4 * k + 2]; t.w = v[4 * k + 3];
        ((F4 *)(p + s * L * E))[k] = t;
      }
    }
  }
}

template <int L, int E>
__global__ void k(int nchains, int n, int stride, unsigned long long *out, float *dump) {
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float *mine = sm + wave * nchains * stride;
  for (int i = threadIdx.x; i < nw * nchains * stride; i += blockDim.x) sm[i] = 1.f + 1e-3f * ((i * 7) & 1023) * (1 + (i >> 10));
  __syncthreads();
  const long long t0 = clock64();
  scan_chains<L, E>(mine, stride, nchains, n);
  const long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) out[wave] = (unsigned long long)(t1 - t0);
  __syncthreads();
  if (dump)
    for (int i = threadIdx.x; i < nchains * stride; i += blockDim.x) dump[i] = sm[i];
}

template <int L, int E>
void run(int waves, int nchains, unsigned long long *d, float *dump) {
  const int n = 1024, stride = n + 16;
  unsigned long long h[16];
  hipFuncSetAttribute((const void *)k<L, E>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; rep++)
    hipLaunchKernelGGL((k<L, E>), dim3(1), dim3(64 * waves), (size_t)waves * nchains * stride * 4, 0, nchains, n, stride, d, dump);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long mx = 0;
  for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
  // check wave 0's chains against a sequential host sum
  std::vector<float> got((size_t)nchains * stride);
  hipMemcpy(got.data(), dump, got.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int c = 0; c < nchains; c++) {
    volatile float acc = 0.f;
    for (int i = 0; i < n; i++) {
      const int idx = c * stride + i;
      const float t = 1.f + 1e-3f * ((idx * 7) & 1023) * (1 + (idx >> 10));
      acc = acc + t;
      float a = acc;
      if (memcmp(&a, &got[idx], 4)) bad++;
    }
  }
  printf("L=%2d E=%2d  %d wave(s) x %d chains: %5.2f cycles/element (slowest wave %llu for 1024)  %s\n", L, E, waves, nchains,
         mx / 1024.0, mx, bad ? "MISMATCH" : "exact");
}

int main() {
  unsigned long long *d;
  float *dump;
  hipMalloc(&d, 1024);
  hipMalloc(&dump, 1 << 20);
  for (int waves : {1, 4, 7}) {
    run<8, 16>(waves, 5, d, dump);
    run<8, 8>(waves, 5, d, dump);
    run<16, 16>(waves, 4, d, dump);
    run<16, 8>(waves, 4, d, dump);
    run<4, 16>(waves, 5, d, dump);
    run<8, 4>(waves, 5, d, dump);
  }
  return 0;
}
