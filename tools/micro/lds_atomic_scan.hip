// Can the LDS do the ordered fp32 running sums itself?  ds_add_rtn_f32 with K consecutive lanes aimed at
// ONE accumulator word: if the LDS serialises same-address lanes in ascending lane order and adds in IEEE
// fp32 (round to nearest even, no flush), each lane gets back the exclusive running sum of the chain --
// 64/K chains x K ordered elements per instruction, no dependent VALU chain, no scan wave.
// This program (a) checks the returned values bit-for-bit against a sequential host sum over many
// instructions and magnitudes, (b) times the instruction for K = 1 .. 64 with 1, 4, 8, 16 waves on a CU.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/micro/lds_atomic_scan tools/micro/lds_atomic_scan.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

__device__ __forceinline__ float lds_add_rtn(float *p, float v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// correctness: one wave, `ninst` instructions; lane l adds in[i*64 + l] to acc[l / K]; out = returned value
__global__ void k_check(int K, int ninst, const float *__restrict__ in, float *__restrict__ out, float *__restrict__ fin) {
  __shared__ float acc[64];
  const int lane = threadIdx.x;
  acc[lane] = 0.f;
  __syncthreads();
  float *p = acc + lane / K;
  for (int i = 0; i < ninst; i++) out[i * 64 + lane] = lds_add_rtn(p, in[i * 64 + lane]);
  __syncthreads();
  fin[lane] = acc[lane];
}

// timing: every wave owns 64/K accumulators (stride `astride` words apart) and issues 256 instructions
template <int UNUSED>
__global__ void k_time(int K, int astride, int iters, unsigned long long *out, float *sink) {
  extern __shared__ float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) sm[i] = 0.f;
  __syncthreads();
  float *p = sm + (wave * (64 / K) + lane / K) * astride;
  float v = 1.0f + 1e-3f * lane, s = 0.f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) s += lds_add_rtn(p, v);
  }
  const long long t1 = clock64();
  if (lane == 0) out[wave] = (unsigned long long)(t1 - t0);
  if (s == 12345.f) sink[threadIdx.x] = s;
}

int main() {
  const int ninst = 512;
  std::vector<float> h_in(ninst * 64), h_out(ninst * 64), h_fin(64);
  float *d_in, *d_out, *d_fin, *sink;
  unsigned long long *d_t;
  hipMalloc(&d_in, h_in.size() * 4);
  hipMalloc(&d_out, h_out.size() * 4);
  hipMalloc(&d_fin, 256);
  hipMalloc(&sink, 1 << 16);
  hipMalloc(&d_t, 1024);
  int total_bad = 0;
  for (int trial = 0; trial < 6; trial++) {
    srand(1234 + trial);
    for (size_t i = 0; i < h_in.size(); i++) {
      // terms as in bark_noise_hybridmp: w = y*y (y >= 1), w*x, w*x*x ... spanning many magnitudes
      const float y = 1.f + (rand() % 100000) * 1e-3f * (trial + 1);
      const float x = (float)(rand() % 1024);
      float t = y * y;
      if (trial & 1) t = t * x;
      if (trial & 2) t = t * x;
      if (trial == 5) t = (rand() & 1) ? -t : t;  // mixed signs (X/XY never are, but check the adder anyway)
      h_in[i] = t;
    }
    hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice);
    for (int K : {1, 2, 4, 6, 8, 12, 16, 32, 64}) {
      if (64 % K) {
        // K that does not divide 64: the last group is short; handled by lane / K anyway
      }
      hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, K, ninst, d_in, d_out, d_fin);
      hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost);
      hipMemcpy(h_fin.data(), d_fin, 256, hipMemcpyDeviceToHost);
      int bad = 0, first_bad = -1;
      float acc[64];
      for (int a = 0; a < 64; a++) acc[a] = 0.f;
      for (int i = 0; i < ninst; i++)
        for (int l = 0; l < 64; l++) {
          volatile float pre = acc[l / K];
          if (memcmp((const void *)&pre, &h_out[i * 64 + l], 4)) {
            if (first_bad < 0) first_bad = i * 64 + l;
            bad++;
          }
          volatile float nx = pre + h_in[i * 64 + l];
          acc[l / K] = nx;
        }
      for (int a = 0; a < (64 + K - 1) / K; a++)
        if (memcmp(&acc[a], &h_fin[a], 4)) bad++;
      printf("trial %d K=%2d: %s (%d mismatches%s", trial, K, bad ? "MISMATCH" : "ordered-exact", bad, bad ? ", first at " : ")\n");
      if (bad) printf("%d: got %.9g)\n", first_bad, h_out[first_bad]);
      total_bad += bad;
    }
  }
  printf("TOTAL mismatches: %d\n", total_bad);
  hipFuncSetAttribute((const void *)k_time<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int waves : {1, 4, 8, 16}) {
    for (int K : {1, 2, 4, 6, 8, 12, 16, 32, 64}) {
      for (int astride : {1, 33}) {
        unsigned long long h[16];
        const int iters = 64;
        for (int rep = 0; rep < 2; rep++)
          hipLaunchKernelGGL(k_time<0>, dim3(1), dim3(64 * waves), 65536, 0, K, astride, iters, d_t, sink);
        hipDeviceSynchronize();
        hipMemcpy(h, d_t, sizeof(h), hipMemcpyDeviceToHost);
        const double per = (double)h[0] / (iters * 16.0);
        printf("waves=%2d K=%2d (%2d accumulators/wave, %2d words apart): %7.1f cycles per instruction per wave, %6.2f chain-elements per cycle per CU\n",
               waves, K, 64 / K, astride, per, 64.0 * waves / per);
      }
    }
  }
  return 0;
}
