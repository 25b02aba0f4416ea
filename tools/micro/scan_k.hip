// The shipped running_sum_rounds (vorbis_amd/csrc/k_noise.h) under the kernel's own conditions: T teams of
// four waves on one CU, wave 0 of each walks five chains while the other three wait at the barrier.
// Variants: raised wave priority for the walking wave; E = 8 / 16.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ivorbis_amd/csrc -o tools/micro/scan_k tools/micro/scan_k.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "k_noise.h"
using namespace vamd;
extern __shared__ __attribute__((aligned(16))) float sm[];

template <int E, int PRIO>
__global__ void k(int n, int reps, unsigned long long *out) {
  const int stride = VAMD_NZ_STRIDE(n);
  for (int i = threadIdx.x; i < 5 * stride; i += blockDim.x) sm[i] = 1.f + 1e-3f * (i & 1023);
  __syncthreads();
  long long t = 0;
  for (int r = 0; r < reps; r++) {
    team_lds_barrier();
    if (threadIdx.x < 64) {
      if (PRIO) __builtin_amdgcn_s_setprio(3);
      const long long t0 = clock64();
      running_sum_rounds<E>(sm, stride, 5, n);
      t += clock64() - t0;
      if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    team_lds_barrier();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned long long)t;
}

template <int E, int PRIO>
void run(int teams, int waves, unsigned long long *d) {
  const int n = 1024, reps = 8;
  static unsigned long long h[4096];
  hipFuncSetAttribute((const void *)k<E, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int rep = 0; rep < 2; rep++)
    hipLaunchKernelGGL((k<E, PRIO>), dim3(teams), dim3(64 * waves), (size_t)5 * VAMD_NZ_STRIDE(n) * 4, 0, n, reps, d);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double mx = 0, av = 0;
  for (int i = 0; i < teams; i++) mx = h[i] > mx ? h[i] : mx, av += h[i];
  printf("E=%2d prio=%d  %d team(s) x %d waves: mean %5.2f max %5.2f cycles/element\n", E, PRIO, teams, waves, av / teams / reps / n,
         mx / reps / n);
}

int main() {
  unsigned long long *d;
  hipMalloc(&d, 4096 * 8);
    for (int teams : {1, 256, 256 * 4, 256 * 7}) {  // (the dispatcher deals teams round the XCDs and CUs: 256 x k teams = k per CU)
    run<16, 0>(teams, 4, d);
    run<16, 1>(teams, 4, d);
    run<8, 0>(teams, 4, d);
    run<16, 0>(teams, 1, d);
  }
  return 0;
}
