// Where do the waves of a four-wave workgroup land?  k_noise's walker is wave 0 of its team: if the dispatcher puts
// wave 0 of every workgroup on the same SIMD, that SIMD carries all the walks of its CU.
// Launch shape as k_noise's: 256-thread workgroups, 20 KB of LDS each, six per CU, persistent for a while.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wave_simd.hip -o tools/micro/wave_simd && tools/micro/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned *out, int spin) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;  // stay resident until the whole grid has been placed
  lds[threadIdx.x] = a;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = (hw & 0xffffu) | ((xcc & 0xf) << 16) | (lds[(threadIdx.x + 64) & 255] == 1.f ? 1u << 31 : 0);
}
int main(int argc, char **argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 6;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int grid = per_cu * p.multiProcessorCount;
  unsigned *d;
  hipMalloc(&d, grid * 16);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 20480);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 20480, 0, d, 200000);
  hipDeviceSynchronize();
  std::vector<unsigned> h(grid * 4);
  hipMemcpy(h.data(), d, grid * 16, hipMemcpyDeviceToHost);
  // histogram: wave index -> SIMD id (HW_ID bits 5:4)
  long hist[4][4] = {};
  for (int b = 0; b < grid; b++)
    for (int w = 0; w < 4; w++) hist[w][(h[b * 4 + w] >> 4) & 3]++;
  printf("workgroups %d (%d per CU), wave index x SIMD id:\n", grid, per_cu);
  for (int w = 0; w < 4; w++) printf("  wave %d: simd0 %ld simd1 %ld simd2 %ld simd3 %ld\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  // per CU: how many wave-0s per SIMD
  long worst[8] = {};
  std::vector<int> cnt(16 * 8 * 16 * 2 * 4, 0);
  for (int b = 0; b < grid; b++) {
    const unsigned v = h[b * 4];
    const int simd = (v >> 4) & 3, cu = (v >> 8) & 15, sh = (v >> 12) & 1, se = (v >> 13) & 7, xcc = (v >> 16) & 15;
    cnt[(((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd]++;
  }
  for (size_t c = 0; c < cnt.size(); c += 4) {
    const int tot = cnt[c] + cnt[c + 1] + cnt[c + 2] + cnt[c + 3];
    if (!tot) continue;
    int mx = 0;
    for (int s = 0; s < 4; s++) mx = cnt[c + s] > mx ? cnt[c + s] : mx;
    worst[mx < 8 ? mx : 7]++;
  }
  printf("CUs by the largest number of wave-0s on one of their SIMDs:");
  for (int i = 1; i < 8; i++) printf("  %d: %ld", i, worst[i]);
  printf("\n");
  return 0;
}
