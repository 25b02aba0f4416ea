// Device facts + a residency census: how many workgroups of a given shape actually run at once.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/devinfo tools/micro/devinfo.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
__global__ void census(int spin, unsigned long long *t, unsigned *xcc) {
  const unsigned long long t0 = wall_clock64();
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  float a = threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 1.f;
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) {
    t[2 * blockIdx.x] = t0;
    t[2 * blockIdx.x + 1] = t1;
    xcc[2 * blockIdx.x] = x;
    xcc[2 * blockIdx.x + 1] = id;
  }
  if (a == 1.2345f) t[0] = 0;
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("name %s  gcn %s\nCUs %d  clockRate %d kHz  maxThreadsPerMP %d  regsPerBlock %d  sharedMemPerBlock %zu  maxSharedPerMP %zu  l2 %d\n", p.name,
         p.gcnArchName, p.multiProcessorCount, p.clockRate, p.maxThreadsPerMultiProcessor, p.regsPerBlock, p.sharedMemPerBlock,
         p.maxSharedMemoryPerMultiProcessor, p.l2CacheSize);
  for (int threads : {64, 256, 1024}) {
    for (int grid : {256, 512, 1024, 2048}) {
      unsigned long long *t;
      unsigned *x;
      hipMalloc(&t, grid * 16);
      hipMalloc(&x, grid * 8);
      hipLaunchKernelGGL(census, dim3(grid), dim3(threads), 0, 0, 200000, t, x);
      hipLaunchKernelGGL(census, dim3(grid), dim3(threads), 0, 0, 200000, t, x);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(2 * grid);
      std::vector<unsigned> hx(2 * grid);
      hipMemcpy(h.data(), t, grid * 16, hipMemcpyDeviceToHost);
      hipMemcpy(hx.data(), x, grid * 8, hipMemcpyDeviceToHost);
      unsigned long long first = ~0ull, lastend = 0;
      for (int i = 0; i < grid; i++) first = std::min(first, h[2 * i]), lastend = std::max(lastend, h[2 * i + 1]);
      // concurrency: how many blocks started within the first 10% of the first block's duration
      const unsigned long long dur = h[1] - h[0];
      int early = 0;
      for (int i = 0; i < grid; i++) early += (h[2 * i] - first) < dur / 2;
      int xc[8] = {0};
      for (int i = 0; i < grid; i++) xc[hx[2 * i] & 7]++;
      printf("threads %4d grid %4d: block duration %.1f us, all done after %.1f us, %d blocks started in the first half-duration; per XCC:", threads, grid,
             dur / 100.0, (lastend - first) / 100.0, early);
      for (int i = 0; i < 8; i++) printf(" %d", xc[i]);
      printf("\n");
      hipFree(t);
      hipFree(x);
    }
  }
  return 0;
}
