// Checks the DPP reductions/scans of vamd_wave.h against plain shuffles.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define VAMD_GPU_BUILD 1
#include "../../vorbis_amd/csrc/vamd_wave.h"
using namespace vamd;
__global__ void k(const int *in, int *out) {
  const int v = in[threadIdx.x];
  out[threadIdx.x] = wave_sum(v);
  out[64 + threadIdx.x] = wave_scan_max(v);
  out[128 + threadIdx.x] = wave_shift_up1(v, -7);
  out[192 + threadIdx.x] = (int)wave_max((float)v);
  out[256 + threadIdx.x] = (int)(wave_or64(1ull << (v & 63)) >> 32);
  out[320 + threadIdx.x] = wave_last(v);
}
int main() {
  int h[64], o[384], *d, *e;
  for (int i = 0; i < 64; i++) h[i] = (i * 37 + 11) % 101 - 20;
  hipMalloc(&d, 256); hipMalloc(&e, 384 * 4);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
  hipMemcpy(o, e, 384 * 4, hipMemcpyDeviceToHost);
  int sum = 0, mx = -1000, bad = 0; unsigned long long orr = 0;
  for (int i = 0; i < 64; i++) { sum += h[i]; orr |= 1ull << (h[i] & 63); }
  for (int i = 0; i < 64; i++) {
    if (h[i] > mx) mx = h[i];
    if (o[i] != sum) bad++, printf("sum lane %d: %d != %d\n", i, o[i], sum);
    if (o[64 + i] != mx) bad++, printf("scan lane %d: %d != %d\n", i, o[64 + i], mx);
    if (o[128 + i] != (i ? h[i - 1] : -7)) bad++, printf("shift lane %d: %d\n", i, o[128 + i]);
    if (o[320 + i] != h[63]) bad++;
  }
  int gmx = -1000; for (int i = 0; i < 64; i++) if (h[i] > gmx) gmx = h[i];
  for (int i = 0; i < 64; i++) { if (o[192 + i] != gmx) bad++, printf("max lane %d: %d != %d\n", i, o[192 + i], gmx);
    if (o[256 + i] != (int)(orr >> 32)) bad++, printf("or lane %d\n", i); }
  printf("bad %d\n", bad);
  return bad != 0;
}
