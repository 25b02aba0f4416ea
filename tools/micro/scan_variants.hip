// Which access width should the running-sum scan use?  One wave, `lanes` chains at stride (n+4)
// floats, 1024 elements each, software-pipelined one group ahead like k_noise.h's running_sum_inplace.
//   hipcc --offload-arch=gfx950 -O3 -o scan_variants tools/micro/scan_variants.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ __attribute__((aligned(16))) float sm[];
struct alignas(16) F4 { float x, y, z, w; };
struct alignas(8) F2 { float x, y; };
#define S4(acc, v) acc += v.x; v.x = acc; acc += v.y; v.y = acc; acc += v.z; v.z = acc; acc += v.w; v.w = acc;
#define S2(acc, v) acc += v.x; v.x = acc; acc += v.y; v.y = acc;

template <int MODE>
__global__ void k(int lanes, int n, unsigned long long *out, float *sink) {
  const int lane = threadIdx.x;
  for (int i = lane; i < 36000; i += 64) sm[i] = 1e-3f * (i & 255);
  __syncthreads();
  float *p = sm + (size_t)lane * (n + 4);
  float acc = 0.f;
  const long long t0 = clock64();
  if (lane < lanes) {
    if (MODE == 0) {  // 16-byte accesses, two register sets of 4 quads (the shipped loop)
      F4 *q = (F4 *)p;
      const int nblk = n >> 4;
      F4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], b0 = q[4], b1 = q[5], b2 = q[6], b3 = q[7];
      for (int b = 0; b < nblk; b += 2) {
        S4(acc, a0) S4(acc, a1) S4(acc, a2) S4(acc, a3)
        q[4 * b] = a0; q[4 * b + 1] = a1; q[4 * b + 2] = a2; q[4 * b + 3] = a3;
        if (b + 2 < nblk) { a0 = q[4 * b + 8]; a1 = q[4 * b + 9]; a2 = q[4 * b + 10]; a3 = q[4 * b + 11]; }
        S4(acc, b0) S4(acc, b1) S4(acc, b2) S4(acc, b3)
        q[4 * b + 4] = b0; q[4 * b + 5] = b1; q[4 * b + 6] = b2; q[4 * b + 7] = b3;
        if (b + 3 < nblk) { b0 = q[4 * b + 12]; b1 = q[4 * b + 13]; b2 = q[4 * b + 14]; b3 = q[4 * b + 15]; }
      }
    } else if (MODE == 1) {  // 8-byte accesses, two register sets of 8 pairs
      F2 *q = (F2 *)p;
      const int nblk = n >> 4;  // 16 values = 8 pairs per block
      F2 a[8], c[8];
#pragma unroll
      for (int i = 0; i < 8; i++) { a[i] = q[i]; c[i] = q[8 + i]; }
      for (int b = 0; b < nblk; b += 2) {
#pragma unroll
        for (int i = 0; i < 8; i++) { S2(acc, a[i]) }
#pragma unroll
        for (int i = 0; i < 8; i++) q[8 * b + i] = a[i];
        if (b + 2 < nblk) {
#pragma unroll
          for (int i = 0; i < 8; i++) a[i] = q[8 * b + 16 + i];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) { S2(acc, c[i]) }
#pragma unroll
        for (int i = 0; i < 8; i++) q[8 * b + 8 + i] = c[i];
        if (b + 3 < nblk) {
#pragma unroll
          for (int i = 0; i < 8; i++) c[i] = q[8 * b + 24 + i];
        }
      }
    } else {  // 4-byte accesses, two register sets of 16
      float a[16], c[16];
      const int nblk = n >> 4;
#pragma unroll
      for (int i = 0; i < 16; i++) { a[i] = p[i]; c[i] = p[16 + i]; }
      for (int b = 0; b < nblk; b += 2) {
#pragma unroll
        for (int i = 0; i < 16; i++) { acc += a[i]; a[i] = acc; }
#pragma unroll
        for (int i = 0; i < 16; i++) p[16 * b + i] = a[i];
        if (b + 2 < nblk) {
#pragma unroll
          for (int i = 0; i < 16; i++) a[i] = p[16 * b + 32 + i];
        }
#pragma unroll
        for (int i = 0; i < 16; i++) { acc += c[i]; c[i] = acc; }
#pragma unroll
        for (int i = 0; i < 16; i++) p[16 * b + 16 + i] = c[i];
        if (b + 3 < nblk) {
#pragma unroll
          for (int i = 0; i < 16; i++) c[i] = p[16 * b + 48 + i];
        }
      }
    }
  }
  const long long t1 = clock64();
  if (lane == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0);
  if (acc == 12345.f) sink[lane] = acc;
}

int main() {
  unsigned long long *d, h;
  float *sink;
  hipMalloc(&d, 64);
  hipMalloc(&sink, 4096);
  const char *names[] = {"16-byte", "8-byte", "4-byte"};
  for (int mode = 0; mode < 3; mode++)
    for (int lanes : {35, 30, 5}) {
      void (*f)(int, int, unsigned long long *, float *) = mode == 0 ? k<0> : mode == 1 ? k<1> : k<2>;
      hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(f, dim3(1), dim3(64), 150 * 1024, 0, lanes, 1024, d, sink);
      hipDeviceSynchronize();
      hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      printf("%-8s accesses, %2d chains: %6.1f cycles per element (1024-element scan: %llu cycles)\n", names[mode], lanes,
             h / 1024.0, h);
    }
  return 0;
}
