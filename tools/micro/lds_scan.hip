// Microbenchmark behind DESIGN.md's statements about the running-sum scan: what do the LDS
// instructions of one wave cost, alone and interleaved with a dependent fp32 add chain?
//   hipcc --offload-arch=gfx950 -O3 -o lds_scan tools/micro/lds_scan.hip && ./lds_scan
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
extern __shared__ __attribute__((aligned(16))) float sm[];
struct alignas(16) F4 { float x, y, z, w; };
struct alignas(8) F2 { float x, y; };

template <int MODE>
__global__ void k(int lanes, int stride_f, int iters, unsigned long long *out, float *sink) {
  const int lane = threadIdx.x;
  float *p = sm + (size_t)lane * stride_f;
  for (int i = lane; i < 40000; i += 64) sm[i] = 1e-3f * i;
  __syncthreads();
  float acc = 0.f;
  F4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const bool act = lane < lanes;
  const long long t0 = clock64();
  if (act) {
    for (int it = 0; it < iters; it++) {
      F4 *q = (F4 *)(p + (it & 31) * 16);
      if (MODE == 0 || MODE == 2) {  // 4 loads b128
        a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3];
      }
      if (MODE == 1 || MODE == 2 || MODE == 4) {  // 16 dependent adds
        acc += a0.x; a0.x = acc; acc += a0.y; a0.y = acc; acc += a0.z; a0.z = acc; acc += a0.w; a0.w = acc;
        acc += a1.x; a1.x = acc; acc += a1.y; a1.y = acc; acc += a1.z; a1.z = acc; acc += a1.w; a1.w = acc;
        acc += a2.x; a2.x = acc; acc += a2.y; a2.y = acc; acc += a2.z; a2.z = acc; acc += a2.w; a2.w = acc;
        acc += a3.x; a3.x = acc; acc += a3.y; a3.y = acc; acc += a3.z; a3.z = acc; acc += a3.w; a3.w = acc;
      }
      if (MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4) {  // 4 stores b128
        q[0] = a0; q[1] = a1; q[2] = a2; q[3] = a3;
      }
      if (MODE == 5) {  // 8 loads + 8 stores b64
        F2 *h = (F2 *)q;
        F2 t0 = h[0], t1 = h[1], t2 = h[2], t3 = h[3], t4 = h[4], t5 = h[5], t6 = h[6], t7 = h[7];
        t0.x += 1.f; t7.y += 1.f;
        h[0] = t0; h[1] = t1; h[2] = t2; h[3] = t3; h[4] = t4; h[5] = t5; h[6] = t6; h[7] = t7;
      }
      if (MODE == 6) {  // 16 loads + 16 stores b32
        float *h = (float *)q;
        float t[16];
#pragma unroll
        for (int i = 0; i < 16; i++) t[i] = h[i];
        t[0] += 1.f; t[15] += 1.f;
#pragma unroll
        for (int i = 0; i < 16; i++) h[i] = t[i];
      }
      if (MODE == 7) {  // the scan on 8-byte accesses
        F2 *h = (F2 *)q;
        F2 t0 = h[0], t1 = h[1], t2 = h[2], t3 = h[3], t4 = h[4], t5 = h[5], t6 = h[6], t7 = h[7];
        acc += t0.x; t0.x = acc; acc += t0.y; t0.y = acc; acc += t1.x; t1.x = acc; acc += t1.y; t1.y = acc;
        acc += t2.x; t2.x = acc; acc += t2.y; t2.y = acc; acc += t3.x; t3.x = acc; acc += t3.y; t3.y = acc;
        acc += t4.x; t4.x = acc; acc += t4.y; t4.y = acc; acc += t5.x; t5.x = acc; acc += t5.y; t5.y = acc;
        acc += t6.x; t6.x = acc; acc += t6.y; t6.y = acc; acc += t7.x; t7.x = acc; acc += t7.y; t7.y = acc;
        h[0] = t0; h[1] = t1; h[2] = t2; h[3] = t3; h[4] = t4; h[5] = t5; h[6] = t6; h[7] = t7;
      }
      if (MODE == 8) {  // 4 loads b128 only
        a0 = q[0]; a1 = q[1]; a2 = q[2]; a3 = q[3];
        acc += a0.x + a1.y + a2.z + a3.w;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  if (lane == 0) out[blockIdx.x] = (unsigned long long)(t1 - t0);
  if (acc == 12345.f) sink[lane] = acc + a0.x + a1.y + a2.z + a3.w;
}

int main() {
  unsigned long long *d;
  float *sink;
  hipMalloc(&d, 8 * 1024);
  hipMalloc(&sink, 4096);
  const int iters = 4096;
  const char *names[] = {"4 ld128 + 4 st128", "16 dependent adds", "4 ld128 + 16 adds + 4 st128 (the scan)",
                         "4 st128 only", "16 adds + 4 st128", "8 ld64 + 8 st64", "16 ld32 + 16 st32", "8 ld64 + 16 adds + 8 st64", "4 ld128 only"};
  hipFuncSetAttribute((const void *)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int mode = 0; mode < 9; mode++)
    for (int cfg = 0; cfg < 6; cfg++) {
      const int lanes[] = {35, 35, 35, 64, 8, 1}, stride[] = {1028, 1030, 1056, 16, 1028, 1028};
      for (int wgs : {1, 1024}) {
        void (*f)(int, int, int, unsigned long long *, float *) =
            mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : mode == 3 ? k<3> : mode == 4 ? k<4> : mode == 5 ? k<5> : mode == 6 ? k<6> : mode == 7 ? k<7> : k<8>;
        hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(f, dim3(wgs), dim3(64), 160 * 1024 - 64, 0, lanes[cfg], stride[cfg], iters, d, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(wgs);
        hipMemcpy(h.data(), d, 8 * wgs, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += v;
        if (wgs == 1 || cfg == 0)
          printf("%-40s lanes %2d stride %4d wgs %4d : %.1f cycles / iteration (16 values)\n", names[mode], lanes[cfg],
                 stride[cfg], wgs, s / wgs / iters);
      }
    }
  return 0;
}
