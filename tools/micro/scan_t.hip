// Lane-per-chain ordered running sums over a chain-interleaved LDS layout (round 2 design of k_noise):
// quad q of chain c lives at S[(q*NCH + c)*4 .. +4), so the 5G chains of a workgroup's G channel-blocks
// are walked by ONE wave, one lane per chain, with 16-byte accesses that are consecutive across the lanes
// (conflict-free), while the per-bin phases still find consecutive elements of a chain on consecutive
// banks when NCH is odd.  Measures shader cycles per element of the 1024-element walk for several
// prefetch depths, alone and beside "worker" waves that keep the LDS pipe and the VALUs busy.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/micro/scan_t tools/micro/scan_t.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ __attribute__((aligned(16))) float sm[];
struct alignas(16) F4 { float x, y, z, w; };

// MODE 0: b128 load, 4 adds, b128 store; P quads in flight
// MODE 1: same, but the adds only (no LDS traffic): the bare dependent-add latency
// MODE 2: b128 load, 4 adds, four b32 stores
// MODE 3: two interleaved chain sets per lane (chains c and c + NCH2): two dependent chains in flight
template <int MODE, int P>
__global__ void k(int nch, int nq, int workers, unsigned long long *out, float *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < nq * nch * 4; i += blockDim.x) sm[i] = 1e-3f * (i & 255) + 1.f;
  int *flag = (int *)(sm + nq * nch * 4);
  if (threadIdx.x == 0) *flag = 0;
  __syncthreads();
  if (wave == 0) {
    float acc = 0.f, acc2 = 0.f;
    const int c = lane < nch ? lane : nch - 1;
    F4 *q = (F4 *)sm + c;
    const long long t0 = clock64();
    if (MODE == 3) {
      const int half = nch / 2;
      const int c1 = lane < half ? lane : half - 1;
      F4 *qa = (F4 *)sm + c1, *qb = (F4 *)sm + c1 + half;
      F4 va[P], vb[P];
#pragma unroll
      for (int p = 0; p < P; p++) va[p] = qa[p * nch], vb[p] = qb[p * nch];
      for (int s = 0; s < nq; s += P) {
#pragma unroll
        for (int p = 0; p < P; p++) {
          F4 a = va[p], b = vb[p];
          a.x = acc + a.x; b.x = acc2 + b.x;
          a.y = a.x + a.y; b.y = b.x + b.y;
          a.z = a.y + a.z; b.z = b.y + b.z;
          a.w = a.z + a.w; b.w = b.z + b.w;
          acc = a.w; acc2 = b.w;
          qa[(s + p) * nch] = a;
          qb[(s + p) * nch] = b;
          if (s + p + P < nq) va[p] = qa[(s + p + P) * nch], vb[p] = qb[(s + p + P) * nch];
        }
      }
    } else {
      F4 v[P];
#pragma unroll
      for (int p = 0; p < P; p++) v[p] = q[p * nch];
      for (int s = 0; s < nq; s += P) {
#pragma unroll
        for (int p = 0; p < P; p++) {
          F4 a = v[p];
          a.x = acc + a.x;
          a.y = a.x + a.y;
          a.z = a.y + a.z;
          a.w = a.z + a.w;
          acc = a.w;
          if (MODE == 0) q[(s + p) * nch] = a;
          if (MODE == 2) {
            float *o = (float *)(q + (s + p) * nch);
            o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
          }
          if (MODE != 1) {
            if (s + p + P < nq) v[p] = q[(s + p + P) * nch];
          } else {
            v[p].x += 1e-7f;  // keep the loop from folding
          }
        }
      }
    }
    const long long t1 = clock64();
    if (lane == 0) {
      out[0] = (unsigned long long)(t1 - t0);
      __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (acc + acc2 == 12345.f) sink[threadIdx.x] = acc;
  } else if (wave <= workers) {
    // a worker: eval-like mix -- ten 4-byte LDS reads and ~40 VALU per "bin", until the scan wave is done
    float s = 0.f;
    int e = lane + 64 * wave;
    unsigned long long n = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        e = (e * 5 + 17) & 1023;
        const int hi = e, lo = (e * 3) & 1023;
        const float *ph = sm + ((hi >> 2) * nch) * 4 + (hi & 3), *pl = sm + ((lo >> 2) * nch) * 4 + (lo & 3);
        const float tN = ph[0] - pl[0], tX = ph[4] - pl[4], tXX = ph[8] - pl[8], tY = ph[12] - pl[12], tXY = ph[16] - pl[16];
        const float A = tY * tXX - tX * tXY, B = tN * tXY - tX * tY, D = tN * tXX - tX * tX;
        s += (A + (float)e * B) / (D + 3.f);
      }
      n++;
    }
    if (lane == 0) out[wave] = n * 4;
    if (s == 12345.f) sink[threadIdx.x] = s;
  }
}

template <int MODE, int P>
void run(const char *name, int nch, int workers, unsigned long long *d, float *sink) {
  unsigned long long h[16];
  const int nq = 256;
  hipFuncSetAttribute((const void *)k<MODE, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; rep++)
    hipLaunchKernelGGL((k<MODE, P>), dim3(1), dim3(64 * (1 + workers)), (size_t)nq * nch * 16 + 64, 0, nch, nq, workers, d, sink);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-34s P=%d chains=%2d workers=%2d: %5.2f cycles/element (%llu for 1024)", name, P, nch, workers, h[0] / 1024.0, h[0]);
  if (workers) printf("  worker bins/wave done meanwhile: %llu (%.1f cycles per wave-bin-eval)", h[1], (double)h[0] / (double)h[1]);
  printf("\n");
}

int main() {
  unsigned long long *d;
  float *sink;
  hipMalloc(&d, 1024);
  hipMalloc(&sink, 1 << 16);
  for (int workers : {0, 3, 7, 11, 15}) {
    run<0, 2>("b128 ld / 4 add / b128 st", 35, workers, d, sink);
    run<0, 4>("b128 ld / 4 add / b128 st", 35, workers, d, sink);
    run<0, 8>("b128 ld / 4 add / b128 st", 35, workers, d, sink);
    run<2, 4>("b128 ld / 4 add / 4 x b32 st", 35, workers, d, sink);
    run<3, 2>("two chain sets per lane", 34, workers, d, sink);
    run<3, 4>("two chain sets per lane", 34, workers, d, sink);
    if (!workers) run<1, 4>("adds only", 35, workers, d, sink);
  }
  return 0;
}
