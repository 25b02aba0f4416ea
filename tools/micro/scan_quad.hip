// How fast can a wave walk ordered fp32 running sums out of LDS when several lanes share a chain?
// Lane j of a chain's L lanes owns quad L*s+j of step s; the running total is handed from lane to
// lane with DPP row shifts so that the adds keep their order.  Variants differ in how the rounds are
// predicated and in the software pipelining.
//   hipcc --offload-arch=gfx950 -O3 -o scan_quad tools/micro/scan_quad.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
extern __shared__ __attribute__((aligned(16))) float sm[];
struct alignas(16) F4 { float x, y, z, w; };

__device__ __forceinline__ float shr1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float shlN(float v, int n) {  // n = 1, 3, 7
  if (n == 1) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xf, 0xf, true));
  if (n == 3) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x103, 0xf, 0xf, true));
  if (n == 15) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x10f, 0xf, 0xf, true));
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x107, 0xf, 0xf, true));
}

// MODE 0: exec-masked rounds, one register set (the first shipped form)
// MODE 1: exec-masked rounds, two register sets (no copies)
// MODE 2: select-based rounds (no branches), two register sets
// MODE 3: every lane runs every round into its own temporaries (no predication on the dependent chain:
//         lane r's round-r result is the valid one and is what the next lane receives); one select tree
//         per step picks each lane's own round afterwards.  Two register sets.
template <int MODE, int L>
__global__ void k(int n, unsigned long long *out, float *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 36000; i += blockDim.x) sm[i] = 1e-3f * (i & 255);
  __syncthreads();
  const int g = lane / L, j = lane % L;
  const int chain = wave * (64 / L) + g;
  F4 *q = (F4 *)(sm + (size_t)chain * (n + 4)) + j;
  const int steps = n / (4 * L);
  float carry = 0.f;
  const long long t0 = clock64();
  if (MODE == 0) {
    F4 a = q[0];
    for (int s = 0; s < steps; s++) {
      F4 nxt = a;
      if (s + 1 < steps) nxt = q[L * (s + 1)];
      float cin = carry;
#pragma unroll
      for (int r = 0; r < L; r++) {
        if (j == r) { a.x = cin + a.x; a.y = a.x + a.y; a.z = a.y + a.z; a.w = a.z + a.w; }
        cin = shr1(a.w);
      }
      carry = shlN(a.w, L - 1);
      q[L * s] = a;
      a = nxt;
    }
  } else if (MODE == 3) {
    F4 a = q[0], b = q[L];
    for (int s = 0; s < steps; s += 2) {
#pragma unroll
      for (int half = 0; half < 2; half++) {
        F4 &v = half ? b : a;
        F4 t[L];
        float cin = carry;
#pragma unroll
        for (int r = 0; r < L; r++) {
          t[r].x = cin + v.x; t[r].y = t[r].x + v.y; t[r].z = t[r].y + v.z; t[r].w = t[r].z + v.w;
          cin = shr1(t[r].w);
        }
        carry = shlN(t[L - 1].w, L - 1);
        F4 o = t[0];
#pragma unroll
        for (int r = 1; r < L; r++) {
          const bool m = j == r;
          o.x = m ? t[r].x : o.x; o.y = m ? t[r].y : o.y; o.z = m ? t[r].z : o.z; o.w = m ? t[r].w : o.w;
        }
        q[L * (s + half)] = o;
        if (s + half + 2 < steps) v = q[L * (s + half + 2)];
      }
    }
  } else {
    F4 a = q[0], b = q[L];
    for (int s = 0; s < steps; s += 2) {
      float cin = carry;
#pragma unroll
      for (int r = 0; r < L; r++) {
        if (MODE == 1) {
          if (j == r) { a.x = cin + a.x; a.y = a.x + a.y; a.z = a.y + a.z; a.w = a.z + a.w; }
        } else {
          const float x = cin + a.x, y = x + a.y, z = y + a.z, w = z + a.w;
          const bool m = j == r;
          a.x = m ? x : a.x; a.y = m ? y : a.y; a.z = m ? z : a.z; a.w = m ? w : a.w;
        }
        cin = shr1(a.w);
      }
      carry = shlN(a.w, L - 1);
      q[L * s] = a;
      if (s + 2 < steps) a = q[L * (s + 2)];
      cin = carry;
#pragma unroll
      for (int r = 0; r < L; r++) {
        if (MODE == 1) {
          if (j == r) { b.x = cin + b.x; b.y = b.x + b.y; b.z = b.y + b.z; b.w = b.z + b.w; }
        } else {
          const float x = cin + b.x, y = x + b.y, z = y + b.z, w = z + b.w;
          const bool m = j == r;
          b.x = m ? x : b.x; b.y = m ? y : b.y; b.z = m ? z : b.z; b.w = m ? w : b.w;
        }
        cin = shr1(b.w);
      }
      carry = shlN(b.w, L - 1);
      q[L * (s + 1)] = b;
      if (s + 3 < steps) b = q[L * (s + 3)];
    }
  }
  const long long t1 = clock64();
  if (lane == 0) out[wave] = (unsigned long long)(t1 - t0);
  if (carry == 12345.f) sink[threadIdx.x] = carry;
}

template <int MODE, int L>
void run(const char *name, int waves, unsigned long long *d, float *sink) {
  unsigned long long h[4];
  hipFuncSetAttribute((const void *)k<MODE, L>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((k<MODE, L>), dim3(1), dim3(64 * waves), 150 * 1024, 0, 1024, d, sink);
  hipDeviceSynchronize();
  hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("%-28s L=%d  %d wave(s) x %2d chains: %5.2f cycles per element of a chain (1024-element scan: %llu cycles)\n", name, L, waves,
         64 / L, h[0] / 1024.0, h[0]);
}

int main() {
  unsigned long long *d;
  float *sink;
  hipMalloc(&d, 64);
  hipMalloc(&sink, 4096);
  for (int waves : {1, 2, 4}) {
    run<3, 4>("all rounds + select, 2 sets", waves, d, sink);
    run<3, 8>("all rounds + select, 2 sets", waves, d, sink);
    if (waves < 4) run<3, 16>("all rounds + select, 2 sets", waves, d, sink);
    run<0, 4>("exec rounds, 1 set", waves, d, sink);
    run<1, 4>("exec rounds, 2 sets", waves, d, sink);
    run<2, 4>("select rounds, 2 sets", waves, d, sink);
    run<1, 2>("exec rounds, 2 sets", waves, d, sink);
    run<2, 2>("select rounds, 2 sets", waves, d, sink);
    run<1, 8>("exec rounds, 2 sets", waves, d, sink);
    run<2, 8>("select rounds, 2 sets", waves, d, sink);
  }
  return 0;
}
