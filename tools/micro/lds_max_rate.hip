// What a no-return LDS max costs by type: ds_max_f32 (what k_tone_seed scatters with) against ds_max_u32 / ds_max_i32
// and a plain ds_write_b32, every lane at its own word (stride `st` words) or K lanes per word.
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/lds_max_rate tools/micro/lds_max_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

enum { OP_F32, OP_U32, OP_I32, OP_ST };
template <int OP>
__global__ void k_time(int st, int K, int iters, unsigned long long *out) {
  extern __shared__ unsigned sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) sm[i] = 0;
  __syncthreads();
  unsigned *p = sm + ((wave * 64 + lane / K * K) * st & 16383);
  unsigned v = 100 + lane;
  const long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      unsigned *q = p + ((u * 67) & 1023);
      if (OP == OP_F32) asm volatile("ds_max_f32 %0, %1" ::"v"((unsigned)(size_t)q), "v"(v) : "memory");
      if (OP == OP_U32) asm volatile("ds_max_u32 %0, %1" ::"v"((unsigned)(size_t)q), "v"(v) : "memory");
      if (OP == OP_I32) asm volatile("ds_max_i32 %0, %1" ::"v"((unsigned)(size_t)q), "v"(v) : "memory");
      if (OP == OP_ST) asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(size_t)q), "v"(v) : "memory");
      v += 3;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  if (lane == 0) out[wave] = (unsigned long long)(t1 - t0);
}

template <int OP>
void run(const char *name, int waves, int st, int K) {
  unsigned long long *d_t, h[32];
  hipMalloc(&d_t, 256);
  const int iters = 256;
  hipLaunchKernelGGL(k_time<OP>, dim3(1), dim3(64 * waves), 65536, 0, st, K, iters, d_t);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_time<OP>, dim3(1), dim3(64 * waves), 65536, 0, st, K, iters, d_t);
  hipDeviceSynchronize();
  hipMemcpy(h, d_t, 8 * waves, hipMemcpyDeviceToHost);
  unsigned long long mx = 0;
  for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
  // clock64 ticks at 100 MHz on this part: report ticks per wave-instruction and per CU
  printf("%-14s waves %2d stride %2d K %2d: %7.2f ticks/instr/wave  %7.3f ticks per instr over the CU\n", name, waves, st, K,
         (double)mx / (iters * 16), (double)mx / (iters * 16 * waves));
  hipFree(d_t);
}

int main() {
  for (int waves : {1, 8, 16}) {
    for (int K : {1, 8}) {
      run<OP_ST>("ds_write_b32", waves, 1, K);
      run<OP_F32>("ds_max_f32", waves, 1, K);
      run<OP_U32>("ds_max_u32", waves, 1, K);
      run<OP_I32>("ds_max_i32", waves, 1, K);
    }
  }
  return 0;
}
