// Two independent MFMA accumulator chains interleaved in inline asm (X0 Y0 [s_nop N] X1 Y1 ...): which N makes the sums
// right (the hardware does not interlock a non-adjacent dependent MFMA), and what rate does one wave per SIMD reach?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
template <int N> __device__ __forceinline__ void nop() { if constexpr (N > 0) asm volatile("s_nop %0" ::"n"(N - 1)); }
// MODE 0: one chain of 18;  MODE 1: two chains of 9 interleaved with s_nop N after every pair;  MODE 2: three chains of 6
template <int MODE, int N, int NT>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* cyc, int iters) {
  const int tid = threadIdx.x;
  u32x4 w[9], f[9];
#pragma unroll
  for (int s = 0; s < 9; s++) {
    w[s] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};                     // bf16 ones
    f[s] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + ((unsigned)s << 16)};  // small constants
  }
  f32x4 x = {0.f, 0.f, 0.f, 0.f}, y = x, z = x;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 9; s++) MF(x, w[s], f[s]);
#pragma unroll
      for (int s = 0; s < 9; s++) MF(y, w[s], f[s]);
    } else if (MODE == 1) {
#pragma unroll
      for (int s = 0; s < 9; s++) { MF(x, w[s], f[s]); MF(y, w[s], f[s]); nop<N>(); }
    } else {
#pragma unroll
      for (int s = 0; s < 6; s++) { MF(x, w[s], f[s]); MF(y, w[s], f[s]); MF(z, w[s], f[s]); nop<N>(); }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(x), "+v"(y), "+v"(z));
  const unsigned long long t1 = clock64();
  out[(blockIdx.x * NT + tid) * 2] = x[0] + x[1] + x[2] + x[3];
  out[(blockIdx.x * NT + tid) * 2 + 1] = y[0] + y[1] + y[2] + y[3] + z[0];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int N, int NT>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * NT * 8); hipMalloc(&cyc, 8);
  const int iters = 200;
  k<MODE, N, NT><<<256, NT>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c; float h[2];
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
  printf("%-18s nop %d waves/CU %d: %6.1f ticks per 18 MFMAs per wave; sums %.6g %.6g\n", name, N, NT / 64, (double)c / iters, h[0], h[1]);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 0, 256>("one chain");
  run<1, 0, 256>("two chains"); run<1, 1, 256>("two chains"); run<1, 2, 256>("two chains"); run<1, 3, 256>("two chains");
  run<1, 4, 256>("two chains"); run<1, 6, 256>("two chains"); run<1, 8, 256>("two chains");
  run<2, 0, 256>("three chains"); run<2, 2, 256>("three chains"); run<2, 4, 256>("three chains");
  run<0, 0, 512>("one chain"); run<1, 4, 512>("two chains");
  return 0;
}
