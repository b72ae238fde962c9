// Microbenchmark: issue interval of dependent v_mfma_f32_16x16x32_bf16 chains (1, 2, 4 independent accumulators),
// with and without one ds_read_b128 per MFMA, one wave per SIMD (256-thread block per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int LDS>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 63);
  __syncthreads();
  bf16x8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (__bf16)(0.01f * (tid + e)); b[e] = (__bf16)(0.02f * e); }
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; j++) acc[j] = f32x4{0, 0, 0, 0};
  const unsigned char* base = lds + (tid & 63) * 80;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int s = 0; s < 16; s++) {
#pragma unroll
      for (int j = 0; j < NACC; j++) {
        bf16x8 bb = b;
        if (LDS) bb = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + ((s * NACC + j) & 63) * 320 + (it & 1) * 16));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, acc[j], 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = clock64();
  f32x4 t = acc[0];
  for (int j = 1; j < NACC; j++) t += acc[j];
  out[blockIdx.x * 256 + tid] = t[0] + t[1] + t[2] + t[3];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int LDS>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 200;
  k<NACC, LDS><<<256, 256>>>(out, cyc, iters);
  k<NACC, LDS><<<256, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s %6.1f clk per MFMA\n", name, (double)c / (iters * 16.0 * NACC));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1, 0>("1 chain, regs");
  run<2, 0>("2 chains, regs");
  run<4, 0>("4 chains, regs");
  run<1, 1>("1 chain, +1 ds_read_b128");
  run<2, 1>("2 chains, +1 ds_read_b128");
  run<4, 1>("4 chains, +1 ds_read_b128");
  return 0;
}
