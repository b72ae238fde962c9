// Microbenchmark: ds_read_b128 throughput with one wave per SIMD (4 waves per CU), 16 reads in flight per wave,
// (a) reads only, (b) reads + a 16-MFMA chain per 16 reads, half-tile software pipelining as in encoder_fused.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int STRIDE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, l = tid & 63, r16 = l & 15, g = l >> 4;
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 63);
  __syncthreads();
  bf16x8 a;
  for (int e = 0; e < 8; e++) a[e] = (__bf16)(0.01f * (tid + e));
  f32x4 acc = {0, 0, 0, 0};
  u32x4 x = {0, 0, 0, 0};
  const unsigned char* base = lds + r16 * STRIDE + 16 * g;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    u32x4 f[16];
    const unsigned char* b2 = base + (it & 7) * 2048;
#pragma unroll
    for (int s = 0; s < 16; s++) f[s] = *reinterpret_cast<const u32x4*>(b2 + (s >> 2) * 20 * STRIDE + (s & 3) * STRIDE);
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 16; s++) x ^= f[s];
    } else {
#pragma unroll
      for (int s = 0; s < 16; s++) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, f[s]), acc, 0, 0, 0);
    }
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3] + (float)(x[0] ^ x[1] ^ x[2] ^ x[3]);
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int STRIDE>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 400;
  k<MODE, STRIDE><<<256, 256>>>(out, cyc, iters);
  k<MODE, STRIDE><<<256, 256>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-40s %6.1f clk per ds_read_b128 (per wave)\n", name, (double)c / (iters * 16.0));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 80>("reads only, pixel stride 80 B");
  run<1, 80>("reads + MFMA chain, stride 80 B");
  run<0, 160>("reads only, pixel stride 160 B");
  run<1, 160>("reads + MFMA chain, stride 160 B");
  run<0, 64>("reads only, stride 64 B (conflicts)");
  return 0;
}
