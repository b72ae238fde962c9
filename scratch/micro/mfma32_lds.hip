// v_mfma_f32_32x32x16_bf16 chains (one accumulator, blocks of 9 as the 32x32x16 conv3) with R ds_read_b128 per 9 MFMAs in
// flight: does the LDS return traffic slow the MFMA stream as it does for 16x16x32 (scratch/micro/mfma_agpr)?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define B9(acc, WC, w, f) asm volatile( \
  "v_mfma_f32_32x32x16_bf16 %0, %10, %1, %0\n v_mfma_f32_32x32x16_bf16 %0, %11, %2, %0\n v_mfma_f32_32x32x16_bf16 %0, %12, %3, %0\n" \
  "v_mfma_f32_32x32x16_bf16 %0, %13, %4, %0\n v_mfma_f32_32x32x16_bf16 %0, %14, %5, %0\n v_mfma_f32_32x32x16_bf16 %0, %15, %6, %0\n" \
  "v_mfma_f32_32x32x16_bf16 %0, %16, %7, %0\n v_mfma_f32_32x32x16_bf16 %0, %17, %8, %0\n v_mfma_f32_32x32x16_bf16 %0, %18, %9, %0\n" \
  : "+v"(acc) : WC((w)[0]), WC((w)[1]), WC((w)[2]), WC((w)[3]), WC((w)[4]), WC((w)[5]), WC((w)[6]), WC((w)[7]), WC((w)[8]), \
    "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]), "v"(f[8]))
template <int R, int STRIDE>
__global__ __launch_bounds__(256) void k(const u32x4* wsrc, float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, l = tid & 63;
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u;
  u32x4 w[18];
  for (int i = 0; i < 18; i++) w[i] = wsrc[i * 64 + l];
  __syncthreads();
  u32x4 fa[9], fb[9];
  const unsigned char* base = lds + STRIDE * (l & 31) + 16 * (l >> 5);
  for (int i = 0; i < 9; i++) { fa[i] = *reinterpret_cast<const u32x4*>(base + i * 32); fb[i] = *reinterpret_cast<const u32x4*>(base + 4096 + i * 32); }
  f32x16 acc = {};
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    for (int i = 0; i < R; i++) fb[i] = *reinterpret_cast<const u32x4*>(base + 4096 + ((it + i) & 15) * 32);
    B9(acc, "a", w, fa);
    for (int i = 0; i < R; i++) fa[i] = *reinterpret_cast<const u32x4*>(base + ((it + i) & 15) * 32);
    B9(acc, "a", (w + 9), fb);
  }
  const unsigned long long t1 = clock64();
  asm volatile("s_nop 15\n s_nop 15" : "+v"(acc));
  out[blockIdx.x * 256 + tid] = acc[0] + acc[5];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int R, int STRIDE>
void run(const char* name, const u32x4* w) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
  const int iters = 500;
  k<R, STRIDE><<<256, 256>>>(w, out, cyc, iters); k<R, STRIDE><<<256, 256>>>(w, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-52s %6.2f clk per MFMA\n", name, (double)c / (iters * 18.0));
}
int main() {
  u32x4* w; (void)hipMalloc(&w, 18 * 64 * 16); (void)hipMemset(w, 0x3c, 18 * 64 * 16);
  run<0, 32>("32x32x16, no LDS reads", w);
  run<9, 32>("32x32x16, 1 read per MFMA, linear (4 cycles)", w);
  run<9, 144>("32x32x16, 1 read per MFMA, 144 B pixel stride", w);
  run<9, 160>("32x32x16, 1 read per MFMA, 160 B stride (2-way)", w);
  run<9, 256>("32x32x16, 1 read per MFMA, 256 B stride (16-way)", w);
  return 0;
}
