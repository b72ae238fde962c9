// Dependent v_mfma_f32_16x16x32_bf16 chains (conv2's shape: 2 accumulators x 8-MFMA blocks): issue rate with the weight
// operand in AGPRs vs VGPRs, alone and with 16 ds_read_b128 per 32 MFMAs in flight (4 waves per CU, one per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define B8(acc, W, w, f) asm volatile( \
  "v_mfma_f32_16x16x32_bf16 %0, %1, %9, %0\n v_mfma_f32_16x16x32_bf16 %0, %2, %10, %0\n v_mfma_f32_16x16x32_bf16 %0, %3, %11, %0\n" \
  "v_mfma_f32_16x16x32_bf16 %0, %4, %12, %0\n v_mfma_f32_16x16x32_bf16 %0, %5, %13, %0\n v_mfma_f32_16x16x32_bf16 %0, %6, %14, %0\n" \
  "v_mfma_f32_16x16x32_bf16 %0, %7, %15, %0\n v_mfma_f32_16x16x32_bf16 %0, %8, %16, %0\n" \
  : "+v"(acc) : W(w[0]), W(w[1]), W(w[2]), W(w[3]), W(w[4]), W(w[5]), W(w[6]), W(w[7]), \
    "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]))
#define WA "a"
#define WV "v"
template <int AG, int LDS>
__global__ __launch_bounds__(256) void k(const u32x4* wsrc, float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, l = tid & 63;
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u;
  u32x4 w[32];
  for (int i = 0; i < 32; i++) w[i] = wsrc[i * 64 + l];
  __syncthreads();
  u32x4 fa[8], fb[8];
  const unsigned char* base = lds + 16 * l;
  for (int i = 0; i < 8; i++) { fa[i] = *reinterpret_cast<const u32x4*>(base + i * 1024); fb[i] = *reinterpret_cast<const u32x4*>(base + 8192 + i * 1024); }
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (LDS) for (int i = 0; i < 8; i++) fb[i] = *reinterpret_cast<const u32x4*>(base + 8192 + ((it + i) & 7) * 1024);
    if (AG) { B8(a0, WA, (w + 0), fa); B8(a1, WA, (w + 8), fa); } else { B8(a0, WV, (w + 0), fa); B8(a1, WV, (w + 8), fa); }
    if (LDS) for (int i = 0; i < 8; i++) fa[i] = *reinterpret_cast<const u32x4*>(base + ((it + i) & 7) * 1024);
    if (AG) { B8(a0, WA, (w + 16), fb); B8(a1, WA, (w + 24), fb); } else { B8(a0, WV, (w + 16), fb); B8(a1, WV, (w + 24), fb); }
  }
  const unsigned long long t1 = clock64();
  asm volatile("s_nop 15" : "+v"(a0), "+v"(a1));
  out[blockIdx.x * 256 + tid] = a0[0] + a1[1];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int AG, int LDS>
void run(const char* name, const u32x4* w) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
  const int iters = 500;
  k<AG, LDS><<<256, 256>>>(w, out, cyc, iters); k<AG, LDS><<<256, 256>>>(w, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-40s %6.2f clk per MFMA\n", name, (double)c / (iters * 32.0));
}
int main() {
  u32x4* w; (void)hipMalloc(&w, 32 * 64 * 16); (void)hipMemset(w, 0x3c, 32 * 64 * 16);
  run<0, 0>("weights in VGPRs, no LDS reads", w);
  run<1, 0>("weights in AGPRs, no LDS reads", w);
  run<0, 1>("weights in VGPRs, 16 reads / 32 MFMAs", w);
  run<1, 1>("weights in AGPRs, 16 reads / 32 MFMAs", w);
  return 0;
}
