// generated: 32x32x16 chains of 9 with the next block's 9 ds_read_b128 issued INSIDE the chain (RPG per MFMA gap)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define OPS(acc, nx, w, cu, addr) : "+v"(acc), "=&v"(nx[0]), "=&v"(nx[1]), "=&v"(nx[2]), "=&v"(nx[3]), "=&v"(nx[4]), "=&v"(nx[5]), "=&v"(nx[6]), "=&v"(nx[7]), "=&v"(nx[8]) \
  : "a"((w)[0]), "a"((w)[1]), "a"((w)[2]), "a"((w)[3]), "a"((w)[4]), "a"((w)[5]), "a"((w)[6]), "a"((w)[7]), "a"((w)[8]), \
    "v"(cu[0]), "v"(cu[1]), "v"(cu[2]), "v"(cu[3]), "v"(cu[4]), "v"(cu[5]), "v"(cu[6]), "v"(cu[7]), "v"(cu[8]), "v"(addr)
#define BLK1(acc, nx, w, cu, addr) asm volatile("s_waitcnt lgkmcnt(0)\n" "v_mfma_f32_32x32x16_bf16 %0, %19, %10, %0\n" "ds_read_b128 %1, %28 offset:0\n" "v_mfma_f32_32x32x16_bf16 %0, %20, %11, %0\n" "ds_read_b128 %2, %28 offset:32\n" "v_mfma_f32_32x32x16_bf16 %0, %21, %12, %0\n" "ds_read_b128 %3, %28 offset:64\n" "v_mfma_f32_32x32x16_bf16 %0, %22, %13, %0\n" "ds_read_b128 %4, %28 offset:96\n" "v_mfma_f32_32x32x16_bf16 %0, %23, %14, %0\n" "ds_read_b128 %5, %28 offset:128\n" "v_mfma_f32_32x32x16_bf16 %0, %24, %15, %0\n" "ds_read_b128 %6, %28 offset:160\n" "v_mfma_f32_32x32x16_bf16 %0, %25, %16, %0\n" "ds_read_b128 %7, %28 offset:192\n" "v_mfma_f32_32x32x16_bf16 %0, %26, %17, %0\n" "ds_read_b128 %8, %28 offset:224\n" "v_mfma_f32_32x32x16_bf16 %0, %27, %18, %0\n" "ds_read_b128 %9, %28 offset:256\n" OPS(acc, nx, w, cu, addr))
#define BLK2(acc, nx, w, cu, addr) asm volatile("s_waitcnt lgkmcnt(0)\n" "v_mfma_f32_32x32x16_bf16 %0, %19, %10, %0\n" "ds_read_b128 %1, %28 offset:0\n" "ds_read_b128 %2, %28 offset:32\n" "v_mfma_f32_32x32x16_bf16 %0, %20, %11, %0\n" "ds_read_b128 %3, %28 offset:64\n" "ds_read_b128 %4, %28 offset:96\n" "v_mfma_f32_32x32x16_bf16 %0, %21, %12, %0\n" "ds_read_b128 %5, %28 offset:128\n" "ds_read_b128 %6, %28 offset:160\n" "v_mfma_f32_32x32x16_bf16 %0, %22, %13, %0\n" "ds_read_b128 %7, %28 offset:192\n" "ds_read_b128 %8, %28 offset:224\n" "v_mfma_f32_32x32x16_bf16 %0, %23, %14, %0\n" "ds_read_b128 %9, %28 offset:256\n" "v_mfma_f32_32x32x16_bf16 %0, %24, %15, %0\n" "v_mfma_f32_32x32x16_bf16 %0, %25, %16, %0\n" "v_mfma_f32_32x32x16_bf16 %0, %26, %17, %0\n" "v_mfma_f32_32x32x16_bf16 %0, %27, %18, %0\n" OPS(acc, nx, w, cu, addr))
#define BLK3(acc, nx, w, cu, addr) asm volatile("s_waitcnt lgkmcnt(0)\n" "v_mfma_f32_32x32x16_bf16 %0, %19, %10, %0\n" "ds_read_b128 %1, %28 offset:0\n" "ds_read_b128 %2, %28 offset:32\n" "ds_read_b128 %3, %28 offset:64\n" "v_mfma_f32_32x32x16_bf16 %0, %20, %11, %0\n" "ds_read_b128 %4, %28 offset:96\n" "ds_read_b128 %5, %28 offset:128\n" "ds_read_b128 %6, %28 offset:160\n" "v_mfma_f32_32x32x16_bf16 %0, %21, %12, %0\n" "ds_read_b128 %7, %28 offset:192\n" "ds_read_b128 %8, %28 offset:224\n" "ds_read_b128 %9, %28 offset:256\n" "v_mfma_f32_32x32x16_bf16 %0, %22, %13, %0\n" "v_mfma_f32_32x32x16_bf16 %0, %23, %14, %0\n" "v_mfma_f32_32x32x16_bf16 %0, %24, %15, %0\n" "v_mfma_f32_32x32x16_bf16 %0, %25, %16, %0\n" "v_mfma_f32_32x32x16_bf16 %0, %26, %17, %0\n" "v_mfma_f32_32x32x16_bf16 %0, %27, %18, %0\n" OPS(acc, nx, w, cu, addr))

template <int RPG, int STRIDE>
__global__ __launch_bounds__(256) void k(const u32x4* wsrc, float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, l = tid & 63;
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u;
  u32x4 w[18];
  for (int i = 0; i < 18; i++) w[i] = wsrc[i * 64 + l];
  __syncthreads();
  u32x4 fa[9], fb[9];
  const unsigned addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds + STRIDE * (l & 31) + 16 * (l >> 5);
  for (int i = 0; i < 9; i++) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(fa[i]) : "v"(addr));
  f32x16 acc = {};
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (RPG == 1) { BLK1(acc, fb, w, fa, addr); BLK1(acc, fa, (w + 9), fb, addr); }
    if (RPG == 2) { BLK2(acc, fb, w, fa, addr); BLK2(acc, fa, (w + 9), fb, addr); }
    if (RPG == 3) { BLK3(acc, fb, w, fa, addr); BLK3(acc, fa, (w + 9), fb, addr); }
  }
  const unsigned long long t1 = clock64();
  asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" : "+v"(acc));
  out[blockIdx.x * 256 + tid] = acc[0] + acc[5];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int RPG, int STRIDE>
void run(const char* name, const u32x4* w) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
  const int iters = 500;
  k<RPG, STRIDE><<<256, 256>>>(w, out, cyc, iters); k<RPG, STRIDE><<<256, 256>>>(w, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-60s %6.2f clk per MFMA\n", name, (double)c / (iters * 18.0));
}
int main() {
  u32x4* w; (void)hipMalloc(&w, 18 * 64 * 16); (void)hipMemset(w, 0x3c, 18 * 64 * 16);
  run<1, 32>("1 read in every gap, linear (4 LDS cycles)", w);
  run<2, 32>("2 reads in the first 5 gaps, linear", w);
  run<3, 32>("3 reads in the first 3 gaps, linear", w);
  run<1, 144>("1 read in every gap, 144 B stride", w);
  run<2, 144>("2 reads in the first 5 gaps, 144 B stride", w);
  run<1, 160>("1 read in every gap, 160 B stride (2-way conflicts)", w);
  run<2, 160>("2 reads in the first 5 gaps, 160 B stride", w);
  return 0;
}
