// generated: 16x16x32, two chains of 8 per block, the next block's 8 fragment reads inside the chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define OPS(a0, a1, nx, w, cu, addr) : "+v"(a0), "+v"(a1), "=&v"(nx[0]), "=&v"(nx[1]), "=&v"(nx[2]), "=&v"(nx[3]), "=&v"(nx[4]), "=&v"(nx[5]), "=&v"(nx[6]), "=&v"(nx[7]) \
  : "a"((w)[0]), "a"((w)[1]), "a"((w)[2]), "a"((w)[3]), "a"((w)[4]), "a"((w)[5]), "a"((w)[6]), "a"((w)[7]), "a"((w)[8]), "a"((w)[9]), "a"((w)[10]), "a"((w)[11]), "a"((w)[12]), "a"((w)[13]), "a"((w)[14]), "a"((w)[15]), \
    "v"(cu[0]), "v"(cu[1]), "v"(cu[2]), "v"(cu[3]), "v"(cu[4]), "v"(cu[5]), "v"(cu[6]), "v"(cu[7]), "v"(addr)
#define BLKA(a0, a1, nx, w, cu, addr) asm volatile("s_waitcnt lgkmcnt(0)\n" "v_mfma_f32_16x16x32_bf16 %0, %10, %26, %0\n" "ds_read_b128 %2, %34 offset:0\n" "v_mfma_f32_16x16x32_bf16 %0, %11, %27, %0\n" "ds_read_b128 %3, %34 offset:64\n" "v_mfma_f32_16x16x32_bf16 %0, %12, %28, %0\n" "ds_read_b128 %4, %34 offset:128\n" "v_mfma_f32_16x16x32_bf16 %0, %13, %29, %0\n" "ds_read_b128 %5, %34 offset:192\n" "v_mfma_f32_16x16x32_bf16 %0, %14, %30, %0\n" "ds_read_b128 %6, %34 offset:256\n" "v_mfma_f32_16x16x32_bf16 %0, %15, %31, %0\n" "ds_read_b128 %7, %34 offset:320\n" "v_mfma_f32_16x16x32_bf16 %0, %16, %32, %0\n" "ds_read_b128 %8, %34 offset:384\n" "v_mfma_f32_16x16x32_bf16 %0, %17, %33, %0\n" "ds_read_b128 %9, %34 offset:448\n" "v_mfma_f32_16x16x32_bf16 %1, %18, %26, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %19, %27, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %20, %28, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %21, %29, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %22, %30, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %23, %31, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %24, %32, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %25, %33, %1\n" OPS(a0, a1, nx, w, cu, addr))
#define BLKB(a0, a1, nx, w, cu, addr) asm volatile("s_waitcnt lgkmcnt(0)\n" "v_mfma_f32_16x16x32_bf16 %0, %10, %26, %0\n" "ds_read_b128 %2, %34 offset:0\n" "v_mfma_f32_16x16x32_bf16 %0, %11, %27, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %12, %28, %0\n" "ds_read_b128 %3, %34 offset:64\n" "v_mfma_f32_16x16x32_bf16 %0, %13, %29, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %14, %30, %0\n" "ds_read_b128 %4, %34 offset:128\n" "v_mfma_f32_16x16x32_bf16 %0, %15, %31, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %16, %32, %0\n" "ds_read_b128 %5, %34 offset:192\n" "v_mfma_f32_16x16x32_bf16 %0, %17, %33, %0\n" "v_mfma_f32_16x16x32_bf16 %1, %18, %26, %1\n" "ds_read_b128 %6, %34 offset:256\n" "v_mfma_f32_16x16x32_bf16 %1, %19, %27, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %20, %28, %1\n" "ds_read_b128 %7, %34 offset:320\n" "v_mfma_f32_16x16x32_bf16 %1, %21, %29, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %22, %30, %1\n" "ds_read_b128 %8, %34 offset:384\n" "v_mfma_f32_16x16x32_bf16 %1, %23, %31, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %24, %32, %1\n" "ds_read_b128 %9, %34 offset:448\n" "v_mfma_f32_16x16x32_bf16 %1, %25, %33, %1\n" OPS(a0, a1, nx, w, cu, addr))
#define BLKC(a0, a1, nx, w, cu, addr) asm volatile("s_waitcnt lgkmcnt(0)\n" "v_mfma_f32_16x16x32_bf16 %0, %10, %26, %0\n" "ds_read_b128 %2, %34 offset:0\n" "ds_read_b128 %3, %34 offset:64\n" "v_mfma_f32_16x16x32_bf16 %0, %11, %27, %0\n" "ds_read_b128 %4, %34 offset:128\n" "ds_read_b128 %5, %34 offset:192\n" "v_mfma_f32_16x16x32_bf16 %0, %12, %28, %0\n" "ds_read_b128 %6, %34 offset:256\n" "ds_read_b128 %7, %34 offset:320\n" "v_mfma_f32_16x16x32_bf16 %0, %13, %29, %0\n" "ds_read_b128 %8, %34 offset:384\n" "ds_read_b128 %9, %34 offset:448\n" "v_mfma_f32_16x16x32_bf16 %0, %14, %30, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %15, %31, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %16, %32, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %17, %33, %0\n" "v_mfma_f32_16x16x32_bf16 %1, %18, %26, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %19, %27, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %20, %28, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %21, %29, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %22, %30, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %23, %31, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %24, %32, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %25, %33, %1\n" OPS(a0, a1, nx, w, cu, addr))
#define BLKD(a0, a1, nx, w, cu, addr) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %10, %26, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %11, %27, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %12, %28, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %13, %29, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %14, %30, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %15, %31, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %16, %32, %0\n" "v_mfma_f32_16x16x32_bf16 %0, %17, %33, %0\n" "v_mfma_f32_16x16x32_bf16 %1, %18, %26, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %19, %27, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %20, %28, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %21, %29, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %22, %30, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %23, %31, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %24, %32, %1\n" "v_mfma_f32_16x16x32_bf16 %1, %25, %33, %1\n" OPS(a0, a1, nx, w, cu, addr))
template <int V, int STRIDE>
__global__ __launch_bounds__(256) void k(const u32x4* wsrc, float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, l = tid & 63;
  for (int i = tid; i < 16384; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u;
  u32x4 w[32];
  for (int i = 0; i < 32; i++) w[i] = wsrc[i * 64 + l];
  __syncthreads();
  u32x4 fa[8], fb[8];
  const unsigned addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds + STRIDE * (l & 15) + 16 * (l >> 4);
  for (int i = 0; i < 8; i++) { asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(fa[i]) : "v"(addr)); asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(fb[i]) : "v"(addr)); }
  f32x4 a0 = {}, a1 = {};
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (V == 0) { BLKA(a0, a1, fb, w, fa, addr); BLKA(a0, a1, fa, (w + 16), fb, addr); }
    if (V == 1) { BLKB(a0, a1, fb, w, fa, addr); BLKB(a0, a1, fa, (w + 16), fb, addr); }
    if (V == 2) { BLKC(a0, a1, fb, w, fa, addr); BLKC(a0, a1, fa, (w + 16), fb, addr); }
    if (V == 3) { BLKD(a0, a1, fb, w, fa, addr); BLKD(a0, a1, fa, (w + 16), fb, addr); }
  }
  const unsigned long long t1 = clock64();
  asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" : "+v"(a0), "+v"(a1));
  out[blockIdx.x * 256 + tid] = a0[0] + a1[1];
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V, int STRIDE>
void run(const char* name, const u32x4* w) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 8);
  const int iters = 500;
  k<V, STRIDE><<<256, 256>>>(w, out, cyc, iters); k<V, STRIDE><<<256, 256>>>(w, out, cyc, iters);
  (void)hipDeviceSynchronize();
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-60s %6.2f clk per MFMA\n", name, (double)c / (iters * 32.0));
}
int main() {
  u32x4* w; (void)hipMalloc(&w, 32 * 64 * 16); (void)hipMemset(w, 0x3c, 32 * 64 * 16);
  run<3, 80>("16x16x32 no reads", w);
  run<0, 80>("16x16x32, 8 reads / 16 MFMAs: 1 in each of the first 8 gaps", w);
  run<1, 80>("16x16x32, 8 reads / 16 MFMAs: 1 in every second gap", w);
  run<2, 80>("16x16x32, 8 reads / 16 MFMAs: 2 in each of the first 4 gaps", w);
  return 0;
}
