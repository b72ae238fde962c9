// Does a wave's LDS fragment traffic overlap with its dependent MFMA chain?  Per iteration: 9 ds_read_b128 (the NEXT
// iteration's B fragments) + 9 dependent v_mfma_f32_16x16x32_bf16 on the CURRENT ones (conv3's half tile).
//   MODE 0: the 9 reads, then the 9 MFMAs          (what encoder_fused.hip does)
//   MODE 1: MFMA, read, MFMA, read, ...             (one read in every MFMA gap)
//   MODE 2: reads only      MODE 3: MFMAs only
// with 4 and 8 waves per CU (one / two per SIMD).  Prints shader clocks per iteration and wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define MF(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define WAITL(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, l = tid & 63, r16 = l & 15, g = l >> 4;
  for (int i = tid; i < 16384; i += NT) reinterpret_cast<float*>(lds)[i] = 0.f;
  __syncthreads();
  const unsigned addr = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)(lds + r16 * 160 + 16 * g + (tid >> 6) * 4096);
  u32x4 w[9], fa[9], fb[9];
#pragma unroll
  for (int s = 0; s < 9; s++) { w[s] = u32x4{0x3c003c00u + s, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; fa[s] = u32x4{0, 0, 0, 0}; fb[s] = fa[s]; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it += 2) {
#define HALF(cur, nxt)                                                                                         \
    if (MODE == 0) {                                                                                             \
      RD(nxt[0], addr, 0); RD(nxt[1], addr, 160); RD(nxt[2], addr, 320); RD(nxt[3], addr, 1440); RD(nxt[4], addr, 1600); \
      RD(nxt[5], addr, 1760); RD(nxt[6], addr, 2880); RD(nxt[7], addr, 3040); RD(nxt[8], addr, 3200);          \
      MF(acc, w[0], cur[0]); MF(acc, w[1], cur[1]); MF(acc, w[2], cur[2]); MF(acc, w[3], cur[3]); MF(acc, w[4], cur[4]); \
      MF(acc, w[5], cur[5]); MF(acc, w[6], cur[6]); MF(acc, w[7], cur[7]); MF(acc, w[8], cur[8]);                \
      WAITL(0);                                                                                                  \
    } else if (MODE == 1) {                                                                                      \
      MF(acc, w[0], cur[0]); RD(nxt[0], addr, 0); MF(acc, w[1], cur[1]); RD(nxt[1], addr, 160);                  \
      MF(acc, w[2], cur[2]); RD(nxt[2], addr, 320); MF(acc, w[3], cur[3]); RD(nxt[3], addr, 1440);               \
      MF(acc, w[4], cur[4]); RD(nxt[4], addr, 1600); MF(acc, w[5], cur[5]); RD(nxt[5], addr, 1760);              \
      MF(acc, w[6], cur[6]); RD(nxt[6], addr, 2880); MF(acc, w[7], cur[7]); RD(nxt[7], addr, 3040);              \
      MF(acc, w[8], cur[8]); RD(nxt[8], addr, 3200);                                                             \
      WAITL(0);                                                                                                  \
    } else if (MODE == 2) {                                                                                      \
      RD(nxt[0], addr, 0); RD(nxt[1], addr, 160); RD(nxt[2], addr, 320); RD(nxt[3], addr, 1440); RD(nxt[4], addr, 1600); \
      RD(nxt[5], addr, 1760); RD(nxt[6], addr, 2880); RD(nxt[7], addr, 3040); RD(nxt[8], addr, 3200);          \
      WAITL(0);                                                                                                  \
    } else {                                                                                                     \
      MF(acc, w[0], cur[0]); MF(acc, w[1], cur[1]); MF(acc, w[2], cur[2]); MF(acc, w[3], cur[3]); MF(acc, w[4], cur[4]); \
      MF(acc, w[5], cur[5]); MF(acc, w[6], cur[6]); MF(acc, w[7], cur[7]); MF(acc, w[8], cur[8]);                \
    }
    HALF(fa, fb)
    HALF(fb, fa)
  }
  asm volatile("s_nop 15" : "+v"(acc));
  const unsigned long long t1 = clock64();
  u32x4 x = fa[0] ^ fb[0] ^ fa[8] ^ fb[8];
  out[blockIdx.x * NT + tid] = acc[0] + acc[1] + acc[2] + acc[3] + (float)(x[0] ^ x[1] ^ x[2] ^ x[3]);
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int NT>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * NT * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  k<MODE, NT><<<256, NT>>>(out, cyc, iters);
  k<MODE, NT><<<256, NT>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s waves/CU %d: %7.1f clk per (9 reads + 9 MFMAs) per wave -> %5.2f MFMAs/clk/CU\n", name, NT / 64, (double)c / iters,
         (NT / 64) * 9.0 * iters / (double)c);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 256>("reads then MFMAs"); run<1, 256>("interleaved"); run<2, 256>("reads only"); run<3, 256>("MFMAs only");
  run<0, 512>("reads then MFMAs"); run<1, 512>("interleaved"); run<2, 512>("reads only"); run<3, 512>("MFMAs only");
  return 0;
}
