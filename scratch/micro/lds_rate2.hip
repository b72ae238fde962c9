// ds_read_b128 rate vs address pattern and waves per SIMD (reads only, 16 in flight per wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int PAT, int NT>
__global__ __launch_bounds__(NT) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, l = tid & 63, r16 = l & 15, g = l >> 4;
  for (int i = tid; i < 16384; i += NT) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 63);
  __syncthreads();
  u32x4 x = {0, 0, 0, 0};
  int off;
  if (PAT == 0) off = 16 * l;                       // linear: 1 KiB contiguous per instruction
  else if (PAT == 1) off = r16 * 80 + 16 * g;       // encoder conv2 pattern
  else if (PAT == 2) off = r16 * 64 + 16 * g;       // 16 rows x 64 B (plain row-major 16x32 bf16 tile)
  else off = (r16 * 64 + 16 * (g ^ (r16 & 3)));     // same, XOR-swizzled chunk
  const unsigned char* base = lds + off;
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    u32x4 f[16];
    const unsigned char* b2 = base + (it & 7) * 1024;
#pragma unroll
    for (int s = 0; s < 16; s++) f[s] = *reinterpret_cast<const u32x4*>(b2 + s * 2048);
#pragma unroll
    for (int s = 0; s < 16; s++) x ^= f[s];
  }
  const unsigned long long t1 = clock64();
  out[blockIdx.x * NT + tid] = (float)(x[0] ^ x[1] ^ x[2] ^ x[3]);
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int PAT, int NT>
void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * NT * 4); hipMalloc(&cyc, 8);
  const int iters = 400;
  k<PAT, NT><<<256, NT>>>(out, cyc, iters);
  k<PAT, NT><<<256, NT>>>(out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per = (double)c / (iters * 16.0);
  printf("%-34s waves/CU %d: %6.1f clk per read per wave -> %6.1f B/clk/CU\n", name, NT / 64, per, (NT / 64) * 1024.0 / per);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 256>("linear");      run<0, 512>("linear");      run<0, 1024>("linear");
  run<1, 256>("r16*80+16g");  run<1, 512>("r16*80+16g");
  run<2, 256>("r16*64+16g");  run<3, 256>("r16*64 swizzled");
  return 0;
}
