// LDS-DMA semantics + rate probes for the encoder's image re-layout (gfx950):
//  1. global_load_lds_dwordx3: where does lane j's 12-byte piece land (M0 + 12 j ?), 4-byte-aligned sources
//  2. global_load_lds_dwordx4 from 8- and 4-byte-aligned sources
//  3. rate: an 84x84x3 bf16 image (42 336 B) streamed per workgroup either verbatim (x4, 1 KiB pieces) or
//     re-laid as space-to-depth [21][21][4 rows][24 B] by x3 pieces (two 12-byte lanes per 24-byte row segment)
// usage: ./dma_x3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

__global__ void sem_x3(const unsigned char* src, const int* lane_src_off, unsigned char* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = 0xEE;
  __syncthreads();
  const unsigned char* p = src + lane_src_off[l];
  const unsigned lds_off = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)(lds + 64);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %0, off\n\ts_waitcnt vmcnt(0)"
               :: "v"(p), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
  __syncthreads();
  for (int i = l; i < 4096; i += 64) out[i] = lds[i];
}
__global__ void sem_x4(const unsigned char* src, const int* lane_src_off, unsigned char* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int l = threadIdx.x;
  for (int i = l; i < 4096; i += 64) lds[i] = 0xEE;
  __syncthreads();
  const unsigned char* p = src + lane_src_off[l];
  const unsigned lds_off = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)(lds + 64);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_waitcnt vmcnt(0)"
               :: "v"(p), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
  __syncthreads();
  for (int i = l; i < 4096; i += 64) out[i] = lds[i];
}

// rate: MODE 0 = verbatim x4; MODE 1 = s2d x3; MODE 2 = saddr-form x4 (SGPR base + constant lane offset)
template <int MODE>
__global__ __launch_bounds__(256) void rate(const unsigned char* __restrict__ img, int n_img, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  constexpr int IMG = 84 * 84 * 6;
  unsigned acc = 0;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds);
  int buf = 0;
  for (int i = blockIdx.x; i < n_img; i += gridDim.x) {
    const unsigned char* src = img + (long)i * IMG;
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 11; q++) {
        const int c0 = q * 256 + w * 64;
        if (c0 < IMG / 16 && c0 + l < IMG / 16) {
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                       :: "v"(src + (long)(c0 + l) * 16), "s"(__builtin_amdgcn_readfirstlane(lds0 + buf * 43008 + c0 * 16)) : "memory");
        }
      }
    } else if (MODE == 2) {
      const unsigned voff = (w * 64 + l) * 16;
#pragma unroll
      for (int q = 0; q < 11; q++) {
        const int c0 = q * 256 + w * 64;
        if (c0 < IMG / 16 && c0 + l < IMG / 16) {
          const unsigned char* sb = src + q * 4096;
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                       :: "v"(voff), "s"(__builtin_amdgcn_readfirstlane(lds0 + buf * 43008 + c0 * 16)), "s"(sb) : "memory");
        }
      }
    } else {
      // s2d: piece index j = ((Y*21 + X)*4 + dy)*2 + half, 12 B each: 3528 pieces = 55.1 wave-instructions of 64
#pragma unroll
      for (int q = 0; q < 14; q++) {
        const int j0 = (q * 4 + w) * 64;
        const int j = j0 + l;
        if (j0 < 3528 && j < 3528) {
          const int half = j & 1, dy = (j >> 1) & 3, yx = j >> 3, Y = yx / 21, X = yx - 21 * Y;
          const unsigned so = ((4 * Y + dy) * 84 + 4 * X) * 6 + 12 * half;
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %0, %2"
                       :: "v"(so), "s"(__builtin_amdgcn_readfirstlane(lds0 + buf * 43008 + j0 * 12)), "s"(src) : "memory");
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    __syncthreads();
    acc += reinterpret_cast<const unsigned*>(lds + (buf ^ 1) * 43008)[tid];
    buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) out[0] = 1;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
  std::vector<unsigned char> h(1 << 16);
  for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned char)(i * 7 + (i >> 8));
  unsigned char *d, *o; int* offs;
  CK(hipMalloc(&d, h.size())); CK(hipMalloc(&o, 4096)); CK(hipMalloc(&offs, 256));
  CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
  std::vector<unsigned char> r(4096);
  auto run = [&](const char* name, bool x3, std::vector<int> lo) {
    CK(hipMemcpy(offs, lo.data(), 256, hipMemcpyHostToDevice));
    if (x3) hipLaunchKernelGGL(sem_x3, dim3(1), dim3(64), 8192, 0, d, offs, o);
    else hipLaunchKernelGGL(sem_x4, dim3(1), dim3(64), 8192, 0, d, offs, o);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost));
    const int sz = x3 ? 12 : 16;
    int bad = 0;
    for (int l = 0; l < 64; l++)
      for (int b = 0; b < sz; b++)
        if (r[64 + l * sz + b] != h[lo[l] + b]) bad++;
    int untouched_ok = 1;
    for (int b = 0; b < 64; b++) if (r[b] != 0xEE) untouched_ok = 0;
    for (int b = 64 + 64 * sz; b < 64 + 64 * sz + 64; b++) if (r[b] != 0xEE) untouched_ok = 0;
    printf("%-44s mismatching bytes %d (of %d), guard bytes intact %d\n", name, bad, 64 * sz, untouched_ok);
    if (bad) {
      printf("   first lanes: ");
      for (int b = 0; b < 48; b++) printf("%02x%s", r[64 + b], (b % sz == sz - 1) ? " | " : " ");
      printf("\n   expected   : ");
      for (int b = 0; b < 48; b++) printf("%02x%s", h[lo[b / sz] + b % sz], (b % sz == sz - 1) ? " | " : " ");
      printf("\n");
    }
  };
  std::vector<int> lo(64);
  for (int l = 0; l < 64; l++) lo[l] = 12 * l;            run("x3 contiguous, 4-B aligned", true, lo);
  for (int l = 0; l < 64; l++) lo[l] = 504 * (l & 3) + 24 * (l >> 3) + 12 * ((l >> 2) & 1) + 1000;  run("x3 scattered rows (s2d pattern)", true, lo);
  for (int l = 0; l < 64; l++) lo[l] = 16 * l;            run("x4 contiguous, 16-B aligned", false, lo);
  for (int l = 0; l < 64; l++) lo[l] = 16 * l + 8;        run("x4 contiguous, 8-B aligned", false, lo);
  for (int l = 0; l < 64; l++) lo[l] = 16 * l + 4;        run("x4 contiguous, 4-B aligned", false, lo);
  for (int l = 0; l < 64; l++) lo[l] = 504 * (l & 7) + 16 * (l >> 3) + 8 * (l & 1);  run("x4 scattered rows, 8-B aligned", false, lo);

  // ---- rate
  const int n_img = 6912;
  unsigned char* img; unsigned* flag;
  CK(hipMalloc(&img, (size_t)n_img * 42336 + 4096)); CK(hipMalloc(&flag, 4));
  CK(hipMemset(img, 1, (size_t)n_img * 42336 + 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, void (*kfn)(const unsigned char*, int, unsigned*)) {
    CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 90000));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kfn, dim3(256), dim3(256), 90000, 0, img, n_img, flag);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kfn, dim3(256), dim3(256), 90000, 0, img, n_img, flag);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("%-36s %.4f ms  %.2f TB/s\n", name, ms, n_img * 42336.0 / ms / 1e9);
  };
  timeit("rate x4 verbatim (vaddr)", rate<0>);
  timeit("rate x4 verbatim (saddr)", rate<2>);
  timeit("rate x3 s2d re-layout", rate<1>);
  return 0;
}
