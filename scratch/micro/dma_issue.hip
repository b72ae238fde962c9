// Issue cost of one LDS-DMA piece (1 KiB per wave-instruction) as the issuing wave sees it, with the whole chip streaming
// images the way the fused encoder does (4 waves per CU, ~11 pieces per wave and image, 256 workgroups):
//   A global_load_lds_dwordx4 voff, s[base:base+1]      (SGPR base + one lane-offset VGPR; the encoder's form)
//   B buffer_load_dwordx4 off, s[srd], soff lds          with ADD_TID_ENABLE in the resource: no address VGPR at all
// usage: ./dma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ img, int n_img, unsigned long long* cyc, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  constexpr int IMG = 84 * 84 * 6;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds);
  unsigned long long spent = 0;
  unsigned acc = 0;
  int buf = 0;
  const unsigned voff = (unsigned)tid * 16u;
  for (int i = blockIdx.x; i < n_img; i += gridDim.x) {
    const unsigned char* src = img + (long)i * IMG;
    const unsigned long long t0 = clock64();
    if (MODE == 2 || MODE == 3) {
      // C: the classic path - global_load_dwordx4 into VGPRs (4 per piece), ds_write_b128 once the data is there.
      //    MODE 2: five pieces in flight, written, five more (20 VGPRs); MODE 3: all ten in flight (40 VGPRs)
      constexpr int G = MODE == 2 ? 5 : 10;
      u32x4 v[G];
      for (int q0 = 0; q0 < 10; q0 += G) {
#pragma unroll
        for (int q = 0; q < G; q++)
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[q]) : "v"(voff), "s"(src + (q0 + q) * 4096) : "memory");
        if (MODE == 2) spent += 0;  // (the wait below is part of what the issuing wave pays in this form)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < G; q++)
          *reinterpret_cast<u32x4*>(lds + buf * 43008 + (q0 + q) * 4096 + tid * 16) = v[q];
      }
    } else if (MODE == 4) {
      // D: as C with ten pieces in flight, but the wait + LDS writes AFTER the image's other work (what a software-pipelined
      //    kernel would do: loads issued early, written late) - only the issue and the writes are charged
      u32x4 v[10];
#pragma unroll
      for (int q = 0; q < 10; q++)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[q]) : "v"(voff), "s"(src + q * 4096) : "memory");
      spent += clock64() - t0;
      for (int r = 0; r < 60; r++) asm volatile("s_sleep 16");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t1 = clock64();
#pragma unroll
      for (int q = 0; q < 10; q++) {
        asm volatile("" : "+v"(v[q]));
        *reinterpret_cast<u32x4*>(lds + buf * 43008 + q * 4096 + tid * 16) = v[q];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      spent += clock64() - t1;
      __syncthreads();
      acc += reinterpret_cast<const unsigned*>(lds + buf * 43008)[tid] + reinterpret_cast<const unsigned*>(lds + buf * 43008 + 40000)[tid & 63];
      buf ^= 1;
      continue;
    } else if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 10; q++) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2"
                     :: "v"(voff), "s"(__builtin_amdgcn_readfirstlane(lds0 + buf * 43008 + q * 4096 + w * 1024)), "s"(src + q * 4096) : "memory");
      }
    } else {
      // resource: base = src (48 bit), stride 16 (bits 48-61 of word 1), num_records = bytes, word3: ADD_TID_ENABLE (bit 23),
      // DATA_FORMAT 32 (bits 15-18 = 4), NUM_FORMAT uint? - the raw-buffer defaults hipcc uses (0x00027000) | add_tid
      const unsigned long long b = (unsigned long long)(src + w * 1024);
      u32x4 srd;
      srd[0] = (unsigned)b;
      srd[1] = (unsigned)(b >> 32) | (16u << 16);
      srd[2] = 42336u - (unsigned)w * 1024u;  // bytes from this wave's base: lanes past the image are clamped (zeros), never fault
      srd[3] = (1u << 23);  // ADD_TID_ENABLE; DATA_FORMAT must be 0 here: with ADD_TID it holds stride bits [17:14]
      u32x4 s;
      s[0] = __builtin_amdgcn_readfirstlane(srd[0]); s[1] = __builtin_amdgcn_readfirstlane(srd[1]);
      s[2] = __builtin_amdgcn_readfirstlane(srd[2]); s[3] = __builtin_amdgcn_readfirstlane(srd[3]);
#pragma unroll
      for (int q = 0; q < 10; q++) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 off, %1, %2 lds"
                     :: "s"(__builtin_amdgcn_readfirstlane(lds0 + buf * 43008 + q * 4096 + w * 1024)), "s"(s), "s"(q * 4096) : "memory");
      }
    }
    spent += clock64() - t0;
    // ~3 us of other work per image, as the encoder has (keeps the memory system at the encoder's load, not saturated)
    for (int r = 0; r < 60; r++) asm volatile("s_sleep 16");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc += reinterpret_cast<const unsigned*>(lds + buf * 43008)[tid] + reinterpret_cast<const unsigned*>(lds + buf * 43008 + 40000)[tid & 63];
    buf ^= 1;
  }
  if (l == 0) atomicAdd(cyc, spent);
  out[blockIdx.x * 256 + tid] = acc;
}
int main() {
  const int n_img = 6912;
  unsigned char* img; unsigned long long* cyc; unsigned* out;
  hipMalloc(&img, (size_t)n_img * 42336 + 8192); hipMalloc(&cyc, 8); hipMalloc(&out, 256 * 256 * 4);
  unsigned char* h = (unsigned char*)malloc((size_t)n_img * 42336);
  for (size_t i = 0; i < (size_t)n_img * 42336; i++) h[i] = (unsigned char)(i * 131 + (i >> 9));
  hipMemcpy(img, h, (size_t)n_img * 42336, hipMemcpyHostToDevice);
  auto run = [&](const char* name, void (*kfn)(const unsigned char*, int, unsigned long long*, unsigned*)) {
    hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 90000);
    unsigned first = 0;
    for (int rep = 0; rep < 3; rep++) {
      hipMemset(cyc, 0, 8);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(kfn, dim3(256), dim3(256), 90000, 0, img, n_img, cyc, out);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      unsigned o; hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
      if (rep == 0) first = o;
      if (rep == 2) printf("%-44s %.1f clk per piece (per wave), kernel %.3f ms, checksum %08x\n", name, (double)c / (256.0 * 4 * 27 * 10), ms, first);
    }
  };
  run("global_load_lds, SGPR base + lane VGPR", k<0>);
  run("buffer_load lds, ADD_TID resource, no VGPR", k<1>);
  run("global_load -> VGPR, wait, ds_write (5 in flight)", k<2>);
  run("global_load -> VGPR, wait, ds_write (10 in flight)", k<3>);
  run("global_load -> VGPR early, ds_write late", k<4>);
  return 0;
}
