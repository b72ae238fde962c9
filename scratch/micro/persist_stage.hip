// Floor of ONE wavefront stage of a persistent action-decoder RNN ("W_hh split across CUs, kept on chip across the
// 15 steps") against the launch-per-stage wavefront the product runs (rnn_gemm_kernel<128,64,2,8>:
// 3 problems x 64 workgroups, 21.7 us per launch in the step, 17 launches).
//
// A persistent stage still has to (1) re-ingest the hidden state the OTHER workgroups have just produced - with 128 x 64
// output tiles and W register/LDS-stationary that is 128 rows x 2048 k x 2 B = 512 KB per workgroup and stage (768 KB today,
// of which 256 KB are weights) - (2) publish its own 128 x 64 bf16 tile so that every XCD can read it, and (3) meet all 191
// other workgroups in a grid-wide barrier with agent-scope release / acquire (per-XCD L2s are not coherent).  This program
// runs exactly that skeleton - no MFMA, no epilogue arithmetic, no weight residency - so its per-stage time is a LOWER bound
// on a persistent stage:
//   mode 0  barrier only            (sc1 write-through tile stores + flat monotonic counter + acquire fence)
//   mode 1  ingest only             (512 KB per workgroup and stage by LDS-DMA through a 2 x 32 KB ring, no barrier)
//   mode 2  ingest + publish + barrier  (the stage skeleton)
// usage: ./persist_stage [stages=17] [reps=20]        build: hipcc --offload-arch=gfx950 -O3 persist_stage.hip -o persist_stage
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define NWG 192          // 3 problems x (2 row tiles x 32 column tiles): one workgroup per CU on 192 of the 256 CUs
#define ROWS 128
#define KDIM 2048

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  // producer side: every wave has drained its (write-through, sc1) stores; one lane arrives; consumer side: relaxed poll,
  // then ONE agent-scope acquire (invalidates this CU's L1), then the workgroup barrier that holds everyone until it is done
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (bounded: a grid that is not fully resident must end, not hang the box; slot [1] records a timeout)
    for (int spin = 0; spin < 4000000 && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; spin++) {
      __builtin_amdgcn_s_sleep(2);
      if (spin == 3999999) counter[1] = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(512) void stage_kernel(const unsigned char* __restrict__ hsrc, unsigned char* __restrict__ hdst,
                                                    unsigned* counter, unsigned* sink, int stages, unsigned epoch0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 2 x 32 KB ring
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int prob = blockIdx.x / 64, t = blockIdx.x % 64, mt = t & 1, nt = t >> 1;
  unsigned acc = 0;
  for (int s = 0; s < stages; s++) {
    // the hidden state of this problem as left by the previous stage: [256 rows][2048] bf16, ping-pong per stage
    const unsigned char* hin = hsrc + ((long)(s & 1) * 3 + prob) * (256L * KDIM * 2) + (long)mt * ROWS * KDIM * 2;
    if (MODE != 0) {
      // 16 K tiles of 128 rows x 128 k: 32 KB each = 32 wave-instructions of 1 KB (4 rows x 256 B), 4 per wave
      auto issue = [&](int kt, int slot) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int row4 = (w + 8 * q) * 4, r = row4 + (l >> 4), c = l & 15;
          const unsigned char* src = hin + (long)r * KDIM * 2 + kt * 256 + c * 16;
          const unsigned lds_off = __builtin_amdgcn_readfirstlane(
              (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)(lds + slot * 32768 + row4 * 256));
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(src), "s"(lds_off) : "memory");
        }
      };
      issue(0, 0);
      for (int kt = 0; kt < 16; kt++) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 1 < 16) issue(kt + 1, (kt + 1) & 1);
        acc += reinterpret_cast<const unsigned*>(lds + (kt & 1) * 32768)[tid];  // (one read per thread: the MFMA loop's reads are not modelled)
      }
    }
    if (MODE != 1) {
      // publish this workgroup's 128 x 64 bf16 output tile (16 KB) write-through: 2 x 16 B per thread
      unsigned char* hout = hdst + ((long)((s + 1) & 1) * 3 + prob) * (256L * KDIM * 2) + ((long)mt * ROWS) * KDIM * 2 + nt * 128;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int e = tid + 512 * q, r = e >> 3, c = e & 7;  // 128 rows x 8 chunks of 16 B
        u32x4 v = {acc + epoch0, (unsigned)s, (unsigned)e, 0u};
        unsigned char* dst = hout + (long)r * KDIM * 2 + c * 16;
        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");
      }
      grid_barrier(counter, epoch0 + (unsigned)(s + 1) * NWG);
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int stages = argc > 1 ? atoi(argv[1]) : 17, reps = argc > 2 ? atoi(argv[2]) : 20;
  const size_t hbytes = 2UL * 3 * 256 * KDIM * 2;
  unsigned char *h;
  unsigned *counter, *sink;
  hipMalloc(&h, hbytes); hipMemset(h, 1, hbytes);
  hipMalloc(&counter, 256); hipMemset(counter, 0, 256);
  hipMalloc(&sink, 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto k0 = stage_kernel<0>; auto k1 = stage_kernel<1>; auto k2 = stage_kernel<2>;
  hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const char* names[3] = {"barrier only (publish 16 KB sc1 + counter + acquire)", "ingest only (512 KB LDS-DMA per workgroup)", "ingest + publish + barrier"};
  unsigned epoch = 0;
  for (int mode = 0; mode < 3; mode++) {
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < reps + 2; r++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k0, dim3(NWG), dim3(512), 65536, 0, h, h, counter, sink, stages, epoch);
      if (mode == 1) hipLaunchKernelGGL(k1, dim3(NWG), dim3(512), 65536, 0, h, h, counter, sink, stages, epoch);
      if (mode == 2) hipLaunchKernelGGL(k2, dim3(NWG), dim3(512), 65536, 0, h, h, counter, sink, stages, epoch);
      hipEventRecord(e1);
      if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\n"); return 1; }
      if (mode != 1) epoch += (unsigned)stages * NWG;
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("mode %d  %-55s %7.2f us per stage (mean), %7.2f (best); launch of %d stages %.1f us\n", mode, names[mode],
           sum / reps / stages * 1e3f, best / stages * 1e3f, stages, sum / reps * 1e3f);
  }
  unsigned host[2] = {0, 0};
  hipMemcpy(host, counter, 8, hipMemcpyDeviceToHost);
  if (host[1]) printf("WARNING: a barrier timed out (grid not co-resident?)\n");
  printf("reference: the product's launch-per-stage wavefront takes 21.7 us per stage in the step (20.4 back to back), of which the\n"
         "MFMA + fragment reads are ~8 us on top of launch + barriers + epilogue (DESIGN.md, round-2 measurements)\n");
  return 0;
}
