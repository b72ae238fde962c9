// Whole-MLP forward in one launch (bf16 MFMA, fp32 accumulate) for the hot path's narrow MLPs:
// goal encoder [32,256,256,32], policy head [64,256,256,256,2A], Q head [64+A,256,256,256,1]
// (reference networks/actors/*.py, critics/*.py: nn.Sequential of Linear + SiLU/ReLU).
//
// As a chain of per-layer GEMM launches each layer costs ~13-25 us of launch + dependent global
// round trips for a few microseconds of work.  Here a workgroup keeps a 64-row block of the batch
// resident: the layer input lives in LDS as bf16, a wave owns 64 output columns (4 N tiles x 4 M
// tiles of accumulators), streams its weight rows straight from global memory into B fragments
// (register double buffer) and writes the fp32 pre-activation / activation rows the backward needs
// plus the bf16 copy that is the next layer's LDS input.
#include "mlp_fused.h"

#include "common.h"

namespace {

constexpr int BMF = 64;   // rows per workgroup
constexpr int XP = 264;   // LDS row pitch (bf16): 528 B -> 16 consecutive rows hit distinct bank groups
constexpr int MAXD = 256; // widest layer

struct MlpFwdArgs {
  const float* x[MF_MAXP];
  const float* params[MF_MAXP];
  const __bf16* pbf[MF_MAXP];  // bf16 copy of the parameter block (same element offsets)
  float* act[MF_MAXP];
  int M[MF_MAXP];
  long zoff[MF_MAXP][MF_MAXL], yoff[MF_MAXP][MF_MAXL];
  long woff[MF_MAXL], boff[MF_MAXL];
  int dims[MF_MAXL + 1], acts[MF_MAXL];
  int L, ldx;
};

// 8 consecutive bf16 weights of output row n (zeros outside the matrix); rows are 8-byte aligned
__device__ __forceinline__ bf16x8 load_w(const __bf16* __restrict__ W, int K, int N, int n, int k) {
  const bool on = n < N && k < K;  // K % 8 == 0: a fragment is inside the row or entirely outside
  const __bf16* q = W + (long)(on ? n : 0) * K + (on ? k : 0);
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(q), hi = *reinterpret_cast<const bf16x4*>(q + 4);
  const __bf16 z = (__bf16)0.f;
  return on ? bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]} : bf16x8{z, z, z, z, z, z, z, z};
}

__global__ __launch_bounds__(256) void mlp_fused_fwd_kernel(MlpFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* X = reinterpret_cast<__bf16*>(smem);  // [2][BMF][XP]
  const int p = blockIdx.y, m0 = blockIdx.x * BMF, M = a.M[p];
  if (m0 >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
  // zero both buffers once: padded K columns are multiplied by zero weights and must stay finite
  for (int e = tid; e < 2 * BMF * XP / 8; e += 256) reinterpret_cast<f32x4*>(X)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  {  // stage the input rows as bf16
    const int K0 = a.dims[0], c8 = K0 / 8;
    const float* x = a.x[p];
    for (int c = tid; c < BMF * c8; c += 256) {
      const int row = c / c8, k = (c - row * c8) * 8;
      if (m0 + row < M) {
        const float* q = x + (long)(m0 + row) * a.ldx + k;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(q), hi = *reinterpret_cast<const f32x4*>(q + 4);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 4; j++) { v[j] = (__bf16)lo[j]; v[4 + j] = (__bf16)hi[j]; }
        *reinterpret_cast<bf16x8*>(X + row * XP + k) = v;
      }
    }
  }
  __syncthreads();
  // LDS-only barrier: the fp32 activation rows written for the backward stay in flight across layers
  // (nothing in this kernel reads them back), __syncthreads() would drain them at every layer
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  const int n0 = 64 * w;
  bf16x8 B[8][4];
  auto load_layer = [&](int l) {  // every weight fragment of the layer in flight at once (K <= 256)
    const int K = a.dims[l], N = a.dims[l + 1];
    const __bf16* Wb = a.pbf[p] + a.woff[l];
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int nt = 0; nt < 4; nt++) B[ks][nt] = load_w(Wb, K, N, n0 + 16 * nt + i, 32 * ks + 8 * g);
  };
  if (n0 < a.dims[1]) load_layer(0);
  int cur = 0;
  for (int l = 0; l < a.L; l++) {
    const int K = a.dims[l], N = a.dims[l + 1], KS = (K + 31) / 32, act = a.acts[l];
    const float* bias = a.params[p] + a.boff[l];
    const __bf16* xin = X + cur * BMF * XP;
    __bf16* xout = X + (cur ^ 1) * BMF * XP;
    f32x4 acc[4][4];
    if (n0 < N) {  // wave-uniform: this wave owns output columns [n0, n0 + 64)
#pragma unroll
      for (int mt = 0; mt < 4; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        if (ks >= KS) break;
        bf16x8 A[4];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) A[mt] = *reinterpret_cast<const bf16x8*>(xin + (16 * mt + i) * XP + 32 * ks + 8 * g);
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
          if (n0 + 16 * nt < N) {
#pragma unroll
            for (int mt = 0; mt < 4; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[mt], B[ks][nt], acc[mt][nt], 0, 0, 0);
          }
      }
    }
    // the next layer's weights travel while this layer's epilogue runs
    if (l + 1 < a.L && n0 < a.dims[l + 2]) load_layer(l + 1);
    if (n0 < N) {
      float* zb = a.zoff[p][l] >= 0 ? a.act[p] + a.zoff[p][l] : nullptr;
      float* yb = a.act[p] + a.yoff[p][l];
#pragma unroll
      for (int nt = 0; nt < 4; nt++) {
        const int col = n0 + 16 * nt + i;
        if (n0 + 16 * nt >= N) continue;
        const bool cok = col < N;
        const float bv = cok ? bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; mt++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = 16 * mt + 4 * g + r;
            const float z = acc[mt][nt][r] + bv, y = act_apply(act, z);
            if (cok && m0 + row < M) {
              const long o = (long)(m0 + row) * N + col;
              if (zb) zb[o] = z;
              yb[o] = y;
            }
            xout[row * XP + col] = (__bf16)(cok ? y : 0.f);
          }
      }
    }
    lds_barrier();
    cur ^= 1;
  }
}

}  // namespace

bool mlp_fused_fwd_ok(int nprob, int L, const int* dims, int ldx) {
  if (nprob < 1 || nprob > MF_MAXP || L < 1 || L > MF_MAXL || ldx % 4) return false;
  for (int l = 0; l < L; l++)
    if (dims[l] % 8 || dims[l] > MAXD || dims[l] < 8) return false;
  return dims[L] >= 1 && dims[L] <= MAXD;
}

int mlp_fused_fwd(int nprob, const float* const* x, int ldx, const float* const* params, const void* const* params_bf16,
                  float* const* act,
                  const int* M, int L, const int* dims, const int* acts, const long* zoff, const long* yoff,
                  const long* woff, const long* boff, hipStream_t st) {
  MlpFwdArgs a{};
  int maxM = 0;
  for (int p = 0; p < nprob; p++) {
    if (((uintptr_t)x[p] & 15) || ((uintptr_t)params[p] & 15)) return TACORL_EINVAL;
    if ((uintptr_t)params_bf16[p] & 7) return TACORL_EINVAL;
    a.x[p] = x[p]; a.params[p] = params[p]; a.pbf[p] = (const __bf16*)params_bf16[p]; a.act[p] = act[p]; a.M[p] = M[p];
    for (int l = 0; l < L; l++) { a.zoff[p][l] = zoff[p * MF_MAXL + l]; a.yoff[p][l] = yoff[p * MF_MAXL + l]; }
    maxM = M[p] > maxM ? M[p] : maxM;
  }
  for (int l = 0; l < L; l++) {
    if (woff[l] % 4) return TACORL_EINVAL;
    a.woff[l] = woff[l]; a.boff[l] = boff[l]; a.dims[l] = dims[l]; a.acts[l] = acts[l];
  }
  a.dims[L] = dims[L]; a.L = L; a.ldx = ldx;
  if (maxM == 0) return TACORL_OK;
  constexpr size_t lds = (size_t)2 * BMF * XP * 2;
  static int once = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_fwd_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : -1;
  if (once) return TACORL_ELAUNCH;
  hipLaunchKernelGGL(mlp_fused_fwd_kernel, dim3((maxM + BMF - 1) / BMF, nprob), dim3(256), lds, st, a);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
