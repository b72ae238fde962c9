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

constexpr int BMF = 32;   // rows per workgroup (32: twice the workgroups of 64 - these kernels are latency-bound,
                          // the Q MLP's 3328 rows give 104 vs 208 workgroups on 256 CUs)
constexpr int MTF = BMF / 16;
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

// bf16-mode activation math: hardware exp / reciprocal (relative error ~1e-6, far below the bf16 operand
// rounding of these kernels); the exact expf / IEEE-division forms cost ~3 us per layer and workgroup here
__device__ __forceinline__ float act_fast(int act, float z) {
  if (act == ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == ACT_SILU) return z * __frcp_rn(1.f + __expf(-z));
  return z;
}
__device__ __forceinline__ float act_grad_fast(int act, float zy) {
  if (act == ACT_RELU) return zy > 0.f ? 1.f : 0.f;
  if (act == ACT_SILU) {
    const float sg = __frcp_rn(1.f + __expf(-zy));
    return sg * (1.f + zy * (1.f - sg));
  }
  return 1.f;
}

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
  if (n0 < a.dims[1]) load_layer(0);  // travels while the input rows are staged
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
  int cur = 0;
  for (int l = 0; l < a.L; l++) {
    const int K = a.dims[l], N = a.dims[l + 1], KS = (K + 31) / 32, act = a.acts[l];
    const float* bias = a.params[p] + a.boff[l];
    const __bf16* xin = X + cur * BMF * XP;
    __bf16* xout = X + (cur ^ 1) * BMF * XP;
    f32x4 acc[MTF][4], bvv[4];
    const bool vec = (N & 3) == 0;
    if (n0 < N) {  // wave-uniform: this wave owns output columns [n0, n0 + 64)
#pragma unroll
      for (int nt = 0; nt < 4; nt++) {  // bias vectors travel under the MFMA loop (clamped address, masked below)
        const int col = n0 + 16 * nt + 4 * g;
        bvv[nt] = *reinterpret_cast<const f32x4*>(bias + (vec && col < N ? col : 0));
      }
#pragma unroll
      for (int mt = 0; mt < MTF; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        if (ks >= KS) break;
        bf16x8 A[MTF];
#pragma unroll
        for (int mt = 0; mt < MTF; mt++) A[mt] = *reinterpret_cast<const bf16x8*>(xin + (16 * mt + i) * XP + 32 * ks + 8 * g);
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
          if (n0 + 16 * nt < N) {
#pragma unroll
            // weights as the A operand: D[n][m] - a lane then holds 4 CONSECUTIVE output columns of one row,
            // so the epilogue moves 16-byte vectors instead of 4-byte gathers
            for (int mt = 0; mt < MTF; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ks][nt], A[mt], acc[mt][nt], 0, 0, 0);
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
        const int col = n0 + 16 * nt + 4 * g;  // this lane's 4 consecutive columns
        if (n0 + 16 * nt >= N) continue;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (vec && col < N) bv = bvv[nt];
        else {
#pragma unroll
          for (int r = 0; r < 4; r++) bv[r] = col + r < N ? bias[col + r] : 0.f;
        }
#pragma unroll
        for (int mt = 0; mt < MTF; mt++) {
          const int row = 16 * mt + i;
          f32x4 z = acc[mt][nt] + bv, y;
#pragma unroll
          for (int r = 0; r < 4; r++) y[r] = col + r < N ? act_fast(act, z[r]) : 0.f;
          if (m0 + row < M) {
            const long o = (long)(m0 + row) * N + col;
            if (vec && col < N) {
              if (zb) *reinterpret_cast<f32x4*>(zb + o) = z;
              *reinterpret_cast<f32x4*>(yb + o) = y;
            } else {
#pragma unroll
              for (int r = 0; r < 4; r++)
                if (col + r < N) { if (zb) zb[o + r] = z[r]; yb[o + r] = y[r]; }
            }
          }
          *reinterpret_cast<bf16x4*>(xout + row * XP + col) = bf16x4{(__bf16)y[0], (__bf16)y[1], (__bf16)y[2], (__bf16)y[3]};
        }
      }
    }
    lds_barrier();
    cur ^= 1;
  }
}

// ---------------------------------------------------------------------------------- backward
// The input-gradient chain dZ_{l-1} = (dZ_l W_l) * act'_{l-1} of the whole MLP in one launch: the same
// 64-row resident structure run through the transposed network.  Wt_l = W_l^T (bf16, [K_l][NP_l], NP_l =
// N_l rounded up to 8, zero padded) comes from mlp_pack_wt_kernel.  Every dZ_l is also written to HBM
// (fp32) for the weight-gradient GEMMs, which no longer sit on the dependent chain.
struct MlpBwdArgs {
  const float* d_out[MF_MAXP];   // [M][ldo] gradient of the MLP output (= dZ of the last layer)
  const float* act[MF_MAXP];     // saved activations of the forward
  const __bf16* wt[MF_MAXP];     // packed transposed weights, all layers
  float* dz[MF_MAXP];            // dZ_l for l = 0 .. L-2 at dzoff[p][l] (floats), [M][dims[l+1]]
  float* d_x[MF_MAXP];           // optional [M][ldd] input gradient
  int M[MF_MAXP];
  long srcoff[MF_MAXP][MF_MAXL];  // activation-derivative source of layer l's output (z for SiLU, y for ReLU), -1 none
  long dzoff[MF_MAXP][MF_MAXL];
  long wtoff[MF_MAXL];            // element offsets into wt[p]
  int dims[MF_MAXL + 1], acts[MF_MAXL];
  int L, ldo, ldd;
};

struct WtPackArgs {
  const float* params[MF_MAXP];
  __bf16* wt[MF_MAXP];
  long woff[MF_MAXL], wtoff[MF_MAXL];
  int dims[MF_MAXL + 1];
  int L;
};
// grid (blocks, L, nprob): Wt[kin][n] = bf16(W[n][kin]), n < NP = roundup(N, 8) (zeros for n >= N)
__global__ __launch_bounds__(256) void mlp_pack_wt_kernel(WtPackArgs a) {
  const int l = blockIdx.y, p = blockIdx.z, K = a.dims[l], N = a.dims[l + 1], NP = (N + 7) / 8 * 8;
  const float* __restrict__ W = a.params[p] + a.woff[l];
  __bf16* __restrict__ T = a.wt[p] + a.wtoff[l];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < K * NP; e += gridDim.x * 256) {
    const int kin = e / NP, n = e - kin * NP;
    T[e] = (__bf16)(n < N ? W[(long)n * K + kin] : 0.f);
  }
}

__global__ __launch_bounds__(256) void mlp_fused_bwd_kernel(MlpBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* X = reinterpret_cast<__bf16*>(smem);  // [2][BMF][XP]
  const int p = blockIdx.y, m0 = blockIdx.x * BMF, M = a.M[p];
  if (m0 >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
  const int n0 = 64 * w;  // this wave's output columns (input features of the layer)
  bf16x8 B[8][4];
  auto load_layer = [&](int l) {
    const int KO = a.dims[l], NP = (a.dims[l + 1] + 7) / 8 * 8;  // outputs, (padded) reduction length
    const __bf16* T = a.wt[p] + a.wtoff[l];
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int nt = 0; nt < 4; nt++) B[ks][nt] = load_w(T, NP, KO, n0 + 16 * nt + i, 32 * ks + 8 * g);
  };
  const int l_last = a.d_x[p] ? 0 : 1;  // layer 0's dgrad only if the input gradient is wanted
  if (a.L - 1 >= l_last && n0 < a.dims[a.L - 1]) load_layer(a.L - 1);  // travels while d_out is staged
  for (int e = tid; e < 2 * BMF * XP / 8; e += 256) reinterpret_cast<f32x4*>(X)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  {  // stage dZ of the last layer (= d_out rows) as bf16
    const int NL = a.dims[a.L];
    const float* d = a.d_out[p];
    for (int c = tid; c < BMF * NL; c += 256) {
      const int row = c / NL, n = c - row * NL;
      if (m0 + row < M) X[row * XP + n] = (__bf16)d[(long)(m0 + row) * a.ldo + n];
    }
  }
  __syncthreads();
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  int cur = 0;
  for (int l = a.L - 1; l >= l_last; l--) {
    const int KO = a.dims[l], NR = a.dims[l + 1], KS = (NR + 31) / 32;
    const __bf16* xin = X + cur * BMF * XP;
    __bf16* xout = X + (cur ^ 1) * BMF * XP;
    f32x4 acc[MTF][4], svv[MTF][4];
    const int pact = l > 0 ? a.acts[l - 1] : ACT_NONE;
    const float* src = (l > 0 && a.srcoff[p][l - 1] >= 0) ? a.act[p] + a.srcoff[p][l - 1] : nullptr;
    const int ldout = l > 0 ? KO : a.ldd;
    const bool vec = (KO & 3) == 0 && (ldout & 3) == 0;
    if (n0 < KO) {
      if (src && vec) {  // the activation-derivative sources travel under the MFMA loop (clamped, masked below)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
#pragma unroll
          for (int mt = 0; mt < MTF; mt++) {
            const int col = n0 + 16 * nt + 4 * g, row = m0 + 16 * mt + i;
            const bool ok = row < M && col < KO;
            svv[mt][nt] = *reinterpret_cast<const f32x4*>(src + (ok ? (long)row * KO + col : 0));
          }
      }
#pragma unroll
      for (int mt = 0; mt < MTF; mt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        if (ks >= KS) break;
        bf16x8 A[MTF];
#pragma unroll
        for (int mt = 0; mt < MTF; mt++) A[mt] = *reinterpret_cast<const bf16x8*>(xin + (16 * mt + i) * XP + 32 * ks + 8 * g);
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
          if (n0 + 16 * nt < KO) {
#pragma unroll
            for (int mt = 0; mt < MTF; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ks][nt], A[mt], acc[mt][nt], 0, 0, 0);
          }
      }
    }
    if (l - 1 >= l_last && n0 < a.dims[l - 1]) load_layer(l - 1);
    if (n0 < KO) {
      float* out = l > 0 ? a.dz[p] + a.dzoff[p][l - 1] : a.d_x[p];
#pragma unroll
      for (int nt = 0; nt < 4; nt++) {
        const int col = n0 + 16 * nt + 4 * g;  // this lane's 4 consecutive columns (D = W^T-frag x dZ-frag)
        if (n0 + 16 * nt >= KO) continue;
#pragma unroll
        for (int mt = 0; mt < MTF; mt++) {
          const int row = 16 * mt + i;
          const bool rok = m0 + row < M;
          f32x4 v = acc[mt][nt];
          if (rok && src) {
            if (vec && col < KO) {
              const f32x4 sv = svv[mt][nt];
#pragma unroll
              for (int r = 0; r < 4; r++) v[r] *= act_grad_fast(pact, sv[r]);
            } else {
#pragma unroll
              for (int r = 0; r < 4; r++)
                if (col + r < KO) v[r] *= act_grad_fast(pact, src[(long)(m0 + row) * KO + col + r]);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (!rok || col + r >= KO) v[r] = 0.f;
          if (rok) {
            float* o = out + (long)(m0 + row) * ldout + col;
            if (vec && col < KO) *reinterpret_cast<f32x4*>(o) = v;
            else {
#pragma unroll
              for (int r = 0; r < 4; r++)
                if (col + r < KO) o[r] = v[r];
            }
          }
          *reinterpret_cast<bf16x4*>(xout + row * XP + col) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
      }
    }
    lds_barrier();
    cur ^= 1;
  }
}

}  // namespace

bool mlp_fused_bwd_ok(int nprob, int L, const int* dims, int ldo, int ldd) {
  if (nprob < 1 || nprob > MF_MAXP || L < 1 || L > MF_MAXL) return false;
  for (int l = 0; l <= L; l++)
    if (dims[l] < 1 || dims[l] > MAXD) return false;
  return true;
}
size_t mlp_fused_wt_elems(int L, const int* dims, long* wtoff) {
  size_t off = 0;
  for (int l = 0; l < L; l++) {
    if (wtoff) wtoff[l] = (long)off;
    off += (size_t)dims[l] * ((dims[l + 1] + 7) / 8 * 8);
    off = (off + 7) & ~(size_t)7;  // 16-byte aligned matrices
  }
  return off;
}
int mlp_fused_bwd(int nprob, const float* const* params, const float* const* act, const float* const* d_out, int ldo,
                  float* const* dz, float* const* d_x, int ldd, void* const* wt, const int* M, int L, const int* dims,
                  const int* acts, const long* srcoff, const long* dzoff, const long* woff, hipStream_t st, int mode) {
  MlpBwdArgs a{};
  WtPackArgs k{};
  long wtoff[MF_MAXL];
  mlp_fused_wt_elems(L, dims, wtoff);
  int maxM = 0;
  for (int p = 0; p < nprob; p++) {
    if ((uintptr_t)wt[p] & 15) return TACORL_EINVAL;
    a.d_out[p] = d_out ? d_out[p] : nullptr; a.act[p] = act ? act[p] : nullptr; a.wt[p] = (const __bf16*)wt[p]; a.dz[p] = dz[p];
    a.d_x[p] = d_x ? d_x[p] : nullptr; a.M[p] = M[p];
    k.params[p] = params[p]; k.wt[p] = (__bf16*)wt[p];
    for (int l = 0; l < L; l++) { a.srcoff[p][l] = srcoff[p * MF_MAXL + l]; a.dzoff[p][l] = dzoff[p * MF_MAXL + l]; }
    maxM = M[p] > maxM ? M[p] : maxM;
  }
  for (int l = 0; l < L; l++) { a.wtoff[l] = wtoff[l]; k.wtoff[l] = wtoff[l]; k.woff[l] = woff[l]; a.acts[l] = acts[l]; }
  for (int l = 0; l <= L; l++) { a.dims[l] = dims[l]; k.dims[l] = dims[l]; }
  a.L = L; k.L = L; a.ldo = ldo; a.ldd = ldd;
  if (maxM == 0) return TACORL_OK;
  if (mode != 2) hipLaunchKernelGGL(mlp_pack_wt_kernel, dim3(64, L, nprob), dim3(256), 0, st, k);
  if (mode == 1) return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
  constexpr size_t lds = (size_t)2 * BMF * XP * 2;
  static int once = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_bwd_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : -1;
  if (once) return TACORL_ELAUNCH;
  hipLaunchKernelGGL(mlp_fused_bwd_kernel, dim3((maxM + BMF - 1) / BMF, nprob), dim3(256), lds, st, a);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

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
