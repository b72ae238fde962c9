// Whole-MLP forward in one launch (bf16 MFMA, fp32 accumulate) for the hot path's narrow MLPs:
// goal encoder [32,256,256,32], policy head [64,256,256,256,2A], Q head [64+A,256,256,256,1]
// (reference networks/actors/*.py, critics/*.py: nn.Sequential of Linear + SiLU/ReLU).
//
// As a chain of per-layer GEMM launches each layer costs ~13-25 us of launch + dependent global
// round trips for a few microseconds of work.  Here a workgroup keeps a 32-row block of the batch
// resident: the layer input lives in LDS as bf16, each of the 16 waves owns 16 output columns, streams its
// weight rows straight from global memory into MFMA fragments (the whole layer in flight at once, the next
// layer's during the epilogue) and writes the fp32 pre-activation / activation rows the backward needs
// plus the bf16 copy that is the next layer's LDS input.  Four waves per SIMD: with one (4 waves x 64
// columns) a wave's LDS reads, fragment waits and epilogue added up with its MFMAs instead of hiding
// behind another wave's - the 8 launches of these kernels on the step's dependent chain took 30 % longer.
#include "mlp_fused.h"

#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BMF = 32;   // rows per workgroup (these kernels are latency-bound: the Q MLP's 3328 rows give 208
                          // workgroups with 32 rows, 104 with 64; measured with 16 waves: 16 rows 1.09-1.10,
                          // 32 rows 1.07, 64 rows 1.08-1.12 ms/step)
constexpr int MTF = BMF / 16;
constexpr int XP = 264;   // LDS row pitch (bf16): 528 B -> 16 consecutive rows hit distinct bank groups
constexpr int MAXD = 256; // widest layer
// 16 waves per workgroup, 16 output columns each (measured at the bench shapes, ms/step: 4 waves 1.183,
// 8 waves 1.122, 16 waves 1.106)
constexpr int MF_NT = 1024, MF_CW = MAXD / (MF_NT / 64), MF_NTW = MF_CW / 16;

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
  if (K & 7) {  // rows not 8-byte aligned (a first layer of width 64 + 7): element loads, masked at the row end
    const __bf16 z0 = (__bf16)0.f;
    bf16x8 v = {z0, z0, z0, z0, z0, z0, z0, z0};
    if (n < N) {
      const __bf16* q = W + (long)n * K;
#pragma unroll
      for (int e = 0; e < 8; e++)
        if (k + e < K) v[e] = q[k + e];
    }
    return v;
  }
  const bool on = n < N && k < K;  // K % 8 == 0: a fragment is inside the row or entirely outside
  const __bf16* q = W + (long)(on ? n : 0) * K + (on ? k : 0);
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(q), hi = *reinterpret_cast<const bf16x4*>(q + 4);
  const __bf16 z = (__bf16)0.f;
  return on ? bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]} : bf16x8{z, z, z, z, z, z, z, z};
}

// Rows per workgroup BMF_: 32 for the step's small batches (latency: more workgroups), 128 for tens of thousands of rows
// (C5's Q networks, 99 k rows each: a workgroup re-streams every layer's weights from L2, 330 KB per 32 rows - 2 GB per
// launch - so four times the rows per weight load; 135 KB of LDS, one workgroup per CU).
template <int BMF_>
__device__ __forceinline__ void mlp_fused_fwd_body(const MlpFwdArgs& a) {
  constexpr int BMF = BMF_, MTF = BMF_ / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* X = reinterpret_cast<__bf16*>(smem);  // [2][BMF][XP]
  const int p = blockIdx.y, m0 = blockIdx.x * BMF, M = a.M[p];
  if (m0 >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
  const int n0 = MF_CW * w;
  bf16x8 B[8][MF_NTW];
  auto load_layer = [&](int l) {  // every weight fragment of the layer in flight at once (K <= 256)
    const int K = a.dims[l], N = a.dims[l + 1];
    const __bf16* Wb = a.pbf[p] + a.woff[l];
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int nt = 0; nt < MF_NTW; nt++) B[ks][nt] = load_w(Wb, K, N, n0 + 16 * nt + i, 32 * ks + 8 * g);
  };
  if (n0 < a.dims[1]) load_layer(0);  // travels while the input rows are staged
  // zero both buffers once: padded K columns are multiplied by zero weights and must stay finite
  for (int e = tid; e < 2 * BMF * XP / 8; e += MF_NT) reinterpret_cast<f32x4*>(X)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  {  // stage the input rows as bf16 (chunks of 8 columns; a ragged last chunk - K0 % 8 != 0 - element by element)
    const int K0 = a.dims[0], c8 = (K0 + 7) / 8;
    const float* x = a.x[p];
    for (int c = tid; c < BMF * c8; c += MF_NT) {
      const int row = c / c8, k = (c - row * c8) * 8;
      if (m0 + row < M) {
        const float* q = x + (long)(m0 + row) * a.ldx + k;
        bf16x8 v;
        if (k + 8 <= K0) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(q), hi = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
          for (int j = 0; j < 4; j++) { v[j] = (__bf16)lo[j]; v[4 + j] = (__bf16)hi[j]; }
        } else {
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = (__bf16)(k + j < K0 ? q[j] : 0.f);
        }
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
    f32x4 acc[MTF][MF_NTW], bvv[MF_NTW];
    const bool vec = (N & 3) == 0;
    if (n0 < N) {  // wave-uniform: this wave owns output columns [n0, n0 + 64)
#pragma unroll
      for (int nt = 0; nt < MF_NTW; nt++) {  // bias vectors travel under the MFMA loop (clamped address, masked below)
        const int col = n0 + 16 * nt + 4 * g;
        bvv[nt] = *reinterpret_cast<const f32x4*>(bias + (vec && col < N ? col : 0));
      }
#pragma unroll
      for (int mt = 0; mt < MTF; mt++)
#pragma unroll
        for (int nt = 0; nt < MF_NTW; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        if (ks >= KS) break;
        bf16x8 A[MTF];
#pragma unroll
        for (int mt = 0; mt < MTF; mt++) A[mt] = *reinterpret_cast<const bf16x8*>(xin + (16 * mt + i) * XP + 32 * ks + 8 * g);
#pragma unroll
        for (int nt = 0; nt < MF_NTW; nt++)
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
      float* yb = a.yoff[p][l] >= 0 ? a.act[p] + a.yoff[p][l] : nullptr;  // lean mode: a hidden SiLU layer's output is not saved
#pragma unroll
      for (int nt = 0; nt < MF_NTW; nt++) {
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
              if (yb) *reinterpret_cast<f32x4*>(yb + o) = y;
            } else {
#pragma unroll
              for (int r = 0; r < 4; r++)
                if (col + r < N) { if (zb) zb[o + r] = z[r]; if (yb) yb[o + r] = y[r]; }
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

__global__ __launch_bounds__(MF_NT) __attribute__((amdgpu_waves_per_eu(5, 8))) void mlp_fused_fwd_kernel(MlpFwdArgs a) {
  mlp_fused_fwd_body<32>(a);
}
__global__ __launch_bounds__(MF_NT) void mlp_fused_fwd_big_kernel(MlpFwdArgs a) { mlp_fused_fwd_body<128>(a); }

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

// (rows per workgroup as in the forward: 32, or 64 for >= 16 k rows - 128 would need 128 accumulator + source registers)
template <int BMF_>
__device__ __forceinline__ void mlp_fused_bwd_body(const MlpBwdArgs& a) {
  constexpr int BMF = BMF_, MTF = BMF_ / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* X = reinterpret_cast<__bf16*>(smem);  // [2][BMF][XP]
  const int p = blockIdx.y, m0 = blockIdx.x * BMF, M = a.M[p];
  if (m0 >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
  const int n0 = MF_CW * w;  // this wave's output columns (input features of the layer)
  bf16x8 B[8][MF_NTW];
  auto load_layer = [&](int l) {
    const int KO = a.dims[l], NP = (a.dims[l + 1] + 7) / 8 * 8;  // outputs, (padded) reduction length
    const __bf16* T = a.wt[p] + a.wtoff[l];
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int nt = 0; nt < MF_NTW; nt++) B[ks][nt] = load_w(T, NP, KO, n0 + 16 * nt + i, 32 * ks + 8 * g);
  };
  const int l_last = a.d_x[p] ? 0 : 1;  // layer 0's dgrad only if the input gradient is wanted
  if (a.L - 1 >= l_last && n0 < a.dims[a.L - 1]) load_layer(a.L - 1);  // travels while d_out is staged
  for (int e = tid; e < 2 * BMF * XP / 8; e += MF_NT) reinterpret_cast<f32x4*>(X)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  {  // stage dZ of the last layer (= d_out rows) as bf16
    const int NL = a.dims[a.L];
    const float* d = a.d_out[p];
    for (int c = tid; c < BMF * NL; c += MF_NT) {
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
    f32x4 acc[MTF][MF_NTW], svv[MTF][MF_NTW];
    const int pact = l > 0 ? a.acts[l - 1] : ACT_NONE;
    const float* src = (l > 0 && a.srcoff[p][l - 1] >= 0) ? a.act[p] + a.srcoff[p][l - 1] : nullptr;
    const int ldout = l > 0 ? KO : a.ldd;
    const bool vec = (KO & 3) == 0 && (ldout & 3) == 0;
    if (n0 < KO) {
      if (src && vec) {  // the activation-derivative sources travel under the MFMA loop (clamped, masked below)
#pragma unroll
        for (int nt = 0; nt < MF_NTW; nt++)
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
        for (int nt = 0; nt < MF_NTW; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        if (ks >= KS) break;
        bf16x8 A[MTF];
#pragma unroll
        for (int mt = 0; mt < MTF; mt++) A[mt] = *reinterpret_cast<const bf16x8*>(xin + (16 * mt + i) * XP + 32 * ks + 8 * g);
#pragma unroll
        for (int nt = 0; nt < MF_NTW; nt++)
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
      for (int nt = 0; nt < MF_NTW; nt++) {
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
__global__ __launch_bounds__(MF_NT) void mlp_fused_bwd_kernel(MlpBwdArgs a) { mlp_fused_bwd_body<32>(a); }
__global__ __launch_bounds__(MF_NT) void mlp_fused_bwd_big_kernel(MlpBwdArgs a) { mlp_fused_bwd_body<64>(a); }

// ---------------------------------------------------------------------------------- weight gradients
// dW_l = dZ_l^T X_l and db_l = colsum(dZ_l) of EVERY layer and network of an MLP site in one launch (+ one reduce
// launch).  As per-layer split-K GEMM launches this was 2 launches per layer, each a latency-bound chain of
// transposed-operand K tiles (a 256 x 256 x 3328 gradient took ~30 us at 2 % of a CU's MFMA rate) - 370 us of
// kernel time per step that sat beside the encoder backward and the action decoder.
// Here a workgroup owns a slice of WG_MS batch rows of one (network, layer) and the WHOLE dW (<= 256 x 256):
// 8 waves x (2 N tiles x 16 K tiles) of 16x16 fp32 accumulators, so every operand fragment read from LDS feeds
// 2 (X) or 16 (dZ) MFMAs; two waves per SIMD, so one wave's LDS traffic and global-load waits hide behind the
// other's MFMAs (with one wave per SIMD they add up: 75 us per launch against 40).  Both operands have the reduction index (the batch row)
// as their slow index in memory: the staging loads 4 rows x 4 columns per lane quad, transposes in registers
// (DPP quad_perm) and writes 4 consecutive rows of one column as bf16 (one 8-byte LDS store), so the MFMA
// fragments are contiguous 16-byte LDS reads.  X is the MFMA A operand (D rows = K index: a lane ends with 4
// consecutive k of one n -> 16-byte slab stores); the bias gradient is one more K tile with an all-ones
// fragment.  Slices are summed by mlp_wgrad_reduce_kernel in slice order (deterministic).
constexpr int WG_MS = 2048;   // most batch rows per workgroup (a.ms: chosen per call, a multiple of WG_MT; 256 at the bench
                              // shapes - the cap only matters for 100 k-row problems, where it keeps the slab count down)
constexpr int WG_MT = 32;     // rows per staged tile (one MFMA k-step)
constexpr int WG_TP = 40;     // bf16 pitch of a transposed row [col][32 m] (80 B: 16 cols hit distinct 16-B slots)
constexpr int WG_OPB = 256 * WG_TP * 2;  // bytes of one staged operand tile

struct MlpWgArgs {
  const float* x[MF_MAXP];      // layer-0 input [M][ldx]
  const float* act[MF_MAXP];    // saved activations (layer l >= 1 input = y_{l-1} at yoff[p][l-1])
  const float* dlast[MF_MAXP];  // gradient of the MLP output [M][ldo] (= dZ of the last layer)
  const float* dz[MF_MAXP];     // dZ_l, l < L-1, at dzoff[p][l] (from the dgrad launch)
  float* slab[MF_MAXP];         // partial gradients: [slice][record]
  int M[MF_MAXP];
  long yoff[MF_MAXP][MF_MAXL], dzoff[MF_MAXP][MF_MAXL];  // yoff < 0: -(zoff + 1) - the layer's output was not saved (lean
                                                          // mode): read its pre-activation there and apply acts[l]
  int acts[MF_MAXL];
  long sloff[MF_MAXL];          // offset of layer l inside a record: [N][K] then [N]
  long rec;                     // floats per record
  int dims[MF_MAXL + 1];
  int L, ldx, ldo, ms;
};

__device__ __forceinline__ float wg_dpp1(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float wg_dpp2(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
}
// 4x4 transpose across a lane quad (lane q enters with row q, leaves with column q)
__device__ __forceinline__ void wg_quad_transpose(f32x4& v, int q) {
  const bool o1 = q & 1, o2 = q & 2;
  const float x0 = wg_dpp1(o1 ? v[0] : v[1]), x1 = wg_dpp1(o1 ? v[2] : v[3]);
  if (o1) { v[0] = x0; v[2] = x1; } else { v[1] = x0; v[3] = x1; }
  const float y0 = wg_dpp2(o2 ? v[0] : v[2]), y1 = wg_dpp2(o2 ? v[1] : v[3]);
  if (o2) { v[0] = y0; v[1] = y1; } else { v[2] = y0; v[3] = y1; }
}

__global__ __launch_bounds__(512) void mlp_wgrad_fused_kernel(MlpWgArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [2 buffers][X^T | dZ^T] tiles
  const int p = blockIdx.z, l = blockIdx.y, m0 = blockIdx.x * a.ms, M = a.M[p];
  if (m0 >= M) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4, q = tid & 3;
  const int qd = (tid >> 2) & 63, jh = tid >> 8;  // column quad, row-group parity of this thread in the staging
  const int K = a.dims[l], N = a.dims[l + 1];
  const long yo = l == 0 ? 0 : a.yoff[p][l - 1];
  const float* X = l == 0 ? a.x[p] : a.act[p] + (yo < 0 ? -(yo + 1) : yo);
  const int actX = l > 0 && yo < 0 ? a.acts[l - 1] : ACT_NONE;  // recompute y = act(z) while staging (bit-identical to the saved y)
  const int ldX = l == 0 ? a.ldx : K;
  const float* Z = l == a.L - 1 ? a.dlast[p] : a.dz[p] + a.dzoff[p][l];
  const int ldZ = l == a.L - 1 ? a.ldo : N;
  const bool vecX = (ldX & 3) == 0 && (K & 3) == 0 && ((uintptr_t)X & 15) == 0;
  const bool vecZ = (ldZ & 3) == 0 && (N & 3) == 0 && ((uintptr_t)Z & 15) == 0;
  const int KT = (K + 15) >> 4, NT = (N + 15) >> 4;
  const int mend = min(M, m0 + a.ms), nsteps = (mend - m0 + WG_MT - 1) / WG_MT;

  // staging: quad qd owns columns 4 qd .. 4 qd + 3, lane q of it row 4 (2 j + jh) + q of the tile (j = 0..3)
  f32x4 rx[4], rz[4];
  auto fetch = [&](const float* S, int ld, int C, bool vec, int mt, f32x4 (&r)[4]) {
    const int c = 4 * qd;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int m = m0 + mt * WG_MT + 4 * (2 * j + jh) + q;
      const bool rok = m < mend;
      const float* src = S + (long)(rok ? m : m0) * ld;
      if (vec) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (c < C ? c : 0));
        const bool ok = rok && c < C;
        r[j] = f32x4{ok ? v[0] : 0.f, ok ? v[1] : 0.f, ok ? v[2] : 0.f, ok ? v[3] : 0.f};
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float v = src[c + e < C ? c + e : 0];
          r[j][e] = rok && c + e < C ? v : 0.f;
        }
      }
    }
  };
  auto stash = [&](f32x4 (&r)[4], unsigned char* T, int act) {  // T: [col][WG_TP] bf16
#pragma unroll
    for (int j = 0; j < 4; j++) {
      f32x4 v = r[j];
      if (act != ACT_NONE) v = f32x4{act_fast(act, v[0]), act_fast(act, v[1]), act_fast(act, v[2]), act_fast(act, v[3])};
      wg_quad_transpose(v, q);  // now: column 4 qd + q, rows 4 (2 j + jh) .. + 3
      *reinterpret_cast<bf16x4*>(T + ((4 * qd + q) * WG_TP + 4 * (2 * j + jh)) * 2) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
  };
  f32x4 acc[2][17];
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int kt = 0; kt < 17; kt++) acc[nt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const __bf16 one = (__bf16)1.0f;
  const bf16x8 ones = {one, one, one, one, one, one, one, one};

  fetch(X, ldX, K, vecX, 0, rx);
  fetch(Z, ldZ, N, vecZ, 0, rz);
  for (int s = 0; s < nsteps; s++) {
    unsigned char* TX = smem + (s & 1) * 2 * WG_OPB;
    unsigned char* TZ = TX + WG_OPB;
    stash(rx, TX, actX);
    stash(rz, TZ, ACT_NONE);
    __syncthreads();  // (the other buffer is still being read by slower waves: two buffers, one barrier per step)
    if (s + 1 < nsteps) { fetch(X, ldX, K, vecX, s + 1, rx); fetch(Z, ldZ, N, vecZ, s + 1, rz); }
    // this wave's N tiles: w, w + 8
    bf16x8 zf[2];
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
      zf[nt] = *reinterpret_cast<const bf16x8*>(TZ + ((16 * (w + 8 * nt) + i) * WG_TP + 8 * g) * 2);
#pragma unroll
    for (int kt = 0; kt < 16; kt++) {
      if (kt < KT) {  // (wave-uniform; a break would leave acc[][] dynamically indexed, i.e. in scratch)
        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(TX + ((16 * kt + i) * WG_TP + 8 * g) * 2);
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
          if (w + 8 * nt < NT) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf, zf[nt], acc[nt][kt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++)  // bias gradient: all-ones X fragment
      if (w + 8 * nt < NT) acc[nt][16] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, zf[nt], acc[nt][16], 0, 0, 0);
  }
  // partial dW [N][K] (this lane: n = 16 tile + i, k = 16 kt + 4 g .. + 3) and db [N]
  float* rec = a.slab[p] + (long)blockIdx.x * a.rec + a.sloff[l];
  const bool vecK = (K & 3) == 0;
#pragma unroll
  for (int nt = 0; nt < 2; nt++) {
    const int n = 16 * (w + 8 * nt) + i;
    if (w + 8 * nt >= NT || n >= N) continue;
#pragma unroll
    for (int kt = 0; kt < 16; kt++) {
      const int k = 16 * kt + 4 * g;
      if (kt < KT) {
        if (vecK) {
          if (k < K) *reinterpret_cast<f32x4*>(rec + (long)n * K + k) = acc[nt][kt];
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (k + r < K) rec[(long)n * K + k + r] = acc[nt][kt][r];
        }
      }
    }
    if (g == 0) rec[(long)N * K + n] = acc[nt][16][0];
  }
}

struct MlpWgReduceArgs {
  const float* slab[MF_MAXP];
  float* grad[MF_MAXP];
  int nslice[MF_MAXP];
  long sloff[MF_MAXL], woff[MF_MAXL], boff[MF_MAXL];
  long rec;
  int dims[MF_MAXL + 1];
  int accumulate;
};
// grid (blocks, L, nprob): grad[wo_l + e] (+)= sum over slices, in slice order
__global__ __launch_bounds__(256) void mlp_wgrad_reduce_kernel(MlpWgReduceArgs a) {
  const int l = blockIdx.y, p = blockIdx.z, K = a.dims[l], N = a.dims[l + 1], ns = a.nslice[p];
  const long nw = (long)N * K, tot = nw + N;
  const float* __restrict__ s0 = a.slab[p] + a.sloff[l];
  float* __restrict__ gr = a.grad[p];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gridDim.x * 256) {
    float t = 0.f;
#pragma unroll 8
    for (int s = 0; s < ns; s++) t += s0[(long)s * a.rec + e];  // slice order: deterministic (loads batch, adds stay ordered)
    float* d = gr + (e < nw ? a.woff[l] + e : a.boff[l] + (e - nw));
    *d = a.accumulate ? *d + t : t;
  }
}

}  // namespace

bool mlp_fused_wgrad_ok(int nprob, int L, const int* dims) {
  if (nprob < 1 || nprob > MF_MAXP || L < 1 || L > MF_MAXL) return false;
  for (int l = 0; l <= L; l++)
    if (dims[l] < 1 || dims[l] > MAXD) return false;
  return true;
}
static long mlp_wgrad_record(int L, const int* dims, long* sloff) {
  long off = 0;
  for (int l = 0; l < L; l++) {
    if (sloff) sloff[l] = off;
    off += (long)dims[l + 1] * dims[l] + dims[l + 1];
    off = (off + 3) & ~3L;
  }
  return off;
}
// Rows per slice: enough slices that the launch has ~128+ workgroups (a 256-row problem as ONE 8-step workgroup
// per layer takes as long as a 3328-row one), at most WG_MS rows, at least one staged tile.
static int mlp_wgrad_rows(int nprob, const int* M, int L) {
  long rows = 0;
  for (int p = 0; p < nprob; p++) rows += M[p] > 0 ? M[p] : 0;
  long ms = rows * L / 128;
  ms = (ms + WG_MT - 1) / WG_MT * WG_MT;
  return (int)(ms < WG_MT ? WG_MT : ms > WG_MS ? WG_MS : ms);
}
size_t mlp_fused_wgrad_slab_floats(int nprob, const int* M, int L, const int* dims) {
  const long rec = mlp_wgrad_record(L, dims, nullptr);
  const int ms = mlp_wgrad_rows(nprob, M, L);
  size_t tot = 0;
  for (int p = 0; p < nprob; p++) tot += (size_t)((M[p] + ms - 1) / ms) * rec;
  return tot;
}
int mlp_fused_wgrad(int nprob, const float* const* x, int ldx, const float* const* act, const float* const* d_out, int ldo,
                    const float* const* dz, float* const* grads, float* slab, const int* M, int L, const int* dims,
                    const long* yoff, const long* dzoff, const long* woff, const long* boff, int accumulate, hipStream_t st,
                    const int* acts) {
  MlpWgArgs a{};
  MlpWgReduceArgs r{};
  a.rec = r.rec = mlp_wgrad_record(L, dims, a.sloff);
  a.L = L; a.ldx = ldx; a.ldo = ldo; r.accumulate = accumulate;
  {
    int Mg[MF_MAXP], n = 0;  // only the problems that take part (same rule as the slab size query when all do)
    for (int p = 0; p < nprob; p++) Mg[n++] = M[p];
    a.ms = mlp_wgrad_rows(n, Mg, L);
  }
  for (int l = 0; l <= L; l++) a.dims[l] = r.dims[l] = dims[l];
  for (int l = 0; l < L; l++) { r.sloff[l] = a.sloff[l]; r.woff[l] = woff[l]; r.boff[l] = boff[l]; a.acts[l] = acts ? acts[l] : ACT_NONE; }
  int n2 = 0, maxs = 0;
  float* sp = slab;
  for (int p = 0; p < nprob; p++) {
    if (!grads[p] || M[p] <= 0) continue;
    const int ns = (M[p] + a.ms - 1) / a.ms;
    a.x[n2] = x[p]; a.act[n2] = act[p]; a.dlast[n2] = d_out[p]; a.dz[n2] = dz[p]; a.M[n2] = M[p]; a.slab[n2] = sp;
    r.slab[n2] = sp; r.grad[n2] = grads[p]; r.nslice[n2] = ns;
    for (int l = 0; l < L; l++) { a.yoff[n2][l] = yoff[p * MF_MAXL + l]; a.dzoff[n2][l] = dzoff[p * MF_MAXL + l]; }
    sp += (size_t)ns * a.rec;
    maxs = ns > maxs ? ns : maxs;
    n2++;
  }
  if (!n2) return 0;
  static int once = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_wgrad_fused_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WG_OPB) == hipSuccess ? 0 : -1;
  if (once) return -1;
  hipLaunchKernelGGL(mlp_wgrad_fused_kernel, dim3(maxs, L, n2), dim3(512), 4 * WG_OPB, st, a);
  hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3(256, L, n2), dim3(256), 0, st, r);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

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
  constexpr size_t lds = (size_t)2 * BMF * XP * 2, lds_big = (size_t)2 * 64 * XP * 2;
  static int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_bwd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_bwd_big_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big) == hipSuccess) ? 0 : -1;
  if (once) return TACORL_ELAUNCH;
  static const int big_rows = [] { const char* e = getenv("TACORL_MLP_BIG_ROWS"); return e ? atoi(e) : 16384; }();
  if (maxM >= big_rows)
    hipLaunchKernelGGL(mlp_fused_bwd_big_kernel, dim3((maxM + 63) / 64, nprob), dim3(MF_NT), lds_big, st, a);
  else
    hipLaunchKernelGGL(mlp_fused_bwd_kernel, dim3((maxM + BMF - 1) / BMF, nprob), dim3(MF_NT), lds, st, a);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

bool mlp_fused_fwd_ok(int nprob, int L, const int* dims, int ldx) {
  if (nprob < 1 || nprob > MF_MAXP || L < 1 || L > MF_MAXL || ldx % 4) return false;
  for (int l = 0; l < L; l++)  // hidden widths: multiples of 8; the input width: any (a ragged row end is handled)
    if ((l > 0 && dims[l] % 8) || dims[l] > MAXD || dims[l] < 8) return false;
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
  constexpr size_t lds = (size_t)2 * BMF * XP * 2, lds_big = (size_t)2 * 128 * XP * 2;
  static int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_fwd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_fwd_big_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big) == hipSuccess) ? 0 : -1;
  if (once) return TACORL_ELAUNCH;
  static const int big_rows = [] { const char* e = getenv("TACORL_MLP_BIG_ROWS"); return e ? atoi(e) : 16384; }();
  if (maxM >= big_rows)
    hipLaunchKernelGGL(mlp_fused_fwd_big_kernel, dim3((maxM + 127) / 128, nprob), dim3(MF_NT), lds_big, st, a);
  else
    hipLaunchKernelGGL(mlp_fused_fwd_kernel, dim3((maxM + BMF - 1) / BMF, nprob), dim3(MF_NT), lds, st, a);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
