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
constexpr int XP = 272;   // LDS row pitch (bf16): 544 B = 34 x 16 B, 34 = 2 (mod 4) and even: the ds_read_b128 lane groups {0-3,12-15,20-27}, .. of a
                          // 16-row fragment read land on 16 distinct 16-byte slots (264 - 33 slots per row - was read as conflict-free for two rounds;
                          // PMC on the many-row forward: 45 % of the LDS-active cycles were bank conflicts)
constexpr int MAXD = 256; // widest layer
// 16 waves per workgroup, 16 output columns each (measured at the bench shapes, ms/step: 4 waves 1.183,
// 8 waves 1.122, 16 waves 1.106)
constexpr int MF_NT = 1024, MF_CW = MAXD / (MF_NT / 64), MF_NTW = MF_CW / 16;

// Gathered layer-0 input: columns [c0, next segment's c0) of row r come from p[(mod ? r % mod : r) * ld + (col - c0)].
// The concatenations the reference builds in front of its MLPs ([enc(obs) | goal_enc(..)], [state | action] with the state
// rows repeated for the n sampled actions - expand_obs, utils/misc.py:132-153) then cost no launch of their own: the
// forward reads its input where the producers left it and, if asked (xw), also writes the assembled fp32 row - the
// weight-gradient launch's layer-0 operand - from the registers it already holds.
struct MlpXSeg { const float* p; int ld, c0, mod, pad; };
struct MlpFwdArgs {
  MlpXSeg seg[MF_MAXP][MF_MAXSEG];
  int nseg[MF_MAXP];            // 0: plain x[p] / ldx
  float* xw[MF_MAXP];           // optional assembled copy [M][ldx]
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
// (v_exp_f32 / v_rcp_f32 directly: `__frcp_rn` is the correctly rounded reciprocal, i.e. hipcc's 11-instruction IEEE
// division sequence - per element, in every epilogue of these kernels)
__device__ __forceinline__ float sigmoid_fast(float z) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * z));
}
__device__ __forceinline__ float act_fast(int act, float z) {
  if (act == ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == ACT_SILU) return z * sigmoid_fast(z);
  return z;
}
__device__ __forceinline__ float act_grad_fast(int act, float zy) {
  if (act == ACT_RELU) return zy > 0.f ? 1.f : 0.f;
  if (act == ACT_SILU) {
    const float sg = sigmoid_fast(zy);
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
  if (a.nseg[p] > 0) {  // gathered input (segment starts are multiples of 8, lds multiples of 4, bases 16-byte aligned)
    const int K0 = a.dims[0], c8 = (K0 + 7) / 8, ns = a.nseg[p];
    float* xw = a.xw[p];
    for (int c = tid; c < BMF * c8; c += MF_NT) {
      const int row = c / c8, k = (c - row * c8) * 8;
      if (m0 + row < M) {
        int sg = 0;
#pragma unroll
        for (int t = 1; t < MF_MAXSEG; t++) sg = (t < ns && k >= a.seg[p][t].c0) ? t : sg;
        const MlpXSeg s = a.seg[p][sg];
        const int cend = sg + 1 < ns ? a.seg[p][sg + 1].c0 : K0;  // the segment's columns end here
        const int r = s.mod ? (m0 + row) % s.mod : m0 + row;
        const float* q = s.p + (long)r * s.ld + (k - s.c0);
        f32x4 lo, hi;
        if (k + 8 <= cend) { lo = *reinterpret_cast<const f32x4*>(q); hi = *reinterpret_cast<const f32x4*>(q + 4); }
        else {
#pragma unroll
          for (int j = 0; j < 4; j++) { lo[j] = k + j < cend ? q[j] : 0.f; hi[j] = k + 4 + j < cend ? q[4 + j] : 0.f; }
        }
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 4; j++) { v[j] = (__bf16)lo[j]; v[4 + j] = (__bf16)hi[j]; }
        *reinterpret_cast<bf16x8*>(X + row * XP + k) = v;
        if (xw) {
          float* o = xw + (long)(m0 + row) * a.ldx + k;
          if (k + 8 <= a.ldx) { *reinterpret_cast<f32x4*>(o) = lo; *reinterpret_cast<f32x4*>(o + 4) = hi; }
          else {
#pragma unroll
            for (int j = 0; j < 4; j++) { if (k + j < a.ldx) o[j] = lo[j]; if (k + 4 + j < a.ldx) o[4 + j] = hi[j]; }
          }
        }
      }
    }
  } else {  // stage the input rows as bf16 (chunks of 8 columns; a ragged last chunk - K0 % 8 != 0 - element by element)
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
// Wt[kin][n] = bf16(W[n][kin]), n < NP = roundup(N, 8) (zeros for n >= N), as 32 x 32 tiles through LDS: rows of W are read
// along kin and rows of Wt written along n, both coalesced (round 5; element by element every read was a 64-byte sector of
// its own - ~70 MB of L2 traffic for the 1.1 M weights of the headline step's five sites, 15 us for their one launch)
__device__ __forceinline__ void pack_wt_tiles(const float* __restrict__ W, __bf16* __restrict__ T, int K, int N, int NP, int b0,
                                              int nb) {
  __shared__ float tile[32][33];
  const int tk = (K + 31) / 32, tn = (NP + 31) / 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int t = b0; t < tk * tn; t += nb) {
    const int k0 = (t / tn) * 32, n0 = (t - (t / tn) * tn) * 32;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n = n0 + ty + 8 * j, k = k0 + tx;
      tile[ty + 8 * j][tx] = (n < N && k < K) ? W[(long)n * K + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int k = k0 + ty + 8 * j, n = n0 + tx;
      if (k < K && n < NP) T[(long)k * NP + n] = (__bf16)tile[tx][ty + 8 * j];
    }
    __syncthreads();
  }
}
// grid (blocks, L, nprob)
__global__ __launch_bounds__(256) void mlp_pack_wt_kernel(WtPackArgs a) {
  const int l = blockIdx.y, p = blockIdx.z, K = a.dims[l], N = a.dims[l + 1], NP = (N + 7) / 8 * 8;
  pack_wt_tiles(a.params[p] + a.woff[l], a.wt[p] + a.wtoff[l], K, N, NP, blockIdx.x, gridDim.x);
}

// ---- weight-only preparation of a step in ONE launch (round 5).  The headline step issued eight 5 - 7 us launches that read
// nothing but parameters - the bf16 mirrors of the five networks' MLP weights (tacorl_to_bf16_batch), the transposed
// weights of four MLP backward sites and of the encoders' FC tail (mlp_pack_wt_kernel, one launch per site) - back to back
// on the step's dependent chain: each a launch latency plus one L2 round trip for a few hundred KB.  Between
// tacorl_prep_batch_begin() and tacorl_prep_batch_end(stream) those entry points only RECORD their job (this thread's
// list); _end issues them as one launch (blockIdx.z = job: a (site, network) transpose or a conversion).
#define PREP_MAXS 6   // transpose sites per launch
#define PREP_MAXB 16  // conversion jobs per launch
struct PrepMultiArgs {
  WtPackArgs site[PREP_MAXS];
  int first[PREP_MAXS + 1];  // jobs [first[s], first[s + 1]) are the networks of site s; conversions follow
  int nsite, ncv;
  const float* csrc[PREP_MAXB];
  __bf16* cdst[PREP_MAXB];
  long cn4[PREP_MAXB];
};
__global__ __launch_bounds__(256) void prep_multi_kernel(PrepMultiArgs a) {
  const int z = blockIdx.z, nt = a.first[a.nsite];
  if (z >= nt) {  // fp32 -> bf16 copy: the (x, y) blocks of the job stride over it
    const int j = z - nt;
    const float* __restrict__ src = a.csrc[j];
    __bf16* __restrict__ dst = a.cdst[j];
    const long n4 = a.cn4[j], step = (long)gridDim.x * gridDim.y * 256;
    for (long q = ((long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; q < n4; q += step) {
      const f32x4 v = reinterpret_cast<const f32x4*>(src)[q];
      reinterpret_cast<bf16x4*>(dst)[q] = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
    return;
  }
  int s = 0;
#pragma unroll
  for (int t = 1; t < PREP_MAXS; t++) s = (t < a.nsite && z >= a.first[t]) ? t : s;
  const WtPackArgs& k = a.site[s];
  const int l = blockIdx.y, pp = z - a.first[s];
  if (l >= k.L) return;
  const int K = k.dims[l], N = k.dims[l + 1], NP = (N + 7) / 8 * 8;
  pack_wt_tiles(k.params[pp] + k.woff[l], k.wt[pp] + k.wtoff[l], K, N, NP, blockIdx.x, gridDim.x);
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

// Several sites' reduces as ONE launch (round 5: the headline step's five reduce launches - Q hidden layers, Q output
// layer, policy head, goal encoders, encoder FC tails; 5 - 8 us each, their only reader is the optimiser - are recorded
// between tacorl_reduce_batch_begin() and _end(stream) and leave together; host side below the kernels).
#define RED_MAXJ 8
struct MlpWgReduceMulti {
  MlpWgReduceArgs job[RED_MAXJ];
  int gx[RED_MAXJ], L[RED_MAXJ], first[RED_MAXJ + 1];  // blocks along x / layers / first blockIdx.z of job j
  int njob;
};
__global__ __launch_bounds__(256) void mlp_wgrad_reduce_multi_kernel(MlpWgReduceMulti m) {
  int j = 0;
#pragma unroll
  for (int t = 1; t < RED_MAXJ; t++) j = (t < m.njob && (int)blockIdx.z >= m.first[t]) ? t : j;
  const int gx = m.gx[j];
  if ((int)blockIdx.y >= m.L[j] || (int)blockIdx.x >= gx) return;
  const MlpWgReduceArgs& a = m.job[j];
  const int l = blockIdx.y, p = blockIdx.z - m.first[j], K = a.dims[l], N = a.dims[l + 1], ns = a.nslice[p];
  const long nw = (long)N * K, tot = nw + N;
  const float* __restrict__ s0 = a.slab[p] + a.sloff[l];
  float* __restrict__ gr = a.grad[p];
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < tot; e += (long)gx * 256) {
    float t = 0.f;
#pragma unroll 8
    for (int s = 0; s < ns; s++) t += s0[(long)s * a.rec + e];  // slice order, as mlp_wgrad_reduce_kernel
    float* d = gr + (e < nw ? a.woff[l] + e : a.boff[l] + (e - nw));
    *d = a.accumulate ? *d + t : t;
  }
}
struct RedState { bool on = false; MlpWgReduceMulti m{}; };
thread_local RedState g_red;
int red_flush(hipStream_t st) {
  MlpWgReduceMulti& m = g_red.m;
  if (m.njob == 0) return 0;
  int gx = 1, Ly = 1;
  for (int j = 0; j < m.njob; j++) { gx = m.gx[j] > gx ? m.gx[j] : gx; Ly = m.L[j] > Ly ? m.L[j] : Ly; }
  hipLaunchKernelGGL(mlp_wgrad_reduce_multi_kernel, dim3(gx, Ly, m.first[m.njob]), dim3(256), 0, st, m);
  m.njob = 0; m.first[0] = 0;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// the reduce of one site: launched, or recorded while a batch is open
int launch_reduce(const MlpWgReduceArgs& r, int gx, int Ly, int nprob, hipStream_t st) {
  if (!g_red.on) {
    hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3(gx, Ly, nprob), dim3(256), 0, st, r);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  MlpWgReduceMulti& m = g_red.m;
  if (m.njob == RED_MAXJ) {
    // table full: this site's reduce leaves now, on ITS stream, behind its own producers.  (Flushing the recorded jobs
    // here instead would sum slabs whose weight-gradient launches sit on other streams that are joined only in front
    // of tacorl_reduce_batch_end - ADVICE r5.)
    hipLaunchKernelGGL(mlp_wgrad_reduce_kernel, dim3(gx, Ly, nprob), dim3(256), 0, st, r);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  m.job[m.njob] = r; m.gx[m.njob] = gx; m.L[m.njob] = Ly; m.first[m.njob + 1] = m.first[m.njob] + nprob;
  m.njob++;
  return 0;
}

// ------------------------------------------------------------------ forward / input gradients, tens of thousands of rows
// C5's Q networks: 99 328 rows each.  With the kernels above (128 / 64 rows per workgroup) the forward wrote every hidden
// pre-activation as fp32 straight from the accumulators - one 64-byte segment per (row, wave): 610 MB in 319 us - and the
// input-gradient chain read them back the same way and wrote every dZ_l as fp32 (428 us).  What the backward needs of a
// hidden SiLU layer is not z but act'(z) (the chain) and y = act(z) rounded to bf16 (the weight gradients' MFMA operand).
// The many-row kernels save exactly those: y as bf16 (the very values the next layer consumed) and s = act'(z) as fp16
// (|s| <= 1.1, 11 significant bits; it multiplies a gradient that is rounded to bf16 right after), both row-major
// [Mp][256] (Mp = M rounded up to 64, zero rows beyond M) - 4 bytes per element instead of 4 (+ 4 for dZ) - and both
// leave through LDS as 16-byte-per-lane coalesced stores.  Same structure as mlp_fused_{fwd,bwd}_body otherwise.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#ifdef MLP_STAMPS  // scratch builds (scratch/mklib2.sh): phase clocks of one workgroup's wave 0, read by tacorl_dbg_mlp_stamps
__device__ unsigned long long g_mlp_stamps[2][64];
#define MLP_STAMP(k, j) do { if (stamp_on) g_mlp_stamps[k][j] = clock64(); } while (0)
#else
#define MLP_STAMP(k, j) do { } while (0)
#endif

struct MlpBigFwdArgs {
  MlpXSeg seg[MF_MAXP][MF_MAXSEG];  // gathered layer-0 input (see MlpFwdArgs)
  int nseg[MF_MAXP];
  float* xw[MF_MAXP];
  const float* x[MF_MAXP];
  const float* params[MF_MAXP];
  const __bf16* pbf[MF_MAXP];
  float* act[MF_MAXP];
  int M[MF_MAXP];
  long ybf[MF_MAXP][MF_MAXL], sbf[MF_MAXP][MF_MAXL];  // float offsets in act[p]: bf16 output copy / fp16 act' copy of hidden layer l
  long yout[MF_MAXP];                                  // float offset of the last layer's fp32 output [M][dims[L]]
  long woff[MF_MAXL], boff[MF_MAXL];
  int dims[MF_MAXL + 1], acts[MF_MAXL];
  int L, ldx;
  int dbg;  // scratch experiments of the persistent kernel (TACORL_MLP_PERS_DBG; 0 in production): 1 = no copy-out stores
};

// BMF_ rows per workgroup: 128 for >= 16 384 rows (C5), 32 for the thousands of rows of the other configurations' Q
// networks (round 5: (3 n + 1) B = 3 328 at B = 256 - more workgroups, same saves, so that their weight gradients can
// take mlp_wgrad_big_kernel too)
template <int BMF_>
__device__ __forceinline__ void mlp_big_fwd_body(const MlpBigFwdArgs& a) {
  constexpr int BMF = BMF_, MTF = BMF / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* X = reinterpret_cast<__bf16*>(smem);  // [2][BMF][XP]
  const int p = blockIdx.y, m0 = blockIdx.x * BMF, M = a.M[p];
  const int Mp = (M + 63) & ~63;
  if (m0 >= Mp) return;  // (a block of padding rows only still writes their zeros)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
  const int n0 = MF_CW * w;
#ifdef MLP_STAMPS
  const bool stamp_on = blockIdx.x == 300 && p == 0 && tid == 0;
#endif
  MLP_STAMP(0, 0);
  // ONE global round trip in front of the first layer: the input rows (8-column chunks, up to two per thread) and the
  // first layer's weights are requested together, LDS is zeroed meanwhile; no barrier of the prologue drains vmcnt
  // (a workgroup starts behind the 393 KB its predecessor on this CU has just stored: every dependent round trip
  // up here costs ~5 us - the two-step prologue of mlp_fused_fwd_body read 11 us of a 52 us workgroup)
  const int K0 = a.dims[0], c8 = (K0 + 7) / 8;  // c8 <= 16 (dims[0] <= 128): BMF * c8 <= 2 * MF_NT
  f32x4 xr[2][2];
  const float* x = a.x[p];
  const int ns = a.nseg[p];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int c = tid + u * MF_NT, row = c / c8, k = (c - row * c8) * 8;
    const bool ok = c < BMF * c8 && m0 + row < M;
    if (ns > 0) {  // gathered input: the chunk's segment, its row (modulo) and the segment's end
      int sg = 0;
#pragma unroll
      for (int t = 1; t < MF_MAXSEG; t++) sg = (t < ns && k >= a.seg[p][t].c0) ? t : sg;
      const MlpXSeg s = a.seg[p][sg];
      const int cend = sg + 1 < ns ? a.seg[p][sg + 1].c0 : K0;
      const int r = s.mod ? (m0 + row) % s.mod : m0 + row;
      const float* q = s.p + (ok ? (long)r * s.ld + (k - s.c0) : 0);
      if (ok && k + 8 <= cend) { xr[u][0] = *reinterpret_cast<const f32x4*>(q); xr[u][1] = *reinterpret_cast<const f32x4*>(q + 4); }
      else {
#pragma unroll
        for (int j = 0; j < 4; j++) { xr[u][0][j] = ok && k + j < cend ? q[j] : 0.f; xr[u][1][j] = ok && k + 4 + j < cend ? q[4 + j] : 0.f; }
      }
      continue;
    }
    const float* q = x + (ok ? (long)(m0 + row) * a.ldx + k : 0);
    if (ok && k + 8 <= a.ldx) { xr[u][0] = *reinterpret_cast<const f32x4*>(q); xr[u][1] = *reinterpret_cast<const f32x4*>(q + 4); }
    else {
#pragma unroll
      for (int j = 0; j < 4; j++) { xr[u][0][j] = ok && k + j < K0 ? q[j] : 0.f; xr[u][1][j] = ok && k + 4 + j < K0 ? q[4 + j] : 0.f; }
    }
  }
  bf16x8 B[8][MF_NTW];
  auto load_layer = [&](int l) {
    const int K = a.dims[l], N = a.dims[l + 1];
    const __bf16* Wb = a.pbf[p] + a.woff[l];
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int nt = 0; nt < MF_NTW; nt++) B[ks][nt] = load_w(Wb, K, N, n0 + 16 * nt + i, 32 * ks + 8 * g);
  };
  if (n0 < a.dims[1]) load_layer(0);
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  for (int e = tid; e < 2 * BMF * XP / 8; e += MF_NT) reinterpret_cast<f32x4*>(X)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  lds_barrier();
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int c = tid + u * MF_NT, row = c / c8, k = (c - row * c8) * 8;
    if (c < BMF * c8 && m0 + row < M) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 4; j++) { v[j] = (__bf16)(k + j < K0 ? xr[u][0][j] : 0.f); v[4 + j] = (__bf16)(k + 4 + j < K0 ? xr[u][1][j] : 0.f); }
      *reinterpret_cast<bf16x8*>(X + row * XP + k) = v;
      if (ns > 0 && a.xw[p]) {  // the assembled fp32 rows (the gather masked them at the segment ends: pad columns are zeros)
        float* o = a.xw[p] + (long)(m0 + row) * a.ldx + k;
        if (k + 8 <= a.ldx) { *reinterpret_cast<f32x4*>(o) = xr[u][0]; *reinterpret_cast<f32x4*>(o + 4) = xr[u][1]; }
        else {
#pragma unroll
          for (int j = 0; j < 4; j++) { if (k + j < a.ldx) o[j] = xr[u][0][j]; if (k + 4 + j < a.ldx) o[4 + j] = xr[u][1][j]; }
        }
      }
    }
  }
  lds_barrier();
  int cur = 0;
  for (int l = 0; l < a.L; l++) {
    MLP_STAMP(0, 8 * l + 1);
    const int K = a.dims[l], N = a.dims[l + 1], KS = (K + 31) / 32, act = a.acts[l];
    const bool hidden = l + 1 < a.L;
    const float* bias = a.params[p] + a.boff[l];
    const __bf16* xin = X + cur * BMF * XP;
    __bf16* xout = X + (cur ^ 1) * BMF * XP;
    f32x4 acc[MTF][MF_NTW], bvv[MF_NTW];
    f16x4 sreg[MTF][MF_NTW];
    const bool vec = (N & 3) == 0;
    if (n0 < N) {
#pragma unroll
      for (int nt = 0; nt < MF_NTW; nt++) {
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
            for (int mt = 0; mt < MTF; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ks][nt], A[mt], acc[mt][nt], 0, 0, 0);
          }
      }
    }
    MLP_STAMP(0, 8 * l + 2);
    if (hidden && n0 < a.dims[l + 2]) load_layer(l + 1);
    // (the copy-out of the previous layer read both buffers: nobody may write xout before every wave is past it)
    lds_barrier();
    MLP_STAMP(0, 8 * l + 3);
    if (hidden) {  // SiLU over all 256 columns (mlp_big_prob_ok): y = z sg, act' = sg + y (1 - sg), one exp / rcp for both
#pragma unroll
      for (int nt = 0; nt < MF_NTW; nt++) {
        const int col = n0 + 16 * nt + 4 * g;
        const f32x4 bv = bvv[nt];
#pragma unroll
        for (int mt = 0; mt < MTF; mt++) {
          const int row = 16 * mt + i;
          const f32x4 z = acc[mt][nt] + bv;
          f32x4 y, sd;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float sg = sigmoid_fast(z[r]);
            y[r] = z[r] * sg;
            sd[r] = sg + y[r] * (1.f - sg);
          }
          *reinterpret_cast<bf16x4*>(xout + row * XP + col) = bf16x4{(__bf16)y[0], (__bf16)y[1], (__bf16)y[2], (__bf16)y[3]};
          sreg[mt][nt] = f16x4{(_Float16)sd[0], (_Float16)sd[1], (_Float16)sd[2], (_Float16)sd[3]};
        }
      }
    } else if (n0 < N) {  // the output layer (1..4 columns, no activation): fp32 rows
      float* yb = a.act[p] + a.yout[p];
#pragma unroll
      for (int nt = 0; nt < MF_NTW; nt++) {
        const int col = n0 + 16 * nt + 4 * g;
        if (n0 + 16 * nt >= N) continue;
#pragma unroll
        for (int mt = 0; mt < MTF; mt++) {
          const int row = 16 * mt + i;
          if (m0 + row < M) {
#pragma unroll
            for (int r = 0; r < 4; r++)
              if (col + r < N) yb[(long)(m0 + row) * N + col + r] = acc[mt][nt][r] + bias[col + r];
          }
        }
      }
    }
    MLP_STAMP(0, 8 * l + 4);
    lds_barrier();  // xout complete, xin dead
    MLP_STAMP(0, 8 * l + 5);
    if (hidden) {
      // act' through the dead input buffer, then both copies leave as 16-byte coalesced stores
      _Float16* sb = reinterpret_cast<_Float16*>(X + cur * BMF * XP);
      if (n0 < N) {
#pragma unroll
        for (int nt = 0; nt < MF_NTW; nt++) {
          const int col = n0 + 16 * nt + 4 * g;
          if (n0 + 16 * nt >= N) continue;
#pragma unroll
          for (int mt = 0; mt < MTF; mt++) *reinterpret_cast<f16x4*>(sb + (16 * mt + i) * XP + col) = sreg[mt][nt];
        }
      }
      lds_barrier();
      MLP_STAMP(0, 8 * l + 6);
      __bf16* y16 = reinterpret_cast<__bf16*>(a.act[p] + a.ybf[p][l]);
      __bf16* s16 = reinterpret_cast<__bf16*>(a.act[p] + a.sbf[p][l]);  // (fp16 payload moved as 16-byte vectors)
      const int c8n = N >> 3;
      for (int c = tid; c < BMF * c8n; c += MF_NT) {
        const int row = c / c8n, k = (c - row * c8n) * 8;
        if (m0 + row < Mp) {
          const __bf16 z0 = (__bf16)0.f;
          bf16x8 yv = {z0, z0, z0, z0, z0, z0, z0, z0}, sv = yv;
          if (m0 + row < M) {
            yv = *reinterpret_cast<const bf16x8*>(xout + row * XP + k);
            sv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(sb) + row * XP + k);
          }
          *reinterpret_cast<bf16x8*>(y16 + (long)(m0 + row) * N + k) = yv;
          *reinterpret_cast<bf16x8*>(s16 + (long)(m0 + row) * N + k) = sv;
        }
      }
    }
    MLP_STAMP(0, 8 * l + 7);
    cur ^= 1;
  }
}
__global__ __launch_bounds__(MF_NT) void mlp_big_fwd_kernel(MlpBigFwdArgs a) { mlp_big_fwd_body<128>(a); }
__global__ __launch_bounds__(MF_NT) __attribute__((amdgpu_waves_per_eu(5, 8))) void mlp_mid_fwd_kernel(MlpBigFwdArgs a) { mlp_big_fwd_body<32>(a); }
__global__ __launch_bounds__(MF_NT) __attribute__((amdgpu_waves_per_eu(5, 8))) void mlp_m64_fwd_kernel(MlpBigFwdArgs a) { mlp_big_fwd_body<64>(a); }

// ---- persistent many-row forward (round 5).  The kernels above re-stream every layer's weights through the CU's L2 port for
// each row block (300 KB per 128 rows, ~10 k clk at the ~29 B/clk a CU ingests), start every block with a prologue behind
// the predecessor's 393 KB of stores, and alternate load / MFMA / epilogue / store phases behind three 16-wave barriers per
// layer: 308 us for C5's 2 x 99 328 rows where the 610 MB of saves alone would take ~120 us (profiles/r04_pmc_sq_c5.md: 8 %
// MFMA-busy, 55 % of the wave cycles waiting).  Here ONE workgroup per CU stays resident and walks over 64-row blocks:
//  * the two 256 x 256 layers' weights live in REGISTERS for the whole launch (8 waves x 32 columns: 128 registers per
//    lane), layer 0's (<= 96 x 256) in LDS in fragment order, biases and the output layer in LDS: no weight traffic per block;
//  * the next block's input rows are requested at the top of a block and converted at its end (no prologue round trip);
//  * act' has its own staging buffer, so a layer is MFMA -> barrier -> epilogue -> barrier -> coalesced copy-out, and the
//    copy-out's stores drain under the next layer's MFMAs;
//  * the 1 .. 4-column output layer is a 32-element dot product per thread on the bf16 row in LDS (no MFMA tile with one
//    valid column).
// Same saves, same layouts, same arithmetic as mlp_big_fwd_body (hidden layers bit-identical; the output layer differs in
// its fp32 summation order only).
constexpr int PF_NT = 512, PF_BM = 64, PF_MT = PF_BM / 16, PF_K0S = 3;
constexpr int PF_XB = 2 * PF_BM * XP * 2, PF_SB = PF_BM * XP * 2, PF_W0B = 8 * PF_K0S * 2 * 1024, PF_MISC = 8192;
constexpr int PF_LDS = PF_XB + PF_SB + PF_W0B + PF_MISC;
static_assert(PF_LDS <= 160 * 1024, "persistent forward: LDS");
typedef unsigned int pf_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NH>  // 256 x 256 hidden layers behind layer 0 (L = NH + 2)
__global__ __launch_bounds__(PF_NT) void mlp_pers_fwd_kernel(MlpBigFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* X = reinterpret_cast<__bf16*>(smem);                         // [2][PF_BM][XP]
  _Float16* S = reinterpret_cast<_Float16*>(smem + PF_XB);             // [PF_BM][XP] act' of the layer being finished
  pf_u32x4* W0 = reinterpret_cast<pf_u32x4*>(smem + PF_XB + PF_SB);    // layer 0: [(wave * PF_K0S + ks) * 2 + nt][lane]
  float* BI = reinterpret_cast<float*>(smem + PF_XB + PF_SB + PF_W0B); // hidden biases [NH + 1][256]
  float* WO = BI + 3 * 256;                                            // output layer [NL <= 4][256], then its bias [4]
  constexpr int L = NH + 2;
#ifdef MLP_PERS_DBG  // scratch builds only (scratch/mklib_file.sh ... -DMLP_PERS_DBG, scratch/r5_mlp_dissect.sh): phases switched off at run time
  const int dbg = a.dbg;
#else
  constexpr int dbg = 0;
#endif
  const int p = blockIdx.y, M = a.M[p], Mp = (M + 63) & ~63, nblk = Mp / PF_BM;
  if ((int)blockIdx.x >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
  const int n0 = 32 * w;
  const int K0 = a.dims[0], KS0 = (K0 + 31) / 32, c8p = 4 * KS0, NL = a.dims[L];
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  // ---- launch-resident operands.  MFMA row r of the wave's tile nt is output column n0 + 8 (r >> 2) + 4 nt + (r & 3): a lane's
  // accumulators (rows 4 g .. 4 g + 3 of both tiles) are then the 8 ADJACENT columns n0 + 8 g .. + 7 - one 16-byte LDS store
  // per 16-row tile for y and one for act' instead of two 8-byte ones each (whose 16 lanes, 544 B apart, hit 8 of 32 banks)
  pf_u32x4 WH[NH][8][2];
#pragma unroll
  for (int h = 0; h < NH; h++) {
    const __bf16* Wb = a.pbf[p] + a.woff[h + 1];
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++) WH[h][ks][nt] = __builtin_bit_cast(pf_u32x4, load_w(Wb, 256, 256, n0 + 8 * (i >> 2) + 4 * nt + (i & 3), 32 * ks + 8 * g));
  }
  {
    const __bf16* Wb = a.pbf[p] + a.woff[0];
#pragma unroll
    for (int ks = 0; ks < PF_K0S; ks++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
        W0[((w * PF_K0S + ks) * 2 + nt) * 64 + lane] = __builtin_bit_cast(pf_u32x4, load_w(Wb, K0, 256, n0 + 8 * (i >> 2) + 4 * nt + (i & 3), 32 * ks + 8 * g));
  }
  for (int e = tid; e < (NH + 1) * 256; e += PF_NT) BI[e] = a.params[p][a.boff[e >> 8] + (e & 255)];
  for (int e = tid; e < NL * 256; e += PF_NT) WO[e] = (float)a.pbf[p][a.woff[L - 1] + e];
  if (tid < 4) WO[4 * 256 + tid] = tid < NL ? a.params[p][a.boff[L - 1] + tid] : 0.f;
  for (int e = tid; e < 2 * PF_BM * XP / 8; e += PF_NT) reinterpret_cast<f32x4*>(X)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (the weights are values of this asm from here on: nothing the compiler could re-load instead of keeping)
#pragma unroll
  for (int h = 0; h < NH; h++)
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++) asm volatile("" : "+v"(WH[h][ks][nt]));

  // ---- input rows of a block: 8-column chunks, two per thread (PF_BM * c8p <= 2 * PF_NT), gathered as in mlp_big_fwd_body
  f32x4 xr[2][2];
  const float* x = a.x[p];
  const int ns = a.nseg[p];
  // (per-thread chunk coordinates are re-derived behind an opaque zero wherever they are used: hoisted out of the block loop
  // they cost ~20 registers of a budget the resident weights leave no room in, i.e. spills inside the loop)
  auto opaque_tid = [&] { int z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return tid + z; };
  auto fetch_x = [&](int m0) {
    const int tv = opaque_tid();
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int c = tv + u * PF_NT, row = c / c8p, k = (c - row * c8p) * 8;
      const bool ok = c < PF_BM * c8p && m0 + row < M && k < K0;
      xr[u][0] = xr[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!ok) continue;
      if (ns > 0) {
        int sg = 0;
#pragma unroll
        for (int t = 1; t < MF_MAXSEG; t++) sg = (t < ns && k >= a.seg[p][t].c0) ? t : sg;
        const MlpXSeg sgm = a.seg[p][sg];
        const int cend = sg + 1 < ns ? a.seg[p][sg + 1].c0 : K0;
        const int r = sgm.mod ? (m0 + row) % sgm.mod : m0 + row;
        const float* q = sgm.p + (long)r * sgm.ld + (k - sgm.c0);
        if (k + 8 <= cend) { xr[u][0] = *reinterpret_cast<const f32x4*>(q); xr[u][1] = *reinterpret_cast<const f32x4*>(q + 4); }
        else {
#pragma unroll
          for (int j = 0; j < 4; j++) { xr[u][0][j] = k + j < cend ? q[j] : 0.f; xr[u][1][j] = k + 4 + j < cend ? q[4 + j] : 0.f; }
        }
        continue;
      }
      const float* q = x + (long)(m0 + row) * a.ldx + k;
      if (k + 8 <= a.ldx) { xr[u][0] = *reinterpret_cast<const f32x4*>(q); xr[u][1] = *reinterpret_cast<const f32x4*>(q + 4); }
      else {
#pragma unroll
        for (int j = 0; j < 4; j++) { xr[u][0][j] = k + j < K0 ? q[j] : 0.f; xr[u][1][j] = k + 4 + j < K0 ? q[4 + j] : 0.f; }
      }
    }
  };
  auto put_x = [&](int m0) {  // -> X[0] as bf16 (every chunk up to layer 0's padded width: the buffer held an activation before)
    const int tv = opaque_tid();
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int c = tv + u * PF_NT, row = c / c8p, k = (c - row * c8p) * 8;
      if (c >= PF_BM * c8p) continue;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 4; j++) { v[j] = (__bf16)(k + j < K0 ? xr[u][0][j] : 0.f); v[4 + j] = (__bf16)(k + 4 + j < K0 ? xr[u][1][j] : 0.f); }
      *reinterpret_cast<bf16x8*>(X + row * XP + k) = v;
      if (ns > 0 && a.xw[p] && m0 + row < M && k < a.ldx) {  // the assembled fp32 rows (pad columns are zeros)
        float* o = a.xw[p] + (long)(m0 + row) * a.ldx + k;
        if (k + 8 <= a.ldx) { *reinterpret_cast<f32x4*>(o) = xr[u][0]; *reinterpret_cast<f32x4*>(o + 4) = xr[u][1]; }
        else {
#pragma unroll
          for (int j = 0; j < 4; j++) { if (k + j < a.ldx) o[j] = xr[u][0][j]; if (k + 4 + j < a.ldx) o[4 + j] = xr[u][1][j]; }
        }
      }
    }
  };

  fetch_x((int)blockIdx.x * PF_BM);
  lds_barrier();  // zeros / layer-0 fragments / biases visible
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int m0 = blk * PF_BM;
    put_x(m0);
    lds_barrier();
    if (blk + (int)gridDim.x < nblk) fetch_x((blk + (int)gridDim.x) * PF_BM);  // in flight during the whole block
    int cur = 0;
#pragma unroll
    for (int l = 0; l <= NH; l++) {
      const __bf16* xin = X + cur * PF_BM * XP;
      __bf16* xout = X + (cur ^ 1) * PF_BM * XP;
      f32x4 acc[PF_MT][2];
#pragma unroll
      for (int mt = 0; mt < PF_MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (dbg & 4) {  // (scratch timing: no MFMA loop)
      } else if (l == 0) {
#pragma unroll
        for (int ks = 0; ks < PF_K0S; ks++) {
          if (ks >= KS0) break;
          bf16x8 A[PF_MT];
#pragma unroll
          for (int mt = 0; mt < PF_MT; mt++) A[mt] = *reinterpret_cast<const bf16x8*>(xin + (16 * mt + i) * XP + 32 * ks + 8 * g);
#pragma unroll
          for (int nt = 0; nt < 2; nt++) {
            const bf16x8 Bf = __builtin_bit_cast(bf16x8, W0[((w * PF_K0S + ks) * 2 + nt) * 64 + lane]);
#pragma unroll
            for (int mt = 0; mt < PF_MT; mt++) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf, A[mt], acc[mt][nt], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
          bf16x8 A[PF_MT];
#pragma unroll
          for (int mt = 0; mt < PF_MT; mt++) A[mt] = *reinterpret_cast<const bf16x8*>(xin + (16 * mt + i) * XP + 32 * ks + 8 * g);
#pragma unroll
          for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int mt = 0; mt < PF_MT; mt++)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WH[l == 0 ? 0 : l - 1][ks][nt]), A[mt], acc[mt][nt], 0, 0, 0);
        }
      }
      lds_barrier();  // the previous layer's copy-out has left S and xout
      if (!(dbg & 8)) {  // (dbg 8, scratch timing: no epilogue)
        const int col = n0 + 8 * g;  // this lane's 8 adjacent columns: tile 0 -> col .. col + 3, tile 1 -> col + 4 .. col + 7
        const f32x4 bv0 = *reinterpret_cast<const f32x4*>(BI + 256 * l + col), bv1 = *reinterpret_cast<const f32x4*>(BI + 256 * l + col + 4);
#pragma unroll
        for (int mt = 0; mt < PF_MT; mt++) {
          const int row = 16 * mt + i;
          bf16x8 yv;
          f16x8 sv;
#pragma unroll
          for (int nt = 0; nt < 2; nt++) {
            const f32x4 z4 = acc[mt][nt] + (nt ? bv1 : bv0);
            // two elements per instruction (v_pk_mul_f32 / v_pk_add_f32): the epilogue is VALU-issue bound - 67 of the kernel's
            // 211 us standalone (profiles/r05_mlp_persistent_dissect.md).  Same operations in the same order as sigmoid_fast /
            // the scalar form (mul, exp2, add, rcp, mul, sub, mul, add: no contraction), so the saves stay bit-identical.
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
              const f32x2 z = {z4[r], z4[r + 1]};
              const f32x2 t = z * f32x2{-1.44269504088896341f, -1.44269504088896341f};
              const f32x2 d = f32x2{(dbg & 2) ? 1.f : __builtin_amdgcn_exp2f(t[0]), (dbg & 2) ? 1.f : __builtin_amdgcn_exp2f(t[1])} + f32x2{1.f, 1.f};
              const f32x2 sg = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
              const f32x2 y = z * sg;
              const f32x2 sd = sg + y * (f32x2{1.f, 1.f} - sg);
              yv[4 * nt + r] = (__bf16)y[0]; yv[4 * nt + r + 1] = (__bf16)y[1];
              sv[4 * nt + r] = (_Float16)sd[0]; sv[4 * nt + r + 1] = (_Float16)sd[1];
            }
          }
          *reinterpret_cast<bf16x8*>(xout + row * XP + col) = yv;
          *reinterpret_cast<f16x8*>(S + row * XP + col) = sv;
        }
      }
      lds_barrier();
      {  // both copies leave as 16-byte coalesced stores (zero rows beyond M up to Mp)
        __bf16* y16 = reinterpret_cast<__bf16*>(a.act[p] + a.ybf[p][l]);
        __bf16* s16 = reinterpret_cast<__bf16*>(a.act[p] + a.sbf[p][l]);
        const int tv = opaque_tid();
#pragma unroll
        for (int u = 0; u < PF_BM * 32 / PF_NT; u++) {
          const int c = tv + u * PF_NT, row = c >> 5, k = (c & 31) * 8;
          const __bf16 z0 = (__bf16)0.f;
          bf16x8 yv = {z0, z0, z0, z0, z0, z0, z0, z0}, sv = yv;
          if (m0 + row < M) {
            yv = *reinterpret_cast<const bf16x8*>(xout + row * XP + k);
            sv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(S) + row * XP + k);
          }
          if (!(dbg & 1)) {
            *reinterpret_cast<bf16x8*>(y16 + (long)(m0 + row) * 256 + k) = yv;
            *reinterpret_cast<bf16x8*>(s16 + (long)(m0 + row) * 256 + k) = sv;
          } else asm volatile("" :: "v"(yv), "v"(sv));
        }
      }
      cur ^= 1;
    }
    {  // output layer: thread = (row, 32-column segment)
      const int tv = opaque_tid(), row = tv >> 3, seg = tv & 7;
      const __bf16* yr = X + cur * PF_BM * XP + row * XP + 32 * seg;
      float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j8 = 0; j8 < 4; j8++) {
        const bf16x8 yv = *reinterpret_cast<const bf16x8*>(yr + 8 * j8);
#pragma unroll
        for (int n = 0; n < 4; n++) {
          if (n < NL) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(WO + 256 * n + 32 * seg + 8 * j8);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(WO + 256 * n + 32 * seg + 8 * j8 + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) { part[n] += (float)yv[e] * w0[e]; part[n] += (float)yv[4 + e] * w1[e]; }
          }
        }
      }
#pragma unroll
      for (int n = 0; n < 4; n++) {
        part[n] += __shfl_xor(part[n], 1);
        part[n] += __shfl_xor(part[n], 2);
        part[n] += __shfl_xor(part[n], 4);
      }
      if (seg == 0 && m0 + row < M) {
        float* yb = a.act[p] + a.yout[p] + (long)(m0 + row) * NL;
#pragma unroll
        for (int n = 0; n < 4; n++)
          if (n < NL) yb[n] = part[n] + WO[4 * 256 + n];
      }
    }
    lds_barrier();  // (the next block's rows overwrite X[0], which the output layer has just read when NH is odd)
  }
}

struct MlpBigBwdArgs {
  const float* d_out[MF_MAXP];
  const float* act[MF_MAXP];
  const __bf16* wt[MF_MAXP];
  float* dz[MF_MAXP];      // dZ_l (bf16 [Mp][dims[l+1]]) at dzoff[p][l] floats, l = 0 .. L-2
  float* d_x[MF_MAXP];     // optional fp32 [M][ldd]
  int M[MF_MAXP];
  long sbf[MF_MAXP][MF_MAXL];  // float offset in act[p] of hidden layer l's fp16 act' copy
  long dzoff[MF_MAXP][MF_MAXL];
  long wtoff[MF_MAXL];
  int dims[MF_MAXL + 1];
  int L, ldo, ldd;
};

template <int BMF_>
__device__ __forceinline__ void mlp_big_bwd_body(const MlpBigBwdArgs& a) {
  constexpr int BMF = BMF_, MTF = BMF / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* X = reinterpret_cast<__bf16*>(smem);                      // [2][BMF][XP]
  _Float16* S = reinterpret_cast<_Float16*>(X + 2 * BMF * XP);      // [BMF][XP] act' of the layer below
  const int p = blockIdx.y, m0 = blockIdx.x * BMF, M = a.M[p];
  if (m0 >= ((M + 63) & ~63)) return;  // (a block of padding rows only still writes their zero dZ rows)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
  const int n0 = MF_CW * w;
  bf16x8 B[8][MF_NTW];
  auto load_layer = [&](int l) {
    const int KO = a.dims[l], NP = (a.dims[l + 1] + 7) / 8 * 8;
    const __bf16* T = a.wt[p] + a.wtoff[l];
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int nt = 0; nt < MF_NTW; nt++) B[ks][nt] = load_w(T, NP, KO, n0 + 16 * nt + i, 32 * ks + 8 * g);
  };
  const int l_last = a.d_x[p] ? 0 : 1;
  // one global round trip in front of the chain (see mlp_big_fwd_kernel): d_out rows and the last layer's W^T together
  const int NL = a.dims[a.L];  // <= 4: BMF * NL <= MF_NT
  float dreg = 0.f;
  {
    const int row = tid / NL, n = tid - row * NL;
    if (tid < BMF * NL && m0 + row < M) dreg = a.d_out[p][(long)(m0 + row) * a.ldo + n];
  }
  if (a.L - 1 >= l_last && n0 < a.dims[a.L - 1]) load_layer(a.L - 1);
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  for (int e = tid; e < 2 * BMF * XP / 8; e += MF_NT) reinterpret_cast<f32x4*>(X)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  lds_barrier();
  if (tid < BMF * NL) { const int row = tid / NL, n = tid - row * NL; X[row * XP + n] = (__bf16)dreg; }
  lds_barrier();
  int cur = 0;
  for (int l = a.L - 1; l >= l_last; l--) {
    const int KO = a.dims[l], NR = a.dims[l + 1], KS = (NR + 31) / 32;
    const __bf16* xin = X + cur * BMF * XP;
    __bf16* xout = X + (cur ^ 1) * BMF * XP;
    f32x4 acc[MTF][MF_NTW];
    const bool masked = l > 0;  // hidden layer l-1's act' multiplies this layer's input gradient
    const int ldout = l > 0 ? KO : a.ldd;
    const bool vec = (KO & 3) == 0 && (ldout & 3) == 0;
    // this workgroup's 64 x KO tile of act' (fp16, row-major: 16 bytes per lane, coalesced) travels under the MFMA loop
    bf16x8 spre[2];
    const int c8n = KO >> 3;  // (KO % 8 == 0 for hidden widths)
    if (masked) {
      const __bf16* s16 = reinterpret_cast<const __bf16*>(a.act[p] + a.sbf[p][l - 1]);
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int c = tid + u * MF_NT, row = c / c8n, k = (c - row * c8n) * 8;
        const bool ok = c < BMF * c8n && m0 + row < M;
        spre[u] = *reinterpret_cast<const bf16x8*>(s16 + (ok ? (long)(m0 + row) * KO + k : 0));
      }
    }
    if (n0 < KO) {
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
    if (masked) {
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int c = tid + u * MF_NT, row = c / c8n, k = (c - row * c8n) * 8;
        if (c < BMF * c8n) *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(S) + row * XP + k) = spre[u];
      }
      lds_barrier();
    }
    if (n0 < KO) {
#pragma unroll
      for (int nt = 0; nt < MF_NTW; nt++) {
        const int col = n0 + 16 * nt + 4 * g;
        if (n0 + 16 * nt >= KO) continue;
#pragma unroll
        for (int mt = 0; mt < MTF; mt++) {
          const int row = 16 * mt + i;
          const bool rok = m0 + row < M;
          f32x4 v = acc[mt][nt];
          if (masked) {
            const f16x4 sv = *reinterpret_cast<const f16x4*>(S + row * XP + col);
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] *= (float)sv[r];
          }
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (!rok || col + r >= KO) v[r] = 0.f;
          if (l == 0 && rok) {
            float* o = a.d_x[p] + (long)(m0 + row) * ldout + col;
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
    if (l > 0) {  // dZ_{l-1} as the chain consumes it leaves from LDS, 16 bytes per lane (zero rows beyond M)
      __bf16* d16 = reinterpret_cast<__bf16*>(a.dz[p] + a.dzoff[p][l - 1]);
      for (int c = tid; c < BMF * c8n; c += MF_NT) {
        const int row = c / c8n, k = (c - row * c8n) * 8;
        *reinterpret_cast<bf16x8*>(d16 + (long)(m0 + row) * KO + k) = *reinterpret_cast<const bf16x8*>(xout + row * XP + k);
      }
    }
    cur ^= 1;
  }
}
__global__ __launch_bounds__(MF_NT) void mlp_big_bwd_kernel(MlpBigBwdArgs a) { mlp_big_bwd_body<64>(a); }
__global__ __launch_bounds__(MF_NT) void mlp_mid_bwd_kernel(MlpBigBwdArgs a) { mlp_big_bwd_body<32>(a); }

// ---- persistent many-row input-gradient chain (round 5): the backward twin of mlp_pers_fwd_kernel, same saves and
// layouts as mlp_big_bwd_body.  One resident workgroup per CU over 64-row blocks; W_l^T of the 256 x 256 layers in
// registers, W_0^T (<= 96 output columns) in LDS in fragment order, the output layer's 1 .. 4 rows in LDS as fp32.
//  * the last layer (dZ_{L-2} = (d_out W_out) * act'_{L-2}, a rank-NL product) is element-parallel: thread = (row, 8 columns),
//    act' read 16 bytes per lane straight from its row-major copy, dZ written to LDS and to HBM from the same registers;
//  * a 256 x 256 layer: act' of the layer below is requested before the MFMA loop, dropped into LDS behind it, applied to the
//    accumulators; dZ leaves from LDS as 16-byte coalesced stores under the next layer's MFMAs;
//  * layer 0 (d_x, fp32): waves 0 .. 5 own one 16-column tile each; the rows leave through LDS (the accumulator layout
//    gives 64-byte segments - request-bound - straight from registers);
//  * the next block's d_out and act'_{L-2} rows are requested during layer 0, where the register budget has room.
constexpr int PB_W0B = 6 * 8 * 1024;                       // W_0^T fragments: [6 n tiles][8 k-steps][64 lanes] x 16 B
constexpr int PB_LDS = PF_XB + PF_SB + PB_W0B + PF_MISC;
static_assert(PB_LDS <= 160 * 1024, "persistent backward: LDS");

template <int NH>
__global__ __launch_bounds__(PF_NT) void mlp_pers_bwd_kernel(MlpBigBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* X = reinterpret_cast<__bf16*>(smem);                          // [2][PF_BM][XP]; X[0] also stages d_x as fp32 [PF_BM][100]
  _Float16* S = reinterpret_cast<_Float16*>(smem + PF_XB);              // [PF_BM][XP] act' of the layer below
  pf_u32x4* W0 = reinterpret_cast<pf_u32x4*>(smem + PF_XB + PF_SB);     // [(nt * 8 + ks)][lane]
  float* WO = reinterpret_cast<float*>(smem + PF_XB + PF_SB + PB_W0B);  // [NL <= 4][256] the output layer's rows (bf16 values)
  float* DQ = WO + 4 * 256;                                             // [PF_BM][4] this block's d_out, bf16-rounded
  constexpr int L = NH + 2;
  const int p = blockIdx.y, M = a.M[p], Mp = (M + 63) & ~63, nblk = Mp / PF_BM;
  if ((int)blockIdx.x >= nblk) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
  const int n0 = 32 * w;
  const int K0 = a.dims[0], NL = a.dims[L], NPL = (NL + 7) / 8 * 8;
  const bool want_dx = a.d_x[p] != nullptr;
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto opaque_tid = [&] { int z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return tid + z; };

  // ---- launch-resident operands: WH[h] = W_{h+1}^T fragments (row n of the packed transpose = input feature n)
  pf_u32x4 WH[NH][8][2];
#pragma unroll
  for (int h = 0; h < NH; h++) {
    const __bf16* T = a.wt[p] + a.wtoff[h + 1];
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)  // (column permutation of the wave's two tiles as in mlp_pers_fwd_kernel: 8 adjacent columns per lane)
        WH[h][ks][nt] = __builtin_bit_cast(pf_u32x4, load_w(T, 256, 256, n0 + 8 * (i >> 2) + 4 * nt + (i & 3), 32 * ks + 8 * g));
  }
  if (want_dx) {
    const __bf16* T = a.wt[p] + a.wtoff[0];  // [K0][256]
    for (int f = w; f < 6 * 8; f += PF_NT / 64) {
      const int nt = f >> 3, ks = f & 7;
      W0[f * 64 + lane] = __builtin_bit_cast(pf_u32x4, load_w(T, 256, K0, 16 * nt + i, 32 * ks + 8 * g));
    }
  }
  {
    const __bf16* T = a.wt[p] + a.wtoff[L - 1];  // [256][NPL]: T[k * NPL + n] = W_out[n][k]
    for (int e = tid; e < 4 * 256; e += PF_NT) {
      const int n = e >> 8, k = e & 255;
      WO[e] = n < NL ? (float)T[k * NPL + n] : 0.f;
    }
  }
#pragma unroll
  for (int h = 0; h < NH; h++)
#pragma unroll
    for (int ks = 0; ks < 8; ks++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++) asm volatile("" : "+v"(WH[h][ks][nt]));

  // act' rows of a block, 16 bytes per lane in row order (chunk c = tid + u * 512: row c >> 5, columns 8 (c & 31) ..)
  const __bf16* s_top = reinterpret_cast<const __bf16*>(a.act[p] + a.sbf[p][L - 2]);
  bf16x8 spre[4];
  float dpre = 0.f;
  auto fetch_top = [&](int m0) {  // the block's d_out and act'_{L-2}
    const int tv = opaque_tid();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int c = tv + u * PF_NT, row = c >> 5, k = (c & 31) * 8;
      spre[u] = *reinterpret_cast<const bf16x8*>(s_top + (m0 + row < M ? (long)(m0 + row) * 256 + k : 0));
    }
    const int row = tv >> 2, n = tv & 3;
    dpre = (tv < PF_BM * 4 && n < NL && m0 + row < M) ? a.d_out[p][(long)(m0 + row) * a.ldo + n] : 0.f;
  };
  fetch_top((int)blockIdx.x * PF_BM);
  lds_barrier();
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int m0 = blk * PF_BM;
    if (tid < PF_BM * 4) DQ[tid] = (float)(__bf16)dpre;
    lds_barrier();
    // ---- last layer: dZ_{L-2}[row][k] = (sum_n d_out[row][n] W_out[n][k]) act'_{L-2}[row][k] -> X[0] and HBM
    {
      __bf16* d16 = reinterpret_cast<__bf16*>(a.dz[p] + a.dzoff[p][L - 2]);
      const int tv = opaque_tid();
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = tv + u * PF_NT, row = c >> 5, k = (c & 31) * 8;
        const bool rok = m0 + row < M;
        const f32x4 dq = *reinterpret_cast<const f32x4*>(DQ + 4 * row);
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < 4; n++) {
          if (n < NL) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(WO + 256 * n + k), w1 = *reinterpret_cast<const f32x4*>(WO + 256 * n + k + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] += dq[n] * w0[e]; v[4 + e] += dq[n] * w1[e]; }
          }
        }
        const _Float16* sv = reinterpret_cast<const _Float16*>(&spre[u]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (__bf16)(rok ? v[e] * (float)sv[e] : 0.f);
        *reinterpret_cast<bf16x8*>(X + row * XP + k) = o;
        *reinterpret_cast<bf16x8*>(d16 + (long)(m0 + row) * 256 + k) = o;
      }
    }
    lds_barrier();
    int cur = 0;
    // ---- the 256 x 256 layers l = L-2 .. 1: dZ_{l-1} = (dZ_l W_l) * act'_{l-1}
#pragma unroll
    for (int l = NH; l >= 1; l--) {
      const __bf16* xin = X + cur * PF_BM * XP;
      __bf16* xout = X + (cur ^ 1) * PF_BM * XP;
      const __bf16* s16 = reinterpret_cast<const __bf16*>(a.act[p] + a.sbf[p][l - 1]);
      bf16x8 sl[4];
      {
        const int tv = opaque_tid();
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int c = tv + u * PF_NT, row = c >> 5, k = (c & 31) * 8;
          sl[u] = *reinterpret_cast<const bf16x8*>(s16 + (m0 + row < M ? (long)(m0 + row) * 256 + k : 0));
        }
      }
      f32x4 acc[PF_MT][2];
#pragma unroll
      for (int mt = 0; mt < PF_MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        bf16x8 A[PF_MT];
#pragma unroll
        for (int mt = 0; mt < PF_MT; mt++) A[mt] = *reinterpret_cast<const bf16x8*>(xin + (16 * mt + i) * XP + 32 * ks + 8 * g);
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
          for (int mt = 0; mt < PF_MT; mt++)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, WH[l - 1][ks][nt]), A[mt], acc[mt][nt], 0, 0, 0);
      }
      {  // act' of the layer below -> S (nobody reads S any more: its last readers were the previous epilogue, two barriers ago)
        const int tv = opaque_tid();
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int c = tv + u * PF_NT, row = c >> 5, k = (c & 31) * 8;
          *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(S) + row * XP + k) = sl[u];
        }
      }
      lds_barrier();  // S complete; the previous copy-out has left xout
      {
        const int col = n0 + 8 * g;
#pragma unroll
        for (int mt = 0; mt < PF_MT; mt++) {
          const int row = 16 * mt + i;
          const bool rok = m0 + row < M;
          const f16x8 sv = *reinterpret_cast<const f16x8*>(S + row * XP + col);
          bf16x8 o;
#pragma unroll
          for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++) o[4 * nt + r] = (__bf16)(rok ? acc[mt][nt][r] * (float)sv[4 * nt + r] : 0.f);
          *reinterpret_cast<bf16x8*>(xout + row * XP + col) = o;
        }
      }
      lds_barrier();
      {
        __bf16* d16 = reinterpret_cast<__bf16*>(a.dz[p] + a.dzoff[p][l - 1]);
        const int tv = opaque_tid();
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int c = tv + u * PF_NT, row = c >> 5, k = (c & 31) * 8;
          *reinterpret_cast<bf16x8*>(d16 + (long)(m0 + row) * 256 + k) = *reinterpret_cast<const bf16x8*>(xout + row * XP + k);
        }
      }
      cur ^= 1;
    }
    // ---- the next block's top rows travel during layer 0
    if (blk + (int)gridDim.x < nblk) fetch_top((blk + (int)gridDim.x) * PF_BM);
    // ---- layer 0: d_x = dZ_0 W_0 (fp32 rows, K0 columns), through LDS
    if (want_dx) {
      const __bf16* xin = X + cur * PF_BM * XP;
      float* stage = reinterpret_cast<float*>(X + (cur ^ 1) * PF_BM * XP);  // [PF_BM][100]
      if (w < 6 && 16 * w < K0) {
        f32x4 acc[PF_MT];
#pragma unroll
        for (int mt = 0; mt < PF_MT; mt++) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
          const bf16x8 Bf = __builtin_bit_cast(bf16x8, W0[(w * 8 + ks) * 64 + lane]);
#pragma unroll
          for (int mt = 0; mt < PF_MT; mt++) {
            const bf16x8 A = *reinterpret_cast<const bf16x8*>(xin + (16 * mt + i) * XP + 32 * ks + 8 * g);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf, A, acc[mt], 0, 0, 0);
          }
        }
#pragma unroll
        for (int mt = 0; mt < PF_MT; mt++) *reinterpret_cast<f32x4*>(stage + (16 * mt + i) * 100 + 16 * w + 4 * g) = acc[mt];
      }
      lds_barrier();
      {
        const int tv = opaque_tid();
        const int c4 = (K0 + 3) / 4;  // 4-column chunks of a row (<= 24)
        for (int c = tv; c < PF_BM * c4; c += PF_NT) {
          const int row = c / c4, k = (c - row * c4) * 4;
          if (m0 + row < M) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 100 + k);
            float* o = a.d_x[p] + (long)(m0 + row) * a.ldd + k;
            if (k + 4 <= K0 && (a.ldd & 3) == 0) *reinterpret_cast<f32x4*>(o) = v;
            else {
#pragma unroll
              for (int r = 0; r < 4; r++)
                if (k + r < K0) o[r] = v[r];
            }
          }
        }
      }
    }
    lds_barrier();  // (the next block's last layer writes X[0] and DQ)
  }
}

// ------------------------------------------------------------------ weight gradients, tens of thousands of rows
// C5's Q networks see (3 n + 1) B = 99 328 rows each.  mlp_wgrad_fused_kernel above reads both operands as fp32, recomputes
// the SiLU of the lean forward per element and re-lays everything through registers: 915 us per step for the two
// networks (61 TFLOP/s, 1.3 TB/s) - bound by the VALU work in front of every MFMA, not by bytes.  Here both operands arrive
// as the bf16 row-major copies their producers leave behind (the forward: y_l = the very bf16 values the next layer
// consumed; the input-gradient chain: dZ_l as the next layer of the chain consumed it) and go HBM -> LDS by LDS-DMA in
// their natural order, 64 rows per stage on a 2-stage ring, 16-byte chunks XOR-swizzled through the choice of global
// chunk per lane; MFMA fragments by ds_read_b64_tr_b16 (the reduction index - the batch row - is the slow index of both:
// rnn_ops.hip's rnn_wgrad_kernel has the derivation).  A workgroup owns ALL 256 dZ columns x 128 x columns of one
// (network, layer) over a slice of rows (dZ is then read twice per layer, x once): 8 waves as 4 (64 dZ columns) x 2
// (64 x columns), 64 accumulator registers; the bias gradient is one more MFMA against ones.  Slices go to the slab
// records mlp_wgrad_reduce_kernel sums in slice order (deterministic).  Layer 0's x (71 columns of fp32) is first
// converted to a zero-padded bf16 [Mp][128] copy (mlp_x_to_bf16_kernel); the last layer (256 -> 1..4 outputs) is a
// weighted column sum of y_{L-2} (mlp_wgrad_out_kernel).
constexpr int BG_R = 64, BG_S = 2;                 // rows per stage, stages
constexpr int BG_OPB = BG_R * 256;                  // bytes of one 128-column operand block of a stage
constexpr int BG_STAGE = 3 * BG_OPB;                // [dZ columns 0..127][dZ columns 128..255][x columns of this tile]
constexpr int BG_NW = 8;

typedef __bf16 bf16x4tr __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 bg_tr_frag8(const unsigned char* lo, const unsigned char* hi) {
  const bf16x4tr a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4tr*)(lo));
  const bf16x4tr b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4tr*)(hi));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct MlpWgBigArgs {
  const __bf16* dz[MF_MAXP][MF_MAXL];  // [Mp][256]
  const __bf16* xb[MF_MAXP][MF_MAXL];  // [Mp][ldxb[l]]
  float* slab[MF_MAXP];
  int Mp[MF_MAXP];
  long sloff[MF_MAXL], rec;
  int K[MF_MAXL], ldxb[MF_MAXL];       // true input width of the layer (columns of dW), row pitch of its bf16 operand
  int tile0[MF_MAXL + 1];              // first x tile (128 columns) of layer l in blockIdx.y
  int nl;                              // layers handled here (0 .. L-2)
  int rps;                             // rows per slice (a multiple of BG_R)
};

__global__ __launch_bounds__(64 * BG_NW) void mlp_wgrad_big_kernel(MlpWgBigArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int p = blockIdx.z, Mp = a.Mp[p];
  const int r_begin = blockIdx.x * a.rps;
  if (r_begin >= Mp) return;
  const int r_end = min(Mp, r_begin + a.rps), nk = (r_end - r_begin) / BG_R;
  int l = 0;
  while (l + 1 < a.nl && (int)blockIdx.y >= a.tile0[l + 1]) l++;
  const int nt = blockIdx.y - a.tile0[l], n0 = nt * 128;  // x columns [n0, n0 + 128)
  const __bf16* __restrict__ dz = a.dz[p][l] + (long)r_begin * 256;
  const int ldx = a.ldxb[l], K = a.K[l];
  const __bf16* __restrict__ xb = a.xb[p][l] + (long)r_begin * ldx + n0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, l16 = lane & 15, q = l16 >> 2, pc = l16 & 3;
  const int wm = w >> 1, wn = w & 1;  // wave grid: 4 (64 dZ columns each) x 2 (64 x columns each)
  constexpr int DPW = 3 * (BG_R / 4) / BG_NW;  // DMA wave-instructions per stage and wave (4 rows x 256 B of one block each)

  auto issue = [&](int kt, int slot) {
#pragma unroll
    for (int i = 0; i < DPW; i++) {
      const int id = w + BG_NW * i;
      const int op = id / (BG_R / 4), row4 = (id % (BG_R / 4)) * 4;
      const int r = row4 + (lane >> 4), cpos = lane & 15, c = cpos ^ (2 * (r & 7));
      const __bf16* src = (op == 2 ? xb + (long)(kt * BG_R + r) * ldx : dz + (long)(kt * BG_R + r) * 256 + 128 * op) + c * 8;
      const unsigned lds_off = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)(
          lds + slot * BG_STAGE + op * BG_OPB + row4 * 256));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src), "s"(lds_off)
                   : "memory");
    }
  };

  f32x4 acc[4][4], bacc[4];  // [x tile ni][dZ tile mi]
#pragma unroll
  for (int mi = 0; mi < 4; mi++) {
    bacc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < 4; ni++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; j++) ones[j] = (__bf16)1.0f;
  if (nk > 0) issue(0, 0);
  // fragment addresses inside a stage: row 4 g + q (+16), 16-byte chunk (2 tile + pc / 2) ^ swizzle, half pc % 2
  const int row = 4 * g + q, sw = 2 * (row & 7), half = 8 * (pc & 1);
  int offA[4], offB[4];
#pragma unroll
  for (int mi = 0; mi < 4; mi++)
    offA[mi] = (wm >> 1) * BG_OPB + row * 256 + (((2 * (4 * (wm & 1) + mi) + (pc >> 1)) ^ sw) << 4) + half;
#pragma unroll
  for (int ni = 0; ni < 4; ni++) offB[ni] = 2 * BG_OPB + row * 256 + (((2 * (4 * wn + ni) + (pc >> 1)) ^ sw) << 4) + half;
  const bool bias_wave = wn == 0 && nt == 0;

  for (int kt = 0; kt < nk; kt++) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of stage kt have landed
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everyone's have; stage kt - 1 is no longer read
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    const unsigned char* st = lds + (kt & 1) * BG_STAGE;
#pragma unroll
    for (int sub = 0; sub < BG_R / 32; sub++) {
      const unsigned char* sb = st + sub * 32 * 256;
      bf16x8 A[4], B[4];
#pragma unroll
      for (int mi = 0; mi < 4; mi++) A[mi] = bg_tr_frag8(sb + offA[mi], sb + offA[mi] + 16 * 256);
#pragma unroll
      for (int ni = 0; ni < 4; ni++) B[ni] = bg_tr_frag8(sb + offB[ni], sb + offB[ni] + 16 * 256);
#pragma unroll
      for (int mi = 0; mi < 4; mi++) {
#pragma unroll
        for (int ni = 0; ni < 4; ni++) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ni], A[mi], acc[ni][mi], 0, 0, 0);
        if (bias_wave) bacc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, A[mi], bacc[mi], 0, 0, 0);
      }
    }
  }
  // D[i][j]: i = x column inside its 16-tile (this lane: 4 g .. 4 g + 3), j = dZ column (lane l16) -> dW[n][k .. k + 3]
  float* rec = a.slab[p] + (long)blockIdx.x * a.rec + a.sloff[l];
  const bool vecK = (K & 3) == 0;
#pragma unroll
  for (int mi = 0; mi < 4; mi++) {
    const int n = 64 * wm + 16 * mi + l16;
#pragma unroll
    for (int ni = 0; ni < 4; ni++) {
      const int k = n0 + 64 * wn + 16 * ni + 4 * g;
      float* o = rec + (long)n * K + k;
      if (vecK) {
        if (k < K) *reinterpret_cast<f32x4*>(o) = acc[ni][mi];
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (k + r < K) o[r] = acc[ni][mi][r];
      }
    }
    if (bias_wave && g == 0) rec[(long)256 * K + n] = bacc[mi][0];
  }
}

// x fp32 [M][ldx] (K0 columns) -> bf16 [Mp][128], zero beyond K0 columns / M rows
struct MlpXbArgs { const float* x[MF_MAXP]; __bf16* xb[MF_MAXP]; int M[MF_MAXP], Mp[MF_MAXP]; int ldx, K0; };
__global__ __launch_bounds__(256) void mlp_x_to_bf16_kernel(MlpXbArgs a) {
  const int p = blockIdx.y, M = a.M[p], Mp = a.Mp[p], K0 = a.K0;
  const float* __restrict__ x = a.x[p];
  __bf16* __restrict__ o = a.xb[p];
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < (long)Mp * 16; c += (long)gridDim.x * 256) {
    const int row = (int)(c >> 4), k = ((int)c & 15) * 8;
    const __bf16 z0 = (__bf16)0.f;
    bf16x8 v = {z0, z0, z0, z0, z0, z0, z0, z0};
    if (row < M && k < K0) {
      const float* q = x + (long)row * a.ldx + k;
      if (k + 8 <= a.ldx && (a.ldx & 3) == 0) {  // inside the row's allocation: two 16-byte loads, masked at the ragged end
        const f32x4 lo = *reinterpret_cast<const f32x4*>(q), hi = *reinterpret_cast<const f32x4*>(q + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) { v[j] = (__bf16)(k + j < K0 ? lo[j] : 0.f); v[4 + j] = (__bf16)(k + 4 + j < K0 ? hi[j] : 0.f); }
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (k + j < K0) v[j] = (__bf16)q[j];
      }
    }
    *reinterpret_cast<bf16x8*>(o + (long)row * 128 + k) = v;
  }
}

// last layer with NL <= 4 outputs: dW[n][k] = sum_r bf16(d_out[r][n]) * y[r][k], db[n] = sum_r bf16(d_out[r][n]) (the operand
// rounding of the MFMA path); y bf16 [Mp][256].  A bandwidth kernel (51 MB of y per network): OUT_RPW rows per workgroup,
// thread = (row group of 8, 8 columns), four 16-byte loads in flight per thread; partials [slice][NL][257] to a slab of
// their own, summed in slice order by mlp_wgrad_reduce_kernel (as a one-layer record).
constexpr int OUT_RPW = 512;
struct MlpWgOutArgs {
  const __bf16* y[MF_MAXP]; const float* dlast[MF_MAXP]; float* slab[MF_MAXP];
  int M[MF_MAXP];
  long rec;
  int NL, ldo, rpw;  // rpw: rows per workgroup (OUT_RPW at tens of thousands of rows, 64 at a few thousand: latency, not bytes)
};
__global__ __launch_bounds__(256) void mlp_wgrad_out_kernel(MlpWgOutArgs a) {
  __shared__ float red[8][4][256 + 1];
  const int p = blockIdx.y, M = a.M[p], r0 = blockIdx.x * a.rpw;
  if (r0 >= M) return;
  const int r1 = min(M, r0 + a.rpw), tid = threadIdx.x, cg = tid & 31, rg = tid >> 5, NL = a.NL;
  float acc[4][8], bsum[4];
#pragma unroll
  for (int n = 0; n < 4; n++) {
    bsum[n] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) acc[n][j] = 0.f;
  }
  const __bf16* __restrict__ y = a.y[p];
  const float* __restrict__ d = a.dlast[p];
  for (int rb = r0 + rg; rb < r1; rb += 32) {
    bf16x8 v[4];
    float dv[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++) {  // (rows beyond the slice: clamped address, zero weight)
      const int r = rb + 8 * u, rc = r < r1 ? r : r0;
      v[u] = *reinterpret_cast<const bf16x8*>(y + (long)rc * 256 + 8 * cg);
#pragma unroll
      for (int n = 0; n < 4; n++) dv[u][n] = (n < NL && r < r1) ? (float)(__bf16)d[(long)rc * a.ldo + n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int n = 0; n < 4; n++)
        if (n < NL) {
          bsum[n] += dv[u][n];
#pragma unroll
          for (int j = 0; j < 8; j++) acc[n][j] += dv[u][n] * (float)v[u][j];
        }
  }
#pragma unroll
  for (int n = 0; n < 4; n++) {
#pragma unroll
    for (int j = 0; j < 8; j++) red[rg][n][8 * cg + j] = acc[n][j];
    if (cg == 0) red[rg][n][256] = bsum[n];
  }
  __syncthreads();
  float* rec = a.slab[p] + (long)blockIdx.x * a.rec;  // one-layer record: [NL][256] then [NL]
  for (int e = tid; e < NL * 257; e += 256) {
    const int n = e / 257, k = e - n * 257;
    float t = 0.f;
#pragma unroll
    for (int g8 = 0; g8 < 8; g8++) t += red[g8][n][k];  // fixed order
    if (k < 256) rec[(long)n * 256 + k] = t; else rec[(long)NL * 256 + n] = t;
  }
}

}  // namespace

// ---- deferred weight-only preparation: host side (kernel: prep_multi_kernel above)
namespace {
struct PrepState {
  bool on = false;
  PrepMultiArgs a{};
  int nprob[PREP_MAXS] = {};
};
thread_local PrepState g_prep;
int prep_flush(hipStream_t st) {
  PrepMultiArgs& a = g_prep.a;
  if (a.nsite == 0 && a.ncv == 0) return TACORL_OK;
  int maxL = 1;
  for (int s = 0; s < a.nsite; s++) maxL = a.site[s].L > maxL ? a.site[s].L : maxL;
  hipLaunchKernelGGL(prep_multi_kernel, dim3(64, maxL, a.first[a.nsite] + a.ncv), dim3(256), 0, st, a);
  a.nsite = 0; a.ncv = 0; a.first[0] = 0;
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
}  // namespace
bool prep_deferring() { return g_prep.on; }
// (stream: where a full list is flushed to - the same stream _end will be given)
int prep_defer_wtpack(const WtPackArgs& k, int nprob, hipStream_t st) {
  PrepMultiArgs& a = g_prep.a;
  if (a.nsite == PREP_MAXS) { const int rc = prep_flush(st); if (rc != TACORL_OK) return rc; }
  a.site[a.nsite] = k;
  a.first[a.nsite + 1] = a.first[a.nsite] + nprob;
  a.nsite++;
  return TACORL_OK;
}
int prep_defer_bf16(const float* src, void* dst, long count, hipStream_t st) {
  PrepMultiArgs& a = g_prep.a;
  if (a.ncv == PREP_MAXB) { const int rc = prep_flush(st); if (rc != TACORL_OK) return rc; }
  a.csrc[a.ncv] = src; a.cdst[a.ncv] = (__bf16*)dst; a.cn4[a.ncv] = count / 4;
  a.ncv++;
  return TACORL_OK;
}
extern "C" int tacorl_reduce_batch_begin(void) {
  if (g_red.on) return TACORL_EINVAL;
  g_red.on = true;
  g_red.m.njob = 0; g_red.m.first[0] = 0;
  return TACORL_OK;
}
extern "C" int tacorl_reduce_batch_end(void* stream) {
  if (!g_red.on) return TACORL_EINVAL;
  g_red.on = false;
  return red_flush((hipStream_t)stream) ? TACORL_ELAUNCH : TACORL_OK;
}
extern "C" int tacorl_prep_batch_begin(void) {
  if (g_prep.on) return TACORL_EINVAL;
  g_prep.on = true;
  g_prep.a.nsite = 0; g_prep.a.ncv = 0; g_prep.a.first[0] = 0;
  return TACORL_OK;
}
extern "C" int tacorl_prep_batch_end(void* stream) {
  if (!g_prep.on) return TACORL_EINVAL;
  g_prep.on = false;
  return prep_flush((hipStream_t)stream);
}


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
  if (launch_reduce(r, 256, L, n2, st)) return -1;
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- many-row weight gradients (mlp_wgrad_big_kernel): eligibility, sizes, launch
static int mlp_big_rows() {
  // 2048 since round 5 (16384 before): below 16 384 rows the same saves come from 32-row workgroups (mlp_mid_*_kernel)
  static const int v = [] { const char* e = getenv("TACORL_MLP_BIG_ROWS"); return e ? atoi(e) : 2048; }();
  return v;
}
bool mlp_big_prob_ok(int M, int L, const int* dims, const int* acts) {
  static const int on = [] { const char* e = getenv("TACORL_MLP_BIG"); return e ? atoi(e) : 1; }();  // A/B switch
  if (!on || L < 2 || L > MF_MAXL || M < mlp_big_rows()) return false;
  if (dims[0] < 8 || dims[0] > 128 || dims[L] < 1 || dims[L] > 4) return false;
  for (int l = 1; l < L; l++)
    if (dims[l] != 256) return false;
  if (acts) {
    for (int l = 0; l + 1 < L; l++)
      if (acts[l] != ACT_SILU) return false;
    if (acts[L - 1] != ACT_NONE) return false;
  }
  return true;
}
#ifdef MLP_STAMPS
extern "C" int tacorl_dbg_mlp_stamps(unsigned long long* dst) {
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_mlp_stamps), sizeof(g_mlp_stamps)) == hipSuccess ? 0 : -1;
}
#endif
// rows per workgroup of the many-row forward / input-gradient kernels: 128 / 64 at >= 16 384 rows, 32 / 32 below
static bool mlp_rows_huge(int maxM) { return maxM >= 16384; }
// experiment switches (rows per workgroup of the many-row forward / chain at >= 16 384 rows: 128 / 64 by default)
static int mlp_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static int mlp_huge_fwd_rows() { static const int v = mlp_env_int("TACORL_MLP_HUGE_FWD_ROWS", 128); return v; }
static int mlp_huge_bwd_rows() { static const int v = mlp_env_int("TACORL_MLP_HUGE_BWD_ROWS", 64); return v; }
template <class Args>
static int mlp_set_gather(Args& a, int p, int q, const MlpXGather* gather, int K0) {  // gather's problem q -> args' problem p
  const int ns = gather ? gather->nseg[q] : 0;
  if (ns < 0 || ns > MF_MAXSEG) return TACORL_EINVAL;
  a.nseg[p] = ns;
  a.xw[p] = ns ? gather->xw[q] : nullptr;
  if ((uintptr_t)a.xw[p] & 15) return TACORL_EINVAL;
  for (int t = 0; t < ns; t++) {
    const int c0 = gather->c0[q][t], ld = gather->ld[q][t];
    if (((uintptr_t)gather->ptr[q][t] & 15) || !gather->ptr[q][t] || ld % 4 || c0 % 8 || (t == 0 ? c0 != 0 : c0 <= gather->c0[q][t - 1]) ||
        c0 >= K0 || gather->mod[q][t] < 0)
      return TACORL_EINVAL;
    a.seg[p][t] = MlpXSeg{gather->ptr[q][t], ld, c0, gather->mod[q][t], 0};
  }
  return TACORL_OK;
}
int mlp_big_fwd(int nprob, const float* const* x, int ldx, const float* const* params, const void* const* params_bf16,
                float* const* act, const int* M, int L, const int* dims, const int* acts, const long* ybf, const long* sbf,
                const long* yout, const long* woff, const long* boff, hipStream_t st, const MlpXGather* gather, const int* gmap) {
  MlpBigFwdArgs a{};
  int maxM = 0;
  for (int p = 0; p < nprob; p++) {
    if (mlp_set_gather(a, p, gmap ? gmap[p] : p, gather, dims[0]) != TACORL_OK) return TACORL_EINVAL;
    if ((!a.nseg[p] && ((uintptr_t)x[p] & 15)) || ((uintptr_t)params[p] & 15) || ((uintptr_t)params_bf16[p] & 7) || ((uintptr_t)act[p] & 15)) return TACORL_EINVAL;
    a.x[p] = x[p]; a.params[p] = params[p]; a.pbf[p] = (const __bf16*)params_bf16[p]; a.act[p] = act[p]; a.M[p] = M[p];
    for (int l = 0; l < L; l++) { a.ybf[p][l] = ybf[p * MF_MAXL + l]; a.sbf[p][l] = sbf[p * MF_MAXL + l]; }
    a.yout[p] = yout[p];
    maxM = M[p] > maxM ? M[p] : maxM;
  }
  for (int l = 0; l < L; l++) {
    if (woff[l] % 4) return TACORL_EINVAL;
    a.woff[l] = woff[l]; a.boff[l] = boff[l]; a.dims[l] = dims[l]; a.acts[l] = acts[l];
  }
  a.dims[L] = dims[L]; a.L = L; a.ldx = ldx;
  if (maxM == 0) return TACORL_OK;
  constexpr size_t lds = (size_t)2 * 128 * XP * 2, lds_mid = (size_t)2 * 32 * XP * 2, lds_64 = (size_t)2 * 64 * XP * 2;
  static int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_big_fwd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_m64_fwd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_64) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_mid_fwd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mid) == hipSuccess) ? 0 : -1;
  if (once) return TACORL_ELAUNCH;
  const int maxMp = (maxM + 63) & ~63;  // (the blocks cover the zero rows up to a multiple of 64 too)
  {
    // tens of thousands of rows: the persistent kernel (one resident workgroup per CU, weights in registers).
    // TACORL_MLP_PERS=0: the per-block kernels below, as before
    const char* pe = getenv("TACORL_MLP_PERS");
    bool pers = mlp_rows_huge(maxM) && (pe ? atoi(pe) : 1) && (L == 3 || L == 4) && dims[0] <= 32 * PF_K0S && dims[L] <= 4;
    for (int l = 0; l + 1 < L && pers; l++) pers = acts[l] == ACT_SILU && dims[l + 1] == 256;
    if (pers) {
      { const char* de = getenv("TACORL_MLP_PERS_DBG"); a.dbg = de ? atoi(de) : 0; }
      static int once_p = (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_pers_fwd_kernel<1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS) == hipSuccess &&
                           hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_pers_fwd_kernel<2>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, PF_LDS) == hipSuccess) ? 0 : -1;
      if (once_p) return TACORL_ELAUNCH;
      static const int ncu = [] { int d = 0, n = 0; if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n < 1) n = 256; return n; }();
      // (TACORL_MLP_PERS_CUS: CUs the resident workgroups take together - fewer than all leaves room for another branch's launches)
      const char* ce = getenv("TACORL_MLP_PERS_CUS");
      int gx = (ce && atoi(ce) > 0 && atoi(ce) < ncu ? atoi(ce) : ncu) / nprob;
      gx = gx < 1 ? 1 : gx;
      gx = gx > maxMp / PF_BM ? maxMp / PF_BM : gx;
      if (L == 3) hipLaunchKernelGGL(mlp_pers_fwd_kernel<1>, dim3(gx, nprob), dim3(PF_NT), PF_LDS, st, a);
      else hipLaunchKernelGGL(mlp_pers_fwd_kernel<2>, dim3(gx, nprob), dim3(PF_NT), PF_LDS, st, a);
      return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
    }
  }
  const int rows = mlp_rows_huge(maxM) ? mlp_huge_fwd_rows() : 32;
  if (rows == 128) hipLaunchKernelGGL(mlp_big_fwd_kernel, dim3((maxMp + 127) / 128, nprob), dim3(MF_NT), lds, st, a);
  else if (rows == 64) hipLaunchKernelGGL(mlp_m64_fwd_kernel, dim3(maxMp / 64, nprob), dim3(MF_NT), lds_64, st, a);
  else hipLaunchKernelGGL(mlp_mid_fwd_kernel, dim3(maxMp / 32, nprob), dim3(MF_NT), lds_mid, st, a);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
// the input-gradient chain of many-row problems (the transposed weights wt[p] are already packed: mlp_fused_bwd mode 1)
int mlp_big_bwd(int nprob, const float* const* act, const float* const* d_out, int ldo, float* const* dz, float* const* d_x,
                int ldd, void* const* wt, const int* M, int L, const int* dims, const long* sbf, const long* dzoff,
                hipStream_t st) {
  MlpBigBwdArgs a{};
  long wtoff[MF_MAXL];
  mlp_fused_wt_elems(L, dims, wtoff);
  int maxM = 0;
  for (int p = 0; p < nprob; p++) {
    if (((uintptr_t)wt[p] | (uintptr_t)dz[p] | (uintptr_t)act[p]) & 15) return TACORL_EINVAL;
    a.d_out[p] = d_out[p]; a.act[p] = act[p]; a.wt[p] = (const __bf16*)wt[p]; a.dz[p] = dz[p];
    a.d_x[p] = d_x ? d_x[p] : nullptr; a.M[p] = M[p];
    for (int l = 0; l < L; l++) { a.sbf[p][l] = sbf[p * MF_MAXL + l]; a.dzoff[p][l] = dzoff[p * MF_MAXL + l]; }
    maxM = M[p] > maxM ? M[p] : maxM;
  }
  for (int l = 0; l < L; l++) a.wtoff[l] = wtoff[l];
  for (int l = 0; l <= L; l++) a.dims[l] = dims[l];
  a.L = L; a.ldo = ldo; a.ldd = ldd;
  if (maxM == 0) return TACORL_OK;
  constexpr size_t lds = (size_t)3 * 64 * XP * 2, lds_mid = (size_t)3 * 32 * XP * 2;
  static int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_big_bwd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_mid_bwd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mid) == hipSuccess) ? 0 : -1;
  if (once) return TACORL_ELAUNCH;
  {
    const char* pe = getenv("TACORL_MLP_PERS_BWD");  // (=0: the per-block kernels below, as before)
    bool pers = mlp_rows_huge(maxM) && (pe ? atoi(pe) : 1) && (L == 3 || L == 4) && dims[0] <= 96 && dims[L] <= 4;
    for (int l = 1; l < L && pers; l++) pers = dims[l] == 256;
    for (int p = 0; p < nprob && pers; p++) pers = !(d_x && d_x[p] && ((uintptr_t)d_x[p] & 15));
    if (pers) {
      static int once_p = (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_pers_bwd_kernel<1>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS) == hipSuccess &&
                           hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_pers_bwd_kernel<2>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS) == hipSuccess) ? 0 : -1;
      if (once_p) return TACORL_ELAUNCH;
      static const int ncu = [] { int d = 0, n = 0; if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n < 1) n = 256; return n; }();
      const int nb = ((maxM + 63) & ~63) / PF_BM;
      const char* ce = getenv("TACORL_MLP_PERS_CUS");
      int gx = (ce && atoi(ce) > 0 && atoi(ce) < ncu ? atoi(ce) : ncu) / nprob;
      gx = gx < 1 ? 1 : gx;
      gx = gx > nb ? nb : gx;
      if (L == 3) hipLaunchKernelGGL(mlp_pers_bwd_kernel<1>, dim3(gx, nprob), dim3(PF_NT), PB_LDS, st, a);
      else hipLaunchKernelGGL(mlp_pers_bwd_kernel<2>, dim3(gx, nprob), dim3(PF_NT), PB_LDS, st, a);
      return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
    }
  }
  if (mlp_rows_huge(maxM) && mlp_huge_bwd_rows() == 64) hipLaunchKernelGGL(mlp_big_bwd_kernel, dim3((maxM + 63) / 64, nprob), dim3(MF_NT), lds, st, a);
  else hipLaunchKernelGGL(mlp_mid_bwd_kernel, dim3(((maxM + 63) & ~63) / 32, nprob), dim3(MF_NT), lds_mid, st, a);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
static int mlp_big_tiles(int L) { return 1 + 2 * (L - 2); }  // layer 0: one 128-column x tile; hidden layers: two
static int mlp_big_rps(int nprob, const int* M, int L) {
  int maxMp = 0, n = 0;
  for (int p = 0; p < nprob; p++)
    if (M[p] > 0) { const int mp = (M[p] + 63) & ~63; maxMp = mp > maxMp ? mp : maxMp; n++; }
  if (!n) return BG_R;
  int nslice = 512 / (mlp_big_tiles(L) * n);
  nslice = nslice < 1 ? 1 : nslice;
  const int stages = maxMp / BG_R;
  const int sps = (stages + nslice - 1) / nslice;
  return (sps < 4 ? 4 : sps) * BG_R;
}
static int mlp_out_rpw(int nprob, const int* M) {
  int maxM = 0;
  for (int p = 0; p < nprob; p++) maxM = M[p] > maxM ? M[p] : maxM;
  return mlp_rows_huge(maxM) ? OUT_RPW : 64;
}
static long mlp_big_out_rec(int L, const int* dims) { return ((long)dims[L] * 257 + 3) & ~3L; }
size_t mlp_big_wgrad_slab_floats(int nprob, const int* M, int L, const int* dims) {
  const long rec = mlp_wgrad_record(L, dims, nullptr), orec = mlp_big_out_rec(L, dims);
  const int rps = mlp_big_rps(nprob, M, L), rpw = mlp_out_rpw(nprob, M);
  size_t tot = 0;
  for (int p = 0; p < nprob; p++)
    if (M[p] > 0) tot += (size_t)((((M[p] + 63) & ~63) + rps - 1) / rps) * rec + (size_t)((M[p] + rpw - 1) / rpw) * orec;
  return tot;
}
size_t mlp_big_xb_bytes(int M) { return (size_t)((M + 63) & ~63) * 128 * 2; }
// ybf[p*MF_MAXL + l]: float offset in act[p] of layer l's bf16 output copy (l < L-1); dz[p] + dzoff[p*MF_MAXL + l]: bf16 dZ_l;
// xb[p]: scratch of mlp_big_xb_bytes(M[p]) bytes for layer 0's input
int mlp_fused_wgrad_big(int nprob, const float* const* x, int ldx, const float* const* act, const float* const* d_out, int ldo,
                        const float* const* dz, float* const* grads, float* slab, void* const* xb, const int* M, int L,
                        const int* dims, const long* ybf, const long* dzoff, const long* woff, const long* boff, int accumulate,
                        hipStream_t st, size_t slab_floats) {
  MlpWgBigArgs a{};
  MlpXbArgs xa{};
  MlpWgOutArgs oa{};
  MlpWgReduceArgs r{}, ro{};
  a.rec = r.rec = mlp_wgrad_record(L, dims, a.sloff);
  oa.rec = ro.rec = mlp_big_out_rec(L, dims);
  r.accumulate = ro.accumulate = accumulate;
  int ng = 0;
  for (int p = 0; p < nprob; p++) ng += (grads[p] && M[p] > 0) ? 1 : 0;
  if (!ng) return 0;
  // rows per slice from EVERY problem of the call, graded or not - the set mlp_big_wgrad_slab_floats sized the slab for
  // (fewer problems would mean more, shorter slices: more records than planned); the capacity is checked below
  a.rps = mlp_big_rps(nprob, M, L);
  a.nl = L - 1;
  for (int l = 0; l <= L; l++) r.dims[l] = dims[l];
  for (int l = 0; l < L; l++) { r.sloff[l] = a.sloff[l]; r.woff[l] = woff[l]; r.boff[l] = boff[l]; }
  // the last layer's partials: a one-layer record of their own (more, finer slices than the MFMA kernel's)
  ro.dims[0] = dims[L - 1]; ro.dims[1] = dims[L]; ro.sloff[0] = 0; ro.woff[0] = woff[L - 1]; ro.boff[0] = boff[L - 1];
  a.tile0[0] = 0;
  for (int l = 0; l + 1 < L; l++) {
    a.K[l] = dims[l]; a.ldxb[l] = l == 0 ? 128 : 256;
    a.tile0[l + 1] = a.tile0[l] + (l == 0 ? 1 : 2);
  }
  xa.ldx = ldx; xa.K0 = dims[0];
  oa.NL = dims[L]; oa.ldo = ldo; oa.rpw = mlp_out_rpw(nprob, M);
  int n2 = 0, maxs = 0, maxso = 0, maxMp = 0;
  float* sp = slab;
  for (int p = 0; p < nprob; p++) {
    if (!grads[p] || M[p] <= 0) continue;
    const int Mp = (M[p] + 63) & ~63, ns = (Mp + a.rps - 1) / a.rps, nso = (M[p] + oa.rpw - 1) / oa.rpw;
    if (((uintptr_t)xb[p] | (uintptr_t)dz[p] | (uintptr_t)act[p]) & 15) return -1;
    a.Mp[n2] = Mp; a.slab[n2] = sp;
    for (int l = 0; l + 1 < L; l++) {
      a.dz[n2][l] = reinterpret_cast<const __bf16*>(dz[p] + dzoff[p * MF_MAXL + l]);
      a.xb[n2][l] = l == 0 ? reinterpret_cast<const __bf16*>(xb[p]) : reinterpret_cast<const __bf16*>(act[p] + ybf[p * MF_MAXL + l - 1]);
    }
    xa.x[n2] = x[p]; xa.xb[n2] = reinterpret_cast<__bf16*>(xb[p]); xa.M[n2] = M[p]; xa.Mp[n2] = Mp;
    r.slab[n2] = sp; r.grad[n2] = grads[p]; r.nslice[n2] = ns;
    sp += (size_t)ns * a.rec;
    oa.y[n2] = reinterpret_cast<const __bf16*>(act[p] + ybf[p * MF_MAXL + L - 2]); oa.dlast[n2] = d_out[p]; oa.slab[n2] = sp; oa.M[n2] = M[p];
    ro.slab[n2] = sp; ro.grad[n2] = grads[p]; ro.nslice[n2] = nso;
    sp += (size_t)nso * oa.rec;
    if ((size_t)(sp - slab) > slab_floats) return -2;  // the records would run past the caller's slab
    maxs = ns > maxs ? ns : maxs; maxso = nso > maxso ? nso : maxso; maxMp = Mp > maxMp ? Mp : maxMp;
    n2++;
  }
  constexpr int lds = BG_S * BG_STAGE;
  static int once = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_wgrad_big_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess ? 0 : -1;
  if (once) return -1;
  const int xblocks = (int)(((long)maxMp * 16 + 255) / 256);
  hipLaunchKernelGGL(mlp_x_to_bf16_kernel, dim3(xblocks > 4096 ? 4096 : xblocks, n2), dim3(256), 0, st, xa);
  hipLaunchKernelGGL(mlp_wgrad_big_kernel, dim3(maxs, a.tile0[L - 1], n2), dim3(64 * BG_NW), lds, st, a);
  hipLaunchKernelGGL(mlp_wgrad_out_kernel, dim3(maxso, n2), dim3(256), 0, st, oa);
  if (launch_reduce(r, 256, L - 1, n2, st) || launch_reduce(ro, 8, 1, n2, st)) return -1;
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
  if (mode == 1 && prep_deferring()) return prep_defer_wtpack(k, nprob, st);  // (tacorl_prep_batch_begin .. _end: one launch for all)
  if (mode != 2) hipLaunchKernelGGL(mlp_pack_wt_kernel, dim3(64, L, nprob), dim3(256), 0, st, k);
  if (mode == 1) return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
  constexpr size_t lds = (size_t)2 * BMF * XP * 2, lds_big = (size_t)2 * 64 * XP * 2;
  static int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_bwd_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
                     hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fused_bwd_big_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big) == hipSuccess) ? 0 : -1;
  if (once) return TACORL_ELAUNCH;
  if (maxM >= mlp_big_rows())
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
                  const long* woff, const long* boff, hipStream_t st, const MlpXGather* gather, const int* gmap) {
  MlpFwdArgs a{};
  int maxM = 0;
  for (int p = 0; p < nprob; p++) {
    if (mlp_set_gather(a, p, gmap ? gmap[p] : p, gather, dims[0]) != TACORL_OK) return TACORL_EINVAL;
    const int ns = a.nseg[p];
    if (ns) {  // (x[p] is not read)
      if ((uintptr_t)params[p] & 15) return TACORL_EINVAL;
    } else if (((uintptr_t)x[p] & 15) || ((uintptr_t)params[p] & 15)) return TACORL_EINVAL;
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
  if (maxM >= mlp_big_rows())
    hipLaunchKernelGGL(mlp_fused_fwd_big_kernel, dim3((maxM + 127) / 128, nprob), dim3(MF_NT), lds_big, st, a);
  else
    hipLaunchKernelGGL(mlp_fused_fwd_kernel, dim3((maxM + BMF - 1) / BMF, nprob), dim3(MF_NT), lds, st, a);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
