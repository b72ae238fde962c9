// y = act(x W^T + bias + addend) for the action decoder's stacked ReLU-RNN (reference
// networks/action_decoders/rnn_models.py:5-16 -> torch nn.RNN(nonlinearity="relu"), hidden 2048):
// the recurrent step (M = batch rows) and the per-layer input projection (M = batch * T rows).
//
// The generic GEMM needs split-K plus a reduce launch for this shape ([256 x 2048] x [2048 x 2048]:
// 16 us + 7.6 us per step, 31 dependent steps per forward) and exposes one global round trip per K tile.
// Here operands are bf16 in HBM and stream through an LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: no staging registers), one output tile per workgroup over the whole K -
// no split, no second launch; bias, the input-projection addend and the activation are applied in the
// epilogue, which also writes the bf16 copy that is the next step's operand.  Tile shapes in use:
// 64 x 32 / 4 stages / 4 waves (single recurrent step, BPTT step), 128 x 64 / 3 stages / 4 waves
// (sequence-wide projections), 128 x 64 / 2 stages / 8 waves (the wavefront batches of 3 - 4 problems:
// a quarter of the workgroups of the 64 x 32 variant at the same launch time; the head and tail launches of a
// wavefront, one or two problems, take 64 x 32 / 2 stages: they would fill a quarter / half of the chip otherwise).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/tacorl_hip.h"
#include "common.h"

namespace {

constexpr int RB_K = 128;
constexpr int ROW_BYTES = RB_K * 2;  // 256 B per staged row (16 chunks of 16 B)

struct RnnArgs {
  const __bf16* x;
  const __bf16* w;
  const float* bias;
  const float* addend;
  const float* mask_src;  // optional [M][N]: result kept where mask_src > 0 (ReLU derivative in BPTT)
  float* y;
  __bf16* yb;
  int M, K, N, ld_add, act;
  // optional twin rows: M2 more rows of the same problem (same W, bias, act) whose activations live in other buffers -
  // rows M .. M + M2 - 1 of the tile grid read x2 / addend2 and write y2 / yb2 (PlayLMP: the logging-only random-plan
  // pass of the action decoder rides in the real pass's launches; at B = 32 the two fill one 64-row tile)
  const __bf16* x2;
  const float* addend2;
  float* y2;
  __bf16* yb2;
  int M2;
  // optional K extension: one more K tile (RB_K columns) whose operands live in other buffers - x_ext [M][RB_K] (x2_ext for
  // the twin rows), w_ext [N][RB_K], zero padded - and a second bias.  The RNN's layer-0 step as ONE contraction over
  // [h_{t-1} | x_t]: W_hh h + W_ih x + b_hh + b_ih (torch nn.RNN's cell, rnn_models.py:5-16) without the separate input
  // projection launch and its fp32 [T B][H] addend (33.5 MB written once and read once per step at the headline shapes).
  const __bf16* x_ext;
  const __bf16* x2_ext;
  const __bf16* w_ext;
  const float* bias2;
};

// One RB_M x RB_N output tile per workgroup (4 waves stacked along M), RB_S-stage ring.
// LDS row r keeps its 16-byte chunk c at position c ^ (r & 15): a fragment read (16 lanes = 16 rows, same
// c) then spreads over all banks.  The DMA writes lane-contiguous, so the swizzle is applied to the
// *global* chunk each lane fetches.
#define RNN_MAXP 4
struct RnnBatch { RnnArgs p[RNN_MAXP]; };

template <int RB_M, int RB_N, int RB_S, int NW = 4>
__global__ __launch_bounds__(64 * NW) void rnn_gemm_kernel(RnnBatch ab, int MT, int NT, int NTX) {
  // XCD-aware tile map (1-D grid of 8 * MT * NTX * nprob workgroups): blocks b and b + 8 share an XCD and its
  // 4 MiB L2, so XCD b % 8 owns the N tiles [NTX (b % 8), NTX (b % 8 + 1)) of every problem - one eighth of
  // each weight matrix (3 x 1 MiB for the three 2048 x 2048 matrices of a wavefront launch), which then stays
  // L2-resident from one recurrent step's launch to the next instead of being re-fetched from the Infinity
  // Cache by every XCD.  Inside an XCD the M tiles of one N tile are adjacent (they share the weight rows).
  // Placement is a speed matter only: any block -> XCD assignment computes the same result.
  // NTX == 0: plain map for FEW N tiles (the output heads: 192 columns = 3 - 6 tiles; the XCD map would leave the XCDs
  // beyond the tile count idle and put every workgroup on the others): block -> (m tile fastest, n tile, problem), so the
  // M tiles spread over all XCDs and each XCD reads the whole (small) weight matrix.
  int mt, ntile, prob;
  if (NTX == 0) {
    const int jx = blockIdx.x;
    mt = jx % MT; ntile = (jx / MT) % NT; prob = jx / (MT * NT);
  } else {
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, rx = jx / MT;
    mt = jx % MT; ntile = xcd * NTX + rx % NTX; prob = rx / NTX;
    if (ntile >= NT) return;
  }
  const RnnArgs a = ab.p[prob];  // by value: keeps the fields in SGPRs (a reference re-loads them in the K loop)
  constexpr int STAGE_BYTES = (RB_M + RB_N) * ROW_BYTES;
  constexpr int DMA_PER_WAVE = (RB_M + RB_N) / 4 / NW;  // wave-instructions per stage and wave (4 rows each)
  constexpr int MI = RB_M / (16 * NW), NI = RB_N / 16;  // 16x16 tiles per wave (NW waves stacked along M)
  static_assert((RB_M + RB_N) % (4 * NW) == 0 && RB_M % (16 * NW) == 0 && RB_N % 16 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
  const int m0 = mt * RB_M, n0 = ntile * RB_N;
  const int nkm = a.K / RB_K, nk = nkm + (a.x_ext ? 1 : 0), mtot = a.M + a.M2;

  auto issue = [&](int kt, int slot) {
    unsigned char* base = lds + slot * STAGE_BYTES;
#pragma unroll
    for (int q = 0; q < DMA_PER_WAVE; q++) {
      const int row4 = (w + NW * q) * 4;  // first of the 4 rows this wave-instruction fills
      const int r = row4 + (lane >> 4), cpos = lane & 15, c = cpos ^ (r & 15);
      const int xm = m0 + r < mtot ? m0 + r : mtot - 1;  // rows past the last one are loaded (clamped) but never stored
      const __bf16* src;
      if (kt < nkm) {
        const __bf16* xrow = xm < a.M ? a.x + (long)xm * a.K : a.x2 + (long)(xm - a.M) * a.K;
        src = (r < RB_M ? xrow : a.w + (long)(n0 + r - RB_M) * a.K) + kt * RB_K + c * 8;
      } else {  // the extension tile
        const __bf16* xrow = xm < a.M ? a.x_ext + (long)xm * RB_K : a.x2_ext + (long)(xm - a.M) * RB_K;
        src = (r < RB_M ? xrow : a.w_ext + (long)(n0 + r - RB_M) * RB_K) + c * 8;
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + row4 * ROW_BYTES), 16, 0, 0);
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int nt = 0; nt < NI; nt++) acc[mi][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < RB_S - 1; s++)
    if (s < nk) issue(s, s);
  const int arow = 16 * MI * w + i;  // this lane's first activation row inside the tile
  for (int kt = 0; kt < nk; kt++) {
    // stage kt has landed once at most the RB_S-2 younger stages of this wave are outstanding
    if (kt + RB_S - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE * (RB_S - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // raw barrier (__syncthreads would also drain the younger DMA stages): every wave's share of stage kt
    // is visible and everyone is done reading stage kt-1
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + RB_S - 1 < nk) issue(kt + RB_S - 1, (kt + RB_S - 1) % RB_S);
    const unsigned char* st = lds + (kt % RB_S) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < RB_K / 32; ks++) {
      const int c = 4 * ks + g;
      bf16x8 X[MI];
#pragma unroll
      for (int mi = 0; mi < MI; mi++) {
        const int r = arow + 16 * mi;
        X[mi] = *reinterpret_cast<const bf16x8*>(st + r * ROW_BYTES + ((c ^ (r & 15)) << 4));
      }
#pragma unroll
      for (int nt = 0; nt < NI; nt++) {
        const int r = RB_M + 16 * nt + i;
        const bf16x8 Wf = *reinterpret_cast<const bf16x8*>(st + r * ROW_BYTES + ((c ^ (r & 15)) << 4));
        // weights as the A operand: D[n][m], a lane ends with 4 consecutive output columns of one row
#pragma unroll
        for (int mi = 0; mi < MI; mi++) acc[mi][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Wf, X[mi], acc[mi][nt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; mi++) {
    int m = m0 + arow + 16 * mi;
    if (m >= mtot) continue;
    const bool twin = m >= a.M;
    if (twin) m -= a.M;
    const float* addend = twin ? a.addend2 : a.addend;
    float* y = twin ? a.y2 : a.y;
    __bf16* yb = twin ? a.yb2 : a.yb;
#pragma unroll
    for (int nt = 0; nt < NI; nt++) {
      const int n = n0 + 16 * nt + 4 * g;
      f32x4 z = acc[mi][nt];
      if (a.bias) z += *reinterpret_cast<const f32x4*>(a.bias + n);
      if (a.bias2) z += *reinterpret_cast<const f32x4*>(a.bias2 + n);
      if (addend) z += *reinterpret_cast<const f32x4*>(addend + (long)m * a.ld_add + n);
#pragma unroll
      for (int r = 0; r < 4; r++) z[r] = act_apply(a.act, z[r]);
      if (a.mask_src && !twin) {
        const f32x4 ms = *reinterpret_cast<const f32x4*>(a.mask_src + (long)m * a.N + n);
#pragma unroll
        for (int r = 0; r < 4; r++) z[r] = ms[r] > 0.f ? z[r] : 0.f;
      }
      *reinterpret_cast<f32x4*>(y + (long)m * a.N + n) = z;
      if (yb) *reinterpret_cast<bf16x4*>(yb + (long)m * a.N + n) = bf16x4{(__bf16)z[0], (__bf16)z[1], (__bf16)z[2], (__bf16)z[3]};
    }
  }
}

// all problems of a batch share (M, K, N)
template <int RB_M, int RB_N, int RB_S, int NW = 4>
int launch_ring(const RnnBatch& ab, int nprob, hipStream_t st, bool plain_map = false) {
  constexpr int lds = RB_S * (RB_M + RB_N) * ROW_BYTES;
  auto kern = rnn_gemm_kernel<RB_M, RB_N, RB_S, NW>;
  static int once = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) ==
                            hipSuccess ? 0 : -1;
  if (once) return TACORL_ELAUNCH;
  const RnnArgs& a = ab.p[0];
  const int MT = (a.M + a.M2 + RB_M - 1) / RB_M, NT = a.N / RB_N, NTX = plain_map ? 0 : (NT + 7) / 8;
  hipLaunchKernelGGL(kern, dim3(plain_map ? MT * NT * nprob : 8 * MT * NTX * nprob), dim3(64 * NW), lds, st, ab, MT, NT, NTX);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
template <int RB_M, int RB_N, int RB_S>
int launch_ring(const RnnArgs& a, hipStream_t st, bool plain_map = false) {
  RnnBatch ab{};
  ab.p[0] = a;
  return launch_ring<RB_M, RB_N, RB_S>(ab, 1, st, plain_map);
}

}  // namespace

// Batched launches as 64 x 32 tiles: a 4-stage ring (96 KB of LDS, one workgroup per CU) when the whole launch is at most
// 192 workgroups (a 256-workgroup single problem of the headline step runs beside other branches, whose kernels the 96 KB
// keep off its CUs: +2.5 us there) - with few rows (PlayLMP at B = 32: M = 64 with the twin rows, 192 workgroups) the launch is a chain of
// 16 L2 round trips per workgroup and three stages in flight instead of one take 12.5 -> ~9 us off each of the step's 33
// launches (1.340 -> 1.279 ms/step, same box) - and the 2-stage ring (48 KB, three workgroups per CU) otherwise.
// TACORL_RNN_SMALL_STAGES = 2 / 3 / 4 overrides (A/B).
static int launch_small(const RnnBatch& ab, int nprob, hipStream_t st) {
  const RnnArgs& a = ab.p[0];
  const int MT = (a.M + a.M2 + 63) / 64, NT = a.N / 32, NTX = (NT + 7) / 8;
  const char* e = getenv("TACORL_RNN_SMALL_STAGES");
  const int stages = e && atoi(e) ? atoi(e) : (8 * MT * NTX * nprob <= 192 ? 4 : 2);  // (0 / unset: by the launch's size)
  if (stages == 4) return launch_ring<64, 32, 4>(ab, nprob, st);
  if (stages == 3) return launch_ring<64, 32, 3>(ab, nprob, st);
  return launch_ring<64, 32, 2>(ab, nprob, st);
}
extern "C" int tacorl_rnn_linear_supported(int M, int K, int N) {
  return M >= 1 && K >= RB_K && K % RB_K == 0 && N >= 32 && N % 32 == 0 ? 1 : 0;
}

extern "C" int tacorl_rnn_linear_fwd(const void* x_bf16, const void* w_bf16, const float* bias, const float* addend,
                                     int ld_add, float* y, void* y_bf16, int M, int K, int N, int act,
                                     tacorl_stream_t stream) {
  if (!tacorl_rnn_linear_supported(M, K, N) || (addend && ld_add % 4)) return TACORL_EINVAL;
  if (((uintptr_t)x_bf16 | (uintptr_t)w_bf16 | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)addend) & 15) return TACORL_EINVAL;
  if ((uintptr_t)y_bf16 & 7) return TACORL_EINVAL;
  RnnArgs a{(const __bf16*)x_bf16, (const __bf16*)w_bf16, bias, addend, nullptr, y, (__bf16*)y_bf16, M, K, N, ld_add, act};
  // a recurrent step (M = batch) wants many small tiles to fill the chip; the sequence-wide input projection
  // (M = batch*T) wants the larger tile (fewer re-reads of W)
  if ((long)M * N >= 256L * 64 * 128 && N % 64 == 0) return launch_ring<128, 64, 3>(a, (hipStream_t)stream);
  // Few output columns over many rows (the decoder's output heads: 3 840 x 2048 -> 192): as 64 x 32 tiles under the XCD map
  // they were 360 one-per-CU workgroups (96 KB ring) on SIX of the eight XCDs - two rounds, 35.5 us in the headline step.
  // 64 x 64 tiles with the plain map: 180 workgroups over all XCDs, one round (round 5; TACORL_RNN_HEADS_TILE=0: as before).
  const char* hte = getenv("TACORL_RNN_HEADS_TILE");  // (read per call: launches are captured once)
  const int heads_tile = hte ? atoi(hte) : 1;
  if (heads_tile == 1 && N % 64 == 0 && N / 64 < 8 && M >= 1024) return launch_ring<64, 64, 3>(a, (hipStream_t)stream, true);
  if (heads_tile == 2 && N / 32 < 8 && M >= 1024) return launch_ring<64, 32, 4>(a, (hipStream_t)stream, true);
  return launch_ring<64, 32, 4>(a, (hipStream_t)stream);
}

/* BPTT step of the ReLU-RNN: y = (x Wt^T + addend) * [mask_src > 0]  with x = dZ_t (bf16 [M][K]), Wt = W_hh^T
 * (bf16 [N][K], from tacorl_transpose_to_bf16), addend = dH_{t-1}, mask_src = h_{t-1}.  Same kernel. */
extern "C" int tacorl_rnn_linear_bwd_step(const void* x_bf16, const void* wt_bf16, const float* addend, int ld_add,
                                          const float* mask_src, float* y, void* y_bf16, int M, int K, int N,
                                          tacorl_stream_t stream) {
  if (!tacorl_rnn_linear_supported(M, K, N) || (addend && ld_add % 4)) return TACORL_EINVAL;
  if (((uintptr_t)x_bf16 | (uintptr_t)wt_bf16 | (uintptr_t)y | (uintptr_t)addend | (uintptr_t)mask_src) & 15) return TACORL_EINVAL;
  if ((uintptr_t)y_bf16 & 7) return TACORL_EINVAL;
  RnnArgs a{(const __bf16*)x_bf16, (const __bf16*)wt_bf16, nullptr, addend, mask_src, y, (__bf16*)y_bf16, M, K, N, ld_add, ACT_NONE};
  return launch_ring<64, 32, 4>(a, (hipStream_t)stream);
}

/* BPTT as a wavefront: nprob <= 4 independent y = (x Wt^T + addend) * [mask_src > 0] of one shape in ONE launch - the
 * recurrent gradient step of every layer plus the projection of the upper layer's dZ onto the lower layer's hidden
 * state (dH_{l-1}[t] = dZ_l[t] W_ih_l: Wt = W_ih_l^T, no mask) - T + 1 launches instead of L (T - 1) + a projection GEMM. */
extern "C" int tacorl_rnn_linear_bwd_batch(int nprob, const void* const* x_bf16, const void* const* wt_bf16,
                                           const float* const* addend, int ld_add, const float* const* mask_src,
                                           float* const* y, void* const* y_bf16, int M, int K, int N, tacorl_stream_t stream) {
  if (nprob < 1 || nprob > RNN_MAXP || !tacorl_rnn_linear_supported(M, K, N) || ld_add % 4) return TACORL_EINVAL;
  RnnBatch ab{};
  for (int p = 0; p < nprob; p++) {
    const float* ad = addend ? addend[p] : nullptr;
    const float* ms = mask_src ? mask_src[p] : nullptr;
    void* yb = y_bf16 ? y_bf16[p] : nullptr;
    if (((uintptr_t)x_bf16[p] | (uintptr_t)wt_bf16[p] | (uintptr_t)y[p] | (uintptr_t)ad | (uintptr_t)ms) & 15) return TACORL_EINVAL;
    if ((uintptr_t)yb & 7) return TACORL_EINVAL;
    ab.p[p] = RnnArgs{(const __bf16*)x_bf16[p], (const __bf16*)wt_bf16[p], nullptr, ad, ms, y[p], (__bf16*)yb, M, K, N, ld_add, ACT_NONE};
  }
  {
    const char* se = getenv("TACORL_RNN_SMALL_UPTO");  // (see tacorl_rnn_linear_fwd_batch)
    if (nprob <= (se ? atoi(se) : 2) && M % 64 == 0 && N % 32 == 0) return launch_small(ab, nprob, (hipStream_t)stream);
  }
  if (M % 128 == 0 && N % 64 == 0) return launch_ring<128, 64, 2, 8>(ab, nprob, (hipStream_t)stream);
  return launch_small(ab, nprob, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients of the RNN's 2048 x 2048 matrices: dW[o][k] = sum_r dZ[r][o] * x[r][k] over r = (time, batch)
// rows (reference: autograd of torch nn.RNN's W_hh / W_ih, rnn_models.py:5-16).  Both operands exist as bf16 row-major
// [R][2048] (the ring GEMMs write those copies) and the reduction index r is the SLOW one of both, so no operand has
// the layout an MFMA fragment wants - the generic wgrad GEMM re-lays them through registers (DPP quad transposes),
// splits R into slabs and reduces them: 151 + 26 us per matrix.  Here: one 128 x 128 tile of dW per workgroup (256
// workgroups for 2048 x 2048: one per CU, whole R, no slabs), 64 rows of both operands per stage streamed into a
// 2-stage LDS ring by LDS-DMA in their natural order (a row's 16-byte chunks XOR-swizzled through the choice of which
// global chunk a lane fetches), fragments by gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group
// reads 4 rows x 16 columns, each lane receives a column - see encoder_bwd_fused.hip).  x as the MFMA's A operand:
// a lane ends with 4 consecutive k of one dW row (16-byte stores).  The bias gradient rides along as one more MFMA
// against a fragment of ones.  Bound by the per-CU LDS-DMA ingest (32 KB per 64 rows), not by MFMA or LDS.
typedef __bf16 bf16x4t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tr_frag8(const unsigned char* lo, const unsigned char* hi) {
  const bf16x4t a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4t*)(lo));
  const bf16x4t b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4t*)(hi));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}
constexpr int WG_T = 128, WG_NW = 8;  // tile edge, waves

template <int WG_R, int WG_S>  // rows per stage, stages
__device__ __forceinline__ void rnn_wgrad_body(const __bf16* __restrict__ dz, int ld_dz, const __bf16* __restrict__ x, int ld_x, int R,
                                               int MT, int NT, float* __restrict__ dw, int ld_w, float* __restrict__ db,
                                               int accumulate) {
  // tile map: XCD b % 8 owns a block of m tiles x n tiles (its operand columns stay in its L2 while all of its
  // workgroups walk down r together); any assignment computes the same result
  int mt, nt;
  {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    if ((MT & 3) == 0 && (NT & 1) == 0) {
      const int mb = MT / 4, nb = NT / 2;  // block shape of an XCD: mb x nb tiles
      mt = (xcd >> 1) * mb + j / nb; nt = (xcd & 1) * nb + j % nb;
    } else {
      const int b = blockIdx.x; mt = b / NT; nt = b % NT;
    }
  }
  // blockIdx.y: row slab (R rows each) with its own output (tacorl_rnn_wgrad_slabs: few output tiles, many rows)
  dz += (long)blockIdx.y * R * ld_dz;
  x += (long)blockIdx.y * R * ld_x;
  dw += (long)blockIdx.y * MT * WG_T * ld_w;
  if (db) db += (long)blockIdx.y * MT * WG_T;
  constexpr int WG_OP_BYTES = WG_R * WG_T * 2, WG_STAGE_BYTES = 2 * WG_OP_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, l16 = lane & 15, q = l16 >> 2, pc = l16 & 3;
  const int wm = w >> 2, wn = w & 3;  // wave grid 2 (m: 64 rows of dW each) x 4 (n: 32 columns each)
  const int m0 = mt * WG_T, n0 = nt * WG_T;
  const int nk = R / WG_R;
  constexpr int DPW = 2 * WG_R / 4 / WG_NW;  // DMA wave-instructions per stage and wave (4 rows of one operand each)

  auto issue = [&](int kt, int slot) {
#pragma unroll
    for (int i = 0; i < DPW; i++) {
      const int id = w + WG_NW * i;            // operand id / (WG_R / 4), row group id % (WG_R / 4)
      const int op = id / (WG_R / 4), row4 = (id % (WG_R / 4)) * 4;
      const int r = row4 + (lane >> 4), cpos = lane & 15, c = cpos ^ (2 * (r & 7));
      const __bf16* src = (op ? x + (long)(kt * WG_R + r) * ld_x + n0 : dz + (long)(kt * WG_R + r) * ld_dz + m0) + c * 8;
      const unsigned lds_off = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)(
          lds + slot * WG_STAGE_BYTES + op * WG_OP_BYTES + row4 * 256));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src), "s"(lds_off)
                   : "memory");
    }
  };

  f32x4 acc[2][4], bacc[4];
#pragma unroll
  for (int mi = 0; mi < 4; mi++) {
    bacc[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < 2; ni++) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; j++) ones[j] = (__bf16)1.0f;
#pragma unroll
  for (int s = 0; s < WG_S - 1; s++)
    if (s < nk) issue(s, s);
  // this lane's fragment addresses inside a stage: row 4 g + q (+16), chunk (2 tile + pc / 2) ^ swizzle, half pc % 2
  const int row = 4 * g + q, sw = 2 * (row & 7), half = 8 * (pc & 1);
  int offA[4], offB[2];
#pragma unroll
  for (int mi = 0; mi < 4; mi++) offA[mi] = row * 256 + (((2 * (4 * wm + mi) + (pc >> 1)) ^ sw) << 4) + half;
#pragma unroll
  for (int ni = 0; ni < 2; ni++) offB[ni] = WG_OP_BYTES + row * 256 + (((2 * (2 * wn + ni) + (pc >> 1)) ^ sw) << 4) + half;

  for (int kt = 0; kt < nk; kt++) {
    if (kt + WG_S - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPW * (WG_S - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + WG_S - 1 < nk) issue(kt + WG_S - 1, (kt + WG_S - 1) % WG_S);
    const unsigned char* st = lds + (kt % WG_S) * WG_STAGE_BYTES;
#pragma unroll
    for (int sub = 0; sub < WG_R / 32; sub++) {
      const unsigned char* sb = st + sub * 32 * 256;
      bf16x8 A[4], B[2];
#pragma unroll
      for (int mi = 0; mi < 4; mi++) A[mi] = tr_frag8(sb + offA[mi], sb + offA[mi] + 16 * 256);
#pragma unroll
      for (int ni = 0; ni < 2; ni++) B[ni] = tr_frag8(sb + offB[ni], sb + offB[ni] + 16 * 256);
#pragma unroll
      for (int mi = 0; mi < 4; mi++) {
#pragma unroll
        for (int ni = 0; ni < 2; ni++) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B[ni], A[mi], acc[ni][mi], 0, 0, 0);
        if (db && wn == 0 && nt == 0) bacc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, A[mi], bacc[mi], 0, 0, 0);
      }
    }
  }
  // D[i][j]: i = x column inside the n tile (rows 4 g .. 4 g + 3 of the lane), j = dZ column (lane l16)
#pragma unroll
  for (int mi = 0; mi < 4; mi++) {
    const int m = m0 + 16 * (4 * wm + mi) + l16;
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
      float* o = dw + (long)m * ld_w + n0 + 16 * (2 * wn + ni) + 4 * g;
      f32x4 v = acc[ni][mi];
      if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(o) = v;
    }
    if (db && wn == 0 && nt == 0 && g == 0) db[m] = (accumulate ? db[m] : 0.f) + bacc[mi][0];
  }
}

template <int WG_R, int WG_S>
__global__ __launch_bounds__(64 * WG_NW) void rnn_wgrad_kernel(const __bf16* __restrict__ dz, int ld_dz, const __bf16* __restrict__ x,
                                                               int ld_x, int R, int MT, int NT, float* __restrict__ dw, int ld_w,
                                                               float* __restrict__ db, int accumulate) {
  rnn_wgrad_body<WG_R, WG_S>(dz, ld_dz, x, ld_x, R, MT, NT, dw, ld_w, db, accumulate);
}
// Several weight gradients of one output shape in ONE launch (blockIdx.z = problem; row counts may differ): the RNN's square
// matrices - W_hh of every layer and W_ih of the upper layers - were one launch each behind the BPTT, every one ingest-bound at
// one 64 KB workgroup per CU; together two workgroups share a CU and one's LDS-DMA runs under the other's MFMAs (round 5).
#define RNN_WG_MAXP 4
struct RnnWgBatch { const __bf16* dz[RNN_WG_MAXP]; const __bf16* x[RNN_WG_MAXP]; float* dw[RNN_WG_MAXP]; float* db[RNN_WG_MAXP]; int R[RNN_WG_MAXP]; };
template <int WG_R, int WG_S>
__global__ __launch_bounds__(64 * WG_NW) void rnn_wgrad_batch_kernel(RnnWgBatch b, int ld_dz, int ld_x, int MT, int NT, int ld_w,
                                                                     int accumulate) {
  const int p = blockIdx.z;
  rnn_wgrad_body<WG_R, WG_S>(b.dz[p], ld_dz, b.x[p], ld_x, b.R[p], MT, NT, b.dw[p], ld_w, b.db[p], accumulate);
}

constexpr int WG_RMIN = 64;
extern "C" int tacorl_rnn_wgrad_supported(int R, int M, int N) {
  return R >= WG_RMIN && R % WG_RMIN == 0 && M >= WG_T && M % WG_T == 0 && N >= WG_T && N % WG_T == 0 ? 1 : 0;
}
template <int WG_R, int WG_S>
static int launch_wgrad(const void* dz_bf16, int ld_dz, const void* x_bf16, int ld_x, int R, int M, int N, float* dw, float* db,
                        int accumulate, hipStream_t st, int slabs = 1) {
  constexpr int lds = WG_S * 2 * WG_R * WG_T * 2;
  auto kern = rnn_wgrad_kernel<WG_R, WG_S>;
  static int once = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess ? 0 : -1;
  if (once) return TACORL_ELAUNCH;
  const int MT = M / WG_T, NT = N / WG_T;
  hipLaunchKernelGGL(kern, dim3(MT * NT, slabs), dim3(64 * WG_NW), lds, st, (const __bf16*)dz_bf16, ld_dz, (const __bf16*)x_bf16, ld_x,
                     R, MT, NT, dw, N, db, accumulate);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
/* dw[M][N] (+)= dz^T x, db[M] (+)= column sums of dz;  dz bf16 [R][ld_dz] (M columns used), x bf16 [R][ld_x] (N columns) */
extern "C" int tacorl_rnn_wgrad(const void* dz_bf16, int ld_dz, const void* x_bf16, int ld_x, int R, int M, int N, float* dw,
                                float* db, int accumulate, tacorl_stream_t stream) {
  if (!tacorl_rnn_wgrad_supported(R, M, N) || ld_dz % 8 || ld_x % 8 || ld_dz < M || ld_x < N) return TACORL_EINVAL;
  if (((uintptr_t)dz_bf16 | (uintptr_t)x_bf16 | (uintptr_t)dw) & 15) return TACORL_EINVAL;
  // measured (R = 3840, 2048 x 2048, us): 64 rows x 2 stages 59.6, x 3 72.2, x 4 58.1, 128 x 2 83.8, 32 x 4 77.2 - the launch
  // moves 491 MB through LDS-DMA at 8.5 TB/s, the same chip-wide ingest rate the ring GEMM's DMA-only run reaches
  return launch_wgrad<64, 2>(dz_bf16, ld_dz, x_bf16, ld_x, R, M, N, dw, db, accumulate, (hipStream_t)stream);
}

/* n <= 4 such gradients of one (M, N) in one launch: dw[p][M][N] (+)= dz[p]^T x[p] over R[p] rows (db[p] may be NULL) */
extern "C" int tacorl_rnn_wgrad_batch(int n, const void* const* dz_bf16, int ld_dz, const void* const* x_bf16, int ld_x, const int* R,
                                      int M, int N, float* const* dw, float* const* db, int accumulate, tacorl_stream_t stream) {
  if (n < 1 || n > RNN_WG_MAXP || ld_dz % 8 || ld_x % 8 || ld_dz < M || ld_x < N) return TACORL_EINVAL;
  RnnWgBatch b{};
  for (int p = 0; p < n; p++) {
    if (!tacorl_rnn_wgrad_supported(R[p], M, N)) return TACORL_EINVAL;
    if (((uintptr_t)dz_bf16[p] | (uintptr_t)x_bf16[p] | (uintptr_t)dw[p]) & 15) return TACORL_EINVAL;
    b.dz[p] = (const __bf16*)dz_bf16[p]; b.x[p] = (const __bf16*)x_bf16[p]; b.dw[p] = dw[p]; b.db[p] = db ? db[p] : nullptr; b.R[p] = R[p];
  }
  constexpr int lds = 2 * 2 * 64 * WG_T * 2;
  auto kern = rnn_wgrad_batch_kernel<64, 2>;
  static int once = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess ? 0 : -1;
  if (once) return TACORL_ELAUNCH;
  const int MT = M / WG_T, NT = N / WG_T;
  hipLaunchKernelGGL(kern, dim3(MT * NT, 1, n), dim3(64 * WG_NW), lds, (hipStream_t)stream, b, ld_dz, ld_x, MT, NT, N, accumulate);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

// out[r][c] (+)= sum over slabs of part[s][r][c] for r < rows (slab pitch Mp rows), in slab order; the bias column likewise
__global__ __launch_bounds__(256) void wgrad_slab_sum_kernel(const float* __restrict__ part, const float* __restrict__ bpart, int slabs,
                                                             int Mp, int N, int rows, float* __restrict__ dw, float* __restrict__ db,
                                                             int accumulate) {
  const long total = (long)rows * N / 4;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total + rows; q += (long)gridDim.x * 256) {
    if (q < total) {
      f32x4 v = accumulate ? *reinterpret_cast<const f32x4*>(dw + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      for (int s_ = 0; s_ < slabs; s_++) v += *reinterpret_cast<const f32x4*>(part + (long)s_ * Mp * N + 4 * q);
      *reinterpret_cast<f32x4*>(dw + 4 * q) = v;
    } else if (db) {
      const int r = (int)(q - total);
      float v = accumulate ? db[r] : 0.f;
      for (int s_ = 0; s_ < slabs; s_++) v += bpart[(long)s_ * Mp + r];
      db[r] = v;
    }
  }
}
extern "C" size_t tacorl_rnn_wgrad_slabs_ws_bytes(int slabs, int Mp, int N) { return (size_t)slabs * Mp * ((size_t)N + 1) * 4; }
/* dw[rows][N] (+)= dz^T x and db[rows] (+)= column sums of dz for a FEW output rows and many reduction rows: dz bf16 [R][ld_dz]
 * with Mp >= rows columns used (a multiple of 128; columns >= rows hold zeros), x bf16 [R][ld_x].  R is cut into `slabs` row
 * ranges (R % (64 slabs) == 0) that run side by side - Mp / 128 x N / 128 output tiles alone would leave most of the chip
 * idle - and a second launch sums their partial results in slab order.  The action decoder's 182 x 2048 output heads
 * (reference action_decoder_logistic.py:44-52): 97 us as the generic split GEMM + slab reduce at 3 840 rows. */
extern "C" int tacorl_rnn_wgrad_slabs(const void* dz_bf16, int ld_dz, const void* x_bf16, int ld_x, int R, int Mp, int N, int rows,
                                      int slabs, float* dw, float* db, int accumulate, void* ws, size_t ws_bytes,
                                      tacorl_stream_t stream) {
  if (slabs < 1 || R % slabs || !tacorl_rnn_wgrad_supported(R / slabs, Mp, N) || rows < 1 || rows > Mp || ld_dz % 8 || ld_x % 8 ||
      ld_dz < Mp || ld_x < N)
    return TACORL_EINVAL;
  if (((uintptr_t)dz_bf16 | (uintptr_t)x_bf16 | (uintptr_t)dw | (uintptr_t)ws) & 15) return TACORL_EINVAL;
  if (ws_bytes < tacorl_rnn_wgrad_slabs_ws_bytes(slabs, Mp, N)) return TACORL_ENOMEM;
  float* part = (float*)ws;
  float* bpart = part + (size_t)slabs * Mp * N;
  const int rc = launch_wgrad<64, 2>(dz_bf16, ld_dz, x_bf16, ld_x, R / slabs, Mp, N, part, bpart, 0, (hipStream_t)stream, slabs);
  if (rc != TACORL_OK) return rc;
  const long total = (long)rows * N / 4 + rows;
  hipLaunchKernelGGL(wgrad_slab_sum_kernel, dim3((int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, part, bpart, slabs, Mp, N, rows, dw, db, accumulate);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

// dst[c][r] = bf16(src[r][c]): 32 x 32 tiles through LDS (R, C multiples of 32)
__global__ __launch_bounds__(256) void transpose_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int R, int C) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; k++) tile[ty + 8 * k][tx] = src[(long)(r0 + ty + 8 * k) * C + c0 + tx];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) dst[(long)(c0 + ty + 8 * k) * R + r0 + tx] = (__bf16)tile[tx][ty + 8 * k];
}
extern "C" int tacorl_transpose_to_bf16(const float* src, void* dst, int R, int C, tacorl_stream_t stream) {
  if (R % 32 || C % 32 || R < 32 || C < 32) return TACORL_EINVAL;
  hipLaunchKernelGGL(transpose_to_bf16_kernel, dim3(C / 32, R / 32), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, R, C);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

// Several transposes in one launch (blockIdx.y = job, blockIdx.x = 32 x 32 tile of the job; jobs with fewer tiles leave
// their surplus blocks idle): the weights-only preparation of the plan recognition's backward (8 launches of 5 us) and of
// the decoder's BPTT (3).
#define TR_MAXJ 16
struct TrBatch { const float* src[TR_MAXJ]; __bf16* dst[TR_MAXJ]; int R[TR_MAXJ], C[TR_MAXJ]; };
__global__ __launch_bounds__(256) void transpose_to_bf16_batch_kernel(TrBatch t) {
  __shared__ float tile[32][33];
  const int j = blockIdx.y, R = t.R[j], C = t.C[j], tx_n = C / 32;
  if ((int)blockIdx.x >= tx_n * (R / 32)) return;
  const float* __restrict__ src = t.src[j];
  __bf16* __restrict__ dst = t.dst[j];
  const int r0 = (blockIdx.x / tx_n) * 32, c0 = (blockIdx.x % tx_n) * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; k++) tile[ty + 8 * k][tx] = src[(long)(r0 + ty + 8 * k) * C + c0 + tx];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) dst[(long)(c0 + ty + 8 * k) * R + r0 + tx] = (__bf16)tile[tx][ty + 8 * k];
}
extern "C" int tacorl_transpose_to_bf16_batch(int n, const float* const* src, void* const* dst, const int* R, const int* C,
                                              tacorl_stream_t stream) {
  if (n < 1 || n > TR_MAXJ) return TACORL_EINVAL;
  TrBatch t{};
  int mx = 0;
  for (int j = 0; j < n; j++) {
    if (R[j] % 32 || C[j] % 32 || R[j] < 32 || C[j] < 32 || !src[j] || !dst[j]) return TACORL_EINVAL;
    t.src[j] = src[j]; t.dst[j] = (__bf16*)dst[j]; t.R[j] = R[j]; t.C[j] = C[j];
    const int tiles = (R[j] / 32) * (C[j] / 32);
    mx = tiles > mx ? tiles : mx;
  }
  hipLaunchKernelGGL(transpose_to_bf16_batch_kernel, dim3(mx, n), dim3(256), 0, (hipStream_t)stream, t);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

// The same with a row count that is no multiple of 32 and a padded destination: dst[c][r] = r < R ? bf16(src[r][c]) : 0 for
// r < ld_dst (the output heads' 182 x H weight as the K-padded W^T operand of the ring GEMM: dH = d_heads W).
__global__ __launch_bounds__(256) void transpose_pad_to_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int R,
                                                                    int C, int ld_dst) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; k++) tile[ty + 8 * k][tx] = r0 + ty + 8 * k < R ? src[(long)(r0 + ty + 8 * k) * C + c0 + tx] : 0.f;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) dst[(long)(c0 + ty + 8 * k) * ld_dst + r0 + tx] = (__bf16)tile[tx][ty + 8 * k];
}
extern "C" int tacorl_transpose_pad_to_bf16(const float* src, void* dst, int R, int C, int ld_dst, tacorl_stream_t stream) {
  if (R < 1 || C % 32 || C < 32 || ld_dst % 32 || ld_dst < R) return TACORL_EINVAL;
  hipLaunchKernelGGL(transpose_pad_to_bf16_kernel, dim3(C / 32, ld_dst / 32), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst,
                     R, C, ld_dst);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
// dst[r][c] = c < cols ? bf16(src[r][c]) : 0 for c < ld_dst: a fp32 matrix as a K-padded bf16 operand; 8 columns per thread
__global__ __launch_bounds__(256) void pad_to_bf16_kernel(const float* __restrict__ src, int ld_src, __bf16* __restrict__ dst,
                                                          int ld_dst, long rows, int cols) {
  const int per = ld_dst / 8;
  const long total = rows * per;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
    const long r = q / per;
    const int c = (int)(q - r * per) * 8;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (__bf16)(c + j < cols ? src[r * ld_src + c + j] : 0.f);
    *reinterpret_cast<bf16x8*>(dst + r * ld_dst + c) = o;
  }
}
extern "C" int tacorl_pad_to_bf16(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, tacorl_stream_t stream) {
  if (rows < 1 || cols < 1 || ld_src < cols || ld_dst < cols || ld_dst % 8 || ((uintptr_t)dst & 15)) return TACORL_EINVAL;
  const long total = rows * (ld_dst / 8);
  hipLaunchKernelGGL(pad_to_bf16_kernel, dim3((int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, src, ld_src, (__bf16*)dst, ld_dst, rows, cols);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

/* nprob <= 4 independent y = act(x W^T + b + addend) of one shape in ONE launch (blockIdx.z = problem), on a
 * 2-stage ring (48 KB of LDS per workgroup, so the workgroups of three problems are co-resident on a CU):
 * the wavefront schedule of a stacked RNN - recurrent steps of both layers and the upper layer's input
 * projection of the next step - as T + 2(L-1) launches instead of L*T + L. */
static int rnn_fwd_batch(int nprob, const void* const* x_bf16, const void* const* w_bf16, const float* const* bias,
                         const float* const* addend, int ld_add, float* const* y, void* const* y_bf16, int M, int K, int N,
                         const int* acts, const void* const* x2_bf16, const float* const* addend2, float* const* y2,
                         void* const* y2_bf16, int M2, tacorl_stream_t stream, const void* const* x_ext = nullptr,
                         const void* const* x2_ext = nullptr, const void* const* w_ext = nullptr, const float* const* bias2 = nullptr) {
  if (nprob < 1 || nprob > RNN_MAXP || !tacorl_rnn_linear_supported(M, K, N) || ld_add % 4 || M2 < 0) return TACORL_EINVAL;
  if (M2 > 0 && (!x2_bf16 || !y2)) return TACORL_EINVAL;
  RnnBatch ab{};
  for (int p = 0; p < nprob; p++) {
    const float* ad = addend ? addend[p] : nullptr;
    const float* bi = bias ? bias[p] : nullptr;
    void* yb = y_bf16 ? y_bf16[p] : nullptr;
    if (((uintptr_t)x_bf16[p] | (uintptr_t)w_bf16[p] | (uintptr_t)y[p] | (uintptr_t)bi | (uintptr_t)ad) & 15) return TACORL_EINVAL;
    if ((uintptr_t)yb & 7) return TACORL_EINVAL;
    ab.p[p] = RnnArgs{(const __bf16*)x_bf16[p], (const __bf16*)w_bf16[p], bi, ad, nullptr, y[p], (__bf16*)yb, M, K, N, ld_add, acts[p]};
    if (x_ext && x_ext[p]) {  // K extension of this problem
      if (!w_ext || !w_ext[p] || (((uintptr_t)x_ext[p] | (uintptr_t)w_ext[p] | (uintptr_t)(bias2 ? bias2[p] : nullptr)) & 15)) return TACORL_EINVAL;
      if (M2 > 0 && x2_bf16[p] && (!x2_ext || !x2_ext[p] || ((uintptr_t)x2_ext[p] & 15))) return TACORL_EINVAL;
      ab.p[p].x_ext = (const __bf16*)x_ext[p]; ab.p[p].w_ext = (const __bf16*)w_ext[p];
      ab.p[p].x2_ext = M2 > 0 && x2_ext ? (const __bf16*)x2_ext[p] : nullptr;
      ab.p[p].bias2 = bias2 ? bias2[p] : nullptr;
    }
    if (M2 > 0 && x2_bf16[p]) {  // (a problem without twin rows: x2[p] == NULL)
      const float* ad2 = addend2 ? addend2[p] : nullptr;
      void* yb2 = y2_bf16 ? y2_bf16[p] : nullptr;
      if (!y2[p] || (((uintptr_t)x2_bf16[p] | (uintptr_t)y2[p] | (uintptr_t)ad2) & 15) || ((uintptr_t)yb2 & 7)) return TACORL_EINVAL;
      ab.p[p].x2 = (const __bf16*)x2_bf16[p]; ab.p[p].addend2 = ad2; ab.p[p].y2 = y2[p]; ab.p[p].yb2 = (__bf16*)yb2;
      ab.p[p].M2 = M2;
    }
  }
  if (M2 > 0) {  // the tile grid is sized by the first problem: give every problem of the launch the same row count
    for (int p = 0; p < nprob; p++)
      if (ab.p[p].M2 != M2) return TACORL_EINVAL;
    M += M2;     // (tile choice below: by the rows of the launch)
  }
  // 128 x 64 tiles, 8 waves (two per SIMD): the launch itself takes as long as with 64 x 32 tiles and 4 waves
  // (20 us for three problems; a third stage changes nothing), but it is 192 workgroups instead of 768 and the
  // step's other branches get through beside it: 1.106 -> 1.089 ms/step
  static const int tile = [] { const char* e = getenv("TACORL_RNN_TILE"); return e ? atoi(e) : 0; }();  // A/B switch
  if (tile == 1 && M % 128 == 0 && N % 128 == 0) return launch_ring<128, 128, 2, 8>(ab, nprob, (hipStream_t)stream);
  if (tile == 3 && M % 64 == 0 && N % 128 == 0) return launch_ring<64, 128, 2, 4>(ab, nprob, (hipStream_t)stream);
  // the first two and the last two launches of a wavefront hold one or two problems: 64 / 128 workgroups of 128 x 64
  // tiles - a quarter / half of the chip, each streaming 768 KB; as 64 x 32 tiles they are 256 / 512 workgroups streaming
  // 384 KB each.  Same-process A/B on the headline step: small tiles for nprob <= 1 / <= 2 / <= 3: -8.5 / -16 / +14 us
  // (three problems as 768 small workgroups crowd the step's other branches).  TACORL_RNN_SMALL_UPTO overrides.
  const char* se = getenv("TACORL_RNN_SMALL_UPTO");
  const int small_upto = se ? atoi(se) : 2;  // problems per launch up to which the small tile is used
  if (nprob <= small_upto && M % 64 == 0 && N % 32 == 0) return launch_small(ab, nprob, (hipStream_t)stream);
  {
    // twin launches of >= 512 rows: three problems as 128 x 64 tiles are 384 workgroups of 96 KB LDS - one and a half
    // rounds over the chip; as 128 x 128 tiles they are 192, each streaming 1 MB instead of 768 KB for twice the outputs
    const char* tw = getenv("TACORL_RNN_TWIN_WIDE");
    if (M2 > 0 && (tw ? atoi(tw) : 1) && nprob > 2 && M >= 512 && M % 128 == 0 && N % 128 == 0)
      return launch_ring<128, 128, 2, 8>(ab, nprob, (hipStream_t)stream);
  }
  if (M % 128 == 0 && N % 64 == 0) return launch_ring<128, 64, 2, 8>(ab, nprob, (hipStream_t)stream);
  return launch_small(ab, nprob, (hipStream_t)stream);
}
extern "C" int tacorl_rnn_linear_fwd_batch(int nprob, const void* const* x_bf16, const void* const* w_bf16,
                                           const float* const* bias, const float* const* addend, int ld_add,
                                           float* const* y, void* const* y_bf16, int M, int K, int N, const int* acts,
                                           tacorl_stream_t stream) {
  return rnn_fwd_batch(nprob, x_bf16, w_bf16, bias, addend, ld_add, y, y_bf16, M, K, N, acts, nullptr, nullptr, nullptr, nullptr, 0,
                       stream);
}
/* The same launch with twin rows: every problem p additionally computes y2[p] = act(x2[p] W[p]^T + b[p] + addend2[p]) for M2
 * more rows whose activations live in other buffers - same weights, one pass over them (PlayLMP.training_step: the
 * logging-only random-plan pass of the action decoder, play_lmp_for_rl.py:243-252, rides in the real pass's launches). */
extern "C" int tacorl_rnn_linear_fwd_batch_twin(int nprob, const void* const* x_bf16, const void* const* x2_bf16,
                                                const void* const* w_bf16, const float* const* bias,
                                                const float* const* addend, const float* const* addend2, int ld_add,
                                                float* const* y, float* const* y2, void* const* y_bf16, void* const* y2_bf16,
                                                int M, int M2, int K, int N, const int* acts, tacorl_stream_t stream) {
  return rnn_fwd_batch(nprob, x_bf16, w_bf16, bias, addend, ld_add, y, y_bf16, M, K, N, acts, x2_bf16, addend2, y2, y2_bf16, M2,
                       stream);
}
/* The same launch with an optional K extension per problem (x_ext[p] != NULL): y[p] = act(x[p] W[p]^T + x_ext[p] w_ext[p]^T + b[p] +
 * bias2[p] + addend[p]) with x_ext [M][128] (x2_ext [M2][128] for the twin rows) and w_ext [N][128] bf16, zero padded beyond the
 * real input width - the RNN cell of layer 0 as one contraction over [h | x] (reference rnn_models.py:5-16: torch nn.RNN,
 * h_t = relu(W_ih x_t + b_ih + W_hh h_{t-1} + b_hh)); M2 = 0 and NULL twin arrays: no twin rows. */
extern "C" int tacorl_rnn_linear_fwd_batch_ext(int nprob, const void* const* x_bf16, const void* const* x2_bf16,
                                               const void* const* w_bf16, const float* const* bias, const float* const* addend,
                                               const float* const* addend2, int ld_add, float* const* y, float* const* y2,
                                               void* const* y_bf16, void* const* y2_bf16, int M, int M2, int K, int N,
                                               const int* acts, const void* const* x_ext, const void* const* x2_ext,
                                               const void* const* w_ext, const float* const* bias2, tacorl_stream_t stream) {
  return rnn_fwd_batch(nprob, x_bf16, w_bf16, bias, addend, ld_add, y, y_bf16, M, K, N, acts, x2_bf16, addend2, y2, y2_bf16, M2,
                       stream, x_ext, x2_ext, w_ext, bias2);
}
