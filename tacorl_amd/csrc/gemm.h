// One LDS-tiled MFMA contraction template for every dense op on the hot path.
//
//   C[p][m][n] = sum_{r in split s} A_p(m, r) * B_p(n, r)        p = problem, s = R-split
//
// Operands come from *loader functors* (implicit im2col, transposed weights, padded
// gathers ...) and results leave through an *epilogue functor* (bias+activation,
// activation-derivative masks, split-K slabs ...).  A loader exposes a logical 2-D
// matrix [i][j] that is contiguous along j in memory and returns 4 consecutive j:
//     load(p, i, j, v[4])        (zero-filled out of range)
// TA/TB say whether the reduction index is j (false: rows are M/N, staged as-is) or
// i (true: the matrix is [R][M] / [R][N] in memory and is transposed while staged).
//
// Tile: BM x BN x 32, 256 threads = 4 waves stacked along M, each wave owning
// (BM/64) x (BN/16) 16x16 MFMA tiles.  Global loads for tile t+1 are issued into
// registers before the MFMAs of tile t (register double buffering, one LDS buffer).
#pragma once
#include "common.h"

#define GEMM_MAXP 16
#define GEMM_BK 32

struct GemmArgs {
  int nprob;
  int nsplit;
  int N;             // common to all problems
  int M[GEMM_MAXP];  // rows per problem
  int R[GEMM_MAXP];  // reduction length per problem
};

// In-place 4x4 transpose across the 4 lanes of a quad: lane q (= lane & 3) enters with row q of a 4x4
// block in v[0..3] and leaves with column q.  Two DPP quad_perm exchanges (lane^1, lane^2), no LDS traffic.
// Used when the reduction index is the slow one in memory ([R][M] operands of the weight gradients): the
// 4 lanes of a quad load 4 consecutive r for the same 4 m, and after the transpose each lane writes 4
// consecutive r of one m with a single LDS store (instead of 4 scalar stores with a stride).
__device__ __forceinline__ float dpp_xor1(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ void quad_transpose(float (&v)[4], int q) {
  const bool o1 = q & 1, o2 = q & 2;
  const float x0 = dpp_xor1(o1 ? v[0] : v[1]), x1 = dpp_xor1(o1 ? v[2] : v[3]);
  if (o1) { v[0] = x0; v[2] = x1; } else { v[1] = x0; v[3] = x1; }
  const float y0 = dpp_xor2(o2 ? v[0] : v[2]), y1 = dpp_xor2(o2 ? v[1] : v[3]);
  if (o2) { v[0] = y0; v[1] = y1; } else { v[2] = y0; v[3] = y1; }
}

template <class L, class = void> struct has_row_state { static constexpr bool value = false; };
template <class L> struct has_row_state<L, decltype((void)L::ROW_STATE)> { static constexpr bool value = L::ROW_STATE; };

template <class L, class = void> struct has_col_state { static constexpr bool value = false; };
template <class L> struct has_col_state<L, decltype((void)L::COL_STATE)> { static constexpr bool value = L::COL_STATE; };

template <class Atom, class LA, class LB, bool TA, bool TB, class Epi, int BM, int BN, int BKT = 0>
__global__ __launch_bounds__(256) void gemm_kernel(LA la, LB lb, Epi epi, GemmArgs g) {
  typedef typename Atom::elem T;
  constexpr int BK = BKT ? BKT : Atom::BK;
  constexpr int LD = BK + Atom::PAD;
  constexpr int MI = BM / 64;
  constexpr int NI = BN / 16;
  constexpr int ACH = BM * BK / 4 / 256;  // float4 chunks per thread, A tile
  constexpr int BCH = (BN * BK / 4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) T As[BM * LD];
  __shared__ __attribute__((aligned(16))) T Bs[BN * LD];

  const int p = blockIdx.z / g.nsplit;
  const int s = blockIdx.z % g.nsplit;
  const int M = g.M[p], N = g.N, R = g.R[p];
  // n tiles are the fast grid index: workgroups that run together write neighbouring column blocks of the SAME rows
  // (whole output rows per DRAM page instead of 256-byte pieces 128 rows apart) and share the A tile in L2
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (m0 >= M) return;
  constexpr int RQ = BK > 64 ? BK : 64;  // split boundaries: multiple of every tile depth in use
  int rs = ((R + g.nsplit - 1) / g.nsplit + RQ - 1) / RQ * RQ;
  const int r_begin = s * rs;
  const int r_end = min(R, r_begin + rs);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  float ra[ACH][4], rb[BCH][4];
  // (loaders with a per-row state - implicit im2col - decompose this thread's ACH rows once, not per K tile)
  auto arows = [&]() {
    if constexpr (!TA && has_row_state<LA>::value) {
      struct R_ { typename LA::Row r[ACH]; } o;
#pragma unroll
      for (int i = 0; i < ACH; i++) o.r[i] = la.row(p, m0 + (tid + i * 256) / (BK / 4));
      return o;
    } else {
      return 0;
    }
  }();
  // (transposed operands of loaders with a column / pixel state - implicit im2col in the weight gradients: this thread's k quad is
  // the same for every tile, its ACH pixels step by BK from one fetch to the next - fetch() is called with r_begin, r_begin + BK, ...)
  auto acols = [&]() {
    if constexpr (TA && has_col_state<LA>::value) {
      static_assert(64 % (BM / 4) == 0, "a thread keeps one k quad over its chunks");
      struct C_ { typename LA::Col c; typename LA::Pix x[ACH]; } o;
      o.c = la.col(p, m0 + ((tid >> 2) % (BM / 4)) * 4);
#pragma unroll
      for (int i = 0; i < ACH; i++) o.x[i] = la.pix(p, r_begin + ((tid + i * 256) / BM) * 4 + (tid & 3));
      return o;
    } else {
      return 0;
    }
  }();
  auto fetch = [&](int r0) {
#pragma unroll
    for (int i = 0; i < ACH; i++) {
      int c = tid + i * 256;
      if constexpr (TA && has_col_state<LA>::value) { la.load_cp(acols.c, acols.x[i], ra[i]); la.advance(acols.x[i], BK); }
      else if constexpr (!TA && has_row_state<LA>::value) la.load_row(arows.r[i], r0 + (c % (BK / 4)) * 4, ra[i]);
      else if (!TA) la.load(p, m0 + c / (BK / 4), r0 + (c % (BK / 4)) * 4, ra[i]);
      else     la.load(p, r0 + (c / BM) * 4 + (c & 3), m0 + ((c >> 2) % (BM / 4)) * 4, ra[i]);  // quad = 4 rows r, same 4 m
    }
#pragma unroll
    for (int i = 0; i < BCH; i++) {
      int c = tid + i * 256;
      if (BN * BK / 4 % 256 != 0 && c >= BN * BK / 4) break;
      if (!TB) lb.load(p, n0 + c / (BK / 4), r0 + (c % (BK / 4)) * 4, rb[i]);
      else     lb.load(p, r0 + (c / BN) * 4 + (c & 3), n0 + ((c >> 2) % (BN / 4)) * 4, rb[i]);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int i = 0; i < ACH; i++) {
      int c = tid + i * 256;
      if (!TA) {
        Atom::st4(&As[(c / (BK / 4)) * LD + (c % (BK / 4)) * 4], ra[i]);
      } else {  // 4x4 transpose inside the lane quad (DPP), then one 4-element store along r
        quad_transpose(ra[i], tid & 3);
        Atom::st4(&As[(((c >> 2) % (BM / 4)) * 4 + (c & 3)) * LD + (c / BM) * 4], ra[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < BCH; i++) {
      int c = tid + i * 256;
      if (BN * BK / 4 % 256 != 0 && c >= BN * BK / 4) break;
      if (!TB) {
        Atom::st4(&Bs[(c / (BK / 4)) * LD + (c % (BK / 4)) * 4], rb[i]);
      } else {
        quad_transpose(rb[i], tid & 3);
        Atom::st4(&Bs[(((c >> 2) % (BN / 4)) * 4 + (c & 3)) * LD + (c / BN) * 4], rb[i]);
      }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NI; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (r_begin < r_end) fetch(r_begin);
  for (int r0 = r_begin; r0 < r_end; r0 += BK) {
    stash();
    __syncthreads();
    if (r0 + BK < r_end) fetch(r0 + BK);
    const T* ap = &As[(wave * MI * 16 + (lane & 15)) * LD + (lane >> 4) * Atom::KPACK];
    const T* bp = &Bs[(lane & 15) * LD + (lane >> 4) * Atom::KPACK];
#pragma unroll
    for (int kk = 0; kk < BK; kk += Atom::KSTEP) {
      typename Atom::frag a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; i++) a[i] = Atom::ld(ap + i * 16 * LD + kk);
#pragma unroll
      for (int j = 0; j < NI; j++) b[j] = Atom::ld(bp + j * 16 * LD + kk);
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++) {
          // epilogues whose output is contiguous along n take the B tile as the MFMA's first operand: D^T, i.e. a lane
          // ends with 4 consecutive columns of ONE row (a 16-byte store / mask load / addend load instead of four
          // scalar ones with 64-bit address arithmetic each).  Same products, same accumulation order over k.
          if constexpr (Epi::VEC) acc[i][j] = Atom::mma(b[j], a[i], acc[i][j]);
          else acc[i][j] = Atom::mma(a[i], b[j], acc[i][j]);
        }
    }
    __syncthreads();
  }
  if constexpr (Epi::VEC) {
#pragma unroll
    for (int i = 0; i < MI; i++) {
      const int m = m0 + wave * MI * 16 + i * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < NI; j++) {
        const int n = n0 + j * 16 + (lane >> 4) * 4;
        if (m >= M) continue;
        if (epi.vec && n + 3 < N) {
          epi.store4(p, s, m, n, acc[i][j]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (n + q < N) epi.store(p, s, m, n + q, acc[i][j][q]);
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
      for (int j = 0; j < NI; j++) {
        int n = n0 + j * 16 + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          int m = m0 + wave * MI * 16 + i * 16 + (lane >> 4) * 4 + q;
          if (m < M && n < N) epi.store(p, s, m, n, acc[i][j][q]);
        }
      }
  }
}

// Host-side launcher: picks the tile for N and the atom for the compute dtype.
enum { DT_F32 = 0, DT_BF16 = 1 };

// DEEPK: also instantiate a 128-deep K tile (bf16 only) for reductions >= 96 per split.  Measured on the
// step's skinny GEMMs (M 256-4096, K 64-272): no gain - a single-K-tile GEMM already costs 13 us of
// launch + dependent round trips (kernarg -> operands -> bias), so no caller enables it.
template <class LA, class LB, bool TA, bool TB, class Epi, bool DEEPK = false>
static int gemm_launch(const LA& la, const LB& lb, const Epi& epi, const GemmArgs& g, int compute_dtype,
                       hipStream_t st) {
  int maxM = 0, maxR = 0;
  for (int i = 0; i < g.nprob; i++) { maxM = g.M[i] > maxM ? g.M[i] : maxM; maxR = g.R[i] > maxR ? g.R[i] : maxR; }
  if (maxM == 0 || g.N == 0) return TACORL_OK;
  if constexpr (DEEPK) {
    if (compute_dtype == DT_BF16 && (maxR + g.nsplit - 1) / g.nsplit >= 96) {
#define GEMM_GO_DEEP(BN_)                                                                                  \
  do {                                                                                                     \
    dim3 grid(cdiv(g.N, BN_), cdiv(maxM, 128), g.nprob * g.nsplit);                                        \
    hipLaunchKernelGGL((gemm_kernel<AtomBF16, LA, LB, TA, TB, Epi, 128, BN_, 128>), grid, dim3(256), 0, st, \
                       la, lb, epi, g);                                                                    \
  } while (0)
      if (g.N > 32) GEMM_GO_DEEP(64);
      else if (g.N > 16) GEMM_GO_DEEP(32);
      else GEMM_GO_DEEP(16);
#undef GEMM_GO_DEEP
      return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
    }
  }
#define GEMM_GO(ATOM, BM_, BN_)                                                              \
  do {                                                                                       \
    dim3 grid(cdiv(g.N, BN_), cdiv(maxM, BM_), g.nprob * g.nsplit);                          \
    hipLaunchKernelGGL((gemm_kernel<ATOM, LA, LB, TA, TB, Epi, BM_, BN_>), grid, dim3(256), \
                       0, st, la, lb, epi, g);                                               \
  } while (0)
#define GEMM_PICK(ATOM)                        \
  do {                                         \
    if (g.N > 32) GEMM_GO(ATOM, 128, 64);      \
    else if (g.N > 16) GEMM_GO(ATOM, 128, 32); \
    else GEMM_GO(ATOM, 128, 16);               \
  } while (0)
  if (compute_dtype == DT_BF16) GEMM_PICK(AtomBF16);
  else GEMM_PICK(AtomF32);
#undef GEMM_PICK
#undef GEMM_GO
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
