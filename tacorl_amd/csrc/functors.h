// Loader / epilogue functors plugged into gemm_kernel (see gemm.h for the contract).
#pragma once
#include "gemm.h"

// ------------------------------------------------------------------ loaders
// Plain row-major fp32 matrix [rows][cols], leading dimension ld.  ones_col: a virtual
// extra column (index == cols) that reads 1.0 - folds the bias gradient into wgrad.
struct RowMajorLoader {
  const float* ptr[GEMM_MAXP];
  int rows[GEMM_MAXP];
  int cols, ld, vec, ones_col;
  __device__ __forceinline__ void load(int p, int i, int j, float (&v)[4]) const {
    const int nr = rows[p];
    if (i >= nr) { v[0] = v[1] = v[2] = v[3] = 0.f; return; }
    const float* q = ptr[p] + (long)i * ld + j;
    if (vec && j + 3 < cols) { load4_as_float<float>(q, v); return; }
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] = (j + t < cols) ? q[t] : ((ones_col && j + t == cols) ? 1.f : 0.f);
  }
};

struct ConvGeom {
  int H, W, C, KH, KW, S, CO, OH, OW;
};

// Implicit im2col of an NHWC tensor: logical [m = (img, oy, ox)][k = (ky, kx, ci)].
// (kx, ci) is one contiguous run of KW*C elements in memory (requires KW*C % 4 == 0).
template <typename InT>
struct ConvColLoader {
  const InT* ptr[GEMM_MAXP];
  int rows[GEMM_MAXP];  // n_img * OH * OW
  ConvGeom g;
  int vec, ones_col;
  __device__ __forceinline__ void load(int p, int m, int k, float (&v)[4]) const {
    const int K = g.KH * g.KW * g.C;
    if (m >= rows[p] || k >= K) {
#pragma unroll
      for (int t = 0; t < 4; t++) v[t] = (ones_col && m < rows[p] && k + t == K) ? 1.f : 0.f;
      return;
    }
    const int px = g.OH * g.OW, run = g.KW * g.C;
    const int img = m / px, pp = m - img * px, oy = pp / g.OW, ox = pp - oy * g.OW;
    const int ky = k / run, r = k - ky * run;
    const InT* q = ptr[p] + (((long)img * g.H + oy * g.S + ky) * g.W + ox * g.S) * g.C + r;
    if (vec) { load4_as_float<InT>(q, v); return; }
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] = (float)q[t];
  }
  // Row state (round 6).  In a forward convolution (TA = false) a thread of gemm_kernel stages the same few im2col rows m for
  // every K tile: the pixel decomposition m -> (image, oy, ox) - two integer divisions and a 64-bit address - is done ONCE
  // per row here, and load_row() is left with k -> (ky, offset in the row's run).  The per-layer conv forward of the
  // geometries the fused encoder does not take (150 x 200: 3.4 ms/step, conv1 alone 0.82 ms at 72 TFLOP/s) was bound by
  // exactly that index arithmetic: ~40 VALU instructions in front of every 4-element load.
  static constexpr bool ROW_STATE = true;
  struct Row { const InT* base; int valid; };
  __device__ __forceinline__ Row row(int p, int m) const {
    Row r;
    r.valid = m < rows[p];
    const int px = g.OH * g.OW, mc = r.valid ? m : 0;
    const int img = mc / px, pp = mc - img * px, oy = pp / g.OW, ox = pp - oy * g.OW;
    r.base = ptr[p] + (((long)img * g.H + oy * g.S) * g.W + ox * g.S) * g.C;
    return r;
  }
  __device__ __forceinline__ void load_row(const Row& rw, int k, float (&v)[4]) const {
    const int K = g.KH * g.KW * g.C;
    if (!rw.valid || k >= K) {
#pragma unroll
      for (int t = 0; t < 4; t++) v[t] = (ones_col && rw.valid && k + t == K) ? 1.f : 0.f;
      return;
    }
    const int run = g.KW * g.C, ky = k / run, r = k - ky * run;
    const InT* q = rw.base + ky * g.W * g.C + r;
    if (vec) { load4_as_float<InT>(q, v); return; }
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] = (float)q[t];
  }
  // Column / pixel state (round 6) for the weight gradients (TA = true: the im2col matrix is the [R = pixel][M = k] operand).
  // There a thread of gemm_kernel keeps ONE k quad for the whole launch and walks down the pixels BK at a time: k -> (ky,
  // offset in the run) is decomposed once (col), a pixel once (pix), and advance() steps the pixel by d with adds and carries -
  // no integer division in the staging loop (the per-layer conv backward of 150 x 200: 183 + 172 us of weight gradients).
  static constexpr bool COL_STATE = true;
  struct Col { int off, k; };              // off < 0: k >= K (zero / ones column)
  struct Pix { const InT* base; int ox, oy, left; };  // left = valid rows from this one on (<= 0: beyond the problem)
  __device__ __forceinline__ Col col(int, int k) const {
    const int K = g.KH * g.KW * g.C, run = g.KW * g.C;
    Col c;
    c.k = k;
    if (k >= K) { c.off = -1; return c; }
    const int ky = k / run;
    c.off = ky * g.W * g.C + (k - ky * run);
    return c;
  }
  __device__ __forceinline__ Pix pix(int p, int m) const {
    Pix x;
    x.left = rows[p] - m;
    const int px = g.OH * g.OW, mc = x.left > 0 ? m : 0;
    const int img = mc / px, pp = mc - img * px;
    x.oy = pp / g.OW; x.ox = pp - x.oy * g.OW;
    x.base = ptr[p] + (((long)img * g.H + x.oy * g.S) * g.W + x.ox * g.S) * g.C;
    return x;
  }
  __device__ __forceinline__ void advance(Pix& x, int d) const {
    x.left -= d;
    x.ox += d;
    x.base += d * g.S * g.C;
    while (x.ox >= g.OW) { x.ox -= g.OW; x.oy++; x.base += (g.S * g.W - g.OW * g.S) * g.C; }
    while (x.oy >= g.OH) { x.oy -= g.OH; x.base += (g.H - g.OH * g.S) * g.W * g.C; }
  }
  __device__ __forceinline__ void load_cp(const Col& c, const Pix& x, float (&v)[4]) const {
    if (x.left <= 0 || c.off < 0) {
      const int K = g.KH * g.KW * g.C;
#pragma unroll
      for (int t = 0; t < 4; t++) v[t] = (ones_col && x.left > 0 && c.k + t == K) ? 1.f : 0.f;
      return;
    }
    const InT* q = x.base + c.off;
    if (vec) { load4_as_float<InT>(q, v); return; }
#pragma unroll
    for (int t = 0; t < 4; t++) v[t] = (float)q[t];
  }
};

// Conv backward-data as a gather.  Problem index = net * S*S + (py*S + px): one GEMM per
// input-pixel parity class.  Logical A[m = (img, y2, x2)][k' = (a, b, co)] =
// dOut[img][y2 - a][x2 - b][co]  (zero outside), where iy = S*y2 + py, ky = py + S*a.
struct ConvDgradALoader {
  const float* ptr[GEMM_MAXP];  // dOut (already multiplied by the activation mask), NHWC
  int n_img[GEMM_MAXP];
  ConvGeom g;
  int TA, TB_;  // taps per axis = ceil(KH / S)
  __device__ __forceinline__ void load(int p, int m, int k, float (&v)[4]) const {
    const int S = g.S, cls = p % (S * S), py = cls / S, px = cls % S;
    const int nh = (g.H - py + S - 1) / S, nw = (g.W - px + S - 1) / S;
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (m >= n_img[p] * nh * nw || k >= TA * TB_ * g.CO) return;
    const int img = m / (nh * nw), pp = m - img * nh * nw, y2 = pp / nw, x2 = pp - y2 * nw;
    const int tap = k / g.CO, co = k - tap * g.CO, a = tap / TB_, b = tap - a * TB_;
    const int oy = y2 - a, ox = x2 - b;
    if (py + S * a >= g.KH || px + S * b >= g.KW || oy < 0 || oy >= g.OH || ox < 0 || ox >= g.OW) return;
    load4_as_float<float>(ptr[p] + (((long)img * g.OH + oy) * g.OW + ox) * g.CO + co, v);
  }
  // Row state (round 6, as ConvColLoader's): a thread of gemm_kernel stages the same few rows m = (img, y2, x2) of its parity
  // class for every K tile - the class geometry and the row's decomposition (four integer divisions) once per launch, the
  // per-load work is k -> (tap, co) and the halo test.  The per-layer conv backward of 150 x 200 spent 222 + 138 us here.
  static constexpr bool ROW_STATE = true;
  struct Row { const float* base; int y2, x2, py, px; };  // base: dOut[img][y2][x2][0]; y2 < 0: no such row
  __device__ __forceinline__ Row row(int p, int m) const {
    const int S = g.S, cls = p % (S * S);
    Row r;
    r.py = cls / S; r.px = cls % S;
    const int nh = (g.H - r.py + S - 1) / S, nw = (g.W - r.px + S - 1) / S;
    if (m >= n_img[p] * nh * nw) { r.y2 = -1; r.x2 = 0; r.base = ptr[p]; return r; }
    const int img = m / (nh * nw), pp = m - img * nh * nw;
    r.y2 = pp / nw; r.x2 = pp - r.y2 * nw;
    r.base = ptr[p] + (((long)img * g.OH + r.y2) * g.OW + r.x2) * g.CO;
    return r;
  }
  __device__ __forceinline__ void load_row(const Row& rw, int k, float (&v)[4]) const {
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (rw.y2 < 0 || k >= TA * TB_ * g.CO) return;
    const int S = g.S;
    const int tap = k / g.CO, co = k - tap * g.CO, a = tap / TB_, b = tap - a * TB_;
    const int oy = rw.y2 - a, ox = rw.x2 - b;
    if (rw.py + S * a >= g.KH || rw.px + S * b >= g.KW || oy < 0 || oy >= g.OH || ox < 0 || ox >= g.OW) return;
    load4_as_float<float>(rw.base - (a * g.OW + b) * g.CO + co, v);
  }
};
// Matching weights: logical [i = k' = (a, b, co)][j = ci] = W[co][py + S a][px + S b][ci].
struct ConvDgradWLoader {
  const float* ptr[GEMM_MAXP];  // W [CO][KH][KW][C]
  ConvGeom g;
  int TA, TB_;
  __device__ __forceinline__ void load(int p, int k, int ci, float (&v)[4]) const {
    const int S = g.S, cls = p % (S * S), py = cls / S, px = cls % S;
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (k >= TA * TB_ * g.CO || ci >= g.C) return;
    const int tap = k / g.CO, co = k - tap * g.CO, a = tap / TB_, b = tap - a * TB_;
    const int ky = py + S * a, kx = px + S * b;
    if (ky >= g.KH || kx >= g.KW) return;
    load4_as_float<float>(ptr[p] + (((long)co * g.KH + ky) * g.KW + kx) * g.C + ci, v);
  }
};

// ---------------------------------------------------------------- epilogues
// y = act(acc + bias[n]); optional pre-activation copy z.
struct BiasActStore {
  const float* bias[GEMM_MAXP];
  const float* addend[GEMM_MAXP];  // optional [M][N] matrix added before the activation (RNN input term)
  float* y[GEMM_MAXP];
  float* z[GEMM_MAXP];
  int ld, act, ld_add;
  int vec;  // host-set: every pointer 16-byte aligned and every leading dimension a multiple of 4 -> store4 is legal
  static constexpr bool VEC = true;  // contiguous along n: the kernel hands a lane 4 consecutive columns of one row
  __device__ __forceinline__ void store(int p, int, int m, int n, float acc) const {
    float zz = acc + (bias[p] ? bias[p][n] : 0.f);
    if (addend[p]) zz += addend[p][(long)m * ld_add + n];
    long o = (long)m * ld + n;
    if (z[p]) z[p][o] = zz;
    y[p][o] = act_apply(act, zz);
  }
  __device__ __forceinline__ void store4(int p, int, int m, int n, f32x4 acc) const {
    f32x4 zz = acc;
    if (bias[p]) zz += *reinterpret_cast<const f32x4*>(bias[p] + n);
    if (addend[p]) zz += *reinterpret_cast<const f32x4*>(addend[p] + (long)m * ld_add + n);
    const long o = (long)m * ld + n;
    if (z[p]) *reinterpret_cast<f32x4*>(z[p] + o) = zz;
    *reinterpret_cast<f32x4*>(y[p] + o) = f32x4{act_apply(act, zz[0]), act_apply(act, zz[1]), act_apply(act, zz[2]), act_apply(act, zz[3])};
  }
};
// split-R forward: partial sums into slab[(s*M + m)*N + n]; finished by bias_act_reduce_kernel
struct SplitStore {
  float* slab[GEMM_MAXP];
  int M[GEMM_MAXP];
  int N;
  int vec;
  static constexpr bool VEC = true;
  __device__ __forceinline__ void store(int p, int s, int m, int n, float acc) const {
    slab[p][((long)s * M[p] + m) * N + n] = acc;
  }
  __device__ __forceinline__ void store4(int p, int s, int m, int n, f32x4 acc) const {
    *reinterpret_cast<f32x4*>(slab[p] + ((long)s * M[p] + m) * N + n) = acc;
  }
};
// out = acc * act'(src)   (src = pre-activation for SiLU, output for ReLU; NULL = identity)
struct DgradStore {
  float* out[GEMM_MAXP];
  const float* src[GEMM_MAXP];
  const float* addend[GEMM_MAXP];  // optional [M][N] added before the mask (BPTT: dH_{t-1})
  int ld, act, ld_add, ld_src;
  int vec;
  static constexpr bool VEC = true;
  __device__ __forceinline__ void store(int p, int, int m, int n, float acc) const {
    long o = (long)m * ld + n;
    if (addend[p]) acc += addend[p][(long)m * ld_add + n];
    out[p][o] = src[p] ? acc * act_grad(act, src[p][(long)m * ld_src + n]) : acc;
  }
  __device__ __forceinline__ void store4(int p, int, int m, int n, f32x4 acc) const {
    if (addend[p]) acc += *reinterpret_cast<const f32x4*>(addend[p] + (long)m * ld_add + n);
    if (src[p]) {
      const f32x4 sv = *reinterpret_cast<const f32x4*>(src[p] + (long)m * ld_src + n);
      acc = f32x4{acc[0] * act_grad(act, sv[0]), acc[1] * act_grad(act, sv[1]), acc[2] * act_grad(act, sv[2]), acc[3] * act_grad(act, sv[3])};
    }
    *reinterpret_cast<f32x4*>(out[p] + (long)m * ld + n) = acc;
  }
};
// conv dgrad: row m of parity class -> NHWC input position; masks with ReLU of the input.
struct ConvDgradStore {
  float* out[GEMM_MAXP];
  const float* src[GEMM_MAXP];  // the conv input activation (post-ReLU) or NULL
  ConvGeom g;
  int vec;  // host-set: C % 4 == 0 and 16-byte aligned pointers -> store4 is legal
  // Round 6: contiguous along n (the input channel) - a lane ends with 4 consecutive channels of one pixel: one row
  // decomposition, one 16-byte mask load and one 16-byte store instead of four of each (the scalar form was most of what the
  // per-layer conv backward of 150 x 200 cost).  Same products, same accumulation order: bit-identical to the scalar form.
  static constexpr bool VEC = true;
  __device__ __forceinline__ long pos(int p, int m) const {
    const int S = g.S, cls = p % (S * S), py = cls / S, px = cls % S;
    const int nh = (g.H - py + S - 1) / S, nw = (g.W - px + S - 1) / S;
    const int img = m / (nh * nw), pp = m - img * nh * nw, y2 = pp / nw, x2 = pp - y2 * nw;
    return (((long)img * g.H + S * y2 + py) * g.W + S * x2 + px) * g.C;
  }
  __device__ __forceinline__ void store(int p, int, int m, int n, float acc) const {
    const long o = pos(p, m) + n;
    out[p][o] = (src[p] && !(src[p][o] > 0.f)) ? 0.f : acc;
  }
  __device__ __forceinline__ void store4(int p, int, int m, int n, f32x4 acc) const {
    const long o = pos(p, m) + n;
    if (src[p]) {
      const f32x4 sv = *reinterpret_cast<const f32x4*>(src[p] + o);
      acc = f32x4{sv[0] > 0.f ? acc[0] : 0.f, sv[1] > 0.f ? acc[1] : 0.f, sv[2] > 0.f ? acc[2] : 0.f, sv[3] > 0.f ? acc[3] : 0.f};
    }
    *reinterpret_cast<f32x4*>(out[p] + o) = acc;
  }
};
// wgrad: GEMM row = k (input feature / im2col column, k == K is the bias column),
// GEMM col = o (output feature).  Slab layout per (problem, split): [O][K+1].
struct WgradStore {
  float* slab[GEMM_MAXP];
  int K, O, nsplit;
  static constexpr bool VEC = false;  // contiguous along the GEMM row (k): the row-per-lane layout is the right one
  __device__ __forceinline__ void store(int p, int s, int k, int o, float acc) const {
    slab[p][((long)s * O + o) * (K + 1) + k] = acc;
  }
};
