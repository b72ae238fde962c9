// Dense ops of the TACO-RL hot path on top of the single MFMA contraction template:
// Linear / conv forward, backward-data, backward-weights (split-R slabs + reduce),
// spatial soft-argmax, and the composite LMPVisionEncoder / MLP forward+backward.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "../../include/tacorl_hip.h"
#include "enc_bwd_fused.h"
#include "mlp_fused.h"
#include "functors.h"

static thread_local char g_err[256] = "";
#define FAIL(code, ...)                        \
  do {                                         \
    snprintf(g_err, sizeof(g_err), __VA_ARGS__); \
    return (code);                             \
  } while (0)
#define CHECK(x)            \
  do {                      \
    int _r = (x);           \
    if (_r != TACORL_OK) return _r; \
  } while (0)

extern "C" const char* tacorl_hip_last_error(void) { return g_err; }
extern "C" int tacorl_hip_version(void) { return 1; }
extern "C" int tacorl_hip_init(int device) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) FAIL(TACORL_EINVAL, "no HIP device %d", device);
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    FAIL(TACORL_EINVAL, "device %d is %s, this library is gfx950-only", device, prop.gcnArchName);
  return TACORL_OK;
}

// Device-side time mark (constant 100 MHz clock): one 1-thread launch that stores the clock into
// marks[slot].  Placed between the launches of a captured step it gives the branch timeline of a graph
// replay without a profiler attached (rocprofv3's kernel trace perturbs the branch overlap).
__global__ void time_mark_kernel(unsigned long long* marks, int slot) { marks[slot] = wall_clock64(); }
extern "C" int tacorl_time_mark(unsigned long long* marks, int slot, tacorl_stream_t stream) {
  if (!marks || slot < 0) FAIL(TACORL_EINVAL, "time_mark: bad arguments");
  hipLaunchKernelGGL(time_mark_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, marks, slot);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

// Calibration aid for event brackets: a 1-thread kernel that runs for `ticks` of that clock and records its own begin and
// end (marks[slot], marks[slot + 1]).  A HIP-event bracket around it reads its duration PLUS the bracket's launch gaps, so
// bracket - (marks[slot + 1] - marks[slot]) is the overhead of a bracket around a kernel of that length (bench.py).
__global__ void time_spin_kernel(unsigned long long* marks, int slot, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  marks[slot] = t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  marks[slot + 1] = wall_clock64();
}
extern "C" int tacorl_time_spin(unsigned long long* marks, int slot, long ticks, tacorl_stream_t stream) {
  if (!marks || slot < 0 || ticks < 0 || ticks > 100000000L) FAIL(TACORL_EINVAL, "time_spin: bad arguments");
  hipLaunchKernelGGL(time_spin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, marks, slot, (unsigned long long)ticks);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

static inline long al4(long x) { return (x + 3) & ~3L; }
static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// ============================================================== linear forward
static int k_linear_fwd(int nprob, const float* const* x, int ldx, const float* const* w,
                        const float* const* b, float* const* y, float* const* z, const int* M, int K,
                        int N, int ldy, int act, int cd, hipStream_t st, const float* const* addend = nullptr,
                        int ld_add = 0) {
  if (nprob < 1 || nprob > GEMM_MAXP) FAIL(TACORL_EINVAL, "linear_fwd: nprob %d", nprob);
  RowMajorLoader la{}, lb{};
  BiasActStore ep{};
  GemmArgs g{};
  g.nprob = nprob; g.nsplit = 1; g.N = N;
  la.cols = K; la.ld = ldx; la.vec = (ldx % 4 == 0); la.ones_col = 0;
  lb.cols = K; lb.ld = K; lb.vec = (K % 4 == 0); lb.ones_col = 0;
  ep.ld = ldy; ep.act = act; ep.ld_add = ld_add;
  ep.vec = ldy % 4 == 0 && ld_add % 4 == 0;
  for (int p = 0; p < nprob; p++) {
    la.ptr[p] = x[p]; la.rows[p] = M[p]; la.vec &= aligned16(x[p]);
    lb.ptr[p] = w[p]; lb.rows[p] = N; lb.vec &= aligned16(w[p]);
    ep.bias[p] = b ? b[p] : nullptr; ep.y[p] = y[p]; ep.z[p] = z ? z[p] : nullptr;
    ep.addend[p] = addend ? addend[p] : nullptr;
    ep.vec &= aligned16(ep.bias[p]) && aligned16(y[p]) && aligned16(ep.z[p]) && aligned16(ep.addend[p]);
    g.M[p] = M[p]; g.R[p] = K;
  }
  return gemm_launch<RowMajorLoader, RowMajorLoader, false, false, BiasActStore>(la, lb, ep, g, cd, st);
}

extern "C" int tacorl_linear_fwd(int nprob, const float* const* x, int ldx, const float* const* w,
                                 const float* const* b, float* const* y, float* const* z, const int* M,
                                 int K, int N, int act, int compute_dtype, tacorl_stream_t stream) {
  return k_linear_fwd(nprob, x, ldx, w, b, y, z, M, K, N, N, act, compute_dtype, (hipStream_t)stream);
}
// Skinny GEMMs (M <= 256, long K: the RNN recurrence) launch too few tiles to hide HBM/L2 latency:
// split the reduction over blocks, then finish bias + addend + activation in a second pass.
struct ReduceTbl {
  const float* slab[GEMM_MAXP];
  const float* bias[GEMM_MAXP];
  const float* addend[GEMM_MAXP];
  float* y[GEMM_MAXP];
  int M[GEMM_MAXP];
};
__global__ void bias_act_reduce_kernel(ReduceTbl t, int nsplit, int N, int ldy, int ld_add, int act) {
  const int p = blockIdx.y, M = t.M[p];
  const long total = (long)M * N;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int m = (int)(e / N), n = (int)(e - (long)m * N);
    float z = t.bias[p] ? t.bias[p][n] : 0.f;
    for (int s = 0; s < nsplit; s++) z += t.slab[p][(long)s * total + e];
    if (t.addend[p]) z += t.addend[p][(long)m * ld_add + n];
    t.y[p][(long)m * ldy + n] = act_apply(act, z);
  }
}
static int splitk_choice(int nprob, const int* M, int K, int N) {
  long maxM = 0;
  for (int p = 0; p < nprob; p++) maxM = M[p] > maxM ? M[p] : maxM;
  const long tiles = (long)cdiv(maxM, 128) * cdiv(N, N > 32 ? 64 : (N > 16 ? 32 : 16)) * nprob;
  if (tiles >= 192 || K < 512) return 1;
  long ns = 512 / tiles;
  if (ns > K / 128) ns = K / 128;
  if (ns > 16) ns = 16;
  return ns < 2 ? 1 : (int)ns;
}
static int k_linear_fwd_splitk(int nprob, const float* const* x, int ldx, const float* const* w, const float* const* b,
                               const float* const* addend, int ld_add, float* const* y, int ldy, const int* M, int K,
                               int N, int act, int cd, void* ws, size_t ws_bytes, hipStream_t st) {
  const int ns = splitk_choice(nprob, M, K, N);
  long need = 0;
  for (int p = 0; p < nprob; p++) need += (long)ns * M[p] * N;
  if (ns == 1 || ws == nullptr || (size_t)need * sizeof(float) > ws_bytes)
    return k_linear_fwd(nprob, x, ldx, w, b, y, nullptr, M, K, N, ldy, act, cd, st, addend, ld_add);
  RowMajorLoader la{}, lb{};
  SplitStore ep{};
  GemmArgs g{};
  ReduceTbl t{};
  g.nprob = nprob; g.nsplit = ns; g.N = N;
  la.cols = K; la.ld = ldx; la.vec = (ldx % 4 == 0); la.ones_col = 0;
  lb.cols = K; lb.ld = K; lb.vec = (K % 4 == 0); lb.ones_col = 0;
  ep.N = N;
  ep.vec = N % 4 == 0 && aligned16(ws);  // slab offsets are multiples of N
  float* cur = (float*)ws;
  int maxM = 0;
  for (int p = 0; p < nprob; p++) {
    la.ptr[p] = x[p]; la.rows[p] = M[p]; la.vec &= aligned16(x[p]);
    lb.ptr[p] = w[p]; lb.rows[p] = N; lb.vec &= aligned16(w[p]);
    ep.slab[p] = cur; ep.M[p] = M[p];
    t.slab[p] = cur; t.bias[p] = b ? b[p] : nullptr; t.addend[p] = addend ? addend[p] : nullptr; t.y[p] = y[p]; t.M[p] = M[p];
    cur += (long)ns * M[p] * N;
    g.M[p] = M[p]; g.R[p] = K;
    maxM = M[p] > maxM ? M[p] : maxM;
  }
  CHECK((gemm_launch<RowMajorLoader, RowMajorLoader, false, false, SplitStore>(la, lb, ep, g, cd, st)));
  const long total = (long)maxM * N;
  dim3 grid((unsigned)(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), nprob);
  hipLaunchKernelGGL(bias_act_reduce_kernel, grid, dim3(256), 0, st, t, ns, N, ldy, ld_add, act);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
extern "C" size_t tacorl_linear_add_fwd_ws_bytes(int nprob, const int* M, int K, int N) {
  const int ns = splitk_choice(nprob, M, K, N);
  if (ns == 1) return 0;
  long need = 0;
  for (int p = 0; p < nprob; p++) need += (long)ns * M[p] * N;
  return (size_t)need * sizeof(float);
}

// y = act(x W^T + b + addend): one ReLU-RNN time step h_t = relu(W_hh h_{t-1} + b_hh + (W_ih x_t + b_ih))
// (torch nn.RNN(nonlinearity="relu"), reference networks/action_decoders/rnn_models.py:5-16)
extern "C" int tacorl_linear_add_fwd(int nprob, const float* const* x, int ldx, const float* const* w,
                                     const float* const* b, const float* const* addend, int ld_add,
                                     float* const* y, int ldy, const int* M, int K, int N, int act,
                                     int compute_dtype, void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  if (nprob < 1 || nprob > GEMM_MAXP) FAIL(TACORL_EINVAL, "linear_add_fwd: nprob %d", nprob);
  return k_linear_fwd_splitk(nprob, x, ldx, w, b, addend, ld_add, y, ldy, M, K, N, act, compute_dtype, ws, ws_bytes,
                             (hipStream_t)stream);
}

// ================================================================ conv forward
static ConvGeom make_geom(int H, int W, int C, int KH, int KW, int S, int CO) {
  ConvGeom g{H, W, C, KH, KW, S, CO, (H - KH) / S + 1, (W - KW) / S + 1};
  return g;
}
static bool conv_vec_ok(const ConvGeom& g) {
  return (g.KW * g.C) % 4 == 0 && (g.W * g.C) % 4 == 0 && (g.S * g.C) % 4 == 0 && (g.H * g.W * g.C) % 4 == 0;
}

template <typename InT>
static int k_conv_fwd(int nprob, const void* const* x, const float* const* w, const float* const* b,
                      float* const* y, const int* n_img, const ConvGeom& cg, int cd, hipStream_t st) {
  if (nprob < 1 || nprob > GEMM_MAXP) FAIL(TACORL_EINVAL, "conv_fwd: nprob %d", nprob);
  if ((cg.KW * cg.C) % 4 != 0) FAIL(TACORL_EINVAL, "conv_fwd: KW*C must be a multiple of 4");
  ConvColLoader<InT> la{};
  RowMajorLoader lb{};
  BiasActStore ep{};
  GemmArgs g{};
  const int K = cg.KH * cg.KW * cg.C;
  g.nprob = nprob; g.nsplit = 1; g.N = cg.CO;
  la.g = cg; la.vec = conv_vec_ok(cg); la.ones_col = 0;
  lb.cols = K; lb.ld = K; lb.vec = 1; lb.ones_col = 0;
  ep.ld = cg.CO; ep.act = ACT_RELU;
  ep.vec = cg.CO % 4 == 0;
  for (int p = 0; p < nprob; p++) {
    la.ptr[p] = (const InT*)x[p]; la.rows[p] = n_img[p] * cg.OH * cg.OW; la.vec &= aligned16(x[p]);
    lb.ptr[p] = w[p]; lb.rows[p] = cg.CO; lb.vec &= aligned16(w[p]);
    ep.bias[p] = b[p]; ep.y[p] = y[p]; ep.z[p] = nullptr;
    ep.vec &= aligned16(b[p]) && aligned16(y[p]);
    g.M[p] = la.rows[p]; g.R[p] = K;
  }
  // (round 6: the 128-deep K tile - gemm_launch's DEEPK - measured here for the 4.8 M-row convolutions of 150 x 200: conv1
  // 678 -> 1 658 us, conv2 / conv3 316 -> 413: sixteen row states + 64 staged floats per thread spill; not used)
  return gemm_launch<ConvColLoader<InT>, RowMajorLoader, false, false, BiasActStore>(la, lb, ep, g, cd, st);
}

static int conv_fwd_any(int nprob, const void* const* x, const float* const* w, const float* const* b,
                        float* const* y, const int* n_img, const ConvGeom& cg, int x_dtype, int cd,
                        hipStream_t st) {
  if (x_dtype == TACORL_BF16) return k_conv_fwd<__bf16>(nprob, x, w, b, y, n_img, cg, cd, st);
  return k_conv_fwd<float>(nprob, x, w, b, y, n_img, cg, cd, st);
}

extern "C" int tacorl_conv2d_relu_fwd(int nprob, const void* const* x, const float* const* w,
                                      const float* const* b, float* const* y, const int* n_img, int H,
                                      int W, int C, int KH, int KW, int S, int CO, int x_dtype,
                                      int compute_dtype, tacorl_stream_t stream) {
  if (H < KH || W < KW) FAIL(TACORL_EINVAL, "conv: image smaller than kernel");
  return conv_fwd_any(nprob, x, w, b, y, n_img, make_geom(H, W, C, KH, KW, S, CO), x_dtype, compute_dtype,
                      (hipStream_t)stream);
}

// ============================================================== backward-data
// linear: dX[m][i] = (sum_o dZ[m][o] W[o][i]) * act'(src[m][i])
static int k_linear_dgrad(int nprob, const float* const* dz, int ld_dz, const float* const* w, float* const* out,
                          int ld_out, const float* const* src, int act_src, const int* M, int O, int I,
                          int cd, hipStream_t st, const float* const* addend = nullptr, int ld_add = 0,
                          int ld_src = -1) {
  RowMajorLoader la{}, lb{};
  DgradStore ep{};
  GemmArgs g{};
  g.nprob = nprob; g.nsplit = 1; g.N = I;
  la.cols = O; la.ld = ld_dz; la.vec = (O % 4 == 0 && ld_dz % 4 == 0); la.ones_col = 0;
  lb.cols = I; lb.ld = I; lb.vec = (I % 4 == 0); lb.ones_col = 0;
  ep.ld = ld_out; ep.act = act_src; ep.ld_add = ld_add; ep.ld_src = ld_src < 0 ? ld_out : ld_src;
  ep.vec = ep.ld % 4 == 0 && ep.ld_add % 4 == 0 && ep.ld_src % 4 == 0;
  for (int p = 0; p < nprob; p++) {
    la.ptr[p] = dz[p]; la.rows[p] = M[p]; la.vec &= aligned16(dz[p]);
    lb.ptr[p] = w[p]; lb.rows[p] = O; lb.vec &= aligned16(w[p]);
    ep.out[p] = out[p]; ep.src[p] = src ? src[p] : nullptr; ep.addend[p] = addend ? addend[p] : nullptr;
    ep.vec &= aligned16(out[p]) && aligned16(ep.src[p]) && aligned16(ep.addend[p]);
    g.M[p] = M[p]; g.R[p] = O;
  }
  return gemm_launch<RowMajorLoader, RowMajorLoader, false, true, DgradStore>(la, lb, ep, g, cd, st);
}

// C-ABI primitives: backward of y = act(x W^T + b) for callers that schedule their own chains
// (the ReLU-RNN BPTT and the transformer backward).
extern "C" int tacorl_linear_dgrad(int nprob, const float* const* dz, int ld_dz, const float* const* w,
                                   float* const* out, int ld_out, const float* const* src, int ld_src, int act_src,
                                   const float* const* addend, int ld_add, const int* M, int O, int I,
                                   int compute_dtype, tacorl_stream_t stream) {
  if (nprob < 1 || nprob > GEMM_MAXP) FAIL(TACORL_EINVAL, "linear_dgrad: nprob %d", nprob);
  return k_linear_dgrad(nprob, dz, ld_dz, w, out, ld_out, src, act_src, M, O, I, compute_dtype, (hipStream_t)stream,
                        addend, ld_add, ld_src);
}

// The same with the reduction split over workgroups when the output is skinny and the reduction long (e.g. the
// transformer FFN's first Linear: [4096 x 2048] . [2048 x 32] is 32 output tiles - 32 CUs streaming 33 MB - 42 us;
// 16-way split + reduce: the whole chip).  Only without an activation-derivative source (the reduce pass adds the addend).
extern "C" size_t tacorl_linear_dgrad_ws_bytes(int nprob, const int* M, int O, int I) {
  return tacorl_linear_add_fwd_ws_bytes(nprob, M, O, I);
}
extern "C" int tacorl_linear_dgrad_splitk(int nprob, const float* const* dz, int ld_dz, const float* const* w,
                                          float* const* out, int ld_out, const float* const* src, int ld_src, int act_src,
                                          const float* const* addend, int ld_add, const int* M, int O, int I,
                                          int compute_dtype, void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  if (nprob < 1 || nprob > GEMM_MAXP) FAIL(TACORL_EINVAL, "linear_dgrad: nprob %d", nprob);
  hipStream_t st = (hipStream_t)stream;
  const int ns = splitk_choice(nprob, M, O, I);
  long need = 0;
  for (int p = 0; p < nprob; p++) need += (long)ns * M[p] * I;
  bool masked = false;
  for (int p = 0; p < nprob; p++) masked |= src && src[p];
  if (ns == 1 || masked || ws == nullptr || (size_t)need * sizeof(float) > ws_bytes)
    return k_linear_dgrad(nprob, dz, ld_dz, w, out, ld_out, src, act_src, M, O, I, compute_dtype, st, addend, ld_add, ld_src);
  RowMajorLoader la{}, lb{};
  SplitStore ep{};
  GemmArgs g{};
  ReduceTbl t{};
  g.nprob = nprob; g.nsplit = ns; g.N = I;
  la.cols = O; la.ld = ld_dz; la.vec = (O % 4 == 0 && ld_dz % 4 == 0); la.ones_col = 0;
  lb.cols = I; lb.ld = I; lb.vec = (I % 4 == 0); lb.ones_col = 0;
  ep.N = I;
  ep.vec = I % 4 == 0 && aligned16(ws);
  float* cur = (float*)ws;
  int maxM = 0;
  for (int p = 0; p < nprob; p++) {
    la.ptr[p] = dz[p]; la.rows[p] = M[p]; la.vec &= aligned16(dz[p]);
    lb.ptr[p] = w[p]; lb.rows[p] = O; lb.vec &= aligned16(w[p]);
    ep.slab[p] = cur; ep.M[p] = M[p];
    t.slab[p] = cur; t.bias[p] = nullptr; t.addend[p] = addend ? addend[p] : nullptr; t.y[p] = out[p]; t.M[p] = M[p];
    cur += (long)ns * M[p] * I;
    g.M[p] = M[p]; g.R[p] = O;
    maxM = M[p] > maxM ? M[p] : maxM;
  }
  CHECK((gemm_launch<RowMajorLoader, RowMajorLoader, false, true, SplitStore>(la, lb, ep, g, compute_dtype, st)));
  const long total = (long)maxM * I;
  dim3 grid((unsigned)(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), nprob);
  hipLaunchKernelGGL(bias_act_reduce_kernel, grid, dim3(256), 0, st, t, ns, I, ld_out, ld_add, ACT_NONE);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

// conv: gather form, one GEMM per input-pixel parity class (S*S classes per net).
static int k_conv_dgrad(int nnets, const float* const* dout, const float* const* w, float* const* din,
                        const float* const* src, const int* n_img, const ConvGeom& cg, int cd,
                        hipStream_t st) {
  const int S = cg.S, ncls = S * S;
  if (nnets * ncls > GEMM_MAXP) FAIL(TACORL_EINVAL, "conv_dgrad: %d nets x %d classes > %d", nnets, ncls, GEMM_MAXP);
  if (cg.C % 4 || cg.CO % 4) FAIL(TACORL_EINVAL, "conv_dgrad: channels must be multiples of 4");
  ConvDgradALoader la{};
  ConvDgradWLoader lb{};
  ConvDgradStore ep{};
  GemmArgs g{};
  const int TA = (cg.KH + S - 1) / S, TB = (cg.KW + S - 1) / S;
  la.g = lb.g = ep.g = cg;
  la.TA = lb.TA = TA; la.TB_ = lb.TB_ = TB;
  g.nprob = nnets * ncls; g.nsplit = 1; g.N = cg.C;
  for (int n = 0; n < nnets; n++)
    for (int c = 0; c < ncls; c++) {
      const int p = n * ncls + c, py = c / S, px = c % S;
      const int nh = (cg.H - py + S - 1) / S, nw = (cg.W - px + S - 1) / S;
      la.ptr[p] = dout[n]; la.n_img[p] = n_img[n];
      lb.ptr[p] = w[n];
      ep.out[p] = din[n]; ep.src[p] = src ? src[n] : nullptr;
      g.M[p] = n_img[n] * nh * nw; g.R[p] = TA * TB * cg.CO;
    }
  ep.vec = 1;  // C % 4 == 0 (checked above); 16-byte aligned buffers or the scalar stores
  for (int n = 0; n < nnets; n++)
    if (((uintptr_t)din[n] | (uintptr_t)(src ? src[n] : nullptr)) & 15) ep.vec = 0;
  return gemm_launch<ConvDgradALoader, ConvDgradWLoader, false, true, ConvDgradStore>(la, lb, ep, g, cd, st);
}

// ============================================================ backward-weights
// dW[o][k] = sum_m dZ[m][o] X(m, k), db[o] = sum_m dZ[m][o]; computed as the GEMM
// C[k][o] with the reduction over m split across blocks into slabs, then reduced.
struct PtrTable {  // device-visible pointer arrays passed by value through a tiny kernel arg
  const float* slab[GEMM_MAXP];
  float* dw[GEMM_MAXP];
  float* db[GEMM_MAXP];
};
__global__ void slab_reduce_kernel_tbl(PtrTable t, int nsplit, int O, int K, int accumulate) {
  const int p = blockIdx.y;
  const float* slab = t.slab[p];
  float* dw = t.dw[p];
  float* db = t.db[p];
  const long per = (long)O * (K + 1);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    int i = 0;
    for (; i + 8 <= nsplit; i += 8) {  // eight slabs' values in flight, summed in slab order as before
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = slab[(i + j) * per + e];
#pragma unroll
      for (int j = 0; j < 8; j++) s += v[j];
    }
    for (; i < nsplit; i++) s += slab[i * per + e];
    const int o = (int)(e / (K + 1)), k = (int)(e - (long)o * (K + 1));
    float* dst = (k == K) ? (db ? db + o : nullptr) : dw + (long)o * K + k;
    if (dst) *dst = accumulate ? *dst + s : s;
  }
}

static int pick_nsplit(int nprob, int K, int O, long maxR) {
  long tiles = (long)cdiv(K + 1, 128) * cdiv(O, O > 32 ? 64 : (O > 16 ? 32 : 16)) * nprob;
  long ns = 768 / (tiles > 0 ? tiles : 1);
  long cap = maxR / 128;
  if (ns > cap) ns = cap;
  if (ns > 128) ns = 128;
  if (ns < 1) ns = 1;
  return (int)ns;
}
static size_t wgrad_ws_bytes(int nprob, int K, int O, long maxR) {
  return (size_t)nprob * pick_nsplit(nprob, K, O, maxR) * O * (K + 1) * sizeof(float);
}

template <class LA>
static int k_wgrad(LA& la, int nprob, const float* const* dz, int ld_dz, const int* R, int K, int O,
                   float* const* dw, float* const* db, int accumulate, void* ws, size_t ws_bytes, int cd,
                   hipStream_t st) {
  long maxR = 0;
  for (int p = 0; p < nprob; p++) maxR = R[p] > maxR ? R[p] : maxR;
  const int ns = pick_nsplit(nprob, K, O, maxR);
  if (wgrad_ws_bytes(nprob, K, O, maxR) > ws_bytes) FAIL(TACORL_ENOMEM, "wgrad: workspace too small");
  RowMajorLoader lb{};
  WgradStore ep{};
  GemmArgs g{};
  PtrTable t{};
  g.nprob = nprob; g.nsplit = ns; g.N = O;
  lb.cols = O; lb.ld = ld_dz; lb.vec = (O % 4 == 0 && ld_dz % 4 == 0); lb.ones_col = 0;
  ep.K = K; ep.O = O; ep.nsplit = ns;
  const long per = (long)O * (K + 1);
  for (int p = 0; p < nprob; p++) {
    lb.ptr[p] = dz[p]; lb.rows[p] = R[p]; lb.vec &= aligned16(dz[p]);
    ep.slab[p] = (float*)ws + (long)p * ns * per;
    t.slab[p] = ep.slab[p]; t.dw[p] = dw[p]; t.db[p] = db ? db[p] : nullptr;
    g.M[p] = K + 1; g.R[p] = R[p];
  }
  CHECK((gemm_launch<LA, RowMajorLoader, true, true, WgradStore>(la, lb, ep, g, cd, st)));
  // up to 1024 workgroups (256 before: the 24-slab sums ran as 4 waves per CU, latency-bound).  PlayLMP, two processes per
  // setting on one box: B = 32 1.195 -> 1.174 ms/step at 1024 (1.172 at 4096), B = 256 2.10 / 2.12 / 2.13 at 256 / 1024 / 4096
  // (process noise +-1 %: wider crowds the step's other branches there) - round 5.  TACORL_SLAB_REDUCE_WGS overrides.
  static const int cap = [] { const char* e = getenv("TACORL_SLAB_REDUCE_WGS"); return e && atoi(e) > 0 ? atoi(e) : 1024; }();
  dim3 grid(cdiv(per, 256) > cap ? cap : cdiv(per, 256), nprob);
  hipLaunchKernelGGL(slab_reduce_kernel_tbl, grid, dim3(256), 0, st, t, ns, O, K, accumulate);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

static int k_linear_wgrad(int nprob, const float* const* x, int ldx, const float* const* dz, int ld_dz, const int* M,
                          int K, int O, float* const* dw, float* const* db, int accumulate, void* ws,
                          size_t ws_bytes, int cd, hipStream_t st) {
  RowMajorLoader la{};
  la.cols = K; la.ld = ldx; la.vec = (ldx % 4 == 0 && K % 4 == 0); la.ones_col = 1;
  for (int p = 0; p < nprob; p++) { la.ptr[p] = x[p]; la.rows[p] = M[p]; la.vec &= aligned16(x[p]); }
  return k_wgrad(la, nprob, dz, ld_dz, M, K, O, dw, db, accumulate, ws, ws_bytes, cd, st);
}

extern "C" size_t tacorl_linear_wgrad_ws_bytes(int nprob, const int* M, int K, int O) {
  long maxM = 0;
  for (int p = 0; p < nprob; p++) maxM = M[p] > maxM ? M[p] : maxM;
  return wgrad_ws_bytes(nprob, K, O, maxM);
}
extern "C" int tacorl_linear_wgrad(int nprob, const float* const* x, int ldx, const float* const* dz, int ld_dz,
                                   const int* M, int K, int O, float* const* dw, float* const* db, int accumulate,
                                   int compute_dtype, void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  if (nprob < 1 || nprob > GEMM_MAXP) FAIL(TACORL_EINVAL, "linear_wgrad: nprob %d", nprob);
  return k_linear_wgrad(nprob, x, ldx, dz, ld_dz, M, K, O, dw, db, accumulate, ws, ws_bytes, compute_dtype,
                        (hipStream_t)stream);
}

template <typename InT>
static int k_conv_wgrad(int nprob, const void* const* x, const float* const* dz, const int* n_img,
                        const ConvGeom& cg, float* const* dw, float* const* db, int accumulate, void* ws,
                        size_t ws_bytes, int cd, hipStream_t st) {
  ConvColLoader<InT> la{};
  la.g = cg; la.vec = conv_vec_ok(cg); la.ones_col = 1;
  int R[GEMM_MAXP];
  for (int p = 0; p < nprob; p++) {
    la.ptr[p] = (const InT*)x[p]; R[p] = la.rows[p] = n_img[p] * cg.OH * cg.OW; la.vec &= aligned16(x[p]);
  }
  return k_wgrad(la, nprob, dz, cg.CO, R, cg.KH * cg.KW * cg.C, cg.CO, dw, db, accumulate, ws, ws_bytes, cd, st);
}

// ========================================================= spatial soft-argmax
// y3: [n][P][64] (post-ReLU conv3, NHWC); out: [n][128] interleaved (x_c, y_c) in pixel units
// (reference networks/visual_encoders/utils.py:39-65).  Per-layer path, any conv3 output size: one WORKGROUP per image
// and all problems of a call in one launch (grid.y = problem); lane = channel (a pixel's 64 channels are one 256-byte
// row: every load / store moves whole rows), wave w takes pixels w, w + 4, ...; the per-channel max and sums meet
// across the four waves in LDS.  (One wave per image with a serial loop over the pixels - 128 waves on the whole chip
// for a 128-image problem - was 57 us forward / 107 us backward per problem at 12 x 12 pixels.)
struct SaArgs {
  const float* y3[GEMM_MAXP];
  const float* temp[GEMM_MAXP];
  const float* sa[GEMM_MAXP];    // backward: forward output
  const float* d_sa[GEMM_MAXP];  // backward: its gradient
  float* out[GEMM_MAXP];         // forward: sa;  backward: dz3
  float* dtp[GEMM_MAXP];         // backward: per-image d(temperature)
  int n[GEMM_MAXP];
};
__global__ __launch_bounds__(256) void softargmax_fwd_kernel(SaArgs a, int OH, int OW) {
  const int p = blockIdx.y, img = blockIdx.x;
  if (img >= a.n[p]) return;
  __shared__ float shm[4][64], shs[4][64], shx[4][64], shy[4][64];
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float t = a.temp[p][0];
  const int P = OH * OW;
  const float* q = a.y3[p] + (long)img * P * 64 + c;
  float mx = -INFINITY;
  for (int i = w; i < P; i += 4) mx = fmaxf(mx, q[i * 64] / t);
  shm[w][c] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(shm[0][c], shm[1][c]), fmaxf(shm[2][c], shm[3][c]));
  float se = 0.f, sx = 0.f, sy = 0.f;
  for (int i = w; i < P; i += 4) {
    const float e = expf(q[i * 64] / t - mx);
    se += e; sx += e * (float)(i % OW); sy += e * (float)(i / OW);
  }
  shs[w][c] = se; shx[w][c] = sx; shy[w][c] = sy;
  __syncthreads();
  if (w == 0) {
    se = ((shs[0][c] + shs[1][c]) + shs[2][c]) + shs[3][c];
    sx = ((shx[0][c] + shx[1][c]) + shx[2][c]) + shx[3][c];
    sy = ((shy[0][c] + shy[1][c]) + shy[2][c]) + shy[3][c];
    f32x2 r = {sx / se, sy / se};
    *reinterpret_cast<f32x2*>(a.out[p] + (long)img * 128 + 2 * c) = r;
  }
}
// d_sa: [n][128]; writes dZ3 = d(y3) masked by ReLU(y3 > 0), and per-image d(temperature).
__global__ __launch_bounds__(256) void softargmax_bwd_kernel(SaArgs a, int OH, int OW) {
  const int p = blockIdx.y, img = blockIdx.x;
  if (img >= a.n[p]) return;
  __shared__ float shm[4][64], shs[4][64], sh[4];
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float t = a.temp[p][0];
  const int P = OH * OW;
  const float* q = a.y3[p] + (long)img * P * 64 + c;
  float* o = a.out[p] + (long)img * P * 64 + c;
  float mx = -INFINITY;
  for (int i = w; i < P; i += 4) mx = fmaxf(mx, q[i * 64] / t);
  shm[w][c] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(shm[0][c], shm[1][c]), fmaxf(shm[2][c], shm[3][c]));
  float se = 0.f;
  for (int i = w; i < P; i += 4) se += expf(q[i * 64] / t - mx);
  shs[w][c] = se;
  __syncthreads();
  se = ((shs[0][c] + shs[1][c]) + shs[2][c]) + shs[3][c];
  const float gx = a.d_sa[p][(long)img * 128 + 2 * c], gy = a.d_sa[p][(long)img * 128 + 2 * c + 1];
  const float dot = gx * a.sa[p][(long)img * 128 + 2 * c] + gy * a.sa[p][(long)img * 128 + 2 * c + 1];
  float dt = 0.f;
  for (int i = w; i < P; i += 4) {
    const float v = q[i * 64], s = v / t;
    const float pr = expf(s - mx) / se;
    const float ds = pr * (gx * (float)(i % OW) + gy * (float)(i / OW) - dot);
    dt -= ds * s / t;
    o[i * 64] = v > 0.f ? ds / t : 0.f;
  }
  dt = wave_sum(dt);
  if (c == 0) sh[w] = dt;
  __syncthreads();
  if (threadIdx.x == 0) a.dtp[p][img] = sh[0] + sh[1] + sh[2] + sh[3];
}
// Same math, all problems of the fused backward in one launch: one workgroup per image, a wave owns 16
// channels x 4 pixel groups, every activation is read once and kept in registers.
#define SAB_MAXI_HUGE 80 // <= 320 pixels (150 x 200: 315), 160 registers of activations / exponentials per lane: the per-layer backward in bf16 mode
#define SAB_MAXI_BIG 36  // pixels per lane: conv3 output <= 52 pixels (49 / 16 / 8 at 84 / 64 / 44x60) with 13, <= 144 (128 x 128) with 36
struct SabArgs {
  const float* y3[EBW_MAXP];
  const float* temp[EBW_MAXP];
  const float* sa[EBW_MAXP];
  const float* d_sa[EBW_MAXP];
  float* dz3[EBW_MAXP];
  float* dtp[EBW_MAXP];
  float* gtemp[EBW_MAXP];
  int n[EBW_MAXP];
};
template <int SAB_MAXI>
__global__ __launch_bounds__(256) void softargmax_bwd_batch_kernel(SabArgs a, int P, int OW) {
  // lane = channel (a pixel's 64 channels are one 256-byte row: every load / store instruction moves whole rows),
  // wave w takes pixels w, w + 4, ...; the per-channel max / sum meet across the four waves in LDS
  const int p = blockIdx.y, img = blockIdx.x;
  if (img >= a.n[p]) return;
  __shared__ float shm[4][64], shs[4][64], sh[4];
  const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float t = a.temp[p][0];
  const float it = 1.0f / t;  // one IEEE division per thread; the per-element v / t were 3 x 13 divisions, each ~10 VALU
  const float* q = a.y3[p] + (long)img * P * 64 + c;
  float* o = a.dz3[p] + (long)img * P * 64 + c;
  // the per-image soft-argmax inputs travel beside the activation rows (they were dependent loads behind two barriers)
  const f32x2 g2 = *reinterpret_cast<const f32x2*>(a.d_sa[p] + (long)img * 128 + 2 * c);
  const f32x2 f2 = *reinterpret_cast<const f32x2*>(a.sa[p] + (long)img * 128 + 2 * c);
  float v[SAB_MAXI], e[SAB_MAXI];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < SAB_MAXI; k++) {
    const int i = w + 4 * k;
    v[k] = i < P ? q[i * 64] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < SAB_MAXI; k++) {
    v[k] *= it;  // s = y / t (exact products of the same two fp32 numbers the forward used are not needed: dz3 is
                 // consumed as a bf16 MFMA operand and the temperature gradient is a sum over 10^5 terms)
    if (w + 4 * k < P) mx = fmaxf(mx, v[k]);
  }
  shm[w][c] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(shm[0][c], shm[1][c]), fmaxf(shm[2][c], shm[3][c]));
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < SAB_MAXI; k++) {
    e[k] = (w + 4 * k < P) ? __expf(v[k] - mx) : 0.f;  // kept: the second pass re-used to recompute every exponential
    se += e[k];
  }
  shs[w][c] = se;
  __syncthreads();
  se = ((shs[0][c] + shs[1][c]) + shs[2][c]) + shs[3][c];
  const float rse = 1.0f / se;
  const float gx = g2[0], gy = g2[1];
  const float dot = gx * f2[0] + gy * f2[1];
  float dt = 0.f;
#pragma unroll
  for (int k = 0; k < SAB_MAXI; k++) {
    const int i = w + 4 * k;
    if (i < P) {
      const int iy = i / OW;
      const float s = v[k], pr = e[k] * rse;
      const float ds = pr * (gx * (float)(i - iy * OW) + gy * (float)iy - dot);
      dt -= ds * s * it;
      o[i * 64] = v[k] != 0.f ? ds * it : 0.f;  // ReLU mask: y3 >= 0, and y3 > 0 <=> y3 / t != 0 for either sign of t
    }
  }
  dt = wave_sum(dt);
  if (c == 0) sh[w] = dt;
  __syncthreads();
  if (threadIdx.x == 0) a.dtp[p][img] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void sum_to_scalar_batch_kernel(SabArgs a, int accumulate) {
  __shared__ float sh[4];
  const int p = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < a.n[p]; i += 256) s += a.dtp[p][i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = sh[0] + sh[1] + sh[2] + sh[3];
    *a.gtemp[p] = accumulate ? *a.gtemp[p] + v : v;
  }
}
__global__ void sum_to_scalar_kernel(const float* __restrict__ part, int n, float* out, int accumulate) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = sh[0] + sh[1] + sh[2] + sh[3];
    *out = accumulate ? *out + v : v;
  }
}

// ================================================================== encoder
enum { E_W1, E_B1, E_W2, E_B2, E_W3, E_B3, E_T, E_FW1, E_FB1, E_FW2, E_FB2, E_N };
extern "C" long tacorl_encoder_param_layout(long* o) {
  const long sz[E_N] = {32 * 8 * 8 * 3, 32, 64 * 4 * 4 * 32, 64, 64 * 3 * 3 * 64, 64, 1, 256 * 128, 256, 32 * 256, 32};
  long off = 0;
  for (int i = 0; i < E_N; i++) { if (o) o[i] = off; off = al4(off + sz[i]); }
  return off;
}
struct EncDims { ConvGeom c1, c2, c3; };
static bool enc_dims(int H, int W, EncDims& d) {
  if (H < 36 || W < 36) return false;  // conv stack needs >= 36 px (8/4, 4/2, 3/1)
  d.c1 = make_geom(H, W, 3, 8, 8, 4, 32);
  d.c2 = make_geom(d.c1.OH, d.c1.OW, 32, 4, 4, 2, 64);
  d.c3 = make_geom(d.c2.OH, d.c2.OW, 64, 3, 3, 1, 64);
  return d.c3.OH >= 1 && d.c3.OW >= 1;
}
extern "C" long tacorl_encoder_act_layout(int n, int H, int W, long* o) {
  EncDims d;
  if (!enc_dims(H, W, d)) return -1;
  const long sz[5] = {(long)n * d.c1.OH * d.c1.OW * 32, (long)n * d.c2.OH * d.c2.OW * 64,
                      (long)n * d.c3.OH * d.c3.OW * 64, (long)n * 128, (long)n * 256};
  long off = 0;
  for (int i = 0; i < 5; i++) { if (o) o[i] = off; off = al4(off + sz[i]); }
  return off;
}

extern "C" int tacorl_encoder_fwd(int nprob, const void* const* img, const float* const* params,
                                  float* const* out, float* const* act, const int* n_img, int H, int W,
                                  int img_dtype, int cd, tacorl_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  EncDims d;
  if (!enc_dims(H, W, d)) FAIL(TACORL_EINVAL, "encoder: image %dx%d too small", H, W);
  if (nprob < 1 || nprob > GEMM_MAXP) FAIL(TACORL_EINVAL, "encoder_fwd: nprob %d", nprob);
  long po[E_N];
  tacorl_encoder_param_layout(po);
  const float *w1[GEMM_MAXP], *b1[GEMM_MAXP], *w2[GEMM_MAXP], *b2[GEMM_MAXP], *w3[GEMM_MAXP], *b3[GEMM_MAXP],
      *fw1[GEMM_MAXP], *fb1[GEMM_MAXP], *fw2[GEMM_MAXP], *fb2[GEMM_MAXP];
  float *y1[GEMM_MAXP], *y2[GEMM_MAXP], *y3[GEMM_MAXP], *sa[GEMM_MAXP], *h1[GEMM_MAXP];
  for (int p = 0; p < nprob; p++) {
    long ao[5];
    tacorl_encoder_act_layout(n_img[p], H, W, ao);
    const float* P = params[p];
    w1[p] = P + po[E_W1]; b1[p] = P + po[E_B1]; w2[p] = P + po[E_W2]; b2[p] = P + po[E_B2];
    w3[p] = P + po[E_W3]; b3[p] = P + po[E_B3]; fw1[p] = P + po[E_FW1]; fb1[p] = P + po[E_FB1];
    fw2[p] = P + po[E_FW2]; fb2[p] = P + po[E_FB2];
    y1[p] = act[p] + ao[0]; y2[p] = act[p] + ao[1]; y3[p] = act[p] + ao[2]; sa[p] = act[p] + ao[3];
    h1[p] = act[p] + ao[4];
  }
  CHECK(conv_fwd_any(nprob, img, w1, b1, y1, n_img, d.c1, img_dtype, cd, st));
  CHECK(conv_fwd_any(nprob, (const void* const*)y1, w2, b2, y2, n_img, d.c2, TACORL_F32, cd, st));
  CHECK(conv_fwd_any(nprob, (const void* const*)y2, w3, b3, y3, n_img, d.c3, TACORL_F32, cd, st));
  {
    SaArgs sg{};
    int mxn = 0;
    for (int p = 0; p < nprob; p++) {
      sg.y3[p] = y3[p]; sg.temp[p] = params[p] + po[E_T]; sg.out[p] = sa[p]; sg.n[p] = n_img[p];
      mxn = n_img[p] > mxn ? n_img[p] : mxn;
    }
    if (mxn > 0) hipLaunchKernelGGL(softargmax_fwd_kernel, dim3(mxn, nprob), dim3(256), 0, st, sg, d.c3.OH, d.c3.OW);
  }
  CHECK(k_linear_fwd(nprob, sa, 128, fw1, fb1, h1, nullptr, n_img, 128, 256, 256, ACT_RELU, cd, st));
  CHECK(k_linear_fwd(nprob, h1, 256, fw2, fb2, out, nullptr, n_img, 256, 32, 32, ACT_NONE, cd, st));
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

static size_t enc_slab_bytes(int nprob, long maxn, const EncDims& d) {
  size_t m = 0, s;
  s = wgrad_ws_bytes(nprob, 192, 32, maxn * d.c1.OH * d.c1.OW); m = s > m ? s : m;
  s = wgrad_ws_bytes(nprob, 512, 64, maxn * d.c2.OH * d.c2.OW); m = s > m ? s : m;
  s = wgrad_ws_bytes(nprob, 576, 64, maxn * d.c3.OH * d.c3.OW); m = s > m ? s : m;
  s = wgrad_ws_bytes(nprob, 128, 256, maxn); m = s > m ? s : m;
  s = wgrad_ws_bytes(nprob, 256, 32, maxn); m = s > m ? s : m;
  return m;
}
// per-problem scratch: d_h1 | d_sa | dz3 | dz2 | dz1 | dtemp_part
static long enc_bwd_scratch_layout(int n, const EncDims& d, long* o) {
  const long sz[6] = {(long)n * 256, (long)n * 128, (long)n * d.c3.OH * d.c3.OW * 64,
                      (long)n * d.c2.OH * d.c2.OW * 64, (long)n * d.c1.OH * d.c1.OW * 32, (long)n};
  long off = 0;
  for (int i = 0; i < 6; i++) { if (o) o[i] = off; off = al4(off + sz[i]); }
  return off;
}
extern "C" size_t tacorl_encoder_bwd_ws_bytes(int nprob, const int* n_img, int H, int W) {
  EncDims d;
  if (!enc_dims(H, W, d)) return 0;
  long tot = 0, maxn = 0;
  for (int p = 0; p < nprob; p++) { tot += enc_bwd_scratch_layout(n_img[p], d, nullptr); maxn = n_img[p] > maxn ? n_img[p] : maxn; }
  return (size_t)tot * sizeof(float) + enc_slab_bytes(nprob, maxn, d);
}

extern "C" int tacorl_encoder_bwd(int nprob, const void* const* img, const float* const* params,
                                  const float* const* act, const float* const* d_out, float* const* grads,
                                  const int* n_img, int H, int W, int img_dtype, int cd, int accumulate,
                                  void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  EncDims d;
  if (!enc_dims(H, W, d)) FAIL(TACORL_EINVAL, "encoder: image %dx%d too small", H, W);
  if (nprob < 1 || nprob * 4 > GEMM_MAXP) FAIL(TACORL_EINVAL, "encoder_bwd: nprob %d (max %d)", nprob, GEMM_MAXP / 4);
  if (ws_bytes < tacorl_encoder_bwd_ws_bytes(nprob, n_img, H, W)) FAIL(TACORL_ENOMEM, "encoder_bwd: workspace too small");
  long po[E_N];
  tacorl_encoder_param_layout(po);
  const float *w2[GEMM_MAXP], *w3[GEMM_MAXP], *fw1[GEMM_MAXP], *fw2[GEMM_MAXP];
  const float *y1[GEMM_MAXP], *y2[GEMM_MAXP], *y3[GEMM_MAXP], *sa[GEMM_MAXP], *h1[GEMM_MAXP];
  float *d_h1[GEMM_MAXP], *d_sa[GEMM_MAXP], *dz3[GEMM_MAXP], *dz2[GEMM_MAXP], *dz1[GEMM_MAXP], *dtp[GEMM_MAXP];
  float *g_w1[GEMM_MAXP], *g_b1[GEMM_MAXP], *g_w2[GEMM_MAXP], *g_b2[GEMM_MAXP], *g_w3[GEMM_MAXP], *g_b3[GEMM_MAXP],
      *g_fw1[GEMM_MAXP], *g_fb1[GEMM_MAXP], *g_fw2[GEMM_MAXP], *g_fb2[GEMM_MAXP];
  float* cur = (float*)ws;
  long maxn = 0;
  for (int p = 0; p < nprob; p++) {
    long ao[5], so[6];
    tacorl_encoder_act_layout(n_img[p], H, W, ao);
    const long tot = enc_bwd_scratch_layout(n_img[p], d, so);
    const float* P = params[p];
    w2[p] = P + po[E_W2]; w3[p] = P + po[E_W3]; fw1[p] = P + po[E_FW1]; fw2[p] = P + po[E_FW2];
    y1[p] = act[p] + ao[0]; y2[p] = act[p] + ao[1]; y3[p] = act[p] + ao[2]; sa[p] = act[p] + ao[3]; h1[p] = act[p] + ao[4];
    d_h1[p] = cur + so[0]; d_sa[p] = cur + so[1]; dz3[p] = cur + so[2]; dz2[p] = cur + so[3]; dz1[p] = cur + so[4]; dtp[p] = cur + so[5];
    cur += tot;
    float* G = grads[p];
    g_w1[p] = G + po[E_W1]; g_b1[p] = G + po[E_B1]; g_w2[p] = G + po[E_W2]; g_b2[p] = G + po[E_B2];
    g_w3[p] = G + po[E_W3]; g_b3[p] = G + po[E_B3]; g_fw1[p] = G + po[E_FW1]; g_fb1[p] = G + po[E_FB1];
    g_fw2[p] = G + po[E_FW2]; g_fb2[p] = G + po[E_FB2];
    maxn = n_img[p] > maxn ? n_img[p] : maxn;
  }
  void* slab = cur;
  const size_t slab_bytes = enc_slab_bytes(nprob, maxn, d);
  // fc2: out = h1 W2^T + b2
  CHECK(k_linear_wgrad(nprob, h1, 256, d_out, 32, n_img, 256, 32, g_fw2, g_fb2, accumulate, slab, slab_bytes, cd, st));
  CHECK(k_linear_dgrad(nprob, d_out, 32, fw2, d_h1, 256, h1, ACT_RELU, n_img, 32, 256, cd, st));
  // fc1: h1 = relu(sa W1^T + b1)
  CHECK(k_linear_wgrad(nprob, sa, 128, (const float* const*)d_h1, 256, n_img, 128, 256, g_fw1, g_fb1, accumulate, slab, slab_bytes, cd, st));
  CHECK(k_linear_dgrad(nprob, (const float* const*)d_h1, 256, fw1, d_sa, 128, nullptr, ACT_NONE, n_img, 256, 128, cd, st));
  // spatial soft-argmax (+ temperature) and ReLU mask of conv3
  {
    SaArgs sg{};
    int mxn = 0;
    for (int p = 0; p < nprob; p++) {
      sg.y3[p] = y3[p]; sg.temp[p] = params[p] + po[E_T]; sg.sa[p] = sa[p]; sg.d_sa[p] = d_sa[p]; sg.out[p] = dz3[p];
      sg.dtp[p] = dtp[p]; sg.n[p] = n_img[p];
      mxn = n_img[p] > mxn ? n_img[p] : mxn;
    }
    const int P3 = d.c3.OH * d.c3.OW;
    if (cd == TACORL_BF16 && nprob <= EBW_MAXP && P3 > 4 * SAB_MAXI_BIG && P3 <= 4 * SAB_MAXI_HUGE) {
      // Round 6 (bf16 mode, conv3 outputs beyond the LDS-resident backward's - 150 x 200: 15 x 21 pixels): the register-resident
      // batch kernel of the fused backward - every activation read once, one exponential per element - instead of three passes
      // with a division per element (89 us -> for 384 images at 150 x 200); the temperature partials of all problems in one launch.
      SabArgs sb{};
      for (int p = 0; p < nprob; p++) {
        sb.y3[p] = y3[p]; sb.temp[p] = params[p] + po[E_T]; sb.sa[p] = sa[p]; sb.d_sa[p] = d_sa[p]; sb.dz3[p] = dz3[p];
        sb.dtp[p] = dtp[p]; sb.gtemp[p] = grads[p] + po[E_T]; sb.n[p] = n_img[p];
      }
      if (mxn > 0) {
        hipLaunchKernelGGL(softargmax_bwd_batch_kernel<SAB_MAXI_HUGE>, dim3(mxn, nprob), dim3(256), 0, st, sb, P3, d.c3.OW);
        hipLaunchKernelGGL(sum_to_scalar_batch_kernel, dim3(nprob), dim3(256), 0, st, sb, accumulate);
      }
    } else {
      if (mxn > 0) hipLaunchKernelGGL(softargmax_bwd_kernel, dim3(mxn, nprob), dim3(256), 0, st, sg, d.c3.OH, d.c3.OW);
      for (int p = 0; p < nprob; p++)
        if (n_img[p] > 0)
          hipLaunchKernelGGL(sum_to_scalar_kernel, dim3(1), dim3(256), 0, st, dtp[p], n_img[p], grads[p] + po[E_T], accumulate);
    }
  }
  // conv3
  CHECK(k_conv_wgrad<float>(nprob, (const void* const*)y2, (const float* const*)dz3, n_img, d.c3, g_w3, g_b3, accumulate, slab, slab_bytes, cd, st));
  CHECK(k_conv_dgrad(nprob, (const float* const*)dz3, w3, dz2, y2, n_img, d.c3, cd, st));
  // conv2
  CHECK(k_conv_wgrad<float>(nprob, (const void* const*)y1, (const float* const*)dz2, n_img, d.c2, g_w2, g_b2, accumulate, slab, slab_bytes, cd, st));
  CHECK(k_conv_dgrad(nprob, (const float* const*)dz2, w2, dz1, y1, n_img, d.c2, cd, st));
  // conv1 (no input gradient: images are data)
  if (img_dtype == TACORL_BF16)
    CHECK(k_conv_wgrad<__bf16>(nprob, img, (const float* const*)dz1, n_img, d.c1, g_w1, g_b1, accumulate, slab, slab_bytes, cd, st));
  else
    CHECK(k_conv_wgrad<float>(nprob, img, (const float* const*)dz1, n_img, d.c1, g_w1, g_b1, accumulate, slab, slab_bytes, cd, st));
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

// ---- bf16 / bf16-image / templated-geometry variant: FC tail and soft-argmax as above, the three
// convolutions' backward through the per-image LDS-resident kernels of encoder_bwd_fused.hip.
static const int kFcDims[3] = {128, 256, 32};
static size_t enc_fc_slab_bytes(int nprob, long maxn) {
  size_t a = wgrad_ws_bytes(nprob, 256, 32, maxn), b = wgrad_ws_bytes(nprob, 128, 256, maxn);
  int Mx[EBW_MAXP];  // the one-launch weight gradients of the FC tail (mlp_fused_wgrad): a record per slice
  for (int p = 0; p < nprob && p < EBW_MAXP; p++) Mx[p] = (int)maxn;
  const size_t c = mlp_fused_wgrad_slab_floats(nprob, Mx, 2, kFcDims) * sizeof(float);
  a = a > b ? a : b;
  return ((a > c ? a : c) + 255) & ~(size_t)255;
}
// Workspace of the fused encoder backward: [per-problem scratch d_h1 | d_sa | dz3 | .. | dtemp][FC W^T (bf16)]
// [FC wgrad slabs][conv backward workspace (ebw_ws_bytes)].
struct EncBwdPlan {
  EncDims d;
  long po[E_N];
  size_t scratch_bytes, fcwt_off[EBW_MAXP], fcwt_each, slab_off, slab_bytes, conv_off, total;
  long so[EBW_MAXP][6], ao[EBW_MAXP][5], sbase[EBW_MAXP];  // scratch / activation offsets (floats)
  long maxn;
};
static const int kFcActs[2] = {ACT_RELU, ACT_NONE};
static bool enc_bwd_plan(int nprob, const int* n_img, int H, int W, EncBwdPlan& pl) {
  if (!enc_dims(H, W, pl.d) || !ebw_supported(H, W) || nprob < 1 || nprob > EBW_MAXP) return false;
  tacorl_encoder_param_layout(pl.po);
  long tot = 0;
  pl.maxn = 0;
  for (int p = 0; p < nprob; p++) {
    pl.sbase[p] = tot;
    tot += enc_bwd_scratch_layout(n_img[p], pl.d, pl.so[p]);
    tacorl_encoder_act_layout(n_img[p], H, W, pl.ao[p]);
    pl.maxn = n_img[p] > pl.maxn ? n_img[p] : pl.maxn;
  }
  size_t b = ((size_t)tot * sizeof(float) + 255) & ~(size_t)255;
  pl.scratch_bytes = b;
  pl.fcwt_each = (mlp_fused_wt_elems(2, kFcDims, nullptr) * 2 + 255) & ~(size_t)255;
  for (int p = 0; p < nprob; p++) { pl.fcwt_off[p] = b; b += pl.fcwt_each; }
  pl.slab_off = b;
  pl.slab_bytes = enc_fc_slab_bytes(nprob, pl.maxn);
  b += pl.slab_bytes;
  pl.conv_off = b;
  pl.total = b + ebw_ws_bytes(nprob, n_img, H, W);
  return true;
}
extern "C" size_t tacorl_encoder_bwd_fused_ws_bytes(int nprob, const int* n_img, int H, int W) {
  EncBwdPlan pl;
  return enc_bwd_plan(nprob, n_img, H, W, pl) ? pl.total : 0;
}
static void enc_bwd_conv_problems(int nprob, const void* const* img, const float* const* params, const float* const* act,
                                  float* const* grads, const int* n_img, const EncBwdPlan& pl, void* ws, EbwProblem* pr) {
  for (int p = 0; p < nprob; p++) {
    const float* P = params[p];
    float* G = grads ? grads[p] : nullptr;
    float* sc = (float*)ws + pl.sbase[p];
    pr[p].img = img ? img[p] : nullptr;
    pr[p].y1 = act ? act[p] + pl.ao[p][0] : nullptr; pr[p].y2 = act ? act[p] + pl.ao[p][1] : nullptr;
    pr[p].dz3 = sc + pl.so[p][2];
    pr[p].w2 = P + pl.po[E_W2]; pr[p].w3 = P + pl.po[E_W3];
    pr[p].g_w1 = G ? G + pl.po[E_W1] : nullptr; pr[p].g_b1 = G ? G + pl.po[E_B1] : nullptr;
    pr[p].g_w2 = G ? G + pl.po[E_W2] : nullptr; pr[p].g_b2 = G ? G + pl.po[E_B2] : nullptr;
    pr[p].g_w3 = G ? G + pl.po[E_W3] : nullptr; pr[p].g_b3 = G ? G + pl.po[E_B3] : nullptr;
    pr[p].n = n_img[p];
  }
}
// FC tail (Linear 128->256 + ReLU -> Linear 256->32): its input-gradient chain through the fused MLP kernel.
// mode 1 = transpose the FC weights only, 2 = chain with the weights already transposed, 0 = both.
static int enc_bwd_fc_chain(int nprob, const float* const* params, const float* const* act, const float* const* d_out,
                            const int* n_img, const EncBwdPlan& pl, void* ws, hipStream_t st, int mode) {
  long wo[2], bo[2], src[MF_MAXP * MF_MAXL], dzo[MF_MAXP * MF_MAXL];
  tacorl_mlp_param_layout(2, kFcDims, wo, bo);
  const float* fcp[EBW_MAXP];
  float *dz[EBW_MAXP], *dsa[EBW_MAXP];
  void* wt[EBW_MAXP];
  for (int p = 0; p < nprob; p++) {
    fcp[p] = params[p] + pl.po[E_FW1];
    src[p * MF_MAXL + 0] = pl.ao[p][4];  // ReLU: derivative source = the layer output h1
    src[p * MF_MAXL + 1] = -1;
    dzo[p * MF_MAXL + 0] = pl.sbase[p] + pl.so[p][0];  // dZ of fc1 = d_h1
    dz[p] = (float*)ws;
    dsa[p] = (float*)ws + pl.sbase[p] + pl.so[p][1];
    wt[p] = (unsigned char*)ws + pl.fcwt_off[p];
  }
  return mlp_fused_bwd(nprob, fcp, act, d_out, 32, dz, dsa, 128, wt, n_img, 2, kFcDims, kFcActs, src, dzo, wo, st, mode);
}

/* Weight-dependent preparation of the fused encoder backward (transposed FC weights, conv W^T fragments):
 * depends only on the parameters, so a caller can run it early, off the dependent chain, and pass
 * prepacked = 1 to the calls below. */
extern "C" int tacorl_encoder_bwd_fused_pack(int nprob, const float* const* params, const int* n_img, int H, int W,
                                             void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  EncBwdPlan pl;
  if (!enc_bwd_plan(nprob, n_img, H, W, pl)) FAIL(TACORL_EINVAL, "encoder_bwd_fused_pack: geometry %dx%d / nprob %d", H, W, nprob);
  if (ws_bytes < pl.total) FAIL(TACORL_ENOMEM, "encoder_bwd_fused_pack: workspace too small");
  EbwProblem pr[EBW_MAXP];
  enc_bwd_conv_problems(nprob, nullptr, params, nullptr, nullptr, n_img, pl, ws, pr);
  int rc = enc_bwd_fc_chain(nprob, params, nullptr, nullptr, n_img, pl, ws, (hipStream_t)stream, 1);
  if (rc == TACORL_OK) rc = ebw_conv_backward(nprob, pr, H, W, 0, (unsigned char*)ws + pl.conv_off, pl.total - pl.conv_off, (hipStream_t)stream, 1);
  if (rc != TACORL_OK) FAIL(rc, "encoder_bwd_fused_pack: launch failed (%d)", rc);
  return TACORL_OK;
}
/* part 1 (dependent chain): d_out -> d(soft-argmax features) through the FC tail, one launch. */
extern "C" int tacorl_encoder_bwd_fused_head(int nprob, const float* const* params, const float* const* act,
                                             const float* const* d_out, const int* n_img, int H, int W, int prepacked,
                                             void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  EncBwdPlan pl;
  if (!enc_bwd_plan(nprob, n_img, H, W, pl)) FAIL(TACORL_EINVAL, "encoder_bwd_fused_head: geometry %dx%d / nprob %d", H, W, nprob);
  if (ws_bytes < pl.total) FAIL(TACORL_ENOMEM, "encoder_bwd_fused_head: workspace too small");
  const int rc = enc_bwd_fc_chain(nprob, params, act, d_out, n_img, pl, ws, (hipStream_t)stream, prepacked ? 2 : 0);
  if (rc != TACORL_OK) FAIL(rc, "encoder_bwd_fused_head: launch failed (%d)", rc);
  return TACORL_OK;
}
/* FC weight / bias gradients (needs part 1's d_h1; not on the dependent chain - any stream after part 1). */
extern "C" int tacorl_encoder_bwd_fused_fc_wgrad(int nprob, const float* const* act, const float* const* d_out,
                                                 float* const* grads, const int* n_img, int H, int W, int accumulate,
                                                 void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  EncBwdPlan pl;
  if (!enc_bwd_plan(nprob, n_img, H, W, pl)) FAIL(TACORL_EINVAL, "encoder_bwd_fused_fc_wgrad: geometry %dx%d / nprob %d", H, W, nprob);
  if (ws_bytes < pl.total) FAIL(TACORL_ENOMEM, "encoder_bwd_fused_fc_wgrad: workspace too small");
  const float *sa[EBW_MAXP], *h1[EBW_MAXP], *d_h1[EBW_MAXP];
  float *g_fw1[EBW_MAXP], *g_fb1[EBW_MAXP], *g_fw2[EBW_MAXP], *g_fb2[EBW_MAXP];
  for (int p = 0; p < nprob; p++) {
    sa[p] = act[p] + pl.ao[p][3]; h1[p] = act[p] + pl.ao[p][4];
    d_h1[p] = (const float*)ws + pl.sbase[p] + pl.so[p][0];
    float* G = grads[p];
    g_fw1[p] = G + pl.po[E_FW1]; g_fb1[p] = G + pl.po[E_FB1]; g_fw2[p] = G + pl.po[E_FW2]; g_fb2[p] = G + pl.po[E_FB2];
  }
  void* slab = (unsigned char*)ws + pl.slab_off;
  if (mlp_fused_wgrad_ok(nprob, 2, kFcDims) &&
      mlp_fused_wgrad_slab_floats(nprob, n_img, 2, kFcDims) * sizeof(float) <= pl.slab_bytes) {
    // both FC layers of every network in one launch + one reduce (they sit on the backward's dependent chain)
    long yoffs[MF_MAXP * MF_MAXL] = {}, dzoffs[MF_MAXP * MF_MAXL] = {};
    const float* dzb[EBW_MAXP];
    for (int p = 0; p < nprob; p++) {
      yoffs[p * MF_MAXL] = pl.ao[p][4];                 // fc1 output h1 inside act[p]
      dzoffs[p * MF_MAXL] = pl.sbase[p] + pl.so[p][0];  // d_h1 inside the workspace
      dzb[p] = (const float*)ws;
    }
    const long wo[2] = {pl.po[E_FW1], pl.po[E_FW2]}, bo[2] = {pl.po[E_FB1], pl.po[E_FB2]};
    if (mlp_fused_wgrad(nprob, sa, 128, act, d_out, 32, dzb, grads, (float*)slab, n_img, 2, kFcDims, yoffs, dzoffs, wo, bo,
                        accumulate, st))
      FAIL(TACORL_ELAUNCH, "encoder_bwd_fused_fc_wgrad: launch failed");
    return TACORL_OK;
  }
  CHECK(k_linear_wgrad(nprob, h1, 256, d_out, 32, n_img, 256, 32, g_fw2, g_fb2, accumulate, slab, pl.slab_bytes, TACORL_BF16, st));
  CHECK(k_linear_wgrad(nprob, sa, 128, d_h1, 256, n_img, 128, 256, g_fw1, g_fb1, accumulate, slab, pl.slab_bytes, TACORL_BF16, st));
  return TACORL_OK;
}
/* part 2 (dependent chain): soft-argmax backward (+ temperature gradient) and the three convolutions. */
extern "C" int tacorl_encoder_bwd_fused_conv(int nprob, const void* const* img, const float* const* params,
                                             const float* const* act, float* const* grads, const int* n_img, int H, int W,
                                             int accumulate, int prepacked, void* ws, size_t ws_bytes,
                                             tacorl_stream_t stream) {
  return tacorl_encoder_bwd_fused_conv_parts(nprob, img, params, act, grads, n_img, H, W, accumulate, prepacked, 127, ws,
                                             ws_bytes, stream);
}
extern "C" int tacorl_encoder_bwd_fused_conv_parts(int nprob, const void* const* img, const float* const* params,
                                                   const float* const* act, float* const* grads, const int* n_img, int H,
                                                   int W, int accumulate, int prepacked, int parts, void* ws,
                                                   size_t ws_bytes, tacorl_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  EncBwdPlan pl;
  if (!enc_bwd_plan(nprob, n_img, H, W, pl)) FAIL(TACORL_EINVAL, "encoder_bwd_fused_conv: geometry %dx%d / nprob %d", H, W, nprob);
  if (ws_bytes < pl.total) FAIL(TACORL_ENOMEM, "encoder_bwd_fused_conv: workspace too small");
  if (pl.d.c3.OH * pl.d.c3.OW > 4 * SAB_MAXI_HUGE) FAIL(TACORL_EINVAL, "encoder_bwd_fused: conv3 output too large");
  EbwProblem pr[EBW_MAXP];
  enc_bwd_conv_problems(nprob, img, params, act, grads, n_img, pl, ws, pr);
  SabArgs sb{};
  for (int p = 0; p < nprob; p++) {
    float* sc = (float*)ws + pl.sbase[p];
    sb.y3[p] = act[p] + pl.ao[p][2]; sb.temp[p] = params[p] + pl.po[E_T]; sb.sa[p] = act[p] + pl.ao[p][3];
    sb.d_sa[p] = sc + pl.so[p][1]; sb.dz3[p] = sc + pl.so[p][2]; sb.dtp[p] = sc + pl.so[p][5];
    sb.gtemp[p] = grads[p] + pl.po[E_T]; sb.n[p] = n_img[p];
  }
  // everything requested on one stream: soft-argmax backward, dgrad3 and wgrad3 are one launch (TACORL_EBW_FUSE3=0: the
  // separate launches, for A/B measurements)
  const char* fe = getenv("TACORL_EBW_FUSE3");  // (read per call: a captured graph keeps what it was captured with)
  const int fuse3 = fe ? atoi(fe) : 1;
  const bool fused3 = fuse3 && (parts & 127) == 127 && ebw_fused3_supported(H, W);
  if (fused3) {
    for (int p = 0; p < nprob; p++) {
      pr[p].y3 = sb.y3[p]; pr[p].temp = sb.temp[p]; pr[p].sa = sb.sa[p]; pr[p].d_sa = sb.d_sa[p];
      pr[p].dtp = sb.dtp[p]; pr[p].g_temp = sb.gtemp[p];
    }
  }
  if (pl.maxn > 0 && (parts & 1) && !fused3) {
    const int P3 = pl.d.c3.OH * pl.d.c3.OW;
    if (P3 <= 4 * 13) hipLaunchKernelGGL(softargmax_bwd_batch_kernel<13>, dim3((unsigned)pl.maxn, nprob), dim3(256), 0, st, sb, P3, pl.d.c3.OW);
    else if (P3 <= 4 * SAB_MAXI_BIG) hipLaunchKernelGGL(softargmax_bwd_batch_kernel<SAB_MAXI_BIG>, dim3((unsigned)pl.maxn, nprob), dim3(256), 0, st, sb, P3, pl.d.c3.OW);
    else hipLaunchKernelGGL(softargmax_bwd_batch_kernel<SAB_MAXI_HUGE>, dim3((unsigned)pl.maxn, nprob), dim3(256), 0, st, sb, P3, pl.d.c3.OW);
    hipLaunchKernelGGL(sum_to_scalar_batch_kernel, dim3(nprob), dim3(256), 0, st, sb, accumulate);
  }
  if (!(parts & EBW_ALL)) return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
  // (a partial call must not re-pack the W^T fragments behind a dgrad that is reading them: prepacked or EBW_ALL)
  if ((parts & EBW_ALL) != EBW_ALL && !prepacked) FAIL(TACORL_EINVAL, "encoder_bwd_fused_conv_parts: partial calls need prepacked fragments");
  const int rc = ebw_conv_backward(nprob, pr, H, W, accumulate, (unsigned char*)ws + pl.conv_off, pl.total - pl.conv_off, st, prepacked ? 2 : 0,
                                   (parts & EBW_ALL) | (fused3 ? EBW_FUSED3 : 0));
  if (rc != TACORL_OK) FAIL(rc, "encoder_bwd_fused_conv: conv backward launch failed (%d)", rc);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
/* everything on one stream: head, FC weight gradients, conv */
extern "C" int tacorl_encoder_bwd_fused(int nprob, const void* const* img, const float* const* params,
                                        const float* const* act, const float* const* d_out, float* const* grads,
                                        const int* n_img, int H, int W, int accumulate, void* ws, size_t ws_bytes,
                                        tacorl_stream_t stream) {
  int rc = tacorl_encoder_bwd_fused_head(nprob, params, act, d_out, n_img, H, W, 0, ws, ws_bytes, stream);
  if (rc == TACORL_OK) rc = tacorl_encoder_bwd_fused_fc_wgrad(nprob, act, d_out, grads, n_img, H, W, accumulate, ws, ws_bytes, stream);
  if (rc == TACORL_OK) rc = tacorl_encoder_bwd_fused_conv(nprob, img, params, act, grads, n_img, H, W, accumulate, 0, ws, ws_bytes, stream);
  return rc;
}

// ====================================================================== MLP
extern "C" long tacorl_mlp_param_layout(int L, const int* dims, long* w_off, long* b_off) {
  long off = 0;
  for (int l = 0; l < L; l++) {
    if (w_off) w_off[l] = off;
    off = al4(off + (long)dims[l + 1] * dims[l]);
    if (b_off) b_off[l] = off;
    off = al4(off + dims[l + 1]);
  }
  return off;
}
extern "C" long tacorl_mlp_act_layout(int M, int L, const int* dims, const int* acts, long* z_off, long* y_off) {
  long off = 0;
  for (int l = 0; l < L; l++) {
    if (z_off) z_off[l] = -1;
    if (acts[l] == ACT_SILU) { if (z_off) z_off[l] = off; off = al4(off + (long)M * dims[l + 1]); }
    if (y_off) y_off[l] = off;
    off = al4(off + (long)M * dims[l + 1]);
  }
  return off;
}
#define MLP_MAXL 8

extern "C" int tacorl_mlp_fwd(int nprob, const float* const* x, int ldx, const float* const* params,
                              float* const* act, const int* M, int L, const int* dims, const int* acts, int cd,
                              tacorl_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (L < 1 || L > MLP_MAXL || nprob < 1 || nprob > GEMM_MAXP) FAIL(TACORL_EINVAL, "mlp_fwd: bad L/nprob");
  long wo[MLP_MAXL], bo[MLP_MAXL];
  tacorl_mlp_param_layout(L, dims, wo, bo);
  for (int l = 0; l < L; l++) {
    const float *xin[GEMM_MAXP], *w[GEMM_MAXP], *b[GEMM_MAXP];
    float *y[GEMM_MAXP], *z[GEMM_MAXP];
    for (int p = 0; p < nprob; p++) {
      long zo[MLP_MAXL], yo[MLP_MAXL];
      tacorl_mlp_act_layout(M[p], L, dims, acts, zo, yo);
      xin[p] = l == 0 ? x[p] : act[p] + yo[l - 1];
      w[p] = params[p] + wo[l]; b[p] = params[p] + bo[l];
      y[p] = act[p] + yo[l]; z[p] = zo[l] >= 0 ? act[p] + zo[l] : nullptr;
    }
    CHECK(k_linear_fwd(nprob, xin, l == 0 ? ldx : dims[l], w, b, y, z, M, dims[l], dims[l + 1], dims[l + 1], acts[l], cd, st));
  }
  return TACORL_OK;
}

// Whole MLP forward in one launch (mlp_fused.hip): bf16 MFMA only; params_bf16[p] = bf16 copy of
// params[p] (tacorl_to_bf16_batch), which the caller refreshes whenever the fp32 block changes.
extern "C" int tacorl_mlp_fwd_fused_supported(int nprob, int L, const int* dims, int ldx) {
  return mlp_fused_fwd_ok(nprob, L, dims, ldx) ? 1 : 0;
}
// lean != 0: the output of a hidden layer whose pre-activation is saved (SiLU) is NOT written - the fused weight-gradient
// launch recomputes it from the pre-activation while staging (tacorl_mlp_bwd_fused_wgrad with the same flag; bit-identical
// operands): at 99 k rows the forward is bound by exactly these stores.  Only for callers whose backward is the fused
// pair (tacorl_mlp_lean_supported).
extern "C" int tacorl_mlp_lean_supported(int nprob, int L, const int* dims, int ldx, int ldo, int ldd) {
  return mlp_fused_fwd_ok(nprob, L, dims, ldx) && mlp_fused_bwd_ok(nprob, L, dims, ldo, ldd) && mlp_fused_wgrad_ok(nprob, L, dims) ? 1 : 0;
}
static int mlp_fwd_fused_impl(int nprob, const float* const* x, int ldx, const float* const* params,
                              const void* const* params_bf16, float* const* act, const int* M, int L,
                              const int* dims, const int* acts, int lean, tacorl_stream_t stream, const MlpXGather* gather) {
  if (!mlp_fused_fwd_ok(nprob, L, dims, ldx)) FAIL(TACORL_EINVAL, "mlp_fwd_fused: shapes not supported");
  long wo[MLP_MAXL], bo[MLP_MAXL];
  tacorl_mlp_param_layout(L, dims, wo, bo);
  // Many-row problems of a lean site (>= 16 384 rows: C5's Q networks) take kernels of their own and save, per hidden
  // layer, a bf16 copy of the output and an fp16 copy of act'(z) in the regions the fp32 z / y would occupy (mlp_fused.h);
  // the site's _dgrad (prepacked bit 1) and _wgrad (lean) calls read exactly that.
  const float *xs[2][MF_MAXP], *ps[2][MF_MAXP];
  const void* pb[2][MF_MAXP];
  float* as[2][MF_MAXP];
  int Ms[2][MF_MAXP], gm[2][MF_MAXP], n[2] = {0, 0};
  long zo[MF_MAXP * MF_MAXL], yo[MF_MAXP * MF_MAXL], ybf[MF_MAXP * MF_MAXL], sbf[MF_MAXP * MF_MAXL], yout[MF_MAXP];
  for (int p = 0; p < nprob; p++) {
    long z1[MLP_MAXL], y1[MLP_MAXL];
    tacorl_mlp_act_layout(M[p], L, dims, acts, z1, y1);
    const int b = (lean && mlp_big_prob_ok(M[p], L, dims, acts)) ? 1 : 0, q = n[b]++;
    gm[b][q] = p;
    xs[b][q] = x ? x[p] : nullptr; ps[b][q] = params[p]; pb[b][q] = params_bf16[p]; as[b][q] = act[p]; Ms[b][q] = M[p];
    for (int l = 0; l < L; l++) {
      if (b) { ybf[q * MF_MAXL + l] = y1[l]; sbf[q * MF_MAXL + l] = z1[l]; }
      else {
        zo[q * MF_MAXL + l] = z1[l];
        yo[q * MF_MAXL + l] = (lean && l + 1 < L && z1[l] >= 0) ? -1 : y1[l];
      }
    }
    if (b) yout[q] = y1[L - 1];
  }
  if (n[1]) {
    const int rc = mlp_big_fwd(n[1], xs[1], ldx, ps[1], pb[1], as[1], Ms[1], L, dims, acts, ybf, sbf, yout, wo, bo, (hipStream_t)stream, gather, gm[1]);
    if (rc != TACORL_OK) FAIL(rc, "mlp_fwd_fused: many-row launch failed (%d)", rc);
  }
  if (n[0]) {
    const int rc = mlp_fused_fwd(n[0], xs[0], ldx, ps[0], pb[0], as[0], Ms[0], L, dims, acts, zo, yo, wo, bo, (hipStream_t)stream, gather, gm[0]);
    if (rc != TACORL_OK) FAIL(rc, "mlp_fwd_fused: launch failed (%d)", rc);
  }
  return TACORL_OK;
}
extern "C" int tacorl_mlp_fwd_fused(int nprob, const float* const* x, int ldx, const float* const* params,
                                    const void* const* params_bf16, float* const* act, const int* M, int L,
                                    const int* dims, const int* acts, int lean, tacorl_stream_t stream) {
  return mlp_fwd_fused_impl(nprob, x, ldx, params, params_bf16, act, M, L, dims, acts, lean, stream, nullptr);
}
/* tacorl_mlp_fwd_fused with a GATHERED layer-0 input: row r of problem p is the concatenation of nseg[p] <= 4 column
 * segments, segment t = columns [seg_c0[4p+t], seg_c0[4p+t+1]) (the last: up to dims[0]) read from
 * seg_ptr[4p+t][(seg_mod ? r % seg_mod : r) * seg_ld + col - seg_c0] - the reference's torch.cat([enc(obs),
 * goal_enc(enc(goal))]) (visual_actor_wrapper.py:41-62), cat(emb, action) (critic.py:92-97) and expand_obs
 * (utils/misc.py:132-153: state rows repeated per sampled action = seg_mod B) without a copy launch in front of the MLP.
 * x_out[p] != NULL: the assembled fp32 rows are also written to x_out[p] ([M][ldx]) - layer 0's operand of the
 * weight-gradient launch.  seg_c0 % 8 == 0, seg_ld % 4 == 0, pointers 16-byte aligned.  Not for many-row problems
 * (tacorl_mlp_fwd_fused_gather_supported == 0: assemble with tacorl_copy_cols_batch and call tacorl_mlp_fwd_fused). */
extern "C" int tacorl_mlp_fwd_fused_gather_supported(int nprob, const int* M, int L, const int* dims, const int* acts, int ldx,
                                                     int lean) {
  (void)M; (void)acts; (void)lean;  // (the many-row kernels gather too since round 5)
  return mlp_fused_fwd_ok(nprob, L, dims, ldx) ? 1 : 0;
}
extern "C" int tacorl_mlp_fwd_fused_gather(int nprob, const int* nseg, const float* const* seg_ptr, const int* seg_ld,
                                           const int* seg_c0, const int* seg_mod, float* const* x_out, int ldx,
                                           const float* const* params, const void* const* params_bf16, float* const* act,
                                           const int* M, int L, const int* dims, const int* acts, int lean,
                                           tacorl_stream_t stream) {
  if (nprob < 1 || nprob > MF_MAXP || !nseg || !seg_ptr || !seg_ld || !seg_c0 || !seg_mod) FAIL(TACORL_EINVAL, "mlp_fwd_fused_gather: arguments");
  MlpXGather g{};
  for (int p = 0; p < nprob; p++) {
    if (nseg[p] < 1 || nseg[p] > MF_MAXSEG) FAIL(TACORL_EINVAL, "mlp_fwd_fused_gather: nseg[%d] = %d", p, nseg[p]);
    g.nseg[p] = nseg[p];
    g.xw[p] = x_out ? x_out[p] : nullptr;
    for (int t = 0; t < nseg[p]; t++) {
      g.ptr[p][t] = seg_ptr[MF_MAXSEG * p + t]; g.ld[p][t] = seg_ld[MF_MAXSEG * p + t];
      g.c0[p][t] = seg_c0[MF_MAXSEG * p + t]; g.mod[p][t] = seg_mod[MF_MAXSEG * p + t];
    }
  }
  return mlp_fwd_fused_impl(nprob, nullptr, ldx, params, params_bf16, act, M, L, dims, acts, lean, stream, &g);
}
struct ToBf16Tbl { const float* src[16]; __bf16* dst[16]; long n4[16]; };
__global__ void to_bf16_batch_kernel(ToBf16Tbl t) {
  const int b = blockIdx.y;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < t.n4[b]; q += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(t.src[b])[q];
    reinterpret_cast<bf16x4*>(t.dst[b])[q] = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  }
}
extern "C" int tacorl_to_bf16_batch(int n, const float* const* src, void* const* dst, const long* count,
                                    tacorl_stream_t stream) {
  if (n < 1 || n > 16) FAIL(TACORL_EINVAL, "to_bf16_batch: n %d", n);
  ToBf16Tbl t{};
  long mx = 0;
  for (int i = 0; i < n; i++) {
    if (count[i] % 4 || !aligned16(src[i]) || ((uintptr_t)dst[i] & 7)) FAIL(TACORL_EINVAL, "to_bf16_batch: count %% 4 and alignment");
    t.src[i] = src[i]; t.dst[i] = (__bf16*)dst[i]; t.n4[i] = count[i] / 4;
    mx = t.n4[i] > mx ? t.n4[i] : mx;
  }
  if (mx == 0) return TACORL_OK;
  if (prep_deferring()) {  // part of the step's one weight-preparation launch (tacorl_prep_batch_end)
    for (int i = 0; i < n; i++)
      if (t.n4[i]) { const int rc = prep_defer_bf16(t.src[i], t.dst[i], 4 * t.n4[i], (hipStream_t)stream); if (rc != TACORL_OK) return rc; }
    return TACORL_OK;
  }
  hipLaunchKernelGGL(to_bf16_batch_kernel, dim3((unsigned)(cdiv(mx, 256) > 1024 ? 1024 : cdiv(mx, 256)), n), dim3(256), 0,
                     (hipStream_t)stream, t);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

// ---- MLP backward, bf16 mode: the input-gradient chain as one launch (mlp_fused.hip) and the weight
// gradients as a separate call, so that a caller can take them off the dependent chain (another stream).
// Workspace: [dZ_l of every problem, l = 0..L-2][transposed bf16 weights][wgrad slabs].
struct MlpBwdWs { long dzoff[MF_MAXP * MF_MAXL]; size_t dz_floats, wt_off[MF_MAXP], wt_bytes_each, xb_off[MF_MAXP], slab_off, slab_bytes, total; };
static MlpBwdWs mlp_bwd_fused_plan(int nprob, const int* M, int L, const int* dims) {
  MlpBwdWs w{};
  long off = 0, maxM = 0;
  for (int p = 0; p < nprob; p++) {
    for (int l = 0; l + 1 < L; l++) { w.dzoff[p * MF_MAXL + l] = off; off += al4((long)M[p] * dims[l + 1]); }
    maxM = M[p] > maxM ? M[p] : maxM;
  }
  w.dz_floats = (size_t)off;
  size_t b = ((size_t)off * sizeof(float) + 255) & ~(size_t)255;
  w.wt_bytes_each = (mlp_fused_wt_elems(L, dims, nullptr) * 2 + 255) & ~(size_t)255;
  for (int p = 0; p < nprob; p++) { w.wt_off[p] = b; b += w.wt_bytes_each; }
  int Mb[MF_MAXP], nb = 0;  // many-row problems: layer 0's input as a bf16 copy for the LDS-DMA weight gradients
  for (int p = 0; p < nprob; p++) {
    w.xb_off[p] = b;
    if (mlp_big_prob_ok(M[p], L, dims, nullptr)) { b += (mlp_big_xb_bytes(M[p]) + 255) & ~(size_t)255; Mb[nb++] = M[p]; }
  }
  w.slab_off = b;
  for (int l = 0; l < L; l++) { size_t s = wgrad_ws_bytes(nprob, dims[l], dims[l + 1], maxM); w.slab_bytes = s > w.slab_bytes ? s : w.slab_bytes; }
  if (mlp_fused_wgrad_ok(nprob, L, dims)) {  // one-launch weight gradients: a record of every layer per 256-row slice
    const size_t s = mlp_fused_wgrad_slab_floats(nprob, M, L, dims) * sizeof(float);
    w.slab_bytes = s > w.slab_bytes ? s : w.slab_bytes;
  }
  if (nb) {
    const size_t s = mlp_big_wgrad_slab_floats(nb, Mb, L, dims) * sizeof(float);
    w.slab_bytes = s > w.slab_bytes ? s : w.slab_bytes;
  }
  w.total = b + w.slab_bytes;
  return w;
}
extern "C" int tacorl_mlp_bwd_fused_supported(int nprob, int L, const int* dims, int ldo, int ldd) {
  return mlp_fused_bwd_ok(nprob, L, dims, ldo, ldd) ? 1 : 0;
}
extern "C" size_t tacorl_mlp_bwd_fused_ws_bytes(int nprob, const int* M, int L, const int* dims) {
  if (nprob < 1 || nprob > MF_MAXP || L < 1 || L > MF_MAXL) return 0;
  return mlp_bwd_fused_plan(nprob, M, L, dims).total;
}
extern "C" int tacorl_mlp_bwd_fused_dgrad(int nprob, const float* const* params, const float* const* act,
                                          const float* const* d_out, int ldo, float* const* d_x, int ldd, const int* M,
                                          int L, const int* dims, const int* acts, int prepacked, void* ws,
                                          size_t ws_bytes, tacorl_stream_t stream) {
  if (!mlp_fused_bwd_ok(nprob, L, dims, ldo, ldd)) FAIL(TACORL_EINVAL, "mlp_bwd_fused: shapes not supported");
  if (acts[L - 1] != ACT_NONE) FAIL(TACORL_EINVAL, "mlp_bwd_fused: last activation must be NONE");
  const MlpBwdWs w = mlp_bwd_fused_plan(nprob, M, L, dims);
  if (ws_bytes < w.total) FAIL(TACORL_ENOMEM, "mlp_bwd_fused: workspace too small");
  long wo[MLP_MAXL], bo[MLP_MAXL], src[MF_MAXP * MF_MAXL];
  tacorl_mlp_param_layout(L, dims, wo, bo);
  // prepacked: bit 0 = the weight transposes are in ws already; bit 1 = the caller's forward and weight gradients run lean:
  // many-row problems then read the fp16 act' copies of the forward and hand dZ_l over as bf16 (mlp_fused.h)
  const float *ps[2][MF_MAXP], *as[2][MF_MAXP], *ds[2][MF_MAXP];
  float *dzs[2][MF_MAXP], *dxs[2][MF_MAXP];
  void* wts[2][MF_MAXP];
  int Ms[2][MF_MAXP], n[2] = {0, 0};
  long dzo[2][MF_MAXP * MF_MAXL], sbf[MF_MAXP * MF_MAXL];
  for (int p = 0; p < nprob; p++) {
    long zo[MLP_MAXL], yo[MLP_MAXL];
    tacorl_mlp_act_layout(M[p], L, dims, acts, zo, yo);
    const int b = ((prepacked & 2) && mlp_big_prob_ok(M[p], L, dims, acts)) ? 1 : 0, q = n[b]++;
    ps[b][q] = params[p]; as[b][q] = act[p]; ds[b][q] = d_out[p]; dzs[b][q] = (float*)ws; dxs[b][q] = d_x ? d_x[p] : nullptr;
    wts[b][q] = (unsigned char*)ws + w.wt_off[p]; Ms[b][q] = M[p];
    for (int l = 0; l < L; l++) {
      dzo[b][q * MF_MAXL + l] = w.dzoff[p * MF_MAXL + l];
      if (b) sbf[q * MF_MAXL + l] = zo[l];
      else src[q * MF_MAXL + l] = acts[l] == ACT_SILU ? zo[l] : (acts[l] == ACT_RELU ? yo[l] : -1);
    }
  }
  hipStream_t st = (hipStream_t)stream;
  if (n[1]) {
    if (!(prepacked & 1)) {
      const int rc = mlp_fused_bwd(n[1], ps[1], nullptr, nullptr, 0, dzs[1], nullptr, 0, wts[1], Ms[1], L, dims, acts, sbf, dzo[1], wo, st, 1);
      if (rc != TACORL_OK) FAIL(rc, "mlp_bwd_fused: weight transposes failed (%d)", rc);
    }
    const int rc = mlp_big_bwd(n[1], as[1], ds[1], ldo, dzs[1], dxs[1], ldd, wts[1], Ms[1], L, dims, sbf, dzo[1], st);
    if (rc != TACORL_OK) FAIL(rc, "mlp_bwd_fused: many-row launch failed (%d)", rc);
  }
  if (n[0]) {
    const int rc = mlp_fused_bwd(n[0], ps[0], as[0], ds[0], ldo, dzs[0], dxs[0], ldd, wts[0], Ms[0], L, dims, acts, src, dzo[0], wo,
                                 st, (prepacked & 1) ? 2 : 0);
    if (rc != TACORL_OK) FAIL(rc, "mlp_bwd_fused: launch failed (%d)", rc);
  }
  return TACORL_OK;
}
/* The weight transposes _dgrad needs, alone: they depend only on the parameters, so a caller can run them
 * early / on another stream and pass prepacked = 1. */
extern "C" int tacorl_mlp_bwd_fused_pack(int nprob, const float* const* params, const int* M, int L, const int* dims,
                                         void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  if (nprob < 1 || nprob > MF_MAXP || L < 1 || L > MF_MAXL) FAIL(TACORL_EINVAL, "mlp_bwd_fused_pack: bad L/nprob");
  const MlpBwdWs w = mlp_bwd_fused_plan(nprob, M, L, dims);
  if (ws_bytes < w.total) FAIL(TACORL_ENOMEM, "mlp_bwd_fused_pack: workspace too small");
  long wo[MLP_MAXL], bo[MLP_MAXL], zero[MF_MAXP * MF_MAXL] = {0};
  int acts[MLP_MAXL] = {0};
  tacorl_mlp_param_layout(L, dims, wo, bo);
  float* dz[MF_MAXP];
  void* wt[MF_MAXP];
  for (int p = 0; p < nprob; p++) { dz[p] = (float*)ws; wt[p] = (unsigned char*)ws + w.wt_off[p]; }
  const int rc = mlp_fused_bwd(nprob, params, nullptr, nullptr, 0, dz, nullptr, 0, wt, M, L, dims, acts, zero, zero, wo,
                               (hipStream_t)stream, 1);
  if (rc != TACORL_OK) FAIL(rc, "mlp_bwd_fused_pack: launch failed (%d)", rc);
  return TACORL_OK;
}
extern "C" int tacorl_mlp_bwd_fused_wgrad(int nprob, const float* const* x, int ldx, const float* const* act,
                                          const float* const* d_out, int ldo, float* const* grads, const int* M, int L,
                                          const int* dims, const int* acts, int accumulate, int lean, void* ws, size_t ws_bytes,
                                          tacorl_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (nprob < 1 || nprob > MF_MAXP || L < 1 || L > MF_MAXL) FAIL(TACORL_EINVAL, "mlp_bwd_fused_wgrad: bad L/nprob");
  const MlpBwdWs w = mlp_bwd_fused_plan(nprob, M, L, dims);
  if (ws_bytes < w.total) FAIL(TACORL_ENOMEM, "mlp_bwd_fused_wgrad: workspace too small");
  long wo[MLP_MAXL], bo[MLP_MAXL];
  tacorl_mlp_param_layout(L, dims, wo, bo);
  void* slab = (unsigned char*)ws + w.slab_off;
  // many-row problems of a lean site: LDS-DMA over the bf16 copies left by the forward and the (lean-flagged) dgrad launch;
  // the others: the one-launch form / per-layer GEMMs below (both groups use the slab region, one after the other)
  const float *xs[2][MF_MAXP], *as[2][MF_MAXP], *ds[2][MF_MAXP], *dzp[2][MF_MAXP];
  float* gs[2][MF_MAXP];
  void* xb[MF_MAXP];
  int Ms[2][MF_MAXP], n[2] = {0, 0};
  long yoffs[2][MF_MAXP * MF_MAXL], dzo[2][MF_MAXP * MF_MAXL];
  for (int p = 0; p < nprob; p++) {
    long zo[MLP_MAXL], yo[MLP_MAXL];
    tacorl_mlp_act_layout(M[p], L, dims, acts, zo, yo);
    const int b = (lean && mlp_big_prob_ok(M[p], L, dims, acts)) ? 1 : 0, q = n[b]++;
    xs[b][q] = x[p]; as[b][q] = act[p]; ds[b][q] = d_out[p]; dzp[b][q] = (const float*)ws; gs[b][q] = grads[p]; Ms[b][q] = M[p];
    if (b) xb[q] = (unsigned char*)ws + w.xb_off[p];
    for (int l = 0; l < L; l++) {
      dzo[b][q * MF_MAXL + l] = w.dzoff[p * MF_MAXL + l];
      yoffs[b][q * MF_MAXL + l] = b ? yo[l] : ((lean && l + 1 < L && zo[l] >= 0) ? -(zo[l] + 1) : yo[l]);
    }
  }
  if (n[1] && mlp_fused_wgrad_big(n[1], xs[1], ldx, as[1], ds[1], ldo, dzp[1], gs[1], (float*)slab, xb, Ms[1], L, dims, yoffs[1],
                                  dzo[1], wo, bo, accumulate, st, w.slab_bytes / sizeof(float)))
    FAIL(TACORL_ELAUNCH, "mlp_bwd_fused_wgrad: many-row launch failed (or its records exceed the planned slab)");
  if (!n[0]) return TACORL_OK;
  if (mlp_fused_wgrad_ok(n[0], L, dims)) {  // every layer and network in one launch (+ one reduce)
    if (mlp_fused_wgrad(n[0], xs[0], ldx, as[0], ds[0], ldo, dzp[0], gs[0], (float*)slab, Ms[0], L, dims, yoffs[0], dzo[0], wo, bo,
                        accumulate, st, acts))
      FAIL(TACORL_ELAUNCH, "mlp_bwd_fused_wgrad: launch failed");
    return TACORL_OK;
  }
  if (lean) FAIL(TACORL_EINVAL, "mlp_bwd_fused_wgrad: lean activations need the one-launch form (tacorl_mlp_lean_supported)");
  for (int l = L - 1; l >= 0; l--) {
    const float *xg[GEMM_MAXP], *dzg[GEMM_MAXP];
    float *dwg[GEMM_MAXP], *dbg[GEMM_MAXP];
    int Mc[GEMM_MAXP], n2 = 0;
    for (int q = 0; q < n[0]; q++) {
      if (!gs[0][q] || Ms[0][q] <= 0) continue;
      long zo[MLP_MAXL], yo[MLP_MAXL];
      tacorl_mlp_act_layout(Ms[0][q], L, dims, acts, zo, yo);
      xg[n2] = l == 0 ? xs[0][q] : as[0][q] + yo[l - 1];
      dzg[n2] = l == L - 1 ? ds[0][q] : (const float*)ws + dzo[0][q * MF_MAXL + l];
      dwg[n2] = gs[0][q] + wo[l]; dbg[n2] = gs[0][q] + bo[l]; Mc[n2] = Ms[0][q]; n2++;
    }
    if (n2) CHECK(k_linear_wgrad(n2, xg, l == 0 ? ldx : dims[l], dzg, l == L - 1 ? ldo : dims[l + 1], Mc, dims[l], dims[l + 1],
                                 dwg, dbg, accumulate, slab, w.slab_bytes, TACORL_BF16, st));
  }
  return TACORL_OK;
}

static void mlp_bwd_sizes(int nprob, const int* M, int L, const int* dims, long& dz_floats, size_t& slab) {
  int maxd = 0;
  for (int l = 0; l <= L; l++) maxd = dims[l] > maxd ? dims[l] : maxd;
  long maxM = 0;
  dz_floats = 0;
  for (int p = 0; p < nprob; p++) { dz_floats += 2 * al4((long)M[p] * maxd); maxM = M[p] > maxM ? M[p] : maxM; }
  slab = 0;
  for (int l = 0; l < L; l++) { size_t s = wgrad_ws_bytes(nprob, dims[l], dims[l + 1], maxM); slab = s > slab ? s : slab; }
}
extern "C" size_t tacorl_mlp_bwd_ws_bytes(int nprob, const int* M, int L, const int* dims) {
  long dz; size_t slab;
  mlp_bwd_sizes(nprob, M, L, dims, dz, slab);
  return (size_t)dz * sizeof(float) + slab;
}

extern "C" int tacorl_mlp_bwd(int nprob, const float* const* x, int ldx, const float* const* params,
                              const float* const* act, const float* const* d_out, int ldo, float* const* grads,
                              float* const* d_x, int ldd, const int* M, int L, const int* dims, const int* acts,
                              int cd, int accumulate, void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (L < 1 || L > MLP_MAXL || nprob < 1 || nprob > GEMM_MAXP) FAIL(TACORL_EINVAL, "mlp_bwd: bad L/nprob");
  if (acts[L - 1] != ACT_NONE) FAIL(TACORL_EINVAL, "mlp_bwd: last activation must be NONE");
  long dzf; size_t slab_bytes;
  mlp_bwd_sizes(nprob, M, L, dims, dzf, slab_bytes);
  if (ws_bytes < (size_t)dzf * sizeof(float) + slab_bytes) FAIL(TACORL_ENOMEM, "mlp_bwd: workspace too small");
  int maxd = 0;
  for (int l = 0; l <= L; l++) maxd = dims[l] > maxd ? dims[l] : maxd;
  long wo[MLP_MAXL], bo[MLP_MAXL];
  tacorl_mlp_param_layout(L, dims, wo, bo);
  float* buf[2][GEMM_MAXP];
  float* cur = (float*)ws;
  for (int p = 0; p < nprob; p++) { buf[0][p] = cur; cur += al4((long)M[p] * maxd); buf[1][p] = cur; cur += al4((long)M[p] * maxd); }
  void* slab = cur;
  bool any_grads = false, any_dx = false;
  for (int p = 0; p < nprob; p++) { any_grads |= grads && grads[p]; any_dx |= d_x && d_x[p]; }
  const float* dz[GEMM_MAXP];
  for (int p = 0; p < nprob; p++) dz[p] = d_out[p];
  for (int l = L - 1; l >= 0; l--) {
    const float *xin[GEMM_MAXP], *w[GEMM_MAXP], *src[GEMM_MAXP];
    float *dw[GEMM_MAXP], *db[GEMM_MAXP], *out[GEMM_MAXP];
    int Mg[GEMM_MAXP], Md[GEMM_MAXP];
    for (int p = 0; p < nprob; p++) {
      long zo[MLP_MAXL], yo[MLP_MAXL];
      tacorl_mlp_act_layout(M[p], L, dims, acts, zo, yo);
      xin[p] = l == 0 ? x[p] : act[p] + yo[l - 1];
      w[p] = params[p] + wo[l];
      const bool hg = grads && grads[p];
      dw[p] = hg ? grads[p] + wo[l] : nullptr; db[p] = hg ? grads[p] + bo[l] : nullptr;
      Mg[p] = hg ? M[p] : 0;
      if (l > 0) {
        src[p] = acts[l - 1] == ACT_SILU ? act[p] + zo[l - 1] : (acts[l - 1] == ACT_RELU ? act[p] + yo[l - 1] : nullptr);
        out[p] = buf[l & 1][p]; Md[p] = M[p];
      } else {
        src[p] = nullptr; out[p] = d_x ? d_x[p] : nullptr; Md[p] = out[p] ? M[p] : 0;
      }
    }
    if (any_grads) {
      // problems without grads get M = 0 rows: their slabs reduce to zeros, so point them at scratch
      const float* xg[GEMM_MAXP]; const float* dzg[GEMM_MAXP]; float* dwg[GEMM_MAXP]; float* dbg[GEMM_MAXP]; int Mc[GEMM_MAXP];
      int n2 = 0;
      for (int p = 0; p < nprob; p++) if (Mg[p] > 0) { xg[n2] = xin[p]; dzg[n2] = dz[p]; dwg[n2] = dw[p]; dbg[n2] = db[p]; Mc[n2] = Mg[p]; n2++; }
      if (n2) CHECK(k_linear_wgrad(n2, xg, l == 0 ? ldx : dims[l], dzg, l == L - 1 ? ldo : dims[l + 1], Mc, dims[l], dims[l + 1], dwg, dbg, accumulate, slab, slab_bytes, cd, st));
    }
    if (l > 0 || any_dx) {
      const float* dzd[GEMM_MAXP]; const float* wd[GEMM_MAXP]; const float* srcd[GEMM_MAXP]; float* outd[GEMM_MAXP]; int Mc[GEMM_MAXP];
      int n2 = 0;
      for (int p = 0; p < nprob; p++) if (Md[p] > 0) { dzd[n2] = dz[p]; wd[n2] = w[p]; srcd[n2] = src[p]; outd[n2] = out[p]; Mc[n2] = Md[p]; n2++; }
      if (n2) CHECK(k_linear_dgrad(n2, dzd, l == L - 1 ? ldo : dims[l + 1], wd, outd, l == 0 ? ldd : dims[l], srcd, l > 0 ? acts[l - 1] : ACT_NONE, Mc, dims[l + 1], dims[l], cd, st));
    }
    if (l > 0) for (int p = 0; p < nprob; p++) dz[p] = buf[l & 1][p];
  }
  return TACORL_OK;
}
