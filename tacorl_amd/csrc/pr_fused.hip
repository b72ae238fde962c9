// Plan-recognition transformer encoder, inference, in ONE launch (reference
// networks/plan_encoders/plan_recognition_transformer.py:36-88: learned position embedding + N post-norm
// nn.TransformerEncoderLayer (ReLU FFN) + mean over time; TACORL runs it frozen / eval, so no dropout and
// nothing to save for a backward).
//
// As separate kernels a layer is 8-9 launches (in-proj, attention, out-proj, add+LayerNorm, FFN1,
// split-K FFN2 + reduce, add+LayerNorm) of a few microseconds of work each: ~180 us for two layers on the
// step's longest dependent chain.  With d_model 32 and T = 16 a whole sequence is one 16-row MFMA tile, so
// here one WORKGROUP owns one sequence for all layers: activations stay in registers (MFMA D layout: lane =
// row, 4 consecutive features) and wave-private LDS scratch.  The four waves run the attention half of a
// layer redundantly (no synchronisation) and split the FFN by hidden chunks; their partial outputs meet in
// LDS behind the only workgroup barrier of a layer.  Weights are the MFMA A operand, streamed from global
// memory (FFN matrices from a bf16 mirror of the parameter block, one phase ahead of their use).
// Optionally the posterior head (fc -> mean_fc composed into one 2A x 32 affine map) and the plan sample
// ride in the same launch.  B = 256: 34 us (one wave per sequence on a quarter of the CUs: 81 us).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tacorl_hip.h"
#include "common.h"

namespace {

#ifndef PR_W2TOP
#define PR_W2TOP 0  // 1: 510 registers - nothing else fits on a CU beside a plan-recognition workgroup (see DESIGN, round 5)
#endif
constexpr int PR_D = 32, PR_T = 16, PR_H = 8, PR_HD = 4, PR_MAXL = 4, PR_CH = 256;
constexpr int XB_P = 40;    // bf16 row pitch of the 32-wide MFMA B operand (80 B: conflict-free b128 reads)
constexpr int HB_P = 264;   // bf16 row pitch of one FFN hidden chunk
constexpr int QKV_P = 100;  // fp32 row pitch of q|k|v

struct PrLayerOff { long in_w, in_b, out_w, out_b, w1, b1, w2, b2, n1w, n1b, n2w, n2b; };
// train mode (PlayLMP, no dropout): what the per-op backward (networks/plan_recognition.py) reads of a layer, written by this
// launch in that path's layouts - batch-major rows r = b T + t, fp32: the layer input, q|k|v, the attention output, the
// out-projection (pre-residual), LayerNorm-1 output and {mean, rstd}, the post-ReLU FFN hidden state, the FFN output
// (pre-residual), LayerNorm-2 {mean, rstd}
struct PrSaveL { float *xin, *qkv, *att, *proj, *x1, *ff1, *ff2, *st1, *st2; };
struct PrArgs {
  const float* emb;     // [B*T][ld_emb], first 32 columns used
  const float* P;       // fp32 parameter block
  const __bf16* Pb;     // bf16 copy of the block (same element offsets)
  float* pooled;        // [B][32] mean over time of the last layer's output
  long pos;             // position_embeddings offset
  PrLayerOff l[PR_MAXL];
  int ld_emb, B, FF, L;
  // optional posterior head + sample in the same launch (all null / 0 when unused)
  const float* Wc;   // [2A][32] composed fc -> mean_fc weight (tacorl_pr_head_compose)
  const float* bc;   // [2A]
  const float* eps;  // [B][A]
  float* head;       // [B][2A]  (mean | var_raw)
  float* plan;       // [B][A]   tanh(mean + eps * std)
  int A;
  float min_std;
  PrSaveL sv[PR_MAXL];
  int save;
};

__device__ __forceinline__ bf16x8 cvt8(const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  return bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
}
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// LayerNorm over the 32 features of each row; v[nt][r] = feature 16 nt + 4 g + r of row i (lanes i, i+16, i+32, i+48)
__device__ __forceinline__ void layer_norm32(f32x4 (&v)[2], const float* w, const float* b, int g, float* st = nullptr) {
  float s = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) s += v[nt][r];
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  const float mean = s / 32.f;
  float q = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) { const float c = v[nt][r] - mean; q += c * c; }
  q += __shfl_xor(q, 16, 64);
  q += __shfl_xor(q, 32, 64);
  const float rstd = 1.0f / sqrtf(q / 32.f + 1e-5f);
  if (st && g == 0) { st[0] = mean; st[1] = rstd; }
#pragma unroll
  for (int nt = 0; nt < 2; nt++) {
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + 16 * nt + 4 * g), bb = *reinterpret_cast<const f32x4*>(b + 16 * nt + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; r++) v[nt][r] = (v[nt][r] - mean) * rstd * ww[r] + bb[r];
  }
}

// RT: 16-row tiles per sequence (T = 16 RT: window 16 - one MFMA tile, as the reference's default - or 32, the real-world
// configuration's window: BASELINE configs[3]; round 4 - at T = 32 the per-op path put 349 us of launches in front of BOTH
// chains of the C4 step).  Everything row-shaped simply exists RT times per lane; the attention has 8 T (head, query) pairs.
// the same with the weight / bias vectors already in registers (fetched at the top of the layer)
__device__ __forceinline__ void layer_norm32_r(f32x4 (&v)[2], const f32x4 (&ww)[2], const f32x4 (&bb)[2], int g, float* st = nullptr) {
  float s = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) s += v[nt][r];
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  const float mean = s / 32.f;
  float q = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) { const float c = v[nt][r] - mean; q += c * c; }
  q += __shfl_xor(q, 16, 64);
  q += __shfl_xor(q, 32, 64);
  const float rstd = 1.0f / sqrtf(q / 32.f + 1e-5f);
  if (st && g == 0) { st[0] = mean; st[1] = rstd; }
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) v[nt][r] = (v[nt][r] - mean) * rstd * ww[nt][r] + bb[nt][r];
}

// Phase clocks of the inference launches (scratch builds with -DPR_STAMPS only): wave 0 of workgroup 0 accumulates shader-clock
// deltas per phase; read back (and reset) with tacorl_pr_stamps_read.
#ifdef PR_STAMPS
__device__ unsigned long long pr_stamps[16];
#define PR_STAMP(k)                                          \
  do {                                                       \
    if (blockIdx.x == 0 && threadIdx.x == 0) {               \
      const unsigned long long t_ = clock64();               \
      atomicAdd(&pr_stamps[k], t_ - t_prev);                 \
      t_prev = t_;                                           \
    }                                                        \
  } while (0)
#else
#define PR_STAMP(k)
#endif

template <int RT>
__global__ __launch_bounds__(256) void pr_encoder_fused_kernel(PrArgs a) {
#ifdef PR_STAMPS
  unsigned long long t_prev = clock64();
#endif
  constexpr int T = 16 * RT;
  __shared__ __attribute__((aligned(16))) __bf16 xb_s[4][T * XB_P];
  __shared__ __attribute__((aligned(16))) unsigned char big_s[4][T * HB_P * 2];  // q|k|v (fp32) or FFN hidden chunk (bf16)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  // One workgroup per sequence.  Its four waves each run the (cheap) attention half of a layer redundantly on
  // private scratch and split the FFN - two thirds of the work, 256 KB of weights per layer - by hidden chunks
  // (wave w: chunks w, w + 4, ...); the partial FFN outputs meet in LDS once per layer.  B workgroups instead
  // of B / 4: at B = 256 the launch covers every CU instead of a quarter of them.
  __shared__ __attribute__((aligned(16))) float ypart[2][4][T * PR_D];  // [layer parity][wave]
  // this wave's FFN1 biases of the layer (its chunks w, w + 4: up to PR_B1C chunks of 256), fetched with the layer's other
  // operands and parked here: read where the 16 column tiles of a chunk need them, they were one more exposed global
  // round trip per chunk
  constexpr int PR_B1C = 2;
  __shared__ __attribute__((aligned(16))) float b1_s[4][PR_B1C * PR_CH];
  const int b = blockIdx.x;
  if (b >= a.B) return;  // block-uniform
  __bf16* xb = xb_s[w];
  float* qkv = reinterpret_cast<float*>(big_s[w]);
  __bf16* hb = reinterpret_cast<__bf16*>(big_s[w]);
  static_assert(T * QKV_P * 4 <= T * HB_P * 2, "q|k|v fits in the hidden-chunk buffer");

  // posterior-head operands of this lane (output j = lane): fetched now, used after the last layer
  const int A_ = a.A, jh = lane < 2 * A_ ? lane : 0;
  f32x4 hw[PR_D / 4];
  float hbias = 0.f, heps = 0.f;
  if (a.Wc) {
#pragma unroll
    for (int d = 0; d < PR_D / 4; d++) hw[d] = *reinterpret_cast<const f32x4*>(a.Wc + jh * PR_D + 4 * d);
    hbias = a.bc[jh];
    heps = a.eps[(long)b * A_ + (lane < A_ ? lane : 0)];
  }
  // x = emb + position embedding, in D layout: x[rt][nt][r] = feature 16 nt + 4 g + r of row (time step) 16 rt + i
  f32x4 x[RT][2];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      const int n = 16 * nt + 4 * g, t = 16 * rt + i;
      x[rt][nt] = *reinterpret_cast<const f32x4*>(a.emb + ((long)b * T + t) * a.ld_emb + n) +
                  *reinterpret_cast<const f32x4*>(a.P + a.pos + t * PR_D + n);
    }
  auto put_xb = [&](const f32x4 (&v)[RT][2]) {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
        *reinterpret_cast<bf16x4*>(xb + (16 * rt + i) * XB_P + 16 * nt + 4 * g) =
            bf16x4{(__bf16)v[rt][nt][0], (__bf16)v[rt][nt][1], (__bf16)v[rt][nt][2], (__bf16)v[rt][nt][3]};
  };
  const int nchunk = a.FF / PR_CH;
  const bool sv0 = a.save && w == 0;  // (the attention half runs redundantly on all four waves: wave 0 writes the saves)
  long rrow[RT];                      // this lane's batch-major rows
#pragma unroll
  for (int rt = 0; rt < RT; rt++) rrow[rt] = (long)b * T + 16 * rt + i;
  for (int l = 0; l < a.L; l++) {
    const PrLayerOff& o = a.l[l];
    const PrSaveL& S = a.sv[l];
    // Every small operand of the layer in ONE global round trip at its top (round 5): projection weights (from the bf16
    // mirror: the very values cvt8 of the fp32 block gives), biases, both LayerNorms' vectors and the first FFN chunk's
    // W1 fragments.  Each used to be fetched where it is consumed, behind an LDS wait that also fences memory: five
    // dependent round trips per layer on a kernel that is nothing but latency (~1.5 us each with 1 024 waves asking for
    // the same lines) - and this launch is on the critical path of BOTH chains of the training step.
    bf16x8 inw[6], outw[2];
    f32x4 inb[6], outb[2], n1w[2], n1b[2], n2w[2], n2b[2], b2v[2];
#pragma unroll
    for (int nt = 0; nt < 6; nt++) {
      inw[nt] = *reinterpret_cast<const bf16x8*>(a.Pb + o.in_w + (long)(16 * nt + i) * PR_D + 8 * g);
      inb[nt] = *reinterpret_cast<const f32x4*>(a.P + o.in_b + 16 * nt + 4 * g);
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      outw[nt] = *reinterpret_cast<const bf16x8*>(a.Pb + o.out_w + (long)(16 * nt + i) * PR_D + 8 * g);
      outb[nt] = *reinterpret_cast<const f32x4*>(a.P + o.out_b + 16 * nt + 4 * g);
      n1w[nt] = *reinterpret_cast<const f32x4*>(a.P + o.n1w + 16 * nt + 4 * g);
      n1b[nt] = *reinterpret_cast<const f32x4*>(a.P + o.n1b + 16 * nt + 4 * g);
      n2w[nt] = *reinterpret_cast<const f32x4*>(a.P + o.n2w + 16 * nt + 4 * g);
      n2b[nt] = *reinterpret_cast<const f32x4*>(a.P + o.n2b + 16 * nt + 4 * g);
      b2v[nt] = *reinterpret_cast<const f32x4*>(a.P + o.b2 + 16 * nt + 4 * g);
    }
    const __bf16* W1 = a.Pb + o.w1;
    const __bf16* W2 = a.Pb + o.w2;
    bf16x8 w1f[16], w2f[8][2];
    const int c0 = w < nchunk ? w : nchunk - 1;  // (a wave without chunks prefetches in bounds and skips the loop)
#pragma unroll
    for (int nt = 0; nt < 16; nt++) w1f[nt] = *reinterpret_cast<const bf16x8*>(W1 + (long)(PR_CH * c0 + 16 * nt + i) * PR_D + 8 * g);
    f32x4 b1pre[PR_B1C];  // lane: floats 4 lane .. 4 lane + 3 of chunk w + 4 j
    const bool b1_staged = nchunk <= 4 * PR_B1C;
    if (b1_staged) {
#pragma unroll
      for (int j = 0; j < PR_B1C; j++) {
        const int c = w + 4 * j;
        b1pre[j] = *reinterpret_cast<const f32x4*>(a.P + o.b1 + PR_CH * (c < nchunk ? c : c0) + 4 * lane);
      }
    }
    if (RT == 1 && PR_W2TOP) {  // (window 16: 64 more registers fit - the first chunk's W2 fragments travel with the rest)
#pragma unroll
      for (int ks = 0; ks < 8; ks++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
          w2f[ks][nt] = *reinterpret_cast<const bf16x8*>(W2 + (long)(16 * nt + i) * a.FF + PR_CH * c0 + 32 * ks + 8 * g);
    }
    if (sv0) {
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) *reinterpret_cast<f32x4*>(S.xin + rrow[rt] * PR_D + 16 * nt + 4 * g) = x[rt][nt];
    }
    PR_STAMP(0);
    // ---- q|k|v = x Win^T + b  (6 N tiles, K = 32)
    put_xb(x);
    lds_sync();
    {
      bf16x8 xf[RT];
#pragma unroll
      for (int rt = 0; rt < RT; rt++) xf[rt] = *reinterpret_cast<const bf16x8*>(xb + (16 * rt + i) * XB_P + 8 * g);
#pragma unroll
      for (int nt = 0; nt < 6; nt++) {
        const bf16x8 wf = inw[nt];
        const f32x4 bias = inb[nt];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
          f32x4 acc = bias;
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[rt], acc, 0, 0, 0);
          *reinterpret_cast<f32x4*>(qkv + (16 * rt + i) * QKV_P + 16 * nt + 4 * g) = acc;
          if (sv0) *reinterpret_cast<f32x4*>(S.qkv + rrow[rt] * 3 * PR_D + 16 * nt + 4 * g) = acc;
        }
      }
    }
    lds_sync();
    PR_STAMP(1);
    // ---- attention: 8 heads x T queries = 8 T (head, query) pairs, 2 RT per lane; result -> xb (bf16)
#pragma unroll
    for (int j = 0; j < 2 * RT; j++) {
      const int p = lane + 64 * j, h = p / T, qi = p % T;
      f32x4 q = *reinterpret_cast<const f32x4*>(qkv + qi * QKV_P + PR_HD * h);
      // 1 / sqrt(head_dim) and log2(e) in one factor: softmax(s) = 2^(s' - max s') / sum, s' = s log2 e (v_exp_f32 is 2^x;
      // expf was ~15 instructions and the 16 IEEE divisions by the sum ~11 each, per (head, query) pair, on all four waves)
#pragma unroll
      for (int e = 0; e < 4; e++) q[e] *= 0.5f * 1.44269504088896341f;
      float s[T], mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < T; t++) {
        const f32x4 k = *reinterpret_cast<const f32x4*>(qkv + t * QKV_P + PR_D + PR_HD * h);
        s[t] = ((q[0] * k[0] + q[1] * k[1]) + q[2] * k[2]) + q[3] * k[3];
        mx = fmaxf(mx, s[t]);
      }
      float se = 0.f;
#pragma unroll
      for (int t = 0; t < T; t++) { s[t] = __builtin_amdgcn_exp2f(s[t] - mx); se += s[t]; }
      const float rse = __builtin_amdgcn_rcpf(se);
      f32x4 ov = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < T; t++) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(qkv + t * QKV_P + 2 * PR_D + PR_HD * h);
        const float pr = s[t] * rse;
#pragma unroll
        for (int e = 0; e < 4; e++) ov[e] += pr * vv[e];
      }
      *reinterpret_cast<bf16x4*>(xb + qi * XB_P + PR_HD * h) = bf16x4{(__bf16)ov[0], (__bf16)ov[1], (__bf16)ov[2], (__bf16)ov[3]};
      if (sv0) *reinterpret_cast<f32x4*>(S.att + ((long)b * T + qi) * PR_D + PR_HD * h) = ov;
    }
    lds_sync();
    PR_STAMP(2);
    // ---- out-projection + residual + LayerNorm 1
    {
      bf16x8 af[RT];
#pragma unroll
      for (int rt = 0; rt < RT; rt++) af[rt] = *reinterpret_cast<const bf16x8*>(xb + (16 * rt + i) * XB_P + 8 * g);
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        const bf16x8 wf = outw[nt];
        const f32x4 bias = outb[nt];
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
          f32x4 acc = bias;
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af[rt], acc, 0, 0, 0);
          if (sv0) *reinterpret_cast<f32x4*>(S.proj + rrow[rt] * PR_D + 16 * nt + 4 * g) = acc;
          x[rt][nt] += acc;
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        layer_norm32_r(x[rt], n1w, n1b, g, sv0 ? S.st1 + 2 * rrow[rt] : nullptr);
        if (sv0) {
#pragma unroll
          for (int nt = 0; nt < 2; nt++) *reinterpret_cast<f32x4*>(S.x1 + rrow[rt] * PR_D + 16 * nt + 4 * g) = x[rt][nt];
        }
      }
    }
    lds_sync();  // every lane has read its out-projection operand before xb is overwritten
    put_xb(x);
    lds_sync();
    PR_STAMP(3);
    // ---- FFN: relu(x W1^T + b1) W2^T + b2, hidden processed in chunks of 256 kept in LDS as bf16
    if (b1_staged) {
#pragma unroll
      for (int j = 0; j < PR_B1C; j++) *reinterpret_cast<f32x4*>(b1_s[w] + PR_CH * j + 4 * lane) = b1pre[j];
      lds_sync();  // (wave-private: this wave wrote it, this wave reads it)
    }
    bf16x8 xf[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++) xf[rt] = *reinterpret_cast<const bf16x8*>(xb + (16 * rt + i) * XB_P + 8 * g);
    f32x4 y[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int nt = 0; nt < 2; nt++)
        y[rt][nt] = w == 0 ? b2v[nt] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = w; c < nchunk; c += 4) {
      // this chunk's W2 fragments travel while FFN1 runs (window 16: the first chunk's came with the layer's other operands)
      if (RT != 1 || !PR_W2TOP || c != w) {
#pragma unroll
        for (int ks = 0; ks < 8; ks++)
#pragma unroll
          for (int nt = 0; nt < 2; nt++)
            w2f[ks][nt] = *reinterpret_cast<const bf16x8*>(W2 + (long)(16 * nt + i) * a.FF + PR_CH * c + 32 * ks + 8 * g);
      }
#pragma unroll
      for (int nt = 0; nt < 16; nt++) {
        const f32x4 bias = b1_staged ? *reinterpret_cast<const f32x4*>(b1_s[w] + PR_CH * ((c - w) >> 2) + 16 * nt + 4 * g)
                                     : *reinterpret_cast<const f32x4*>(a.P + o.b1 + PR_CH * c + 16 * nt + 4 * g);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
          f32x4 hacc = bias;
          hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[nt], xf[rt], hacc, 0, 0, 0);
          const f32x4 hr = {fmaxf(hacc[0], 0.f), fmaxf(hacc[1], 0.f), fmaxf(hacc[2], 0.f), fmaxf(hacc[3], 0.f)};
          *reinterpret_cast<bf16x4*>(hb + (16 * rt + i) * HB_P + 16 * nt + 4 * g) = bf16x4{(__bf16)hr[0], (__bf16)hr[1], (__bf16)hr[2], (__bf16)hr[3]};
          if (a.save) *reinterpret_cast<f32x4*>(S.ff1 + rrow[rt] * a.FF + PR_CH * c + 16 * nt + 4 * g) = hr;  // (each wave its own chunks)
        }
      }
      lds_sync();
      // the next chunk's W1 fragments travel while FFN2 runs
      if (c + 4 < nchunk) {
#pragma unroll
        for (int nt = 0; nt < 16; nt++)
          w1f[nt] = *reinterpret_cast<const bf16x8*>(W1 + (long)(PR_CH * (c + 4) + 16 * nt + i) * PR_D + 8 * g);
      }
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
          const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hb + (16 * rt + i) * HB_P + 32 * ks + 8 * g);
#pragma unroll
          for (int nt = 0; nt < 2; nt++) y[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks][nt], hf, y[rt][nt], 0, 0, 0);
        }
      }
      lds_sync();  // hidden chunk consumed before the next one overwrites it
    }
    PR_STAMP(4);
    // the four partial FFN outputs meet in LDS (fixed summation order: every wave ends with the same bits).
    // One barrier per layer: the buffer alternates with the layer, and a wave can only reach layer l + 2's write
    // after every wave has passed layer l + 1's barrier, i.e. finished reading layer l's partials.
    {
      float* yp = ypart[l & 1][w];
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) *reinterpret_cast<f32x4*>(yp + (16 * rt + i) * PR_D + 16 * nt + 4 * g) = y[rt][nt];
      __syncthreads();
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
          f32x4 t = *reinterpret_cast<const f32x4*>(ypart[l & 1][0] + (16 * rt + i) * PR_D + 16 * nt + 4 * g);
#pragma unroll
          for (int ww = 1; ww < 4; ww++) t += *reinterpret_cast<const f32x4*>(ypart[l & 1][ww] + (16 * rt + i) * PR_D + 16 * nt + 4 * g);
          if (sv0) *reinterpret_cast<f32x4*>(S.ff2 + rrow[rt] * PR_D + 16 * nt + 4 * g) = t;
          x[rt][nt] += t;
        }
    }
#pragma unroll
    for (int rt = 0; rt < RT; rt++) layer_norm32_r(x[rt], n2w, n2b, g, sv0 ? S.st2 + 2 * rrow[rt] : nullptr);
    PR_STAMP(5);
  }
  if (w != 0) return;  // every wave holds the same result: wave 0 writes it
  // ---- mean over the T time steps (row tiles in order, then lanes i = 0..15 of each g)
#pragma unroll
  for (int nt = 0; nt < 2; nt++) {
    f32x4 s = x[0][nt];
#pragma unroll
    for (int rt = 1; rt < RT; rt++) s += x[rt][nt];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1)
#pragma unroll
      for (int r = 0; r < 4; r++) s[r] += __shfl_xor(s[r], off, 64);
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) s[r] *= (1.0f / T);
      *reinterpret_cast<f32x4*>(a.pooled + (long)b * PR_D + 16 * nt + 4 * g) = s;
      if (a.Wc) *reinterpret_cast<f32x4*>(qkv + 16 * nt + 4 * g) = s;
    }
  }
  if (!a.Wc) return;
  // ---- posterior head on the pooled vector: head = Wc pooled + bc (lane j = output j), then
  // std = softplus(var_raw) + min_std, plan = tanh(mean + eps * std)  (plan_recognition_transformer.py:89-104)
  lds_sync();
  const int A = A_;
  float h = hbias;
#pragma unroll
  for (int d = 0; d < PR_D; d += 4) {
    const f32x4 wv = hw[d / 4], pv = *reinterpret_cast<const f32x4*>(qkv + d);
    h += wv[0] * pv[0] + wv[1] * pv[1] + wv[2] * pv[2] + wv[3] * pv[3];
  }
  if (lane < 2 * A) a.head[(long)b * 2 * A + lane] = h;
  const float vr = __shfl(h, (lane + A) & 63, 64);
  if (lane < A) {
    const float sd = (vr > 20.f ? vr : log1pf(expf(vr))) + a.min_std;
    a.plan[(long)b * A + lane] = tanhf(h + heps * sd);
  }
}

// ------------------------------------------------------------------------------------------------ d_model 64 (two cameras)
// The same launch for d_model = 64 (the real-world configuration: two cameras x 32 features, 8 heads of 8, window 32 -
// BASELINE configs[3]), inference only (frozen LMP inside TACORL; round 4: the per-op path put 349 us of launches in front of
// BOTH chains of the C4 step).  Same structure - one workgroup per sequence, activations in registers in the MFMA D layout
// (four 16-feature tiles per row), the attention half redundantly on the four waves, the FFN split by hidden chunks - with
// K = 64 contractions as two k-steps, 128-wide hidden chunks (16 + 16 weight fragments in registers) and ONE partial-sum
// buffer (two barriers per layer: 32 KB; two buffers would not fit beside 4 x 30 KB of wave scratch).
constexpr int D6 = 64, HD6 = 8, CH6 = 128;
constexpr int XB6_P = 72;    // bf16 row pitch of the 64-wide B operand (144 B)
constexpr int HB6_P = 136;   // bf16 row pitch of a 128-wide hidden chunk (272 B)
constexpr int QKV6_P = 196;  // fp32 row pitch of q|k|v (192 + 4)

__device__ __forceinline__ void layer_norm64(f32x4 (&v)[4], const float* w, const float* b, int g) {
  float s = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) s += v[nt][r];
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  const float mean = s / 64.f;
  float q = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) { const float c = v[nt][r] - mean; q += c * c; }
  q += __shfl_xor(q, 16, 64);
  q += __shfl_xor(q, 32, 64);
  const float rstd = 1.0f / sqrtf(q / 64.f + 1e-5f);
#pragma unroll
  for (int nt = 0; nt < 4; nt++) {
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + 16 * nt + 4 * g), bb = *reinterpret_cast<const f32x4*>(b + 16 * nt + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; r++) v[nt][r] = (v[nt][r] - mean) * rstd * ww[r] + bb[r];
  }
}

__device__ __forceinline__ void layer_norm64(f32x4 (&v)[4], const f32x4 (&ww)[4], const f32x4 (&bb)[4]) {  // same sums, operands in registers
  float s = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) s += v[nt][r];
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  const float mean = s / 64.f;
  float q = 0.f;
#pragma unroll
  for (int nt = 0; nt < 4; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) { const float c = v[nt][r] - mean; q += c * c; }
  q += __shfl_xor(q, 16, 64);
  q += __shfl_xor(q, 32, 64);
  const float rstd = 1.0f / sqrtf(q / 64.f + 1e-5f);
#pragma unroll
  for (int nt = 0; nt < 4; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) v[nt][r] = (v[nt][r] - mean) * rstd * ww[nt][r] + bb[nt][r];
}

template <int RT>
__global__ __launch_bounds__(256) void pr_encoder_fused64_kernel(PrArgs a) {
#ifdef PR_STAMPS
  unsigned long long t_prev = clock64();
#endif
  constexpr int T = 16 * RT;
  constexpr int BIG = (T * QKV6_P * 4 > T * HB6_P * 2) ? T * QKV6_P * 4 : T * HB6_P * 2;
  __shared__ __attribute__((aligned(16))) __bf16 xb_s[4][T * XB6_P];
  __shared__ __attribute__((aligned(16))) unsigned char big_s[4][BIG];  // q|k|v (fp32) or FFN hidden chunk (bf16)
  __shared__ __attribute__((aligned(16))) float ypart[4][T * D6];
  // The attention is SPLIT over the four waves (round 5; redundant before, as in the d_model-32 kernel): wave w owns heads 2 w and
  // 2 w + 1 - the q / k / v column tiles w, 4 + w, 8 + w of the in-projection, 2 T (head, query) pairs - and writes its 16 columns
  // of the result into this shared operand of the out-projection.  At T = 32 every lane read 2 x 32 B of k and of v per key as
  // broadcast ds_read_b128 (1 KB of LDS return path for 32 distinct bytes), four waves at once: 35 % of the launch (phase clocks).
  __shared__ __attribute__((aligned(16))) __bf16 xatt[T * XB6_P];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  if (b >= a.B) return;  // block-uniform
  __bf16* xb = xb_s[w];
  float* qkv = reinterpret_cast<float*>(big_s[w]);
  __bf16* hb = reinterpret_cast<__bf16*>(big_s[w]);
  // x[rt][nt][r] = feature 16 nt + 4 g + r of row (time step) 16 rt + i
  f32x4 x[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
      const int n = 16 * nt + 4 * g, t = 16 * rt + i;
      x[rt][nt] = *reinterpret_cast<const f32x4*>(a.emb + ((long)b * T + t) * a.ld_emb + n) +
                  *reinterpret_cast<const f32x4*>(a.P + a.pos + t * D6 + n);
    }
  auto put_xb = [&](const f32x4 (&v)[RT][4]) {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int nt = 0; nt < 4; nt++)
        *reinterpret_cast<bf16x4*>(xb + (16 * rt + i) * XB6_P + 16 * nt + 4 * g) =
            bf16x4{(__bf16)v[rt][nt][0], (__bf16)v[rt][nt][1], (__bf16)v[rt][nt][2], (__bf16)v[rt][nt][3]};
  };
  auto get_xf = [&](bf16x8 (&xf)[RT][2]) {
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) xf[rt][ks] = *reinterpret_cast<const bf16x8*>(xb + (16 * rt + i) * XB6_P + 32 * ks + 8 * g);
  };
  const int nchunk = a.FF / CH6;
  for (int l = 0; l < a.L; l++) {
    const PrLayerOff& o = a.l[l];
    PR_STAMP(0);
    // Operands are requested a phase ahead, in front of the LDS waits (which also fence memory for the compiler): fetched where
    // they are consumed, each phase of this latency-bound launch began with an exposed L2 round trip (round 5; the d_model-32
    // kernel above does the same).  Projection weights come from the bf16 mirror - the very values cvt8 of the fp32 block gives.
    bf16x8 inw[3][2];
    f32x4 inb[3];
#pragma unroll
    for (int k3 = 0; k3 < 3; k3++) {
      const int nt = 4 * k3 + w;
      inw[k3][0] = *reinterpret_cast<const bf16x8*>(a.Pb + o.in_w + (long)(16 * nt + i) * D6 + 8 * g);
      inw[k3][1] = *reinterpret_cast<const bf16x8*>(a.Pb + o.in_w + (long)(16 * nt + i) * D6 + 32 + 8 * g);
      inb[k3] = *reinterpret_cast<const f32x4*>(a.P + o.in_b + 16 * nt + 4 * g);
    }
    // ---- q|k|v = x Win^T + b  (this wave's 3 of the 12 N tiles, K = 64)
    put_xb(x);
    lds_sync();
    {
      bf16x8 xf[RT][2];
      get_xf(xf);
#pragma unroll
      for (int k3 = 0; k3 < 3; k3++) {
        const int nt = 4 * k3 + w;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
          f32x4 acc = inb[k3];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(inw[k3][0], xf[rt][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(inw[k3][1], xf[rt][1], acc, 0, 0, 0);
          *reinterpret_cast<f32x4*>(qkv + (16 * rt + i) * QKV6_P + 16 * nt + 4 * g) = acc;
        }
      }
    }
    // (requested here, used behind the attention: out-projection and LayerNorm 1)
    bf16x8 outw[4][2];
    f32x4 outb[4], n1w[4], n1b[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
      outw[nt][0] = *reinterpret_cast<const bf16x8*>(a.Pb + o.out_w + (long)(16 * nt + i) * D6 + 8 * g);
      outw[nt][1] = *reinterpret_cast<const bf16x8*>(a.Pb + o.out_w + (long)(16 * nt + i) * D6 + 32 + 8 * g);
      outb[nt] = *reinterpret_cast<const f32x4*>(a.P + o.out_b + 16 * nt + 4 * g);
      n1w[nt] = *reinterpret_cast<const f32x4*>(a.P + o.n1w + 16 * nt + 4 * g);
      n1b[nt] = *reinterpret_cast<const f32x4*>(a.P + o.n1b + 16 * nt + 4 * g);
    }
    lds_sync();
    PR_STAMP(1);
    // ---- attention: this wave's 2 heads x T queries, one (head, query) pair per lane (T = 16: the upper half of the wave repeats
    // the lower one and does not store), head_dim 8; result -> xatt (bf16)
    {
      const int p = lane & (2 * T - 1), h = 2 * w + p / T, qi = p % T;
      f32x4 q0 = *reinterpret_cast<const f32x4*>(qkv + qi * QKV6_P + HD6 * h), q1 = *reinterpret_cast<const f32x4*>(qkv + qi * QKV6_P + HD6 * h + 4);
      const float sc = 0.35355339059327379f * 1.44269504088896341f;  // 1 / sqrt(8) and log2 e: the softmax runs on 2^x (round 5)
      q0 *= sc; q1 *= sc;
      float s[T], mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < T; t++) {
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(qkv + t * QKV6_P + D6 + HD6 * h), k1 = *reinterpret_cast<const f32x4*>(qkv + t * QKV6_P + D6 + HD6 * h + 4);
        s[t] = (((q0[0] * k0[0] + q0[1] * k0[1]) + q0[2] * k0[2]) + q0[3] * k0[3]) + (((q1[0] * k1[0] + q1[1] * k1[1]) + q1[2] * k1[2]) + q1[3] * k1[3]);
        mx = fmaxf(mx, s[t]);
      }
      float se = 0.f;
#pragma unroll
      for (int t = 0; t < T; t++) { s[t] = __builtin_amdgcn_exp2f(s[t] - mx); se += s[t]; }
      const float rse = __builtin_amdgcn_rcpf(se);
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < T; t++) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(qkv + t * QKV6_P + 2 * D6 + HD6 * h), v1 = *reinterpret_cast<const f32x4*>(qkv + t * QKV6_P + 2 * D6 + HD6 * h + 4);
        const float pr = s[t] * rse;
        o0 += pr * v0; o1 += pr * v1;
      }
      if (lane < 2 * T)
        *reinterpret_cast<bf16x8*>(xatt + qi * XB6_P + HD6 * h) =
            bf16x8{(__bf16)o0[0], (__bf16)o0[1], (__bf16)o0[2], (__bf16)o0[3], (__bf16)o1[0], (__bf16)o1[1], (__bf16)o1[2], (__bf16)o1[3]};
    }
    // (requested here, used behind the out-projection: this wave's first hidden chunk of W1)
    const __bf16* W1 = a.Pb + o.w1;
    const __bf16* W2 = a.Pb + o.w2;
    bf16x8 w1f[8][2], w2f[4][4];
    f32x4 b1f[8];  // (the chunk's FFN1 biases travel with its W1 fragments: fetched inside the chunk they were one more exposed round trip)
    auto fetch_w1 = [&](int c) {
#pragma unroll
      for (int nt = 0; nt < 8; nt++) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++) w1f[nt][ks] = *reinterpret_cast<const bf16x8*>(W1 + (long)(CH6 * c + 16 * nt + i) * D6 + 32 * ks + 8 * g);
        b1f[nt] = *reinterpret_cast<const f32x4*>(a.P + o.b1 + CH6 * c + 16 * nt + 4 * g);
      }
    };
    auto fetch_w2 = [&](int c) {
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
          w2f[ks][nt] = *reinterpret_cast<const bf16x8*>(W2 + (long)(16 * nt + i) * a.FF + CH6 * c + 32 * ks + 8 * g);
    };
    const int c0 = w < nchunk ? w : nchunk - 1;  // (a wave without chunks prefetches in bounds and skips the loop)
    fetch_w1(c0);
    __syncthreads();  // the four waves' columns of the attention result (the previous layer's readers of xatt passed the two
                      // barriers of its FFN exchange)
    PR_STAMP(2);
    // ---- out-projection + residual + LayerNorm 1
    {
      bf16x8 af[RT][2];
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int ks = 0; ks < 2; ks++) af[rt][ks] = *reinterpret_cast<const bf16x8*>(xatt + (16 * rt + i) * XB6_P + 32 * ks + 8 * g);
#pragma unroll
      for (int nt = 0; nt < 4; nt++) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
          f32x4 acc = outb[nt];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(outw[nt][0], af[rt][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(outw[nt][1], af[rt][1], acc, 0, 0, 0);
          x[rt][nt] += acc;
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; rt++) layer_norm64(x[rt], n1w, n1b);
    }
    // (requested here, used behind the two waits below: the first chunk's W2 fragments, the FFN output bias)
    fetch_w2(c0);
    f32x4 b2v[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) b2v[nt] = *reinterpret_cast<const f32x4*>(a.P + o.b2 + 16 * nt + 4 * g);
    put_xb(x);  // (the out-projection read xatt, not xb)
    lds_sync();
    PR_STAMP(3);
    // ---- FFN: relu(x W1^T + b1) W2^T + b2, hidden in chunks of 128 kept in LDS as bf16
    bf16x8 xf[RT][2];
    get_xf(xf);
    f32x4 y[RT][4];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
      for (int nt = 0; nt < 4; nt++)
        y[rt][nt] = w == 0 ? b2v[nt] : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = w; c < nchunk; c += 4) {
#pragma unroll
      for (int nt = 0; nt < 8; nt++) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
          f32x4 hacc = b1f[nt];
          hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[nt][0], xf[rt][0], hacc, 0, 0, 0);
          hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[nt][1], xf[rt][1], hacc, 0, 0, 0);
          *reinterpret_cast<bf16x4*>(hb + (16 * rt + i) * HB6_P + 16 * nt + 4 * g) =
              bf16x4{(__bf16)fmaxf(hacc[0], 0.f), (__bf16)fmaxf(hacc[1], 0.f), (__bf16)fmaxf(hacc[2], 0.f), (__bf16)fmaxf(hacc[3], 0.f)};
        }
      }
      if (c + 4 < nchunk) fetch_w1(c + 4);  // (its registers are free: in flight under the second half of this chunk)
      lds_sync();
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
          const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hb + (16 * rt + i) * HB6_P + 32 * ks + 8 * g);
#pragma unroll
          for (int nt = 0; nt < 4; nt++) y[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks][nt], hf, y[rt][nt], 0, 0, 0);
        }
      }
      if (c + 4 < nchunk) fetch_w2(c + 4);  // (in flight under the first half of the next chunk)
      lds_sync();  // hidden chunk consumed before the next one overwrites it
    }
    PR_STAMP(4);
    // (requested in front of the workgroup barriers below: LayerNorm 2)
    f32x4 n2w[4], n2b[4];
#pragma unroll
    for (int nt = 0; nt < 4; nt++) {
      n2w[nt] = *reinterpret_cast<const f32x4*>(a.P + o.n2w + 16 * nt + 4 * g);
      n2b[nt] = *reinterpret_cast<const f32x4*>(a.P + o.n2b + 16 * nt + 4 * g);
    }
    // the four partial FFN outputs meet in LDS (fixed summation order: every wave ends with the same bits)
    {
      float* yp = ypart[w];
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) *reinterpret_cast<f32x4*>(yp + (16 * rt + i) * D6 + 16 * nt + 4 * g) = y[rt][nt];
      __syncthreads();
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int nt = 0; nt < 4; nt++) {
          f32x4 t = *reinterpret_cast<const f32x4*>(ypart[0] + (16 * rt + i) * D6 + 16 * nt + 4 * g);
#pragma unroll
          for (int ww = 1; ww < 4; ww++) t += *reinterpret_cast<const f32x4*>(ypart[ww] + (16 * rt + i) * D6 + 16 * nt + 4 * g);
          x[rt][nt] += t;
        }
      __syncthreads();  // (one buffer: everyone has read the partials before the next layer's are written)
    }
#pragma unroll
    for (int rt = 0; rt < RT; rt++) layer_norm64(x[rt], n2w, n2b);
    PR_STAMP(5);
  }
  if (w != 0) return;  // every wave holds the same result: wave 0 writes it
  // ---- mean over the T time steps
#pragma unroll
  for (int nt = 0; nt < 4; nt++) {
    f32x4 s = x[0][nt];
#pragma unroll
    for (int rt = 1; rt < RT; rt++) s += x[rt][nt];
#pragma unroll
    for (int off = 1; off < 16; off <<= 1)
#pragma unroll
      for (int r = 0; r < 4; r++) s[r] += __shfl_xor(s[r], off, 64);
    if (i == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++) s[r] *= (1.0f / T);
      *reinterpret_cast<f32x4*>(a.pooled + (long)b * D6 + 16 * nt + 4 * g) = s;
      if (a.Wc) *reinterpret_cast<f32x4*>(qkv + 16 * nt + 4 * g) = s;
    }
  }
  if (!a.Wc) return;
  // ---- posterior head on the pooled vector: head = Wc pooled + bc (lane j = output j), std = softplus(var_raw) + min_std,
  // plan = tanh(mean + eps * std)  (plan_recognition_transformer.py:89-104)
  lds_sync();
  const int A = a.A, jh = lane < 2 * A ? lane : 0;
  float h = a.bc[jh];
#pragma unroll
  for (int d = 0; d < D6; d += 4) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(a.Wc + jh * D6 + d), pv = *reinterpret_cast<const f32x4*>(qkv + d);
    h += wv[0] * pv[0] + wv[1] * pv[1] + wv[2] * pv[2] + wv[3] * pv[3];
  }
  if (lane < 2 * A) a.head[(long)b * 2 * A + lane] = h;
  const float vr = __shfl(h, (lane + A) & 63, 64);
  if (lane < A) {
    const float sd = (vr > 20.f ? vr : log1pf(expf(vr))) + a.min_std;
    a.plan[(long)b * A + lane] = tanhf(h + a.eps[(long)b * A + lane] * sd);
  }
}

// ------------------------------------------------------------------------------------------------ backward (train mode)
// The input-gradient chain of the same encoder in ONE launch (round 4): from the gradient of the time-pooled output back to
// the gradient of the position-embedded input, through both layers - LayerNorm 2, FFN (the 2048-wide hidden gradient in
// 256-column chunks split over the four waves, masked by the saved post-ReLU hidden state), LayerNorm 1, out-projection,
// the 8-head attention, in-projection - reading what tacorl_pr_encoder_fused_train saved.  As per-op launches this chain
// was ~9 dependent launches per layer (LayerNorm backward + column sums, two FFN input-gradient GEMMs - one split-K with
// its reduce -, out-projection, attention, in-projection) on PlayLMP.training_step's critical path.  The weight gradients
// stay per-op GEMMs on the caller's side stream: this launch writes their dZ operands (LayerNorm-2 / LayerNorm-1 input
// gradients, the masked hidden gradient, d(q|k|v)) and per-sequence partial sums of the LayerNorm weight / bias gradients
// (pr_ln_reduce_kernel adds them up in batch order).  Same structure as the forward: one workgroup per sequence,
// activations in registers in the MFMA D layout, the cheap parts redundantly on all four waves, one barrier per layer.
// The FFN needs W1^T [32][FF] and W2^T [FF][32] as bf16 (tacorl_transpose_to_bf16, weights only); the 32 x 32 / 96 x 32
// projection matrices are gathered transposed from the fp32 block.
constexpr int DQ_P = 104;  // bf16 row pitch of d(q|k|v) as the in-projection's MFMA B operand (208 B)
struct PrBwdL {
  const float *xin, *qkv, *att, *proj, *x1, *ff1, *ff2, *st1, *st2;  // saved by the train-mode forward
  float *dvb, *d_ff1, *dv1b, *d_qkv;                                  // dZ operands of the weight-gradient GEMMs
  const __bf16 *w1t, *w2t;                                            // W1^T [32][FF], W2^T [FF][32]
  const __bf16 *wot, *wint;                                           // Wo^T [32][32], Win^T [32][96] (or null: gathered from the fp32 block)
};
struct PrBwdArgs {
  const float* P;
  const float* d_pool;  // [B][32]; or null: d_pool = d_head Wc computed here
  const float* d_head;  // [B][A2]
  const float* Wc;      // [A2][32] composed head weight (tacorl_pr_head_compose)
  int A2;
  float* dx;            // [B*T][32]
  float* lnpart;        // [L][2][B][64]: per-sequence (dw | db) partials of LayerNorm 1 / 2
  PrLayerOff l[PR_MAXL];
  PrBwdL s[PR_MAXL];
  int B, FF, L;
};

// LayerNorm(xa + xr) backward for this lane's 8 features of row i; dy in / dv out in the D layout; part: [64] (dw | db)
// partial of this sequence (written by the lanes i == 0), or null
__device__ __forceinline__ void ln_bwd32(const f32x4 (&dy)[2], const float* xa, const float* xr, const float* st, const float* w,
                                         long rrow, int i, int g, f32x4 (&dv)[2], float* part) {
  const float mean = st[2 * rrow], rstd = st[2 * rrow + 1];
  f32x4 xh[2], gv[2];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; nt++) {
    const long o = rrow * PR_D + 16 * nt + 4 * g;
    const f32x4 u = *reinterpret_cast<const f32x4*>(xa + o) + *reinterpret_cast<const f32x4*>(xr + o);
    const f32x4 ww = *reinterpret_cast<const f32x4*>(w + 16 * nt + 4 * g);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      xh[nt][r] = (u[r] - mean) * rstd;
      gv[nt][r] = dy[nt][r] * ww[r];
      s1 += gv[nt][r];
      s2 += gv[nt][r] * xh[nt][r];
    }
  }
  s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
  s1 *= (1.f / 32.f); s2 *= (1.f / 32.f);
#pragma unroll
  for (int nt = 0; nt < 2; nt++)
#pragma unroll
    for (int r = 0; r < 4; r++) dv[nt][r] = rstd * (gv[nt][r] - s1 - xh[nt][r] * s2);
  if (part) {  // sums over the 16 rows (lanes i) of dy * xhat and dy, per feature
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float gw = dy[nt][r] * xh[nt][r], gb = dy[nt][r];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { gw += __shfl_xor(gw, off, 64); gb += __shfl_xor(gb, off, 64); }
        if (i == 0) { part[16 * nt + 4 * g + r] = gw; part[32 + 16 * nt + 4 * g + r] = gb; }
      }
  }
}

__global__ __launch_bounds__(256) void pr_encoder_bwd_fused_kernel(PrBwdArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 xb_s[4][PR_T * XB_P];
  __shared__ __attribute__((aligned(16))) unsigned char big_s[4][PR_T * HB_P * 2];  // q|k|v (fp32) or hidden-gradient chunk (bf16)
  __shared__ __attribute__((aligned(16))) float datt_s[4][PR_T * 36];
  __shared__ __attribute__((aligned(16))) float pm_s[4][4][2][PR_T * 17];            // per wave: 4 heads x (p, ds)
  __shared__ __attribute__((aligned(16))) __bf16 dqb_s[4][PR_T * DQ_P];
  __shared__ __attribute__((aligned(16))) float ypart[2][4][PR_T * PR_D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  if (b >= a.B) return;
  __bf16* xb = xb_s[w];
  float* qkv = reinterpret_cast<float*>(big_s[w]);
  __bf16* hb = reinterpret_cast<__bf16*>(big_s[w]);
  float* datt = datt_s[w];
  __bf16* dqb = dqb_s[w];
  const bool sv0 = w == 0;
  const long rrow = (long)b * PR_T + i;
  const int nchunk = a.FF / PR_CH;
  auto put_xb = [&](const f32x4 (&v)[2]) {
#pragma unroll
    for (int nt = 0; nt < 2; nt++)
      *reinterpret_cast<bf16x4*>(xb + i * XB_P + 16 * nt + 4 * g) =
          bf16x4{(__bf16)v[nt][0], (__bf16)v[nt][1], (__bf16)v[nt][2], (__bf16)v[nt][3]};
  };
  // gradient of the mean over time: every row takes d_pool / T
  f32x4 dx[2];
  if (a.d_pool) {
#pragma unroll
    for (int nt = 0; nt < 2; nt++) dx[nt] = *reinterpret_cast<const f32x4*>(a.d_pool + (long)b * PR_D + 16 * nt + 4 * g) * (1.0f / PR_T);
  } else {  // d_pool = d_head Wc (fp32; the composed head of the forward)
    dx[0] = dx[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < a.A2; j++) {
      const float dh = a.d_head[(long)b * a.A2 + j];
#pragma unroll
      for (int nt = 0; nt < 2; nt++) dx[nt] += dh * *reinterpret_cast<const f32x4*>(a.Wc + (long)j * PR_D + 16 * nt + 4 * g);
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++) dx[nt] *= (1.0f / PR_T);
  }
  for (int l = a.L - 1; l >= 0; l--) {
    const PrLayerOff& o = a.l[l];
    const PrBwdL& S = a.s[l];
    float* lp = a.lnpart + ((long)(2 * l) * a.B + b) * 64;  // LayerNorm 1 partial of this sequence; LayerNorm 2: + B * 64
    // ---- LayerNorm 2: y = LN(x1 + ff2)
    f32x4 dv[2];
    ln_bwd32(dx, S.x1, S.ff2, S.st2, a.P + o.n2w, rrow, i, g, dv, sv0 ? lp + (long)a.B * 64 : nullptr);
    if (sv0) {
#pragma unroll
      for (int nt = 0; nt < 2; nt++) *reinterpret_cast<f32x4*>(S.dvb + rrow * PR_D + 16 * nt + 4 * g) = dv[nt];
    }
    // ---- FFN: d_hidden = (dv W2) * [hidden > 0] in 256-column chunks (written out for linear1's weight gradient),
    //      d_x1 += d_hidden W1
    put_xb(dv);
    lds_sync();
    const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xb + i * XB_P + 8 * g);
    f32x4 y[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    for (int c = w; c < nchunk; c += 4) {
      bf16x8 w1tf[8][2];
#pragma unroll
      for (int ks = 0; ks < 8; ks++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
          w1tf[ks][nt] = *reinterpret_cast<const bf16x8*>(S.w1t + (long)(16 * nt + i) * a.FF + PR_CH * c + 32 * ks + 8 * g);
#pragma unroll
      for (int nt = 0; nt < 16; nt++) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(S.w2t + (long)(PR_CH * c + 16 * nt + i) * PR_D + 8 * g);
        const long ho = rrow * a.FF + PR_CH * c + 16 * nt + 4 * g;
        const f32x4 hm = *reinterpret_cast<const f32x4*>(S.ff1 + ho);  // post-ReLU hidden state: the mask
        f32x4 hacc = {0.f, 0.f, 0.f, 0.f};
        hacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, hacc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) hacc[r] = hm[r] > 0.f ? hacc[r] : 0.f;
        *reinterpret_cast<f32x4*>(S.d_ff1 + ho) = hacc;
        *reinterpret_cast<bf16x4*>(hb + i * HB_P + 16 * nt + 4 * g) = bf16x4{(__bf16)hacc[0], (__bf16)hacc[1], (__bf16)hacc[2], (__bf16)hacc[3]};
      }
      lds_sync();
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        const bf16x8 hf = *reinterpret_cast<const bf16x8*>(hb + i * HB_P + 32 * ks + 8 * g);
#pragma unroll
        for (int nt = 0; nt < 2; nt++) y[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1tf[ks][nt], hf, y[nt], 0, 0, 0);
      }
      lds_sync();
    }
    f32x4 dx1[2];
    {
      float* yp = ypart[l & 1][w];
#pragma unroll
      for (int nt = 0; nt < 2; nt++) *reinterpret_cast<f32x4*>(yp + i * PR_D + 16 * nt + 4 * g) = y[nt];
      __syncthreads();
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        f32x4 t = *reinterpret_cast<const f32x4*>(ypart[l & 1][0] + i * PR_D + 16 * nt + 4 * g);
#pragma unroll
        for (int ww = 1; ww < 4; ww++) t += *reinterpret_cast<const f32x4*>(ypart[l & 1][ww] + i * PR_D + 16 * nt + 4 * g);
        dx1[nt] = dv[nt] + t;  // (the residual: LayerNorm 2's input is x1 + ff2)
      }
    }
    // ---- LayerNorm 1: x1 = LN(xin + proj)
    f32x4 dv1[2];
    ln_bwd32(dx1, S.xin, S.proj, S.st1, a.P + o.n1w, rrow, i, g, dv1, sv0 ? lp : nullptr);
    if (sv0) {
#pragma unroll
      for (int nt = 0; nt < 2; nt++) *reinterpret_cast<f32x4*>(S.dv1b + rrow * PR_D + 16 * nt + 4 * g) = dv1[nt];
    }
    // ---- out-projection: d_att = dv1 Wo  (Wo^T gathered from the fp32 block: 32 x 32)
    put_xb(dv1);
    lds_sync();
    {
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(xb + i * XB_P + 8 * g);
#pragma unroll
      for (int nt = 0; nt < 2; nt++) {
        bf16x8 wf;
        if (S.wot) {
          wf = *reinterpret_cast<const bf16x8*>(S.wot + (16 * nt + i) * PR_D + 8 * g);
        } else {
#pragma unroll
          for (int q = 0; q < 8; q++) wf[q] = (__bf16)a.P[o.out_w + (long)(8 * g + q) * PR_D + 16 * nt + i];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc, 0, 0, 0);
        *reinterpret_cast<f32x4*>(datt + i * 36 + 16 * nt + 4 * g) = acc;
      }
    }
    // the saved q|k|v of this sequence -> LDS (16 rows x 96 floats: six 16-byte pieces per lane)
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const int e = lane + 64 * k, row = e / 24, c4 = e - row * 24;
      *reinterpret_cast<f32x4*>(qkv + row * QKV_P + 4 * c4) = *reinterpret_cast<const f32x4*>(S.qkv + ((long)b * PR_T + row) * 3 * PR_D + 4 * c4);
    }
    lds_sync();
    // ---- attention backward: four heads per pass, lane = (head, row)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int hl = lane >> 4, h = 4 * j + hl, qi = lane & 15;
      float* pp = pm_s[w][hl][0];
      float* pd = pm_s[w][hl][1];
      f32x4 qs = *reinterpret_cast<const f32x4*>(qkv + qi * QKV_P + PR_HD * h);
      qs *= 0.5f;  // 1 / sqrt(head_dim)
      const f32x4 dO = *reinterpret_cast<const f32x4*>(datt + qi * 36 + PR_HD * h);
      float ws_[PR_T], dp[PR_T], mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < PR_T; t++) {
        const f32x4 kt = *reinterpret_cast<const f32x4*>(qkv + t * QKV_P + PR_D + PR_HD * h);
        const f32x4 vt = *reinterpret_cast<const f32x4*>(qkv + t * QKV_P + 2 * PR_D + PR_HD * h);
        ws_[t] = ((qs[0] * kt[0] + qs[1] * kt[1]) + qs[2] * kt[2]) + qs[3] * kt[3];
        dp[t] = ((dO[0] * vt[0] + dO[1] * vt[1]) + dO[2] * vt[2]) + dO[3] * vt[3];
        mx = fmaxf(mx, ws_[t]);
      }
      float se = 0.f, dd = 0.f;
#pragma unroll
      for (int t = 0; t < PR_T; t++) { ws_[t] = expf(ws_[t] - mx); se += ws_[t]; dd += ws_[t] * dp[t]; }
      dd /= se;
      f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < PR_T; t++) {
        const float pj = ws_[t] / se, ds = pj * (dp[t] - dd);
        const f32x4 kt = *reinterpret_cast<const f32x4*>(qkv + t * QKV_P + PR_D + PR_HD * h);
        dq += ds * kt;
        pp[qi * 17 + t] = pj;
        pd[qi * 17 + t] = ds;
      }
      lds_sync();
      // this lane as key / value row qi: dk = scale sum_r ds[r][qi] q_r, dv = sum_r p[r][qi] dO_r
      f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dvv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < PR_T; r++) {
        f32x4 qr = *reinterpret_cast<const f32x4*>(qkv + r * QKV_P + PR_HD * h);
        const f32x4 dor = *reinterpret_cast<const f32x4*>(datt + r * 36 + PR_HD * h);
        qr *= 0.5f;
        dk += pd[r * 17 + qi] * qr;
        dvv += pp[r * 17 + qi] * dor;
      }
      dq *= 0.5f;
      if (sv0) {
        float* dst = S.d_qkv + ((long)b * PR_T + qi) * 3 * PR_D + PR_HD * h;
        *reinterpret_cast<f32x4*>(dst) = dq;
        *reinterpret_cast<f32x4*>(dst + PR_D) = dk;
        *reinterpret_cast<f32x4*>(dst + 2 * PR_D) = dvv;
      }
      __bf16* db_ = dqb + qi * DQ_P + PR_HD * h;
      *reinterpret_cast<bf16x4*>(db_) = bf16x4{(__bf16)dq[0], (__bf16)dq[1], (__bf16)dq[2], (__bf16)dq[3]};
      *reinterpret_cast<bf16x4*>(db_ + PR_D) = bf16x4{(__bf16)dk[0], (__bf16)dk[1], (__bf16)dk[2], (__bf16)dk[3]};
      *reinterpret_cast<bf16x4*>(db_ + 2 * PR_D) = bf16x4{(__bf16)dvv[0], (__bf16)dvv[1], (__bf16)dvv[2], (__bf16)dvv[3]};
      lds_sync();  // (p / ds of this pass consumed before the next pass overwrites them; d(q|k|v) visible)
    }
    // ---- in-projection: dx = d(q|k|v) Win + dv1  (K = 96; Win^T gathered from the fp32 block)
#pragma unroll
    for (int nt = 0; nt < 2; nt++) {
      f32x4 acc = dv1[nt];
#pragma unroll
      for (int ks = 0; ks < 3; ks++) {
        const bf16x8 df = *reinterpret_cast<const bf16x8*>(dqb + i * DQ_P + 32 * ks + 8 * g);
        bf16x8 wf;
        if (S.wint) {
          wf = *reinterpret_cast<const bf16x8*>(S.wint + (16 * nt + i) * (3 * PR_D) + 32 * ks + 8 * g);
        } else {
#pragma unroll
          for (int q = 0; q < 8; q++) wf[q] = (__bf16)a.P[o.in_w + (long)(32 * ks + 8 * g + q) * PR_D + 16 * nt + i];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, df, acc, 0, 0, 0);
      }
      dx[nt] = acc;
    }
    lds_sync();  // every lane has read its operands before the next layer overwrites xb / the scratch buffers
  }
  if (sv0) {
#pragma unroll
    for (int nt = 0; nt < 2; nt++) *reinterpret_cast<f32x4*>(a.dx + rrow * PR_D + 16 * nt + 4 * g) = dx[nt];
  }
}

// LayerNorm weight / bias gradients: sum the per-sequence partials in batch order.  grid = 2 L blocks of 256 threads:
// thread = (column 0..63, row phase 0..3); out[blockIdx.x] -> (dw, db) pointers of that LayerNorm.
struct PrLnReduceArgs { const float* part; float* dw[2 * PR_MAXL]; float* db[2 * PR_MAXL]; int B; };
__global__ __launch_bounds__(256) void pr_ln_reduce_kernel(PrLnReduceArgs a) {
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const float* in = a.part + (long)blockIdx.x * a.B * 64;
  float s = 0.f;
  for (int r = ph; r < a.B; r += 4) s += in[(long)r * 64 + c];
  red[ph][c] = s;
  __syncthreads();
  if (ph == 0) {
    const float t = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
    if (c < 32) a.dw[blockIdx.x][c] = t; else a.db[blockIdx.x][c - 32] = t;
  }
}

// Wc = W_head W_fc ([2A][D]), bc = W_head b_fc + b_head: the two bias-only Linear layers after the time pooling
// (fc: D -> FC, mean_fc: FC -> 2A, no activation between them) are one affine map.  Weights only, so it runs off
// the dependent chain; fixed reduction order (deterministic).  grid = 2A blocks of 256 threads, D = 32.
__global__ __launch_bounds__(256) void pr_head_compose_kernel(const float* __restrict__ w_fc, const float* __restrict__ b_fc,
                                                              const float* __restrict__ w_head, const float* __restrict__ b_head,
                                                              float* __restrict__ Wc, float* __restrict__ bc, int FC, int D) {
  __shared__ float part[8][64 + 1];
  const int NG = 256 / D;  // feature groups of threads: 8 (D = 32) or 4 (D = 64)
  const int j = blockIdx.x, d = threadIdx.x % D, fg = threadIdx.x / D;
  float acc = 0.f, accb = 0.f;
#pragma unroll 16
  for (int f = fg; f < FC; f += NG) {  // independent iterations: 16 x 3 loads in flight per thread
    const float wh = w_head[(long)j * FC + f];
    acc += wh * w_fc[(long)f * D + d];
    accb += wh * b_fc[f];
  }
  part[fg][d] = acc;
  if (d == 0) part[fg][D] = accb;
  __syncthreads();
  if (threadIdx.x <= D) {
    float t = 0.f;
    for (int q = 0; q < NG; q++) t += part[q][threadIdx.x];
    if (threadIdx.x < D) Wc[j * D + threadIdx.x] = t;
    else bc[j] = t + b_head[j];
  }
}

}  // namespace

extern "C" int tacorl_pr_encoder_fused_supported(int D, int T, int H, int FF, int L) {
  return (D == PR_D || D == D6) && (T == PR_T || T == 2 * PR_T) && H == PR_H && FF >= PR_CH && FF % PR_CH == 0 && L >= 1 &&
                 L <= PR_MAXL
             ? 1 : 0;
}
/* the train-mode launch (activations saved for the backward) exists for d_model 32 */
extern "C" int tacorl_pr_encoder_fused_train_supported(int D, int T, int H, int FF, int L) {
  return D == PR_D && tacorl_pr_encoder_fused_supported(D, T, H, FF, L) ? 1 : 0;
}
// offsets: [position_embeddings, then per layer: in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias,
// linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.weight, norm1.bias, norm2.weight, norm2.bias]
// (element offsets into params / params_bf16, each a multiple of 4).
static int pr_encoder_fused_launch(const float* emb, int ld_emb, const float* params, const void* params_bf16,
                                   const long* offsets, float* pooled, int B, int D, int T, int H, int FF, int L,
                                   const float* Wc, const float* bc, const float* eps, float* head, float* plan, int A,
                                   float min_std, tacorl_stream_t stream, float* const* save = nullptr) {
  if (!tacorl_pr_encoder_fused_supported(D, T, H, FF, L) || ld_emb % 4 || B < 1) return TACORL_EINVAL;
  if (((uintptr_t)emb | (uintptr_t)params | (uintptr_t)params_bf16 | (uintptr_t)pooled) & 15) return TACORL_EINVAL;
  PrArgs a{};
  a.emb = emb; a.P = params; a.Pb = (const __bf16*)params_bf16; a.pooled = pooled; a.ld_emb = ld_emb; a.B = B; a.FF = FF; a.L = L;
  a.pos = offsets[0];
  for (int l = 0; l < L; l++) {
    const long* q = offsets + 1 + 12 * l;
    for (int k = 0; k < 12; k++)
      if (q[k] % 4) return TACORL_EINVAL;
    a.l[l] = PrLayerOff{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11]};
  }
  if (Wc) {
    if (!bc || !eps || !head || !plan || A < 1 || 2 * A > 64 || ((uintptr_t)Wc & 15)) return TACORL_EINVAL;
    a.Wc = Wc; a.bc = bc; a.eps = eps; a.head = head; a.plan = plan; a.A = A; a.min_std = min_std;
  }
  if (save) {
    for (int l = 0; l < L; l++) {
      float* const* q = save + 9 * l;
      for (int k = 0; k < 9; k++)
        if (!q[k] || ((uintptr_t)q[k] & 15)) return TACORL_EINVAL;
      a.sv[l] = PrSaveL{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]};
    }
    a.save = 1;
  }
  if (D == D6) {
    if (save) return TACORL_EINVAL;  // (inference only)
    if (T == PR_T) hipLaunchKernelGGL(pr_encoder_fused64_kernel<1>, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(pr_encoder_fused64_kernel<2>, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  } else if (T == PR_T) hipLaunchKernelGGL(pr_encoder_fused_kernel<1>, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(pr_encoder_fused_kernel<2>, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
/* The same launch in train mode (no dropout): also writes, per layer, what the per-op backward reads.  save[9 l + k], k = 0..8:
 * layer input [B T][32], q|k|v [B T][96], attention output [B T][32], out-projection [B T][32], LayerNorm-1 output [B T][32],
 * post-ReLU FFN hidden [B T][FF], FFN output [B T][32], LayerNorm-1 {mean, rstd} [B T][2], LayerNorm-2 {mean, rstd} [B T][2]
 * (fp32, 16-byte aligned) - the tensors tacorl_linear_add_fwd / attention_fwd / add_layernorm_fwd leave behind. */
extern "C" int tacorl_pr_encoder_fused_train(const float* emb, int ld_emb, const float* params, const void* params_bf16,
                                             const long* offsets, float* pooled, int B, int D, int T, int H, int FF, int L,
                                             float* const* save, tacorl_stream_t stream) {
  if (!save) return TACORL_EINVAL;
  return pr_encoder_fused_launch(emb, ld_emb, params, params_bf16, offsets, pooled, B, D, T, H, FF, L, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, 0, 0.f, stream, save);
}
/* Train mode with the posterior head and the plan sample in the launch (as tacorl_pr_encoder_fused_sample: head = Wc pooled +
 * bc with the composed (Wc, bc) of tacorl_pr_head_compose).  The backward takes d_pool = d_head Wc the same way
 * (tacorl_pr_encoder_bwd_fused, d_head / Wc arguments); fc's and mean_fc's weight gradients need fc_out = pooled W_fc^T + b_fc
 * and d_fc = d_head W_head as per-op GEMMs - off the dependent chain. */
extern "C" int tacorl_pr_encoder_fused_train_sample(const float* emb, int ld_emb, const float* params, const void* params_bf16,
                                                    const long* offsets, float* pooled, int B, int D, int T, int H, int FF,
                                                    int L, float* const* save, const float* Wc, const float* bc,
                                                    const float* eps, float* head, float* plan, int A, float min_std,
                                                    tacorl_stream_t stream) {
  if (!save || !Wc) return TACORL_EINVAL;
  return pr_encoder_fused_launch(emb, ld_emb, params, params_bf16, offsets, pooled, B, D, T, H, FF, L, Wc, bc, eps, head, plan,
                                 A, min_std, stream, save);
}
extern "C" int tacorl_pr_encoder_fused(const float* emb, int ld_emb, const float* params, const void* params_bf16,
                                       const long* offsets, float* pooled, int B, int D, int T, int H, int FF, int L,
                                       tacorl_stream_t stream) {
  return pr_encoder_fused_launch(emb, ld_emb, params, params_bf16, offsets, pooled, B, D, T, H, FF, L, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, 0, 0.f, stream);
}
extern "C" int tacorl_pr_encoder_fused_sample(const float* emb, int ld_emb, const float* params, const void* params_bf16,
                                              const long* offsets, float* pooled, int B, int D, int T, int H, int FF,
                                              int L, const float* Wc, const float* bc, const float* eps, float* head,
                                              float* plan, int A, float min_std, tacorl_stream_t stream) {
  if (!Wc) return TACORL_EINVAL;
  return pr_encoder_fused_launch(emb, ld_emb, params, params_bf16, offsets, pooled, B, D, T, H, FF, L, Wc, bc, eps, head,
                                 plan, A, min_std, stream);
}
extern "C" int tacorl_pr_head_compose(const float* w_fc, const float* b_fc, const float* w_head, const float* b_head,
                                      float* Wc, float* bc, int D, int FC, int A2, tacorl_stream_t stream) {
  if ((D != PR_D && D != D6) || FC < 1 || A2 < 1 || !w_fc || !b_fc || !w_head || !b_head || !Wc || !bc) return TACORL_EINVAL;
  hipLaunchKernelGGL(pr_head_compose_kernel, dim3(A2), dim3(256), 0, (hipStream_t)stream, w_fc, b_fc, w_head, b_head, Wc, bc,
                     FC, D);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

/* Train-mode backward of the same encoder: the whole input-gradient chain in ONE launch (+ one small reduce), from the
 * gradient of the time-pooled output d_pool [B][32] to dx [B T][32].  saved[9 l + k]: what tacorl_pr_encoder_fused_train
 * wrote; dz[4 l + k], k = 0..3: outputs for the per-op weight-gradient GEMMs - LayerNorm-2 input gradient [B T][32]
 * (linear2's dZ), the masked hidden gradient [B T][FF] (linear1's dZ), LayerNorm-1 input gradient [B T][32] (out-proj's
 * dZ), d(q|k|v) [B T][96] (in-proj's dZ); wt[4 l + {0,1,2,3}]: W1^T [32][FF], W2^T [FF][32], Wo^T [32][32], Win^T [32][96] as bf16
 * (tacorl_transpose_to_bf16; the last two may be NULL);
 * ln_part: scratch of L * 2 * B * 64 floats; ln_grads[4 l + k]: norm1.weight, norm1.bias, norm2.weight, norm2.bias gradients.
 * d_pool == NULL: d_pool = d_head Wc is computed in the launch (d_head [B][A2], Wc [A2][32] from tacorl_pr_head_compose).
 * Reference: autograd through plan_recognition_transformer.py:70-88 (nn.TransformerEncoderLayer, post-norm, ReLU). */
extern "C" int tacorl_pr_encoder_bwd_fused(const float* params, const long* offsets, const float* d_pool, const float* d_head,
                                           const float* Wc, int A2, float* dx, const float* const* saved, float* const* dz,
                                           const void* const* wt, float* ln_part, float* const* ln_grads, int B, int D, int T,
                                           int H, int FF, int L, tacorl_stream_t stream) {
  // wt[4 l + {0, 1, 2, 3}]: linear1.weight^T, linear2.weight^T, out_proj.weight^T [32][32], in_proj_weight^T [32][96] as bf16
  // (the last two may be NULL: gathered transposed from the fp32 block inside the launch)
  // backward: d_model 32 (PR_D: 64-column LayerNorm partials, [32][*] transposes and saves) and window 16 only - the
  // forward also takes d_model 64 / window 32, which must be refused here instead of running out of bounds
  if (!tacorl_pr_encoder_fused_train_supported(D, T, H, FF, L) || D != PR_D || T != PR_T || B < 1) return TACORL_EINVAL;
  if (!params || !dx || !saved || !dz || !wt || !ln_part || !ln_grads) return TACORL_EINVAL;
  if (!d_pool && (!d_head || !Wc || A2 < 1)) return TACORL_EINVAL;
  if (((uintptr_t)params | (uintptr_t)d_pool | (uintptr_t)dx | (uintptr_t)ln_part | (uintptr_t)Wc) & 15) return TACORL_EINVAL;
  PrBwdArgs a{};
  PrLnReduceArgs r{};
  a.P = params; a.d_pool = d_pool; a.d_head = d_head; a.Wc = Wc; a.A2 = A2; a.dx = dx; a.lnpart = ln_part; a.B = B; a.FF = FF; a.L = L;
  r.part = ln_part; r.B = B;
  for (int l = 0; l < L; l++) {
    const long* q = offsets + 1 + 12 * l;
    for (int k = 0; k < 12; k++)
      if (q[k] % 4) return TACORL_EINVAL;
    a.l[l] = PrLayerOff{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11]};
    const float* const* sv = saved + 9 * l;
    float* const* z = dz + 4 * l;
    for (int k = 0; k < 9; k++)
      if (!sv[k] || ((uintptr_t)sv[k] & 15)) return TACORL_EINVAL;
    for (int k = 0; k < 4; k++)
      if (!z[k] || ((uintptr_t)z[k] & 15) || !ln_grads[4 * l + k]) return TACORL_EINVAL;
    if (!wt[4 * l] || !wt[4 * l + 1] ||
        (((uintptr_t)wt[4 * l] | (uintptr_t)wt[4 * l + 1] | (uintptr_t)wt[4 * l + 2] | (uintptr_t)wt[4 * l + 3]) & 15))
      return TACORL_EINVAL;
    a.s[l] = PrBwdL{sv[0], sv[1], sv[2], sv[3], sv[4], sv[5], sv[6], sv[7], sv[8], z[0], z[1], z[2], z[3],
                    (const __bf16*)wt[4 * l], (const __bf16*)wt[4 * l + 1], (const __bf16*)wt[4 * l + 2], (const __bf16*)wt[4 * l + 3]};
    r.dw[2 * l] = ln_grads[4 * l]; r.db[2 * l] = ln_grads[4 * l + 1];
    r.dw[2 * l + 1] = ln_grads[4 * l + 2]; r.db[2 * l + 1] = ln_grads[4 * l + 3];
  }
  hipLaunchKernelGGL(pr_encoder_bwd_fused_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(pr_ln_reduce_kernel, dim3(2 * L), dim3(256), 0, (hipStream_t)stream, r);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

#ifdef PR_STAMPS
extern "C" int tacorl_pr_stamps_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pr_stamps), sizeof(pr_stamps)) != hipSuccess) return TACORL_ELAUNCH;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(pr_stamps), z, sizeof(z)) != hipSuccess) return TACORL_ELAUNCH;
  }
  return TACORL_OK;
}
#endif
