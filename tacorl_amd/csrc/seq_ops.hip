// Sequence-side kernels of the plan-recognition transformer (reference
// networks/plan_encoders/plan_recognition_transformer.py:70-105 + torch's post-norm
// nn.TransformerEncoderLayer) and of the plan sampling.  The GEMMs go through
// tacorl_linear_fwd; these are the small latency-bound pieces between them.
#include <stdio.h>

#include "../../include/tacorl_hip.h"
#include "common.h"

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH)

// out[r][0:D] = x[r][0:D] + add[r % T][0:D];  out[r][D:Dp] = add[r % T][D:Dp]  (zero-padded x)
__global__ void add_rows_bcast_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ add,
                                      float* __restrict__ out, int R, int T, int D, int Dp) {
  const long total = (long)R * Dp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / Dp), d = (int)(i - (long)r * Dp);
    out[i] = (d < D ? x[(long)r * ldx + d] : 0.f) + add[(long)(r % T) * Dp + d];
  }
}
extern "C" int tacorl_add_rows_bcast(const float* x, int ldx, const float* add, float* out, int R, int T, int D, int Dp,
                                     tacorl_stream_t stream) {
  if (R <= 0) return TACORL_OK;
  const long total = (long)R * Dp;
  hipLaunchKernelGGL(add_rows_bcast_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                     add, out, R, T, D, Dp);
  return LAUNCH_OK();
}

// Multi-head self-attention core for short sequences (T <= 64, head_dim <= 16).
// qkv: [B*T][3D] (q | k | v, heads contiguous inside each), out: [B*T][D].
// One thread per (b, head, query): scores = (q/sqrt(hd)) . k, softmax over keys, weighted sum of v.
#define ATT_MAX_T 64
#define ATT_MAX_HD 16
// keep (may be NULL): attention-probability dropout of nn.MultiheadAttention in train mode, uint8 [B][H][T][T]
// keep flags; a kept probability is scaled by ks = 1/(1-p)  (out = (softmax(..) o keep * ks) v).
__global__ void attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, int B, int T, int D, int H,
                                     const unsigned char* __restrict__ keep, float ks) {
  const int hd = D / H;
  const long total = (long)B * H * T;
  const float scale = 1.0f / sqrtf((float)hd);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T), h = (int)((i / T) % H), b = (int)(i / ((long)T * H));
    const float* base = qkv + (long)b * T * 3 * D;
    float q[ATT_MAX_HD], o[ATT_MAX_HD], s[ATT_MAX_T];
    for (int e = 0; e < hd; e++) { q[e] = base[(long)t * 3 * D + h * hd + e] * scale; o[e] = 0.f; }
    float mx = -INFINITY;
    for (int j = 0; j < T; j++) {
      float a = 0.f;
      for (int e = 0; e < hd; e++) a += q[e] * base[(long)j * 3 * D + D + h * hd + e];
      s[j] = a; mx = fmaxf(mx, a);
    }
    float se = 0.f;
    for (int j = 0; j < T; j++) { s[j] = expf(s[j] - mx); se += s[j]; }
    const unsigned char* kp = keep ? keep + (((long)b * H + h) * T + t) * T : nullptr;
    for (int j = 0; j < T; j++) {
      const float p = s[j] / se * (kp ? (kp[j] ? ks : 0.f) : 1.f);
      for (int e = 0; e < hd; e++) o[e] += p * base[(long)j * 3 * D + 2 * D + h * hd + e];
    }
    for (int e = 0; e < hd; e++) out[((long)b * T + t) * D + h * hd + e] = o[e];
  }
}
static int attention_fwd_launch(const float* qkv, float* out, int B, int T, int D, int H, const unsigned char* keep, float ks,
                                tacorl_stream_t stream) {
  if (T > ATT_MAX_T || D % H || D / H > ATT_MAX_HD) return TACORL_EINVAL;
  const long total = (long)B * H * T;
  if (total <= 0) return TACORL_OK;
  hipLaunchKernelGGL(attention_fwd_kernel, dim3((int)((total + 127) / 128)), dim3(128), 0, (hipStream_t)stream, qkv, out,
                     B, T, D, H, keep, ks);
  return LAUNCH_OK();
}
extern "C" int tacorl_attention_fwd(const float* qkv, float* out, int B, int T, int D, int H, tacorl_stream_t stream) {
  return attention_fwd_launch(qkv, out, B, T, D, H, nullptr, 1.f, stream);
}
extern "C" int tacorl_attention_dropout_fwd(const float* qkv, float* out, const unsigned char* keep, float keep_scale, int B,
                                            int T, int D, int H, tacorl_stream_t stream) {
  if (!keep) return TACORL_EINVAL;
  return attention_fwd_launch(qkv, out, B, T, D, H, keep, keep_scale, stream);
}

// x[i] = keep[i] ? x[i] * keep_scale : 0 in place: nn.Dropout in train mode with the keep mask as an explicit input
// (applied to an activation in the forward and to its gradient in the backward).
__global__ void dropout_mul_kernel(float* __restrict__ x, const unsigned char* __restrict__ keep, float ks, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    x[i] = keep[i] ? x[i] * ks : 0.f;
}
extern "C" int tacorl_dropout_mul(float* x, const unsigned char* keep, float keep_scale, long n, tacorl_stream_t stream) {
  if (n <= 0) return TACORL_OK;
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(dropout_mul_kernel, dim3((int)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, x, keep,
                     keep_scale, n);
  return LAUNCH_OK();
}

// y = LayerNorm(x + res) * w + b  (eps inside sqrt, biased variance: torch F.layer_norm);
// stats[r] = {mean, rstd} kept for the backward.  One wave per row, D <= 256.
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                float* __restrict__ y, float* __restrict__ stats, int R,
                                                                int D, float eps) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  float v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int d = lane + 64 * i;
    v[i] = d < D ? x[(long)r * D + d] + (res ? res[(long)r * D + d] : 0.f) : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int d = lane + 64 * i;
    const float c = d < D ? v[i] - mean : 0.f;
    q += c * c;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int d = lane + 64 * i;
    if (d < D) y[(long)r * D + d] = (v[i] - mean) * rstd * w[d] + b[d];
  }
  if (stats && lane == 0) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
}
extern "C" int tacorl_add_layernorm_fwd(const float* x, const float* res, const float* w, const float* b, float* y,
                                        float* stats, int R, int D, float eps, tacorl_stream_t stream) {
  if (D > 256) return TACORL_EINVAL;
  if (R <= 0) return TACORL_OK;
  hipLaunchKernelGGL(add_layernorm_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, res, w, b, y,
                     stats, R, D, eps);
  return LAUNCH_OK();
}

// out[b][d] = mean_t x[b*T + t][d]
__global__ void mean_over_t_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int T, int D) {
  const long total = (long)B * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / D), d = (int)(i - (long)b * D);
    float s = 0.f;
    for (int t = 0; t < T; t++) s += x[((long)b * T + t) * D + d];
    out[i] = s / (float)T;
  }
}
extern "C" int tacorl_mean_over_t(const float* x, float* out, int B, int T, int D, tacorl_stream_t stream) {
  const long total = (long)B * D;
  if (total <= 0) return TACORL_OK;
  hipLaunchKernelGGL(mean_over_t_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, B,
                     T, D);
  return LAUNCH_OK();
}

// Posterior head: head[b] = [mean (A) | var_raw (A)]; std = softplus(var_raw) + min_std;
// plan = tanh(mean + eps*std)  (plan_recognition_transformer.py:100-104, distributions.py:137-140).
__global__ void pr_sample_kernel(const float* __restrict__ head, const float* __restrict__ eps, float* __restrict__ plan,
                                 float* __restrict__ mu_out, float* __restrict__ std_out, int B, int A, float min_std) {
  const long total = (long)B * A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / A), j = (int)(i - (long)b * A);
    const float mu = head[(long)b * 2 * A + j], vr = head[(long)b * 2 * A + A + j];
    const float sd = (vr > 20.f ? vr : log1pf(expf(vr))) + min_std;
    if (plan) plan[i] = tanhf(mu + eps[i] * sd);
    if (mu_out) mu_out[i] = mu;
    if (std_out) std_out[i] = sd;
  }
}
extern "C" int tacorl_pr_sample(const float* head, const float* eps, float* plan, float* mu_out, float* std_out, int B,
                                int A, float min_std, tacorl_stream_t stream) {
  const long total = (long)B * A;
  if (total <= 0) return TACORL_OK;
  hipLaunchKernelGGL(pr_sample_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, head, eps,
                     plan, mu_out, std_out, B, A, min_std);
  return LAUNCH_OK();
}

// =============================================================== action decoder glue
// x_seq[(t*B + b)] = [plan[b] (P) | emb[(b*T + t)] (E)]  for t < Tm  (time-major RNN input;
// reference action_decoder_logistic.py:279-281 with perceptual_emb = emb[:, :-1])
__global__ void build_ad_input_kernel(const float* __restrict__ plan, const float* __restrict__ emb, int ld_emb,
                                      float* __restrict__ out, int B, int T, int Tm, int P, int E) {
  const int W = P + E;
  const long total = (long)Tm * B * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % W);
    const long r = i / W;
    const int b = (int)(r % B), t = (int)(r / B);
    out[i] = c < P ? plan[(long)b * P + c] : emb[((long)b * T + t) * ld_emb + (c - P)];
  }
}
extern "C" int tacorl_build_ad_input(const float* plan, const float* emb, int ld_emb, float* out, int B, int T, int Tm,
                                     int P, int E, tacorl_stream_t stream) {
  const long total = (long)Tm * B * (P + E);
  if (total <= 0) return TACORL_OK;
  hipLaunchKernelGGL(build_ad_input_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, plan,
                     emb, ld_emb, out, B, T, Tm, P, E);
  return LAUNCH_OK();
}

// The same rows as bf16, zero padded to 128 columns: out[(t*B + b)][0..127] = bf16([plan[b] | emb[b*T + t] | 0 ...]) - the
// K-extension operand of the ring GEMM's layer-0 step (tacorl_rnn_linear_fwd_batch_ext), 16 bytes per thread.
__global__ __launch_bounds__(256) void build_ad_input_bf16_kernel(const float* __restrict__ plan, const float* __restrict__ emb,
                                                                  int ld_emb, __bf16* __restrict__ out, int B, int T, int Tm, int P,
                                                                  int E) {
  const long total = (long)Tm * B * 16;
  const int K = P + E;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = ((int)i & 15) * 8;
    const long r = i >> 4;
    const int b = (int)(r % B), t = (int)(r / B);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = c + j;
      v[j] = (__bf16)(k < P ? plan[(long)b * P + k] : (k < K ? emb[((long)b * T + t) * ld_emb + (k - P)] : 0.f));
    }
    *reinterpret_cast<bf16x8*>(out + r * 128 + c) = v;
  }
}
extern "C" int tacorl_build_ad_input_bf16(const float* plan, const float* emb, int ld_emb, void* out_bf16, int B, int T, int Tm,
                                          int P, int E, tacorl_stream_t stream) {
  if (P < 0 || E < 1 || P + E > 128 || B < 1 || Tm < 1 || ((uintptr_t)out_bf16 & 15)) return TACORL_EINVAL;
  const long total = (long)Tm * B * 16;
  hipLaunchKernelGGL(build_ad_input_bf16_kernel, dim3((int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, plan, emb, ld_emb, (__bf16*)out_bf16, B, T, Tm, P, E);
  return LAUNCH_OK();
}

// PlayLMP.training_step: the gradient entering the encoders, assembled in ONE launch.  d_emb[(b T + t)][c] already holds the
// action decoder's share; this adds the plan recognition's (dx, first D_in columns), the plan proposal's state share on the
// window's first frame (dS[b][c]) and its goal share on the last frame (dgin[b][c]) - in that order, as the four accumulating
// copies it replaces did - writes the sum back and hands every camera its 32 columns (f_dout[cam][(b T + t)][0..31]).
// (reference play_lmp_for_rl.py:200-257: autograd sums these paths into perceptual_emb.grad)
struct DembArgs { float* f_dout[8]; };
__global__ __launch_bounds__(256) void plmp_demb_finish_kernel(float* __restrict__ d_emb, const float* __restrict__ dx, int ld_dx,
                                                               int D_in, const float* __restrict__ dS, int ld_ds,
                                                               const float* __restrict__ dgin, DembArgs fo, int B, int T, int Ec) {
  const int q4 = Ec / 4;
  const long total = (long)B * T * q4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / q4;
    const int c = (int)(i - r * q4) * 4, b = (int)(r / T), t = (int)(r - (long)b * T);
    f32x4 v = *reinterpret_cast<const f32x4*>(d_emb + r * Ec + c);
    if (c < D_in) v += *reinterpret_cast<const f32x4*>(dx + r * ld_dx + c);
    if (t == 0) v += *reinterpret_cast<const f32x4*>(dS + (long)b * ld_ds + c);
    if (t == T - 1) v += *reinterpret_cast<const f32x4*>(dgin + (long)b * Ec + c);
    *reinterpret_cast<f32x4*>(d_emb + r * Ec + c) = v;
    float* o = fo.f_dout[c >> 5];
    if (o) *reinterpret_cast<f32x4*>(o + r * 32 + (c & 31)) = v;
  }
}
extern "C" int tacorl_plmp_demb_finish(float* d_emb, const float* dx, int ld_dx, int D_in, const float* dS, int ld_ds,
                                       const float* dgin, float* const* f_dout, int ncam, int B, int T, int Ec,
                                       tacorl_stream_t stream) {
  if (ncam < 1 || ncam > 8 || Ec != 32 * ncam || D_in % 4 || D_in > Ec || ld_dx % 4 || ld_ds % 4 || B < 1 || T < 1) return TACORL_EINVAL;
  if (((uintptr_t)d_emb | (uintptr_t)dx | (uintptr_t)dS | (uintptr_t)dgin) & 15) return TACORL_EINVAL;
  DembArgs fo{};
  for (int j = 0; j < ncam; j++) {
    if ((uintptr_t)f_dout[j] & 15) return TACORL_EINVAL;
    fo.f_dout[j] = f_dout[j];
  }
  const long total = (long)B * T * (Ec / 4);
  hipLaunchKernelGGL(plmp_demb_finish_kernel, dim3((int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, d_emb, dx, ld_dx, D_in, dS, ld_ds, dgin, fo, B, T, Ec);
  return LAUNCH_OK();
}

// The RNN's layer-0 input projection straight from (plan, frame embeddings): xin[t*B + b][n] = b_ih[n] +
// sum_k bf16(x[t*B + b][k]) bf16(W_ih[n][k]),  x = [plan[b] | emb[b*T + t]]  (K = P + E <= 64) - build_ad_input + the generic
// GEMM were two launches (4 + 21 us: a K = 48 contraction is all epilogue) at the head of the action-decoder branch.
// W is the MFMA A operand (a lane ends with 4 consecutive n of one row: 16-byte stores), a wave owns 16 rows x 512
// columns, a workgroup 64 rows; bf16 operand rounding and fp32 accumulation as the generic bf16 path.
__global__ __launch_bounds__(256) void ad_input_proj_kernel(const float* __restrict__ plan, const float* __restrict__ emb,
                                                            int ld_emb, const float* __restrict__ W, const float* __restrict__ bias,
                                                            float* __restrict__ out, int B, int T, int Tm, int P, int E, int H) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, m = lane & 15, g = lane >> 4;
  const int R = Tm * B, K = P + E;
  const int r = blockIdx.x * 64 + 16 * w + m, rc = r < R ? r : R - 1;
  const int b = rc % B, t = rc / B;
  bf16x8 X[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = 32 * ks + 8 * g + j;
      const float v = k < P ? plan[(long)b * P + k] : (k < K ? emb[((long)b * T + t) * ld_emb + (k - P)] : 0.f);
      X[ks][j] = (__bf16)v;
    }
  const int n_begin = blockIdx.y * 512, n_end = min(H, n_begin + 512);
  // four column tiles per iteration, every W load of the group in flight before the first conversion (one tile at a time
  // the loop was a chain of L2 round trips: 32 x ~1.5 us per wave)
  constexpr int NU = 4;
  const bool k2 = 32 + 8 * g < K;  // this lane's second k-step holds real columns (K = 48: g < 2)
  for (int n0 = n_begin; n0 < n_end; n0 += 16 * NU) {
    f32x4 raw[NU][4];
    f32x4 acc[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int nn = min(n0 + 16 * u, n_end - 16);
      const float* wr = W + (long)(nn + m) * K + 8 * g;  // A operand: row nn + m of W, 8 consecutive k per k-step
      raw[u][0] = *reinterpret_cast<const f32x4*>(wr);
      raw[u][1] = *reinterpret_cast<const f32x4*>(wr + 4);
      raw[u][2] = k2 ? *reinterpret_cast<const f32x4*>(wr + 32) : f32x4{0.f, 0.f, 0.f, 0.f};
      raw[u][3] = k2 ? *reinterpret_cast<const f32x4*>(wr + 36) : f32x4{0.f, 0.f, 0.f, 0.f};
      acc[u] = *reinterpret_cast<const f32x4*>(bias + nn + 4 * g);  // D rows = n: this lane's 4 consecutive columns
    }
#pragma unroll
    for (int u = 0; u < NU; u++) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        bf16x8 A;
#pragma unroll
        for (int j = 0; j < 8; j++) A[j] = (__bf16)raw[u][2 * ks + (j >> 2)][j & 3];
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, X[ks], acc[u], 0, 0, 0);
      }
      if (r < R && n0 + 16 * u < n_end) *reinterpret_cast<f32x4*>(out + (long)r * H + n0 + 16 * u + 4 * g) = acc[u];
    }
  }
}
extern "C" int tacorl_ad_input_proj(const float* plan, const float* emb, int ld_emb, const float* w_ih, const float* b_ih,
                                    float* out, int B, int T, int Tm, int P, int E, int H, tacorl_stream_t stream) {
  // (K % 8 == 0 and 16-byte aligned W rows: the fragment loads are 16-byte vectors; columns 32.. need whole 8-groups)
  // (P + E >= 32: the first k-step reads W columns 0..31 of every row unmasked)
  if (P + E < 32 || P + E > 64 || (P + E) % 8 || H % 16 || H < 16 || B < 1 || Tm < 1) return TACORL_EINVAL;
  if (((uintptr_t)out | (uintptr_t)b_ih | (uintptr_t)w_ih) & 15) return TACORL_EINVAL;
  const int R = B * Tm;
  hipLaunchKernelGGL(ad_input_proj_kernel, dim3((R + 63) / 64, (H + 511) / 512), dim3(256), 0, (hipStream_t)stream, plan, emb,
                     ld_emb, w_ih, b_ih, out, B, T, Tm, P, E, H);
  return LAUNCH_OK();
}

// Discretised logistic mixture NLL + gripper cross-entropy, forward and backward fused
// (reference action_decoder_logistic.py:110-235; bounds +-1, num_classes bins, n_mix mixtures).
// heads[(t*B+b)] = [means (Da*K) | log_scales (Da*K) | logit_probs (Da*K) | gripper (2)], ld = ldh.
// actions: batch-major [B][T][Da+1] (the first Tm steps are used).  One thread per (row, action dim).
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus_(float x) { return x > 20.f ? x : log1pf(expf(x)); }

#define LM_MAXK 16
__global__ __launch_bounds__(256) void logistic_mixture_kernel(const float* __restrict__ heads, int ldh,
                                                               const float* __restrict__ actions, float* __restrict__ d_heads,
                                                               float* __restrict__ partial, int B, int T, int Tm, int Da,
                                                               int K, float half_bin, float log_bins_half,
                                                               float gripper_alpha, float grad_scale) {
  __shared__ float sh[4];
  const int R = Tm * B;
  const long total = (long)R * Da;
  float loss = 0.f, hits = 0.f;
  const float gR = grad_scale / (float)R;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int a = (int)(i % Da);
    const long r = i / Da;
    const int b = (int)(r % B), t = (int)(r / B);
    const float* h = heads + r * ldh;
    const float x = actions[((long)b * T + t) * (Da + 1) + a];
    float lp[LM_MAXK], gm[LM_MAXK], gs[LM_MAXK], pl[LM_MAXK];
    float mxl = -INFINITY;
    for (int k = 0; k < K; k++) { pl[k] = h[2 * Da * K + a * K + k]; mxl = fmaxf(mxl, pl[k]); }
    float sel = 0.f;
    for (int k = 0; k < K; k++) sel += expf(pl[k] - mxl);
    const float lse_l = mxl + logf(sel);
    float mx = -INFINITY;
    for (int k = 0; k < K; k++) {
      const float m = h[a * K + k], lsr = h[Da * K + a * K + k];
      const float ls = fmaxf(lsr, -5.0f);
      const float c = x - m, inv = expf(-ls);
      const float plus_in = inv * (c + half_bin), min_in = inv * (c - half_bin), mid_in = inv * c;
      const float cp = sigmoidf_(plus_in), cm = sigmoidf_(min_in), delta = cp - cm;
      float v, dm, ds;
      if (x < -1.0f + 1e-3f) {
        v = plus_in - softplus_(plus_in);
        const float g = 1.f - cp; dm = g * (-inv); ds = g * (-plus_in);
      } else if (x > 1.0f - 1e-3f) {
        v = -softplus_(min_in);
        const float g = -cm; dm = g * (-inv); ds = g * (-min_in);
      } else if (delta > 1e-5f) {
        v = logf(fmaxf(delta, 1e-12f));
        const float gp = cp * (1.f - cp) / delta, gq = -cm * (1.f - cm) / delta;
        dm = (gp + gq) * (-inv); ds = gp * (-plus_in) + gq * (-min_in);
      } else {
        v = mid_in - ls - 2.f * softplus_(mid_in) - log_bins_half;
        const float g = 1.f - 2.f * sigmoidf_(mid_in); dm = g * (-inv); ds = g * (-mid_in) - 1.f;
      }
      if (lsr < -5.0f) ds = 0.f;
      lp[k] = v + (pl[k] - lse_l); gm[k] = dm; gs[k] = ds;
      mx = fmaxf(mx, lp[k]);
    }
    float se = 0.f;
    for (int k = 0; k < K; k++) se += expf(lp[k] - mx);
    loss -= mx + logf(se);
    if (d_heads) {
      float* d = d_heads + r * ldh;
      for (int k = 0; k < K; k++) {
        const float w = expf(lp[k] - mx) / se, p = expf(pl[k] - lse_l);
        d[a * K + k] = -gR * w * gm[k];
        d[Da * K + a * K + k] = -gR * w * gs[k];
        d[2 * Da * K + a * K + k] = -gR * (w - p);
      }
    }
    if (a == 0) {  // gripper cross-entropy for this row (nn.CrossEntropyLoss, mean over rows)
      const float g0 = h[3 * Da * K], g1 = h[3 * Da * K + 1], mg = fmaxf(g0, g1);
      const float lz = mg + logf(expf(g0 - mg) + expf(g1 - mg));
      const int y = actions[((long)b * T + t) * (Da + 1) + Da] == -1.0f ? 0 : (int)actions[((long)b * T + t) * (Da + 1) + Da];
      loss += gripper_alpha * (lz - (y ? g1 : g0));
      hits += ((g1 > g0) ? 1 : 0) == y ? 1.f : 0.f;  // gripper_bounds[argmax] vs ground truth (play_lmp_for_rl.py:166-176)
      if (d_heads) {
        float* d = d_heads + r * ldh;
        d[3 * Da * K] = gR * gripper_alpha * (expf(g0 - lz) - (y == 0 ? 1.f : 0.f));
        d[3 * Da * K + 1] = gR * gripper_alpha * (expf(g1 - lz) - (y == 1 ? 1.f : 0.f));
      }
    }
  }
  loss = wave_sum(loss);
  hits = wave_sum(hits);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = loss;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = hits;
  __syncthreads();
  if (threadIdx.x == 0) partial[gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// The same loss with the K mixtures of a (row, action dim) pair on 16 lanes (K <= 16): the per-mixture log-probability -
// a dozen transcendentals - is computed once per lane and the two logsumexps over k are 16-lane butterflies, instead of
// one thread looping over 10 mixtures (15.8 us for 3 MB of heads, on the logging-only action-decoder branch that is the
// last thing of the step to finish).  Sums over k are taken in butterfly order (fixed: deterministic), not 0..K-1.
__device__ __forceinline__ float grp16_max(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 16));
  return v;
}
__device__ __forceinline__ float grp16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
  return v;
}
__global__ __launch_bounds__(256) void logistic_mixture_k16_kernel(const float* __restrict__ heads, int ldh,
                                                                   const float* __restrict__ actions, float* __restrict__ d_heads,
                                                                   float* __restrict__ partial, int B, int T, int Tm, int Da,
                                                                   int K, float half_bin, float log_bins_half,
                                                                   float gripper_alpha, float grad_scale) {
  __shared__ float sh[4];
  const int R = Tm * B, k = threadIdx.x & 15;
  const long total = (long)R * Da;
  float loss = 0.f, hits = 0.f;
  const float gR = grad_scale / (float)R;
  const bool on = k < K;
  for (long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4); i < total; i += (long)gridDim.x * 16) {
    const int a = (int)(i % Da);
    const long r = i / Da;
    const int b = (int)(r % B), t = (int)(r / B);
    const float* h = heads + r * ldh;
    const float x = actions[((long)b * T + t) * (Da + 1) + a];
    const float pl = on ? h[2 * Da * K + a * K + k] : -INFINITY;
    const float mxl = grp16_max(pl);
    const float lse_l = mxl + logf(grp16_sum(on ? expf(pl - mxl) : 0.f));
    float lp = -INFINITY, dm = 0.f, ds = 0.f;
    if (on) {
      const float m = h[a * K + k], lsr = h[Da * K + a * K + k];
      const float ls = fmaxf(lsr, -5.0f);
      const float c = x - m, inv = expf(-ls);
      const float plus_in = inv * (c + half_bin), min_in = inv * (c - half_bin), mid_in = inv * c;
      const float cp = sigmoidf_(plus_in), cm = sigmoidf_(min_in), delta = cp - cm;
      float v;
      if (x < -1.0f + 1e-3f) {
        v = plus_in - softplus_(plus_in);
        const float g = 1.f - cp; dm = g * (-inv); ds = g * (-plus_in);
      } else if (x > 1.0f - 1e-3f) {
        v = -softplus_(min_in);
        const float g = -cm; dm = g * (-inv); ds = g * (-min_in);
      } else if (delta > 1e-5f) {
        v = logf(fmaxf(delta, 1e-12f));
        const float gp = cp * (1.f - cp) / delta, gq = -cm * (1.f - cm) / delta;
        dm = (gp + gq) * (-inv); ds = gp * (-plus_in) + gq * (-min_in);
      } else {
        v = mid_in - ls - 2.f * softplus_(mid_in) - log_bins_half;
        const float g = 1.f - 2.f * sigmoidf_(mid_in); dm = g * (-inv); ds = g * (-mid_in) - 1.f;
      }
      if (lsr < -5.0f) ds = 0.f;
      lp = v + (pl - lse_l);
    }
    const float mx = grp16_max(lp);
    const float ek = on ? expf(lp - mx) : 0.f;
    const float se = grp16_sum(ek);
    if (k == 0) loss -= mx + logf(se);
    if (d_heads && on) {
      float* d = d_heads + r * ldh;
      const float w = ek / se, p = expf(pl - lse_l);
      d[a * K + k] = -gR * w * dm;
      d[Da * K + a * K + k] = -gR * w * ds;
      d[2 * Da * K + a * K + k] = -gR * (w - p);
    }
    if (a == 0 && k == 0) {  // gripper cross-entropy for this row (nn.CrossEntropyLoss, mean over rows)
      const float g0 = h[3 * Da * K], g1 = h[3 * Da * K + 1], mg = fmaxf(g0, g1);
      const float lz = mg + logf(expf(g0 - mg) + expf(g1 - mg));
      const int y = actions[((long)b * T + t) * (Da + 1) + Da] == -1.0f ? 0 : (int)actions[((long)b * T + t) * (Da + 1) + Da];
      loss += gripper_alpha * (lz - (y ? g1 : g0));
      hits += ((g1 > g0) ? 1 : 0) == y ? 1.f : 0.f;
      if (d_heads) {
        float* d = d_heads + r * ldh;
        d[3 * Da * K] = gR * gripper_alpha * (expf(g0 - lz) - (y == 0 ? 1.f : 0.f));
        d[3 * Da * K + 1] = gR * gripper_alpha * (expf(g1 - lz) - (y == 1 ? 1.f : 0.f));
      }
    }
  }
  loss = wave_sum(loss);
  hits = wave_sum(hits);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = loss;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = hits;
  __syncthreads();
  if (threadIdx.x == 0) partial[gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// out[0] = scale * sum(partial[0:n]), out[1] = scale * sum(partial[n:2n])
__global__ void scaled_sum_kernel(const float* __restrict__ partial, int n, float scale, float* out) {
  __shared__ float sh[4];
  for (int k = 0; k < 2; k++) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[k * n + i];
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[k] = (sh[0] + sh[1] + sh[2] + sh[3]) * scale;
  }
}
extern "C" size_t tacorl_logistic_mixture_ws_bytes(int B, int Tm, int Da) {
  (void)B; (void)Tm; (void)Da;
  return (size_t)1024 * 2 * sizeof(float);  // per-block partials of (loss, gripper hits): at most 1024 blocks
}
extern "C" int tacorl_logistic_mixture_loss(const float* heads, int ldh, const float* actions, float* d_heads,
                                            float* loss_out, int B, int T, int Tm, int Da, int K, int num_classes,
                                            float gripper_alpha, float grad_scale, void* ws, size_t ws_bytes,
                                            tacorl_stream_t stream) {
  if (K > LM_MAXK) return TACORL_EINVAL;
  if (ws_bytes < tacorl_logistic_mixture_ws_bytes(B, Tm, Da)) return TACORL_ENOMEM;
  // 16 lanes per (row, action dim) pair: 16 pairs per block (TACORL_LM_SCALAR=1: one thread per pair, the first version)
  const char* lme = getenv("TACORL_LM_SCALAR");  // (read per call: a captured graph keeps what it was captured with)
  const int scalar = lme ? atoi(lme) : 0;
  long blocks = scalar ? ((long)B * Tm * Da + 255) / 256 : ((long)B * Tm * Da + 15) / 16;
  if (blocks > 1024) blocks = 1024;
  if (blocks <= 0) return TACORL_OK;
  if (scalar)
    hipLaunchKernelGGL(logistic_mixture_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, heads, ldh, actions,
                       d_heads, (float*)ws, B, T, Tm, Da, K, 1.0f / (float)(num_classes - 1),
                       logf((float)(num_classes - 1) / 2.f), gripper_alpha, grad_scale);
  else
    hipLaunchKernelGGL(logistic_mixture_k16_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, heads, ldh, actions,
                       d_heads, (float*)ws, B, T, Tm, Da, K, 1.0f / (float)(num_classes - 1),
                       logf((float)(num_classes - 1) / 2.f), gripper_alpha, grad_scale);
  // loss_out == NULL: the per-block partials stay in ws and tacorl_logistic_mixture_finish sums them when the scalar is read
  // (a logging-only loss: the one-block sum was the last launch of the step's action-decoder branch, 9 us of launch latency
  // per step for a number the host looks at every log_every_n_steps)
  if (loss_out)
    hipLaunchKernelGGL(scaled_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, (int)blocks,
                       1.0f / (float)(B * Tm), loss_out);
  return LAUNCH_OK();
}
extern "C" int tacorl_logistic_mixture_finish(const void* ws, size_t ws_bytes, int B, int Tm, int Da, float* loss_out,
                                              tacorl_stream_t stream) {
  if (!loss_out || !ws) return TACORL_EINVAL;
  if (ws_bytes < tacorl_logistic_mixture_ws_bytes(B, Tm, Da)) return TACORL_ENOMEM;
  const char* lme = getenv("TACORL_LM_SCALAR");
  const int scalar = lme ? atoi(lme) : 0;
  long blocks = scalar ? ((long)B * Tm * Da + 255) / 256 : ((long)B * Tm * Da + 15) / 16;
  if (blocks > 1024) blocks = 1024;
  if (blocks <= 0) return TACORL_OK;
  hipLaunchKernelGGL(scaled_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, (int)blocks,
                     1.0f / (float)(B * Tm), loss_out);
  return LAUNCH_OK();
}

// ActionDecoderLogistic._sample (reference action_decoder_logistic.py:238-266): Gumbel-max choice of the mixture
// component, inversion sampling of that logistic, argmax gripper class.  One thread per (row, action dim).
// heads as in logistic_mixture_kernel; rand_a [R][Da][K], rand_b [R][Da] U(0,1) draws in the heads' row order;
// out [R][Da+1] = [actions | gripper command -1/+1].
__global__ void logistic_mixture_sample_kernel(const float* __restrict__ heads, int ldh, const float* __restrict__ rand_a,
                                               const float* __restrict__ rand_b, float* __restrict__ out, int R, int Da, int K) {
  const long total = (long)R * Da;
  const float r1 = 1e-5f, r2 = 1.0f - 1e-5f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int a = (int)(i % Da);
    const long r = i / Da;
    const float* h = heads + r * ldh;
    const float* mean = h + a * K;
    const float* lsc = h + (Da + a) * K;
    const float* lg = h + (2 * Da + a) * K;
    float best = -INFINITY;
    int arg = 0;
    for (int k = 0; k < K; k++) {
      const float u = (r1 - r2) * rand_a[i * K + k] + r2;
      const float t = lg[k] - logf(-logf(u));
      if (t > best) { best = t; arg = k; }  // first maximum, as torch.argmax
    }
    const float ls = fmaxf(lsc[arg], -5.0f);  // LOG_SIG_MIN (:18,292)
    const float u = (r1 - r2) * rand_b[i] + r2;
    out[r * (Da + 1) + a] = mean[arg] + expf(ls) * (logf(u) - logf(1.0f - u));
    if (a == 0) out[r * (Da + 1) + Da] = h[3 * Da * K + 1] > h[3 * Da * K] ? 1.0f : -1.0f;
  }
}
extern "C" int tacorl_logistic_mixture_sample(const float* heads, int ldh, const float* rand_a, const float* rand_b,
                                              float* out, int R, int Da, int K, tacorl_stream_t stream) {
  if (K > LM_MAXK || R < 0) return TACORL_EINVAL;
  const long total = (long)R * Da;
  if (total <= 0) return TACORL_OK;
  long blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(logistic_mixture_sample_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, heads, ldh, rand_a,
                     rand_b, out, R, Da, K);
  return LAUNCH_OK();
}

// ====================================================================== backward glue
// out = (dy (+ add)) * [h > 0]    (ReLU-RNN: last BPTT step, no recurrent term yet)
__global__ void relu_mask_mul_kernel(const float* __restrict__ dy, const float* __restrict__ add,
                                     const float* __restrict__ h, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = h[i] > 0.f ? dy[i] + (add ? add[i] : 0.f) : 0.f;
}
extern "C" int tacorl_relu_mask_mul(const float* dy, const float* add, const float* h, float* out, long n,
                                    tacorl_stream_t stream) {
  if (n <= 0) return TACORL_OK;
  hipLaunchKernelGGL(relu_mask_mul_kernel, dim3((int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, add, h, out, n);
  return LAUNCH_OK();
}

// Backward of build_ad_input: d_plan[b] = sum_t dx[(t,b)][0:P]; d_emb[(b*T+t)][0:E] (+)= dx[(t,b)][P:], t < Tm.
// accumulate = 2: overwrite AND zero the rows t >= Tm (the whole [B T][E] block is defined afterwards: no fill launch in front).
__global__ void ad_input_bwd_kernel(const float* __restrict__ dx, float* __restrict__ d_plan, float* __restrict__ d_emb,
                                    int ld_emb, int B, int T, int Tm, int P, int E, int accumulate) {
  const int W = P + E, Te = accumulate == 2 ? T : Tm;
  const long total = (long)B * (P + (long)Te * E);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (i < (long)B * P) {
      const int b = (int)(i / P), c = (int)(i % P);
      float s = 0.f;
      for (int t = 0; t < Tm; t++) s += dx[((long)t * B + b) * W + c];
      d_plan[i] = s;
    } else {
      const long j = i - (long)B * P;
      const int c = (int)(j % E);
      const long r = j / E;
      const int t = (int)(r % Te), b = (int)(r / Te);
      float* d = d_emb + ((long)b * T + t) * ld_emb + c;
      const float v = t < Tm ? dx[((long)t * B + b) * W + P + c] : 0.f;
      *d = accumulate == 1 ? *d + v : v;
    }
  }
}
extern "C" int tacorl_ad_input_bwd(const float* dx, float* d_plan, float* d_emb, int ld_emb, int B, int T, int Tm, int P,
                                   int E, int accumulate, tacorl_stream_t stream) {
  const long total = (long)B * (P + (long)(accumulate == 2 ? T : Tm) * E);
  if (total <= 0) return TACORL_OK;
  hipLaunchKernelGGL(ad_input_bwd_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx,
                     d_plan, d_emb, ld_emb, B, T, Tm, P, E, accumulate);
  return LAUNCH_OK();
}

// dst[(b*T+t)][d] (+)= src[b][d] * scale   (backward of the mean over time)
__global__ void bcast_over_t_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int T, int D,
                                    float scale, int accumulate) {
  const long total = (long)B * T * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const long b = i / ((long)T * D);
    const float v = src[b * D + d] * scale;
    dst[i] = accumulate ? dst[i] + v : v;
  }
}
extern "C" int tacorl_bcast_over_t(const float* src, float* dst, int B, int T, int D, float scale, int accumulate,
                                   tacorl_stream_t stream) {
  const long total = (long)B * T * D;
  if (total <= 0) return TACORL_OK;
  hipLaunchKernelGGL(bcast_over_t_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     B, T, D, scale, accumulate);
  return LAUNCH_OK();
}

// Attention backward, one block of T threads per (batch, head); probabilities are recomputed.
// keep / ks as in the forward: out_i = sum_j p_ij w_ij v_j with w_ij = keep_ij * ks, so dP_ij = w_ij (dO_i . v_j).
__global__ void attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ d_out,
                                     float* __restrict__ d_qkv, int B, int T, int D, int H,
                                     const unsigned char* __restrict__ keep, float ks) {
  __shared__ float s_m[ATT_MAX_T], s_l[ATT_MAX_T], s_dd[ATT_MAX_T];
  const int hd = D / H, b = blockIdx.x / H, h = blockIdx.x % H, i = threadIdx.x;
  const float scale = 1.0f / sqrtf((float)hd);
  const float* base = qkv + (long)b * T * 3 * D;
  const float* dob = d_out + (long)b * T * D;
  float* dqb = d_qkv + (long)b * T * 3 * D;
  float q[ATT_MAX_HD], dO[ATT_MAX_HD], acc[ATT_MAX_HD];
  const unsigned char* kb = keep ? keep + ((long)b * H + h) * T * T : nullptr;
#define ATT_W(r, j) (kb ? (kb[(r) * T + (j)] ? ks : 0.f) : 1.f)
  if (i < T) {
    for (int e = 0; e < hd; e++) { q[e] = base[(long)i * 3 * D + h * hd + e] * scale; dO[e] = dob[(long)i * D + h * hd + e]; }
    float mx = -INFINITY;
    for (int j = 0; j < T; j++) {
      float a = 0.f;
      for (int e = 0; e < hd; e++) a += q[e] * base[(long)j * 3 * D + D + h * hd + e];
      mx = fmaxf(mx, a);
    }
    float l = 0.f, dd = 0.f;
    for (int j = 0; j < T; j++) {
      float a = 0.f, dp = 0.f;
      for (int e = 0; e < hd; e++) {
        a += q[e] * base[(long)j * 3 * D + D + h * hd + e];
        dp += dO[e] * base[(long)j * 3 * D + 2 * D + h * hd + e];
      }
      const float w = expf(a - mx);
      l += w; dd += w * dp * ATT_W(i, j);
    }
    dd /= l;
    s_m[i] = mx; s_l[i] = l; s_dd[i] = dd;
    // dq_i = scale * sum_j dS_ij k_j
    for (int e = 0; e < hd; e++) acc[e] = 0.f;
    for (int j = 0; j < T; j++) {
      float a = 0.f, dp = 0.f;
      for (int e = 0; e < hd; e++) {
        a += q[e] * base[(long)j * 3 * D + D + h * hd + e];
        dp += dO[e] * base[(long)j * 3 * D + 2 * D + h * hd + e];
      }
      const float ds = expf(a - mx) / l * (dp * ATT_W(i, j) - dd);
      for (int e = 0; e < hd; e++) acc[e] += ds * base[(long)j * 3 * D + D + h * hd + e];
    }
    for (int e = 0; e < hd; e++) dqb[(long)i * 3 * D + h * hd + e] = acc[e] * scale;
  }
  __syncthreads();
  if (i < T) {  // this thread now owns key/value j = i
    float k[ATT_MAX_HD], v[ATT_MAX_HD], dk[ATT_MAX_HD], dv[ATT_MAX_HD];
    for (int e = 0; e < hd; e++) {
      k[e] = base[(long)i * 3 * D + D + h * hd + e]; v[e] = base[(long)i * 3 * D + 2 * D + h * hd + e];
      dk[e] = 0.f; dv[e] = 0.f;
    }
    for (int r = 0; r < T; r++) {
      float a = 0.f, dp = 0.f;
      for (int e = 0; e < hd; e++) {
        const float qs = base[(long)r * 3 * D + h * hd + e] * scale;
        a += qs * k[e];
        dp += dob[(long)r * D + h * hd + e] * v[e];
      }
      const float p = expf(a - s_m[r]) / s_l[r];
      const float wr = ATT_W(r, i);
      const float ds = p * (dp * wr - s_dd[r]);
      for (int e = 0; e < hd; e++) {
        dk[e] += ds * base[(long)r * 3 * D + h * hd + e] * scale;
        dv[e] += p * wr * dob[(long)r * D + h * hd + e];
      }
    }
    for (int e = 0; e < hd; e++) {
      dqb[(long)i * 3 * D + D + h * hd + e] = dk[e];
      dqb[(long)i * 3 * D + 2 * D + h * hd + e] = dv[e];
    }
  }
}
#undef ATT_W
// The same backward for the configured plan recognition (T = 16 key / query positions, head_dim 4; plan_recognition_transformer.py:
// d_model 32, 8 heads): 16 lanes per (sequence, head) pair - lane = query row in the first half, key / value row in the second -
// four pairs per wave, operands staged once in LDS as 16-byte rows (the general kernel above runs 16 of 64 threads per block
// through run-time loops over scalar global loads: 43 / 59 us at B = 32 / 256 on PlayLMP's dependent chain).
__global__ __launch_bounds__(256) void attention_bwd_t16_kernel(const float* __restrict__ qkv, const float* __restrict__ d_out,
                                                                float* __restrict__ d_qkv, int npairs, int D, int H,
                                                                const unsigned char* __restrict__ keep, float ks) {
  __shared__ __attribute__((aligned(16))) float rows[16][16 * 16];   // per pair: [row][q(4) | k(4) | v(4) | dO(4)]
  __shared__ __attribute__((aligned(16))) float pm[16][2][16 * 17];  // per pair: p[r][j], ds[r][j] (pitch 17)
  const int lp = threadIdx.x >> 4, i = threadIdx.x & 15, pair = blockIdx.x * 16 + lp;
  const bool on = pair < npairs;
  const int b = on ? pair / H : 0, h = on ? pair % H : 0;
  const float* base = qkv + ((long)b * 16 + i) * 3 * D + 4 * h;
  const f32x4 q = *reinterpret_cast<const f32x4*>(base), k = *reinterpret_cast<const f32x4*>(base + D),
              v = *reinterpret_cast<const f32x4*>(base + 2 * D), dO = *reinterpret_cast<const f32x4*>(d_out + ((long)b * 16 + i) * D + 4 * h);
  float* R = rows[lp];
  *reinterpret_cast<f32x4*>(R + i * 16) = q * 0.5f;  // 1 / sqrt(head_dim)
  *reinterpret_cast<f32x4*>(R + i * 16 + 4) = k;
  *reinterpret_cast<f32x4*>(R + i * 16 + 8) = v;
  *reinterpret_cast<f32x4*>(R + i * 16 + 12) = dO;
  __syncthreads();
  const unsigned char* kb = keep && on ? keep + (long)pair * 256 : nullptr;
  const f32x4 qs = q * 0.5f;
  float w[16], dp[16], mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const f32x4 kj = *reinterpret_cast<const f32x4*>(R + j * 16 + 4), vj = *reinterpret_cast<const f32x4*>(R + j * 16 + 8);
    w[j] = ((qs[0] * kj[0] + qs[1] * kj[1]) + qs[2] * kj[2]) + qs[3] * kj[3];
    dp[j] = ((dO[0] * vj[0] + dO[1] * vj[1]) + dO[2] * vj[2]) + dO[3] * vj[3];
    if (kb) dp[j] *= kb[i * 16 + j] ? ks : 0.f;
    mx = fmaxf(mx, w[j]);
  }
  float l = 0.f, dd = 0.f;
#pragma unroll
  for (int j = 0; j < 16; j++) { w[j] = expf(w[j] - mx); l += w[j]; dd += w[j] * dp[j]; }
  dd /= l;
  f32x4 dq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const float pj = w[j] / l, ds = pj * (dp[j] - dd);
    const f32x4 kj = *reinterpret_cast<const f32x4*>(R + j * 16 + 4);
    dq += ds * kj;
    pm[lp][0][i * 17 + j] = kb ? pj * (kb[i * 16 + j] ? ks : 0.f) : pj;  // the (dropped) probability that multiplied v_j
    pm[lp][1][i * 17 + j] = ds;
  }
  __syncthreads();
  // this lane as key / value row i: dk_i = scale sum_r ds[r][i] q_r, dv_i = sum_r p[r][i] dO_r
  f32x4 dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const f32x4 qr = *reinterpret_cast<const f32x4*>(R + r * 16), dor = *reinterpret_cast<const f32x4*>(R + r * 16 + 12);
    dk += pm[lp][1][r * 17 + i] * qr;   // (qr already carries the scale)
    dv += pm[lp][0][r * 17 + i] * dor;
  }
  if (on) {
    float* o = d_qkv + ((long)b * 16 + i) * 3 * D + 4 * h;
    *reinterpret_cast<f32x4*>(o) = dq * 0.5f;
    *reinterpret_cast<f32x4*>(o + D) = dk;
    *reinterpret_cast<f32x4*>(o + 2 * D) = dv;
  }
}
static int attention_bwd_launch(const float* qkv, const float* d_out, float* d_qkv, int B, int T, int D, int H,
                                const unsigned char* keep, float ks, tacorl_stream_t stream) {
  if (T > ATT_MAX_T || D % H || D / H > ATT_MAX_HD) return TACORL_EINVAL;
  if (B <= 0) return TACORL_OK;
  if (T == 16 && D / H == 4 && D % 4 == 0 && !(((uintptr_t)qkv | (uintptr_t)d_out | (uintptr_t)d_qkv) & 15)) {
    const int npairs = B * H;
    hipLaunchKernelGGL(attention_bwd_t16_kernel, dim3((npairs + 15) / 16), dim3(256), 0, (hipStream_t)stream, qkv, d_out, d_qkv,
                       npairs, D, H, keep, ks);
    return LAUNCH_OK();
  }
  hipLaunchKernelGGL(attention_bwd_kernel, dim3(B * H), dim3(64), 0, (hipStream_t)stream, qkv, d_out, d_qkv, B, T, D, H, keep,
                     ks);
  return LAUNCH_OK();
}
extern "C" int tacorl_attention_bwd(const float* qkv, const float* d_out, float* d_qkv, int B, int T, int D, int H,
                                    tacorl_stream_t stream) {
  return attention_bwd_launch(qkv, d_out, d_qkv, B, T, D, H, nullptr, 1.f, stream);
}
extern "C" int tacorl_attention_dropout_bwd(const float* qkv, const float* d_out, float* d_qkv, const unsigned char* keep,
                                            float keep_scale, int B, int T, int D, int H, tacorl_stream_t stream) {
  if (!keep) return TACORL_EINVAL;
  return attention_bwd_launch(qkv, d_out, d_qkv, B, T, D, H, keep, keep_scale, stream);
}

// LayerNorm(x + res) backward.  dv (gradient of x + res) and per-block partial sums of dw, db.
// partial: [nblocks][2*D]; finished by tacorl_colsum.
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ res, const float* __restrict__ w,
                                                                const float* __restrict__ stats, float* __restrict__ dv,
                                                                float* __restrict__ partial, int R, int D) {
  __shared__ float sh[4][2 * 256];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wv;
  float gw[4] = {0, 0, 0, 0}, gb[4] = {0, 0, 0, 0};
  if (r < R) {
    const float mean = stats[2 * r], rstd = stats[2 * r + 1];
    float xh[4], g[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int d = lane + 64 * i;
      if (d < D) {
        xh[i] = (x[(long)r * D + d] + (res ? res[(long)r * D + d] : 0.f) - mean) * rstd;
        const float dyv = dy[(long)r * D + d];
        g[i] = dyv * w[d];
        gw[i] = dyv * xh[i]; gb[i] = dyv;
        s1 += g[i]; s2 += g[i] * xh[i];
      } else { xh[i] = 0.f; g[i] = 0.f; }
    }
    s1 = wave_sum(s1) / (float)D; s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int d = lane + 64 * i;
      if (d < D) dv[(long)r * D + d] = rstd * (g[i] - s1 - xh[i] * s2);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int d = lane + 64 * i;
    if (d < D) { sh[wv][d] = gw[i]; sh[wv][256 + d] = gb[i]; }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) {
    partial[(long)blockIdx.x * 2 * D + d] = sh[0][d] + sh[1][d] + sh[2][d] + sh[3][d];
    partial[(long)blockIdx.x * 2 * D + D + d] = sh[0][256 + d] + sh[1][256 + d] + sh[2][256 + d] + sh[3][256 + d];
  }
}
// column sums of the [rows][2D] partial matrix: first D columns -> dw, last D -> db
// (256 threads per 64 columns: four row phases with eight independent loads in flight each, summed in fixed order - one
// thread per column over 128 dependent-latency rows took 30 us, four times per PlayLMP step at B = 32)
__global__ __launch_bounds__(256) void colsum2_kernel(const float* __restrict__ in, int rows, int D, float* __restrict__ dw,
                                                      float* __restrict__ db, int accumulate) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, ph = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < 2 * D) {
    int r = ph;
    for (; r + 28 < rows; r += 32) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = in[(long)(r + 4 * u) * 2 * D + c];
#pragma unroll
      for (int u = 0; u < 8; u++) s += t[u];
    }
    for (; r < rows; r += 4) s += in[(long)r * 2 * D + c];
  }
  part[ph][cl] = s;
  __syncthreads();
  if (ph == 0 && c < 2 * D) {
    const float t = ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl];
    float* o = c < D ? dw + c : db + (c - D);
    *o = accumulate ? *o + t : t;
  }
}
// first stage for tall partial matrices: block b sums rows [64 b, 64 b + 64) -> out[b][2D]
__global__ void colsum2_stage_kernel(const float* __restrict__ in, int rows, int D, float* __restrict__ out) {
  const int c = threadIdx.x, r0 = blockIdx.x * 64, r1 = min(rows, r0 + 64);
  if (c >= 2 * D) return;
  float s = 0.f;
  for (int r = r0; r < r1; r++) s += in[(long)r * 2 * D + c];
  out[(long)blockIdx.x * 2 * D + c] = s;
}
extern "C" size_t tacorl_add_layernorm_bwd_ws_bytes(int R, int D) {
  const size_t nb = (size_t)((R + 3) / 4);
  return (nb + (nb + 63) / 64) * 2 * D * sizeof(float);
}
extern "C" int tacorl_add_layernorm_bwd(const float* dy, const float* x, const float* res, const float* w,
                                        const float* stats, float* dv, float* dw, float* db, int R, int D,
                                        int accumulate, void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  if (D > 256) return TACORL_EINVAL;
  if (R <= 0) return TACORL_OK;
  if (ws_bytes < tacorl_add_layernorm_bwd_ws_bytes(R, D)) return TACORL_ENOMEM;
  const int nb = (R + 3) / 4;
  hipLaunchKernelGGL(add_layernorm_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, dy, x, res, w, stats, dv,
                     (float*)ws, R, D);
  const float* part = (const float*)ws;
  int rows = nb;
  if (nb > 128) {  // two-level column sum (fixed order): one thread per column over 1000+ rows took 235 us
    float* p2 = (float*)ws + (size_t)nb * 2 * D;
    rows = (nb + 63) / 64;
    hipLaunchKernelGGL(colsum2_stage_kernel, dim3(rows), dim3(2 * D), 0, (hipStream_t)stream, part, nb, D, p2);
    part = p2;
  }
  hipLaunchKernelGGL(colsum2_kernel, dim3((2 * D + 63) / 64), dim3(256), 0, (hipStream_t)stream, part, rows, D, dw, db,
                     accumulate);
  return LAUNCH_OK();
}

// Balanced KL between the plan-recognition posterior q and the plan-proposal prior p on the
// underlying Normals (reference play_lmp_for_rl.py:259-301), forward + backward fused:
//   kl = alpha*KL(sg(q)||p) + (1-alpha)*KL(q||sg(p)),  mean over batch;  loss = beta*kl.
// head_q = [mean | var_raw] (std = softplus+min_std), head_p = [mean_raw | log_std_raw] (policy clamps).
// d_head_* receive d(beta*kl)/d(raw head).  One block of 16 waves (B A = 4 096 elements at B = 256: four per thread; as 256
// threads the launch took 22 us there).
__global__ __launch_bounds__(1024) void gauss_kl_kernel(const float* __restrict__ hq, const float* __restrict__ hp,
                                                       float* __restrict__ dq, float* __restrict__ dp, int B, int A,
                                                       float alpha, float beta, float min_std, int balanced,
                                                       float grad_scale, float* out /*[2]: kl, beta*kl*/) {
  __shared__ float sh[16];
  float s = 0.f;
  const float c = beta * grad_scale / (float)B;
  for (int i = threadIdx.x; i < B * A; i += 1024) {
    const int b = i / A, j = i - b * A;
    const float mq = hq[(long)b * 2 * A + j], vr = hq[(long)b * 2 * A + A + j];
    const float sq = (vr > 20.f ? vr : log1pf(expf(vr))) + min_std;
    const float mr = hp[(long)b * 2 * A + j], lr = hp[(long)b * 2 * A + A + j];
    const float mp = fminf(fmaxf(mr, -9.f), 9.f), sp = expf(fminf(fmaxf(lr, -5.f), 2.f));
    const float vratio = (sq / sp) * (sq / sp), d = mq - mp, t1 = (d / sp) * (d / sp);
    const float kl = 0.5f * (vratio + t1 - 1.f - logf(vratio));
    s += kl;  // both terms have the same value; only their gradients differ
    // gradient of KL wrt posterior (m1,s1) and prior (m2,s2)
    const float g_m1 = d / (sp * sp), g_s1 = sq / (sp * sp) - 1.f / sq;
    const float g_m2 = -d / (sp * sp), g_s2 = -(sq * sq) / (sp * sp * sp) - (d * d) / (sp * sp * sp) + 1.f / sp;
    const float wq = balanced ? (1.f - alpha) : 1.f, wp = balanced ? alpha : 1.f;
    const float sig = vr > 20.f ? 1.f : 1.f / (1.f + expf(-vr));  // softplus'
    dq[(long)b * 2 * A + j] = c * wq * g_m1;
    dq[(long)b * 2 * A + A + j] = c * wq * g_s1 * sig;
    dp[(long)b * 2 * A + j] = (mr >= -9.f && mr <= 9.f) ? c * wp * g_m2 : 0.f;
    dp[(long)b * 2 * A + A + j] = (lr >= -5.f && lr <= 2.f) ? c * wp * g_s2 * sp : 0.f;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; w++) t += sh[w];
    const float kl = t / (float)B;
    out[0] = kl; out[1] = kl * beta;
  }
}
extern "C" int tacorl_gauss_kl_balanced(const float* head_q, const float* head_p, float* d_head_q, float* d_head_p,
                                        int B, int A, float kl_alpha, float kl_beta, float min_std, int balanced,
                                        float grad_scale, float* out2, tacorl_stream_t stream) {
  hipLaunchKernelGGL(gauss_kl_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, head_q, head_p, d_head_q, d_head_p, B,
                     A, kl_alpha, kl_beta, min_std, balanced, grad_scale, out2);
  return LAUNCH_OK();
}

// rsample backward: plan = tanh(mean + eps*std), std = softplus(var_raw)+min_std:
// d_head_q[mean] += dplan*(1-a^2);  d_head_q[var_raw] += dplan*(1-a^2)*eps*sigmoid(var_raw)
__global__ void pr_sample_bwd_kernel(const float* __restrict__ head, const float* __restrict__ eps,
                                     const float* __restrict__ d_plan, float* __restrict__ d_head, int B, int A,
                                     float min_std) {
  const long total = (long)B * A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / A), j = (int)(i - (long)b * A);
    const float mu = head[(long)b * 2 * A + j], vr = head[(long)b * 2 * A + A + j];
    const float sd = (vr > 20.f ? vr : log1pf(expf(vr))) + min_std;
    const float a = tanhf(mu + eps[i] * sd), g = d_plan[i] * (1.f - a * a);
    const float sig = vr > 20.f ? 1.f : 1.f / (1.f + expf(-vr));
    d_head[(long)b * 2 * A + j] += g;
    d_head[(long)b * 2 * A + A + j] += g * eps[i] * sig;
  }
}
extern "C" int tacorl_pr_sample_bwd(const float* head, const float* eps, const float* d_plan, float* d_head, int B,
                                    int A, float min_std, tacorl_stream_t stream) {
  const long total = (long)B * A;
  if (total <= 0) return TACORL_OK;
  hipLaunchKernelGGL(pr_sample_bwd_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, head,
                     eps, d_plan, d_head, B, A, min_std);
  return LAUNCH_OK();
}
