/* libtacorl_hip.so - C ABI of the MI355X (gfx950) TACO-RL offline-training hot path.
 *
 * The reference (ErickRosete/tacorl) is pure Python and has no FFI for this path
 * (SURVEY.md section 8b): every entry point below replaces a stock-torch op sequence
 * inside the reference module named in its comment.  Contract for ALL functions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch tensors);
 *     the caller keeps it alive until `stream` has been synchronised;
 *   - work is enqueued asynchronously on `stream`; nothing allocates, nothing
 *     synchronises (hipGraph-capturable); scratch comes from the caller through
 *     (ws, ws_bytes) and is sized by the matching *_ws_bytes query;
 *   - return 0 on success, negative errno-style code otherwise (never throws);
 *     tacorl_hip_last_error() gives the text for the calling thread;
 *   - "problem batches": arrays of `nprob` pointers describe independent problems of
 *     the same geometry (actor / q1 / q2 / target nets) that run in ONE launch.
 *   - fp32 row-major tensors; images NHWC; conv weights [CO][KH][KW][CI].
 */
#ifndef TACORL_HIP_H
#define TACORL_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* tacorl_stream_t; /* hipStream_t */

#define TACORL_MAXP 16 /* max problems per batched call */
enum { TACORL_F32 = 0, TACORL_BF16 = 1 };               /* storage / MFMA operand dtype */
enum { TACORL_ACT_NONE = 0, TACORL_ACT_RELU = 1, TACORL_ACT_SILU = 2 };

int tacorl_hip_version(void);
int tacorl_hip_init(int device);             /* idempotent; checks the device is gfx950 */
const char* tacorl_hip_last_error(void);
/* tracing aid: marks[slot] = device wall clock (100 MHz ticks) when the stream reaches this point */
int tacorl_time_mark(unsigned long long* marks, int slot, tacorl_stream_t stream);
/* calibration aid: a 1-thread kernel that runs for `ticks` of that clock and stores its own begin / end in marks[slot],
 * marks[slot + 1] - an event bracket around it minus that difference is the bracket's overhead for a kernel of that length */
int tacorl_time_spin(unsigned long long* marks, int slot, long ticks, tacorl_stream_t stream);

/* ---- primitives ------------------------------------------------------------------ */
/* y = act(x W^T + b), optional pre-activation z.  Replaces nn.Linear (+F.silu / nn.ReLU):
 * reference networks/actor_critic/actor.py:252-270, critic.py:92-97, goal_encoder.py:17-23. */
int tacorl_linear_fwd(int nprob, const float* const* x, int ldx, const float* const* w,
                      const float* const* b, float* const* y, float* const* z, const int* M,
                      int K, int N, int act, int compute_dtype, tacorl_stream_t stream);
/* y = act(x W^T + b + addend), y with leading dim ldy: one ReLU-RNN time step
 * (torch nn.RNN relu; reference networks/action_decoders/rnn_models.py:5-16). */
size_t tacorl_linear_add_fwd_ws_bytes(int nprob, const int* M, int K, int N);
/* ws (may be NULL): scratch for the split-reduction path used when M is too small to fill the chip. */
int tacorl_linear_add_fwd(int nprob, const float* const* x, int ldx, const float* const* w,
                          const float* const* b, const float* const* addend, int ld_add, float* const* y,
                          int ldy, const int* M, int K, int N, int act, int compute_dtype, void* ws,
                          size_t ws_bytes, tacorl_stream_t stream);
/* y = act(x W^T + b + addend) with bf16 operands resident in HBM (x [M][K], W [N][K]; K % 128 == 0,
 * N % 32 == 0): the stacked ReLU-RNN of the action decoder (reference rnn_models.py:5-16, torch nn.RNN) -
 * recurrent step (M = batch) and per-layer input projection (M = batch*T).  One launch, no split-K: a
 * 4-stage LDS-DMA ring streams K.  Writes y fp32 and, if y_bf16 != NULL, the bf16 copy that is the next
 * step's operand.  bf16 MFMA, fp32 accumulate. */
int tacorl_rnn_linear_supported(int M, int K, int N);
int tacorl_rnn_linear_fwd(const void* x_bf16, const void* w_bf16, const float* bias, const float* addend,
                          int ld_add, float* y, void* y_bf16, int M, int K, int N, int act,
                          tacorl_stream_t stream);
/* nprob <= 4 independent y = act(x W^T + b + addend) of one shape (M, K, N) in one launch, one activation per
 * problem: the wavefront schedule of the stacked RNN (launch s = recurrent step s-2l of every layer l plus
 * the input projection of step s-2l+1 of layers l >= 1). */
int tacorl_rnn_linear_fwd_batch(int nprob, const void* const* x_bf16, const void* const* w_bf16,
                                const float* const* bias, const float* const* addend, int ld_add,
                                float* const* y, void* const* y_bf16, int M, int K, int N, const int* acts,
                                tacorl_stream_t stream);
/* tacorl_rnn_linear_fwd_batch with twin rows: problem p additionally computes y2[p] = act(x2[p] W[p]^T + b[p] + addend2[p])
 * for M2 more rows held in other buffers - the same weights, read once (every problem of the launch carries twin rows).
 * PlayLMP.training_step: the logging-only random-plan pass of the action decoder (reference play_lmp_for_rl.py:243-252)
 * rides in the launches of the real pass. */
int tacorl_rnn_linear_fwd_batch_twin(int nprob, const void* const* x_bf16, const void* const* x2_bf16,
                                     const void* const* w_bf16, const float* const* bias, const float* const* addend,
                                     const float* const* addend2, int ld_add, float* const* y, float* const* y2,
                                     void* const* y_bf16, void* const* y2_bf16, int M, int M2, int K, int N,
                                     const int* acts, tacorl_stream_t stream);
/* The same launch with an optional K extension per problem (x_ext[p] != NULL; NULL arrays / entries: none):
 * y[p] = act(x[p] W[p]^T + x_ext[p] w_ext[p]^T + b[p] + bias2[p] + addend[p]), x_ext bf16 [M][128] (x2_ext [M2][128] for the twin
 * rows), w_ext bf16 [N][128], zero padded beyond the real width.  Layer 0's cell of the action decoder's ReLU-RNN as ONE
 * contraction over [h_{t-1} | x_t] - reference networks/action_decoders/rnn_models.py:5-16 (torch nn.RNN: h_t = relu(W_ih x_t +
 * b_ih + W_hh h_{t-1} + b_hh)) - instead of a separate input-projection launch + fp32 addend.  M2 = 0: no twin rows. */
int tacorl_rnn_linear_fwd_batch_ext(int nprob, const void* const* x_bf16, const void* const* x2_bf16,
                                    const void* const* w_bf16, const float* const* bias, const float* const* addend,
                                    const float* const* addend2, int ld_add, float* const* y, float* const* y2,
                                    void* const* y_bf16, void* const* y2_bf16, int M, int M2, int K, int N,
                                    const int* acts, const void* const* x_ext, const void* const* x2_ext,
                                    const void* const* w_ext, const float* const* bias2, tacorl_stream_t stream);
/* out_bf16[(t*B + b)][0..127] = bf16([plan[b] (P) | emb[(b*T + t)] (E) | zeros]) for t < Tm: the time-major RNN input rows
 * (reference action_decoder_logistic.py:279-281) as the K-extension operand above. */
int tacorl_build_ad_input_bf16(const float* plan, const float* emb, int ld_emb, void* out_bf16, int B, int T, int Tm,
                               int P, int E, tacorl_stream_t stream);
/* BPTT step of the same RNN: y = (x Wt^T + addend) * [mask_src > 0], x = dZ_t (bf16), Wt = W_hh^T (bf16,
 * tacorl_transpose_to_bf16), addend = dH_{t-1}, mask_src = h_{t-1}; y fp32 + bf16 copy (next step's x). */
int tacorl_rnn_linear_bwd_step(const void* x_bf16, const void* wt_bf16, const float* addend, int ld_add,
                               const float* mask_src, float* y, void* y_bf16, int M, int K, int N,
                               tacorl_stream_t stream);
/* Weight gradient of the RNN's square matrices from the bf16 copies the ring GEMMs leave behind (replaces, for this
 * shape, tacorl_linear_wgrad's split-R slabs + reduce; reference: autograd of nn.RNN's weight_hh / weight_ih,
 * networks/action_decoders/rnn_models.py:5-16): dw[M][N] (+)= dz^T x, db[M] (+)= column sums of dz (db may be NULL).
 * dz bf16 [R][ld_dz], x bf16 [R][ld_x]; R % 64 == 0, M % 128 == 0, N % 128 == 0. */
int tacorl_rnn_wgrad_supported(int R, int M, int N);
int tacorl_rnn_wgrad(const void* dz_bf16, int ld_dz, const void* x_bf16, int ld_x, int R, int M, int N, float* dw,
                     float* db, int accumulate, tacorl_stream_t stream);
/* n <= 4 such gradients of one output shape in ONE launch (row counts R[p] may differ; db[p] may be NULL): the weight
 * gradients of nn.RNN's square matrices - weight_hh of every layer, weight_ih of the upper ones (reference rnn_models.py:5-16,
 * autograd) - behind the BPTT. */
int tacorl_rnn_wgrad_batch(int n, const void* const* dz_bf16, int ld_dz, const void* const* x_bf16, int ld_x, const int* R,
                           int M, int N, float* const* dw, float* const* db, int accumulate, tacorl_stream_t stream);
/* The same products for a FEW output rows and many reduction rows (the action decoder's 182 x H output heads): dz bf16
 * [R][ld_dz] with Mp >= rows columns in use (Mp % 128 == 0, columns >= rows zero), x bf16 [R][ld_x]; R is cut into `slabs` row
 * ranges (R % (64 slabs) == 0) that run side by side, a second launch sums the partial results in slab order into
 * dw[rows][N] / db[rows].  ws: tacorl_rnn_wgrad_slabs_ws_bytes(slabs, Mp, N). */
size_t tacorl_rnn_wgrad_slabs_ws_bytes(int slabs, int Mp, int N);
int tacorl_rnn_wgrad_slabs(const void* dz_bf16, int ld_dz, const void* x_bf16, int ld_x, int R, int Mp, int N, int rows,
                           int slabs, float* dw, float* db, int accumulate, void* ws, size_t ws_bytes, tacorl_stream_t stream);
/* The BPTT wavefront launch: nprob <= 4 problems y[p] = (x[p] Wt[p]^T + addend[p]) * [mask_src[p] > 0] (addend / mask
 * entries may be NULL) of one shape, fp32 y + optional bf16 copy: recurrent gradient steps of all layers and the
 * projection dH_{l-1}[t] = dZ_l[t] W_ih_l (Wt = W_ih_l^T) side by side (reference: autograd of nn.RNN, rnn_models.py:5-16). */
int tacorl_rnn_linear_bwd_batch(int nprob, const void* const* x_bf16, const void* const* wt_bf16,
                                const float* const* addend, int ld_add, const float* const* mask_src,
                                float* const* y, void* const* y_bf16, int M, int K, int N, tacorl_stream_t stream);
/* dst[c][r] = bf16(src[r][c]); R, C multiples of 32. */
int tacorl_transpose_to_bf16(const float* src, void* dst, int R, int C, tacorl_stream_t stream);
/* n <= 16 such transposes (dst[j][c][r] = bf16(src[j][r][c]), R[j] x C[j], multiples of 32) in ONE launch: the weights-only
 * preparation of the plan recognition's and the action decoder's backward (W^T operands of their input-gradient GEMMs;
 * autograd of nn.Linear / nn.RNN in the reference - plan_recognition_transformer.py:70-105, rnn_models.py:5-16). */
int tacorl_transpose_to_bf16_batch(int n, const float* const* src, void* const* dst, const int* R, const int* C,
                                   tacorl_stream_t stream);
/* dst[c][r] = r < R ? bf16(src[r][c]) : 0 for r < ld_dst (C, ld_dst multiples of 32): a weight whose row count is no multiple
 * of the ring GEMM's k-step as its K-padded transposed operand (the action decoder's 182 x H output heads, reference
 * action_decoder_logistic.py:44-52, in dH = d_heads W). */
int tacorl_transpose_pad_to_bf16(const float* src, void* dst, int R, int C, int ld_dst, tacorl_stream_t stream);
/* dst[r][c] = c < cols ? bf16(src[r][c]) : 0 for c < ld_dst (ld_dst % 8 == 0): a fp32 matrix as a K-padded bf16 operand. */
int tacorl_pad_to_bf16(const float* src, int ld_src, void* dst, int ld_dst, long rows, int cols, tacorl_stream_t stream);

/* Backward primitives of y = act(x W^T + b):
 *   dgrad: out[m][i] = (sum_o dz[m][o] W[o][i] + addend[m][i]) * act'(src[m][i])
 *   wgrad: dw[o][k] (+)= sum_m dz[m][o] x[m][k], db[o] (+)= sum_m dz[m][o]  (split-R slabs in ws). */
int tacorl_linear_dgrad(int nprob, const float* const* dz, int ld_dz, const float* const* w,
                        float* const* out, int ld_out, const float* const* src, int ld_src, int act_src,
                        const float* const* addend, int ld_add, const int* M, int O, int I,
                        int compute_dtype, tacorl_stream_t stream);
/* tacorl_linear_dgrad with the reduction split over workgroups (+ a reduce pass) when the output is skinny and O long;
 * falls back to the single-pass form when a src mask is given or ws is too small (tacorl_linear_dgrad_ws_bytes). */
size_t tacorl_linear_dgrad_ws_bytes(int nprob, const int* M, int O, int I);
int tacorl_linear_dgrad_splitk(int nprob, const float* const* dz, int ld_dz, const float* const* w,
                               float* const* out, int ld_out, const float* const* src, int ld_src, int act_src,
                               const float* const* addend, int ld_add, const int* M, int O, int I,
                               int compute_dtype, void* ws, size_t ws_bytes, tacorl_stream_t stream);
size_t tacorl_linear_wgrad_ws_bytes(int nprob, const int* M, int K, int O);
int tacorl_linear_wgrad(int nprob, const float* const* x, int ldx, const float* const* dz, int ld_dz,
                        const int* M, int K, int O, float* const* dw, float* const* db, int accumulate,
                        int compute_dtype, void* ws, size_t ws_bytes, tacorl_stream_t stream);
/* y = relu(conv2d(x, w) + b), no padding.  Replaces nn.Conv2d + nn.ReLU,
 * reference networks/visual_encoders/encoder.py:369-390. x_dtype: image storage dtype. */
int tacorl_conv2d_relu_fwd(int nprob, const void* const* x, const float* const* w,
                           const float* const* b, float* const* y, const int* n_img, int H, int W,
                           int C, int KH, int KW, int S, int CO, int x_dtype, int compute_dtype,
                           tacorl_stream_t stream);

/* ---- LMPVisionEncoder (reference encoder.py:349-419, utils.py:22-76) --------------- */
/* Packed parameter block (floats, every tensor 4-float aligned):
 *   0 conv1.w[32][8][8][3] 1 conv1.b 2 conv2.w[64][4][4][32] 3 conv2.b 4 conv3.w[64][3][3][64]
 *   5 conv3.b 6 temperature[1] 7 fc1.w[256][128] 8 fc1.b 9 fc2.w[32][256] 10 fc2.b
 * Writes the 11 offsets; returns the block size in floats. */
long tacorl_encoder_param_layout(long* offsets11);
/* Saved-activation block per problem: y1 | y2 | y3 | softargmax(128) | fc1(256). Writes the 5
 * offsets (floats), returns the block size in floats. */
long tacorl_encoder_act_layout(int n_img, int H, int W, long* offsets5);
int tacorl_encoder_fwd(int nprob, const void* const* img, const float* const* params,
                       float* const* out /*[n][32]*/, float* const* act, const int* n_img, int H,
                       int W, int img_dtype, int compute_dtype, tacorl_stream_t stream);
size_t tacorl_encoder_bwd_ws_bytes(int nprob, const int* n_img, int H, int W);
/* grads: blocks in the parameter layout (overwritten, or accumulated if accumulate != 0). */
int tacorl_encoder_bwd(int nprob, const void* const* img, const float* const* params,
                       const float* const* act, const float* const* d_out, float* const* grads,
                       const int* n_img, int H, int W, int img_dtype, int compute_dtype,
                       int accumulate, void* ws, size_t ws_bytes, tacorl_stream_t stream);

/* Fused inference forward (no saved activations): one launch, bf16 MFMA, register-stationary weights,
 * LDS-resident intermediates; bf16 NHWC images only; optionally saves the activations a backward needs.  `packed` = tacorl_encoder_pack_weights output
 * (tacorl_encoder_fused_wpk_bytes() bytes per network; re-pack whenever the fp32 block changes). */
long tacorl_encoder_fused_wpk_bytes(void);
int tacorl_encoder_fused_supported(int H, int W);
/* What the fused forward writes into a problem's act block for this geometry: 0 - not a fused geometry; 1 - y1 / y2 as bf16 at
 * the start of their slots (tacorl_encoder_bwd_fused* read them); 2 - every activation as fp32, i.e. exactly what the per-layer
 * tacorl_encoder_fwd leaves and tacorl_encoder_bwd reads (a geometry of csrc/encoder_ring.hip - conv1 output too large for the
 * LDS - for which tacorl_encoder_bwd_fused_ws_bytes() is 0.  150 x 200, the un-resized rgb_static of the reference's
 * experiment=tacorl_real_world, config/.../rl_real_world_train.yaml:2-10, has the LDS-resident backward and reports 1). */
int tacorl_encoder_fused_act_format(int H, int W);
int tacorl_encoder_pack_weights(int nprob, const float* const* params, void* const* packed,
                                tacorl_stream_t stream);
/* act: NULL, or per-problem pointers (NULL entries allowed) to tacorl_encoder_act_layout blocks that
 * receive the fp32 activations tacorl_encoder_bwd needs.  Saved activations (act[p] != NULL, layout of
 * tacorl_encoder_act_layout): y3, the soft-argmax features and fc1 as fp32; y1 and y2 as **bf16** at the start of their
 * fp32-sized slots - the values the next layer consumed - which is what tacorl_encoder_bwd_fused* read (geometries of
 * tacorl_encoder_fused_act_format() == 2: all five as fp32, for tacorl_encoder_bwd). */
int tacorl_encoder_fwd_fused(int nprob, const void* const* img, const void* const* packed,
                             const float* const* params, float* const* out, float* const* act,
                             const int* n_img, int H, int W, tacorl_stream_t stream);
/* The same launch on at most max_workgroups workgroups (0 = one per CU): fewer than the CU count leaves CUs to a concurrent
 * branch of the caller's graph - TACORL issues the frozen LMP window's problems first, forks the plan recognition ->
 * action decoder branch (reference tacorl.py:142-179, 206-233: neither depends on the actor / critic encoders) and runs the
 * update's own encoder problems beside it on 160 workgroups.  The launch is shared out by WORK UNITS (an image costs 64, or 72
 * with saved activations; a workgroup may serve the tail of one problem and the head of the next): which workgroup serves
 * which image changes nothing in the results.  max_workgroups, when given, must be at least nprob (TACORL_EINVAL otherwise). */
int tacorl_encoder_fwd_fused_wg(int nprob, const void* const* img, const void* const* packed,
                                const float* const* params, float* const* out, float* const* act,
                                const int* n_img, int H, int W, int max_workgroups, tacorl_stream_t stream);

/* Backward twin of the fused forward (same preconditions: bf16 NHWC images, bf16 MFMA, a geometry
 * tacorl_encoder_fused_supported() accepts; at most 4 problems): FC tail and soft-argmax as
 * tacorl_encoder_bwd, the three convolutions' dgrad/wgrad as per-image LDS-resident kernels. Same
 * argument meaning as tacorl_encoder_bwd. Replaces autograd through
 * networks/vision/lmp_vision_network.py:32-66. */
size_t tacorl_encoder_bwd_fused_ws_bytes(int nprob, const int* n_img, int H, int W);
int tacorl_encoder_bwd_fused(int nprob, const void* const* img, const float* const* params,
                             const float* const* act, const float* const* d_out,
                             float* const* grads, const int* n_img, int H, int W, int accumulate,
                             void* ws, size_t ws_bytes, tacorl_stream_t stream);
/* The same backward in parts that share `ws`, so that a caller can keep only the dependent chain on its
 * main stream:  _pack (weight-only preparation: any time after the last optimiser step) -> _head (FC tail
 * input-gradient chain, one launch) -> _conv (soft-argmax + conv backward); _fc_wgrad (FC weight
 * gradients) may run on another stream any time after _head.  prepacked = 1: _pack already ran for the
 * current weights. */
int tacorl_encoder_bwd_fused_pack(int nprob, const float* const* params, const int* n_img, int H, int W,
                                  void* ws, size_t ws_bytes, tacorl_stream_t stream);
int tacorl_encoder_bwd_fused_head(int nprob, const float* const* params, const float* const* act,
                                  const float* const* d_out, const int* n_img, int H, int W,
                                  int prepacked, void* ws, size_t ws_bytes, tacorl_stream_t stream);
int tacorl_encoder_bwd_fused_fc_wgrad(int nprob, const float* const* act, const float* const* d_out,
                                      float* const* grads, const int* n_img, int H, int W,
                                      int accumulate, void* ws, size_t ws_bytes, tacorl_stream_t stream);
int tacorl_encoder_bwd_fused_conv(int nprob, const void* const* img, const float* const* params,
                                  const float* const* act, float* const* grads, const int* n_img, int H,
                                  int W, int accumulate, int prepacked, void* ws, size_t ws_bytes,
                                  tacorl_stream_t stream);
/* The same, a subset of its launches (parts, a bit mask: 1 soft-argmax backward, 2 dgrad3, 4 wgrad3, 8 dgrad2,
 * 16 wgrad2, 32 wgrad1, 64 slab reduce; 127 = all), so that a caller can put the weight gradients of conv3 / conv2 on a
 * second stream beside the dgrad3 -> dgrad2 -> wgrad1 chain.  The caller orders the streams: dgrad2 and wgrad2 read
 * dgrad3's output, wgrad1 reads dgrad2's, the reduce reads every wgrad's slabs.  Partial calls need prepacked = 1. */
int tacorl_encoder_bwd_fused_conv_parts(int nprob, const void* const* img, const float* const* params,
                                        const float* const* act, float* const* grads, const int* n_img, int H,
                                        int W, int accumulate, int prepacked, int parts, void* ws,
                                        size_t ws_bytes, tacorl_stream_t stream);

/* ---- MLP = chain of Linear(dims[l] -> dims[l+1]) + acts[l]  ------------------------ */
/* Parameter block: for each layer W[out][in] then b[out], each 4-float aligned. */
long tacorl_mlp_param_layout(int n_layers, const int* dims, long* w_off, long* b_off);
/* Activation block for M rows: per layer [z_l if SiLU][y_l]; y_off[n_layers-1] is the output. */
long tacorl_mlp_act_layout(int M, int n_layers, const int* dims, const int* acts, long* z_off,
                           long* y_off);
int tacorl_mlp_fwd(int nprob, const float* const* x, int ldx, const float* const* params,
                   float* const* act, const int* M, int n_layers, const int* dims,
                   const int* acts, int compute_dtype, tacorl_stream_t stream);
/* The same forward as ONE launch (bf16 MFMA only; <= 8 problems, <= 4 layers, every layer input width a
 * multiple of 8 and all widths <= 256, ldx % 4 == 0 - query with _supported).  params_bf16[p] is a bf16
 * copy of params[p] at the same element offsets (tacorl_to_bf16_batch); the caller refreshes it
 * whenever the fp32 block changes.  Saved activations are identical in layout to tacorl_mlp_fwd's. */
int tacorl_mlp_fwd_fused_supported(int nprob, int n_layers, const int* dims, int ldx);
/* lean != 0: hidden-layer outputs whose pre-activation is saved are not written; the fused weight-gradient launch
 * (same flag) recomputes them.  Only where tacorl_mlp_lean_supported() - forward, input-gradient chain and one-launch
 * weight gradients all fused - and nothing else reads those hidden outputs. */
int tacorl_mlp_lean_supported(int nprob, int n_layers, const int* dims, int ldx, int ldo, int ldd);
int tacorl_mlp_fwd_fused(int nprob, const float* const* x, int ldx, const float* const* params,
                         const void* const* params_bf16, float* const* act, const int* M,
                         int n_layers, const int* dims, const int* acts, int lean, tacorl_stream_t stream);
/* tacorl_mlp_fwd_fused with a GATHERED layer-0 input: row r of problem p is the concatenation of nseg[p] <= 4 column
 * segments; segment t supplies columns [seg_c0[4p+t], seg_c0[4p+t+1]) (the last one up to dims[0]) from
 * seg_ptr[4p+t][(seg_mod[4p+t] ? r % seg_mod[4p+t] : r) * seg_ld[4p+t] + col - seg_c0[4p+t]].  This is the reference's
 * torch.cat([enc(obs), goal_enc(enc(goal))]) (networks/actor_critic/visual_actor_wrapper.py:41-62,
 * visual_critic_wrapper.py:50-71), torch.cat([emb, action]) (critic.py:92-97) and expand_obs (utils/misc.py:132-153:
 * state rows repeated for every sampled action = seg_mod B) read where the producers left them, with no copy launch in
 * front of the MLP.  x_out != NULL and x_out[p] != NULL: the assembled fp32 rows are also written to x_out[p] ([M][ldx]),
 * layer 0's operand of the weight-gradient launch.  seg_c0 % 8 == 0, seg_ld % 4 == 0, pointers 16-byte aligned.
 * _supported == 0 (a many-row problem of a lean site, or shapes the fused forward does not take): assemble with
 * tacorl_copy_cols_batch and call tacorl_mlp_fwd_fused / tacorl_mlp_fwd. */
int tacorl_mlp_fwd_fused_gather_supported(int nprob, const int* M, int n_layers, const int* dims, const int* acts,
                                          int ldx, int lean);
int tacorl_mlp_fwd_fused_gather(int nprob, const int* nseg, const float* const* seg_ptr, const int* seg_ld,
                                const int* seg_c0, const int* seg_mod, float* const* x_out, int ldx,
                                const float* const* params, const void* const* params_bf16, float* const* act,
                                const int* M, int n_layers, const int* dims, const int* acts, int lean,
                                tacorl_stream_t stream);
/* dst[i][:] = bf16(src[i][:]) for n <= 16 buffers in one launch (count[i] % 4 == 0). */
int tacorl_to_bf16_batch(int n, const float* const* src, void* const* dst, const long* count,
                         tacorl_stream_t stream);
size_t tacorl_mlp_bwd_ws_bytes(int nprob, const int* M, int n_layers, const int* dims);
/* d_out: gradient w.r.t. the last layer's output [M][dims[n_layers]], leading dim ldo (last act
 * must be NONE).
 * grads[p] may be NULL (skip weight gradients); d_x[p] may be NULL (skip input gradient). */
int tacorl_mlp_bwd(int nprob, const float* const* x, int ldx, const float* const* params,
                   const float* const* act, const float* const* d_out, int ldo, float* const* grads,
                   float* const* d_x, int ldd, const int* M, int n_layers, const int* dims,
                   const int* acts, int compute_dtype, int accumulate, void* ws, size_t ws_bytes,
                   tacorl_stream_t stream);
/* bf16 mode, <= 8 problems, <= 4 layers, widths <= 256: the backward split in two calls that share `ws`.
 *   _dgrad: the whole input-gradient chain dZ_{l-1} = (dZ_l W_l) * act'_{l-1} in ONE launch (plus a weight
 *           transpose launch); leaves every dZ_l in ws and writes d_x[p] (NULL = not wanted).
 *   _wgrad: weight / bias gradients from those dZ_l (grads[p] NULL = skip the problem); may run on
 *           another stream once _dgrad has finished - it is not on the dependent chain of the step.
 * Together they equal tacorl_mlp_bwd (autograd through the Linear/SiLU/ReLU stacks of
 * networks/actor_critic/{actor,critic}.py and visual_encoders/goal_encoder.py). */
int tacorl_mlp_bwd_fused_supported(int nprob, int n_layers, const int* dims, int ldo, int ldd);
size_t tacorl_mlp_bwd_fused_ws_bytes(int nprob, const int* M, int n_layers, const int* dims);
/* prepacked: bit 0 = weight transposes already in ws (tacorl_mlp_bwd_fused_pack); bit 1 = this MLP site runs lean
 * (tacorl_mlp_fwd_fused and _wgrad with lean != 0): at >= 16 384 rows (C5's Q networks: 99 328) the forward then also
 * leaves bf16 copies of the hidden outputs, _dgrad leaves its dZ_l as bf16, and _wgrad streams both by LDS-DMA
 * (mlp_wgrad_big_kernel) - the three calls must agree on the flag. */
int tacorl_mlp_bwd_fused_dgrad(int nprob, const float* const* params, const float* const* act,
                               const float* const* d_out, int ldo, float* const* d_x, int ldd,
                               const int* M, int n_layers, const int* dims, const int* acts,
                               int prepacked, void* ws, size_t ws_bytes, tacorl_stream_t stream);
/* The weight transposes _dgrad needs, alone (they depend only on the parameters: run them early / on
 * another stream, then pass prepacked = 1 to _dgrad). */
int tacorl_mlp_bwd_fused_pack(int nprob, const float* const* params, const int* M, int n_layers,
                              const int* dims, void* ws, size_t ws_bytes, tacorl_stream_t stream);
/* Weight-only preparation of a step as ONE launch.  Between _begin() and _end(stream) - same host thread - the entry
 * points whose launches read nothing but parameters (tacorl_to_bf16_batch, tacorl_mlp_bwd_fused_pack and the FC-tail
 * transposes of tacorl_encoder_bwd_fused_pack) only record their jobs; _end issues them together on `stream`, which must
 * be the stream those calls were given.  Replaces, in the reference's terms, nothing: torch.nn.Linear's backward
 * (networks/actors/actor.py:28-60, critics/critic.py:92-97 via autograd) has no such step - the transposed bf16 copies
 * are this implementation's MFMA operands.  _begin inside an open batch and _end without one return TACORL_EINVAL. */
int tacorl_prep_batch_begin(void);
/* The same for the slab reduces that finish the one-launch MLP weight gradients (tacorl_mlp_bwd_fused_wgrad,
 * tacorl_encoder_bwd_fused_fc_wgrad): between tacorl_reduce_batch_begin() and _end(stream) they are recorded, and _end
 * sums every site's slabs in one launch - the gradients are complete only after _end.  Their only readers are the
 * gradient all-reduce and the optimiser (reference cql_offline_lightning.py:519-542: the three optimizer.step() calls). */
int tacorl_reduce_batch_begin(void);
int tacorl_reduce_batch_end(tacorl_stream_t stream);
int tacorl_prep_batch_end(tacorl_stream_t stream);
int tacorl_mlp_bwd_fused_wgrad(int nprob, const float* const* x, int ldx, const float* const* act,
                               const float* const* d_out, int ldo, float* const* grads, const int* M,
                               int n_layers, const int* dims, const int* acts, int accumulate, int lean,
                               void* ws, size_t ws_bytes, tacorl_stream_t stream);

/* ---- data movement ----------------------------------------------------------------- */
/* n images (image i at src + i*img_pitch floats; NCHW if src_nchw else NHWC) -> contiguous NHWC
 * dst (fp32 / bf16).  Replaces the states[:,0] / states[:,-1] / goal gathers of
 * TACORL.get_rl_batch (reference modules/tacorl/tacorl.py:142-179). */
int tacorl_pack_images(const float* src, long img_pitch, int src_nchw, void* dst, int dst_dtype,
                       int n, int C, int H, int W, tacorl_stream_t stream);
/* Up to 8 NCHW (C = 3, H*W % 4 == 0, 16-byte aligned) -> NHWC packs in one launch: the B*T window frames
 * and the obs / goal / next-obs images of a TACORL step (reference get_rl_batch, tacorl.py:142-179). */
int tacorl_pack_images_batch(int njobs, const float* const* src, const long* img_pitch, void* const* dst,
                             const int* n_img, int dst_dtype, int H, int W, tacorl_stream_t stream);
/* The same with second destinations for WINDOW jobs (window[j] = T >= 2, n_img[j] % T == 0): image i = b T + t of job j is
 * also written to image b of dst_first[j] (t == 0) or of dst_last[j] (t == T - 1).  TACO-RL's transition (s, s') is the
 * window's first and last frame (reference modules/tacorl/tacorl.py:147-168, get_rl_batch), so the step's obs / next
 * images are packed from the one read of the window instead of being read again: 2 B of 19 B fp32 frames per step.
 * window == NULL or window[j] == 0: a plain job. */
int tacorl_pack_images_window_batch(int njobs, const float* const* src, const long* img_pitch, void* const* dst,
                                    const int* n_img, void* const* dst_first, void* const* dst_last,
                                    const int* window, int dst_dtype, int H, int W, tacorl_stream_t stream);
/* The dataset's own frame format: uint8 HWC (reference datamodule/dataset/play_dataset.py) -> NHWC fp32 / bf16 with the
 * reference's transform pipeline applied on the way, ToTensor (x / 255) and Normalize(0.5, 0.5) ((t - 0.5) / 0.5) in
 * fp32 (config .../rl_train.yaml:12-14): bit-identical to packing the transformed fp32 frames, a quarter of the
 * bytes read.  Jobs as in tacorl_pack_images_batch; pitches in bytes; H*W*3 % 16 == 0, 16-byte aligned. */
int tacorl_pack_images_u8_batch(int njobs, const void* const* src, const long* img_pitch_bytes, void* const* dst,
                                const int* n_img, int dst_dtype, int H, int W, tacorl_stream_t stream);
/* The same pack reading its images BY FRAME INDEX out of a uint8 dataset resident in device (or mapped host) memory:
 * image i of job j is frame index[j][i * index_stride[j]] of the dataset at src[j] (pitch = bytes per frame); index[j]
 * NULL = image i (the plain pack).  One pass from the replay store into the encoder's image buffers - the window
 * frames (stride 1) and the obs / goal / next images (strides T, 1, T over the same id table) of a TACORL step -
 * instead of a gather into a uint8 batch followed by the pack (reference datamodule/dataset/play_dataset.py:115-169,
 * 357-419 assembles the window on the host).  Ids are NOT range-checked here: tacorl_amd/data/replay.py checks them. */
int tacorl_pack_images_u8_gather_batch(int njobs, const void* const* src, const long* img_pitch_bytes,
                                       const long* const* index, const int* index_stride, void* const* dst,
                                       const int* n_img, int dst_dtype, int H, int W, tacorl_stream_t stream);
/* Replay data path on the GPU.
 * tacorl_gather_frames_u8: dst[i] = frames[index[i]] - a step's window / goal frames out of the uint8 HWC dataset
 * resident in HBM (reference datamodule/dataset/play_dataset.py:357-419 loads them from per-frame .npz files);
 * frame_bytes % 16 == 0, index: device int64[n].
 * tacorl_pack_images_u8_aug_batch: the reference's train-time image pipeline on the way into the NHWC image
 * buffers (config/datamodule/transform_manager/transforms/rl_train.yaml): RandomShiftsAug (utils/transforms.py:
 * 265-299; shift[i] = (sx, sy) in [0, 2*pad], the reference's randint draw - an integer shift of the replicate-padded
 * frame), x/255, torchvision ColorJitter (utils/transforms.py:302-330; jitter[i] = {brightness factor, contrast
 * factor, hue shift, the 4 drawn positions of fn_idx (0 brightness, 1 contrast, 2 saturation = unused, 3 hue),
 * apply flag}), Normalize(0.5, 0.5).  shift / jitter: device tables per job, NULL (array or entry) = that stage off. */
int tacorl_gather_frames_u8(const void* frames, long frame_bytes, const long* index, void* dst, long n,
                            tacorl_stream_t stream);
int tacorl_pack_images_u8_aug_batch(int njobs, const void* const* src, const long* img_pitch_bytes,
                                    void* const* dst, const int* const* shift, const float* const* jitter,
                                    const int* n_img, int dst_dtype, int H, int W, int pad,
                                    tacorl_stream_t stream);
/* The whole train pipeline of rl_train.yaml including its first stage, torchvision.transforms.Resize (:3-4,16-17: the
 * 200x200 camera frames -> 128x128 static / 84x84 gripper): source frames src_H x src_W uint8 HWC (pitch = bytes per
 * source frame), bilinear with align_corners = False and no antialias on the 0..255 values - what torchvision's
 * tensor path computes through torch.nn.functional.interpolate - then RandomShiftsAug on the H x W result, x/255,
 * ColorJitter, Normalize.  src_H == H and src_W == W: no resize (tacorl_pack_images_u8_aug_gather_batch). */
int tacorl_pack_images_u8_resize_aug_gather_batch(int njobs, const void* const* src, const long* img_pitch_bytes,
                                                  const long* const* index, const int* index_stride, void* const* dst,
                                                  const int* const* shift, const float* const* jitter, const int* n_img,
                                                  int dst_dtype, int src_H, int src_W, int H, int W, int pad,
                                                  tacorl_stream_t stream);
/* ... and the augmenting pack with the same optional frame-index tables (shift / jitter tables are per IMAGE i). */
int tacorl_pack_images_u8_aug_gather_batch(int njobs, const void* const* src, const long* img_pitch_bytes,
                                           const long* const* index, const int* index_stride, void* const* dst,
                                           const int* const* shift, const float* const* jitter, const int* n_img,
                                           int dst_dtype, int H, int W, int pad, tacorl_stream_t stream);
/* reward = done = float(disp == 1) (done may be NULL) and acts_dst[0:n_acts] = acts_src[0:n_acts], one launch:
 * the small tensors of TACORL.get_rl_batch (reference modules/tacorl/tacorl.py:142-179).
 * disp_dtype: 0 float32, 1 int64, 2 int32, 3 uint8 / bool. */
int tacorl_stage_transition(const void* disp, int disp_dtype, float* reward, float* done, int B,
                            const float* acts_src, float* acts_dst, long n_acts, tacorl_stream_t stream);
/* dst[r][0:cols] (+)= src[r % src_row_mod][0:cols]  (src_row_mod <= 0: r).  expand_obs on
 * embeddings instead of images (reference utils/misc.py:132-153) and torch.cat plumbing. */
int tacorl_copy_cols(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols,
                     int src_row_mod, int accumulate, tacorl_stream_t stream);
/* n <= 32 mutually independent tacorl_copy_cols in one launch (no copy may read what another writes). */
int tacorl_copy_cols_batch(int n, const float* const* src, const int* ld_src, float* const* dst,
                           const int* ld_dst, const int* rows, const int* cols,
                           const int* src_row_mod, const int* accumulate, tacorl_stream_t stream);
/* out[b][c] = sum_{j<reps} in[j*B+b][c] : gradient of that broadcast. */
int tacorl_reduce_rows_mod(const float* in, int ld_in, float* out, int ld_out, int B, int cols,
                           int reps, tacorl_stream_t stream);
/* n <= 4 such reductions of one shape in one launch. */
int tacorl_reduce_rows_mod_batch(int n, const float* const* in, int ld_in, float* const* out, int ld_out, int B,
                                 int cols, int reps, tacorl_stream_t stream);
/* dst[r][0:A] = 2u-1 (last dim snapped to +-1 for a discrete gripper);
 * reference modules/cql/cql_offline_lightning.py:243-250. */
int tacorl_uniform_actions(const float* u01, float* dst, int ld_dst, int rows, int A,
                           int discrete_gripper, tacorl_stream_t stream);

/* ---- tanh-Gaussian policy head (reference actor.py:65-156, utils/distributions.py:61-153) --- */
/* head[m] = [mean_raw(Ac) | log_std_raw(Ac) | gripper logits(2)?].  For k<n, m<M:
 * a = tanh(mu + sd*eps[k][m]) -> act_out[(k*M+m)*ld_act ..], log pi -> logp[k*M+m].
 * gumbel_u (n,M,2) U(0,1) or NULL: discrete gripper (hard_rsample: the rsample(hard=True) index). */
int tacorl_tanh_normal_sample(const float* head, int ld_head, const float* eps, const float* gumbel_u,
                              int hard_rsample, float* act_out, int ld_act, float* logp, int* grip_idx,
                              int n, int M, int Ac, tacorl_stream_t stream);
/* njobs <= 6 tacorl_tanh_normal_sample calls sharing (M, Ac, ld_head, ld_act) in one launch, optionally with
 * the tacorl_uniform_actions(u01, u_dst, ld_act, u_rows, A, u_discrete) of the step (u01 NULL = none). */
int tacorl_tanh_normal_sample_batch(int njobs, const float* const* head, int ld_head,
                                    const float* const* eps, const float* const* gumbel_u,
                                    const int* hard_rsample, float* const* act_out, int ld_act,
                                    float* const* logp, int* const* grip_idx, const int* n, int M, int Ac,
                                    const float* u01, float* u_dst, int u_rows, int A, int u_discrete,
                                    tacorl_stream_t stream);

/* ---- plan-recognition transformer glue (reference plan_recognition_transformer.py:70-105) ---- */
/* out[r] = [x[r] (D, zero-padded to Dp)] + add[r % T]  (position embeddings). */
int tacorl_add_rows_bcast(const float* x, int ldx, const float* add, float* out, int R, int T, int D,
                          int Dp, tacorl_stream_t stream);
/* softmax(q k^T / sqrt(hd)) v per (batch, head); qkv [B*T][3D], out [B*T][D]; T <= 64, hd <= 16. */
int tacorl_attention_fwd(const float* qkv, float* out, int B, int T, int D, int H, tacorl_stream_t stream);
/* Train mode of the same (reference plan_recognition_transformer.py:49-54: nn.TransformerEncoderLayer(dropout=p);
 * nn.MultiheadAttention drops attention probabilities): keep = uint8 [B][H][T][T] keep flags (an explicit input:
 * the caller draws them), kept probabilities are scaled by keep_scale = 1/(1-p). */
int tacorl_attention_dropout_fwd(const float* qkv, float* out, const unsigned char* keep, float keep_scale, int B,
                                 int T, int D, int H, tacorl_stream_t stream);
int tacorl_attention_dropout_bwd(const float* qkv, const float* d_out, float* d_qkv, const unsigned char* keep,
                                 float keep_scale, int B, int T, int D, int H, tacorl_stream_t stream);
/* nn.Dropout in train mode with an explicit keep mask, in place: x[i] = keep[i] ? x[i] * keep_scale : 0
 * (an activation in the forward, its gradient in the backward; :60,87 and the encoder layers' dropout1/2/ffn). */
int tacorl_dropout_mul(float* x, const unsigned char* keep, float keep_scale, long n, tacorl_stream_t stream);
/* y = LayerNorm(x + res); stats[r] = {mean, rstd} (may be NULL). D <= 256. */
int tacorl_add_layernorm_fwd(const float* x, const float* res, const float* w, const float* b, float* y,
                             float* stats, int R, int D, float eps, tacorl_stream_t stream);
/* Frozen / eval plan-recognition encoder in ONE launch: position embedding + L post-norm transformer
 * encoder layers (ReLU FFN) + mean over time -> pooled [B][D]; d_model 32 or 64 (one or two cameras), T 16 or 32, 8 heads,
 * FF % 256 == 0, L <= 4 (reference plan_recognition_transformer.py:36-88).  One workgroup per sequence, bf16 MFMA, nothing
 * saved for a backward.  offsets: [position_embeddings, then per layer in_proj_weight, in_proj_bias,
 * out_proj.weight, out_proj.bias, linear1.weight, linear1.bias, linear2.weight, linear2.bias, norm1.weight,
 * norm1.bias, norm2.weight, norm2.bias] - element offsets into params and its bf16 copy params_bf16. */
int tacorl_pr_encoder_fused_supported(int D, int T, int H, int FF, int L);
/* the train-mode launches (tacorl_pr_encoder_fused_train / _train_sample) exist for d_model 32 (T 16 or 32); the one-launch
 * backward for T 16 */
int tacorl_pr_encoder_fused_train_supported(int D, int T, int H, int FF, int L);
int tacorl_pr_encoder_fused(const float* emb, int ld_emb, const float* params, const void* params_bf16,
                            const long* offsets, float* pooled, int B, int D, int T, int H, int FF, int L,
                            tacorl_stream_t stream);
/* The same launch extended by the posterior head and the plan sample: head = Wc pooled + bc with (Wc, bc) the
 * composition of the two bias-only Linear layers after the pooling (fc, mean_fc: no activation between them;
 * tacorl_pr_head_compose, weights only, off the dependent chain), std = softplus(var_raw) + min_std,
 * plan = tanh(mean + eps * std)  (plan_recognition_transformer.py:89-104).  2A <= 64. */
int tacorl_pr_encoder_fused_sample(const float* emb, int ld_emb, const float* params, const void* params_bf16,
                                   const long* offsets, float* pooled, int B, int D, int T, int H, int FF, int L,
                                   const float* Wc, const float* bc, const float* eps, float* head, float* plan,
                                   int A, float min_std, tacorl_stream_t stream);
/* Train mode of the same launch (PlayLMP.training_step with plan-recognition dropout 0): additionally writes, per layer, the
 * tensors the per-op backward reads, in the per-op forward's layouts (fp32, batch-major rows).  save[9 l + k], k = 0..8: layer
 * input [B T][32], q|k|v [B T][96], attention output [B T][32], out-projection [B T][32], LayerNorm-1 output [B T][32],
 * post-ReLU FFN hidden [B T][FF], FFN output [B T][32], LayerNorm-1 {mean, rstd} [B T][2], LayerNorm-2 {mean, rstd} [B T][2]. */
int tacorl_pr_encoder_fused_train(const float* emb, int ld_emb, const float* params, const void* params_bf16,
                                  const long* offsets, float* pooled, int B, int D, int T, int H, int FF, int L,
                                  float* const* save, tacorl_stream_t stream);
/* Train-mode backward of the same encoder: the input-gradient chain of all layers in ONE launch (+ one small reduce), from
 * d_pool [B][32] (gradient of the time-pooled output) to dx [B T][32] (gradient of the position-embedded input).
 * saved[9 l + k]: what tacorl_pr_encoder_fused_train wrote.  dz[4 l + k], k = 0..3, outputs - the dZ operands of the per-op
 * weight-gradient GEMMs: LayerNorm-2 input gradient [B T][32] (linear2), masked hidden gradient [B T][FF] (linear1),
 * LayerNorm-1 input gradient [B T][32] (out_proj), d(q|k|v) [B T][96] (in_proj).  wt[4 l + {0, 1, 2, 3}]: linear1.weight^T
 * [32][FF], linear2.weight^T [FF][32], out_proj.weight^T [32][32], in_proj_weight^T [32][96] as bf16
 * (tacorl_transpose_to_bf16; the last two may be NULL: gathered from the fp32 block inside the launch).  ln_part: scratch of L * 2 * B * 64 floats;
 * ln_grads[4 l + k]: norm1.weight, norm1.bias, norm2.weight, norm2.bias gradients (written, not accumulated).
 * d_pool == NULL: d_pool = d_head Wc computed in the launch (d_head [B][A2], Wc [A2][32] from tacorl_pr_head_compose).
 * Reference: autograd through plan_recognition_transformer.py:70-88 (nn.TransformerEncoderLayer, post-norm, ReLU). */
int tacorl_pr_encoder_bwd_fused(const float* params, const long* offsets, const float* d_pool, const float* d_head,
                                const float* Wc, int A2, float* dx, const float* const* saved, float* const* dz,
                                const void* const* wt, float* ln_part, float* const* ln_grads, int B, int D, int T, int H,
                                int FF, int L, tacorl_stream_t stream);
/* tacorl_pr_encoder_fused_train with the posterior head and the plan sample in the launch (head = Wc pooled + bc, composed
 * by tacorl_pr_head_compose; plan = tanh(mean + eps * std)); its backward: tacorl_pr_encoder_bwd_fused with d_pool == NULL. */
int tacorl_pr_encoder_fused_train_sample(const float* emb, int ld_emb, const float* params, const void* params_bf16,
                                         const long* offsets, float* pooled, int B, int D, int T, int H, int FF, int L,
                                         float* const* save, const float* Wc, const float* bc, const float* eps,
                                         float* head, float* plan, int A, float min_std, tacorl_stream_t stream);
int tacorl_pr_head_compose(const float* w_fc, const float* b_fc, const float* w_head, const float* b_head,
                           float* Wc, float* bc, int D, int FC, int A2, tacorl_stream_t stream);
int tacorl_mean_over_t(const float* x, float* out, int B, int T, int D, tacorl_stream_t stream);
/* head [B][2A] = [mean | var_raw]; std = softplus(var_raw)+min_std; plan = tanh(mean + eps*std). */
int tacorl_pr_sample(const float* head, const float* eps, float* plan, float* mu_out, float* std_out,
                     int B, int A, float min_std, tacorl_stream_t stream);

/* ---- action decoder (reference networks/action_decoders/action_decoder_logistic.py) ---------- */
/* Time-major RNN input x[(t*B+b)] = [plan[b] | emb[b*T+t]], t < Tm (:279-281). */
int tacorl_build_ad_input(const float* plan, const float* emb, int ld_emb, float* out, int B, int T,
                          int Tm, int P, int E, tacorl_stream_t stream);
/* bf16 compute: the RNN's layer-0 input projection straight from (plan, frame embeddings), without x_seq:
 * out[t*B + b][n] = b_ih[n] + sum_k bf16(x[k]) bf16(w_ih[n][k]), x = [plan[b] (P) | emb[b*T + t] (E)], P + E <= 64, H % 16 == 0
 * (reference action_decoder_logistic.py:279-281 + nn.RNN's weight_ih_l0). */
int tacorl_ad_input_proj(const float* plan, const float* emb, int ld_emb, const float* w_ih, const float* b_ih,
                         float* out, int B, int T, int Tm, int P, int E, int H, tacorl_stream_t stream);
size_t tacorl_logistic_mixture_ws_bytes(int B, int Tm, int Da);
/* Discretised-logistic-mixture NLL + gripper CE (:110-235), forward + backward fused.
 * heads[(t*B+b)] = [means Da*K | log_scales Da*K | logit_probs Da*K | gripper 2]; actions batch-major
 * [B][T][Da+1]; d_heads may be NULL (loss only); loss_out: two device floats {loss, gripper accuracy}. */
int tacorl_logistic_mixture_loss(const float* heads, int ldh, const float* actions, float* d_heads,
                                 float* loss_out, int B, int T, int Tm, int Da, int K, int num_classes,
                                 float gripper_alpha, float grad_scale, void* ws, size_t ws_bytes,
                                 tacorl_stream_t stream);
/* loss_out may be NULL: the per-block partial sums then stay in ws, and this call - any time before the next
 * tacorl_logistic_mixture_loss on the same ws, same (B, Tm, Da) - writes {loss, gripper accuracy} (a loss that is only
 * logged, reference tacorl.py:262-270, need not cost a launch per step). */
int tacorl_logistic_mixture_finish(const void* ws, size_t ws_bytes, int B, int Tm, int Da, float* loss_out,
                                   tacorl_stream_t stream);
/* ActionDecoderLogistic._sample (:238-266), the action of `act` at rollout time: Gumbel-max mixture component,
 * inversion sampling of the chosen logistic, argmax gripper class.  rand_a [R][Da][K], rand_b [R][Da]: the
 * caller's U(0,1) draws in the heads' row order; out [R][Da+1] = [actions | gripper -1/+1]. */
int tacorl_logistic_mixture_sample(const float* heads, int ldh, const float* rand_a, const float* rand_b,
                                   float* out, int R, int Da, int K, tacorl_stream_t stream);

/* ---- backward glue: ReLU-RNN BPTT, transformer, seq-VAE KL ---------------------------------- */
int tacorl_relu_mask_mul(const float* dy, const float* add, const float* h, float* out, long n,
                         tacorl_stream_t stream);
/* accumulate: 0 = overwrite rows t < Tm, 1 = add to them, 2 = overwrite them and zero the rows Tm <= t < T */
int tacorl_ad_input_bwd(const float* dx, float* d_plan, float* d_emb, int ld_emb, int B, int T, int Tm,
                        int P, int E, int accumulate, tacorl_stream_t stream);
/* PlayLMP.training_step (reference play_lmp_for_rl.py:200-257): the gradient entering the encoders in one launch.
 * d_emb [(b T + t)][Ec] (holding the action decoder's share) += dx[..][0..D_in) (plan recognition) + dS[b][0..Ec) on t = 0
 * (plan proposal, state) + dgin[b][0..Ec) on t = T - 1 (plan proposal, goal), in that order; camera j's 32 columns also go
 * to f_dout[j] [(b T + t)][32] (NULL: skipped).  Ec = 32 ncam. */
int tacorl_plmp_demb_finish(float* d_emb, const float* dx, int ld_dx, int D_in, const float* dS, int ld_ds,
                            const float* dgin, float* const* f_dout, int ncam, int B, int T, int Ec,
                            tacorl_stream_t stream);
int tacorl_bcast_over_t(const float* src, float* dst, int B, int T, int D, float scale, int accumulate,
                        tacorl_stream_t stream);
int tacorl_attention_bwd(const float* qkv, const float* d_out, float* d_qkv, int B, int T, int D, int H,
                         tacorl_stream_t stream);
size_t tacorl_add_layernorm_bwd_ws_bytes(int R, int D);
int tacorl_add_layernorm_bwd(const float* dy, const float* x, const float* res, const float* w,
                             const float* stats, float* dv, float* dw, float* db, int R, int D,
                             int accumulate, void* ws, size_t ws_bytes, tacorl_stream_t stream);
/* Balanced KL(q||p) of PlayLMP.compute_kl_loss (reference play_lmp_for_rl.py:259-301), fwd+bwd:
 * out2 = {kl, kl_beta*kl}; d_head_* = d(kl_beta*kl)/d(raw heads) * grad_scale. */
int tacorl_gauss_kl_balanced(const float* head_q, const float* head_p, float* d_head_q, float* d_head_p,
                             int B, int A, float kl_alpha, float kl_beta, float min_std, int balanced,
                             float grad_scale, float* out2, tacorl_stream_t stream);
int tacorl_pr_sample_bwd(const float* head, const float* eps, const float* d_plan, float* d_head, int B,
                         int A, float min_std, tacorl_stream_t stream);

/* device-resident metric record written by the loss kernels (names = reference self.log keys) */
enum {
  TACORL_LG_ALPHA_LOSS = 0, TACORL_LG_ALPHA, TACORL_LG_ACTOR_LOSS, TACORL_LG_BELL1, TACORL_LG_BELL2,
  TACORL_LG_CONS1, TACORL_LG_CONS2, TACORL_LG_Q1LOSS, TACORL_LG_Q2LOSS, TACORL_LG_ALPHA_P,
  TACORL_LG_ALPHA_P_LOSS, TACORL_LG_Q1_DATA, TACORL_LG_Q1_RAND, TACORL_LG_Q1_POL, TACORL_LG_Q2_DATA,
  TACORL_LG_Q2_RAND, TACORL_LG_Q2_POL, TACORL_LG_ACTION_LOSS, TACORL_LG_COUNT
};
/* alpha_loss and d/dlog_alpha (cql_offline_lightning.py:447-449); grad scaled by grad_scale. */
int tacorl_alpha_loss(const float* logp, int B, const float* log_alpha, float target_entropy,
                      float grad_scale, float* g_log_alpha, float* logs, tacorl_stream_t stream);
/* One GPU: the same loss / gradient and Adam's step on log_alpha in ONE launch (adam arithmetic of tacorl_adam_step for
 * n = 1 without clipping; the logged loss uses the pre-step log_alpha, reference cql_offline_lightning.py:447-455). */
int tacorl_alpha_loss_step(const float* logp, int B, float* log_alpha, float target_entropy, float* g_log_alpha,
                           float* logs, float* m, float* v, float lr, int* step_counter, tacorl_stream_t stream);
/* Q phase actor loss mean(alpha*logpi - min(q1,q2)) and dL/dq_i (:463-466). */
int tacorl_actor_qmin(const float* q1, const float* q2, const float* logp, int B, const float* log_alpha,
                      float* dq1, float* dq2, float grad_scale, float* logs, tacorl_stream_t stream);
/* dL_actor/d head.  g_act1/2: dL/da from the critics (Q phase) or NULL; value: dataset action for
 * the BC phase (:459-461) or NULL. */
int tacorl_actor_head_bwd(const float* head, int ld_head, const float* eps, const float* logp,
                          const float* g_act1, const float* g_act2, int ld_g, const float* value,
                          int ld_value, const int* grip_idx, const float* log_alpha, float grad_scale,
                          float* d_head, int B, int Ac, int has_grip, float* logs, tacorl_stream_t stream);

/* ---- Bellman + CQL logsumexp (+ Lagrange), forward and backward fused (:284-406) -------------- */
size_t tacorl_cql_ws_bytes(int B);
/* q_i / dq_i: rows [data B | random nB | current-policy nB | next-policy nB], sample-major (k*B+b). */
int tacorl_cql_loss(const float* q1, const float* q2, float* dq1, float* dq2, const float* tq1,
                    const float* tq2, const float* logp_cur, const float* logp_nxt, const float* next_logp,
                    const float* reward, const float* done, const float* log_alpha,
                    const float* log_alpha_prime, int B, int n, int A, float discount, float reward_scale,
                    float temp, float cons_w, float gap, int deterministic_backup, float grad_scale,
                    float* g_log_alpha_prime, float* logs, void* ws, size_t ws_bytes, tacorl_stream_t stream);

/* ---- clip_grad_norm_ + Adam + Polyak over one flat block (:229-232, 519-542, 553-574) --------- */
size_t tacorl_adam_ws_bytes(long n);
/* max_norm <= 0: no clipping; target NULL: no soft update; step_counter: device int (t-1). */
int tacorl_adam_step(float* param, const float* grad, float* m, float* v, long n, float lr,
                     float max_norm, int* step_counter, float* target, float tau, void* ws,
                     size_t ws_bytes, tacorl_stream_t stream);
/* nb <= 8 parameter blocks (each with its own lr / max_norm / step counter / optional Polyak target) in
 * two launches (norms + step counters, updates); results bit-identical to nb tacorl_adam_step calls. */
size_t tacorl_adam_batch_ws_bytes(int nb);
int tacorl_adam_step_batch(int nb, float* const* param, const float* const* grad, float* const* m,
                           float* const* v, const long* n, const float* lr, const float* max_norm,
                           int* const* step_counter, float* const* target, const float* tau, void* ws,
                           size_t ws_bytes, tacorl_stream_t stream);
/* The same update; mirror[b] / target_mirror[b] (either array or any entry may be NULL): bf16 copies of the UPDATED
 * parameter block / Polyak target at the same element offsets, written by the update launch itself - the fused MLP
 * kernels' MFMA operand (params_bf16 of tacorl_mlp_fwd_fused) is then current when the step ends and the next step needs
 * no tacorl_to_bf16_batch launch at the head of its dependent chain. */
int tacorl_adam_step_batch_mirror(int nb, float* const* param, const float* const* grad, float* const* m,
                                  float* const* v, const long* n, const float* lr, const float* max_norm,
                                  int* const* step_counter, float* const* target, const float* tau,
                                  void* const* mirror, void* const* target_mirror, void* ws, size_t ws_bytes,
                                  tacorl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
