// Launchers of the per-image, LDS-resident convolution backward of the LMPVisionEncoder
// (encoder_bwd_fused.hip), called from the tacorl_encoder_bwd_fused composite in dense_ops.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#define EBW_MAXP 8  // problems (networks x cameras of one geometry) per call (round 5: 4 before - two cameras were two launch sequences)

struct EbwProblem {
  const void* img;   // [n][H][W][3] bf16
  const void* y1;    // [n][OH1*OW1][32] saved conv1 output (post-ReLU), bf16 (tacorl_encoder_fwd_fused's format)
  const void* y2;    // [n][OH2*OW2][64] bf16
  const float* dz3;  // [n][OH3*OW3][64] gradient of conv3's pre-activation (fp32)
  const float* w2;   // [64][4][4][32] fp32
  const float* w3;   // [64][3][3][64] fp32
  float *g_w1, *g_b1, *g_w2, *g_b2, *g_w3, *g_b3;
  int n;
  // EBW_FUSED3 only (soft-argmax backward inside the conv3 launch): conv3 output, temperature, soft-argmax output and
  // its gradient, per-workgroup scratch for the temperature-gradient partials (>= 256 floats), d(temperature)
  const float *y3, *temp, *sa, *d_sa;
  float *dtp, *g_temp;
};

bool ebw_supported(int H, int W);
size_t ebw_ws_bytes(int nprob, const int* n_img, int H, int W);
// dW1,db1,dW2,db2,dW3,db3 (+)= conv backward of every problem; deterministic (fixed reduction order).
// mode: 0 = pack the W^T fragments, then run; 1 = pack only (needs pr[].w2 / w3 / n only); 2 = run with the
// fragments already packed in ws for this step's weights.
// parts: which launches of the conv backward to issue on `st` (all of them = EBW_ALL); a caller that spreads them
// over two streams orders them with events (dgrad2 and wgrad2 read dgrad3's output, wgrad1 reads dgrad2's, the
// reduce reads every wgrad's slabs).
enum { EBW_DGRAD3 = 2, EBW_WGRAD3 = 4, EBW_DGRAD2 = 8, EBW_WGRAD2 = 16, EBW_WGRAD1 = 32, EBW_REDUCE = 64, EBW_ALL = 126,
       // with EBW_ALL: soft-argmax backward + dgrad3 + wgrad3 as ONE launch (pr[].dz3 is then not read; the caller does not
       // launch the soft-argmax backward or its temperature sum - the reduce adds the temperature gradient)
       EBW_FUSED3 = 128 };
bool ebw_fused3_supported(int H, int W);
int ebw_conv_backward(int nprob, const EbwProblem* pr, int H, int W, int accumulate, void* ws, size_t ws_bytes,
                      hipStream_t st, int mode, int parts = EBW_ALL);
