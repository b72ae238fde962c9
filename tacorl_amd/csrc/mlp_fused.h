// Single-launch MLP forward (mlp_fused.hip), used by tacorl_mlp_fwd when the shapes qualify.
#pragma once
#include <hip/hip_runtime.h>

#define MF_MAXP 8  // problems per launch
#define MF_MAXL 4  // layers
#define MF_MAXSEG 4  // column segments of a gathered layer-0 input

// Gathered layer-0 input of the fused forward (nseg[p] == 0: plain x[p]).  Segment t of problem p supplies columns
// [c0[p][t], c0[p][t+1]) (the last one up to dims[0]) of row r from ptr[p][t][(mod ? r % mod : r) * ld + col - c0];
// c0 % 8 == 0, ld % 4 == 0, pointers 16-byte aligned.  xw[p] != NULL: the assembled fp32 rows are also written there
// ([M][ldx]) for the weight-gradient launch.
struct MlpXGather {
  const float* ptr[MF_MAXP][MF_MAXSEG];
  int ld[MF_MAXP][MF_MAXSEG], c0[MF_MAXP][MF_MAXSEG], mod[MF_MAXP][MF_MAXSEG];
  int nseg[MF_MAXP];
  float* xw[MF_MAXP];
};

// bf16 compute only: hidden widths multiples of 8, every width <= 256 (the input width may be ragged, e.g. 64 + 7).
bool mlp_fused_fwd_ok(int nprob, int L, const int* dims, int ldx);
// zoff / yoff: [nprob][MF_MAXL] offsets (floats) into act[p] (tacorl_mlp_act_layout; zoff < 0 = not saved);
// woff / boff: [L] offsets into params[p] (tacorl_mlp_param_layout).
// params_bf16[p]: bf16 copy of params[p] (same element offsets, 8-byte aligned): the MFMA B operand.
int mlp_fused_fwd(int nprob, const float* const* x, int ldx, const float* const* params, const void* const* params_bf16,
                  float* const* act,
                  const int* M, int L, const int* dims, const int* acts, const long* zoff, const long* yoff,
                  const long* woff, const long* boff, hipStream_t st, const MlpXGather* gather = nullptr,
                  const int* gmap = nullptr);  // gmap[p]: the gather's problem index of this launch's problem p

// Backward input-gradient chain in one launch (+ one weight-transpose pack launch): writes dZ_l (fp32,
// [M][dims[l+1]]) for l = 0..L-2 at dz[p] + dzoff[p*MF_MAXL + l] and, if d_x[p] != NULL, the input gradient.
// The last layer's dZ is d_out itself.  srcoff[p*MF_MAXL + l] = offset in act[p] of the activation
// derivative source of layer l's output (pre-activation for SiLU, output for ReLU), -1 for identity.
// wt[p]: scratch of mlp_fused_wt_elems() bf16 elements (16-byte aligned).
bool mlp_fused_bwd_ok(int nprob, int L, const int* dims, int ldo, int ldd);
size_t mlp_fused_wt_elems(int L, const int* dims, long* wtoff);
int mlp_fused_bwd(int nprob, const float* const* params, const float* const* act, const float* const* d_out, int ldo,
                  float* const* dz, float* const* d_x, int ldd, void* const* wt, const int* M, int L, const int* dims,
                  const int* acts, const long* srcoff, const long* dzoff, const long* woff, hipStream_t st, int mode);
// Deferred weight-only preparation (tacorl_prep_batch_begin / _end, mlp_fused.hip): while deferring, mlp_fused_bwd(mode 1)
// and tacorl_to_bf16_batch record their jobs instead of launching them.
bool prep_deferring();
int prep_defer_bf16(const float* src, void* dst, long count, hipStream_t st);
// mode: 0 = transpose the weights, then run the chain; 1 = transpose only (d_out / act / dz unused);
//       2 = chain only (wt already holds this step's transposed weights)

// Weight / bias gradients of every layer and network of an MLP site in one launch + one reduce launch
// (dims <= 256).  dz[p] + dzoff[p*MF_MAXL + l]: dZ_l from the dgrad launch (l < L-1); d_out: dZ of the last
// layer; yoff[p*MF_MAXL + l]: offset of layer l's output in act[p].  slab: mlp_fused_wgrad_slab_floats() floats.
bool mlp_fused_wgrad_ok(int nprob, int L, const int* dims);
size_t mlp_fused_wgrad_slab_floats(int nprob, const int* M, int L, const int* dims);
int mlp_fused_wgrad(int nprob, const float* const* x, int ldx, const float* const* act, const float* const* d_out, int ldo,
                    const float* const* dz, float* const* grads, float* slab, const int* M, int L, const int* dims,
                    const long* yoff, const long* dzoff, const long* woff, const long* boff, int accumulate, hipStream_t st,
                    const int* acts = nullptr);
// yoff[...] < 0 (forward: do not save that layer's output; wgrad: -(zoff + 1), recompute it from the pre-activation with acts[l])

// ---- many-row problems (>= TACORL_MLP_BIG_ROWS = 16384 rows: C5's Q networks, 99 328): kernels of their own.  Eligible:
// L >= 2, hidden widths 256 with SiLU, dims[0] <= 128, dims[L] <= 4.  A hidden layer l saves, instead of fp32 z / y:
//   ybf[p*MF_MAXL + l]: bf16 [Mp][256] copy of its output (the next layer's MFMA operand and the weight gradients' x operand),
//   sbf[p*MF_MAXL + l]: fp16 [Mp][256] copy of act'(z)   (Mp = M rounded up to 64; float offsets into act[p]);
// the input-gradient chain leaves dZ_l as bf16 [Mp][256] at dz[p] + dzoff (floats), and the weight gradients stream both by
// LDS-DMA (mlp_wgrad_big_kernel).  yout[p]: float offset of the last layer's fp32 output.
bool mlp_big_prob_ok(int M, int L, const int* dims, const int* acts);
int mlp_big_fwd(int nprob, const float* const* x, int ldx, const float* const* params, const void* const* params_bf16,
                float* const* act, const int* M, int L, const int* dims, const int* acts, const long* ybf, const long* sbf,
                const long* yout, const long* woff, const long* boff, hipStream_t st, const MlpXGather* gather = nullptr,
                const int* gmap = nullptr);
int mlp_big_bwd(int nprob, const float* const* act, const float* const* d_out, int ldo, float* const* dz, float* const* d_x,
                int ldd, void* const* wt, const int* M, int L, const int* dims, const long* sbf, const long* dzoff,
                hipStream_t st);
size_t mlp_big_wgrad_slab_floats(int nprob, const int* M, int L, const int* dims);
size_t mlp_big_xb_bytes(int M);
int mlp_fused_wgrad_big(int nprob, const float* const* x, int ldx, const float* const* act, const float* const* d_out, int ldo,
                        const float* const* dz, float* const* grads, float* slab, void* const* xb, const int* M, int L,
                        const int* dims, const long* ybf, const long* dzoff, const long* woff, const long* boff, int accumulate,
                        hipStream_t st, size_t slab_floats);
