// Single-launch MLP forward (mlp_fused.hip), used by tacorl_mlp_fwd when the shapes qualify.
#pragma once
#include <hip/hip_runtime.h>

#define MF_MAXP 8  // problems per launch
#define MF_MAXL 4  // layers

// bf16 compute only: every layer input width a multiple of 8 and <= 256, output widths <= 256.
bool mlp_fused_fwd_ok(int nprob, int L, const int* dims, int ldx);
// zoff / yoff: [nprob][MF_MAXL] offsets (floats) into act[p] (tacorl_mlp_act_layout; zoff < 0 = not saved);
// woff / boff: [L] offsets into params[p] (tacorl_mlp_param_layout).
// params_bf16[p]: bf16 copy of params[p] (same element offsets, 8-byte aligned): the MFMA B operand.
int mlp_fused_fwd(int nprob, const float* const* x, int ldx, const float* const* params, const void* const* params_bf16,
                  float* const* act,
                  const int* M, int L, const int* dims, const int* acts, const long* zoff, const long* yoff,
                  const long* woff, const long* boff, hipStream_t st);
