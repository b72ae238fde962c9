// Shared device helpers for the TACO-RL hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define TACORL_OK 0
#define TACORL_EINVAL (-22)
#define TACORL_ENOMEM (-12)
#define TACORL_ELAUNCH (-5)

// activation ids shared with the host side (include/tacorl_hip.h)
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

// Reductions over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), result in every lane: quad xor 1, quad
// xor 2, row_half_mirror, row_mirror - four VALU-side lane exchanges, no LDS crossbar (ds_bpermute) round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float x) {
  x += dpp_move<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_move<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_move<0x141>(x);  // row_half_mirror: the other quad of this half row
  x += dpp_move<0x140>(x);  // row_mirror: the other half row
  return x;
}
__device__ __forceinline__ float row16_max(float x) {
  x = fmaxf(x, dpp_move<0xB1>(x));
  x = fmaxf(x, dpp_move<0x4E>(x));
  x = fmaxf(x, dpp_move<0x141>(x));
  x = fmaxf(x, dpp_move<0x140>(x));
  return x;
}

__device__ __forceinline__ float act_apply(int act, float z) {
  if (act == ACT_RELU) return z > 0.f ? z : 0.f;
  if (act == ACT_SILU) return z / (1.f + expf(-z));
  return z;
}
// derivative given the pre-activation z (SiLU) or the output y (ReLU: y > 0 <=> z > 0)
__device__ __forceinline__ float act_grad(int act, float zy) {
  if (act == ACT_RELU) return zy > 0.f ? 1.f : 0.f;
  if (act == ACT_SILU) {
    float s = 1.f / (1.f + expf(-zy));
    return s * (1.f + zy * (1.f - s));
  }
  return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------- MFMA atoms
// Both atoms compute a 16x16 tile; lane l holds A[row l&15][k .. k+KPACK) with
// k = KPACK*(l>>4), the same for B[col l&15][...] (B is kept [N][K], K contiguous),
// and C/D: col = l&15, row = 4*(l>>4) + reg   (cdna_hip_programming.md section 3).
struct AtomF32 {  // v_mfma_f32_16x16x4_f32: exact fp32 (k-ordered fmaf chain)
  typedef float elem;
  static constexpr int KPACK = 1;  // consecutive k per lane
  static constexpr int KSTEP = 4;  // k per instruction
  static constexpr int BK = 32;    // k per LDS tile
  static constexpr int PAD = 2;    // LDS row pad (elems): stride 34 -> conflict-free ds_read_b32
  typedef float frag;
  static __device__ __forceinline__ elem cvt(float x) { return x; }
  static __device__ __forceinline__ frag ld(const elem* p) { return *p; }
  // 4 consecutive k of one row: LD = 34 floats -> 8-byte aligned only
  static __device__ __forceinline__ void st4(elem* d, const float (&v)[4]) {
    *reinterpret_cast<f32x2*>(d) = f32x2{v[0], v[1]};
    *reinterpret_cast<f32x2*>(d + 2) = f32x2{v[2], v[3]};
  }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
};
struct AtomBF16 {  // v_mfma_f32_16x16x32_bf16: bf16 operands, fp32 accumulate
  typedef __bf16 elem;
  static constexpr int KPACK = 8;
  static constexpr int KSTEP = 32;
  static constexpr int BK = 32;  // (64 measured 6 % slower on the whole step: fewer resident blocks per CU)
  static constexpr int PAD = 8;  // 16 B
  typedef bf16x8 frag;
  static __device__ __forceinline__ elem cvt(float x) { return (__bf16)x; }
  static __device__ __forceinline__ frag ld(const elem* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static __device__ __forceinline__ void st4(elem* d, const float (&v)[4]) {  // one ds_write_b64
    *reinterpret_cast<bf16x4*>(d) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  }
  static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ void load4_as_float(const T* p, float (&v)[4]);
template <>
__device__ __forceinline__ void load4_as_float<float>(const float* p, float (&v)[4]) {
  f32x4 t = *reinterpret_cast<const f32x4*>(p);
  v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <>
__device__ __forceinline__ void load4_as_float<__bf16>(const __bf16* p, float (&v)[4]) {
  bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
  v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
