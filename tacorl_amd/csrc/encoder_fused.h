// Shared pieces of the fused LMPVisionEncoder forward kernels (encoder_fused.hip: whole-image / banded conv1 geometries,
// encoder_ring.hip: geometries whose conv1 output does not fit the LDS): launch table, packed-weight fragment offsets, the
// inline-asm MFMA forms with AGPR-resident weights and the one-instruction DPP row reductions.
#pragma once
#include "../../include/tacorl_hip.h"
#include "common.h"

#define EF_CHUNK 16     // images per FC batch = the 16 columns of the FC MFMA tiles (8 left half of every tile empty)
constexpr int cmaxi(int a, int b) { return a > b ? a : b; }
#define EF_MAXCH 12     // 16-byte chunks per thread for one image (<= 49 152 B: up to ~90x90x3 bf16)
#define EF_MAXP 16
#define ACT1_STRIDE 80   // bytes per conv1-output pixel (32 ch bf16 + 16 pad)
#define SA_STRIDE 272    // bytes per image of soft-argmax features (128 bf16 + 16 pad)
#define H1_STRIDE 528    // bytes per image of fc1 output (256 bf16 + 16 pad)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct EFProblem {
  const __bf16* img;   // [n][H][W][3]
  const u32x4* wpk;    // packed bf16 fragments (tacorl_encoder_pack_weights)
  const float* params; // fp32 block (biases, temperature)
  float* out;          // [n][32]
  float* act;          // optional saved activations (tacorl_encoder_act_layout) for a later backward
  long a_y2, a_y3, a_sa, a_h1;  // float offsets of y2 / y3 / soft-argmax / fc1 inside act (y1 at 0)
  int n_img;
  int cost;            // launch balance: cost of one image of this problem in 1/64 image (64, or EF_ACT_COST with saved activations)
  long ustart;         // first work unit of this problem's images on the launch's unit line (see ef_partition)
};
struct EFArgs {
  EFProblem p[EF_MAXP];
  long utotal;         // work units of the launch
  int nprob;
  int H, W, OH1, OW1, OH2, OW2, OH3, OW3;
  int img_bytes;   // H*W*3*2
  int lds_img;     // bytes reserved per image buffer (multiple of 16)
  int act_bf16;    // encoder_ring.hip: saved y1 / y2 as bf16 at the start of their slots (1: what tacorl_encoder_bwd_fused* read) or fp32 (0)
};

// packed-fragment offsets (in 16-byte units, 64 lanes per fragment)
#define WP_C1 0                      // [2 ntile][6 kstep][64]
#define WP_C2 (WP_C1 + 2 * 6 * 64)   // [4][16][64]
#define WP_C3 (WP_C2 + 4 * 16 * 64)  // [4][18][64]
#define WP_F1 (WP_C3 + 4 * 18 * 64)  // [16][4][64]
#define WP_F2 (WP_F1 + 16 * 4 * 64)  // [2][8][64]
#define WP_TOTAL (WP_F2 + 2 * 8 * 64)

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
// max(x, 0) as ONE v_max_f32: fmaxf() costs two (hipcc first canonicalises x with v_max x, x - signalling-NaN
// quieting the accumulators here cannot need); 20 of them per 16-pixel tile add up at one wave per SIMD
__device__ __forceinline__ float relu1(float x) {
  float r;
  asm("v_max_f32_e32 %0, 0, %1" : "=v"(r) : "v"(x));
  return r;
}
// One butterfly step of a 16-lane (DPP row) reduction on FOUR independent values in four instructions: v_op_dpp reads its
// first source through the lane permutation, so a step is one instruction instead of v_mov_dpp + v_op (hipcc fuses only
// some of them).  Four values per statement: a VALU result needs two wait states before a DPP read of it, and inline
// asm gets no hazard padding - the three other instructions of the group provide them.
#define EF_DPP4(OP, CTRL, x)                                                                  \
  asm volatile(OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
               OP " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
               OP " %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
               OP " %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf"                             \
               : "+v"((x)[0]), "+v"((x)[1]), "+v"((x)[2]), "+v"((x)[3]))
// (the wait states in front of the first step sit INSIDE its asm statement, behind the operands: as a statement of its own -
// `asm volatile("s_nop 1")` - nothing kept hipcc from scheduling the VALU instruction that produces x[0] between the nop and
// the first DPP read of x[0]; it did once the code around the soft-argmax changed in round 6, and channel q = 0 of every
// lane came out wrong)
#define EF_DPP4_FIRST(OP, CTRL, x)                                                            \
  asm volatile("s_nop 1\n\t"                                                                   \
               OP " %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
               OP " %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
               OP " %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
               OP " %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf"                             \
               : "+v"((x)[0]), "+v"((x)[1]), "+v"((x)[2]), "+v"((x)[3]))
#define EF_ROW16_4(OP, x)                          \
  do {                                             \
    EF_DPP4_FIRST(OP, "quad_perm:[1,0,3,2]", x);   \
    EF_DPP4(OP, "quad_perm:[2,3,0,1]", x);         \
    EF_DPP4(OP, "row_half_mirror", x);             \
    EF_DPP4(OP, "row_mirror", x);                  \
  } while (0)
__device__ __forceinline__ u32x2 pack4_bf16(float a, float b, float c, float d) {
  bf16x4 t = {(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d};
  return __builtin_bit_cast(u32x2, t);
}

#ifndef EF_VAR
#define EF_VAR 0   // scratch timing builds only (see encoder_fused.hip)
#endif
// MFMA with the weight fragment held in AGPRs (the conv2/conv3 weights fill 136 AGPRs; as plain
// builtin operands hipcc keeps them in arch VGPRs, runs out, and serialises every LDS read behind one
// shared destination register).  Inline asm is invisible to hipcc's hazard recogniser, so the chain
// brackets itself: s_nop before the first MFMA (VALU-written accumulator) and after the last one
// (MFMA result read by VALU) - cdna_hip_programming.md section 5.7.
#if EF_VAR & 2
#define MFMA_AW(acc, wfrag, bfrag) asm volatile("" : "+v"(acc) : "a"(wfrag), "v"(bfrag))
#else
#define MFMA_AW(acc, wfrag, bfrag) \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wfrag), "v"(bfrag))
#endif
#define MFMA_CHAIN_BEGIN(acc) asm volatile("s_nop 1" : "+v"(acc))
// First MFMA of a chain: the bias registers are its C operand and the accumulator only its destination - no four
// v_mov per chain to seed the accumulator (22 chains per image), and no VALU-write -> MFMA-read wait either: the
// bias registers were written once, before the image loop.
#if EF_VAR & 2
#define MFMA_FIRST_AW(acc, wfrag, bfrag, bias) asm volatile("" : "=v"(acc) : "a"(wfrag), "v"(bfrag), "v"(bias))
#else
#define MFMA_FIRST_AW(acc, wfrag, bfrag, bias) \
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=v"(acc) : "a"(wfrag), "v"(bfrag), "v"(bias))
#endif
// (12 wait states: what an 8-pass XDL result needs before a non-MFMA reader, cdna_hip_programming.md section 5.7 item 2;
// v_mfma_f32_16x16x32_bf16 issues every ~17 clk in a dependent chain, i.e. is a 4-pass op - 20 states were used before)
#define MFMA_CHAIN_END(acc) asm volatile("s_nop 11" : "+v"(acc))
#if EF_VAR & 2
#define EF_MFMA_TXT(...) ""
#else
#define EF_MFMA_TXT(...) __VA_ARGS__
#endif

// encoder_ring.hip
int ef_ring_supported(int H, int W);
int ef_ring_launch(EFArgs& a, int nb, int H, int W, hipStream_t st);
