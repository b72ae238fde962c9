// Fused LMPVisionEncoder forward (inference / no-grad problems): conv8x8s4+ReLU -> conv4x4s2+ReLU ->
// conv3x3s1+ReLU -> spatial soft-argmax -> FC128-256+ReLU -> FC256-32, one launch, bf16 MFMA with
// fp32 accumulation, every intermediate on chip (reference networks/visual_encoders/encoder.py:369-419,
// utils.py:39-65).
//
// Design (MI355X_MICROARCH / cdna_hip_programming guides):
//  * weights are REGISTER-stationary (248 AGPRs, fed to the MFMAs by inline asm, loaded once per workgroup and
//    reused for every image it processes): conv1 - all 32 channels in every wave, the waves split the pixel
//    tiles; conv2 - a wave owns 32 output channels and every second pixel tile, so each im2col fragment read
//    from LDS feeds two MFMAs; conv3 - a wave owns 16 channels and reads every pixel tile (a quarter of fc1,
//    half of fc2 from global memory per chunk of 8 images).  One wave per SIMD: that is what bounds the kernel;
//  * activations live in LDS: the raw NHWC bf16 image (double-buffered, next image prefetched through
//    registers while the current one is computed), conv1 output [pixel][32ch] with an 80-byte pixel
//    stride and conv2 output [pixel][64ch] with a 160-byte stride - strides chosen so that the
//    im2col fragment reads (ds_read_b128, 16 consecutive output pixels x 8 channels) hit 16 distinct
//    16-byte slots per lane group (conflict-free);
//  * MFMA orientation D[channel][pixel] = W[channel][k] * im2col[k][pixel]: a lane ends up with 4
//    consecutive channels of one pixel -> one 8-byte LDS store per tile;
//  * conv3's accumulators never leave registers: bias+ReLU+soft-argmax (max / sum-exp / E[x], E[y])
//    are reduced with 4 xor-shuffles across the 16 pixel lanes;
//  * the FC tail runs once per chunk of 8 images (pixels -> images on the MFMA column axis).
// HBM traffic = the bf16 image (42 336 B @84x84) + 128 B of output per image.
#include <stdio.h>
#include <stdlib.h>

#include "encoder_fused.h"
#include "enc_bwd_fused.h"

#ifndef EF_DEFER
#define EF_DEFER 1      // 0: every pixel in the per-image tiles, as rounds 1-5 (A/B builds)
#endif
#define EF_DCHUNK 8     // images per leftover-pixel tile (8 of its 16 MFMA columns; 16 would not fit the LDS)
#ifndef EF_SETUP_COST
#define EF_SETUP_COST 128 // launch balance: the prologue of a workgroup's second problem (248 weight registers per lane from L2, first image), in 1/64 image: measured ~2 images (a 27-image workgroup that crosses: 256 k clk against 238 k)
#endif
#ifndef EF_ACT_COST
#define EF_ACT_COST 72  // cost of an image whose activations are also stored, in 1/64 of a plain image: 10.7 k vs 9.5 k clk (per-workgroup clocks, -DEF_BLKCLK)
                        // (49 KB of stores through a ~14 B/clk per-CU store path; per-workgroup clocks of a -DEF_BLKCLK build,
                        // scratch/run_fused.py: critical workgroup 335 k clk at 64, 313 k at 72, 292 k at 80, 295 k at 84; mean 279 k)
#endif


// ------------------------------------------------------------------ weight packing
// frag(j, s, l)[e] = W[16 j + (l & 15)][32 s + 8 (l >> 4) + e]   (W row-major [N][K], fp32 -> bf16)
// conv1 uses a permuted K order so that the im2col LDS address of lane group g is base + g*rowbytes +
// an immediate: k-step s, group g, element e  <->  ky = 4 (s / 3) + g, (kx, ci) run offset 8 (s % 3) + e.
struct EFPackArgs {
  const float* params[16];
  u32x4* out[16];
};
__global__ void ef_pack_kernel(EFPackArgs a, long o_w1, long o_w2, long o_w3, long o_f1, long o_f2) {
  const float* __restrict__ params = a.params[blockIdx.y];
  u32x4* __restrict__ out = a.out[blockIdx.y];
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;  // fragment id
  if (f >= WP_TOTAL / 64) return;
  const float* W; int K, j, s;
  if (f < WP_C2 / 64) { W = params + o_w1; K = 192; j = f / 6; s = f % 6; }
  else if (f < WP_C3 / 64) { int r = f - WP_C2 / 64; W = params + o_w2; K = 512; j = r / 16; s = r % 16; }
  else if (f < WP_F1 / 64) { int r = f - WP_C3 / 64; W = params + o_w3; K = 576; j = r / 18; s = r % 18; }
  else if (f < WP_F2 / 64) { int r = f - WP_F1 / 64; W = params + o_f1; K = 128; j = r / 4; s = r % 4; }
  else { int r = f - WP_F2 / 64; W = params + o_f2; K = 256; j = r / 8; s = r % 8; }
  const float* src = W + (long)(16 * j + (l & 15)) * K + 32 * s + 8 * (l >> 4);
  // conv1 also permutes the output channels between its two tiles: row 4 a + r of tile j is channel 8 a + 4 j + r,
  // so that a lane's accumulators (rows 4 g .. 4 g + 3 of both tiles) are the 8 ADJACENT channels 8 g .. 8 g + 7
  // (one 16-byte activation store per pixel tile instead of two 8-byte ones)
  if (f < WP_C2 / 64)
    src = W + (long)(8 * ((l & 15) >> 2) + 4 * j + (l & 3)) * K + (4 * (s / 3) + (l >> 4)) * 24 + 8 * (s % 3);
  // conv2: a wave owns the fragment pair (2 cg, 2 cg + 1) = channels 32 cg .. 32 cg + 31, permuted the same way:
  // row 4 a + r of fragment 2 cg + ct is channel 32 cg + 8 a + 4 ct + r
  else if (f < WP_C3 / 64)
    src = W + (long)(32 * (j >> 1) + 8 * ((l & 15) >> 2) + 4 * (j & 1) + (l & 3)) * K + 32 * s + 8 * (l >> 4);
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; e++) v[e] = (__bf16)src[e];
  out[(long)f * 64 + l] = __builtin_bit_cast(u32x4, v);
}

extern "C" long tacorl_encoder_fused_wpk_bytes(void) { return (long)WP_TOTAL * 16; }
extern "C" int tacorl_encoder_pack_weights(int nprob, const float* const* params, void* const* packed,
                                           tacorl_stream_t stream) {
  long po[11];
  tacorl_encoder_param_layout(po);
  if (nprob < 1 || nprob > 16) return TACORL_EINVAL;
  EFPackArgs a{};
  for (int p = 0; p < nprob; p++) { a.params[p] = params[p]; a.out[p] = (u32x4*)packed[p]; }
  hipLaunchKernelGGL(ef_pack_kernel, dim3((WP_TOTAL / 64 + 3) / 4, nprob), dim3(256), 0, (hipStream_t)stream, a, po[0],
                     po[2], po[4], po[7], po[9]);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

// ---- MFMA chains as ONE asm statement each (operands: %0 accumulator, then the N weight fragments, the N im2col fragments, the bias).  hipcc pads an
// `s_nop 0` between two adjacent asm statements and a counted s_waitcnt in front of each one whose fragment is the
// youngest LDS read: per 16x16x32 MFMA that was 1.3 scalar instructions, each a 4-5 clk issue slot at one wave per SIMD.
#define MFMA6_FIRST(acc, bias, w, o, f) \
  asm volatile(EF_MFMA_TXT("v_mfma_f32_16x16x32_bf16 %0, %1, %7, %13\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %2, %8, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %3, %9, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %4, %10, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %5, %11, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %6, %12, %0\n\t") \
      : "=&v"(acc) : "a"((w)[(o) + 0]), "a"((w)[(o) + 1]), "a"((w)[(o) + 2]), "a"((w)[(o) + 3]), "a"((w)[(o) + 4]), "a"((w)[(o) + 5]), "v"((f)[0]), "v"((f)[1]), "v"((f)[2]), "v"((f)[3]), "v"((f)[4]), "v"((f)[5]), "v"(bias))
#define MFMA6_MORE(acc, w, o, f) \
  asm volatile(EF_MFMA_TXT("v_mfma_f32_16x16x32_bf16 %0, %1, %7, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %2, %8, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %3, %9, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %4, %10, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %5, %11, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %6, %12, %0\n\t") \
      : "+v"(acc) : "a"((w)[(o) + 0]), "a"((w)[(o) + 1]), "a"((w)[(o) + 2]), "a"((w)[(o) + 3]), "a"((w)[(o) + 4]), "a"((w)[(o) + 5]), "v"((f)[0]), "v"((f)[1]), "v"((f)[2]), "v"((f)[3]), "v"((f)[4]), "v"((f)[5]))
#define MFMA8_FIRST(acc, bias, w, o, f) \
  asm volatile(EF_MFMA_TXT("v_mfma_f32_16x16x32_bf16 %0, %1, %9, %17\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %2, %10, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %3, %11, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %4, %12, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %5, %13, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %6, %14, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %7, %15, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %8, %16, %0\n\t") \
      : "=&v"(acc) : "a"((w)[(o) + 0]), "a"((w)[(o) + 1]), "a"((w)[(o) + 2]), "a"((w)[(o) + 3]), "a"((w)[(o) + 4]), "a"((w)[(o) + 5]), "a"((w)[(o) + 6]), "a"((w)[(o) + 7]), "v"((f)[0]), "v"((f)[1]), "v"((f)[2]), "v"((f)[3]), "v"((f)[4]), "v"((f)[5]), "v"((f)[6]), "v"((f)[7]), "v"(bias))
#define MFMA8_MORE(acc, w, o, f) \
  asm volatile(EF_MFMA_TXT("v_mfma_f32_16x16x32_bf16 %0, %1, %9, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %2, %10, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %3, %11, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %4, %12, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %5, %13, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %6, %14, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %7, %15, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %8, %16, %0\n\t") \
      : "+v"(acc) : "a"((w)[(o) + 0]), "a"((w)[(o) + 1]), "a"((w)[(o) + 2]), "a"((w)[(o) + 3]), "a"((w)[(o) + 4]), "a"((w)[(o) + 5]), "a"((w)[(o) + 6]), "a"((w)[(o) + 7]), "v"((f)[0]), "v"((f)[1]), "v"((f)[2]), "v"((f)[3]), "v"((f)[4]), "v"((f)[5]), "v"((f)[6]), "v"((f)[7]))
#define MFMA9_FIRST(acc, bias, w, o, f) \
  asm volatile(EF_MFMA_TXT("v_mfma_f32_16x16x32_bf16 %0, %1, %10, %19\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %2, %11, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %3, %12, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %4, %13, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %5, %14, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %6, %15, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %7, %16, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %8, %17, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %9, %18, %0\n\t") \
      : "=&v"(acc) : "a"((w)[(o) + 0]), "a"((w)[(o) + 1]), "a"((w)[(o) + 2]), "a"((w)[(o) + 3]), "a"((w)[(o) + 4]), "a"((w)[(o) + 5]), "a"((w)[(o) + 6]), "a"((w)[(o) + 7]), "a"((w)[(o) + 8]), "v"((f)[0]), "v"((f)[1]), "v"((f)[2]), "v"((f)[3]), "v"((f)[4]), "v"((f)[5]), "v"((f)[6]), "v"((f)[7]), "v"((f)[8]), "v"(bias))
#define MFMA9_MORE(acc, w, o, f) \
  asm volatile(EF_MFMA_TXT("v_mfma_f32_16x16x32_bf16 %0, %1, %10, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %2, %11, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %3, %12, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %4, %13, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %5, %14, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %6, %15, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %7, %16, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %8, %17, %0\n\t" \
      "v_mfma_f32_16x16x32_bf16 %0, %9, %18, %0\n\t") \
      : "+v"(acc) : "a"((w)[(o) + 0]), "a"((w)[(o) + 1]), "a"((w)[(o) + 2]), "a"((w)[(o) + 3]), "a"((w)[(o) + 4]), "a"((w)[(o) + 5]), "a"((w)[(o) + 6]), "a"((w)[(o) + 7]), "a"((w)[(o) + 8]), "v"((f)[0]), "v"((f)[1]), "v"((f)[2]), "v"((f)[3]), "v"((f)[4]), "v"((f)[5]), "v"((f)[6]), "v"((f)[7]), "v"((f)[8]))

// ------------------------------------------------------------------------- kernel

template <int H_, int W_>
struct EFGeom {
  static constexpr int H = H_, W = W_;
  static constexpr int OH1 = (H - 8) / 4 + 1, OW1 = (W - 8) / 4 + 1;
  static constexpr int OH2 = (OH1 - 4) / 2 + 1, OW2 = (OW1 - 4) / 2 + 1;
  static constexpr int OH3 = OH2 - 2, OW3 = OW2 - 2;
  static constexpr int IMG_BYTES = H * W * 6;
  static constexpr int NPX1 = OH1 * OW1, NPX2 = OH2 * OW2, NPX3 = OH3 * OW3;
  // Row pitches of the two activation images in LDS.  Pixel strides 80 / 160 B alone are conflict-free only inside one
  // image row; a 16-pixel tile spans 2-3 rows of a 9- or 7-pixel-wide output, and with dense rows the wrap put two
  // lanes of a ds_read_b128 lane group on the same banks: 7.3 / 7.0 LDS cycles per read instead of 4 (measured,
  // scratch/micro/lds_pat; model scratch/micro/bank_sim.py, which also found these pads - the smallest that give 4.0
  // for every tile and tap of the geometry).
  static constexpr int PAD1 = (H == 84 && W == 84) ? 16 : (H == 44 && W == 60) ? 0 : 48;
  static constexpr int PX2 = 160;  // bytes per conv2-output pixel (64 ch bf16 + 32 pad)
  static constexpr int PAD2 = (H == 44 && W == 60) ? 32 : 192;
  static constexpr int PITCH1 = OW1 * 80 + PAD1, PITCH2 = OW2 * PX2 + PAD2;
  static constexpr int ACT1_BYTES = OH1 * PITCH1, ACT2_BYTES = OH2 * PITCH2;
  // conv1 tiles of 16 output pixels with x stride 2 (all-even-x tiles, all-odd-x tiles, one mixed tail tile): the two
  // image rows a 32-lane ds_read_b64 group touches are 504 B = 63 (odd) 8-byte units apart, pixels 2 apart are 6
  // units apart (even), so the group's 32 accesses fall on 32 different 8-byte units mod 256 B - conflict-free, and the
  // reads are issued as ds_read_b64 pairs at 256 B/clk (hipcc's merged ds_read2_b64 runs at 128 B/clk: 8 LDS cycles
  // per fragment against 2 x 2.08; this file is compiled with the load-store-opt target feature off, build.py).
  // Row pitches other than 504 B do not have the property (bank_sim.py).
  static constexpr bool S2 = (H == 84 && W == 84);
  // Round 6 - the LEFTOVER PIXEL is deferred.  84 x 84: conv2 has 81 = 5 x 16 + 1 output pixels and conv3 49 = 3 x 16 + 1, so a
  // sixth conv2 tile and a fourth conv3 tile carried ONE pixel each (15.6 % / 23.4 % of those layers' MFMAs, fragment
  // reads and epilogues were padding).  The last conv2 pixel (OH2-1, OW2-1) feeds only the last conv3 pixel, so both can
  // wait: per image the kernel runs 5 / 3 full tiles, keeps the two pixels' im2col rows (D1: the 4 x 4 x 32 window of
  // conv1's output, D2: the 3 x 3 x 64 window of conv2's) and the soft-argmax state of the other 48 pixels (PS: running
  // max / sum-exp / E[x] / E[y] per channel) in LDS, and once per EF_DCHUNK images ONE conv2 tile and ONE conv3 tile run
  // over the leftover pixels of those images (image = MFMA column), whose result is merged into the soft-argmax state.
  // PS lives in the FC tail's h1 buffer (dead until the tail, which follows the merge).
  static constexpr int D1S = 16 * 64 + 16, D2S = 9 * 128 + 16;   // per-image pitches: 16-byte slots r16 apart -> conflict-free ds_read_b128
  static constexpr bool DEFER = EF_DEFER != 0 && NPX2 % 16 == 1 && NPX3 % 16 == 1 && NPX2 > 16 && NPX3 > 16 &&
                                2 * ((IMG_BYTES + 15) & ~15) + ACT1_BYTES + ACT2_BYTES + EF_CHUNK * (272 + 528) +
                                        EF_DCHUNK * (D1S + D2S) <= 160 * 1024;
  static constexpr int DEFER_BYTES = DEFER ? EF_DCHUNK * (D1S + D2S) : 0;
  static constexpr int NPX2M = DEFER ? NPX2 - 1 : NPX2, NPX3M = DEFER ? NPX3 - 1 : NPX3;  // pixels of the per-image tiles
  static constexpr int FIXED_BYTES = ACT1_BYTES + ACT2_BYTES + EF_CHUNK * (272 + 528) + DEFER_BYTES;
  // The image buffer is double: two whole images when they fit beside the activations (84 x 84: 2 x 42 KB); otherwise
  // conv1 runs over BANDS of BR output rows = 4 BR + 4 image rows (128 x 128: 8 bands of 20 rows, 2 x 15 KB; the 4
  // rows two bands share are fetched twice - from L2), the next band streaming in while the current one is computed.
  static constexpr bool WHOLE = 2 * ((IMG_BYTES + 15) & ~15) + FIXED_BYTES <= 160 * 1024;
  static constexpr int BR = WHOLE ? OH1 : 4, NB = (OH1 + BR - 1) / BR;
  static constexpr int BAND_ROWS = WHOLE ? H : 4 * BR + 4;
  static constexpr int BAND_BYTES = BAND_ROWS * W * 6, LDS_IMG = (BAND_BYTES + 15) & ~15;
  static constexpr int NPXB = BR * OW1;  // conv1 output pixels of a full band
  static constexpr int LDS_BYTES = 2 * LDS_IMG + FIXED_BYTES;
  static constexpr bool OK = (WHOLE ? IMG_BYTES % 16 == 0 : (W * 6) % 16 == 0) && BAND_BYTES <= EF_MAXCH * 256 * 16 &&
                             LDS_BYTES <= 160 * 1024 && OH3 >= 1 && OW3 >= 1;
};

// Phase timing for kernel work (scratch builds with -DEF_STAMPS only; the product build has no trace of it):
// wave 0 of block 0 accumulates shader-clock deltas per phase, read back with tacorl_ef_stamps_read.
#ifndef EF_VAR
#define EF_VAR 0
#endif
#ifndef EF_X   // scratch experiments (timing only, wrong results): 1 = no image DMA inside the loop, 2 = conv1 does not store its
#define EF_X 0 // activations, 4 = every fetch re-reads one L2-resident image, 8 = no per-image barriers, 16 = no soft-argmax arithmetic
#endif
#if EF_X & 8
#define EF_IMG_BARRIER() do { } while (0)
#else
#define EF_IMG_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#if defined(EF_STAMPS) || defined(EF_BLKCLK)  // scratch builds: per-workgroup shader clocks from entry to exit
__device__ unsigned long long ef_blk[512];    // [0, 256) clocks, [256, 512) images served
extern "C" int tacorl_ef_blk_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ef_blk), sizeof(ef_blk)) == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}
#endif
#ifdef EF_STAMPS
__device__ unsigned long long ef_stamps[64];
#ifndef EF_STAMP_WAVE
#define EF_STAMP_WAVE 0
#endif
#define STAMP(k)                                                    \
  do {                                                              \
    if (blockIdx.x == 0 && threadIdx.x == 64 * EF_STAMP_WAVE) {     \
      const unsigned long long t_ = clock64();                      \
      atomicAdd(&ef_stamps[k], t_ - t_prev);                        \
      t_prev = t_;                                                  \
    }                                                               \
  } while (0)
extern "C" int tacorl_ef_stamps_read(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(ef_stamps), sizeof(ef_stamps)) != hipSuccess) return TACORL_ELAUNCH;
  if (reset) {
    unsigned long long z[64] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(ef_stamps), z, sizeof(z)) != hipSuccess) return TACORL_ELAUNCH;
  }
  return TACORL_OK;
}
#else
#define STAMP(k)
#endif

template <int H_, int W_>
__global__ __launch_bounds__(256, 1) void encoder_fused_kernel(EFArgs a_) {
  typedef EFGeom<H_, W_> G;
  struct { int W, OW1, OW2, OW3, OH1, OH2, OH3, img_bytes, lds_img, nprob; const EFProblem* p; } a;
  a.W = G::W; a.OW1 = G::OW1; a.OW2 = G::OW2; a.OW3 = G::OW3; a.OH1 = G::OH1; a.OH2 = G::OH2; a.OH3 = G::OH3;
  a.img_bytes = G::IMG_BYTES; a.lds_img = G::LDS_IMG; a.nprob = a_.nprob; a.p = a_.p;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // Round 6 - launch balance by WORK UNITS, not by whole workgroups per problem.  The launch lasts as long as its slowest
  // workgroup; with a workgroup bound to one problem (rounds 2-5: ceil(images / workgroups) per problem, bisected budget)
  // the image counts per workgroup came out 24 .. 29 at the bench shapes and the critical workgroup ran 7 % over the mean
  // (per-workgroup shader clocks, -DEF_BLKCLK: 257 k against 240 k).  Now the problems' images lie on ONE line of work
  // units (an image = its problem's cost; EF_SETUP_COST dead units in front of every problem but the first stand for the
  // weight reload of a workgroup that crosses into it), workgroup b owns the units [b U / n, (b + 1) U / n) and serves the
  // images that START inside them - a contiguous run of images of one problem, or the tail of one problem and the head of
  // the next (then it reloads its register-stationary weights once).  Which workgroup serves which image changes nothing
  // in the results (images are independent; FC chunks and leftover-pixel groups are per-column).
  const unsigned long u_lo = (unsigned long)blockIdx.x * (unsigned long)a_.utotal / gridDim.x;
  const unsigned long u_hi = (unsigned long)(blockIdx.x + 1) * (unsigned long)a_.utotal / gridDim.x;
#if defined(EF_STAMPS) || defined(EF_BLKCLK)
  const unsigned long long t_entry = clock64();
  int n_served = 0;
#endif
#ifdef EF_STAMPS
  unsigned long long t_prev = t_entry;
#endif
#pragma unroll 1
  for (int pi = 0; pi < a_.nprob; pi++) {
  const long seg_s = a_.p[pi].ustart, seg_e = seg_s + (long)a_.p[pi].n_img * a_.p[pi].cost;
  if ((long)u_hi <= seg_s || (long)u_lo >= seg_e) continue;
  const int pcost = a_.p[pi].cost;
  const long f0 = (long)u_lo > seg_s ? ((long)u_lo - seg_s + pcost - 1) / pcost : 0;
  const long f1 = (long)u_hi < seg_e ? ((long)u_hi - seg_s + pcost - 1) / pcost : a_.p[pi].n_img;
  if (f1 <= f0) continue;
  const EFProblem P = a_.p[pi];
  // (the lane indices are derived behind an opaque zero INSIDE the problem loop: everything that depends on them is then
  // computed per problem, as the one-problem kernel computed it once - hoisted out of this loop by LICM the same values
  // overflowed the register file, and 60 serialised scratch reloads made a workgroup's prologue 35 k clocks instead of 12 k)
  int tid;
  asm volatile("v_mov_b32 %0, 0" : "=v"(tid));
  tid += threadIdx.x;
  const int w = tid >> 6, l = tid & 63, r16 = l & 15, g = l >> 4;
  // (this run of images is [f0, f1): the loop below walks it with `worker` = its first image and stride `nworkers` = 1)
  const int nworkers = 1;
  const long worker = f0, seg_end = f1;

  unsigned char* act1 = lds + 2 * a.lds_img;
  const int npx1 = a.OH1 * a.OW1, npx2 = a.OH2 * a.OW2, npx3 = a.OH3 * a.OW3;
  unsigned char* act2 = act1 + G::ACT1_BYTES;
  unsigned char* sa = act2 + G::ACT2_BYTES;
  unsigned char* h1 = sa + EF_CHUNK * SA_STRIDE;
  // leftover-pixel deferral (G::DEFER): D1 / D2 behind the FC staging, the soft-argmax state in h1 (dead outside the FC tail)
  unsigned char* const d1 = h1 + EF_CHUNK * H1_STRIDE;
  unsigned char* const d2 = d1 + EF_DCHUNK * G::D1S;
  unsigned char* const ps = h1;
  static_assert(!G::DEFER || EF_DCHUNK * 1024 <= EF_CHUNK * H1_STRIDE, "the soft-argmax state fits the FC tail's h1 buffer");
  static_assert(EF_CHUNK % EF_DCHUNK == 0, "a leftover-pixel group never straddles an FC chunk");

  // ---- register-stationary weights and biases
  long po[11];
  {
    const long sz[11] = {32 * 8 * 8 * 3, 32, 64 * 4 * 4 * 32, 64, 64 * 3 * 3 * 64, 64, 1, 256 * 128, 256, 32 * 256, 32};
    long off = 0;
    for (int i = 0; i < 11; i++) { po[i] = off; off = (off + sz[i] + 3) & ~3L; }
  }
  // conv2 work split: wave = (channel half cg, pixel-tile parity ph).  It owns 32 output channels (two MFMA tiles)
  // and every second 16-pixel tile, so each im2col fragment read from LDS feeds TWO MFMAs: with one wave per SIMD an
  // LDS read costs the wave ~32 clk and adds to its MFMA time, and conv2 sat at its read count (96 per image).
  // conv3 keeps 16 channels per wave (its second fragment set would not fit the register file: 248 of 256 AGPRs
  // are weights now).
  const int cg = w & 1, ph = w >> 1;
  u32x4 wc1a[6], wc1b[6], wc2a[16], wc2b[16], wc3[18];  // all AGPR-resident (248 AGPRs)
#pragma unroll
  for (int s = 0; s < 6; s++) { wc1a[s] = P.wpk[WP_C1 + s * 64 + l]; wc1b[s] = P.wpk[WP_C1 + (6 + s) * 64 + l]; }
#pragma unroll
  for (int s = 0; s < 16; s++) {
    wc2a[s] = P.wpk[WP_C2 + ((2 * cg) * 16 + s) * 64 + l];
    wc2b[s] = P.wpk[WP_C2 + ((2 * cg + 1) * 16 + s) * 64 + l];
  }
#pragma unroll
  for (int s = 0; s < 18; s++) wc3[s] = P.wpk[WP_C3 + (w * 18 + s) * 64 + l];
  float b1[2][4], b2[2][4], b3[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    b1[0][q] = P.params[po[1] + 8 * g + q];  // conv1 tile j, row 4 g + q  <->  channel 8 g + 4 j + q (ef_pack_kernel)
    b1[1][q] = P.params[po[1] + 8 * g + 4 + q];
    b2[0][q] = P.params[po[3] + 32 * cg + 8 * g + q];  // fragment 2 cg + ct, row 4 g + q <-> channel 32 cg + 8 g + 4 ct + q
    b2[1][q] = P.params[po[3] + 32 * cg + 8 * g + 4 + q];
    b3[q] = P.params[po[5] + 16 * w + 4 * g + q];
  }
  const f32x4 bias1a = {b1[0][0], b1[0][1], b1[0][2], b1[0][3]}, bias1b = {b1[1][0], b1[1][1], b1[1][2], b1[1][3]},
              bias2a = {b2[0][0], b2[0][1], b2[0][2], b2[0][3]}, bias2b = {b2[1][0], b2[1][1], b2[1][2], b2[1][3]},
              bias3 = {b3[0], b3[1], b3[2], b3[3]};
  const float temp = P.params[po[6]];
  // Every load the COMPILER knows about must have completed before the image loop: hipcc's wait-count pass does not
  // see the asm-issued LDS-DMA pieces, so a "vmcnt(7)" it places at the first in-loop use of a bias register to
  // cover these few prologue loads in fact waits for all but the 7 youngest of the NEXT image's pieces - an HBM
  // round trip exposed at the top of every image (round 2's kernel had four such waits in its loop).
  asm volatile("" ::"v"(bias1a), "v"(bias1b), "v"(bias2a), "v"(bias2b), "v"(bias3), "v"(temp));

#if EF_VAR & 1
  u32x4 kconst;  // scratch builds: a benign fragment (bf16 1/128 everywhere) instead of every LDS fragment read
  asm volatile("v_mov_b32 %0, 0x3c003c00" : "=v"(kconst[0]));
  kconst[1] = kconst[2] = kconst[3] = kconst[0];
#endif
  // conv1 k offsets (bytes): lane group g reads image row 4 (s / 3) + g, 8-element group s % 3 of the 24-run
  const int k1g = g * a.W * 6;

  // ---- image streaming: LDS-DMA (global_load_lds_dwordx4: no staging registers).  One wave-instruction
  // moves 64 x 16 B = 1 KiB from a contiguous global span to a contiguous LDS span (wave-uniform LDS base).
  // SGPR-base form: address = uniform 64-bit base (image + 4 KiB step) + one constant per-lane byte offset, LDS
  // destination in M0 - no per-piece vector address arithmetic (the flat-pointer form cost 4 VALU + 6 SALU per piece).
  const unsigned dma_voff = G::WHOLE ? (unsigned)l * 16u : (unsigned)tid * 16u;
  const unsigned dma_voff_l = (unsigned)l * 16u;
  const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds;
  // pieces [i0, i1) of (image, band).  Band geometries: piece i = the 256-lane span of 4 KiB at offset 4096 i, this wave's
  // KiB of it.  Whole-image geometries: the image is NPC KiB-pieces q, wave w issues DQ_LO or DQ_HI consecutive ones from
  // dq0 - conv1's tiles do not divide by four (84 x 84: 25 tiles = 7 + 6 + 6 + 6) and a piece costs the issuing wave ~100
  // clk, a quarter of a tile: the waves with the extra tile issue fewer pieces (6 / 12 / 12 / 12 instead of 11 each),
  // so that all four reach the conv1 -> conv2 barrier together
  constexpr int NPC = (G::IMG_BYTES + 1023) / 1024;
  constexpr int NT1W = (G::NPXB + 15) >> 4, PER1W = (NT1W + 3) >> 2, NFULL = NT1W - 4 * (PER1W - 1);  // waves 0 .. NFULL-1 own PER1W tiles
  constexpr int DQ_BAL = NFULL == 4 ? (NPC + 3) / 4 : (NPC + 4 * NFULL + 3) / 4;  // (a tile ~ four pieces)
  constexpr int DQ_HI = PER1W > 1 && DQ_BAL > 2 * (PER1W - 1) ? 2 * (PER1W - 1) : DQ_BAL;  // at most two pieces per tile's first chain
  constexpr int DQ_LO = NFULL == 4 ? DQ_HI : cmaxi((NPC - (4 - NFULL) * DQ_HI + NFULL - 1) / NFULL, 0);
  static_assert(!G::WHOLE || NFULL * DQ_LO + (4 - NFULL) * DQ_HI >= NPC, "every KiB of the image has an issuer");
  const int wu = __builtin_amdgcn_readfirstlane(w);  // (scalar: the DMA's base registers derive from it)
  const int dqn = wu < NFULL ? DQ_LO : DQ_HI, dq0 = wu < NFULL ? wu * DQ_LO : NFULL * DQ_LO + (wu - NFULL) * DQ_HI;
  auto dma_pieces = [&](long img_idx, int band, int buf, int i0, int i1) {
    const int row0 = G::WHOLE ? 0 : 4 * G::BR * band;                                // first image row of the band
    const int rows = G::WHOLE ? G::H : min(G::BAND_ROWS, G::H - row0);
    const int n16 = G::WHOLE ? (G::IMG_BYTES >> 4) : (rows * G::W * 6) >> 4;          // 16-byte chunks to move
    const unsigned char* src = reinterpret_cast<const unsigned char*>(P.img) + img_idx * G::IMG_BYTES + row0 * (G::W * 6);
    if constexpr (!G::WHOLE) {
      // Band geometries: wave w moves the band's KiB pieces 4 w .. 4 w + 3 - FOUR CONSECUTIVE KiB, one M0 / base setup, the
      // pieces by the instruction's offset field (which moves the global and the LDS address alike).  Round 4: the pieces
      // of a wave used to be 4 KiB apart (piece i = the wave's KiB of the workgroup's i-th 4-KiB span): the same bytes, but
      // the whole launch took 0.146 ms instead of 0.127 - per-wave contiguity is worth 11 %, the shared M0 setup 2 % (same
      // box, scratch builds).  (i0, i1 are not used: a band is always issued whole.)
      static_assert(G::WHOLE || G::BAND_BYTES <= 4 * 4096, "a band is at most four KiB pieces per wave");
      const int p0 = 4 * wu, left = n16 - p0 * 64, nfull = left <= 0 ? 0 : (left >= 256 ? 4 : left >> 6);
      const unsigned dstw = lds_base + buf * a.lds_img + p0 * 1024;
      const unsigned char* srcw = src + p0 * 1024;
      const unsigned m0v = __builtin_amdgcn_readfirstlane(dstw);
      unsigned keep_m0;
#define EF_BAND_ASM(LOADS)                                                                                              \
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t" LOADS "s_mov_b32 m0, %0"                         \
               : "=&s"(keep_m0) : "v"(dma_voff_l), "s"(m0v), "s"(srcw) : "memory")
#define EF_LD(OFF) "global_load_lds_dwordx4 %1, %3 offset:" #OFF "\n\t"
      if (nfull == 4) EF_BAND_ASM(EF_LD(0) EF_LD(1024) EF_LD(2048) EF_LD(3072));
      else if (nfull == 3) EF_BAND_ASM(EF_LD(0) EF_LD(1024) EF_LD(2048));
      else if (nfull == 2) EF_BAND_ASM(EF_LD(0) EF_LD(1024));
      else if (nfull == 1) EF_BAND_ASM(EF_LD(0));
      const int part = left - 64 * nfull;  // 16-byte chunks of a last, partial KiB of this wave's share
      if (nfull < 4 && part > 0 && l < part) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep_m0)
                     : "v"(dma_voff_l), "s"(__builtin_amdgcn_readfirstlane(dstw + nfull * 1024)), "s"(srcw + nfull * 1024)
                     : "memory");
      }
#undef EF_LD
#undef EF_BAND_ASM
      return;
    }
    const unsigned dst0 = lds_base + buf * a.lds_img + (G::WHOLE ? dq0 : w) * 1024;
    if constexpr (G::WHOLE) src += dq0 * 1024;
#pragma unroll
    for (int i = i0; i < i1; i++) {
      if (i >= (G::WHOLE ? DQ_HI : EF_MAXCH)) break;
      constexpr int STEP = G::WHOLE ? 1024 : 4096;
      const int c0 = G::WHOLE ? (dq0 + i) * 64 : i * 256 + w * 64;  // first chunk of this wave-instruction (wave-uniform)
      // (whole image: only the last KiB piece is partial - one fixed lane mask instead of a comparison per piece)
      constexpr int LASTN = (G::IMG_BYTES >> 4) - (NPC - 1) * 64;
      const bool wave_on = G::WHOLE ? (i < dqn && dq0 + i < NPC) : c0 < n16;
      if (wave_on) {
        if (G::WHOLE ? (dq0 + i < NPC - 1 || l < LASTN) : (c0 + l < n16)) {
          // inline asm, not __builtin_amdgcn_global_load_lds: with the builtin in the loop hipcc degrades every
          // LDS wait of the conv phases to lgkmcnt(0) (it cannot tell the DMA's LDS writes from the fragment
          // reads), which exposes the full LDS latency at each tile; the ordering against the readers of this
          // buffer is explicit anyway (vmcnt(0) + barrier at the end of the image / band).
          unsigned keep_m0;  // M0 is saved and restored around the piece: hipcc rejects "m0" as a clobber (reserved register)
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep_m0)
                       : "v"(dma_voff), "s"(__builtin_amdgcn_readfirstlane(dst0 + i * STEP)), "s"(src + i * STEP)
                       : "memory");
        }
      }
    }
  };
  auto dma_load = [&](long img_idx, int band, int buf) { dma_pieces(img_idx, band, buf, 0, G::WHOLE ? DQ_HI : EF_MAXCH); };
  // Round 6 - LEAN pieces for the whole-image geometries' in-loop DMA.  The kernel is bound by instruction issue (one wave per
  // SIMD issues one instruction per ~4 clk whatever its kind; the launch takes 0.102 ms with every MFMA removed against
  // 0.127 with them), and a piece as issued by dma_pieces() is ~11 instructions: a has_next branch, the 64-bit source
  // address (s_add / s_addc), M0 saved, set, a wait state, restored, a lane-mask test for "is this the image's last,
  // partial KiB".  Here: the wave's share of the NEXT image is addressed from one (source, LDS) base per 4 KiB - the
  // instruction's 12-bit offset field moves both addresses - set when the group's first piece is issued; M0 simply STAYS
  // (hipcc never touches it in this kernel: checked on the built code object by tests/test_abi_cpu.py); the last image of a
  // workgroup fetches itself again instead of branching around every piece; only the one piece that can be the image's
  // last KiB carries a lane mask.  A piece is then ONE instruction, + 4 per group of four.
  constexpr int DQ0_LAST = NFULL == 4 ? 3 * DQ_HI : NFULL * DQ_LO + (3 - NFULL) * DQ_HI;  // first KiB of wave 3's share
  constexpr int ILAST = NPC - 1 - DQ0_LAST;                                                // its piece that holds the last KiB
  const unsigned char* lean_src = nullptr;
  unsigned lean_dst = 0;
  auto lean_begin = [&](long img_idx, int buf) {
    lean_src = reinterpret_cast<const unsigned char*>(P.img) + img_idx * G::IMG_BYTES + dq0 * 1024;
    lean_dst = lds_base + buf * a.lds_img + dq0 * 1024;
  };
#define EF_LEAN_LD(OFF) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #OFF :: "v"(dma_voff), "s"(gsrc) : "memory")
  auto lean_piece = [&](int i) {  // piece i of this wave's share (i is a compile-time constant at every call site)
    if (!G::WHOLE || i >= DQ_HI) return;
    const bool on = i < DQ_LO || wu >= NFULL;       // (waves 0 .. NFULL - 1 issue DQ_LO pieces, the others DQ_HI)
    if (!on) return;
    if (i > ILAST && wu == 3) return;               // beyond the image's last KiB
    const unsigned char* gsrc = lean_src + (i & ~3) * 1024;
    if ((i & 3) == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(__builtin_amdgcn_readfirstlane(lean_dst + i * 1024)) : "memory");
    constexpr int LASTN = (G::IMG_BYTES >> 4) - (NPC - 1) * 64;  // 16-byte chunks of the image's last KiB
    if (i == ILAST && wu == 3 && l >= LASTN) return;
    switch (i & 3) {
      case 0: EF_LEAN_LD(0); break;
      case 1: EF_LEAN_LD(1024); break;
      case 2: EF_LEAN_LD(2048); break;
      default: EF_LEAN_LD(3072); break;
    }
  };
#undef EF_LEAN_LD

  // images this workgroup processes: worker, worker + nworkers, ... (image granularity: at most one image
  // of imbalance); the FC tail runs after every EF_CHUNK processed images (slots) or at the end.
  long cur = worker;
  int it = 0, buf = 0;
  dma_load(cur, 0, 0);
  // vmcnt(0) as the BUILTIN: it empties the compiler's own scoreboard too (an asm wait does not) - see the note at the
  // bias registers above; the weights load straight into AGPRs and would otherwise be waited for inside the loop
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  STAMP(0);  // prologue: weights to registers, first image

  while (true) {
    const int slot = it & (EF_CHUNK - 1);
    const int dslot = it & (EF_DCHUNK - 1);  // position inside the leftover-pixel group (G::DEFER)
    const long nxt = cur + nworkers;
    const bool has_next = nxt < seg_end;
    // Every conv phase is software-pipelined by hand: the LDS reads of the next half-tile are issued
    // before the MFMA chain of the current one (one wave per SIMD: nothing else hides LDS latency).
    // ------------------------------------------------ conv1: 8x8 stride 4, 3 -> 32, band by band
#pragma unroll 1
    for (int band = 0; band < G::NB; band++) {
      // the other buffer was last read by the previous band's (or image's) conv1, a barrier ago
      // The next image (band) streams in beside this one's conv1.  Whole-image geometries spread their ~11 pieces per
      // wave over the conv1 tiles (two per tile, below): issued in one burst the pieces queue behind one another in the
      // CU's memory pipeline and the issuing wave stands still for ~100 clk per piece (1 000 clk per image).
#if !(EF_X & 1)
      if (band + 1 < G::NB) dma_load(cur, band + 1, buf ^ 1);
      else if (has_next && !G::WHOLE) dma_load(nxt, 0, buf ^ 1);
#endif
#if EF_X & 4  // scratch: every fetch re-reads the workgroup's FIRST image (L2-resident): the DMA without its HBM traffic
      if (G::WHOLE) lean_begin(worker, buf ^ 1);
#else
      if (G::WHOLE) lean_begin(has_next ? nxt : cur, buf ^ 1);  // (the workgroup's last image fetches itself again: no branch per piece)
#endif
      STAMP(1);  // DMA issue
      const unsigned char* ib = lds + buf * a.lds_img;
      constexpr int NT1 = (G::NPXB + 15) >> 4, PER1 = (NT1 + 3) >> 2;
      const int npxb = G::WHOLE ? npx1 : min(G::BR, a.OH1 - band * G::BR) * a.OW1;  // output pixels of this band
      const int nt1 = G::WHOLE ? NT1 : (npxb + 15) >> 4, px0 = band * G::NPXB;
      // pixel of lane r16 in tile mt: (oy, ox) inside the band, clamped to a valid pixel; false for padding lanes
      auto px1 = [&](int mt, int& oy, int& ox) -> bool {
        if (G::S2) {
          constexpr int half = G::OW1 / 2, NE = G::OH1 * half, TE = NE / 16, REM = NE % 16;
          static_assert(!G::S2 || (G::OW1 % 2 == 0 && REM <= 8 && G::WHOLE), "stride-2 conv1 tiles");
          int par, n; bool ok = true;
          if (mt < 2 * TE) { par = mt >= TE ? 1 : 0; n = 16 * (mt - par * TE) + r16; }
          else { par = r16 >> 3; n = 16 * TE + (r16 & 7); ok = (r16 & 7) < REM; n = min(n, NE - 1); }
          oy = n / half; ox = 2 * (n - oy * half) + par;
          return ok;
        } else {
          const int pb = mt * 16 + r16, pc = min(pb, npxb - 1);
          oy = pc / a.OW1; ox = pc - oy * a.OW1;
          return pb < npxb;
        }
      };
      auto base1 = [&](int oy, int ox) { return ib + ((4 * oy * a.W + 4 * ox) * 3) * 2 + k1g; };
      auto ld1 = [&](const unsigned char* base, u32x4 (&bf)[6]) {
#pragma unroll
        for (int s = 0; s < 6; s++) {
          const int off = (s / 3) * 4 * G::W * 6 + (s % 3) * 16;  // compile-time immediate
#if EF_VAR & 1
          bf[s] = kconst; (void)base;
#else
          const u32x2 lo = *reinterpret_cast<const u32x2*>(base + off);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(base + off + 8);
          bf[s] = u32x4{lo[0], lo[1], hi[0], hi[1]};
#endif
        }
      };
      auto ld1one = [&](const unsigned char* base, int s) -> u32x4 {
        const int off = (s / 3) * 4 * G::W * 6 + (s % 3) * 16;  // compile-time immediate
#if EF_VAR & 1
        (void)base; return kconst;
#else
        const u32x2 lo = *reinterpret_cast<const u32x2*>(base + off);
        const u32x2 hi = *reinterpret_cast<const u32x2*>(base + off + 8);
        return u32x4{lo[0], lo[1], hi[0], hi[1]};
#endif
      };
      // Per-wave tiles i = 0 .. PER1 - 1 (tile w + 4 i); tile i lives in fragment set i & 1.  A tile is two chains over
      // the same six fragments; the second chain refills them in place with the fragments of tile i + 2 (two 8-byte
      // reads per MFMA gap - see conv3), the first chain carries the tile's share of the next image's DMA pieces (after
      // its 2nd and 4th MFMA: a piece stands ~100 clk at issue, in the shadow of a running MFMA part of that is hidden)
      // and, for whole-image geometries, the EPILOGUE OF THE PREVIOUS TILE after its 5th MFMA: that accumulator pair is
      // long past its read-after-MFMA hazard there, and the ~25 VALU/LDS instructions sit in MFMA shadows.
      // DMA pieces per conv1 tile: a wave's DQ_HI (DQ_LO) pieces over the PER1 - 1 (PER1) tiles it certainly has
      constexpr int DPT = !G::WHOLE ? 0 : PER1 > 1 ? cmaxi((DQ_HI + PER1 - 2) / (PER1 - 1), (DQ_LO + PER1 - 1) / PER1) : DQ_HI;
      static_assert(DPT <= 2, "a tile's first chain places at most two DMA pieces");
      constexpr bool OV = G::WHOLE;  // (band geometries: a tile's existence is a run-time fact for every i - plain order)
      f32x4 A0[PER1], A1[PER1];
      int OY[PER1], OX[PER1];
      bool OK[PER1];
#pragma unroll
      for (int i = 0; i < PER1; i++) OK[i] = px1(min(w + 4 * i, nt1 - 1), OY[i], OX[i]);  // (clamped: a tile beyond the last)
      auto epi1 = [&](int i) {
        if (OK[i]) {
          const f32x4 acc0 = A0[i], acc1 = A1[i];
          const int oyi = band * G::BR + OY[i], ox = OX[i];  // row inside the image
          // (ReLU as v_pk_max_i16 on the packed bf16 halves the instruction count and measured 4 % SLOWER: a dependent
          // conversion -> VOP3P pair per register instead of independent v_max_f32 ahead of the conversions)
          const f32x4 r0 = {relu1(acc0[0]), relu1(acc0[1]), relu1(acc0[2]), relu1(acc0[3])};
          const f32x4 r1 = {relu1(acc1[0]), relu1(acc1[1]), relu1(acc1[2]), relu1(acc1[3])};
          const u32x2 lo = pack4_bf16(r0[0], r0[1], r0[2], r0[3]), hi = pack4_bf16(r1[0], r1[1], r1[2], r1[3]);
          const u32x4 pk = {lo[0], lo[1], hi[0], hi[1]};
#if !(EF_X & 2)
          *reinterpret_cast<u32x4*>(act1 + oyi * G::PITCH1 + ox * ACT1_STRIDE + (8 * g) * 2) = pk;  // channels 8 g .. 8 g + 7
#else
          asm volatile("" :: "v"(pk));
#endif
          // saved for the backward as bf16 - the very value the next layer consumes (ReLU mask of the dgrad, im2col operand
          // of the wgrad) - at the start of y1's fp32-sized slot: one 16-byte store, half the bytes of the fp32 copy
          if (P.act)
            *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(P.act) + (long)cur * npx1 * 32 + (unsigned)((oyi * a.OW1 + ox) * 32 + 8 * g)) = pk;  // uniform base + 32-bit lane offset
        }
      };
      u32x4 fa[6], fb[6];
      auto dma_of_tile = [&](int i) {
#if !(EF_X & 1)
        if (G::WHOLE) {
#pragma unroll
          for (int q = 0; q < DPT; q++) lean_piece(i * DPT + q);
        }
#endif
      };
      // the two chains of tile i; prev >= 0: that tile's epilogue rides in the first chain
      auto chains1 = [&](int i, u32x4 (&bf)[6], int prev) {
        const bool do_refill = i + 2 < PER1;
        const unsigned char* refill = base1(OY[do_refill ? i + 2 : i], OX[do_refill ? i + 2 : i]);
        __builtin_amdgcn_sched_barrier(0);
        MFMA_FIRST_AW(A0[i], wc1a[0], bf[0], bias1a);  // each accumulator chain stays strictly back-to-back (asm MFMAs get
#pragma unroll                                           // no compiler hazard handling: interleaving two chains returned wrong sums)
        for (int s = 1; s < 6; s++) {
          MFMA_AW(A0[i], wc1a[s], bf[s]);
#if !(EF_X & 1)
          if (G::WHOLE && (s == 1 || s == 3) && (s >> 1) < DPT) lean_piece(i * DPT + (s >> 1));
#endif
          if (s == 4 && prev >= 0) {
            asm volatile("" : "+v"(A0[prev]), "+v"(A1[prev]));
            epi1(prev);
          }
          if (i == 0 && PER1 > 1) {  // fragment set b = tile 1, six fragments over the five gaps
            const unsigned char* b1 = base1(OY[1], OX[1]);
            if (s == 1) fb[0] = ld1one(b1, 0);
            if (s == 2) { fb[1] = ld1one(b1, 1); fb[2] = ld1one(b1, 2); }
            if (s == 3) fb[3] = ld1one(b1, 3);
            if (s == 4) { fb[4] = ld1one(b1, 4); fb[5] = ld1one(b1, 5); }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        MFMA_FIRST_AW(A1[i], wc1b[0], bf[0], bias1b);
#pragma unroll
        for (int s = 1; s < 6; s++) {
          MFMA_AW(A1[i], wc1b[s], bf[s]);
          if (do_refill) bf[s - 1] = ld1one(refill, s - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (do_refill) bf[5] = ld1one(refill, 5);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto finish1 = [&](int i) {  // a tile whose epilogue no later chain carries
        MFMA_CHAIN_END(A1[i]);
        asm volatile("" : "+v"(A0[i]));  // (its first chain ended six MFMAs earlier)
        epi1(i);
      };
      ld1(base1(OY[0], OX[0]), fa);  // (tile 1's fragments arrive in the gaps of tile 0's first chain)
#pragma unroll
      for (int i = 0; i < PER1; i++) {
        // every wave has tiles 0 .. PER1 - 2 of a whole image; the last one (and any tile of a band) is a run-time fact
        const bool exists = (OV && i + 1 < PER1) || w + 4 * i < nt1;
        if (exists) {
          if (i & 1) chains1(i, fb, OV && i > 0 ? i - 1 : -1); else chains1(i, fa, OV && i > 0 ? i - 1 : -1);
          if (!OV || i + 1 == PER1) finish1(i);
        } else {
          dma_of_tile(i);
          if (OV && i > 0) finish1(i - 1);  // (whole image, no last tile in this wave: its predecessor is still open)
        }
      }
      static_assert(!G::WHOLE || ((PER1 - (NFULL < 4 ? 1 : 0)) * DPT >= DQ_HI && PER1 * DPT >= DQ_LO), "every piece of the next image is issued");
      if (band + 1 < G::NB) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next band has landed
        __syncthreads();                                  // ... everyone's, and everyone is done reading this band
        buf ^= 1;
      }
    }
    STAMP(2);  // conv1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    EF_IMG_BARRIER();
    STAMP(3);  // barrier after conv1

    // ------------------------------------------------ conv2: 4x4 stride 2, 32 -> 64 (wave = 32 channels x half the tiles)
    {
      constexpr int NT2 = (G::NPX2M + 15) >> 4;
      auto base2 = [&](int mt) {
        const int pc = min(mt * 16 + r16, npx2 - 1);
        const int oy = pc / a.OW2, ox = pc - oy * a.OW2;
        return act1 + (2 * oy) * G::PITCH1 + 2 * ox * ACT1_STRIDE + 16 * g;
      };
      auto ld2 = [&](const unsigned char* base, int h, u32x4 (&bf)[8]) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int s = 8 * h + i, ky = s >> 2, kx = s & 3;
#if EF_VAR & 1
          bf[i] = kconst; (void)base;
#else
          bf[i] = *reinterpret_cast<const u32x4*>(base + ky * G::PITCH1 + kx * ACT1_STRIDE);
#endif
        }
      };
      u32x4 fa[8], fb[8];
      // Two accumulator chains per tile (fragments 2 cg and 2 cg + 1) in blocks of 8 back-to-back MFMAs: inside a
      // block every MFMA takes the previous one's result as its C (forwarded); a chain is resumed only after the
      // other chain's 8 MFMAs - far beyond the wait states a non-adjacent dependent MFMA needs (hipcc pads
      // nothing for inline asm).
      auto ld2one = [&](const unsigned char* base, int s) -> u32x4 {
#if EF_VAR & 1
        (void)base; return kconst;
#else
        return *reinterpret_cast<const u32x4*>(base + (s >> 2) * G::PITCH1 + (s & 3) * ACT1_STRIDE);
#endif
      };
      constexpr int PER2 = (NT2 + 1) / 2;  // tiles per wave: ph, ph + 2, ...; with an odd count the last one only for ph == 0
      f32x4 C0[PER2], C1[PER2];
      auto epi2 = [&](int t2) {
        const f32x4 acc0 = C0[t2], acc1 = C1[t2];
        // (store addresses re-derived behind an opaque zero: hoisted out of the image loop they are spilled, and a
        // scratch reload inside the loop waits on vmcnt, i.e. on the next image's DMA)
        int oz;
        asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
        const int pm = (ph + 2 * t2) * 16 + r16 + oz;
        if (pm < npx2) {
          const f32x4 r0 = {relu1(acc0[0]), relu1(acc0[1]), relu1(acc0[2]), relu1(acc0[3])};
          const f32x4 r1 = {relu1(acc1[0]), relu1(acc1[1]), relu1(acc1[2]), relu1(acc1[3])};
          const u32x2 lo = pack4_bf16(r0[0], r0[1], r0[2], r0[3]), hi = pack4_bf16(r1[0], r1[1], r1[2], r1[3]);
          const u32x4 pk = {lo[0], lo[1], hi[0], hi[1]};
          const int oy2 = pm / a.OW2, ox2 = pm - oy2 * a.OW2;
          *reinterpret_cast<u32x4*>(act2 + oy2 * G::PITCH2 + ox2 * G::PX2 + (32 * cg + 8 * g) * 2) = pk;
          if (P.act) {
            // bf16 at the start of y2's slot; wave-uniform base + 32-bit lane offset (a 64-bit per-lane address would be
            // hoisted and spilled)
            __bf16* y = reinterpret_cast<__bf16*>(P.act + P.a_y2) + (long)cur * (npx2 * 64) + (unsigned)(pm * 64 + 32 * cg + 8 * g);
            *reinterpret_cast<u32x4*>(y) = pk;
          }
        }
      };
      auto finish2 = [&](int t2) {
        MFMA_CHAIN_END(C1[t2]);  // (acc0's last MFMA is 8 MFMAs older: covered)
        asm volatile("" : "+v"(C0[t2]));
        epi2(t2);
      };
      if constexpr (G::DEFER) {
        // 5 full tiles: wave (cg, ph) takes tiles ph and ph + 2 whole (both channel tiles: a fragment read feeds two MFMAs)
        // and ONE channel tile - fragment 2 cg + ph - of the last tile, whose reads feed one MFMA each: 80 MFMAs per wave.
        static_assert(!G::DEFER || (NT2 == 5 && PER2 == 3 && G::NPX2M == 80), "leftover-pixel deferral: five full conv2 tiles");
        // the leftover pixel's im2col row - conv1's output over rows / columns 2 (O?2 - 1) .. + 3 - goes to D1 now: read
        // ahead of the first tile's fragment burst, stored behind it (LDS returns in order: the wait is the burst's own)
        // (lane-constant address parts are re-derived behind an opaque zero: hoisted out of the image loop they are spilled,
        // and a scratch reload inside the loop waits on vmcnt, i.e. on the next image's DMA)
        u32x4 d1v;
        const bool d1w = wu == 3;
        int lz;
        asm volatile("v_mov_b32 %0, 0" : "=v"(lz));
        lz += l;
        if (d1w) d1v = *reinterpret_cast<const u32x4*>(act1 + (2 * (G::OH2 - 1) + (lz >> 4)) * G::PITCH1 +
                                                        (2 * (G::OW2 - 1) + ((lz >> 2) & 3)) * ACT1_STRIDE + (lz & 3) * 16);
        const unsigned char* bcur = base2(ph);
        const unsigned char* const bfirst = bcur;
        ld2(bcur, 0, fa);
        if (d1w) *reinterpret_cast<u32x4*>(d1 + dslot * G::D1S + (lz >> 2) * 64 + (lz & 3) * 16) = d1v;
        auto epi2h = [&]() {  // the last tile's half: 4 channels 32 cg + 8 g + 4 ph .. + 3 of pixel 64 + r16
          const f32x4 acc = C0[2];
          int oz;
          asm volatile("s_mov_b32 %0, 0" : "=s"(oz));
          const int pm = (NT2 - 1) * 16 + r16 + oz;
          const int gq = (l + oz) >> 4;
          const u32x2 pk = pack4_bf16(relu1(acc[0]), relu1(acc[1]), relu1(acc[2]), relu1(acc[3]));
          const int oy2 = pm / a.OW2, ox2 = pm - oy2 * a.OW2;
          *reinterpret_cast<u32x2*>(act2 + oy2 * G::PITCH2 + ox2 * G::PX2 + (32 * cg + 8 * gq + 4 * ph) * 2) = pk;
          if (P.act) {
            __bf16* y = reinterpret_cast<__bf16*>(P.act + P.a_y2) + (long)cur * (npx2 * 64) + (unsigned)(pm * 64 + 32 * cg + 8 * gq + 4 * ph);
            *reinterpret_cast<u32x2*>(y) = pk;
          }
        };
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {
          const int mt = ph + 2 * t2;
          bcur = base2(t2 == 0 ? mt + 2 : NT2 - 1);  // tile t2 refills its fragment registers with the next tile's
          __builtin_amdgcn_sched_barrier(0);
          MFMA_FIRST_AW(C0[t2], wc2a[0], fa[0], bias2a);
#pragma unroll
          for (int i = 1; i < 8; i++) {
            MFMA_AW(C0[t2], wc2a[i], fa[i]);
            if (i == 4 && t2 > 0) {
              asm volatile("" : "+v"(C0[t2 - 1]), "+v"(C1[t2 - 1]));
              epi2(t2 - 1);
            }
            if (t2 == 0) fb[i - 1] = ld2one(bfirst, 8 + i - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (t2 == 0) fb[7] = ld2one(bfirst, 15);
          MFMA_FIRST_AW(C1[t2], wc2b[0], fa[0], bias2b);
#pragma unroll
          for (int i = 1; i < 8; i++) {
            MFMA_AW(C1[t2], wc2b[i], fa[i]);
            fa[i - 1] = ld2one(bcur, i - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          MFMA_AW(C0[t2], wc2a[8], fb[0]);
          fa[7] = ld2one(bcur, 7);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 1; i < 8; i++) MFMA_AW(C0[t2], wc2a[8 + i], fb[i]);
          MFMA_AW(C1[t2], wc2b[8], fb[0]);
#pragma unroll
          for (int i = 1; i < 8; i++) {
            MFMA_AW(C1[t2], wc2b[8 + i], fb[i]);
            fb[i - 1] = ld2one(bcur, 8 + i - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
          fb[7] = ld2one(bcur, 15);
          __builtin_amdgcn_sched_barrier(0);
        }
        // the last tile's channel tile 2 cg + ph: one chain of 16, the previous tile's epilogue behind its fifth MFMA (ONE
        // call site between two wave-uniform branches: every site's hoisted address registers count against a file that is full)
        if (ph == 0) {
          MFMA_FIRST_AW(C0[2], wc2a[0], fa[0], bias2a);
#pragma unroll
          for (int i = 1; i < 5; i++) MFMA_AW(C0[2], wc2a[i], fa[i]);
        } else {
          MFMA_FIRST_AW(C0[2], wc2b[0], fa[0], bias2b);
#pragma unroll
          for (int i = 1; i < 5; i++) MFMA_AW(C0[2], wc2b[i], fa[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(C0[1]), "+v"(C1[1]));
        epi2(1);
        __builtin_amdgcn_sched_barrier(0);
        if (ph == 0) {
#pragma unroll
          for (int i = 5; i < 16; i++) { if (i < 8) MFMA_AW(C0[2], wc2a[i], fa[i]); else MFMA_AW(C0[2], wc2a[i], fb[i - 8]); }
        } else {
#pragma unroll
          for (int i = 5; i < 16; i++) { if (i < 8) MFMA_AW(C0[2], wc2b[i], fa[i]); else MFMA_AW(C0[2], wc2b[i], fb[i - 8]); }
        }
        MFMA_CHAIN_END(C0[2]);
        epi2h();
      } else if (ph < NT2) {
        const unsigned char* bcur = base2(ph);
        const unsigned char* const bfirst = bcur;
        ld2(bcur, 0, fa);  // (the first tile's second half arrives in the gaps of its first block)
#pragma unroll
        for (int t2 = 0; t2 < PER2; t2++) {
          const int mt = ph + 2 * t2;
          const bool exists = t2 + 1 < PER2 || NT2 % 2 == 0 || mt < NT2;  // (only an odd count's last tile is a run-time fact)
          if (exists) {
            // the second chain over a fragment set refills it in place with the next tile's fragments, one read per MFMA
            // gap (see conv3); the last tile re-reads its own (a conditional load would push the fragments to scratch)
            bcur = base2(mt + 2 < NT2 ? mt + 2 : mt);
            // (with an even tile count every wave's last tile is known at compile time: no refill there - its stale
            // reads would only delay the next phase's first fragment reads into the same registers)
            const bool refill = NT2 % 2 != 0 || t2 + 1 < NT2 / 2;
            __builtin_amdgcn_sched_barrier(0);
            // first block: the previous tile's epilogue rides behind its fifth MFMA (as in conv1)
            MFMA_FIRST_AW(C0[t2], wc2a[0], fa[0], bias2a);
#pragma unroll
            for (int i = 1; i < 8; i++) {
              MFMA_AW(C0[t2], wc2a[i], fa[i]);
              if (i == 4 && t2 > 0) {
                asm volatile("" : "+v"(C0[t2 - 1]), "+v"(C1[t2 - 1]));
                epi2(t2 - 1);
              }
              if (t2 == 0) fb[i - 1] = ld2one(bfirst, 8 + i - 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (t2 == 0) fb[7] = ld2one(bfirst, 15);
            MFMA_FIRST_AW(C1[t2], wc2b[0], fa[0], bias2b);
#pragma unroll
            for (int i = 1; i < 8; i++) {
              MFMA_AW(C1[t2], wc2b[i], fa[i]);
              if (refill) fa[i - 1] = ld2one(bcur, i - 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            MFMA_AW(C0[t2], wc2a[8], fb[0]);
            if (refill) fa[7] = ld2one(bcur, 7);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 1; i < 8; i++) MFMA_AW(C0[t2], wc2a[8 + i], fb[i]);
            MFMA_AW(C1[t2], wc2b[8], fb[0]);
#pragma unroll
            for (int i = 1; i < 8; i++) {
              MFMA_AW(C1[t2], wc2b[8 + i], fb[i]);
              if (refill) fb[i - 1] = ld2one(bcur, 8 + i - 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (refill) fb[7] = ld2one(bcur, 15);
            __builtin_amdgcn_sched_barrier(0);
            if (t2 + 1 == PER2) finish2(t2);
          } else if (t2 > 0) {
            finish2(t2 - 1);  // (odd tile count, ph == 1: the predecessor is still open)
          }
        }
      }
    }
    STAMP(4);  // conv2
    // Whole-image geometries without saved activations: the next image's DMA pieces (issued during conv1, thousands of
    // clocks ago) are waited for HERE, so that this barrier also publishes "the next image has landed" and the
    // end-of-image barrier can go - the waves then absorb their skew (fc2 runs on two of them) instead of meeting a
    // fourth time per image.  With saved activations the wait would also cover conv2's global stores: old scheme there.
    const bool early_land = G::WHOLE && P.act == nullptr;
#if EF_X & 32  // scratch: nobody waits for the next image's DMA (what would a landing wait that never stalls be worth?)
    if (early_land) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    if (early_land) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    EF_IMG_BARRIER();
    STAMP(5);  // barrier after conv2

    // ------------------------------------------------ conv3: 3x3 stride 1, 64 -> 64 + soft-argmax in registers
    {
      constexpr int NT3 = (G::NPX3M + 15) >> 4;
      const float inv_t = (1.0f / temp) * 1.44269504088896f;  // log2(e) folded in: the soft-argmax exponentials are exp2
      // (G::DEFER) the 3 x 3 x 64 window of the leftover conv3 pixel, minus its last pixel - the leftover conv2 pixel, which
      // the group's conv2 tile writes later - goes to D2: read ahead of the fragment burst, stored behind it
      u32x4 d2v;
      const bool d2w = G::DEFER && wu == 2;
      int lz3 = 0;
      if (G::DEFER) { asm volatile("v_mov_b32 %0, 0" : "=v"(lz3)); lz3 += l; }  // (opaque: see conv2's D1 copy)
      if (d2w) {
        const int j = lz3 >> 3, ky = (j * 11) >> 5, kx = j - 3 * ky;  // pixels 0 .. 7 of the window (j / 3 for j < 8), 16-byte part l & 7
        d2v = *reinterpret_cast<const u32x4*>(act2 + (G::OH3 - 1 + ky) * G::PITCH2 + (G::OW3 - 1 + kx) * G::PX2 + (lz3 & 7) * 16);
      }
      f32x4 v3[NT3];
      float fx[NT3], fy[NT3];
      auto base3 = [&](int mt, int& ox, int& oy) {
        const int pc = min(mt * 16 + r16, npx3 - 1);
        oy = pc / a.OW3; ox = pc - oy * a.OW3;
        return act2 + oy * G::PITCH2 + ox * G::PX2 + 16 * g;
      };
      auto ld3 = [&](const unsigned char* base, int h, u32x4 (&bf)[9]) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
          const int s = 9 * h + i, tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky;
#if EF_VAR & 1
          bf[i] = kconst; (void)base;
#else
          bf[i] = *reinterpret_cast<const u32x4*>(base + ky * G::PITCH2 + kx * G::PX2 + 64 * (s & 1));
#endif
        }
      };
      // Fragment reads INSIDE the MFMA chain: a ds_read_b128 issued while an MFMA is executing is free, the same read
      // issued between two chains costs the lone wave of a SIMD ~16 clk of issue (scratch/micro/mfma16_il: 16.6 vs
      // 24.4 clk per MFMA at conv2's ratio).  So a tile's chain refills its own fragment registers, in place, with the NEXT
      // tile's data: fragment i is overwritten right after MFMA i + 1 has been issued (the chain is dependent, so MFMA
      // i has completed by then), one read per MFMA gap, pinned by scheduling barriers; hipcc still counts the waits.
      u32x4 fa[9], fb[9];
      int ox, oy;
      const unsigned char* bcur = base3(0, ox, oy);
      const unsigned char* const bfirst = bcur;
      ld3(bcur, 0, fa);  // (only the first half in a burst: the first tile's chain fetches its own second half in its gaps)
      if (d2w) *reinterpret_cast<u32x4*>(d2 + dslot * G::D2S + (lz3 >> 3) * 128 + (lz3 & 7) * 16) = d2v;
      auto ld3one = [&](const unsigned char* base, int s) -> u32x4 {
        const int tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky;
#if EF_VAR & 1
        (void)base; return kconst;
#else
        return *reinterpret_cast<const u32x4*>(base + ky * G::PITCH2 + kx * G::PX2 + 64 * (s & 1));
#endif
      };
      // a tile's epilogue (ReLU, temperature scale, the saved-activation store) runs inside the NEXT tile's chain, after its
      // fifth MFMA: the finished accumulator is long past its read-after-MFMA hazard there (no s_nop 11), and the VALU
      // work sits in MFMA shadows instead of between two chains
      f32x4 accs[NT3];
      auto epi3 = [&](int mt) {
        const f32x4 acc = accs[mt];
        const bool ok = mt * 16 + r16 < G::NPX3M;
        int oz3;
        asm volatile("s_mov_b32 %0, 0" : "=s"(oz3));  // (opaque: the store's lane offset is not hoisted out of the image loop as a 64-bit pointer - and spilled)
        if (P.act && ok)
          *reinterpret_cast<f32x4*>(P.act + P.a_y3 + (long)cur * npx3 * 64 + (unsigned)((mt * 16 + r16 + oz3) * 64 + 16 * w + 4 * g)) =
              f32x4{relu1(acc[0]), relu1(acc[1]), relu1(acc[2]), relu1(acc[3])};
#pragma unroll
        for (int q = 0; q < 4; q++)  // (only the last tile has padding lanes)
          v3[mt][q] = (mt * 16 + 15 < G::NPX3M || ok) ? relu1(acc[q]) * inv_t : -INFINITY;
      };
#pragma unroll
      for (int mt = 0; mt < NT3; mt++) {
        fx[mt] = (float)ox; fy[mt] = (float)oy;
        const bool more = mt + 1 < NT3;
        if (more) bcur = base3(mt + 1, ox, oy);
        __builtin_amdgcn_sched_barrier(0);
        MFMA_FIRST_AW(accs[mt], wc3[0], fa[0], bias3);
#pragma unroll
        for (int i = 1; i < 18; i++) {
          if (i < 9) MFMA_AW(accs[mt], wc3[i], fa[i]); else MFMA_AW(accs[mt], wc3[i], fb[i - 9]);
          if (mt == 0 && i <= 9) fb[i - 1] = ld3one(bfirst, 9 + i - 1);
          if (more) { if (i - 1 < 9) fa[i - 1] = ld3one(bcur, i - 1); else fb[i - 10] = ld3one(bcur, i - 1); }
          if (i == 4 && mt > 0) { asm volatile("" : "+v"(accs[mt - 1])); epi3(mt - 1); }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (more) fb[8] = ld3one(bcur, 17);
        __builtin_amdgcn_sched_barrier(0);
      }
      MFMA_CHAIN_END(accs[NT3 - 1]);
      epi3(NT3 - 1);
      STAMP(6);  // conv3 MFMA part
      // pass 1: per-channel max over the image (tiles in registers, then the 16 pixel lanes)
      float mx[4], se[4], sx[4], sy[4];
#if EF_X & 16
#pragma unroll
      for (int q = 0; q < 4; q++) { mx[q] = v3[0][q]; se[q] = v3[1][q]; sx[q] = v3[2][q]; sy[q] = v3[NT3 - 1][q]; }
#else
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float m = v3[0][q];
#pragma unroll
        for (int mt = 1; mt < NT3; mt++) m = fmaxf(m, v3[mt][q]);
        mx[q] = m; se[q] = 0.f; sx[q] = 0.f; sy[q] = 0.f;
      }
      EF_ROW16_4("v_max_f32_dpp", mx);
      // pass 2: exp and the three sums (exp2(-inf) = 0 masks the padded pixels) on packed fp32 pairs (v_pk_add / v_pk_fma:
      // 12 instructions per tile instead of 28 - with contraction off every `s += e * f` was a multiply and an add)
      {
        const f32x2 m01 = {mx[0], mx[1]}, m23 = {mx[2], mx[3]};
        f32x2 se01 = {0.f, 0.f}, se23 = {0.f, 0.f}, sxy[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int mt = 0; mt < NT3; mt++) {
          const f32x2 d01 = f32x2{v3[mt][0], v3[mt][1]} - m01, d23 = f32x2{v3[mt][2], v3[mt][3]} - m23;
          const f32x2 e01 = {__builtin_amdgcn_exp2f(d01[0]), __builtin_amdgcn_exp2f(d01[1])};
          const f32x2 e23 = {__builtin_amdgcn_exp2f(d23[0]), __builtin_amdgcn_exp2f(d23[1])};
          se01 += e01; se23 += e23;
          const f32x2 fxy = {fx[mt], fy[mt]};
          sxy[0] = __builtin_elementwise_fma(f32x2{e01[0], e01[0]}, fxy, sxy[0]);
          sxy[1] = __builtin_elementwise_fma(f32x2{e01[1], e01[1]}, fxy, sxy[1]);
          sxy[2] = __builtin_elementwise_fma(f32x2{e23[0], e23[0]}, fxy, sxy[2]);
          sxy[3] = __builtin_elementwise_fma(f32x2{e23[1], e23[1]}, fxy, sxy[3]);
        }
        se[0] = se01[0]; se[1] = se01[1]; se[2] = se23[0]; se[3] = se23[1];
#pragma unroll
        for (int q = 0; q < 4; q++) { sx[q] = sxy[q][0]; sy[q] = sxy[q][1]; }
      }
      EF_ROW16_4("v_add_f32_dpp", se);
      EF_ROW16_4("v_add_f32_dpp", sx);
      EF_ROW16_4("v_add_f32_dpp", sy);
#endif
      if constexpr (G::DEFER) {
        // running (max, sum exp, sum exp x, sum exp y) of this image's first 48 pixels, channel 16 w + 4 g + q: the
        // leftover pixel joins them when its group's tiles have run
        if (r16 == 0) {
#pragma unroll
          for (int q = 0; q < 4; q++)
            *reinterpret_cast<f32x4*>(ps + dslot * 1024 + (16 * w + 4 * g + q) * 16) = f32x4{mx[q], se[q], sx[q], sy[q]};
        }
      } else
      if (r16 == 0) {
        // features interleaved [x_c, y_c]; channels 16 w + 4 g + q
        float fx_[4], fy_[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float r = __builtin_amdgcn_rcpf(se[q]);
          fx_[q] = sx[q] * r; fy_[q] = sy[q] * r;
        }
        // the lane's 8 features are adjacent: one 16-byte LDS store (and two 16-byte global ones), not eight 2-byte ones
        const u32x2 lo = pack4_bf16(fx_[0], fy_[0], fx_[1], fy_[1]), hi = pack4_bf16(fx_[2], fy_[2], fx_[3], fy_[3]);
        *reinterpret_cast<u32x4*>(sa + slot * SA_STRIDE + 4 * (16 * w + 4 * g)) = u32x4{lo[0], lo[1], hi[0], hi[1]};
        if (P.act) {
          float* f = P.act + P.a_sa + (long)cur * 128 + (unsigned)(2 * (16 * w + 4 * g));  // uniform base + 32-bit lane offset
          *reinterpret_cast<f32x4*>(f) = f32x4{fx_[0], fy_[0], fx_[1], fy_[1]};
          *reinterpret_cast<f32x4*>(f + 4) = f32x4{fx_[2], fy_[2], fx_[3], fy_[3]};
        }
      }
    }

    STAMP(7);  // soft-argmax
    // ------------------------------------------------ the group's leftover pixels: one conv2 tile, one conv3 tile, the merge
    if constexpr (G::DEFER) {
      if (dslot == EF_DCHUNK - 1 || !has_next) {
        int ld_;
        asm volatile("v_mov_b32 %0, 0" : "=v"(ld_));  // (opaque lane index: nothing of this block is hoisted out of the image loop)
        ld_ += l;
        const int r16 = ld_ & 15, g = ld_ >> 4;
        const int nd = dslot + 1, jimg = min(r16, nd - 1);  // images of the group = MFMA columns (clamped: the others' results are dropped)
        const long img0 = cur - (long)dslot * nworkers;      // the group's first image; image j = img0 + j nworkers
        // conv2, pixel NPX2 - 1, channel tile 2 cg + ph of this wave (D1 of this image was stored before the conv2 -> conv3
        // barrier; the earlier images' long before)
        {
          const unsigned char* bA = d1 + jimg * G::D1S + 16 * g;
          u32x4 f[16];
#pragma unroll
          for (int s2 = 0; s2 < 16; s2++) f[s2] = *reinterpret_cast<const u32x4*>(bA + s2 * 64);
          f32x4 acc;
          __builtin_amdgcn_sched_barrier(0);
          if (ph == 0) {
            MFMA_FIRST_AW(acc, wc2a[0], f[0], bias2a);
#pragma unroll
            for (int s2 = 1; s2 < 16; s2++) MFMA_AW(acc, wc2a[s2], f[s2]);
          } else {
            MFMA_FIRST_AW(acc, wc2b[0], f[0], bias2b);
#pragma unroll
            for (int s2 = 1; s2 < 16; s2++) MFMA_AW(acc, wc2b[s2], f[s2]);
          }
          MFMA_CHAIN_END(acc);
          const u32x2 pk = pack4_bf16(relu1(acc[0]), relu1(acc[1]), relu1(acc[2]), relu1(acc[3]));
          if (r16 < nd) {
            *reinterpret_cast<u32x2*>(d2 + r16 * G::D2S + 8 * 128 + (32 * cg + 8 * g + 4 * ph) * 2) = pk;  // window pixel 8
            if (P.act) {
              __bf16* y = reinterpret_cast<__bf16*>(P.act + P.a_y2) + img0 * (npx2 * 64) +
                          (unsigned)(r16 * (unsigned)nworkers * (unsigned)(npx2 * 64) + (npx2 - 1) * 64 + 32 * cg + 8 * g + 4 * ph);
              *reinterpret_cast<u32x2*>(y) = pk;
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // D2 complete (and every wave's soft-argmax state of this image is in LDS)
        // conv3, pixel NPX3 - 1, this wave's 16 channels; then the merge into the state and the features
        {
          const unsigned char* bB = d2 + jimg * G::D2S + 16 * g;
          u32x4 f[18];
#pragma unroll
          for (int s3 = 0; s3 < 18; s3++) f[s3] = *reinterpret_cast<const u32x4*>(bB + (s3 >> 1) * 128 + 64 * (s3 & 1));
          f32x4 st[4];
#pragma unroll
          for (int q = 0; q < 4; q++) st[q] = *reinterpret_cast<const f32x4*>(ps + jimg * 1024 + (16 * w + 4 * g + q) * 16);
          f32x4 acc;
          __builtin_amdgcn_sched_barrier(0);
          MFMA_FIRST_AW(acc, wc3[0], f[0], bias3);
#pragma unroll
          for (int s3 = 1; s3 < 18; s3++) MFMA_AW(acc, wc3[s3], f[s3]);
          MFMA_CHAIN_END(acc);
          const float inv_t = (1.0f / temp) * 1.44269504088896f;
          const f32x4 y3 = {relu1(acc[0]), relu1(acc[1]), relu1(acc[2]), relu1(acc[3])};
          float fx_[4], fy_[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float v = y3[q] * inv_t, m = st[q][0];
            const float M = fmaxf(m, v);
            const float ea = __builtin_amdgcn_exp2f(m - M), eb = __builtin_amdgcn_exp2f(v - M);
            const float se = st[q][1] * ea + eb;
            const float sx = st[q][2] * ea + eb * (float)(G::OW3 - 1), sy = st[q][3] * ea + eb * (float)(G::OH3 - 1);
            const float r = __builtin_amdgcn_rcpf(se);
            fx_[q] = sx * r; fy_[q] = sy * r;
          }
          if (r16 < nd) {
            const u32x2 lo = pack4_bf16(fx_[0], fy_[0], fx_[1], fy_[1]), hi = pack4_bf16(fx_[2], fy_[2], fx_[3], fy_[3]);
            *reinterpret_cast<u32x4*>(sa + (slot - dslot + r16) * SA_STRIDE + 4 * (16 * w + 4 * g)) = u32x4{lo[0], lo[1], hi[0], hi[1]};
            if (P.act) {
              const unsigned jo = (unsigned)r16 * (unsigned)nworkers;  // image offset of this lane's column
              *reinterpret_cast<f32x4*>(P.act + P.a_y3 + img0 * (npx3 * 64) + (unsigned)(jo * (unsigned)(npx3 * 64) + (npx3 - 1) * 64 + 16 * w + 4 * g)) = y3;
              float* fo = P.act + P.a_sa + img0 * 128 + (unsigned)(jo * 128u + 2 * (16 * w + 4 * g));
              *reinterpret_cast<f32x4*>(fo) = f32x4{fx_[0], fy_[0], fx_[1], fy_[1]};
              *reinterpret_cast<f32x4*>(fo + 4) = f32x4{fx_[2], fy_[2], fx_[3], fy_[3]};
            }
          }
        }
      }
    }
    // ------------------------------------------------ FC tail once per chunk
    const bool chunk_done = (slot == EF_CHUNK - 1) || !has_next;
    if (chunk_done) {
      // fc1's 16 weight fragments travel while the slower waves finish their soft-argmax (the conv fragment
      // registers are free here); issued after the barrier their L2 round trip was exposed once per chunk
      u32x4 wf1[4][4];
      int lf;
      asm volatile("v_mov_b32 %0, 0" : "=v"(lf));  // (opaque lane index for the tail's weight / bias addresses: not hoisted out of the image loop)
      lf += tid;
      const int w = lf >> 6, l = lf & 63, g = l >> 4;
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int s = 0; s < 4; s++) wf1[j][s] = P.wpk[WP_F1 + ((4 * w + j) * 4 + s) * 64 + l];
      // ... and so do both layers' biases and fc2's fragments: loaded where they were used, each one exposed an L2 round
      // trip on the chunk's critical path (the tail took ~6 000 clk per chunk for 24 MFMAs)
      f32x4 bf1[4];
#pragma unroll
      for (int j = 0; j < 4; j++) bf1[j] = *reinterpret_cast<const f32x4*>(P.params + po[8] + 16 * (4 * w + j) + 4 * g);
      u32x4 wf2[8];
#pragma unroll
      for (int s = 0; s < 8; s++) wf2[s] = P.wpk[WP_F2 + ((w & 1) * 8 + s) * 64 + l];
      const f32x4 bf2 = *reinterpret_cast<const f32x4*>(P.params + po[10] + 16 * (w & 1) + 4 * g);
      // (LDS-only barriers: __syncthreads() also drains vmcnt, i.e. waits for every global store still in flight)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // soft-argmax features of the whole chunk are in LDS
      const int n_in_chunk = slot + 1;
      {
        const unsigned char* base = sa + min(r16, EF_CHUNK - 1) * SA_STRIDE + 16 * g;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          const u32x4 (&wf)[4] = wf1[j];
          const float c0 = bf1[j][0], c1 = bf1[j][1], c2 = bf1[j][2], c3 = bf1[j][3];
#pragma unroll
          for (int s = 0; s < 4; s++)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wf[s]), as_bf16x8(*reinterpret_cast<const u32x4*>(base + 64 * s)), acc, 0, 0, 0);
          const f32x4 r = {fmaxf(acc[0] + c0, 0.f), fmaxf(acc[1] + c1, 0.f), fmaxf(acc[2] + c2, 0.f), fmaxf(acc[3] + c3, 0.f)};
          if (r16 < EF_CHUNK)
            *reinterpret_cast<u32x2*>(h1 + r16 * H1_STRIDE + (16 * (4 * w + j) + 4 * g) * 2) = pack4_bf16(r[0], r[1], r[2], r[3]);
          if (P.act && r16 < n_in_chunk)
            *reinterpret_cast<f32x4*>(P.act + P.a_h1 + (worker + (long)(it - slot) * nworkers) * 256 +
                                      (unsigned)(r16 * (unsigned)nworkers * 256u + 16 * (4 * w + j) + 4 * g)) = r;  // (32-bit lane offset: < 16 * 256 workers * 1 KB)
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (w < 2) {
        const unsigned char* base = h1 + min(r16, EF_CHUNK - 1) * H1_STRIDE + 16 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float c0 = bf2[0], c1 = bf2[1], c2 = bf2[2], c3 = bf2[3];
#pragma unroll
        for (int s = 0; s < 8; s++)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wf2[s]), as_bf16x8(*reinterpret_cast<const u32x4*>(base + 64 * s)), acc, 0, 0, 0);
        int lq;
        asm volatile("v_mov_b32 %0, 0" : "=v"(lq));  // (opaque lane index: the store's lane offset is not hoisted out of the image loop - and spilled)
        lq += l;
        const int r16 = lq & 15, g = lq >> 4;
        if (r16 < n_in_chunk) {
          float* o = P.out + (worker + (long)(it - slot) * nworkers) * 32 + (unsigned)(r16 * (unsigned)nworkers * 32u + 16 * w + 4 * g);
          f32x4 r = {acc[0] + c0, acc[1] + c1, acc[2] + c2, acc[3] + c3};
          *reinterpret_cast<f32x4*>(o) = r;
        }
      }
    }
    STAMP(8);  // FC tail (every 8th image)
    if (!has_next) {
      // (the last image's conv1 fetched that image once more - lean pieces never branch on has_next: an LDS-DMA write must
      // not still be in flight when the workgroup's LDS is handed to the next one)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      break;
    }
    if (!early_land) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next image has landed
      __syncthreads();                                  // ... and everyone's
    }
    STAMP(9);  // next image landed + end-of-image barrier
    cur = nxt; it++; buf ^= 1;
  }
#if defined(EF_STAMPS) || defined(EF_BLKCLK)
  n_served += it + 1;
#endif
  __syncthreads();  // (a workgroup that goes on with the next problem: nobody still reads what its prologue overwrites)
  }  // problems of this workgroup
#if defined(EF_STAMPS) || defined(EF_BLKCLK)
  if (threadIdx.x == 0 && blockIdx.x < 256) { ef_blk[blockIdx.x] = clock64() - t_entry; ef_blk[256 + blockIdx.x] = n_served; }
#endif
}


#define EF_GEOMS(X) X(84, 84) X(64, 64) X(44, 60) X(128, 128)

extern "C" int tacorl_encoder_fused_supported(int H, int W) {
#define X(h, w) if (H == h && W == w) return EFGeom<h, w>::OK ? 1 : 0;
  EF_GEOMS(X)
#undef X
  return ef_ring_supported(H, W);  // geometries whose conv1 output does not fit the LDS: encoder_ring.hip (no saved activations)
}

template <int H, int W>
static int ef_launch(EFArgs& a, int nb, hipStream_t st) {
  static bool attr_set = false;
  auto kfn = encoder_fused_kernel<H, W>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  constexpr size_t lds_bytes = EFGeom<H, W>::LDS_BYTES;
  hipLaunchKernelGGL(kfn, dim3(nb), dim3(256), lds_bytes, st, a);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

/* What the fused forward writes into a problem's act block for this geometry: 0 - not a fused geometry; 1 - y1 / y2 as bf16 at the
 * start of their slots (what tacorl_encoder_bwd_fused reads); 2 - everything fp32 (what the per-layer tacorl_encoder_bwd reads). */
extern "C" int tacorl_encoder_fused_act_format(int H, int W) {
  if (ef_ring_supported(H, W)) return ebw_supported(H, W) ? 1 : 2;  // (a ring geometry with / without the LDS-resident backward)
  return tacorl_encoder_fused_supported(H, W) ? 1 : 0;
}

extern "C" int tacorl_encoder_fwd_fused(int nprob, const void* const* img, const void* const* packed,
                                        const float* const* params, float* const* out, float* const* act,
                                        const int* n_img, int H, int W, tacorl_stream_t stream) {
  return tacorl_encoder_fwd_fused_wg(nprob, img, packed, params, out, act, n_img, H, W, 0, stream);
}
/* The same launch on at most max_workgroups workgroups (0: one per CU) - fewer than the CU count leaves CUs to a concurrent
 * branch of the caller's graph (TACORL: the update's own encoder problems beside the plan recognition, engine.encode_split). */
extern "C" int tacorl_encoder_fwd_fused_wg(int nprob, const void* const* img, const void* const* packed,
                                           const float* const* params, float* const* out, float* const* act,
                                           const int* n_img, int H, int W, int max_workgroups, tacorl_stream_t stream) {
  if (nprob < 1 || nprob > EF_MAXP || !tacorl_encoder_fused_supported(H, W)) return TACORL_EINVAL;
  if (max_workgroups < 0 || (max_workgroups > 0 && max_workgroups < nprob)) return TACORL_EINVAL;
  EFArgs a{};
  a.nprob = nprob; a.H = H; a.W = W;
  long total = 0;
  for (int p = 0; p < nprob; p++) total += n_img[p];
  if (total == 0) return TACORL_OK;
  // One workgroup per CU (or max_workgroups of them), each owning an equal share of the launch's WORK UNITS (see the kernel):
  // an image costs 64 units, EF_ACT_COST when its activations are also stored; EF_SETUP_COST dead units lie in front of
  // every problem but the first - the weight reload of a workgroup that crosses from one problem into the next.
  const int budget = max_workgroups > 0 && max_workgroups < 256 ? max_workgroups : 256;
  const bool ring = ef_ring_supported(H, W) != 0;
  a.act_bf16 = tacorl_encoder_fused_act_format(H, W) == 1;
  // (ring geometries: an image is ~5 x the work of an 84 x 84 one, the prologue is the same)
  const int setup_cost = ring ? EF_SETUP_COST / 4 : EF_SETUP_COST;
  long units = 0;
  int first = 1;
  for (int p = 0; p < nprob; p++) {
    a.p[p].img = (const __bf16*)img[p]; a.p[p].wpk = (const u32x4*)packed[p]; a.p[p].params = params[p];
    a.p[p].out = out[p]; a.p[p].n_img = n_img[p];
    a.p[p].act = act ? act[p] : nullptr;
    a.p[p].cost = a.p[p].act ? (ring ? (a.act_bf16 ? 84 : 96) : EF_ACT_COST) : 64;  // (ring, 150 x 200: 243 KB of stores per image with bf16 y1 / y2, 406 KB as fp32)
    if (n_img[p] > 0 && !first) units += setup_cost;
    a.p[p].ustart = units;
    units += (long)a.p[p].cost * n_img[p];
    if (n_img[p] > 0) first = 0;
    if (a.p[p].act) {
      long ao[5];
      tacorl_encoder_act_layout(n_img[p], H, W, ao);
      a.p[p].a_y2 = ao[1]; a.p[p].a_y3 = ao[2]; a.p[p].a_sa = ao[3]; a.p[p].a_h1 = ao[4];
    }
  }
  a.utotal = units;
  const int nb = (int)(total < budget ? total : budget);  // (never more workgroups than images)
  if (ring) return ef_ring_launch(a, nb, H, W, (hipStream_t)stream);
#define X(h, w) if (H == h && W == w) return ef_launch<h, w>(a, nb, (hipStream_t)stream);
  EF_GEOMS(X)
#undef X
  return TACORL_EINVAL;
}
