// Convolution backward of the LMPVisionEncoder (conv 8x8/4 -> 4x4/2 -> 3x3/1) as per-image,
// LDS-resident MFMA kernels for gfx950 - the backward twin of encoder_fused.hip.
//
// Replaces, for bf16 compute + bf16 images + a templated camera geometry, the generic-GEMM conv
// backward of tacorl_encoder_bwd (implicit-im2col gather loaders with per-element index arithmetic).
// Reference semantics: autograd of nn.Conv2d/ReLU in networks/vision/lmp_vision_network.py:32-48.
//
//   dgrad  dX[y][x][ci] = sum_{ky,kx,co} dZ[(y-ky)/S][(x-kx)/S][co] W[co][ky][kx][ci], times ReLU mask.
//          Per stride-phase class (y%S, x%S) a dense GEMM  M = class pixels, N = ci, K = taps x co.
//          dZ of one image lives in LDS as NHWC with a zero halo, so every A fragment is one aligned
//          ds_read_b128 at a compile-time offset; W^T fragments stay in registers for the whole kernel.
//   wgrad  dW[co][tap] = sum_pixels dZ[pixel][co] * im2col[pixel][tap].  The reduction index of an MFMA is
//          the lane-contiguous one, and it is the *pixel* here: both operands stay in LDS in their natural NHWC
//          order and the fragments come out of gfx950's transposing read (see ebw_wgrad_tr_kernel; round 1 re-laid
//          both operands as pixel-major planes with 2-byte LDS->LDS moves, 2x slower per kernel).
//          dW accumulates in registers across the images of a workgroup; one partial slab per workgroup,
//          summed in fixed order by ebw_reduce_tr_kernel (deterministic, no atomics).
//   The conv1 / conv2 activations (ReLU mask of the dgrads, im2col operand of the wgrads) arrive as bf16 - the
//   fused forward saves them that way - dZ3 as fp32 (soft-argmax backward), dZ2 / dZ1 as bf16 (written here).
//   The bias gradient rides on the same MFMAs with an all-ones B fragment.
#include "enc_bwd_fused.h"

#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

#ifndef EBW_ROLL_IMG
#define EBW_ROLL_IMG 0  // 1: conv1 weight gradient with its k-steps as a loop - 210 -> 111 registers at 84 x 84 (it then fits beside a ring-GEMM workgroup) but 24.3 -> 27.7 us and the step +9 us (round 5, same-box A/B)
#endif
constexpr int NT = 512;  // 8 waves: 2 per SIMD
constexpr int NW = NT / 64;

constexpr int cmax(int a, int b) { return a > b ? a : b; }
// smallest p >= need with p % 16 == 8: rows 16 B aligned and 16 consecutive rows spread over all LDS banks
constexpr int pitch8(int need) { return (need % 16 <= 8) ? need - need % 16 + 8 : need - need % 16 + 24; }

template <int KH_, int S_, int CI_, int CO_, int IH_, int IW_>
struct ConvL {
  static constexpr int KH = KH_, KW = KH_, S = S_, CI = CI_, CO = CO_, IH = IH_, IW = IW_;
  static constexpr int OH = (IH - KH) / S + 1, OW = (IW - KW) / S + 1;
  static constexpr int PH = (IH + S - 1) / S, PW = (IW + S - 1) / S;  // space-to-depth plane
  static constexpr int KA = KH / S, KB = KW / S;                      // taps per stride-phase class
  static constexpr int TAPS = KH * KW * CI;
  static constexpr int NPL = CI * S * S;
  // wgrad planes: row pitch PWP (multiple of 8) and one copy per horizontal tap offset b = kx / S, copy b
  // holding plane[Y][X + b] at [Y][X] - every B fragment (8 pixels of one tap) is then a 16-byte
  // ALIGNED ds_read_b128 (2-byte aligned reads of a single copy measured 2.3x slower per kernel)
  static constexpr int PWP = (PW + 7) / 8 * 8;
  static constexpr int KQ = (OH * PWP + 31) / 32 * 32;  // padded pixel-reduction length, q = oy*PWP + ox
  static constexpr int PLP = pitch8(cmax(PH * PWP, (KA - 1) * PWP + KQ));  // plane pitch (elements)
  static constexpr int NCOPY = KB;
  static constexpr int DZP = KQ + 8;  // dZ plane pitch
  static constexpr int MT = CO / 16, NTL = TAPS / 16;
  static constexpr int SLABF = CO * TAPS + CO;  // floats per partial slab (dW | db)
  // dgrad
  // PP: dZ pixel pitch in the dgrad's halo image: 160 B for CO = 64 - with ds_read_b128's lane groups ({0-3,12-15,20-27}, ...)
  // the 16 (pixel, k-group) pairs of a group then hit 16 distinct 16-byte slots; CO + 8 (144 B) cost 59 % conflict cycles
  // C: pixels per halo row = PW + 8: a 16-pixel tile spans two or three rows, and with PP = 80 (160 B) the 16 lanes of a
  // ds_read_b128 group hit 16 distinct 16-byte slots only if the row-to-row gap is a multiple of 8 pixels (PW + HB, the
  // minimum, costs 7.3 - 7.4 LDS cycles per read instead of 4: scratch/micro/bank_sim.py; the dgrads are LDS-bound)
  static constexpr int HA = KA - 1, HB = KB - 1, R = PH + HA, C = PW + 8, PP = CO + 16;
  static_assert(HB <= 8, "halo columns");
  static constexpr int KS = KA * KB * CO / 32;
  static constexpr int NCLS = S * S, NTI = CI / 16;
  static constexpr int MTC = (PH * PW + 15) / 16;
  static constexpr int NCOMBO = NCLS * NTI;
  static_assert(TAPS % 16 == 0 && CO % 32 == 0 && KH % S == 0, "tile shapes");
};

template <int H, int W>
struct Geo {
  using L1 = ConvL<8, 4, 3, 32, H, W>;
  using L2 = ConvL<4, 2, 32, 64, L1::OH, L1::OW>;
  using L3 = ConvL<3, 1, 64, 64, L2::OH, L2::OW>;
};

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  const __bf16 x = (__bf16)a, y = (__bf16)b;
  return (uint32_t)__builtin_bit_cast(unsigned short, x) | ((uint32_t)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
__device__ __forceinline__ void load8(const __bf16* p, float (&v)[8]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = (float)a[j];
}
__device__ __forceinline__ bf16x8 load8v(const float* p) {
  float v[8];
  load8(p, v);
  bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] = (__bf16)v[j];
  return r;
}
__device__ __forceinline__ bf16x8 load8v(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// ===================================================================== wgrad
struct WgArgs {
  const void* in[EBW_MAXP];
  const void* dz[EBW_MAXP];
  int n[EBW_MAXP];
  float* slab;
  int wpp;  // workgroups per problem
};

// ===================================================================== wgrad, transposing LDS reads
// Same contraction as ebw_wgrad_kernel - dW[co][tap] = sum_pix dZ[pix][co] * im2col[pix][tap] - without any
// re-layout of the operands: dZ and the layer input sit in LDS in their natural NHWC order (a coalesced 16-byte
// copy with a padded pixel pitch; the bf16 image verbatim) and the MFMA fragments, whose reduction index is the
// PIXEL, come out of gfx950's transposing read (ds_read_b64_tr_b16: a 16-lane group reads 4 rows x 16 columns of
// 16-bit elements and each lane receives one COLUMN; every lane supplies the address of its own row, so the "rows" of
// the im2col operand are simply the input pixels of four consecutive output pixels).  One read gives a lane 4 of its
// 8 k-values; the pixel order inside a 32-pixel k-step is permuted identically for both operands (group g takes pixels
// 4g..4g+3 and 16+4g..16+4g+3), which a sum does not see, so that a 32-lane half reads 8 CONSECUTIVE pixels whose
// padded pitches (160 B) spread over all 64 banks.
// Replaces: per image ~700 VALU + 180 LDS instructions per wave of plane building (2-byte LDS->LDS moves, 57 % bank
// conflict cycles) in front of 60 MFMAs.
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
#define LDS_BF4(p) ((__attribute__((address_space(3))) bf16x4v*)(p))
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* lo, const __bf16* hi) {
  const bf16x4v a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_BF4(lo));
  const bf16x4v b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_BF4(hi));
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// wave tiling of a layer: MGRP x NGRP x KGRP = 8 waves; a wave owns MT/MGRP m-tiles (16 output channels each) x
// NTL/NGRP n-tiles (16 taps each) and the k-steps s with s % KGRP == its k group
template <class L> struct TrTile;
template <int H, int W> struct TrTile<ConvL<3, 1, 64, 64, H, W>> { static constexpr int MGRP = 2, NGRP = 4, KGRP = 1, NBUF = 2, CIP = 80; };
template <int H, int W> struct TrTile<ConvL<4, 2, 32, 64, H, W>> { static constexpr int MGRP = 2, NGRP = 4, KGRP = 1, NBUF = 2, CIP = 40; };
template <int H, int W> struct TrTile<ConvL<8, 4, 3, 32, H, W>> { static constexpr int MGRP = 1, NGRP = 4, KGRP = 2, NBUF = 1, CIP = 3; };

template <class L>
struct TrGeo {
  using T = TrTile<L>;
  static constexpr bool IMG = L::CI < 8;
  static constexpr int DZR = L::CO + 16;                       // dZ row pitch (elements): 160 B / 96 B
  static constexpr int CIP = T::CIP;                           // input pixel pitch (elements)
  static constexpr int NPIX_ALL = L::OH * L::OW;
  // conv1 of a large camera (128 x 128: 95 KB of dZ1 + the 98 KB image) does not fit LDS even single-buffered: the image
  // is then processed in NBAND bands of BR output rows = (BR - 1) S + KH input rows; dW simply keeps accumulating
  // Round 6: any layer may be banded, in up to 4 bands (150 x 200: conv1 in 3 - 58 KB of dZ1 + 62 KB of image rows per band -,
  // conv2 in 2 - 36 KB of dZ2 + 78 KB of conv1-output rows; conv3 fits whole): the smallest band count whose single buffer fits.
  static constexpr size_t bytes_for(int nband) {
    const int br = (L::OH + nband - 1) / nband, rows = nband == 1 ? L::IH : (br - 1) * L::S + L::KH;
    return ((size_t)((br * L::OW + 31) / 32 * 32) * DZR + (size_t)(rows * L::IW * CIP + 7) / 8 * 8) * 2;
  }
  static constexpr size_t WHOLE_BYTES = bytes_for(1);
  static constexpr int NBAND = bytes_for(1) <= 160 * 1024 ? 1 : bytes_for(2) <= 160 * 1024 ? 2 : bytes_for(3) <= 160 * 1024 ? 3 : 4;
  static constexpr int BR = (L::OH + NBAND - 1) / NBAND;       // output rows per band (the last band may have fewer)
  static constexpr int BROWS = NBAND == 1 ? L::IH : (BR - 1) * L::S + L::KH;  // input rows staged per band
  static constexpr int NPIX = BR * L::OW, KS = (NPIX + 31) / 32, KQ = KS * 32;
  static constexpr int DZ_EL = KQ * DZR;
  static constexpr int IN_EL = (BROWS * L::IW * CIP + 7) / 8 * 8;
  static constexpr int BUF_EL = DZ_EL + IN_EL;
  static constexpr int NBUF = (T::NBUF == 2 && (size_t)2 * BUF_EL * 2 <= 160 * 1024 && NBAND == 1) ? 2 : 1;
  static constexpr int MPW = L::MT / T::MGRP, NPW = L::NTL / T::NGRP, TPW = MPW * NPW;
  static constexpr size_t lds_bytes = (size_t)NBUF * BUF_EL * 2;
  static constexpr int SLABF = T::KGRP * (L::CO * L::TAPS + L::CO);  // floats per workgroup slab
  static_assert(L::MT % T::MGRP == 0 && L::NTL % T::NGRP == 0 && T::MGRP * T::NGRP * T::KGRP == NW, "wave tiling");
  static_assert(lds_bytes <= 160 * 1024, "wgrad tile does not fit LDS");
};

template <class L, class InT, class DzT>
__global__ __launch_bounds__(NT) void ebw_wgrad_tr_kernel(WgArgs a) {
  using G = TrGeo<L>;
  using T = TrTile<L>;
  constexpr bool IMG = L::CI < 8;  // conv1: the bf16 NHWC image (3 channels), copied verbatim
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* lds = reinterpret_cast<__bf16*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, l16 = lane & 15, q = l16 >> 2, pc = l16 & 3;
  const int p = blockIdx.x / a.wpp, j0 = blockIdx.x - p * a.wpp;
  const int n_img = a.n[p];
  const int ng = w % T::NGRP, mg = (w / T::NGRP) % T::MGRP, kg = w / (T::NGRP * T::MGRP);
  // zero once: the dZ rows of the padded pixels (never written again) - and everything else, so that no lane ever
  // feeds an MFMA from uninitialised LDS
  for (int e = tid; e < (int)(G::lds_bytes / 4); e += NT) reinterpret_cast<uint32_t*>(lds)[e] = 0u;
  // per-lane tap offsets of the wave's n-tiles (elements, relative to the pixel's first input element)
  int toff[G::NPW];
#pragma unroll
  for (int j = 0; j < G::NPW; j++) {
    const int nt = ng * G::NPW + j;
    if constexpr (IMG) {
      const int tap = 16 * nt + 4 * pc, ky = tap / (L::KW * L::CI), rem = tap - ky * (L::KW * L::CI);
      toff[j] = ky * L::IW * L::CI + rem;
    } else {
      constexpr int CPT16 = L::CI / 16;
      const int kk = nt / CPT16, c0 = 16 * (nt - kk * CPT16), ky = kk / L::KW, kx = kk - ky * L::KW;
      toff[j] = (ky * L::IW + kx) * G::CIP + c0 + 4 * pc;
    }
  }
  f32x4 acc[G::MPW][G::NPW], bacc[G::MPW];
#pragma unroll
  for (int i = 0; i < G::MPW; i++) {
    bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < G::NPW; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; j++) ones[j] = (__bf16)1.0f;
  const InT* in = reinterpret_cast<const InT*>(a.in[p]);
  const DzT* dz = reinterpret_cast<const DzT*>(a.dz[p]);
  // ---- staging: 8-element chunks, global -> registers (in flight during the MFMAs) -> LDS
  constexpr int DCG = L::CO / 8, DCH = G::NPIX * DCG, DCPT = (DCH + NT - 1) / NT;
  constexpr int ICH = G::BROWS * L::IW * L::CI / 8, ICPT = (ICH + NT - 1) / NT;  // (one band's input rows; BROWS = IH without bands)
  static_assert((G::BROWS * L::IW * L::CI) % 8 == 0 && (G::NBAND == 1 || (G::BR * L::S * L::IW * L::CI) % 8 == 0), "image / band bytes");
  bf16x8 dpre[DCPT], ipre[ICPT];
  // band b of an image: output rows [b BR, ...), dZ chunks [0, dch), input chunks [0, ich) from input row b BR S
  auto band_rows = [&](int b) { return G::NBAND == 1 ? L::OH : (L::OH - b * G::BR < G::BR ? L::OH - b * G::BR : G::BR); };
  auto fetch = [&](int img, int b) {
    const int dch = G::NBAND == 1 ? DCH : band_rows(b) * L::OW * DCG;
    const DzT* d = dz + ((long)img * G::NPIX_ALL + (long)b * G::BR * L::OW) * L::CO;
#pragma unroll
    for (int r = 0; r < DCPT; r++) {
      const int c = tid + r * NT;
      dpre[r] = load8v(d + 8 * (c < dch ? c : dch - 1));  // unconditional (clamped): keeps dpre in registers
    }
    const int ich = G::NBAND == 1 ? ICH : ((band_rows(b) - 1) * L::S + L::KH) * L::IW * L::CI / 8;
    const InT* s = in + (long)img * L::IH * L::IW * L::CI + (long)b * G::BR * L::S * L::IW * L::CI;
#pragma unroll
    for (int r = 0; r < ICPT; r++) {
      const int c = tid + r * NT;
      ipre[r] = load8v(s + 8 * (c < ich ? c : ich - 1));
    }
  };
  auto put = [&](__bf16* buf, int b) {
    const int dch = G::NBAND == 1 ? DCH : band_rows(b) * L::OW * DCG;
#pragma unroll
    for (int r = 0; r < DCPT; r++) {
      const int c = tid + r * NT;
      if (c < dch) *reinterpret_cast<bf16x8*>(buf + (c / DCG) * G::DZR + 8 * (c % DCG)) = dpre[r];
      else if (G::NBAND > 1 && c < DCH) *reinterpret_cast<bf16x8*>(buf + (c / DCG) * G::DZR + 8 * (c % DCG)) = bf16x8{};  // a shorter band: its missing rows are zero
    }
    __bf16* ib = buf + G::DZ_EL;
    const int ich = G::NBAND == 1 ? ICH : ((band_rows(b) - 1) * L::S + L::KH) * L::IW * L::CI / 8;
#pragma unroll
    for (int r = 0; r < ICPT; r++) {
      const int c = tid + r * NT;
      if (c < ich) {
        if constexpr (IMG) *reinterpret_cast<bf16x8*>(ib + 8 * c) = ipre[r];
        else *reinterpret_cast<bf16x8*>(ib + (c / (L::CI / 8)) * G::CIP + 8 * (c % (L::CI / 8))) = ipre[r];
      }
    }
  };
  auto compute = [&](const __bf16* buf, int b) {
    const __bf16* dzs = buf;
    const __bf16* ins = buf + G::DZ_EL;
    const int npix = G::NBAND == 1 ? G::NPIX : band_rows(b) * L::OW;
    auto kstep = [&](int s) {
      if (T::KGRP > 1 && (s % T::KGRP) != kg) return;
      // this lane's two pixels of the k-step (dZ rows of padded pixels are zero; their input address is clamped)
      const int P0 = 32 * s + 4 * g + q, P1 = P0 + 16;
      const int Q0 = P0 < npix ? P0 : npix - 1, Q1 = P1 < npix ? P1 : npix - 1;
      const int b0 = ((Q0 / L::OW) * L::S * L::IW + (Q0 % L::OW) * L::S) * G::CIP;
      const int b1 = ((Q1 / L::OW) * L::S * L::IW + (Q1 % L::OW) * L::S) * G::CIP;
      bf16x8 A[G::MPW];
#pragma unroll
      for (int i = 0; i < G::MPW; i++) {
        const int col = 16 * (mg * G::MPW + i) + 4 * pc;
        A[i] = tr_frag(dzs + P0 * G::DZR + col, dzs + P1 * G::DZR + col);
      }
#pragma unroll
      for (int j = 0; j < G::NPW; j++) {
        const bf16x8 B = tr_frag(ins + b0 + toff[j], ins + b1 + toff[j]);
#pragma unroll
        for (int i = 0; i < G::MPW; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B, acc[i][j], 0, 0, 0);
      }
      if (ng == 0) {
#pragma unroll
        for (int i = 0; i < G::MPW; i++) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], ones, bacc[i], 0, 0, 0);
      }
    };
    // (fully unrolled the k-steps hoist their pixel -> address arithmetic: at 7 steps - conv2 of a 128 x 128 camera, 196
    // output pixels - beside 48 staging registers that was 256 registers + 54 spilled ones reloaded per image; round 5)
    if constexpr (G::KS > 4 && (!IMG || EBW_ROLL_IMG || G::KS > 16)) {  // (conv1 of 150 x 200 in 3 bands: 19 k-steps - unrolled they spill 74 registers)
#pragma unroll 1
      for (int s = 0; s < G::KS; s++) kstep(s);
    } else {
#pragma unroll
      for (int s = 0; s < G::KS; s++) kstep(s);
    }
  };
  __syncthreads();  // zero fill done
  if constexpr (G::NBUF == 2) {
    if (j0 < n_img) { fetch(j0, 0); put(lds, 0); }
    __syncthreads();
    int k = 0;
    for (int img = j0; img < n_img; img += a.wpp, k ^= 1) {
      const int nxt = img + a.wpp;
      if (nxt < n_img) fetch(nxt, 0);
      compute(lds + k * G::BUF_EL, 0);
      if (nxt < n_img) put(lds + (k ^ 1) * G::BUF_EL, 0);
      __syncthreads();
    }
  } else {
    if (j0 < n_img) fetch(j0, 0);
    for (int img = j0; img < n_img; img += a.wpp) {
#pragma unroll 1
      for (int b = 0; b < G::NBAND; b++) {
        __syncthreads();  // the previous unit's fragments are consumed
        put(lds, b);
        __syncthreads();
        if (b + 1 < G::NBAND) fetch(img, b + 1);
        else if (img + a.wpp < n_img) fetch(img + a.wpp, 0);
        compute(lds, b);
      }
    }
  }
  // partial slab in accumulator order: dW tiles [wave][mi][nj][lane][4], then db [KGRP][CO]
  float* sl = a.slab + (long)blockIdx.x * G::SLABF;
#pragma unroll
  for (int i = 0; i < G::MPW; i++)
#pragma unroll
    for (int j = 0; j < G::NPW; j++)
      *reinterpret_cast<f32x4*>(sl + (((w * G::MPW + i) * G::NPW + j) * 64 + lane) * 4) = acc[i][j];
  if (ng == 0 && l16 == 0) {
#pragma unroll
    for (int i = 0; i < G::MPW; i++)
#pragma unroll
      for (int r = 0; r < 4; r++)
        sl[T::KGRP * L::CO * L::TAPS + kg * L::CO + 16 * (mg * G::MPW + i) + 4 * g + r] = bacc[i][r];
  }
}

// Fixed-order sum of the transposing-read kernels' slabs (all three layers, all problems, one launch): one thread per
// OUTPUT element, over the k groups and then the workgroups.
struct RdTrArgs {
  const float* slab[3];
  float* gw[3][EBW_MAXP];
  float* gb[3][EBW_MAXP];
  int co[3], taps[3], mpw[3], npw[3], mgrp[3], ngrp[3], kgrp[3], slabf[3];
  int ileave[3];  // n-tiles interleaved over the wave groups (ebw_l3_kernel) instead of blocked
  int wpp, accumulate;
  // fused conv3 launch: its per-workgroup temperature-gradient partials are summed here (one otherwise idle block)
  const float* dtp[EBW_MAXP];
  float* g_temp[EBW_MAXP];
  int ndt[EBW_MAXP];  // partials per problem (0 = no temperature sum in this launch)
};
__global__ __launch_bounds__(256) void ebw_reduce_tr_kernel(RdTrArgs a) {
  // slab order (coalesced reads): 64 consecutive slab floats per block, the 4 waves take every 4th workgroup slab,
  // partial sums added 0..3 (fixed order); the k groups of a layer are summed by the same thread
  __shared__ float sh[4][64];
  const int l = blockIdx.y % 3, p = blockIdx.y / 3;
  const int nW = a.co[l] * a.taps[l], per = nW + a.co[l], KG = a.kgrp[l];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), kq = threadIdx.x >> 6;
  if (a.ndt[p] > 0 && l == 0 && blockIdx.x == gridDim.x - 1) {  // (beyond conv1's elements: the host checks)
    float s = 0.f;
    for (int i = threadIdx.x; i < a.ndt[p]; i += 256) s += a.dtp[p][i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[0][kq] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float v = ((sh[0][0] + sh[0][1]) + sh[0][2]) + sh[0][3];
      *a.g_temp[p] = a.accumulate ? *a.g_temp[p] + v : v;
    }
    return;
  }
  if (blockIdx.x * 64 >= per) return;
  float sum = 0.f;
  if (e < per) {
    const float* s0 = a.slab[l] + (long)p * a.wpp * a.slabf[l];
    for (int kg = 0; kg < KG; kg++) {
      // element e of k group kg: dW tiles of the group's waves are contiguous (wave index = kg * MGRP * NGRP + ...)
      const float* s = s0 + (e < nW ? (long)kg * nW + e : (long)KG * nW + kg * a.co[l] + (e - nW));
#pragma unroll 8
      for (int k = kq; k < a.wpp; k += 4) sum += s[(long)k * a.slabf[l]];
    }
  }
  sh[kq][threadIdx.x & 63] = sum;
  __syncthreads();
  if (kq == 0 && e < per) {
    const int t = threadIdx.x;
    const float v = ((sh[0][t] + sh[1][t]) + sh[2][t]) + sh[3][t];
    float* o;
    if (e < nW) {  // slab element (wave-in-group, mi, nj, lane, r) -> dW[co][tap]
      const int r = e & 3, lane = (e >> 2) & 63, tt = e >> 8;
      const int nj = tt % a.npw[l], mi = (tt / a.npw[l]) % a.mpw[l], wg = tt / (a.npw[l] * a.mpw[l]);
      const int ng = wg % a.ngrp[l], mg = wg / a.ngrp[l];
      const int co = 16 * (mg * a.mpw[l] + mi) + 4 * (lane >> 4) + r;
      const int tap = 16 * (a.ileave[l] ? nj * a.ngrp[l] + ng : ng * a.npw[l] + nj) + (lane & 15);
      o = a.gw[l][p] + (long)co * a.taps[l] + tap;
    } else {
      o = a.gb[l][p] + (e - nW);
    }
    *o = a.accumulate ? *o + v : v;
  }
}

// ===================================================================== dgrad
struct DgArgs {
  const void* dz[EBW_MAXP];    // [n][OH*OW][CO]
  const __bf16* yin[EBW_MAXP];  // [n][IH*IW][CI] layer input (post-ReLU, bf16 as the fused forward saved it): mask
  const uint4* wpk[EBW_MAXP];  // packed W^T fragments
  __bf16* dx[EBW_MAXP];        // [n][IH*IW][CI] masked input gradient = dZ of the previous layer
  int n[EBW_MAXP];
  int wpp;
};
struct PkArgs {
  const float* w[EBW_MAXP];
  uint4* out[EBW_MAXP];
};

// wpk[(combo*KS + ks)*64 + lane]: combo = class*NTI + ntile; the lane's 8 consecutive co of
// W[co][py + S*a][px + S*b][16*ntile + (lane&15)], ks = (a*KB + b)*(CO/32) + half.
template <class L>
__global__ __launch_bounds__(256) void ebw_pack_kernel(PkArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= L::NCOMBO * L::KS * 64) return;
  const int lane = idx & 63, ks = (idx >> 6) % L::KS, combo = (idx >> 6) / L::KS;
  const int cls = combo / L::NTI, nt = combo % L::NTI, py = cls / L::S, px = cls % L::S;
  const int tp = ks / (L::CO / 32), hf = ks % (L::CO / 32), ta = tp / L::KB, tb = tp % L::KB;
  const int ky = py + L::S * ta, kx = px + L::S * tb, ci = 16 * nt + (lane & 15), co0 = 32 * hf + 8 * (lane >> 4);
  const float* w = a.w[blockIdx.y];
  uint32_t v[4];
#pragma unroll
  for (int e = 0; e < 4; e++)
    v[e] = pack2(w[(((co0 + 2 * e) * L::KH + ky) * L::KW + kx) * L::CI + ci],
                 w[(((co0 + 2 * e + 1) * L::KH + ky) * L::KW + kx) * L::CI + ci]);
  a.out[blockIdx.y][idx] = make_uint4(v[0], v[1], v[2], v[3]);
}

// both dgrads' fragments in one launch (they were two 3 - 6 us launches on the step's chain): blocks [0, B2) pack L2's
template <class LA, class LB>
__global__ __launch_bounds__(256) void ebw_pack2_kernel(PkArgs a, PkArgs b) {
  constexpr int BA = (LA::NCOMBO * LA::KS * 64 + 255) / 256;
  const bool first = (int)blockIdx.x < BA;
  const int idx = (first ? blockIdx.x : blockIdx.x - BA) * 256 + threadIdx.x;
  auto body = [&](auto tag, const PkArgs& k) {
    using L = decltype(tag);
    if (idx >= L::NCOMBO * L::KS * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) % L::KS, combo = (idx >> 6) / L::KS;
    const int cls = combo / L::NTI, nt = combo % L::NTI, py = cls / L::S, px = cls % L::S;
    const int tp = ks / (L::CO / 32), hf = ks % (L::CO / 32), ta = tp / L::KB, tb = tp % L::KB;
    const int ky = py + L::S * ta, kx = px + L::S * tb, ci = 16 * nt + (lane & 15), co0 = 32 * hf + 8 * (lane >> 4);
    const float* w = k.w[blockIdx.y];
    uint32_t v[4];
#pragma unroll
    for (int e = 0; e < 4; e++)
      v[e] = pack2(w[(((co0 + 2 * e) * L::KH + ky) * L::KW + kx) * L::CI + ci],
                   w[(((co0 + 2 * e + 1) * L::KH + ky) * L::KW + kx) * L::CI + ci]);
    k.out[blockIdx.y][idx] = make_uint4(v[0], v[1], v[2], v[3]);
  };
  if (first) body(LA{}, a); else body(LB{}, b);
}

template <class L>
constexpr int dgrad_mask_bytes() { return (L::IH * L::IW * L::CI / 8 + 15) / 16 * 16; }
// two buffers (the next image's dZ / mask are put while this one is computed) where they fit, one otherwise (150 x 200: the
// halo image of one conv2 / conv3 gradient is 100 / 94 KB)
template <class L>
constexpr int dgrad_nbuf() { return (size_t)2 * L::R * L::C * L::PP * 2 + 2 * dgrad_mask_bytes<L>() <= 160 * 1024 ? 2 : 1; }
template <class L>
constexpr size_t dgrad_lds_bytes() { return (size_t)dgrad_nbuf<L>() * (L::R * L::C * L::PP * 2 + dgrad_mask_bytes<L>()); }

template <class L, class DzT>
__global__ __launch_bounds__(NT) void ebw_dgrad_kernel(DgArgs a) {
  constexpr int MPARTS = NW / L::NCOMBO, MTW = (L::MTC + MPARTS - 1) / MPARTS;
  static_assert(NW % L::NCOMBO == 0, "wave tiling");
  constexpr int BUF = L::R * L::C * L::PP, NB = dgrad_nbuf<L>();
  constexpr int NCG = L::CO / 8, NCH = L::OH * L::OW * NCG, CPT = (NCH + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* dzh = reinterpret_cast<__bf16*>(smem);  // 2 x [R][C][PP], zero halo
  // ReLU mask of the layer input, one bit per element ([pixel][ci] order), 2 buffers: the fp32
  // activations are read once, coalesced, by the staging threads instead of 4 B gathers in the epilogue
  constexpr int MKB = dgrad_mask_bytes<L>(), NYC = L::IH * L::IW * L::CI / 8, CPY = (NYC + NT - 1) / NT;
  static_assert(L::CI % 8 == 0, "mask bytes hold 8 channels of one pixel");
  unsigned char* mkb = smem + (size_t)NB * BUF * 2;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, g = lane >> 4;
  const int p = blockIdx.x / a.wpp, j0 = blockIdx.x - p * a.wpp;
  const int n_img = a.n[p];
  for (int e = tid; e < NB * BUF / 2; e += NT) reinterpret_cast<uint32_t*>(dzh)[e] = 0u;
  const int combo = w % L::NCOMBO, mpart = w / L::NCOMBO, cls = combo / L::NTI, nt = combo % L::NTI;
  const int py = cls / L::S, px = cls % L::S;
  bf16x8 Bf[L::KS];
#pragma unroll
  for (int ks = 0; ks < L::KS; ks++) {
    const uint4 t = a.wpk[p][(combo * L::KS + ks) * 64 + lane];
    Bf[ks] = __builtin_bit_cast(bf16x8, t);
  }
  int abase[MTW];
#pragma unroll
  for (int t = 0; t < MTW; t++) {
    int m = 16 * (mpart + t * MPARTS) + i;
    m = m < L::PH * L::PW ? m : L::PH * L::PW - 1;
    const int Y = m / L::PW, X = m - Y * L::PW;
    abase[t] = (Y * L::C + X) * L::PP + 8 * g;  // halo pixel of tap (KA-1, KB-1): every tap offset below is >= 0 (an immediate)
  }
  // W^T is the MFMA A operand (D rows = input channels): a lane ends with 4 CONSECUTIVE channels of one pixel,
  // i.e. one 8-byte store and one mask nibble per tile instead of four 2-byte stores and four mask bytes.
  // First output element (relative to the image) of tile t's accumulator, -1 = padding:
  int oidx[MTW];
#pragma unroll
  for (int t = 0; t < MTW; t++) {
    const int m = 16 * (mpart + t * MPARTS) + i, Y = m / L::PW, X = m - Y * L::PW;
    const int y = L::S * Y + py, x = L::S * X + px;
    oidx[t] = (m < L::PH * L::PW && y < L::IH && x < L::IW) ? (y * L::IW + x) * L::CI + 16 * nt + 4 * g : -1;
  }
  const DzT* dz = reinterpret_cast<const DzT*>(a.dz[p]);
  const __bf16* yin = a.yin[p];
  __bf16* dx = a.dx[p];
  bf16x8 pre[CPT];
  uint4 ypre[CPY];  // 8 bf16 activations per chunk, raw bits
  auto fetch = [&](int img) {
#pragma unroll
    for (int r = 0; r < CPT; r++) {
      const int c = tid + r * NT;
      if (c < NCH) pre[r] = load8v(dz + ((long)img * L::OH * L::OW + c / NCG) * L::CO + 8 * (c % NCG));
    }
#pragma unroll
    for (int r = 0; r < CPY; r++) {
      const int c = tid + r * NT;
      ypre[r] = *reinterpret_cast<const uint4*>(yin + (long)img * L::IH * L::IW * L::CI + 8 * (c < NYC ? c : NYC - 1));
    }
  };
  auto put = [&](__bf16* buf, unsigned char* mb) {
#pragma unroll
    for (int r = 0; r < CPY; r++) {
      const int c = tid + r * NT;
      if (c < NYC) {
        // y > 0 on the bf16 bit pattern: 0x0001 .. 0x7fff (post-ReLU values: no negatives, no NaN)
        const uint32_t d[4] = {ypre[r].x, ypre[r].y, ypre[r].z, ypre[r].w};
        unsigned b = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          b |= ((((d[j] & 0xffffu) - 1u) & 0xffffu) < 0x7fffu ? 1u : 0u) << (2 * j);
          b |= ((((d[j] >> 16) - 1u) & 0xffffu) < 0x7fffu ? 1u : 0u) << (2 * j + 1);
        }
        mb[c] = (unsigned char)b;
      }
    }
#pragma unroll
    for (int r = 0; r < CPT; r++) {
      const int c = tid + r * NT;
      if (c < NCH) {
        const int pix = c / NCG, cg = c - pix * NCG, oy = pix / L::OW, ox = pix - oy * L::OW;
        *reinterpret_cast<bf16x8*>(buf + ((oy + L::HA) * L::C + ox + L::HB) * L::PP + 8 * cg) = pre[r];
      }
    }
  };
  __syncthreads();  // zero fill done
  if (j0 < n_img) { fetch(j0); put(dzh, mkb); }
  __syncthreads();
  int k = 0;
  for (int img = j0; img < n_img; img += a.wpp, k ^= (NB - 1)) {
    const int nxt = img + a.wpp;
    if (nxt < n_img) fetch(nxt);
    const __bf16* buf = dzh + k * BUF;
    const unsigned char* mb = mkb + k * MKB;
    const long ibase = (long)img * L::IH * L::IW * L::CI;
#pragma unroll
    for (int t = 0; t < MTW; t++) {
      const int mt = mpart + t * MPARTS;
      if (mt >= L::MTC) break;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < L::KS; ks++) {
        const int tp = ks / (L::CO / 32), hf = ks % (L::CO / 32), ta = tp / L::KB, tb = tp % L::KB;
        const bf16x8 A = *reinterpret_cast<const bf16x8*>(buf + abase[t] + ((L::HA - ta) * L::C + L::HB - tb) * L::PP + 32 * hf);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf[ks], A, acc, 0, 0, 0);
      }
      if (oidx[t] >= 0) {
        const unsigned bits = (unsigned)mb[oidx[t] >> 3] >> (oidx[t] & 7);  // 4 channels: one nibble (CI % 8 == 0)
        *reinterpret_cast<bf16x4*>(dx + ibase + oidx[t]) =
            bf16x4{(__bf16)((bits & 1) ? acc[0] : 0.f), (__bf16)((bits & 2) ? acc[1] : 0.f),
                   (__bf16)((bits & 4) ? acc[2] : 0.f), (__bf16)((bits & 8) ? acc[3] : 0.f)};
      }
    }
    if constexpr (NB == 2) {
      if (nxt < n_img) put(dzh + (k ^ 1) * BUF, mkb + (k ^ 1) * MKB);
      __syncthreads();
    } else {  // one buffer: everyone is done reading it, then the next image (already in registers) goes in
      __syncthreads();
      if (nxt < n_img) put(dzh, mkb);
      __syncthreads();
    }
  }
}


// ===================================================================== conv3, everything in one launch
// soft-argmax backward -> dZ3 -> { dgrad3 (-> masked dZ2), wgrad3 } per image, dZ3 never leaving LDS: replaces four
// launches of the dependent chain (soft-argmax backward, its temperature-gradient sum, ebw_dgrad_kernel<L3>,
// ebw_wgrad_tr_kernel<L3>) - each of which paid its own launch, prologue and first-load latency for ~1 us of work per
// image - and two of the three passes over y3 / dZ3 / y2 (HBM traffic is what bounds these kernels: 340 MB of
// activations + 166 MB of slabs at B=256 against ~0.1 ms of MFMA-free time).
//   dZ3 lives in the dgrad's zero-halo image [R][C][PP]; the transposing reads of the weight gradient address its rows
//   through a per-lane pixel -> halo-offset table (padded pixels point at a halo corner, which is zero).
//   The ReLU mask of dgrad3's output is read from the staged y2 itself (8 bytes per lane and tile).
//   Reference semantics: SpatialSoftmax.backward + nn.Conv2d/ReLU autograd, networks/vision/lmp_vision_network.py:32-72.
struct L3Args {
  const float* y3[EBW_MAXP];    // [n][P3][64] conv3 output (post-ReLU, fp32)
  const float* temp[EBW_MAXP];  // learned temperature (1 float)
  const float* sa[EBW_MAXP];    // [n][128] soft-argmax output (x, y interleaved per channel)
  const float* d_sa[EBW_MAXP];  // [n][128] its gradient
  const __bf16* y2[EBW_MAXP];   // [n][IH*IW][64] conv2 output (post-ReLU)
  const uint4* wpk[EBW_MAXP];   // packed W3^T fragments (ebw_pack_kernel)
  __bf16* dz2[EBW_MAXP];        // out: [n][IH*IW][64] masked gradient of conv2's pre-activation
  float* dtp[EBW_MAXP];         // out: per-workgroup partial of the temperature gradient [wpp]
  float* slab;                  // out: weight-gradient slabs, ebw_wgrad_tr_kernel<L3>'s layout
  int n[EBW_MAXP];
  int wpp;
};
// Wave roles: 8 waves per workgroup - waves 0-3 own dgrad3 (one input-channel tile each, W3^T fragments in registers,
// TG3 pixel tiles in flight), waves 4-7 own wgrad3 (64 output channels x a quarter of the taps each: 144 + 16 accumulator
// registers; one wave of each role per SIMD, 108 + 80 MFMAs per image).  With
// every wave doing both (round 3's first version) the 72 fragment + 80 accumulator registers left nothing to read ahead
// with: dgrad3 ran as 54 x (ds_read, wait, MFMA), 38 us per launch; roles keep the two register sets in different waves
// and let the two phases of an image overlap on the MFMA pipe.  All waves share the soft-argmax and the staging.
#ifdef L3_STAMPS  // scratch builds: shader / wall clocks of workgroup 0 per phase, summed over its images
__device__ unsigned long long l3_st[16];
extern "C" int tacorl_l3_stamps_read(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(l3_st), sizeof(l3_st)) == hipSuccess ? 0 : -1;
}
#define L3_ST(i) do { if (tid == 64 * w && blockIdx.x == 0 && (w == 0 || w == ND3)) { const unsigned long long t_ = clock64(); st[i] += t_ - tl; tl = t_; } } while (0)
#else
#define L3_ST(i) do { } while (0)
#endif
constexpr int NT3 = 512, NW3 = NT3 / 64, ND3 = 4;
struct L3Tile { static constexpr int MGRP = 1, NGRP = 4, KGRP = 1; };  // wgrad3's wave tiling here (TrTile: 2 x 4 over 8 waves)
template <class L>
struct L3Geo {
  using TG = TrGeo<L>;
  static constexpr int P3 = L::OH * L::OW, K3 = (P3 + NW3 - 1) / NW3;  // soft-argmax: wave w owns pixels w, w + 8, ...
  static constexpr int HALO = L::R * L::C * L::PP;                     // elements
  static constexpr int CIP = TG::CIP, Y2E = (L::IH * L::IW * CIP + 7) / 8 * 8;
  static constexpr int KS = (P3 + 31) / 32;                            // weight-gradient k-steps (32 pixels)
  static constexpr int Z2E = L::IH * L::IW * L::CI;                     // dZ2 of one image (elements), staged for coalesced stores
  static constexpr int MPW = L::MT / L3Tile::MGRP, NPW = L::NTL / L3Tile::NGRP;
  static constexpr size_t lds_bytes = ((size_t)HALO + 2 * Y2E + 2 * Z2E) * 2 + 2 * NW3 * 64 * 4 + 64;
  static_assert(L::S == 1 && L::CI == 64 && L::CO == 64 && L::NCOMBO == ND3, "conv3 of the LMPVisionEncoder");
  static constexpr bool ok = K3 <= 8 && lds_bytes <= 160 * 1024;  // (128 x 128: 144 conv3 pixels - the soft-argmax's registers spill)
};

template <class L>
__global__ __launch_bounds__(NT3) void ebw_l3_kernel(L3Args a) {
  using G = L3Geo<L>;
  using TG = TrGeo<L>;
  using T = L3Tile;
  static_assert(T::MGRP * T::NGRP * T::KGRP == NW3 - ND3, "wgrad waves");
  static_assert(G::ok, "geometry not supported by the fused conv3 backward");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* dzh = reinterpret_cast<__bf16*>(smem);                     // dZ3, zero halo
  __bf16* y2s = dzh + G::HALO;                                      // 2 x [IH*IW][CIP]
  __bf16* z2s = y2s + 2 * G::Y2E;                                   // 2 x [IH*IW][64]: dZ2 on its way out
  float* shm = reinterpret_cast<float*>(z2s + 2 * G::Z2E);          // [NW3][64] per-wave maxima
  float* shs = shm + NW3 * 64;                                      // [NW3][64] per-wave sums
  float* shd = shs + NW3 * 64;                                      // [NW3]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, l16 = lane & 15, q = l16 >> 2, pc = l16 & 3;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the role branch below must be a uniform branch
  const int p = blockIdx.x / a.wpp, j0 = blockIdx.x - p * a.wpp;
  const int n_img = a.n[p];
  // only the halo image needs zeros (its border and the pad columns are never written); y2's pad columns are never read
  for (int e = tid; e < G::HALO / 8; e += NT3) reinterpret_cast<uint4*>(dzh)[e] = make_uint4(0u, 0u, 0u, 0u);
  const float it = 1.0f / a.temp[p][0];
  const float* y3 = a.y3[p];
  const __bf16* y2 = a.y2[p];
  constexpr int ICH = L::IH * L::IW * (L::CI / 8), ICPT = (ICH + NT3 - 1) / NT3;
  // One image loop per role (same barrier sequence in both): role 0 = dgrad3, role 1 = wgrad3.
  auto run = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;
    float vn[G::K3];
    f32x2 gn, fn;
    bf16x8 ipre[ICPT];
    // the next image's soft-argmax inputs are fetched once this image's are dead (after the soft-argmax phase: the two
    // sets never overlap in registers), its y2 at the top of the iteration
    auto fetch3 = [&](int img) {
#pragma unroll
      for (int k = 0; k < G::K3; k++) {
        const int i = w + NW3 * k;
        vn[k] = y3[((long)img * G::P3 + (i < G::P3 ? i : G::P3 - 1)) * 64 + lane];
      }
      gn = *reinterpret_cast<const f32x2*>(a.d_sa[p] + (long)img * 128 + 2 * lane);
      fn = *reinterpret_cast<const f32x2*>(a.sa[p] + (long)img * 128 + 2 * lane);
    };
    auto fetch = [&](int img) {
#pragma unroll
      for (int r = 0; r < ICPT; r++) {
        const int c = tid + r * NT3;
        ipre[r] = *reinterpret_cast<const bf16x8*>(y2 + (long)img * L::IH * L::IW * L::CI + 8 * (c < ICH ? c : ICH - 1));
      }
    };
    auto put = [&](__bf16* buf) {
#pragma unroll
      for (int r = 0; r < ICPT; r++) {
        const int c = tid + r * NT3;
        if (c < ICH) *reinterpret_cast<bf16x8*>(buf + (c / (L::CI / 8)) * G::CIP + 8 * (c % (L::CI / 8))) = ipre[r];
      }
    };
    // dZ2 leaves through LDS: the dgrad waves write their tiles into z2s, and EVERY thread stores 16 contiguous bytes of
    // the PREVIOUS image behind the next soft-argmax phase - the stores' acknowledgements (in-order vmcnt: any later wait
    // for a prefetch waits for them too) then have a whole iteration to arrive; issued by the dgrad waves themselves at
    // the end of their phase they stalled the loop top by a store round trip per image
    constexpr int ZCH = G::Z2E / 8, ZCPT = (ZCH + NT3 - 1) / NT3;
    auto flush = [&](int img, const __bf16* zb) {
#pragma unroll
      for (int r = 0; r < ZCPT; r++) {
        const int c = tid + r * NT3;
        if (c < ZCH) *reinterpret_cast<bf16x8*>(a.dz2[p] + (long)img * G::Z2E + 8 * c) = *reinterpret_cast<const bf16x8*>(zb + 8 * c);
      }
    };
    // ---- role state
    // dgrad3: wave = input-channel tile nt, all pixel tiles; W3^T fragments stay in registers
    constexpr int MT = L::MTC, TG3 = 2, NGRP3 = (MT + TG3 - 1) / TG3;  // pixel tiles, processed TG3 at a time
    bf16x8 Bf[ROLE == 0 ? L::KS : 1];
    int abase[ROLE == 0 ? MT : 1];
    int obase = 0;
    if constexpr (ROLE == 0) {
      const int nt = w;
#pragma unroll
      for (int ks = 0; ks < L::KS; ks++) Bf[ks] = __builtin_bit_cast(bf16x8, a.wpk[p][(nt * L::KS + ks) * 64 + lane]);
#pragma unroll
      for (int t = 0; t < MT; t++) {
        const int m0 = 16 * t + l16;
        const int m = m0 < L::PH * L::PW ? m0 : L::PH * L::PW - 1;
        const int Y = m / L::PW, X = m - Y * L::PW;
        abase[t] = (Y * L::C + X) * L::PP + 8 * g;  // halo pixel (Y, X) = tap (KA-1, KB-1): every tap offset below is >= 0 (an immediate)
      }
      // S = 1: output pixel m = 16 t + l16, 4 consecutive input channels: element index obase + t * 16 * CI
      obase = l16 * L::CI + 16 * nt + 4 * g;
    }
    // wgrad3 (TrTile): n-tiles interleaved over the wave groups (tile j of group ng = j NGRP + ng = tap j, channels 16 ng ..):
    // the tap part of a B-fragment address is a compile-time constant, the lane part one register
    static_assert(L::CI / 16 == T::NGRP, "one tap per n-tile round");
    const int wv = w - ND3, ng = wv % T::NGRP, mg = (wv / T::NGRP) % T::MGRP;
    const int coff = 16 * ng + 4 * pc;
    int zr0[ROLE == 1 ? G::KS : 1], zr1[ROLE == 1 ? G::KS : 1], ir0[ROLE == 1 ? G::KS : 1], ir1[ROLE == 1 ? G::KS : 1];
    f32x4 acc[ROLE == 1 ? G::MPW : 1][ROLE == 1 ? G::NPW : 1], bacc[ROLE == 1 ? G::MPW : 1];
    bf16x8 ones;
    if constexpr (ROLE == 1) {
#pragma unroll
      for (int s = 0; s < G::KS; s++) {  // dZ3 halo rows / input rows of this lane's two pixels per k-step
        const int P0 = 32 * s + 4 * g + q, P1 = P0 + 16;
        const int Q0 = P0 < G::P3 ? P0 : G::P3 - 1, Q1 = P1 < G::P3 ? P1 : G::P3 - 1;
        zr0[s] = P0 < G::P3 ? ((P0 / L::OW + L::HA) * L::C + P0 % L::OW + L::HB) * L::PP : 0;  // halo corner: zero
        zr1[s] = P1 < G::P3 ? ((P1 / L::OW + L::HA) * L::C + P1 % L::OW + L::HB) * L::PP : 0;
        ir0[s] = ((Q0 / L::OW) * L::IW + Q0 % L::OW) * G::CIP;
        ir1[s] = ((Q1 / L::OW) * L::IW + Q1 % L::OW) * G::CIP;
      }
#pragma unroll
      for (int i = 0; i < G::MPW; i++) {
        bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < G::NPW; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 8; j++) ones[j] = (__bf16)1.0f;
    }
    float dt = 0.f;
#ifdef L3_STAMPS
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, tl = clock64();
    const unsigned long long w0 = wall_clock64(), c0 = tl;
#endif
    __syncthreads();  // zero fill done
    if (j0 < n_img) { fetch(j0); fetch3(j0); put(y2s); }
    __syncthreads();
    int kb = 0;
    L3_ST(0);  // prologue
    for (int img = j0; img < n_img; img += a.wpp, kb ^= 1) {
      const int nxt = img + a.wpp;
      int wl = w;  // opaque copy: the pixel coordinates derived from it are scalar arithmetic per image; hoisted out of the
      asm volatile("" : "+s"(wl));  // loop (as floats in VGPRs, one set per role) they were what spilled
      float v[G::K3];
#pragma unroll
      for (int k = 0; k < G::K3; k++) v[k] = vn[k];
      const float gx = gn[0], gy = gn[1], dot = gn[0] * fn[0] + gn[1] * fn[1];
      // every use of the previous prefetch sits above this point: a consumer that sinks below the loads issued next waits
      // for THEM as well (in-order vmcnt) - hipcc had moved `dot` behind the second barrier: a full HBM latency per image
      asm volatile("" : : "v"(gx), "v"(gy), "v"(dot) : "memory");
#pragma unroll
      for (int k = 0; k < G::K3; k++) asm volatile("" : : "v"(v[k]) : "memory");
      if (nxt < n_img) { fetch(nxt); fetch3(nxt); }  // both prefetches have the whole iteration to land
      const __bf16* yb = y2s + kb * G::Y2E;
      // ---- soft-argmax backward (softargmax_bwd_batch_kernel's arithmetic): dZ3 as bf16 into the halo image
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < G::K3; k++) {
        v[k] *= it;
        if (wl + NW3 * k < G::P3) mx = fmaxf(mx, v[k]);
      }
      // one barrier for both statistics: every wave publishes its own maximum and its sum of exponentials RELATIVE to
      // that maximum; the channel's sum is then sum_w s_w exp(m_w - M)
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < G::K3; k++) se += (wl + NW3 * k < G::P3) ? __expf(v[k] - mx) : 0.f;
      shm[wl * 64 + lane] = mx;
      shs[wl * 64 + lane] = se;
      __syncthreads();
      {
        float mw[NW3], sw[NW3];  // every read in flight before the first use (hipcc issued them one round trip at a time)
#pragma unroll
        for (int ww = 0; ww < NW3; ww++) { mw[ww] = shm[ww * 64 + lane]; sw[ww] = shs[ww * 64 + lane]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ww = 0; ww < NW3; ww++) mx = fmaxf(mx, mw[ww]);
        se = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW3; ww++) se += sw[ww] * __expf(mw[ww] - mx);  // (a wave without pixels: 0 * exp(-inf) = 0)
      }
      const float rse = 1.0f / se;
#pragma unroll
      for (int k = 0; k < G::K3; k++) {
        const int i = wl + NW3 * k;
        if (i < G::P3) {
          const int iy = i / L::OW;
          const float pr = __expf(v[k] - mx) * rse;  // (recomputed: K3 registers matter more here than K3 exponentials)
          const float ds = pr * (gx * (float)(i - iy * L::OW) + gy * (float)iy - dot);
          dt -= ds * v[k] * it;
          dzh[((iy + L::HA) * L::C + (i - iy * L::OW) + L::HB) * L::PP + lane] = (__bf16)(v[k] != 0.f ? ds * it : 0.f);  // ReLU mask: y3 > 0 <=> y3 / t != 0
        }
      }
      L3_ST(1);  // soft-argmax
      if (img != j0) flush(img - a.wpp, z2s + (kb ^ 1) * G::Z2E);
      __syncthreads();  // dZ3 of this image complete
      L3_ST(2);  // flush + barrier
      if constexpr (ROLE == 0) {
        // ---- dgrad3: dZ2 = (dZ3 (*) W3^T) masked by y2 > 0; TG3 pixel tiles share every fragment step
        __bf16* zw = z2s + kb * G::Z2E;
#pragma unroll
        for (int tg = 0; tg < NGRP3; tg++) {
          f32x4 o[TG3];
#pragma unroll
          for (int u = 0; u < TG3; u++) o[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          // A fragments: the first HK steps of every tile of the group are read before the first MFMA, each slot is
          // refilled with step ks + HK as soon as its MFMA has issued (the compiler's own schedule read each fragment
          // right in front of its MFMA: one LDS round trip per two MFMAs, 72 clk per MFMA)
          constexpr int HK = L::KS / 2;
          static_assert(L::KS % 2 == 0, "fragment steps in two halves");
          auto aoff = [&](int ks) {
            const int tp = ks / (L::CO / 32), hf = ks % (L::CO / 32), ta = tp / L::KB, tb = tp % L::KB;
            return ((L::HA - ta) * L::C + L::HB - tb) * L::PP + 32 * hf;
          };
          bf16x8 A[TG3][HK];
#pragma unroll
          for (int ks = 0; ks < HK; ks++)
#pragma unroll
            for (int u = 0; u < TG3; u++)
              A[u][ks] = *reinterpret_cast<const bf16x8*>(dzh + abase[tg * TG3 + u < MT ? tg * TG3 + u : MT - 1] + aoff(ks));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < L::KS; ks++) {
#pragma unroll
            for (int u = 0; u < TG3; u++) {
              o[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bf[ks], A[u][ks % HK], o[u], 0, 0, 0);
              if (ks + HK < L::KS)
                A[u][ks % HK] = *reinterpret_cast<const bf16x8*>(dzh + abase[tg * TG3 + u < MT ? tg * TG3 + u : MT - 1] + aoff(ks + HK));
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int u = 0; u < TG3; u++) {
            const int t = tg * TG3 + u;
            if (t >= MT) continue;
            const int oi = obase + t * 16 * L::CI;
            if (oi < L::PH * L::PW * L::CI) {
              const uint2 yv = *reinterpret_cast<const uint2*>(yb + (oi >> 6) * G::CIP + (oi & 63));
              const bool m0 = (((yv.x & 0xffffu) - 1u) & 0xffffu) < 0x7fffu, m1 = (((yv.x >> 16) - 1u) & 0xffffu) < 0x7fffu;
              const bool m2 = (((yv.y & 0xffffu) - 1u) & 0xffffu) < 0x7fffu, m3 = (((yv.y >> 16) - 1u) & 0xffffu) < 0x7fffu;
              *reinterpret_cast<bf16x4*>(zw + oi) =
                  bf16x4{(__bf16)(m0 ? o[u][0] : 0.f), (__bf16)(m1 ? o[u][1] : 0.f), (__bf16)(m2 ? o[u][2] : 0.f),
                         (__bf16)(m3 ? o[u][3] : 0.f)};
            }
          }
        }
      } else {
        // ---- wgrad3: dW3[co][tap] += sum_pixels dZ3[pixel][co] y2[pixel @ tap]
#pragma unroll
        for (int s = 0; s < G::KS; s++) {
          // the fragments of a k-step are read ahead of their MFMAs in three batches (all 9 tap fragments at once, or 5 + 4, spilled - and a scratch reload waits for the prefetch in flight)
          constexpr int JB = (G::NPW + 2) / 3;
          bf16x8 A[G::MPW], B[JB];
#pragma unroll
          for (int i = 0; i < G::MPW; i++) {
            const int col = 16 * (mg * G::MPW + i) + 4 * pc;
            A[i] = tr_frag(dzh + zr0[s] + col, dzh + zr1[s] + col);
          }
#pragma unroll
          for (int jb = 0; jb < G::NPW; jb += JB) {
#pragma unroll
            for (int j = jb; j < jb + JB && j < G::NPW; j++) {
              const int tofs = ((j / L::KW) * L::IW + j % L::KW) * G::CIP;  // tap j = (ky, kx)
              B[j - jb] = tr_frag(yb + ir0[s] + coff + tofs, yb + ir1[s] + coff + tofs);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = jb; j < jb + JB && j < G::NPW; j++)
#pragma unroll
              for (int i = 0; i < G::MPW; i++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], B[j - jb], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (ng == 0) {
#pragma unroll
            for (int i = 0; i < G::MPW; i++) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], ones, bacc[i], 0, 0, 0);
          }
        }
      }
      L3_ST(3);  // role phase
      if (nxt < n_img) put(y2s + (kb ^ 1) * G::Y2E);
      __syncthreads();  // dZ3 / y2 of this image consumed, the next y2 staged
      L3_ST(4);  // put + barrier
    }
    if (j0 < n_img) flush(j0 + ((n_img - 1 - j0) / a.wpp) * a.wpp, z2s + (kb ^ 1) * G::Z2E);  // the last image's dZ2 (its buffer: kb flipped once more)
    if constexpr (ROLE == 1) {
      // ---- weight-gradient slab (ebw_wgrad_tr_kernel's order with wave index wv)
      float* sl = a.slab + (long)blockIdx.x * TG::SLABF;
#pragma unroll
      for (int i = 0; i < G::MPW; i++)
#pragma unroll
        for (int j = 0; j < G::NPW; j++)
          *reinterpret_cast<f32x4*>(sl + (((wv * G::MPW + i) * G::NPW + j) * 64 + lane) * 4) = acc[i][j];
      if (ng == 0 && l16 == 0) {
#pragma unroll
        for (int i = 0; i < G::MPW; i++)
#pragma unroll
          for (int r = 0; r < 4; r++) sl[L::CO * L::TAPS + 16 * (mg * G::MPW + i) + 4 * g + r] = bacc[i][r];
      }
    }
    // ---- temperature-gradient partial of this workgroup
    dt = wave_sum(dt);
    if (lane == 0) shd[w] = dt;
    __syncthreads();
    if (tid == 0 && j0 < n_img) {  // (workgroups without an image have nothing to add; dtp holds min(wpp, n) partials)
      float sum = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW3; ww++) sum += shd[ww];
      a.dtp[p][j0] = sum;
    }
#ifdef L3_STAMPS
    L3_ST(5);  // epilogue
    if (tid == 64 * w && blockIdx.x == 0 && (w == 0 || w == ND3)) {
      const int o = w == 0 ? 0 : 8;
      for (int i = 0; i < 6; i++) l3_st[o + i] = st[i];
      l3_st[o + 6] = clock64() - c0; l3_st[o + 7] = wall_clock64() - w0;
    }
#endif
  };
  if (w < ND3) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
}

// ====================================================================== host
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int cdivi(long a, long b) { return (int)((a + b - 1) / b); }
constexpr int cdivi_c(int a, int b) { return (a + b - 1) / b; }

struct WsPlan {
  size_t wpk2[EBW_MAXP], wpk3[EBW_MAXP], dz2[EBW_MAXP], dz1[EBW_MAXP], slab1, slab2, slab3, total;
  int wpp;
};
template <class G>
WsPlan plan(int nprob, const int* n) {
  WsPlan w{};
  long maxn = 1;
  for (int p = 0; p < nprob; p++) maxn = n[p] > maxn ? n[p] : maxn;
  // one workgroup per CU (256 CUs), never more: a second round of workgroups would double the time
  const int per = 256 / nprob > 1 ? 256 / nprob : 1;
  w.wpp = (int)(per < maxn ? per : maxn);
  size_t off = 0;
  for (int p = 0; p < nprob; p++) {
    w.wpk2[p] = off; off += al256((size_t)G::L2::NCOMBO * G::L2::KS * 64 * 16);
    w.wpk3[p] = off; off += al256((size_t)G::L3::NCOMBO * G::L3::KS * 64 * 16);
    w.dz2[p] = off; off += al256((size_t)n[p] * G::L3::IH * G::L3::IW * 64 * 2);
    w.dz1[p] = off; off += al256((size_t)n[p] * G::L2::IH * G::L2::IW * 32 * 2);
  }
  const size_t nwg = (size_t)nprob * w.wpp;
  w.slab1 = off; off += al256(nwg * TrGeo<typename G::L1>::SLABF * 4);
  w.slab2 = off; off += al256(nwg * TrGeo<typename G::L2>::SLABF * 4);
  w.slab3 = off; off += al256(nwg * TrGeo<typename G::L3>::SLABF * 4);
  w.total = off;
  return w;
}

template <class K>
int set_lds(K kern, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) ==
                 hipSuccess ? 0 : -1;
}

template <class L, class InT, class DzT>
int launch_wgrad_tr(const WgArgs& a, int nwg, hipStream_t st) {
  auto kern = ebw_wgrad_tr_kernel<L, InT, DzT>;
  constexpr size_t lds = TrGeo<L>::lds_bytes;
  static int once = set_lds(kern, lds);
  if (once) return TACORL_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), lds, st, a);
  return TACORL_OK;
}
template <class L, class DzT>
int launch_dgrad(const DgArgs& a, int nwg, hipStream_t st) {
  auto kern = ebw_dgrad_kernel<L, DzT>;
  constexpr size_t lds = dgrad_lds_bytes<L>();
  static int once = set_lds(kern, lds);
  if (once) return TACORL_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), lds, st, a);
  return TACORL_OK;
}
template <class L>
int launch_l3(const L3Args& a, int nwg, hipStream_t st) {
  auto kern = ebw_l3_kernel<L>;
  constexpr size_t lds = L3Geo<L>::lds_bytes;
  static int once = set_lds(kern, lds);
  if (once) return TACORL_ELAUNCH;
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT3), lds, st, a);
  return TACORL_OK;
}
template <class G>
int run(int nprob, const EbwProblem* pr, int accumulate, void* ws, size_t ws_bytes, hipStream_t st, int mode, int parts) {
  using L1 = typename G::L1; using L2 = typename G::L2; using L3 = typename G::L3;
  int n[EBW_MAXP];
  for (int p = 0; p < nprob; p++) n[p] = pr[p].n;
  const WsPlan w = plan<G>(nprob, n);
  if (ws_bytes < w.total) return TACORL_ENOMEM;
  unsigned char* base = (unsigned char*)ws;
  const int nwg = nprob * w.wpp;
  // W^T fragments for the two dgrads
  PkArgs k2{}, k3{};
  for (int p = 0; p < nprob; p++) {
    k2.w[p] = pr[p].w2; k2.out[p] = (uint4*)(base + w.wpk2[p]);
    k3.w[p] = pr[p].w3; k3.out[p] = (uint4*)(base + w.wpk3[p]);
  }
  if (mode != 2) {
    hipLaunchKernelGGL((ebw_pack2_kernel<L2, L3>), dim3(cdivi(L2::NCOMBO * L2::KS * 64, 256) + cdivi(L3::NCOMBO * L3::KS * 64, 256), nprob),
                       dim3(256), 0, st, k2, k3);
  }
  if (mode == 1) return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
  DgArgs d3{}, d2{};
  WgArgs g3{}, g2{}, g1{};
  float *gw1[EBW_MAXP], *gb1[EBW_MAXP], *gw2[EBW_MAXP], *gb2[EBW_MAXP], *gw3[EBW_MAXP], *gb3[EBW_MAXP];
  for (int p = 0; p < nprob; p++) {
    __bf16* dz2 = (__bf16*)(base + w.dz2[p]);
    __bf16* dz1 = (__bf16*)(base + w.dz1[p]);
    d3.dz[p] = pr[p].dz3; d3.yin[p] = (const __bf16*)pr[p].y2; d3.wpk[p] = k3.out[p]; d3.dx[p] = dz2; d3.n[p] = n[p];
    d2.dz[p] = dz2; d2.yin[p] = (const __bf16*)pr[p].y1; d2.wpk[p] = k2.out[p]; d2.dx[p] = dz1; d2.n[p] = n[p];
    g3.in[p] = pr[p].y2; g3.dz[p] = pr[p].dz3; g3.n[p] = n[p];
    g2.in[p] = pr[p].y1; g2.dz[p] = dz2; g2.n[p] = n[p];
    g1.in[p] = pr[p].img; g1.dz[p] = dz1; g1.n[p] = n[p];
    gw1[p] = pr[p].g_w1; gb1[p] = pr[p].g_b1; gw2[p] = pr[p].g_w2; gb2[p] = pr[p].g_b2;
    gw3[p] = pr[p].g_w3; gb3[p] = pr[p].g_b3;
  }
  d3.wpp = d2.wpp = g3.wpp = g2.wpp = g1.wpp = w.wpp;
  // the dgrads need < 40 KB of LDS and ~70 registers: two of their workgroups fit on a CU and hide each other's global
  // latency (a workgroup's image loop waits for the next image's loads every iteration).  TACORL_EBW_DG2=0: one per CU
  static const int dg2 = [] { const char* e = getenv("TACORL_EBW_DG2"); return e ? atoi(e) : 2; }();
  long maxn = 1;
  for (int p = 0; p < nprob; p++) maxn = n[p] > maxn ? n[p] : maxn;
  auto per_cu = [&](size_t lds) { return dg2 > 1 && (size_t)dg2 * lds <= 160 * 1024 ? dg2 : 1; };  // 128 x 128: one
  const int k3w = per_cu(dgrad_lds_bytes<L3>()), k2w = per_cu(dgrad_lds_bytes<L2>());
  d3.wpp = (int)((long)k3w * w.wpp < maxn ? (long)k3w * w.wpp : maxn);
  d2.wpp = (int)((long)k2w * w.wpp < maxn ? (long)k2w * w.wpp : maxn);
  g1.slab = (float*)(base + w.slab1); g2.slab = (float*)(base + w.slab2); g3.slab = (float*)(base + w.slab3);
  int rc;
  const bool fused3 = (parts & EBW_FUSED3) != 0;
  if (fused3 && (parts & EBW_ALL) != EBW_ALL) return TACORL_EINVAL;
  {
    if (fused3) {
      L3Args f{};
      for (int p = 0; p < nprob; p++) {
        f.y3[p] = pr[p].y3; f.temp[p] = pr[p].temp; f.sa[p] = pr[p].sa; f.d_sa[p] = pr[p].d_sa;
        f.y2[p] = (const __bf16*)pr[p].y2; f.wpk[p] = k3.out[p]; f.dz2[p] = (__bf16*)(base + w.dz2[p]);
        f.dtp[p] = pr[p].dtp; f.n[p] = n[p];
      }
      f.slab = g3.slab; f.wpp = w.wpp;
      if constexpr (L3Geo<L3>::ok) {
        if ((rc = launch_l3<L3>(f, nwg, st))) return rc;
      } else {
        return TACORL_EINVAL;  // ebw_fused3_supported() said no
      }
    } else {
      if ((parts & EBW_DGRAD3) && (rc = launch_dgrad<L3, float>(d3, nprob * d3.wpp, st))) return rc;
      if ((parts & EBW_WGRAD3) && (rc = launch_wgrad_tr<L3, __bf16, float>(g3, nwg, st))) return rc;
    }
    if ((parts & EBW_DGRAD2) && (rc = launch_dgrad<L2, __bf16>(d2, nprob * d2.wpp, st))) return rc;
    if ((parts & EBW_WGRAD2) && (rc = launch_wgrad_tr<L2, __bf16, __bf16>(g2, nwg, st))) return rc;
    if ((parts & EBW_WGRAD1) && (rc = launch_wgrad_tr<L1, __bf16, __bf16>(g1, nwg, st))) return rc;
    if (!(parts & EBW_REDUCE)) return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
    RdTrArgs t{};
    t.slab[0] = g1.slab; t.slab[1] = g2.slab; t.slab[2] = g3.slab;
    auto fill = [&](int l, auto geo, int co, int taps) {
      using TG = decltype(geo);
      t.co[l] = co; t.taps[l] = taps; t.mpw[l] = TG::MPW; t.npw[l] = TG::NPW; t.mgrp[l] = TG::T::MGRP;
      t.ngrp[l] = TG::T::NGRP; t.kgrp[l] = TG::T::KGRP; t.slabf[l] = TG::SLABF;
    };
    fill(0, TrGeo<L1>{}, L1::CO, L1::TAPS); fill(1, TrGeo<L2>{}, L2::CO, L2::TAPS); fill(2, TrGeo<L3>{}, L3::CO, L3::TAPS);
    t.wpp = w.wpp; t.accumulate = accumulate;
    for (int p = 0; p < nprob; p++) {
      t.gw[0][p] = gw1[p]; t.gb[0][p] = gb1[p]; t.gw[1][p] = gw2[p]; t.gb[1][p] = gb2[p];
      t.gw[2][p] = gw3[p]; t.gb[2][p] = gb3[p];
    }
    constexpr int maxper_t = cmax(L1::SLABF, cmax(L2::SLABF, L3::SLABF));
    static_assert(cdivi_c(maxper_t, 64) - 1 > cdivi_c(L1::SLABF, 64), "the temperature block must lie beyond conv1's elements");
    if (fused3) {
      for (int p = 0; p < nprob; p++) { t.dtp[p] = pr[p].dtp; t.g_temp[p] = pr[p].g_temp; t.ndt[p] = n[p] < w.wpp ? n[p] : w.wpp; }
      t.ileave[2] = 1; t.mpw[2] = L3Geo<L3>::MPW; t.npw[2] = L3Geo<L3>::NPW; t.mgrp[2] = L3Tile::MGRP; t.ngrp[2] = L3Tile::NGRP;
    }
    hipLaunchKernelGGL(ebw_reduce_tr_kernel, dim3(cdivi(maxper_t, 64), 3 * nprob), dim3(256), 0, st, t);
    return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
  }
}

}  // namespace

// the geometries encoder_fused.hip is instantiated for (the cameras of the reference's configs)
#define EBW_GEOMS(X) X(84, 84) X(64, 64) X(44, 60) X(128, 128) X(150, 200)

// (conv3 outputs of up to 64 pixels: the soft-argmax keeps a wave's pixels of the image in registers; 128 x 128 - 144
// pixels - spilled 155 registers and takes the separate launches)
bool ebw_fused3_supported(int H, int W) {
#define X(h, w) if (H == h && W == w) return L3Geo<Geo<h, w>::L3>::ok;
  EBW_GEOMS(X)
#undef X
  return false;
}
bool ebw_supported(int H, int W) {
#define X(h, w) if (H == h && W == w) return true;
  EBW_GEOMS(X)
#undef X
  return false;
}
size_t ebw_ws_bytes(int nprob, const int* n_img, int H, int W) {
  if (nprob < 1 || nprob > EBW_MAXP) return 0;
#define X(h, w) if (H == h && W == w) return plan<Geo<h, w>>(nprob, n_img).total;
  EBW_GEOMS(X)
#undef X
  return 0;
}
int ebw_conv_backward(int nprob, const EbwProblem* pr, int H, int W, int accumulate, void* ws, size_t ws_bytes,
                      hipStream_t st, int mode, int parts) {
  if (nprob < 1 || nprob > EBW_MAXP) return TACORL_EINVAL;
#define X(h, w) if (H == h && W == w) return run<Geo<h, w>>(nprob, pr, accumulate, ws, ws_bytes, st, mode, parts);
  EBW_GEOMS(X)
#undef X
  return TACORL_EINVAL;
}
