// Fused LMPVisionEncoder forward for camera geometries whose conv1 output does NOT fit the LDS beside conv2's:
// 150 x 200, the un-resized rgb_static of experiment=tacorl_real_world (reference
// config/datamodule/transform_manager/transforms/rl_real_world_train.yaml:2-10; the network is encoder.py:369-419,
// utils.py:39-65 as in encoder_fused.hip).  Same launch table, packed weights, work split and arithmetic as
// encoder_fused_kernel - conv8x8s4+ReLU -> conv4x4s2+ReLU -> conv3x3s1+ReLU -> spatial soft-argmax -> FC+ReLU -> FC, bf16 MFMA
// with fp32 accumulation, weights register-stationary in 248 AGPRs, one workgroup of four waves per CU - with a different
// residency plan for the activations:
//  * the image arrives in BANDS of 20 rows (4 conv1 output rows), double-buffered by LDS-DMA: 2 x 24 000 B;
//  * conv1's output lives in a RING of 8 rows (row r in slot r & 7; 31 KB instead of 141 KB): after band b conv1 rows
//    4 b - 4 .. 4 b + 3 are present, which is what the conv2 pixels that became computable with this band need - conv2 row r
//    reads conv1 rows 2 r .. 2 r + 3, rows 2 b - 1 and 2 b are new, and a 16-pixel tile reaches back at most one conv2 row
//    (OW2 >= 16) to row 2 b - 2.  conv2 therefore runs band by band, over the 16-pixel tiles of the flat pixel index whose
//    last pixel has become computable (no padding except in the image's last tile); a lane addresses its window through
//    TWO bases (rows 2 r, 2 r + 1 and rows 2 r + 2, 2 r + 3: the ring is even, a pair never wraps), taps stay immediates;
//  * conv2's output is resident whole (17 x 23 x 64 at a 160-byte pixel pitch: 66 KB); conv3 walks its 20 tiles with the
//    accumulators of ONE tile live and an ONLINE soft-argmax per lane (running max / sum exp / sum exp x / sum exp y over the
//    pixels the lane has seen, rescaled when the maximum moves - v3[NT3] of encoder_fused would be 80 registers here); the 16
//    pixel lanes are merged once per image.  A tile's soft-argmax update runs inside the next tile's MFMA chain.
// Every fragment set is double: the next tile's reads are issued between the running chain's MFMAs into the other set.
// Problems a backward follows save their activations from the same launch: y3, the soft-argmax features and fc1 as fp32 at
// their tacorl_encoder_act_layout offsets; y1 / y2 as bf16 at the start of their slots where the geometry has the LDS-resident
// backward (EFArgs::act_bf16 - what encoder_fused_kernel saves and tacorl_encoder_bwd_fused* read; 150 x 200 since later in
// round 6), as fp32 for the per-layer tacorl_encoder_bwd otherwise.  tacorl_encoder_fused_act_format() tells the caller which.
#include <stdio.h>
#include <stdlib.h>

#include "encoder_fused.h"

#define ER_BR 4  // conv1 output rows per band
#define ER_R1 8  // conv1 ring rows (a power of two, >= ER_BR + 4)

template <int H_, int W_>
struct ERGeom {
  static constexpr int H = H_, W = W_;
  static constexpr int OH1 = (H - 8) / 4 + 1, OW1 = (W - 8) / 4 + 1;
  static constexpr int OH2 = (OH1 - 4) / 2 + 1, OW2 = (OW1 - 4) / 2 + 1;
  static constexpr int OH3 = OH2 - 2, OW3 = OW2 - 2;
  static constexpr int ROW_BYTES = W * 6, IMG_BYTES = H * ROW_BYTES;
  static constexpr int NPX1 = OH1 * OW1, NPX2 = OH2 * OW2, NPX3 = OH3 * OW3;
  static constexpr int NB = (OH1 + ER_BR - 1) / ER_BR;
  static constexpr int BAND_ROWS = 4 * ER_BR + 4, BAND_BYTES = BAND_ROWS * ROW_BYTES, LDS_IMG = (BAND_BYTES + 15) & ~15;
  static constexpr int NPXB = ER_BR * OW1, NT1 = (NPXB + 15) >> 4, PER1 = (NT1 + 3) >> 2;
  // row pads from scratch/micro/bank_sim.py (ring of 8 / whole image): 4.56 and 4.2 LDS cycles per ds_read_b128 over every
  // tile and tap of 150 x 200 (dense rows: 6.4 / 7.0)
  static constexpr int PAD1 = 96, PITCH1 = OW1 * ACT1_STRIDE + PAD1, ACT1_BYTES = ER_R1 * PITCH1;
  static constexpr int PX2 = 160, PAD2 = 192, PITCH2 = OW2 * PX2 + PAD2, ACT2_BYTES = OH2 * PITCH2;
  static constexpr int NT2 = (NPX2 + 15) >> 4, NT3 = (NPX3 + 15) >> 4;
  static constexpr int LDS_BYTES = 2 * LDS_IMG + ACT1_BYTES + ACT2_BYTES + EF_CHUNK * (SA_STRIDE + H1_STRIDE);
  static constexpr int N16 = BAND_BYTES / 16, NPIECE = (N16 + 255) / 256;  // KiB pieces of a band per wave
  // conv2 tiles whose last pixel is computable once band b's conv1 rows exist (cumulative)
  static constexpr int t2_upto(int b) {
    if (b < 0) return 0;
    if (b >= NB - 1) return NT2;
    const int rows = 2 * b + 1 < OH2 ? 2 * b + 1 : OH2;
    return (rows * OW2) >> 4;
  }
  static constexpr bool bands_ok() {
    for (int b = 0; b < NB; b++)
      if (t2_upto(b) - t2_upto(b - 1) > 4) return false;  // at most two tiles per wave and band
    return true;
  }
  static constexpr bool OK = ROW_BYTES % 16 == 0 && LDS_BYTES <= 160 * 1024 && OH3 >= 1 && OW3 >= 1 && OW2 >= 16 &&
                             OH1 % ER_BR == 0 && NT1 >= 4 && bands_ok() && (ER_R1 & (ER_R1 - 1)) == 0;
};

template <int H_, int W_>
__global__ __launch_bounds__(256, 1) void encoder_ring_kernel(EFArgs a_) {
  typedef ERGeom<H_, W_> G;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // launch balance by work units, as encoder_fused_kernel: workgroup b serves the images that start inside its share of the
  // launch's unit line - a run of one problem, or the tail of one and the head of the next (weights reloaded once)
  const unsigned long u_lo = (unsigned long)blockIdx.x * (unsigned long)a_.utotal / gridDim.x;
  const unsigned long u_hi = (unsigned long)(blockIdx.x + 1) * (unsigned long)a_.utotal / gridDim.x;
#pragma unroll 1
  for (int pi = 0; pi < a_.nprob; pi++) {
    const long seg_s = a_.p[pi].ustart, seg_e = seg_s + (long)a_.p[pi].n_img * a_.p[pi].cost;
    if ((long)u_hi <= seg_s || (long)u_lo >= seg_e) continue;
    const int pcost = a_.p[pi].cost;
    const long f0 = (long)u_lo > seg_s ? ((long)u_lo - seg_s + pcost - 1) / pcost : 0;
    const long f1 = (long)u_hi < seg_e ? ((long)u_hi - seg_s + pcost - 1) / pcost : a_.p[pi].n_img;
    if (f1 <= f0) continue;
    const EFProblem P = a_.p[pi];
    // (lane indices behind an opaque zero, per problem: what depends on them is not hoisted out of this loop - see
    // encoder_fused_kernel)
    int tid;
    asm volatile("v_mov_b32 %0, 0" : "=v"(tid));
    tid += threadIdx.x;
    const int w = tid >> 6, l = tid & 63, r16 = l & 15, g = l >> 4;
    const int wu = __builtin_amdgcn_readfirstlane(w);
    // conv2: wave = (channel half, tile parity), as encoder_fused_kernel.  Measured and rejected (same-box A/B, 2 368 images,
    // scratch/r6_ring_ab.sh): wave = ONE 16-channel tile of every pixel tile - balanced for any tile count per band, where the
    // parity split idles a wave for a 32-MFMA tile whenever a band brings an odd count - 0.2628 -> 0.2815 ms: every fragment
    // read then feeds one MFMA instead of two, and reads are what a lone wave per SIMD pays for; the band's DMA pieces issued
    // by waves 1 - 3 only (wave 0 carries conv1's 13th tile) - 0.2626 -> 0.2690 ms.
    const int cg = w & 1, ph = w >> 1;

    unsigned char* const act1 = lds + 2 * G::LDS_IMG;
    unsigned char* const act2 = act1 + G::ACT1_BYTES;
    unsigned char* const sa = act2 + G::ACT2_BYTES;
    unsigned char* const h1 = sa + EF_CHUNK * SA_STRIDE;

    // ---- register-stationary weights and biases (fragment layouts: ef_pack_kernel)
    long po[11];
    {
      const long sz[11] = {32 * 8 * 8 * 3, 32, 64 * 4 * 4 * 32, 64, 64 * 3 * 3 * 64, 64, 1, 256 * 128, 256, 32 * 256, 32};
      long off = 0;
      for (int i = 0; i < 11; i++) { po[i] = off; off = (off + sz[i] + 3) & ~3L; }
    }
    u32x4 wc1a[6], wc1b[6], wc2a[16], wc2b[16], wc3[18];  // AGPR-resident (248 AGPRs)
#pragma unroll
    for (int s = 0; s < 6; s++) { wc1a[s] = P.wpk[WP_C1 + s * 64 + l]; wc1b[s] = P.wpk[WP_C1 + (6 + s) * 64 + l]; }
#pragma unroll
    for (int s = 0; s < 16; s++) {
      wc2a[s] = P.wpk[WP_C2 + ((2 * cg) * 16 + s) * 64 + l];
      wc2b[s] = P.wpk[WP_C2 + ((2 * cg + 1) * 16 + s) * 64 + l];
    }
#pragma unroll
    for (int s = 0; s < 18; s++) wc3[s] = P.wpk[WP_C3 + (w * 18 + s) * 64 + l];
    f32x4 bias1a, bias1b, bias2a, bias2b, bias3;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      bias1a[q] = P.params[po[1] + 8 * g + q];  // conv1 tile j, row 4 g + q <-> channel 8 g + 4 j + q
      bias1b[q] = P.params[po[1] + 8 * g + 4 + q];
      bias2a[q] = P.params[po[3] + 32 * cg + 8 * g + q];  // fragment 2 cg + ct, row 4 g + q <-> channel 32 cg + 8 g + 4 ct + q
      bias2b[q] = P.params[po[3] + 32 * cg + 8 * g + 4 + q];
      bias3[q] = P.params[po[5] + 16 * w + 4 * g + q];
    }
    const float temp = P.params[po[6]];
    asm volatile("" ::"v"(bias1a), "v"(bias1b), "v"(bias2a), "v"(bias2b), "v"(bias3), "v"(temp));

    // ---- image bands by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB per wave-instruction, uniform 64-bit base
    // + per-lane offset, LDS destination in M0).  Wave w moves the band's chunks [256 NPIECE w, 256 NPIECE (w + 1)).
    const unsigned dma_voff = (unsigned)l * 16u;
    const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)lds;
    // pieces [i0, i1) of this wave's share of (image, band)
    auto dma_pieces = [&](long img_idx, int band, int bufi, int i0, int i1) {
      const int row0 = 4 * ER_BR * band;
      const int rows = min(G::BAND_ROWS, G::H - row0);
      const int n16 = rows * (G::ROW_BYTES / 16);
      const unsigned char* src = reinterpret_cast<const unsigned char*>(P.img) + img_idx * G::IMG_BYTES + row0 * G::ROW_BYTES;
      const int c0w = wu * (G::NPIECE * 64);
#pragma unroll
      for (int i = i0; i < i1; i++) {
        if (i >= G::NPIECE) break;
        const int c0 = c0w + 64 * i;  // first chunk of this piece (wave-uniform)
        if (c0 < n16) {
          if (c0 + l < n16) {
            unsigned keep_m0;  // (M0 saved and restored around the piece: hipcc rejects "m0" as a clobber)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep_m0)
                         : "v"(dma_voff), "s"(__builtin_amdgcn_readfirstlane(lds_base + bufi * G::LDS_IMG + c0 * 16)),
                           "s"(src + (long)c0 * 16)
                         : "memory");
          }
        }
      }
    };
    auto dma_band = [&](long img_idx, int band, int bufi) { dma_pieces(img_idx, band, bufi, 0, G::NPIECE); };
    // In the image loop a band's pieces ride in conv1's MFMA chains, DPT per tile behind the second and fourth MFMA of the tile's
    // first chain (every wave has at least PER1 - 1 tiles): issued in one burst in front of conv1 a piece stands 100 - 185 clk at
    // issue (MI355X_MICROARCH.md, LDS-DMA piece issue cost), in the shadow of a running MFMA part of that is hidden.
    constexpr int DPT = (G::NPIECE + G::PER1 - 2) / (G::PER1 - 1);
    static_assert(G::PER1 >= 2 && DPT <= 2 && G::NT1 >= 4 * (G::PER1 - 1), "a tile's first chain places at most two DMA pieces");

    long cur = f0;
    int it = 0, buf = 0;
    dma_band(cur, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) as the builtin: it also empties hipcc's own scoreboard (weights, biases)
    __syncthreads();

    while (true) {
      const int slot = it & (EF_CHUNK - 1);
      const bool has_next = cur + 1 < f1;
      int t2_lo = 0;  // conv2 tiles done so far
#pragma unroll 1
      for (int band = 0; band < G::NB; band++) {
        // the other buffer was last read by the previous band's conv1, two barriers ago
        // what streams in beside this band's conv1: the next band, or the next image's first (wave-uniform)
        const bool dma_on = band + 1 < G::NB || has_next;
        const long dma_img = band + 1 < G::NB ? cur : cur + 1;
        const int dma_bnd = band + 1 < G::NB ? band + 1 : 0;
        // ------------------------------------------------ conv1: 8x8 stride 4, 3 -> 32, this band's 4 rows
        {
          const unsigned char* ib = lds + buf * G::LDS_IMG;
          const int k1g = g * G::ROW_BYTES;  // lane group g reads image row 4 (s / 3) + g
          auto px1 = [&](int mt, int& oy, int& ox) -> bool {
            const int pb = mt * 16 + r16, pc = min(pb, G::NPXB - 1);
            oy = pc / G::OW1; ox = pc - oy * G::OW1;
            return pb < G::NPXB;
          };
          auto base1 = [&](int oy, int ox) { return ib + (4 * oy * G::W + 4 * ox) * 6 + k1g; };
          auto ld1one = [&](const unsigned char* base, int s) -> u32x4 {
            const int off = (s / 3) * 4 * G::ROW_BYTES + (s % 3) * 16;  // compile-time immediate
            const u32x2 lo = *reinterpret_cast<const u32x2*>(base + off);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(base + off + 8);
            return u32x4{lo[0], lo[1], hi[0], hi[1]};
          };
          int OY[G::PER1], OX[G::PER1];
          bool OKP[G::PER1];
#pragma unroll
          for (int i = 0; i < G::PER1; i++) OKP[i] = px1(min(w + 4 * i, G::NT1 - 1), OY[i], OX[i]);
          u32x4 fa[6], fb[6];
          {
            const unsigned char* b0 = base1(OY[0], OX[0]);
#pragma unroll
            for (int s = 0; s < 6; s++) fa[s] = ld1one(b0, s);
          }
          auto tile1 = [&](int i, u32x4 (&fc)[6], u32x4 (&fn)[6]) {
            const bool pre = i + 1 < G::PER1;  // (a tile beyond the band's last is clamped: harmless reads)
            const unsigned char* nb = base1(OY[pre ? i + 1 : i], OX[pre ? i + 1 : i]);
            f32x4 A0, A1;
            __builtin_amdgcn_sched_barrier(0);
            MFMA_FIRST_AW(A0, wc1a[0], fc[0], bias1a);
#pragma unroll
            for (int s = 1; s < 6; s++) {
              MFMA_AW(A0, wc1a[s], fc[s]);
              if (pre) fn[s - 1] = ld1one(nb, s - 1);
              if ((s == 1 || s == 3) && (s >> 1) < DPT && dma_on) dma_pieces(dma_img, dma_bnd, buf ^ 1, i * DPT + (s >> 1), i * DPT + (s >> 1) + 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            MFMA_FIRST_AW(A1, wc1b[0], fc[0], bias1b);
            if (pre) fn[5] = ld1one(nb, 5);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 1; s < 6; s++) MFMA_AW(A1, wc1b[s], fc[s]);
            MFMA_CHAIN_END(A1);
            asm volatile("" : "+v"(A0));  // (its chain ended six MFMAs earlier)
            if (OKP[i]) {
              const int oyi = band * ER_BR + OY[i];
              const u32x2 lo = pack4_bf16(relu1(A0[0]), relu1(A0[1]), relu1(A0[2]), relu1(A0[3]));
              const u32x2 hi = pack4_bf16(relu1(A1[0]), relu1(A1[1]), relu1(A1[2]), relu1(A1[3]));
              *reinterpret_cast<u32x4*>(act1 + (oyi & (ER_R1 - 1)) * G::PITCH1 + OX[i] * ACT1_STRIDE + 16 * g) =
                  u32x4{lo[0], lo[1], hi[0], hi[1]};  // channels 8 g .. 8 g + 7
              if (P.act) {  // saved for the backward (uniform base + 32-bit lane offset), [pixel][32]:
                if (a_.act_bf16) {  // ... as bf16 at the start of y1's slot - the very value conv2 consumed (tacorl_encoder_bwd_fused*)
                  *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(P.act) + (long)cur * (G::NPX1 * 32) +
                                            (unsigned)((oyi * G::OW1 + OX[i]) * 32 + 8 * g)) = u32x4{lo[0], lo[1], hi[0], hi[1]};
                } else {            // ... as fp32 (the per-layer tacorl_encoder_bwd)
                  float* y = P.act + (long)cur * (G::NPX1 * 32) + (unsigned)((oyi * G::OW1 + OX[i]) * 32 + 8 * g);
                  *reinterpret_cast<f32x4*>(y) = f32x4{relu1(A0[0]), relu1(A0[1]), relu1(A0[2]), relu1(A0[3])};
                  *reinterpret_cast<f32x4*>(y + 4) = f32x4{relu1(A1[0]), relu1(A1[1]), relu1(A1[2]), relu1(A1[3])};
                }
              }
            }
          };
#pragma unroll
          for (int i = 0; i < G::PER1; i++) {
            if (wu + 4 * i < G::NT1) {
              if (i & 1) tile1(i, fb, fa); else tile1(i, fa, fb);
            }
          }
        }
        // this wave's conv1 rows are written; the barrier publishes them and "everyone is done reading this band" (the next
        // band's pieces have until the barrier behind conv2 to land)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf ^= 1;

        // ------------------------------------------------ conv2: 4x4 stride 2, 32 -> 64, the tiles this band completed
        {
          const int t2_hi = band >= G::NB - 1 ? G::NT2 : (min(2 * band + 1, G::OH2) * G::OW2) >> 4;
          const int t0 = t2_lo + ((t2_lo ^ ph) & 1);  // this wave's first tile: parity ph (wave-uniform)
          const int t0u = __builtin_amdgcn_readfirstlane(t0), t2_hiu = __builtin_amdgcn_readfirstlane(t2_hi);
          auto base2 = [&](int t, const unsigned char*& lo, const unsigned char*& hi) {
            const int pc = min(t * 16 + r16, G::NPX2 - 1);
            const int oy = pc / G::OW2, ox = pc - oy * G::OW2;
            const int off = 2 * ox * ACT1_STRIDE + 16 * g;
            lo = act1 + ((2 * oy) & (ER_R1 - 1)) * G::PITCH1 + off;
            hi = act1 + ((2 * oy + 2) & (ER_R1 - 1)) * G::PITCH1 + off;
          };
          auto ld2one = [&](const unsigned char* lo, const unsigned char* hi, int s) -> u32x4 {
            const int ky = s >> 2, kx = s & 3;
            return *reinterpret_cast<const u32x4*>((ky < 2 ? lo : hi) + (ky & 1) * G::PITCH1 + kx * ACT1_STRIDE);
          };
          auto tile2 = [&](int t, bool pre, u32x4 (&fc)[16], u32x4 (&fn)[16]) {
            const unsigned char *nlo, *nhi;
            base2(pre ? t + 2 : t, nlo, nhi);
            f32x4 C0, C1;
            __builtin_amdgcn_sched_barrier(0);
            MFMA_FIRST_AW(C0, wc2a[0], fc[0], bias2a);
#pragma unroll
            for (int s = 1; s < 16; s++) {
              MFMA_AW(C0, wc2a[s], fc[s]);
              if (pre) fn[s - 1] = ld2one(nlo, nhi, s - 1);
              __builtin_amdgcn_sched_barrier(0);
            }
            MFMA_FIRST_AW(C1, wc2b[0], fc[0], bias2b);
            if (pre) fn[15] = ld2one(nlo, nhi, 15);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 1; s < 16; s++) MFMA_AW(C1, wc2b[s], fc[s]);
            MFMA_CHAIN_END(C1);
            asm volatile("" : "+v"(C0));
            const int pm = t * 16 + r16;
            if (pm < G::NPX2) {
              const u32x2 lo = pack4_bf16(relu1(C0[0]), relu1(C0[1]), relu1(C0[2]), relu1(C0[3]));
              const u32x2 hi = pack4_bf16(relu1(C1[0]), relu1(C1[1]), relu1(C1[2]), relu1(C1[3]));
              const int oy2 = pm / G::OW2, ox2 = pm - oy2 * G::OW2;
              *reinterpret_cast<u32x4*>(act2 + oy2 * G::PITCH2 + ox2 * G::PX2 + (32 * cg + 8 * g) * 2) =
                  u32x4{lo[0], lo[1], hi[0], hi[1]};
              if (P.act) {
                if (a_.act_bf16) {
                  *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(P.act + P.a_y2) + (long)cur * (G::NPX2 * 64) +
                                            (unsigned)(pm * 64 + 32 * cg + 8 * g)) = u32x4{lo[0], lo[1], hi[0], hi[1]};
                } else {
                  float* y = P.act + P.a_y2 + (long)cur * (G::NPX2 * 64) + (unsigned)(pm * 64 + 32 * cg + 8 * g);
                  *reinterpret_cast<f32x4*>(y) = f32x4{relu1(C0[0]), relu1(C0[1]), relu1(C0[2]), relu1(C0[3])};
                  *reinterpret_cast<f32x4*>(y + 4) = f32x4{relu1(C1[0]), relu1(C1[1]), relu1(C1[2]), relu1(C1[3])};
                }
              }
            }
          };
          if (t0u < t2_hiu) {
            u32x4 fa[16], fb[16];
            const unsigned char *lo, *hi;
            base2(t0u, lo, hi);
#pragma unroll
            for (int s = 0; s < 16; s++) fa[s] = ld2one(lo, hi, s);
            const bool two = t0u + 2 < t2_hiu;
            if (two) {
              tile2(t0u, true, fa, fb);
              tile2(t0u + 2, false, fb, fa);
            } else {
              tile2(t0u, false, fa, fb);
            }
          }
          t2_lo = t2_hi;
        }
        // this wave's share of the next band has landed; conv2's reads of the ring are over (the next band overwrites four
        // slots), its rows are visible
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }

      // ------------------------------------------------ conv3: 3x3 stride 1, 64 -> 64, online soft-argmax per lane
      {
        const float inv_t = (1.0f / temp) * 1.44269504088896f;  // log2(e) folded in: the exponentials are exp2
        constexpr float NEG = -3.0e38f;  // "no pixel yet" / padding lanes (finite: differences of two of them are 0, not NaN;
                                         // such a lane's sums vanish when the lanes are merged against the row's real maximum)
        float m[4], se[4], sx[4], sy[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { m[q] = NEG; se[q] = 0.f; sx[q] = 0.f; sy[q] = 0.f; }
        auto base3 = [&](int mt, float& fxv, float& fyv, bool& ok) {
          const int pb = mt * 16 + r16, pc = min(pb, G::NPX3 - 1);
          const int oy = pc / G::OW3, ox = pc - oy * G::OW3;
          ok = pb < G::NPX3; fxv = (float)ox; fyv = (float)oy;
          return act2 + oy * G::PITCH2 + ox * G::PX2 + 16 * g;
        };
        auto ld3one = [&](const unsigned char* base, int s) -> u32x4 {
          const int tap = s >> 1, ky = tap / 3, kx = tap - 3 * ky;
          return *reinterpret_cast<const u32x4*>(base + ky * G::PITCH2 + kx * G::PX2 + 64 * (s & 1));
        };
        // one pixel per lane joins the lane's running state
        auto join = [&](const f32x4& acc, float fxv, float fyv, bool ok) {
          if (P.act && ok) {  // y3 (fp32, [pixel][64]); the lane's pixel index back from its coordinates
            const int px = (int)fyv * G::OW3 + (int)fxv;
            *reinterpret_cast<f32x4*>(P.act + P.a_y3 + (long)cur * (G::NPX3 * 64) + (unsigned)(px * 64 + 16 * w + 4 * g)) =
                f32x4{relu1(acc[0]), relu1(acc[1]), relu1(acc[2]), relu1(acc[3])};
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float v = ok ? relu1(acc[q]) * inv_t : NEG;
            const float M = fmaxf(m[q], v);
            const float ea = __builtin_amdgcn_exp2f(m[q] - M), eb = __builtin_amdgcn_exp2f(v - M);
            m[q] = M;
            se[q] = __builtin_fmaf(se[q], ea, eb);
            sx[q] = __builtin_fmaf(sx[q], ea, eb * fxv);
            sy[q] = __builtin_fmaf(sy[q], ea, eb * fyv);
          }
        };
        u32x4 fa[18], fb[18];
        float fx0, fy0, fx1 = 0.f, fy1 = 0.f;
        bool ok0, ok1 = false;
        f32x4 acc0, acc1 = {0.f, 0.f, 0.f, 0.f};
        {
          const unsigned char* b0 = base3(0, fx0, fy0, ok0);
#pragma unroll
          for (int s = 0; s < 18; s++) fa[s] = ld3one(b0, s);
        }
        // tile mt on fragment set fc, the next tile's fragments into fn, the PREVIOUS tile's pixels joined behind the fifth MFMA
        // (its accumulator is long past the read-after-MFMA wait there, and the VALU work sits in MFMA shadows)
        auto tile3 = [&](int mt, u32x4 (&fc)[18], u32x4 (&fn)[18], f32x4& acc, float& nfx, float& nfy, bool& nok,
                         const f32x4& pacc, float pfx, float pfy, bool pok, bool have_prev) {
          const unsigned char* nb = base3(min(mt + 1, G::NT3 - 1), nfx, nfy, nok);
          __builtin_amdgcn_sched_barrier(0);
          MFMA_FIRST_AW(acc, wc3[0], fc[0], bias3);
#pragma unroll
          for (int s = 1; s < 18; s++) {
            MFMA_AW(acc, wc3[s], fc[s]);
            fn[s - 1] = ld3one(nb, s - 1);
            if (s == 4 && have_prev) {
              asm volatile("" : "+v"(const_cast<f32x4&>(pacc)));
              join(pacc, pfx, pfy, pok);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          fn[17] = ld3one(nb, 17);
          __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll 1
        for (int mt = 0; mt < G::NT3; mt += 2) {
          tile3(mt, fa, fb, acc0, fx1, fy1, ok1, acc1, fx1, fy1, ok1, mt > 0);
          if (mt + 1 < G::NT3) {
            float nfx, nfy; bool nok;
            tile3(mt + 1, fb, fa, acc1, nfx, nfy, nok, acc0, fx0, fy0, ok0, true);
            fx0 = nfx; fy0 = nfy; ok0 = nok;
          }
        }
        if (G::NT3 & 1) { MFMA_CHAIN_END(acc0); join(acc0, fx0, fy0, ok0); }
        else { MFMA_CHAIN_END(acc1); join(acc1, fx1, fy1, ok1); }
        // merge the 16 pixel lanes of a row: common maximum, rescaled sums
        float mx[4] = {m[0], m[1], m[2], m[3]};
        EF_ROW16_4("v_max_f32_dpp", mx);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float r = __builtin_amdgcn_exp2f(m[q] - mx[q]);
          se[q] *= r; sx[q] *= r; sy[q] *= r;
        }
        EF_ROW16_4("v_add_f32_dpp", se);
        EF_ROW16_4("v_add_f32_dpp", sx);
        EF_ROW16_4("v_add_f32_dpp", sy);
        if (r16 == 0) {  // features interleaved [x_c, y_c]; channels 16 w + 4 g + q
          float fx_[4], fy_[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float r = __builtin_amdgcn_rcpf(se[q]);
            fx_[q] = sx[q] * r; fy_[q] = sy[q] * r;
          }
          const u32x2 lo = pack4_bf16(fx_[0], fy_[0], fx_[1], fy_[1]), hi = pack4_bf16(fx_[2], fy_[2], fx_[3], fy_[3]);
          *reinterpret_cast<u32x4*>(sa + slot * SA_STRIDE + 4 * (16 * w + 4 * g)) = u32x4{lo[0], lo[1], hi[0], hi[1]};
          if (P.act) {
            float* f = P.act + P.a_sa + (long)cur * 128 + (unsigned)(2 * (16 * w + 4 * g));
            *reinterpret_cast<f32x4*>(f) = f32x4{fx_[0], fy_[0], fx_[1], fy_[1]};
            *reinterpret_cast<f32x4*>(f + 4) = f32x4{fx_[2], fy_[2], fx_[3], fy_[3]};
          }
        }
      }

      // ------------------------------------------------ FC tail once per chunk of EF_CHUNK images (images = MFMA columns)
      const bool chunk_done = (slot == EF_CHUNK - 1) || !has_next;
      if (chunk_done) {
        u32x4 wf1[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int s = 0; s < 4; s++) wf1[j][s] = P.wpk[WP_F1 + ((4 * w + j) * 4 + s) * 64 + l];
        f32x4 bf1[4];
#pragma unroll
        for (int j = 0; j < 4; j++) bf1[j] = *reinterpret_cast<const f32x4*>(P.params + po[8] + 16 * (4 * w + j) + 4 * g);
        u32x4 wf2[8];
#pragma unroll
        for (int s = 0; s < 8; s++) wf2[s] = P.wpk[WP_F2 + ((w & 1) * 8 + s) * 64 + l];
        const f32x4 bf2 = *reinterpret_cast<const f32x4*>(P.params + po[10] + 16 * (w & 1) + 4 * g);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // soft-argmax features of the whole chunk are in LDS
        const int n_in_chunk = slot + 1;
        const long img0 = f0 + (it - slot);  // first image of the chunk
        {
          const unsigned char* base = sa + r16 * SA_STRIDE + 16 * g;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; s++)
              acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wf1[j][s]), as_bf16x8(*reinterpret_cast<const u32x4*>(base + 64 * s)), acc, 0, 0, 0);
            const f32x4 r = {fmaxf(acc[0] + bf1[j][0], 0.f), fmaxf(acc[1] + bf1[j][1], 0.f), fmaxf(acc[2] + bf1[j][2], 0.f),
                             fmaxf(acc[3] + bf1[j][3], 0.f)};
            *reinterpret_cast<u32x2*>(h1 + r16 * H1_STRIDE + (16 * (4 * w + j) + 4 * g) * 2) = pack4_bf16(r[0], r[1], r[2], r[3]);
            if (P.act && r16 < n_in_chunk)
              *reinterpret_cast<f32x4*>(P.act + P.a_h1 + img0 * 256 + (unsigned)(r16 * 256 + 16 * (4 * w + j) + 4 * g)) = r;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (w < 2) {
          const unsigned char* base = h1 + r16 * H1_STRIDE + 16 * g;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 8; s++)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wf2[s]), as_bf16x8(*reinterpret_cast<const u32x4*>(base + 64 * s)), acc, 0, 0, 0);
          if (r16 < n_in_chunk) {
            float* o = P.out + img0 * 32 + (unsigned)(r16 * 32 + 16 * w + 4 * g);
            *reinterpret_cast<f32x4*>(o) = f32x4{acc[0] + bf2[0], acc[1] + bf2[1], acc[2] + bf2[2], acc[3] + bf2[3]};
          }
        }
        // (the next chunk's first soft-argmax store comes a whole image - many barriers - later: no barrier here)
      }
      if (!has_next) break;
      cur++; it++;
    }
    __syncthreads();  // (a workgroup that goes on with the next problem: nobody still reads what its prologue overwrites)
  }
}

#define ER_GEOMS(X) X(150, 200)

int ef_ring_supported(int H, int W) {
#define X(h, w) if (H == h && W == w) return ERGeom<h, w>::OK ? 1 : 0;
  ER_GEOMS(X)
#undef X
  return 0;
}

template <int H, int W>
static int er_launch(EFArgs& a, int nb, hipStream_t st) {
  static bool attr_set = false;
  auto kfn = encoder_ring_kernel<H, W>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  constexpr size_t lds_bytes = ERGeom<H, W>::LDS_BYTES;
  hipLaunchKernelGGL(kfn, dim3(nb), dim3(256), lds_bytes, st, a);
  return hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH;
}

int ef_ring_launch(EFArgs& a, int nb, int H, int W, hipStream_t st) {
#define X(h, w) if (H == h && W == w) return er_launch<h, w>(a, nb, st);
  ER_GEOMS(X)
#undef X
  return TACORL_EINVAL;
}
