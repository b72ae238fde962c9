// HBM-/latency-bound kernels of the CQL / TACO-RL step: image packing, tanh-Gaussian
// sampling + log-prob, the fused Bellman + CQL logsumexp loss (fwd+bwd in one pass, wave
// shuffle reductions), the twin-Q min for the actor loss, and the fused clip+Adam+Polyak.
#include <stdio.h>

#include "../../include/tacorl_hip.h"
#include "common.h"

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH)

__device__ __forceinline__ float softplusf(float x) {  // F.softplus, threshold 20
  return x > 20.f ? x : log1pf(expf(x));
}
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// =================================================================== pack_images
// src: n images, image i at src + i*img_pitch (elements); layout NCHW (src_nchw) or NHWC.
// dst: contiguous NHWC, fp32 or bf16.
template <typename OutT>
__global__ void pack_images_kernel(const float* __restrict__ src, long img_pitch, int src_nchw,
                                   OutT* __restrict__ dst, int n, int C, int HW) {
  const long total = (long)n * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long img = i / HW;
    const int px = (int)(i - img * HW);
    const float* s = src + img * img_pitch;
    OutT* d = dst + i * C;
    for (int c = 0; c < C; c++) d[c] = (OutT)(src_nchw ? s[(long)c * HW + px] : s[(long)px * C + c]);
  }
}
extern "C" int tacorl_pack_images(const float* src, long img_pitch, int src_nchw, void* dst, int dst_dtype,
                                  int n, int C, int H, int W, tacorl_stream_t stream) {
  if (n <= 0) return TACORL_OK;
  const long total = (long)n * H * W;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dst_dtype == TACORL_BF16)
    hipLaunchKernelGGL(pack_images_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, img_pitch,
                       src_nchw, (__bf16*)dst, n, C, H * W);
  else
    hipLaunchKernelGGL(pack_images_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, img_pitch,
                       src_nchw, (float*)dst, n, C, H * W);
  return LAUNCH_OK();
}

// Several NCHW fp32 -> NHWC packs in one launch (blockIdx.y = job), 4 pixels per thread: three 16 B
// plane reads, 24 B (bf16) or 48 B (fp32) of contiguous interleaved output.  C = 3, H*W % 4 == 0.
#define PACK_MAXJ 8
struct PackTbl {
  const float* src[PACK_MAXJ];
  void* dst[PACK_MAXJ];
  long pitch[PACK_MAXJ];
  int n[PACK_MAXJ];
  // optional second destinations of a WINDOW job (dupT[j] = T > 0): image i = b T + t also goes to dupA[j] + b (t == 0)
  // or dupB[j] + b (t == T - 1) - the transition's obs / next images ARE the window's first / last frames (reference
  // tacorl.py:147-168 get_rl_batch: s = states[:, 0], s' = states[:, -1]), so they are read once
  void* dupA[PACK_MAXJ];
  void* dupB[PACK_MAXJ];
  int dupT[PACK_MAXJ];
};
template <typename OutT>
__global__ void pack_nchw3_batch_kernel(PackTbl t, int HW) {
  const int j = blockIdx.y, q4 = HW / 4;
  const long total = (long)t.n[j] * q4;
  const float* __restrict__ src = t.src[j];
  OutT* __restrict__ dst = reinterpret_cast<OutT*>(t.dst[j]);
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const long img = q / q4;
    const int p = (int)(q - img * q4) * 4;
    const float* s = src + img * t.pitch[j] + p;
    const f32x4 r = *reinterpret_cast<const f32x4*>(s), g = *reinterpret_cast<const f32x4*>(s + HW),
                b = *reinterpret_cast<const f32x4*>(s + 2 * HW);
    const float v[12] = {r[0], g[0], b[0], r[1], g[1], b[1], r[2], g[2], b[2], r[3], g[3], b[3]};
    OutT* d = dst + (img * HW + p) * 3;
    OutT* d2 = nullptr;
    if (t.dupT[j] > 0) {
      const long b = img / t.dupT[j];
      const int tt = (int)(img - b * t.dupT[j]);
      OutT* base = tt == 0 ? reinterpret_cast<OutT*>(t.dupA[j]) : (tt == t.dupT[j] - 1 ? reinterpret_cast<OutT*>(t.dupB[j]) : nullptr);
      if (base) d2 = base + (b * HW + p) * 3;
    }
    if constexpr (sizeof(OutT) == 2) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const bf16x4 o = bf16x4{(__bf16)v[4 * k], (__bf16)v[4 * k + 1], (__bf16)v[4 * k + 2], (__bf16)v[4 * k + 3]};
        *reinterpret_cast<bf16x4*>(d + 4 * k) = o;
        if (d2) *reinterpret_cast<bf16x4*>(d2 + 4 * k) = o;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const f32x4 o = f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
        *reinterpret_cast<f32x4*>(d + 4 * k) = o;
        if (d2) *reinterpret_cast<f32x4*>(d2 + 4 * k) = o;
      }
    }
  }
}
extern "C" int tacorl_pack_images_batch(int njobs, const float* const* src, const long* img_pitch, void* const* dst,
                                        const int* n_img, int dst_dtype, int H, int W, tacorl_stream_t stream) {
  return tacorl_pack_images_window_batch(njobs, src, img_pitch, dst, n_img, nullptr, nullptr, nullptr, dst_dtype, H, W, stream);
}
extern "C" int tacorl_pack_images_window_batch(int njobs, const float* const* src, const long* img_pitch, void* const* dst,
                                               const int* n_img, void* const* dst_first, void* const* dst_last,
                                               const int* window, int dst_dtype, int H, int W, tacorl_stream_t stream) {
  if (njobs < 1 || njobs > PACK_MAXJ || (H * W) % 4) return TACORL_EINVAL;
  PackTbl t{};
  long mx = 0;
  int m = 0;
  for (int j = 0; j < njobs; j++) {
    if (n_img[j] <= 0) continue;
    if (((uintptr_t)src[j] & 15) || ((uintptr_t)dst[j] & 15) || img_pitch[j] % 4) return TACORL_EINVAL;
    const int T = window ? window[j] : 0;
    if (T < 0 || (T > 0 && (T < 2 || n_img[j] % T || !dst_first || !dst_last || !dst_first[j] || !dst_last[j] ||
                            (((uintptr_t)dst_first[j] | (uintptr_t)dst_last[j]) & 15))))
      return TACORL_EINVAL;
    t.dupT[m] = T; t.dupA[m] = T ? dst_first[j] : nullptr; t.dupB[m] = T ? dst_last[j] : nullptr;
    t.src[m] = src[j]; t.dst[m] = dst[j]; t.pitch[m] = img_pitch[j]; t.n[m] = n_img[j];
    const long tot = (long)n_img[j] * (H * W / 4);
    mx = tot > mx ? tot : mx;
    m++;
  }
  if (m == 0) return TACORL_OK;
  const int blocks = (int)((mx + 255) / 256 > 8192 ? 8192 : (mx + 255) / 256);
  if (dst_dtype == TACORL_BF16)
    hipLaunchKernelGGL(pack_nchw3_batch_kernel<__bf16>, dim3(blocks, m), dim3(256), 0, (hipStream_t)stream, t, H * W);
  else
    hipLaunchKernelGGL(pack_nchw3_batch_kernel<float>, dim3(blocks, m), dim3(256), 0, (hipStream_t)stream, t, H * W);
  return LAUNCH_OK();
}

// ===================================================================== uint8 frames (dataset format)
// The play dataset stores frames as uint8 HWC (reference datamodule/dataset/play_dataset.py, one .npz per frame);
// the reference's CPU transform pipeline turns them into fp32 CHW with ToTensor (x / 255) and Normalize(0.5, 0.5)
// ((t - 0.5) / 0.5) before the batch reaches the module (config/.../rl_train.yaml:12-14).  Shipping the uint8
// frames and normalising here - the same two fp32 operations, then the cast to the image dtype - reads a quarter
// of the bytes (and a quarter of the host link) and needs no layout change: the source is already NHWC.
// 16 bytes per thread; H*W*3 % 16 == 0, pitches % 16 == 0, 16-byte aligned pointers.
struct PackU8Tbl {
  const unsigned char* src[PACK_MAXJ];
  void* dst[PACK_MAXJ];
  long pitch[PACK_MAXJ];  // bytes between images
  const long* idx[PACK_MAXJ];  // optional frame ids: image i is frame idx[i * istride] of the dataset at src (replay gather)
  int istride[PACK_MAXJ];
  int n[PACK_MAXJ];
};
typedef unsigned int pk_u32x4 __attribute__((ext_vector_type(4)));
template <typename OutT>
__global__ void pack_u8_batch_kernel(PackU8Tbl t, int chunks) {  // chunks = H*W*3 / 16
  const int j = blockIdx.y;
  const long total = (long)t.n[j] * chunks;
  const unsigned char* __restrict__ src = t.src[j];
  OutT* __restrict__ dst = reinterpret_cast<OutT*>(t.dst[j]);
  const long* __restrict__ idx = t.idx[j];
  const int istride = t.istride[j];
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const long img = q / chunks;
    const int c = (int)(q - img * chunks);
    const long frame = idx ? idx[img * istride] : img;
    const pk_u32x4 v = *reinterpret_cast<const pk_u32x4*>(src + frame * t.pitch[j] + (long)c * 16);
    float f[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const float x = (float)((v[e >> 2] >> (8 * (e & 3))) & 0xffu);
      f[e] = (x / 255.0f - 0.5f) / 0.5f;  // ToTensor, Normalize(0.5, 0.5): the reference's fp32 operations
    }
    OutT* d = dst + (img * chunks + c) * 16;
    if constexpr (sizeof(OutT) == 2) {
#pragma unroll
      for (int k = 0; k < 4; k++)
        *reinterpret_cast<bf16x4*>(d + 4 * k) = bf16x4{(__bf16)f[4 * k], (__bf16)f[4 * k + 1], (__bf16)f[4 * k + 2], (__bf16)f[4 * k + 3]};
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) *reinterpret_cast<f32x4*>(d + 4 * k) = f32x4{f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]};
    }
  }
}
extern "C" int tacorl_pack_images_u8_batch(int njobs, const void* const* src, const long* img_pitch_bytes, void* const* dst,
                                           const int* n_img, int dst_dtype, int H, int W, tacorl_stream_t stream) {
  return tacorl_pack_images_u8_gather_batch(njobs, src, img_pitch_bytes, nullptr, nullptr, dst, n_img, dst_dtype, H, W, stream);
}
extern "C" int tacorl_pack_images_u8_gather_batch(int njobs, const void* const* src, const long* img_pitch_bytes,
                                                  const long* const* index, const int* index_stride, void* const* dst,
                                                  const int* n_img, int dst_dtype, int H, int W, tacorl_stream_t stream) {
  const long bytes = (long)H * W * 3;
  if (njobs < 1 || njobs > PACK_MAXJ || bytes % 16) return TACORL_EINVAL;
  PackU8Tbl t{};
  long mx = 0;
  int m = 0;
  for (int j = 0; j < njobs; j++) {
    if (n_img[j] <= 0) continue;
    if (((uintptr_t)src[j] & 15) || ((uintptr_t)dst[j] & 15) || img_pitch_bytes[j] % 16) return TACORL_EINVAL;
    t.src[m] = (const unsigned char*)src[j]; t.dst[m] = dst[j]; t.pitch[m] = img_pitch_bytes[j]; t.n[m] = n_img[j];
    t.idx[m] = index ? index[j] : nullptr;
    t.istride[m] = (index && index[j] && index_stride) ? index_stride[j] : 1;
    if (t.idx[m] && t.istride[m] < 1) return TACORL_EINVAL;
    const long tot = (long)n_img[j] * (bytes / 16);
    mx = tot > mx ? tot : mx;
    m++;
  }
  if (m == 0) return TACORL_OK;
  const int blocks = (int)((mx + 255) / 256 > 8192 ? 8192 : (mx + 255) / 256);
  if (dst_dtype == TACORL_BF16)
    hipLaunchKernelGGL(pack_u8_batch_kernel<__bf16>, dim3(blocks, m), dim3(256), 0, (hipStream_t)stream, t, (int)(bytes / 16));
  else
    hipLaunchKernelGGL(pack_u8_batch_kernel<float>, dim3(blocks, m), dim3(256), 0, (hipStream_t)stream, t, (int)(bytes / 16));
  return LAUNCH_OK();
}

// ===================================================================== small transition tensors
// reward = done = (disp == 1) as float (TACORL.get_rl_batch, reference tacorl.py:142-179) and the action window
// copy, in one launch instead of four (compare, two casts, copy) in front of every step.
// disp_dtype: 0 = float32, 1 = int64, 2 = int32, 3 = uint8 / bool.
__global__ void stage_transition_kernel(const void* __restrict__ disp, int disp_dtype, float* __restrict__ reward,
                                        float* __restrict__ done, int B, const float* __restrict__ acts_src,
                                        float* __restrict__ acts_dst, long n_acts) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) {
    bool one;
    if (disp_dtype == 0) one = static_cast<const float*>(disp)[i] == 1.0f;
    else if (disp_dtype == 1) one = static_cast<const long long*>(disp)[i] == 1;
    else if (disp_dtype == 2) one = static_cast<const int*>(disp)[i] == 1;
    else one = static_cast<const unsigned char*>(disp)[i] == 1;
    const float r = one ? 1.0f : 0.0f;
    reward[i] = r;
    if (done) done[i] = r;
  }
  for (long k = i; k < n_acts; k += (long)gridDim.x * blockDim.x) acts_dst[k] = acts_src[k];
}
extern "C" int tacorl_stage_transition(const void* disp, int disp_dtype, float* reward, float* done, int B,
                                       const float* acts_src, float* acts_dst, long n_acts, tacorl_stream_t stream) {
  if (!disp || !reward || B < 1 || disp_dtype < 0 || disp_dtype > 3 || (n_acts > 0 && (!acts_src || !acts_dst))) return TACORL_EINVAL;
  const long work = n_acts > B ? n_acts : B;
  const int blocks = (int)((work + 255) / 256 > 1024 ? 1024 : (work + 255) / 256);
  hipLaunchKernelGGL(stage_transition_kernel, dim3(blocks < (B + 255) / 256 ? (B + 255) / 256 : blocks), dim3(256), 0,
                     (hipStream_t)stream, disp, disp_dtype, reward, done, B, acts_src, acts_dst, n_acts);
  return LAUNCH_OK();
}

// ===================================================================== copy_cols
__global__ void copy_cols_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst,
                                 int rows, int cols, int src_row_mod, int accumulate) {
  const long total = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    const int rs = src_row_mod > 0 ? r % src_row_mod : r;
    const float v = src[(long)rs * ld_src + c];
    float* d = dst + (long)r * ld_dst + c;
    *d = accumulate ? *d + v : v;
  }
}
// dst[r][0:cols] (+)= src[r % src_row_mod][0:cols]; pointers are pre-offset to the first column.
extern "C" int tacorl_copy_cols(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols,
                                int src_row_mod, int accumulate, tacorl_stream_t stream) {
  if (rows <= 0 || cols <= 0) return TACORL_OK;
  const long total = (long)rows * cols;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(copy_cols_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, dst, ld_dst, rows,
                     cols, src_row_mod, accumulate);
  return LAUNCH_OK();
}

// Up to 32 independent copy_cols in one launch (blockIdx.y = descriptor): the state / Q-input
// assembly is a dozen 32-256 column concatenations, each far smaller than a launch.
#define COPY_MAXB 32
struct CopyTbl {
  const float* src[COPY_MAXB];
  float* dst[COPY_MAXB];
  int ld_src[COPY_MAXB], ld_dst[COPY_MAXB], rows[COPY_MAXB], cols[COPY_MAXB], mod[COPY_MAXB], acc[COPY_MAXB];
};
__global__ void copy_cols_batch_kernel(CopyTbl t) {
  const int d = blockIdx.y, cols = t.cols[d], mod = t.mod[d], acc = t.acc[d];
  const long total = (long)t.rows[d] * cols;
  const float* __restrict__ src = t.src[d];
  float* __restrict__ dst = t.dst[d];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    const int rs = mod > 0 ? r % mod : r;
    const float v = src[(long)rs * t.ld_src[d] + c];
    float* o = dst + (long)r * t.ld_dst[d] + c;
    *o = acc ? *o + v : v;
  }
}
extern "C" int tacorl_copy_cols_batch(int n, const float* const* src, const int* ld_src, float* const* dst,
                                      const int* ld_dst, const int* rows, const int* cols, const int* src_row_mod,
                                      const int* accumulate, tacorl_stream_t stream) {
  if (n < 0 || n > COPY_MAXB) return TACORL_EINVAL;
  CopyTbl t{};
  long maxt = 0;
  int m = 0;
  for (int i = 0; i < n; i++) {
    if (rows[i] <= 0 || cols[i] <= 0) continue;
    t.src[m] = src[i]; t.dst[m] = dst[i]; t.ld_src[m] = ld_src[i]; t.ld_dst[m] = ld_dst[i]; t.rows[m] = rows[i];
    t.cols[m] = cols[i]; t.mod[m] = src_row_mod ? src_row_mod[i] : 0; t.acc[m] = accumulate ? accumulate[i] : 0;
    const long tot = (long)rows[i] * cols[i];
    maxt = tot > maxt ? tot : maxt;
    m++;
  }
  if (m == 0) return TACORL_OK;
  const int blocks = (int)((maxt + 255) / 256 > 512 ? 512 : (maxt + 255) / 256);
  hipLaunchKernelGGL(copy_cols_batch_kernel, dim3(blocks, m), dim3(256), 0, (hipStream_t)stream, t);
  return LAUNCH_OK();
}

// out[b][c] = sum_j in[(j*B + b)][c], j < reps   (gradient of a broadcast over samples).
// A workgroup owns 64 consecutive output elements; its four waves split the repetitions (wave w: j = w, w + 4, ...,
// eight loads in flight each) and meet in LDS, summed in wave order: fixed order, run-to-run identical.  (One thread
// per output element walking all repetitions was 390 us for 97 repetitions of 1024 x 64 - two workgroups per CU, each
// thread a chain of 97 round trips.)
__device__ __forceinline__ void reduce_rows_mod_body(const float* __restrict__ in, int ld_in, float* __restrict__ out, int ld_out,
                                                     int B, int cols, int reps) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long total = (long)B * cols;
  for (long i0 = (long)blockIdx.x * 64; i0 < total; i0 += (long)gridDim.x * 64) {
    const long i = i0 + lane;
    const bool on = i < total;
    const int b = on ? (int)(i / cols) : 0, c = on ? (int)(i - (long)b * cols) : 0;
    float s = 0.f;
    if (on) {
#pragma unroll 8
      for (int j = w; j < reps; j += 4) s += in[((long)j * B + b) * ld_in + c];
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && on) out[(long)b * ld_out + c] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void reduce_rows_mod_kernel(const float* __restrict__ in, int ld_in, float* __restrict__ out,
                                                              int ld_out, int B, int cols, int reps) {
  reduce_rows_mod_body(in, ld_in, out, ld_out, B, cols, reps);
}
static int rrm_blocks(long total) {
  const long nb = (total + 63) / 64;
  return (int)(nb < 8192 ? nb : 8192);
}
extern "C" int tacorl_reduce_rows_mod(const float* in, int ld_in, float* out, int ld_out, int B, int cols, int reps,
                                      tacorl_stream_t stream) {
  if (B <= 0) return TACORL_OK;
  hipLaunchKernelGGL(reduce_rows_mod_kernel, dim3(rrm_blocks((long)B * cols)), dim3(256), 0, (hipStream_t)stream, in,
                     ld_in, out, ld_out, B, cols, reps);
  return LAUNCH_OK();
}

// the same for up to 4 (in, out) pairs of one shape in one launch (blockIdx.y = pair): the critics' state gradients
struct RrmTbl { const float* in[4]; float* out[4]; };
__global__ __launch_bounds__(256) void reduce_rows_mod_batch_kernel(RrmTbl t, int ld_in, int ld_out, int B, int cols, int reps) {
  reduce_rows_mod_body(t.in[blockIdx.y], ld_in, t.out[blockIdx.y], ld_out, B, cols, reps);
}
extern "C" int tacorl_reduce_rows_mod_batch(int n, const float* const* in, int ld_in, float* const* out, int ld_out, int B,
                                            int cols, int reps, tacorl_stream_t stream) {
  if (n < 1 || n > 4) return TACORL_EINVAL;
  if (B <= 0) return TACORL_OK;
  RrmTbl t{};
  for (int k = 0; k < n; k++) { t.in[k] = in[k]; t.out[k] = out[k]; }
  hipLaunchKernelGGL(reduce_rows_mod_batch_kernel, dim3(rrm_blocks((long)B * cols), n), dim3(256), 0, (hipStream_t)stream, t,
                     ld_in, ld_out, B, cols, reps);
  return LAUNCH_OK();
}

// uniform random actions: dst[r][0:A] = 2u - 1 (last dim snapped to +-1 if discrete gripper)
// reference cql_offline_lightning.py:243-250
__global__ void uniform_actions_kernel(const float* __restrict__ u01, float* __restrict__ dst, int ld_dst, int rows,
                                       int A, int discrete_gripper) {
  const long total = (long)rows * A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / A), c = (int)(i - (long)r * A);
    float v = u01[i] * 2.0f - 1.0f;
    if (discrete_gripper && c == A - 1) v = v >= 0.f ? 1.f : -1.f;
    dst[(long)r * ld_dst + c] = v;
  }
}
extern "C" int tacorl_uniform_actions(const float* u01, float* dst, int ld_dst, int rows, int A, int discrete_gripper,
                                      tacorl_stream_t stream) {
  if (rows <= 0) return TACORL_OK;
  const long total = (long)rows * A;
  hipLaunchKernelGGL(uniform_actions_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u01,
                     dst, ld_dst, rows, A, discrete_gripper);
  return LAUNCH_OK();
}

// ================================================================ tanh-Gaussian
// head[m] = [mean_raw (Ac) | log_std_raw (Ac) | gripper logits (2, optional)]
// reference actor.py:259-265 (clamps), utils/distributions.py:78-140 (TanhNormal).
__device__ __forceinline__ void head_stats(const float* h, int j, int Ac, float& mu, float& sd) {
  mu = fminf(fmaxf(h[j], -9.0f), 9.0f);
  sd = expf(fminf(fmaxf(h[Ac + j], -5.0f), 2.0f));
}
__device__ __forceinline__ float normal_lp(float z, float mu, float sd) {
  const float d = z - mu;
  return -(d * d) / (2.f * (sd * sd)) - logf(sd) - 0.9189385332046727f;
}
__device__ __forceinline__ float tanh_corr(float z) {  // -2(log2 - z - softplus(-2z))
  return -2.f * (0.6931471805599453f - z - softplusf(-2.f * z));
}

// One thread per (sample k, row m): a = tanh(mu + sd*eps) written to act_out[(k*M+m)*ld_act .. +Ac),
// log pi to logp[k*M+m].  gumbel_u (n,M,2) U(0,1) adds the discrete gripper (actor.py:118-132 /
// :83-97): index = argmax(norm_logits - log(-log u)) or, for hard_rsample, the relaxed-categorical
// argmax at temperature 0.5; action +-1 appended, its log-softmax added to log pi.
__global__ void tanh_normal_sample_kernel(const float* __restrict__ head, int ld_head, const float* __restrict__ eps,
                                          const float* __restrict__ gumbel_u, int hard_rsample,
                                          float* __restrict__ act_out, int ld_act, float* __restrict__ logp,
                                          int* __restrict__ grip_idx, int n, int M, int Ac) {
  const long total = (long)n * M;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i % M);
    const float* h = head + (long)m * ld_head;
    float lp = 0.f, corr = 0.f;
    for (int j = 0; j < Ac; j++) {
      float mu, sd;
      head_stats(h, j, Ac, mu, sd);
      const float z = mu + eps[i * Ac + j] * sd;
      act_out[i * ld_act + j] = tanhf(z);
      lp += normal_lp(z, mu, sd);
      corr += (0.6931471805599453f - z - softplusf(-2.f * z));
    }
    lp += -2.f * corr;
    if (gumbel_u) {
      const float l0 = h[2 * Ac], l1 = h[2 * Ac + 1];
      const float mx = fmaxf(l0, l1);
      const float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
      const float n0 = l0 - lse, n1 = l1 - lse;
      float u0 = gumbel_u[i * 2], u1 = gumbel_u[i * 2 + 1];
      int idx;
      if (hard_rsample) {
        const float e = 1.1920929e-07f;
        u0 = fminf(fmaxf(u0, e), 1.f - e); u1 = fminf(fmaxf(u1, e), 1.f - e);
        const float s0 = (n0 - logf(-logf(u0))) / 0.5f, s1 = (n1 - logf(-logf(u1))) / 0.5f;
        idx = s1 > s0 ? 1 : 0;
      } else {
        idx = (n1 - logf(-logf(u1))) > (n0 - logf(-logf(u0))) ? 1 : 0;
      }
      // log_softmax of the normalised logits
      const float mm = fmaxf(n0, n1), l2 = mm + logf(expf(n0 - mm) + expf(n1 - mm));
      lp += (idx ? n1 : n0) - l2;
      act_out[i * ld_act + Ac] = idx ? 1.f : -1.f;
      if (grip_idx) grip_idx[i] = idx;
    }
    logp[i] = lp;
  }
}
extern "C" int tacorl_tanh_normal_sample(const float* head, int ld_head, const float* eps, const float* gumbel_u,
                                         int hard_rsample, float* act_out, int ld_act, float* logp, int* grip_idx,
                                         int n, int M, int Ac, tacorl_stream_t stream) {
  if (n <= 0 || M <= 0) return TACORL_OK;
  const long total = (long)n * M;
  hipLaunchKernelGGL(tanh_normal_sample_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     head, ld_head, eps, gumbel_u, hard_rsample, act_out, ld_act, logp, grip_idx, n, M, Ac);
  return LAUNCH_OK();
}

// All tanh-Gaussian draws of a step in one launch (blockIdx.y = job): 32 lanes per (sample, row), one
// per action dimension, log pi reduced with xor-shuffles - the one-thread-per-row kernel above leaves
// 1-4 workgroups with a serial loop over the dimensions (~13 us per call, 4 calls per step).
// Job j < njobs: same arguments as tacorl_tanh_normal_sample.  An optional job of uniform actions
// (u01 -> 2u-1, reference cql_offline_lightning.py:355-362) rides along as the last blockIdx.y.
#define TNS_MAXJ 6
struct TnsTbl {
  const float* head[TNS_MAXJ];
  const float* eps[TNS_MAXJ];
  const float* gumbel[TNS_MAXJ];
  float* act_out[TNS_MAXJ];
  float* logp[TNS_MAXJ];
  int* grip[TNS_MAXJ];
  int hard[TNS_MAXJ], n[TNS_MAXJ];
  const float* u01;
  float* u_dst;
  int u_rows, u_disc;
};
__global__ __launch_bounds__(256) void tanh_normal_sample_batch_kernel(TnsTbl t, int njobs, int M, int Ac, int ld_head,
                                                                       int ld_act, int A) {
  const int job = blockIdx.y;
  if (job == njobs) {  // uniform actions
    const long total = (long)t.u_rows * A;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int r = (int)(i / A), c = (int)(i - (long)r * A);
      float v = t.u01[i] * 2.0f - 1.0f;
      if (t.u_disc && c == A - 1) v = v >= 0.f ? 1.f : -1.f;
      t.u_dst[(long)r * ld_act + c] = v;
    }
    return;
  }
  const long rows = (long)t.n[job] * M;
  const int j = threadIdx.x & 31;
  for (long i = (long)blockIdx.x * 8 + (threadIdx.x >> 5); i < rows; i += (long)gridDim.x * 8) {
    const int m = (int)(i % M);
    const float* h = t.head[job] + (long)m * ld_head;
    float lp = 0.f;
    if (j < Ac) {
      float mu, sd;
      head_stats(h, j, Ac, mu, sd);
      const float z = mu + t.eps[job][i * Ac + j] * sd;
      t.act_out[job][i * ld_act + j] = tanhf(z);
      lp = normal_lp(z, mu, sd) + tanh_corr(z);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) lp += __shfl_xor(lp, o, 64);
    if (j == 0) {
      if (t.gumbel[job]) {
        const float l0 = h[2 * Ac], l1 = h[2 * Ac + 1];
        const float mx = fmaxf(l0, l1);
        const float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
        const float n0 = l0 - lse, n1 = l1 - lse;
        float u0 = t.gumbel[job][i * 2], u1 = t.gumbel[job][i * 2 + 1];
        int idx;
        if (t.hard[job]) {
          const float e = 1.1920929e-07f;
          u0 = fminf(fmaxf(u0, e), 1.f - e); u1 = fminf(fmaxf(u1, e), 1.f - e);
          const float s0 = (n0 - logf(-logf(u0))) / 0.5f, s1 = (n1 - logf(-logf(u1))) / 0.5f;
          idx = s1 > s0 ? 1 : 0;
        } else {
          idx = (n1 - logf(-logf(u1))) > (n0 - logf(-logf(u0))) ? 1 : 0;
        }
        const float mm = fmaxf(n0, n1), l2 = mm + logf(expf(n0 - mm) + expf(n1 - mm));
        lp += (idx ? n1 : n0) - l2;
        t.act_out[job][i * ld_act + Ac] = idx ? 1.f : -1.f;
        if (t.grip[job]) t.grip[job][i] = idx;
      }
      t.logp[job][i] = lp;
    }
  }
}
extern "C" int tacorl_tanh_normal_sample_batch(int njobs, const float* const* head, int ld_head, const float* const* eps,
                                               const float* const* gumbel_u, const int* hard_rsample,
                                               float* const* act_out, int ld_act, float* const* logp,
                                               int* const* grip_idx, const int* n, int M, int Ac, const float* u01,
                                               float* u_dst, int u_rows, int A, int u_discrete, tacorl_stream_t stream) {
  if (njobs < 1 || njobs > TNS_MAXJ || Ac > 32 || M <= 0) return TACORL_EINVAL;
  TnsTbl t{};
  long maxrows = 0;
  for (int j = 0; j < njobs; j++) {
    t.head[j] = head[j]; t.eps[j] = eps[j]; t.gumbel[j] = gumbel_u ? gumbel_u[j] : nullptr; t.act_out[j] = act_out[j];
    t.logp[j] = logp[j]; t.grip[j] = grip_idx ? grip_idx[j] : nullptr; t.hard[j] = hard_rsample[j]; t.n[j] = n[j];
    maxrows = (long)n[j] * M > maxrows ? (long)n[j] * M : maxrows;
  }
  t.u01 = u01; t.u_dst = u_dst; t.u_rows = u01 ? u_rows : 0; t.u_disc = u_discrete;
  const long urows8 = u01 ? ((long)u_rows * A + 255) / 256 : 0;
  long blocks = (maxrows + 7) / 8;
  if (urows8 > blocks) blocks = urows8;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(tanh_normal_sample_batch_kernel, dim3((unsigned)blocks, njobs + (u01 ? 1 : 0)), dim3(256), 0,
                     (hipStream_t)stream, t, njobs, M, Ac, ld_head, ld_act, A);
  return LAUNCH_OK();
}

// ------------------------------------------------------------------ actor losses
// logs slots (device float buffer, also the host-visible metric record)
enum {
  LG_ALPHA_LOSS = 0, LG_ALPHA, LG_ACTOR_LOSS, LG_BELL1, LG_BELL2, LG_CONS1, LG_CONS2, LG_Q1LOSS, LG_Q2LOSS,
  LG_ALPHA_P, LG_ALPHA_P_LOSS, LG_Q1_DATA, LG_Q1_RAND, LG_Q1_POL, LG_Q2_DATA, LG_Q2_RAND, LG_Q2_POL, LG_ACTION_LOSS,
  LG_COUNT
};

// alpha_loss = -mean(log_alpha * (logpi + target_entropy));  d/dlog_alpha = -mean(logpi + H)
// (cql_offline_lightning.py:447-449).  Single block.
__global__ __launch_bounds__(256) void alpha_loss_kernel(const float* __restrict__ logp, int B, const float* log_alpha,
                                                         float target_entropy, float inv_world, float* g_log_alpha,
                                                         float* logs) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += logp[i] + target_entropy;
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) {
    const float mean = s / (float)B;
    g_log_alpha[0] = -mean * inv_world;
    logs[LG_ALPHA_LOSS] = -(log_alpha[0] * mean);
  }
}
// The same loss and gradient, then Adam's step on log_alpha in the same launch (one GPU: the gradient needs no
// all-reduce between the two; with several ranks tacorl_alpha_loss + collective #1 + tacorl_adam_step stay separate).
// alpha_loss is logged with the PRE-step log_alpha as the reference does; the update is adam_kernel's arithmetic for a
// single element without clipping (bit-identical to tacorl_adam_step(n = 1, max_norm = 0)).
__global__ __launch_bounds__(256) void alpha_loss_step_kernel(const float* __restrict__ logp, int B, float* log_alpha,
                                                              float target_entropy, float* g_log_alpha, float* logs,
                                                              float* m, float* v, float lr, int* step_counter) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += logp[i] + target_entropy;
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) {
    const float mean = s / (float)B, la = log_alpha[0];
    const float gi = -mean * 1.0f;  // (grad_scale = 1 / world = 1)
    g_log_alpha[0] = gi;
    logs[LG_ALPHA_LOSS] = -(la * mean);
    const int t = step_counter[0] + 1;
    step_counter[0] = t;
    const double bc1 = 1.0 - pow(0.9, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
    const float step_size = (float)((double)lr / bc1), rsq_bc2 = (float)sqrt(bc2);
    const float mi = m[0] * 0.9f + gi * 0.1f;
    const float vi = v[0] * 0.999f + (gi * gi) * 0.001f;
    m[0] = mi; v[0] = vi;
    const float denom = sqrtf(vi) / rsq_bc2 + 1e-8f;
    log_alpha[0] = la - step_size * (mi / denom);
  }
}
extern "C" int tacorl_alpha_loss_step(const float* logp, int B, float* log_alpha, float target_entropy,
                                      float* g_log_alpha, float* logs, float* m, float* v, float lr, int* step_counter,
                                      tacorl_stream_t stream) {
  hipLaunchKernelGGL(alpha_loss_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, B, log_alpha,
                     target_entropy, g_log_alpha, logs, m, v, lr, step_counter);
  return LAUNCH_OK();
}
extern "C" int tacorl_alpha_loss(const float* logp, int B, const float* log_alpha, float target_entropy,
                                 float grad_scale, float* g_log_alpha, float* logs, tacorl_stream_t stream) {
  hipLaunchKernelGGL(alpha_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logp, B, log_alpha,
                     target_entropy, grad_scale, g_log_alpha, logs);
  return LAUNCH_OK();
}

// Q phase (cql_offline_lightning.py:463-466): actor_loss = mean(alpha*logpi - min(q1,q2)).
// Writes d(actor_loss)/dq_i (torch.min tie rule: split evenly) and the loss.  Single block.
__global__ __launch_bounds__(256) void actor_qmin_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                                         const float* __restrict__ logp, int B, const float* log_alpha,
                                                         float* __restrict__ dq1, float* __restrict__ dq2,
                                                         float grad_scale, float* logs) {
  __shared__ float sh[4];
  const float alpha = expf(log_alpha[0]);
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float a = q1[i], b = q2[i];
    s += alpha * logp[i] - fminf(a, b);
    const float g = -grad_scale / (float)B;
    dq1[i] = a < b ? g : (a == b ? 0.5f * g : 0.f);
    dq2[i] = b < a ? g : (a == b ? 0.5f * g : 0.f);
  }
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) { logs[LG_ACTOR_LOSS] = s / (float)B; logs[LG_ALPHA] = alpha; }
}
extern "C" int tacorl_actor_qmin(const float* q1, const float* q2, const float* logp, int B, const float* log_alpha,
                                 float* dq1, float* dq2, float grad_scale, float* logs, tacorl_stream_t stream) {
  hipLaunchKernelGGL(actor_qmin_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, q1, q2, logp, B, log_alpha, dq1,
                     dq2, grad_scale, logs);
  return LAUNCH_OK();
}

// Gradient of the actor loss w.r.t. the policy head (rsample path, actor.py:106-111):
//   L = (1/B) sum_b [ alpha*logpi_b  - Qmin_b            ]   (Q phase:  g_act = dL/da from the critics)
//   L = (1/B) sum_b [ alpha*logpi_b  - logp_data_b       ]   (BC phase: value = dataset action)
// d logpi/d mu_j = 2 a_j, d logpi/d logsd_j = -1 + 2 a_j eps_j sd_j (the Normal terms cancel through
// z = mu + sd*eps), d a_j/d mu_j = 1 - a_j^2, clamp masks as torch.clamp.
// bc: also accumulates mean(alpha*logpi - logp_data) into logs (single block when bc).
// One workgroup of 1024 threads, one thread per (row, action dim) element (consecutive lanes = consecutive dims: the
// head / eps / gradient loads coalesce and are independent of each other; with a thread per row and a serial loop over
// the dims this was 21 us of dependent, strided round trips on the update's chain), then one thread per row for the
// gripper logits and the per-row loss term.
__global__ __launch_bounds__(1024) void actor_head_bwd_kernel(
    const float* __restrict__ head, int ld_head, const float* __restrict__ eps, const float* __restrict__ logp,
    const float* __restrict__ g_act1, const float* __restrict__ g_act2, int ld_g, const float* __restrict__ value,
    int ld_value, const int* __restrict__ grip_idx, const float* log_alpha, float grad_scale,
    float* __restrict__ d_head, int B, int Ac, int has_grip, float* logs) {
  __shared__ float sh[16];
  const float alpha = expf(log_alpha[0]);
  const float gl = alpha * grad_scale / (float)B;  // dL/dlogpi
  const float c = -grad_scale / (float)B;
  float loss = 0.f;
  const int total = B * Ac;
#pragma unroll 2
  for (int e0 = threadIdx.x; e0 < total; e0 += 1024) {
    const int b = e0 / Ac, j = e0 - b * Ac;
    const float* h = head + (long)b * ld_head;
    float* dh = d_head + (long)b * ld_head;
    float mu, sd;
    head_stats(h, j, Ac, mu, sd);
    const float e = eps[(long)b * Ac + j];
    const float a = tanhf(mu + e * sd);
    float gm = gl * 2.f * a, gs = gl * (-1.f + 2.f * a * e * sd);
    if (g_act1) {
      float ga = g_act1[(long)b * ld_g + j] + (g_act2 ? g_act2[(long)b * ld_g + j] : 0.f);
      gm += ga * (1.f - a * a);
      gs += ga * (1.f - a * a) * e * sd;
    }
    if (value) {  // - (1/B) d logp_data: TanhNormal.log_prob(value), distributions.py:98-109
      float v = fminf(fmaxf(value[(long)b * ld_value + j], -0.999f), 0.999f);
      const float zd = 0.5f * logf(fmaxf(1.f + v, 1e-6f) / fmaxf(1.f - v, 1e-6f));
      const float d = zd - mu, var = sd * sd;
      loss -= normal_lp(zd, mu, sd) + tanh_corr(zd);
      gm += c * (d / var);
      gs += c * ((d * d) / var - 1.f);
    }
    const float mr = h[j], lr = h[Ac + j];
    dh[j] = (mr >= -9.f && mr <= 9.f) ? gm : 0.f;
    dh[Ac + j] = (lr >= -5.f && lr <= 2.f) ? gs : 0.f;
  }
  for (int b = threadIdx.x; b < B; b += 1024) {
    if (has_grip) {
      // log pi (and, in BC, log p_data) contain log_softmax(logits)[idx]: d/dlogits = onehot - softmax
      const float* h = head + (long)b * ld_head;
      float* dh = d_head + (long)b * ld_head;
      const float l0 = h[2 * Ac], l1 = h[2 * Ac + 1], mx = fmaxf(l0, l1);
      const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
      const int idx = grip_idx[b];
      float g0 = gl * ((idx == 0 ? 1.f : 0.f) - p0), g1 = gl * ((idx == 1 ? 1.f : 0.f) - p1);
      if (value) {
        const int vi = (int)(value[(long)b * ld_value + Ac] / 2.f + 0.5f);
        g0 += c * ((vi == 0 ? 1.f : 0.f) - p0); g1 += c * ((vi == 1 ? 1.f : 0.f) - p1);
        loss -= (vi ? l1 : l0) - (mx + logf(e0 + e1));
      }
      dh[2 * Ac] = g0; dh[2 * Ac + 1] = g1;
    }
    if (value) loss += alpha * logp[b];
  }
  if (value) {
    loss = wave_sum(loss);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int i = 0; i < 16; i++) t += sh[i];
      logs[LG_ACTOR_LOSS] = t / (float)B; logs[LG_ALPHA] = alpha;
    }
  }
}
extern "C" int tacorl_actor_head_bwd(const float* head, int ld_head, const float* eps, const float* logp,
                                     const float* g_act1, const float* g_act2, int ld_g, const float* value,
                                     int ld_value, const int* grip_idx, const float* log_alpha, float grad_scale,
                                     float* d_head, int B, int Ac, int has_grip, float* logs,
                                     tacorl_stream_t stream) {
  hipLaunchKernelGGL(actor_head_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, head, ld_head, eps, logp,
                     g_act1, g_act2, ld_g, value, ld_value, grip_idx, log_alpha, grad_scale, d_head, B, Ac, has_grip,
                     logs);
  return LAUNCH_OK();
}

// ================================================================= CQL + Bellman
// q_i: Q outputs for rows [data B | rand nB | cur nB | nxt nB] (sample-major k*B+b).
// One wave per sample b: the 3n logits live one per lane (3n <= 64), logsumexp / softmax by
// wave shuffles (reference cql_offline_lightning.py:284-314, 357-398).  Writes dL_i/dq for every
// row, per-wave partial sums; the last phase (block 0 of a second launch) folds the partials into
// the logged scalars and the Lagrange gradient.
struct CqlArgs {
  const float* q[2];       // main-row Q outputs
  float* dq[2];            // gradients, same layout
  const float* tq[2];      // target Q(next, a')  [B]
  const float* logp_cur;   // [n*B]
  const float* logp_nxt;   // [n*B]
  const float* next_logp;  // [B] log pi(a'|s') (used when !deterministic_backup)
  const float* reward;     // [B] float
  const float* done;       // [B] float (0/1)
  const float* log_alpha;
  const float* log_alpha_prime;  // NULL when !with_lagrange
  float* partial;          // [nblocks*4][12]
  int B, n, A;
  float discount, reward_scale, temp, cons_w, gap, grad_scale;
  int deterministic_backup;
};
enum { CP_BELL1, CP_BELL2, CP_LSE1, CP_LSE2, CP_D1, CP_D2, CP_R1, CP_R2, CP_P1, CP_P2, CP_N };

__global__ __launch_bounds__(256) void cql_loss_kernel(CqlArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int B = a.B, n = a.n, nc = 3 * n;
  const float alpha = expf(a.log_alpha[0]);
  const float alpha_p = a.log_alpha_prime ? fminf(fmaxf(expf(a.log_alpha_prime[0]), 0.f), 1000000.f) : 1.f;
  const float rand_density = (float)a.A * -0.6931471805599453f;  // log(0.5^A)
  float acc[CP_N];
#pragma unroll
  for (int i = 0; i < CP_N; i++) acc[i] = 0.f;
  for (int b = blockIdx.x * 4 + wv; b < B; b += gridDim.x * 4) {
    float qn = fminf(a.tq[0][b], a.tq[1][b]);
    if (!a.deterministic_backup) qn -= alpha * a.next_logp[b];
    const float y = a.reward_scale * a.reward[b] + (1.f - a.done[b]) * a.discount * qn;
    // which logits this lane owns: j = lane + 64 s (s < 2) -> group g = j / n, sample k = j % n
    // (3n <= 128: n = 4 in the TACORL configs, n = 32 in the CQL baseline)
    int gq[2]; bool on[2]; long row[2]; float sub[2];
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int j = lane + 64 * s, g = j / n, k = j - g * n;
      gq[s] = g; on[s] = j < nc;
      row[s] = (long)B + ((long)g * n + k) * B + b;
      sub[s] = 0.f;
      if (on[s]) sub[s] = g == 0 ? rand_density : (g == 1 ? a.logp_cur[(long)k * B + b] : a.logp_nxt[(long)k * B + b]);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const float qd = a.q[i][b];
      float qv[2], lg[2], e[2];
#pragma unroll
      for (int s = 0; s < 2; s++) {
        qv[s] = on[s] ? a.q[i][row[s]] : 0.f;
        lg[s] = on[s] ? (qv[s] - sub[s]) / a.temp : -INFINITY;
      }
      const float mx = wave_max(fmaxf(lg[0], lg[1]));
#pragma unroll
      for (int s = 0; s < 2; s++) e[s] = on[s] ? expf(lg[s] - mx) : 0.f;
      const float se = wave_sum(e[0] + e[1]);
      const float lse = mx + logf(se);
#pragma unroll
      for (int s = 0; s < 2; s++)
        if (on[s]) a.dq[i][row[s]] = a.grad_scale * alpha_p * a.cons_w * (e[s] / se) / (float)B;
      const float sr = wave_sum((on[0] && gq[0] == 0 ? qv[0] : 0.f) + (on[1] && gq[1] == 0 ? qv[1] : 0.f));
      const float sp = wave_sum((on[0] && gq[0] == 1 ? qv[0] : 0.f) + (on[1] && gq[1] == 1 ? qv[1] : 0.f));
      if (lane == 0) {
        const float d = qd - y;
        a.dq[i][b] = a.grad_scale * (2.f * d - a.cons_w * alpha_p) / (float)B;
        acc[CP_BELL1 + i] += d * d; acc[CP_LSE1 + i] += lse; acc[CP_D1 + i] += qd;
        acc[CP_R1 + i] += sr; acc[CP_P1 + i] += sp;
      }
    }
  }
  if (lane == 0) {
    float* o = a.partial + ((long)blockIdx.x * 4 + wv) * CP_N;
#pragma unroll
    for (int i = 0; i < CP_N; i++) o[i] = acc[i];
  }
}
__global__ __launch_bounds__(256) void cql_finish_kernel(CqlArgs a, int nparts, float* g_log_alpha_prime, float* logs) {
  __shared__ float sh[4];
  float tot[CP_N];
  for (int i = 0; i < CP_N; i++) {
    float s = 0.f;
    for (int j = threadIdx.x; j < nparts; j += 256) s += a.partial[(long)j * CP_N + i];
    tot[i] = block_sum_256(s, sh);
  }
  if (threadIdx.x != 0) return;
  const float B = (float)a.B, nB = (float)a.B * (float)a.n;
  const float ea = a.log_alpha_prime ? expf(a.log_alpha_prime[0]) : 1.f;
  const float alpha_p = a.log_alpha_prime ? fminf(fmaxf(ea, 0.f), 1000000.f) : 1.f;
  float cons[2], bell[2];
  for (int i = 0; i < 2; i++) {
    bell[i] = tot[CP_BELL1 + i] / B;
    cons[i] = tot[CP_LSE1 + i] / B * a.cons_w * a.temp - tot[CP_D1 + i] / B * a.cons_w;
  }
  if (a.log_alpha_prime) {
    const float raw = (cons[0] - a.gap) + (cons[1] - a.gap);
    cons[0] = alpha_p * (cons[0] - a.gap); cons[1] = alpha_p * (cons[1] - a.gap);
    logs[LG_ALPHA_P] = alpha_p;
    logs[LG_ALPHA_P_LOSS] = (-cons[0] - cons[1]) * 0.5f;
    // d/dlog_alpha' of -(alpha'(c1-gap) + alpha'(c2-gap))/2 ; clamp passes gradient inside [0,1e6]
    g_log_alpha_prime[0] = (ea >= 0.f && ea <= 1000000.f) ? -0.5f * raw * ea * a.grad_scale : 0.f;
  }
  logs[LG_BELL1] = bell[0]; logs[LG_BELL2] = bell[1];
  logs[LG_CONS1] = cons[0]; logs[LG_CONS2] = cons[1];
  logs[LG_Q1LOSS] = bell[0] + cons[0]; logs[LG_Q2LOSS] = bell[1] + cons[1];
  logs[LG_Q1_DATA] = tot[CP_D1] / B; logs[LG_Q2_DATA] = tot[CP_D2] / B;
  logs[LG_Q1_RAND] = tot[CP_R1] / nB; logs[LG_Q2_RAND] = tot[CP_R2] / nB;
  logs[LG_Q1_POL] = tot[CP_P1] / nB; logs[LG_Q2_POL] = tot[CP_P2] / nB;
}
extern "C" size_t tacorl_cql_ws_bytes(int B) {
  const int blocks = (B + 3) / 4 > 256 ? 256 : (B + 3) / 4;
  return (size_t)blocks * 4 * CP_N * sizeof(float);
}
extern "C" int tacorl_cql_loss(const float* q1, const float* q2, float* dq1, float* dq2, const float* tq1,
                               const float* tq2, const float* logp_cur, const float* logp_nxt,
                               const float* next_logp, const float* reward, const float* done,
                               const float* log_alpha, const float* log_alpha_prime, int B, int n, int A,
                               float discount, float reward_scale, float temp, float cons_w, float gap,
                               int deterministic_backup, float grad_scale, float* g_log_alpha_prime, float* logs,
                               void* ws, size_t ws_bytes, tacorl_stream_t stream) {
  if (3 * n > 128 || n < 1) return TACORL_EINVAL;
  if (ws_bytes < tacorl_cql_ws_bytes(B)) return TACORL_ENOMEM;
  CqlArgs a{};
  a.q[0] = q1; a.q[1] = q2; a.dq[0] = dq1; a.dq[1] = dq2; a.tq[0] = tq1; a.tq[1] = tq2;
  a.logp_cur = logp_cur; a.logp_nxt = logp_nxt; a.next_logp = next_logp; a.reward = reward; a.done = done;
  a.log_alpha = log_alpha; a.log_alpha_prime = log_alpha_prime; a.partial = (float*)ws;
  a.B = B; a.n = n; a.A = A; a.discount = discount; a.reward_scale = reward_scale; a.temp = temp; a.cons_w = cons_w;
  a.gap = gap; a.grad_scale = grad_scale; a.deterministic_backup = deterministic_backup;
  const int blocks = (B + 3) / 4 > 256 ? 256 : (B + 3) / 4;
  hipLaunchKernelGGL(cql_loss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(cql_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, blocks * 4, g_log_alpha_prime,
                     logs);
  return LAUNCH_OK();
}

// =========================================================== clip + Adam + Polyak
// torch.optim.Adam (betas .9/.999, eps 1e-8) + clip_grad_norm_ (L2, eps 1e-6) + soft target
// update (cql_offline_lightning.py:229-232, 519-542), over one flat parameter block.
// state = {step (as float), } lives on the device so a captured graph replays correctly.
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, long n, float* partial) {
  __shared__ float sh[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += g[i] * g[i];
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                   const float* partial, int nparts, float max_norm,
                                                   int* step_counter, float* __restrict__ target, float tau) {
  __shared__ float sh[4];
  float coef = 1.f;
  if (max_norm > 0.f) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partial[i];
    s = block_sum_256(s, sh);
    coef = fminf(max_norm / (sqrtf(s) + 1e-6f), 1.0f);
  }
  const int t = step_counter[0] + 1;
  if (gridDim.x == 1) {  // a one-block launch (log_alpha) bumps its own counter: one launch less on the dependent chain
    __syncthreads();     // every thread has read the counter
    if (threadIdx.x == 0) step_counter[0] = t;
  }
  __shared__ float bcs[2];  // the bias-correction factors once per block (two double-precision pow() per thread before round 5)
  if (threadIdx.x == 0) {
    const double bc1 = 1.0 - pow(0.9, (double)t), bc2 = 1.0 - pow(0.999, (double)t);
    bcs[0] = (float)((double)lr / bc1);
    bcs[1] = (float)sqrt(bc2);
  }
  __syncthreads();
  const float step_size = bcs[0], rsq_bc2 = bcs[1];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * coef;
    const float mi = m[i] * 0.9f + gi * 0.1f;
    const float vi = v[i] * 0.999f + (gi * gi) * 0.001f;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / rsq_bc2 + 1e-8f;
    const float pn = p[i] - step_size * (mi / denom);
    p[i] = pn;
    if (target) target[i] = target[i] * (1.0f - tau) + pn * tau;
  }
}
__global__ void bump_step_kernel(int* step_counter) { step_counter[0] += 1; }

// Several parameter blocks in two launches (norms + step counters, updates) instead of three each:
// the optimiser phase is the tail of the step's dependent chain.  Same per-block partitioning as
// tacorl_adam_step, so results are bit-identical to it.
#define ADAM_MAXB 8
struct AdamTbl {
  float* p[ADAM_MAXB];
  const float* g[ADAM_MAXB];
  float* m[ADAM_MAXB];
  float* v[ADAM_MAXB];
  float* target[ADAM_MAXB];
  int* step[ADAM_MAXB];
  float* partial[ADAM_MAXB];
  long n[ADAM_MAXB];
  float lr[ADAM_MAXB], max_norm[ADAM_MAXB], tau[ADAM_MAXB];
  int blocks[ADAM_MAXB];
  float* coef;                 // [ADAM_MAXB][2]: step size and sqrt(bias correction 2) of this step, written by the norm launch
  __bf16* mirror[ADAM_MAXB];   // optional bf16 copies of the updated parameters / of the updated Polyak target, same
  __bf16* tmirror[ADAM_MAXB];  // element offsets (the fused MLP kernels' MFMA operand: no conversion launch next step)
};
__global__ __launch_bounds__(256) void sqnorm_partial_batch_kernel(AdamTbl t) {
  __shared__ float sh[4];
  const int b = blockIdx.y;
  // the step counters advance here, one launch ahead of the update that reads them (no bump launch behind it) - and with
  // them the two bias-correction factors: two double-precision pow() per THREAD of the update launch were most of its
  // 16 us (round 5); the same double arithmetic, once per block of parameters
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int st = t.step[b][0] + 1;
    t.step[b][0] = st;
    const double bc1 = 1.0 - pow(0.9, (double)st), bc2 = 1.0 - pow(0.999, (double)st);
    t.coef[2 * b] = (float)((double)t.lr[b] / bc1);
    t.coef[2 * b + 1] = (float)sqrt(bc2);
  }
  if ((int)blockIdx.x >= t.blocks[b] || !(t.max_norm[b] > 0.f)) return;
  const float* __restrict__ g = t.g[b];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < t.n[b]; i += (long)t.blocks[b] * 256) s += g[i] * g[i];
  s = block_sum_256(s, sh);
  if (threadIdx.x == 0) t.partial[b][blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void adam_batch_kernel(AdamTbl t) {
  __shared__ float sh[4];
  const int b = blockIdx.y, nb = t.blocks[b];
  if ((int)blockIdx.x >= nb) return;
  float coef = 1.f;
  if (t.max_norm[b] > 0.f) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += t.partial[b][i];
    s = block_sum_256(s, sh);
    coef = fminf(t.max_norm[b] / (sqrtf(s) + 1e-6f), 1.0f);
  }
  const float step_size = t.coef[2 * b], rsq_bc2 = t.coef[2 * b + 1];  // (of the step counter the norm launch advanced)
  float* __restrict__ p = t.p[b];
  const float* __restrict__ g = t.g[b];
  float* __restrict__ m = t.m[b];
  float* __restrict__ v = t.v[b];
  float* __restrict__ target = t.target[b];
  const float tau = t.tau[b];
  // the same arithmetic per element as tacorl_adam_step (bit-identical); four elements per thread and access when the
  // block is 16-byte aligned - the update is the tail of the step's dependent chain, 38 MB of traffic that took 20 us as
  // seven 4-byte accesses per element
  auto upd = [&](float pi, float gr, float& mi, float& vi, float& ti, bool has_t) {
    const float gi = gr * coef;
    mi = mi * 0.9f + gi * 0.1f;
    vi = vi * 0.999f + (gi * gi) * 0.001f;
    const float denom = sqrtf(vi) / rsq_bc2 + 1e-8f;
    const float pn = pi - step_size * (mi / denom);
    if (has_t) ti = ti * (1.0f - tau) + pn * tau;
    return pn;
  };
  const long n = t.n[b];
  __bf16* __restrict__ mir = t.mirror[b];
  __bf16* __restrict__ tmir = target ? t.tmirror[b] : nullptr;
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)target) & 15) == 0) &&
                   ((((uintptr_t)mir | (uintptr_t)tmir) & 7) == 0);
  const long n4 = vec ? n / 4 : 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)nb * 256) {
    f32x4 P = reinterpret_cast<f32x4*>(p)[i], M = reinterpret_cast<f32x4*>(m)[i], V = reinterpret_cast<f32x4*>(v)[i];
    const f32x4 G = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 T = target ? reinterpret_cast<f32x4*>(target)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float mi = M[e], vi = V[e], ti = T[e];
      P[e] = upd(P[e], G[e], mi, vi, ti, target != nullptr);
      M[e] = mi; V[e] = vi; T[e] = ti;
    }
    reinterpret_cast<f32x4*>(m)[i] = M; reinterpret_cast<f32x4*>(v)[i] = V; reinterpret_cast<f32x4*>(p)[i] = P;
    if (target) reinterpret_cast<f32x4*>(target)[i] = T;
    if (mir) reinterpret_cast<bf16x4*>(mir)[i] = bf16x4{(__bf16)P[0], (__bf16)P[1], (__bf16)P[2], (__bf16)P[3]};
    if (tmir) reinterpret_cast<bf16x4*>(tmir)[i] = bf16x4{(__bf16)T[0], (__bf16)T[1], (__bf16)T[2], (__bf16)T[3]};
  }
  for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)nb * 256) {
    float mi = m[i], vi = v[i], ti = target ? target[i] : 0.f;
    const float pn = upd(p[i], g[i], mi, vi, ti, target != nullptr);
    m[i] = mi; v[i] = vi; p[i] = pn;
    if (target) target[i] = ti;
    if (mir) mir[i] = (__bf16)pn;
    if (tmir) tmir[i] = (__bf16)ti;
  }
}
extern "C" size_t tacorl_adam_batch_ws_bytes(int nb) { return ((size_t)nb * 1024 + 2 * ADAM_MAXB) * sizeof(float); }
extern "C" int tacorl_adam_step_batch(int nb, float* const* param, const float* const* grad, float* const* m,
                                      float* const* v, const long* n, const float* lr, const float* max_norm,
                                      int* const* step_counter, float* const* target, const float* tau, void* ws,
                                      size_t ws_bytes, tacorl_stream_t stream) {
  return tacorl_adam_step_batch_mirror(nb, param, grad, m, v, n, lr, max_norm, step_counter, target, tau, nullptr, nullptr, ws,
                                       ws_bytes, stream);
}
extern "C" int tacorl_adam_step_batch_mirror(int nb, float* const* param, const float* const* grad, float* const* m,
                                             float* const* v, const long* n, const float* lr, const float* max_norm,
                                             int* const* step_counter, float* const* target, const float* tau,
                                             void* const* mirror, void* const* target_mirror, void* ws, size_t ws_bytes,
                                             tacorl_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (nb < 1 || nb > ADAM_MAXB) return TACORL_EINVAL;
  if (ws_bytes < tacorl_adam_batch_ws_bytes(nb)) return TACORL_ENOMEM;
  AdamTbl t{};
  t.coef = (float*)ws + (long)nb * 1024;
  int maxb = 0;
  bool any_clip = false;
  for (int b = 0; b < nb; b++) {
    long blocks = (n[b] + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    t.p[b] = param[b]; t.g[b] = grad[b]; t.m[b] = m[b]; t.v[b] = v[b]; t.target[b] = target ? target[b] : nullptr;
    t.step[b] = step_counter[b]; t.partial[b] = (float*)ws + (long)b * 1024; t.n[b] = n[b]; t.lr[b] = lr[b];
    t.max_norm[b] = max_norm[b]; t.tau[b] = tau ? tau[b] : 0.f; t.blocks[b] = (int)blocks;
    t.mirror[b] = mirror ? (__bf16*)mirror[b] : nullptr; t.tmirror[b] = target_mirror ? (__bf16*)target_mirror[b] : nullptr;
    if (((uintptr_t)t.mirror[b] | (uintptr_t)t.tmirror[b]) & 1) return TACORL_EINVAL;
    maxb = (int)blocks > maxb ? (int)blocks : maxb;
    any_clip |= max_norm[b] > 0.f;
  }
  (void)any_clip;  // the norm launch also advances the step counters, so it always runs
  hipLaunchKernelGGL(sqnorm_partial_batch_kernel, dim3(maxb, nb), dim3(256), 0, st, t);
  hipLaunchKernelGGL(adam_batch_kernel, dim3(maxb, nb), dim3(256), 0, st, t);
  return LAUNCH_OK();
}

extern "C" size_t tacorl_adam_ws_bytes(long n) {
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  return (size_t)blocks * sizeof(float);
}
extern "C" int tacorl_adam_step(float* param, const float* grad, float* m, float* v, long n, float lr,
                                float max_norm, int* step_counter, float* target, float tau, void* ws,
                                size_t ws_bytes, tacorl_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (max_norm > 0.f) {
    if (ws_bytes < (size_t)blocks * sizeof(float)) return TACORL_ENOMEM;
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3((int)blocks), dim3(256), 0, st, grad, n, (float*)ws);
  }
  hipLaunchKernelGGL(adam_kernel, dim3((int)blocks), dim3(256), 0, st, param, grad, m, v, n, lr, (const float*)ws,
                     (int)blocks, max_norm, step_counter, target, tau);
  if (blocks > 1) hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, st, step_counter);
  return LAUNCH_OK();
}
