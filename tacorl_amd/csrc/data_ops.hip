// Replay data path on the GPU: the dataset's uint8 HWC frames stay resident in HBM
// (a CALVIN-sized play dataset is ~50 GB of 84x84x3 frames: it fits in one MI355X's 288 GB), a step's windows are
// gathered by frame index, and the reference's train-time image pipeline runs on the way into the encoder's NHWC
// image buffers: RandomShiftsAug -> x/255 -> ColorJitter(brightness, contrast, hue) -> Normalize(0.5, 0.5)
// (config/datamodule/transform_manager/transforms/rl_train.yaml; utils/transforms.py:265-330).  All random draws are
// explicit inputs (per-image tables), so the kernels are deterministic functions that goldens can pin.
// Pure byte / elementwise work: HBM-bound, coalesced 16-byte accesses, no MFMA.
#include "../../include/tacorl_hip.h"
#include "common.h"

#define LAUNCH_OK() (hipGetLastError() == hipSuccess ? TACORL_OK : TACORL_ELAUNCH)

namespace {

// dst[i] = frames[index[i]] : n frames of frame_bytes (multiple of 16), one 16-byte chunk per thread iteration
__global__ void gather_frames_u8_kernel(const unsigned char* __restrict__ base, long frame_bytes, const long* __restrict__ index,
                                        unsigned char* __restrict__ dst, long n, int chunks) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const long total = n * chunks;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const long i = q / chunks;
    const int c = (int)(q - i * chunks);
    *reinterpret_cast<u32x4*>(dst + i * frame_bytes + (long)c * 16) =
        *reinterpret_cast<const u32x4*>(base + index[i] * frame_bytes + (long)c * 16);
  }
}

// torchvision.transforms.functional (_functional_tensor.py) colour operations on one RGB pixel in [0, 1]
__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
__device__ __forceinline__ float gray_of(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }
__device__ __forceinline__ void adjust_hue(float& r, float& g, float& b, float hf) {
  // _rgb2hsv
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr / (eqc ? 1.f : maxc);
  const float div = eqc ? 1.f : cr;
  const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
  const float hr = (maxc == r) ? (bc - gc) : 0.f;
  const float hg = ((maxc == g) && (maxc != r)) ? (2.f + rc - bc) : 0.f;
  const float hb = ((maxc != g) && (maxc != r)) ? (4.f + gc - rc) : 0.f;
  float h = fmodf((hr + hg + hb) / 6.f + 1.f, 1.f);
  // shift: (h + hue_factor) % 1.0 with python's non-negative remainder
  h = h + hf;
  h = h - floorf(h);
  // _hsv2rgb
  const float v = maxc;
  const float h6 = h * 6.f, fi = floorf(h6), f = h6 - fi;
  const int i = ((int)fi) % 6;
  const float p = clamp01(v * (1.f - s)), q = clamp01(v * (1.f - f * s)), t = clamp01(v * (1.f - (1.f - f) * s));
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

struct AugJob {
  const unsigned char* src;  // uint8 HWC frames
  long pitch;                // bytes between frames
  void* dst;                 // NHWC normalised frames
  const int* shift;          // [n][2] (sx, sy) in [0, 2*pad], or NULL: no shift
  const float* jitter;       // [n][8] {brightness, contrast, hue, order0..3, apply}, or NULL: no colour jitter
  const long* idx;           // optional frame ids: image i is frame idx[i * istride] of the dataset at src
  int istride;
  int n;
};
#define AUG_MAXJ 8
struct AugTbl { AugJob j[AUG_MAXJ]; };

// One workgroup per image.  RandomShiftsAug with integer shifts = a clamped (replicate-padded) translation; the
// colour operations run in the drawn order; adjust_contrast needs the image's mean grey level AFTER the operations
// that precede it, hence the first pass (a block reduction) when contrast is drawn.
// torchvision.transforms.Resize on a float tensor = torch.nn.functional.interpolate(mode="bilinear", align_corners=False),
// no antialias (the reference's torchvision generation; rl_train.yaml:3-4,16-17: 200x200 -> 128x128 static, -> 84x84
// gripper, applied to the 0..255 float frame BEFORE RandomShiftsAug): ATen's upsample_bilinear2d,
//   src = max(scale * (dst + 0.5) - 0.5, 0), scale = in / out (fp32); i0 = floor(src), i1 = i0 + (i0 < in - 1);
//   l1 = src - i0, l0 = 1 - l1;  out = h0 * (w0 * p00 + w1 * p01) + h1 * (w0 * p10 + w1 * p11)
__device__ __forceinline__ void bilinear_axis(int dst, float scale, int in, int& i0, int& i1, float& l0, float& l1) {
  const float src = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

template <typename OutT>
__global__ __launch_bounds__(256) void pack_u8_aug_kernel(AugTbl t, int H, int W, int pad, int Hs, int Ws) {
  const AugJob jb = t.j[blockIdx.y];
  const int img = blockIdx.x;
  if (img >= jb.n) return;
  __shared__ float red[4];
  const unsigned char* __restrict__ src = jb.src + (jb.idx ? jb.idx[(long)img * jb.istride] : (long)img) * jb.pitch;
  OutT* __restrict__ dst = reinterpret_cast<OutT*>(jb.dst) + (long)img * H * W * 3;
  const int sx = jb.shift ? jb.shift[2 * img] - pad : 0, sy = jb.shift ? jb.shift[2 * img + 1] - pad : 0;
  float bf = 1.f, cf = 1.f, hf = 0.f;
  unsigned ord = 0u;  // the drawn order, 2 bits per position: op k = (ord >> 2k) & 3
  int cpos = 4;       // position of the contrast operation
  bool jit = false;
  if (jb.jitter) {
    const float* p = jb.jitter + 8L * img;
    bf = p[0]; cf = p[1]; hf = p[2];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned op = (unsigned)(int)p[3 + k] & 3u;
      ord |= op << (2 * k);
      if (op == 1u) cpos = k;
    }
    jit = p[7] != 0.f;
  }
  const int npx = H * W;
  const bool resize = Hs != H || Ws != W;
  const float sch = (float)Hs / (float)H, scw = (float)Ws / (float)W;
  auto load = [&](int px, float& r, float& g, float& b) {
    const int y = px / W, x = px - y * W;
    const int ys = min(max(y + sy, 0), H - 1), xs = min(max(x + sx, 0), W - 1);  // RandomShiftsAug on the (resized) frame
    if (!resize) {
      const unsigned char* s = src + ((long)ys * W + xs) * 3;
      r = (float)s[0] / 255.0f; g = (float)s[1] / 255.0f; b = (float)s[2] / 255.0f;  // ScaleImageTensor / ToTensor
      return;
    }
    int y0, y1, x0, x1;
    float h0, h1, w0, w1;
    bilinear_axis(ys, sch, Hs, y0, y1, h0, h1);
    bilinear_axis(xs, scw, Ws, x0, x1, w0, w1);
    const unsigned char *p00 = src + ((long)y0 * Ws + x0) * 3, *p01 = src + ((long)y0 * Ws + x1) * 3,
                        *p10 = src + ((long)y1 * Ws + x0) * 3, *p11 = src + ((long)y1 * Ws + x1) * 3;
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; c++)
      v[c] = h0 * (w0 * (float)p00[c] + w1 * (float)p01[c]) + h1 * (w0 * (float)p10[c] + w1 * (float)p11[c]);
    r = v[0] / 255.0f; g = v[1] / 255.0f; b = v[2] / 255.0f;
  };
  float mean = 0.f;
  if (jit) {
    float part = 0.f;
    for (int px = threadIdx.x; px < npx; px += 256) {
      float r, g, b;
      load(px, r, g, b);
      for (int k = 0; k < cpos; k++) {  // the operations drawn before the contrast adjustment
        const unsigned op = (ord >> (2 * k)) & 3u;
        if (op == 0u) { r = clamp01(bf * r); g = clamp01(bf * g); b = clamp01(bf * b); }
        else if (op == 3u) adjust_hue(r, g, b, hf);
      }
      part += gray_of(r, g, b);
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    mean = (((red[0] + red[1]) + red[2]) + red[3]) / (float)npx;
  }
  for (int px = threadIdx.x; px < npx; px += 256) {
    float r, g, b;
    load(px, r, g, b);
    if (jit) {
      for (int k = 0; k < 4; k++) {
        const unsigned op = (ord >> (2 * k)) & 3u;
        if (op == 0u) { r = clamp01(bf * r); g = clamp01(bf * g); b = clamp01(bf * b); }
        else if (op == 1u) { r = clamp01(cf * r + (1.f - cf) * mean); g = clamp01(cf * g + (1.f - cf) * mean); b = clamp01(cf * b + (1.f - cf) * mean); }
        else if (op == 3u) adjust_hue(r, g, b, hf);
      }
    }
    r = (r - 0.5f) / 0.5f; g = (g - 0.5f) / 0.5f; b = (b - 0.5f) / 0.5f;  // Normalize(0.5, 0.5)
    dst[3L * px] = (OutT)r; dst[3L * px + 1] = (OutT)g; dst[3L * px + 2] = (OutT)b;
  }
}

}  // namespace

extern "C" int tacorl_gather_frames_u8(const void* frames, long frame_bytes, const long* index, void* dst, long n,
                                       tacorl_stream_t stream) {
  if (!frames || !index || !dst || frame_bytes <= 0 || frame_bytes % 16 || ((uintptr_t)frames & 15) || ((uintptr_t)dst & 15))
    return TACORL_EINVAL;
  if (n <= 0) return TACORL_OK;
  const int chunks = (int)(frame_bytes / 16);
  const long blocks = (n * chunks + 255) / 256;
  hipLaunchKernelGGL(gather_frames_u8_kernel, dim3((int)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)frames, frame_bytes, index, (unsigned char*)dst, n, chunks);
  return LAUNCH_OK();
}

extern "C" int tacorl_pack_images_u8_aug_batch(int njobs, const void* const* src, const long* img_pitch_bytes,
                                               void* const* dst, const int* const* shift, const float* const* jitter,
                                               const int* n_img, int dst_dtype, int H, int W, int pad,
                                               tacorl_stream_t stream) {
  return tacorl_pack_images_u8_aug_gather_batch(njobs, src, img_pitch_bytes, nullptr, nullptr, dst, shift, jitter, n_img,
                                                dst_dtype, H, W, pad, stream);
}
extern "C" int tacorl_pack_images_u8_aug_gather_batch(int njobs, const void* const* src, const long* img_pitch_bytes,
                                                      const long* const* index, const int* index_stride,
                                                      void* const* dst, const int* const* shift, const float* const* jitter,
                                                      const int* n_img, int dst_dtype, int H, int W, int pad,
                                                      tacorl_stream_t stream) {
  return tacorl_pack_images_u8_resize_aug_gather_batch(njobs, src, img_pitch_bytes, index, index_stride, dst, shift, jitter,
                                                       n_img, dst_dtype, H, W, H, W, pad, stream);
}
extern "C" int tacorl_pack_images_u8_resize_aug_gather_batch(int njobs, const void* const* src, const long* img_pitch_bytes,
                                                             const long* const* index, const int* index_stride,
                                                             void* const* dst, const int* const* shift,
                                                             const float* const* jitter, const int* n_img, int dst_dtype,
                                                             int src_H, int src_W, int H, int W, int pad,
                                                             tacorl_stream_t stream) {
  const int Hs = src_H, Ws = src_W;
  if (njobs < 1 || njobs > AUG_MAXJ || H < 1 || W < 1 || Hs < 1 || Ws < 1 || pad < 0) return TACORL_EINVAL;
  AugTbl t{};
  int m = 0, mx = 0;
  for (int j = 0; j < njobs; j++) {
    if (n_img[j] <= 0) continue;
    if (!src[j] || !dst[j]) return TACORL_EINVAL;
    const long* ix = index ? index[j] : nullptr;
    const int is = (ix && index_stride) ? index_stride[j] : 1;
    if (ix && is < 1) return TACORL_EINVAL;
    t.j[m] = AugJob{(const unsigned char*)src[j], img_pitch_bytes[j], dst[j], shift ? shift[j] : nullptr,
                    jitter ? jitter[j] : nullptr, ix, is, n_img[j]};
    mx = n_img[j] > mx ? n_img[j] : mx;
    m++;
  }
  if (m == 0) return TACORL_OK;
  if (dst_dtype == TACORL_BF16)
    hipLaunchKernelGGL(pack_u8_aug_kernel<__bf16>, dim3(mx, m), dim3(256), 0, (hipStream_t)stream, t, H, W, pad, Hs, Ws);
  else
    hipLaunchKernelGGL(pack_u8_aug_kernel<float>, dim3(mx, m), dim3(256), 0, (hipStream_t)stream, t, H, W, pad, Hs, Ws);
  return LAUNCH_OK();
}
