// dd_input.hip -- the input side of a training step on the device:
//   dd_prepare_frames   uint8 HWC frames of a batch of triplets -> ('color',f,0) and ('color_aug',f,0): ToTensor, horizontal
//                       flip, torchvision-ColorJitter with per-frame parameters (reference datasets/base_dataset.py:83-95,118-131,
//                       159-164; torchvision.transforms.functional_tensor for the arithmetic)
//   dd_pyramid_down2    one level of the target pyramid: clamp(bicubic-antialias resize by 1/2, 0, 1)  (Trainer.py:722-734, :80)
// Byte-streaming kernels: one thread per pixel, coalesced along W, 3 B read + 24 B written per pixel and frame; the only
// reduction is the grey-level mean that `adjust_contrast` blends with (per frame, fixed-order two-level sum).
// The reference does all of this on the host, per sample, in the DataLoader workers (two of them, options.py:38-41).
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"

namespace dd {

constexpr int IN_NT = 256;
constexpr int IN_BPI = 32;       // workgroups per frame in the grey-mean pass

struct Rgb {
  float r, g, b;
};

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float grey(const Rgb& c) { return 0.2989f * c.r + 0.587f * c.g + 0.114f * c.b; }      // rgb_to_grayscale

// _blend(img1, img2, ratio) = clamp(ratio * img1 + (1 - ratio) * img2, 0, 1)
__device__ __forceinline__ Rgb blend(const Rgb& a, float other, float ratio) {
  const float q = 1.f - ratio;
  return {clamp01(ratio * a.r + q * other), clamp01(ratio * a.g + q * other), clamp01(ratio * a.b + q * other)};
}

// adjust_hue on a float image: _rgb2hsv, h = (h + hue) % 1, _hsv2rgb -- same formulas, same special cases
__device__ __forceinline__ Rgb shift_hue(const Rgb& c, float hue) {
  const float maxc = fmaxf(c.r, fmaxf(c.g, c.b)), minc = fminf(c.r, fminf(c.g, c.b));
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr / (eqc ? 1.f : maxc);
  const float div = eqc ? 1.f : cr;
  const float rc = (maxc - c.r) / div, gc = (maxc - c.g) / div, bc = (maxc - c.b) / div;
  const bool is_r = maxc == c.r, is_g = maxc == c.g;
  const float hr = is_r ? (bc - gc) : 0.f;
  const float hg = (is_g && !is_r) ? (2.f + rc - bc) : 0.f;
  const float hb = (!is_g && !is_r) ? (4.f + gc - rc) : 0.f;
  float h = fmodf((hr + hg + hb) / 6.f + 1.f, 1.f);
  h = h + hue;
  h = h - floorf(h);                                   // python-style % 1.0 (result in [0,1))
  const float v = maxc;
  const float i6 = floorf(h * 6.f);
  const float f = h * 6.f - i6;
  int i = static_cast<int>(i6) % 6;
  if (i < 0) i += 6;
  const float p = clamp01(v * (1.f - s)), q = clamp01(v * (1.f - s * f)), t = clamp01(v * (1.f - s * (1.f - f)));
  switch (i) {
    case 0: return {v, t, p};
    case 1: return {q, v, p};
    case 2: return {p, v, t};
    case 3: return {p, q, v};
    case 4: return {t, p, v};
    default: return {v, p, q};
  }
}

struct Jitter {          // params row: apply, order[4] (0 brightness, 1 contrast, 2 saturation, 3 hue), b, c, s, h
  int apply, order[4];
  float bri, con, sat, hue;
};

__device__ __forceinline__ Jitter load_jitter(const float* __restrict__ p) {
  Jitter j;
  j.apply = p[0] > 0.5f;
#pragma unroll
  for (int k = 0; k < 4; ++k) j.order[k] = static_cast<int>(p[1 + k] + 0.5f);
  j.bri = p[5]; j.con = p[6]; j.sat = p[7]; j.hue = p[8];
  return j;
}

// the ops of ColorJitter.forward from position `from` up to (not including) `to`; `mean` feeds adjust_contrast
__device__ __forceinline__ Rgb jitter_ops(Rgb c, const Jitter& j, int from, int to, float mean) {
  for (int k = from; k < to; ++k) {
    const int op = j.order[k];
    if (op == 0) c = blend(c, 0.f, j.bri);
    else if (op == 1) c = blend(c, mean, j.con);
    else if (op == 2) c = blend(c, grey(c), j.sat);
    else c = shift_hue(c, j.hue);
  }
  return c;
}

__device__ __forceinline__ int contrast_position(const Jitter& j) {
  int pos = 4;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (j.order[k] == 1) pos = k;
  return pos;
}

__device__ __forceinline__ Rgb load_u8(const uint8_t* __restrict__ frame, int p) {
  const uint8_t* q = frame + (size_t)p * 3;
  return {static_cast<float>(q[0]) / 255.f, static_cast<float>(q[1]) / 255.f, static_cast<float>(q[2]) / 255.f};
}

// grey-level sum of the image as adjust_contrast sees it (after the ops drawn in front of it); IN_BPI partials per frame
__global__ __launch_bounds__(IN_NT) void jitter_mean_kernel(const uint8_t* __restrict__ frames, const float* __restrict__ params, int n,
                                                             float* __restrict__ partial) {
  __shared__ float red[IN_NT / 64];
  const int bf = blockIdx.y;
  const Jitter j = load_jitter(params + bf * 9);
  const int cpos = contrast_position(j);
  float acc = 0.f;
  if (j.apply && cpos < 4) {
    const uint8_t* frame = frames + (size_t)bf * n * 3;
    for (int p = blockIdx.x * IN_NT + threadIdx.x; p < n; p += IN_BPI * IN_NT) acc += grey(jitter_ops(load_u8(frame, p), j, 0, cpos, 0.f));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[bf * IN_BPI + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ToTensor + flip -> color; + the jitter -> color_aug.  Outputs are frame-major (F,B,3,H,W): one contiguous tensor per frame.
__global__ __launch_bounds__(IN_NT) void jitter_apply_kernel(const uint8_t* __restrict__ frames, const float* __restrict__ params,
                                                              const int32_t* __restrict__ flip, int B, int F, int H, int W,
                                                              const float* __restrict__ partial, float* __restrict__ color,
                                                              float* __restrict__ color_aug) {
  __shared__ float s_mean;
  const int bf = blockIdx.y, b = bf / F, f = bf % F;
  const int n = H * W;
  const Jitter j = load_jitter(params + bf * 9);
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < IN_BPI; ++i) s += partial[bf * IN_BPI + i];
    s_mean = s / static_cast<float>(n);
  }
  __syncthreads();
  const int p = blockIdx.x * IN_NT + threadIdx.x;
  if (p >= n) return;
  const int y = p / W, x = p - y * W;
  const int xs = flip[b] ? W - 1 - x : x;
  const Rgb c = load_u8(frames + (size_t)bf * n * 3, y * W + xs);
  const size_t o = ((size_t)f * B + b) * 3 * n + p;
  color[o] = c.r; color[o + n] = c.g; color[o + 2 * (size_t)n] = c.b;
  const Rgb a = j.apply ? jitter_ops(c, j, 0, 4, s_mean) : c;
  color_aug[o] = a.r; color_aug[o + n] = a.g; color_aug[o + 2 * (size_t)n] = a.b;
}

// Keys cubic (a = -0.5), the filter of upsample_bicubic2d_aa
__device__ __forceinline__ float cubic_aa(float x) {
  x = fabsf(x);
  if (x < 1.f) return (1.5f * x - 2.5f) * x * x + 1.f;
  if (x < 2.f) return ((-0.5f * x + 2.5f) * x - 4.f) * x + 2.f;
  return 0.f;
}

// the (at most) eight taps of output index i of a 2x antialiased down-scale: input indices 2i-3 .. 2i+4 clipped to the image,
// weights cubic((j + 0.5 - centre) / 2) normalised over the taps that remain (ATen _compute_indices_weights_aa)
__device__ __forceinline__ void aa_taps(int i, int in_size, float w[8]) {
  float total = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int jx = 2 * i - 3 + t;
    const float v = (jx >= 0 && jx < in_size) ? cubic_aa((static_cast<float>(t) - 3.5f) * 0.5f) : 0.f;
    w[t] = v;
    total += v;
  }
  const float inv = 1.f / total;
#pragma unroll
  for (int t = 0; t < 8; ++t) w[t] *= inv;
}

__global__ __launch_bounds__(IN_NT) void pyramid_down2_kernel(const float* __restrict__ src, int H, int W, float* __restrict__ dst) {
  const int h = H / 2, w = W / 2;
  const int p = blockIdx.x * IN_NT + threadIdx.x;
  if (p >= h * w) return;
  const int oy = p / w, ox = p - oy * w;
  float wx[8], wy[8];
  aa_taps(ox, W, wx);
  aa_taps(oy, H, wy);
  const float* plane = src + (size_t)blockIdx.y * H * W;
  float acc = 0.f;
#pragma unroll
  for (int ty = 0; ty < 8; ++ty) {
    const int jy = min(max(2 * oy - 3 + ty, 0), H - 1);        // clipped taps carry weight 0
    const float* row = plane + (size_t)jy * W;
    float r = 0.f;
#pragma unroll
    for (int tx = 0; tx < 8; ++tx) r += wx[tx] * row[min(max(2 * ox - 3 + tx, 0), W - 1)];      // horizontal pass first, like ATen
    acc += wy[ty] * r;
  }
  dst[(size_t)blockIdx.y * h * w + p] = clamp01(acc);
}

}  // namespace dd

using namespace dd;

extern "C" size_t dd_prepare_frames_workspace_bytes(int B, int F) { return (size_t)B * F * IN_BPI * sizeof(float); }

extern "C" int dd_prepare_frames(const uint8_t* frames_u8, const float* params, const int32_t* flip, int B, int F, int H, int W,
                                 float* color, float* color_aug, float* workspace, void* stream_) {
  if (!frames_u8 || !params || !flip || !color || !color_aug || !workspace || B < 1 || F < 1 || H < 1 || W < 1)
    return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int n = H * W;
  hipLaunchKernelGGL(jitter_mean_kernel, dim3(IN_BPI, B * F), dim3(IN_NT), 0, stream, frames_u8, params, n, workspace);
  hipLaunchKernelGGL(jitter_apply_kernel, dim3((n + IN_NT - 1) / IN_NT, B * F), dim3(IN_NT), 0, stream, frames_u8, params, flip, B, F, H, W,
                     workspace, color, color_aug);
  return (int)hipGetLastError();
}

// (B,3,H,W) -> (B,H,W,3): a thread takes four pixels -- one float4 from each plane, three float4 of interleaved pixels out (both sides
// coalesced).  HBM-bound: 24 bytes per pixel (two source frames of the KITTI batch: 70.8 MB).
__global__ __launch_bounds__(IN_NT) void pack_rgb_kernel(const float* __restrict__ planar, int n4, float* __restrict__ packed) {
  const int q = blockIdx.x * IN_NT + threadIdx.x, b = blockIdx.y;
  if (q >= n4) return;
  const float4* src = reinterpret_cast<const float4*>(planar) + (size_t)b * 3 * n4;
  const float4 r = src[q], g = src[n4 + q], bl = src[2 * n4 + q];
  float4* dst = reinterpret_cast<float4*>(packed) + ((size_t)b * n4 + q) * 3;
  dst[0] = make_float4(r.x, g.x, bl.x, r.y);
  dst[1] = make_float4(g.y, bl.y, r.z, g.z);
  dst[2] = make_float4(bl.z, r.w, g.w, bl.w);
}

extern "C" int dd_pack_rgb(const float* planar, int B, int H, int W, float* packed, void* stream_) {
  if (!planar || !packed || B < 1 || H < 1 || W < 1 || ((H * W) & 3) || (reinterpret_cast<unsigned long long>(planar) & 15ull) ||
      (reinterpret_cast<unsigned long long>(packed) & 15ull))
    return (int)hipErrorInvalidValue;
  const int n4 = H * W / 4;
  hipLaunchKernelGGL(pack_rgb_kernel, dim3((n4 + IN_NT - 1) / IN_NT, B), dim3(IN_NT), 0, static_cast<hipStream_t>(stream_), planar, n4, packed);
  return (int)hipGetLastError();
}

extern "C" int dd_pyramid_down2(const float* src, int planes, int H, int W, float* dst, void* stream_) {
  if (!src || !dst || planes < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return (int)hipErrorInvalidValue;
  const int n = (H / 2) * (W / 2);
  hipLaunchKernelGGL(pyramid_down2_kernel, dim3((n + IN_NT - 1) / IN_NT, planes), dim3(IN_NT), 0, static_cast<hipStream_t>(stream_), src, H, W, dst);
  return (int)hipGetLastError();
}
