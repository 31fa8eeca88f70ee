// dd_decoder.hip -- the glue between two 3x3 convolutions of the disparity decoders as ONE pass over the tensor.
//
// Reference (networks/depth_decoder.py:40-53 Monodepth2, :98-113 Lite-Mono; networks/layers.py:84-121 ConvBlock / Conv3x3 / upsample):
//     x = ELU(conv_a(...))                       ConvBlock
//     x = upsample(x, scale_factor=2)            nearest (Monodepth2) or bilinear, align_corners=False (Lite-Mono)
//     x = cat((x, skip), 1)                      encoder feature of the finer level, when there is one
//     x = ReflectionPad2d(1)(x)                  first thing the next ConvBlock / Conv3x3 does
// -- four element-wise kernels that each read and write a full-resolution tensor (ATen's channels-last bilinear kernel alone
// runs at 0.3 TB/s).  Here the padded, concatenated tensor is written once from the low-resolution pre-activation and the skip
// feature; the backward gathers (no atomics): every low-resolution element collects the <= 4x4 full-resolution positions its
// value was spread to, every one of them the <= 2x2 padded positions that mirror onto it.
//   MODE 0: nearest x2   MODE 1: bilinear x2 (PyTorch's area_pixel source index, clamped at 0)   MODE 2: same size (ELU + pad only)
// Channels-last tensors, channel counts multiples of 4, fp32 / fp16 / bf16 storage (dd_half.h), fp32 arithmetic.
#include <hip/hip_runtime.h>

#include "dd_half.h"
#include "dd_math.h"

namespace dd {

constexpr int DEC_NT = 256;

__device__ __forceinline__ float elu1(float v) { return v <= 0.f ? expf(v) - 1.f : v; }          // alpha = 1 (nn.ELU default)
__device__ __forceinline__ float elu1_grad(float v) { return v <= 0.f ? expf(v) : 1.f; }
template <bool ELU>
__device__ __forceinline__ float4 act4(float4 v) {
  if (ELU) return make_float4(elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w));
  return v;
}

// PyTorch's bilinear source position for output index o at scale 1/2 (aten/native/UpSample.h area_pixel_compute_source_index,
// align_corners=False, cubic=False): i0 = floor(src), i1 = i0 + (i0 < n-1), weight of i1 = src - i0
__device__ __forceinline__ void bilinear_src(int o, int n, int& i0, int& i1, float& l1) {
  const float src = fmaxf((o + 0.5f) * 0.5f - 0.5f, 0.f);
  i0 = (int)src;
  i1 = i0 + (i0 < n - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

template <typename T, int MODE, bool ELU>
__global__ __launch_bounds__(DEC_NT) void up_cat_pad_kernel(const T* __restrict__ x, const T* __restrict__ skip, int h, int w, int C1q, int C2q,
                                                            T* __restrict__ out) {
  const int H = MODE == 2 ? h : 2 * h, W = MODE == 2 ? w : 2 * w, Wp = W + 2, Cq = C1q + C2q;
  const int yo = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * DEC_NT + threadIdx.x;            // position inside the padded row, in groups of four channels
  if (j >= Wp * Cq) return;
  const int xo = j / Cq, cq = j - xo * Cq;
  const int Y = dd_reflect(yo - 1, H), X = dd_reflect(xo - 1, W);
  float4 v;
  if (cq >= C1q) {
    v = IO<T>::load4(skip, (((long long)b * H + Y) * W + X) * C2q + (cq - C1q));
  } else if (MODE == 2) {
    v = act4<ELU>(IO<T>::load4(x, (((long long)b * h + Y) * w + X) * C1q + cq));
  } else if (MODE == 0) {
    v = act4<ELU>(IO<T>::load4(x, (((long long)b * h + (Y >> 1)) * w + (X >> 1)) * C1q + cq));
  } else {
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_src(Y, h, y0, y1, ly);
    bilinear_src(X, w, x0, x1, lx);
    const long long r0 = ((long long)b * h + y0) * w, r1 = ((long long)b * h + y1) * w;
    const float4 a = act4<ELU>(IO<T>::load4(x, (r0 + x0) * C1q + cq)), bb = act4<ELU>(IO<T>::load4(x, (r0 + x1) * C1q + cq));
    const float4 c = act4<ELU>(IO<T>::load4(x, (r1 + x0) * C1q + cq)), d = act4<ELU>(IO<T>::load4(x, (r1 + x1) * C1q + cq));
    const float hy = 1.f - ly, hx = 1.f - lx;
    v.x = hy * (hx * a.x + lx * bb.x) + ly * (hx * c.x + lx * d.x);
    v.y = hy * (hx * a.y + lx * bb.y) + ly * (hx * c.y + lx * d.y);
    v.z = hy * (hx * a.z + lx * bb.z) + ly * (hx * c.z + lx * d.z);
    v.w = hy * (hx * a.w + lx * bb.w) + ly * (hx * c.w + lx * d.w);
  }
  IO<T>::store4(out, ((long long)b * (H + 2) + yo) * Wp * Cq + j, v);
}

// d loss / d cat(...)[b, Y, X, 4 channels]: the padded positions that mirror onto (Y, X)
template <typename T>
__device__ __forceinline__ float4 pad_fold(const T* __restrict__ g, int b, int Y, int X, int H, int W, int Cq, int cq) {
  const int Hp = H + 2, Wp = W + 2;
  const int ys[2] = {Y + 1, Y == 1 ? 0 : (Y == H - 2 ? Hp - 1 : -1)};
  const int xs[2] = {X + 1, X == 1 ? 0 : (X == W - 2 ? Wp - 1 : -1)};
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int d = 0; d < 2; ++d)
      if (ys[a] >= 0 && xs[d] >= 0) {
        const float4 t = IO<T>::load4(g, (((long long)b * Hp + ys[a]) * Wp + xs[d]) * Cq + cq);
        acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
      }
  return acc;
}

// full-resolution indices that read low-resolution index i, and with which total weight (the adjoint of bilinear_src / nearest)
template <int MODE>
__device__ __forceinline__ int up_taps(int i, int n, int (&o)[4], float (&wt)[4]) {
  if (MODE == 0) {
    o[0] = 2 * i; o[1] = 2 * i + 1; wt[0] = wt[1] = 1.f;
    return 2;
  }
  int cnt = 0;
  if (i >= 1) { o[cnt] = 2 * i - 1; wt[cnt++] = 0.25f; }                     // odd row of i-1: its second tap
  o[cnt] = 2 * i; wt[cnt++] = i == 0 ? 1.f : 0.75f;                            // source clamped at 0: row 0 reads x[0] alone
  o[cnt] = 2 * i + 1; wt[cnt++] = i == n - 1 ? 1.f : 0.75f;                    // last row: both taps are x[n-1]
  if (i + 1 <= n - 1) { o[cnt] = 2 * i + 2; wt[cnt++] = 0.25f; }
  return cnt;
}

template <typename T, int MODE, bool ELU>
__global__ __launch_bounds__(DEC_NT) void up_cat_pad_bwd_x_kernel(const T* __restrict__ g, const T* __restrict__ x, int h, int w, int C1q, int C2q,
                                                                  T* __restrict__ gx) {
  const int H = MODE == 2 ? h : 2 * h, W = MODE == 2 ? w : 2 * w, Cq = C1q + C2q;
  const int yi = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * DEC_NT + threadIdx.x;
  if (j >= w * C1q) return;
  const int xi = j / C1q, cq = j - xi * C1q;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (MODE == 2) {
    acc = pad_fold<T>(g, b, yi, xi, H, W, Cq, cq);
  } else {
    int oy[4], ox[4];
    float wy[4], wx[4];
    const int ny = up_taps<MODE>(yi, h, oy, wy), nx = up_taps<MODE>(xi, w, ox, wx);
    for (int a = 0; a < ny; ++a)
      for (int d = 0; d < nx; ++d) {
        const float4 t = pad_fold<T>(g, b, oy[a], ox[d], H, W, Cq, cq);
        const float wgt = wy[a] * wx[d];
        acc.x += wgt * t.x; acc.y += wgt * t.y; acc.z += wgt * t.z; acc.w += wgt * t.w;
      }
  }
  const long long at = ((long long)b * h + yi) * w * C1q + j;
  if (ELU) {
    const float4 v = IO<T>::load4(x, at);
    acc.x *= elu1_grad(v.x); acc.y *= elu1_grad(v.y); acc.z *= elu1_grad(v.z); acc.w *= elu1_grad(v.w);
  }
  IO<T>::store4(gx, at, acc);
}

template <typename T>
__global__ __launch_bounds__(DEC_NT) void up_cat_pad_bwd_skip_kernel(const T* __restrict__ g, int H, int W, int C1q, int C2q, T* __restrict__ gs) {
  const int Y = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * DEC_NT + threadIdx.x;
  if (j >= W * C2q) return;
  const int X = j / C2q, cq = j - X * C2q;
  IO<T>::store4(gs, ((long long)b * H + Y) * W * C2q + j, pad_fold<T>(g, b, Y, X, H, W, C1q + C2q, C1q + cq));
}

template <typename T>
static void up_cat_pad_launch(const void* x_, const void* skip_, int B, int h, int w, int C1, int C2, int mode, int elu, void* out_, hipStream_t s) {
  const T* x = static_cast<const T*>(x_);
  const T* skip = static_cast<const T*>(skip_);
  T* out = static_cast<T*>(out_);
  const int H = mode == 2 ? h : 2 * h, W = mode == 2 ? w : 2 * w, C1q = C1 / 4, C2q = C2 / 4;
  const dim3 grid(((W + 2) * (C1q + C2q) + DEC_NT - 1) / DEC_NT, H + 2, B), block(DEC_NT);
#define DD_UCP(M, E) hipLaunchKernelGGL((up_cat_pad_kernel<T, M, E>), grid, block, 0, s, x, skip, h, w, C1q, C2q, out)
  if (mode == 0) { if (elu) DD_UCP(0, true); else DD_UCP(0, false); }
  else if (mode == 1) { if (elu) DD_UCP(1, true); else DD_UCP(1, false); }
  else { if (elu) DD_UCP(2, true); else DD_UCP(2, false); }
#undef DD_UCP
}

template <typename T>
static void up_cat_pad_bwd_launch(const void* g_, const void* x_, int B, int h, int w, int C1, int C2, int mode, int elu, void* gx_, void* gs_,
                                  hipStream_t s) {
  const T* g = static_cast<const T*>(g_);
  const T* x = static_cast<const T*>(x_);
  T* gx = static_cast<T*>(gx_);
  T* gs = static_cast<T*>(gs_);
  const int H = mode == 2 ? h : 2 * h, W = mode == 2 ? w : 2 * w, C1q = C1 / 4, C2q = C2 / 4;
  if (gx) {
    const dim3 grid((w * C1q + DEC_NT - 1) / DEC_NT, h, B), block(DEC_NT);
#define DD_UCPB(M, E) hipLaunchKernelGGL((up_cat_pad_bwd_x_kernel<T, M, E>), grid, block, 0, s, g, x, h, w, C1q, C2q, gx)
    if (mode == 0) { if (elu) DD_UCPB(0, true); else DD_UCPB(0, false); }
    else if (mode == 1) { if (elu) DD_UCPB(1, true); else DD_UCPB(1, false); }
    else { if (elu) DD_UCPB(2, true); else DD_UCPB(2, false); }
#undef DD_UCPB
  }
  if (gs && C2q > 0)
    hipLaunchKernelGGL(up_cat_pad_bwd_skip_kernel<T>, dim3((W * C2q + DEC_NT - 1) / DEC_NT, H, B), dim3(DEC_NT), 0, s, g, H, W, C1q, C2q, gs);
}

}  // namespace dd

using namespace dd;

static bool up_cat_pad_ok(const void* x, int B, int h, int w, int C1, int C2, int mode, int dtype) {
  const int H = mode == 2 ? h : 2 * h;
  return x && B >= 1 && B <= 65535 && h >= 2 && w >= 2 && H >= 4 && H + 2 <= 65535 && C1 >= 4 && C1 % 4 == 0 && C2 >= 0 && C2 % 4 == 0 && mode >= 0 &&
         mode <= 2 && dtype >= 0 && dtype <= 2;
}

extern "C" int dd_up_cat_pad_t(const void* x, const void* skip, int B, int h, int w, int C1, int C2, int mode, int elu, void* out, int dtype,
                               void* stream) {
  if (!up_cat_pad_ok(x, B, h, w, C1, C2, mode, dtype) || !out || (C2 > 0 && !skip)) return (int)hipErrorInvalidValue;
  DD_DISPATCH_DTYPE(dtype, up_cat_pad_launch, x, skip, B, h, w, C1, C2, mode, elu, out, static_cast<hipStream_t>(stream));
  return (int)hipGetLastError();
}

extern "C" int dd_up_cat_pad_bwd_t(const void* g_out, const void* x, int B, int h, int w, int C1, int C2, int mode, int elu, void* g_x, void* g_skip,
                                   int dtype, void* stream) {
  if (!up_cat_pad_ok(x, B, h, w, C1, C2, mode, dtype) || !g_out || (!g_x && !g_skip)) return (int)hipErrorInvalidValue;
  DD_DISPATCH_DTYPE(dtype, up_cat_pad_bwd_launch, g_out, x, B, h, w, C1, C2, mode, elu, g_x, g_skip, static_cast<hipStream_t>(stream));
  return (int)hipGetLastError();
}
