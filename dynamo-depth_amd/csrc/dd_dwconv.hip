// Depth-wise dilated 3x3 convolution, channels-last, for LiteMono's DilatedConv blocks
// (reference: networks/depth_encoder.py:168-181 CDilated, used with groups == channels, stride 1, padding == dilation, no bias).
// MIOpen has no dilated grouped convolution, so ATen falls back to an NCHW-only kernel (two layout copies per block) and the
// dilation-1 blocks go through a batched-GEMM weight gradient that takes >1 ms on a 23 MB tensor. This is HBM-bound byte work:
// one float4 (four channels) per thread, rows of the NHWC tensor read fully coalesced, the nine taps come from L1/L2.
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_half.h"

namespace dd {

constexpr int DW_NT = 256;
constexpr int DW_ROWS = 4;          // output rows per thread: amortises the weight staging
constexpr int DW_MAX_C = 512;

// out[b,y,x,c] = sum_t w[c, t] * in[b, y + (ty-1) d, x + (tx-1) d, c]   (FLIP: taps mirrored = the data gradient)
// T: storage type of in / out (fp32, or fp16 / bf16 under autocast); the weights are the fp32 master copy, accumulation fp32
template <bool FLIP, typename T>
__global__ __launch_bounds__(DW_NT) void dwconv3x3_kernel(const T* __restrict__ in, const float* __restrict__ w, int H, int W, int C,
                                                           int dil, T* __restrict__ out) {
  __shared__ float4 wt[9 * DW_MAX_C / 4];                     // [tap][c4]
  const int C4 = C >> 2;
  for (int i = threadIdx.x; i < 9 * C; i += DW_NT) {
    const int c = i / 9, t = i - c * 9;
    reinterpret_cast<float*>(wt)[(FLIP ? 8 - t : t) * C + c] = w[i];
  }
  __syncthreads();
  const int j = blockIdx.x * DW_NT + threadIdx.x;             // x * C4 + c4 inside a row
  if (j >= W * C4) return;
  const int x = j / C4, c4 = j - x * C4;
  const int b = blockIdx.z, y0 = blockIdx.y * DW_ROWS;
  const long long img = (long long)b * H * W * C4;
  float4 k[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) k[t] = wt[t * C4 + c4];
  // taps outside the image read a clamped (valid) address and are multiplied by 0: no divergent loads, so all 36 loads of the
  // four rows are in flight together instead of nine at a time behind branches
  int xo[3];
  float xm[3];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int xx = x + (tx - 1) * dil;
    xm[tx] = (xx >= 0 && xx < W) ? 1.f : 0.f;
    xo[tx] = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
  }
#pragma unroll
  for (int r = 0; r < DW_ROWS; ++r) {
    const int y = y0 + r;
    const int yc = y < H ? y : H - 1;                          // rows past the end recompute the last row and are not stored
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const int yy = yc + (ty - 1) * dil;
      const float ym = (yy >= 0 && yy < H) ? 1.f : 0.f;
      const int yo = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const float4 v = IO<T>::load4(in, img + ((long long)yo * W + xo[tx]) * C4 + c4);
        const float m = ym * xm[tx];
        const float4 q = k[ty * 3 + tx];
        acc.x = fmaf(q.x * m, v.x, acc.x); acc.y = fmaf(q.y * m, v.y, acc.y); acc.z = fmaf(q.z * m, v.z, acc.z); acc.w = fmaf(q.w * m, v.w, acc.w);
      }
    }
    if (y < H) IO<T>::store4(out, img + ((long long)y * W + x) * C4 + c4, acc);
  }
}

// weight gradient, stage 1: one block per image row; thread = (pixel lane, c4); 36 running sums per thread, folded over the
// pixel lanes through LDS in a fixed order, one [9][C] record per row
template <typename T>
__global__ __launch_bounds__(DW_NT) void dwconv3x3_wgrad_rows_kernel(const T* __restrict__ g, const T* __restrict__ in, int H, int W,
                                                                      int C, int dil, int lanes, float* __restrict__ partial) {
  extern __shared__ float red[];                               // [lanes][9][C]
  const int C4 = C >> 2;
  const int lane = threadIdx.x / C4, c4 = threadIdx.x - lane * C4;
  const int y = blockIdx.x, b = blockIdx.y;
  const long long grow = ((long long)b * H + y) * W * C4, img = (long long)b * H * W * C4;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < lanes) {
    for (int x = lane; x < W; x += lanes) {
      const float4 gv = IO<T>::load4(g, grow + (long long)x * C4 + c4);
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const int yy = y + (ty - 1) * dil;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const int xx = x + (tx - 1) * dil;
          if (xx < 0 || xx >= W) continue;
          const float4 v = IO<T>::load4(in, img + ((long long)yy * W + xx) * C4 + c4);
          float4& a = acc[ty * 3 + tx];
          a.x = fmaf(gv.x, v.x, a.x); a.y = fmaf(gv.y, v.y, a.y); a.z = fmaf(gv.z, v.z, a.z); a.w = fmaf(gv.w, v.w, a.w);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) reinterpret_cast<float4*>(red)[(lane * 9 + t) * C4 + c4] = acc[t];
  }
  __syncthreads();
  float* dst = partial + ((size_t)b * H + y) * 9 * C;
  for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * 9 * C + i];
    dst[i] = s;
  }
}

// stage 2: gw[c][t] = sum over rows of partial[row][t][c], fixed order, a few rows-slices per output for parallelism
constexpr int DW_FOLD = 8;
__global__ __launch_bounds__(DW_NT) void dwconv3x3_wgrad_fold_kernel(const float* __restrict__ partial, int rows, int C, float* __restrict__ gw) {
  __shared__ float part[DW_NT];
  const int per = DW_NT / DW_FOLD;                             // outputs per block
  const int o = blockIdx.x * per + threadIdx.x % per, slice = threadIdx.x / per;
  float s = 0.f;
  if (o < 9 * C) {
#pragma unroll 8
    for (int r = slice; r < rows; r += DW_FOLD) s += partial[(size_t)r * 9 * C + o];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if (slice == 0 && o < 9 * C) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < DW_FOLD; ++k) tot += part[k * per + threadIdx.x];
    const int t = o / C, c = o - t * C;
    gw[c * 9 + t] = tot;
  }
}

static inline bool dw_dims_ok(int B, int H, int W, int C, int dil) {
  return B >= 1 && H >= 1 && W >= 1 && C >= 4 && (C & 3) == 0 && C <= DW_MAX_C && dil >= 1 && B <= 65535 && H <= 65535;
}

}  // namespace dd

using namespace dd;

template <typename T>
static void dw_launch(bool flip, const void* in, const float* w, int B, int H, int W, int C, int dil, void* out, hipStream_t s) {
  const dim3 grid((W * (C >> 2) + DW_NT - 1) / DW_NT, (H + DW_ROWS - 1) / DW_ROWS, B);
  if (flip) hipLaunchKernelGGL((dwconv3x3_kernel<true, T>), grid, dim3(DW_NT), 0, s, static_cast<const T*>(in), w, H, W, C, dil, static_cast<T*>(out));
  else hipLaunchKernelGGL((dwconv3x3_kernel<false, T>), grid, dim3(DW_NT), 0, s, static_cast<const T*>(in), w, H, W, C, dil, static_cast<T*>(out));
}

static int launch_dw(bool flip, const void* in, const float* w, int B, int H, int W, int C, int dil, void* out, int dtype, void* stream) {
  if (!in || !w || !out || !dw_dims_ok(B, H, W, C, dil) || dtype < 0 || dtype > 2) return (int)hipErrorInvalidValue;
  DD_DISPATCH_DTYPE(dtype, dw_launch, flip, in, w, B, H, W, C, dil, out, static_cast<hipStream_t>(stream));
  return (int)hipGetLastError();
}

extern "C" int dd_dwconv3x3_nhwc_t(const void* x, const float* weight, int B, int H, int W, int C, int dilation, void* out, int dtype, void* stream) {
  return launch_dw(false, x, weight, B, H, W, C, dilation, out, dtype, stream);
}

extern "C" int dd_dwconv3x3_nhwc_bwd_data_t(const void* g_out, const float* weight, int B, int H, int W, int C, int dilation, void* g_x, int dtype,
                                            void* stream) {
  return launch_dw(true, g_out, weight, B, H, W, C, dilation, g_x, dtype, stream);
}

extern "C" int dd_dwconv3x3_nhwc(const float* x, const float* weight, int B, int H, int W, int C, int dilation, float* out, void* stream) {
  return launch_dw(false, x, weight, B, H, W, C, dilation, out, 0, stream);
}

extern "C" int dd_dwconv3x3_nhwc_bwd_data(const float* g_out, const float* weight, int B, int H, int W, int C, int dilation, float* g_x,
                                          void* stream) {
  return launch_dw(true, g_out, weight, B, H, W, C, dilation, g_x, 0, stream);
}

extern "C" size_t dd_dwconv3x3_workspace_bytes(int B, int H, int C) { return (size_t)B * H * 9 * C * sizeof(float); }

template <typename T>
static void dw_wgrad_launch(const void* g_out, const void* x, int B, int H, int W, int C, int dil, int lanes, int threads, size_t lds, float* partial,
                            hipStream_t s) {
  hipLaunchKernelGGL((dwconv3x3_wgrad_rows_kernel<T>), dim3(H, B), dim3(threads), lds, s, static_cast<const T*>(g_out), static_cast<const T*>(x), H, W,
                     C, dil, lanes, partial);
}

extern "C" int dd_dwconv3x3_nhwc_bwd_weight_t(const void* g_out, const void* x, int B, int H, int W, int C, int dilation, float* g_weight,
                                              void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  if (!g_out || !x || !g_weight || !workspace || !dw_dims_ok(B, H, W, C, dilation) || dtype < 0 || dtype > 2) return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_dwconv3x3_workspace_bytes(B, H, C)) return (int)hipErrorInvalidValue;
  const int C4 = C >> 2;
  int lanes = DW_NT / C4;
  if (lanes > W) lanes = W;
  if (lanes < 1) lanes = 1;
  const int threads = lanes * C4;                              // <= 256 since C <= 512 -> C4 <= 128
  const size_t lds = (size_t)lanes * 9 * C * sizeof(float);
  float* partial = static_cast<float*>(workspace);
  hipStream_t s = static_cast<hipStream_t>(stream);
  DD_DISPATCH_DTYPE(dtype, dw_wgrad_launch, g_out, x, B, H, W, C, dilation, lanes, threads, lds, partial, s);
  const int per = DW_NT / DW_FOLD;
  hipLaunchKernelGGL(dwconv3x3_wgrad_fold_kernel, dim3((9 * C + per - 1) / per), dim3(DW_NT), 0, s, partial, B * H, C, g_weight);
  return (int)hipGetLastError();
}

extern "C" int dd_dwconv3x3_nhwc_bwd_weight(const float* g_out, const float* x, int B, int H, int W, int C, int dilation, float* g_weight,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  return dd_dwconv3x3_nhwc_bwd_weight_t(g_out, x, B, H, W, C, dilation, g_weight, workspace, workspace_bytes, 0, stream);
}

// ---- data gradient of a 3x3 convolution with ONE output channel (the disparity heads `dispconv`, reference
// networks/depth_decoder.py:49-51,95-97: Conv3x3(num_ch_dec[s], 1) after a reflection pad) ---------------------------------
// MIOpen falls back to its naive kernel here (165 us for a 47 MB gradient).  It is an outer product: every input pixel
// gathers nine scalars of g and scales them with its channel's nine weights -- one float4 store per thread, g from L1.
namespace dd {

__global__ __launch_bounds__(DW_NT) void conv3x3_cout1_bwd_data_kernel(const float* __restrict__ g, const float* __restrict__ w, int Hi, int Wi,
                                                                        int C, int pad, int Ho, int Wo, float* __restrict__ gx) {
  __shared__ float4 wt[9 * DW_MAX_C / 4];                     // [tap][c4]
  const int C4 = C >> 2;
  for (int i = threadIdx.x; i < 9 * C; i += DW_NT) {
    const int c = i / 9, t = i - c * 9;
    reinterpret_cast<float*>(wt)[t * C + c] = w[i];           // w: [1][C][3][3]
  }
  __syncthreads();
  const int j = blockIdx.x * DW_NT + threadIdx.x;
  if (j >= Wi * C4) return;
  const int x = j / C4, c4 = j - x * C4;
  const int y = blockIdx.y, b = blockIdx.z;
  const float* gb = g + (size_t)b * Ho * Wo;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yo = y - ky + pad;
    if (yo < 0 || yo >= Ho) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int xo = x - kx + pad;
      if (xo < 0 || xo >= Wo) continue;
      const float gv = gb[(size_t)yo * Wo + xo];
      const float4 q = wt[(ky * 3 + kx) * C4 + c4];
      acc.x = fmaf(q.x, gv, acc.x); acc.y = fmaf(q.y, gv, acc.y); acc.z = fmaf(q.z, gv, acc.z); acc.w = fmaf(q.w, gv, acc.w);
    }
  }
  reinterpret_cast<float4*>(gx)[((size_t)(b * Hi + y) * Wi + x) * C4 + c4] = acc;
}

}  // namespace dd

extern "C" int dd_conv3x3_cout1_bwd_data(const float* g_out, const float* weight, int B, int Hi, int Wi, int C, int padding, float* g_x,
                                         void* stream) {
  if (!g_out || !weight || !g_x || B < 1 || B > 65535 || Hi < 3 || Wi < 3 || Hi > 65535 || C < 4 || (C & 3) || C > dd::DW_MAX_C ||
      padding < 0 || padding > 1)
    return (int)hipErrorInvalidValue;
  const int Ho = Hi + 2 * padding - 2, Wo = Wi + 2 * padding - 2;
  const dim3 grid((Wi * (C >> 2) + dd::DW_NT - 1) / dd::DW_NT, Hi, B);
  hipLaunchKernelGGL(dd::conv3x3_cout1_bwd_data_kernel, grid, dim3(dd::DW_NT), 0, static_cast<hipStream_t>(stream), g_out, weight, Hi, Wi, C,
                     padding, Ho, Wo, g_x);
  return (int)hipGetLastError();
}
