// dd_conv_head.hip -- the disparity heads: a 3x3, stride-1 convolution to ONE output channel on an input that already carries its
// reflection padding (reference networks/depth_decoder.py:49-51,95-97 `Conv3x3(num_ch_dec[s], 1)`, networks/layers.py:103-121; the
// padded tensor comes from dd_up_cat_pad).  Per step the heads run on 36 images (the target frame's pass and the statistics-only pass
// of frames -1/+1) at up to 96x320 with 32 channels: 97 MB in, 3 MB out -- ~20 us of bytes -- where the library's implicit GEMM with its
// 16-wide N tile takes 91-160 us (forward) and 90 us (weight gradient).  The data gradient is dd_conv3x3_cout1_bwd_data (dd_dwconv.hip).
//   forward:         out[b,h,w] = bias + sum_{kh,kw,c} x[b,h+kh,w+kw,c] * w[c,kh,kw]
//   weight gradient: gw[kh,kw,c] = sum_{b,h,w} g[b,h,w] * x[b,h+kh,w+kw,c],  gb = sum g
// Both walk the tensor the same way: C/4 lanes per pixel (each lane owns four channels: 16-byte loads, 128-256 contiguous bytes per
// pixel, consecutive pixels contiguous), a workgroup covers a strip of 256/(C/4) columns and walks down `rows` output rows with a
// rolling 3-row window in registers, so every input element is loaded once per column strip (three strips share a column through the
// cache).  Forward: nine 4-wide dot products, a shuffle reduction over the pixel's lanes.  Weight gradient: 36 accumulators per lane,
// reduced over the workgroup's pixels in LDS, one partial per workgroup, folded in a fixed order in two levels (no atomics).
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"

namespace dd {

constexpr int CH_NT = 256;
constexpr int CH_ROWS = 8;
constexpr int CH_FOLD = 32;

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }

template <int LPP>
__global__ __launch_bounds__(CH_NT) void conv_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, long long s_ci, long long s_kh,
                                                              long long s_kw, const float* __restrict__ bias, int Hp, int Wp, float* __restrict__ out) {
  constexpr int PX = CH_NT / LPP, C = 4 * LPP;
  const int Ho = Hp - 2, Wo = Wp - 2;
  const int p = threadIdx.x / LPP, q = threadIdx.x % LPP;
  const int b = blockIdx.z, h0 = blockIdx.y * CH_ROWS;
  const int wo = blockIdx.x * PX + p;
  const int wc = wo < Wo ? wo : Wo - 1;                  // lanes beyond the row compute a valid column and do not store
  float4 wk[3][3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float* wp = w + kh * s_kh + kw * s_kw + (long long)(4 * q) * s_ci;
      wk[kh][kw] = make_float4(wp[0], wp[s_ci], wp[2 * s_ci], wp[3 * s_ci]);
    }
  const float4* xb = reinterpret_cast<const float4*>(x) + ((long long)b * Hp * Wp) * LPP;
  float4 win[3][3];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) win[r][kw] = xb[((long long)(h0 + r) * Wp + wc + kw) * LPP + q];
  const float bv = bias ? bias[0] : 0.f;
  const int h1 = h0 + CH_ROWS < Ho ? h0 + CH_ROWS : Ho;
  for (int h = h0; h < h1; ++h) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) win[2][kw] = xb[((long long)(h + 2) * Wp + wc + kw) * LPP + q];
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) acc += dot4(win[kh][kw], wk[kh][kw]);
#pragma unroll
    for (int m = 1; m < LPP; m <<= 1) acc += __shfl_xor(acc, m, 64);
    if (q == 0 && wo < Wo) out[((long long)b * Ho + h) * Wo + wo] = acc + bv;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) { win[0][kw] = win[1][kw]; win[1][kw] = win[2][kw]; }
  }
  (void)C;
}

// partial[block][9 * C + 1]: (kh, kw, c) then the sum of g
template <int LPP>
__global__ __launch_bounds__(CH_NT) void conv_head_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g, int Hp, int Wp,
                                                                float* __restrict__ part) {
  constexpr int PX = CH_NT / LPP, C = 4 * LPP, N = 9 * C + 1;
  __shared__ float s_red[CH_NT * 37];                    // 36 accumulators + the lane's share of sum(g), stride 37 (odd)
  const int Ho = Hp - 2, Wo = Wp - 2;
  const int p = threadIdx.x / LPP, q = threadIdx.x % LPP;
  const int b = blockIdx.z, h0 = blockIdx.y * CH_ROWS;
  const int wo = blockIdx.x * PX + p;
  const bool live = wo < Wo;
  const int wc = live ? wo : Wo - 1;
  const float4* xb = reinterpret_cast<const float4*>(x) + ((long long)b * Hp * Wp) * LPP;
  float4 win[3][3], a[3][3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) a[kh][kw] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) win[r][kw] = xb[((long long)(h0 + r) * Wp + wc + kw) * LPP + q];
  float gs = 0.f;
  const int h1 = h0 + CH_ROWS < Ho ? h0 + CH_ROWS : Ho;
  for (int h = h0; h < h1; ++h) {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) win[2][kw] = xb[((long long)(h + 2) * Wp + wc + kw) * LPP + q];
    const float gv = live ? g[((long long)b * Ho + h) * Wo + wo] : 0.f;
    gs += gv;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        a[kh][kw].x = fmaf(gv, win[kh][kw].x, a[kh][kw].x); a[kh][kw].y = fmaf(gv, win[kh][kw].y, a[kh][kw].y);
        a[kh][kw].z = fmaf(gv, win[kh][kw].z, a[kh][kw].z); a[kh][kw].w = fmaf(gv, win[kh][kw].w, a[kh][kw].w);
      }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) { win[0][kw] = win[1][kw]; win[1][kw] = win[2][kw]; }
  }
  float* mine = s_red + threadIdx.x * 37;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int t = (kh * 3 + kw) * 4;
      mine[t] = a[kh][kw].x; mine[t + 1] = a[kh][kw].y; mine[t + 2] = a[kh][kw].z; mine[t + 3] = a[kh][kw].w;
    }
  mine[36] = q == 0 ? gs : 0.f;
  __syncthreads();
  // element e < 9*C: (tap, c) -> lane q = c / 4, slot tap * 4 + c % 4, summed over the PX pixels in order; element 9*C: sum of g
  const long long blk = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  for (int e = threadIdx.x; e < N; e += CH_NT) {
    float s = 0.f;
    if (e < 9 * C) {
      const int tap = e / C, c = e - tap * C, qq = c >> 2, slot = tap * 4 + (c & 3);
#pragma unroll 4
      for (int pp = 0; pp < PX; ++pp) s += s_red[(pp * LPP + qq) * 37 + slot];
    } else {
#pragma unroll 4
      for (int pp = 0; pp < PX; ++pp) s += s_red[(pp * LPP) * 37 + 36];
    }
    part[blk * N + e] = s;
  }
}

// column sums of partial[nparts][N] in a fixed order, two levels (see dd_conv_small.hip)
__global__ __launch_bounds__(CH_NT) void conv_head_fold1_kernel(const float* __restrict__ part, int nparts, int N, float* __restrict__ slices) {
  const int e = blockIdx.x * CH_NT + threadIdx.x, sl = blockIdx.y;
  if (e >= N) return;
  float a0 = 0.f, a1 = 0.f;
  int i = sl;
  for (; i + CH_FOLD < nparts; i += 2 * CH_FOLD) { a0 += part[(long long)i * N + e]; a1 += part[(long long)(i + CH_FOLD) * N + e]; }
  if (i < nparts) a0 += part[(long long)i * N + e];
  slices[(long long)sl * N + e] = a0 + a1;
}
__global__ __launch_bounds__(CH_NT) void conv_head_fold2_kernel(const float* __restrict__ slices, int N, float* __restrict__ gw, float* __restrict__ gb) {
  const int e = blockIdx.x * CH_NT + threadIdx.x;
  if (e >= N) return;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < CH_FOLD; ++i) a[i & 3] += slices[(long long)i * N + e];
  const float v = (a[0] + a[1]) + (a[2] + a[3]);
  if (e < N - 1) gw[e] = v;                               // (kh, kw, c): the memory order of a channels-last (1, C, 3, 3) weight
  else if (gb) gb[0] = v;
}

static inline int head_lpp(int C) { return C == 32 ? 8 : (C == 64 ? 16 : 0); }

}  // namespace dd

extern "C" int dd_conv_head_supported(int C) { return dd::head_lpp(C) != 0; }

extern "C" size_t dd_conv_head_workspace_bytes(int B, int Hp, int Wp, int C) {
  const int lpp = dd::head_lpp(C);
  if (!lpp || Hp < 3 || Wp < 3) return 0;
  const int px = dd::CH_NT / lpp, Ho = Hp - 2, Wo = Wp - 2;
  const size_t blocks = (size_t)B * ((Ho + dd::CH_ROWS - 1) / dd::CH_ROWS) * ((Wo + px - 1) / px);
  return (blocks + dd::CH_FOLD) * (size_t)(9 * C + 1) * sizeof(float);
}

extern "C" int dd_conv_head_fwd(const float* x_padded, const float* weight, long long s_ci, long long s_kh, long long s_kw, const float* bias, int B, int Hp,
                                int Wp, int C, float* out, void* stream) {
  const int lpp = dd::head_lpp(C);
  if (!x_padded || !weight || !out || !lpp || B < 1 || Hp < 3 || Wp < 3) return (int)hipErrorInvalidValue;
  const int px = dd::CH_NT / lpp, Ho = Hp - 2, Wo = Wp - 2;
  const dim3 grid((Wo + px - 1) / px, (Ho + dd::CH_ROWS - 1) / dd::CH_ROWS, B);
  hipStream_t st = (hipStream_t)stream;
  if (lpp == 8) hipLaunchKernelGGL((dd::conv_head_fwd_kernel<8>), grid, dim3(dd::CH_NT), 0, st, x_padded, weight, s_ci, s_kh, s_kw, bias, Hp, Wp, out);
  else hipLaunchKernelGGL((dd::conv_head_fwd_kernel<16>), grid, dim3(dd::CH_NT), 0, st, x_padded, weight, s_ci, s_kh, s_kw, bias, Hp, Wp, out);
  return (int)hipGetLastError();
}

extern "C" int dd_conv_head_bwd_weight(const float* x_padded, const float* g_out, int B, int Hp, int Wp, int C, float* g_weight, float* g_bias,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  const int lpp = dd::head_lpp(C);
  if (!x_padded || !g_out || !g_weight || !workspace || !lpp || B < 1 || Hp < 3 || Wp < 3) return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_conv_head_workspace_bytes(B, Hp, Wp, C)) return (int)hipErrorInvalidValue;
  const int px = dd::CH_NT / lpp, Ho = Hp - 2, Wo = Wp - 2, N = 9 * C + 1;
  const dim3 grid((Wo + px - 1) / px, (Ho + dd::CH_ROWS - 1) / dd::CH_ROWS, B);
  const int blocks = (int)(grid.x * grid.y * grid.z);
  float* part = static_cast<float*>(workspace);
  float* slices = part + (size_t)blocks * N;
  hipStream_t st = (hipStream_t)stream;
  if (lpp == 8) hipLaunchKernelGGL((dd::conv_head_wgrad_kernel<8>), grid, dim3(dd::CH_NT), 0, st, x_padded, g_out, Hp, Wp, part);
  else hipLaunchKernelGGL((dd::conv_head_wgrad_kernel<16>), grid, dim3(dd::CH_NT), 0, st, x_padded, g_out, Hp, Wp, part);
  hipLaunchKernelGGL(dd::conv_head_fold1_kernel, dim3((N + dd::CH_NT - 1) / dd::CH_NT, dd::CH_FOLD), dim3(dd::CH_NT), 0, st, part, blocks, N, slices);
  hipLaunchKernelGGL(dd::conv_head_fold2_kernel, dim3((N + dd::CH_NT - 1) / dd::CH_NT), dim3(dd::CH_NT), 0, st, slices, N, g_weight, g_bias);
  return (int)hipGetLastError();
}
