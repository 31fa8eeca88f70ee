// 3x3 / stride 1 / zero-pad 1 convolutions with a handful of channels on full-resolution channels-last tensors: the last level of
// the two motion decoders (reference networks/motion_decoder.py:31-41,60-72 -- refine_motion_conv5: Conv2d(9+out_dim -> 9, 3x3),
// Conv2d(9 -> 9, 3x3) on the raw 9-channel frame stack at 192x640).  With 9-12 channels the contraction is 81-108 deep and
// 9-12 wide: MIOpen's implicit-GEMM kernels need 250-400 us per call on a problem whose tensors (53-71 MB) stream in ~25 us.
// This is a register-tiled direct convolution: one thread = two vertically adjacent output pixels x all output channels, the
// input tile (+1 halo) staged once in LDS with an odd pixel stride (conflict-free), the weights read through uniform
// (scalar) loads of a fully unrolled (tap, cin, cout) nest, outputs staged through LDS so that global stores are flat rows.
// The data gradient is the same kernel with the channel roles swapped and the taps mirrored.
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"

namespace dd {

constexpr int SC_TH = 16, SC_TW = 32, SC_NT = 256;           // 16x32 output tile, 256 threads, 2 rows per thread

// FWD: out[b,y,x,co] = bias[co] + sum w[co,ci,ky,kx] * in[b,y+ky-1,x+kx-1,ci]          (CI = in channels, CO = out channels)
// BWD: out[b,y,x,ci] =            sum w[co,ci,ky,kx] * in[b,y-ky+1,x-kx+1,co]          (CI = channels of `in` = conv's Cout)
// `w` is always the convolution's own [Cout][Cin][3][3] tensor.
template <int CI, int CO, bool BWD>
__global__ __launch_bounds__(SC_NT) void conv3x3_small_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                               const float* __restrict__ bias, int H, int W, float* __restrict__ out) {
  constexpr int PS = CI | 1;                                   // odd pixel stride in LDS
  constexpr int TWP = SC_TW + 2, THP = SC_TH + 2;
  constexpr int IN_FLOATS = THP * TWP * PS, OUT_FLOATS = SC_TH * SC_TW * CO;
  __shared__ float tile[IN_FLOATS > OUT_FLOATS ? IN_FLOATS : OUT_FLOATS];
  const int b = blockIdx.z, Y0 = blockIdx.y * SC_TH, X0 = blockIdx.x * SC_TW;
  const float* src = in + (size_t)b * H * W * CI;
  // ---- stage the input tile: rows of (TW+2)*CI contiguous floats in global memory --------------------------------------
  for (int i = threadIdx.x; i < THP * TWP * CI; i += SC_NT) {
    const int row = i / (TWP * CI), rem = i - row * (TWP * CI);
    const int px = rem / CI, c = rem - px * CI;
    const int Y = Y0 - 1 + row, X = X0 - 1 + px;
    const bool inside = Y >= 0 && Y < H && X >= 0 && X < W;
    tile[(row * TWP + px) * PS + c] = inside ? src[((size_t)Y * W + X) * CI + c] : 0.f;
  }
  __syncthreads();
  const int lx = threadIdx.x % SC_TW, ly = (threadIdx.x / SC_TW) * 2;   // rows ly, ly+1
  float acc[2][CO];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[p][o] = (!BWD && bias) ? bias[o] : 0.f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      // FWD reads in[y+ky-1, x+kx-1]; BWD reads in[y-ky+1, x-kx+1] = tile offset (2-ky, 2-kx)
      const int oy = BWD ? 2 - ky : ky, ox = BWD ? 2 - kx : kx;
      const float* p0 = tile + ((ly + oy) * TWP + lx + ox) * PS;
      const float* p1 = p0 + TWP * PS;
#pragma unroll
      for (int i = 0; i < CI; ++i) {
        const float v0 = p0[i], v1 = p1[i];
#pragma unroll
        for (int o = 0; o < CO; ++o) {
          // uniform address -> scalar load; FWD: w[o][i][ky][kx] (Cin = CI); BWD: w[i][o][ky][kx] (Cin = CO)
          const float wt = BWD ? w[((i * CO + o) * 3 + ky) * 3 + kx] : w[((o * CI + i) * 3 + ky) * 3 + kx];
          acc[0][o] = fmaf(wt, v0, acc[0][o]);
          acc[1][o] = fmaf(wt, v1, acc[1][o]);
        }
      }
    }
  __syncthreads();                                             // the input tile is dead: reuse it for the outputs
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int o = 0; o < CO; ++o) tile[((ly + p) * SC_TW + lx) * CO + o] = acc[p][o];
  __syncthreads();
  float* dst = out + (size_t)b * H * W * CO;
  const int cols = (X0 + SC_TW <= W ? SC_TW : W - X0) * CO;   // floats of one tile row that are inside the image
  for (int i = threadIdx.x; i < SC_TH * SC_TW * CO; i += SC_NT) {
    const int row = i / (SC_TW * CO), rem = i - row * (SC_TW * CO);
    const int Y = Y0 + row;
    if (Y < H && rem < cols) dst[((size_t)Y * W + X0) * CO + rem] = tile[i];
  }
}

// Weight gradient: gw[co][ci][ky][kx] = sum over pixels g[p][co] * in[p + tap][ci].  One block per 16x32 tile stages both tiles
// in LDS; thread t < 9*CO*CIG owns (tap, co, group of 4 input channels): 4 running sums; per-tile records, folded afterwards
// in a fixed order.
template <int CI, int CO>
__global__ __launch_bounds__(SC_NT) void conv3x3_small_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ g, int H, int W,
                                                                     int tiles_x, int tiles_y, float* __restrict__ partial) {
  constexpr int PS = CI | 1, GS = CO | 1;
  constexpr int TWP = SC_TW + 2, THP = SC_TH + 2;
  constexpr int CIG = (CI + 3) / 4;                            // groups of four input channels
  constexpr int ITEMS = 9 * CO * CIG;
  static_assert(ITEMS <= 2 * SC_NT, "two items per thread at most");
  __shared__ float tin[THP * TWP * PS];
  __shared__ float tg[SC_TH * SC_TW * GS];
  const int b = blockIdx.z, Y0 = blockIdx.y * SC_TH, X0 = blockIdx.x * SC_TW;
  const float* src = in + (size_t)b * H * W * CI;
  const float* gsrc = g + (size_t)b * H * W * CO;
  for (int i = threadIdx.x; i < THP * TWP * CI; i += SC_NT) {
    const int row = i / (TWP * CI), rem = i - row * (TWP * CI);
    const int px = rem / CI, c = rem - px * CI;
    const int Y = Y0 - 1 + row, X = X0 - 1 + px;
    tin[(row * TWP + px) * PS + c] = (Y >= 0 && Y < H && X >= 0 && X < W) ? src[((size_t)Y * W + X) * CI + c] : 0.f;
  }
  for (int i = threadIdx.x; i < SC_TH * SC_TW * CO; i += SC_NT) {
    const int row = i / (SC_TW * CO), rem = i - row * (SC_TW * CO);
    const int px = rem / CO, c = rem - px * CO;
    const int Y = Y0 + row, X = X0 + px;
    tg[(row * SC_TW + px) * GS + c] = (Y < H && X < W) ? gsrc[((size_t)Y * W + X) * CO + c] : 0.f;
  }
  __syncthreads();
  float* rec = partial + ((size_t)(b * tiles_y + blockIdx.y) * tiles_x + blockIdx.x) * (9 * CO * CI);
  for (int item = threadIdx.x; item < ITEMS; item += SC_NT) {
    const int tap = item / (CO * CIG), r = item - tap * (CO * CIG);
    const int co = r / CIG, cg = r - co * CIG;
    const int ky = tap / 3, kx = tap - ky * 3;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int y = 0; y < SC_TH; ++y) {
      const float* gp = tg + (y * SC_TW) * GS + co;
      const float* ip = tin + ((y + ky) * TWP + kx) * PS + cg * 4;
#pragma unroll 8
      for (int x = 0; x < SC_TW; ++x) {
        const float gv = gp[x * GS];
        const float* q = ip + x * PS;
        a0 = fmaf(gv, q[0], a0);
        if (cg * 4 + 1 < CI) a1 = fmaf(gv, q[1], a1);
        if (cg * 4 + 2 < CI) a2 = fmaf(gv, q[2], a2);
        if (cg * 4 + 3 < CI) a3 = fmaf(gv, q[3], a3);
      }
    }
    const float a[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (cg * 4 + j < CI) rec[((co * CI + cg * 4 + j) * 3 + ky) * 3 + kx] = a[j];
  }
}

// gw[i] = sum over tiles of partial[tile][i]: one wave per output, fixed order
__global__ __launch_bounds__(256) void conv3x3_small_wfold_kernel(const float* __restrict__ partial, int ntiles, int n, float* __restrict__ gw) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (o >= n) return;
  float s = 0.f;
  for (int t = l; t < ntiles; t += 64) s += partial[(size_t)t * n + o];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (l == 0) gw[o] = s;
}

template <int CI, int CO, bool BWD>
static int launch_small(const float* in, const float* w, const float* bias, int B, int H, int W, float* out, hipStream_t s) {
  const dim3 grid((W + SC_TW - 1) / SC_TW, (H + SC_TH - 1) / SC_TH, B);
  hipLaunchKernelGGL((conv3x3_small_kernel<CI, CO, BWD>), grid, dim3(SC_NT), 0, s, in, w, bias, H, W, out);
  return (int)hipGetLastError();
}

template <int CI, int CO>
static int launch_small_wgrad(const float* in, const float* g, int B, int H, int W, float* gw, float* partial, hipStream_t s) {
  const int tx = (W + SC_TW - 1) / SC_TW, ty = (H + SC_TH - 1) / SC_TH;
  hipLaunchKernelGGL((conv3x3_small_wgrad_kernel<CI, CO>), dim3(tx, ty, B), dim3(SC_NT), 0, s, in, g, H, W, tx, ty, partial);
  const int n = 9 * CO * CI;
  hipLaunchKernelGGL(conv3x3_small_wfold_kernel, dim3((n + 3) / 4), dim3(256), 0, s, partial, tx * ty * B, n, gw);
  return (int)hipGetLastError();
}

}  // namespace dd

using namespace dd;

// channel pairs of the reference's networks: (12 -> 9) flow decoder, (10 -> 9) mask decoder, (9 -> 9) both
extern "C" int dd_conv3x3_small_supported(int c_in, int c_out) {
  return (c_out == 9 && (c_in == 9 || c_in == 10 || c_in == 12)) ? 1 : 0;
}

extern "C" int dd_conv3x3_small_fwd(const float* x, const float* weight, const float* bias, int B, int H, int W, int c_in, int c_out,
                                    float* out, void* stream) {
  if (!x || !weight || !out || B < 1 || H < 1 || W < 1 || B > 65535 || !dd_conv3x3_small_supported(c_in, c_out)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c_in == 12) return launch_small<12, 9, false>(x, weight, bias, B, H, W, out, s);
  if (c_in == 10) return launch_small<10, 9, false>(x, weight, bias, B, H, W, out, s);
  return launch_small<9, 9, false>(x, weight, bias, B, H, W, out, s);
}

extern "C" int dd_conv3x3_small_bwd_data(const float* g_out, const float* weight, int B, int H, int W, int c_in, int c_out, float* g_x,
                                         void* stream) {
  if (!g_out || !weight || !g_x || B < 1 || H < 1 || W < 1 || B > 65535 || !dd_conv3x3_small_supported(c_in, c_out)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // kernel's CI = channels of g_out (= c_out), kernel's CO = channels of g_x (= c_in)
  if (c_in == 12) return launch_small<9, 12, true>(g_out, weight, nullptr, B, H, W, g_x, s);
  if (c_in == 10) return launch_small<9, 10, true>(g_out, weight, nullptr, B, H, W, g_x, s);
  return launch_small<9, 9, true>(g_out, weight, nullptr, B, H, W, g_x, s);
}

extern "C" size_t dd_conv3x3_small_workspace_bytes(int B, int H, int W, int c_in, int c_out) {
  const size_t tiles = (size_t)B * ((H + SC_TH - 1) / SC_TH) * ((W + SC_TW - 1) / SC_TW);
  return tiles * 9 * c_in * c_out * sizeof(float);
}

extern "C" int dd_conv3x3_small_bwd_weight(const float* x, const float* g_out, int B, int H, int W, int c_in, int c_out, float* g_weight,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !g_out || !g_weight || !workspace || B < 1 || H < 1 || W < 1 || B > 65535 || !dd_conv3x3_small_supported(c_in, c_out))
    return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_conv3x3_small_workspace_bytes(B, H, W, c_in, c_out)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  if (c_in == 12) return launch_small_wgrad<12, 9>(x, g_out, B, H, W, g_weight, partial, s);
  if (c_in == 10) return launch_small_wgrad<10, 9>(x, g_out, B, H, W, g_weight, partial, s);
  return launch_small_wgrad<9, 9>(x, g_out, B, H, W, g_weight, partial, s);
}
