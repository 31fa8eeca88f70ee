// dd_conv_small.hip -- the full-resolution convolutions of the motion decoders (reference networks/motion_decoder.py:24-33,57-66: at the
// finest level `refine_motion_conv5` is two 3x3 convolutions on 9-12 channels of the 192x640 input stack and `refine_motion_redu5` a 1x1
// reduction to 3 / 1 channels, for the flow decoder and for the mask decoder).  A (12, 9..12, 192, 640) fp32 tensor is 53-71 MB: forward,
// data gradient and weight gradient are each ~25 us of bytes and 2-3 GFLOP.  MIOpen's implicit-GEMM kernels pad such channel counts up
// to their 16/32-wide tiles: 130-160 us forward, 180-200 us data gradient, 150-160 us weight gradient per convolution
// (profiles/r04_small_convs.txt) -- ~2.5 ms per step for twelve 3x3 and twelve 1x1 launches.  Here:
//  * forward / data gradient: a DIRECT convolution, one output pixel (all its channels) per thread, the input tile of a workgroup
//    (8 x 32 pixels + halo) staged once through LDS with an odd per-pixel stride (no bank conflicts between neighbouring pixels), the
//    weights read through the scalar cache (uniform addresses: `s_load` + FMAs with an SGPR operand), channel counts compile-time so
//    that the 972 multiply-adds of a pixel are straight-line code.  The data gradient is the same kernel on flipped, transposed weights.
//  * weight gradient: a GEMM  dW[(tap, ci), co] = sum over pixels  patch[pixel, (tap, ci)] * g[pixel, co]  with K = 1.5 M pixels -- on
//    the matrix pipe (v_mfma_f32_16x16x4_f32: four pixels per instruction), persistent workgroups accumulating in registers, one
//    partial per workgroup, folded in a fixed order (bit-reproducible; MIOpen's kernels split K with float atomics).  One extra patch
//    row of ones yields the bias gradient in the same pass.
// Channels-last fp32 only; everything else stays with the library (these are network layers, not the loss path).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dynamo_hip.h"

namespace dd {

constexpr int CS_NT = 256, CS_TH = 8, CS_TW = 32;
constexpr int CS_MAXC = 16;
constexpr int CS_WG_BLOCKS = 1024;                      // persistent workgroups of the weight-gradient kernel
typedef float cs_f4 __attribute__((ext_vector_type(4)));

// wp[(tap * cin_k + ci) * cout_k + co]: the weights in the order the direct kernel walks them.
//   transpose 0 (forward):        cin_k = cin,  cout_k = cout, wp = w[co][ci][kh][kw]
//   transpose 1 (data gradient):  cin_k = cout, cout_k = cin,  wp[(tap*cout + co)*cin + ci] = w[co][ci][ks-1-kh][ks-1-kw]
__global__ __launch_bounds__(CS_NT) void conv_small_prep_kernel(const float* __restrict__ w, long long s_co, long long s_ci, long long s_kh, long long s_kw,
                                                                int cout, int cin, int ks, int transpose, float* __restrict__ wp) {
  const int i = blockIdx.x * CS_NT + threadIdx.x;
  const int n = ks * ks * cin * cout;
  if (i >= n) return;
  int co, ci, tap;
  if (!transpose) { co = i % cout; ci = (i / cout) % cin; tap = i / (cout * cin); }
  else { ci = i % cin; co = (i / cin) % cout; tap = i / (cout * cin); }
  int kh = tap / ks, kw = tap % ks;
  if (transpose) { kh = ks - 1 - kh; kw = ks - 1 - kw; }
  wp[i] = w[co * s_co + ci * s_ci + kh * s_kh + kw * s_kw];
}

// stages rows [h0-R, h0+TH+R) x columns [w0-R, w0+TW+R) of image b (zero outside the image) at s[(r * TWH + px) * cs + c].
// A tile row is one contiguous run of TWH * cin floats in memory: coalesced whatever the channel count.  All loads of the tile are
// issued before the first LDS store (a load -> store loop runs at one memory latency per row).
template <int R>
__device__ __forceinline__ void stage_tile(const float* __restrict__ x, int b, int H, int W, int cin, int cs, int h0, int w0, float* __restrict__ s) {
  constexpr int TH = CS_TH + 2 * R, TWH = CS_TW + 2 * R;
  constexpr int IT = (TWH * CS_MAXC + CS_NT - 1) / CS_NT;
  const int rowf = TWH * cin;
  const unsigned magic = (65536u + (unsigned)cin - 1u) / (unsigned)cin;          // i / cin for i < 4096 (i * cin < 65536)
  int dst[IT];
  bool ok[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int i = threadIdx.x + it * CS_NT;
    const int px = (int)(((unsigned)i * magic) >> 16), c = i - px * cin;
    const int ww = w0 - R + px;
    dst[it] = px * cs + c;
    ok[it] = i < rowf && ww >= 0 && ww < W;
  }
  float v[TH][IT];
#pragma unroll
  for (int r = 0; r < TH; ++r) {
    const int hh = h0 - R + r;
    const bool rin = hh >= 0 && hh < H;
    const long long row0 = ((long long)(b * H + hh) * W + (w0 - R)) * cin;
#pragma unroll
    for (int it = 0; it < IT; ++it) v[r][it] = (rin && ok[it]) ? x[row0 + threadIdx.x + it * CS_NT] : 0.f;
  }
#pragma unroll
  for (int r = 0; r < TH; ++r) {
#pragma unroll
    for (int it = 0; it < IT; ++it)
      if (threadIdx.x + it * CS_NT < rowf) s[r * TWH * cs + dst[it]] = v[r][it];
  }
}

template <int KS, int CI, int CO>
__global__ __launch_bounds__(CS_NT) void conv_small_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, int H,
                                                           int W, float* __restrict__ y) {
  constexpr int R = KS / 2, TWH = CS_TW + 2 * R, CSTR = CI | 1;
  extern __shared__ float s_x[];
  const int b = blockIdx.z, h0 = blockIdx.y * CS_TH, w0 = blockIdx.x * CS_TW;
  stage_tile<R>(x, b, H, W, CI, CSTR, h0, w0, s_x);
  __syncthreads();
  const int ty = threadIdx.x / CS_TW, tx = threadIdx.x % CS_TW;
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = bias ? bias[co] : 0.f;
#pragma unroll
  for (int kh = 0; kh < KS; ++kh) {
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) {
      const float* sp = s_x + ((ty + kh) * TWH + tx + kw) * CSTR;
      const float* wq = wp + (kh * KS + kw) * CI * CO;
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        const float xv = sp[ci];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = fmaf(xv, wq[ci * CO + co], acc[co]);
      }
    }
  }
  const int h = h0 + ty, w = w0 + tx;
  if (h < H && w < W) {
    float* dst = y + ((long long)(b * H + h) * W + w) * CO;
#pragma unroll
    for (int co = 0; co < CO; ++co) dst[co] = acc[co];
  }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------
// rows m of the patch matrix: m = tap * cin + ci for m < KS*KS*cin, then ONE row of ones (bias gradient), then zero rows up to MT*16.
// v_mfma_f32_16x16x4_f32: lane l supplies A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16]; it holds D[4 * (l / 16) + r][l % 16], r = 0..3.
template <int KS, int MT>
__global__ __launch_bounds__(CS_NT) void conv_small_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g, int B, int H, int W, int cin, int cout,
                                                                 int tiles_x, int tiles_y, float* __restrict__ part) {
  constexpr int R = KS / 2, TH = CS_TH + 2 * R, TWH = CS_TW + 2 * R;
  extern __shared__ float s_mem[];
  const int cs = cin | 1;
  float* s_x = s_mem;                                   // TH * TWH * cs floats, then [one = 1.0][zero = 0.0]
  const int n_x = TH * TWH * cs;
  float* s_g = s_mem + ((n_x + 2 + 3) & ~3);            // CS_NT pixels x 16 channels (zero beyond cout)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, mi = lane & 15, kq = lane >> 4;
  const int m_real = KS * KS * cin;
  // per lane and M-tile: where row m = mt*16 + mi of the patch sits relative to the pixel's LDS base (or an absolute slot: one / zero)
  int off[MT], rel[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mt * 16 + mi;
    if (m < m_real) {
      const int tap = m / cin, ci = m - tap * cin;
      off[mt] = ((tap / KS) * TWH + (tap % KS)) * cs + ci;
      rel[mt] = 1;
    } else {
      off[mt] = m == m_real ? n_x : n_x + 1;
      rel[mt] = 0;
    }
  }
  cs_f4 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = cs_f4{0.f, 0.f, 0.f, 0.f};
  const int ntiles = B * tiles_y * tiles_x;
  // (round 5: fetching the next tile into registers in front of the matrix loop -- 46 more registers, 140-160 in all -- made the kernel
  // SLOWER: 111 / 96 / 93 us against 90 / 80 / 75; it lives on its occupancy, the other resident workgroups cover the staging)
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int b = tile / (tiles_y * tiles_x), trem = tile - b * tiles_y * tiles_x;
    const int h0 = (trem / tiles_x) * CS_TH, w0 = (trem % tiles_x) * CS_TW;
    __syncthreads();                                    // the previous tile's readers are done
    stage_tile<R>(x, b, H, W, cin, cs, h0, w0, s_x);
    if (threadIdx.x == 0) { s_x[n_x] = 1.f; s_x[n_x + 1] = 0.f; }
    {
      // g tile: pixel p = threadIdx.x -> (py, px); 16 slots per pixel
      const int py = threadIdx.x / CS_TW, px = threadIdx.x % CS_TW;
      const int h = h0 + py, w = w0 + px;
      const bool in = h < H && w < W;
      const float* gp = g + ((long long)(b * H + h) * W + w) * cout;
#pragma unroll
      for (int n = 0; n < 16; ++n) s_g[threadIdx.x * 16 + n] = (in && n < cout) ? gp[n] : 0.f;
    }
    __syncthreads();
#pragma unroll 2
    for (int ks4 = 0; ks4 < 16; ++ks4) {                // the wave's 64 pixels, four per matrix instruction
      const int p = wave * 64 + ks4 * 4 + kq;
      const int base = ((p / CS_TW) * TWH + (p % CS_TW)) * cs;
      const float bv = s_g[p * 16 + mi];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float av = s_x[rel[mt] * base + off[mt]];
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[mt], 0, 0, 0);
      }
    }
  }
  // the four waves' accumulators meet in LDS; one partial (MT*16 x 16) per workgroup
  __syncthreads();
  float* s_acc = s_mem;                                 // 4 x MT*256 floats (fits: see conv_small_wgrad_lds)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s_acc[(wave * MT + mt) * 256 + (4 * kq + r) * 16 + mi] = acc[mt][r];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < MT * 256; i += CS_NT) {
    const float v = (s_acc[i] + s_acc[MT * 256 + i]) + (s_acc[2 * MT * 256 + i] + s_acc[3 * MT * 256 + i]);
    part[(long long)blockIdx.x * MT * 256 + i] = v;
  }
}

// dW and the bias gradient from the per-workgroup partials, in a fixed order, two levels deep (one thread adding all ~1000 partials of
// its element runs at one memory latency per four loads: 100 us for 7 MB): slice (mt, sl) adds every CS_FOLD-th partial starting at sl,
// the finishing pass adds the CS_FOLD slices and scatters into the weight's layout.
constexpr int CS_FOLD = 32;
__global__ __launch_bounds__(CS_NT) void conv_small_wgrad_fold1_kernel(const float* __restrict__ part, int nparts, int MT, float* __restrict__ slices) {
  const int mt = blockIdx.x, sl = blockIdx.y, e = threadIdx.x;
  const long long stride = (long long)MT * 256;
  const float* p = part + (long long)mt * 256 + e;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = sl;
  for (; i + 3 * CS_FOLD < nparts; i += 4 * CS_FOLD) {
    a0 += p[(long long)i * stride]; a1 += p[(long long)(i + CS_FOLD) * stride];
    a2 += p[(long long)(i + 2 * CS_FOLD) * stride]; a3 += p[(long long)(i + 3 * CS_FOLD) * stride];
  }
  for (; i < nparts; i += CS_FOLD) a0 += p[(long long)i * stride];
  slices[((long long)sl * MT + mt) * 256 + e] = (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(CS_NT) void conv_small_wgrad_fold2_kernel(const float* __restrict__ slices, int MT, int cin, int cout, int ks,
                                                                       float* __restrict__ gw, float* __restrict__ gb) {
  const int mt = blockIdx.x, e = threadIdx.x;            // element e = row * 16 + col of M-tile mt
  const float* p = slices + (long long)mt * 256 + e;
  const long long stride = (long long)MT * 256;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < CS_FOLD; ++i) a[i & 3] += p[i * stride];
  const float v = (a[0] + a[1]) + (a[2] + a[3]);
  const int m = mt * 16 + e / 16, co = e % 16, m_real = ks * ks * cin;
  if (co >= cout) return;
  if (m < m_real) {
    const int tap = m / cin, ci = m - tap * cin;
    gw[((long long)co * ks * ks + tap) * cin + ci] = v;                 // (cout, kh, kw, cin): a channels-last weight's memory order
  } else if (m == m_real && gb) {
    gb[co] = v;
  }
}

static inline size_t conv_small_lds_direct(int ks, int ci) {
  const int R = ks / 2;
  return (size_t)(CS_TH + 2 * R) * (CS_TW + 2 * R) * (ci | 1) * sizeof(float);
}
static inline int conv_small_mt(int ks, int cin) { return (ks * ks * cin + 1 + 15) / 16; }
static inline size_t conv_small_lds_wgrad(int ks, int cin) {
  const int R = ks / 2, MT = conv_small_mt(ks, cin);
  const size_t n_x = (size_t)(CS_TH + 2 * R) * (CS_TW + 2 * R) * (cin | 1);
  const size_t stage = ((n_x + 2 + 3) & ~(size_t)3) + (size_t)CS_NT * 16;
  const size_t fold = (size_t)4 * MT * 256;
  return (stage > fold ? stage : fold) * sizeof(float);
}

// the (kernel size, in-channels, out-channels) the direct kernel is instantiated for: the motion decoders' finest level with three or
// four input planes per image (9 / 12 input channels), both decoders, forward and (roles swapped) data gradient
#define DD_CS_PAIRS(X) \
  X(3, 12, 9) X(3, 10, 9) X(3, 9, 9) X(3, 9, 12) X(3, 9, 10) X(3, 16, 12) X(3, 13, 12) X(3, 12, 12) X(3, 12, 16) X(3, 12, 13) \
  X(1, 9, 3) X(1, 9, 1) X(1, 3, 9) X(1, 1, 9) X(1, 12, 3) X(1, 12, 1) X(1, 3, 12) X(1, 1, 12)

static int direct_supported(int ks, int ci, int co) {
#define X(K, I, O) if (ks == K && ci == I && co == O) return 1;
  DD_CS_PAIRS(X)
#undef X
  return 0;
}

static hipError_t launch_direct(int ks, int ci, int co, const float* x, const float* wp, const float* bias, int B, int H, int W, float* y, hipStream_t st) {
  const dim3 grid((W + CS_TW - 1) / CS_TW, (H + CS_TH - 1) / CS_TH, B);
  const size_t lds = conv_small_lds_direct(ks, ci);
#define X(K, I, O) \
  if (ks == K && ci == I && co == O) { hipLaunchKernelGGL((conv_small_kernel<K, I, O>), grid, dim3(CS_NT), lds, st, x, wp, bias, H, W, y); return hipGetLastError(); }
  DD_CS_PAIRS(X)
#undef X
  return hipErrorInvalidValue;
}

}  // namespace dd

extern "C" int dd_conv_small_supported(int ks, int cin, int cout) {
  if (!(ks == 1 || ks == 3) || cin < 1 || cout < 1 || cin > dd::CS_MAXC || cout > dd::CS_MAXC) return 0;
  const int mt = dd::conv_small_mt(ks, cin);
  if (!(mt == 1 || mt == 6 || mt == 7 || mt == 8 || mt == 10)) return 0;
  return dd::direct_supported(ks, cin, cout) && dd::direct_supported(ks, cout, cin);
}

extern "C" size_t dd_conv_small_workspace_bytes(int ks, int cin, int cout) {
  const size_t wp = (size_t)ks * ks * cin * cout * sizeof(float);
  const size_t parts = (size_t)(dd::CS_WG_BLOCKS + dd::CS_FOLD) * dd::conv_small_mt(ks, cin) * 256 * sizeof(float);
  return ((wp + 255) & ~(size_t)255) + parts;
}

static int conv_small_direct(const float* x, const float* weight, long long s_co, long long s_ci, long long s_kh, long long s_kw, const float* bias, int B, int H,
                             int W, int cin, int cout, int ks, int transpose, float* y, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !weight || !y || !ws || B < 1 || H < 1 || W < 1) return (int)hipErrorInvalidValue;
  if (!dd_conv_small_supported(ks, cin, cout) || ws_bytes < dd_conv_small_workspace_bytes(ks, cin, cout)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  float* wp = reinterpret_cast<float*>(ws);
  const int n = ks * ks * cin * cout;
  hipLaunchKernelGGL(dd::conv_small_prep_kernel, dim3((n + dd::CS_NT - 1) / dd::CS_NT), dim3(dd::CS_NT), 0, st, weight, s_co, s_ci, s_kh, s_kw, cout, cin, ks,
                     transpose, wp);
  // the direct kernel's (in, out) channel counts: swapped for the data gradient
  const hipError_t e = transpose ? dd::launch_direct(ks, cout, cin, x, wp, nullptr, B, H, W, y, st) : dd::launch_direct(ks, cin, cout, x, wp, bias, B, H, W, y, st);
  return (int)e;
}

extern "C" int dd_conv_small_fwd(const float* x, const float* weight, long long s_co, long long s_ci, long long s_kh, long long s_kw, const float* bias, int B,
                                 int H, int W, int cin, int cout, int ks, float* y, void* workspace, size_t workspace_bytes, void* stream) {
  return conv_small_direct(x, weight, s_co, s_ci, s_kh, s_kw, bias, B, H, W, cin, cout, ks, 0, y, workspace, workspace_bytes, stream);
}

extern "C" int dd_conv_small_bwd_data(const float* g_out, const float* weight, long long s_co, long long s_ci, long long s_kh, long long s_kw, int B, int H, int W,
                                      int cin, int cout, int ks, float* g_x, void* workspace, size_t workspace_bytes, void* stream) {
  return conv_small_direct(g_out, weight, s_co, s_ci, s_kh, s_kw, nullptr, B, H, W, cin, cout, ks, 1, g_x, workspace, workspace_bytes, stream);
}

extern "C" int dd_conv_small_bwd_weight(const float* x, const float* g_out, int B, int H, int W, int cin, int cout, int ks, float* g_weight, float* g_bias,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !g_out || !g_weight || !workspace || B < 1 || H < 1 || W < 1) return (int)hipErrorInvalidValue;
  if (!dd_conv_small_supported(ks, cin, cout) || workspace_bytes < dd_conv_small_workspace_bytes(ks, cin, cout)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const size_t wp = ((size_t)ks * ks * cin * cout * sizeof(float) + 255) & ~(size_t)255;
  float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + wp);
  const int tiles_x = (W + dd::CS_TW - 1) / dd::CS_TW, tiles_y = (H + dd::CS_TH - 1) / dd::CS_TH;
  const int ntiles = B * tiles_x * tiles_y;
  const int blocks = ntiles < dd::CS_WG_BLOCKS ? ntiles : dd::CS_WG_BLOCKS;
  const int MT = dd::conv_small_mt(ks, cin);
  const size_t lds = dd::conv_small_lds_wgrad(ks, cin);
#define DD_CS_WG(K, M) \
  if (ks == K && MT == M) hipLaunchKernelGGL((dd::conv_small_wgrad_kernel<K, M>), dim3(blocks), dim3(dd::CS_NT), lds, st, x, g_out, B, H, W, cin, cout, tiles_x, tiles_y, part);
  DD_CS_WG(1, 1) DD_CS_WG(3, 6) DD_CS_WG(3, 7) DD_CS_WG(3, 8) DD_CS_WG(3, 10)
#undef DD_CS_WG
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  float* slices = part + (size_t)dd::CS_WG_BLOCKS * MT * 256;
  hipLaunchKernelGGL(dd::conv_small_wgrad_fold1_kernel, dim3(MT, dd::CS_FOLD), dim3(dd::CS_NT), 0, st, part, blocks, MT, slices);
  hipLaunchKernelGGL(dd::conv_small_wgrad_fold2_kernel, dim3(MT), dim3(dd::CS_NT), 0, st, slices, MT, cin, cout, ks, g_weight, g_bias);
  return (int)hipGetLastError();
}
