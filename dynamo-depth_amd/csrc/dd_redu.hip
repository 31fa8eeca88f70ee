// dd_redu.hip -- the 1x1 reductions of the motion decoders (reference networks/motion_decoder.py:33,66: `refine_motion_redu{level}` =
// Conv2d(2*ch, out_dim, 1) applied to cat(a, b) of the level's two 3x3 convolutions; out_dim 3 for the flow decoder, 1 for the mask
// decoder; ch = 512, 256, 128, 64, 64 at 6x20 ... 96x320).  A reduction to 1 / 3 channels is a few dot products per pixel -- bytes, not
// flops -- and the library spends a padded implicit-GEMM launch on each half and each gradient (the halves are convolved where they lie
// instead of concatenated: motion_decoder.redu_split): per level and decoder 2 forward launches + an add, 4 gradient launches behind
// zero-fills, 2 bias sums, ~120 launches and ~1.2 ms per step for ~0.3 ms of bytes.  Here the reduction is ONE operator:
//   forward          y[p, co] = bias[co] + sum_c a[p, c] Wa[co, c] + sum_c b[p, c] Wb[co, c]              one launch
//   data gradients   ga[p, c] = sum_co g[p, co] Wa[co, c],  gb likewise                                    one launch
//   weight gradient  gW[co, c] = sum_p g[p, co] a[p, c] (and b), g_bias[co] = sum_p g[p, co]               one pass + a fixed-order fold
// a, b: [P, C] (channels-last tensors as matrices), weight: [cout, 2C] rows (the memory of a (cout, 2C, 1, 1) tensor in either layout).
// LPP = min(C/4, 64) lanes per pixel, each lane owns C/(4 LPP) channel quads (16-byte accesses, a pixel's channels contiguous);
// weights staged in LDS; sums over a pixel's lanes by shuffles, over pixels in registers -> LDS -> one partial per workgroup.
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"

namespace dd {

constexpr int RD_NT = 256;
constexpr int RD_MAX_BLOCKS = 1024;
constexpr int RD_FOLD = 32;

__device__ __forceinline__ float rd_dot4(const float4 a, const float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }

// s_w: [cout][2 * C4] float4
template <int C4>
__device__ __forceinline__ void rd_stage_weights(const float* __restrict__ w, int cout, float4* s_w) {
  for (int i = threadIdx.x; i < cout * 2 * C4; i += RD_NT) s_w[i] = reinterpret_cast<const float4*>(w)[i];
  __syncthreads();
}

template <int COUT, int LPP, int QUADS>
__global__ __launch_bounds__(RD_NT) void redu_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ w,
                                                         const float* __restrict__ bias, long long P, float* __restrict__ y) {
  constexpr int C4 = LPP * QUADS, PX = RD_NT / LPP;
  __shared__ float4 s_w[COUT * 2 * C4];
  rd_stage_weights<C4>(w, COUT, s_w);
  const int pl = threadIdx.x / LPP, q = threadIdx.x % LPP;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float bv[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) bv[co] = bias ? bias[co] : 0.f;
  const long long groups = (P + PX - 1) / PX;
  for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const long long p = grp * PX + pl;
    const long long pc = p < P ? p : P - 1;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
#pragma unroll
    for (int j = 0; j < QUADS; ++j) {
      const int c4 = q + LPP * j;
      const float4 av = a4[pc * C4 + c4], bb = b4[pc * C4 + c4];
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[co] += rd_dot4(av, s_w[co * 2 * C4 + c4]) + rd_dot4(bb, s_w[co * 2 * C4 + C4 + c4]);
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
#pragma unroll
      for (int m = 1; m < LPP; m <<= 1) acc[co] += __shfl_xor(acc[co], m, 64);
    }
    if (q == 0 && p < P) {
#pragma unroll
      for (int co = 0; co < COUT; ++co) y[p * COUT + co] = acc[co] + bv[co];
    }
  }
}

template <int COUT, int LPP, int QUADS>
__global__ __launch_bounds__(RD_NT) void redu_bwd_data_kernel(const float* __restrict__ g, const float* __restrict__ w, long long P, float* __restrict__ ga,
                                                              float* __restrict__ gb) {
  constexpr int C4 = LPP * QUADS, PX = RD_NT / LPP;
  __shared__ float4 s_w[COUT * 2 * C4];
  rd_stage_weights<C4>(w, COUT, s_w);
  const int pl = threadIdx.x / LPP, q = threadIdx.x % LPP;
  const long long groups = (P + PX - 1) / PX;
  for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const long long p = grp * PX + pl;
    if (p >= P) continue;
    float gv[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) gv[co] = g[p * COUT + co];
#pragma unroll
    for (int j = 0; j < QUADS; ++j) {
      const int c4 = q + LPP * j;
      float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        const float4 wa = s_w[co * 2 * C4 + c4], wb = s_w[co * 2 * C4 + C4 + c4];
        ra.x = fmaf(gv[co], wa.x, ra.x); ra.y = fmaf(gv[co], wa.y, ra.y); ra.z = fmaf(gv[co], wa.z, ra.z); ra.w = fmaf(gv[co], wa.w, ra.w);
        rb.x = fmaf(gv[co], wb.x, rb.x); rb.y = fmaf(gv[co], wb.y, rb.y); rb.z = fmaf(gv[co], wb.z, rb.z); rb.w = fmaf(gv[co], wb.w, rb.w);
      }
      if (ga) reinterpret_cast<float4*>(ga)[p * C4 + c4] = ra;
      if (gb) reinterpret_cast<float4*>(gb)[p * C4 + c4] = rb;
    }
  }
}

// part[block][COUT * 2C + COUT]: gW rows ([co][a channels | b channels]) then the bias gradient
template <int COUT, int LPP, int QUADS>
__global__ __launch_bounds__(RD_NT) void redu_wgrad_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g, long long P,
                                                           float* __restrict__ part) {
  constexpr int C4 = LPP * QUADS, PX = RD_NT / LPP, C = 4 * C4, N = COUT * 2 * C + COUT;
  constexpr int SLOTS = COUT * QUADS * 2 * 4 + COUT, STR = SLOTS | 1;
  extern __shared__ float s_red[];                       // RD_NT * STR floats
  const int pl = threadIdx.x / LPP, q = threadIdx.x % LPP;
  const float4* a4 = reinterpret_cast<const float4*>(a);
  const float4* b4 = reinterpret_cast<const float4*>(b);
  float4 accA[COUT][QUADS], accB[COUT][QUADS];
  float gs[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
    gs[co] = 0.f;
#pragma unroll
    for (int j = 0; j < QUADS; ++j) accA[co][j] = accB[co][j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const long long groups = (P + PX - 1) / PX;
  for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const long long p = grp * PX + pl;
    if (p >= P) continue;
    float gv[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) { gv[co] = g[p * COUT + co]; gs[co] += gv[co]; }
#pragma unroll
    for (int j = 0; j < QUADS; ++j) {
      const int c4 = q + LPP * j;
      const float4 av = a4[p * C4 + c4], bb = b4[p * C4 + c4];
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        accA[co][j].x = fmaf(gv[co], av.x, accA[co][j].x); accA[co][j].y = fmaf(gv[co], av.y, accA[co][j].y);
        accA[co][j].z = fmaf(gv[co], av.z, accA[co][j].z); accA[co][j].w = fmaf(gv[co], av.w, accA[co][j].w);
        accB[co][j].x = fmaf(gv[co], bb.x, accB[co][j].x); accB[co][j].y = fmaf(gv[co], bb.y, accB[co][j].y);
        accB[co][j].z = fmaf(gv[co], bb.z, accB[co][j].z); accB[co][j].w = fmaf(gv[co], bb.w, accB[co][j].w);
      }
    }
  }
  float* mine = s_red + threadIdx.x * STR;
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
#pragma unroll
    for (int j = 0; j < QUADS; ++j) {
      const int t = ((co * QUADS + j) * 2) * 4;
      mine[t] = accA[co][j].x; mine[t + 1] = accA[co][j].y; mine[t + 2] = accA[co][j].z; mine[t + 3] = accA[co][j].w;
      mine[t + 4] = accB[co][j].x; mine[t + 5] = accB[co][j].y; mine[t + 6] = accB[co][j].z; mine[t + 7] = accB[co][j].w;
    }
    mine[COUT * QUADS * 8 + co] = q == 0 ? gs[co] : 0.f;
  }
  __syncthreads();
  // element e < COUT*2C: (co, half, c) -> lane q = (c/4) % LPP, quad j = (c/4) / LPP, slot ((co*QUADS + j)*2 + half)*4 + c%4; summed over the PX pixel lanes
  for (int e = threadIdx.x; e < N; e += RD_NT) {
    float s = 0.f;
    if (e < COUT * 2 * C) {
      const int co = e / (2 * C), r = e - co * 2 * C, half = r / C, c = r - half * C;
      const int cq = c >> 2, qq = cq % LPP, j = cq / LPP, slot = ((co * QUADS + j) * 2 + half) * 4 + (c & 3);
#pragma unroll 4
      for (int pp = 0; pp < PX; ++pp) s += s_red[(pp * LPP + qq) * STR + slot];
    } else {
      const int co = e - COUT * 2 * C;
#pragma unroll 4
      for (int pp = 0; pp < PX; ++pp) s += s_red[(pp * LPP) * STR + COUT * QUADS * 8 + co];
    }
    part[(long long)blockIdx.x * N + e] = s;
  }
}

__global__ __launch_bounds__(RD_NT) void redu_fold1_kernel(const float* __restrict__ part, int nparts, int N, float* __restrict__ slices) {
  const int e = blockIdx.x * RD_NT + threadIdx.x, sl = blockIdx.y;
  if (e >= N) return;
  float a0 = 0.f, a1 = 0.f;
  int i = sl;
  for (; i + RD_FOLD < nparts; i += 2 * RD_FOLD) { a0 += part[(long long)i * N + e]; a1 += part[(long long)(i + RD_FOLD) * N + e]; }
  if (i < nparts) a0 += part[(long long)i * N + e];
  slices[(long long)sl * N + e] = a0 + a1;
}
__global__ __launch_bounds__(RD_NT) void redu_fold2_kernel(const float* __restrict__ slices, int N, int nw, float* __restrict__ gw, float* __restrict__ gbias) {
  const int e = blockIdx.x * RD_NT + threadIdx.x;
  if (e >= N) return;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < RD_FOLD; ++i) a[i & 3] += slices[(long long)i * N + e];
  const float v = (a[0] + a[1]) + (a[2] + a[3]);
  if (e < nw) gw[e] = v;
  else if (gbias) gbias[e - nw] = v;
}

struct ReduShape { int lpp, quads; };
static inline ReduShape redu_shape(int C) {
  ReduShape s = {0, 0};
  if (C == 64) s = {16, 1};
  else if (C == 128) s = {32, 1};
  else if (C == 256) s = {64, 1};
  else if (C == 512) s = {64, 2};
  return s;
}
static inline int redu_blocks(long long P, int lpp) {
  const long long groups = (P + RD_NT / lpp - 1) / (RD_NT / lpp);
  long long nb = (groups + 7) / 8;                        // at least eight pixel groups per workgroup (the weights are staged per workgroup)
  if (nb < 1) nb = 1;
  return (int)(nb > RD_MAX_BLOCKS ? RD_MAX_BLOCKS : nb);
}

}  // namespace dd

extern "C" int dd_redu_supported(int C, int cout) { return dd::redu_shape(C).lpp != 0 && (cout == 1 || cout == 3); }

extern "C" size_t dd_redu_workspace_bytes(long long P, int C, int cout) {
  if (!dd_redu_supported(C, cout) || P < 1) return 0;
  const size_t N = (size_t)cout * 2 * C + cout;
  return ((size_t)dd::redu_blocks(P, dd::redu_shape(C).lpp) + dd::RD_FOLD) * N * sizeof(float);
}

#define DD_REDU_DISPATCH(KERNEL, GRID, LDS, ...)                                                                                             \
  do {                                                                                                                                       \
    const dd::ReduShape sh_ = dd::redu_shape(C);                                                                                             \
    if (cout == 1) {                                                                                                                         \
      if (sh_.lpp == 16) hipLaunchKernelGGL((dd::KERNEL<1, 16, 1>), GRID, dim3(dd::RD_NT), LDS(1, 1), st, __VA_ARGS__);                         \
      else if (sh_.lpp == 32) hipLaunchKernelGGL((dd::KERNEL<1, 32, 1>), GRID, dim3(dd::RD_NT), LDS(1, 1), st, __VA_ARGS__);                    \
      else if (sh_.quads == 1) hipLaunchKernelGGL((dd::KERNEL<1, 64, 1>), GRID, dim3(dd::RD_NT), LDS(1, 1), st, __VA_ARGS__);                   \
      else hipLaunchKernelGGL((dd::KERNEL<1, 64, 2>), GRID, dim3(dd::RD_NT), LDS(1, 2), st, __VA_ARGS__);                                        \
    } else {                                                                                                                                 \
      if (sh_.lpp == 16) hipLaunchKernelGGL((dd::KERNEL<3, 16, 1>), GRID, dim3(dd::RD_NT), LDS(3, 1), st, __VA_ARGS__);                         \
      else if (sh_.lpp == 32) hipLaunchKernelGGL((dd::KERNEL<3, 32, 1>), GRID, dim3(dd::RD_NT), LDS(3, 1), st, __VA_ARGS__);                    \
      else if (sh_.quads == 1) hipLaunchKernelGGL((dd::KERNEL<3, 64, 1>), GRID, dim3(dd::RD_NT), LDS(3, 1), st, __VA_ARGS__);                   \
      else hipLaunchKernelGGL((dd::KERNEL<3, 64, 2>), GRID, dim3(dd::RD_NT), LDS(3, 2), st, __VA_ARGS__);                                        \
    }                                                                                                                                        \
  } while (0)
#define DD_REDU_NO_LDS(co, qd) 0
#define DD_REDU_WG_LDS(co, qd) ((size_t)dd::RD_NT * ((((co) * (qd) * 8 + (co))) | 1) * sizeof(float))

extern "C" int dd_redu_fwd(const float* a, const float* b, const float* weight, const float* bias, long long P, int C, int cout, float* y, void* stream) {
  if (!a || !b || !weight || !y || P < 1 || !dd_redu_supported(C, cout)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(dd::redu_blocks(P, dd::redu_shape(C).lpp));
  DD_REDU_DISPATCH(redu_fwd_kernel, grid, DD_REDU_NO_LDS, a, b, weight, bias, P, y);
  return (int)hipGetLastError();
}

extern "C" int dd_redu_bwd_data(const float* g_out, const float* weight, long long P, int C, int cout, float* g_a, float* g_b, void* stream) {
  if (!g_out || !weight || (!g_a && !g_b) || P < 1 || !dd_redu_supported(C, cout)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(dd::redu_blocks(P, dd::redu_shape(C).lpp));
  DD_REDU_DISPATCH(redu_bwd_data_kernel, grid, DD_REDU_NO_LDS, g_out, weight, P, g_a, g_b);
  return (int)hipGetLastError();
}

extern "C" int dd_redu_bwd_weight(const float* a, const float* b, const float* g_out, long long P, int C, int cout, float* g_weight, float* g_bias,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  if (!a || !b || !g_out || !g_weight || !workspace || P < 1 || !dd_redu_supported(C, cout)) return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_redu_workspace_bytes(P, C, cout)) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = dd::redu_blocks(P, dd::redu_shape(C).lpp);
  const int nw = cout * 2 * C, N = nw + cout;
  float* part = static_cast<float*>(workspace);
  float* slices = part + (size_t)blocks * N;
  const dim3 grid(blocks);
  DD_REDU_DISPATCH(redu_wgrad_kernel, grid, DD_REDU_WG_LDS, a, b, g_out, P, part);
  hipLaunchKernelGGL(dd::redu_fold1_kernel, dim3((N + dd::RD_NT - 1) / dd::RD_NT, dd::RD_FOLD), dim3(dd::RD_NT), 0, st, part, blocks, N, slices);
  hipLaunchKernelGGL(dd::redu_fold2_kernel, dim3((N + dd::RD_NT - 1) / dd::RD_NT), dim3(dd::RD_NT), 0, st, slices, N, nw, g_weight, g_bias);
  return (int)hipGetLastError();
}
