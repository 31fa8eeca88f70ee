// dd_pw_gemm.hip -- LiteMono's point-wise Linears (reference networks/depth_encoder.py:200-203,216-224,262-272: pwconv1 -> GELU -> pwconv2
// on a channels-last (B,H,W,C) tensor, C = 64 / 128 / 224, hidden 6C) at fp32 accuracy on the bf16 matrix pipe (gfx950).
//
// The BLAS back-end runs these GEMMs on the fp32 MFMA forms and they are short in one dimension (K = C or N = C): 70-100 us each for
// 140 MB of hidden activations, with an element-wise exact-GELU pass (read + write of the hidden tensor) between them.  Here:
//   * y[M,N] = act_in(x[M,K]) . W[N,K]^T + bias with every fp32 operand split exactly into three bf16 pieces and six partial products
//     on v_mfma_f32_32x32x16_bf16 (dd_split.h; the arithmetic of dd_conv_mfma.hip, pinned by tests/test_split_bf16.py);
//   * act_in = exact GELU applied to the A operand while it is split ("GELU prologue"): the second Linear reads the first one's
//     pre-activation, the activated tensor never exists in memory;
//   * no LDS and no barrier: the A fragment of a lane is 8 consecutive floats of ONE row (32 bytes, straight from global memory, next
//     k step prefetched), the B fragments come in fragment order from a pack (dd_mlp_pack, once per step, L2-resident); a wave owns
//     32*RB rows and walks over its share of the 32-column blocks, NG at a time.
// Bound: HBM for the wide side (the 6C-wide tensor is written or read once, 4 bytes per element); the GELU prologue costs ~40 VALU
// instructions per element (erff) -- about the time of the tensor's own HBM read.
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_split.h"

namespace dd {
namespace pw {

using cm::bf8;
using cm::f16v;
using cm::split2;

constexpr int FRAG_U4 = 64;          // uint4 per fragment (64 lanes x 16 bytes)

__device__ __forceinline__ float gelu_exact(float v) {        // ATen's GeluCUDAKernelImpl (approximate = none), same device erff
  return (v * 0.5f) * (1.f + erff(v * 0.70710678118654752440f));
}

// pack layout: [n block][k step][piece][lane] x 16 bytes.  lane l of a fragment holds, for output column nb * 32 + (l & 31), the
// inputs ks * 16 + (l >> 5) * 8 + 0..7.  The weight is addressed through two element strides: (s_n, s_k) = (K, 1) packs W (N,K) for
// x . W^T, (1, N') packs the transpose of a (N',N) matrix for the data gradient.
struct PackRegion {
  const float* w;
  long long s_n, s_k;
  int N, K, first;                   // first fragment (n block, k step) of the region inside the launch
  uint4* out;
};
struct PackArgs {
  PackRegion r[4];
  int count, total;
};

__global__ __launch_bounds__(256) void pw_pack_kernel(const PackArgs a) {
  const int gid = blockIdx.x * 256 + threadIdx.x, lane = gid & 63;
  int f = gid >> 6;
  if (f >= a.total) return;
  int ri = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (i < a.count && f >= a.r[i].first) ri = i;
  PackRegion rg = a.r[0];
#pragma unroll
  for (int i = 1; i < 4; ++i)
    if (ri == i) rg = a.r[i];
  f -= rg.first;
  const int KS = rg.K >> 4, nb = f / KS, ks = f - nb * KS;
  const int n = nb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = n < rg.N ? rg.w[(long long)n * rg.s_n + (long long)(k0 + e) * rg.s_k] : 0.f;
  unsigned p[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split2(v[2 * q], v[2 * q + 1], p[0][q], p[1][q], p[2][q]);
  uint4* dst = rg.out + ((size_t)f * 3) * FRAG_U4 + lane;
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) dst[pc * FRAG_U4] = make_uint4(p[pc][0], p[pc][1], p[pc][2], p[pc][3]);
}

template <int RB, int NG, bool GELU_IN>
__global__ __launch_bounds__(256, 2) void pw_gemm_kernel(const float* __restrict__ x, const uint4* __restrict__ pack, const float* __restrict__ bias,
                                                         int M, int K, int N, float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r0 = ((int)blockIdx.x * 4 + wave) * (32 * RB);
  if (r0 >= M) return;               // wave-uniform; the kernel has no barrier
  const int KS = K >> 4, NBLK = (N + 31) >> 5, ngroups = (NBLK + NG - 1) / NG;
  const float* ap[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int row = min(r0 + rb * 32 + (lane & 31), M - 1);      // rows beyond M repeat the last one (never stored)
    ap[rb] = x + (size_t)row * K + (lane >> 5) * 8;
  }
  for (int grp = blockIdx.y; grp < ngroups; grp += gridDim.y) {
    const uint4* bp[NG];
#pragma unroll
    for (int n = 0; n < NG; ++n) bp[n] = pack + (size_t)min(grp * NG + n, NBLK - 1) * KS * (3 * FRAG_U4) + lane;
    f16v acc[RB][NG];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int n = 0; n < NG; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][n][r] = 0.f;
    float4 ar[RB][2];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      ar[rb][0] = *reinterpret_cast<const float4*>(ap[rb]);
      ar[rb][1] = *reinterpret_cast<const float4*>(ap[rb] + 4);
    }
    for (int ks = 0; ks < KS; ++ks) {
      uint4 bf[NG][3];
#pragma unroll
      for (int n = 0; n < NG; ++n)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) bf[n][pc] = bp[n][(ks * 3 + pc) * FRAG_U4];
      uint4 af[RB][3];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        float v[8] = {ar[rb][0].x, ar[rb][0].y, ar[rb][0].z, ar[rb][0].w, ar[rb][1].x, ar[rb][1].y, ar[rb][1].z, ar[rb][1].w};
        if (GELU_IN) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = gelu_exact(v[e]);
        }
        unsigned p[3][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split2(v[2 * q], v[2 * q + 1], p[0][q], p[1][q], p[2][q]);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) af[rb][pc] = make_uint4(p[pc][0], p[pc][1], p[pc][2], p[pc][3]);
      }
      {
        const int kn = min(ks + 1, KS - 1);        // the last step re-reads itself: the loop body stays branch-free
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          ar[rb][0] = *reinterpret_cast<const float4*>(ap[rb] + kn * 16);
          ar[rb][1] = *reinterpret_cast<const float4*>(ap[rb] + kn * 16 + 4);
        }
      }
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int n = 0; n < NG; ++n)
            acc[rb][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, af[rb][cm::kPieceA[t]]), __builtin_bit_cast(bf8, bf[n][cm::kPieceB[t]]),
                                                                 acc[rb][n], 0, 0, 0);
    }
    // C layout of 32x32: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int n = 0; n < NG; ++n) {
      const int co = (grp * NG + n) * 32 + (lane & 31);
      if (grp * NG + n >= NBLK || co >= N) continue;
      const float bv = bias ? bias[co] : 0.f;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < M) y[(size_t)row * N + co] = acc[rb][n][r] + bv;
        }
      }
    }
  }
}

template <int RB, int NG, bool GELU_IN>
static int launch(const float* x, const void* pack, const float* bias, int M, int K, int N, float* y, hipStream_t stream) {
  const int tiles = (M + 32 * RB - 1) / (32 * RB), ngroups = ((N + 31) / 32 + NG - 1) / NG;
  int split = (4096 + tiles - 1) / tiles;              // enough waves for 256 CUs x 4 SIMDs x a few
  split = split < 1 ? 1 : (split > ngroups ? ngroups : split);
  hipLaunchKernelGGL((pw_gemm_kernel<RB, NG, GELU_IN>), dim3((tiles + 3) / 4, split), dim3(256), 0, stream, x, static_cast<const uint4*>(pack), bias, M, K, N, y);
  return (int)hipGetLastError();
}

}  // namespace pw
}  // namespace dd

extern "C" size_t dd_pw_gemm_pack_bytes(int N, int K) { return (size_t)((N + 31) / 32) * (K / 16) * 3 * 1024; }

extern "C" int dd_mlp_pack(const float* w1, long long s1_n, long long s1_k, const float* w2, long long s2_n, long long s2_k, int C, int hidden, void* pack_fwd1,
                           void* pack_fwd2, void* pack_bwd2, void* pack_bwd1, void* stream) {
  using namespace dd::pw;
  if (!w1 || !w2 || C < 16 || hidden < 16 || C % 16 || hidden % 16) return (int)hipErrorInvalidValue;
  PackArgs a;
  a.count = 0;
  a.total = 0;
  auto add = [&](const float* w, long long s_n, long long s_k, int N, int K, void* out) {
    if (!out) return;
    PackRegion& r = a.r[a.count++];
    r.w = w; r.s_n = s_n; r.s_k = s_k; r.N = N; r.K = K; r.first = a.total; r.out = static_cast<uint4*>(out);
    a.total += ((N + 31) / 32) * (K / 16);
  };
  add(w1, s1_n, s1_k, hidden, C, pack_fwd1);           // pre  = y . W1^T          W1 (hidden, C)
  add(w2, s2_n, s2_k, C, hidden, pack_fwd2);           // out  = act(pre) . W2^T   W2 (C, hidden)
  add(w2, s2_k, s2_n, hidden, C, pack_bwd2);           // g_post = g . W2          "weight" (n = hidden, k = C) = W2^T
  add(w1, s1_k, s1_n, C, hidden, pack_bwd1);           // g_y  = g_pre . W1        "weight" (n = C, k = hidden) = W1^T
  if (a.count == 0) return (int)hipErrorInvalidValue;
  for (int i = a.count; i < 4; ++i) a.r[i] = a.r[0];
  hipLaunchKernelGGL(pw_pack_kernel, dim3((a.total * 64 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

extern "C" int dd_pw_gemm(const float* x, const void* pack, const float* bias, int M, int K, int N, int gelu_in, float* y, void* stream) {
  using namespace dd::pw;
  if (!x || !pack || !y || M < 1 || K < 16 || K % 16 || N < 1) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<unsigned long long>(x) & 15ull) || (size_t)M * K >= (1ull << 40) || (size_t)M * N >= (1ull << 40)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool narrow = (N + 31) / 32 <= 2;              // one or two column blocks: two row blocks per wave instead of four column blocks
  if (gelu_in) return narrow ? launch<2, 2, true>(x, pack, bias, M, K, N, y, s) : launch<1, 4, true>(x, pack, bias, M, K, N, y, s);
  return narrow ? launch<2, 2, false>(x, pack, bias, M, K, N, y, s) : launch<2, 4, false>(x, pack, bias, M, K, N, y, s);
}
