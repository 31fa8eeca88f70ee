// dd_pw_gemm.hip -- LiteMono's point-wise Linears (reference networks/depth_encoder.py:200-203,216-224,262-272: pwconv1 -> GELU -> pwconv2
// on a channels-last (B,H,W,C) tensor, C = 64 / 128 / 224, hidden 6C) at fp32 accuracy on the bf16 matrix pipe (gfx950).
//
// The BLAS back-end runs these GEMMs on the fp32 MFMA forms and they are short in one dimension (K = C or N = C): 70-100 us each for
// 140 MB of hidden activations, with an element-wise exact-GELU pass (read + write of the hidden tensor) between them.  Here:
//   * y[M,N] = act_in(x[M,K]) . W[N,K]^T + bias with every fp32 operand split exactly into three bf16 pieces and six partial products
//     on v_mfma_f32_32x32x16_bf16 (dd_split.h; the arithmetic of dd_conv_mfma.hip, pinned by tests/test_split_bf16.py);
//   * act_in = the erf GELU applied to the A operand while it is split ("GELU prologue"): the second Linear reads the first one's
//     pre-activation, the activated tensor never exists in memory.  erf is a branch-free two-range polynomial (8e-8 absolute error,
//     fitted and checked against scipy on the CPU: ~25 VALU instructions where the device library's erff takes ~55) -- with the
//     library's the prologue, not HBM, bounded the kernel;
//   * a workgroup (NW waves) owns 32*RB*NW rows and NG 32-column blocks; K goes in chunks of 32: the chunk of A is fetched with
//     coalesced 16-byte loads (eight lanes per row; prefetched one chunk ahead), activated, split ONCE and staged in LDS as three bf16
//     planes with an 80-byte row pitch (conflict-free 16-byte fragment reads); the B fragments come in fragment order from a pack
//     (dd_mlp_pack, once per step) through LDS, each wave fetching its share.  (The first version read every lane's A fragment straight
//     from global memory -- 32 cache lines per load instruction -- and every wave its own B fragments: correct, and bound by the
//     texture addresser at 2-3x the time of this one.)
// Bound: HBM for the wide side (the 6C-wide tensor is written or read once, 4 bytes per element).
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_attr.h"
#include "dd_split.h"

namespace dd {
namespace pw {

using cm::bf8;
using cm::f16v;
using cm::split2;

constexpr int FRAG_U4 = 64;          // uint4 per fragment (64 lanes x 16 bytes)

// pack layout: [n block][k step][piece][lane] x 16 bytes.  lane l of a fragment holds, for output column nb * 32 + (l & 31), the
// inputs ks * 16 + (l >> 5) * 8 + 0..7.  The weight is addressed through two element strides: (s_n, s_k) = (K, 1) packs W (N,K) for
// x . W^T, (1, N') packs the transpose of a (N',N) matrix for the data gradient.
struct PackRegion {
  const float* w;
  long long s_n, s_k;
  int N, K, first;                   // first fragment (n block, k step) of the region inside the launch
  int fused;                         // 1: the second Linear's operand for mlp_fwd_kernel -- [k block of 32][n block][step][piece][lane],
                                     // the lane's eight k in the order in which the first GEMM's accumulator registers hold them
  uint4* out;
};
constexpr int MAX_REGIONS = 5;
struct PackArgs {
  PackRegion r[MAX_REGIONS];
  int count, total;
};

// The C layout of a 32x32 accumulator puts row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) in register r.  mlp_fwd_kernel computes the hidden
// tile TRANSPOSED (hidden index = accumulator row, pixel = column = lane & 31) and feeds registers 8 j .. 8 j + 7 of a lane as the
// eight k values of step j of the second GEMM without moving them: element e of step j of lane-half hh is hidden index
__host__ __device__ inline int fused_k(int j, int hh, int e) { return (e & 3) + 8 * (2 * j + (e >> 2)) + 4 * hh; }

__global__ __launch_bounds__(256) void pw_pack_kernel(const PackArgs a) {
  const int gid = blockIdx.x * 256 + threadIdx.x, lane = gid & 63;
  int f = gid >> 6;
  if (f >= a.total) return;
  int ri = 0;
#pragma unroll
  for (int i = 1; i < MAX_REGIONS; ++i)
    if (i < a.count && f >= a.r[i].first) ri = i;
  PackRegion rg = a.r[0];
#pragma unroll
  for (int i = 1; i < MAX_REGIONS; ++i)
    if (ri == i) rg = a.r[i];
  f -= rg.first;
  const int KS = rg.K >> 4;
  int nb, kk[8];
  if (rg.fused) {
    const int NBLK = (rg.N + 31) >> 5, j = f & 1, hb = (f >> 1) / NBLK;
    nb = (f >> 1) - hb * NBLK;
#pragma unroll
    for (int e = 0; e < 8; ++e) kk[e] = hb * 32 + fused_k(j, lane >> 5, e);
  } else {
    nb = f / KS;
    const int ks = f - nb * KS;
#pragma unroll
    for (int e = 0; e < 8; ++e) kk[e] = ks * 16 + (lane >> 5) * 8 + e;
  }
  const int n = nb * 32 + (lane & 31);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = n < rg.N ? rg.w[(long long)n * rg.s_n + (long long)kk[e] * rg.s_k] : 0.f;
  unsigned p[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split2(v[2 * q], v[2 * q + 1], p[0][q], p[1][q], p[2][q]);
  uint4* dst = rg.out + ((size_t)f * 3) * FRAG_U4 + lane;
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) dst[pc * FRAG_U4] = make_uint4(p[pc][0], p[pc][1], p[pc][2], p[pc][3]);
}

// erf(a), branch-free: a (1 + P(a^2)) for |a| <= 0.93, sign(a) (1 - exp(t R(t) - t)) beyond, t = min(|a|, 4.05).  Coefficients:
// least squares on Chebyshev nodes in float64, rounded to fp32; evaluated in fp32 the maximum error against scipy.special.erf over
// [-6, 6] is 8.3e-8 absolute (1.0e-7 relative) and the GELU below is within 4.5e-7 of float64 for |x| <= 6.
__device__ __forceinline__ float erf_poly(float a) {
  const float t = fminf(fabsf(a), 4.05f), s = a * a;
  float r = -0.0005972102517262101f;
  r = fmaf(r, s, 0.004989944398403168f);
  r = fmaf(r, s, -0.02676478400826454f);
  r = fmaf(r, s, 0.11281778663396835f);
  r = fmaf(r, s, -0.37612491846084595f);
  r = fmaf(r, s, 0.12837915122509003f);
  const float small = fmaf(r, a, a);
  float q = 9.955634823199944e-07f;
  q = fmaf(q, t, -3.347820893395692e-05f);
  q = fmaf(q, t, 0.0004920329665765166f);
  q = fmaf(q, t, -0.004276696592569351f);
  q = fmaf(q, t, 0.025077397003769875f);
  q = fmaf(q, t, -0.10777396708726883f);
  q = fmaf(q, t, -0.6342049241065979f);
  q = fmaf(q, t, -0.12888674437999725f);
  q = fmaf(q, t, -t);
  const float large = copysignf(1.f - __expf(q), a);
  return t > 0.93f ? large : small;
}

__device__ __forceinline__ float gelu_erf(float v) { return (v * 0.5f) * (1.f + erf_poly(v * 0.70710678118654752440f)); }

constexpr int KC = 32;               // contraction elements per chunk: two MFMA steps
constexpr int RSTR = 80;             // bytes per row and piece in LDS: 32 bf16 + 16 bytes of padding
constexpr int FRAG = 1024;           // bytes per fragment

template <int NW, int RB, int NG>
constexpr int lds_bytes() { return 3 * (32 * RB * NW) * RSTR + NG * 6 * FRAG; }

template <int NW, int RB, int NG, bool GELU_IN>
__global__ __launch_bounds__(NW * 64, 2) void pw_gemm_kernel(const float* __restrict__ x, const uint4* __restrict__ pack, const float* __restrict__ bias,
                                                             int M, int K, int N, float* __restrict__ y) {
  constexpr int NT = NW * 64, MT = 32 * RB * NW;
  constexpr int A_PIECE = MT * RSTR, A_BYTES = 3 * A_PIECE;
  constexpr int BFR = NG * 6;                    // B fragments per chunk: [column block][step][piece]
  constexpr int BR = (BFR + NW - 1) / NW;        // rounds: wave w moves fragment w + NW r
  constexpr int PA = MT * 8 / NT;                // 16-byte loads of A per thread and chunk (eight per row)
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* const s_b = smem + A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = (int)blockIdx.x * MT, nb0 = (int)blockIdx.y * NG;
  const int KS = K >> 4, NCH = K / KC, NBLK = (N + 31) >> 5;

  const float* a_src[PA];
  int a_lds[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int i = tid + p * NT, r = i >> 3, q = i & 7;
    a_src[p] = x + (size_t)min(row0 + r, M - 1) * K + q * 4;          // rows beyond M repeat the last one (never stored)
    a_lds[p] = r * RSTR + q * 8;
  }
  const uint4* b_src[BR];
#pragma unroll
  for (int r = 0; r < BR; ++r) {
    const int f = wave + NW * r < BFR ? wave + NW * r : 0, n = f / 6, j = f - n * 6;
    b_src[r] = pack + (size_t)min(nb0 + n, NBLK - 1) * KS * 192 + j * 64 + lane;
  }
  float4 areg[PA];
  uint4 breg[BR];
  auto fetch = [&](int c) {
#pragma unroll
    for (int p = 0; p < PA; ++p) areg[p] = *reinterpret_cast<const float4*>(a_src[p] + c * KC);
#pragma unroll
    for (int r = 0; r < BR; ++r) breg[r] = b_src[r][(size_t)c * 384];        // two steps x three pieces x 64 lanes
  };
  auto stage = [&]() {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      float v[4] = {areg[p].x, areg[p].y, areg[p].z, areg[p].w};
      if (GELU_IN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
      }
      unsigned a1, a2, a3, b1, b2, b3;
      split2(v[0], v[1], a1, a2, a3);
      split2(v[2], v[3], b1, b2, b3);
      *reinterpret_cast<uint2*>(smem + a_lds[p]) = make_uint2(a1, b1);
      *reinterpret_cast<uint2*>(smem + A_PIECE + a_lds[p]) = make_uint2(a2, b2);
      *reinterpret_cast<uint2*>(smem + 2 * A_PIECE + a_lds[p]) = make_uint2(a3, b3);
    }
#pragma unroll
    for (int r = 0; r < BR; ++r)
      if (wave + NW * r < BFR) *reinterpret_cast<uint4*>(s_b + (wave + NW * r) * FRAG + lane * 16) = breg[r];
  };
  // LDS-only barrier: this wave's LDS operations have completed, its global prefetch stays in flight
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  f16v acc[RB][NG];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int n = 0; n < NG; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][n][r] = 0.f;

  const unsigned char* a_lane = smem + (wave * 32 * RB + (lane & 31)) * RSTR + (lane >> 5) * 16;
  const unsigned char* b_lane = s_b + lane * 16;
  fetch(0);
  for (int c = 0; c < NCH; ++c) {
    lds_barrier();                         // the previous chunk's fragment reads are done
    stage();
    lds_barrier();
    fetch(min(c + 1, NCH - 1));            // lands under this chunk's MFMAs (the last chunk re-reads itself: no branch)
#pragma unroll
    for (int ksi = 0; ksi < 2; ++ksi) {
      uint4 af[RB][3], bf[NG][3];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) af[rb][pc] = *reinterpret_cast<const uint4*>(a_lane + rb * 32 * RSTR + ksi * 32 + pc * A_PIECE);
#pragma unroll
      for (int n = 0; n < NG; ++n)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) bf[n][pc] = *reinterpret_cast<const uint4*>(b_lane + ((n * 2 + ksi) * 3 + pc) * FRAG);
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int n = 0; n < NG; ++n)
            acc[rb][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, af[rb][cm::kPieceA[t]]), __builtin_bit_cast(bf8, bf[n][cm::kPieceB[t]]),
                                                                 acc[rb][n], 0, 0, 0);
    }
  }
  // C layout of 32x32: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int n = 0; n < NG; ++n) {
    const int co = (nb0 + n) * 32 + (lane & 31);
    if (nb0 + n >= NBLK || co >= N) continue;
    const float bv = bias ? bias[co] : 0.f;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + (wave * RB + rb) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < M) y[(size_t)row * N + co] = acc[rb][n][r] + bv;
      }
    }
  }
}

template <int NW, int RB, int NG, bool GELU_IN>
static int launch(const float* x, const void* pack, const float* bias, int M, int K, int N, float* y, hipStream_t stream) {
  constexpr int MT = 32 * RB * NW;
  auto kern = pw_gemm_kernel<NW, RB, NG, GELU_IN>;
  static dd::LdsAttrOnce lds_attr;          // per instantiation and device (dd_attr.h)
  if (const int rc = lds_attr.ensure(reinterpret_cast<const void*>(kern), (int)((lds_bytes<NW, RB, NG>())))) return rc;
  const int NBLK = (N + 31) / 32;
  constexpr int lds = lds_bytes<NW, RB, NG>();
  hipLaunchKernelGGL(kern, dim3((M + MT - 1) / MT, (NBLK + NG - 1) / NG), dim3(NW * 64), lds, stream, x, static_cast<const uint4*>(pack), bias,
                     M, K, N, y);
  return (int)hipGetLastError();
}

// Which tile a workgroup takes.  Up to two column blocks (N <= 64): 64 rows per wave on two column blocks; more: 32 rows per wave on
// four column blocks (with K > N -- the wide tensor is the operand A -- N <= 128 is then ONE group: A is read, and its GELU evaluated,
// once).  Four waves per workgroup when that still gives ~400 workgroups, two otherwise.
template <bool GELU_IN>
static int dispatch(const float* x, const void* pack, const float* bias, int M, int K, int N, float* y, hipStream_t s) {
  const int NBLK = (N + 31) / 32;
  if (NBLK <= 2) {
    const int groups = 1;
    if ((M + 255) / 256 * groups >= 400) return launch<4, 2, 2, GELU_IN>(x, pack, bias, M, K, N, y, s);
    return launch<2, 2, 2, GELU_IN>(x, pack, bias, M, K, N, y, s);
  }
  const int groups = (NBLK + 3) / 4;
  if ((M + 127) / 128 * groups >= 400) return launch<4, 1, 4, GELU_IN>(x, pack, bias, M, K, N, y, s);
  return launch<2, 1, 4, GELU_IN>(x, pack, bias, M, K, N, y, s);
}

// ---- the whole block forward in one kernel (forward-only passes: the statistics-only side batch, evaluation) -------------------
// out (M,C) = GELU(x . W1^T + b1) . W2^T + b2 without the hidden tensor ever leaving the chip.  A wave owns 32 rows; their x fragments
// are split once and stay in registers.  Per block of 32 hidden units: the first GEMM TRANSPOSED (A operand = W1's fragments, B
// operand = the x fragments: accumulator row = hidden unit, column = pixel), bias + GELU on the sixteen accumulator registers, which
// then ARE the A operand of the second GEMM (pixel = lane & 31 in both layouts; registers 8 j .. 8 j + 7 are the eight k of step j in
// the order fused_k() gives, and dd_mlp_pack lays W2 out in that order) -- no transposition, no LDS round trip for activations.  The
// weight fragments of a hidden block (W1: C/16 x 3, W2: C/32 x 2 x 3 KB) go through LDS, shared by the four waves, prefetched into
// registers one block ahead.  HBM traffic: x once, out once (35 MB at stage 1 against ~600 for two GEMMs and an element-wise pass).
template <int C>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(const float* __restrict__ x, const uint4* __restrict__ pack1, const uint4* __restrict__ pack2,
                                                         const float* __restrict__ b1, const float* __restrict__ b2, int M, float* __restrict__ y) {
  constexpr int KS1 = C / 16, NB2 = C / 32, HID = 6 * C, NHB = HID / 32;
  constexpr int NF1 = KS1 * 3, NF2 = NB2 * 6, NF = NF1 + NF2;      // fragments per hidden block
  constexpr int BR = (NF + 3) / 4;
  extern __shared__ __align__(16) unsigned char smem[];            // [NF fragments] | b1 (HID floats)
  float* const s_b1 = reinterpret_cast<float*>(smem + NF * FRAG);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hh = lane >> 5;
  const int row0 = (int)blockIdx.x * 128 + wave * 32;

  for (int i = tid; i < HID; i += 256) s_b1[i] = b1[i];
  // x fragments of this wave's 32 rows (B operand of the first GEMM): lane = pixel lane & 31, channels ks * 16 + hh * 8 .. + 7
  uint4 xf[KS1][3];
  {
    const float* xp = x + (size_t)min(row0 + (lane & 31), M - 1) * C + hh * 8;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const float4 v0 = *reinterpret_cast<const float4*>(xp + ks * 16), v1 = *reinterpret_cast<const float4*>(xp + ks * 16 + 4);
      unsigned p[3][4];
      split2(v0.x, v0.y, p[0][0], p[1][0], p[2][0]);
      split2(v0.z, v0.w, p[0][1], p[1][1], p[2][1]);
      split2(v1.x, v1.y, p[0][2], p[1][2], p[2][2]);
      split2(v1.z, v1.w, p[0][3], p[1][3], p[2][3]);
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) xf[ks][pc] = make_uint4(p[pc][0], p[pc][1], p[pc][2], p[pc][3]);
    }
  }
  // weight fragments of hidden block hb: round r moves fragment f = wave + 4 r (W1's first, then W2's)
  typedef unsigned u4v __attribute__((ext_vector_type(4)));        // (HIP's uint4 struct kept this array in scratch)
  u4v breg[BR];
  const u4v* w_src[BR];
  int w_step[BR];                                                   // uint4 between two hidden blocks of the round's source
#pragma unroll
  for (int r = 0; r < BR; ++r) {
    const int f = wave + 4 * r < NF ? wave + 4 * r : 0;             // wave-uniform
    w_src[r] = reinterpret_cast<const u4v*>(f < NF1 ? pack1 + f * 64 : pack2 + (f - NF1) * 64) + lane;
    w_step[r] = (f < NF1 ? NF1 : NF2) * 64;
  }
  auto fetch = [&](int hb) {
#pragma unroll
    for (int r = 0; r < BR; ++r) breg[r] = w_src[r][(size_t)hb * w_step[r]];
  };
  auto store_w = [&]() {
#pragma unroll
    for (int r = 0; r < BR; ++r)
      if (wave + 4 * r < NF) *reinterpret_cast<u4v*>(smem + (wave + 4 * r) * FRAG + lane * 16) = breg[r];
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  f16v acc2[NB2];
#pragma unroll
  for (int n = 0; n < NB2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[n][r] = 0.f;
  const unsigned char* w_lane = smem + lane * 16;

  fetch(0);
  for (int hb = 0; hb < NHB; ++hb) {
    lds_barrier();                       // the previous block's fragment reads are done (first pass: b1 is in place after the next one)
    store_w();
    lds_barrier();
    fetch(min(hb + 1, NHB - 1));
    // first GEMM, transposed: h^T (32 hidden x 32 pixels) = W1 block . x^T
    f16v acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      uint4 wf[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) wf[pc] = *reinterpret_cast<const uint4*>(w_lane + (ks * 3 + pc) * FRAG);
#pragma unroll
      for (int t = 0; t < 6; ++t)
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, wf[cm::kPieceA[t]]), __builtin_bit_cast(bf8, xf[ks][cm::kPieceB[t]]), acc1, 0, 0, 0);
    }
    // bias + GELU on the accumulator: register r holds hidden unit hb * 32 + (r & 3) + 8 (r >> 2) + 4 hh of pixel lane & 31
    float hv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bq = *reinterpret_cast<const float4*>(s_b1 + hb * 32 + 8 * q + 4 * hh);
      hv[4 * q + 0] = gelu_erf(acc1[4 * q + 0] + bq.x);
      hv[4 * q + 1] = gelu_erf(acc1[4 * q + 1] + bq.y);
      hv[4 * q + 2] = gelu_erf(acc1[4 * q + 2] + bq.z);
      hv[4 * q + 3] = gelu_erf(acc1[4 * q + 3] + bq.w);
    }
    // second GEMM: out (32 pixels x C) += h (A operand, straight from the registers) . W2 block
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned p[3][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) split2(hv[8 * j + 2 * q], hv[8 * j + 2 * q + 1], p[0][q], p[1][q], p[2][q]);
      uint4 hf[3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) hf[pc] = make_uint4(p[pc][0], p[pc][1], p[pc][2], p[pc][3]);
#pragma unroll
      for (int n = 0; n < NB2; ++n) {
        uint4 wf[3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) wf[pc] = *reinterpret_cast<const uint4*>(w_lane + (NF1 + (n * 2 + j) * 3 + pc) * FRAG);
#pragma unroll
        for (int t = 0; t < 6; ++t)
          acc2[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, hf[cm::kPieceA[t]]), __builtin_bit_cast(bf8, wf[cm::kPieceB[t]]), acc2[n], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NB2; ++n) {
    const int co = n * 32 + (lane & 31);
    const float bv = b2[co];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
      if (row < M) y[(size_t)row * C + co] = acc2[n][r] + bv;
    }
  }
}

template <int C>
static int launch_fused(const float* x, const void* pack1, const void* pack2, const float* b1, const float* b2, int M, float* y, hipStream_t stream) {
  constexpr int lds = ((C / 16) * 3 + (C / 32) * 6) * FRAG + 6 * C * 4;
  auto kern = mlp_fwd_kernel<C>;
  static dd::LdsAttrOnce lds_attr;          // per instantiation and device (dd_attr.h)
  if (const int rc = lds_attr.ensure(reinterpret_cast<const void*>(kern), (int)(lds))) return rc;
  hipLaunchKernelGGL(kern, dim3((M + 127) / 128), dim3(256), lds, stream, x, static_cast<const uint4*>(pack1), static_cast<const uint4*>(pack2), b1, b2, M, y);
  return (int)hipGetLastError();
}

// backward of the activation between the two Linears, one pass: post = GELU(pre) (the second Linear's weight gradient wants the
// activated tensor, which the forward never wrote) and g <- g * GELU'(pre) in place (cdf + x pdf, ATen's GeluBackwardCUDAKernelImpl formula on erf_poly)
__global__ __launch_bounds__(256) void gelu_pair_kernel(const float4* __restrict__ pre, float4* __restrict__ g, float4* __restrict__ post, size_t quads) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= quads) return;
  const float4 x4 = pre[i], g4 = g[i];
  const float xv[4] = {x4.x, x4.y, x4.z, x4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
  float pv[4], dv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = xv[e];
    const float er = 1.f + erf_poly(x * 0.70710678118654752440f);
    const float pdf = __expf(-0.5f * x * x) * 0.3989422804014327f;        // M_2_SQRTPI * M_SQRT1_2 * 0.5
    pv[e] = (x * 0.5f) * er;                                              // the forward's arithmetic, bit for bit
    dv[e] = gv[e] * fmaf(x, pdf, 0.5f * er);
  }
  post[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
  g[i] = make_float4(dv[0], dv[1], dv[2], dv[3]);
}

}  // namespace pw
}  // namespace dd

extern "C" size_t dd_pw_gemm_pack_bytes(int N, int K) { return (size_t)((N + 31) / 32) * (K / 16) * 3 * 1024; }

extern "C" int dd_mlp_pack(const float* w1, long long s1_n, long long s1_k, const float* w2, long long s2_n, long long s2_k, int C, int hidden, void* pack_fwd1,
                           void* pack_fwd2, void* pack_bwd2, void* pack_bwd1, void* pack_fwd2_fused, void* stream) {
  using namespace dd::pw;
  if (!w1 || !w2 || C < 1 || hidden < 1) return (int)hipErrorInvalidValue;
  PackArgs a;
  a.count = 0;
  a.total = 0;
  bool bad = false;
  auto add = [&](const float* w, long long s_n, long long s_k, int N, int K, void* out, int fused = 0) {
    if (!out) return;
    if (K < 16 || K % 16 || (fused && K % 32)) { bad = true; return; }        // the contraction goes in steps of sixteen
    PackRegion& r = a.r[a.count++];
    r.w = w; r.s_n = s_n; r.s_k = s_k; r.N = N; r.K = K; r.first = a.total; r.fused = fused; r.out = static_cast<uint4*>(out);
    a.total += ((N + 31) / 32) * (K / 16);
  };
  add(w1, s1_n, s1_k, hidden, C, pack_fwd1);           // pre  = y . W1^T          W1 (hidden, C)
  add(w2, s2_n, s2_k, C, hidden, pack_fwd2);           // out  = act(pre) . W2^T   W2 (C, hidden)
  add(w2, s2_k, s2_n, hidden, C, pack_bwd2);           // g_post = g . W2          "weight" (n = hidden, k = C) = W2^T
  add(w1, s1_k, s1_n, C, hidden, pack_bwd1);           // g_y  = g_pre . W1        "weight" (n = C, k = hidden) = W1^T
  add(w2, s2_n, s2_k, C, hidden, pack_fwd2_fused, 1);  // dd_mlp_fwd's second operand: W2 by hidden block, k in accumulator-register order
  if (a.count == 0 || bad) return (int)hipErrorInvalidValue;
  for (int i = a.count; i < MAX_REGIONS; ++i) a.r[i] = a.r[0];
  hipLaunchKernelGGL(pw_pack_kernel, dim3((a.total * 64 + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return (int)hipGetLastError();
}

extern "C" int dd_mlp_fwd_supported(int C) { return (C == 64 || C == 128) ? 1 : 0; }

extern "C" int dd_mlp_fwd(const float* x, const void* pack_fwd1, const void* pack_fwd2_fused, const float* b1, const float* b2, int M, int C, float* y, void* stream) {
  if (!x || !pack_fwd1 || !pack_fwd2_fused || !b1 || !b2 || !y || M < 1 || (reinterpret_cast<unsigned long long>(x) & 15ull)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (C) {
    case 64: return dd::pw::launch_fused<64>(x, pack_fwd1, pack_fwd2_fused, b1, b2, M, y, s);
    case 128: return dd::pw::launch_fused<128>(x, pack_fwd1, pack_fwd2_fused, b1, b2, M, y, s);
    default: return (int)hipErrorInvalidValue;
  }
}

extern "C" int dd_gelu_pair(const float* pre, float* g_inout, float* post, size_t n, void* stream) {
  if (!pre || !g_inout || !post || n == 0 || (n & 3)) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<unsigned long long>(pre) | reinterpret_cast<unsigned long long>(g_inout) | reinterpret_cast<unsigned long long>(post)) & 15ull)
    return (int)hipErrorInvalidValue;
  const size_t quads = n >> 2;
  hipLaunchKernelGGL(dd::pw::gelu_pair_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(pre), reinterpret_cast<float4*>(g_inout), reinterpret_cast<float4*>(post), quads);
  return (int)hipGetLastError();
}

extern "C" int dd_pw_gemm(const float* x, const void* pack, const float* bias, int M, int K, int N, int gelu_in, float* y, void* stream) {
  using namespace dd::pw;
  if (!x || !pack || !y || M < 1 || K < 32 || K % 32 || N < 1) return (int)hipErrorInvalidValue;       // chunks of 32
  if ((reinterpret_cast<unsigned long long>(x) & 15ull) || (size_t)M * K >= (1ull << 40) || (size_t)M * N >= (1ull << 40)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return gelu_in ? dd::pw::dispatch<true>(x, pack, bias, M, K, N, y, s) : dd::pw::dispatch<false>(x, pack, bias, M, K, N, y, s);
}
