// dd_conv_mfma.hip -- 3x3 stride-1 convolutions of the motion decoders / ResNet blocks at fp32 accuracy on the bf16 matrix pipe (gfx950).
//
// MIOpen runs these (reference networks/motion_decoder.py:24-33,57-66: two 3x3 convolutions on 64-512 channels per level, two
// decoders) on the fp32 MFMA forms, whose peak is 1/16 of the bf16 forms' (157 TFLOP/s against 2.5 PFLOP/s), at 85-110 TFLOP/s.
// Here every fp32 operand is split EXACTLY into three bf16 pieces, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1),
// x3 = x - x1 - x2: 8 + 8 + 8 significand bits, the residuals are exact in fp32), and the product x * w is the six partial
// products x1w1 + x1w2 + x2w1 + x1w3 + x2w2 + x3w1 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Every bf16 x bf16
// product is exact in the accumulator's fp32; the three dropped cross terms are below 2^-26 |x w| -- less than the rounding of an
// fp32 product.  The result has the accuracy of an fp32 FMA chain (tests/test_conv_mfma_gpu.py: against float64, beside
// MIOpen's fp32 result) at 6/16 of the fp32 form's matrix-pipe time.
//
// Implicit GEMM, M = output pixels, N = output channels, K = 9 taps x input channels:
//   * a workgroup (256 threads, 4 waves) owns 8 rows x 32 columns of output pixels of one image and 32*NB output channels;
//     wave w owns tile rows 2w, 2w+1 (two 32-pixel M blocks) and all NB N blocks: 2*NB accumulators of 32x32;
//   * the input halo (10 x 34 pixels) of one 16-channel chunk is split once and staged in LDS as three bf16 planes,
//     [pixel][16 channels + pad] with a 48-byte pixel stride: the A fragment of tap (ty,tx) is one ds_read_b128 per piece at a
//     constant offset, conflict-free (3 x 16 bytes: 16 consecutive pixels fall on 16 distinct 16-byte bank groups);
//   * the weights are split and laid out in fragment order once per step (conv_mfma_pack_kernel): per (chunk, tap) the workgroup
//     fetches 3*NB KB straight into LDS (global_load_lds_dwordx4, double-buffered) while the previous tap's MFMAs run;
//   * the next chunk's halo is fetched into registers under the last taps of the current one.
// LDS: 48 960 B of halo + 2 x 3*NB KB of weights (61 KB at NB = 2): two workgroups per CU, one staging while the other multiplies.
// The data gradient is the same kernel on the output gradient with the weights packed transposed and flipped (pad' = 2 - pad).
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"

namespace dd {
namespace cm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) float fl2;
typedef __attribute__((ext_vector_type(16))) float f16v;

constexpr int TH = 8, TW = 32, NT = 256;
constexpr int HH = TH + 2, HW = TW + 2, HN = HH * HW;      // halo
constexpr int CK = 16;                                      // channels per chunk = the K of one MFMA
constexpr int PSTR = 48;                                    // bytes per halo pixel and piece: 16 bf16 + 16 bytes of padding
constexpr int A_PIECE = HN * PSTR;                          // 16 320
constexpr int A_BYTES = 3 * A_PIECE;                        // 48 960
constexpr int FRAG = 1024;                                  // one B fragment: 64 lanes x 16 bytes
constexpr int PRE = (HN * 4 + NT - 1) / NT;                 // float4 loads per thread and chunk (6)

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: round to nearest even, a in the low half
  return __builtin_bit_cast(unsigned, __builtin_convertvector(fl2{a, b}, bf2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// two fp32 values -> their three bf16 pieces, packed pairwise
__device__ __forceinline__ void split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = pack_bf16(a, b);
  const float ra = a - lo_f(p1), rb = b - hi_f(p1);      // exact
  p2 = pack_bf16(ra, rb);
  p3 = pack_bf16(ra - lo_f(p2), rb - hi_f(p2));           // exact residual, representable in bf16
}

// N blocks of 32 output channels per workgroup: all of them up to 96 channels (the activations are staged and split once), else 64 per workgroup
__host__ __device__ inline int blocks_for(int n_out) { return n_out <= 32 ? 1 : (n_out <= 64 ? 2 : (n_out <= 96 ? 3 : 2)); }

template <int NB>
constexpr int lds_bytes() { return A_BYTES + 2 * 3 * NB * FRAG; }

// pack layout: [n tile][chunk][tap][n block in tile][piece][lane] x 16 bytes.  lane l of a fragment holds, for output channel
// (tile * NB + block) * 32 + (l & 31), the input channels chunk * 16 + (l >> 5) * 8 + 0..7 of the tap.
// transposed = 0: out = cout, in = cin, tap as stored (forward).  transposed = 1: out = cin, in = cout, tap mirrored (data gradient).
__global__ __launch_bounds__(256) void conv_mfma_pack_kernel(const float* __restrict__ w, long long s_co, long long s_ci, long long s_kh, long long s_kw,
                                                             int cout, int cin, uint4* __restrict__ pack_fwd, uint4* __restrict__ pack_bwd,
                                                             int frags_fwd, int frags_bwd) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int lane = gid & 63;
  int f = gid >> 6;                                      // (tile, chunk, tap, block) of either pack
  const bool bwd = f >= frags_fwd;
  if (bwd) f -= frags_fwd;
  if (bwd ? (f >= frags_bwd || !pack_bwd) : !pack_fwd) return;
  const int n_out = bwd ? cin : cout, k_in = bwd ? cout : cin;
  const int nchunks = (k_in + CK - 1) / CK, NB = blocks_for(n_out);
  const int blk = f % NB, tap = (f / NB) % 9, chunk = (f / (NB * 9)) % nchunks, tile = f / (NB * 9 * nchunks);
  const int o = (tile * NB + blk) * 32 + (lane & 31);
  const int i0 = chunk * CK + (lane >> 5) * 8;
  const int kh = tap / 3, kw = tap % 3;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int i = i0 + e;
    float val = 0.f;
    if (o < n_out && i < k_in)
      val = bwd ? w[(long long)i * s_co + (long long)o * s_ci + (2 - kh) * s_kh + (2 - kw) * s_kw]
                : w[(long long)o * s_co + (long long)i * s_ci + kh * s_kh + kw * s_kw];
    v[e] = val;
  }
  unsigned p[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split2(v[2 * q], v[2 * q + 1], p[0][q], p[1][q], p[2][q]);
  uint4* dst = (bwd ? pack_bwd : pack_fwd) + ((size_t)f * 3) * 64 + lane;
#pragma unroll
  for (int pc = 0; pc < 3; ++pc) dst[pc * 64] = make_uint4(p[pc][0], p[pc][1], p[pc][2], p[pc][3]);
}

// y (B,Ho,Wo,n_out) = conv3x3(x (B,Hi,Wi,k_in) zero-extended, pack) + bias;  Ho = Hi + 2 pad - 2, pad in 0..2
template <int NB>
__global__ __launch_bounds__(NT, 2) void conv_mfma_kernel(const float* __restrict__ x, const uint4* __restrict__ pack, const float* __restrict__ bias,
                                                          int Hi, int Wi, int Ho, int Wo, int k_in, int n_out, int pad, int tiles_x, int tiles_y,
                                                          float* __restrict__ y) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* const s_b = smem + A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, ntile = blockIdx.y;
  int tile = blockIdx.x;
  {
    // XCD x = blockIdx.x % 8 gets a contiguous band of tiles: neighbours share halos (and the packed weights) in one L2
    const int ntiles = gridDim.x, xc = tile & 7, base = ntiles >> 3, extra = ntiles & 7;
    tile = xc * base + min(xc, extra) + (tile >> 3);
  }
  const int X0 = (tile % tiles_x) * TW, Y0 = (tile / tiles_x) * TH;
  const int nchunks = (k_in + CK - 1) / CK;
  const float* xb = x + (size_t)b * Hi * Wi * k_in;
  const char* pk = reinterpret_cast<const char*>(pack) + (size_t)ntile * nchunks * 9 * (3 * NB * FRAG);

  // this thread's halo positions (the same for every chunk): index i = tid + j * NT -> pixel i >> 2, channel quad i & 3
  int g_off[PRE];          // element offset of (pixel, quad) in x, or -1 outside the image
  int l_off[PRE];          // byte offset in a piece plane, or -1 beyond the halo
#pragma unroll
  for (int j = 0; j < PRE; ++j) {
    const int i = tid + j * NT, px = i >> 2, q = i & 3;
    const int hy = px / HW, hx = px - hy * HW;
    const int Y = Y0 - pad + hy, X = X0 - pad + hx;
    l_off[j] = px < HN ? px * PSTR + q * 8 : -1;
    g_off[j] = (px < HN && Y >= 0 && Y < Hi && X >= 0 && X < Wi) ? (Y * Wi + X) * k_in + q * 4 : -1;
  }
  float4 pre[PRE];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < PRE; ++j) {
      const int c0 = chunk * CK + (((tid + j * NT) & 3) << 2);
      pre[j] = (g_off[j] >= 0 && c0 < k_in) ? *reinterpret_cast<const float4*>(xb + g_off[j] + chunk * CK) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int j = 0; j < PRE; ++j) {
      if (l_off[j] >= 0) {
        unsigned a1, a2, a3, b1, b2, b3;
        split2(pre[j].x, pre[j].y, a1, a2, a3);
        split2(pre[j].z, pre[j].w, b1, b2, b3);
        *reinterpret_cast<uint2*>(smem + l_off[j]) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(smem + A_PIECE + l_off[j]) = make_uint2(a2, b2);
        *reinterpret_cast<uint2*>(smem + 2 * A_PIECE + l_off[j]) = make_uint2(a3, b3);
      }
    }
  };
  // the B fragments of step s (= chunk * 9 + tap) -> buffer s & 1: 3 * NB wave-wide 16-byte loads, dealt round-robin to the four waves
  auto fetch_b = [&](int s) {
    const char* src = pk + (size_t)s * (3 * NB * FRAG);
    unsigned char* dst = s_b + (s & 1) * (3 * NB * FRAG);
#pragma unroll
    for (int f = 0; f < (3 * NB + 3) / 4; ++f) {
      const int fr = wave + 4 * f;
      if (fr < 3 * NB)
        __builtin_amdgcn_global_load_lds(src + fr * FRAG + lane * 16, reinterpret_cast<uint4*>(dst + fr * FRAG), 16, 0, 0);
    }
  };

  f16v acc[2][NB];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // A fragment of this lane: tile row 2*wave + m, column lane & 31, channels (lane >> 5) * 8 .. + 7 of the chunk
  const unsigned char* a_lane = smem + ((2 * wave) * HW + (lane & 31)) * PSTR + (lane >> 5) * 16;
  const unsigned char* b_lane = s_b + lane * 16;

  fetch(0);
  fetch_b(0);
  const int nsteps = nchunks * 9;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();                       // the previous chunk's A reads are done
    stage();
    __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): this step's weights have landed
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = chunk * 9 + tap;
      const int ty = tap / 3, tx = tap % 3;
      uint4 af[2][3], bfr[NB][3];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) af[m][pc] = *reinterpret_cast<const uint4*>(a_lane + ((m + ty) * HW + tx) * PSTR + pc * A_PIECE);
      const unsigned char* bb = b_lane + (s & 1) * (3 * NB * FRAG);
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) bfr[n][pc] = *reinterpret_cast<const uint4*>(bb + (n * 3 + pc) * FRAG);
      // The next step's weights go out BEHIND this step's fragment reads: the compiler waits for an LDS-bound load in front of the
      // next LDS read it cannot tell apart from the destination -- here that is the next step's, behind the MFMAs and the barrier.
      if (s + 1 < nsteps) fetch_b(s + 1);
      if (tap == 5 && chunk + 1 < nchunks) fetch(chunk + 1);
      // six partial products per accumulator, the small ones first; consecutive MFMAs go to different accumulators
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, af[m][PA[t]]), __builtin_bit_cast(bf8, bfr[n][PB[t]]), acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);           // the wait below stays behind the MFMAs: they cover the loads' latency
      if (tap < 8) {
        __builtin_amdgcn_s_waitcnt(0x0F70);      // next tap's weights (and nothing else is outstanding except the halo prefetch: it
        __syncthreads();                         // completes with them -- issued three taps earlier)
      }
    }
  }

  // C layout of 32x32: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int co = (ntile * NB + n) * 32 + (lane & 31);
    if (co >= n_out) continue;
    const float bv = bias ? bias[co] : 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int Y = Y0 + 2 * wave + m;
      if (Y >= Ho) continue;
      float* row = y + (((size_t)b * Ho + Y) * Wo) * n_out + co;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int X = X0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (X < Wo) row[(size_t)X * n_out] = acc[m][n][r] + bv;
      }
    }
  }
}

static size_t pack_bytes(int n_out, int k_in) {
  const int NB = blocks_for(n_out), tiles = (n_out + 32 * NB - 1) / (32 * NB), nchunks = (k_in + CK - 1) / CK;
  return (size_t)tiles * nchunks * 9 * 3 * NB * FRAG;
}

template <int NB>
static int launch(const float* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, float* y, hipStream_t stream) {
  const int Ho = Hi + 2 * pad - 2, Wo = Wi + 2 * pad - 2;
  const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
  auto kern = conv_mfma_kernel<NB>;
  static bool attr_set = false;
  if (!attr_set) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes<NB>());
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid(tiles_x * tiles_y, (n_out + 32 * NB - 1) / (32 * NB), B);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds_bytes<NB>(), stream, x, static_cast<const uint4*>(pack), bias, Hi, Wi, Ho, Wo, k_in, n_out, pad, tiles_x,
                     tiles_y, y);
  return (int)hipGetLastError();
}

}  // namespace cm
}  // namespace dd

extern "C" int dd_conv3x3_mfma_supported(int cin, int cout) {
  return (cin >= 16 && cout >= 16 && cin % 4 == 0 && cin <= 1024 && cout <= 1024) ? 1 : 0;
}

extern "C" size_t dd_conv3x3_mfma_pack_bytes(int n_out, int k_in) { return dd::cm::pack_bytes(n_out, k_in); }

extern "C" int dd_conv3x3_mfma_pack(const float* weight, long long s_co, long long s_ci, long long s_kh, long long s_kw, int cout, int cin, void* pack_fwd,
                                    void* pack_bwd_data, void* stream) {
  using namespace dd::cm;
  if (!weight || (!pack_fwd && !pack_bwd_data) || cout < 1 || cin < 1) return (int)hipErrorInvalidValue;
  const int frags_fwd = pack_fwd ? (int)(pack_bytes(cout, cin) / (3 * FRAG)) : 0;
  const int frags_bwd = pack_bwd_data ? (int)(pack_bytes(cin, cout) / (3 * FRAG)) : 0;
  const int threads = (frags_fwd + frags_bwd) * 64;
  // one launch, two regions: the forward pack's fragments first; each region decodes its own (tile, chunk, tap, block)
  hipLaunchKernelGGL(conv_mfma_pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), weight, s_co, s_ci, s_kh, s_kw, cout,
                     cin, static_cast<uint4*>(pack_fwd), static_cast<uint4*>(pack_bwd_data), frags_fwd, frags_bwd);
  return (int)hipGetLastError();
}

extern "C" int dd_conv3x3_mfma(const float* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, float* y,
                               void* stream) {
  using namespace dd::cm;
  if (!x || !pack || !y || B < 1 || pad < 0 || pad > 2 || Hi + 2 * pad < 3 || Wi + 2 * pad < 3 || k_in % 4 || k_in < 4 || n_out < 1) return (int)hipErrorInvalidValue;
  if ((size_t)Hi * Wi * k_in >= (1ull << 31) || (reinterpret_cast<unsigned long long>(x) & 15ull)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (blocks_for(n_out)) {
    case 1: return launch<1>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
    case 2: return launch<2>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
    default: return launch<3>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
  }
}
