// dd_conv_half.hip -- 3x3 stride-1 convolutions of the half-precision (fp16 / bf16) networks on the matrix pipe (gfx950).
//
// BASELINE.json config 5 runs its networks in half precision ("fp16 (CDNA4 MFMA conv)": the ResNet encoders' BasicBlock convolutions,
// the decoders' Conv3x3s -- reference networks/resnet_encoder.py:95-135, networks/depth_decoder.py:10-55, networks/motion_decoder.py:24-33,
// 48-66; the reference itself has no AMP, the half-precision step is this build's row g1).  Under autocast these layers ran on the
// library (MIOpen / CK: 290-480 TFLOP/s forward, 265-350 data gradient at config 5's shapes, profiles/r06_library_half_convs_config5.txt --
// 11-19 % of the dense half-precision MFMA peak); dd_conv_mfma.hip's kernels serve fp32 networks only (six partial products per
// multiply-add).  Here the operands ARE half precision: one v_mfma_f32_32x32x16_{f16,bf16} per 32 x 32 x 16 block, fp32 accumulation,
// the result rounded once (to nearest even) on its way out -- the arithmetic of the library's kernels (tests/test_conv_half_gpu.py
// holds both to float64 on the same inputs).
//
// Implicit GEMM, M = output pixels, N = output channels, K = 9 taps x input channels, laid out for ONE product per operand pair:
//   * a workgroup (256 threads, 4 waves) owns 8 rows x 32 columns of output pixels of one image and 32*NB output channels (NB <= 2);
//     wave w owns tile rows 2w, 2w+1 and all NB channel blocks: 2*NB accumulators of 32 x 32;
//   * K advances in chunks of 32 input channels (two K-16 MFMA steps per tap).  A chunk's input halo (10 x 34 pixels x 32 channels,
//     80-byte pixel stride: every A fragment is one conflict-free ds_read_b128 at a constant offset) AND all of the chunk's weight
//     fragments (9 taps x 2 steps x NB KB, fragment order, packed once per step from the fp32 master weights by
//     conv_half_pack_kernel: the cast that autocast would launch per layer happens there) are staged together: TWO LDS barriers per
//     chunk of 36*NB MFMAs per wave, none inside it -- the fp32 kernel's barrier per tap would cost as much as the four MFMAs
//     between two of them here;
//   * the next chunk's halo and weights are fetched into registers under the current chunk's MFMAs (plain loads with counted waits;
//     LDS-only barriers keep them in flight);
//   * 27 KB of halo + 18*NB KB of weights (64 KB at NB = 2): two workgroups per CU, one staging while the other multiplies.
// The data gradient is the same kernel on the output gradient with the weights packed transposed and mirrored (pad' = 2 - pad).
// The weight gradient of these layers stays with the library's half-precision kernel (hipops/functions.py: HalfConvFn).
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_attr.h"

namespace dd {
namespace ch {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;
typedef __attribute__((ext_vector_type(2))) float fl2;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) unsigned u4;        // native vector (HIP's uint4 struct arrays do not always leave scratch)

constexpr int TH = 8, TW = 32, NT = 256;
constexpr int HH = TH + 2, HW = TW + 2, HN = HH * HW;      // halo: 340 pixels
#ifndef DD_CH_KS
#define DD_CH_KS 1
#endif
constexpr int KS = DD_CH_KS;                                // K-16 MFMA steps per tap and chunk
constexpr int CK = 16 * KS;                                 // input channels per chunk
constexpr int PSTR = 32 * KS + 16;                          // bytes per halo pixel: the chunk's halves + 16 bytes of padding (80 = 5 x 16, 48 = 3 x 16: conflict-free)
constexpr int A_BYTES = HN * PSTR;                          // 27 200 (KS = 2)
constexpr int FRAG = 1024;                                  // one B fragment: 64 lanes x 16 bytes
constexpr int OCT = 2 * KS;                                 // channel octets (16-byte items) per pixel and chunk
constexpr int A_ITEMS = HN * OCT;                           // 16-byte items of a chunk's halo (pixel, channel octet)
constexpr int A_PRE = (A_ITEMS + NT - 1) / NT;              // 6 per thread (the last round is partial)
#ifndef DD_CH_MINWG
#define DD_CH_MINWG 3
#endif

__host__ __device__ inline int blocks_for(int n_out) { return n_out <= 32 ? 1 : 2; }
template <int NB>
constexpr int b_bytes() { return 9 * KS * NB * FRAG; }      // a chunk's weight fragments: [tap][K step][n block][lane] x 16 bytes
template <int NB>
constexpr int b_pre() { return (b_bytes<NB>() / 16 + NT - 1) / NT; }
template <int NB>
constexpr int lds_bytes() { return A_BYTES + b_bytes<NB>(); }

template <bool F16>
__device__ __forceinline__ unsigned pack2(float a, float b) {          // two fp32 -> two halves of the kernel's type, round to nearest even
  if (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector(fl2{a, b}, h2));
  return __builtin_bit_cast(unsigned, __builtin_convertvector(fl2{a, b}, bf2));
}
template <bool F16>
__device__ __forceinline__ unsigned short pack1(float a) {
  return static_cast<unsigned short>(pack2<F16>(a, 0.f) & 0xffffu);
}

// pack layout: [n tile][chunk][tap][K step][n block in tile][lane] x 16 bytes.  Lane l of a fragment holds, for output channel
// (tile * NB + block) * 32 + (l & 31), the input channels chunk * 32 + step * 16 + (l >> 5) * 8 + 0..7 of the tap, converted from the fp32
// weights.  transposed = 0: out = cout, in = cin, tap as stored (forward).  transposed = 1: out = cin, in = cout, tap mirrored (data gradient).
template <bool F16>
__global__ __launch_bounds__(256) void conv_half_pack_kernel(const float* __restrict__ w, long long s_co, long long s_ci, long long s_kh, long long s_kw,
                                                             int cout, int cin, uint4* __restrict__ pack_fwd, uint4* __restrict__ pack_bwd,
                                                             int frags_fwd, int frags_bwd) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int lane = gid & 63;
  int f = gid >> 6;
  const bool bwd = f >= frags_fwd;
  if (bwd) f -= frags_fwd;
  if (bwd ? (f >= frags_bwd || !pack_bwd) : !pack_fwd) return;
  const int n_out = bwd ? cin : cout, k_in = bwd ? cout : cin;
  const int nchunks = (k_in + CK - 1) / CK, NB = blocks_for(n_out);
  const int blk = f % NB, ks = (f / NB) % KS, tap = (f / (KS * NB)) % 9, chunk = (f / (9 * KS * NB)) % nchunks, tile = f / (9 * KS * NB * nchunks);
  const int o = (tile * NB + blk) * 32 + (lane & 31);
  const int i0 = chunk * CK + ks * 16 + (lane >> 5) * 8;
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
  (bwd ? pack_bwd : pack_fwd)[(size_t)f * 64 + lane] =
      make_uint4(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]), pack2<F16>(v[4], v[5]), pack2<F16>(v[6], v[7]));
}

template <bool F16>
__device__ __forceinline__ f16v mfma(const u4& a, const u4& b, const f16v& c) {
  if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}

// y (B,Ho,Wo,n_out) = conv3x3(x (B,Hi,Wi,k_in) zero-extended, pack) + bias;  Ho = Hi + 2 pad - 2, pad in 0..2; x, y in the half type
template <int NB, bool F16>
__global__ __launch_bounds__(NT, DD_CH_MINWG) void conv_half_kernel(const unsigned short* __restrict__ x, const uint4* __restrict__ pack, const float* __restrict__ bias,
                                                          int Hi, int Wi, int Ho, int Wo, int k_in, int n_out, int pad, int tiles_x, int tiles_y,
                                                          unsigned short* __restrict__ y) {
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
  const unsigned short* xb = x + (size_t)b * Hi * Wi * k_in;
  const char* pk = reinterpret_cast<const char*>(pack) + (size_t)ntile * nchunks * b_bytes<NB>();

  // this thread's halo items: item i = tid + j * NT -> pixel i / OCT, channel octet i % OCT.  Their offsets are recomputed per chunk from
  // the item index (a dozen integer instructions per item against 36 MFMAs per chunk) instead of living in 2 * A_PRE registers across the
  // MFMA block: with the fragments read one tap ahead only (DD_CH_TAP_FENCE) the kernel fits 128 VGPRs -- four workgroups per CU.
  auto item_lds = [&](int j) -> int {          // byte offset in the halo plane, or -1 beyond the halo
    const int i = tid + j * NT, px = i / OCT, q = i % OCT;
    return px < HN ? px * PSTR + q * 16 : -1;
  };
  auto item_gmem = [&](int j) -> int {         // element offset of (pixel, octet) in x, or -1 outside the image
    const int i = tid + j * NT, px = i / OCT, q = i % OCT;
    const int hy = px / HW, hx = px - hy * HW;
    const int Y = Y0 - pad + hy, X = X0 - pad + hx;
    return (px < HN && Y >= 0 && Y < Hi && X >= 0 && X < Wi) ? (Y * Wi + X) * k_in + q * 8 : -1;
  };
  // Every thread issues exactly A_PRE + BP loads per chunk (positions outside the image or beyond the last channel re-read the image's
  // first pixel and are zeroed when staged; the weight items beyond a partial last round re-read the chunk's first): straight-line code,
  // the compiler's counted waits stay exact.
  u4 pre[A_PRE];
#ifdef DD_CH_DEEP
  u4 pre1[A_PRE];              // a second halo in flight: the halo of chunk c + 2 is fetched while chunk c is multiplied (weights: c + 1, from L2)
#endif
  auto fetch_a = [&](int chunk, u4 (&pre)[A_PRE]) {
#pragma unroll
    for (int j = 0; j < A_PRE; ++j) {
      const int c0 = chunk * CK + (((tid + j * NT) % OCT) << 3);
      const int go = item_gmem(j);
      const int off = (go >= 0 && c0 < k_in) ? go + chunk * CK : 0;
      pre[j] = *reinterpret_cast<const u4*>(xb + off);
    }
  };
  auto stage_a = [&](int chunk, const u4 (&pre)[A_PRE]) {
#pragma unroll
    for (int j = 0; j < A_PRE; ++j) {
      const int lo = item_lds(j);
      if (lo >= 0) {
        const int c0 = chunk * CK + (((tid + j * NT) % OCT) << 3);
        const bool in = item_gmem(j) >= 0 && c0 < k_in;
        *reinterpret_cast<u4*>(smem + lo) = in ? pre[j] : u4{0u, 0u, 0u, 0u};
      }
    }
  };
  constexpr int BP = b_pre<NB>(), B_ITEMS = b_bytes<NB>() / 16;
  u4 bpre[BP];
  auto fetch_b = [&](int chunk) {
    const char* src = pk + (size_t)chunk * b_bytes<NB>();
#pragma unroll
    for (int r = 0; r < BP; ++r) {
      const int idx = tid + r * NT;
      bpre[r] = *reinterpret_cast<const u4*>(src + (idx < B_ITEMS ? idx : tid) * 16);
    }
  };
  auto stage_b = [&]() {
#pragma unroll
    for (int r = 0; r < BP; ++r) {
      const int idx = tid + r * NT;
      if (idx < B_ITEMS) *reinterpret_cast<u4*>(s_b + idx * 16) = bpre[r];
    }
  };
  // LDS-only barrier: every LDS operation of this wave has completed, global loads stay in flight (__syncthreads() drains them)
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  f16v acc[2][NB];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // A fragment of this lane: tile row 2*wave + m, column lane & 31, channels step * 16 + (lane >> 5) * 8 .. + 7 of the chunk
  const unsigned char* a_lane = smem + ((2 * wave) * HW + (lane & 31)) * PSTR + (lane >> 5) * 16;
  const unsigned char* b_lane = s_b + lane * 16;

  auto multiply = [&]() {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ty = tap / 3, tx = tap % 3;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        u4 af[2], bf[NB];
#pragma unroll
        for (int m = 0; m < 2; ++m) af[m] = *reinterpret_cast<const u4*>(a_lane + ((m + ty) * HW + tx) * PSTR + ks * 32);
#pragma unroll
        for (int n = 0; n < NB; ++n) bf[n] = *reinterpret_cast<const u4*>(b_lane + ((tap * KS + ks) * NB + n) * FRAG);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[m][n] = mfma<F16>(af[m], bf[n], acc[m][n]);
      }
#ifdef DD_CH_TAP_FENCE
      if (tap % DD_CH_TAP_FENCE == DD_CH_TAP_FENCE - 1) __builtin_amdgcn_sched_barrier(0);     // caps the fragments read ahead (registers)
#endif
    }
  };
  const int last = nchunks - 1;
#ifdef DD_CH_DEEP
  // Chunks in pairs, two halo register sets: set 0 holds the even chunks, set 1 the odd ones; an odd chunk count runs one all-zero
  // chunk more (its halo is staged as zeros: c0 >= k_in; its weights are the last chunk's -- the products vanish).
  fetch_a(0, pre);
  fetch_a(1, pre1);
  fetch_b(0);
  for (int chunk = 0; chunk < nchunks; chunk += 2) {
    lds_barrier();
    stage_a(chunk, pre);
    stage_b();
    lds_barrier();
    fetch_a(chunk + 2, pre);
    fetch_b(min(chunk + 1, last));
    __builtin_amdgcn_sched_barrier(0);
    multiply();
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();
    stage_a(chunk + 1, pre1);
    stage_b();
    lds_barrier();
    fetch_a(chunk + 3, pre1);
    fetch_b(min(chunk + 2, last));
    __builtin_amdgcn_sched_barrier(0);
    multiply();
    __builtin_amdgcn_sched_barrier(0);
  }
#else
  fetch_a(0, pre);
  fetch_b(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    lds_barrier();                         // the previous chunk's fragment reads are done
    stage_a(chunk, pre);
    stage_b();
    lds_barrier();
    fetch_a(min(chunk + 1, last), pre);    // (the last chunk re-reads itself: no branch around the loads)
    fetch_b(min(chunk + 1, last));
    __builtin_amdgcn_sched_barrier(0);     // the loads go out IN FRONT of the MFMAs (left alone, the scheduler sinks them behind the block:
                                           // shorter live ranges, and the whole round trip exposed in front of the next staging)
    multiply();
    __builtin_amdgcn_sched_barrier(0);     // ... and the waits for them stay behind the block
  }
#endif

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
      unsigned short* row = y + (((size_t)b * Ho + Y) * Wo) * n_out + co;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int X = X0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (X < Wo) row[(size_t)X * n_out] = pack1<F16>(acc[m][n][r] + bv);
      }
    }
  }
}

static size_t pack_bytes(int n_out, int k_in) {
  const int NB = blocks_for(n_out), tiles = (n_out + 32 * NB - 1) / (32 * NB), nchunks = (k_in + CK - 1) / CK;
  return (size_t)tiles * nchunks * 9 * KS * NB * FRAG;
}

template <int NB, bool F16>
static int launch(const void* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, void* y, hipStream_t stream) {
  const int Ho = Hi + 2 * pad - 2, Wo = Wi + 2 * pad - 2;
  const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
  auto kern = conv_half_kernel<NB, F16>;
  static dd::LdsAttrOnce lds_attr;          // per instantiation and device (dd_attr.h)
  if (const int rc = lds_attr.ensure(reinterpret_cast<const void*>(kern), lds_bytes<NB>())) return rc;
  dim3 grid(tiles_x * tiles_y, (n_out + 32 * NB - 1) / (32 * NB), B);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds_bytes<NB>(), stream, static_cast<const unsigned short*>(x), static_cast<const uint4*>(pack), bias, Hi, Wi, Ho,
                     Wo, k_in, n_out, pad, tiles_x, tiles_y, static_cast<unsigned short*>(y));
  return (int)hipGetLastError();
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------
// g_w[co][tap][ci] = sum over pixels of g[pixel][co] * x[pixel + tap - pad][ci]: M = co, N = ci, K = pixels, in the half type with fp32
// accumulation -- and an fp32 RESULT: the master weights' gradient leaves the kernel in their own precision (the library's
// half-precision weight gradient is rounded to the half type and promoted again by a cast launch; its split-K variants also need a
// zero-fill launch and add with atomics).  The structure is dd_conv_mfma.hip's conv_wgrad_kernel: the operands want K contiguous per
// lane, the tensors have the channels contiguous -- tiles are TRANSPOSED on their way into LDS ([channel][pixel], two horizontally
// adjacent pixels per 32-bit store); a tap's shift along the row is a shift of the fragment by one or two 16-bit values, formed in
// registers from one aligned read of ten pixels (v_alignbit_b32); a workgroup (4 waves) owns a 64 (co) x 64 (ci) block for a contiguous
// range of tiles, 36 accumulators, nine per wave, kept in registers across the tiles; ONE fp32 partial per workgroup, folded in a fixed
// order by conv_half_wgrad_fold_kernel (bit-reproducible).  With ONE product per operand pair (the fp32 kernel: six) a 2 x 16-pixel tile
// would be nine MFMAs per wave between two barriers: the tile is 8 rows x 16 columns here (eight K-16 steps, 72 MFMAs per wave and tile).
namespace wg {
constexpr int TR = 8, TC = 16;                    // output-gradient pixels per tile: 8 rows x 16 columns = eight K-16 steps
constexpr int XR = TR + 2, XC = TC + 2;           // halo of x
constexpr int XROW = 24;                          // halo row pitch in elements (48 bytes: every row starts 16-byte aligned)
constexpr int GSTR = TR * TC * 2 + 16;            // bytes per channel of the g tile (272 = 17 x 16: conflict-free fragment reads)
constexpr int XSTR = XR * XROW * 2 + 16;          // bytes per channel of the x tile (496 = 31 x 16)
constexpr int G_BYTES = 64 * GSTR, X_BYTES = 64 * XSTR;
constexpr int LDS = G_BYTES + X_BYTES;            // 49 152 bytes
constexpr int GITEMS = 8 * TR * (TC / 2);         // (channel octet, row, pixel pair) items of the g tile: 512
constexpr int GJ = GITEMS / NT;                   // 2 per thread
constexpr int XITEMS = 8 * XR * (XC / 2);         // ... of the x tile: 720
constexpr int XJ = (XITEMS + NT - 1) / NT;        // 3 per thread (the last round is partial)
constexpr int BLOCK = 64 * 9 * 64;                // floats of one partial
static_assert(GITEMS % NT == 0, "whole rounds of g items");
}  // namespace wg

template <bool F16>
__global__ __launch_bounds__(NT, 2) void conv_half_wgrad_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ g, int B, int Hi, int Wi,
                                                                int Ho, int Wo, int cin, int cout, int pad, int tiles_x, int tiles_y, float* __restrict__ partial) {
  using namespace wg;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* const s_g = smem;
  unsigned char* const s_x = smem + G_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = blockIdx.x, nks = gridDim.x, ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
  const int total = B * tiles_y * tiles_x;
  const int t_begin = (int)((long long)total * ks / nks), t_end = (int)((long long)total * (ks + 1) / nks);

  // staging items of this thread (the same for every tile): the pixel pair is the fastest index -- consecutive lanes store consecutive
  // 32-bit words of one channel row (conflict-free)
  int gq[GJ], grow[GJ], gpair[GJ], g_lds[GJ];
#pragma unroll
  for (int j = 0; j < GJ; ++j) {
    const int item = tid + j * NT;
    gq[j] = item / (TR * (TC / 2));
    const int rem = item - gq[j] * (TR * (TC / 2));
    grow[j] = rem / (TC / 2);
    gpair[j] = rem - grow[j] * (TC / 2);
    g_lds[j] = (8 * gq[j]) * GSTR + (grow[j] * TC + 2 * gpair[j]) * 2;
  }
  int xq[XJ], xrow[XJ], xpair[XJ], x_lds[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int item = tid + j * NT;
    xq[j] = item / (XR * (XC / 2));
    const int rem = item - xq[j] * (XR * (XC / 2));
    xrow[j] = rem / (XC / 2);
    xpair[j] = rem - xrow[j] * (XC / 2);
    x_lds[j] = item < XITEMS ? (8 * xq[j]) * XSTR + (xrow[j] * XROW + 2 * xpair[j]) * 2 : -1;
  }

  u4 pg[GJ][2], px[XJ][2];
  const u4 zero = {0u, 0u, 0u, 0u};
  auto fetch = [&](int t) {
    const int b = t / (tiles_y * tiles_x), r = t - b * (tiles_y * tiles_x);
    const int Y0 = (r / tiles_x) * TR, X0 = (r - (r / tiles_x) * tiles_x) * TC;
#pragma unroll
    for (int j = 0; j < GJ; ++j) {
      const int Y = Y0 + grow[j], X = X0 + 2 * gpair[j];
      const unsigned short* src = g + (((size_t)b * Ho + Y) * Wo + X) * cout + co0 + 8 * gq[j];
      const bool ok = co0 + 8 * gq[j] < cout && Y < Ho;
      pg[j][0] = (ok && X < Wo) ? *reinterpret_cast<const u4*>(src) : zero;
      pg[j][1] = (ok && X + 1 < Wo) ? *reinterpret_cast<const u4*>(src + cout) : zero;
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int Y = Y0 - pad + xrow[j], X = X0 - pad + 2 * xpair[j];
      const bool ok = x_lds[j] >= 0 && ci0 + 8 * xq[j] < cin && Y >= 0 && Y < Hi;
      const unsigned short* src = x + (((ptrdiff_t)b * Hi + Y) * Wi + X) * cin + ci0 + 8 * xq[j];
      px[j][0] = (ok && X >= 0 && X < Wi) ? *reinterpret_cast<const u4*>(src) : zero;
      px[j][1] = (ok && X + 1 >= 0 && X + 1 < Wi) ? *reinterpret_cast<const u4*>(src + cin) : zero;
    }
  };
  // eight channels of two adjacent pixels -> eight 32-bit words {pixel X, pixel X + 1}, one per channel row
  auto put = [&](unsigned char* plane, int stride, int off, const u4& a, const u4& b) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned lo = (a[q] & 0xffffu) | (b[q] << 16), hi = (a[q] >> 16) | (b[q] & 0xffff0000u);
      *reinterpret_cast<unsigned*>(plane + off + (2 * q) * stride) = lo;
      *reinterpret_cast<unsigned*>(plane + off + (2 * q + 1) * stride) = hi;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int j = 0; j < GJ; ++j) put(s_g, GSTR, g_lds[j], pg[j][0], pg[j][1]);
#pragma unroll
    for (int j = 0; j < XJ; ++j)
      if (x_lds[j] >= 0) put(s_x, XSTR, x_lds[j], px[j][0], px[j][1]);
  };

  // This wave's nine accumulators (dd_conv_mfma.hip): a GROUP is (ci block nb, tap row ty), its three taps tx = 0, 1, 2 read the same ten
  // pixels of a row of x shifted by 0 / 1 / 2.  Groups 0..5 = (nb 0, ty 0..2), (nb 1, ty 0..2).  Wave w owns group w on both co blocks
  // (six accumulators) and half of group 4 + (w >> 1): of its six (tx, co block) pairs, w even takes (0,0) (0,1) (1,0), w odd (2,0) (2,1) (1,1).
  const int ngroups = (cin - ci0 > 32) ? 6 : 3;
  const int g_full = wave, g_half = 4 + (wave >> 1), half = wave & 1;
  const bool full_on = g_full < ngroups, half_on = g_half < ngroups;
  auto group_off = [&](int gi) { return (gi / 3) * 32 * XSTR + (gi % 3) * (XROW * 2); };
  const int off_full = group_off(g_full), off_half = group_off(g_half);
  f16v acc[9];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  const unsigned char* a_lane = s_g + (lane & 31) * GSTR + (lane >> 5) * 16;
  const unsigned char* b_lane = s_x + (lane & 31) * XSTR + (lane >> 5) * 16;
  auto shifted = [](const u4& w0, unsigned w1, int tx) -> u4 {
    if (tx == 0) return w0;
    if (tx == 2) return u4{w0[1], w0[2], w0[3], w1};
    return u4{__builtin_amdgcn_alignbit(w0[1], w0[0], 16), __builtin_amdgcn_alignbit(w0[2], w0[1], 16), __builtin_amdgcn_alignbit(w0[3], w0[2], 16),
              __builtin_amdgcn_alignbit(w1, w0[3], 16)};
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  if (t_begin < t_end) fetch(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    lds_barrier();                        // the previous tile's fragment reads are done
    stage();
    lds_barrier();
    fetch(min(t + 1, t_end - 1));         // lands under this tile's MFMAs (the last tile re-reads itself: no branch around the loads)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int row = 0; row < TR; ++row) {
      u4 af[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) af[m] = *reinterpret_cast<const u4*>(a_lane + m * 32 * GSTR + row * (TC * 2));
      if (full_on) {                      // wave-uniform
        const unsigned char* src = b_lane + off_full + row * (XROW * 2);
        const u4 w0 = *reinterpret_cast<const u4*>(src);
        const unsigned w1 = *reinterpret_cast<const unsigned*>(src + 16);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const u4 bf = shifted(w0, w1, tx);
#pragma unroll
          for (int m = 0; m < 2; ++m) acc[tx * 2 + m] = mfma<F16>(af[m], bf, acc[tx * 2 + m]);
        }
      }
      if (half_on) {
        const unsigned char* src = b_lane + off_half + row * (XROW * 2);
        const u4 w0 = *reinterpret_cast<const u4*>(src);
        const unsigned w1 = *reinterpret_cast<const unsigned*>(src + 16);
        // pairs (tx, m): half 0 -> (0,0) (0,1) (1,0); half 1 -> (2,0) (2,1) (1,1)
        const u4 s0 = shifted(w0, w1, 0), s2 = shifted(w0, w1, 2), b1 = shifted(w0, w1, 1);
        u4 bd, am;                        // component-wise selects (an indexed choice between register vectors goes through scratch)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bd[q] = half == 0 ? s0[q] : s2[q];
          am[q] = half == 0 ? af[0][q] : af[1][q];
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[6 + m] = mfma<F16>(af[m], bd, acc[6 + m]);
        acc[8] = mfma<F16>(am, b1, acc[8]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  // the partial of this workgroup: [co 64][tap 9][ci 64]; C layout: column (ci) = lane & 31, row (co) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* P = partial + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * nks + ks) * BLOCK;
  auto put_acc = [&](const f16v& A, int gi, int tx, int m) {
    const int nb = gi / 3, tap = (gi % 3) * 3 + tx;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      P[(co * 9 + tap) * 64 + nb * 32 + (lane & 31)] = A[r];
    }
  };
  if (full_on) {
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int m = 0; m < 2; ++m) put_acc(acc[tx * 2 + m], g_full, tx, m);
  } else {
    // (a workgroup whose ci group has 32 or fewer channels: the groups of nb 1 do not exist; their slots of the partial are never read)
  }
  if (half_on) {
    put_acc(acc[6], g_half, half == 0 ? 0 : 2, 0);
    put_acc(acc[7], g_half, half == 0 ? 0 : 2, 1);
    put_acc(acc[8], g_half, 1, half);
  }
}

// g_weight (cout,3,3,cin) fp32 = sum over the nks partials of every (co group, ci group), in order (dd_conv_mfma.hip's fold)
__global__ __launch_bounds__(256) void conv_half_wgrad_fold_kernel(const float* __restrict__ partial, int nks, int cin, int cout, int ci_groups,
                                                                   float* __restrict__ gw) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;                  // (co, tap, ci) of the result
  const bool in = idx < cout * 9 * cin;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (in) {
    const int ci = idx % cin, tap = (idx / cin) % 9, co = idx / (9 * cin);
    const int z = co >> 6, yb = ci >> 6;
    const float* P = partial + ((size_t)z * ci_groups + yb) * nks * wg::BLOCK + ((co & 63) * 9 + tap) * 64 + (ci & 63);
    int k = wave;
    for (; k + 28 < nks; k += 32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += P[(size_t)(k + 4 * j) * wg::BLOCK];
    }
    for (; k < nks; k += 4) s[0] += P[(size_t)k * wg::BLOCK];
  }
  red[wave][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (wave == 0 && in) gw[idx] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

static int wgrad_splits(int B, int Ho, int Wo, int cin, int cout) {
  const int groups = ((cin + 63) / 64) * ((cout + 63) / 64);
  const int tiles = B * ((Ho + wg::TR - 1) / wg::TR) * ((Wo + wg::TC - 1) / wg::TC);
  int n = 512 / groups;                        // two workgroups per CU in all ...
  if (n > tiles / 3) n = tiles / 3;            // ... of at least three tiles (24 K-16 steps) each
  return n < 1 ? 1 : n;
}

}  // namespace ch
}  // namespace dd

// both channel counts in eights (16-byte octets of halves; the data gradient swaps the roles), dtype DD_DTYPE_F16 / DD_DTYPE_BF16
extern "C" int dd_conv3x3_half_supported(int cin, int cout) {
  return (cin >= 16 && cout >= 16 && cin % 8 == 0 && cout % 8 == 0 && cin <= 2048 && cout <= 2048) ? 1 : 0;
}

extern "C" size_t dd_conv3x3_half_pack_bytes(int n_out, int k_in) { return dd::ch::pack_bytes(n_out, k_in); }

extern "C" int dd_conv3x3_half_pack(const float* weight, long long s_co, long long s_ci, long long s_kh, long long s_kw, int cout, int cin, int dtype,
                                    void* pack_fwd, void* pack_bwd_data, void* stream) {
  using namespace dd::ch;
  if (!weight || (!pack_fwd && !pack_bwd_data) || cout < 1 || cin < 1 || (dtype != DD_DTYPE_F16 && dtype != DD_DTYPE_BF16)) return (int)hipErrorInvalidValue;
  const int frags_fwd = pack_fwd ? (int)(pack_bytes(cout, cin) / FRAG) : 0;
  const int frags_bwd = pack_bwd_data ? (int)(pack_bytes(cin, cout) / FRAG) : 0;
  const int threads = (frags_fwd + frags_bwd) * 64;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == DD_DTYPE_F16)
    hipLaunchKernelGGL(conv_half_pack_kernel<true>, dim3((threads + 255) / 256), dim3(256), 0, s, weight, s_co, s_ci, s_kh, s_kw, cout, cin,
                       static_cast<uint4*>(pack_fwd), static_cast<uint4*>(pack_bwd_data), frags_fwd, frags_bwd);
  else
    hipLaunchKernelGGL(conv_half_pack_kernel<false>, dim3((threads + 255) / 256), dim3(256), 0, s, weight, s_co, s_ci, s_kh, s_kw, cout, cin,
                       static_cast<uint4*>(pack_fwd), static_cast<uint4*>(pack_bwd_data), frags_fwd, frags_bwd);
  return (int)hipGetLastError();
}

extern "C" int dd_conv3x3_half(const void* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, int dtype, void* y,
                               void* stream) {
  using namespace dd::ch;
  if (!x || !pack || !y || B < 1 || pad < 0 || pad > 2 || Hi + 2 * pad < 3 || Wi + 2 * pad < 3 || k_in % 8 || k_in < 8 || n_out < 1 ||
      (dtype != DD_DTYPE_F16 && dtype != DD_DTYPE_BF16))
    return (int)hipErrorInvalidValue;
  if ((size_t)Hi * Wi * k_in >= (1ull << 31) || (reinterpret_cast<unsigned long long>(x) & 15ull)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool f16 = dtype == DD_DTYPE_F16;
  if (blocks_for(n_out) == 1) return f16 ? launch<1, true>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s) : launch<1, false>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
  return f16 ? launch<2, true>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s) : launch<2, false>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
}

extern "C" size_t dd_conv3x3_half_wgrad_workspace_bytes(int B, int Ho, int Wo, int cin, int cout) {
  const size_t groups = (size_t)((cin + 63) / 64) * ((cout + 63) / 64);
  return groups * dd::ch::wgrad_splits(B, Ho, Wo, cin, cout) * dd::ch::wg::BLOCK * sizeof(float);
}

extern "C" int dd_conv3x3_half_bwd_weight(const void* x, const void* g_out, int B, int Hi, int Wi, int cin, int cout, int pad, int dtype, float* g_weight,
                                          void* workspace, size_t workspace_bytes, void* stream) {
  using namespace dd::ch;
  if (!x || !g_out || !g_weight || !workspace || B < 1 || pad < 0 || pad > 1 || cin % 8 || cout % 8 || cin < 8 || cout < 8 ||
      (dtype != DD_DTYPE_F16 && dtype != DD_DTYPE_BF16))
    return (int)hipErrorInvalidValue;
  const int Ho = Hi + 2 * pad - 2, Wo = Wi + 2 * pad - 2;
  if (Ho < 1 || Wo < 1 || workspace_bytes < dd_conv3x3_half_wgrad_workspace_bytes(B, Ho, Wo, cin, cout)) return (int)hipErrorInvalidValue;
  if ((size_t)B * Hi * Wi * cin >= (1ull << 31) || (size_t)B * Ho * Wo * cout >= (1ull << 31)) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<unsigned long long>(x) | reinterpret_cast<unsigned long long>(g_out)) & 15ull) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_x = (Wo + wg::TC - 1) / wg::TC, tiles_y = (Ho + wg::TR - 1) / wg::TR;
  const int nks = wgrad_splits(B, Ho, Wo, cin, cout), ci_groups = (cin + 63) / 64, co_groups = (cout + 63) / 64;
  const unsigned short* xs = static_cast<const unsigned short*>(x);
  const unsigned short* gs = static_cast<const unsigned short*>(g_out);
  if (dtype == DD_DTYPE_F16)
    hipLaunchKernelGGL(conv_half_wgrad_kernel<true>, dim3(nks, ci_groups, co_groups), dim3(NT), wg::LDS, s, xs, gs, B, Hi, Wi, Ho, Wo, cin, cout, pad, tiles_x, tiles_y,
                       static_cast<float*>(workspace));
  else
    hipLaunchKernelGGL(conv_half_wgrad_kernel<false>, dim3(nks, ci_groups, co_groups), dim3(NT), wg::LDS, s, xs, gs, B, Hi, Wi, Ho, Wo, cin, cout, pad, tiles_x, tiles_y,
                       static_cast<float*>(workspace));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(conv_half_wgrad_fold_kernel, dim3((cout * 9 * cin + 63) / 64), dim3(256), 0, s, static_cast<const float*>(workspace), nks, cin, cout, ci_groups,
                     g_weight);
  return (int)hipGetLastError();
}
