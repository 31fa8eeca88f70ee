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
constexpr int CK = 32;                                      // input channels per chunk = two K-16 MFMA steps per tap
constexpr int PSTR = 80;                                    // bytes per halo pixel: 32 halves + 16 bytes of padding (5 x 16: conflict-free)
constexpr int A_BYTES = HN * PSTR;                          // 27 200
constexpr int FRAG = 1024;                                  // one B fragment: 64 lanes x 16 bytes
constexpr int A_ITEMS = HN * 4;                             // 16-byte items of a chunk's halo (pixel, channel octet)
constexpr int A_PRE = (A_ITEMS + NT - 1) / NT;              // 6 per thread (the last round is partial)

__host__ __device__ inline int blocks_for(int n_out) { return n_out <= 32 ? 1 : 2; }
template <int NB>
constexpr int b_bytes() { return 9 * 2 * NB * FRAG; }       // a chunk's weight fragments: [tap][K step][n block][lane] x 16 bytes
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
  const int blk = f % NB, ks = (f / NB) % 2, tap = (f / (2 * NB)) % 9, chunk = (f / (18 * NB)) % nchunks, tile = f / (18 * NB * nchunks);
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
__global__ __launch_bounds__(NT, 2) void conv_half_kernel(const unsigned short* __restrict__ x, const uint4* __restrict__ pack, const float* __restrict__ bias,
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

  // this thread's halo items (the same for every chunk): item i = tid + j * NT -> pixel i >> 2, channel octet i & 3
  int g_off[A_PRE];          // element offset of (pixel, octet) in x, or -1 outside the image
  int l_off[A_PRE];          // byte offset in the halo plane, or -1 beyond the halo
#pragma unroll
  for (int j = 0; j < A_PRE; ++j) {
    const int i = tid + j * NT, px = i >> 2, q = i & 3;
    const int hy = px / HW, hx = px - hy * HW;
    const int Y = Y0 - pad + hy, X = X0 - pad + hx;
    l_off[j] = px < HN ? px * PSTR + q * 16 : -1;
    g_off[j] = (px < HN && Y >= 0 && Y < Hi && X >= 0 && X < Wi) ? (Y * Wi + X) * k_in + q * 8 : -1;
  }
  // Every thread issues exactly A_PRE + BP loads per chunk (positions outside the image or beyond the last channel re-read the image's
  // first pixel and are zeroed when staged; the weight items beyond a partial last round re-read the chunk's first): straight-line code,
  // the compiler's counted waits stay exact.
  u4 pre[A_PRE];
  auto fetch_a = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < A_PRE; ++j) {
      const int c0 = chunk * CK + (((tid + j * NT) & 3) << 3);
      const int off = (g_off[j] >= 0 && c0 < k_in) ? g_off[j] + chunk * CK : 0;
      pre[j] = *reinterpret_cast<const u4*>(xb + off);
    }
  };
  auto stage_a = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < A_PRE; ++j) {
      if (l_off[j] >= 0) {
        const int c0 = chunk * CK + (((tid + j * NT) & 3) << 3);
        const bool in = g_off[j] >= 0 && c0 < k_in;
        *reinterpret_cast<u4*>(smem + l_off[j]) = in ? pre[j] : u4{0u, 0u, 0u, 0u};
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

  fetch_a(0);
  fetch_b(0);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    lds_barrier();                         // the previous chunk's fragment reads are done
    stage_a(chunk);
    stage_b();
    lds_barrier();
    {
      const int nxt = min(chunk + 1, nchunks - 1);          // (the last chunk re-reads itself: no branch around the loads)
      fetch_a(nxt);
      fetch_b(nxt);
    }
    __builtin_amdgcn_sched_barrier(0);     // the loads go out IN FRONT of the MFMAs (left alone, the scheduler sinks them behind the block:
                                           // shorter live ranges, and the whole round trip exposed in front of the next staging)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ty = tap / 3, tx = tap % 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        u4 af[2], bf[NB];
#pragma unroll
        for (int m = 0; m < 2; ++m) af[m] = *reinterpret_cast<const u4*>(a_lane + ((m + ty) * HW + tx) * PSTR + ks * 32);
#pragma unroll
        for (int n = 0; n < NB; ++n) bf[n] = *reinterpret_cast<const u4*>(b_lane + ((tap * 2 + ks) * NB + n) * FRAG);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n) acc[m][n] = mfma<F16>(af[m], bf[n], acc[m][n]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // ... and the waits for them stay behind the block
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
  return (size_t)tiles * nchunks * 18 * NB * FRAG;
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
