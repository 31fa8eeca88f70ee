// dd_conv_mfma.hip -- 3x3 stride-1 convolutions of the motion decoders / ResNet blocks at fp32 accuracy on the bf16 matrix pipe (gfx950).
//
// MIOpen runs these (reference networks/motion_decoder.py:24-33,57-66: two 3x3 convolutions on 64-512 channels per level, two
// decoders) on the fp32 MFMA forms, whose peak is 1/16 of the bf16 forms' (157 TFLOP/s against 2.5 PFLOP/s), at 85-110 TFLOP/s.
// Here every fp32 operand is split EXACTLY into three bf16 pieces, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1),
// x3 = x - x1 - x2: 8 + 8 + 8 significand bits, the residuals are exact in fp32), and the product x * w is the six partial
// products x1w1 + x1w2 + x2w1 + x1w3 + x2w2 + x3w1 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Every bf16 x bf16
// product is exact in the accumulator's fp32; the three dropped cross terms are at most 2^-23 |x w| (2^-24.4 measured:
// tests/test_split_bf16.py, the arithmetic restated in oracle/ref_split_bf16.py) -- the size of the rounding of ONE fp32 product.  The result has the accuracy of an fp32 FMA chain (tests/test_conv_mfma_gpu.py: against float64, beside
// MIOpen's fp32 result) at 6/16 of the fp32 form's matrix-pipe time.
//
// Implicit GEMM, M = output pixels, N = output channels, K = 9 taps x input channels:
//   * a workgroup (256 threads, 4 waves) owns 8 rows x 32 columns of output pixels of one image and 32*NB output channels;
//     wave w owns tile rows 2w, 2w+1 (two 32-pixel M blocks) and all NB N blocks: 2*NB accumulators of 32x32;
//   * the input halo (10 x 34 pixels) of one 16-channel chunk is split once and staged in LDS as three bf16 planes,
//     [pixel][16 channels + pad] with a 48-byte pixel stride: the A fragment of tap (ty,tx) is one ds_read_b128 per piece at a
//     constant offset, conflict-free (3 x 16 bytes: 16 consecutive pixels fall on 16 distinct 16-byte bank groups);
//   * the weights are split and laid out in fragment order once per step (conv_mfma_pack_kernel): per (chunk, tap) the workgroup
//     fetches 3*NB KB two steps ahead (registers -> LDS, double-buffered);
//   * software pipeline: the fragments of step s + 1 are read into a second register set while the MFMAs of step s run, the next
//     chunk's halo is fetched into registers seven steps ahead; one LDS-only barrier per step.
// Tried and measured no better (round 5): starting the second resident workgroup of every CU 4-35 us late so that the two are out of
// phase (165-199 against 167 us).  Tried and measured slower: persistent workgroups that walk over several tiles, fetch the next tile's first halo under the
// current tile's last chunk and issue a finished tile's stores behind the next tile's staging -- 193 against 175 us: on gfx9 stores
// count in vmcnt like loads, in order, so the first counted wait of the next tile (its weights) also waits for the 64 stores to land.
// LDS: 48 960 B of halo + 2 x 4*ceil(3*NB/4) KB of weights (65 KB at NB = 2): two workgroups per CU, one staging while the other multiplies.
// The data gradient is the same kernel on the output gradient with the weights packed transposed and flipped (pad' = 2 - pad).
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "../../include/dynamo_hip.h"
#include "dd_attr.h"
#include "dd_split.h"

// -DDD_CM_EXP=<bits>: timing experiments only (wrong results) -- scripts/microbench/conv_mfma_variants.hip
#ifndef DD_CM_EXP
#define DD_CM_EXP 0
#endif

namespace dd {
namespace cm {

constexpr int TH = 8, TW = 32, NT = 256;
constexpr int HH = TH + 2, HW = TW + 2, HN = HH * HW;      // halo
constexpr int CK = 16;                                      // channels per chunk = the K of one MFMA
// Halo layout in LDS, per piece plane.  DD_CM_SWZ = 0 (rounds 5's): 48 bytes per pixel = 16 bf16 + 16 bytes of padding -- 16 consecutive
// pixels fall on 16 distinct 16-byte bank groups, every fragment address is a compile-time constant away from the lane's base; 49 KB of
// halo + 16 KB of weights: TWO workgroups per CU.  DD_CM_SWZ = 1 (round 6): 32 bytes per pixel, no padding, the two 16-byte halves of a
// pixel swapped where bit 2 of its index is set (eight consecutive pixels then cover all eight 16-byte groups of a 128-byte LDS clock
// whatever the tap offset): 32.6 KB + 16 KB = 49 KB -- THREE workgroups per CU (the lever that took dd_conv_half.hip from behind the
// library to ahead of it), with ONE fragment set (DD_CM_ONESET: 158 VGPRs); the fragment address costs five integer instructions per
// read instead of none.  MEASURED (profiles/r06_conv_mfma_swizzle.txt): 175.9 against 185.5 us forward and 154.9 against 160.2 data
// gradient at 12x64x64x96x320, but 69.2 against 60.3 at 12x128x128x24x80 (the single fragment set exposes the LDS round trip where a
// workgroup has few steps), flat kernel unchanged; in the step 333.8 / 334.5 against 332.8 / 331.4 img/s -- inside the noise.  These
// kernels are MFMA-heavy (24 MFMAs per step): occupancy is not their bound.  Shipped: 0.
#ifndef DD_CM_SWZ
#define DD_CM_SWZ 0
#endif
#ifndef DD_CM_ONESET
#define DD_CM_ONESET DD_CM_SWZ
#endif
constexpr int PSTR = DD_CM_SWZ ? 32 : 48;                   // bytes per halo pixel and piece
constexpr int A_PIECE = HN * PSTR;
constexpr int A_BYTES = 3 * A_PIECE;
// byte offset, inside a piece plane, of the 16-byte half `half` (channels half * 8 .. + 7 of the chunk) of halo pixel px
__device__ __forceinline__ int halo_addr(int px, int half) {
  return DD_CM_SWZ ? px * 32 + ((half ^ ((px >> 2) & 1)) << 4) : px * 48 + half * 16;
}
// Partial products per multiply-add (the `products` argument of the _n entry points) -> bf16 pieces of each operand that take part:
// 6 = x1w1 + x1w2 + x2w1 + x1w3 + x2w2 + x3w1 (three pieces: fp32 accuracy, the default); 3 = x1w1 + x1w2 + x2w1 (two pieces, 2^-16:
// what torch.set_float32_matmul_precision("high") names "bf16x3"); 1 = x1w1 (operands rounded to bf16, fp32 accumulation: "medium").
__host__ __device__ constexpr int pieces_of(int products) { return products == 6 ? 3 : (products == 3 ? 2 : 1); }
constexpr int FRAG = 1024;                                  // one B fragment: 64 lanes x 16 bytes
constexpr int PRE = (HN * 4 + NT - 1) / NT;                 // float4 loads per thread and chunk (6)

// N blocks of 32 output channels per workgroup: all of them up to 96 channels (the activations are staged and split once), else 64 per workgroup
__host__ __device__ inline int blocks_for(int n_out) { return n_out <= 32 ? 1 : (n_out <= 64 ? 2 : (n_out <= 96 ? 3 : 2)); }
// fragments (one per wave: 64 lanes x 3 pieces x 16 bytes) of the pack of an (n_out, k_in) layer
__host__ __device__ inline int frags_of(int n_out, int k_in) {
  const int NB = blocks_for(n_out);
  return ((n_out + 32 * NB - 1) / (32 * NB)) * ((k_in + CK - 1) / CK) * 9 * NB;
}

// one weight buffer: the 3 * NB fragments of a step, rounded up to whole rounds of four (one fragment per wave and round; the
// surplus slots take the clamped re-reads of waves without a fragment in the last round -- loads and stores stay unconditional)
template <int NB>
constexpr int b_buf_bytes() { return ((3 * NB + 3) / 4) * 4 * FRAG; }

template <int NB>
constexpr int lds_bytes() { return A_BYTES + 2 * b_buf_bytes<NB>(); }

// pack layout: [n tile][chunk][tap][n block in tile][piece][lane] x 16 bytes.  lane l of a fragment holds, for output channel
// (tile * NB + block) * 32 + (l & 31), the input channels chunk * 16 + (l >> 5) * 8 + 0..7 of the tap.
// transposed = 0: out = cout, in = cin, tap as stored (forward).  transposed = 1: out = cin, in = cout, tap mirrored (data gradient).
__device__ __forceinline__ void pack_fragment(const float* __restrict__ w, long long s_co, long long s_ci, long long s_kh, long long s_kw, int cout, int cin,
                                              uint4* __restrict__ pack_fwd, uint4* __restrict__ pack_bwd, int frags_fwd, int frags_bwd, int gid) {
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

__global__ __launch_bounds__(256) void conv_mfma_pack_kernel(const float* __restrict__ w, long long s_co, long long s_ci, long long s_kh, long long s_kw,
                                                             int cout, int cin, uint4* __restrict__ pack_fwd, uint4* __restrict__ pack_bwd,
                                                             int frags_fwd, int frags_bwd) {
  pack_fragment(w, s_co, s_ci, s_kh, s_kw, cout, cin, pack_fwd, pack_bwd, frags_fwd, frags_bwd, blockIdx.x * 256 + threadIdx.x);
}

// The packs of MANY layers in one launch (round 6: a network's 3x3 layers were 8-21 pack launches of 4-8 us per forward, each in front of
// its convolution on the network's stream).  jobs: PACK_JOB_WORDS 64-bit words per layer -- weight pointer, its four element strides, cout,
// cin, pack_fwd, pack_bwd (0: none), the first workgroup of the layer; block_job: the layer of every workgroup.  The fragment a thread
// writes, and every byte of it, is what conv_mfma_pack_kernel writes for that layer.
constexpr int PACK_JOB_WORDS = 10;
__global__ __launch_bounds__(256) void conv_mfma_pack_many_kernel(const long long* __restrict__ jobs, const int* __restrict__ block_job) {
  const long long* j = jobs + (size_t)block_job[blockIdx.x] * PACK_JOB_WORDS;
  const int cout = (int)j[5], cin = (int)j[6];
  uint4* const pf = reinterpret_cast<uint4*>(j[7]);
  uint4* const pb = reinterpret_cast<uint4*>(j[8]);
  const int frags_fwd = pf ? frags_of(cout, cin) : 0, frags_bwd = pb ? frags_of(cin, cout) : 0;
  pack_fragment(reinterpret_cast<const float*>(j[0]), j[1], j[2], j[3], j[4], cout, cin, pf, pb, frags_fwd, frags_bwd,
                (blockIdx.x - (int)j[9]) * 256 + threadIdx.x);
}

// y (B,Ho,Wo,n_out) = conv3x3(x (B,Hi,Wi,k_in) zero-extended, pack) + bias;  Ho = Hi + 2 pad - 2, pad in 0..2
template <int NB, int NP>
__global__ __launch_bounds__(NT, (DD_CM_SWZ && NB <= 2) ? 3 : 2) void conv_mfma_kernel(const float* __restrict__ x, const uint4* __restrict__ pack, const float* __restrict__ bias,
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
    l_off[j] = px < HN ? halo_addr(px, q >> 1) + (q & 1) * 8 : -1;
    g_off[j] = (px < HN && Y >= 0 && Y < Hi && X >= 0 && X < Wi) ? (Y * Wi + X) * k_in + q * 4 : -1;
  }
  // Every thread issues exactly PRE loads per chunk (positions outside the image or beyond the last channel read the image's first
  // pixel and are zeroed when staged): the counted s_waitcnt below relies on it.
  float4 pre[PRE];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < PRE; ++j) {
      const int c0 = chunk * CK + (((tid + j * NT) & 3) << 2);
      const int off = (g_off[j] >= 0 && c0 < k_in) ? g_off[j] + chunk * CK : 0;
      pre[j] = *reinterpret_cast<const float4*>(xb + off);
    }
  };
  auto stage = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < PRE; ++j) {
      if (l_off[j] >= 0) {
        const int c0 = chunk * CK + (((tid + j * NT) & 3) << 2);
        const bool in = g_off[j] >= 0 && c0 < k_in;
        const float4 v = in ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned a1, a2, a3, b1, b2, b3;
        split2(v.x, v.y, a1, a2, a3);
        split2(v.z, v.w, b1, b2, b3);
        *reinterpret_cast<uint2*>(smem + l_off[j]) = make_uint2(a1, b1);
        if (pieces_of(NP) > 1) *reinterpret_cast<uint2*>(smem + A_PIECE + l_off[j]) = make_uint2(a2, b2);
        if (pieces_of(NP) > 2) *reinterpret_cast<uint2*>(smem + 2 * A_PIECE + l_off[j]) = make_uint2(a3, b3);
      }
    }
  };
  // The B fragments of step s (= chunk * 9 + tap) -> buffer s & 1, through registers: 3 * NB KB per step, 16 bytes per thread and round.
  // (global_load_lds would save the round trip, but the compiler cannot tell its LDS destination from the fragment reads and puts a
  // full vmcnt(0) in front of every LDS read that follows one -- which also drains the halo prefetch.  With plain loads it counts.)
  constexpr int BR = (3 * NB + 3) / 4;             // rounds: wave w moves fragment w + 4 r (64 lanes x 16 bytes) in round r
  uint4 breg[BR];
#pragma unroll
  for (int r = 0; r < BR; ++r) breg[r] = make_uint4(0u, 0u, 0u, 0u);
  auto fetch_b = [&](int s) {
    const char* src = pk + (size_t)s * (3 * NB * FRAG) + lane * 16;
#pragma unroll
    for (int r = 0; r < BR; ++r)       // unconditional (a wave without a fragment in the last round re-reads the step's first): straight-line
      breg[r] = *reinterpret_cast<const uint4*>(src + (wave + 4 * r < 3 * NB ? wave + 4 * r : 0) * FRAG);         // code, exact wait counts
  };
  auto store_b = [&](int s) {
    unsigned char* dst = s_b + (s & 1) * b_buf_bytes<NB>() + lane * 16;
#pragma unroll
    for (int r = 0; r < BR; ++r) *reinterpret_cast<uint4*>(dst + (wave + 4 * r) * FRAG) = breg[r];
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

  // A fragment of this lane: tile row 2*wave + m, column lane & 31, channels (lane >> 5) * 8 .. + 7 of the chunk
  const int a_pix = (2 * wave) * HW + (lane & 31), a_half = lane >> 5;
  const unsigned char* b_lane = s_b + lane * 16;

  // Software pipeline.  The fragments of step s + 1 are read into a second register set and the weights of step s + 2 are fetched into
  // the buffer step s has finished with WHILE the MFMAs of step s run; the barrier at the end of a step finds everything in place.
  uint4 af[DD_CM_ONESET ? 1 : 2][2][3], bfr[DD_CM_ONESET ? 1 : 2][NB][3];
  auto read_frags = [&](int set, int tap, int s) {
    const int ty = tap / 3, tx = tap % 3;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const unsigned char* ap = smem + halo_addr(a_pix + (m + ty) * HW + tx, a_half);
#pragma unroll
      for (int pc = 0; pc < pieces_of(NP); ++pc) af[set][m][pc] = *reinterpret_cast<const uint4*>(ap + pc * A_PIECE);
    }
    const unsigned char* bb = b_lane + (s & 1) * b_buf_bytes<NB>();
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int pc = 0; pc < pieces_of(NP); ++pc) bfr[set][n][pc] = *reinterpret_cast<const uint4*>(bb + (n * 3 + pc) * FRAG);
  };

  const int nsteps = nchunks * 9;
  fetch_b(0);
  fetch(0);
  store_b(0);
  fetch_b(1);                              // nsteps >= 9
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    lds_barrier();                         // the previous chunk's fragment reads are done
    stage(chunk);
    store_b(chunk * 9 + 1);                // the weights of the chunk's second step (its first step's went in one step earlier)
    lds_barrier();
    if (!DD_CM_ONESET) read_frags(0, 0, chunk * 9);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = chunk * 9 + tap, cur = DD_CM_ONESET ? 0 : (tap & 1);
      // No branch in this loop body: the loads of the last steps / last chunk are clamped re-reads, so that the compiler's wait
      // counters stay exact (a conditional load makes every later wait a vmcnt(0), which drains the halo prefetch with the weights).
      // DD_CM_ONESET (with the swizzled layout: three workgroups per CU): ONE fragment set, read at the top of its own step -- the second
      // set's 48 registers are what stands between 180 VGPRs and the 168 of three waves per SIMD, and with three workgroups resident the
      // other waves' MFMAs cover this wave's LDS round trip.
      if (DD_CM_ONESET) read_frags(0, tap, s);
      else if (tap < 8 && !(DD_CM_EXP & 2)) read_frags(cur ^ 1, tap + 1, s + 1);
      if (!(DD_CM_EXP & 4)) fetch_b(min(s + 2, nsteps - 1));
      if (tap == 1 && !(DD_CM_EXP & 16)) fetch(min(chunk + 1, nchunks - 1));           // seven steps ahead of its use
      // (measured: pinning these reads in front of the MFMAs with a sched_barrier -- two live fragment sets, 220 registers -- is 6 %
      // SLOWER than letting the scheduler sink them behind the last use of the current set: the second workgroup of the CU covers
      // the LDS latency, and the barrier skew between four SIMDs does not shrink)
      // NP partial products per accumulator, the small ones first; consecutive MFMAs go to different accumulators
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 6 - NP; t < 6; ++t)         // NP = 6: all of them (fp32 accuracy); 3: x2w1 + x1w2 + x1w1 ("bf16x3"); 1: x1w1 (bf16 operands)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n)
            if (!(DD_CM_EXP & 8))
              acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, af[(DD_CM_EXP & 2) ? 0 : cur][m][PA[t]]),
                                                                  __builtin_bit_cast(bf8, bfr[(DD_CM_EXP & 2) ? 0 : cur][n][PB[t]]), acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);           // the stores and waits stay behind the MFMAs: they cover the loads' latency
      if (tap < 8) {
        if (!(DD_CM_EXP & 4)) store_b(s + 2);      // into the buffer of step s, whose fragments were read before the last barrier
        if (!(DD_CM_EXP & 1)) lds_barrier();
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

// ---- small images: flat pixel tiles + a split of the contraction --------------------------------------------------------------------
// The ResNet encoders' and motion decoders' deep levels (12 x 40 and 6 x 20 images, 256 / 512 channels: ~22 convolutions of the headline
// step) have M = B*H*W = 1 440 ... 11 520 output pixels against K = 9*C = 2 304 ... 4 608: 8 x 32-pixel tiles waste half of such an
// image, and 45 ... 180 M blocks do not fill 256 CUs.  Here a workgroup owns 256 CONSECUTIVE pixels of the flattened batch (wave w: the
// 32-pixel blocks 2w, 2w+1), 64 output channels, and a RANGE of the 16-channel chunks (blockIdx.z of `splits`): the window of
// 256 + 2 (W + 1) flat pixels of a chunk is split and staged as in conv_mfma_kernel (same 48-byte pixel stride, same LDS budget: W <= 40);
// tap (ty, tx) of a pixel is the window pixel (ty - 1) W + (tx - 1) further on -- or, where that neighbour lies outside the pixel's
// image (zero padding), a pixel slot of zeros at the end of the window: ONE select per fragment address instead of masking registers.
// Partial sums of the splits go to the workspace and conv_flat_fold_kernel adds them in split order (+ bias): bit-reproducible.
// pad = 1 only (the data gradient of a pad-1 convolution is a pad-1 convolution on the mirrored, transposed pack).
constexpr int FLAT_MT = 256;                                 // pixels per workgroup
constexpr int FLAT_ZERO = HN - 1;                            // window slot that holds zeros
constexpr int FLAT_MAX_W = (FLAT_ZERO - FLAT_MT) / 2 - 1;    // 40

template <int NB, int NP>
__global__ __launch_bounds__(NT, (DD_CM_SWZ && NB <= 2) ? 3 : 2) void conv_mfma_flat_kernel(const float* __restrict__ x, const uint4* __restrict__ pack, const float* __restrict__ bias,
                                                               int H, int W, int M, int k_in, int n_out, int splits, float* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* const s_b = smem + A_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p0 = (int)blockIdx.x * FLAT_MT, ntile = blockIdx.y, split = blockIdx.z;
  const int nchunks = (k_in + CK - 1) / CK;
  const int c_begin = (int)((long long)nchunks * split / splits), c_end = (int)((long long)nchunks * (split + 1) / splits);
  const int win = FLAT_MT + 2 * (W + 1), q0 = p0 - (W + 1);          // window: flat pixels q0 .. q0 + win - 1
  const char* pk = reinterpret_cast<const char*>(pack) + (size_t)ntile * nchunks * 9 * (3 * NB * FRAG);

  int g_off[PRE], l_off[PRE];
#pragma unroll
  for (int j = 0; j < PRE; ++j) {
    const int i = tid + j * NT, px = i >> 2, q = i & 3;
    const int fp = q0 + px;
    l_off[j] = px < win ? halo_addr(px, q >> 1) + (q & 1) * 8 : -1;
    g_off[j] = (px < win && fp >= 0 && fp < M) ? fp * k_in + q * 4 : -1;
  }
  if (tid < 9) {          // the zero pixel: three pieces x PSTR bytes
    const int pc = tid / 3, part = tid % 3;
    if (part * 16 < PSTR) *reinterpret_cast<uint4*>(smem + pc * A_PIECE + FLAT_ZERO * PSTR + part * 16) = make_uint4(0u, 0u, 0u, 0u);
  }
  float4 pre[PRE];
  auto fetch = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < PRE; ++j) {
      const int c0 = chunk * CK + (((tid + j * NT) & 3) << 2);
      const int off = (g_off[j] >= 0 && c0 < k_in) ? g_off[j] + chunk * CK : 0;
      pre[j] = *reinterpret_cast<const float4*>(x + off);
    }
  };
  auto stage = [&](int chunk) {
#pragma unroll
    for (int j = 0; j < PRE; ++j) {
      if (l_off[j] >= 0) {
        const int c0 = chunk * CK + (((tid + j * NT) & 3) << 2);
        const bool in = g_off[j] >= 0 && c0 < k_in;
        const float4 v = in ? pre[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned a1, a2, a3, b1, b2, b3;
        split2(v.x, v.y, a1, a2, a3);
        split2(v.z, v.w, b1, b2, b3);
        *reinterpret_cast<uint2*>(smem + l_off[j]) = make_uint2(a1, b1);
        if (pieces_of(NP) > 1) *reinterpret_cast<uint2*>(smem + A_PIECE + l_off[j]) = make_uint2(a2, b2);
        if (pieces_of(NP) > 2) *reinterpret_cast<uint2*>(smem + 2 * A_PIECE + l_off[j]) = make_uint2(a3, b3);
      }
    }
  };
  constexpr int BR = (3 * NB + 3) / 4;
  uint4 breg[BR];
#pragma unroll
  for (int r = 0; r < BR; ++r) breg[r] = make_uint4(0u, 0u, 0u, 0u);
  auto fetch_b = [&](int s) {
    const char* src = pk + (size_t)s * (3 * NB * FRAG) + lane * 16;
#pragma unroll
    for (int r = 0; r < BR; ++r) breg[r] = *reinterpret_cast<const uint4*>(src + (wave + 4 * r < 3 * NB ? wave + 4 * r : 0) * FRAG);
  };
  auto store_b = [&](int s) {
    unsigned char* dst = s_b + (s & 1) * b_buf_bytes<NB>() + lane * 16;
#pragma unroll
    for (int r = 0; r < BR; ++r) *reinterpret_cast<uint4*>(dst + (wave + 4 * r) * FRAG) = breg[r];
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  f16v acc[2][NB];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  // this lane's pixel of M block m: which of its nine neighbours lie inside its image (bit ty * 3 + tx), and its window address
  unsigned tapmask[2];
  int a_base[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int loc = (2 * wave + m) * 32 + (lane & 31), p = p0 + loc;
    const int rem = p % (H * W), yy = rem / W, xx = rem - yy * W;
    unsigned mk = 0u;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int Y = yy + t / 3 - 1, X = xx + t % 3 - 1;
      if (p < M && Y >= 0 && Y < H && X >= 0 && X < W) mk |= 1u << t;
    }
    tapmask[m] = mk;
    a_base[m] = (W + 1) + loc;             // window pixel of this lane's output pixel
  }
  const unsigned char* b_lane = s_b + lane * 16;

  uint4 af[DD_CM_ONESET ? 1 : 2][2][3], bfr[DD_CM_ONESET ? 1 : 2][NB][3];
  auto read_frags = [&](int set, int tap, int s) {
    const int toff = (tap / 3 - 1) * W + (tap % 3 - 1);                  // wave-uniform, in pixels
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int addr = halo_addr(((tapmask[m] >> tap) & 1u) ? a_base[m] + toff : FLAT_ZERO, lane >> 5);
#pragma unroll
      for (int pc = 0; pc < pieces_of(NP); ++pc) af[set][m][pc] = *reinterpret_cast<const uint4*>(smem + addr + pc * A_PIECE);
    }
    const unsigned char* bb = b_lane + (s & 1) * b_buf_bytes<NB>();
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int pc = 0; pc < pieces_of(NP); ++pc) bfr[set][n][pc] = *reinterpret_cast<const uint4*>(bb + (n * 3 + pc) * FRAG);
  };

  const int s_last = c_end * 9 - 1;
  fetch_b(c_begin * 9);
  fetch(c_begin);
  store_b(c_begin * 9);
  fetch_b(min(c_begin * 9 + 1, s_last));
  for (int chunk = c_begin; chunk < c_end; ++chunk) {
    lds_barrier();
    stage(chunk);
    store_b(chunk * 9 + 1);
    lds_barrier();
    if (!DD_CM_ONESET) read_frags(0, 0, chunk * 9);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int s = chunk * 9 + tap, cur = DD_CM_ONESET ? 0 : (tap & 1);
      if (DD_CM_ONESET) read_frags(0, tap, s);
      else if (tap < 8) read_frags(cur ^ 1, tap + 1, s + 1);
      fetch_b(min(s + 2, s_last));
      if (tap == 1) fetch(min(chunk + 1, c_end - 1));
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 6 - NP; t < 6; ++t)         // NP = 6: all of them (fp32 accuracy); 3: x2w1 + x1w2 + x1w1 ("bf16x3"); 1: x1w1 (bf16 operands)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NB; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, af[cur][m][PA[t]]), __builtin_bit_cast(bf8, bfr[cur][n][PB[t]]), acc[m][n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (tap < 8) {
        store_b(s + 2);
        lds_barrier();
      }
    }
  }

  // splits == 1: the result (+ bias); else this split's partial sums, [split][pixel][channel]
  float* dst = out + (splits > 1 ? (size_t)split * M * n_out : 0);
#pragma unroll
  for (int n = 0; n < NB; ++n) {
    const int co = (ntile * NB + n) * 32 + (lane & 31);
    if (co >= n_out) continue;
    const float bv = (splits == 1 && bias) ? bias[co] : 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int p = p0 + (2 * wave + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (p < M) dst[(size_t)p * n_out + co] = acc[m][n][r] + bv;
      }
    }
  }
}

// y = bias + the partial sums of the splits, in split order (float4 per thread; M * n_out is a multiple of 4)
__global__ __launch_bounds__(256) void conv_flat_fold_kernel(const float* __restrict__ part, const float* __restrict__ bias, int total4, int n_out, int splits,
                                                             float* __restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const float4* P = reinterpret_cast<const float4*>(part);
  float4 a = P[i];
  for (int s = 1; s < splits; ++s) {
    const float4 v = P[(size_t)s * total4 + i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (bias) {
    const int c = (i * 4) % n_out;
    a.x += bias[c]; a.y += bias[c + 1]; a.z += bias[c + 2]; a.w += bias[c + 3];
  }
  reinterpret_cast<float4*>(y)[i] = a;
}

static int flat_splits(int M, int k_in, int n_out) {
  const int wgs = ((M + FLAT_MT - 1) / FLAT_MT) * ((n_out + 63) / 64), nchunks = (k_in + CK - 1) / CK;
  const char* env = getenv("DD_FLAT_SPLITS");
  int s = env ? atoi(env) : (480 + wgs / 2) / wgs;            // about 480 workgroups in all (measured: profiles/r05_conv_flat.txt)
  if (s > nchunks / 2) s = nchunks / 2;                     // at least two chunks (18 steps) per workgroup
  return s < 1 ? 1 : s;
}

static size_t pack_bytes(int n_out, int k_in) { return (size_t)frags_of(n_out, k_in) * 3 * FRAG; }

template <int NB, int NP>
static int launch(const float* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, float* y, hipStream_t stream) {
  const int Ho = Hi + 2 * pad - 2, Wo = Wi + 2 * pad - 2;
  const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
  auto kern = conv_mfma_kernel<NB, NP>;
  static dd::LdsAttrOnce lds_attr;          // per instantiation and device (dd_attr.h)
  if (const int rc = lds_attr.ensure(reinterpret_cast<const void*>(kern), (int)(lds_bytes<NB>()))) return rc;
  dim3 grid(tiles_x * tiles_y, (n_out + 32 * NB - 1) / (32 * NB), B);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds_bytes<NB>(), stream, x, static_cast<const uint4*>(pack), bias, Hi, Wi, Ho, Wo, k_in, n_out, pad, tiles_x,
                     tiles_y, y);
  return (int)hipGetLastError();
}

template <int NP>
static int launch_blocks(const float* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, float* y, hipStream_t s) {
  switch (blocks_for(n_out)) {
    case 1: return launch<1, NP>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
    case 2: return launch<2, NP>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
    default: return launch<3, NP>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
  }
}

template <int NP>
static int launch_flat(const float* x, const void* pack, const float* bias, int H, int W, int M, int k_in, int n_out, int splits, float* out, hipStream_t s) {
  auto kern = conv_mfma_flat_kernel<2, NP>;
  static dd::LdsAttrOnce lds_attr;          // per instantiation and device (dd_attr.h)
  if (const int rc = lds_attr.ensure(reinterpret_cast<const void*>(kern), (int)(lds_bytes<2>()))) return rc;
  hipLaunchKernelGGL(kern, dim3((M + FLAT_MT - 1) / FLAT_MT, (n_out + 63) / 64, splits), dim3(NT), lds_bytes<2>(), s, x, static_cast<const uint4*>(pack), bias, H, W, M, k_in,
                     n_out, splits, out);
  return (int)hipGetLastError();
}

// ---- weight gradient ------------------------------------------------------------------------------------------------------------
// g_w[co][tap][ci] = sum over pixels of g[pixel][co] * x[pixel + tap - pad][ci]: M = co, N = ci, K = pixels.  The matrix operands want K
// contiguous per lane, the tensors have the channels contiguous: the tiles are TRANSPOSED on their way into LDS ([channel][pixel], two
// horizontally adjacent pixels per 32-bit store), split into the three bf16 pieces as in the forward.  A tap's shift along the row is a
// shift of the fragment by one or two bf16 values: formed in registers from one aligned read of ten pixels (gfx950 does serve
// unaligned ds_read_b128 -- scripts/microbench/lds_unaligned.hip: 144 against 84 cycles -- and the first version used it).
// A workgroup (4 waves) owns a 64 (co) x 64 (ci) block of the result for a contiguous range of 2 x 16-pixel tiles: 36 accumulators
// (2 co blocks x 2 ci blocks x 9 taps), nine per wave (one (ci block, tap row) group on both co blocks + half of another), kept in
// registers across the tiles; it writes ONE partial, conv_wgrad_fold_kernel adds the partials in a fixed order (no atomics: bit-reproducible).
namespace wg {
constexpr int TR = 2, TC = 16;                    // output-gradient pixels per tile: 2 rows x 16 columns = two K-16 steps
constexpr int XR = TR + 2, XC = TC + 2;           // halo of x
constexpr int XROW = 24;                          // halo row pitch in elements (48 bytes: every row starts 16-byte aligned)
constexpr int GSTR = TR * TC * 2 + 16;            // bytes per channel of the g tile (80 = 5 x 16: conflict-free fragment reads)
constexpr int XSTR = XR * XROW * 2 + 16;          // bytes per channel of the x tile (208 = 13 x 16)
constexpr int G_PIECE = 64 * GSTR, X_PIECE = 64 * XSTR;
constexpr int LDS = 3 * (G_PIECE + X_PIECE);      // 55 296 bytes: two workgroups per CU
constexpr int XITEMS = XR * (XC / 2) * 16;        // (row, pixel pair, channel quad) items of the x tile: 576
constexpr int XJ = (XITEMS + NT - 1) / NT;        // 3 per thread (the last round is partial)
constexpr int BLOCK = 64 * 9 * 64;                // floats of one partial
}  // namespace wg

template <int NP>
__global__ __launch_bounds__(NT, 2) void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g, int B, int Hi, int Wi, int Ho, int Wo,
                                                           int cin, int cout, int pad, int tiles_x, int tiles_y, float* __restrict__ partial) {
  using namespace wg;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* const s_g = smem;
  unsigned char* const s_x = smem + 3 * G_PIECE;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = blockIdx.x, nks = gridDim.x, ci0 = blockIdx.y * 64, co0 = blockIdx.z * 64;
  const int total = B * tiles_y * tiles_x;
  const int t_begin = (int)((long long)total * ks / nks), t_end = (int)((long long)total * (ks + 1) / nks);

  // staging items of this thread (the same for every tile)
  const int gq = tid >> 4, grow = (tid >> 3) & 1, gpair = tid & 7;
  const int g_lds = (4 * gq) * GSTR + (grow * TC + 2 * gpair) * 2;
  int xq[XJ], xrow[XJ], xpair[XJ], x_lds[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int item = tid + j * NT;
    xq[j] = item / (XR * (XC / 2));
    const int rem = item - xq[j] * (XR * (XC / 2));
    xrow[j] = rem / (XC / 2);
    xpair[j] = rem - xrow[j] * (XC / 2);
    x_lds[j] = item < XITEMS ? (4 * xq[j]) * XSTR + (xrow[j] * XROW + 2 * xpair[j]) * 2 : -1;
  }
  const bool g_ch = co0 + 4 * gq < cout;

  float4 pg[2], px[XJ][2];
  auto fetch = [&](int t) {
    const int b = t / (tiles_y * tiles_x), r = t - b * (tiles_y * tiles_x);
    const int Y0 = (r / tiles_x) * TR, X0 = (r - (r / tiles_x) * tiles_x) * TC;
    {
      const int Y = Y0 + grow, X = X0 + 2 * gpair;
      const float* src = g + (((size_t)b * Ho + Y) * Wo + X) * cout + co0 + 4 * gq;
      const bool ok = g_ch && Y < Ho;
      pg[0] = (ok && X < Wo) ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
      pg[1] = (ok && X + 1 < Wo) ? *reinterpret_cast<const float4*>(src + cout) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int Y = Y0 - pad + xrow[j], X = X0 - pad + 2 * xpair[j];
      const bool ok = x_lds[j] >= 0 && ci0 + 4 * xq[j] < cin && Y >= 0 && Y < Hi;
      const float* src = x + (((ptrdiff_t)b * Hi + Y) * Wi + X) * cin + ci0 + 4 * xq[j];
      px[j][0] = (ok && X >= 0 && X < Wi) ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
      px[j][1] = (ok && X + 1 >= 0 && X + 1 < Wi) ? *reinterpret_cast<const float4*>(src + cin) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto put = [&](unsigned char* plane, int piece_bytes, int stride, int off, const float4& a, const float4& b) {
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned p1, p2, p3;
      split2(av[c], bv[c], p1, p2, p3);          // {pixel X, pixel X + 1} of one channel: one 32-bit store per piece
      *reinterpret_cast<unsigned*>(plane + off + c * stride) = p1;
      if (pieces_of(NP) > 1) *reinterpret_cast<unsigned*>(plane + piece_bytes + off + c * stride) = p2;
      if (pieces_of(NP) > 2) *reinterpret_cast<unsigned*>(plane + 2 * piece_bytes + off + c * stride) = p3;
    }
  };
  auto stage = [&]() {
    put(s_g, G_PIECE, GSTR, g_lds, pg[0], pg[1]);
#pragma unroll
    for (int j = 0; j < XJ; ++j)
      if (x_lds[j] >= 0) put(s_x, X_PIECE, XSTR, x_lds[j], px[j][0], px[j][1]);
  };

  // This wave's nine accumulators.  A GROUP is (ci block nb, tap row ty): its three taps tx = 0, 1, 2 read the same 10 pixels of a row
  // of x shifted by 0 / 1 / 2 -- one aligned 16-byte read + one 4-byte read give all three fragments (tx = 1 through four
  // v_alignbit_b32, tx = 2 is a renaming), where three separate reads were two unaligned ones (the LDS pipe was 72 % busy).
  // Groups 0..5 = (nb 0, ty 0..2), (nb 1, ty 0..2).  Wave w owns group w on both co blocks (six accumulators) and half of group
  // 4 + (w >> 1): of its six (tx, co block) pairs, w even takes (0,0) (0,1) (1,0), w odd (2,0) (2,1) (1,1).
  // With 32 or fewer input channels in this ci group only the groups of nb 0 exist.
  const int ngroups = (cin - ci0 > 32) ? 6 : 3;
  const int g_full = wave, g_half = 4 + (wave >> 1), half = wave & 1;
  const bool full_on = g_full < ngroups, half_on = g_half < ngroups;
  auto group_off = [&](int g) { return (g / 3) * 32 * XSTR + (g % 3) * (XROW * 2); };
  const int off_full = group_off(g_full), off_half = group_off(g_half);
  f16v acc[9];                             // [tx * 2 + m] of the full group, then the three pairs of the half group
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

  const unsigned char* a_lane = s_g + (lane & 31) * GSTR + (lane >> 5) * 16;
  const unsigned char* b_lane = s_x + (lane & 31) * XSTR + (lane >> 5) * 16;
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
  // fragment of tap column tx from the ten pixels w0 (eight) | w1 (two) of one piece
  auto shifted = [](const uint4& w0, unsigned w1, int tx) -> uint4 {
    if (tx == 0) return w0;
    if (tx == 2) return make_uint4(w0.y, w0.z, w0.w, w1);
    return make_uint4(__builtin_amdgcn_alignbit(w0.y, w0.x, 16), __builtin_amdgcn_alignbit(w0.z, w0.y, 16), __builtin_amdgcn_alignbit(w0.w, w0.z, 16),
                      __builtin_amdgcn_alignbit(w1, w0.w, 16));
  };

  if (t_begin < t_end) fetch(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();                      // the previous tile's fragment reads are done
    stage();
    __syncthreads();
    if (t + 1 < t_end) fetch(t + 1);      // lands under this tile's MFMAs
#pragma unroll
    for (int row = 0; row < TR; ++row) {
      uint4 af[2][3];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pc = 0; pc < pieces_of(NP); ++pc) af[m][pc] = *reinterpret_cast<const uint4*>(a_lane + m * 32 * GSTR + row * (TC * 2) + pc * G_PIECE);
      uint4 w0[3];
      unsigned w1[3];
      if (full_on) {                      // wave-uniform
#pragma unroll
        for (int pc = 0; pc < pieces_of(NP); ++pc) {
          const unsigned char* src = b_lane + off_full + row * (XROW * 2) + pc * X_PIECE;
          w0[pc] = *reinterpret_cast<const uint4*>(src);
          w1[pc] = *reinterpret_cast<const unsigned*>(src + 16);
        }
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          uint4 bf[3];
#pragma unroll
          for (int pc = 0; pc < pieces_of(NP); ++pc) bf[pc] = shifted(w0[pc], w1[pc], tx);
#pragma unroll
          for (int tt = 6 - NP; tt < 6; ++tt)
#pragma unroll
            for (int m = 0; m < 2; ++m)
              acc[tx * 2 + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, af[m][PA[tt]]), __builtin_bit_cast(bf8, bf[PB[tt]]),
                                                                        acc[tx * 2 + m], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);      // one tap column's fragments at a time (formed ahead they cost 24 registers: spills)
        }
      }
      if (half_on) {
#pragma unroll
        for (int pc = 0; pc < pieces_of(NP); ++pc) {
          const unsigned char* src = b_lane + off_half + row * (XROW * 2) + pc * X_PIECE;
          w0[pc] = *reinterpret_cast<const uint4*>(src);
          w1[pc] = *reinterpret_cast<const unsigned*>(src + 16);
        }
        // pairs (tx, m): half 0 -> (0,0) (0,1) (1,0); half 1 -> (1,1) (2,0) (2,1): the tap column with both co blocks, then the single
        {
          const int txd = half == 0 ? 0 : 2;                 // wave-uniform
          uint4 bf[3];
#pragma unroll
          for (int pc = 0; pc < pieces_of(NP); ++pc) bf[pc] = txd == 0 ? shifted(w0[pc], w1[pc], 0) : shifted(w0[pc], w1[pc], 2);
#pragma unroll
          for (int tt = 6 - NP; tt < 6; ++tt)
#pragma unroll
            for (int m = 0; m < 2; ++m)
              acc[6 + m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, af[m][PA[tt]]), __builtin_bit_cast(bf8, bf[PB[tt]]), acc[6 + m], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          uint4 am[3];
#pragma unroll
          for (int pc = 0; pc < pieces_of(NP); ++pc) {
            bf[pc] = shifted(w0[pc], w1[pc], 1);
            am[pc] = half == 0 ? af[0][pc] : af[1][pc];
          }
#pragma unroll
          for (int tt = 6 - NP; tt < 6; ++tt)
            acc[8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, am[PA[tt]]), __builtin_bit_cast(bf8, bf[PB[tt]]), acc[8], 0, 0, 0);
        }
      }
    }
  }

  // the partial of this workgroup: [co 64][tap 9][ci 64]; C layout: column (ci) = lane & 31, row (co) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* P = partial + (((size_t)blockIdx.z * gridDim.y + blockIdx.y) * nks + ks) * BLOCK;
  auto put_acc = [&](const f16v& A, int g, int tx, int m) {
    const int nb = g / 3, tap = (g % 3) * 3 + tx;
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
  }
  if (half_on) {
    put_acc(acc[6], g_half, half == 0 ? 0 : 2, 0);
    put_acc(acc[7], g_half, half == 0 ? 0 : 2, 1);
    put_acc(acc[8], g_half, 1, half);
  }
}

// g_weight (cout,3,3,cin) = sum over the nks partials of every (co group, ci group), in order
// A workgroup takes 64 consecutive elements of the result; its four waves take the partials k = wave, wave + 4, ... (eight loads in
// flight per lane) and meet in LDS in wave order.  (The first version gave one thread the whole chain of up to 512 partials: 144
// workgroups of dependent loads, 32 us for 75 MB at 64 x 64 channels.)
__global__ __launch_bounds__(256) void conv_wgrad_fold_kernel(const float* __restrict__ partial, int nks, int cin, int cout, int ci_groups, float* __restrict__ gw) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;                  // (co, tap, ci) of the result
  const bool in = idx < cout * 9 * cin;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (in) {
    const int ci = idx % cin, tap = (idx / cin) % 9, co = idx / (9 * cin);
    const int z = co >> 6, y = ci >> 6;
    const float* P = partial + ((size_t)z * ci_groups + y) * nks * wg::BLOCK + ((co & 63) * 9 + tap) * 64 + (ci & 63);
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
  int n = 512 / groups;                        // two workgroups per CU in all (measured: 128 / 192 / 256 / 384 in all are slower at every
  if (n > tiles / 11) n = tiles / 11;          // shape but 12 x 64 x 64 x 48 x 160, which this bound sends to 256: at least eleven tiles each)
  return n < 1 ? 1 : n;
}

}  // namespace cm
}  // namespace dd

extern "C" int dd_conv3x3_mfma_supported(int cin, int cout) {
  // both counts in fours: the data gradient is the same kernel with the roles swapped (k_in = cout)
  return (cin >= 16 && cout >= 16 && cin % 4 == 0 && cout % 4 == 0 && cin <= 1024 && cout <= 1024) ? 1 : 0;
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

extern "C" int dd_conv3x3_mfma_pack_many_job_words(void) { return dd::cm::PACK_JOB_WORDS; }

extern "C" int dd_conv3x3_mfma_pack_many_blocks(int cout, int cin, int want_fwd, int want_bwd_data) {
  using namespace dd::cm;
  if (cout < 1 || cin < 1) return 0;
  return ((want_fwd ? frags_of(cout, cin) : 0) + (want_bwd_data ? frags_of(cin, cout) : 0) + 3) / 4;      // four fragments per 256-thread workgroup
}

extern "C" int dd_conv3x3_mfma_pack_many(const long long* jobs, const int* block_job, int n_blocks, void* stream) {
  using namespace dd::cm;
  if (!jobs || !block_job || n_blocks < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(conv_mfma_pack_many_kernel, dim3(n_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), jobs, block_job);
  return (int)hipGetLastError();
}

extern "C" int dd_conv3x3_mfma_n(const float* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, int products,
                                 float* y, void* stream) {
  using namespace dd::cm;
  if (!x || !pack || !y || B < 1 || pad < 0 || pad > 2 || Hi + 2 * pad < 3 || Wi + 2 * pad < 3 || k_in % 4 || k_in < 4 || n_out < 1) return (int)hipErrorInvalidValue;
  if ((size_t)Hi * Wi * k_in >= (1ull << 31) || (reinterpret_cast<unsigned long long>(x) & 15ull)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (products) {
    case 6: return launch_blocks<6>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
    case 3: return launch_blocks<3>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
    case 1: return launch_blocks<1>(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, y, s);
    default: return (int)hipErrorInvalidValue;
  }
}

extern "C" int dd_conv3x3_mfma(const float* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, float* y,
                               void* stream) {
  return dd_conv3x3_mfma_n(x, pack, bias, B, Hi, Wi, k_in, n_out, pad, 6, y, stream);
}

extern "C" int dd_conv3x3_mfma_flat_supported(int B, int H, int W, int k_in, int n_out) {
  using namespace dd::cm;
  // (n_out: the channel counts for which dd_conv3x3_mfma_pack lays out two 32-channel blocks per tile, blocks_for() == 2)
  return (B >= 1 && H >= 1 && W >= 1 && W <= FLAT_MAX_W && k_in >= 32 && k_in % 4 == 0 && k_in <= 1024 && n_out % 4 == 0 && n_out <= 1024 &&
          blocks_for(n_out) == 2 && n_out > 32 &&
          (size_t)B * H * W * (size_t)(k_in > n_out ? k_in : n_out) < (1ull << 31)) ? 1 : 0;
}

extern "C" size_t dd_conv3x3_mfma_flat_workspace_bytes(int B, int H, int W, int k_in, int n_out) {
  using namespace dd::cm;
  const int M = B * H * W, s = flat_splits(M, k_in, n_out);
  return s > 1 ? (size_t)s * M * n_out * sizeof(float) : 16;
}

extern "C" int dd_conv3x3_mfma_flat_n(const float* x, const void* pack, const float* bias, int B, int H, int W, int k_in, int n_out, int products, float* y,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  using namespace dd::cm;
  if (!x || !pack || !y || !dd_conv3x3_mfma_flat_supported(B, H, W, k_in, n_out) || (reinterpret_cast<unsigned long long>(x) & 15ull) ||
      (reinterpret_cast<unsigned long long>(y) & 15ull) || (products != 6 && products != 3 && products != 1))
    return (int)hipErrorInvalidValue;
  const int M = B * H * W, splits = flat_splits(M, k_in, n_out);
  if (splits > 1 && (!workspace || workspace_bytes < dd_conv3x3_mfma_flat_workspace_bytes(B, H, W, k_in, n_out) || (reinterpret_cast<unsigned long long>(workspace) & 15ull)))
    return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* out = splits > 1 ? static_cast<float*>(workspace) : y;
  const int rc = products == 6 ? launch_flat<6>(x, pack, bias, H, W, M, k_in, n_out, splits, out, s)
               : products == 3 ? launch_flat<3>(x, pack, bias, H, W, M, k_in, n_out, splits, out, s)
                               : launch_flat<1>(x, pack, bias, H, W, M, k_in, n_out, splits, out, s);
  if (rc != 0 || splits == 1) return rc;
  const int total4 = M * n_out / 4;
  hipLaunchKernelGGL(conv_flat_fold_kernel, dim3((total4 + 255) / 256), dim3(256), 0, s, static_cast<const float*>(workspace), bias, total4, n_out, splits, y);
  return (int)hipGetLastError();
}

extern "C" int dd_conv3x3_mfma_flat(const float* x, const void* pack, const float* bias, int B, int H, int W, int k_in, int n_out, float* y, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return dd_conv3x3_mfma_flat_n(x, pack, bias, B, H, W, k_in, n_out, 6, y, workspace, workspace_bytes, stream);
}

extern "C" size_t dd_conv3x3_mfma_wgrad_workspace_bytes(int B, int Ho, int Wo, int cin, int cout) {
  const size_t groups = (size_t)((cin + 63) / 64) * ((cout + 63) / 64);
  return groups * dd::cm::wgrad_splits(B, Ho, Wo, cin, cout) * dd::cm::wg::BLOCK * sizeof(float);
}

extern "C" int dd_conv3x3_mfma_bwd_weight_n(const float* x, const float* g_out, int B, int Hi, int Wi, int cin, int cout, int pad, int products, float* g_weight,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  using namespace dd::cm;
  if (!x || !g_out || !g_weight || !workspace || B < 1 || pad < 0 || pad > 1 || cin % 4 || cout % 4 || cin < 4 || cout < 4) return (int)hipErrorInvalidValue;
  if (products != 6 && products != 3 && products != 1) return (int)hipErrorInvalidValue;
  const int Ho = Hi + 2 * pad - 2, Wo = Wi + 2 * pad - 2;
  if (Ho < 1 || Wo < 1 || workspace_bytes < dd_conv3x3_mfma_wgrad_workspace_bytes(B, Ho, Wo, cin, cout)) return (int)hipErrorInvalidValue;
  if ((size_t)B * Hi * Wi * cin >= (1ull << 31) || (size_t)B * Ho * Wo * cout >= (1ull << 31)) return (int)hipErrorInvalidValue;
  if ((reinterpret_cast<unsigned long long>(x) | reinterpret_cast<unsigned long long>(g_out)) & 15ull) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int tiles_x = (Wo + wg::TC - 1) / wg::TC, tiles_y = (Ho + wg::TR - 1) / wg::TR;
  const int nks = wgrad_splits(B, Ho, Wo, cin, cout), ci_groups = (cin + 63) / 64, co_groups = (cout + 63) / 64;
  const dim3 grid(nks, ci_groups, co_groups);
  float* part = static_cast<float*>(workspace);
  if (products == 6)
    hipLaunchKernelGGL(conv_wgrad_kernel<6>, grid, dim3(NT), wg::LDS, s, x, g_out, B, Hi, Wi, Ho, Wo, cin, cout, pad, tiles_x, tiles_y, part);
  else if (products == 3)
    hipLaunchKernelGGL(conv_wgrad_kernel<3>, grid, dim3(NT), wg::LDS, s, x, g_out, B, Hi, Wi, Ho, Wo, cin, cout, pad, tiles_x, tiles_y, part);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<1>, grid, dim3(NT), wg::LDS, s, x, g_out, B, Hi, Wi, Ho, Wo, cin, cout, pad, tiles_x, tiles_y, part);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(conv_wgrad_fold_kernel, dim3((cout * 9 * cin + 63) / 64), dim3(256), 0, s, static_cast<const float*>(workspace), nks, cin, cout, ci_groups,
                     g_weight);
  return (int)hipGetLastError();
}

extern "C" int dd_conv3x3_mfma_bwd_weight(const float* x, const float* g_out, int B, int Hi, int Wi, int cin, int cout, int pad, float* g_weight, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  return dd_conv3x3_mfma_bwd_weight_n(x, g_out, B, Hi, Wi, cin, cout, pad, 6, g_weight, workspace, workspace_bytes, stream);
}
