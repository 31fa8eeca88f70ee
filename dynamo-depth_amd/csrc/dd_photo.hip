// dd_photo.hip -- fused view-synthesis photometric loss, forward AND backward in one pass (gfx950).
//
// One workgroup owns a TH x TW tile of target pixels of one image at one scale and does, without
// touching HBM for any intermediate:
//   A  up-sample disp/flow/mask -> depth -> back-project -> (flow compose) -> rigid transform ->
//      project -> bilinear border warp of both source frames, for the tile plus a 2-pixel halo
//      (the halo pixels' warps are recomputed, they belong to the neighbouring tiles);
//      warped colours go to LDS, the owners keep d(colour)/d(u,v) and the geometry in registers;
//   B  SSIM(3x3, reflect) + L1 for tile + 1-pixel halo, min over frames (+ identity/automask),
//      loss accumulation, and the three per-channel coefficients of d(loss)/d(warped colour);
//   C  adjoint of the reflect-padded box filter (gather over the 9 windows that contain a pixel),
//      chain rule through warp / projection / pose / flow composition / depth, adjoint of the
//      bilinear up-sampling as a separable gather in LDS -> the tile's low-res footprint goes to the
//      workspace with plain stores (photo_combine_kernel adds the overlapping footprints in a fixed order);
//   R  block reduction of the loss sums and of d(loss)/dT through an LDS transpose -> one record per block.
// photo_finalize_kernel folds the per-block records deterministically.  No float atomics anywhere.
//
// The kernel is VALU-issue bound, so the arithmetic is laid out for the packed fp32 pipe: the two source frames of a
// pixel are the two elements of an `f2` (dd_pair.h) through every stage -- geometry, the four bilinear taps, the SSIM
// window sums, the adjoint gather, the pose-gradient outer products -- and their warped colours sit interleaved in LDS so
// that one ds_read_b64 fetches both.  The reflect padding of the SSIM window is materialised in LDS (the row/column just
// outside the image holds the mirrored values), so every window is nine constant offsets from the centre.
//
// Replaces Trainer.generate_images_pred + the photometric part of Trainer.compute_losses and their
// autograd (reference Trainer.py:215-352,384-386,413-423; tools.py:191-257,291-298) -- thousands of
// ATen launches per step in the reference.  The arithmetic is dd_math.h / dd_pair.h.
#include <hip/hip_runtime.h>
#include <cstring>

#include "../../include/dynamo_hip.h"
#include "dd_attr.h"
#include "dd_fuse.h"
#include "dd_pair.h"

#ifdef DD_EXP_NOBARRIER     // timing experiment only (wrong results): what do the workgroup barriers cost?
#define __syncthreads() do { } while (0)
#endif

namespace dd {

// TH x TW tiles (dd_fuse.h: 16x32 = 512 threads, ~77 KB LDS, two workgroups per CU)
constexpr int NT = TH * TW;       // one thread per target pixel
constexpr int RH = TH + 4, RW = TW + 4;       // region with 2-pixel halo (warped colours, target)
constexpr int R2N = RH * RW;
constexpr int CH_ = TH + 2, CW_ = TW + 2;     // centres with 1-pixel halo (SSIM, selection, coefficients)
constexpr int R1N = CH_ * CW_;
constexpr int RING = R2N - TH * TW;           // halo pixels of the region
constexpr int CRING = R1N - TH * TW;          // halo centres
constexpr int FPW_MAX = TW / 2 + 2;                          // widest low-res footprint of a tile (scale 1)

constexpr int LRN_MAX = (TH / 2) * (TW / 2);                  // low-res pixels inside a tile at scale >= 1
constexpr int NWAVES = NT / 64;
constexpr int NRED = 30;          // photo, n_warp, cons[2], delta[2], gT[2][12]  (+ NSMOOTH smoothness sums with the fused smoothness)

constexpr int LOWH = RH / 2 + 2, LOWW = RW / 2 + 2;           // staged low-res region (scale >= 1) incl. halo taps
constexpr int LOWN = LOWH * LOWW;
static_assert(RING <= NT && CRING <= NT, "one pass over the halo ring");
#ifndef DD_EXP_TAPS
#define DD_EXP_TAPS 0
#endif
#ifndef DD_RING_LDS
#define DD_RING_LDS 0            // 1 (experiment, round 5): the halo ring's source taps go global -> LDS directly and share the owners' memory
#endif                           // round trip in stage A -- correct (same values) and SLOWER: profiles/r05_photo_ring_through_lds.txt
constexpr int RING_WAVES = (RING + 63) / 64;
constexpr int RING_STAGE_OFF = (9 * LOWN + 63) / 64 * 64;                    // floats: behind the staged low-res inputs
constexpr int RING_STAGE_FLOATS = (24 + 8) * RING_WAVES * 64;                // 24 taps + four f2 weights per ring pixel
static_assert(TH % 8 == 0 && TW % 16 == 0 && NT % 64 == 0 && NT <= 1024, "tile shape");
static_assert(LOWN <= 256 && NT >= 512, "low-res staging takes two planes per pass, 256 threads each");
static_assert(NWAVES * 4 >= NRED && NWAVES * 5 >= NRED + NSMOOTH && NWAVES * 5 == DD_PARTIAL_STRIDE && REC_SMOOTH == NRED,
              "four (five with the smoothness sums) values of the block record per wave");

// wave64 sum on the VALU with DPP (quad swaps, row rotations, row broadcasts) instead of six ds_bpermute round trips
// through the LDS pipe per value; the total is read from lane 63 and returned uniformly.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}

__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add<0xb1, 0xf>(v);     // quad_perm:[1,0,3,2]
  v = dpp_add<0x4e, 0xf>(v);     // quad_perm:[2,3,0,1]
  v = dpp_add<0x124, 0xf>(v);    // row_ror:4
  v = dpp_add<0x128, 0xf>(v);    // row_ror:8   -> every lane holds its row-of-16 sum
  v = dpp_add<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// load through a wave-uniform base pointer and a 32-bit BYTE offset: global_load_dword v, v_off, s[base:base+1] -- no
// 64-bit address arithmetic on the VALU (the offsets inside one image stay far below 4 GB)
__device__ __forceinline__ float ldg(const float* base, unsigned byte_off) {
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// Keeps the instruction stream in source order across this point: the asm memory clobber pins the loads at instruction
// selection, sched_barrier pins everything in the machine scheduler.  Used to cap the number of LDS values in flight where
// the default schedule (all loads first) would spill the per-pixel state.
#ifndef DD_NO_ORDER
#define DD_ORDER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DD_ORDER() do { } while (0)
#endif
// ... and an accumulator named here has to be complete at this point (its arithmetic cannot sink below the loads that follow)
#define DD_PIN(x) asm volatile("" : "+v"(x))

// v[lane] + v[lane + 1] inside a row of 16 lanes (row_shl:1, out-of-row reads give 0)
__device__ __forceinline__ float add_right_neighbour(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true);
  return v + __builtin_bit_cast(float, moved);
}

// -DDD_ISA_MARKS: comment markers in the ISA between the stages (scripts/isa_stage_count.py weighs the static counts)
#ifdef DD_ISA_MARKS
#define DD_ISA(name) asm volatile("; DDMARK " name ::: "memory")
#else
#define DD_ISA(name) do { } while (0)
#endif

#ifdef DD_STAGE_TIMING
__device__ unsigned long long g_stage_cycles[8];
#define DD_STAGE_MARK(i)                                              \
  do {                                                                \
    if (threadIdx.x == 0) {                                           \
      const unsigned long long t_now = clock64();                     \
      atomicAdd(&g_stage_cycles[i], t_now - t_prev);                  \
      t_prev = t_now;                                                 \
    }                                                                 \
  } while (0)
#else
#define DD_STAGE_MARK(i) do { } while (0)
#endif

// SSIM + L1 of both frames at the centre whose region index is `li`, from the LDS planes (x: interleaved frame pairs,
// y: target).  The reflect padding is already in the planes: nine constant offsets.  WITH_GRAD: the backward coefficients of both
// frames (gscale x d ssim/d mean_x, 2 d ssim/d mean_xx, d ssim/d mean_xy per channel) go straight to the LDS planes at cf.
typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));   // three adjacent floats at a 4-byte-aligned address (global_load_dwordx3)
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));   // two adjacent floats at a 4-byte-aligned LDS address (ds_read2_b32)

template <bool WITH_GRAD>
__device__ __forceinline__ f2 rho_pair(const f2* __restrict__ s_x, const float* __restrict__ s_y, int li, float alpha, float gscale, f2* cf) {
  f2 ssum = sp2(0.f), l1 = sp2(0.f);
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const f2* xp = s_x + ch * R2N + li;
    const float* yp = s_y + ch * R2N + li;
    f2 sx = sp2(0.f), sxx = sp2(0.f), sxy = sp2(0.f);
    float sy = 0.f, syy = 0.f;             // the target's sums take the taps in the SAME order as the frames' (see ssim_value2:
#pragma unroll                             // a window with x == y bit for bit must give sxx == sxy == syy bit for bit)
    for (int j = -1; j <= 1; ++j) {
      const f2 ya = *reinterpret_cast<const f2u*>(yp + j * RW - 1);
      const float yb = yp[j * RW + 1];
      const f2 xa = xp[j * RW - 1], xb = xp[j * RW], xc = xp[j * RW + 1];
      sx += xa; sxx += xa * xa; sxy += xa * sp2(ya[0]);
      sx += xb; sxx += xb * xb; sxy += xb * sp2(ya[1]);
      sx += xc; sxx += xc * xc; sxy += xc * sp2(yb);
      sy += ya[0]; syy += ya[0] * ya[0];
      sy += ya[1]; syy += ya[1] * ya[1];
      sy += yb; syy += yb * yb;
    }
    SsimGrad2 sg;
    ssum += ssim_value2<WITH_GRAD>(sx, sxx, sxy, sy, syy, gscale, sg);
    if (WITH_GRAD) {            // cf: this centre's slot of the [9][R1N] coefficient planes
      cf[(ch * 3 + 0) * R1N] = sg.dmu;
      cf[(ch * 3 + 1) * R1N] = sg.dxx2;
      cf[(ch * 3 + 2) * R1N] = sg.dxy;
    }
    l1 += abs2(sp2(yp[0]) - xp[0]);
    DD_PIN(ssum);
    DD_PIN(l1);
    DD_ORDER();     // one channel at a time: interleaving the three keeps ~60 transient VGPRs alive and spills the owner state
  }
  return sp2(alpha) * (ssum * sp2(1.f / 3.f)) + sp2(1.f - alpha) * (l1 * sp2(1.f / 3.f));
}

// sign(x) * e for 0 <= e <= 1 (an edge weight): x * 2^126 * 2^126 saturated to [-e, e] -- sign2's arithmetic with the weight as the
// clamp bound (one v_med3 per value instead of a clamp to +-1 and a multiply); sign(0) = 0, e = 0 gives 0
__device__ __forceinline__ f2 signed_weight2(f2 x, f2 e) {
  const f2 big = (x * sp2(8.507059173023462e37f)) * sp2(8.507059173023462e37f);
  return mk2(__builtin_amdgcn_fmed3f(big[0], -e[0], e[0]), __builtin_amdgcn_fmed3f(big[1], -e[1], e[1]));
}

// does full-res coordinate c take part in the align_corners=False bilinear down-sampling by 2^shift?
__device__ __forceinline__ bool down_tap(int c, int shift) {
  const int blk = 1 << shift, r = c & (blk - 1);
  return (r == (blk >> 1) - 1) || (r == (blk >> 1));
}

// LDS.  Regions are reused across stages (the barriers in the kernel body separate the lifetimes):
//   pred+tgt  : warped colours / target colours (stages 0..C1)  ->  per-pixel gradient planes G (stage C2, scale >= 1)
//   coef      : backward coefficients (stages B..C1)             ->  x-reduced gradient planes Hx (stage C3)
//   everything: the transposed block reduction (stage R)
struct LdsLayout {
  f2 pred[3 * R2N];               // warped source colours {first, second frame} (source copies during the automask pre-pass)
  float tgt[3 * R2N];             // target colours
  union {
    f2 coef[9 * R1N];             // [3 channels x (A, B, C)][centre]: weighted backward coefficients of BOTH frames (the gather
                                  // applies the selection) -- written channel by channel, nothing of it waits in registers
    float low[9 * LOWN];          // stage 0..A only: staged low-res inputs (scale >= 1): disp, flow[2][3], mask[2]
                                  // (shared tensors: disp | {flow x, flow y} | {flow z, mask})
  };
  union {
    f2 lr[2 * 5 * LRN_MAX];       // FLOW_MASK: low-res residual flow (3) + grid difference (2), frame pairs; two row slots (upper / lower tap row)
    float idmin[R1N];             // automask: min over frames of the identity reprojection loss (+noise)
  };
  signed char sel[(R1N + 15) / 16 * 16];   // selected frame per centre (-1: identity won / outside the image)
};
static_assert(9 * TH * TW * sizeof(float) <= sizeof(f2) * 3 * R2N + sizeof(float) * 3 * R2N, "gradient planes must fit into pred+tgt");
static_assert((RING_STAGE_OFF + RING_STAGE_FLOATS) * sizeof(float) <= sizeof(f2) * 9 * R1N, "the ring's tap staging must fit into the coefficient planes");
static_assert(9 * TH * FPW_MAX * sizeof(float) <= sizeof(f2) * 9 * R1N, "x-reduced planes must fit into coef");
static_assert(sizeof(LdsLayout) >= NT * NWAVES * 4 * sizeof(float), "the transposed reduction spans the whole layout");
static_assert(sizeof(LdsLayout) <= 80 * 1024, "two workgroups per CU");
// dynamic LDS of a launch: the layout; the transposed reduction of the fused-smoothness instantiations parks 37 planes of NT floats
// in it (74 KB) -- NOT DD_PARTIAL_STRIDE planes: exactly 80 KB per workgroup left one workgroup per CU (measured: +13 % kernel time)
constexpr size_t lds_bytes(bool) { return sizeof(LdsLayout); }
static_assert(sizeof(LdsLayout) >= NT * (NRED + NSMOOTH) * sizeof(float), "the transposed reduction with the smoothness sums spans the layout");

// bilinear up-sampling taps of one full-res pixel, as offsets into a staged LOWH x LOWW region
struct LowTap {
  int o00, o01, o10, o11;
  float wx0, wx1, wy0, wy1;
};

__device__ __forceinline__ float low_eval(const float* __restrict__ plane, const LowTap& t) {
  // same association as ATen: wy0*(wx0*a + wx1*b) + wy1*(wx0*c + wx1*d)
  return t.wy0 * (t.wx0 * plane[t.o00] + t.wx1 * plane[t.o01]) + t.wy1 * (t.wx0 * plane[t.o10] + t.wx1 * plane[t.o11]);
}

__device__ __forceinline__ f2 low_eval2(const f2* __restrict__ plane, const LowTap& t) {
  return sp2(t.wy0) * (sp2(t.wx0) * plane[t.o00] + sp2(t.wx1) * plane[t.o01]) + sp2(t.wy1) * (sp2(t.wx0) * plane[t.o10] + sp2(t.wx1) * plane[t.o11]);
}

// gradient / staged-plane channels: disp | flow (3 per frame, or 3 when both frames read one tensor) | mask (likewise)
template <int MODE, bool SHARED>
struct Channels {
  static constexpr int FLOW = MODE == MODE_RIGID ? 0 : (SHARED ? 3 : 6);
  static constexpr int MASK = MODE == MODE_FLOW_MASK ? (SHARED ? 1 : 2) : 0;
  static constexpr int N = 1 + FLOW + MASK;
  static constexpr int MASK0 = 1 + FLOW;      // first mask channel
};

// The auto-mask's identity reprojection loss (Trainer.py:325-340: the photometric loss of the UN-warped source frames against the
// target) does not depend on the scale: one pass per step writes it for both frames, the tile kernel's three scale passes read two
// floats per centre.  Same tile geometry, same LDS planes (sources interleaved as frame pairs, reflect padding materialised) and the
// same rho_pair arithmetic as the tile kernel's own evaluation of round 4: bit-identical values.  out: (B,2,H,W).
__global__ __launch_bounds__(NT) void photo_identity_kernel(const float* __restrict__ target, const float* __restrict__ src0, const float* __restrict__ src1,
                                                           int H, int W, float alpha, float* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  f2* const s_pred = reinterpret_cast<f2*>(smem_raw);                                       // [3][R2N]
  float* const s_tgt = reinterpret_cast<float*>(smem_raw + sizeof(f2) * 3 * R2N);          // [3][R2N]
  const int tid = threadIdx.x, b = blockIdx.y, N = H * W;
  const int tiles_x = (W + TW - 1) / TW;
  int tile = blockIdx.x;
  {
    const int ntiles = gridDim.x, x = tile & 7, base = ntiles >> 3, extra = ntiles & 7;       // XCD bands, as the tile kernel
    tile = x * base + min(x, extra) + (tile >> 3);
  }
  const int X0 = (tile % tiles_x) * TW, Y0 = (tile / tiles_x) * TH;
  const float* tgt_g = target + (size_t)b * 3 * N;
  const float* s0 = src0 + (size_t)b * 3 * N;
  const float* s1 = src1 + (size_t)b * 3 * N;
  for (int i = tid; i < R2N; i += NT) {
    const int ry = i / RW, rx = i - ry * RW;
    const int Y = dd_reflect(min(max(Y0 - 2 + ry, -1), H), H), X = dd_reflect(min(max(X0 - 2 + rx, -1), W), W);
    const unsigned o = (unsigned)(__mul24(Y, W) + X) * 4u;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      s_tgt[ch * R2N + i] = ldg(tgt_g + ch * (unsigned)N, o);
      s_pred[ch * R2N + i] = mk2(ldg(s0 + ch * (unsigned)N, o), ldg(s1 + ch * (unsigned)N, o));
    }
  }
  __syncthreads();
  const int lx = tid % TW, ly = tid / TW;
  if (X0 + lx < W && Y0 + ly < H) {
    const f2 rho = rho_pair<false>(s_pred, s_tgt, (ly + 2) * RW + (lx + 2), alpha, 0.f, nullptr);
    const size_t o = (size_t)b * 2 * N + (size_t)(Y0 + ly) * W + (X0 + lx);
    out[o] = rho[0];
    out[o + N] = rho[1];
  }
}

#ifndef DD_MIN_WAVES
#define DD_MIN_WAVES 1
#endif
// OUT: the materialised outputs of a log step (colour, sample grid, depth, ...) are written; the training step proper
// runs the instantiation without them (fewer live pointers: no scalar-register spills).
// SHARED: both frames read the same flow tensor (the sign rides on ts) and the same mask tensor, and their gradients
// go to the same buffers -- what networks.Model publishes; 5 instead of 9 low-res planes to stage and to back-project.
// SMOOTH (with GRAD; frames sharing their tensors, or the rigid mode): the edge-aware smoothness of the scale-0 disparity / flow /
// mask (tools.py:311-326, Trainer.py:355-359,380-381,401-402) is evaluated in the store stage -- the target tile sits in LDS, the
// thread holds its pixel's gradient -- and its gradient is added before the pixel's ONE store; seven more block sums.
template <int MODE, bool AUTOMASK, bool GRAD, bool SHARED, bool OUT, bool SMOOTH = false>
__global__ __launch_bounds__(NT, DD_MIN_WAVES) void photo_tile_kernel(const DDPhotoArgs a, const DepthParams dp, const ImageDims dim,
                                                                     const FootprintInfo fp, const FuseInfo fuse, const SideInfo side) {
  static_assert(!SMOOTH || (GRAD && (SHARED || MODE == MODE_RIGID)), "fused smoothness: gradient pass, one tensor per group");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  LdsLayout& S = *reinterpret_cast<LdsLayout*>(smem_raw);
  using CHN = Channels<MODE, SHARED>;
  constexpr int NCH = CHN::N;

  const int tid = threadIdx.x;
  const int H = a.H, W = a.W, N = H * W;
  const int tiles_x = (W + TW - 1) / TW;
#ifdef DD_SCALE_INNER
  // (build switch, round 6) ONE-dimensional grid: ntiles * B * S tile workgroups, then the side workgroups.  Workgroup i runs on XCD i % 8;
  // every XCD takes a contiguous range of the work list ordered (image, tile, scale) with the SCALE innermost: the three scale passes of a
  // tile -- same target tile, neighbouring source pixels -- follow one another on one XCD and share its L2 (the default order launches them
  // 2 880 workgroups apart: the launch fetches its 60 MB of frames four times over).  A bijection: results do not depend on it.
  const int ntiles = tiles_x * ((H + TH - 1) / TH), work = ntiles * a.B * a.num_scales;
  const int wid = blockIdx.x;
  const bool side_wg = wid >= work;
  int b, si, tile = 0;
  if (side_wg) { b = (wid - work) % a.B; si = (wid - work) / a.B; }
  else {
    const int x = wid & 7, base = work >> 3, extra = work & 7;
    const int w = x * base + min(x, extra) + (wid >> 3);
    si = w % a.num_scales;
    tile = (w / a.num_scales) % ntiles;
    b = w / (a.num_scales * ntiles);
  }
  const DDPhotoScale& sc = a.scale[si];
  if (SMOOTH && side.on && side_wg) {
#else
  const int b = blockIdx.y;
  const int si = blockIdx.z;
  const DDPhotoScale& sc = a.scale[si];
  const int ntiles = SMOOTH ? (int)gridDim.x - side.on : (int)gridDim.x;
  if (SMOOTH && side.on && (int)blockIdx.x == ntiles) {
#endif
    // ---- the extra workgroup of (image b, scale si): RANSAC candidates + disparity sum (dd_fuse.h) -- nothing of the tile path ----
    const SideScale& ss = side.sc[si];
    const int h = sc.h, w = sc.w, n = h * w;
    if (ss.inv_K && tid < side.max_it) {
      float cv[3];
      const int j = b * side.max_it + tid;
      ground_candidate_solve(sc.disp, ss.inv_K, ss.rand_idx, a.B, h, w, ss.rows, side.np, side.max_it, dp, j, cv);
#pragma unroll
      for (int i = 0; i < 3; ++i) ss.cand[(size_t)j * 3 + i] = cv[i];
    }
    if (ss.mean_partial) {
      const float* pl = sc.disp + (size_t)b * n;
      float acc = 0.f;
      if ((n & 3) == 0 && (reinterpret_cast<unsigned long long>(pl) & 15ull) == 0) {
#pragma unroll 4
        for (int i = tid * 4; i < n; i += NT * 4) {
          const float4 v = *reinterpret_cast<const float4*>(pl + i);
          acc += (v.x + v.y) + (v.z + v.w);
        }
      } else {
#pragma unroll 4
        for (int i = tid; i < n; i += NT) acc += pl[i];
      }
      float* red = reinterpret_cast<float*>(smem_raw);
      acc = wave_sum(acc);
      if ((tid & 63) == 0) red[tid >> 6] = acc;
      __syncthreads();
      if (tid < 32) {
        float tot = 0.f;
        if (tid == 0) {
#pragma unroll
          for (int k = 0; k < NWAVES; ++k) tot += red[k];
        }
        ss.mean_partial[b * 32 + tid] = tot;          // plane_mean folds 32 partials per image: one sum and 31 zeros
      }
    }
    return;
  }
  // XCD-aware tile order: workgroup i runs on XCD i % 8 (each XCD has its own L2).  Give every XCD a contiguous
  // band of the image so that neighbouring tiles -- which re-read each other's 2-pixel halo and the same source rows --
  // share an L2 instead of each missing separately.  Pure permutation of blockIdx.x: correctness does not depend on it.
#ifndef DD_SCALE_INNER
  int tile = blockIdx.x;
  {
    // XCD x = blockIdx.x % 8 runs workgroups x, x + 8, ...: it gets the contiguous band that starts behind the bands of XCDs 0..x-1
    // (the first ntiles % 8 XCDs hold one tile more) -- a bijection for every tile count (round 3 only remapped multiples of 8:
    // the 15 x 20 tiles of 320x480 and the 16 x 18 of 288x512 ran unmapped)
    const int x = tile & 7, base = ntiles >> 3, extra = ntiles & 7;
    tile = x * base + min(x, extra) + (tile >> 3);
  }
#endif
  const int X0 = (tile % tiles_x) * TW, Y0 = (tile / tiles_x) * TH;
  const int shift = sc.shift, h = sc.h, w = sc.w, n = h * w;
  const float ratio = 1.f / static_cast<float>(1 << shift);
  const float alpha = a.ssim_weight;
  Intrinsics cam;
  load_intrinsics(cam, a.K + b * 16, a.inv_K + b * 16);
  PairT Tm;
  load_pair_T(Tm, a.T[0] + b * 16, a.T[1] + b * 16);
  f2 tsv = sp2(1.f);
  if (MODE != MODE_RIGID) tsv = mk2(a.ts[0] ? a.ts[0][b] : 1.f, a.ts[1] ? a.ts[1][b] : 1.f);
  const float* tgt_g = a.target + (size_t)b * 3 * N;
  // Source frames: pixel-interleaved copies (B,H,W,3) when the caller hands them over (DDPhotoArgs.source_packed, dd_pack_rgb) -- one
  // global_load_dwordx3 per bilinear tap and frame, 8 gathers per pixel instead of 24 and a third of the cache lines they touch:
  // the gather's divergence, not its latency, is what the warp stage pays for (profiles/r05_photo_gather_ablation.txt) -- else the
  // planar tensors of the reference boundary.
  const bool packed = !DD_RING_LDS && a.source_packed[0] != nullptr && a.source_packed[1] != nullptr;
  const float* src0_g = (packed ? a.source_packed[0] : a.source[0]) + (size_t)b * 3 * N;
  const float* src1_g = (packed ? a.source_packed[1] : a.source[1]) + (size_t)b * 3 * N;
  const float* disp_g = sc.disp + (size_t)b * n;
  // plane p of the low-res inputs.  separate tensors: 0 disp | 1..3 flow f0 | 4..6 flow f1 | 7 mask f0 | 8 mask f1
  //                                 shared tensors  : 0 disp | 1..3 flow | 4 mask
  auto plane_ptr = [&](int p) -> const float* {
    if (p == 0) return disp_g;
    if (p < 1 + CHN::FLOW) return sc.flow[(p - 1) / 3] + ((size_t)b * 3 + (p - 1) % 3) * n;
    return sc.mask[p - CHN::MASK0] + (size_t)b * n;
  };

  // footprint of the tile's own pixels on the low-res grid (gradient side), scale >= 1 only
  const int fy0 = max((Y0 >> shift) - 1, 0), fx0 = max((X0 >> shift) - 1, 0);
  const int fph = (TH >> shift) + 2, fpw = (TW >> shift) + 2;
  const int lrh = TH >> shift, lrw = TW >> shift;           // low-res pixels inside the tile (shift >= 1)
  // staged low-res region covering the taps of the tile + 2-pixel halo
  const int lfy0 = max((Y0 >> shift) - 2, 0), lfx0 = max((X0 >> shift) - 2, 0);

#ifdef DD_STAGE_TIMING
  unsigned long long t_prev = clock64();
#endif
  DD_ISA("stage0 1.0");
  // ---- stage 0: stage the target region and (scale >= 1) the low-res inputs in LDS ---------------------
  // Positions one step outside the image receive the mirrored pixel (ReflectionPad2d(1)); positions further out are never
  // read by a centre inside the image, they get some valid pixel.  (The auto-mask's identity reprojection loss is scale-independent
  // and comes from photo_identity_kernel, once per step: round 4 staged target + sources here and re-evaluated it for every scale.)
  // staged low-res planes.  separate tensors: float [9][LOWN].  shared tensors: disp float [LOWN] | {flow x, flow y} f2 [LOWN] |
  // {flow z, mask} f2 [LOWN] -- two packed bilinear evaluations instead of four scalar ones
  float* const low_d = S.low;
  f2* const low_a = reinterpret_cast<f2*>(S.low + LOWN);
  f2* const low_b = reinterpret_cast<f2*>(S.low + 3 * LOWN);
  if (shift > 0) {
    // waves 0-3 and waves 4-7 take different planes (the plane pointers stay wave-uniform)
    const int half = __builtin_amdgcn_readfirstlane(tid >> 8), r = tid & 255;
    const int qy = lfy0 + r / LOWW, qx = lfx0 + r % LOWW;
    const bool ok = (r < LOWN) && (qy < h) && (qx < w);
    const unsigned o = ok ? (unsigned)(__mul24(qy, w) + qx) * 4u : 0u;
    if (SHARED) {
      if (r < LOWN) {
        const float* fl = sc.flow[0] + (size_t)b * 3 * n;
        if (half == 0) {
          low_d[r] = ok ? ldg(disp_g, o) : 0.f;
          low_a[r] = ok ? mk2(ldg(fl, o), ldg(fl + n, o)) : sp2(0.f);
        } else {
          const float mz = MODE == MODE_FLOW_MASK ? ldg(sc.mask[0] + (size_t)b * n, o) : 0.f;
          low_b[r] = ok ? mk2(ldg(fl + 2 * n, o), mz) : sp2(0.f);
        }
      }
    } else {
#pragma unroll
      for (int p0 = 0; p0 < NCH; p0 += 2) {
        const int p = p0 + half;
        if (p < NCH && r < LOWN) S.low[p * LOWN + r] = ok ? ldg(plane_ptr(p), o) : 0.f;
      }
    }
  }
  __syncthreads();
  DD_STAGE_MARK(0);

  // The target goes global -> LDS directly (global_load_lds_dword: no VGPR round trip) and the loads stay in flight across
  // stage A -- the target is first read in stage B; the wait sits in front of that stage's barrier.
  {
    for (int i0 = 0; i0 < R2N; i0 += NT) {
      const int i = i0 + tid;
      if (i < R2N) {
        const int ry = i / RW, rx = i - ry * RW;
        const int Y = dd_reflect(min(max(Y0 - 2 + ry, -1), H), H), X = dd_reflect(min(max(X0 - 2 + rx, -1), W), W);
        const unsigned o = (unsigned)(__mul24(Y, W) + X) * 4u;
        const int wave_base = __builtin_amdgcn_readfirstlane(i - (tid & 63));     // the wave's 64 positions are contiguous in LDS
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const char*>(tgt_g + ch * (unsigned)N) + o, S.tgt + ch * R2N + wave_base, 4, 0, 0);
      }
    }
  }

  DD_ISA("setup 1.0");
  // Own pixel / own centre of this thread.  Threads of a tile that overhangs the image compute on the clamped pixel (every
  // value stays finite, no divergent region to merge) and are kept out of every store and sum by `own`.
  const int lx = tid % TW, ly = tid / TW;
  const bool own = (X0 + lx < W) && (Y0 + ly < H);
  const int oX = min(X0 + lx, W - 1), oY = min(Y0 + ly, H - 1);
  const int op = __mul24(oY, W) + oX;
  const int oli = (ly + 2) * RW + (lx + 2);          // region index of the own pixel
  const int oci = (ly + 1) * CW_ + (lx + 1);         // centre index of the own pixel
  // Halo centre of this thread (0 <= rt < CRING): top row, bottom row, left column, right column of the CH_ x CW_ block.
  // The part-time jobs of a tile (halo centres, low-res pixels, the two up-sampling-adjoint passes) start at different
  // waves, and the halo centres alternate between waves 0-1 and 2-3 from tile to tile: a wave keeps its SIMD for life, and
  // giving every extra to waves 0 and 1 left two of the four SIMDs ~10 % more work than the others.
  const int rt = tid - ((tile & 1) << 7);
  const bool ring_lane = static_cast<unsigned>(rt) < static_cast<unsigned>(CRING);
  int hcy = 0, hcx = 0;
  if (rt < 2 * CW_) { hcy = rt < CW_ ? 0 : CH_ - 1; hcx = rt < CW_ ? rt : rt - CW_; }
  else { const int r3 = rt - 2 * CW_; hcy = 1 + (r3 >> 1); hcx = (r3 & 1) ? CW_ - 1 : 0; }
  const int hY = Y0 - 1 + hcy, hX = X0 - 1 + hcx;
  const bool hc_in = ring_lane && (hY >= 0) && (hY < H) && (hX >= 0) && (hX < W);
  const int hci = hcy * CW_ + hcx, hli = (hcy + 1) * RW + (hcx + 1);

  DD_ISA("automask 1.0");
  // ---- automask: min over frames of the identity reprojection loss (+ this scale's tie-break noise, Trainer.py:339) at the
  // thread's own centre and at its halo centre; the loss itself is photo_identity_kernel's (a.workspace), the loads are issued
  // here and first used in stage B
  float id_own = 0.f, id_ring = 0.f;
  if (AUTOMASK) {
    const float* idr = fp.idrho + (size_t)b * 2 * N;
    auto idmin_at = [&](int Y, int X) -> float {
      const unsigned o = (unsigned)(__mul24(Y, W) + X) * 4u;
      f2 rho = mk2(ldg(idr, o), ldg(idr + N, o));
      if (sc.noise) {
        const float* nz = sc.noise + (size_t)b * 2 * N;
        rho += mk2(ldg(nz, o), ldg(nz + N, o)) * sp2(0.00001f);
      }
      return rho[1] < rho[0] ? rho[1] : rho[0];
    };
    id_own = idmin_at(oY, oX);
    if (GRAD && hc_in) id_ring = idmin_at(hY, hX);
  }

  DD_ISA("setupA 1.0");
  // ---- stage A: geometry + warp ---------------------------------------------------------------------
  auto make_tap = [&](int X, int Y) -> LowTap {
    LowTap t;
    const Tap1 tx = resize_tap(X, w, ratio), ty = resize_tap(Y, h, ratio);
    t.o00 = (ty.i0 - lfy0) * LOWW + (tx.i0 - lfx0);
    t.o01 = (ty.i0 - lfy0) * LOWW + (tx.i1 - lfx0);
    t.o10 = (ty.i1 - lfy0) * LOWW + (tx.i0 - lfx0);
    t.o11 = (ty.i1 - lfy0) * LOWW + (tx.i1 - lfx0);
    t.wx0 = tx.w0; t.wx1 = tx.w1; t.wy0 = ty.w0; t.wy1 = ty.w1;
    return t;
  };
  // Warp of one target pixel against both source frames.  li = its index in the region.  Writes the warped colours to LDS
  // (and to the mirrored positions just outside the image) when `store`, returns the geometry.
  // up-sampled disparity / flow / mask of one pixel: identity at scale 0 (coalesced loads), LDS taps otherwise
  auto pixel_inputs = [&](int X, int Y, float& d, f2 c[3], f2& m) {
    const unsigned pb = (unsigned)(__mul24(Y, W) + X) * 4u;
    c[0] = c[1] = c[2] = sp2(0.f);
    m = sp2(1.f);
    if (shift == 0) {
      d = ldg(disp_g, pb);
      if (MODE != MODE_RIGID) {
        const float* f0 = sc.flow[0] + (size_t)b * 3 * n;
        const float* f1 = sc.flow[1] + (size_t)b * 3 * n;
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = (SHARED ? sp2(ldg(f0 + k * n, pb)) : mk2(ldg(f0 + k * n, pb), ldg(f1 + k * n, pb))) * tsv;
      }
      if (MODE == MODE_FLOW_MASK)
        m = SHARED ? sp2(ldg(sc.mask[0] + (size_t)b * n, pb)) : mk2(ldg(sc.mask[0] + (size_t)b * n, pb), ldg(sc.mask[1] + (size_t)b * n, pb));
    } else {
      const LowTap t = make_tap(X, Y);
      d = low_eval(low_d, t);
      if (SHARED) {
        const f2 fa = low_eval2(low_a, t), fb = low_eval2(low_b, t);
        c[0] = sp2(fa[0]) * tsv; c[1] = sp2(fa[1]) * tsv; c[2] = sp2(fb[0]) * tsv;
        if (MODE == MODE_FLOW_MASK) m = sp2(fb[1]);
      } else {
        if (MODE != MODE_RIGID) {
#pragma unroll
          for (int k = 0; k < 3; ++k) c[k] = mk2(low_eval(S.low + (1 + k) * LOWN, t), low_eval(S.low + (4 + k) * LOWN, t)) * tsv;
        }
        if (MODE == MODE_FLOW_MASK) m = mk2(low_eval(S.low + 7 * LOWN, t), low_eval(S.low + 8 * LOWN, t));
      }
    }
  };
  // depth -> back-projection -> (flow composition) -> rigid transform -> projection -> tap addresses and weights of both frames
  auto pixel_geometry = [&](int X, int Y, float d, const f2 c[3], f2 m, float& Z_out, PairGeom& g, PairSide& sd) -> SampleCoord2 {
    const float Z = dd_rcp(dp.lo + dp.span * d);
    Z_out = Z;
    float ray[3], P[3];
    pixel_ray(cam, X, Y, ray);
#pragma unroll
    for (int k = 0; k < 3; ++k) P[k] = Z * ray[k];
    frame_geometry2<MODE>(cam, Tm, P, c, m, dim, a.eps, g, sd);
    return sample_coord2(g.proj.u, g.proj.v, W, H);
  };
  // the warped colours of a region pixel go to LDS, and to the mirrored positions just outside the image (reflect padding: the pixel
  // one step inside the border is also the value one step outside it)
  auto store_warped = [&](int X, int Y, int ry, int rx, const f2 xval[3]) {
    const int li = ry * RW + rx;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) S.pred[ch * R2N + li] = xval[ch];
    const int mxo = X == 1 ? -2 : ((X == W - 2 && rx + 2 < RW) ? 2 : 0);
    const int myo = Y == 1 ? -2 * RW : ((Y == H - 2 && ry + 2 < RH) ? 2 * RW : 0);
    if (mxo | myo) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        if (mxo) S.pred[ch * R2N + li + mxo] = xval[ch];
        if (myo) S.pred[ch * R2N + li + myo] = xval[ch];
        if (mxo && myo) S.pred[ch * R2N + li + myo + mxo] = xval[ch];
      }
    }
  };
  // Warp of one target pixel against both source frames.  Writes the warped colours to LDS when `store`, returns the geometry.
  auto warp_pixel = [&](int X, int Y, int ry, int rx, bool store, float d, const f2 c[3], f2 m, float& Z_out, PairGeom& g, PairSide& sd,
                        f2 dvx[3], f2 dvy[3], f2 xval[3]) {
    const SampleCoord2 scd = pixel_geometry(X, Y, d, c, m, Z_out, g, sd);
    // all 24 source taps are issued before any is consumed (memory-level parallelism)
    f2 v00[3], v01[3], v10[3], v11[3];
#if DD_EXP_TAPS == 1      // timing experiment only (wrong results): every tap reads the pixel's own position -- what does the gather's divergence cost?
    const unsigned a00 = (unsigned)(__mul24(Y, W) + X) * 4u, a01 = a00, a10 = a00, a11 = a00, b00 = a00, b01 = a00, b10 = a00, b11 = a00;
#else
    const unsigned a00 = scd.o00[0], a01 = a00 + scd.dxb[0], a10 = a00 + scd.dyb[0], a11 = a10 + scd.dxb[0];
    const unsigned b00 = scd.o00[1], b01 = b00 + scd.dxb[1], b10 = b00 + scd.dyb[1], b11 = b10 + scd.dxb[1];
#endif
    if (packed) {
      const char* q0 = reinterpret_cast<const char*>(src0_g);
      const char* q1 = reinterpret_cast<const char*>(src1_g);
      // (plane offsets are bytes of fp32 elements: x 3 for the 12-byte pixels)
      const f3u t00a = *reinterpret_cast<const f3u*>(q0 + a00 * 3u), t01a = *reinterpret_cast<const f3u*>(q0 + a01 * 3u);
      const f3u t10a = *reinterpret_cast<const f3u*>(q0 + a10 * 3u), t11a = *reinterpret_cast<const f3u*>(q0 + a11 * 3u);
      const f3u t00b = *reinterpret_cast<const f3u*>(q1 + b00 * 3u), t01b = *reinterpret_cast<const f3u*>(q1 + b01 * 3u);
      const f3u t10b = *reinterpret_cast<const f3u*>(q1 + b10 * 3u), t11b = *reinterpret_cast<const f3u*>(q1 + b11 * 3u);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        v00[ch] = mk2(t00a[ch], t00b[ch]); v01[ch] = mk2(t01a[ch], t01b[ch]);
        v10[ch] = mk2(t10a[ch], t10b[ch]); v11[ch] = mk2(t11a[ch], t11b[ch]);
      }
    } else {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float* p0 = src0_g + ch * (unsigned)N;
        const float* p1 = src1_g + ch * (unsigned)N;
#if DD_EXP_TAPS == 2      // timing experiment only (wrong results): no source loads at all
        const float fk = __builtin_bit_cast(float, a00 + b01 + (unsigned)ch);
        v00[ch] = mk2(fk, fk + 1.f); v01[ch] = mk2(fk + 2.f, fk); v10[ch] = mk2(fk, fk + 3.f); v11[ch] = mk2(fk + 4.f, fk);
#elif DD_EXP_TAPS == 3    // timing experiment only (wrong results): one load per frame and channel instead of four
        v00[ch] = mk2(ldg(p0, a00), ldg(p1, b00));
        v01[ch] = v00[ch] + sp2(__builtin_bit_cast(float, a01)); v10[ch] = v00[ch] + sp2(__builtin_bit_cast(float, a10)); v11[ch] = v00[ch] + sp2(__builtin_bit_cast(float, b11));
#else
        v00[ch] = mk2(ldg(p0, a00), ldg(p1, b00));
        v01[ch] = mk2(ldg(p0, a01), ldg(p1, b01));
        v10[ch] = mk2(ldg(p0, a10), ldg(p1, b10));
        v11[ch] = mk2(ldg(p0, a11), ldg(p1, b11));
#endif
      }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) xval[ch] = sample_taps2(scd, v00[ch], v01[ch], v10[ch], v11[ch], dvx[ch], dvy[ch]);
    if (store) store_warped(X, Y, ry, rx, xval);
  };

#ifdef DD_DEPHASE         // experiment (round 5): the two workgroups of a CU start together and, with equal lifetimes, stay in phase (both in
  {                       // the load-bound warp stage, then both in the VALU-bound stages): delay every second first-generation workgroup
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (lin < 512u && ((lin >> DD_DEPHASE) & 1u)) {
#pragma unroll 1
      for (int k = 0; k < DD_DEPHASE_N; ++k) __builtin_amdgcn_s_sleep(127);
    }
  }
#endif
  DD_ISA("warp_halo 0.5");
#ifdef DD_PRIO_WARP       // experiment (round 5): waves 0-3 carry the workgroup's critical path (ring pass, then their own pixel)
  if (tid < 256) __builtin_amdgcn_s_setprio(DD_PRIO_WARP);
#endif
  // Halo ring: one pixel, both frames, per thread of waves 0-3, forward only, as a pass of its own IN FRONT of the owners' pass (its
  // transient registers are gone before the owner state comes alive): waves 0-3 go through inputs -> geometry -> 24 gathered taps ->
  // interpolation twice, one after the other, while waves 4-7 wait at the barrier (the warp stage is 46-66 % of a workgroup's life).
  // -DDD_RING_LDS=1 (experiment): the ring's taps go global -> LDS directly (global_load_lds_dword: no VGPR holds them), its four tap
  // weights wait in LDS too, the owner's pass is issued right behind -- ONE memory round trip covers both -- and the ring is interpolated
  // from LDS afterwards (staging area: the coefficient planes', not live before stage B; the owner's inputs are read first: the compiler
  // puts a full vmcnt(0) in front of every LDS read behind an LDS-direct load).  Same values (the parity tests pass), but 4 % SLOWER
  // without spills (disp_init, motion_init) and 25 % slower where the ring's finish meets the live owner state (fine_tune: 15 VGPRs
  // spilled): the two round trips of waves 0-3 are NOT what the warp stage waits for -- it is throughput (issue slots shared with the
  // CU's other workgroup, the texture path), not the latency of a dependent chain.
  auto ring_pixel = [&](int r, int& ry, int& rx) -> bool {      // region position of ring pixel r; is it one (inside the image)?
    ry = rx = 0;
    if (r < 2 * RW) { ry = r / RW; rx = r % RW; }
    else if (r < 4 * RW) { const int r2 = r - 2 * RW; ry = RH - 2 + r2 / RW; rx = r2 % RW; }
    else if (r < RING) { const int r3 = r - 4 * RW; ry = 2 + (r3 >> 2); const int k = r3 & 3; rx = k < 2 ? k : RW - 4 + k; }
    const int Y = Y0 - 2 + ry, X = X0 - 2 + rx;
    return r < RING && Y >= 0 && Y < H && X >= 0 && X < W;
  };
  int ring_ry, ring_rx;
  const bool ring_on = ring_pixel(tid, ring_ry, ring_rx);
  const int ring_Y = Y0 - 2 + ring_ry, ring_X = X0 - 2 + ring_rx;
  float d_own;
  f2 c_own[3], m_own;
#if DD_RING_LDS
  float* const ring_taps = S.low + RING_STAGE_OFF;                               // [24 taps][RING_WAVES][64 lanes]
  f2* const ring_wgt = reinterpret_cast<f2*>(ring_taps + 24 * RING_WAVES * 64);   // [4][RING_WAVES * 64]
  pixel_inputs(oX, oY, d_own, c_own, m_own);
  if (ring_on) {
    float d, Zu;
    f2 c[3], m;
    PairGeom gu;
    PairSide su;
    pixel_inputs(ring_X, ring_Y, d, c, m);
    const SampleCoord2 scd = pixel_geometry(ring_X, ring_Y, d, c, m, Zu, gu, su);
    ring_wgt[0 * RING_WAVES * 64 + tid] = scd.w00;
    ring_wgt[1 * RING_WAVES * 64 + tid] = scd.w01;
    ring_wgt[2 * RING_WAVES * 64 + tid] = scd.w10;
    ring_wgt[3 * RING_WAVES * 64 + tid] = scd.w11;
    const unsigned a00 = scd.o00[0], a01 = a00 + scd.dxb[0], a10 = a00 + scd.dyb[0], a11 = a10 + scd.dxb[0];
    const unsigned b00 = scd.o00[1], b01 = b00 + scd.dxb[1], b10 = b00 + scd.dyb[1], b11 = b10 + scd.dxb[1];
    float* const dst = ring_taps + __builtin_amdgcn_readfirstlane(tid & ~63);    // the hardware adds lane * 4
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const char* p0 = reinterpret_cast<const char*>(src0_g + ch * (unsigned)N);
      const char* p1 = reinterpret_cast<const char*>(src1_g + ch * (unsigned)N);
      float* const dk = dst + ch * 8 * RING_WAVES * 64;
      __builtin_amdgcn_global_load_lds(p0 + a00, dk + 0 * RING_WAVES * 64, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(p1 + b00, dk + 1 * RING_WAVES * 64, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(p0 + a01, dk + 2 * RING_WAVES * 64, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(p1 + b01, dk + 3 * RING_WAVES * 64, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(p0 + a10, dk + 4 * RING_WAVES * 64, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(p1 + b10, dk + 5 * RING_WAVES * 64, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(p0 + a11, dk + 6 * RING_WAVES * 64, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(p1 + b11, dk + 7 * RING_WAVES * 64, 4, 0, 0);
    }
  }
#else
  if (ring_on) {
    float d, Zu;
    f2 c[3], m, du[3], dw[3], xv[3];
    PairGeom gu;
    PairSide su;
    pixel_inputs(ring_X, ring_Y, d, c, m);
    warp_pixel(ring_X, ring_Y, ring_ry, ring_rx, true, d, c, m, Zu, gu, su, du, dw, xv);
  }
  pixel_inputs(oX, oY, d_own, c_own, m_own);
#endif
  // owner state (one pixel per thread)
  float Zs;
  f2 mval, dvx[3], dvy[3];
  PairGeom geo;
  f2 acc_cons = sp2(0.f), acc_delta = sp2(0.f);
  f2 lrv[5] = {sp2(0.f), sp2(0.f), sp2(0.f), sp2(0.f), sp2(0.f)};   // this pixel's share of the low-res residual / grid difference
  const f2 ownf = sp2(own ? 1.f : 0.f);

  DD_ISA("warp_own 1.0");
  {
    PairSide sd;
    f2 xval[3];
    warp_pixel(oX, oY, ly + 2, lx + 2, own, d_own, c_own, m_own, Zs, geo, sd, dvx, dvy, xval);
    mval = m_own;
#if DD_RING_LDS
    DD_ISA("warp_halo_finish 0.5");
    int tid2 = tid, fy, fx;
    asm volatile("" : "+v"(tid2));             // the ring position is formed again (five registers less across the owner's pass)
    if (ring_pixel(tid2, fy, fx)) {
      const int ring_ry = fy, ring_rx = fx, ring_Y = Y0 - 2 + fy, ring_X = X0 - 2 + fx;
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the ring's taps have landed (they were issued in front of the owner's)
      f2 xv[3];
      const f2 w00 = ring_wgt[0 * RING_WAVES * 64 + tid], w01 = ring_wgt[1 * RING_WAVES * 64 + tid];
      const f2 w10 = ring_wgt[2 * RING_WAVES * 64 + tid], w11 = ring_wgt[3 * RING_WAVES * 64 + tid];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float* tk = ring_taps + ch * 8 * RING_WAVES * 64 + tid;
        const f2 v00 = mk2(tk[0 * RING_WAVES * 64], tk[1 * RING_WAVES * 64]), v01 = mk2(tk[2 * RING_WAVES * 64], tk[3 * RING_WAVES * 64]);
        const f2 v10 = mk2(tk[4 * RING_WAVES * 64], tk[5 * RING_WAVES * 64]), v11 = mk2(tk[6 * RING_WAVES * 64], tk[7 * RING_WAVES * 64]);
        xv[ch] = v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11;         // sample_taps2's value
        DD_PIN(xv[ch]);
        DD_ORDER();          // one channel's eight taps in registers at a time: the owner state is alive here
      }
      store_warped(ring_X, ring_Y, ring_ry, ring_rx, xv);
    }
#endif
    if (OUT && own) {
      if (sc.out_depth) sc.out_depth[(size_t)b * N + op] = Zs;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        if (sc.out_color[f]) {
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) sc.out_color[f][((size_t)b * 3 + ch) * N + op] = xval[ch][f];
        }
        if (sc.out_sample[f])
          reinterpret_cast<float2*>(sc.out_sample[f])[(size_t)b * N + op] =
              make_float2(grid_normalise(geo.proj.u[f], dim.inv_wm1), grid_normalise(geo.proj.v[f], dim.inv_hm1));
      }
    }
    if (MODE == MODE_FLOW_MASK) {
      if (shift == 0) {
        // the low-res pixel IS this pixel: c_consistency and disp_mag directly
        const unsigned pb = (unsigned)op * 4u;
        const float valid = ldg(disp_g, pb) > a.disp_thr ? 1.f : 0.f;
        const f2 mk = SHARED ? sp2(ldg(sc.mask[0] + (size_t)b * n, pb)) : mk2(ldg(sc.mask[0] + (size_t)b * n, pb), ldg(sc.mask[1] + (size_t)b * n, pb));
        const f2 om = (sp2(valid) * ownf) * (sp2(1.f) - mk);
#pragma unroll
        for (int k = 0; k < 3; ++k) acc_cons += om * abs2(sd.r[k]);
        const f2 delta = sd.dgx * sd.dgx + sd.dgy * sd.dgy;
        acc_delta += delta * ownf;
        if (own) {
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            if (OUT && sc.out_resid[f]) {
#pragma unroll
              for (int k = 0; k < 3; ++k) sc.out_resid[f][((size_t)b * 3 + k) * n + op] = sd.r[k][f];
            }
            if (sc.out_delta[f]) sc.out_delta[f][(size_t)b * n + op] = delta[f];
          }
        }
      } else {
        // align_corners=False down-sampling by 2^s = mean of the 2x2 centre pixels of each block.  The two taps of a row
        // are adjacent lanes (one DPP add, all lanes take part); the two rows go to two LDS slots that stage L adds in a
        // fixed order: deterministic and free of LDS atomics.
        const f2 quarter = sp2((own && down_tap(oX, shift) && down_tap(oY, shift)) ? 0.25f : 0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) lrv[k] = quarter * sd.r[k];
        lrv[3] = quarter * sd.dgx;
        lrv[4] = quarter * sd.dgy;
      }
    }
  }
#ifdef DD_PRIO_WARP
  __builtin_amdgcn_s_setprio(0);
#endif
  DD_ISA("lr_down 1.0");
  if (MODE == MODE_FLOW_MASK && shift > 0) {
    const bool left = own && ((oX & ((1 << shift) - 1)) == (1 << (shift - 1)) - 1) && down_tap(oY, shift);
    const int slot = (oY & ((1 << shift) - 1)) == (1 << (shift - 1)) ? 1 : 0;
    const int q = (((oY - Y0) >> shift) * lrw) + ((oX - X0) >> shift);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const f2 pair = mk2(add_right_neighbour(lrv[k][0]), add_right_neighbour(lrv[k][1]));
      if (left) S.lr[(slot * 5 + k) * LRN_MAX + q] = pair;
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): the target planes have landed in LDS
  __syncthreads();
  DD_STAGE_MARK(2);

  DD_ISA("ssim_own 1.0");
#ifdef DD_PRIO_SSIM       // experiment: the VALU-dense stages issue ahead of the other workgroup's load-bound stages
  __builtin_amdgcn_s_setprio(DD_PRIO_SSIM);
#endif
  // ---- stage B: SSIM + L1, selection, loss, backward coefficients ------------------------------------
  float acc_photo = 0.f, acc_nwarp = 0.f;
  // One centre, branch-free.  A centre outside the image evaluates the window of the tile's first pixel instead (always
  // inside the image: finite data, finite coefficients) and is deselected afterwards, so that the gather can apply the
  // selection as a multiplication by 0 / weight.  Returns the selected frame (-1: identity won / outside the image) and
  // the selected loss.
  auto centre = [&](int ci, int li, bool in, float idb, float& best_out) -> int {
    const float wgt = sc.w_photo * alpha * (1.f / 27.f);     // the coefficients come out weighted
#ifdef DD_EXP_NO_SSIM      // timing experiment only (wrong results): what do the SSIM window sums cost?
    const f2 rho = S.pred[in ? li : 2 * RW + 2] * sp2(wgt) + sp2(S.tgt[li]);
    if (GRAD) { for (int k = 0; k < 9; ++k) S.coef[ci + k * R1N] = rho; }
#else
    const f2 rho = rho_pair<GRAD>(S.pred, S.tgt, in ? li : 2 * RW + 2, alpha, wgt, S.coef + ci);
#endif
    const bool second = rho[1] < rho[0];
    float best = second ? rho[1] : rho[0];
    bool warped = in;
    if (AUTOMASK) {
      const bool idwin = idb <= best;            // identity entries precede the warped ones in the cat: ties go to them
      best = idwin ? idb : best;
      warped = in && !idwin;
    }
    best_out = in ? best : 0.f;
    const int bf = warped ? (second ? 1 : 0) : -1;
    if (GRAD) S.sel[ci] = (signed char)bf;
    return bf;
  };
  {
    float best;
    const int bf = centre(oci, oli, own, id_own, best);
    acc_photo += best;
    acc_nwarp += bf >= 0 ? 1.f : 0.f;
    if (AUTOMASK && OUT && own && sc.out_idsel) sc.out_idsel[(size_t)b * N + op] = bf >= 0 ? 1.f : 0.f;
  }
  DD_ISA("ssim_ring 0.25");
  if (GRAD && ring_lane) {
    float best;
    centre(hci, hli, hc_in, id_ring, best);
  }

#ifdef DD_PRIO_SSIM
  __builtin_amdgcn_s_setprio(0);
#endif
  DD_ISA("stageL 0.25");
  // ---- stage L: c_consistency and disp_mag on the tile's low-res pixels (scale >= 1) -------------------
  if (MODE == MODE_FLOW_MASK && shift > 0) {
    for (int q = (tid + NT - 128) & (NT - 1); q < lrh * lrw; q += NT) {     // starts at wave 2
      const int qy = (Y0 >> shift) + (q >> (5 - shift)), qx = (X0 >> shift) + (q & (lrw - 1));     // lrw = TW >> shift, TW = 32
      if (qy < h && qx < w) {
        const int gq = __mul24(qy, w) + qx;
        const float valid = disp_g[gq] > a.disp_thr ? 1.f : 0.f;
        const f2 mk = SHARED ? sp2(sc.mask[0][(size_t)b * n + gq]) : mk2(sc.mask[0][(size_t)b * n + gq], sc.mask[1][(size_t)b * n + gq]);
        const f2 om = sp2(valid) * (sp2(1.f) - mk);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const f2 rv = S.lr[k * LRN_MAX + q] + S.lr[(5 + k) * LRN_MAX + q];
          acc_cons += om * abs2(rv);
          S.lr[k * LRN_MAX + q] = sp2(sc.w_cons) * om * sign2(rv);
#pragma unroll
          for (int f = 0; f < 2; ++f)
            if (OUT && sc.out_resid[f]) sc.out_resid[f][((size_t)b * 3 + k) * n + gq] = rv[f];
        }
        const f2 dx = S.lr[3 * LRN_MAX + q] + S.lr[(5 + 3) * LRN_MAX + q];
        const f2 dy = S.lr[4 * LRN_MAX + q] + S.lr[(5 + 4) * LRN_MAX + q];
        const f2 delta = dx * dx + dy * dy;
        acc_delta += delta;
#pragma unroll
        for (int f = 0; f < 2; ++f)
          if (sc.out_delta[f]) sc.out_delta[f][(size_t)b * n + gq] = delta[f];
      }
    }
  }
  __syncthreads();
  DD_STAGE_MARK(3);

  DD_ISA("gather 1.0");
#ifdef DD_PRIO_GATHER
  __builtin_amdgcn_s_setprio(DD_PRIO_GATHER);
#endif
  // ---- stage C: backward ------------------------------------------------------------------------------
  f2 gTacc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) gTacc[k] = sp2(0.f);
  float sm_acc[NSMOOTH];          // SMOOTH: sx_d, sy_d, dot_d | sx_c, sy_c | sx_m, sy_m of this pixel (scale 0)
#pragma unroll
  for (int k = 0; k < NSMOOTH; ++k) sm_acc[k] = 0.f;

  if (GRAD) {
    float gch[NCH];           // d loss / d (up-sampled disp, flow, mask) of this pixel
    {
      const int X = oX, Y = oY;
      // Adjoint of the reflect-padded 3x3 box filter, gather form.  Centres outside the image carry sel = -1
      // (no bounds tests needed); a border centre counts its inner neighbour twice (reflection), which is the
      // closed-form weight below.  A thread outside the image gets weight 0 everywhere: all its gradients come out
      // as exact zeros.
      f2 Sc[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) Sc[k] = sp2(0.f);
      const float wy_lo = Y == 1 ? 2.f : 1.f, wy_hi = Y == H - 2 ? 2.f : 1.f;
      const float o1 = own ? 1.f : 0.f;
      const float wxv[3] = {X == 1 ? 2.f * o1 : o1, o1, X == W - 2 ? 2.f * o1 : o1};
      const int own_sel = own ? (int)S.sel[oci] : -1;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dxx = 0; dxx < 3; ++dxx) {
          const int ci = oci + (dy - 1) * CW_ + (dxx - 1);
          const int sl = S.sel[ci];
          const float wgt = (dy == 0 ? wy_lo : (dy == 2 ? wy_hi : 1.f)) * wxv[dxx];
          const f2 mm = mk2(sl == 0 ? wgt : 0.f, sl == 1 ? wgt : 0.f);     // every stored coefficient is finite (see centre())
#pragma unroll
          for (int k = 0; k < 9; ++k) Sc[k] += mm * S.coef[k * R1N + ci];
#pragma unroll
          for (int k = 0; k < 9; ++k) DD_PIN(Sc[k]);
          DD_ORDER();    // one centre at a time: nine f2 in flight, not eighty-one
        }
      DD_ISA("chain 1.0");
      const float wl1 = own ? sc.w_photo * (1.f - alpha) * (1.f / 3.f) : 0.f;
      const f2 l1w = mk2(own_sel == 0 ? wl1 : 0.f, own_sel == 1 ? wl1 : 0.f);
      f2 gu = sp2(0.f), gv = sp2(0.f);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const f2 xv = S.pred[ch * R2N + oli];                 // re-read: frees registers across stage B
        const float yv = S.tgt[ch * R2N + oli];
        const f2 gx = Sc[ch * 3 + 0] + xv * Sc[ch * 3 + 1] + sp2(yv) * Sc[ch * 3 + 2] + l1w * sign2(xv - sp2(yv));
        gu += gx * dvx[ch];
        gv += gx * dvy[ch];
      }
      f2 gr_extra[3] = {sp2(0.f), sp2(0.f), sp2(0.f)};
      if (MODE == MODE_FLOW_MASK) {
        if (shift == 0) {
          const unsigned pb = (unsigned)op * 4u;
          const float vw = (own && ldg(disp_g, pb) > a.disp_thr) ? sc.w_cons : 0.f;
          const f2 mk = SHARED ? sp2(ldg(sc.mask[0] + (size_t)b * n, pb)) : mk2(ldg(sc.mask[0] + (size_t)b * n, pb), ldg(sc.mask[1] + (size_t)b * n, pb));
          const f2 vm = sp2(vw) * (sp2(1.f) - mk);
#pragma unroll
          for (int k = 0; k < 3; ++k) gr_extra[k] = vm * sign2(geo.r[k]);
        } else {
          const f2 quarter = sp2((own && down_tap(X, shift) && down_tap(Y, shift)) ? 0.25f : 0.f);
          const int q = (((Y - Y0) >> shift) * lrw) + ((X - X0) >> shift);
#pragma unroll
          for (int k = 0; k < 3; ++k) gr_extra[k] = quarter * S.lr[k * LRN_MAX + q];
        }
      }
      float ray[3], P[3];
      pixel_ray(cam, X, Y, ray);
#pragma unroll
      for (int k = 0; k < 3; ++k) P[k] = Zs * ray[k];
      PairGrad pg;
      frame_geometry_bwd2<MODE>(cam, Tm, P, mval, geo, gu, gv, gr_extra, pg);
#pragma unroll
      for (int k = 0; k < 12; ++k) gTacc[k] = pg.gT[k];
      const float gPtot[3] = {hsum(pg.gP[0]), hsum(pg.gP[1]), hsum(pg.gP[2])};
      gch[0] = depth_bwd(dp, gPtot, ray, Zs);
      if (MODE != MODE_RIGID) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const f2 gk = pg.gc[k] * tsv;
          if (SHARED) gch[1 + k] = hsum(gk);
          else { gch[1 + k] = gk[0]; gch[4 + k] = gk[1]; }
        }
      }
      if (MODE == MODE_FLOW_MASK) {
        if (SHARED) gch[CHN::MASK0] = hsum(pg.gm);
        else { gch[CHN::MASK0] = pg.gm[0]; gch[CHN::MASK0 + 1] = pg.gm[1]; }
      }
    }
#ifdef DD_PRIO_GATHER
    __builtin_amdgcn_s_setprio(0);
#endif
    DD_ISA("store 1.0");
    auto grad_ptr = [&](int ch) -> float* {
      if (ch == 0) return sc.g_disp + (size_t)b * n;
      if (ch < 1 + CHN::FLOW) return sc.g_flow[(ch - 1) / 3] + ((size_t)b * 3 + (ch - 1) % 3) * n;
      return sc.g_mask[ch - CHN::MASK0] + (size_t)b * n;
    };
    if (shift == 0) {
      // up-sampling is the identity and every element has exactly one owner: plain coalesced stores into the
      // gradient buffers.  A device-scope float atomic would cost one 32-64 B fabric write per 4 useful
      // bytes (measured: WRITE_SIZE 4.3x the algorithmic bytes).  Buffers that the caller aliases between the two
      // frames (the shared motion mask) receive the sum.
      if (SMOOTH && own && fuse.on) {
        // ---- edge-aware smoothness of this pixel (scale 0: the pyramid level IS the target, which sits in LDS with its halo).
        // A pixel owns its right and lower difference terms; its gradient also takes the left / upper neighbours' terms.  The
        // disparity enters un-normalised: |d_p - d_q| / (mean + eps) = |d_p/(mean+eps) - d_q/(mean+eps)| up to rounding, the sign of
        // the difference is the same, so the per-image factor 1/(mean + eps) is applied to the folded sums and in the finishing
        // pass of the gradient (dd_reg.hip: image_fold_body / disp_finish_body) -- no mean is needed in front of this kernel.
        const FuseScale& fz = fuse.sc[si];
        const bool has_r = oX + 1 < W, has_l = oX > 0, has_d = oY + 1 < H, has_u = oY > 0;
        const unsigned pb = (unsigned)op * 4u;
        const unsigned o_r = pb + (has_r ? 4u : 0u), o_l = pb - (has_l ? 4u : 0u);
        const unsigned o_d = pb + (has_d ? (unsigned)W * 4u : 0u), o_u = pb - (has_u ? (unsigned)W * 4u : 0u);
        // every plane's five values first, unconditionally and before the edge weights are formed: 5 x NCH loads in flight at once
        // (behind the per-group `if` they went out one channel at a time -- five dependent round trips at the end of the kernel)
        float av[NCH][5];
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          const float* src = plane_ptr(k);
          av[k][0] = ldg(src, pb); av[k][1] = ldg(src, o_r); av[k][2] = ldg(src, o_l); av[k][3] = ldg(src, o_d); av[k][4] = ldg(src, o_u);
        }
        float dr = 0.f, dl = 0.f, dd_ = 0.f, du = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float* tp = S.tgt + ch * R2N + oli;
          const float cc = tp[0];
          dr += dd_abs(cc - tp[1]); dl += dd_abs(tp[-1] - cc); dd_ += dd_abs(cc - tp[RW]); du += dd_abs(tp[-RW] - cc);
        }
        // exp(-mean_c |dI|), zero where the neighbour is outside the image (the term does not exist)
        const float third = -1.f / 3.f;
        const float e_r = has_r ? __expf(dr * third) : 0.f, e_l = has_l ? __expf(dl * third) : 0.f;
        const float e_d = has_d ? __expf(dd_ * third) : 0.f, e_u = has_u ? __expf(du * third) : 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          const int grp = k == 0 ? 0 : (k < 1 + CHN::FLOW ? 1 : 2);
          const float a_c = av[k][0];
          const f2 dh = mk2(a_c - av[k][1], av[k][2] - a_c), dv = mk2(a_c - av[k][3], av[k][4] - a_c);      // {own term, the neighbour's term}
          const f2 sh = signed_weight2(dh, mk2(e_r, e_l)), sv = signed_weight2(dv, mk2(e_d, e_u));
          const float g = (sh[0] - sh[1]) * fz.wx[grp] + (sv[0] - sv[1]) * fz.wy[grp];          // weights 0: the group is not smoothed in this phase
          if (fz.wx[grp] != 0.f) {                     // uniform
            sm_acc[grp == 0 ? 0 : (grp == 1 ? 3 : 5)] += dd_abs(dh[0]) * e_r;
            sm_acc[grp == 0 ? 1 : (grp == 1 ? 4 : 6)] += dd_abs(dv[0]) * e_d;
            if (k == 0) {
              fz.g_tmp[(size_t)b * n + op] = g;        // d/d(normalised disparity): the finishing pass divides by mean + eps
              sm_acc[2] = g * a_c;
            }
          }
          if (k > 0) gch[k] += g;
        }
      }
      if (own) {
        grad_ptr(0)[op] = gch[0];
        if (SHARED) {
#pragma unroll
          for (int k = 1; k < NCH; ++k) grad_ptr(k)[op] = gch[k];
        } else {
          if (MODE != MODE_RIGID) {
            const bool alias = sc.g_flow[0] == sc.g_flow[1];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              if (alias) grad_ptr(1 + k)[op] = gch[1 + k] + gch[4 + k];
              else { grad_ptr(1 + k)[op] = gch[1 + k]; grad_ptr(4 + k)[op] = gch[4 + k]; }
            }
          }
          if (MODE == MODE_FLOW_MASK) {
            if (sc.g_mask[0] == sc.g_mask[1]) grad_ptr(7)[op] = gch[7] + gch[8];
            else { grad_ptr(7)[op] = gch[7]; grad_ptr(8)[op] = gch[8]; }
          }
        }
      }
    } else {
      // Adjoint of the bilinear up-sampling WITHOUT atomics on the LDS: park the per-pixel gradients, then every
      // low-res footprint element gathers its contributions, x first (separable), then y.
      float* G = reinterpret_cast<float*>(S.pred);      // [NCH][TH*TW], spans pred+tgt
      float* Hx = reinterpret_cast<float*>(S.coef);     // [NCH][TH][FPW_MAX]
      __syncthreads();                    // all reads of pred/tgt/coef are done
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) G[ch * (TH * TW) + tid] = gch[ch];
      __syncthreads();
      DD_STAGE_MARK(4);
      // The weight with which full-res X feeds low-res q is the tent 1 - |s - q| around the (clamped) source coordinate
      // s = clamp((X + 0.5) / 2^shift - 0.5, 0, w - 1): the same numbers the forward taps use, bit for bit (every operand is a
      // multiple of 2^-(shift+1), all sums are exact), without re-deriving the tap pair per element.
      const int blk = 1 << shift;
      const float inv_fpw = 1.f / static_cast<float>(fpw);
      for (int i = (tid + NT / 2) & (NT - 1); i < TH * fpw; i += NT) {       // starts at wave 4
        const int r = static_cast<int>((static_cast<float>(i) + 0.5f) * inv_fpw), j = i - __mul24(r, fpw);
        const int q = fx0 + j;
        float acc[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) acc[ch] = 0.f;
        const int xa = max(X0, blk * q - (blk >> 1)), xb = min(min(X0 + TW, W), blk * q + 3 * (blk >> 1));
        const float qf = static_cast<float>(q), hi = static_cast<float>(w - 1);
        float sX = (static_cast<float>(xa) + 0.5f) * ratio - 0.5f;
        const float* gp = G + r * TW + (xa - X0);
        for (int X = xa; X < xb; ++X, sX += ratio, ++gp) {
          const float wt = fmaxf(0.f, 1.f - fabsf(clamp_coord(sX, hi) - qf));
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) acc[ch] = fmaf(wt, gp[ch * (TH * TW)], acc[ch]);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) Hx[(ch * TH + r) * FPW_MAX + j] = acc[ch];
      }
      __syncthreads();
      for (int i = (tid + NT - 128) & (NT - 1); i < fph * fpw; i += NT) {    // starts at wave 2
        const int jy = static_cast<int>((static_cast<float>(i) + 0.5f) * inv_fpw), j = i - __mul24(jy, fpw);
        const int qy = fy0 + jy, qx = fx0 + j;
        if (qy >= h || qx >= w) continue;            // never read by the combine pass
        float acc[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) acc[ch] = 0.f;
        const int ya = max(Y0, blk * qy - (blk >> 1)), yb = min(min(Y0 + TH, H), blk * qy + 3 * (blk >> 1));
        const float qf = static_cast<float>(qy), hi = static_cast<float>(h - 1);
        float sY = (static_cast<float>(ya) + 0.5f) * ratio - 0.5f;
        const float* hp = Hx + (ya - Y0) * FPW_MAX + j;
        for (int Y = ya; Y < yb; ++Y, sY += ratio, hp += FPW_MAX) {
          const float wt = fmaxf(0.f, 1.f - fabsf(clamp_coord(sY, hi) - qf));
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) acc[ch] = fmaf(wt, hp[ch * (TH * FPW_MAX)], acc[ch]);
        }
        // the footprint rim overlaps the neighbouring tiles' footprints: photo_combine_kernel adds them up
        float* dst = fp.base + fp.off[si] + ((size_t)(b * ntiles + tile) * NCH) * (fph * fpw) + jy * fpw + j;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) dst[(size_t)ch * (fph * fpw)] = acc[ch];
      }
    }
  }

  DD_STAGE_MARK(5);
  DD_ISA("reduce 1.0");
  // ---- stage R: block reduction -> one record per block ---------------------------------------------
  // Transpose through LDS: every thread parks its 30 values ([value][thread], conflict-free plain stores), then wave g adds
  // up values 4g..4g+3 over the 512 threads (eight reads per value and lane) and finishes with four DPP wave sums --
  // ~60 VALU instructions per thread instead of 30 seven-step DPP reductions.
  // record: [0] photo, [1] n_warp, [2..3] cons, [4..5] delta, [6 + f*12 + k] gT
  {
    float* R = reinterpret_cast<float*>(smem_raw);         // [32][NT] ([40][NT] with the smoothness sums)
    __syncthreads();                                        // every reader of the LDS regions is done
    R[0 * NT + tid] = acc_photo;
    R[1 * NT + tid] = acc_nwarp;
    R[2 * NT + tid] = acc_cons[0];
    R[3 * NT + tid] = acc_cons[1];
    R[4 * NT + tid] = acc_delta[0];
    R[5 * NT + tid] = acc_delta[1];
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int k = 0; k < 12; ++k) R[(6 + f * 12 + k) * NT + tid] = gTacc[k][f];
    if (SMOOTH) {
#pragma unroll
      for (int k = 0; k < NSMOOTH; ++k) R[(REC_SMOOTH + k) * NT + tid] = sm_acc[k];
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int VPW = SMOOTH ? 5 : 4, NVAL = SMOOTH ? NRED + NSMOOTH : NRED;     // values per wave; value v lands in rec[v]
    float s5[VPW];
#pragma unroll
    for (int v = 0; v < VPW; ++v) {
      float acc = 0.f;
      if (wave * VPW + v < NVAL) {        // wave-uniform
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) acc += R[(wave * VPW + v) * NT + i * 64 + lane];
        acc = wave_sum(acc);
      }
      s5[v] = acc;
    }
    if (lane == 0) {
      const size_t rec = ((size_t)si * a.B + b) * ntiles + tile;
      float* dst = a.workspace + rec * DD_PARTIAL_STRIDE + wave * VPW;
      if (SMOOTH) {
#pragma unroll
        for (int v = 0; v < VPW; ++v) dst[v] = s5[v];
      } else {
        *reinterpret_cast<float4*>(dst) = make_float4(s5[0], s5[1], s5[2], s5[3]);
      }
    }
  }
  DD_STAGE_MARK(6);
}

// Folds the per-block records: blocks [0,S) produce sums[s][*]; blocks [S, S+B) produce g_T[.][b].
__global__ __launch_bounds__(256) void photo_finalize_kernel(const float* __restrict__ partials, int S_, int B, int tiles,
                                                             float* __restrict__ sums, float* g_T0, float* g_T1) {
  __shared__ float red[4 * 24];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float acc[24];
#pragma unroll
  for (int k = 0; k < 24; ++k) acc[k] = 0.f;
  int nvals;
  if ((int)blockIdx.x < S_) {
    nvals = 6;
    const int s = blockIdx.x;
    for (int i = tid; i < B * tiles; i += 256) {
      const float* rec = partials + ((size_t)s * B * tiles + i) * DD_PARTIAL_STRIDE;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += rec[k];
    }
  } else {
    nvals = 24;
    const int b = blockIdx.x - S_;
    for (int i = tid; i < S_ * tiles; i += 256) {
      const int s = i / tiles, t = i % tiles;
      const float* rec = partials + (((size_t)s * B + b) * tiles + t) * DD_PARTIAL_STRIDE + 6;
#pragma unroll
      for (int k = 0; k < 24; ++k) acc[k] += rec[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const float r = wave_sum(acc[k]);
    if (lane == 0) red[wave * 24 + k] = r;
  }
  __syncthreads();
  if (tid < nvals) {
    const float r = red[tid] + red[24 + tid] + red[48 + tid] + red[72 + tid];
    if ((int)blockIdx.x < S_) {
      // record order: photo, n_warp, cons0, cons1, delta0, delta1 -> sums order: photo, cons0, cons1, delta0, delta1, n_warp
      const int map[6] = {0, 5, 1, 2, 3, 4};
      sums[blockIdx.x * DD_SUMS_STRIDE + map[tid]] = r;
    } else if (g_T0) {
      const int b = blockIdx.x - S_;
      float* dst = (tid < 12 ? g_T0 : g_T1) + b * 16;
      dst[tid % 12] = r;
    }
  }
  if ((int)blockIdx.x >= S_ && g_T0 && tid < 8) {
    const int b = blockIdx.x - S_;
    (tid < 4 ? g_T0 : g_T1)[b * 16 + 12 + (tid & 3)] = 0.f;
  }
  if ((int)blockIdx.x < S_ && tid >= 6 && tid < DD_SUMS_STRIDE) sums[blockIdx.x * DD_SUMS_STRIDE + tid] = 0.f;
}

// Sums, for every low-res pixel of every scale >= 1, the footprint partials of the tiles that overlap it (at most two
// per axis), always in the same order, and writes the gradient (single owner: plain store).
// NCH 1: disp | 4: + shared flow | 5: + shared mask | 7: + flow per frame | 9: + mask per frame
template <int NCH>
__global__ __launch_bounds__(256) void photo_combine_kernel(const DDPhotoArgs a, const FootprintInfo fp, int tiles_x, int tiles_y) {
  const int si = blockIdx.z, b = blockIdx.y;
  const DDPhotoScale& sc = a.scale[si];
  const int shift = sc.shift, h = sc.h, w = sc.w, n = h * w;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (shift == 0 || q >= n) return;
  const int qy = q / w, qx = q - qy * w;
  const int lrh = TH >> shift, lrw = TW >> shift, fph = lrh + 2, fpw = lrw + 2;
  float acc[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) acc[ch] = 0.f;
  const int ty_hi = min(qy / lrh + 1, tiles_y - 1), tx_hi = min(qx / lrw + 1, tiles_x - 1);
  for (int ty = max(qy / lrh - 1, 0); ty <= ty_hi; ++ty) {
    const int jy = qy - max(ty * lrh - 1, 0);
    if (jy < 0 || jy >= fph) continue;
    for (int tx = max(qx / lrw - 1, 0); tx <= tx_hi; ++tx) {
      const int j = qx - max(tx * lrw - 1, 0);
      if (j < 0 || j >= fpw) continue;
      const int tile = ty * tiles_x + tx;
      const float* src = fp.base + fp.off[si] + ((size_t)(b * (tiles_x * tiles_y) + tile) * NCH) * (fph * fpw) + jy * fpw + j;
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) acc[ch] += src[(size_t)ch * (fph * fpw)];
    }
  }
  sc.g_disp[(size_t)b * n + q] = acc[0];
  if (NCH == 4 || NCH == 5) {
#pragma unroll
    for (int k = 0; k < 3; ++k) sc.g_flow[0][((size_t)b * 3 + k) * n + q] = acc[1 + k];
    if (NCH == 5) sc.g_mask[0][(size_t)b * n + q] = acc[4];
  }
  if (NCH >= 7) {
    const bool alias = sc.g_flow[0] == sc.g_flow[1];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (alias) sc.g_flow[0][((size_t)b * 3 + k) * n + q] = acc[1 + k] + acc[4 + k];
      else { sc.g_flow[0][((size_t)b * 3 + k) * n + q] = acc[1 + k]; sc.g_flow[1][((size_t)b * 3 + k) * n + q] = acc[4 + k]; }
    }
  }
  if (NCH >= 9) {
    if (sc.g_mask[0] == sc.g_mask[1]) sc.g_mask[0][(size_t)b * n + q] = acc[7] + acc[8];
    else { sc.g_mask[0][(size_t)b * n + q] = acc[7]; sc.g_mask[1][(size_t)b * n + q] = acc[8]; }
  }
}

// do both frames read (and differentiate into) the same flow and mask tensors at every scale?
bool frames_share_tensors(const DDPhotoArgs& a) {
  if (a.mode == DD_MODE_RIGID) return false;
  for (int s = 0; s < a.num_scales; ++s) {
    const DDPhotoScale& sc = a.scale[s];
    if (sc.flow[0] != sc.flow[1] || sc.g_flow[0] != sc.g_flow[1]) return false;
    if (a.mode == DD_MODE_FLOW_MASK && (sc.mask[0] != sc.mask[1] || sc.g_mask[0] != sc.g_mask[1])) return false;
  }
  return true;
}

int gradient_channels(const DDPhotoArgs& a) {
  const bool sh = frames_share_tensors(a);
  return a.mode == DD_MODE_RIGID ? 1 : (a.mode == DD_MODE_FLOW ? (sh ? 4 : 7) : (sh ? 5 : 9));
}

size_t footprint_floats(const DDPhotoArgs& a, long long off[DD_MAX_SCALES]) {
  const size_t tiles = (size_t)((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
  const int nch = gradient_channels(a);
  size_t total = 0;
  for (int s = 0; s < a.num_scales; ++s) {
    off[s] = (long long)total;
    const int shift = a.scale[s].shift;
    if (shift > 0 && a.want_grad) total += tiles * a.B * nch * (size_t)((TH >> shift) + 2) * ((TW >> shift) + 2);
  }
  return total;
}

// does any scale ask for a materialised output (a log step)?
static bool wants_outputs(const DDPhotoArgs& a) {
  for (int s = 0; s < a.num_scales; ++s) {
    const DDPhotoScale& sc = a.scale[s];
    if (sc.out_depth || sc.out_idsel || sc.out_color[0] || sc.out_color[1] || sc.out_sample[0] || sc.out_sample[1] || sc.out_resid[0] ||
        sc.out_resid[1])
      return true;
  }
  return false;
}

// dd_photo_timing(): HIP events around the tile kernel alone, on the stream it is launched on (bench.py's roofline leg)
struct TileTimer {
  static constexpr int CAP = 256;
  bool on = false;
  int n = 0;
  hipEvent_t ev[CAP][2];
  int created = 0;
};
static TileTimer g_timer;

static bool timer_slot(hipStream_t stream, hipEvent_t*& pair) {
  if (!g_timer.on || g_timer.n >= TileTimer::CAP) return false;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return false;
  if (g_timer.n >= g_timer.created) {
    if (hipEventCreate(&g_timer.ev[g_timer.created][0]) != hipSuccess || hipEventCreate(&g_timer.ev[g_timer.created][1]) != hipSuccess) return false;
    ++g_timer.created;
  }
  pair = g_timer.ev[g_timer.n++];
  return true;
}

// the tile kernel alone, with the HIP-event bracket of dd_photo_timing
template <int MODE, bool AUTOMASK, bool GRAD, bool SHARED, bool OUT, bool SMOOTH>
static int launch_tile(const DDPhotoArgs& a, const FuseInfo& fuse, const SideInfo& side, hipStream_t stream) {
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH, tiles = tiles_x * tiles_y;
#ifdef DD_SCALE_INNER
  dim3 grid(tiles * a.B * a.num_scales + ((SMOOTH && side.on) ? a.B * a.num_scales : 0));
#else
  dim3 grid(tiles + ((SMOOTH && side.on) ? 1 : 0), a.B, a.num_scales);
#endif
  auto kern = photo_tile_kernel<MODE, AUTOMASK, GRAD, SHARED, OUT, SMOOTH>;
  static dd::LdsAttrOnce lds_attr;          // per instantiation and device (dd_attr.h)
  if (const int rc = lds_attr.ensure(reinterpret_cast<const void*>(kern), (int)((int)lds_bytes(SMOOTH)))) return rc;
  FootprintInfo fp;
  const size_t fp_floats = footprint_floats(a, fp.off);
  fp.base = a.workspace + (size_t)tiles * a.B * a.num_scales * DD_PARTIAL_STRIDE;
  fp.idrho = AUTOMASK ? fp.base + fp_floats : nullptr;
  if (AUTOMASK) {
    hipLaunchKernelGGL(photo_identity_kernel, dim3(tiles, a.B), dim3(NT), (sizeof(f2) + sizeof(float)) * 3 * R2N, stream, a.target, a.source[0],
                       a.source[1], a.H, a.W, a.ssim_weight, fp.base + fp_floats);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  hipEvent_t* timed = nullptr;
  const bool timing = GRAD && timer_slot(stream, timed);
  if (timing) (void)hipEventRecord(timed[0], stream);
  hipLaunchKernelGGL(kern, grid, dim3(NT), lds_bytes(SMOOTH), stream, a, depth_params(a.min_depth, a.max_depth), image_dims(a.W, a.H), fp, fuse, side);
  if (timing) (void)hipEventRecord(timed[1], stream);
  return (int)hipGetLastError();
}

// part: 0 = every launch, 1 = the tile kernel alone, 2 = what follows it (dd_photo_loss_part)
template <int MODE, bool AUTOMASK, bool GRAD, bool SHARED, bool OUT>
static int launch_photo(const DDPhotoArgs& a, hipStream_t stream, int part) {
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH, tiles = tiles_x * tiles_y;
  FootprintInfo fp;
  footprint_floats(a, fp.off);
  fp.base = a.workspace + (size_t)tiles * a.B * a.num_scales * DD_PARTIAL_STRIDE;
  fp.idrho = nullptr;
  hipError_t e = hipSuccess;
  if (part != 2) {
    FuseInfo none;
    memset(&none, 0, sizeof(none));
    SideInfo no_side;
    memset(&no_side, 0, sizeof(no_side));
    const int rc = launch_tile<MODE, AUTOMASK, GRAD, SHARED, OUT, false>(a, none, no_side, stream);
    if (rc) return rc;
  }
  if (part == 1) return 0;
  if (GRAD) {
    int max_n = 0;
    for (int s = 0; s < a.num_scales; ++s)
      if (a.scale[s].shift > 0) max_n = max(max_n, a.scale[s].h * a.scale[s].w);
    if (max_n > 0) {
      constexpr int NCH = Channels<MODE, SHARED>::N;
      hipLaunchKernelGGL((photo_combine_kernel<NCH>), dim3((max_n + 255) / 256, a.B, a.num_scales), dim3(256), 0, stream, a, fp, tiles_x, tiles_y);
      e = hipGetLastError();
      if (e != hipSuccess) return (int)e;
    }
  }
  hipLaunchKernelGGL(photo_finalize_kernel, dim3(a.num_scales + a.B), dim3(256), 0, stream, a.workspace, a.num_scales,
                     a.B, tiles, a.sums, a.want_grad ? a.g_T[0] : nullptr, a.want_grad ? a.g_T[1] : nullptr);
  return (int)hipGetLastError();
}

// dd_fused_loss: the gradient pass with the scale-0 smoothness in the store stage (dd_fuse.h)
template <int MODE, bool AUTOMASK, bool SHARED>
static int launch_tile_fused_g(const DDPhotoArgs& a, const FuseInfo& fuse, const SideInfo& side, hipStream_t stream) {
  return wants_outputs(a) ? launch_tile<MODE, AUTOMASK, true, SHARED, true, true>(a, fuse, side, stream)
                          : launch_tile<MODE, AUTOMASK, true, SHARED, false, true>(a, fuse, side, stream);
}

static int photo_args_ok(const DDPhotoArgs* a) {
  if (!a || a->abi_version != DD_ABI_VERSION) return 0;
  if (a->num_scales < 1 || a->num_scales > DD_MAX_SCALES || a->B < 1 || !a->workspace || !a->sums) return 0;
  if (a->H < 4 || a->W < 4 || a->H > 4096 || a->W > 4096) return 0;   // 24-bit index products, fp32-exact plane offsets
  for (int s = 0; s < a->num_scales; ++s) {
    const DDPhotoScale& sc = a->scale[s];
    if (sc.shift < 0 || sc.shift > 3 || sc.h != (a->H >> sc.shift) || sc.w != (a->W >> sc.shift)) return 0;
    if ((a->H % (1 << sc.shift)) || (a->W % (1 << sc.shift))) return 0;
  }
  return 1;
}

int launch_tile_fused(const DDPhotoArgs& a, const FuseInfo& fuse, const SideInfo& side, hipStream_t stream) {
  if (!photo_args_ok(&a) || !a.want_grad) return (int)hipErrorInvalidValue;
  const bool sh = frames_share_tensors(a);
  switch (a.mode) {
    case DD_MODE_RIGID:
      return a.automask ? launch_tile_fused_g<MODE_RIGID, true, false>(a, fuse, side, stream) : launch_tile_fused_g<MODE_RIGID, false, false>(a, fuse, side, stream);
    case DD_MODE_FLOW:
      if (a.automask || !sh) return (int)hipErrorInvalidValue;
      return launch_tile_fused_g<MODE_FLOW, false, true>(a, fuse, side, stream);
    case DD_MODE_FLOW_MASK:
      if (a.automask || !sh) return (int)hipErrorInvalidValue;
      return launch_tile_fused_g<MODE_FLOW_MASK, false, true>(a, fuse, side, stream);
    default:
      return (int)hipErrorInvalidValue;
  }
}

template <int MODE, bool AUTOMASK, bool SHARED>
static int launch_photo_g(const DDPhotoArgs& a, hipStream_t stream, int part) {
  if (wants_outputs(a))
    return a.want_grad ? launch_photo<MODE, AUTOMASK, true, SHARED, true>(a, stream, part) : launch_photo<MODE, AUTOMASK, false, SHARED, true>(a, stream, part);
  return a.want_grad ? launch_photo<MODE, AUTOMASK, true, SHARED, false>(a, stream, part) : launch_photo<MODE, AUTOMASK, false, SHARED, false>(a, stream, part);
}

}  // namespace dd

#ifdef DD_STAGE_TIMING
// debug build only (make TIMING=1): cumulative shader cycles of thread 0 per stage, summed over all blocks
extern "C" int dd_debug_stage_cycles(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(dd::g_stage_cycles), 8 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(dd::g_stage_cycles), z, sizeof(z));
  }
  return (int)e;
}
#endif

extern "C" size_t dd_photo_workspace_bytes(const DDPhotoArgs* a) {
  const size_t tiles = (size_t)((a->W + dd::TW - 1) / dd::TW) * ((a->H + dd::TH - 1) / dd::TH);
  long long off[DD_MAX_SCALES];
  const size_t identity = a->automask ? (size_t)a->B * 2 * a->H * a->W : 0;        // photo_identity_kernel's output
  return (tiles * a->B * a->num_scales * DD_PARTIAL_STRIDE + dd::footprint_floats(*a, off) + identity) * sizeof(float);
}

extern "C" int dd_photo_timing(int enable) {
  dd::g_timer.on = enable != 0;
  dd::g_timer.n = 0;
  return 0;
}

extern "C" int dd_photo_timing_read(float* mean_us, int* launches, int skip) {
  using dd::g_timer;
  double tot = 0.0;
  int cnt = 0;
  for (int i = skip < 0 ? 0 : skip; i < g_timer.n; ++i) {
    hipError_t e = hipEventSynchronize(g_timer.ev[i][1]);
    if (e != hipSuccess) return (int)e;
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, g_timer.ev[i][0], g_timer.ev[i][1]);
    if (e != hipSuccess) return (int)e;
    tot += ms;
    ++cnt;
  }
  if (mean_us) *mean_us = cnt ? (float)(tot / cnt * 1e3) : 0.f;
  if (launches) *launches = cnt;
  g_timer.n = 0;
  return 0;
}

static int photo_loss_impl(const DDPhotoArgs* a, void* stream_, int part) {
  using namespace dd;
  if (!photo_args_ok(a)) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const bool sh = frames_share_tensors(*a);
  switch (a->mode) {
    case DD_MODE_RIGID:
      return a->automask ? launch_photo_g<MODE_RIGID, true, false>(*a, stream, part) : launch_photo_g<MODE_RIGID, false, false>(*a, stream, part);
    case DD_MODE_FLOW:
      if (a->automask) return (int)hipErrorInvalidValue;
      return sh ? launch_photo_g<MODE_FLOW, false, true>(*a, stream, part) : launch_photo_g<MODE_FLOW, false, false>(*a, stream, part);
    case DD_MODE_FLOW_MASK:
      if (a->automask) return (int)hipErrorInvalidValue;
      return sh ? launch_photo_g<MODE_FLOW_MASK, false, true>(*a, stream, part) : launch_photo_g<MODE_FLOW_MASK, false, false>(*a, stream, part);
    default:
      return (int)hipErrorInvalidValue;
  }
}

extern "C" int dd_photo_loss(const DDPhotoArgs* a, void* stream) { return photo_loss_impl(a, stream, 0); }

extern "C" int dd_photo_loss_part(const DDPhotoArgs* a, void* stream, int part) {
  if (part < 0 || part > 2) return (int)hipErrorInvalidValue;
  return photo_loss_impl(a, stream, part);
}
