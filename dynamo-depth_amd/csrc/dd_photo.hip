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
//   R  DPP wave64 + LDS reduction of the loss sums and of d(loss)/dT -> one record per block.
// photo_finalize_kernel folds the per-block records deterministically.  No float atomics anywhere.
//
// Replaces Trainer.generate_images_pred + the photometric part of Trainer.compute_losses and their
// autograd (reference Trainer.py:215-352,384-386,413-423; tools.py:191-257,291-298) -- thousands of
// ATen launches per step in the reference.  The arithmetic is dd_math.h.
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_math.h"

namespace dd {

#ifndef DD_TH
#define DD_TH 16         // 16x32 tiles = 512 threads, ~68 KB LDS: two workgroups per CU (16 waves) whose barrier-separated
#define DD_TW 32         // stages interleave; measured 5-9 % faster than one 16x64 / 1024-thread workgroup per CU
#define DD_MIN_WAVES 4   // <= 128 VGPRs so that both workgroups fit
#endif
constexpr int TH = DD_TH;         // tile height (target pixels); multiple of 8 (coarsest scale block)
constexpr int TW = DD_TW;         // tile width
constexpr int NT = TH * TW;       // one thread per target pixel: 1024 threads = 16 waves = 4 per SIMD
constexpr int RH = TH + 4, RW = TW + 4;       // region with 2-pixel halo (warped colours, target)
constexpr int R2N = RH * RW;
constexpr int CH_ = TH + 2, CW_ = TW + 2;     // centres with 1-pixel halo (SSIM, selection, coefficients)
constexpr int R1N = CH_ * CW_;
constexpr int RING = R2N - TH * TW;
constexpr int FPW_MAX = TW / 2 + 2;                          // widest low-res footprint of a tile (scale 1)

constexpr int LRN_MAX = (TH / 2) * (TW / 2);                  // low-res pixels inside a tile at scale >= 1
constexpr int NWAVES = NT / 64;
constexpr int NRED = 30;          // photo, n_warp, cons[2], delta[2], gT[2][12]

constexpr int LOWH = RH / 2 + 2, LOWW = RW / 2 + 2;           // staged low-res region (scale >= 1) incl. halo taps
constexpr int LOWN = LOWH * LOWW;
static_assert(2 * RING <= NT, "one pass over the halo ring, one (pixel, frame) item per thread");
static_assert(TH % 8 == 0 && TW % 8 == 0 && NT % 64 == 0 && NT <= 1024, "tile shape");

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

// SSIM + L1 of both frames at one centre, from LDS planes.  cf != nullptr also returns the backward
// coefficients (d ssim/d mean_x, 2 d ssim/d mean_xx, d ssim/d mean_xy per channel).
template <bool WITH_GRAD>
__device__ __forceinline__ void rho_pair(const float* __restrict__ s_x, const float* __restrict__ s_y, const int ry[3],
                                         const int rx[3], int centre, float alpha, float rho[2], float cf[2][9]) {
  float ssum[2] = {0.f, 0.f}, l1[2] = {0.f, 0.f};
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float* yp = s_y + ch * R2N;
    float yv[9];
    float sy = 0.f, syy = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float v = yp[ry[j] + rx[i]];
        yv[j * 3 + i] = v;
        sy += v;
        syy += v * v;
      }
    const float yc = yp[centre];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const float* xp = s_x + (f * 3 + ch) * R2N;
      SsimStats st;
      st.sx = 0.f; st.sxx = 0.f; st.sxy = 0.f; st.sy = sy; st.syy = syy;
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float v = xp[ry[j] + rx[i]];
          st.sx += v;
          st.sxx += v * v;
          st.sxy += v * yv[j * 3 + i];
        }
      if (WITH_GRAD) {
        SsimGrad sg;
        ssum[f] += ssim_value(st, &sg);
        cf[f][ch * 3 + 0] = sg.dmu;
        cf[f][ch * 3 + 1] = 2.f * sg.dxx;
        cf[f][ch * 3 + 2] = sg.dxy;
      } else {
        ssum[f] += ssim_value(st, nullptr);
      }
      l1[f] += dd_abs(yc - xp[centre]);
    }
  }
#pragma unroll
  for (int f = 0; f < 2; ++f) rho[f] = alpha * (ssum[f] * (1.f / 3.f)) + (1.f - alpha) * (l1[f] * (1.f / 3.f));
}

// does full-res coordinate c take part in the align_corners=False bilinear down-sampling by 2^shift?
__device__ __forceinline__ bool down_tap(int c, int shift) {
  const int blk = 1 << shift, r = c & (blk - 1);
  return (r == (blk >> 1) - 1) || (r == (blk >> 1));
}

// LDS.  Regions are reused across stages (the barriers in the kernel body separate the lifetimes):
//   pred+tgt  : warped colours / target colours (stages 0..C1)  ->  per-pixel gradient planes G (stage C2, scale >= 1)
//   coef      : backward coefficients (stages B..C1)             ->  x-reduced gradient planes Hx (stage C3)
struct LdsLayout {
  float pred[2 * 3 * R2N];       // warped source colours (identity copies during the automask pre-pass)
  float tgt[3 * R2N];            // target colours
  float coef[9 * R1N];           // backward coefficients of the selected frame
  int sel[R1N];                  // selected frame per centre (-1: identity won / outside the image)
  float idmin[R1N];              // automask: min over frames of the identity reprojection loss (+noise)
  float lr[2 * 2 * 5 * LRN_MAX]; // low-res residual flow (3) + grid difference (2) per frame; two row slots (upper / lower tap row)
  float low[9 * LOWN];           // staged low-res inputs (scale >= 1): disp, flow[2][3], mask[2]
  float red[NWAVES * NRED];
};
static_assert(9 * TH * TW <= (2 * 3 + 3) * R2N, "gradient planes must fit into pred+tgt");
static_assert(9 * TH * FPW_MAX <= 9 * R1N, "x-reduced planes must fit into coef");

// bilinear up-sampling taps of one full-res pixel, as offsets into a staged LOWH x LOWW region
// Per-tile low-res gradient footprints (scale >= 1) go to the workspace with plain stores; photo_combine_kernel sums the
// <= 4 tiles that overlap each low-res pixel in a fixed order: deterministic gradients, no device atomics.
struct FootprintInfo {
  float* base;                    // workspace area behind the per-block records
  long long off[DD_MAX_SCALES];   // float offset of scale si (unused for shift == 0)
};

struct LowTap {
  int o00, o01, o10, o11;
  float wx0, wx1, wy0, wy1;
};

__device__ __forceinline__ float low_eval(const float* __restrict__ plane, const LowTap& t) {
  // same association as ATen: wy0*(wx0*a + wx1*b) + wy1*(wx0*c + wx1*d)
  return t.wy0 * (t.wx0 * plane[t.o00] + t.wx1 * plane[t.o01]) + t.wy1 * (t.wx0 * plane[t.o10] + t.wx1 * plane[t.o11]);
}

#ifndef DD_MIN_WAVES
#define DD_MIN_WAVES 1
#endif
template <int MODE, bool AUTOMASK, bool GRAD>
__global__ __launch_bounds__(NT, DD_MIN_WAVES) void photo_tile_kernel(const DDPhotoArgs a, const DepthParams dp, const FootprintInfo fp) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  LdsLayout& S = *reinterpret_cast<LdsLayout*>(smem_raw);

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int si = blockIdx.z;
  const DDPhotoScale& sc = a.scale[si];
  const int H = a.H, W = a.W, N = H * W;
  const int tiles_x = (W + TW - 1) / TW;
  // XCD-aware tile order: workgroup i runs on XCD i % 8 (each XCD has its own L2).  Give every XCD a contiguous
  // band of the image so that neighbouring tiles -- which re-read each other's 2-pixel halo and the same source rows --
  // share an L2 instead of each missing separately.  Pure permutation of blockIdx.x: correctness does not depend on it.
  int tile = blockIdx.x;
  {
    const int ntiles = gridDim.x, per = (ntiles + 7) >> 3;
    const int remapped = (tile & 7) * per + (tile >> 3);
    if ((ntiles & 7) == 0) tile = remapped;
  }
  const int X0 = (tile % tiles_x) * TW, Y0 = (tile / tiles_x) * TH;
  const int shift = sc.shift, h = sc.h, w = sc.w, n = h * w;
  const float ratio = 1.f / static_cast<float>(1 << shift);
  const float alpha = a.ssim_weight;
  const ImageDims dim = image_dims(W, H);

  Intrinsics cam;
  load_intrinsics(cam, a.K + b * 16, a.inv_K + b * 16);
  const float* Tm[2] = {a.T[0] + b * 16, a.T[1] + b * 16};
  float tsv[2] = {1.f, 1.f};
  if (MODE != MODE_RIGID) {
#pragma unroll
    for (int f = 0; f < 2; ++f) tsv[f] = a.ts[f] ? a.ts[f][b] : 1.f;
  }
  const float* tgt_g = a.target + (size_t)b * 3 * N;
  const float* src_g[2] = {a.source[0] + (size_t)b * 3 * N, a.source[1] + (size_t)b * 3 * N};
  const float* disp_g = sc.disp + (size_t)b * n;
  constexpr int NCH = 1 + (MODE != MODE_RIGID ? 6 : 0) + (MODE == MODE_FLOW_MASK ? 2 : 0);   // gradient channels
  constexpr int NPL = NCH;                                                                     // staged low-res planes
  // plane p of the low-res inputs: 0 disp | 1..3 flow f0 | 4..6 flow f1 | 7 mask f0 | 8 mask f1
  auto plane_ptr = [&](int p) -> const float* {
    if (p == 0) return disp_g;
    if (p < 7) return sc.flow[(p - 1) / 3] + ((size_t)b * 3 + (p - 1) % 3) * n;
    return sc.mask[p - 7] + (size_t)b * n;
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
  // ---- stage 0: stage the target region and (scale >= 1) the low-res inputs in LDS ---------------------
  for (int i = tid; i < R2N; i += NT) {
    const int Y = Y0 - 2 + i / RW, X = X0 - 2 + i % RW;
    const bool in = (Y >= 0) && (Y < H) && (X >= 0) && (X < W);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) S.tgt[ch * R2N + i] = in ? tgt_g[(size_t)ch * N + Y * W + X] : 0.f;
    if (AUTOMASK) {
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
          S.pred[(f * 3 + ch) * R2N + i] = in ? src_g[f][(size_t)ch * N + Y * W + X] : 0.f;
    }
  }
  if (shift > 0) {
    for (int i = tid; i < NPL * LOWN; i += NT) {
      const int p = i / LOWN, r = i - p * LOWN;
      const int qy = lfy0 + r / LOWW, qx = lfx0 + r % LOWW;
      S.low[i] = (qy < h && qx < w) ? plane_ptr(p)[qy * w + qx] : 0.f;
    }
  }
  __syncthreads();
  DD_STAGE_MARK(0);

  // ---- automask pre-pass: identity reprojection loss at every centre -------------------------------
  if (AUTOMASK) {
    for (int i = tid; i < R1N; i += NT) {
      const int cy = i / CW_, cx = i % CW_;
      const int Y = Y0 - 1 + cy, X = X0 - 1 + cx;
      float v = 0.f;
      if (Y >= 0 && Y < H && X >= 0 && X < W) {
        int ry[3], rx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          ry[d] = (dd_reflect(Y + d - 1, H) - (Y0 - 2)) * RW;
          rx[d] = dd_reflect(X + d - 1, W) - (X0 - 2);
        }
        float rho[2];
        rho_pair<false>(S.pred, S.tgt, ry, rx, (cy + 1) * RW + cx + 1, alpha, rho, nullptr);
        if (sc.noise) {
          rho[0] += sc.noise[((size_t)b * 2 + 0) * N + Y * W + X] * 0.00001f;
          rho[1] += sc.noise[((size_t)b * 2 + 1) * N + Y * W + X] * 0.00001f;
        }
        v = rho[1] < rho[0] ? rho[1] : rho[0];
      }
      S.idmin[i] = v;
    }
    __syncthreads();
    DD_STAGE_MARK(1);
  }

  // ---- stage A: geometry + warp ---------------------------------------------------------------------
  // value of low-res plane `pl` at full-res pixel (X,Y): identity at scale 0 (one coalesced load), LDS taps otherwise
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
  auto lowres = [&](int pl, const LowTap& t, int p) -> float {
    return shift == 0 ? plane_ptr(pl)[p] : low_eval(S.low + pl * LOWN, t);
  };
  // geometry of one (pixel, frame): returns the sample coordinate, fills g
  auto frame_geo = [&](int f, const LowTap& t, int p, const float P[3], FrameGeom& g, float& m_out) -> SampleCoord {
    float c[3] = {0.f, 0.f, 0.f}, m = 1.f;
    if (MODE != MODE_RIGID) {
#pragma unroll
      for (int k = 0; k < 3; ++k) c[k] = lowres(1 + f * 3 + k, t, p) * tsv[f];
    }
    if (MODE == MODE_FLOW_MASK) m = lowres(7 + f, t, p);
    m_out = m;
    frame_geometry<MODE>(cam, Tm[f], P, c, m, dim, a.eps, g);
    return sample_coord(g.gnx, g.gny, W, H);
  };

  // owner state (one pixel per thread)
  const int lx = tid % TW, ly = tid / TW;
  const int oX = X0 + lx, oY = Y0 + ly;
  const bool own = (oX < W) && (oY < H);
  const int op = oY * W + oX;
  float Zs = 0.f, mval[2] = {1.f, 1.f}, xval[2][3], dvx[2][3], dvy[2][3];
  FrameGeom geo[2];
  float acc_cons[2] = {0.f, 0.f}, acc_delta[2] = {0.f, 0.f};
  float lrv[2][5] = {{0.f, 0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f, 0.f}};   // this pixel's share of the low-res residual / grid difference

  if (own) {
    LowTap t;
    if (shift > 0) t = make_tap(oX, oY);
    const float d = lowres(0, t, op);
    const float Z = dd_rcp(dp.lo + dp.span * d);
    Zs = Z;
    float ray[3], P[3];
    pixel_ray(cam, oX, oY, ray);
#pragma unroll
    for (int k = 0; k < 3; ++k) P[k] = Z * ray[k];
    if (sc.out_depth) sc.out_depth[(size_t)b * N + op] = Z;
    SampleCoord scd[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) scd[f] = frame_geo(f, t, op, P, geo[f], mval[f]);
    // all 24 source taps are issued before any is consumed (memory-level parallelism)
    const int li = (oY - (Y0 - 2)) * RW + (oX - (X0 - 2));
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        xval[f][ch] = sample_plane(src_g[f] + (size_t)ch * N, scd[f], W, H, dvx[f][ch], dvy[f][ch]);
        S.pred[(f * 3 + ch) * R2N + li] = xval[f][ch];
      }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const FrameGeom& g = geo[f];
      if (sc.out_color[f]) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) sc.out_color[f][((size_t)b * 3 + ch) * N + op] = xval[f][ch];
      }
      if (sc.out_sample[f]) reinterpret_cast<float2*>(sc.out_sample[f])[(size_t)b * N + op] = make_float2(g.gnx, g.gny);
      if (MODE == MODE_FLOW_MASK) {
        if (shift == 0) {
          // the low-res pixel IS this pixel: c_consistency and disp_mag directly
          const float valid = disp_g[op] > a.disp_thr ? 1.f : 0.f;
          const float om = 1.f - sc.mask[f][(size_t)b * n + op];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            acc_cons[f] += valid * om * dd_abs(g.r[k]);
            if (sc.out_resid[f]) sc.out_resid[f][((size_t)b * 3 + k) * n + op] = g.r[k];
          }
          const float dx = g.ego_gn[0] - g.cmp_gn[0], dy = g.ego_gn[1] - g.cmp_gn[1];
          const float delta = dx * dx + dy * dy;
          acc_delta[f] += delta;
          if (sc.out_delta[f]) sc.out_delta[f][(size_t)b * n + op] = delta;
        } else if (down_tap(oX, shift) && down_tap(oY, shift)) {
#pragma unroll
          for (int k = 0; k < 3; ++k) lrv[f][k] = 0.25f * g.r[k];
          lrv[f][3] = 0.25f * (g.ego_gn[0] - g.cmp_gn[0]);
          lrv[f][4] = 0.25f * (g.ego_gn[1] - g.cmp_gn[1]);
        }
      }
    }
  }
  if (MODE == MODE_FLOW_MASK && shift > 0) {
    // align_corners=False down-sampling by 2^s = mean of the 2x2 centre pixels of each block.  The two taps of a row
    // are adjacent lanes (one shuffle, all lanes take part); the two rows go to two LDS slots that stage L adds in a
    // fixed order: deterministic and free of LDS atomics.
    const bool left = own && ((oX & ((1 << shift) - 1)) == (1 << (shift - 1)) - 1) && down_tap(oY, shift);
    const int slot = (oY & ((1 << shift) - 1)) == (1 << (shift - 1)) ? 1 : 0;
    const int q = (((oY - Y0) >> shift) * lrw) + ((oX - X0) >> shift);
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const float pair = lrv[f][k] + __shfl_down(lrv[f][k], 1, 64);
        if (left) S.lr[((slot * 2 + f) * 5 + k) * LRN_MAX + q] = pair;
      }
  }
  // halo ring: one (pixel, frame) item per thread, forward only
  if (tid < 2 * RING) {
    const int f = tid >= RING ? 1 : 0;
    const int r = tid - f * RING;
    int ry, rx;
    if (r < 2 * RW) { ry = r / RW; rx = r % RW; }
    else if (r < 4 * RW) { const int r2 = r - 2 * RW; ry = RH - 2 + r2 / RW; rx = r2 % RW; }
    else { const int r3 = r - 4 * RW; ry = 2 + (r3 >> 2); const int k = r3 & 3; rx = k < 2 ? k : RW - 4 + k; }
    const int Y = Y0 - 2 + ry, X = X0 - 2 + rx;
    if (Y >= 0 && Y < H && X >= 0 && X < W) {
      const int p = Y * W + X;
      LowTap t;
      if (shift > 0) t = make_tap(X, Y);
      const float Z = dd_rcp(dp.lo + dp.span * lowres(0, t, p));
      float ray[3], P[3];
      pixel_ray(cam, X, Y, ray);
#pragma unroll
      for (int k = 0; k < 3; ++k) P[k] = Z * ray[k];
      FrameGeom g;
      float m_unused;
      const SampleCoord scd = f == 0 ? frame_geo(0, t, p, P, g, m_unused) : frame_geo(1, t, p, P, g, m_unused);
      const float* sp = f == 0 ? src_g[0] : src_g[1];
      const int li = ry * RW + rx;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float dxu, dyu;
        S.pred[(f * 3 + ch) * R2N + li] = sample_plane(sp + (size_t)ch * N, scd, W, H, dxu, dyu);
      }
    }
  }
  __syncthreads();
  DD_STAGE_MARK(2);

  // ---- stage B: SSIM + L1, selection, loss, backward coefficients ------------------------------------
  float acc_photo = 0.f, acc_nwarp = 0.f;
  for (int i = tid; i < R1N; i += NT) {
    const int cy = i / CW_, cx = i % CW_;
    const int Y = Y0 - 1 + cy, X = X0 - 1 + cx;
    int bf = -1;
    float cf[2][9];
    if (Y >= 0 && Y < H && X >= 0 && X < W) {
      int ry[3], rx[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        ry[d] = (dd_reflect(Y + d - 1, H) - (Y0 - 2)) * RW;
        rx[d] = dd_reflect(X + d - 1, W) - (X0 - 2);
      }
      float rho[2];
      rho_pair<GRAD>(S.pred, S.tgt, ry, rx, (cy + 1) * RW + cx + 1, alpha, rho, cf);
      float best = rho[0];
      bf = 0;
      if (rho[1] < best) { best = rho[1]; bf = 1; }
      if (AUTOMASK) {
        const float idb = S.idmin[i];
        if (idb <= best) { best = idb; bf = -1; }   // identity entries precede the warped ones in the cat: ties go to them
      }
      const bool interior = (cy >= 1) && (cy <= TH) && (cx >= 1) && (cx <= TW);
      if (interior) {
        acc_photo += best;
        acc_nwarp += bf >= 0 ? 1.f : 0.f;
        if (AUTOMASK && sc.out_idsel) sc.out_idsel[(size_t)b * N + Y * W + X] = bf >= 0 ? 1.f : 0.f;
      }
    }
    if (GRAD) {
      S.sel[i] = bf;
      const float wgt = sc.w_photo * alpha * (1.f / 27.f);
#pragma unroll
      for (int k = 0; k < 9; ++k) S.coef[k * R1N + i] = bf >= 0 ? wgt * (bf == 0 ? cf[0][k] : cf[1][k]) : 0.f;
    }
  }

  // ---- stage L: c_consistency and disp_mag on the tile's low-res pixels (scale >= 1) -------------------
  if (MODE == MODE_FLOW_MASK && shift > 0) {
    for (int q = tid; q < lrh * lrw; q += NT) {
      const int qy = (Y0 >> shift) + q / lrw, qx = (X0 >> shift) + q % lrw;
      if (qy < h && qx < w) {
        const int gq = qy * w + qx;
        const float valid = disp_g[gq] > a.disp_thr ? 1.f : 0.f;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const float om = 1.f - sc.mask[f][(size_t)b * n + gq];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const float rv = S.lr[(f * 5 + k) * LRN_MAX + q] + S.lr[((2 + f) * 5 + k) * LRN_MAX + q];
            acc_cons[f] += valid * om * dd_abs(rv);
            S.lr[(f * 5 + k) * LRN_MAX + q] = sc.w_cons * valid * om * dd_sign(rv);
            if (sc.out_resid[f]) sc.out_resid[f][((size_t)b * 3 + k) * n + gq] = rv;
          }
          const float dx = S.lr[(f * 5 + 3) * LRN_MAX + q] + S.lr[((2 + f) * 5 + 3) * LRN_MAX + q];
          const float dy = S.lr[(f * 5 + 4) * LRN_MAX + q] + S.lr[((2 + f) * 5 + 4) * LRN_MAX + q];
          const float delta = dx * dx + dy * dy;
          acc_delta[f] += delta;
          if (sc.out_delta[f]) sc.out_delta[f][(size_t)b * n + gq] = delta;
        }
      }
    }
  }
  __syncthreads();
  DD_STAGE_MARK(3);

  // ---- stage C: backward ------------------------------------------------------------------------------
  float gTacc[2][12];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int k = 0; k < 12; ++k) gTacc[f][k] = 0.f;

  if (GRAD) {
    float gch[NCH];           // d loss / d (up-sampled disp, flow[2][3], mask[2]) of this pixel
#pragma unroll
    for (int k = 0; k < NCH; ++k) gch[k] = 0.f;
    if (own) {
      const int X = oX, Y = oY;
      // Adjoint of the reflect-padded 3x3 box filter, gather form.  Centres outside the image carry sel = -1
      // (no bounds tests needed); a border centre counts its inner neighbour twice (reflection), which is the
      // closed-form weight below.  The selection mask and the multiplicity fold into one factor per centre.
      float Sc[2][9];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int k = 0; k < 9; ++k) Sc[f][k] = 0.f;
      const float wy_lo = Y == 1 ? 2.f : 1.f, wy_hi = Y == H - 2 ? 2.f : 1.f;
      const float wxv[3] = {X == 1 ? 2.f : 1.f, 1.f, X == W - 2 ? 2.f : 1.f};
      const int ci0 = (Y - (Y0 - 1)) * CW_ + (X - (X0 - 1));
#pragma unroll 1
      for (int dy = 0; dy < 3; ++dy)          // rolled on purpose: full unrolling keeps 81 LDS values in flight and spills
#pragma unroll
        for (int dxx = 0; dxx < 3; ++dxx) {
          const int ci = ci0 + (dy - 1) * CW_ + (dxx - 1);
          const int sl = S.sel[ci];
          const float wgt = (dy == 0 ? wy_lo : (dy == 2 ? wy_hi : 1.f)) * wxv[dxx];
          const float m0 = sl == 0 ? wgt : 0.f, m1 = sl == 1 ? wgt : 0.f;
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            const float v = S.coef[k * R1N + ci];
            Sc[0][k] = fmaf(m0, v, Sc[0][k]);
            Sc[1][k] = fmaf(m1, v, Sc[1][k]);
          }
        }
      const int own_sel = S.sel[ci0];
      const int li = (Y - (Y0 - 2)) * RW + (X - (X0 - 2));
      float ray[3], P[3], gPtot[3] = {0.f, 0.f, 0.f};
      pixel_ray(cam, X, Y, ray);
#pragma unroll
      for (int k = 0; k < 3; ++k) P[k] = Zs * ray[k];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float gu = 0.f, gv = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float xv = S.pred[(f * 3 + ch) * R2N + li], yv = S.tgt[ch * R2N + li];   // re-read: frees six registers across stage B
          float gx = Sc[f][ch * 3 + 0] + xv * Sc[f][ch * 3 + 1] + yv * Sc[f][ch * 3 + 2];
          if (own_sel == f) gx += sc.w_photo * (1.f - alpha) * (1.f / 3.f) * dd_sign(xv - yv);
          gu += gx * dvx[f][ch];
          gv += gx * dvy[f][ch];
        }
        float gr_extra[3] = {0.f, 0.f, 0.f};
        if (MODE == MODE_FLOW_MASK) {
          if (shift == 0) {
            const float vm = (disp_g[op] > a.disp_thr ? sc.w_cons : 0.f) * (1.f - sc.mask[f][(size_t)b * n + op]);
#pragma unroll
            for (int k = 0; k < 3; ++k) gr_extra[k] = vm * dd_sign(geo[f].r[k]);
          } else if (down_tap(X, shift) && down_tap(Y, shift)) {
            const int q = (((Y - Y0) >> shift) * lrw) + ((X - X0) >> shift);
#pragma unroll
            for (int k = 0; k < 3; ++k) gr_extra[k] = 0.25f * S.lr[(f * 5 + k) * LRN_MAX + q];
          }
        }
        PixelGrad pg;
        frame_geometry_bwd<MODE>(cam, Tm[f], P, mval[f], geo[f], gu, gv, gr_extra, pg);
#pragma unroll
        for (int k = 0; k < 3; ++k) gPtot[k] += pg.gP[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) gTacc[f][k] += pg.gT[k];
        if (MODE != MODE_RIGID) {
#pragma unroll
          for (int k = 0; k < 3; ++k) gch[1 + f * 3 + k] = pg.gc[k] * tsv[f];
        }
        if (MODE == MODE_FLOW_MASK) gch[7 + f] = pg.gm;
      }
      gch[0] = depth_bwd(dp, gPtot, ray, Zs);
    }
    auto grad_ptr = [&](int ch) -> float* {
      if (ch == 0) return sc.g_disp + (size_t)b * n;
      if (ch < 7) return sc.g_flow[(ch - 1) / 3] + ((size_t)b * 3 + (ch - 1) % 3) * n;
      return sc.g_mask[ch - 7] + (size_t)b * n;
    };
    if (shift == 0) {
      // up-sampling is the identity and every element has exactly one owner: plain coalesced stores into the
      // (zeroed) gradient buffers.  A device-scope float atomic would cost one 32-64 B fabric write per 4 useful
      // bytes (measured: WRITE_SIZE 4.3x the algorithmic bytes).  Buffers that the caller aliases between the two
      // frames (the shared motion mask) receive the sum.
      if (own) {
        grad_ptr(0)[op] = gch[0];
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
    } else {
      // Adjoint of the bilinear up-sampling WITHOUT atomics on the LDS: park the per-pixel gradients, then every
      // low-res footprint element gathers its contributions, x first (separable), then y.
      float* G = S.pred;                  // [NCH][TH*TW], spans pred+tgt
      float* Hx = S.coef;                 // [NCH][TH][FPW_MAX]
      __syncthreads();                    // all reads of pred/tgt/coef/sel are done
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) G[ch * (TH * TW) + tid] = gch[ch];
      __syncthreads();
      DD_STAGE_MARK(4);
      const int blk = 1 << shift;
      for (int i = tid; i < TH * fpw; i += NT) {
        const int r = i / fpw, j = i - r * fpw;
        const int q = fx0 + j;
        float acc[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) acc[ch] = 0.f;
        const int xa = max(X0, blk * q - (blk >> 1)), xb = min(min(X0 + TW, W), blk * q + 3 * (blk >> 1));
        for (int X = xa; X < xb; ++X) {
          const Tap1 t = resize_tap(X, w, ratio);
          const float wt = (t.i0 == q ? t.w0 : 0.f) + (t.i1 == q ? t.w1 : 0.f);
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) acc[ch] = fmaf(wt, G[ch * (TH * TW) + r * TW + (X - X0)], acc[ch]);
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) Hx[(ch * TH + r) * FPW_MAX + j] = acc[ch];
      }
      __syncthreads();
      for (int i = tid; i < fph * fpw; i += NT) {
        const int jy = i / fpw, j = i - jy * fpw;
        const int qy = fy0 + jy, qx = fx0 + j;
        if (qy >= h || qx >= w) continue;            // never read by the combine pass
        float acc[NCH];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) acc[ch] = 0.f;
        const int ya = max(Y0, blk * qy - (blk >> 1)), yb = min(min(Y0 + TH, H), blk * qy + 3 * (blk >> 1));
        for (int Y = ya; Y < yb; ++Y) {
          const Tap1 t = resize_tap(Y, h, ratio);
          const float wt = (t.i0 == qy ? t.w0 : 0.f) + (t.i1 == qy ? t.w1 : 0.f);
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) acc[ch] = fmaf(wt, Hx[(ch * TH + (Y - Y0)) * FPW_MAX + j], acc[ch]);
        }
        // the footprint rim overlaps the neighbouring tiles' footprints: photo_combine_kernel adds them up
        float* dst = fp.base + fp.off[si] + ((size_t)(b * gridDim.x + tile) * NCH) * (fph * fpw) + jy * fpw + j;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) dst[(size_t)ch * (fph * fpw)] = acc[ch];
      }
    }
  }

  DD_STAGE_MARK(5);
  // ---- stage R: block reduction -> one record per block ---------------------------------------------
  float vals[NRED];
  vals[0] = acc_photo; vals[1] = acc_nwarp;
  vals[2] = acc_cons[0]; vals[3] = acc_cons[1]; vals[4] = acc_delta[0]; vals[5] = acc_delta[1];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int k = 0; k < 12; ++k) vals[6 + f * 12 + k] = gTacc[f][k];
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < NRED; ++k) {
    const float r = wave_sum(vals[k]);
    if (lane == 0) S.red[wave * NRED + k] = r;
  }
  __syncthreads();
  if (tid < NRED) {
    float r = 0.f;
#pragma unroll
    for (int wv = 0; wv < NWAVES; ++wv) r += S.red[wv * NRED + tid];
    const size_t rec = ((size_t)si * a.B + b) * gridDim.x + tile;
    a.workspace[rec * DD_PARTIAL_STRIDE + tid] = r;
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

static size_t footprint_floats(const DDPhotoArgs& a, long long off[DD_MAX_SCALES]) {
  const size_t tiles = (size_t)((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
  const int nch = a.mode == DD_MODE_RIGID ? 1 : (a.mode == DD_MODE_FLOW ? 7 : 9);
  size_t total = 0;
  for (int s = 0; s < a.num_scales; ++s) {
    off[s] = (long long)total;
    const int shift = a.scale[s].shift;
    if (shift > 0 && a.want_grad) total += tiles * a.B * nch * (size_t)((TH >> shift) + 2) * ((TW >> shift) + 2);
  }
  return total;
}

template <int MODE, bool AUTOMASK, bool GRAD>
static int launch_photo(const DDPhotoArgs& a, hipStream_t stream) {
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH, tiles = tiles_x * tiles_y;
  dim3 grid(tiles, a.B, a.num_scales);
  auto kern = photo_tile_kernel<MODE, AUTOMASK, GRAD>;
  static bool attr_set = false;   // per-instantiation; the attribute is a property of the code object
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(LdsLayout));
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  FootprintInfo fp;
  footprint_floats(a, fp.off);
  fp.base = a.workspace + (size_t)tiles * a.B * a.num_scales * DD_PARTIAL_STRIDE;
  hipLaunchKernelGGL(kern, grid, dim3(NT), sizeof(LdsLayout), stream, a, depth_params(a.min_depth, a.max_depth), fp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (GRAD) {
    int max_n = 0;
    for (int s = 0; s < a.num_scales; ++s)
      if (a.scale[s].shift > 0) max_n = max(max_n, a.scale[s].h * a.scale[s].w);
    if (max_n > 0) {
      constexpr int NCH = 1 + (MODE != MODE_RIGID ? 6 : 0) + (MODE == MODE_FLOW_MASK ? 2 : 0);
      hipLaunchKernelGGL((photo_combine_kernel<NCH>), dim3((max_n + 255) / 256, a.B, a.num_scales), dim3(256), 0, stream, a, fp, tiles_x, tiles_y);
      e = hipGetLastError();
      if (e != hipSuccess) return (int)e;
    }
  }
  hipLaunchKernelGGL(photo_finalize_kernel, dim3(a.num_scales + a.B), dim3(256), 0, stream, a.workspace, a.num_scales,
                     a.B, tiles, a.sums, a.want_grad ? a.g_T[0] : nullptr, a.want_grad ? a.g_T[1] : nullptr);
  return (int)hipGetLastError();
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
  return (tiles * a->B * a->num_scales * DD_PARTIAL_STRIDE + dd::footprint_floats(*a, off)) * sizeof(float);
}

extern "C" int dd_photo_loss(const DDPhotoArgs* a, void* stream_) {
  using namespace dd;
  if (!a || a->abi_version != DD_ABI_VERSION) return (int)hipErrorInvalidValue;
  if (a->num_scales < 1 || a->num_scales > DD_MAX_SCALES || a->B < 1 || !a->workspace || !a->sums) return (int)hipErrorInvalidValue;
  for (int s = 0; s < a->num_scales; ++s) {
    const DDPhotoScale& sc = a->scale[s];
    if (sc.shift < 0 || sc.shift > 3 || sc.h != (a->H >> sc.shift) || sc.w != (a->W >> sc.shift)) return (int)hipErrorInvalidValue;
    if ((a->H % (1 << sc.shift)) || (a->W % (1 << sc.shift))) return (int)hipErrorInvalidValue;
  }
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const bool g = a->want_grad != 0;
  switch (a->mode) {
    case DD_MODE_RIGID:
      if (a->automask) return g ? launch_photo<MODE_RIGID, true, true>(*a, stream) : launch_photo<MODE_RIGID, true, false>(*a, stream);
      return g ? launch_photo<MODE_RIGID, false, true>(*a, stream) : launch_photo<MODE_RIGID, false, false>(*a, stream);
    case DD_MODE_FLOW:
      if (a->automask) return (int)hipErrorInvalidValue;
      return g ? launch_photo<MODE_FLOW, false, true>(*a, stream) : launch_photo<MODE_FLOW, false, false>(*a, stream);
    case DD_MODE_FLOW_MASK:
      if (a->automask) return (int)hipErrorInvalidValue;
      return g ? launch_photo<MODE_FLOW_MASK, false, true>(*a, stream) : launch_photo<MODE_FLOW_MASK, false, false>(*a, stream);
    default:
      return (int)hipErrorInvalidValue;
  }
}
