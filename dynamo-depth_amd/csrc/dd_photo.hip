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
//      bilinear up-sampling accumulated in LDS, then one global atomic per low-res pixel per tile;
//   R  wave64 shuffle + LDS reduction of the loss sums and of d(loss)/dT -> one record per block.
// A second tiny kernel folds the per-block records deterministically.
//
// Replaces Trainer.generate_images_pred + the photometric part of Trainer.compute_losses and their
// autograd (reference Trainer.py:215-352,384-386,413-423; tools.py:191-257,291-298) -- thousands of
// ATen launches per step in the reference.  The arithmetic is dd_math.h.
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_math.h"

namespace dd {

constexpr int TH = 16;            // tile height (target pixels)
constexpr int TW = 64;            // tile width  (one wave64 per row -> coalesced 256 B rows)
#ifndef DD_PHOTO_NT
#define DD_PHOTO_NT 512
#endif
constexpr int NT = DD_PHOTO_NT;   // threads per workgroup
constexpr int NPT = TH * TW / NT; // interior pixels owned per thread
constexpr int RH = TH + 4, RW = TW + 4;       // region with 2-pixel halo (warped colours, target)
constexpr int R2N = RH * RW;
constexpr int CH_ = TH + 2, CW_ = TW + 2;     // centres with 1-pixel halo (SSIM, selection, coefficients)
constexpr int R1N = CH_ * CW_;
constexpr int RING = R2N - TH * TW;
constexpr int FPH_MAX = TH / 2 + 2, FPW_MAX = TW / 2 + 2;   // low-res footprint of a tile at scale >= 1
constexpr int FPN_MAX = FPH_MAX * FPW_MAX;
constexpr int LRN_MAX = (TH / 2) * (TW / 2);                  // low-res pixels inside a tile at scale >= 1
constexpr int NWAVES = NT / 64;
constexpr int NRED = 30;          // photo, n_warp, cons[2], delta[2], gT[2][12]

static_assert(NPT * NT == TH * TW, "tile must be divisible among threads");
static_assert(RING <= NT, "one pass over the halo ring");

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

struct LdsLayout {
  float pred[2 * 3 * R2N];       // warped source colours (identity copies during the automask pre-pass)
  float tgt[3 * R2N];            // target colours
  float coef[9 * R1N];           // backward coefficients of the selected frame
  int sel[R1N];                  // selected frame per centre (-1: identity won / outside the image)
  float idmin[R1N];              // automask: min over frames of the identity reprojection loss (+noise)
  float gacc[9 * FPN_MAX];       // low-res gradient accumulators: disp, flow[2][3], mask[2]
  float lr[2 * 5 * LRN_MAX];     // low-res residual flow (3) and grid difference (2) per frame
  float red[NWAVES * NRED];
};

template <int MODE, bool AUTOMASK, bool GRAD>
__global__ __launch_bounds__(NT) void photo_tile_kernel(const DDPhotoArgs a, const DepthParams dp) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  LdsLayout& S = *reinterpret_cast<LdsLayout*>(smem_raw);

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int si = blockIdx.z;
  const DDPhotoScale& sc = a.scale[si];
  const int H = a.H, W = a.W, N = H * W;
  const int tiles_x = (W + TW - 1) / TW;
  const int tile = blockIdx.x;
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

  // footprint of this tile on the low-res grid (taps of the up-sampling), scale >= 1 only
  const int fy0 = max((Y0 >> shift) - 1, 0), fx0 = max((X0 >> shift) - 1, 0);
  const int fph = (TH >> shift) + 2, fpw = (TW >> shift) + 2;
  const int lrh = TH >> shift, lrw = TW >> shift;           // low-res pixels inside the tile (shift >= 1)
  constexpr int NCH = 1 + (MODE != MODE_RIGID ? 6 : 0) + (MODE == MODE_FLOW_MASK ? 2 : 0);

#ifdef DD_STAGE_TIMING
  unsigned long long t_prev = clock64();
#endif
  // ---- stage 0: clear accumulators, stage the target region --------------------------------------
  if (GRAD && shift > 0)
    for (int i = tid; i < NCH * FPN_MAX; i += NT) S.gacc[i] = 0.f;
  if (MODE == MODE_FLOW_MASK && shift > 0)
    for (int i = tid; i < 2 * 5 * LRN_MAX; i += NT) S.lr[i] = 0.f;
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
  // owned interior pixels: column tid % TW, rows NPT*(tid / TW) + j
  const int lx = tid % TW, ly0 = (tid / TW) * NPT;
  float Zs[NPT], mval[NPT][2], cval[NPT][2][3], xval[NPT][2][3], dvx[NPT][2][3], dvy[NPT][2][3];
  FrameGeom geo[NPT][2];
  bool own[NPT];
  float acc_cons[2] = {0.f, 0.f}, acc_delta[2] = {0.f, 0.f};

  auto warp_pixel = [&](int X, int Y, bool owner, int j) __attribute__((always_inline)) {
    const int p = Y * W + X;
    const float d = resize_eval(disp_g, X, Y, h, w, ratio);
    const float Z = dd_rcp(dp.lo + dp.span * d);
    float ray[3], P[3];
    pixel_ray(cam, X, Y, ray);
#pragma unroll
    for (int k = 0; k < 3; ++k) P[k] = Z * ray[k];
    const int li = (Y - (Y0 - 2)) * RW + (X - (X0 - 2));
    if (owner && sc.out_depth) sc.out_depth[(size_t)b * N + p] = Z;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float c[3] = {0.f, 0.f, 0.f}, m = 1.f;
      if (MODE != MODE_RIGID) {
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = resize_eval(sc.flow[f] + ((size_t)b * 3 + k) * n, X, Y, h, w, ratio) * tsv[f];
      }
      if (MODE == MODE_FLOW_MASK) m = resize_eval(sc.mask[f] + (size_t)b * n, X, Y, h, w, ratio);
      FrameGeom g;
      frame_geometry<MODE>(cam, Tm[f], P, c, m, dim, a.eps, g);
      const SampleCoord scd = sample_coord(g.gnx, g.gny, W, H);
      float xv[3], ddx[3], ddy[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        xv[ch] = sample_plane(src_g[f] + (size_t)ch * N, scd, W, H, ddx[ch], ddy[ch]);
        S.pred[(f * 3 + ch) * R2N + li] = xv[ch];
      }
      if (owner) {
        // j is a compile-time constant at every call site with owner == true
        Zs[j] = Z;
        mval[j][f] = m;
        geo[j][f] = g;
#pragma unroll
        for (int k = 0; k < 3; ++k) { cval[j][f][k] = c[k]; xval[j][f][k] = xv[k]; dvx[j][f][k] = ddx[k]; dvy[j][f][k] = ddy[k]; }
        if (sc.out_color[f]) {
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) sc.out_color[f][((size_t)b * 3 + ch) * N + p] = xv[ch];
        }
        if (sc.out_sample[f]) {
          float2 gn = make_float2(g.gnx, g.gny);
          reinterpret_cast<float2*>(sc.out_sample[f])[(size_t)b * N + p] = gn;
        }
        if (MODE == MODE_FLOW_MASK) {
          if (shift == 0) {
            // the low-res pixel IS this pixel: c_consistency and disp_mag in registers
            const float valid = disp_g[p] > a.disp_thr ? 1.f : 0.f;
            const float om = 1.f - sc.mask[f][(size_t)b * n + p];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              acc_cons[f] += valid * om * dd_abs(g.r[k]);
              if (sc.out_resid[f]) atomicAdd(&sc.out_resid[f][((size_t)b * 3 + k) * n + p], g.r[k]);
            }
            const float dx = g.ego_gn[0] - g.cmp_gn[0], dy = g.ego_gn[1] - g.cmp_gn[1];
            const float delta = dx * dx + dy * dy;
            acc_delta[f] += delta;
            if (sc.out_delta[f]) atomicAdd(&sc.out_delta[f][(size_t)b * n + p], delta);
          } else if (down_tap(X, shift) && down_tap(Y, shift)) {
            const int q = (((Y - Y0) >> shift) * lrw) + ((X - X0) >> shift);
#pragma unroll
            for (int k = 0; k < 3; ++k) atomicAdd(&S.lr[(f * 5 + k) * LRN_MAX + q], 0.25f * g.r[k]);
            atomicAdd(&S.lr[(f * 5 + 3) * LRN_MAX + q], 0.25f * (g.ego_gn[0] - g.cmp_gn[0]));
            atomicAdd(&S.lr[(f * 5 + 4) * LRN_MAX + q], 0.25f * (g.ego_gn[1] - g.cmp_gn[1]));
          }
        }
      }
    }
  };

#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int X = X0 + lx, Y = Y0 + ly0 + j;
    own[j] = (X < W) && (Y < H);
    if (own[j]) warp_pixel(X, Y, true, j);
  }
  if (tid < RING) {
    int ry, rx;
    if (tid < 2 * RW) { ry = tid / RW; rx = tid % RW; }
    else if (tid < 4 * RW) { const int r2 = tid - 2 * RW; ry = RH - 2 + r2 / RW; rx = r2 % RW; }
    else { const int r3 = tid - 4 * RW; ry = 2 + (r3 >> 2); const int k = r3 & 3; rx = k < 2 ? k : RW - 4 + k; }
    const int Y = Y0 - 2 + ry, X = X0 - 2 + rx;
    if (Y >= 0 && Y < H && X >= 0 && X < W) warp_pixel(X, Y, false, 0);
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
            const float rv = S.lr[(f * 5 + k) * LRN_MAX + q];
            acc_cons[f] += valid * om * dd_abs(rv);
            S.lr[(f * 5 + k) * LRN_MAX + q] = sc.w_cons * valid * om * dd_sign(rv);
            if (sc.out_resid[f]) atomicAdd(&sc.out_resid[f][((size_t)b * 3 + k) * n + gq], rv);
          }
          const float dx = S.lr[(f * 5 + 3) * LRN_MAX + q], dy = S.lr[(f * 5 + 4) * LRN_MAX + q];
          const float delta = dx * dx + dy * dy;
          acc_delta[f] += delta;
          if (sc.out_delta[f]) atomicAdd(&sc.out_delta[f][(size_t)b * n + gq], delta);
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
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      if (!own[j]) continue;
      const int X = X0 + lx, Y = Y0 + ly0 + j;
      const int p = Y * W + X;
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
      for (int k = 0; k < 3; ++k) P[k] = Zs[j] * ray[k];
      const Tap2 tap = resize_tap2(X, Y, h, w, ratio);
      // offsets of the four up-sampling taps inside the tile's LDS footprint (scale >= 1)
      int fo[4];
      if (shift > 0) {
        const Tap1 tx = resize_tap(X, w, ratio), ty = resize_tap(Y, h, ratio);
        fo[0] = (ty.i0 - fy0) * FPW_MAX + (tx.i0 - fx0);
        fo[1] = (ty.i0 - fy0) * FPW_MAX + (tx.i1 - fx0);
        fo[2] = (ty.i1 - fy0) * FPW_MAX + (tx.i0 - fx0);
        fo[3] = (ty.i1 - fy0) * FPW_MAX + (tx.i1 - fx0);
      }
      const float tw[4] = {tap.w00, tap.w01, tap.w10, tap.w11};
      auto scatter = [&](float* gbase, int chan, float gval) __attribute__((always_inline)) {
        if (shift == 0) {
          gbase[p] += gval;                         // exactly one owner per element at scale 0
        } else {
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) atomicAdd(&S.gacc[chan * FPN_MAX + fo[t4]], tw[t4] * gval);
        }
      };
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        float gu = 0.f, gv = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float xv = xval[j][f][ch], yv = S.tgt[ch * R2N + li];
          float gx = Sc[f][ch * 3 + 0] + xv * Sc[f][ch * 3 + 1] + yv * Sc[f][ch * 3 + 2];
          if (own_sel == f) gx += sc.w_photo * (1.f - alpha) * (1.f / 3.f) * dd_sign(xv - yv);
          gu += gx * dvx[j][f][ch];
          gv += gx * dvy[j][f][ch];
        }
        float gr_extra[3] = {0.f, 0.f, 0.f};
        if (MODE == MODE_FLOW_MASK) {
          if (shift == 0) {
            const float vm = (disp_g[p] > a.disp_thr ? sc.w_cons : 0.f) * (1.f - sc.mask[f][(size_t)b * n + p]);
#pragma unroll
            for (int k = 0; k < 3; ++k) gr_extra[k] = vm * dd_sign(geo[j][f].r[k]);
          } else if (down_tap(X, shift) && down_tap(Y, shift)) {
            const int q = (((Y - Y0) >> shift) * lrw) + ((X - X0) >> shift);
#pragma unroll
            for (int k = 0; k < 3; ++k) gr_extra[k] = 0.25f * S.lr[(f * 5 + k) * LRN_MAX + q];
          }
        }
        PixelGrad pg;
        frame_geometry_bwd<MODE>(cam, Tm[f], P, mval[j][f], geo[j][f], gu, gv, gr_extra, pg);
#pragma unroll
        for (int k = 0; k < 3; ++k) gPtot[k] += pg.gP[k];
#pragma unroll
        for (int k = 0; k < 12; ++k) gTacc[f][k] += pg.gT[k];
        if (MODE != MODE_RIGID) {
#pragma unroll
          for (int k = 0; k < 3; ++k) scatter(sc.g_flow[f] + ((size_t)b * 3 + k) * n, 1 + f * 3 + k, pg.gc[k] * tsv[f]);
        }
        if (MODE == MODE_FLOW_MASK) scatter(sc.g_mask[f] + (size_t)b * n, 7 + f, pg.gm);
      }
      scatter(sc.g_disp + (size_t)b * n, 0, depth_bwd(dp, gPtot, ray, Zs[j]));
    }
    if (shift > 0) {
      __syncthreads();
      DD_STAGE_MARK(4);
      for (int i = tid; i < NCH * fph * fpw; i += NT) {
        const int chan = i / (fph * fpw), r = i % (fph * fpw);
        const int qy = fy0 + r / fpw, qx = fx0 + r % fpw;
        if (qy >= h || qx >= w) continue;
        const float v = S.gacc[chan * FPN_MAX + (r / fpw) * FPW_MAX + (r % fpw)];
        if (v == 0.f) continue;
        float* dst;
        if (chan == 0) dst = sc.g_disp + (size_t)b * n;
        else if (chan < 7) dst = sc.g_flow[(chan - 1) / 3] + ((size_t)b * 3 + (chan - 1) % 3) * n;
        else dst = sc.g_mask[chan - 7] + (size_t)b * n;
        atomicAdd(&dst[qy * w + qx], v);
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

template <int MODE, bool AUTOMASK, bool GRAD>
static int launch_photo(const DDPhotoArgs& a, hipStream_t stream) {
  const int tiles = ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
  dim3 grid(tiles, a.B, a.num_scales);
  auto kern = photo_tile_kernel<MODE, AUTOMASK, GRAD>;
  static bool attr_set = false;   // per-instantiation; the attribute is a property of the code object
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)sizeof(LdsLayout));
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(NT), sizeof(LdsLayout), stream, a, depth_params(a.min_depth, a.max_depth));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
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
  return tiles * a->B * a->num_scales * DD_PARTIAL_STRIDE * sizeof(float);
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
