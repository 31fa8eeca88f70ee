// photo_host.cpp -- TEST-ONLY host executor of csrc/dd_math.h + csrc/dd_pair.h.
//
// Walks whole images pixel by pixel with the very functions the HIP kernels call
// (dynamo-depth_amd/csrc/dd_math.h compiled by g++), behind the same DDPhotoArgs struct as
// dd_photo_loss() but with HOST pointers.  tests/test_hostmath.py compares it with the oracle's
// autograd, which validates the explicit backward formulas on a machine without a GPU.
// It is not part of the product path (nothing under dynamo-depth_amd/ links or loads it).
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../dynamo-depth_amd/csrc/dd_pair.h"
#include "../../include/dynamo_hip.h"

using namespace dd;

namespace {

template <int MODE>
void run_scale(const DDPhotoArgs& a, const DDPhotoScale& sc, float* sums, std::vector<double>& gT) {
  const int B = a.B, H = a.H, W = a.W, N = H * W, h = sc.h, w = sc.w, n = h * w;
  const float ratio = 1.f / static_cast<float>(1 << sc.shift);
  const DepthParams dp = depth_params(a.min_depth, a.max_depth);
  const float alpha = a.ssim_weight;
  const ImageDims dim = image_dims(W, H);
  double photo_sum = 0, cons_sum[2] = {0, 0}, delta_sum[2] = {0, 0}, n_warp = 0;

  std::vector<float> pred(2 * 3 * N), dvx(2 * 3 * N), dvy(2 * 3 * N);
  std::vector<PairGeom> geom(N);
  std::vector<float> Zs(N);
  std::vector<f2> mvals(N);
  std::vector<float> rho(2 * N), coef(9 * N), idmin(N);
  std::vector<int> sel(N);
  std::vector<float> resid(2 * 3 * n), dgn(2 * 2 * n), gresid(2 * 3 * n);

  for (int b = 0; b < B; ++b) {
    Intrinsics cam;
    load_intrinsics(cam, a.K + b * 16, a.inv_K + b * 16);
    const float* tgt = a.target + (size_t)b * 3 * N;
    const float* disp = sc.disp + (size_t)b * n;
    std::fill(resid.begin(), resid.end(), 0.f);
    std::fill(dgn.begin(), dgn.end(), 0.f);
    // ---- stage A: geometry + warp --------------------------------------------------------
    for (int Y = 0; Y < H; ++Y)
      for (int X = 0; X < W; ++X) {
        const int p = Y * W + X;
        const float d = resize_eval(disp, X, Y, h, w, ratio);
        const float Z = 1.f / (dp.lo + dp.span * d);
        Zs[p] = Z;
        if (sc.out_depth) sc.out_depth[(size_t)b * N + p] = Z;
        float ray[3], P[3];
        pixel_ray(cam, X, Y, ray);
        for (int k = 0; k < 3; ++k) P[k] = Z * ray[k];
        // both source frames at once through the pair arithmetic of dd_pair.h (what the HIP kernel executes)
        {
          PairT Tm;
          load_pair_T(Tm, a.T[0] + b * 16, a.T[1] + b * 16);
          f2 c[3] = {sp2(0.f), sp2(0.f), sp2(0.f)}, m = sp2(1.f);
          if (MODE != MODE_RIGID) {
            const f2 tsv = mk2(a.ts[0] ? a.ts[0][b] : 1.f, a.ts[1] ? a.ts[1][b] : 1.f);
            for (int k = 0; k < 3; ++k)
              c[k] = mk2(resize_eval(sc.flow[0] + ((size_t)b * 3 + k) * n, X, Y, h, w, ratio),
                         resize_eval(sc.flow[1] + ((size_t)b * 3 + k) * n, X, Y, h, w, ratio)) * tsv;
          }
          if (MODE == MODE_FLOW_MASK)
            m = mk2(resize_eval(sc.mask[0] + (size_t)b * n, X, Y, h, w, ratio), resize_eval(sc.mask[1] + (size_t)b * n, X, Y, h, w, ratio));
          PairGeom& g = geom[p];
          PairSide sd;
          frame_geometry2<MODE>(cam, Tm, P, c, m, dim, a.eps, g, sd);
          mvals[p] = m;
          const SampleCoord2 scd = sample_coord2(g.proj.u, g.proj.v, W, H);
          const f2 gnx = grid_normalise2(g.proj.u, dim.inv_wm1), gny = grid_normalise2(g.proj.v, dim.inv_hm1);
          for (int ch = 0; ch < 3; ++ch) {
            const float* p0 = a.source[0] + ((size_t)b * 3 + ch) * N + scd.o00[0] / 4;
            const float* p1 = a.source[1] + ((size_t)b * 3 + ch) * N + scd.o00[1] / 4;
            const unsigned dx0 = scd.dxb[0] / 4, dx1 = scd.dxb[1] / 4, dy0 = scd.dyb[0] / 4, dy1 = scd.dyb[1] / 4;
            f2 dx2, dy2;
            const f2 v = sample_taps2(scd, mk2(p0[0], p1[0]), mk2(p0[dx0], p1[dx1]), mk2(p0[dy0], p1[dy1]),
                                      mk2(p0[dy0 + dx0], p1[dy1 + dx1]), dx2, dy2);
            for (int f = 0; f < 2; ++f) {
              pred[(f * 3 + ch) * N + p] = v[f];
              dvx[(f * 3 + ch) * N + p] = dx2[f];
              dvy[(f * 3 + ch) * N + p] = dy2[f];
              if (sc.out_color[f]) sc.out_color[f][((size_t)b * 3 + ch) * N + p] = v[f];
            }
          }
          for (int f = 0; f < 2; ++f) {
            if (sc.out_sample[f]) {
              sc.out_sample[f][((size_t)b * N + p) * 2 + 0] = gnx[f];
              sc.out_sample[f][((size_t)b * N + p) * 2 + 1] = gny[f];
            }
            if (MODE == MODE_FLOW_MASK) {
              // bilinear down-sampling (align_corners=False) to (h,w): for power-of-two ratios the two taps per
              // axis are the centre pair of each block with weight 1/2 (identity at scale 0)
              const int blk = 1 << sc.shift;
              bool cx = true, cy = true;
              if (blk > 1) {
                cx = (X % blk == blk / 2 - 1) || (X % blk == blk / 2);
                cy = (Y % blk == blk / 2 - 1) || (Y % blk == blk / 2);
              }
              if (cx && cy) {
                const float wt = blk > 1 ? 0.25f : 1.f;
                const int q = (Y >> sc.shift) * w + (X >> sc.shift);
                for (int k = 0; k < 3; ++k) resid[(f * 3 + k) * n + q] += wt * sd.r[k][f];
                dgn[(f * 2 + 0) * n + q] += wt * sd.dgx[f];
                dgn[(f * 2 + 1) * n + q] += wt * sd.dgy[f];
              }
            }
          }
        }
      }
    // ---- consistency + delta on the low-res grid ---------------------------------------------
    if (MODE == MODE_FLOW_MASK) {
      for (int f = 0; f < 2; ++f)
        for (int q = 0; q < n; ++q) {
          const float valid = disp[q] > a.disp_thr ? 1.f : 0.f;
          const float om = 1.f - sc.mask[f][(size_t)b * n + q];
          for (int k = 0; k < 3; ++k) {
            const float rv = resid[(f * 3 + k) * n + q];
            cons_sum[f] += valid * om * dd_abs(rv);
            gresid[(f * 3 + k) * n + q] = sc.w_cons * valid * om * dd_sign(rv);
            if (sc.out_resid[f]) sc.out_resid[f][((size_t)b * 3 + k) * n + q] += rv;
          }
          const float dx = dgn[(f * 2 + 0) * n + q], dy = dgn[(f * 2 + 1) * n + q];
          const float delta = dx * dx + dy * dy;
          delta_sum[f] += delta;
          if (sc.out_delta[f]) sc.out_delta[f][(size_t)b * n + q] += delta;
        }
    }
    // ---- stage B: SSIM + L1, selection, backward coefficients ------------------------------
    for (int Y = 0; Y < H; ++Y)
      for (int X = 0; X < W; ++X) {
        const int p = Y * W + X;
        int ry[3], rx[3];
        for (int d = 0; d < 3; ++d) { ry[d] = dd_reflect(Y + d - 1, H); rx[d] = dd_reflect(X + d - 1, W); }
        float rho_f[2], cf[2][9];
        const float wgt = sc.w_photo * alpha / 3.f / 9.f;
        {
          f2 ssum = sp2(0.f), l1 = sp2(0.f);
          for (int ch = 0; ch < 3; ++ch) {
            const float* yp = tgt + (size_t)ch * N;
            f2 sx = sp2(0.f), sxx = sp2(0.f), sxy = sp2(0.f);
            float sy = 0.f, syy = 0.f;
            for (int j = 0; j < 3; ++j)
              for (int i = 0; i < 3; ++i) {
                const int o = ry[j] * W + rx[i];
                const f2 xv = mk2(pred[(0 * 3 + ch) * N + o], pred[(1 * 3 + ch) * N + o]);
                const float yv = yp[o];
                sx += xv; sxx += xv * xv; sxy += xv * sp2(yv); sy += yv; syy += yv * yv;
              }
            SsimGrad2 sg;
            ssum += ssim_value2<true>(sx, sxx, sxy, sy, syy, wgt, sg);
            l1 += abs2(sp2(yp[p]) - mk2(pred[(0 * 3 + ch) * N + p], pred[(1 * 3 + ch) * N + p]));
            for (int f = 0; f < 2; ++f) { cf[f][ch * 3 + 0] = sg.dmu[f]; cf[f][ch * 3 + 1] = sg.dxx2[f]; cf[f][ch * 3 + 2] = sg.dxy[f]; }
          }
          const f2 rho2 = sp2(alpha) * (ssum * sp2(1.f / 3.f)) + sp2(1.f - alpha) * (l1 * sp2(1.f / 3.f));
          rho_f[0] = rho2[0]; rho_f[1] = rho2[1];
        }
        float best = rho_f[0];
        int bf = 0;
        if (rho_f[1] < best) { best = rho_f[1]; bf = 1; }
        if (a.automask) {
          float idb = 0.f;
          for (int f = 0; f < 2; ++f) {
            float ssum = 0.f, l1 = 0.f;
            for (int ch = 0; ch < 3; ++ch) {
              const float* xp = a.source[f] + ((size_t)b * 3 + ch) * N;
              const float* yp = tgt + (size_t)ch * N;
              SsimStats st = {0, 0, 0, 0, 0};
              for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i) {
                  const float xv = xp[ry[j] * W + rx[i]], yv = yp[ry[j] * W + rx[i]];
                  st.sx += xv; st.sy += yv; st.sxx += xv * xv; st.syy += yv * yv; st.sxy += xv * yv;
                }
              ssum += ssim_value(st, nullptr);
              l1 += dd_abs(yp[p] - xp[p]);
            }
            float v = alpha * (ssum / 3.f) + (1.f - alpha) * (l1 / 3.f);
            if (sc.noise) v += sc.noise[((size_t)b * 2 + f) * N + p] * 0.00001f;
            idb = (f == 0) ? v : (v < idb ? v : idb);
          }
          if (idb <= best) { best = idb; bf = -1; }   // identity entries come first in the cat -> win ties
          if (sc.out_idsel) sc.out_idsel[(size_t)b * N + p] = bf >= 0 ? 1.f : 0.f;
        }
        photo_sum += best;
        n_warp += bf >= 0 ? 1 : 0;
        sel[p] = bf;
        for (int k = 0; k < 9; ++k) coef[k * N + p] = bf >= 0 ? cf[bf][k] : 0.f;   // already weighted (gscale)
      }
    if (!a.want_grad) continue;
    // ---- stage C: adjoint of the box filter, warp backward, scatter ---------------------------
    for (int Y = 0; Y < H; ++Y)
      for (int X = 0; X < W; ++X) {
        const int p = Y * W + X;
        float S[2][9];
        memset(S, 0, sizeof(S));
        for (int cy = Y - 1; cy <= Y + 1; ++cy) {
          if (cy < 0 || cy >= H) continue;
          const int my = reflect_multiplicity(cy, Y, H);
          for (int cx = X - 1; cx <= X + 1; ++cx) {
            if (cx < 0 || cx >= W) continue;
            const int mult = my * reflect_multiplicity(cx, X, W);
            const int q = cy * W + cx;
            if (sel[q] < 0 || mult == 0) continue;
            for (int k = 0; k < 9; ++k) S[sel[q]][k] += static_cast<float>(mult) * coef[k * N + q];
          }
        }
        float ray[3], P[3], gPtot[3] = {0, 0, 0};
        pixel_ray(cam, X, Y, ray);
        for (int k = 0; k < 3; ++k) P[k] = Zs[p] * ray[k];
        const Tap2 tap = resize_tap2(X, Y, h, w, ratio);
        {
          f2 gu = sp2(0.f), gv = sp2(0.f);
          for (int ch = 0; ch < 3; ++ch) {
            const float yv = tgt[(size_t)ch * N + p];
            for (int f = 0; f < 2; ++f) {
              const float xv = pred[(f * 3 + ch) * N + p];
              float gx = S[f][ch * 3 + 0] + xv * S[f][ch * 3 + 1] + yv * S[f][ch * 3 + 2];
              if (sel[p] == f) gx += sc.w_photo * (1.f - alpha) / 3.f * dd_sign(xv - yv);
              gu[f] += gx * dvx[(f * 3 + ch) * N + p];
              gv[f] += gx * dvy[(f * 3 + ch) * N + p];
            }
          }
          f2 gr_extra[3] = {sp2(0.f), sp2(0.f), sp2(0.f)};
          if (MODE == MODE_FLOW_MASK) {
            const int blk = 1 << sc.shift;
            bool cx = true, cy = true;
            if (blk > 1) {
              cx = (X % blk == blk / 2 - 1) || (X % blk == blk / 2);
              cy = (Y % blk == blk / 2 - 1) || (Y % blk == blk / 2);
            }
            if (cx && cy) {
              const float wt = blk > 1 ? 0.25f : 1.f;
              const int q = (Y >> sc.shift) * w + (X >> sc.shift);
              for (int k = 0; k < 3; ++k) gr_extra[k] = mk2(wt * gresid[(0 * 3 + k) * n + q], wt * gresid[(1 * 3 + k) * n + q]);
            }
          }
          PairT Tm;
          load_pair_T(Tm, a.T[0] + b * 16, a.T[1] + b * 16);
          PairGrad pg;
          frame_geometry_bwd2<MODE>(cam, Tm, P, mvals[p], geom[p], gu, gv, gr_extra, pg);
          for (int k = 0; k < 3; ++k) gPtot[k] = hsum(pg.gP[k]);
          for (int f = 0; f < 2; ++f) {
            for (int k = 0; k < 12; ++k) gT[(b * 2 + f) * 12 + k] += pg.gT[k][f];
            if (MODE != MODE_RIGID) {
              const float tsv = a.ts[f] ? a.ts[f][b] : 1.f;
              for (int k = 0; k < 3; ++k) {
                float* gf = sc.g_flow[f] + ((size_t)b * 3 + k) * n;
                const float gk = pg.gc[k][f] * tsv;
                gf[tap.o00] += tap.w00 * gk; gf[tap.o01] += tap.w01 * gk;
                gf[tap.o10] += tap.w10 * gk; gf[tap.o11] += tap.w11 * gk;
              }
            }
            if (MODE == MODE_FLOW_MASK) {
              float* gmk = sc.g_mask[f] + (size_t)b * n;
              const float gm = pg.gm[f];
              gmk[tap.o00] += tap.w00 * gm; gmk[tap.o01] += tap.w01 * gm;
              gmk[tap.o10] += tap.w10 * gm; gmk[tap.o11] += tap.w11 * gm;
            }
          }
        }
        const float gd = depth_bwd(dp, gPtot, ray, Zs[p]);
        float* gdp = sc.g_disp + (size_t)b * n;
        gdp[tap.o00] += tap.w00 * gd; gdp[tap.o01] += tap.w01 * gd;
        gdp[tap.o10] += tap.w10 * gd; gdp[tap.o11] += tap.w11 * gd;
      }
  }
  sums[0] = static_cast<float>(photo_sum);
  sums[1] = static_cast<float>(cons_sum[0]);
  sums[2] = static_cast<float>(cons_sum[1]);
  sums[3] = static_cast<float>(delta_sum[0]);
  sums[4] = static_cast<float>(delta_sum[1]);
  sums[5] = static_cast<float>(n_warp);
}

}  // namespace

extern "C" int dd_photo_loss_host(const DDPhotoArgs* a) {
  if (a->abi_version != DD_ABI_VERSION) return -1;
  std::vector<double> gT((size_t)a->B * 2 * 12, 0.0);
  for (int s = 0; s < a->num_scales; ++s) {
    float* sums = a->sums + s * DD_SUMS_STRIDE;
    for (int k = 0; k < DD_SUMS_STRIDE; ++k) sums[k] = 0.f;
    switch (a->mode) {
      case DD_MODE_RIGID: run_scale<MODE_RIGID>(*a, a->scale[s], sums, gT); break;
      case DD_MODE_FLOW: run_scale<MODE_FLOW>(*a, a->scale[s], sums, gT); break;
      case DD_MODE_FLOW_MASK: run_scale<MODE_FLOW_MASK>(*a, a->scale[s], sums, gT); break;
      default: return -2;
    }
  }
  if (a->want_grad)
    for (int f = 0; f < 2; ++f)
      for (int b = 0; b < a->B; ++b)
        for (int k = 0; k < 16; ++k)
          a->g_T[f][b * 16 + k] = k < 12 ? static_cast<float>(gT[(b * 2 + f) * 12 + k]) : 0.f;
  return 0;
}
