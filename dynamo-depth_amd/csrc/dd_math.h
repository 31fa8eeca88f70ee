// dd_math.h -- per-pixel arithmetic of the Dynamo-Depth view-synthesis loss, forward and explicit
// backward, shared by every HIP kernel in this directory.
//
// Everything here is `DD_HD` (host + device) so that tests/hostmath/ can compile the very same
// functions with g++ and check them against the oracle's autograd on a GPU-less machine.  The
// kernels add tiling, LDS staging and reductions around these functions; they do not restate them.
//
// Reference semantics followed (paths relative to the reference checkout):
//   up-sampling      utils.py:98-101  -> F.interpolate(bilinear, align_corners=False)
//   disp -> depth    tools.py:291-298
//   back-projection  tools.py:191-197
//   projection       tools.py:211-224
//   flow composition Trainer.py:248-277
//   warp             Trainer.py:281   -> F.grid_sample(bilinear, border, align_corners=True)
//   SSIM + L1        tools.py:243-257, Trainer.py:413-423
// SURVEY.md Appendix A lists the exact formulas and the sub-gradient conventions of the torch ops.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DD_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define DD_HD inline
#endif

namespace dd {

constexpr float kSsimC1 = 0.0001f;   // 0.01^2
constexpr float kSsimC2 = 0.0009f;   // 0.03^2

// Flow-composition mode of Trainer.generate_images_pred (phase flags, Trainer.py:466-490).
enum : int {
  MODE_RIGID = 0,     // disp_init:   S = T*P
  MODE_FLOW = 1,      // motion_init: S = P + c            (T unused, Trainer.py:270-271)
  MODE_FLOW_MASK = 2  // mask_init / fine_tune: S = T*(P + m*(c - (T*P - P)))
};

// 1/x: v_rcp_f32 (1 ulp) refined by one Newton step on the device -- an IEEE division costs ~10 VALU instructions and
// the warp/SSIM path has dozens per pixel; the result stays within 1 ulp of the correctly rounded quotient.
DD_HD float dd_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.f), r, r);
#else
  return 1.f / x;
#endif
}

DD_HD float dd_floor(float x) { return floorf(x); }
DD_HD float dd_abs(float x) { return fabsf(x); }
DD_HD float dd_sign(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }  // abs'(0) = 0
DD_HD int dd_reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }  // ReflectionPad2d

// ------------------------------------------------------------------------------------------------
// bilinear resize taps, align_corners=False (ATen area_pixel_compute_source_index, clamped at 0)
// ------------------------------------------------------------------------------------------------
struct Tap1 {
  int i0, i1;
  float w0, w1;
};

// destination index `dst` of an axis resized from `src_n` to `dst_n` samples; ratio = src_n/dst_n
DD_HD Tap1 resize_tap(int dst, int src_n, float ratio) {
  float s = ratio * (static_cast<float>(dst) + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  Tap1 t;
  t.i0 = static_cast<int>(s);
  t.i1 = t.i0 + ((t.i0 < src_n - 1) ? 1 : 0);
  t.w1 = s - static_cast<float>(t.i0);
  t.w0 = 1.f - t.w1;
  return t;
}

struct Tap2 {
  int o00, o01, o10, o11;      // offsets inside one (h,w) plane
  float w00, w01, w10, w11;
};

DD_HD Tap2 resize_tap2(int X, int Y, int h, int w, float ratio) {
  const Tap1 tx = resize_tap(X, w, ratio), ty = resize_tap(Y, h, ratio);
  Tap2 t;
  t.o00 = ty.i0 * w + tx.i0;
  t.o01 = ty.i0 * w + tx.i1;
  t.o10 = ty.i1 * w + tx.i0;
  t.o11 = ty.i1 * w + tx.i1;
  t.w00 = ty.w0 * tx.w0;
  t.w01 = ty.w0 * tx.w1;
  t.w10 = ty.w1 * tx.w0;
  t.w11 = ty.w1 * tx.w1;
  return t;
}

// value of a low-res plane at a full-res pixel; same association as ATen:
//   w0y*(w0x*a + w1x*b) + w1y*(w0x*c + w1x*d)
DD_HD float resize_eval(const float* plane, int X, int Y, int h, int w, float ratio) {
  const Tap1 tx = resize_tap(X, w, ratio), ty = resize_tap(Y, h, ratio);
  const float* r0 = plane + ty.i0 * w;
  const float* r1 = plane + ty.i1 * w;
  return ty.w0 * (tx.w0 * r0[tx.i0] + tx.w1 * r0[tx.i1]) + ty.w1 * (tx.w0 * r1[tx.i0] + tx.w1 * r1[tx.i1]);
}

// ------------------------------------------------------------------------------------------------
// camera
// ------------------------------------------------------------------------------------------------
struct Intrinsics {
  float A[9];    // inv_K[:3,:3] row-major
  float K[12];   // K[:3,:4] row-major
};

DD_HD void load_intrinsics(Intrinsics& c, const float* K44, const float* invK44) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c.A[i * 3 + j] = invK44[i * 4 + j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) c.K[i * 4 + j] = K44[i * 4 + j];
}

struct DepthParams {
  float lo;      // 1/max_depth
  float span;    // 1/min_depth - 1/max_depth
};

DD_HD DepthParams depth_params(float min_depth, float max_depth) {
  // the reference evaluates these in Python doubles, then multiplies an fp32 tensor (tools.py:294-296)
  const double lo = 1.0 / static_cast<double>(max_depth), hi = 1.0 / static_cast<double>(min_depth);
  DepthParams p;
  p.lo = static_cast<float>(lo);
  p.span = static_cast<float>(hi - lo);
  return p;
}

DD_HD void pixel_ray(const Intrinsics& c, int X, int Y, float ray[3]) {
  const float x = static_cast<float>(X), y = static_cast<float>(Y);
  for (int i = 0; i < 3; ++i) ray[i] = c.A[i * 3 + 0] * x + c.A[i * 3 + 1] * y + c.A[i * 3 + 2];
}

// rows 0..2 of a 4x4 rigid transform applied to a point (w = 1)
DD_HD void rigid_apply(const float* T, const float p[3], float q[3]) {
  for (int i = 0; i < 3; ++i) q[i] = T[i * 4 + 0] * p[0] + T[i * 4 + 1] * p[1] + T[i * 4 + 2] * p[2] + T[i * 4 + 3];
}

struct Proj {
  float u, v;        // pixel coordinates
  float inv_den;     // 1 / (z + eps)
};

DD_HD Proj project_point(const Intrinsics& c, const float s[3], float eps) {
  const float q0 = c.K[0] * s[0] + c.K[1] * s[1] + c.K[2] * s[2] + c.K[3];
  const float q1 = c.K[4] * s[0] + c.K[5] * s[1] + c.K[6] * s[2] + c.K[7];
  const float q2 = c.K[8] * s[0] + c.K[9] * s[1] + c.K[10] * s[2] + c.K[11];
  Proj p;
  p.inv_den = dd_rcp(q2 + eps);
  p.u = q0 * p.inv_den;
  p.v = q1 * p.inv_den;
  return p;
}

// gradient of (u,v) w.r.t. the projected 3-D point
DD_HD void project_point_bwd(const Intrinsics& c, const Proj& p, float gu, float gv, float gs[3]) {
  const float g0 = gu * p.inv_den, g1 = gv * p.inv_den;
  const float g2 = -(gu * p.u + gv * p.v) * p.inv_den;
  for (int k = 0; k < 3; ++k) gs[k] = g0 * c.K[k] + g1 * c.K[4 + k] + g2 * c.K[8 + k];
}

// the normalised sampling grid stored by the reference (tools.py:217-221)
DD_HD float grid_normalise(float pix, float inv_size_m1) { return (pix * inv_size_m1 - 0.5f) * 2.f; }

struct ImageDims {
  int W, H;
  float inv_wm1, inv_hm1;     // 1/(W-1), 1/(H-1)
};

DD_HD ImageDims image_dims(int W, int H) {
  ImageDims d;
  d.W = W; d.H = H;
  d.inv_wm1 = 1.f / static_cast<float>(W - 1);
  d.inv_hm1 = 1.f / static_cast<float>(H - 1);
  return d;
}

// ------------------------------------------------------------------------------------------------
// grid_sample(bilinear, padding_mode='border', align_corners=True)
// ------------------------------------------------------------------------------------------------
struct SampleCoord {
  int x0, y0;        // top-left tap
  float ax, ay;      // weights of the right / bottom taps
  float bx, by;      // weights of the left / top taps ((x0+1)-ix, as ATen computes them)
  float passx, passy;  // 1 where d(ix)/d(u) = 1, 0 where the border clip kills the gradient
};

DD_HD SampleCoord sample_coord(float gnx, float gny, int W, int H) {
  SampleCoord s;
  float ix = ((gnx + 1.f) * 0.5f) * static_cast<float>(W - 1);
  float iy = ((gny + 1.f) * 0.5f) * static_cast<float>(H - 1);
  const float mx = static_cast<float>(W - 1), my = static_cast<float>(H - 1);
  s.passx = (ix > 0.f && ix < mx) ? 1.f : 0.f;      // clip_coordinates_set_grad: <=0 or >=max -> 0
  s.passy = (iy > 0.f && iy < my) ? 1.f : 0.f;
  ix = ix < 0.f ? 0.f : (ix > mx ? mx : ix);
  iy = iy < 0.f ? 0.f : (iy > my ? my : iy);
  // NaN coordinates (z + eps == 0) fall through the comparisons; park them on pixel 0 with no gradient
  if (!(ix == ix)) { ix = 0.f; s.passx = 0.f; }
  if (!(iy == iy)) { iy = 0.f; s.passy = 0.f; }
  const float fx = dd_floor(ix), fy = dd_floor(iy);
  s.x0 = static_cast<int>(fx);
  s.y0 = static_cast<int>(fy);
  s.ax = ix - fx;
  s.ay = iy - fy;
  s.bx = (fx + 1.f) - ix;
  s.by = (fy + 1.f) - iy;
  return s;
}

// Samples one channel plane; also returns d(value)/d(ix), d(value)/d(iy) (already gated by the clip).
DD_HD float sample_plane(const float* plane, const SampleCoord& s, int W, int H, float& dvx, float& dvy) {
  // Border mode clamps the coordinate into [0, W-1]: the x+1 tap leaves the image only when ix == W-1 exactly, where its
  // weight ax is 0 and the clip gate passx is 0 -- so its value never reaches a result, and reading the in-range neighbour
  // instead (a finite number) gives the same output as the reference's "out of bounds -> 0" without a divergent load.
  const int dx = (s.x0 + 1) <= (W - 1) ? 1 : 0, dy = (s.y0 + 1) <= (H - 1) ? W : 0;
  const float* r0 = plane + s.y0 * W + s.x0;
  const float v00 = r0[0];
  const float v01 = r0[dx];
  const float v10 = r0[dy];
  const float v11 = r0[dy + dx];
  dvx = s.passx * (s.by * (v01 - v00) + s.ay * (v11 - v10));
  dvy = s.passy * (s.bx * (v10 - v00) + s.ax * (v11 - v01));
  return v00 * (s.bx * s.by) + v01 * (s.ax * s.by) + v10 * (s.bx * s.ay) + v11 * (s.ax * s.ay);
}

// ------------------------------------------------------------------------------------------------
// per-pixel, per-source-frame geometry: forward record + backward
// ------------------------------------------------------------------------------------------------
struct FrameGeom {
  Proj proj;          // projection of the final sample point
  float gnx, gny;     // normalised grid ('sample' output)
  float Pp[3];        // MODE_FLOW_MASK: P + m*r   (point the rigid transform is applied to)
  float r[3];         // MODE_FLOW_MASK / MODE_FLOW: residual flow c - ego   (zeros in MODE_RIGID)
  float ego[3];       // T*P - P
  float ego_gn[2];    // normalised grid of pi(T*P)       ('sample_ego', detached)
  float cmp_gn[2];    // normalised grid of pi(P + c)     ('sample_complete', detached)
};

// P = Z*ray; c = ts*up(flow) (unused in MODE_RIGID); m = up(mask) (MODE_FLOW_MASK only)
template <int MODE>
DD_HD void frame_geometry(const Intrinsics& cam, const float* T, const float P[3], const float c[3], float m,
                          const ImageDims& dim, float eps, FrameGeom& g) {
  float Q[3];
  rigid_apply(T, P, Q);       // MODE_FLOW needs it only for the ego / sample_ego side outputs
  for (int k = 0; k < 3; ++k) g.ego[k] = Q[k] - P[k];
  float S[3];
  if (MODE == MODE_RIGID) {
    for (int k = 0; k < 3; ++k) { S[k] = Q[k]; g.r[k] = 0.f; g.Pp[k] = P[k]; }
  } else {
    float Pc[3];
    for (int k = 0; k < 3; ++k) { g.r[k] = c[k] - g.ego[k]; Pc[k] = P[k] + c[k]; }
    const Proj pe = project_point(cam, Q, eps), pc = project_point(cam, Pc, eps);
    g.ego_gn[0] = grid_normalise(pe.u, dim.inv_wm1); g.ego_gn[1] = grid_normalise(pe.v, dim.inv_hm1);
    g.cmp_gn[0] = grid_normalise(pc.u, dim.inv_wm1); g.cmp_gn[1] = grid_normalise(pc.v, dim.inv_hm1);
    if (MODE == MODE_FLOW) {
      for (int k = 0; k < 3; ++k) { S[k] = Pc[k]; g.Pp[k] = Pc[k]; }
    } else {
      for (int k = 0; k < 3; ++k) g.Pp[k] = P[k] + g.r[k] * m;
      rigid_apply(T, g.Pp, S);
    }
  }
  g.proj = project_point(cam, S, eps);
  g.gnx = grid_normalise(g.proj.u, dim.inv_wm1);
  g.gny = grid_normalise(g.proj.v, dim.inv_hm1);
}

// Accumulated gradients of one pixel w.r.t. its own inputs.
struct PixelGrad {
  float gP[3];       // w.r.t. the back-projected point (summed over frames by the caller)
  float gc[3];       // w.r.t. c = ts*up(flow)
  float gm;          // w.r.t. m = up(mask)
  float gT[12];      // w.r.t. rows 0..2 of T (row-major 3x4)
};

// gu, gv: d loss / d (ix, iy) of the main sample (already border-gated).
// gr_extra: additional upstream on the residual flow r (c_consistency term); zeros otherwise.
template <int MODE>
DD_HD void frame_geometry_bwd(const Intrinsics& cam, const float* T, const float P[3], float m, const FrameGeom& g,
                              float gu, float gv, const float gr_extra[3], PixelGrad& out) {
  float gS[3];
  project_point_bwd(cam, g.proj, gu, gv, gS);
  for (int k = 0; k < 3; ++k) { out.gP[k] = 0.f; out.gc[k] = 0.f; }
  out.gm = 0.f;
  for (int k = 0; k < 12; ++k) out.gT[k] = 0.f;
  if (MODE == MODE_RIGID) {
    for (int i = 0; i < 3; ++i) {
      for (int k = 0; k < 3; ++k) { out.gT[i * 4 + k] = gS[i] * P[k]; out.gP[k] += T[i * 4 + k] * gS[i]; }
      out.gT[i * 4 + 3] = gS[i];
    }
  } else if (MODE == MODE_FLOW) {
    for (int k = 0; k < 3; ++k) { out.gP[k] = gS[k]; out.gc[k] = gS[k]; }
  } else {
    float gPp[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < 3; ++i) {
      for (int k = 0; k < 3; ++k) { out.gT[i * 4 + k] = gS[i] * g.Pp[k]; gPp[k] += T[i * 4 + k] * gS[i]; }
      out.gT[i * 4 + 3] = gS[i];
    }
    float gr[3], gQ[3];
    for (int k = 0; k < 3; ++k) {
      out.gP[k] = gPp[k];
      out.gm += gPp[k] * g.r[k];
      gr[k] = gPp[k] * m + gr_extra[k];
      out.gc[k] = gr[k];
      gQ[k] = -gr[k];            // r = c - (Q - P)
      out.gP[k] += gr[k];
    }
    for (int i = 0; i < 3; ++i) {
      for (int k = 0; k < 3; ++k) { out.gT[i * 4 + k] += gQ[i] * P[k]; out.gP[k] += T[i * 4 + k] * gQ[i]; }
      out.gT[i * 4 + 3] += gQ[i];
    }
  }
}

// d loss/d disp(full-res, up-sampled) from d loss/d P:  P = Z*ray, Z = 1/(lo + span*d)
DD_HD float depth_bwd(const DepthParams& dp, const float gP[3], const float ray[3], float Z) {
  const float gZ = gP[0] * ray[0] + gP[1] * ray[1] + gP[2] * ray[2];
  return -gZ * Z * Z * dp.span;
}

// ------------------------------------------------------------------------------------------------
// SSIM(3x3 box, reflect pad) + L1
// ------------------------------------------------------------------------------------------------
struct SsimStats {   // window SUMS (not means) over the 9 reflect-padded taps
  float sx, sy, sxx, syy, sxy;
};

struct SsimGrad {    // d ssim / d (mean_x, mean_xx, mean_xy), zero outside the clamp's pass band
  float dmu, dxx, dxy;
};

DD_HD float ssim_value(const SsimStats& w, SsimGrad* grad) {
  const float inv9 = 1.f / 9.f;
  const float mx = w.sx * inv9, my = w.sy * inv9;
  const float vx = w.sxx * inv9 - mx * mx;
  const float vy = w.syy * inv9 - my * my;
  const float vxy = w.sxy * inv9 - mx * my;
  const float a1 = 2.f * mx * my + kSsimC1, a2 = 2.f * vxy + kSsimC2;
  const float b1 = mx * mx + my * my + kSsimC1, b2 = vx + vy + kSsimC2;
  const float n = a1 * a2, d = b1 * b2;
  const float inv_d = dd_rcp(d);
  const float q = n * inv_d;                           // n/d
  const float val = (1.f - q) * 0.5f;
  if (grad) {
    const bool pass = (val >= 0.f) && (val <= 1.f);   // torch.clamp passes gradient on the closed interval
    if (pass) {
      // d(n/d) = (dn - q*dd)/d ; value = (1 - n/d)/2
      const float dn_dmu = 2.f * my * (a2 - a1), dd_dmu = 2.f * mx * (b2 - b1);
      grad->dmu = -0.5f * (dn_dmu - q * dd_dmu) * inv_d;
      grad->dxx = 0.5f * q * b1 * inv_d;               // -0.5 * (0 - q*b1)/d
      grad->dxy = -0.5f * (2.f * a1) * inv_d;
    } else {
      grad->dmu = grad->dxx = grad->dxy = 0.f;
    }
  }
  return val < 0.f ? 0.f : (val > 1.f ? 1.f : val);
}

// multiplicity with which pixel `p` appears among the reflect-padded taps {c-1, c, c+1} of centre `c`
DD_HD int reflect_multiplicity(int c, int p, int n) {
  int m = 0;
  for (int d = -1; d <= 1; ++d) m += (dd_reflect(c + d, n) == p) ? 1 : 0;
  return m;
}

}  // namespace dd
