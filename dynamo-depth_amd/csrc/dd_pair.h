// dd_pair.h -- the per-pixel arithmetic of dd_math.h for BOTH source frames at once.
//
// The two source frames (-1, +1) of a target pixel go through identical arithmetic on different data.  Holding the two
// values of every per-frame quantity in one two-float vector (`f2`: element 0 = first source frame, element 1 = second)
// lets the gfx950 compiler emit the packed fp32 VALU forms (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two fp32 results
// per lane per issue slot) with the pair living in an aligned VGPR pair from the start -- no re-packing moves.  The
// photometric kernel is VALU-issue bound, so this halves the cost of everything per-frame.
//
// Same formulas, same association order as dd_math.h (which stays the single-frame statement used by the stand-alone
// operators); `DD_HD` as there, so tests/hostmath/ runs these functions on the CPU (g++ vector extension) against the oracle.
//
// Reference semantics: see the header of dd_math.h (tools.py:191-257,291-298; Trainer.py:248-281,413-423).
#pragma once

#include "dd_math.h"

namespace dd {

#if defined(__clang__)
typedef float f2 __attribute__((ext_vector_type(2)));
#else
typedef float f2 __attribute__((vector_size(8)));
#endif

DD_HD f2 mk2(float a, float b) { f2 r = {a, b}; return r; }
DD_HD f2 sp2(float a) { f2 r = {a, a}; return r; }

// 1/x per element: v_rcp_f32 + one Newton step (the step is packed)
DD_HD f2 rcp2(f2 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const f2 r = mk2(__builtin_amdgcn_rcpf(x[0]), __builtin_amdgcn_rcpf(x[1]));
  return (sp2(1.f) - x * r) * r + r;
#else
  return mk2(1.f / x[0], 1.f / x[1]);
#endif
}

// n / d per element as IEEE division delivers it in all but last-bit corner cases: v_rcp_f32 (1 ulp), the product, and one
// residual correction q + r (n - q d) with fused multiply-adds.  In particular n == d gives exactly 1 -- the flat-patch SSIM
// (n == d, clamp((1 - n/d)/2, 0, 1) on its lower edge, tools.py:255-257) is exactly 0 as in the reference; the bare reciprocal
// leaves 1 - 2^-24 there.  r: the reciprocal, for the gradient factors that the reference forms as further divisions by d.
DD_HD f2 div2(f2 n, f2 d, f2& r) {
#if defined(__HIP_DEVICE_COMPILE__)
  r = mk2(__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]));
  const f2 q = n * r;
  return (n - q * d) * r + q;
#else
  r = mk2(1.f / d[0], 1.f / d[1]);
  return mk2(n[0] / d[0], n[1] / d[1]);
#endif
}

DD_HD f2 abs2(f2 x) { return mk2(dd_abs(x[0]), dd_abs(x[1])); }
// sign(x) with sign(0) = 0 (abs'(0) = 0 in torch): x * 2^126 * 2^126 saturated to [-1, 1] -- exact for every non-zero x
// including the denormals (2^-149 * 2^252 >= 1; one factor alone would leave a denormal difference at a fraction), two packed
// multiplies + two v_med3 instead of four compares and four selects
DD_HD f2 sign2(f2 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const f2 big = (x * sp2(8.507059173023462e37f)) * sp2(8.507059173023462e37f);
  return mk2(__builtin_amdgcn_fmed3f(big[0], -1.f, 1.f), __builtin_amdgcn_fmed3f(big[1], -1.f, 1.f));
#else
  return mk2(dd_sign(x[0]), dd_sign(x[1]));
#endif
}
DD_HD float hsum(f2 x) { return x[0] + x[1]; }

// rows 0..2 of both frames' rigid transforms: T[i*4+k] = {T_first[i][k], T_second[i][k]}
struct PairT {
  f2 m[12];
};

DD_HD void load_pair_T(PairT& t, const float* Ta, const float* Tb) {
  for (int k = 0; k < 12; ++k) t.m[k] = mk2(Ta[k], Tb[k]);
}

// T * [p;1] for a frame-independent point
DD_HD void rigid_apply2(const PairT& T, const float p[3], f2 q[3]) {
  for (int i = 0; i < 3; ++i) q[i] = T.m[i * 4 + 0] * sp2(p[0]) + T.m[i * 4 + 1] * sp2(p[1]) + T.m[i * 4 + 2] * sp2(p[2]) + T.m[i * 4 + 3];
}

// T * [p;1] for a per-frame point
DD_HD void rigid_apply2(const PairT& T, const f2 p[3], f2 q[3]) {
  for (int i = 0; i < 3; ++i) q[i] = T.m[i * 4 + 0] * p[0] + T.m[i * 4 + 1] * p[1] + T.m[i * 4 + 2] * p[2] + T.m[i * 4 + 3];
}

struct Proj2 {
  f2 u, v, inv_den;
};

DD_HD Proj2 project_point2(const Intrinsics& c, const f2 s[3], float eps) {
  const f2 q0 = sp2(c.K[0]) * s[0] + sp2(c.K[1]) * s[1] + sp2(c.K[2]) * s[2] + sp2(c.K[3]);
  const f2 q1 = sp2(c.K[4]) * s[0] + sp2(c.K[5]) * s[1] + sp2(c.K[6]) * s[2] + sp2(c.K[7]);
  const f2 q2 = sp2(c.K[8]) * s[0] + sp2(c.K[9]) * s[1] + sp2(c.K[10]) * s[2] + sp2(c.K[11]);
  Proj2 p;
  p.inv_den = rcp2(q2 + sp2(eps));          // v_rcp_f32 + one Newton step (tools.py:216 divides; z + eps is not clamped)
  p.u = q0 * p.inv_den;
  p.v = q1 * p.inv_den;
  return p;
}

DD_HD void project_point_bwd2(const Intrinsics& c, const Proj2& p, f2 gu, f2 gv, f2 gs[3]) {
  const f2 g0 = gu * p.inv_den, g1 = gv * p.inv_den;
  const f2 g2 = -(gu * p.u + gv * p.v) * p.inv_den;
  for (int k = 0; k < 3; ++k) gs[k] = g0 * sp2(c.K[k]) + g1 * sp2(c.K[4 + k]) + g2 * sp2(c.K[8 + k]);
}

DD_HD f2 grid_normalise2(f2 pix, float inv_size_m1) { return (pix * sp2(inv_size_m1) - sp2(0.5f)) * sp2(2.f); }

// ------------------------------------------------------------------------------------------------
// grid_sample(bilinear, border, align_corners=True), both frames
// ------------------------------------------------------------------------------------------------
struct SampleCoord2 {
  unsigned o00[2], dxb[2], dyb[2];   // BYTE offset of the top-left tap inside a (H,W) fp32 plane, and the steps to the right / lower taps
  f2 w00, w01, w10, w11;             // tap weights ((x0+1)-ix etc., as ATen forms them)
  f2 pby, pay, pbx, pax;             // weights of the spatial derivative, already gated by the border clip
};

// clamp to [0, hi] with NaN -> 0 (a NaN coordinate, z + eps == 0, parks on pixel 0)
DD_HD float clamp_coord(float v, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fmed3f(v, 0.f, hi);          // IEEE mode: a NaN operand yields min3 of the others = 0
#else
  return (v == v) ? (v < 0.f ? 0.f : (v > hi ? hi : v)) : 0.f;
#endif
}

// Border mode clamps the coordinate into [0, W-1]: the x+1 tap leaves the image only when ix == W-1 exactly, where its
// weight is 0 and the clip gate is 0 -- its value never reaches a result, so the in-range neighbour is read instead (step 0)
// and no load diverges.  Plane offsets are formed in fp32 (exact: H*W < 2^24, checked by the entry point).
// (ixr, iyr): the projected pixel coordinates.  The reference stores them normalised to [-1,1] and grid_sample
// un-normalises again ((g+1)/2*(W-1)); that round trip moves a coordinate by a few ulp (~1e-4 px at W=640) and is not repeated.
DD_HD SampleCoord2 sample_coord2(f2 ixr, f2 iyr, int W, int H) {
  SampleCoord2 s;
  const float mx = static_cast<float>(W - 1), my = static_cast<float>(H - 1);
  f2 ix, iy, fx, fy, passx, passy;
  for (int e = 0; e < 2; ++e) {
    passx[e] = (ixr[e] > 0.f && ixr[e] < mx) ? 1.f : 0.f;      // clip_coordinates_set_grad: <=0 or >=max (or NaN) -> 0
    passy[e] = (iyr[e] > 0.f && iyr[e] < my) ? 1.f : 0.f;
    ix[e] = clamp_coord(ixr[e], mx);
    iy[e] = clamp_coord(iyr[e], my);
    fx[e] = dd_floor(ix[e]);
    fy[e] = dd_floor(iy[e]);
  }
  const f2 ax = ix - fx, ay = iy - fy;
  const f2 bx = (fx + sp2(1.f)) - ix, by = (fy + sp2(1.f)) - iy;
  const f2 of = fy * sp2(static_cast<float>(W)) + fx;
  for (int e = 0; e < 2; ++e) {
    s.o00[e] = static_cast<unsigned>(of[e]) * 4u;
    s.dxb[e] = fx[e] < mx ? 4u : 0u;
    s.dyb[e] = fy[e] < my ? static_cast<unsigned>(W) * 4u : 0u;
  }
  s.w00 = bx * by; s.w01 = ax * by; s.w10 = bx * ay; s.w11 = ax * ay;
  s.pby = passx * by; s.pay = passx * ay; s.pbx = passy * bx; s.pax = passy * ax;
  return s;
}

// value and spatial derivative of one channel from the four taps of both frames
DD_HD f2 sample_taps2(const SampleCoord2& s, f2 v00, f2 v01, f2 v10, f2 v11, f2& dvx, f2& dvy) {
  dvx = s.pby * (v01 - v00) + s.pay * (v11 - v10);
  dvy = s.pbx * (v10 - v00) + s.pax * (v11 - v01);
  return v00 * s.w00 + v01 * s.w01 + v10 * s.w10 + v11 * s.w11;
}

// ------------------------------------------------------------------------------------------------
// geometry of one target pixel against both source frames
// ------------------------------------------------------------------------------------------------
struct PairGeom {        // what the backward pass needs
  Proj2 proj;            // projection of the final sample point
  f2 Pp[3];              // MODE_FLOW_MASK: P + m*r (the point T is applied to); otherwise unused
  f2 r[3];               // MODE_FLOW_MASK: residual flow c - ego
};

struct PairSide {        // forward-only side products
  f2 dgx, dgy;           // MODE_FLOW*: sample_ego - sample_complete (difference of the normalised grids)
  f2 r[3];               // MODE_FLOW*: residual flow
};

template <int MODE>
DD_HD void frame_geometry2(const Intrinsics& cam, const PairT& T, const float P[3], const f2 c[3], f2 m, const ImageDims& dim,
                           float eps, PairGeom& g, PairSide& sd) {
  f2 S[3];
  if (MODE == MODE_RIGID) {
    rigid_apply2(T, P, S);
  } else {
    f2 Q[3], Pc[3];
    rigid_apply2(T, P, Q);
    for (int k = 0; k < 3; ++k) {
      sd.r[k] = c[k] - (Q[k] - sp2(P[k]));
      Pc[k] = sp2(P[k]) + c[k];
    }
    const Proj2 pe = project_point2(cam, Q, eps), pc = project_point2(cam, Pc, eps);
    // (2(a/(W-1) - 0.5)) - (2(b/(W-1) - 0.5)) = (a - b) * 2/(W-1)
    sd.dgx = (pe.u - pc.u) * sp2(2.f * dim.inv_wm1);
    sd.dgy = (pe.v - pc.v) * sp2(2.f * dim.inv_hm1);
    if (MODE == MODE_FLOW) {
      for (int k = 0; k < 3; ++k) S[k] = Pc[k];
    } else {
      for (int k = 0; k < 3; ++k) { g.r[k] = sd.r[k]; g.Pp[k] = sp2(P[k]) + sd.r[k] * m; }
      rigid_apply2(T, g.Pp, S);
    }
  }
  g.proj = project_point2(cam, S, eps);
}

struct PairGrad {
  f2 gP[3];        // w.r.t. the back-projected point, per frame (the caller adds the two)
  f2 gc[3];        // w.r.t. c = ts*up(flow)
  f2 gm;           // w.r.t. m = up(mask)
  f2 gT[12];       // w.r.t. rows 0..2 of T
};

// gu, gv: d loss / d (ix, iy) of the main sample (already border-gated); gr_extra: upstream of c_consistency on r
template <int MODE>
DD_HD void frame_geometry_bwd2(const Intrinsics& cam, const PairT& T, const float P[3], f2 m, const PairGeom& g, f2 gu, f2 gv,
                               const f2 gr_extra[3], PairGrad& out) {
  f2 gS[3];
  project_point_bwd2(cam, g.proj, gu, gv, gS);
  const f2 zero = sp2(0.f);
  for (int k = 0; k < 3; ++k) { out.gP[k] = zero; out.gc[k] = zero; }
  out.gm = zero;
  if (MODE == MODE_RIGID) {
    for (int i = 0; i < 3; ++i) {
      for (int k = 0; k < 3; ++k) { out.gT[i * 4 + k] = gS[i] * sp2(P[k]); out.gP[k] += T.m[i * 4 + k] * gS[i]; }
      out.gT[i * 4 + 3] = gS[i];
    }
  } else if (MODE == MODE_FLOW) {
    for (int k = 0; k < 12; ++k) out.gT[k] = zero;
    for (int k = 0; k < 3; ++k) { out.gP[k] = gS[k]; out.gc[k] = gS[k]; }
  } else {
    f2 gPp[3] = {zero, zero, zero};
    for (int i = 0; i < 3; ++i) {
      for (int k = 0; k < 3; ++k) { out.gT[i * 4 + k] = gS[i] * g.Pp[k]; gPp[k] += T.m[i * 4 + k] * gS[i]; }
      out.gT[i * 4 + 3] = gS[i];
    }
    f2 gr[3], gQ[3];
    for (int k = 0; k < 3; ++k) {
      out.gP[k] = gPp[k];
      out.gm += gPp[k] * g.r[k];
      gr[k] = gPp[k] * m + gr_extra[k];
      out.gc[k] = gr[k];
      gQ[k] = -gr[k];            // r = c - (Q - P)
      out.gP[k] += gr[k];
    }
    for (int i = 0; i < 3; ++i) {
      for (int k = 0; k < 3; ++k) { out.gT[i * 4 + k] += gQ[i] * sp2(P[k]); out.gP[k] += T.m[i * 4 + k] * gQ[i]; }
      out.gT[i * 4 + 3] += gQ[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// SSIM(3x3 box, reflect pad) of both warped frames against the target
// ------------------------------------------------------------------------------------------------
struct SsimGrad2 {   // gscale * d ssim / d (mean_x, mean_xx (doubled), mean_xy), zero outside the clamp's pass band
  f2 dmu, dxx2, dxy;
};

// window SUMS over the 9 taps: sx, sxx, sxy per frame; sy, syy of the target.  The gradient comes out multiplied by
// `gscale` (the caller's loss weight) with the clamp's pass band folded into the same factor.
template <bool WITH_GRAD>
DD_HD f2 ssim_value2(f2 sx, f2 sxx, f2 sxy, float sy, float syy, float gscale, SsimGrad2& grad) {
  const float inv9 = 1.f / 9.f;
  const float my = sy * inv9;
  const float vy = syy * inv9 - my * my;
  const f2 mx = sx * sp2(inv9);
  const f2 vx = sxx * sp2(inv9) - mx * mx;
  const f2 vxy = sxy * sp2(inv9) - mx * sp2(my);
  // x and y enter every pair (a1, b1), (a2, b2) through the same sequence of roundings: a window with x == y bit for bit
  // (identical frames, the auto-mask's identity term on a static scene) gives n == d bit for bit, hence q == 1 and an SSIM
  // of exactly 0, as the reference's unfused arithmetic does (2 mu_x mu_y = mu_x^2 + mu_y^2 there too)
  const f2 mxy = mx * sp2(my);
  const f2 a1 = mxy + (mxy + sp2(kSsimC1)), a2 = vxy + (vxy + sp2(kSsimC2));
  const f2 b1 = mx * mx + (sp2(my * my) + sp2(kSsimC1)), b2 = vx + (sp2(vy) + sp2(kSsimC2));
  const f2 n = a1 * a2, d = b1 * b2;
  f2 inv_d;
  const f2 q = div2(n, d, inv_d);
  const f2 val = (sp2(1.f) - q) * sp2(0.5f);
  f2 out;
#if defined(__HIP_DEVICE_COMPILE__)
  // clamp(val, 0, 1) as one v_med3_f32 per frame (two compares + two selects otherwise).  A NaN value comes out as 0 here where the
  // compare chain passes it on: the L1 term of the same pixel (|y - x| of the same NaN colour) carries it into the loss either way.
  for (int e = 0; e < 2; ++e) out[e] = __builtin_amdgcn_fmed3f(val[e], 0.f, 1.f);
#else
  for (int e = 0; e < 2; ++e) out[e] = val[e] < 0.f ? 0.f : (val[e] > 1.f ? 1.f : val[e]);
#endif
  if (WITH_GRAD) {
    // torch.clamp passes gradient on the closed interval [0,1]: exactly where clamping changed nothing
    const f2 hp = inv_d * mk2(out[0] == val[0] ? gscale : 0.f, out[1] == val[1] ? gscale : 0.f);    // gscale * pass / d
    const f2 dn_dmu = sp2(2.f * my) * (a2 - a1), dd_dmu = sp2(2.f) * mx * (b2 - b1);
    // d(n/d) = (dn - q*dd)/d ; value = (1 - n/d)/2
    grad.dmu = sp2(0.5f) * (q * dd_dmu - dn_dmu) * hp;
    grad.dxx2 = q * b1 * hp;                    // 2 * (0.5 * q * b1 / d)
    grad.dxy = -(a1 * hp);                      // -0.5 * (2 a1) / d
  }
  return out;
}

}  // namespace dd
