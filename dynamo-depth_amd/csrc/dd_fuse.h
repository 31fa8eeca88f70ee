// dd_fuse.h -- what dd_photo.hip and dd_reg.hip share for dd_fused_loss (round 5): the tile geometry of the photometric kernel
// (the regulariser side adds up its low-res gradient footprints), the smoothness request the tile kernel carries out in its store
// stage at scale 0, and the tile kernel's launcher.  Private to csrc/: nothing here is part of the C ABI.
//
// dd_fused_loss is "the fused warp + SSIM + smoothness loss" of the north star in five launches (ten before):
//   1 photo_tile_kernel       warp + SSIM + selection + backward of every scale, and at scale 0 the edge-aware smoothness of the
//                             disparity / flow / mask (value sums + gradient, added to the pixel's gradient before its ONE store)
//   2 fused_post_kernel       tasks: low-res footprint sums (scales >= 1) + their smoothness in the same pass | fold of the tile
//                             records | per-image disparity sums | RANSAC candidates + scoring (matrix pipe)
//   3 fused_mid_kernel        tasks: static-pixel counts | per-image scalars (smoothness sums, mean, winning plane)
//   4 fused_fin_kernel        tasks: sparsity gradient | disparity-gradient finish (normalisation adjoint + ground hinge)
//   5 fused_finish_kernel     hinge fold + per-image folds + the losses dict values
// Reference: Trainer.py:215-411, tools.py:76-164,191-257,291-326.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_math.h"

namespace dd {

#ifndef DD_TH
#define DD_TH 16         // 16x32 tiles = 512 threads, ~77 KB LDS: two workgroups per CU (16 waves) whose barrier-separated
#define DD_TW 32         // stages interleave; measured 5-9 % faster than one 16x64 / 1024-thread workgroup per CU
#define DD_MIN_WAVES 4   // <= 128 VGPRs so that both workgroups fit
#endif
constexpr int TH = DD_TH;         // tile height (target pixels); multiple of 8 (coarsest scale block)
constexpr int TW = DD_TW;         // tile width

// Per-tile low-res gradient footprints (scale >= 1) go to the workspace with plain stores; the combine pass sums the
// <= 4 tiles that overlap each low-res pixel in a fixed order: deterministic gradients, no device atomics.
struct FootprintInfo {
  float* base;                    // workspace area behind the per-block records
  long long off[DD_MAX_SCALES];   // float offset of scale si (unused for shift == 0)
  const float* idrho;             // auto-mask: (B,2,H,W) identity reprojection loss of both frames (photo_identity_kernel), behind the footprints
};

// record of one tile-kernel workgroup (DD_PARTIAL_STRIDE floats): [0] photo, [1] n_warp, [2..3] cons, [4..5] delta,
// [6 + f*12 + k] gT, and with the fused smoothness (scale 0 only, zeros otherwise):
constexpr int REC_SMOOTH = 30;    // [30] sx_d [31] sy_d [32] dot_d [33] sx_c [34] sy_c [35] sx_m [36] sy_m   (raw, un-normalised)
constexpr int NSMOOTH = 7;

// smoothness groups: 0 disparity (mean-normalised, C = 1) | 1 complete flow (C = 3) | 2 motion mask (C = 1)
struct FuseScale {
  float wx[3], wy[3];             // weight / (B*C*h*(w-1)), weight / (B*C*(h-1)*w) per group; 0 = the group is off
  float* g_tmp;                   // (B,h,w): d total / d (normalised disparity) -- the mean's adjoint is applied by the finishing pass
};
struct FuseInfo {
  int on;                         // any group at any scale
  FuseScale sc[DD_MAX_SCALES];
};

// disp_b == points plane 0 when invK_b == nullptr: then the three coordinates are read from a (3,h*w) tensor
__device__ __forceinline__ void ground_point(const float* __restrict__ disp_b, const float* __restrict__ invK_b, DepthParams dp,
                                             int w, int pix, float P[3], int n = 0) {
  if (invK_b == nullptr) {
    P[0] = disp_b[pix]; P[1] = disp_b[n + pix]; P[2] = disp_b[2 * n + pix];
    return;
  }
  const int y = pix / w, x = pix % w;
  const float Z = 1.f / (dp.lo + dp.span * disp_b[pix]);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    P[i] = Z * (invK_b[i * 4 + 0] * static_cast<float>(x) + invK_b[i * 4 + 1] * static_cast<float>(y) + invK_b[i * 4 + 2]);
}

// (AtA + 1e-6 on EVERY entry)^-1 At B (tools.py:152), solved in double to stay clear of the conditioning of 5 nearby points
// candidate j = b*max_it + it: least squares through its np points of image b (tools.py:141-154)
__device__ __forceinline__ void ground_candidate_solve(const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                       const int32_t* __restrict__ rand_idx, int B, int h, int w, int rows, int np, int max_it,
                                                       DepthParams dp, int j, float out[3]) {
  const int b = j / max_it, it = j % max_it;
  const int n = h * w, base = (h - rows) * w;
  double M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, r[3] = {0, 0, 0};
  for (int k = 0; k < np; ++k) {
    const int idx = rand_idx[(size_t)b * max_it * np + it * np + k];
    float P[3];
    ground_point(disp + (size_t)b * n * (inv_K ? 1 : 3), inv_K ? inv_K + b * 16 : nullptr, dp, w, base + idx, P, n);
    const double av[3] = {P[0], P[2], 1.0};
    for (int i = 0; i < 3; ++i) {
      for (int l = 0; l < 3; ++l) M[i][l] += av[i] * av[l];
      r[i] += av[i] * P[1];
    }
  }
  for (int i = 0; i < 3; ++i)
    for (int l = 0; l < 3; ++l) M[i][l] += 1e-6;
  // 3x3 inverse by cofactors
  const double c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1], c01 = M[1][2] * M[2][0] - M[1][0] * M[2][2],
               c02 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
  const double det = M[0][0] * c00 + M[0][1] * c01 + M[0][2] * c02;
  const double id = 1.0 / det;
  const double inv[3][3] = {
      {c00 * id, (M[0][2] * M[2][1] - M[0][1] * M[2][2]) * id, (M[0][1] * M[1][2] - M[0][2] * M[1][1]) * id},
      {c01 * id, (M[0][0] * M[2][2] - M[0][2] * M[2][0]) * id, (M[0][2] * M[1][0] - M[0][0] * M[1][2]) * id},
      {c02 * id, (M[0][1] * M[2][0] - M[0][0] * M[2][1]) * id, (M[0][0] * M[1][1] - M[0][1] * M[1][0]) * id}};
  for (int i = 0; i < 3; ++i) out[i] = static_cast<float>(inv[i][0] * r[0] + inv[i][1] * r[1] + inv[i][2] * r[2]);
}

// Photo-independent preparation that rides in the tile kernel's launch (dd_fused_loss): ONE extra workgroup per (image, scale) --
// blockIdx.x == the tile count -- solves the image's max_it RANSAC candidates (tools.py:141-154) and sums its disparity plane (the
// mean of Trainer.py:358); the passes behind the tile kernel then start from finished candidates and sums instead of computing them in
// a launch (or in every scoring workgroup) of their own.
struct SideScale {
  const float* inv_K;             // (B,4,4) of this scale; nullptr: no ground term
  const int32_t* rand_idx;        // (B, max_it*np)
  float* cand;                    // (B*max_it, 3) out
  float* mean_partial;            // (B, 32) out: [b*32] = plane sum, the rest zeros (plane_mean folds 32 partials); nullptr: no mean
  int rows;
};
struct SideInfo {
  int on;                         // 1: the launch carries the extra workgroups (gridDim.x = tiles + 1)
  int np, max_it;
  SideScale sc[DD_MAX_SCALES];
};

// launches photo_tile_kernel (part 1) / nothing else: the caller (dd_fused_loss) issues the passes behind it.
// Returns hipErrorInvalidValue when the arguments do not take the fused path (the caller falls back).
int launch_tile_fused(const DDPhotoArgs& a, const FuseInfo& fuse, const SideInfo& side, hipStream_t stream);
bool frames_share_tensors(const DDPhotoArgs& a);
int gradient_channels(const DDPhotoArgs& a);
size_t footprint_floats(const DDPhotoArgs& a, long long off[DD_MAX_SCALES]);

}  // namespace dd
