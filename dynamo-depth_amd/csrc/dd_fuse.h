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

// launches photo_tile_kernel (part 1) / nothing else: the caller (dd_fused_loss) issues the passes behind it.
// Returns hipErrorInvalidValue when the arguments do not take the fused path (the caller falls back).
int launch_tile_fused(const DDPhotoArgs& a, const FuseInfo& fuse, hipStream_t stream);
bool frames_share_tensors(const DDPhotoArgs& a);
int gradient_channels(const DDPhotoArgs& a);
size_t footprint_floats(const DDPhotoArgs& a, long long off[DD_MAX_SCALES]);

}  // namespace dd
