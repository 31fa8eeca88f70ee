/* dynamo_hip.h -- C ABI of libdynamo_hip.so, the MI355X (gfx950) implementation of Dynamo-Depth's
 * per-step view-synthesis loss path.
 *
 * The reference has no FFI: its boundary is the Python operator surface of tools.py / Trainer.py
 * (SURVEY.md section 8(b)).  Each entry point below names the reference code it replaces; the
 * Python side (dynamo-depth_amd/hipops/) binds them with ctypes -- see INTEGRATION.md for the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 NCHW data unless stated otherwise;
 *   - the caller owns all memory (inputs, outputs, workspaces); the library allocates nothing, keeps
 *     no global state and holds no pointer past return;
 *   - launches are asynchronous on `stream` (a hipStream_t passed as void*), never synchronise, and
 *     are therefore hipGraph-capturable;
 *   - return value: 0 on success, otherwise a hipError_t (dd_error_string() decodes it);
 *   - "accumulate" outputs are added to (read-modify-write by the single owner of each element; no float atomic is left in
 *     the library, only integer counters) and must be initialised by the caller.
 */
#ifndef DYNAMO_HIP_H_
#define DYNAMO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DD_MAX_SCALES 4
#define DD_NUM_SRC 2                 /* source frames (-1, +1); reference options.py frame_ids default */
#define DD_ABI_VERSION 2

/* flow-composition mode = phase flags of Trainer.setup_phase (Trainer.py:466-490) */
#define DD_MODE_RIGID 0              /* disp_init   : bool_CmpFlow=False, bool_MotMask=False */
#define DD_MODE_FLOW 1               /* motion_init : bool_CmpFlow=True,  bool_MotMask=False */
#define DD_MODE_FLOW_MASK 2          /* mask_init / fine_tune : both True                      */

/* per-block partial record written by dd_photo_loss's tile kernel (floats; 30 used, +7 smoothness sums under dd_fused_loss) */
#define DD_PARTIAL_STRIDE 40
/* per-scale sums produced by dd_photo_loss (floats):
 *   [0] sum over B*H*W of the selected photometric loss          (Trainer.py:352 before .mean())
 *   [1],[2] sum over B*3*h*w of valid*(1-mask)*|residual_flow|   per source frame (Trainer.py:386)
 *   [3],[4] sum over B*h*w of disp_mag                           per source frame (Trainer.py:396-397)
 *   [5] number of pixels whose minimum was a warped frame (automask statistics)
 */
#define DD_SUMS_STRIDE 8

typedef struct DDPhotoScale {
  int shift;                         /* scale s: (h,w) = (H>>s, W>>s) */
  int h, w;
  float w_photo;                     /* weight of sums[0] in the differentiated total  */
  float w_cons;                      /* weight of sums[1]+sums[2] in the differentiated total */
  const float* disp;                 /* (B,1,h,w)  outputs[('disp',0,s)] */
  const float* flow[DD_NUM_SRC];     /* (B,3,h,w)  outputs[('complete_flow',f,s)]  (modes 1,2) */
  const float* mask[DD_NUM_SRC];     /* (B,1,h,w)  outputs[('motion_mask',f,s)]    (mode 2)    */
  const float* noise;                /* (B,2,H,W)  tie-break noise of Trainer.py:339, or NULL  */
  /* gradients of the weighted total (NULL when args.want_grad == 0).  Buffers must be zeroed by the caller; at
   * shift == 0 every element is overwritten by its single owner, at shift > 0 contributions are added atomically.
   * The two frames may share one buffer (the reference shares the motion mask between frames): it receives the sum. */
  float* g_disp;                     /* (B,1,h,w) */
  float* g_flow[DD_NUM_SRC];         /* (B,3,h,w) */
  float* g_mask[DD_NUM_SRC];         /* (B,1,h,w) */
  /* optional materialised outputs (NULL = skip) -- the dict entries of SURVEY.md Appendix B */
  float* out_color[DD_NUM_SRC];      /* (B,3,H,W)  outputs[('color',f,s)]                    */
  float* out_sample[DD_NUM_SRC];     /* (B,H,W,2)  outputs[('sample',f,s)]                   */
  float* out_depth;                  /* (B,1,H,W)  outputs[('depth',0,s)]                    */
  float* out_idsel;                  /* (B,H,W)    outputs['identity_selection/s'] (automask) */
  float* out_resid[DD_NUM_SRC];      /* (B,3,h,w)  outputs[('residual_flow',f,s)]  accumulate */
  float* out_delta[DD_NUM_SRC];      /* (B,h,w)    disp_mag of Trainer.py:396      accumulate */
} DDPhotoScale;

typedef struct DDPhotoArgs {
  int abi_version;                   /* DD_ABI_VERSION */
  int B, H, W;
  int num_scales;
  int mode;                          /* DD_MODE_* */
  int automask;                      /* Trainer.bool_automask (Trainer.py:117) */
  int want_grad;
  float min_depth, max_depth;        /* options.py:182-189 */
  float ssim_weight;                 /* options.py:115-118 */
  float eps;                         /* Project3D eps, tools.py:203 */
  float disp_thr;                    /* mask_disp_thrd, options.py:119-122 */
  const float* target;               /* (B,3,H,W) inputs[('color',0,0)] */
  const float* source[DD_NUM_SRC];   /* (B,3,H,W) inputs[('color',f,0)] */
  const float* source_packed[DD_NUM_SRC]; /* optional (NULL, or both set): (B,H,W,3) pixel-interleaved copies of source[f] (dd_pack_rgb) --
                                        same values; the warp gathers one 12-byte pixel per bilinear tap instead of three planes */
  const float* K;                    /* (B,4,4) inputs[('K',0)] */
  const float* inv_K;                /* (B,4,4) inputs[('inv_K',0)] */
  const float* T[DD_NUM_SRC];        /* (B,4,4) outputs[('cam_T_cam',0,f)], last row assumed (0,0,0,1) */
  const float* ts[DD_NUM_SRC];       /* (B,) fp32 inputs[('ts',f)] or NULL (=1) */
  float* g_T[DD_NUM_SRC];            /* (B,4,4) gradient w.r.t. T, overwritten (row 3 = 0); NULL if !want_grad */
  float* sums;                       /* (num_scales, DD_SUMS_STRIDE) overwritten */
  float* workspace;                  /* dd_photo_workspace_bytes() bytes, contents undefined */
  DDPhotoScale scale[DD_MAX_SCALES];
} DDPhotoArgs;

/* Replaces, in one launch pair, Trainer.generate_images_pred (Trainer.py:215-287) + the photometric /
 * automask / c_consistency part of Trainer.compute_losses (Trainer.py:316-352,384-386) + their autograd
 * backward: utils.interp, tools.disp_to_depth (tools.py:291), BackprojectDepth.forward (tools.py:191),
 * Project3D.forward (tools.py:211), F.grid_sample (Trainer.py:281), SSIM.forward (tools.py:243),
 * compute_reprojection_loss (Trainer.py:413). */
int dd_photo_loss(const DDPhotoArgs* args, void* stream);
size_t dd_photo_workspace_bytes(const DDPhotoArgs* args);
/* Measurement aid (bench.py's roofline leg): while enabled, every gradient-carrying dd_photo_loss call that is not under
 * stream capture records a HIP event pair around its photo_tile_kernel launch, on the stream the kernel is launched on
 * (at most 256 pairs between reads).  dd_photo_timing_read waits for the recorded launches, returns their mean duration
 * (ignoring the first `skip`) and forgets them.  Not thread-safe; no reference counterpart. */
int dd_photo_timing(int enable);
int dd_photo_timing_read(float* mean_us, int* launches, int skip);
/* dd_photo_loss in two calls: part 1 launches photo_tile_kernel alone, part 2 the launches that follow it (pyramid combine,
 * finalize); part 0 = dd_photo_loss.  A caller that records the step into hipGraphs can leave part 1 out of the recording and
 * issue it from the host between two graphs (the arguments are fixed addresses), where dd_photo_timing can bracket it with
 * events -- segments.SegmentedStep with time_tile_kernel, used by bench.py's roofline leg.  Same arguments for both parts. */
int dd_photo_loss_part(const DDPhotoArgs* args, void* stream, int part);

/* Edge-aware smoothness, forward + gradient in one pass.  Replaces tools.compute_smooth_loss
 * (tools.py:311-326) and, with normalise=1, the mean-normalisation of Trainer.py:357-359.
 *   inp (B,C,h,w), img (B,3,h,w) or NULL.
 *   sums[0] = sum |dx inp| e^{-mean_c|dx img|}, sums[1] = same in y  (caller divides by the two counts)
 *   g_inp (B,C,h,w) accumulate:  += weight * d( sums[0]/(B*C*h*(w-1)) + sums[1]/(B*C*(h-1)*w) )/d inp,
 *   or NULL.  workspace: dd_smooth_workspace_bytes(B,C,h,w). */
int dd_smooth_loss(const float* inp, const float* img, int B, int C, int h, int w, int normalise, float weight,
                   float* g_inp, float* sums, float* workspace, void* stream);
size_t dd_smooth_workspace_bytes(int B, int C, int h, int w);

/* Motion-mask sparsity (Trainer.py:393-399): BCE-with-logits(prob, 0) over pixels whose disp_mag is below the
 * batch-global mean, only if every image keeps at least one such pixel.
 *   delta (B,h,w) from dd_photo_loss.out_delta, delta_sum = pointer to its global sum (device scalar),
 *   prob (B,1,h,w); out[0] = loss value (mean over static pixels, 0 if gated off), out[1] = #static pixels;
 *   g_prob (B,1,h,w) accumulate: weight * d loss/d prob.  workspace: dd_sparsity_workspace_bytes(B,h,w). */
int dd_sparsity_loss(const float* delta, const float* delta_sum, const float* prob, int B, int h, int w, float weight,
                     float* g_prob, float* out, float* workspace, void* stream);
size_t dd_sparsity_workspace_bytes(int B, int h, int w);

/* Above-ground term (Trainer.py:361-364,425-461; tools.GroundPlane tools.py:76-164).
 *   disp (B,1,h,w), inv_K (B,4,4) of this scale, rand_idx (B, max_it*np_per_it) int32 indices into the bottom
 *   int(g_prior*h) rows (the reference draws them with the host NumPy RNG, tools.py:125-127).
 *   out[0] = sum of min(disp - ground_disp, 0) over valid pixels (caller: -out[0]/(B*h*w)/2^s),
 *   plane (B,3) = best plane parameters (before the +tol shift), g_disp (B,1,h,w) accumulate with `weight`
 *   = d(total)/d(out[0]) .  workspace: dd_ground_workspace_bytes(B,h,w,max_it). */
int dd_ground_loss(const float* disp, const float* inv_K, const int32_t* rand_idx, int B, int h, int w,
                   int np_per_it, int max_it, float tol, float g_prior, float min_depth, float max_depth,
                   float weight, float* g_disp, float* plane, float* out, float* workspace, void* stream);
size_t dd_ground_workspace_bytes(int B, int h, int w, int max_it);

/* The two halves of dd_ground_loss separately (parity tests pin each against the reference on its own, tests/test_ground_pin.py):
 * dd_ground_candidates: the least-squares plane of every RANSAC sample (tools.py:141-154 `calc_param`) -> cand (B*max_it,3),
 *   candidate j = b*max_it + it drawn from image b's ground points;
 * dd_ground_select: the reference's decision rule on GIVEN candidates (tools.py:129-137: candidate j scored on image j mod B,
 *   inlier fraction |dist| < tol, first maximum) and everything behind it (Trainer.py:361-364,436-461): counts (B*max_it) inliers
 *   per candidate, plane (B,3), out[0] as dd_ground_loss, g_disp accumulated (may be NULL).  Same workspace. */
int dd_ground_candidates(const float* disp, const float* inv_K, const int32_t* rand_idx, int B, int h, int w, int np_per_it, int max_it,
                         float g_prior, float min_depth, float max_depth, float* cand, void* stream);
int dd_ground_select(const float* disp, const float* inv_K, const float* cand, int B, int h, int w, int max_it, float tol, float g_prior,
                     float min_depth, float max_depth, float weight, float* g_disp, int32_t* counts, float* plane, float* out,
                     float* workspace, void* stream);

/* tools.GroundPlane.forward (tools.py:85-101) on an explicit point map: points (B,3,h,w) ->
 * dist (B,1,h,w) vertical distance to the best RANSAC plane, plane (B,3).  Same workspace size as above. */
int dd_ground_plane(const float* points, const int32_t* rand_idx, int B, int h, int w, int np_per_it, int max_it,
                    float tol, float g_prior, float* dist, float* plane, float* workspace, void* stream);

/* Folds the raw sums written by the kernels above into the `losses` dict values of Trainer.compute_losses
 * (Trainer.py:404-409) in one tiny launch (no host round trip, weights are launch-time scalars):
 *   term[s][t] = sum_i [scale_of[i]==s && term_of[i]==t] * norm[i] * res[i]
 *   out[1+t]   = sum_s term[s][t]                    ('loss_term/<name>')
 *   out[8+s]   = sum_t coef[t] * term[s][t]          ('loss_term/<s>')
 *   loss[0]    = sum_s out[8+s] / num_scales         ('loss')          out[0] = loss[0] as well
 * res entries with term_of[i] < 0 are ignored. */
#define DD_NUM_TERMS 7
#define DD_MAX_RES 128
typedef struct DDAssembleArgs {
  int n;                              /* entries of res */
  int num_scales;
  float coef[DD_NUM_TERMS];           /* losses['loss_coef/<name>'] in options.py g_* order */
  float norm[DD_MAX_RES];
  int8_t term_of[DD_MAX_RES];
  int8_t scale_of[DD_MAX_RES];
} DDAssembleArgs;
int dd_assemble_losses(const float* res, const DDAssembleArgs* args, float* loss, float* out, void* stream);

/* All regularisers of Trainer.compute_losses for every scale in five launches (the per-term entry points above take two to
 * four launches per term and scale -- about 45 per step at three scales); the smoothness of ALL smoothed tensors of a scale is
 * one pass that forms the edge weights once per pixel: edge-aware smoothness of disp / flow / mask
 * (tools.py:311-326, Trainer.py:355-359,380-381,401-402), mask sparsity (Trainer.py:393-399), ground term (Trainer.py:361-364,
 * 425-461).  Same per-element arithmetic as dd_smooth_loss / dd_sparsity_loss / dd_ground_loss; the smoothness sums are folded
 * per entry instead of per channel (a different, equally fixed order: the last bits of the value may differ from dd_smooth_loss).
 * A `smooth` entry with inp == NULL is skipped; the caller merges entries whose tensors are shared between the two frames
 * (weight = sum of the frames' weights).  A sparsity entry with prob == NULL and a ground entry with disp == NULL are skipped.
 * Raw sums go to res[scale * DD_REG_RES_STRIDE + slot]:
 *   slot 2*k, 2*k+1 (k < DD_REG_SMOOTH): smooth entry k -> sum |dx| e^-|dx img|, sum |dy| e^-|dy img|
 *   slot 10 + 2*f, 11 + 2*f           : sparsity frame f -> loss value (mean over static pixels, 0 if gated off), #static
 *   slot 14                            : ground -> sum of min(disp - ground_disp, 0)
 * Gradients are accumulated into the g_* buffers with the given weights, exactly like the per-term entry points. */
#define DD_REG_SMOOTH 5
#define DD_REG_RES_STRIDE 16
typedef struct DDRegSmooth {
  const float* inp;                  /* (B,C,h,w) */
  float* g_inp;                      /* accumulate, or NULL */
  int C;
  int normalise;                     /* 1: divide by the per-image mean first (disparity, C == 1) */
  float weight;                      /* d(total)/d(mean_x + mean_y) */
} DDRegSmooth;
typedef struct DDRegScale {
  int h, w;
  const float* img;                  /* (B,3,h,w) target pyramid level */
  DDRegSmooth smooth[DD_REG_SMOOTH];
  const float* delta[DD_NUM_SRC];    /* (B,h,w) disp_mag from dd_photo_loss */
  const float* delta_sum[DD_NUM_SRC];/* device scalar: its sum over the batch */
  const float* prob[DD_NUM_SRC];     /* (B,1,h,w) motion_prob, NULL = no sparsity term for this frame */
  float* g_prob[DD_NUM_SRC];
  float w_sparsity[DD_NUM_SRC];
  const float* disp;                 /* ground term: (B,1,h,w), NULL = off */
  float* g_disp;
  const float* inv_K;                /* (B,4,4) of this scale */
  const int32_t* rand_idx;           /* (B, max_it*np_per_it) */
  float* plane;                      /* (B,3) out */
  float w_ground;
} DDRegScale;
typedef struct DDRegArgs {
  int abi_version;
  int B, num_scales;
  int np_per_it, max_it;             /* RANSAC (options.py:198-213) */
  float tol, g_prior, min_depth, max_depth;
  float* res;                        /* (num_scales, DD_REG_RES_STRIDE), zeroed by the caller; the slots of active terms are overwritten */
  float* workspace;                  /* dd_reg_workspace_bytes() bytes */
  DDRegScale scale[DD_MAX_SCALES];
} DDRegArgs;
int dd_reg_losses(const DDRegArgs* args, void* stream);
size_t dd_reg_workspace_bytes(const DDRegArgs* args);
/* dd_reg_losses followed by dd_assemble_losses on the same `res` record (args->res; the caller's photometric sums already sit
 * behind the regulariser slots) with one launch less: the fold of the ground-hinge partials runs inside the assembling kernel.
 * Four launches for the regularisers of all scales + one for the loss assembly (reference Trainer.py:355-409). */
int dd_reg_losses_finish(const DDRegArgs* args, const DDAssembleArgs* assemble, float* loss, float* out, void* stream);

/* THE fused loss: dd_photo_loss + dd_reg_losses_finish on the same arguments in FIVE launches instead of ten (round 5) -- the whole of
 * Trainer.generate_images_pred + Trainer.compute_losses and their backward (Trainer.py:215-411, tools.py:76-164,191-257,291-326):
 *   1  the photometric tile kernel; at scale 0 the edge-aware smoothness of the disparity / flow / mask (tools.py:311-326,
 *      Trainer.py:355-359,380-381,401-402) is evaluated in its store stage and added to the pixel's gradient before its one store;
 *   2  one launch of tasks: low-res gradient footprints summed AND the smoothness of the scales >= 1 in the same pass | fold of the
 *      tile records | per-image disparity sums | RANSAC candidates + inlier counts (tools.py:114-154);
 *   3  static-pixel counts (Trainer.py:393-399) | per-image scalars: smoothness sums, mean, winning plane;
 *   4  sparsity gradient | mean-normalisation adjoint + ground hinge (Trainer.py:361-364,425-461);
 *   5  the losses dict values (Trainer.py:404-409).
 * `photo` and `reg` are filled exactly as for the two calls they replace (same workspaces: dd_photo_workspace_bytes /
 * dd_reg_workspace_bytes; photo->sums behind the regulariser slots of reg->res as dd_reg_losses_finish expects).  The request takes
 * this pipeline when it is a gradient pass whose two frames share their flow / mask tensors (or the rigid mode), every smoothness
 * entry is one of the photometric tensors (disparity, mean-normalised | flow | mask) accumulating into the photometric gradient
 * buffer, the scale-0 pyramid level is the target image and the rows of the scales >= 1 are whole 16-byte aligned quads --
 * dd_fused_loss_supported says so (0: no; 1: yes; 2: yes, and every element of every gradient buffer named in the two argument blocks
 * is written by exactly one plain store, so the caller need not zero them first -- motion_prob's gradient included, which under the
 * other entry points is an accumulate-type output); dd_fused_loss returns hipErrorNotSupported (801) otherwise and launches nothing:
 * the caller then issues dd_photo_loss + dd_reg_losses_finish.  The disparity smoothness is evaluated on the raw disparity and divided
 * by (mean + eps) per image afterwards (the reference divides first: equal up to rounding); gradients of one element are written
 * once (no read-modify-write of the flow / mask planes).  dd_fused_loss_part: 1 = the tile kernel alone, 2 = the launches behind it
 * (see dd_photo_loss_part). */
int dd_fused_loss(const DDPhotoArgs* photo, const DDRegArgs* reg, const DDAssembleArgs* assemble, float* loss, float* out, void* stream);
int dd_fused_loss_part(const DDPhotoArgs* photo, const DDRegArgs* reg, const DDAssembleArgs* assemble, float* loss, float* out, void* stream,
                       int part);
int dd_fused_loss_supported(const DDPhotoArgs* photo, const DDRegArgs* reg);

/* ---- baseline JPEG decoding of a batch of frames (csrc/dd_jpeg.hip; SURVEY.md 8(f) row 1, first stage) ------------------------
 * Replaces datasets/base_dataset.py:13-18 `pil_loader` (PIL / libjpeg decode of the three frames of every sample in the DataLoader
 * workers, base_dataset.py:140-147) for files that are already at the training resolution -- the reference's `downsample` image
 * type.  The result is PIL's, bit for bit: sequential Huffman decoding, ISLOW integer IDCT, "fancy" triangle chroma up-sampling,
 * 16-bit fixed-point YCbCr -> RGB.
 * One header record per image, filled on the host from the marker segments (hipops/jpeg.py parse_header): */
typedef struct DDJpegHeader {
  int32_t data_offset;               /* first byte of the entropy-coded segment, relative to the image's first byte */
  int32_t data_end;                  /* file length */
  int32_t width, height;
  int32_t restart_interval;          /* MCUs between RSTn markers, 0 = none */
  int32_t ncomp;
  int32_t h[3], v[3], tq[3], td[3], ta[3];
  int32_t reserved[3];               /* qt at byte 96: the kernels read the tables 16 bytes at a time; sizeof = 1696 = 106 x 16 */
  uint16_t qt[4][64];                /* quantisation tables, natural (de-zigzagged) order */
  uint8_t bits[4][16];               /* Huffman tables dc0, dc1, ac0, ac1: number of codes per length ... */
  uint8_t vals[4][256];              /* ... and the symbols in code order (T.81 B.2.4.2) */
} DDJpegHeader;
/* data: image i occupies data[i*stride, i*stride + headers[i].data_end); all images H x W with `ncomp` components of sampling (h,v)
 * (luma 1x1 / 2x1 / 1x2 / 2x2, chroma 1x1).  rgb: (n_images, H, W, 3) uint8, what dd_prepare_frames reads.  Three launches. */
size_t dd_jpeg_workspace_bytes(int n_images, int H, int W, int ncomp, const int* h, const int* v);
int dd_jpeg_decode(const unsigned char* data, long long stride, const DDJpegHeader* headers, int n_images, int H, int W, int ncomp, const int* h,
                   const int* v, unsigned char* rgb, void* workspace, size_t workspace_bytes, void* stream);

/* transforms.Resize((H, W), BICUBIC) on PIL frames (datasets/base_dataset.py:80,147) = Pillow's Image.resize, on the device, bit for
 * bit: src (n_images, Hs, Ws, 3) uint8 -> dst (slots, H, W, 3) uint8, image i written to slot dst_slot[i] (NULL: slot i).  The tap
 * tables are Pillow's precompute_coeffs + normalize_coeffs_8bpc (hipops/resize.py builds them): *_bounds (out, 2) = first input
 * index and tap count per output index, *_coef (out, ksize) 22-bit fixed-point weights; the tables of an axis whose size does not
 * change may be NULL (Pillow skips that pass).  workspace: dd_resize_workspace_bytes (the horizontal pass's output). */
size_t dd_resize_workspace_bytes(int n_images, int Hs, int Ws, int H, int W);
int dd_resize_bicubic(const unsigned char* src, int n_images, int Hs, int Ws, unsigned char* dst, const int32_t* dst_slot, int H, int W,
                      const int32_t* h_bounds, const int32_t* h_coef, int h_ksize, const int32_t* v_bounds, const int32_t* v_coef, int v_ksize,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- operator-level entry points: the tools.py modules one by one (forward; *_bwd = autograd) ---- */

/* tools.BackprojectDepth.forward (tools.py:191-197): depth (B,1,h,w), inv_K (B,4,4) -> points (B,4,h*w) */
int dd_backproject(const float* depth, const float* inv_K, int B, int h, int w, float* points, void* stream);
int dd_backproject_bwd(const float* g_points, const float* inv_K, int B, int h, int w, float* g_depth, void* stream);

/* tools.Project3D.forward (tools.py:211-224): points (B,4,N), K (B,4,4), T (B,4,4) or NULL ->
 * pix (B,h,w,2) normalised to [-1,1], ego (B,3,N) */
int dd_project3d(const float* points, const float* K, const float* T, int B, int h, int w, float eps,
                 float* pix, float* ego, void* stream);
/* g_points (B,4,N) overwritten; g_T (B,4,4) overwritten (NULL when T is NULL); workspace B*nblocks*12 floats */
int dd_project3d_bwd(const float* points, const float* K, const float* T, const float* g_pix, const float* g_ego,
                     int B, int h, int w, float eps, float* g_points, float* g_T, float* workspace, void* stream);
size_t dd_project3d_workspace_bytes(int B, int h, int w);

/* tools.SSIM.forward (tools.py:243-257): x, y (B,C,H,W) -> (B,C,H,W);  *_bwd: g_x, g_y overwritten (either may be NULL) */
int dd_ssim(const float* x, const float* y, int B, int C, int H, int W, float* out, void* stream);
int dd_ssim_bwd(const float* x, const float* y, const float* g_out, int B, int C, int H, int W,
                float* g_x, float* g_y, void* stream);

/* tools.disp_to_depth (tools.py:291-298) elementwise over n values */
int dd_disp_to_depth(const float* disp, size_t n, float min_depth, float max_depth, float* scaled, float* depth,
                     void* stream);

/* networks.layers.transformation_from_parameters (networks/layers.py:7-82): axisangle, translation (B,3) -> (B,4,4) */
int dd_pose_matrix(const float* axisangle, const float* translation, int B, int invert, float* T, void* stream);
int dd_pose_matrix_bwd(const float* axisangle, const float* translation, const float* g_T, int B, int invert,
                       float* g_axisangle, float* g_translation, void* stream);

/* Bias gradient of a convolution whose output gradient is channels-last: out[c] = sum over rows of x[row*C + c]
 * (x = (B,H,W,C) memory, rows = B*H*W).  Replaces ATen's generic reduction, which is ~100x off the HBM roofline for
 * small C (the full-resolution 9-channel convs of networks/motion_decoder.py); any C >= 1 (C > 256, the bias gradients of
 * LiteMono's point-wise Linears -- networks/depth_encoder.py:200-203 --, takes a column-per-thread kernel).
 * workspace: dd_channel_sum_workspace_bytes(C). */
int dd_channel_sum_nhwc(const float* x, long long rows, int C, float* out, float* workspace, void* stream);
size_t dd_channel_sum_workspace_bytes(int C);

/* nn.ReflectionPad2d(1) of networks.layers.Conv3x3 (reference networks/layers.py:100-115) for channels-last tensors:
 * x (B,H,W,C) memory -> out (B,H+2,W+2,C); *_bwd is its adjoint (g_out (B,H+2,W+2,C) -> g_x (B,H,W,C), overwritten). */
int dd_reflect_pad1_nhwc(const float* x, int B, int H, int W, int C, float* out, void* stream);
int dd_reflect_pad1_nhwc_bwd(const float* g_out, int B, int H, int W, int C, float* g_x, void* stream);

/* Depth-wise (groups == channels) 3x3 convolution, stride 1, zero padding == dilation, no bias, on channels-last tensors:
 * LiteMono's CDilated inside DilatedConv (reference networks/depth_encoder.py:168-181, 197-199 -- nn.Conv2d(dim, dim, 3,
 * padding=d, dilation=d, groups=dim, bias=False)). x, out, g_*: [B,H,W,C] floats (the memory of a channels_last NCHW tensor),
 * weight/g_weight: [C,1,3,3] contiguous; C a multiple of 4, <= 512. The weight gradient is a fixed-order two-level sum
 * (one record per image row in `workspace`, dd_dwconv3x3_workspace_bytes(B,H,C) bytes), so it is run-to-run reproducible. */
int dd_dwconv3x3_nhwc(const float* x, const float* weight, int B, int H, int W, int C, int dilation, float* out, void* stream);
int dd_dwconv3x3_nhwc_bwd_data(const float* g_out, const float* weight, int B, int H, int W, int C, int dilation, float* g_x, void* stream);
int dd_dwconv3x3_nhwc_bwd_weight(const float* g_out, const float* x, int B, int H, int W, int C, int dilation, float* g_weight,
                                 void* workspace, size_t workspace_bytes, void* stream);
size_t dd_dwconv3x3_workspace_bytes(int B, int H, int C);

/* Data gradient of a 3x3, stride-1 convolution with ONE output channel (the disparity heads: networks/depth_decoder.py:49-51,
 * 95-97, `Conv3x3(num_ch_dec[s], 1)`): g_out [B,Ho,Wo] (one channel), weight [1,C,3,3], g_x [B,Hi,Wi,C] channels-last,
 * Ho = Hi + 2*padding - 2, padding 0 (input already reflection-padded) or 1.  C a multiple of 4, <= 512. */
int dd_conv3x3_cout1_bwd_data(const float* g_out, const float* weight, int B, int Hi, int Wi, int C, int padding, float* g_x, void* stream);

/* Input side of a training step on the device.  The reference prepares every sample on the host in the
 * DataLoader workers: ToTensor, torchvision ColorJitter on the float tensor of each frame (a fresh draw per frame,
 * datasets/base_dataset.py:83-95,159-164), horizontal flip (:118-131); Trainer.apply_img_resize then builds the target pyramid
 * (Trainer.py:722-734).  Here the loader hands over the decoded uint8 frames and the drawn parameters.
 *   frames_u8 (B,F,H,W,3) uint8 RGB;  params (B,F,9) fp32 rows [apply, fn_idx[4] (0 brightness 1 contrast 2 saturation 3 hue),
 *   brightness, contrast, saturation, hue];  flip (B) int32.
 *   color, color_aug (F,B,3,H,W) fp32 in [0,1]: ('color',f,0) and ('color_aug',f,0) of frame f = one contiguous (B,3,H,W) block.
 * workspace: dd_prepare_frames_workspace_bytes(B,F).  Two launches. */
int dd_prepare_frames(const uint8_t* frames_u8, const float* params, const int32_t* flip, int B, int F, int H, int W, float* color,
                      float* color_aug, float* workspace, void* stream);
size_t dd_prepare_frames_workspace_bytes(int B, int F);
/* One pyramid level: dst = clamp(F.interpolate(src, (H/2,W/2), 'bicubic', align_corners=False, antialias=True), 0, 1) -- the tensor
 * Resize(BICUBIC) of Trainer.py:80 as ATen computes it (Keys cubic a = -0.5, support widened by the scale, border taps
 * renormalised, horizontal then vertical).  src (planes,H,W), dst (planes,H/2,W/2); H, W even. */
int dd_pyramid_down2(const float* src, int planes, int H, int W, float* dst, void* stream);
/* (B,3,H,W) planar fp32 -> (B,H,W,3) pixel-interleaved fp32, same values: the layout DDPhotoArgs.source_packed takes.  The
 * reference has no counterpart (F.grid_sample reads the planar tensor, Trainer.py:281); the input side issues it once per step
 * for the two source frames, the photometric kernel then gathers from it at every scale.  H*W must be a multiple of 4. */
int dd_pack_rgb(const float* planar, int B, int H, int W, float* packed, void* stream);

/* tools.DepthMetrics.forward without a mask (tools.py:16-73) and compute_errors (tools.py:269-288): sparse-LiDAR depth
 * metrics with per-image median scaling -- SURVEY.md 8(f) row 2, the accuracy gate of the evaluation.
 * disp [B,1,H,W] (outputs['disp_scaled',0,0]); lidar [B,M,3] = (row, col, depth) in ground-truth pixels, padded;
 * valid [B,M] floats (0 = padding); gt_dim [B,2] int32 (height, width) on the DEVICE; img_bound: 4 doubles on the HOST
 * (opt.eval_img_bound: up, down, left, right fractions). per_sample [B,8] = abs_rel, sq_rel, rms, log_rms, a1, a2, a3,
 * kept-point count; mean [7] = batch mean as the reference returns it. A sample with no kept point yields NaNs (the
 * reference raises). workspace: dd_depth_metrics_workspace_bytes(B, M). One workgroup per sample, no host sync. */
int dd_depth_metrics(const float* disp, int B, int H, int W, const float* lidar, const float* valid, int M, const int* gt_dim,
                     const double* img_bound, float min_depth, float max_depth, float* per_sample, float* mean,
                     void* workspace, size_t workspace_bytes, void* stream);
size_t dd_depth_metrics_workspace_bytes(int B, int M);
/* The mask branch of tools.DepthMetrics.forward (tools.py:23-25,58-72): `mask` (B,mask_h,mask_w) uint8 labels in ground-truth pixels
 * (the loaders' sem_mask / mot_mask).  per_label [B,256,8] = for every label that occurs among a sample's kept LiDAR points, the seven
 * errors over those points and their count (zero rows otherwise); the caller forms sum_b err*cnt and sum_b cnt per label.
 * workspace: dd_depth_metrics_masked_workspace_bytes(B, M). */
int dd_depth_metrics_masked(const float* disp, int B, int H, int W, const float* lidar, const float* valid, int M, const int* gt_dim,
                            const double* img_bound, float min_depth, float max_depth, const uint8_t* mask, int mask_h, int mask_w,
                            float* per_sample, float* mean, float* per_label, void* workspace, size_t workspace_bytes, void* stream);
size_t dd_depth_metrics_masked_workspace_bytes(int B, int M);

/* Training-mode nn.BatchNorm2d on a channels-last tensor, with the activation that follows it and an optional residual add
 * fused into the normalisation pass: out = act(bn(x) [+ residual]).  Covers torchvision BasicBlock's bn1->relu and
 * bn2(+identity)->relu as used by networks/resnet_encoder.py:42-88, the stem bn1->relu, and LiteMono's BNGELU / DilatedConv.bn1
 * (networks/depth_encoder.py:137-148,197-199).  x, residual, out, g_*: [rows, C] floats (rows = B*H*W of an NHWC tensor),
 * C a multiple of 4, <= 512.  act: 0 none, 1 ReLU, 2 GELU (erf; not with a residual).  running_mean/var (may both be NULL) are
 * updated in place with `momentum` (unbiased variance, as PyTorch); save_mean/save_invstd [C] feed the backward.
 * Backward: g_x always; g_residual (NULL unless a residual was given and act != 0) = gradient after the activation;
 * `out` (backward): the ReLU mask is read from it when given -- REQUIRED when a residual was added in the forward; NULL for act == 1 without a residual: the mask is recomputed from x (the forward's own fmaf, the same bits), one read pass less.  Batch statistics: fp32 within a chunk of rows, fp64 across chunks, fixed order.
 * workspace: dd_bn_workspace_bytes(C).  Two launches forward, two backward. */
int dd_bn_act_fwd(const float* x, const float* residual, long long rows, int C, const float* gamma, const float* beta, float eps,
                  float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd, int act,
                  float* out, void* workspace, size_t workspace_bytes, void* stream);
int dd_bn_act_bwd(const float* x, const float* g_out, const float* out, long long rows, int C, const float* gamma, const float* beta,
                  const float* save_mean, const float* save_invstd, int act, float* g_x, float* g_residual, float* g_gamma,
                  float* g_beta, void* workspace, size_t workspace_bytes, void* stream);
size_t dd_bn_workspace_bytes(int C);

/* Element-typed variants for the tensors of an autocast forward (config "fp16 convs" of BASELINE.json): `dtype` selects the type of
 * the activation / gradient tensors -- DD_DTYPE_F32, DD_DTYPE_F16, DD_DTYPE_BF16 -- read and written four elements at a time;
 * statistics, affine parameters, sums and every intermediate stay fp32.  The fp32 entry points above are dtype = 0 of these. */
#define DD_DTYPE_F32 0
#define DD_DTYPE_F16 1
#define DD_DTYPE_BF16 2
int dd_bn_act_fwd_t(const void* x, const void* residual, long long rows, int C, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, float* save_mean, float* save_invstd, int act, void* out, int dtype,
                    void* workspace, size_t workspace_bytes, void* stream);
int dd_bn_act_bwd_t(const void* x, const void* g_out, const void* out, long long rows, int C, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd, int act, void* g_x, void* g_residual, float* g_gamma, float* g_beta,
                    int dtype, void* workspace, size_t workspace_bytes, void* stream);
int dd_channel_sum_nhwc_t(const void* x, long long rows, int C, float* out, int dtype, float* workspace, void* stream);
int dd_reflect_pad1_nhwc_t(const void* x, int B, int H, int W, int C, void* out, int dtype, void* stream);
int dd_reflect_pad1_nhwc_bwd_t(const void* g_out, int B, int H, int W, int C, void* g_x, int dtype, void* stream);

/* The glue between two 3x3 convolutions of the disparity decoders (reference networks/depth_decoder.py:40-53 Monodepth2,
 * :98-113 Lite-Mono; networks/layers.py:84-121) in one pass, channels-last:
 *     out = ReflectionPad2d(1)( cat( up( act(x) ), skip ) )        out: (B, H+2, W+2, C1+C2)
 * x: (B,h,w,C1) -- with elu != 0 the PRE-activation output of the ConvBlock's convolution, act = ELU(alpha=1), else act = identity;
 * up (mode): 0 nearest x2 (F.interpolate(scale_factor=2, mode="nearest")), 1 bilinear x2 (mode="bilinear", align_corners=False),
 * 2 none (H,W = h,w: ELU + padding only); skip: (B,H,W,C2) or NULL with C2 = 0.  C1, C2 multiples of 4.
 * *_bwd: g_out (B,H+2,W+2,C1+C2) -> g_x (B,h,w,C1) (times ELU'(x) when elu) and g_skip (B,H,W,C2); either may be NULL; gather
 * form, no atomics, overwritten.  dtype: DD_DTYPE_*. */
int dd_up_cat_pad_t(const void* x, const void* skip, int B, int h, int w, int C1, int C2, int mode, int elu, void* out, int dtype, void* stream);
int dd_up_cat_pad_bwd_t(const void* g_out, const void* x, int B, int h, int w, int C1, int C2, int mode, int elu, void* g_x, void* g_skip,
                        int dtype, void* stream);

/* LayerNorm over the last (channel) axis of a [rows, C] matrix -- LiteMono's LayerNorm(data_format="channels_last")
 * (networks/depth_encoder.py:101-128, used by LGFI at :241,:252): y = (x - mean) * rstd * gamma + beta, biased variance, eps
 * inside the square root.  C a multiple of 4, <= 256.  mean, rstd: [rows], kept for the backward.
 * Backward: g_x [rows, C]; g_gamma_beta [2*C] = (d gamma, d beta), fixed-order column sums.  workspace:
 * dd_layer_norm_workspace_bytes(C).  One launch forward, two backward. */
int dd_layer_norm_fwd(const float* x, long long rows, int C, const float* gamma, const float* beta, float eps, float* y, float* mean,
                      float* rstd, void* stream);
int dd_layer_norm_bwd(const float* x, const float* g_out, const float* gamma, const float* mean, const float* rstd, long long rows, int C,
                      float* g_x, float* g_gamma_beta, void* workspace, size_t workspace_bytes, void* stream);
size_t dd_layer_norm_workspace_bytes(int C);

/* Backward of LiteMono's layer-scale residual out = res + y * scale[b,c] (scale = gamma x stochastic-depth factor; reference
 * networks/depth_encoder.py:219-226,266-274): g_y[b,r,c] = g_out[b,r,c] * scale[b,c] and g_scale[b,c] = sum_r g_out * y, one pass.
 * g_out, y, g_y: [B, rows, C] (rows = H*W of a channels-last image), scale, g_scale: [B, C]; C a multiple of 4, <= 1024.
 * workspace: dd_layer_scale_workspace_bytes(B, C).  The residual's own gradient is g_out itself. */
int dd_layer_scale_bwd(const float* g_out, const float* y, const float* scale, int B, int rows, int C, float* g_y, float* g_scale,
                       void* workspace, size_t workspace_bytes, void* stream);
size_t dd_layer_scale_workspace_bytes(int B, int C);

/* The same LiteMono hooks on the tensors of an autocast forward (BASELINE.json config 5 with the LiteMono depth net; round 3):
 * `dtype` is the storage type of the activations and their gradients (0 fp32, 1 fp16, 2 bf16, as for dd_bn_act_*_t); the
 * affine / convolution weights stay the fp32 master copies, statistics and every sum are fp32.  Half types move 8 bytes per
 * access (four channels). */
int dd_layer_norm_fwd_t(const void* x, long long rows, int C, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd,
                        int dtype, void* stream);
int dd_layer_norm_bwd_t(const void* x, const void* g_out, const float* gamma, const float* mean, const float* rstd, long long rows, int C,
                        void* g_x, float* g_gamma_beta, void* workspace, size_t workspace_bytes, int dtype, void* stream);
int dd_layer_scale_bwd_t(const void* g_out, const void* y, const float* scale, int B, int rows, int C, void* g_y, float* g_scale, void* workspace,
                         size_t workspace_bytes, int dtype, void* stream);
/* forward of the layer-scale residual, out = res + y * scale[b, c] (all (B,rows,C) channels-last, scale (B,C) fp32), evaluated in
 * fp32 whatever the storage type: the layer-scale parameters start at 1e-6, below fp16's normal range */
int dd_layer_scale_fwd_t(const void* res, const void* y, const float* scale, int B, int rows, int C, void* out, int dtype, void* stream);
int dd_dwconv3x3_nhwc_t(const void* x, const float* weight, int B, int H, int W, int C, int dilation, void* out, int dtype, void* stream);
int dd_dwconv3x3_nhwc_bwd_data_t(const void* g_out, const float* weight, int B, int H, int W, int C, int dilation, void* g_x, int dtype, void* stream);
int dd_dwconv3x3_nhwc_bwd_weight_t(const void* g_out, const void* x, int B, int H, int W, int C, int dilation, float* g_weight, void* workspace,
                                   size_t workspace_bytes, int dtype, void* stream);

/* The 1x1 reductions of the motion decoders as ONE operator (reference networks/motion_decoder.py:33,66: `refine_motion_redu{level}` =
 * Conv2d(2*ch, out_dim, 1) on cat(a, b)): y[p,co] = bias[co] + sum_c a[p,c] W[co,c] + sum_c b[p,c] W[co,C+c].
 * a, b, g_a, g_b: [P, C] fp32 (channels-last tensors as matrices, P = B*H*W); y, g_out: [P, cout]; weight, g_weight: [cout, 2C] rows (the
 * memory of a (cout, 2C, 1, 1) tensor in either layout); bias, g_bias: [cout] or NULL.  C in {64, 128, 256, 512}, cout in {1, 3}
 * (dd_redu_supported).  Forward one launch, data gradients one launch (either of g_a / g_b may be NULL), weight + bias gradient three
 * launches (per-workgroup partials, two-level fixed-order fold: bit-reproducible); workspace: dd_redu_workspace_bytes(P, C, cout). */
int dd_redu_supported(int C, int cout);
size_t dd_redu_workspace_bytes(long long P, int C, int cout);
int dd_redu_fwd(const float* a, const float* b, const float* weight, const float* bias, long long P, int C, int cout, float* y, void* stream);
int dd_redu_bwd_data(const float* g_out, const float* weight, long long P, int C, int cout, float* g_a, float* g_b, void* stream);
int dd_redu_bwd_weight(const float* a, const float* b, const float* g_out, long long P, int C, int cout, float* g_weight, float* g_bias, void* workspace,
                       size_t workspace_bytes, void* stream);

/* The disparity heads: 3x3, stride-1 convolution to ONE output channel on an input that already carries its reflection padding
 * (reference networks/depth_decoder.py:49-51,95-97 `Conv3x3(num_ch_dec[s], 1)`; the data gradient is dd_conv3x3_cout1_bwd_data).
 * x_padded (B,Hp,Wp,C) channels-last fp32, C = 32 or 64 (dd_conv_head_supported); out / g_out (B,Hp-2,Wp-2).  weight (1,C,3,3) addressed
 * through its element strides; g_weight (3,3,C) dense = the memory order of a channels-last (1,C,3,3) weight; g_bias (1) or NULL.
 * One launch forward; three for the weight + bias gradient (per-workgroup partials, two-level fixed-order fold: bit-reproducible);
 * workspace: dd_conv_head_workspace_bytes(B, Hp, Wp, C). */
int dd_conv_head_supported(int C);
size_t dd_conv_head_workspace_bytes(int B, int Hp, int Wp, int C);
int dd_conv_head_fwd(const float* x_padded, const float* weight, long long s_ci, long long s_kh, long long s_kw, const float* bias, int B, int Hp, int Wp,
                     int C, float* out, void* stream);
int dd_conv_head_bwd_weight(const float* x_padded, const float* g_out, int B, int Hp, int Wp, int C, float* g_weight, float* g_bias, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Convolutions on a handful of channels at full resolution -- the finest level of the motion decoders (reference
 * networks/motion_decoder.py:24-33,57-66: `refine_motion_conv5` = two 3x3 convolutions on 9-12 channels, `refine_motion_redu5` a 1x1
 * reduction to 3 / 1 channels, at 192x640).  stride 1, padding ks/2, dilation 1, groups 1, ks = 1 or 3, fp32.
 * x (B,H,W,cin), y / g_out (B,H,W,cout), g_x (B,H,W,cin): channels-last, dense.  weight: (cout,cin,ks,ks) addressed through its four
 * element strides (any layout).  g_weight: (cout,ks,ks,cin) dense -- the memory order of a channels-last weight tensor; g_bias (cout) or NULL.
 * dd_conv_small_supported: 1 when the (ks, cin, cout) combination is instantiated (forward AND data gradient), else the caller keeps
 * the library's convolution.  workspace: dd_conv_small_workspace_bytes(ks, cin, cout), private to the call's stream.
 * Forward / data gradient: two launches (weight re-ordering, direct convolution); weight gradient: three launches (matrix-pipe
 * partials per workgroup, a two-level fixed-order fold that also yields the bias gradient).  No atomics: bit-reproducible. */
int dd_conv_small_supported(int ks, int cin, int cout);
size_t dd_conv_small_workspace_bytes(int ks, int cin, int cout);
int dd_conv_small_fwd(const float* x, const float* weight, long long s_co, long long s_ci, long long s_kh, long long s_kw, const float* bias, int B, int H,
                      int W, int cin, int cout, int ks, float* y, void* workspace, size_t workspace_bytes, void* stream);
int dd_conv_small_bwd_data(const float* g_out, const float* weight, long long s_co, long long s_ci, long long s_kh, long long s_kw, int B, int H, int W, int cin,
                           int cout, int ks, float* g_x, void* workspace, size_t workspace_bytes, void* stream);
int dd_conv_small_bwd_weight(const float* x, const float* g_out, int B, int H, int W, int cin, int cout, int ks, float* g_weight, float* g_bias,
                             void* workspace, size_t workspace_bytes, void* stream);

/* 3x3 stride-1 convolutions on 16+ channels at fp32 accuracy on the bf16 matrix pipe -- the motion decoders' refinement convolutions
 * (reference networks/motion_decoder.py:24-33,57-66; 64-512 channels per level) and the ResNet encoders' basic blocks (reference
 * networks/resnet_encoder.py via torchvision BasicBlock).  Every fp32 operand is split exactly into three bf16 pieces and each product is
 * formed from six v_mfma_f32_32x32x16_bf16 partial products with fp32 accumulation: the dropped cross terms are at most 2^-23 |x w| (the size of one fp32 rounding), the
 * result has the accuracy of an fp32 FMA chain (csrc/dd_conv_mfma.hip).
 * x (B,Hi,Wi,k_in), y (B,Ho,Wo,n_out): channels-last, dense, fp32; Ho = Hi + 2 pad - 2 (pad 0: pre-padded input, 1: 'same', 2: the data
 * gradient of a pad-0 convolution); x is zero-extended.  k_in % 4 == 0, x 16-byte aligned.
 * dd_conv3x3_mfma_pack: weight (cout,cin,3,3) addressed through its four element strides -> the split weights in matrix-fragment order,
 * pack_fwd (dd_conv3x3_mfma_pack_bytes(cout, cin) bytes) for the forward and / or pack_bwd_data (dd_conv3x3_mfma_pack_bytes(cin, cout)),
 * transposed and mirrored, for the data gradient g_x = dd_conv3x3_mfma(g_out, pack_bwd_data, NULL, ..., k_in = cout, n_out = cin, 2 - pad).
 * dd_conv3x3_mfma_bwd_weight (pad 0 or 1, cin % 4 == cout % 4 == 0): g_weight (cout,3,3,cin) dense -- the memory order of a channels-last
 * weight -- from x and g_out (B,Ho,Wo,cout), the same split arithmetic with the pixels as the contraction; workspace
 * dd_conv3x3_mfma_wgrad_workspace_bytes(B, Ho, Wo, cin, cout), private to the call's stream: one partial per workgroup, folded in a
 * fixed order by a second launch.  No atomics anywhere: every result is bit-reproducible. */
int dd_conv3x3_mfma_supported(int cin, int cout);
size_t dd_conv3x3_mfma_pack_bytes(int n_out, int k_in);
int dd_conv3x3_mfma_pack(const float* weight, long long s_co, long long s_ci, long long s_kh, long long s_kw, int cout, int cin, void* pack_fwd,
                         void* pack_bwd_data, void* stream);
/* The _n forms take the number of partial products per multiply-add: 6 (what the forms without _n compute: fp32 accuracy), 3 = the two
 * leading bf16 pieces of each operand, x1w1 + x1w2 + x2w1 (relative error of a product <= 2^-16: PyTorch's
 * torch.set_float32_matmul_precision("high"), "bf16x3" -- more accurate than the TF32 arithmetic cuDNN runs the reference's fp32
 * convolutions in by default on Ampere-class GPUs), 1 = operands rounded to bf16 ("medium").  Same packs, same layouts, same workspaces. */
int dd_conv3x3_mfma_n(const float* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, int products, float* y,
                      void* stream);
/* The packs of MANY layers in one launch (a network's 3x3 layers at the top of its forward pass instead of one pack launch in front of
 * every convolution).  jobs (device memory): dd_conv3x3_mfma_pack_many_job_words() = 10 64-bit words per layer --
 * { weight pointer, s_co, s_ci, s_kh, s_kw, cout, cin, pack_fwd pointer, pack_bwd_data pointer (0: none), first workgroup of the layer };
 * a layer takes dd_conv3x3_mfma_pack_many_blocks(cout, cin, want_fwd, want_bwd_data) consecutive workgroups; block_job (device memory,
 * n_blocks entries): the layer index of every workgroup.  Every byte written is what dd_conv3x3_mfma_pack writes for that layer
 * (tests/test_conv_mfma_gpu.py::test_packs_of_many_layers_in_one_launch_are_the_single_packs). */
int dd_conv3x3_mfma_pack_many_job_words(void);
int dd_conv3x3_mfma_pack_many_blocks(int cout, int cin, int want_fwd, int want_bwd_data);
int dd_conv3x3_mfma_pack_many(const long long* jobs, const int* block_job, int n_blocks, void* stream);
int dd_conv3x3_mfma(const float* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, float* y, void* stream);
/* The same convolution (pad 1) for SMALL images -- the encoders' and motion decoders' deep levels, 12 x 40 and 6 x 20 pixels with 256 / 512
 * channels -- where 8 x 32-pixel tiles waste half an image and B*H*W / 32 M blocks do not fill the chip: flat 256-pixel tiles of the whole
 * batch, the contraction split across workgroups, partial sums folded in split order (bit-reproducible).  W <= 40, k_in % 4 == 0,
 * n_out % 4 == 0 and 33..64 or > 96 (the two-block pack layout); same packs as dd_conv3x3_mfma (forward: pack_fwd; data gradient: pack_bwd_data on g_out with k_in = cout, n_out = cin).
 * workspace: dd_conv3x3_mfma_flat_workspace_bytes(...) bytes, private to the call's stream. */
int dd_conv3x3_mfma_flat_supported(int B, int H, int W, int k_in, int n_out);
size_t dd_conv3x3_mfma_flat_workspace_bytes(int B, int H, int W, int k_in, int n_out);
int dd_conv3x3_mfma_flat(const float* x, const void* pack, const float* bias, int B, int H, int W, int k_in, int n_out, float* y, void* workspace,
                         size_t workspace_bytes, void* stream);
int dd_conv3x3_mfma_flat_n(const float* x, const void* pack, const float* bias, int B, int H, int W, int k_in, int n_out, int products, float* y,
                           void* workspace, size_t workspace_bytes, void* stream);
size_t dd_conv3x3_mfma_wgrad_workspace_bytes(int B, int Ho, int Wo, int cin, int cout);
int dd_conv3x3_mfma_bwd_weight(const float* x, const float* g_out, int B, int Hi, int Wi, int cin, int cout, int pad, float* g_weight, void* workspace,
                               size_t workspace_bytes, void* stream);
int dd_conv3x3_mfma_bwd_weight_n(const float* x, const float* g_out, int B, int Hi, int Wi, int cin, int cout, int pad, int products, float* g_weight,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* The same 3x3 stride-1 convolution for HALF-PRECISION networks (BASELINE.json config 5: "fp16 (CDNA4 MFMA conv)"; the layers of reference
 * networks/resnet_encoder.py:95-135 via torchvision BasicBlock, networks/depth_decoder.py:10-55, networks/motion_decoder.py:24-33,48-66 when
 * the networks run under autocast -- the reference itself has no AMP).  x (B,Hi,Wi,k_in) and y (B,Ho,Wo,n_out) are channels-last, dense, in
 * the half type `dtype` (DD_DTYPE_F16 or DD_DTYPE_BF16); ONE v_mfma_f32_32x32x16_{f16,bf16} per operand pair, fp32 accumulation, fp32 bias
 * (may be NULL), the result rounded to nearest even once (csrc/dd_conv_half.hip).  k_in % 8 == 0, x 16-byte aligned, pad 0..2 as above.
 * dd_conv3x3_half_pack: the fp32 master weight (cout,cin,3,3) addressed through its four element strides -> half-precision fragments,
 * pack_fwd (dd_conv3x3_half_pack_bytes(cout, cin) bytes) and / or pack_bwd_data (dd_conv3x3_half_pack_bytes(cin, cout)), transposed and
 * mirrored, for the data gradient g_x = dd_conv3x3_half(g_out, pack_bwd_data, NULL, ..., k_in = cout, n_out = cin, 2 - pad, ...): the
 * cast autocast would launch per layer happens in the pack.  Bit-reproducible (no atomics, fixed order). */
int dd_conv3x3_half_supported(int cin, int cout);
size_t dd_conv3x3_half_pack_bytes(int n_out, int k_in);
int dd_conv3x3_half_pack(const float* weight, long long s_co, long long s_ci, long long s_kh, long long s_kw, int cout, int cin, int dtype,
                         void* pack_fwd, void* pack_bwd_data, void* stream);
int dd_conv3x3_half(const void* x, const void* pack, const float* bias, int B, int Hi, int Wi, int k_in, int n_out, int pad, int dtype, void* y,
                    void* stream);
/* Its weight gradient (pad 0 or 1, cin % 8 == cout % 8 == 0): g_weight (cout,3,3,cin) dense **fp32** -- the master weights' precision; the
 * products are half x half on the matrix pipe with fp32 accumulation, the pixels as the contraction -- from x and g_out (B,Ho,Wo,cout) in
 * the half type; workspace dd_conv3x3_half_wgrad_workspace_bytes(B, Ho, Wo, cin, cout) bytes, private to the call's stream: one fp32 partial
 * per workgroup, folded in a fixed order by a second launch (no atomics, no zero-fill, no promotion pass; bit-reproducible). */
size_t dd_conv3x3_half_wgrad_workspace_bytes(int B, int Ho, int Wo, int cin, int cout);
int dd_conv3x3_half_bwd_weight(const void* x, const void* g_out, int B, int Hi, int Wi, int cin, int cout, int pad, int dtype, float* g_weight,
                               void* workspace, size_t workspace_bytes, void* stream);

/* LiteMono's point-wise Linears (reference networks/depth_encoder.py:200-203 `pwconv1` / `act` / `pwconv2`, applied at :216-224 and
 * :262-272: nn.Linear(C, 6C) -> nn.GELU() -> nn.Linear(6C, C) on a channels-last (B,H,W,C) tensor, C = 64 / 128 / 224) at fp32 accuracy
 * on the bf16 matrix pipe, with the split arithmetic of dd_conv3x3_mfma (csrc/dd_pw_gemm.hip).
 * dd_pw_gemm: y (M,N) = act(x (M,K)) . W^T + bias, dense row-major fp32; K % 16 == 0, x 16-byte aligned; `pack` holds the split W in
 * matrix-fragment order (dd_pw_gemm_pack_bytes(N, K) bytes); gelu_in != 0 applies the exact (erf) GELU to x while it is split -- the
 * second Linear reads the first one's pre-activation and the activated tensor is never written.  bias may be NULL.
 * dd_mlp_pack: ONE launch packs up to four operands of a block from w1 (hidden,C) and w2 (C,hidden), each addressed through its two
 * element strides: pack_fwd1 (N = hidden, K = C), pack_fwd2 (N = C, K = hidden), and for the data gradients pack_bwd2 = w2 transposed
 * (g_post = g . w2: N = hidden, K = C) and pack_bwd1 = w1 transposed (g_y = g_pre . w1: N = C, K = hidden); NULL skips an operand.
 * dd_mlp_fwd (C = 64 or 128: dd_mlp_fwd_supported): the whole block forward out (M,C) = GELU(x . w1^T + b1) . w2^T + b2 in ONE kernel for
 * passes that keep nothing for a backward (the statistics-only side batch of Trainer.py:215-222's three-frame depth pass, evaluation):
 * the 6C-wide hidden tile stays in accumulator registers between the two GEMMs; operands pack_fwd1 and pack_fwd2_fused
 * (dd_pw_gemm_pack_bytes(C, hidden) bytes: w2 by hidden block, contraction order = the accumulator's register order).
 * dd_gelu_pair: the activation's backward in one pass over n elements (n % 4 == 0, 16-byte aligned): post = GELU(pre) (what the second
 * Linear's weight gradient contracts with) and g_inout <- g_inout * GELU'(pre), ATen's arithmetic.  Everything is bit-reproducible. */
size_t dd_pw_gemm_pack_bytes(int N, int K);
int dd_mlp_pack(const float* w1, long long s1_n, long long s1_k, const float* w2, long long s2_n, long long s2_k, int C, int hidden, void* pack_fwd1,
                void* pack_fwd2, void* pack_bwd2, void* pack_bwd1, void* pack_fwd2_fused, void* stream);
int dd_mlp_fwd_supported(int C);
int dd_mlp_fwd(const float* x, const void* pack_fwd1, const void* pack_fwd2_fused, const float* b1, const float* b2, int M, int C, float* y, void* stream);
int dd_pw_gemm(const float* x, const void* pack, const float* bias, int M, int K, int N, int gelu_in, float* y, void* stream);
int dd_gelu_pair(const float* pre, float* g_inout, float* post, size_t n, void* stream);

/* The Adam update of every parameter tensor of a step in ONE launch behind a one-thread-per-tensor prologue (reference Trainer.py:150
 * `optimizer.step()` on torch.optim.Adam, Trainer.py:492-497).  `records` (device memory): one per parameter
 * tensor -- dense fp32 arrays of n elements each, `step` the tensor's step counter as torch keeps it for capturable optimizers (a
 * float on the device; the prologue adds 1 to it, as optimizer.step() does, and evaluates the bias corrections from it).
 * `block_map` (device memory): n_blocks pairs (record index, chunk index), chunk = dd_adam_chunk() elements; the caller lists every
 * chunk of every record once.  `aux`: 8 * n_records bytes of device scratch.  grad_scale / found_inf: NULL, or GradScaler's device
 * scalars (gradients are divided by *grad_scale on the fly -- the gradient arrays are NOT written back --, and nothing is updated,
 * the counters included, when *found_inf != 0).  Update rule and operation order of torch's kernel; bit-reproducible. */
typedef struct DDAdamRecord {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  float* step;
  long long n;
} DDAdamRecord;
int dd_adam_chunk(void);
int dd_adam_multi(const DDAdamRecord* records, int n_records, const int* block_map, int n_blocks, void* aux, double lr, double beta1, double beta2,
                  double eps, double weight_decay, const float* grad_scale, const float* found_inf, void* stream);

const char* dd_error_string(int code);
int dd_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DYNAMO_HIP_H_ */
