// tools.DepthMetrics on the device (reference tools.py:6-73 + compute_errors tools.py:269-288), mask=None case.
// The reference loops over the batch in Python: per sample one full-resolution bilinear resize of the disparity to the
// ground-truth size (375x1242 on KITTI), boolean-mask gathers, two torch.median calls and ~20 small reductions with
// .item() syncs.  Here: one workgroup per sample, no host sync.  Only the <= 25 000 LiDAR pixels are interpolated; the two
// medians are exact (radix select on the float bit patterns, all values are positive); the seven means accumulate in fp64.
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"

namespace dd {

constexpr int DM_NT = 1024;
constexpr int DM_WAVES = DM_NT / 64;

__device__ __forceinline__ double dm_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum, result valid in every thread; `red` holds DM_WAVES doubles
__device__ double dm_block_sum(double v, double* red) {
  v = dm_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int w = 0; w < DM_WAVES; ++w) r += red[w];
  return r;
}

// k-th smallest (0-based) of the entries of v[0..M) that are > 0 (dropped points carry -1); all kept values are positive
// floats, whose bit patterns order like unsigned integers.  Four 8-bit passes, a 256-bin LDS histogram each.
__device__ float dm_select(const float* __restrict__ v, int M, int k, unsigned* hist, unsigned* state) {
  unsigned prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += DM_NT) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < M; i += DM_NT) {
      const float f = v[i];
      const unsigned u = __float_as_uint(f);
      if (f > 0.f && (u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int rem = k;
      unsigned bin = 0;
      for (; bin < 255; ++bin) {
        const int c = (int)hist[bin];
        if (rem < c) break;
        rem -= c;
      }
      state[0] = bin;
      state[1] = (unsigned)rem;
    }
    __syncthreads();
    prefix |= state[0] << shift;
    mask |= 255u << shift;
    k = (int)state[1];
  }
  return __uint_as_float(prefix);
}

// F.interpolate(..., mode='bilinear', align_corners=False) source tap of destination index d (ATen: scale = in/out in float,
// src = max(scale*(d+0.5)-0.5, 0))
__device__ __forceinline__ void dm_tap(int d, float scale, int in_size, int& i0, int& i1, float& w1) {
  float src = scale * (static_cast<float>(d) + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = static_cast<int>(src);
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  w1 = src - static_cast<float>(i0);
}

// MASKED: `mask` (B, mask_h, mask_w) uint8 labels in ground-truth pixels; per_label (B,256,8) receives, for every label that
// occurs among the sample's kept LiDAR points, the seven errors over those points and their count (tools.py:58-72).
template <bool MASKED>
__global__ __launch_bounds__(DM_NT) void depth_metrics_kernel(const float* __restrict__ disp, int H, int W, const float* __restrict__ lidar,
                                                              const float* __restrict__ valid, int M, const int* __restrict__ gt_dim,
                                                              double b_up, double b_down, double b_left, double b_right, float min_depth,
                                                              float max_depth, float* __restrict__ per_sample, float* __restrict__ ws,
                                                              const uint8_t* __restrict__ mask, int mask_h, int mask_w,
                                                              float* __restrict__ per_label) {
  __shared__ unsigned hist[256];
  __shared__ unsigned label_count[256];
  __shared__ unsigned state[2];
  __shared__ double red[DM_WAVES];
  const int b = blockIdx.x;
  const float* dp = disp + (size_t)b * H * W;
  const float* pts = lidar + (size_t)b * M * 3;
  const float* vl = valid + (size_t)b * M;
  float* w_gt = ws + (size_t)b * (MASKED ? 3 : 2) * M;
  float* w_pd = w_gt + M;
  int* w_label = reinterpret_cast<int*>(w_pd + M);           // MASKED only
  if (MASKED) {
    for (int i = threadIdx.x; i < 256; i += DM_NT) label_count[i] = 0;
    for (int i = threadIdx.x; i < 256 * 8; i += DM_NT) per_label[(size_t)b * 256 * 8 + i] = 0.f;
    __syncthreads();
  }
  const int gh = gt_dim[b * 2], gw = gt_dim[b * 2 + 1];
  // int(self.img_bound[i] * gt_height): Python float (double) product truncated
  const int up = (int)(b_up * gh), down = (int)(b_down * gh);
  const int left = (int)(b_left * gw), right = (int)(b_right * gw);
  const float sy = (float)H / (float)gh, sx = (float)W / (float)gw;

  double cnt = 0.0;
  for (int i = threadIdx.x; i < M; i += DM_NT) {
    const float r = pts[i * 3], c = pts[i * 3 + 1], z = pts[i * 3 + 2];
    const bool keep = vl[i] != 0.f && r >= (float)up && r < (float)down && c >= (float)left && c < (float)right && z > min_depth && z < max_depth;
    float g = -1.f, p = -1.f;
    if (keep) {
      const int row = (int)r, col = (int)c;                 // .long(): truncation
      int y0, y1, x0, x1;
      float wy, wx;
      dm_tap(row, sy, H, y0, y1, wy);
      dm_tap(col, sx, W, x0, x1, wx);
      // ATen's upsample_bilinear2d: w0 = 1 - w1; value = wy0*(wx0*a + wx1*b) + wy1*(wx0*c + wx1*d)
      const float top = (1.f - wx) * dp[y0 * W + x0] + wx * dp[y0 * W + x1];
      const float bot = (1.f - wx) * dp[y1 * W + x0] + wx * dp[y1 * W + x1];
      const float dv = (1.f - wy) * top + wy * bot;
      g = z;
      p = 1.f / dv;
      cnt += 1.0;
      if (MASKED) {
        const int lab = (row < mask_h && col < mask_w) ? mask[((size_t)b * mask_h + row) * mask_w + col] : 0;
        w_label[i] = lab;
        atomicAdd(&label_count[lab], 1u);                   // integer counts: order does not matter
      }
    }
    w_gt[i] = g;
    w_pd[i] = p;
  }
  const int n = (int)dm_block_sum(cnt, red);
  float* out = per_sample + b * 8;
  if (n == 0) {                                             // the reference fails on an empty selection; report NaNs
    if (threadIdx.x < 7) out[threadIdx.x] = __uint_as_float(0x7fc00000u);
    if (threadIdx.x == 7) out[7] = 0.f;
    return;
  }
  __threadfence_block();
  const int k = (n - 1) >> 1;                               // torch.median: the lower of the two middle values
  const float med_gt = dm_select(w_gt, M, k, hist, state);
  const float med_pd = dm_select(w_pd, M, k, hist, state);
  const float ratio = med_gt / med_pd;

  // the seven error sums over the kept points (label < 0) or over the kept points carrying `label`; every thread gets the totals
  auto error_sums = [&](int label, double tot[7]) {
    double s[7] = {0, 0, 0, 0, 0, 0, 0};                    // abs_rel, sq_rel, sq, sq_log, a1, a2, a3
    for (int i = threadIdx.x; i < M; i += DM_NT) {
      const float g = w_gt[i];
      if (!(g > 0.f)) continue;
      if (MASKED && label >= 0 && w_label[i] != label) continue;
      float p = w_pd[i] * ratio;
      p = p < min_depth ? min_depth : (p > max_depth ? max_depth : p);
      const float t1 = g / p, t2 = p / g;
      const float th = t1 > t2 ? t1 : t2;
      const float d = g - p;
      const float dl = logf(g) - logf(p);
      s[0] += (double)(fabsf(d) / g);
      s[1] += (double)((d * d) / g);
      s[2] += (double)(d * d);
      s[3] += (double)(dl * dl);
      s[4] += th < 1.25f ? 1.0 : 0.0;
      s[5] += th < 1.5625f ? 1.0 : 0.0;
      s[6] += th < 1.953125f ? 1.0 : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) tot[j] = dm_block_sum(s[j], red);
  };
  auto write_errors = [&](float* dst, const double tot[7], double count) {
    const double inv = 1.0 / count;
    dst[0] = (float)(tot[0] * inv);
    dst[1] = (float)(tot[1] * inv);
    dst[2] = sqrtf((float)(tot[2] * inv));
    dst[3] = sqrtf((float)(tot[3] * inv));
    dst[4] = (float)(tot[4] * inv);
    dst[5] = (float)(tot[5] * inv);
    dst[6] = (float)(tot[6] * inv);
    dst[7] = (float)count;
  };
  double tot[7];
  error_sums(-1, tot);
  if (threadIdx.x == 0) write_errors(out, tot, (double)n);
  if (MASKED) {
    for (int lab = 0; lab < 256; ++lab) {                     // uniform: only the labels that occur cost a pass
      const unsigned c = label_count[lab];
      if (c == 0) continue;
      error_sums(lab, tot);
      if (threadIdx.x == 0) write_errors(per_label + ((size_t)b * 256 + lab) * 8, tot, (double)c);
    }
  }
}

__global__ void depth_metrics_mean_kernel(const float* __restrict__ per_sample, int B, float* __restrict__ mean) {
  const int j = threadIdx.x;
  if (j >= 7) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += per_sample[b * 8 + j];   // metrics[m] += errs[i] in batch order, then / B
  mean[j] = acc / (float)B;
}

}  // namespace dd

using namespace dd;

extern "C" size_t dd_depth_metrics_workspace_bytes(int B, int M) { return (size_t)B * 2 * M * sizeof(float); }
extern "C" size_t dd_depth_metrics_masked_workspace_bytes(int B, int M) { return (size_t)B * 3 * M * sizeof(float); }

extern "C" int dd_depth_metrics(const float* disp, int B, int H, int W, const float* lidar, const float* valid, int M, const int* gt_dim,
                                const double* img_bound, float min_depth, float max_depth, float* per_sample, float* mean,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!disp || !lidar || !valid || !gt_dim || !img_bound || !per_sample || !mean || !workspace || B < 1 || H < 1 || W < 1 || M < 1)
    return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_depth_metrics_workspace_bytes(B, M)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(depth_metrics_kernel<false>, dim3(B), dim3(DM_NT), 0, s, disp, H, W, lidar, valid, M, gt_dim, img_bound[0], img_bound[1],
                     img_bound[2], img_bound[3], min_depth, max_depth, per_sample, static_cast<float*>(workspace), (const uint8_t*)nullptr, 0, 0,
                     (float*)nullptr);
  hipLaunchKernelGGL(depth_metrics_mean_kernel, dim3(1), dim3(64), 0, s, per_sample, B, mean);
  return (int)hipGetLastError();
}

extern "C" int dd_depth_metrics_masked(const float* disp, int B, int H, int W, const float* lidar, const float* valid, int M, const int* gt_dim,
                                       const double* img_bound, float min_depth, float max_depth, const uint8_t* mask, int mask_h, int mask_w,
                                       float* per_sample, float* mean, float* per_label, void* workspace, size_t workspace_bytes, void* stream) {
  if (!disp || !lidar || !valid || !gt_dim || !img_bound || !per_sample || !mean || !mask || !per_label || !workspace || B < 1 || H < 1 || W < 1 ||
      M < 1 || mask_h < 1 || mask_w < 1)
    return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_depth_metrics_masked_workspace_bytes(B, M)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(depth_metrics_kernel<true>, dim3(B), dim3(DM_NT), 0, s, disp, H, W, lidar, valid, M, gt_dim, img_bound[0], img_bound[1],
                     img_bound[2], img_bound[3], min_depth, max_depth, per_sample, static_cast<float*>(workspace), mask, mask_h, mask_w, per_label);
  hipLaunchKernelGGL(depth_metrics_mean_kernel, dim3(1), dim3(64), 0, s, per_sample, B, mean);
  return (int)hipGetLastError();
}
