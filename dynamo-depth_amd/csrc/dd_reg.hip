// dd_reg.hip -- regularisers of Trainer.compute_losses on the low-res network outputs (gfx950):
//   dd_smooth_loss    edge-aware smoothness, value + gradient in one pass   (tools.py:311-326, Trainer.py:355-359,380-381,401-402)
//   dd_sparsity_loss  masked BCE-with-logits of the motion probability       (Trainer.py:393-399)
//   dd_ground_loss    RANSAC ground plane + above-ground hinge               (tools.py:76-164, Trainer.py:361-364,425-461)
// All three are HBM-bound streaming kernels over (B,C,h,w) tensors: coalesced row-major reads, neighbours
// come from L1/L2, per-block partial sums go through wave64 shuffles and are folded in a fixed order
// (no float atomics on the loss values).  No host synchronisation anywhere -- the reference needs three
// (tools.py:125-127,137; Trainer.py:398-399).
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_math.h"

namespace dd {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of up to NV values per thread; result valid in thread 0..NV-1 (value k in thread k)
template <int NV, int NTHREADS>
__device__ __forceinline__ float block_sum(float (&v)[NV], float* red /* NV * NTHREADS/64 */) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float r = wsum(v[k]);
    if (lane == 0) red[wave * NV + k] = r;
  }
  __syncthreads();
  float out = 0.f;
  if (threadIdx.x < NV) {
#pragma unroll
    for (int wv = 0; wv < NTHREADS / 64; ++wv) out += red[wv * NV + threadIdx.x];
  }
  __syncthreads();
  return out;
}

// =================================================================================================
// smoothness
// =================================================================================================
constexpr int SM_NT = 256;

// per-image mean of a (B,1,h,w) tensor (Trainer.py:358), two-level so that the whole chip takes part:
// MEAN_BPI blocks per image write partial sums; consumers fold the MEAN_BPI partials in a fixed order.
constexpr int MEAN_BPI = 32;

__device__ __forceinline__ void plane_sum_body(int bx, int by, int gx, const float* __restrict__ x, int n, float* __restrict__ partial) {
  __shared__ float red[SM_NT / 64];
  const float* p = x + (size_t)by * n;
  float v[1] = {0.f};
  for (int i = bx * SM_NT + threadIdx.x; i < n; i += MEAN_BPI * SM_NT) v[0] += p[i];
  const float s = block_sum<1, SM_NT>(v, red);
  if (threadIdx.x == 0) partial[by * MEAN_BPI + bx] = s;
}

__global__ __launch_bounds__(SM_NT) void plane_sum_kernel(const float* __restrict__ x, int n, float* __restrict__ partial) {
  plane_sum_body(blockIdx.x, blockIdx.y, gridDim.x, x, n, partial);
}

__device__ __forceinline__ float plane_mean(const float* __restrict__ partial, int b, int n) {
  float s = 0.f;
#pragma unroll 8
  for (int i = 0; i < MEAN_BPI; ++i) s += partial[b * MEAN_BPI + i];
  return s / static_cast<float>(n);
}

// one thread per element of inp; writes the gradient w.r.t. the (normalised) input and per-block partials
template <bool HAS_IMG, bool NORMALISE>
__device__ __forceinline__ void smooth_body(int bx, int by, int gx, const float* __restrict__ inp, const float* __restrict__ img, int C,
                                                        int h, int w, const float* __restrict__ mean, float wx_scale,
                                                        float wy_scale, float* __restrict__ g_inp,
                                                        float* __restrict__ partials) {
  __shared__ float red[3 * SM_NT / 64];
  const int n = h * w;
  const int bc = by;                 // b*C + c
  const int b = bc / C;
  const int p = bx * SM_NT + threadIdx.x;
  float acc[3] = {0.f, 0.f, 0.f};            // sum_x, sum_y, sum g_a*d (normalised case)
  if (p < n) {
    const int y = p / w, x = p % w;
    const float* a = inp + (size_t)bc * n;
    const float* im = HAS_IMG ? img + (size_t)b * 3 * n : nullptr;
    float inv = 1.f;
    if (NORMALISE) inv = 1.f / (plane_mean(mean, b, n) + 1e-7f);
    const float ac = a[p] * inv;
    auto edge_w = [&](int q0, int q1) -> float {
      if (!HAS_IMG) return 1.f;
      const float d = dd_abs(im[q0] - im[q1]) + dd_abs(im[n + q0] - im[n + q1]) + dd_abs(im[2 * n + q0] - im[2 * n + q1]);
      return __expf(-d / 3.f);
    };
    float g = 0.f;
    if (x + 1 < w) {           // term owned by this pixel: |a[p] - a[p+1]| * wx[p]
      const float d = ac - a[p + 1] * inv, e = edge_w(p, p + 1);
      acc[0] += dd_abs(d) * e;
      g += dd_sign(d) * e * wx_scale;
    }
    if (x > 0) {               // term owned by the left neighbour
      const float d = a[p - 1] * inv - ac, e = edge_w(p - 1, p);
      g -= dd_sign(d) * e * wx_scale;
    }
    if (y + 1 < h) {
      const float d = ac - a[p + w] * inv, e = edge_w(p, p + w);
      acc[1] += dd_abs(d) * e;
      g += dd_sign(d) * e * wy_scale;
    }
    if (y > 0) {
      const float d = a[p - w] * inv - ac, e = edge_w(p - w, p);
      g -= dd_sign(d) * e * wy_scale;
    }
    // NORMALISE: g_out is a temporary holding d/d(normalised input); otherwise it is the caller's accumulator
    if (g_inp) {
      if (NORMALISE) g_inp[(size_t)bc * n + p] = g;
      else g_inp[(size_t)bc * n + p] += g;
    }
    if (NORMALISE) acc[2] = g * a[p];
  }
  const float r = block_sum<3, SM_NT>(acc, red);
  if (threadIdx.x < 3) partials[((size_t)bc * gx + bx) * 4 + threadIdx.x] = r;
}

template <bool HAS_IMG, bool NORMALISE>
__global__ __launch_bounds__(SM_NT) void smooth_kernel(const float* __restrict__ inp, const float* __restrict__ img, int C,
                                                        int h, int w, const float* __restrict__ mean, float wx_scale,
                                                        float wy_scale, float* __restrict__ g_inp,
                                                        float* __restrict__ partials) {
  smooth_body<HAS_IMG, NORMALISE>(blockIdx.x, blockIdx.y, gridDim.x, inp, img, C, h, w, mean, wx_scale, wy_scale, g_inp, partials);
}

// folds the partials into sums[0..1]; with NORMALISE also turns d/d(normalised) into d/d(disp):
//   d = a*(m+eps)  ->  g_d = g_a/(m+eps) - (sum_p g_a[p] d[p]) / ((m+eps)^2 * n)
template <bool NORMALISE>
__device__ __forceinline__ void smooth_finish_body(int bx, int by, int gx, const float* __restrict__ partials, int nblk, int BC, int n,
                                                               const float* __restrict__ mean, const float* __restrict__ g_tmp,
                                                               float* __restrict__ g_inp, float* __restrict__ sums) {
  __shared__ float red[2 * SM_NT / 64];
  __shared__ float dot_s;
  const int bc = by;
  if (bx == 0 && bc == 0) {
    float v[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < BC * nblk; i += SM_NT) { v[0] += partials[(size_t)i * 4]; v[1] += partials[(size_t)i * 4 + 1]; }
    const float r = block_sum<2, SM_NT>(v, red);
    if (threadIdx.x < 2) sums[threadIdx.x] = r;
  }
  if (NORMALISE && g_inp) {
    float v[1] = {0.f};
    for (int i = threadIdx.x; i < nblk; i += SM_NT) v[0] += partials[((size_t)bc * nblk + i) * 4 + 2];
    const float r = block_sum<1, SM_NT>(v, red);
    if (threadIdx.x == 0) dot_s = r;
    __syncthreads();
    const int p = bx * SM_NT + threadIdx.x;
    if (p < n) {
      const float me = plane_mean(mean, bc, n) + 1e-7f;     // C == 1 in the normalised case: bc == b
      const size_t i = (size_t)bc * n + p;
      g_inp[i] += g_tmp[i] / me - dot_s / (me * me * static_cast<float>(n));
    }
  }
}

template <bool NORMALISE>
__global__ __launch_bounds__(SM_NT) void smooth_finish_kernel(const float* __restrict__ partials, int nblk, int BC, int n,
                                                               const float* __restrict__ mean, const float* __restrict__ g_tmp,
                                                               float* __restrict__ g_inp, float* __restrict__ sums) {
  smooth_finish_body<NORMALISE>(blockIdx.x, blockIdx.y, gridDim.x, partials, nblk, BC, n, mean, g_tmp, g_inp, sums);
}

// =================================================================================================
// sparsity
// =================================================================================================
constexpr int SP_NT = 256;
constexpr int SP_BPI = 32;     // blocks per image in the counting pass

DD_HD float softplus(float x) { return (x > 0.f ? x : 0.f) + log1pf(expf(-dd_abs(x))); }

__device__ __forceinline__ void sparsity_count_body(int bx, int by, int gx, const float* __restrict__ delta, const float* __restrict__ delta_sum,
                                                                const float* __restrict__ prob, int n, float inv_total,
                                                                float* __restrict__ partials) {
  __shared__ float red[2 * SP_NT / 64];
  const int b = by;
  const float thr = delta_sum[0] * inv_total;            // disp_mag.mean() over the whole batch (Trainer.py:397)
  float v[2] = {0.f, 0.f};
  for (int p = bx * SP_NT + threadIdx.x; p < n; p += SP_BPI * SP_NT) {
    const size_t i = (size_t)b * n + p;
    if (delta[i] < thr) { v[0] += 1.f; v[1] += softplus(prob[i]); }
  }
  const float r = block_sum<2, SP_NT>(v, red);
  if (threadIdx.x < 2) partials[((size_t)b * SP_BPI + bx) * 2 + threadIdx.x] = r;
}

__global__ __launch_bounds__(SP_NT) void sparsity_count_kernel(const float* __restrict__ delta, const float* __restrict__ delta_sum,
                                                                const float* __restrict__ prob, int n, float inv_total,
                                                                float* __restrict__ partials) {
  sparsity_count_body(blockIdx.x, blockIdx.y, gridDim.x, delta, delta_sum, prob, n, inv_total, partials);
}

__device__ __forceinline__ void sparsity_grad_body(int bx, int by, int gx, const float* __restrict__ delta, const float* __restrict__ delta_sum,
                                                               const float* __restrict__ prob, int B, int n, float inv_total,
                                                               float weight, const float* __restrict__ partials,
                                                               float* __restrict__ g_prob, float* __restrict__ out) {
  __shared__ float s_cnt, s_sum;
  __shared__ int s_gate;
  if (threadIdx.x < 64) {
    // one wave folds the B*SP_BPI records in a fixed order; an image with zero static pixels closes the gate
    float cnt = 0.f, sm = 0.f;
    int gate = 1;
    for (int b = 0; b < B; ++b) {
      float c = 0.f, s2 = 0.f;
      for (int i = threadIdx.x; i < SP_BPI; i += 64) { c += partials[((size_t)b * SP_BPI + i) * 2]; s2 += partials[((size_t)b * SP_BPI + i) * 2 + 1]; }
      c = wsum(c); s2 = wsum(s2);
      if (c <= 0.f) gate = 0;
      cnt += c; sm += s2;
    }
    if (threadIdx.x == 0) { s_cnt = cnt; s_sum = sm; s_gate = gate; }
  }
  __syncthreads();
  const float cnt = s_cnt;
  const bool gate = s_gate != 0;
  if (bx == 0 && by == 0 && threadIdx.x == 0) {
    out[0] = gate ? s_sum / cnt : 0.f;
    out[1] = cnt;
  }
  if (!gate || !g_prob) return;
  const float thr = delta_sum[0] * inv_total;
  const int b = by;
  const int p = bx * SP_NT + threadIdx.x;
  if (p < n) {
    const size_t i = (size_t)b * n + p;
    if (delta[i] < thr) {
      const float x = prob[i];
      g_prob[i] += weight / cnt * (1.f / (1.f + expf(-x)));      // d softplus = sigmoid
    }
  }
}

__global__ __launch_bounds__(SP_NT) void sparsity_grad_kernel(const float* __restrict__ delta, const float* __restrict__ delta_sum,
                                                               const float* __restrict__ prob, int B, int n, float inv_total,
                                                               float weight, const float* __restrict__ partials,
                                                               float* __restrict__ g_prob, float* __restrict__ out) {
  sparsity_grad_body(blockIdx.x, blockIdx.y, gridDim.x, delta, delta_sum, prob, B, n, inv_total, weight, partials, g_prob, out);
}

// =================================================================================================
// ground plane
// =================================================================================================
constexpr int GP_NT = 256;
constexpr int GP_MAX_IT = 128;

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

// one thread per RANSAC candidate: least squares y = w1*x + w2*z + w3 through np points (tools.py:141-154),
// (AtA + 1e-6 on EVERY entry)^-1 At B, solved in double to stay clear of the conditioning of 5 nearby points
__device__ __forceinline__ void ground_candidates_body(int bx, int by, int gx, const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                                   const int32_t* __restrict__ rand_idx, int B, int h, int w,
                                                                   int rows, int np, int max_it, DepthParams dp,
                                                                   float* __restrict__ cand /* (B*max_it,3) */,
                                                                   int* __restrict__ counts /* (B*max_it), zeroed here */) {
  const int j = bx * GP_NT + threadIdx.x;
  if (j >= B * max_it) return;
  counts[j] = 0;
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
  for (int i = 0; i < 3; ++i) cand[(size_t)j * 3 + i] = static_cast<float>(inv[i][0] * r[0] + inv[i][1] * r[1] + inv[i][2] * r[2]);
}

__global__ __launch_bounds__(GP_NT) void ground_candidates_kernel(const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                                   const int32_t* __restrict__ rand_idx, int B, int h, int w,
                                                                   int rows, int np, int max_it, DepthParams dp,
                                                                   float* __restrict__ cand /* (B*max_it,3) */,
                                                                   int* __restrict__ counts /* (B*max_it), zeroed here */) {
  ground_candidates_body(blockIdx.x, blockIdx.y, gridDim.x, disp, inv_K, rand_idx, B, h, w, rows, np, max_it, dp, cand, counts);
}

// scores every candidate against the ground points of ONE image.  The reference pairs candidate
// j = b*max_it + it with the points of image (j mod B) -- `points.repeat(max_it,1,1)` at tools.py:130 tiles
// the batch while the candidates are image-major -- and that pairing is reproduced here.
__device__ __forceinline__ void ground_score_body(int bx, int by, int gx, const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                              const float* __restrict__ cand, int B, int h, int w, int rows,
                                                              int max_it, float tol, DepthParams dp,
                                                              int* __restrict__ counts /* (B*max_it) zeroed */) {
  __shared__ float s_c[GP_MAX_IT * 3];
  __shared__ int s_j[GP_MAX_IT];
  __shared__ int s_cnt[GP_MAX_IT];
  const int img = by;
  const int n = h * w, base = (h - rows) * w, ng = rows * w;
  // the candidates scored on image `img`: all j in [0, B*max_it) with j % B == img  (exactly max_it of them)
  for (int k = threadIdx.x; k < max_it; k += GP_NT) {
    const int j = img + k * B;
    s_j[k] = j;
    s_c[k * 3 + 0] = cand[(size_t)j * 3 + 0];
    s_c[k * 3 + 1] = cand[(size_t)j * 3 + 1];
    s_c[k * 3 + 2] = cand[(size_t)j * 3 + 2];
    s_cnt[k] = 0;
  }
  __syncthreads();
  const int q = bx * GP_NT + threadIdx.x;
  float P[3] = {0.f, 0.f, 0.f};
  const bool live = q < ng;
  if (live) ground_point(disp + (size_t)img * n * (inv_K ? 1 : 3), inv_K ? inv_K + img * 16 : nullptr, dp, w, base + q, P, n);
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < max_it; ++k) {
    const float dist = P[0] * s_c[k * 3 + 0] + P[2] * s_c[k * 3 + 1] + s_c[k * 3 + 2] - P[1];
    const bool in = live && (dd_abs(dist) < tol);
    const unsigned long long m = __ballot(in);
    if (lane == 0 && m) atomicAdd(&s_cnt[k], __popcll(m));
  }
  __syncthreads();
  for (int k = threadIdx.x; k < max_it; k += GP_NT)
    if (s_cnt[k]) atomicAdd(&counts[s_j[k]], s_cnt[k]);
}

__global__ __launch_bounds__(GP_NT) void ground_score_kernel(const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                              const float* __restrict__ cand, int B, int h, int w, int rows,
                                                              int max_it, float tol, DepthParams dp,
                                                              int* __restrict__ counts /* (B*max_it) zeroed */) {
  ground_score_body(blockIdx.x, blockIdx.y, gridDim.x, disp, inv_K, cand, B, h, w, rows, max_it, tol, dp, counts);
}

// picks the best candidate per image (first maximum, like argmax), evaluates the hinge and its gradient
__device__ __forceinline__ void ground_hinge_body(int bx, int by, int gx, const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                              const float* __restrict__ cand, const int* __restrict__ counts,
                                                              int h, int w, int max_it, float tol, float max_depth,
                                                              DepthParams dp, float weight, float* __restrict__ g_disp,
                                                              float* __restrict__ plane, float* __restrict__ partials) {
  __shared__ float red[GP_NT / 64];
  __shared__ float s_w[3];
  const int b = by, n = h * w;
  if (threadIdx.x == 0) {
    int best = 0, bc = counts[b * max_it];
    for (int k = 1; k < max_it; ++k) {
      const int c = counts[b * max_it + k];
      if (c > bc) { bc = c; best = k; }
    }
    for (int i = 0; i < 3; ++i) s_w[i] = cand[((size_t)b * max_it + best) * 3 + i];
    if (bx == 0)
      for (int i = 0; i < 3; ++i) plane[b * 3 + i] = s_w[i];
  }
  __syncthreads();
  const float w1 = s_w[0], w2 = s_w[1], w3 = s_w[2] + tol;      // Trainer.py:437-438
  const int p = bx * GP_NT + threadIdx.x;
  float v[1] = {0.f};
  if (p < n) {
    const int y = p / w, x = p % w;
    const float* A = inv_K + b * 16;
    float ray[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) ray[i] = A[i * 4 + 0] * static_cast<float>(x) + A[i * 4 + 1] * static_cast<float>(y) + A[i * 4 + 2];
    float gd = w3 / (ray[1] - ray[0] * w1 - ray[2] * w2);
    const bool invalid = (gd < 0.f) || (gd > max_depth);        // NaN compares false -> stays, like the reference
    if (invalid) gd = max_depth;
    if (gd != max_depth) {
      const float gdisp = (1.f / gd - dp.lo) / dp.span;
      const float diff = disp[(size_t)b * n + p] - gdisp;
      if (!(diff > 0.f)) {                                       // disp_diff[disp_diff > 0] = 0
        v[0] = diff;
        if (g_disp) g_disp[(size_t)b * n + p] += weight;
      }
    }
  }
  const float r = block_sum<1, GP_NT>(v, red);
  if (threadIdx.x == 0) partials[(size_t)b * gx + bx] = r;
}

__global__ __launch_bounds__(GP_NT) void ground_hinge_kernel(const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                              const float* __restrict__ cand, const int* __restrict__ counts,
                                                              int h, int w, int max_it, float tol, float max_depth,
                                                              DepthParams dp, float weight, float* __restrict__ g_disp,
                                                              float* __restrict__ plane, float* __restrict__ partials) {
  ground_hinge_body(blockIdx.x, blockIdx.y, gridDim.x, disp, inv_K, cand, counts, h, w, max_it, tol, max_depth, dp, weight, g_disp, plane, partials);
}

// tools.GroundPlane.forward: vertical distance of every point to the best plane (tools.py:96-101,103-111)
__global__ __launch_bounds__(GP_NT) void ground_dist_kernel(const float* __restrict__ points, const float* __restrict__ cand,
                                                             const int* __restrict__ counts, int n, int max_it,
                                                             float* __restrict__ dist, float* __restrict__ plane) {
  __shared__ float s_w[3];
  const int b = blockIdx.y;
  if (threadIdx.x == 0) {
    int best = 0, bc = counts[b * max_it];
    for (int k = 1; k < max_it; ++k) {
      const int c = counts[b * max_it + k];
      if (c > bc) { bc = c; best = k; }
    }
    for (int i = 0; i < 3; ++i) s_w[i] = cand[((size_t)b * max_it + best) * 3 + i];
    if (blockIdx.x == 0)
      for (int i = 0; i < 3; ++i) plane[b * 3 + i] = s_w[i];
  }
  __syncthreads();
  const int p = blockIdx.x * GP_NT + threadIdx.x;
  if (p < n) {
    const float* P = points + (size_t)b * 3 * n;
    dist[(size_t)b * n + p] = P[p] * s_w[0] + P[2 * n + p] * s_w[1] + s_w[2] - P[n + p];
  }
}

__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ partials, int count, float* __restrict__ out) {
  __shared__ float red[4];
  float v[1] = {0.f};
  for (int i = threadIdx.x; i < count; i += 256) v[0] += partials[i];
  const float r = block_sum<1, 256>(v, red);
  if (threadIdx.x == 0) out[0] = r;
}

// one wave: folds the raw sums into the losses dict values (see dd_assemble_losses in dynamo_hip.h)
__global__ __launch_bounds__(64) void assemble_kernel(const float* __restrict__ res, const DDAssembleArgs a, float* __restrict__ loss,
                                                      float* __restrict__ out) {
  __shared__ float term[DD_MAX_SCALES][DD_NUM_TERMS];
  const int t = threadIdx.x;
  if (t < DD_MAX_SCALES * DD_NUM_TERMS) {
    const int s = t / DD_NUM_TERMS, k = t % DD_NUM_TERMS;
    float acc = 0.f;
    for (int i = 0; i < a.n; ++i)
      if (a.term_of[i] == k && a.scale_of[i] == s) acc += a.norm[i] * res[i];
    term[s][k] = acc;
  }
  __syncthreads();
  if (t < DD_NUM_TERMS) {
    float acc = 0.f;
    for (int s = 0; s < a.num_scales; ++s) acc += term[s][t];
    out[1 + t] = acc;
  }
  if (t == 32) {
    float total = 0.f;
    for (int s = 0; s < a.num_scales; ++s) {
      float acc = 0.f;
      for (int k = 0; k < DD_NUM_TERMS; ++k) acc += a.coef[k] * term[s][k];
      out[1 + DD_NUM_TERMS + s] = acc;
      total += acc / static_cast<float>(a.num_scales);
    }
    out[0] = total;
    loss[0] = total;
  }
}

// =================================================================================================
// all regularisers of all scales in three or four launches (dd_reg_losses)
// =================================================================================================
// The per-term kernels above run as TASKS of three stage kernels: a task owns a contiguous range of workgroups of the launch
// and maps it onto the 2-D grid the stand-alone kernel would have had.  Stage 1: per-image disparity means, static-pixel
// counts, RANSAC candidates.  Stage 2 (needs stage 1): smoothness value + gradient, sparsity gradient, candidate scoring.
// Stage 3 (needs stage 2): mean-normalisation adjoint + smoothness sums.  Stage 4: ground hinge (+ its fixed-order fold by the
// last workgroup to finish) -- after stage 3 because both add to the disparity gradient.  Same bodies, same reduction orders,
// same results as the per-term entry points.
constexpr int RT_NT = 256;
static_assert(SM_NT == RT_NT && SP_NT == RT_NT && GP_NT == RT_NT, "one workgroup size for every task");
enum : int { K_MEAN = 0, K_SPCOUNT, K_GCAND, K_SMOOTH, K_SPGRAD, K_GSCORE, K_SMFIN, K_GHINGE };
constexpr int REG_MAX_TASKS = 32;

struct RegTask {
  int first;            // first workgroup of the task
  int gx;               // width of its virtual grid
  short kind;
  signed char scale, idx;
};

struct RegTasks {
  int n;
  RegTask t[REG_MAX_TASKS];
};

struct RegOffsets {     // float offsets into DDRegArgs.workspace
  long long mean[DD_MAX_SCALES];
  long long sm_part[DD_MAX_SCALES][DD_REG_SMOOTH];
  long long sm_gtmp[DD_MAX_SCALES][DD_REG_SMOOTH];
  long long sp_part[DD_MAX_SCALES][DD_NUM_SRC];
  long long g_cand[DD_MAX_SCALES], g_counts[DD_MAX_SCALES], g_part[DD_MAX_SCALES], g_done[DD_MAX_SCALES];
};

__global__ __launch_bounds__(RT_NT) void reg_stage_kernel(const DDRegArgs a, const RegOffsets off, const RegTasks tasks) {
  int ti = 0;
  for (int i = 1; i < tasks.n; ++i)
    if ((int)blockIdx.x >= tasks.t[i].first) ti = i;
  const RegTask t = tasks.t[ti];
  const int vb = (int)blockIdx.x - t.first, gx = t.gx, bx = vb % gx, by = vb / gx;
  const int s = t.scale, k = t.idx;
  const DDRegScale& sc = a.scale[s];
  const int B = a.B, h = sc.h, w = sc.w, n = h * w;
  const int nblk = (n + RT_NT - 1) / RT_NT;
  float* ws = a.workspace;
  float* res = a.res + s * DD_REG_RES_STRIDE;
  const float inv_total = 1.f / (static_cast<float>(B) * static_cast<float>(n));
  const int rows = static_cast<int>(a.g_prior * static_cast<float>(h));
  const DepthParams dp = depth_params(a.min_depth, a.max_depth);
  switch (t.kind) {
    case K_MEAN:
      plane_sum_body(bx, by, gx, sc.smooth[k].inp, n, ws + off.mean[s]);
      break;
    case K_SPCOUNT:
      sparsity_count_body(bx, by, gx, sc.delta[k], sc.delta_sum[k], sc.prob[k], n, inv_total, ws + off.sp_part[s][k]);
      break;
    case K_GCAND:
      if (bx == 0 && threadIdx.x == 0) *reinterpret_cast<int*>(ws + off.g_done[s]) = 0;
      ground_candidates_body(bx, by, gx, sc.disp, sc.inv_K, sc.rand_idx, B, h, w, rows, a.np_per_it, a.max_it, dp, ws + off.g_cand[s],
                             reinterpret_cast<int*>(ws + off.g_counts[s]));
      break;
    case K_SMOOTH: {
      const DDRegSmooth& sm = sc.smooth[k];
      const float cnt = static_cast<float>(B) * static_cast<float>(sm.C);
      const float wx = sm.weight / (cnt * h * (w - 1)), wy = sm.weight / (cnt * (h - 1) * w);
      if (sm.normalise)
        smooth_body<true, true>(bx, by, gx, sm.inp, sc.img, sm.C, h, w, ws + off.mean[s], wx, wy, sm.g_inp ? ws + off.sm_gtmp[s][k] : nullptr,
                                ws + off.sm_part[s][k]);
      else
        smooth_body<true, false>(bx, by, gx, sm.inp, sc.img, sm.C, h, w, nullptr, wx, wy, sm.g_inp, ws + off.sm_part[s][k]);
      break;
    }
    case K_SPGRAD:
      // k == 2: both frames read one motion_prob tensor and accumulate into one gradient buffer (what networks.Model
      // publishes) -- the same thread then handles the element for frame 0 and frame 1 in turn; two tasks would race
      for (int f = (k == 2 ? 0 : k); f <= (k == 2 ? 1 : k); ++f)
        sparsity_grad_body(bx, by, gx, sc.delta[f], sc.delta_sum[f], sc.prob[f], B, n, inv_total, sc.w_sparsity[f], ws + off.sp_part[s][f],
                           sc.g_prob[f], res + 10 + 2 * f);
      break;
    case K_GSCORE:
      ground_score_body(bx, by, gx, sc.disp, sc.inv_K, ws + off.g_cand[s], B, h, w, rows, a.max_it, a.tol, dp,
                        reinterpret_cast<int*>(ws + off.g_counts[s]));
      break;
    case K_SMFIN: {
      const DDRegSmooth& sm = sc.smooth[k];
      if (sm.normalise)
        smooth_finish_body<true>(bx, by, gx, ws + off.sm_part[s][k], nblk, B * sm.C, n, ws + off.mean[s], ws + off.sm_gtmp[s][k], sm.g_inp, res + 2 * k);
      else
        smooth_finish_body<false>(bx, by, gx, ws + off.sm_part[s][k], nblk, B * sm.C, n, nullptr, nullptr, sm.g_inp, res + 2 * k);
      break;
    }
    case K_GHINGE: {
      float* partials = ws + off.g_part[s];
      ground_hinge_body(bx, by, gx, sc.disp, sc.inv_K, ws + off.g_cand[s], reinterpret_cast<const int*>(ws + off.g_counts[s]), h, w, a.max_it,
                        a.tol, a.max_depth, dp, sc.w_ground, sc.g_disp, sc.plane, partials);
      // the last workgroup to arrive folds all partials in a fixed order: the sum does not depend on which one it is
      __shared__ int s_last;
      __shared__ float red[RT_NT / 64];
      if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(reinterpret_cast<int*>(ws + off.g_done[s]), 1) == gx * B - 1;
      }
      __syncthreads();
      if (s_last) {
        __threadfence();
        float v[1] = {0.f};
        for (int i = threadIdx.x; i < gx * B; i += RT_NT) v[0] += *reinterpret_cast<volatile float*>(partials + i);     // written by other workgroups of this launch
        const float r = block_sum<1, RT_NT>(v, red);
        if (threadIdx.x == 0) res[14] = r;
      }
      break;
    }
    default:
      break;
  }
}

constexpr int REG_STAGES = 4;
struct RegPlan {
  RegOffsets off;
  RegTasks stage[REG_STAGES];
  int blocks[REG_STAGES];
  size_t floats;
};

static int reg_plan(const DDRegArgs& a, RegPlan& p) {
  size_t total = 0;
  auto take = [&](size_t nfloats) { const size_t o = total; total += (nfloats + 63) / 64 * 64; return (long long)o; };
  for (int st = 0; st < REG_STAGES; ++st) { p.stage[st].n = 0; p.blocks[st] = 0; }
  auto add = [&](int st, int kind, int s, int idx, int gx, int gy) -> int {
    RegTasks& T = p.stage[st];
    if (T.n >= REG_MAX_TASKS) return 1;
    RegTask& t = T.t[T.n++];
    t.first = p.blocks[st]; t.gx = gx; t.kind = (short)kind; t.scale = (signed char)s; t.idx = (signed char)idx;
    p.blocks[st] += gx * gy;
    return 0;
  };
  int bad = 0;
  for (int s = 0; s < a.num_scales; ++s) {
    const DDRegScale& sc = a.scale[s];
    if (sc.h < 2 || sc.w < 2) return 1;
    const int n = sc.h * sc.w, nblk = (n + RT_NT - 1) / RT_NT;
    int normalised = -1;
    for (int k = 0; k < DD_REG_SMOOTH; ++k) {
      const DDRegSmooth& sm = sc.smooth[k];
      if (!sm.inp) continue;
      if (!sc.img || sm.C < 1 || (sm.normalise && sm.C != 1)) return 1;
      if (sm.normalise) {
        if (normalised >= 0) return 1;                    // one mean buffer per scale
        normalised = k;
        p.off.mean[s] = take((size_t)a.B * MEAN_BPI);
        p.off.sm_gtmp[s][k] = take((size_t)a.B * n);
        bad |= add(0, K_MEAN, s, k, MEAN_BPI, a.B);
      }
      p.off.sm_part[s][k] = take((size_t)a.B * sm.C * nblk * 4);
      bad |= add(1, K_SMOOTH, s, k, nblk, a.B * sm.C);
      bad |= sm.normalise ? add(2, K_SMFIN, s, k, nblk, a.B * sm.C) : add(2, K_SMFIN, s, k, 1, 1);
    }
    const bool shared_prob = sc.prob[0] && sc.prob[0] == sc.prob[1];
    for (int f = 0; f < DD_NUM_SRC; ++f) {
      if (!sc.prob[f]) continue;
      if (!sc.delta[f] || !sc.delta_sum[f]) return 1;
      p.off.sp_part[s][f] = take((size_t)a.B * SP_BPI * 2);
      bad |= add(0, K_SPCOUNT, s, f, SP_BPI, a.B);
      if (!shared_prob) bad |= add(1, K_SPGRAD, s, f, nblk, a.B);
    }
    if (shared_prob) bad |= add(1, K_SPGRAD, s, 2, nblk, a.B);
    if (sc.disp) {
      if (!sc.inv_K || !sc.rand_idx || !sc.plane || a.max_it < 1 || a.max_it > GP_MAX_IT) return 1;
      const int rows = (int)(a.g_prior * (float)sc.h);
      if (rows < 1) return 1;
      p.off.g_cand[s] = take((size_t)a.B * a.max_it * 3);
      p.off.g_counts[s] = take((size_t)a.B * a.max_it);
      p.off.g_part[s] = take((size_t)a.B * nblk);
      p.off.g_done[s] = take(1);
      bad |= add(0, K_GCAND, s, 0, (a.B * a.max_it + RT_NT - 1) / RT_NT, 1);
      bad |= add(1, K_GSCORE, s, 0, (rows * sc.w + RT_NT - 1) / RT_NT, a.B);
      bad |= add(3, K_GHINGE, s, 0, nblk, a.B);
    }
  }
  p.floats = total;
  return bad;
}

}  // namespace dd

using namespace dd;

static inline int last_error() { return (int)hipGetLastError(); }

// ------------------------------------------------------------------------------------------------
extern "C" size_t dd_smooth_workspace_bytes(int B, int C, int h, int w) {
  const size_t nblk = ((size_t)h * w + SM_NT - 1) / SM_NT;
  return ((size_t)B * C * nblk * 4 + (size_t)B * MEAN_BPI + (size_t)B * C * h * w) * sizeof(float);
}

extern "C" int dd_smooth_loss(const float* inp, const float* img, int B, int C, int h, int w, int normalise, float weight,
                              float* g_inp, float* sums, float* workspace, void* stream_) {
  if (!inp || !sums || !workspace || B < 1 || C < 1 || h < 2 || w < 2) return (int)hipErrorInvalidValue;
  if (normalise && C != 1) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int n = h * w, nblk = (n + SM_NT - 1) / SM_NT;
  float* partials = workspace;
  float* mean = workspace + (size_t)B * C * nblk * 4;
  float* g_tmp = mean + (size_t)B * MEAN_BPI;
  const float wx = weight / ((float)B * C * h * (w - 1)), wy = weight / ((float)B * C * (h - 1) * w);
  dim3 grid(nblk, B * C);
  if (normalise) {
    hipLaunchKernelGGL(plane_sum_kernel, dim3(MEAN_BPI, B), dim3(SM_NT), 0, stream, inp, n, mean);
    if (img) hipLaunchKernelGGL((smooth_kernel<true, true>), grid, dim3(SM_NT), 0, stream, inp, img, C, h, w, mean, wx, wy, g_inp ? g_tmp : nullptr, partials);
    else hipLaunchKernelGGL((smooth_kernel<false, true>), grid, dim3(SM_NT), 0, stream, inp, img, C, h, w, mean, wx, wy, g_inp ? g_tmp : nullptr, partials);
    hipLaunchKernelGGL((smooth_finish_kernel<true>), grid, dim3(SM_NT), 0, stream, partials, nblk, B * C, n, mean, g_tmp, g_inp, sums);
  } else {
    if (img) hipLaunchKernelGGL((smooth_kernel<true, false>), grid, dim3(SM_NT), 0, stream, inp, img, C, h, w, mean, wx, wy, g_inp, partials);
    else hipLaunchKernelGGL((smooth_kernel<false, false>), grid, dim3(SM_NT), 0, stream, inp, img, C, h, w, mean, wx, wy, g_inp, partials);
    hipLaunchKernelGGL((smooth_finish_kernel<false>), dim3(1, 1), dim3(SM_NT), 0, stream, partials, nblk, B * C, n, mean, g_tmp, g_inp, sums);
  }
  return last_error();
}

// ------------------------------------------------------------------------------------------------
extern "C" size_t dd_sparsity_workspace_bytes(int B, int h, int w) {
  (void)h; (void)w;
  return (size_t)B * SP_BPI * 2 * sizeof(float);
}

extern "C" int dd_sparsity_loss(const float* delta, const float* delta_sum, const float* prob, int B, int h, int w, float weight,
                                float* g_prob, float* out, float* workspace, void* stream_) {
  if (!delta || !delta_sum || !prob || !out || !workspace || B < 1) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int n = h * w;
  const float inv_total = 1.f / ((float)B * n);
  hipLaunchKernelGGL(sparsity_count_kernel, dim3(SP_BPI, B), dim3(SP_NT), 0, stream, delta, delta_sum, prob, n, inv_total, workspace);
  hipLaunchKernelGGL(sparsity_grad_kernel, dim3((n + SP_NT - 1) / SP_NT, B), dim3(SP_NT), 0, stream, delta, delta_sum, prob, B, n,
                     inv_total, weight, workspace, g_prob, out);
  return last_error();
}

// ------------------------------------------------------------------------------------------------
extern "C" size_t dd_ground_workspace_bytes(int B, int h, int w, int max_it) {
  const size_t nblk = ((size_t)h * w + GP_NT - 1) / GP_NT;
  return ((size_t)B * max_it * 3 + (size_t)B * max_it + (size_t)B * nblk) * sizeof(float);
}

extern "C" int dd_ground_loss(const float* disp, const float* inv_K, const int32_t* rand_idx, int B, int h, int w, int np_per_it,
                              int max_it, float tol, float g_prior, float min_depth, float max_depth, float weight, float* g_disp,
                              float* plane, float* out, float* workspace, void* stream_) {
  if (!disp || !inv_K || !rand_idx || !plane || !out || !workspace || max_it < 1 || max_it > GP_MAX_IT) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int rows = (int)(g_prior * (float)h);
  if (rows < 1) return (int)hipErrorInvalidValue;
  const int n = h * w, ng = rows * w, nblk = (n + GP_NT - 1) / GP_NT;
  const DepthParams dp = depth_params(min_depth, max_depth);
  float* cand = workspace;
  int* counts = reinterpret_cast<int*>(workspace + (size_t)B * max_it * 3);
  float* partials = workspace + (size_t)B * max_it * 4;
  hipLaunchKernelGGL(ground_candidates_kernel, dim3((B * max_it + GP_NT - 1) / GP_NT), dim3(GP_NT), 0, stream, disp, inv_K, rand_idx,
                     B, h, w, rows, np_per_it, max_it, dp, cand, counts);
  hipLaunchKernelGGL(ground_score_kernel, dim3((ng + GP_NT - 1) / GP_NT, B), dim3(GP_NT), 0, stream, disp, inv_K, cand, B, h, w, rows,
                     max_it, tol, dp, counts);
  hipLaunchKernelGGL(ground_hinge_kernel, dim3(nblk, B), dim3(GP_NT), 0, stream, disp, inv_K, cand, counts, h, w, max_it, tol,
                     max_depth, dp, weight, g_disp, plane, partials);
  hipLaunchKernelGGL(fold_kernel, dim3(1), dim3(256), 0, stream, partials, B * nblk, out);
  return last_error();
}

// ------------------------------------------------------------------------------------------------
extern "C" int dd_assemble_losses(const float* res, const DDAssembleArgs* args, float* loss, float* out, void* stream_) {
  if (!res || !args || !loss || !out || args->n < 0 || args->n > DD_MAX_RES || args->num_scales < 1 || args->num_scales > DD_MAX_SCALES)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(assemble_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_), res, *args, loss, out);
  return last_error();
}

// ------------------------------------------------------------------------------------------------
extern "C" int dd_ground_plane(const float* points, const int32_t* rand_idx, int B, int h, int w, int np_per_it, int max_it,
                               float tol, float g_prior, float* dist, float* plane, float* workspace, void* stream_) {
  if (!points || !rand_idx || !dist || !plane || !workspace || max_it < 1 || max_it > GP_MAX_IT) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int rows = (int)(g_prior * (float)h);
  if (rows < 1) return (int)hipErrorInvalidValue;
  const int n = h * w, ng = rows * w;
  const DepthParams dp = {0.f, 0.f};
  float* cand = workspace;
  int* counts = reinterpret_cast<int*>(workspace + (size_t)B * max_it * 3);
  hipLaunchKernelGGL(ground_candidates_kernel, dim3((B * max_it + GP_NT - 1) / GP_NT), dim3(GP_NT), 0, stream, points, (const float*)nullptr,
                     rand_idx, B, h, w, rows, np_per_it, max_it, dp, cand, counts);
  hipLaunchKernelGGL(ground_score_kernel, dim3((ng + GP_NT - 1) / GP_NT, B), dim3(GP_NT), 0, stream, points, (const float*)nullptr, cand, B,
                     h, w, rows, max_it, tol, dp, counts);
  hipLaunchKernelGGL(ground_dist_kernel, dim3((n + GP_NT - 1) / GP_NT, B), dim3(GP_NT), 0, stream, points, cand, counts, n, max_it, dist,
                     plane);
  return last_error();
}

// ------------------------------------------------------------------------------------------------
extern "C" size_t dd_reg_workspace_bytes(const DDRegArgs* a) {
  RegPlan p;
  if (!a || a->num_scales < 1 || a->num_scales > DD_MAX_SCALES || reg_plan(*a, p)) return 0;
  return (p.floats > 0 ? p.floats : 1) * sizeof(float);
}

extern "C" int dd_reg_losses(const DDRegArgs* a, void* stream_) {
  if (!a || a->abi_version != DD_ABI_VERSION || a->B < 1 || a->num_scales < 1 || a->num_scales > DD_MAX_SCALES || !a->res || !a->workspace)
    return (int)hipErrorInvalidValue;
  RegPlan p;
  if (reg_plan(*a, p)) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  for (int st = 0; st < REG_STAGES; ++st) {
    if (p.blocks[st] == 0) continue;
    hipLaunchKernelGGL(reg_stage_kernel, dim3(p.blocks[st]), dim3(RT_NT), 0, stream, *a, p.off, p.stage[st]);
    const int e = last_error();
    if (e) return e;
  }
  return 0;
}
