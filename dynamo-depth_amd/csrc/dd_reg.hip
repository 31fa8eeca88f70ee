// dd_reg.hip -- regularisers of Trainer.compute_losses on the low-res network outputs (gfx950):
//   dd_smooth_loss    edge-aware smoothness, value + gradient in one pass   (tools.py:311-326, Trainer.py:355-359,380-381,401-402)
//   dd_sparsity_loss  masked BCE-with-logits of the motion probability       (Trainer.py:393-399)
//   dd_ground_loss    RANSAC ground plane + above-ground hinge               (tools.py:76-164, Trainer.py:361-364,425-461)
// All three are HBM-bound streaming kernels over (B,C,h,w) tensors: coalesced row-major reads, neighbours
// come from L1/L2, per-block partial sums go through wave64 shuffles and are folded in a fixed order
// (no float atomics on the loss values).  No host synchronisation anywhere -- the reference needs three
// (tools.py:125-127,137; Trainer.py:398-399).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/dynamo_hip.h"
#include "dd_fuse.h"
#include "dd_math.h"

namespace dd {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// the same sum on the VALU with DPP (quad swaps, row rotations, row broadcasts) instead of six ds_bpermute round trips through
// the LDS pipe per value -- for the tasks that reduce many values per workgroup.  A different (equally fixed) association.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wsum_dpp(float v) {
  v = dpp_add<0xb1, 0xf>(v);     // quad_perm:[1,0,3,2]
  v = dpp_add<0x4e, 0xf>(v);     // quad_perm:[2,3,0,1]
  v = dpp_add<0x124, 0xf>(v);    // row_ror:4
  v = dpp_add<0x128, 0xf>(v);    // row_ror:8   -> every lane holds its row-of-16 sum
  v = dpp_add<0x142, 0xa>(v);    // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

template <int NV, int NTHREADS>
__device__ __forceinline__ float block_sum_dpp(float (&v)[NV], float* red /* NV * NTHREADS/64 */) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float r = wsum_dpp(v[k]);
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
// per-image record of a scale (floats): [0] dot, [1] mean + eps, [2..4] winning plane | dd_fused_loss: [5] sx_d/(mean+eps), [6] sy_d/(mean+eps),
// [7] sx_c, [8] sy_c, [9] sx_m, [10] sy_m  (the image's smoothness sums, folded from the block records)
constexpr int PRE_STRIDE = 16;

constexpr int RT_NT_FUSED = 256;      // workgroup size of every task of the dd_fused_loss passes

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

// mean of image b's plane from its MEAN_BPI partial sums.  Called by EVERY thread of the workgroup (there is a barrier inside):
// the records are fetched once per workgroup and every thread folds them from LDS in the same fixed order.
__device__ __forceinline__ float plane_mean(const float* __restrict__ partial, int b, int n) {
  __shared__ __align__(16) float s_part[MEAN_BPI];
  if (threadIdx.x < MEAN_BPI) s_part[threadIdx.x] = partial[b * MEAN_BPI + threadIdx.x];
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MEAN_BPI; ++i) s += s_part[i];
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
  float inv = 1.f;
  if (NORMALISE) inv = 1.f / (plane_mean(mean, b, n) + 1e-7f);
  if (p < n) {
    const int y = p / w, x = p % w;
    const float* a = inp + (size_t)bc * n;
    const float* im = HAS_IMG ? img + (size_t)b * 3 * n : nullptr;
    // One load phase: the four neighbours are read unconditionally (index clamped to the pixel itself at the border) and the
    // border predicates select afterwards -- with the loads inside `if (x + 1 < w)`-style branches every branch waited for
    // its own memory round trip, and the task was bound by four of those in a row.
    const bool has_r = x + 1 < w, has_l = x > 0, has_d = y + 1 < h, has_u = y > 0;
    const int pr = has_r ? p + 1 : p, pl = has_l ? p - 1 : p, pd = has_d ? p + w : p, pu = has_u ? p - w : p;
    const float a_c = a[p], a_r = a[pr], a_l = a[pl], a_d = a[pd], a_u = a[pu];
    float ic[3], ir[3], il[3], id[3], iu[3];
    if (HAS_IMG) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        ic[ch] = im[ch * n + p]; ir[ch] = im[ch * n + pr]; il[ch] = im[ch * n + pl]; id[ch] = im[ch * n + pd]; iu[ch] = im[ch * n + pu];
      }
    }
    auto edge_w = [&](const float (&q0)[3], const float (&q1)[3]) -> float {
      if (!HAS_IMG) return 1.f;
      const float d = dd_abs(q0[0] - q1[0]) + dd_abs(q0[1] - q1[1]) + dd_abs(q0[2] - q1[2]);
      return __expf(-d / 3.f);
    };
    const float ac = a_c * inv;
    float g = 0.f;
    {                          // term owned by this pixel: |a[p] - a[p+1]| * wx[p]
      const float d = ac - a_r * inv, e = edge_w(ic, ir);
      if (has_r) { acc[0] += dd_abs(d) * e; g += dd_sign(d) * e * wx_scale; }
    }
    {                          // term owned by the left neighbour
      const float d = a_l * inv - ac, e = edge_w(il, ic);
      if (has_l) g -= dd_sign(d) * e * wx_scale;
    }
    {
      const float d = ac - a_d * inv, e = edge_w(ic, id);
      if (has_d) { acc[1] += dd_abs(d) * e; g += dd_sign(d) * e * wy_scale; }
    }
    {
      const float d = a_u * inv - ac, e = edge_w(iu, ic);
      if (has_u) g -= dd_sign(d) * e * wy_scale;
    }
    // NORMALISE: g_out is a temporary holding d/d(normalised input); otherwise it is the caller's accumulator
    if (g_inp) {
      if (NORMALISE) g_inp[(size_t)bc * n + p] = g;
      else g_inp[(size_t)bc * n + p] += g;
    }
    if (NORMALISE) acc[2] = g * a_c;
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
    const float me = plane_mean(mean, bc, n) + 1e-7f;       // C == 1 in the normalised case: bc == b
    if (p < n) {
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

template <int PXT = 1>
__device__ __forceinline__ void sparsity_grad_body(int bx, int by, int gx, const float* __restrict__ delta, const float* __restrict__ delta_sum,
                                                               const float* __restrict__ prob, int B, int n, float inv_total,
                                                               float weight, const float* __restrict__ partials,
                                                               float* __restrict__ g_prob, float* __restrict__ out) {
  __shared__ float s_cnt, s_sum;
  __shared__ int s_gate;
  // the B*SP_BPI records come in with one coalesced pass (every workgroup of the launch repeats this fold: a chain of
  // dependent global loads per image made it the longest part of the task), then ...
  constexpr int REC_CAP = 16 * SP_BPI * 2;           // B <= 16 through LDS (4 KB); larger batches read the records in place
  __shared__ float s_rec[REC_CAP];
  const bool staged = B * SP_BPI * 2 <= REC_CAP;
  if (staged)
    for (int i = threadIdx.x; i < B * SP_BPI * 2; i += SP_NT) s_rec[i] = partials[i];
  __syncthreads();            // also separates two calls from one workgroup (shared motion_prob: both frames in turn)
  const float* rec = staged ? s_rec : partials;
  if (threadIdx.x < 64) {
    // ... one wave folds them in a fixed order; an image with zero static pixels closes the gate
    float cnt = 0.f, sm = 0.f;
    int gate = 1;
    for (int b = 0; b < B; ++b) {
      float c = 0.f, s2 = 0.f;
      for (int i = threadIdx.x; i < SP_BPI; i += 64) { c += rec[((size_t)b * SP_BPI + i) * 2]; s2 += rec[((size_t)b * SP_BPI + i) * 2 + 1]; }
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
#pragma unroll
  for (int it = 0; it < PXT; ++it) {
    const int p = (bx * PXT + it) * SP_NT + threadIdx.x;
    if (p < n) {
      const size_t i = (size_t)b * n + p;
      const float dl = delta[i], x = prob[i], g0 = g_prob[i];     // one load phase; the store is conditional
      if (dl < thr) g_prob[i] = g0 + weight / cnt * (1.f / (1.f + expf(-x)));      // d softplus = sigmoid
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
static_assert(GP_NT == 2 * GP_MAX_IT, "ground_count_body splits the workgroup into two halves of GP_MAX_IT candidates");

// one thread per RANSAC candidate: least squares y = w1*x + w2*z + w3 through np points (tools.py:141-154),
// (AtA + 1e-6 on EVERY entry)^-1 At B, solved in double to stay clear of the conditioning of 5 nearby points
__device__ __forceinline__ void ground_candidates_body(int bx, int by, int gx, const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                                   const int32_t* __restrict__ rand_idx, int B, int h, int w,
                                                                   int rows, int np, int max_it, DepthParams dp,
                                                                   float* __restrict__ cand /* (B*max_it,3) */,
                                                                   int* __restrict__ counts /* (B*max_it), zeroed here */) {
  const int j = bx * GP_NT + threadIdx.x;
  if (j >= B * max_it) return;
  if (counts) counts[j] = 0;
  float cv[3];
  ground_candidate_solve(disp, inv_K, rand_idx, B, h, w, rows, np, max_it, dp, j, cv);
  for (int i = 0; i < 3; ++i) cand[(size_t)j * 3 + i] = cv[i];
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
// `part` == nullptr (per-term entry points): one point per thread, counts[] (zeroed) collects the inliers with atomics.
// `part` != nullptr (dd_reg_losses): the workgroup covers GS_SLABS * GP_NT points and leaves its max_it inlier counts as one
// record part[(img * gx + bx) * max_it + k] (plain stores; ground_count_body adds the records up).  The candidate planes
// sit in the lanes of each wave and are broadcast with v_readlane, a thread keeps GS_SLABS points in registers, and lane k
// accumulates the wave's count for candidates k and k + 64: no memory traffic and no atomics inside the loop.
constexpr int GS_SLABS = 4;
__device__ __forceinline__ void ground_score_body(int bx, int by, int gx, const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                              const float* __restrict__ cand, int B, int h, int w, int rows,
                                                              int max_it, float tol, DepthParams dp,
                                                              int* __restrict__ counts /* (B*max_it) zeroed */, int* __restrict__ part = nullptr) {
  const int img = by;
  const int n = h * w, base = (h - rows) * w, ng = rows * w;
  const int lane = threadIdx.x & 63;
  const float* disp_b = disp + (size_t)img * n * (inv_K ? 1 : 3);
  const float* invK_b = inv_K ? inv_K + img * 16 : nullptr;
  if (part) {
    __shared__ int s_wave[GP_NT / 64][GP_MAX_IT];
    float P[GS_SLABS][3];
#pragma unroll
    for (int sl = 0; sl < GS_SLABS; ++sl) {
      const int q = (bx * GS_SLABS + sl) * GP_NT + threadIdx.x;
      P[sl][0] = 0.f; P[sl][1] = 3e38f; P[sl][2] = 0.f;            // a point beyond the data: |distance| is huge for every plane
      if (q < ng) ground_point(disp_b, invK_b, dp, w, base + q, P[sl], n);
    }
    // lane k holds the planes of candidates k and k + 64; the loop broadcasts them with v_readlane (k is wave-uniform)
    float pl[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int k = lane + 64 * hf;
      if (k < max_it) {
        const float* c = cand + (size_t)(img + k * B) * 3;
        pl[hf][0] = c[0]; pl[hf][1] = c[1]; pl[hf][2] = c[2];
      }
    }
    auto bcast = [](float v, int k) -> float { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); };
    int cnt[2] = {0, 0};             // inliers of candidate `lane` / `lane + 64` seen by this wave
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int kn = min(max_it - 64 * hf, 64);
      for (int k = 0; k < kn; ++k) {
        const float c0 = bcast(pl[hf][0], k), c1 = bcast(pl[hf][1], k), c2 = bcast(pl[hf][2], k);
        int tot = 0;
#pragma unroll
        for (int sl = 0; sl < GS_SLABS; ++sl) {
          const float dist = P[sl][0] * c0 + P[sl][2] * c1 + c2 - P[sl][1];
          tot += __popcll(__ballot(dd_abs(dist) < tol));
        }
        if (lane == k) cnt[hf] += tot;
      }
    }
    const int lo = cnt[0], hi = cnt[1];
    const int wave = threadIdx.x >> 6;
    s_wave[wave][lane] = lo;
    s_wave[wave][lane + 64] = hi;
    __syncthreads();
    for (int k = threadIdx.x; k < max_it; k += GP_NT) {
      int t = 0;
#pragma unroll
      for (int wv = 0; wv < GP_NT / 64; ++wv) t += s_wave[wv][k];
      part[((size_t)img * gx + bx) * max_it + k] = t;
    }
    return;
  }
  __shared__ float s_c[GP_MAX_IT * 3];
  __shared__ int s_j[GP_MAX_IT];
  __shared__ int s_cnt[GP_MAX_IT];
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
  if (live) ground_point(disp_b, invK_b, dp, w, base + q, P, n);
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

// The same scoring on the matrix pipe (dd_reg_losses): distances of N points to max_it planes are a (N x 4)(4 x max_it) product --
// rows [x, z, 1, -y], columns [w1, w2, w3, 1] -- and that, unlike the 3x3 window sums of the photometric kernel, IS a dense product:
// two v_mfma_f32_32x32x2_f32 per 32 points x 32 candidates (exact fp32 products, fp32 accumulation), then |d| < tol as a compare +
// add-with-carry per value.  Lane l supplies point l % 32 (both halves of the wave compute the same 32 points; k = l / 32 selects the
// coordinate) and candidate column l % 32; its 16 results are 16 points of ONE candidate column, so a lane counts in registers and
// the two halves meet once at the end.  0.25 cycles per (point, candidate) against ~0.8 for the readlane / ballot form above, which
// made this task the second largest of the regularisers (24 us; scripts/reg_task_costs.sh).  Same records as ground_score_body.
typedef float gs_f16v __attribute__((ext_vector_type(16)));
// rand_idx != nullptr (dd_fused_loss): the workgroup first SOLVES the max_it candidates it scores (candidate img + k*B, thread k; five
// gathered points and a 3x3 solve in fp64 each -- identical arithmetic in every workgroup, hence identical planes) instead of reading
// them from a launch in front of this one; workgroup 0 of an image also publishes them in `cand_out` for the winner's look-up.
__device__ __forceinline__ void ground_score_mfma_body(int bx, int img, int gx, const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                       const float* __restrict__ cand, int B, int h, int w, int rows, int max_it, float tol,
                                                       DepthParams dp, int* __restrict__ part, const int32_t* __restrict__ rand_idx = nullptr,
                                                       int np = 0, float* __restrict__ cand_out = nullptr) {
  __shared__ int s_wave[GP_NT / 64][GP_MAX_IT];
  __shared__ float s_cand[GP_MAX_IT * 3];
  const int n = h * w, base = (h - rows) * w, ng = rows * w;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, hi = lane >> 5;
  const float* disp_b = disp + (size_t)img * n;
  const float* invK_b = inv_K + img * 16;
  if (rand_idx) {             // uniform
    if ((int)threadIdx.x < max_it) {
      const int jc = img + (int)threadIdx.x * B;
      float cv[3];
      ground_candidate_solve(disp, inv_K, rand_idx, B, h, w, rows, np, max_it, dp, jc, cv);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        s_cand[threadIdx.x * 3 + i] = cv[i];
        if (bx == 0 && cand_out) cand_out[(size_t)jc * 3 + i] = cv[i];
      }
    }
    __syncthreads();
  }
  // candidate columns: tile t holds candidates t*32 .. t*32+31 (of the max_it scored on this image: j' = img + k*B, tools.py:130)
  float b1[4], b2[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int k = t * 32 + j;
    float c0 = 0.f, c1 = 0.f, c2 = 3e38f;               // a padding column: every distance huge
    if (k < max_it) {
      if (rand_idx) { c0 = s_cand[k * 3]; c1 = s_cand[k * 3 + 1]; c2 = s_cand[k * 3 + 2]; }
      else {
        const float* c = cand + (size_t)(img + k * B) * 3;
        c0 = c[0]; c1 = c[1]; c2 = c[2];
      }
    }
    b1[t] = hi ? c1 : c0;           // k = 0: w1, k = 1: w2      (pairs with [x, z])
    b2[t] = hi ? 1.f : c2;          // k = 0: w3, k = 1: 1       (pairs with [1, -y])
  }
  int cnt[4] = {0, 0, 0, 0};
  constexpr int TILES = GS_SLABS * GP_NT / (GP_NT / 64) / 32;       // 32-point tiles per wave: the workgroup covers GS_SLABS * GP_NT points
  // the wave's points first, all loads in flight together (one dependent load per tile inside the product loop made the kernel
  // latency-bound: 19 us for ~6 us of matrix work)
  __shared__ float2 s_pts[TILES][GP_NT];                 // (parked per thread: a register array would need the product loop fully unrolled -- 452 VGPRs)
#pragma unroll
  for (int tile = 0; tile < TILES; ++tile) {
    const int q = ((bx * (GP_NT / 64) + wave) * TILES + tile) * 32 + j;
    float P[3] = {0.f, 3e38f, 0.f};                     // a point beyond the data: |distance| is huge for every plane
    if (q < ng) ground_point(disp_b, invK_b, dp, w, base + q, P);
    s_pts[tile][threadIdx.x] = make_float2(hi ? P[2] : P[0], hi ? -P[1] : 1.f);
  }
#pragma unroll 1
  for (int tile = 0; tile < TILES; ++tile) {
    const float2 av = s_pts[tile][threadIdx.x];          // the thread's own slot: no barrier needed
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      gs_f16v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b2[t], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1[t], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) cnt[t] += (dd_abs(acc[r]) < tol) ? 1 : 0;
    }
  }
  // the two halves of the wave hold the same candidate columns over different point rows
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tot = cnt[t] + __shfl_xor(cnt[t], 32, 64);
    if (hi == 0) s_wave[wave][t * 32 + j] = tot;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < max_it; k += GP_NT) {
    int tsum = 0;
#pragma unroll
    for (int wv = 0; wv < GP_NT / 64; ++wv) tsum += s_wave[wv][k];
    part[((size_t)img * gx + bx) * max_it + k] = tsum;
  }
}

// ... as a kernel of its own, one launch for all scales (between stage 1 and stage 2): the 64 accumulator registers of four candidate
// tiles would otherwise set the occupancy of every task of reg_stage_kernel (95 -> 128+ VGPRs).
struct ScoreScale {
  const float* disp;
  const float* inv_K;
  const float* cand;
  int* part;
  int h, w, rows, gx, first;
};
struct ScoreArgs {
  ScoreScale sc[DD_MAX_SCALES];
  int num_scales, B, max_it;
  float tol;
  DepthParams dp;
};
__global__ __launch_bounds__(GP_NT) void ground_score_all_kernel(const ScoreArgs a) {
  int si = 0;
#pragma unroll
  for (int i = 1; i < DD_MAX_SCALES; ++i)
    if (i < a.num_scales && (int)blockIdx.x >= a.sc[i].first) si = i;
  switch (si) {       // constant offsets into the kernel-argument block
#define DD_GS_SCALE(I) case I: { const int vb = (int)blockIdx.x - a.sc[I].first; \
      ground_score_mfma_body(vb % a.sc[I].gx, vb / a.sc[I].gx, a.sc[I].gx, a.sc[I].disp, a.sc[I].inv_K, a.sc[I].cand, a.B, a.sc[I].h, a.sc[I].w, a.sc[I].rows, \
                             a.max_it, a.tol, a.dp, a.sc[I].part); break; }
    DD_GS_SCALE(0) DD_GS_SCALE(1) DD_GS_SCALE(2) DD_GS_SCALE(3)
#undef DD_GS_SCALE
    default: break;
  }
}

// counts[img + k*B] = sum over the gx records of image img (the pairing of ground_score_body), one workgroup per image:
// thread (k, half) adds every second record, the two halves meet in LDS.  Integer sums: any order gives the same result.
__device__ __forceinline__ void ground_count_body(int by, int gx, const int* __restrict__ part, int B, int max_it, int* __restrict__ counts) {
  __shared__ int s_half[GP_MAX_IT];
  const int img = by, k = threadIdx.x & (GP_MAX_IT - 1), half = threadIdx.x / GP_MAX_IT;     // GP_NT == 2 * GP_MAX_IT
  int acc = 0;
  if (k < max_it) {
#pragma unroll 8
    for (int r = half; r < gx; r += 2) acc += part[((size_t)img * gx + r) * max_it + k];
  }
  if (half == 1) s_half[k] = acc;
  __syncthreads();
  if (half == 0 && k < max_it) counts[img + k * B] = acc + s_half[k];
}

__global__ __launch_bounds__(GP_NT) void ground_score_kernel(const float* __restrict__ disp, const float* __restrict__ inv_K,
                                                              const float* __restrict__ cand, int B, int h, int w, int rows,
                                                              int max_it, float tol, DepthParams dp,
                                                              int* __restrict__ counts /* (B*max_it) zeroed */) {
  ground_score_body(blockIdx.x, blockIdx.y, gridDim.x, disp, inv_K, cand, B, h, w, rows, max_it, tol, dp, counts);
}

// index of the first maximum of counts[0..m) (argmax semantics), by the whole workgroup: every thread takes a strided share,
// then one LDS max over keys (count << 32 | ~index).  A serial scan by one thread costs m dependent global loads per
// workgroup, which dominated the hinge pass.  Counts are inlier numbers (>= 0).
__device__ __forceinline__ int first_argmax(const int* __restrict__ counts, int m) {
  __shared__ unsigned long long s_key;
  if (threadIdx.x == 0) s_key = 0ull;
  __syncthreads();
  unsigned long long key = 0ull;
  for (int k = threadIdx.x; k < m; k += GP_NT) {
    const unsigned long long kk = (static_cast<unsigned long long>(static_cast<unsigned>(counts[k])) << 32) | (0xFFFFFFFFu - static_cast<unsigned>(k));
    key = kk > key ? kk : key;
  }
  if (threadIdx.x < m) atomicMax(&s_key, key);
  __syncthreads();
  return static_cast<int>(0xFFFFFFFFu - static_cast<unsigned>(s_key & 0xFFFFFFFFull));
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
  const int best = first_argmax(counts + b * max_it, max_it);
  if (threadIdx.x < 3) {
    s_w[threadIdx.x] = cand[((size_t)b * max_it + best) * 3 + threadIdx.x];
    if (bx == 0) plane[b * 3 + threadIdx.x] = s_w[threadIdx.x];
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
    const float dv = disp[(size_t)b * n + p], g0 = g_disp ? g_disp[(size_t)b * n + p] : 0.f;     // one load phase
    float gd = w3 / (ray[1] - ray[0] * w1 - ray[2] * w2);
    const bool invalid = (gd < 0.f) || (gd > max_depth);        // NaN compares false -> stays, like the reference
    if (invalid) gd = max_depth;
    if (gd != max_depth) {
      const float gdisp = (1.f / gd - dp.lo) / dp.span;
      const float diff = dv - gdisp;
      if (!(diff > 0.f)) {                                       // disp_diff[disp_diff > 0] = 0
        v[0] = diff;
        if (g_disp) g_disp[(size_t)b * n + p] = g0 + weight;
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
  const int best = first_argmax(counts + b * max_it, max_it);
  if (threadIdx.x < 3) {
    s_w[threadIdx.x] = cand[((size_t)b * max_it + best) * 3 + threadIdx.x];
    if (blockIdx.x == 0) plane[b * 3 + threadIdx.x] = s_w[threadIdx.x];
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
  __shared__ float s_res[DD_MAX_RES];
  const int t = threadIdx.x;
  // the raw sums come in with ONE coalesced load phase (the loop below used to read res[i] under a per-thread predicate: 128
  // exec-masked global loads in a row, each waiting for its own round trip -- 13 us for a one-workgroup kernel)
  for (int i = t; i < DD_MAX_RES; i += 64) s_res[i] = i < a.n ? res[i] : 0.f;
  __syncthreads();
  if (t < DD_MAX_SCALES * DD_NUM_TERMS) {
    const int s = t / DD_NUM_TERMS, k = t % DD_NUM_TERMS;
    float acc = 0.f;
    // fully unrolled: every a.* access sits at a constant offset of the kernel-argument block (with a run-time index the
    // compiler copies the whole struct into scratch, per thread); branch-free: a record that is not this thread's adds +0
#pragma unroll
    for (int i = 0; i < DD_MAX_RES; ++i) {
      const float v = a.norm[i] * s_res[i];
      acc += (i < a.n && a.term_of[i] == k && a.scale_of[i] == s) ? v : 0.f;
    }
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

// dd_reg_losses_finish: stage 5 and the assembling in ONE workgroup -- the hinge partials of every scale are folded in a fixed
// order into their res slot, then the same arithmetic as assemble_kernel.
struct HingeFold {
  const float* part[DD_MAX_SCALES];     // per-workgroup hinge sums of the scale (nullptr: no ground term)
  int count[DD_MAX_SCALES];
};

__global__ __launch_bounds__(256) void finish_kernel(float* __restrict__ res, const HingeFold hf, const DDAssembleArgs a, float* __restrict__ loss,
                                                     float* __restrict__ out) {
  __shared__ float red[4];
  __shared__ float term[DD_MAX_SCALES][DD_NUM_TERMS];
  for (int s = 0; s < a.num_scales; ++s) {
    if (!hf.part[s]) continue;             // uniform
    float v[1] = {0.f};
    for (int i = threadIdx.x; i < hf.count[s]; i += 256) v[0] += hf.part[s][i];
    const float r = block_sum<1, 256>(v, red);
    if (threadIdx.x == 0) res[s * DD_REG_RES_STRIDE + 14] = r;
  }
  __threadfence_block();
  __syncthreads();
  __shared__ float s_res[DD_MAX_RES];
  const int t = threadIdx.x;
  // one coalesced load phase for the raw sums (see assemble_kernel), behind the fold above (same workgroup: the barrier orders it)
  if (t < DD_MAX_RES) s_res[t] = t < a.n ? res[t] : 0.f;
  __syncthreads();
  if (t < DD_MAX_SCALES * DD_NUM_TERMS) {
    const int s = t / DD_NUM_TERMS, k = t % DD_NUM_TERMS;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < DD_MAX_RES; ++i) {
      const float v = a.norm[i] * s_res[i];
      acc += (i < a.n && a.term_of[i] == k && a.scale_of[i] == s) ? v : 0.f;
    }
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

// dd_fused_loss's last launch: finish_kernel with (i) the hinge fold of scale s on wave s (no barrier chain over the scales) and
// (ii) the fold of the per-image smoothness sums (image_fold_body) into their res slots: thread (s, group, x|y) adds the B values in
// image order.  slots: 4 bits per (scale, group) = res slot of the group's x sum inside the scale's block, 15 = group off.
struct ImageSums {
  const float* pre_all;              // [scale][image][PRE_STRIDE]
  unsigned long long slots;
  int B;
};
__global__ __launch_bounds__(256) void fused_finish_kernel(float* __restrict__ res, const HingeFold hf, const ImageSums im, const DDAssembleArgs a,
                                                           float* __restrict__ loss, float* __restrict__ out) {
  __shared__ float term[DD_MAX_SCALES][DD_NUM_TERMS];
  __shared__ float s_res[DD_MAX_RES];
  const int t = threadIdx.x, lane = t & 63;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  static_assert(DD_MAX_SCALES == 4, "one wave per scale");
  // every global load of the kernel is issued up front -- the raw sums, the hinge partials, the per-image sums are independent --
  // and the folded values go straight into the LDS copy of `res` (finish_kernel stores them to `res` and reads them back: one more
  // global round trip in a kernel that is nothing but round trips)
  const float r0 = (t < DD_MAX_RES && t < a.n) ? res[t] : 0.f;
  float hv = 0.f;
  const bool hinge = wv < a.num_scales && hf.part[wv] != nullptr;
  if (hinge) {
    // four loads in flight per lane (a run-time trip count leaves one dependent round trip per iteration: 12 of them at scale 0)
    const float* part = hf.part[wv];
    const int cnt = hf.count[wv];
    float h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f;
    int i = lane;
    for (; i + 192 < cnt; i += 256) { h0 += part[i]; h1 += part[i + 64]; h2 += part[i + 128]; h3 += part[i + 192]; }
    for (; i < cnt; i += 64) h0 += part[i];
    hv = (h0 + h1) + (h2 + h3);
  }
  // per-image smoothness sums: thread (part, combo) adds images part, part + 8, ... of combo = (scale, group, x|y); the eight parts
  // meet in LDS in a fixed order
  __shared__ float s_im[8][32];
  const int combo = t & 31, ipart = t >> 5;
  int islot = -1;
  {
    const int s = combo / 6, g = (combo % 6) >> 1, xy = combo & 1;
    float iv = 0.f;
    if (im.pre_all && combo < DD_MAX_SCALES * 6 && s < a.num_scales) {
      const int slot = (int)((im.slots >> (4 * (s * 3 + g))) & 15ull);
      if (slot != 15) {
        islot = s * DD_REG_RES_STRIDE + slot + xy;
        float v0 = 0.f, v1 = 0.f;
        int b = ipart;
        for (; b + 8 < im.B; b += 16) { v0 += im.pre_all[((size_t)s * im.B + b) * PRE_STRIDE + 5 + 2 * g + xy]; v1 += im.pre_all[((size_t)s * im.B + b + 8) * PRE_STRIDE + 5 + 2 * g + xy]; }
        if (b < im.B) v0 += im.pre_all[((size_t)s * im.B + b) * PRE_STRIDE + 5 + 2 * g + xy];
        iv = v0 + v1;
      }
    }
    s_im[ipart][combo] = iv;
  }
  if (t < DD_MAX_RES) s_res[t] = r0;
  __syncthreads();
  if (hinge) {
    hv = wsum_dpp(hv);
    if (lane == 0) { s_res[wv * DD_REG_RES_STRIDE + 14] = hv; res[wv * DD_REG_RES_STRIDE + 14] = hv; }
  }
  if (ipart == 0 && islot >= 0) {
    float iv = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) iv += s_im[k][combo];
    s_res[islot] = iv; res[islot] = iv;
  }
  __syncthreads();
  if (t < DD_MAX_SCALES * DD_NUM_TERMS) {
    const int s = t / DD_NUM_TERMS, k = t % DD_NUM_TERMS;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < DD_MAX_RES; ++i) {
      const float v = a.norm[i] * s_res[i];
      acc += (i < a.n && a.term_of[i] == k && a.scale_of[i] == s) ? v : 0.f;
    }
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
// smoothness of ALL entries of a scale in one pass (dd_reg_losses)
// =================================================================================================
// One thread per low-res pixel of one image handles every smoothed channel of the scale (disparity | flow x 3 | mask, or the
// per-frame variants): the colour neighbourhood is loaded once and the four edge weights exp(-mean_c |dI|) are formed once for
// all NCH channels (the per-entry kernel evaluated them per channel: 5x the exps and 5x the colour loads in fine_tune).  All
// loads of the thread -- 15 colour values, 5 x NCH inputs -- are issued before the first is consumed.  Same per-term
// arithmetic as smooth_body; the block's partial sums are per ENTRY (channels of an entry summed inside the thread).
// Entry layout: DDRegScale.smooth[k], k in ascending order; channel ch of the scale = (entry, channel) in that order.
constexpr int SMA_MAX_ENTRIES = DD_REG_SMOOTH;
constexpr int SMA_MAX_CH = 9;
// Pixels per thread of the element-wise tasks of dd_reg_losses.  The tasks are bound by the number of workgroups, not by bytes:
// a workgroup spends most of its life in its prologue / epilogue (fold of per-image records, barriers, a 15-value block
// reduction) -- at one pixel per thread the stage kernels retired ~160 workgroups per microsecond whatever they carried
// (23 000 workgroups = 149 us in stage 2).  More pixels per workgroup amortise that part.
constexpr int SMA_PXT = 4;      // smoothness pass: four pixels per thread -- consecutive ones (16-byte loads, shared neighbours) when the rows allow it
constexpr int SPG_PXT = 8;      // sparsity gradient
constexpr int FIN_PXT = 8;      // disparity-gradient finish (normalisation adjoint + ground hinge)

// one smoothed channel of a scale, resolved on the host (reg_plan): the kernel reads it with a compile-time channel index, so
// every field stays in scalar registers
// wave-uniform values derived through run-time control flow, pinned to scalar registers
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
template <class T>
__device__ __forceinline__ T* uni(T* q) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(q);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v)), hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
  return reinterpret_cast<T*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

template <int NCH>
__device__ __forceinline__ void smooth_all_body(int bx, int b, int gx, const DDRegScale& sc, int B, const float* __restrict__ mean,
                                                float* __restrict__ g_tmp /* normalised entry: d/d(normalised), (B,n) */,
                                                float* const (&part)[SMA_MAX_ENTRIES] /* per entry: [(b*gx+bx)*4 + j] */) {
  __shared__ float red[3 * SMA_MAX_ENTRIES * SM_NT / 64];
  const int h = sc.h, w = sc.w, n = h * w;
  // channel table: channel ch of the scale = (entry k, channel c of it), entries in ascending order.  Every value is
  // wave-uniform; uni() keeps it in scalar registers although it comes out of run-time control flow.
  const float* in[NCH];
  float* gp[NCH];
  float wxs[NCH], wys[NCH];
  int ent[NCH];
  bool nrm[NCH];
  {
    int ch = 0;
#pragma unroll
    for (int k = 0; k < DD_REG_SMOOTH; ++k) {
      const DDRegSmooth& sm = sc.smooth[k];
      const int Cc = sm.inp ? sm.C : 0;
      const float cnt = static_cast<float>(B) * static_cast<float>(Cc > 0 ? Cc : 1);
      const float wx = sm.weight / (cnt * h * (w - 1)), wy = sm.weight / (cnt * (h - 1) * w);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (c >= Cc) continue;
        // ch is a run-time value here; the assignment below is unrolled over the compile-time slot index
#pragma unroll
        for (int slot = 0; slot < NCH; ++slot) {
          if (slot != ch) continue;
          in[slot] = uni(sm.inp + ((size_t)b * Cc + c) * n);
          nrm[slot] = sm.normalise != 0;
          gp[slot] = uni(sm.g_inp ? (sm.normalise ? g_tmp + (size_t)b * n : sm.g_inp + ((size_t)b * Cc + c) * n) : (float*)nullptr);
          wxs[slot] = uni(wx); wys[slot] = uni(wy);
          ent[slot] = k;
        }
        ++ch;
      }
    }
  }
  float inv_mean = 1.f;
  {
    bool any = false;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) any = any || nrm[ch];
    if (any) inv_mean = 1.f / (plane_mean(mean, b, n) + 1e-7f);      // barrier inside: uniform condition
  }
  float acc[3 * SMA_MAX_ENTRIES];
#pragma unroll
  for (int i = 0; i < 3 * SMA_MAX_ENTRIES; ++i) acc[i] = 0.f;
#pragma unroll
  for (int it = 0; it < SMA_PXT; ++it) {
    const int p = (bx * SMA_PXT + it) * SM_NT + threadIdx.x;
    if (p >= n) continue;
    const int y = p / w, x = p - y * w;
    const bool has_r = x + 1 < w, has_l = x > 0, has_d = y + 1 < h, has_u = y > 0;
    const int pr = has_r ? p + 1 : p, pl = has_l ? p - 1 : p, pd = has_d ? p + w : p, pu = has_u ? p - w : p;
    const float* im = sc.img + (size_t)b * 3 * n;
    float ic[3], ir[3], il[3], id[3], iu[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      ic[ch] = im[ch * n + p]; ir[ch] = im[ch * n + pr]; il[ch] = im[ch * n + pl]; id[ch] = im[ch * n + pd]; iu[ch] = im[ch * n + pu];
    }
    auto edge_w = [](const float (&q0)[3], const float (&q1)[3]) -> float {
      const float d = dd_abs(q0[0] - q1[0]) + dd_abs(q0[1] - q1[1]) + dd_abs(q0[2] - q1[2]);
      return __expf(-d / 3.f);
    };
    // the channels go in groups of at most five: every load of a group is issued before the first is consumed, and the
    // registers of one group bound the kernel's footprint (nine channels at once cost the stage kernel an occupancy step)
    constexpr int GROUP = 5;
    float e_r = 0.f, e_l = 0.f, e_d = 0.f, e_u = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += GROUP) {
      constexpr int dummy = 0;
      (void)dummy;
      float a_c[GROUP], a_r[GROUP], a_l[GROUP], a_d[GROUP], a_u[GROUP], g_old[GROUP];
#pragma unroll
      for (int j = 0; j < GROUP; ++j) {
        const int ch = c0 + j < NCH ? c0 + j : NCH - 1;
        a_c[j] = in[ch][p]; a_r[j] = in[ch][pr]; a_l[j] = in[ch][pl]; a_d[j] = in[ch][pd]; a_u[j] = in[ch][pu];
        g_old[j] = (c0 + j < NCH && gp[ch] && !nrm[ch]) ? gp[ch][p] : 0.f;
      }
      if (c0 == 0) { e_r = edge_w(ic, ir); e_l = edge_w(il, ic); e_d = edge_w(ic, id); e_u = edge_w(iu, ic); }
#pragma unroll
      for (int j = 0; j < GROUP; ++j) {
        if (c0 + j >= NCH) continue;           // compile-time
        const int ch = c0 + j;
        const float inv = nrm[ch] ? inv_mean : 1.f;
        const float ac = a_c[j] * inv;
        float g = 0.f, sx = 0.f, sy = 0.f;
        {
          const float d = ac - a_r[j] * inv;
          if (has_r) { sx = dd_abs(d) * e_r; g += dd_sign(d) * e_r * wxs[ch]; }
        }
        {
          const float d = a_l[j] * inv - ac;
          if (has_l) g -= dd_sign(d) * e_l * wxs[ch];
        }
        {
          const float d = ac - a_d[j] * inv;
          if (has_d) { sy = dd_abs(d) * e_d; g += dd_sign(d) * e_d * wys[ch]; }
        }
        {
          const float d = a_u[j] * inv - ac;
          if (has_u) g -= dd_sign(d) * e_u * wys[ch];
        }
        if (gp[ch]) gp[ch][p] = nrm[ch] ? g : g_old[j] + g;
        // the entry index is wave-uniform but not a compile-time constant: a select chain keeps acc[] in registers
#pragma unroll
        for (int e = 0; e < SMA_MAX_ENTRIES; ++e) {
          const bool mine = ent[ch] == e;
          acc[3 * e + 0] += mine ? sx : 0.f;
          acc[3 * e + 1] += mine ? sy : 0.f;
          acc[3 * e + 2] += (mine && nrm[ch]) ? g * a_c[j] : 0.f;
        }
      }
    }
  }
  const float r = block_sum<3 * SMA_MAX_ENTRIES, SM_NT>(acc, red);
#pragma unroll
  for (int e = 0; e < SMA_MAX_ENTRIES; ++e) {
    const int j = (int)threadIdx.x - 3 * e;
    if (j >= 0 && j < 3 && part[e]) part[e][((size_t)b * gx + bx) * 4 + j] = r;
  }
}

// The same pass with the thread's four pixels CONSECUTIVE in a row (w % 4 == 0, 16-byte aligned planes -- decided by the planner):
// per plane a thread issues three 16-byte loads (its row, the row above, the row below) and two 4-byte loads (the pixel left and
// right of the quad) instead of twenty 4-byte ones, the five horizontal differences / edge weights of the quad are formed once
// (a pixel's left term IS its left neighbour's right term: |a - b| = |b - a| bit for bit), and gradients leave as 16-byte stores.
// Per-pixel arithmetic, its order, and the workgroup's pixel set are those of smooth_all_body; only the order in which a thread
// adds its four pixels into the block's partial sums is its own.  A kernel of its own (smooth_quad_kernel, one launch for all
// scales, between stage 1 and stage 2): inside reg_stage_kernel its registers (130 for five channels) would set the occupancy of
// every other task, and four instantiations next to the other bodies made the compiler copy the 1 KB argument block to scratch.
struct SmoothQuadScale {
  const float* img;                  // (B,3,h,w) colour pyramid level
  const float* mean;                 // per-image partial sums of the normalised entry (plane_mean), or nullptr
  const float* in[SMA_MAX_CH];       // channel ch of image 0
  float* gp[SMA_MAX_CH];             // its gradient plane of image 0 (normalised channel: the d/d(normalised) temporary), or nullptr
  float* part[SMA_MAX_ENTRIES];      // per entry: block records [(b*gx+bx)*4 + j]
  int bstride[SMA_MAX_CH];           // floats between two images of the channel's tensor / gradient
  float wx[SMA_MAX_CH], wy[SMA_MAX_CH];
  signed char ent[SMA_MAX_CH], nrm[SMA_MAX_CH];
  int h, w, gx, first, npad;         // gx workgroups per image; first workgroup of the scale inside the launch (a multiple of 8); its
};                                   // workgroup count rounded up to a multiple of 8 (the XCD band remap)
struct SmoothQuadArgs {
  SmoothQuadScale sc[DD_MAX_SCALES];
  int num_scales, B;
};

template <int NCH>
__device__ __forceinline__ void smooth_quad_body(int bx, int b, const SmoothQuadScale& q) {
  __shared__ float red[3 * SMA_MAX_ENTRIES * SM_NT / 64];
  const int h = q.h, w = q.w, n = h * w, gx = q.gx;
  float inv_mean = 1.f;
  if (q.mean) inv_mean = 1.f / (plane_mean(q.mean, b, n) + 1e-7f);      // barrier inside: uniform condition
  float acc[3 * SMA_MAX_ENTRIES];
#pragma unroll
  for (int i = 0; i < 3 * SMA_MAX_ENTRIES; ++i) acc[i] = 0.f;
  const int p0 = (bx * SM_NT + (int)threadIdx.x) * 4;
  if (p0 < n) {
    const int y = p0 / w, x0 = p0 - y * w;
    const bool has_l = x0 > 0, has_r = x0 + 4 < w, has_u = y > 0, has_d = y + 1 < h;
    const int pu = has_u ? p0 - w : p0, pd = has_d ? p0 + w : p0, pl = has_l ? p0 - 1 : p0, pr = has_r ? p0 + 4 : p0 + 3;
    auto ld4 = [](const float* ptr) -> float4 { return *reinterpret_cast<const float4*>(ptr); };
    // ---- edge weights: five horizontal (left of pixel 0 ... right of pixel 3), four down, four up ----
    float dh[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, dd[4] = {0.f, 0.f, 0.f, 0.f}, du[4] = {0.f, 0.f, 0.f, 0.f};
    {
      const float* im = q.img + (size_t)b * 3 * n;
      float4 C4[3], U4[3], D4[3];
      float cl[3], cr[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        C4[ch] = ld4(im + ch * n + p0); U4[ch] = ld4(im + ch * n + pu); D4[ch] = ld4(im + ch * n + pd);
        cl[ch] = im[ch * n + pl]; cr[ch] = im[ch * n + pr];
      }
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float c[6] = {cl[ch], C4[ch].x, C4[ch].y, C4[ch].z, C4[ch].w, cr[ch]};
        const float u[4] = {U4[ch].x, U4[ch].y, U4[ch].z, U4[ch].w}, d[4] = {D4[ch].x, D4[ch].y, D4[ch].z, D4[ch].w};
#pragma unroll
        for (int k = 0; k < 5; ++k) dh[k] += dd_abs(c[k] - c[k + 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { dd[i] += dd_abs(c[i + 1] - d[i]); du[i] += dd_abs(u[i] - c[i + 1]); }
      }
    }
    float eh[5], ed[4], eu[4];
#pragma unroll
    for (int k = 0; k < 5; ++k) eh[k] = __expf(-dh[k] / 3.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) { ed[i] = __expf(-dd[i] / 3.f); eu[i] = __expf(-du[i] / 3.f); }
    // ---- the channels, in groups whose loads are all issued before the first is consumed ----
    constexpr int GROUP = 3;
#pragma unroll
    for (int c0 = 0; c0 < NCH; c0 += GROUP) {
      float4 A4[GROUP], AU4[GROUP], AD4[GROUP], G4[GROUP];
      float al[GROUP], ar[GROUP];
#pragma unroll
      for (int j = 0; j < GROUP; ++j) {
        const int ch = c0 + j < NCH ? c0 + j : NCH - 1;
        const float* src = q.in[ch] + (size_t)b * q.bstride[ch];
        A4[j] = ld4(src + p0); AU4[j] = ld4(src + pu); AD4[j] = ld4(src + pd);
        al[j] = src[pl]; ar[j] = src[pr];
        G4[j] = (c0 + j < NCH && q.gp[ch] && !q.nrm[ch]) ? ld4(q.gp[ch] + (size_t)b * q.bstride[ch] + p0) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < GROUP; ++j) {
        if (c0 + j >= NCH) continue;           // compile-time
        const int ch = c0 + j;
        const bool nrm = q.nrm[ch] != 0;
        const float inv = nrm ? inv_mean : 1.f, wxs = q.wx[ch], wys = q.wy[ch];
        const float raw[6] = {al[j], A4[j].x, A4[j].y, A4[j].z, A4[j].w, ar[j]};
        const float up[4] = {AU4[j].x, AU4[j].y, AU4[j].z, AU4[j].w}, dn[4] = {AD4[j].x, AD4[j].y, AD4[j].z, AD4[j].w};
        const float old[4] = {G4[j].x, G4[j].y, G4[j].z, G4[j].w};
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = raw[k] * inv;
        float hd[5];                         // hd[k] = v[k] - v[k+1]: right term of pixel k-1, left term of pixel k
#pragma unroll
        for (int k = 0; k < 5; ++k) hd[k] = v[k] - v[k + 1];
        float gout[4], sx = 0.f, sy = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool r_ok = i < 3 || has_r, l_ok = i > 0 || has_l;
          float g = 0.f;
          if (r_ok) { sx += dd_abs(hd[i + 1]) * eh[i + 1]; g += dd_sign(hd[i + 1]) * eh[i + 1] * wxs; }
          if (l_ok) g -= dd_sign(hd[i]) * eh[i] * wxs;
          {
            const float d = v[i + 1] - dn[i] * inv;
            if (has_d) { sy += dd_abs(d) * ed[i]; g += dd_sign(d) * ed[i] * wys; }
          }
          {
            const float d = up[i] * inv - v[i + 1];
            if (has_u) g -= dd_sign(d) * eu[i] * wys;
          }
          gout[i] = nrm ? g : old[i] + g;
          dot += g * raw[i + 1];
        }
        if (q.gp[ch]) *reinterpret_cast<float4*>(q.gp[ch] + (size_t)b * q.bstride[ch] + p0) = make_float4(gout[0], gout[1], gout[2], gout[3]);
        const int e_of = q.ent[ch];
#pragma unroll
        for (int e = 0; e < SMA_MAX_ENTRIES; ++e) {
          const bool mine = e_of == e;
          acc[3 * e + 0] += mine ? sx : 0.f;
          acc[3 * e + 1] += mine ? sy : 0.f;
          acc[3 * e + 2] += (mine && nrm) ? dot : 0.f;
        }
      }
    }
  }
  const float r = block_sum_dpp<3 * SMA_MAX_ENTRIES, SM_NT>(acc, red);
#pragma unroll
  for (int e = 0; e < SMA_MAX_ENTRIES; ++e) {
    const int j = (int)threadIdx.x - 3 * e;
    if (j >= 0 && j < 3 && q.part[e]) q.part[e][((size_t)b * gx + bx) * 4 + j] = r;
  }
}

template <int NCH>
__global__ __launch_bounds__(SM_NT) void smooth_quad_kernel(const SmoothQuadArgs a) {
  int si = 0;
#pragma unroll
  for (int i = 1; i < DD_MAX_SCALES; ++i)
    if (i < a.num_scales && (int)blockIdx.x >= a.sc[i].first) si = i;
  // every a.sc[...] access below sits at a constant offset of the kernel-argument block (a run-time index would make the compiler
  // copy the block to scratch)
  switch (si) {
  // XCD-aware order: workgroup i runs on XCD i % 8 (own L2 each).  A workgroup covers 1024 consecutive pixels and reads the rows above
  // and below them: with the natural order those rows belong to workgroups on OTHER XCDs and every plane was fetched ~1.7x (PMC:
  // 156 MB fetched per launch for 93 MB of planes).  Each XCD gets a contiguous band of the scale's workgroups instead.
#define DD_SMQ_SCALE(I) case I: { const int vb = (int)blockIdx.x - a.sc[I].first; \
      const int vbr = (vb & 7) * (a.sc[I].npad >> 3) + (vb >> 3); \
      if (vbr < a.sc[I].gx * a.B) smooth_quad_body<NCH>(vbr % a.sc[I].gx, vbr / a.sc[I].gx, a.sc[I]); break; }
    DD_SMQ_SCALE(0) DD_SMQ_SCALE(1) DD_SMQ_SCALE(2) DD_SMQ_SCALE(3)
#undef DD_SMQ_SCALE
    default: break;
  }
}

// res[2k], res[2k+1] of one entry: fixed-order fold of its B*nblk block records (one workgroup)
__device__ __forceinline__ void smooth_fold_body(const float* __restrict__ partials, int count, float* __restrict__ sums) {
  __shared__ float red[2 * SM_NT / 64];
  float v[2] = {0.f, 0.f};
  for (int i = threadIdx.x; i < count; i += SM_NT) { v[0] += partials[(size_t)i * 4]; v[1] += partials[(size_t)i * 4 + 1]; }
  const float r = block_sum<2, SM_NT>(v, red);
  if (threadIdx.x < 2) sums[threadIdx.x] = r;
}

// Per-image scalars of the disparity finish, ONCE per image (one workgroup each, stage 3): the dot product and mean of the
// mean-normalisation adjoint, and the winning RANSAC plane -- the inlier counts of the image's max_it candidates are folded here from
// the scoring records (candidate j = b*max_it + it was scored on image j mod B: tools.py:130) and the first maximum taken.  Round 3
// had every workgroup of the finishing pass (60 per image at scale 0) redo the fold of the dot product, the mean and a 100-entry
// argmax behind four barriers before it touched a pixel; that prologue was most of the pass.  pre[b*8 ..]: dot, mean+eps, w1, w2, w3.
__device__ __forceinline__ void disp_pre_body(int b, const DDRegScale& sc, int nblk, bool normalised, bool ground, const float* __restrict__ mean,
                                              const float* __restrict__ sm_part, const float* __restrict__ cand, const int* __restrict__ cpart,
                                              int rec_per_img, int B, int max_it, float* __restrict__ pre) {
  __shared__ float red[GP_NT / 64];
  __shared__ int s_half[GP_MAX_IT];
  __shared__ unsigned long long s_key;
  const int n = sc.h * sc.w;
  float dot = 0.f, me = 1.f;
  if (normalised) {
    float v[1] = {0.f};
    for (int i = threadIdx.x; i < nblk; i += GP_NT) v[0] += sm_part[((size_t)b * nblk + i) * 4 + 2];
    const float r = block_sum<1, GP_NT>(v, red);       // valid in thread 0
    dot = r;
    me = plane_mean(mean, b, n) + 1e-7f;
    if (threadIdx.x == 0) { pre[b * PRE_STRIDE + 0] = dot; pre[b * PRE_STRIDE + 1] = me; }
  }
  if (!ground) return;               // uniform
  if (threadIdx.x == 0) s_key = 0ull;
  const int it = threadIdx.x & (GP_MAX_IT - 1), half = threadIdx.x / GP_MAX_IT;     // GP_NT == 2 * GP_MAX_IT
  int acc = 0;
  if (it < max_it) {
    const int j = b * max_it + it, img = j % B, k = j / B;
#pragma unroll 8
    for (int r = half; r < rec_per_img; r += 2) acc += cpart[((size_t)img * rec_per_img + r) * max_it + k];
  }
  if (half == 1) s_half[it] = acc;
  __syncthreads();
  if (half == 0 && it < max_it) {
    const unsigned long long key = (static_cast<unsigned long long>(static_cast<unsigned>(acc + s_half[it])) << 32) | (0xFFFFFFFFu - static_cast<unsigned>(it));
    atomicMax(&s_key, key);            // first maximum: the larger ~index wins among equal counts
  }
  __syncthreads();
  const int best = static_cast<int>(0xFFFFFFFFu - static_cast<unsigned>(s_key & 0xFFFFFFFFull));
  if (threadIdx.x < 3) {
    const float wv = cand[((size_t)b * max_it + best) * 3 + threadIdx.x];
    pre[b * PRE_STRIDE + 2 + threadIdx.x] = wv;
    sc.plane[b * 3 + threadIdx.x] = wv;
  }
}

// Last pass over the disparity gradient of a scale: the adjoint of the mean-normalisation of d_smooth (Trainer.py:357-359)
//   g_d += g_a / (m + eps) - (sum_p g_a[p] d[p]) / ((m + eps)^2 n)
// and the above-ground hinge (Trainer.py:361-364,425-461) in ONE read-modify-write (they were two passes over g_disp).
__device__ __forceinline__ void disp_finish_body(int bx, int b, int gx, const DDRegScale& sc, bool normalised, bool ground,
                                                 const float* __restrict__ pre, const float* __restrict__ g_tmp,
                                                 float* __restrict__ g_norm /* sm.g_inp or nullptr */, float tol,
                                                 float max_depth, DepthParams dp, float* __restrict__ hinge_part) {
  __shared__ float red[GP_NT / 64];
  const int h = sc.h, w = sc.w, n = h * w;
  // the image's scalars from disp_pre_body (wave-uniform loads: no fold, no barrier in front of the pixels)
  float me = 1.f, dot = 0.f;
  if (normalised && g_norm) { dot = pre[b * PRE_STRIDE + 0]; me = pre[b * PRE_STRIDE + 1]; }
  float w1 = 0.f, w2 = 0.f, w3 = 0.f;
  if (ground) { w1 = pre[b * PRE_STRIDE + 2]; w2 = pre[b * PRE_STRIDE + 3]; w3 = pre[b * PRE_STRIDE + 4] + tol; }      // Trainer.py:437-438
  float* g_disp = ground ? sc.g_disp : g_norm;         // the same buffer when both are active (checked by the planner)
  float v[1] = {0.f};
#pragma unroll 2
  for (int it = 0; it < FIN_PXT; ++it) {
    const int p = (bx * FIN_PXT + it) * GP_NT + threadIdx.x;
    if (p >= n) break;
    const size_t i = (size_t)b * n + p;
    float g0 = g_disp ? g_disp[i] : 0.f;
    const float gt = (normalised && g_norm) ? g_tmp[i] : 0.f;
    const float dv = ground ? sc.disp[i] : 0.f;
    if (normalised && g_norm) g0 += gt / me - dot / (me * me * static_cast<float>(n));
    if (ground) {
      const int y = p / w, x = p - y * w;
      const float* A = sc.inv_K + b * 16;
      float ray[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) ray[k] = A[k * 4 + 0] * static_cast<float>(x) + A[k * 4 + 1] * static_cast<float>(y) + A[k * 4 + 2];
      float gd = w3 / (ray[1] - ray[0] * w1 - ray[2] * w2);
      const bool invalid = (gd < 0.f) || (gd > max_depth);        // NaN compares false -> stays, like the reference
      if (invalid) gd = max_depth;
      if (gd != max_depth) {
        const float gdisp = (1.f / gd - dp.lo) / dp.span;
        const float diff = dv - gdisp;
        if (!(diff > 0.f)) {                                       // disp_diff[disp_diff > 0] = 0
          v[0] += diff;
          g0 += sc.w_ground;
        }
      }
    }
    if (g_disp) g_disp[i] = g0;
  }
  if (ground) {
    const float r = block_sum<1, GP_NT>(v, red);
    if (threadIdx.x == 0) hinge_part[(size_t)b * gx + bx] = r;
  }
}

// =================================================================================================
// dd_fused_loss (round 5): the passes behind the photometric tile kernel (dd_fuse.h)
// =================================================================================================
// One scale of the pass that finishes the low-res gradients (scales >= 1): a thread owns four CONSECUTIVE low-res pixels of one image,
// adds up the <= 4 tile footprints that overlap each (same order as photo_combine_kernel: bit-identical photometric gradients) and
// evaluates the edge-aware smoothness of the scale's channels on the same quad (the arithmetic of smooth_quad_body, disparity
// un-normalised: see photo_tile_kernel) -- the smoothness gradient is added BEFORE the quad's one 16-byte store instead of in a
// read-modify-write pass of its own.  Channels: 0 disparity | 1..3 flow | 4 mask (NCH = 1 | 4 | 5).
struct CombineScale {
  const float* img;                  // (B,3,h,w) pyramid level
  const float* fp;                   // this scale's footprint area of the tile kernel's workspace
  const float* in[5];                // input plane of image 0
  float* gp[5];                      // gradient plane of image 0 (stored here)
  float* g_tmp;                      // (B,n) d/d(normalised disparity), or nullptr when the disparity is not smoothed
  float* part;                       // block records [(b*gx + bx) * 8 + j], j < NSMOOTH
  int bstride[5];
  float wx[3], wy[3];                // per smoothness group (disparity | flow | mask); 0 = off
  int h, w, shift, gx, first, npad;  // gx workgroups per image; first workgroup inside the task range (a multiple of 8); count padded to 8
};
struct CombineArgs {
  CombineScale sc[DD_MAX_SCALES];
  int B, tiles_x, tiles_y;
};

template <int NCH>
__device__ __forceinline__ void combine_smooth_body(int bx, int b, const CombineScale& q, int tiles_x, int tiles_y) {
  __shared__ float red[NSMOOTH * SM_NT / 64];
  const int h = q.h, w = q.w, n = h * w, gx = q.gx, shift = q.shift;
  float acc[NSMOOTH];
#pragma unroll
  for (int i = 0; i < NSMOOTH; ++i) acc[i] = 0.f;
  const int p0 = (bx * SM_NT + (int)threadIdx.x) * 4;
  if (p0 < n) {
    const int y = p0 / w, x0 = p0 - y * w;
    // ---- footprint sums of the quad as straight-line code: every load of the thread is in flight at once (the tile walk as nested
    // run-time loops left one dependent round trip per tap: 20 us for this task).  A low-res pixel lies inside ONE tile and, on the
    // tile's first / last row or column, also on the rim of the neighbour's footprint: at most two taps per axis.  The quad's four
    // pixels share the row taps and the main column tile (quads are aligned, tile widths are multiples of four): per row tap one
    // unaligned 16-byte load of the main tile plus the left neighbour's rim for pixel 0 and the right neighbour's for pixel 3.
    // Same order of additions as photo_combine_kernel (tile rows ascending, tile columns ascending): bit-identical sums -- the taps
    // left out here are the columns / rows of the border tiles' footprints that no pixel maps to (exact zeros).
    float gsum[NCH][4];
    {
      static_assert(TW == 32 && TH == 16, "tile shifts below");
      const int sw = 5 - shift, sh = 4 - shift;                      // log2 of the low-res tile width / height
      const int lrh = TH >> shift, lrw = TW >> shift, fph = lrh + 2, fpw = lrw + 2, fpn = fph * fpw;
      const int ty0 = y >> sh, tx0 = x0 >> sw;
      const bool up = ((y & (lrh - 1)) == 0) && ty0 > 0, dn = ((y & (lrh - 1)) == lrh - 1) && ty0 + 1 < tiles_y;
      const bool lft = ((x0 & (lrw - 1)) == 0) && tx0 > 0, rgt = (((x0 + 3) & (lrw - 1)) == lrw - 1) && tx0 + 1 < tiles_x;
      const int tyA = up ? ty0 - 1 : ty0, tyB = up ? ty0 : ty0 + 1;  // first / second row tap (the second exists with up || dn)
      const bool hasB = up || dn;
      auto row_off = [&](int ty) { return ((b * tiles_y + ty) * tiles_x * NCH) * fpn + (y - max((ty << sh) - 1, 0)) * fpw; };
      const int rA = row_off(tyA), rB = hasB ? row_off(tyB) : rA;
      const int cM = tx0 * NCH * fpn + (x0 - max((tx0 << sw) - 1, 0));
      const int cL = lft ? (tx0 - 1) * NCH * fpn + (x0 - max(((tx0 - 1) << sw) - 1, 0)) : cM;
      const int cR = rgt ? (tx0 + 1) * NCH * fpn : cM;              // column 0 of the right neighbour's footprint
      typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
      f4u mA[NCH], mB[NCH];
      float lA[NCH], rAv[NCH], lB[NCH], rBv[NCH];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        const float* base = q.fp + (size_t)ch * fpn;
        mA[ch] = *reinterpret_cast<const f4u*>(base + rA + cM);
        mB[ch] = *reinterpret_cast<const f4u*>(base + rB + cM);
        lA[ch] = base[rA + cL]; rAv[ch] = base[rA + cR]; lB[ch] = base[rB + cL]; rBv[ch] = base[rB + cR];
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
        if (lft) g0 += lA[ch];
        g0 += mA[ch].x; g1 += mA[ch].y; g2 += mA[ch].z; g3 += mA[ch].w;
        if (rgt) g3 += rAv[ch];
        if (hasB) {
          if (lft) g0 += lB[ch];
          g0 += mB[ch].x; g1 += mB[ch].y; g2 += mB[ch].z; g3 += mB[ch].w;
          if (rgt) g3 += rBv[ch];
        }
        gsum[ch][0] = g0; gsum[ch][1] = g1; gsum[ch][2] = g2; gsum[ch][3] = g3;
      }
    }
    auto st4 = [](float* ptr, const float (&g)[4]) { *reinterpret_cast<float4*>(ptr) = make_float4(g[0], g[1], g[2], g[3]); };
    const bool any_smooth = q.wx[0] != 0.f || q.wx[1] != 0.f || q.wx[2] != 0.f;       // uniform
    if (!any_smooth) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) st4(q.gp[ch] + (size_t)b * q.bstride[ch] + p0, gsum[ch]);
    } else {
      const bool has_l = x0 > 0, has_r = x0 + 4 < w, has_u = y > 0, has_d = y + 1 < h;
      const int pu = has_u ? p0 - w : p0, pd = has_d ? p0 + w : p0, pl = has_l ? p0 - 1 : p0, pr = has_r ? p0 + 4 : p0 + 3;
      auto ld4 = [](const float* ptr) -> float4 { return *reinterpret_cast<const float4*>(ptr); };
      float dh[5] = {0.f, 0.f, 0.f, 0.f, 0.f}, dd[4] = {0.f, 0.f, 0.f, 0.f}, du[4] = {0.f, 0.f, 0.f, 0.f};
      {
        const float* im = q.img + (size_t)b * 3 * n;
        float4 C4[3], U4[3], D4[3];
        float cl[3], cr[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          C4[ch] = ld4(im + ch * n + p0); U4[ch] = ld4(im + ch * n + pu); D4[ch] = ld4(im + ch * n + pd);
          cl[ch] = im[ch * n + pl]; cr[ch] = im[ch * n + pr];
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const float c[6] = {cl[ch], C4[ch].x, C4[ch].y, C4[ch].z, C4[ch].w, cr[ch]};
          const float u[4] = {U4[ch].x, U4[ch].y, U4[ch].z, U4[ch].w}, d[4] = {D4[ch].x, D4[ch].y, D4[ch].z, D4[ch].w};
#pragma unroll
          for (int k = 0; k < 5; ++k) dh[k] += dd_abs(c[k] - c[k + 1]);
#pragma unroll
          for (int i = 0; i < 4; ++i) { dd[i] += dd_abs(c[i + 1] - d[i]); du[i] += dd_abs(u[i] - c[i + 1]); }
        }
      }
      float eh[5], ed[4], eu[4];
#pragma unroll
      for (int k = 0; k < 5; ++k) eh[k] = __expf(-dh[k] / 3.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) { ed[i] = __expf(-dd[i] / 3.f); eu[i] = __expf(-du[i] / 3.f); }
      constexpr int GROUP = 3;
#pragma unroll
      for (int c0 = 0; c0 < NCH; c0 += GROUP) {
        float4 A4[GROUP], AU4[GROUP], AD4[GROUP];
        float al[GROUP], ar[GROUP];
#pragma unroll
        for (int j = 0; j < GROUP; ++j) {
          const int ch = c0 + j < NCH ? c0 + j : NCH - 1;
          const float* src = q.in[ch] + (size_t)b * q.bstride[ch];
          A4[j] = ld4(src + p0); AU4[j] = ld4(src + pu); AD4[j] = ld4(src + pd);
          al[j] = src[pl]; ar[j] = src[pr];
        }
#pragma unroll
        for (int j = 0; j < GROUP; ++j) {
          if (c0 + j >= NCH) continue;           // compile-time
          const int ch = c0 + j;
          const int grp = ch == 0 ? 0 : (ch < 4 ? 1 : 2);
          const float wxs = q.wx[grp], wys = q.wy[grp];
          float (&gph)[4] = gsum[ch];            // the photometric gradient of the quad
          if (wxs != 0.f) {                      // uniform: the group is smoothed in this phase
            const float v[6] = {al[j], A4[j].x, A4[j].y, A4[j].z, A4[j].w, ar[j]};
            const float up[4] = {AU4[j].x, AU4[j].y, AU4[j].z, AU4[j].w}, dn[4] = {AD4[j].x, AD4[j].y, AD4[j].z, AD4[j].w};
            float hd[5];                         // hd[k] = v[k] - v[k+1]: right term of pixel k-1, left term of pixel k
#pragma unroll
            for (int k = 0; k < 5; ++k) hd[k] = v[k] - v[k + 1];
            float gout[4], sx = 0.f, sy = 0.f, dot = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const bool r_ok = i < 3 || has_r, l_ok = i > 0 || has_l;
              float g = 0.f;
              if (r_ok) { sx += dd_abs(hd[i + 1]) * eh[i + 1]; g += dd_sign(hd[i + 1]) * eh[i + 1] * wxs; }
              if (l_ok) g -= dd_sign(hd[i]) * eh[i] * wxs;
              {
                const float d = v[i + 1] - dn[i];
                if (has_d) { sy += dd_abs(d) * ed[i]; g += dd_sign(d) * ed[i] * wys; }
              }
              {
                const float d = up[i] - v[i + 1];
                if (has_u) g -= dd_sign(d) * eu[i] * wys;
              }
              gout[i] = g;
              dot += g * v[i + 1];
            }
            if (ch == 0) {
              st4(q.g_tmp + (size_t)b * n + p0, gout);
              acc[0] += sx; acc[1] += sy; acc[2] += dot;
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) gph[i] += gout[i];
              acc[grp == 1 ? 3 : 5] += sx; acc[grp == 1 ? 4 : 6] += sy;
            }
          }
          st4(q.gp[ch] + (size_t)b * q.bstride[ch] + p0, gph);
        }
      }
    }
  }
  const float r = block_sum_dpp<NSMOOTH, SM_NT>(acc, red);
  if ((int)threadIdx.x < NSMOOTH) q.part[((size_t)b * gx + bx) * 8 + threadIdx.x] = r;
}

// fold of the tile kernel's block records (photo_finalize_kernel of dd_photo.hip as a task): workgroups [0,S) produce sums[s][*],
// workgroups [S, S+B) the pose gradients of image b.  Same order of additions as photo_finalize_kernel.
struct TileFoldArgs {
  const float* partials;
  float* sums;
  float* g_T[2];
  int S, B, tiles;
};
__device__ __forceinline__ void tile_fold_body(int vb, const TileFoldArgs& f) {
  __shared__ float red[4 * 24];
  if (vb >= f.S + f.B) return;         // (a debug build with the task switched off)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int S_ = f.S, B = f.B, tiles = f.tiles;
  float acc[24];
#pragma unroll
  for (int k = 0; k < 24; ++k) acc[k] = 0.f;
  int nvals;
  if (vb < S_) {
    nvals = 6;
    for (int i = tid; i < B * tiles; i += 256) {
      const float* rec = f.partials + ((size_t)vb * B * tiles + i) * DD_PARTIAL_STRIDE;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc[k] += rec[k];
    }
  } else {
    nvals = 24;
    const int b = vb - S_;
    for (int i = tid; i < S_ * tiles; i += 256) {
      const int s = i / tiles, t = i % tiles;
      const float* rec = f.partials + (((size_t)s * B + b) * tiles + t) * DD_PARTIAL_STRIDE + 6;
#pragma unroll
      for (int k = 0; k < 24; ++k) acc[k] += rec[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const float r = wsum_dpp(acc[k]);
    if (lane == 0) red[wave * 24 + k] = r;
  }
  __syncthreads();
  if (tid < nvals) {
    const float r = red[tid] + red[24 + tid] + red[48 + tid] + red[72 + tid];
    if (vb < S_) {
      const int map[6] = {0, 5, 1, 2, 3, 4};      // record order -> sums order (photo, cons0, cons1, delta0, delta1, n_warp)
      f.sums[vb * DD_SUMS_STRIDE + map[tid]] = r;
    } else if (f.g_T[0]) {
      float* dst = (tid < 12 ? f.g_T[0] : f.g_T[1]) + (vb - S_) * 16;
      dst[tid % 12] = r;
    }
  }
  if (vb >= S_ && f.g_T[0] && tid < 8) (tid < 4 ? f.g_T[0] : f.g_T[1])[(vb - S_) * 16 + 12 + (tid & 3)] = 0.f;
  if (vb < S_ && tid >= 6 && tid < DD_SUMS_STRIDE) f.sums[vb * DD_SUMS_STRIDE + tid] = 0.f;
}

// second launch of dd_fused_loss.  Task ranges: footprint sums + smoothness (scales >= 1; memory-bound) | tile-record fold |
// candidate scoring (all scales; matrix pipe + VALU) | per-image disparity sums.  The memory-bound workgroups are dispatched FIRST:
// with the scoring in front its ~750 workgroups took every slot of the chip and the rest of the launch ran behind them (38 us for
// 19 + 11 + 7 us of tasks).
struct PostScore {
  ScoreScale sc[DD_MAX_SCALES];
  const int32_t* rand_idx[DD_MAX_SCALES];
  float* cand_out[DD_MAX_SCALES];
  int num_scales, B, max_it, np;
  float tol;
  DepthParams dp;
};
struct PostMean {
  const float* inp[DD_MAX_SCALES];   // the scale's disparity (nullptr: no mean needed)
  float* partial[DD_MAX_SCALES];
  int n[DD_MAX_SCALES], first[DD_MAX_SCALES];
};
struct PostArgs {
  PostScore score;
  CombineArgs comb;
  TileFoldArgs fold;
  PostMean mean;
  int num_scales, first_fold, first_score, first_mean;      // combine tasks start at workgroup 0
};

template <int NCH>
__global__ __launch_bounds__(RT_NT_FUSED) void fused_post_kernel(const PostArgs a) {
  const int blk = (int)blockIdx.x;
  if (blk < a.first_fold) {
    int si = 0;
#pragma unroll
    for (int i = 1; i < DD_MAX_SCALES; ++i)
      if (i < a.num_scales && a.comb.sc[i].gx > 0 && blk >= a.comb.sc[i].first) si = i;
    switch (si) {
    // XCD band remap as in smooth_quad_kernel: a workgroup reads the rows above and below its own 1024 pixels
#define DD_PC_SCALE(I) case I: { const int vb = blk - a.comb.sc[I].first; \
      const int vbr = (vb & 7) * (a.comb.sc[I].npad >> 3) + (vb >> 3); \
      if (a.comb.sc[I].gx > 0 && vbr < a.comb.sc[I].gx * a.comb.B) combine_smooth_body<NCH>(vbr % a.comb.sc[I].gx, vbr / a.comb.sc[I].gx, a.comb.sc[I], a.comb.tiles_x, a.comb.tiles_y); break; }
      DD_PC_SCALE(0) DD_PC_SCALE(1) DD_PC_SCALE(2) DD_PC_SCALE(3)
#undef DD_PC_SCALE
      default: break;
    }
  } else if (blk < a.first_score) {
    tile_fold_body(blk - a.first_fold, a.fold);
  } else if (blk < a.first_mean) {
    const int sb = blk - a.first_score;
    int si = 0;
#pragma unroll
    for (int i = 1; i < DD_MAX_SCALES; ++i)
      if (i < a.num_scales && a.score.sc[i].gx > 0 && sb >= a.score.sc[i].first) si = i;
    switch (si) {       // constant offsets into the kernel-argument block
#define DD_PS_SCALE(I) case I: { const int vb = sb - a.score.sc[I].first; \
      if (a.score.sc[I].gx > 0) ground_score_mfma_body(vb % a.score.sc[I].gx, vb / a.score.sc[I].gx, a.score.sc[I].gx, a.score.sc[I].disp, a.score.sc[I].inv_K, a.score.sc[I].cand, a.score.B, \
                             a.score.sc[I].h, a.score.sc[I].w, a.score.sc[I].rows, a.score.max_it, a.score.tol, a.score.dp, a.score.sc[I].part, \
                             a.score.rand_idx[I], a.score.np, a.score.cand_out[I]); break; }
      DD_PS_SCALE(0) DD_PS_SCALE(1) DD_PS_SCALE(2) DD_PS_SCALE(3)
#undef DD_PS_SCALE
      default: break;
    }
  } else {
    const int mb = blk - a.first_mean;
    int si = 0;
#pragma unroll
    for (int i = 1; i < DD_MAX_SCALES; ++i)
      if (i < a.num_scales && a.mean.inp[i] && mb >= a.mean.first[i]) si = i;
    switch (si) {
#define DD_PM_SCALE(I) case I: { const int vb = mb - a.mean.first[I]; \
      if (a.mean.inp[I]) plane_sum_body(vb % MEAN_BPI, vb / MEAN_BPI, MEAN_BPI, a.mean.inp[I], a.mean.n[I], a.mean.partial[I]); break; }
      DD_PM_SCALE(0) DD_PM_SCALE(1) DD_PM_SCALE(2) DD_PM_SCALE(3)
#undef DD_PM_SCALE
      default: break;
    }
  }
}

// dd_fused_loss: the sparsity term's two passes for frames that share ONE motion_prob tensor (what networks.Model publishes), both
// frames in the same pass over 16-byte quads, every load of a thread issued before the first is consumed.  Counting pass: a workgroup
// covers SP2_PXT * 256 consecutive pixels of one image and leaves one record {count_0, softplus sum_0, count_1, softplus sum_1}.
constexpr int SP2_PXT = 16;
__device__ __forceinline__ void sparsity_count2_body(int bx, int b, int gx, const DDRegScale& sc, int n, float inv_total, float* __restrict__ part) {
  __shared__ float red[4 * SP_NT / 64];
  const float thr0 = sc.delta_sum[0][0] * inv_total, thr1 = sc.delta_sum[1][0] * inv_total;       // disp_mag.mean() over the batch (Trainer.py:397)
  const float* d0 = sc.delta[0] + (size_t)b * n;
  const float* d1 = sc.delta[1] + (size_t)b * n;
  const float* pr = sc.prob[0] + (size_t)b * n;
  float4 A[SP2_PXT / 4], Bv[SP2_PXT / 4], X[SP2_PXT / 4];
#pragma unroll
  for (int it = 0; it < SP2_PXT / 4; ++it) {
    const int p0 = ((bx * (SP2_PXT / 4) + it) * SP_NT + (int)threadIdx.x) * 4;
    const int q0 = p0 < n ? p0 : 0;
    A[it] = *reinterpret_cast<const float4*>(d0 + q0); Bv[it] = *reinterpret_cast<const float4*>(d1 + q0); X[it] = *reinterpret_cast<const float4*>(pr + q0);
  }
  float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < SP2_PXT / 4; ++it) {
    const int p0 = ((bx * (SP2_PXT / 4) + it) * SP_NT + (int)threadIdx.x) * 4;
    if (p0 >= n) continue;
    const float a[4] = {A[it].x, A[it].y, A[it].z, A[it].w}, c[4] = {Bv[it].x, Bv[it].y, Bv[it].z, Bv[it].w}, x[4] = {X[it].x, X[it].y, X[it].z, X[it].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sp = softplus(x[i]);
      if (a[i] < thr0) { v[0] += 1.f; v[1] += sp; }
      if (c[i] < thr1) { v[2] += 1.f; v[3] += sp; }
    }
  }
  const float r = block_sum_dpp<4, SP_NT>(v, red);
  if (threadIdx.x < 4) part[((size_t)b * gx + bx) * 4 + threadIdx.x] = r;
}

// Gradient pass: prologue = fold of the counting records per image (wave w takes images w, w+4, ...; fixed order), the gate of
// Trainer.py:398 (every image keeps a static pixel) per frame; then d(mean softplus)/d prob of both frames in one plain 16-byte
// store per quad (motion_prob receives no other gradient: nothing to read-modify-write).  res[10+2f] value, res[11+2f] #static.
__device__ __forceinline__ void sparsity_grad2_body(int bx, int b, int gx, const DDRegScale& sc, int B, int n, float inv_total, const float* __restrict__ part,
                                                    int gxc, float* __restrict__ res) {
  __shared__ float s_w[SP_NT / 64][4];
  __shared__ int s_g[SP_NT / 64][2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
    float tc0 = 0.f, ts0 = 0.f, tc1 = 0.f, ts1 = 0.f;
    int g0 = 1, g1 = 1;
    for (int bi = wave; bi < B; bi += SP_NT / 64) {
      float c0 = 0.f, s0 = 0.f, c1 = 0.f, s1 = 0.f;
      for (int i = lane; i < gxc; i += 64) {
        const float4 r = *reinterpret_cast<const float4*>(part + ((size_t)bi * gxc + i) * 4);
        c0 += r.x; s0 += r.y; c1 += r.z; s1 += r.w;
      }
      c0 = wsum_dpp(c0); s0 = wsum_dpp(s0); c1 = wsum_dpp(c1); s1 = wsum_dpp(s1);
      if (c0 <= 0.f) g0 = 0;
      if (c1 <= 0.f) g1 = 0;
      tc0 += c0; ts0 += s0; tc1 += c1; ts1 += s1;
    }
    if (lane == 0) { s_w[wave][0] = tc0; s_w[wave][1] = ts0; s_w[wave][2] = tc1; s_w[wave][3] = ts1; s_g[wave][0] = g0; s_g[wave][1] = g1; }
  }
  __syncthreads();
  float cnt[2] = {0.f, 0.f}, sm[2] = {0.f, 0.f};
  bool gate[2] = {true, true};
#pragma unroll
  for (int wv = 0; wv < SP_NT / 64; ++wv) {
    cnt[0] += s_w[wv][0]; sm[0] += s_w[wv][1]; cnt[1] += s_w[wv][2]; sm[1] += s_w[wv][3];
    gate[0] = gate[0] && s_g[wv][0] != 0; gate[1] = gate[1] && s_g[wv][1] != 0;
  }
  if (bx == 0 && b == 0 && threadIdx.x < 2) {
    const int f = threadIdx.x;
    res[10 + 2 * f] = gate[f] ? sm[f] / cnt[f] : 0.f;
    res[11 + 2 * f] = cnt[f];
  }
  float* gp = sc.g_prob[0];
  if (!gp) return;
  const float k0 = gate[0] ? sc.w_sparsity[0] / cnt[0] : 0.f, k1 = gate[1] ? sc.w_sparsity[1] / cnt[1] : 0.f;
  const float thr0 = sc.delta_sum[0][0] * inv_total, thr1 = sc.delta_sum[1][0] * inv_total;
  const float* d0 = sc.delta[0] + (size_t)b * n;
  const float* d1 = sc.delta[1] + (size_t)b * n;
  const float* pr = sc.prob[0] + (size_t)b * n;
  float4 A[SP2_PXT / 4], Bv[SP2_PXT / 4], X[SP2_PXT / 4];
#pragma unroll
  for (int it = 0; it < SP2_PXT / 4; ++it) {
    const int p0 = ((bx * (SP2_PXT / 4) + it) * SP_NT + (int)threadIdx.x) * 4;
    const int q0 = p0 < n ? p0 : 0;
    A[it] = *reinterpret_cast<const float4*>(d0 + q0); Bv[it] = *reinterpret_cast<const float4*>(d1 + q0); X[it] = *reinterpret_cast<const float4*>(pr + q0);
  }
#pragma unroll
  for (int it = 0; it < SP2_PXT / 4; ++it) {
    const int p0 = ((bx * (SP2_PXT / 4) + it) * SP_NT + (int)threadIdx.x) * 4;
    if (p0 >= n) continue;
    const float a[4] = {A[it].x, A[it].y, A[it].z, A[it].w}, c[4] = {Bv[it].x, Bv[it].y, Bv[it].z, Bv[it].w}, x[4] = {X[it].x, X[it].y, X[it].z, X[it].w};
    float g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sg = 1.f / (1.f + expf(-x[i]));                      // d softplus = sigmoid
      g[i] = (a[i] < thr0 ? k0 * sg : 0.f) + (c[i] < thr1 ? k1 * sg : 0.f);
    }
    *reinterpret_cast<float4*>(gp + (size_t)b * n + p0) = make_float4(g[0], g[1], g[2], g[3]);
  }
}

// disp_finish_body on 16-byte quads: two quads per thread, every load issued before the first is consumed (same arithmetic per pixel,
// same block records)
__device__ __forceinline__ void disp_finish4_body(int bx, int b, int gx, const DDRegScale& sc, bool normalised, bool ground,
                                                  const float* __restrict__ pre, const float* __restrict__ g_tmp, float* __restrict__ g_norm, float tol,
                                                  float max_depth, DepthParams dp, float* __restrict__ hinge_part) {
  __shared__ float red[GP_NT / 64];
  const int h = sc.h, w = sc.w, n = h * w;
  float me = 1.f, dot = 0.f;
  const bool nrm = normalised && g_norm;
  if (nrm) { dot = pre[b * PRE_STRIDE + 0]; me = pre[b * PRE_STRIDE + 1]; }
  float w1 = 0.f, w2 = 0.f, w3 = 0.f;
  if (ground) { w1 = pre[b * PRE_STRIDE + 2]; w2 = pre[b * PRE_STRIDE + 3]; w3 = pre[b * PRE_STRIDE + 4] + tol; }      // Trainer.py:437-438
  float* g_disp = ground ? sc.g_disp : g_norm;
  const float inv_me = 1.f / me, shiftc = dot / (me * me * static_cast<float>(n));
  constexpr int Q = FIN_PXT / 4;
  float4 G[Q], T[Q], D[Q];
#pragma unroll
  for (int it = 0; it < Q; ++it) {
    const int p0 = ((bx * Q + it) * GP_NT + (int)threadIdx.x) * 4;
    const size_t i0 = (size_t)b * n + (p0 < n ? p0 : 0);
    G[it] = g_disp ? *reinterpret_cast<const float4*>(g_disp + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
    T[it] = nrm ? *reinterpret_cast<const float4*>(g_tmp + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
    D[it] = ground ? *reinterpret_cast<const float4*>(sc.disp + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float v[1] = {0.f};
  const float* A = sc.inv_K + b * 16;
#pragma unroll
  for (int it = 0; it < Q; ++it) {
    const int p0 = ((bx * Q + it) * GP_NT + (int)threadIdx.x) * 4;
    if (p0 >= n) continue;
    float g0[4] = {G[it].x, G[it].y, G[it].z, G[it].w};
    const float gt[4] = {T[it].x, T[it].y, T[it].z, T[it].w}, dv[4] = {D[it].x, D[it].y, D[it].z, D[it].w};
    const int y = p0 / w, x0 = p0 - y * w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (nrm) g0[i] += gt[i] / me - shiftc;
      if (ground) {
        const float xf = static_cast<float>(x0 + i), yf = static_cast<float>(y);
        float ray[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) ray[k] = A[k * 4 + 0] * xf + A[k * 4 + 1] * yf + A[k * 4 + 2];
        float gd = w3 / (ray[1] - ray[0] * w1 - ray[2] * w2);
        const bool invalid = (gd < 0.f) || (gd > max_depth);        // NaN compares false -> stays, like the reference
        if (invalid) gd = max_depth;
        if (gd != max_depth) {
          const float gdisp = (1.f / gd - dp.lo) / dp.span;
          const float diff = dv[i] - gdisp;
          if (!(diff > 0.f)) {                                       // disp_diff[disp_diff > 0] = 0
            v[0] += diff;
            g0[i] += sc.w_ground;
          }
        }
      }
    }
    if (g_disp) *reinterpret_cast<float4*>(g_disp + (size_t)b * n + p0) = make_float4(g0[0], g0[1], g0[2], g0[3]);
  }
  (void)inv_me;
  if (ground) {
    const float r = block_sum_dpp<1, GP_NT>(v, red);
    if (threadIdx.x == 0) hinge_part[(size_t)b * gx + bx] = r;
  }
}

// Per-image scalars of a scale for dd_fused_loss, ONE workgroup per (scale, image), third launch: the image's smoothness sums folded
// from the block records (scale 0: the tile kernel's records of the image's tiles; scales >= 1: combine_smooth_body's), the mean of
// its disparity, and -- like disp_pre_body -- the winning RANSAC plane.  The mean normalisation enters HERE: the disparity's sums
// were taken on the raw values, sum |d_p - d_q| e / (mean + eps) is what Trainer.py:357-359 + tools.py:311-326 evaluate.
struct FoldSource {
  const float* rec;       // first record of image 0
  int count;              // records per image
  int stride;             // floats between two records
  int img_stride;         // floats between two images' first records
  int base;               // index of sx_d inside a record
};
__device__ __forceinline__ void image_fold_body(int b, const DDRegScale& sc, const FoldSource fs, bool normalised, bool ground,
                                                const float* __restrict__ mean, const float* __restrict__ cand, const int* __restrict__ cpart,
                                                int rec_per_img, int B, int max_it, float* __restrict__ pre) {
  __shared__ float red[NSMOOTH * GP_NT / 64];
  __shared__ int s_half[GP_MAX_IT];
  __shared__ unsigned long long s_key;
  const int n = sc.h * sc.w;
  float v[NSMOOTH];
#pragma unroll
  for (int k = 0; k < NSMOOTH; ++k) v[k] = 0.f;
  if (fs.rec) {
    for (int i = threadIdx.x; i < fs.count; i += GP_NT) {
      const float* r = fs.rec + (size_t)b * fs.img_stride + (size_t)i * fs.stride + fs.base;
#pragma unroll
      for (int k = 0; k < NSMOOTH; ++k) v[k] += r[k];
    }
  }
  const float r = block_sum_dpp<NSMOOTH, GP_NT>(v, red);       // value k valid in thread k
  float me = 1.f;
  if (normalised) me = plane_mean(mean, b, n) + 1e-7f;           // (barrier inside: uniform condition)
  if ((int)threadIdx.x < NSMOOTH) {
    const int k = threadIdx.x;
    // [0] dot [1] mean+eps [5] sx_d/me [6] sy_d/me [7..10] flow / mask sums
    if (k == 2) { pre[b * PRE_STRIDE + 0] = r; pre[b * PRE_STRIDE + 1] = me; }
    else if (k < 2) pre[b * PRE_STRIDE + 5 + k] = r / me;
    else pre[b * PRE_STRIDE + 4 + k] = r;
  }
  if (!ground) return;               // uniform
  if (threadIdx.x == 0) s_key = 0ull;
  const int it = threadIdx.x & (GP_MAX_IT - 1), half = threadIdx.x / GP_MAX_IT;     // GP_NT == 2 * GP_MAX_IT
  int acc = 0;
  if (it < max_it) {
    const int j = b * max_it + it, img = j % B, k = j / B;
#pragma unroll 8
    for (int rr = half; rr < rec_per_img; rr += 2) acc += cpart[((size_t)img * rec_per_img + rr) * max_it + k];
  }
  if (half == 1) s_half[it] = acc;
  __syncthreads();
  if (half == 0 && it < max_it) {
    const unsigned long long key = (static_cast<unsigned long long>(static_cast<unsigned>(acc + s_half[it])) << 32) | (0xFFFFFFFFu - static_cast<unsigned>(it));
    atomicMax(&s_key, key);            // first maximum: the larger ~index wins among equal counts
  }
  __syncthreads();
  const int best = static_cast<int>(0xFFFFFFFFu - static_cast<unsigned>(s_key & 0xFFFFFFFFull));
  if (threadIdx.x < 3) {
    const float wv = cand[((size_t)b * max_it + best) * 3 + threadIdx.x];
    pre[b * PRE_STRIDE + 2 + threadIdx.x] = wv;
    sc.plane[b * 3 + threadIdx.x] = wv;
  }
}

// =================================================================================================
// all regularisers of all scales in up to five launches (dd_reg_losses)
// =================================================================================================
// The per-term bodies above run as TASKS of one stage kernel launched per stage: a task owns a contiguous range of workgroups of
// the launch and maps it onto a 2-D grid (x: blocks of a plane, y: image).  Stage 1: per-image disparity means, static-pixel
// counts, RANSAC candidates.  Stage 2 (needs stage 1): smoothness of ALL entries of a scale in one pass (value + gradient),
// sparsity gradient, candidate scoring.  Stage 3 (needs stage 2): smoothness sums per entry, inlier counts per candidate.
// Stage 4: ONE pass over the disparity gradient -- mean-normalisation adjoint + ground hinge.  Stage 5: fixed-order fold of the
// hinge partials; dd_reg_losses_finish folds them inside the assembling kernel instead (one launch less).  No global atomics.
// Round 3: stage 2 evaluated the edge weights once per smoothed CHANNEL (5 tasks per scale in fine_tune, 45 000 workgroups of
// ~30 loads each, 155 us) and the disparity gradient was rewritten twice (stages 3 and 4).
constexpr int RT_NT = 256;
static_assert(SM_NT == RT_NT && SP_NT == RT_NT && GP_NT == RT_NT, "one workgroup size for every task");
enum : int { K_MEAN = 0, K_SPCOUNT, K_GCAND, K_SMOOTHALL, K_SPGRAD, K_GSCORE, K_SMFOLD, K_GCOUNT, K_DISPFIN, K_GFOLD, K_DISPPRE, K_IMGFOLD,
              K_SPCOUNT2, K_SPGRAD2, K_DISPFIN4 };
constexpr int REG_MAX_TASKS = 32;

struct RegTask {
  int first;            // first workgroup of the task
  int gx;               // width of its virtual grid
  int gx2;              // K_GCOUNT: records per image
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
  long long g_cand[DD_MAX_SCALES], g_counts[DD_MAX_SCALES], g_part[DD_MAX_SCALES], g_cpart[DD_MAX_SCALES], d_pre[DD_MAX_SCALES];
  long long post_part[DD_MAX_SCALES];        // dd_fused_loss: combine_smooth_body's block records (scales >= 1)
  FoldSource fold[DD_MAX_SCALES];            // dd_fused_loss: where image_fold_body finds the scale's smoothness records
};

__global__ __launch_bounds__(RT_NT) void reg_stage_kernel(const DDRegArgs a, const RegOffsets off, const RegTasks tasks) {
  int ti = 0;
  for (int i = 1; i < tasks.n; ++i)
    if ((int)blockIdx.x >= tasks.t[i].first) ti = i;
  const RegTask t = tasks.t[ti];
  const int gx = t.gx;
  const int s = t.scale, k = t.idx;
  const DDRegScale& sc = a.scale[s];
  const int B = a.B, h = sc.h, w = sc.w, n = h * w;
  const int nblk_sm = (n + RT_NT * SMA_PXT - 1) / (RT_NT * SMA_PXT);          // smoothness records per image
  float* ws = a.workspace;
  float* res = a.res + s * DD_REG_RES_STRIDE;
  const float inv_total = 1.f / (static_cast<float>(B) * static_cast<float>(n));
  const int rows = static_cast<int>(a.g_prior * static_cast<float>(h));
  const DepthParams dp = depth_params(a.min_depth, a.max_depth);
  const int vb = (int)blockIdx.x - t.first, bx = vb % gx, by = vb / gx;
  switch (t.kind) {
    case K_MEAN:
      plane_sum_body(bx, by, gx, sc.smooth[k].inp, n, ws + off.mean[s]);
      break;
    case K_SPCOUNT:
      sparsity_count_body(bx, by, gx, sc.delta[k], sc.delta_sum[k], sc.prob[k], n, inv_total, ws + off.sp_part[s][k]);
      break;
    case K_GCAND:
      ground_candidates_body(bx, by, gx, sc.disp, sc.inv_K, sc.rand_idx, B, h, w, rows, a.np_per_it, a.max_it, dp, ws + off.g_cand[s],
                             reinterpret_cast<int*>(ws + off.g_counts[s]));
      break;
    case K_SMOOTHALL: {
      float* part[SMA_MAX_ENTRIES];
      int normalised = -1;
#pragma unroll
      for (int e = 0; e < SMA_MAX_ENTRIES; ++e) {
        part[e] = sc.smooth[e].inp ? ws + off.sm_part[s][e] : nullptr;
        if (sc.smooth[e].inp && sc.smooth[e].normalise) normalised = e;
      }
      const float* mean = normalised >= 0 ? ws + off.mean[s] : nullptr;
      float* g_tmp = normalised >= 0 ? ws + off.sm_gtmp[s][normalised] : nullptr;
      switch (k) {           // k = number of smoothed channels of the scale
#define DD_SMA_CASE(N) case N: smooth_all_body<N>(bx, by, gx, sc, B, mean, g_tmp, part); break;
        DD_SMA_CASE(1) DD_SMA_CASE(2) DD_SMA_CASE(3) DD_SMA_CASE(4) DD_SMA_CASE(5) DD_SMA_CASE(6) DD_SMA_CASE(7) DD_SMA_CASE(8) DD_SMA_CASE(9)
#undef DD_SMA_CASE
        default: break;
      }
      break;
    }
    case K_SPGRAD:
      // k == 2: both frames read one motion_prob tensor and accumulate into one gradient buffer (what networks.Model
      // publishes) -- the same thread then handles the element for frame 0 and frame 1 in turn; two tasks would race
      for (int f = (k == 2 ? 0 : k); f <= (k == 2 ? 1 : k); ++f)
        sparsity_grad_body<SPG_PXT>(bx, by, gx, sc.delta[f], sc.delta_sum[f], sc.prob[f], B, n, inv_total, sc.w_sparsity[f], ws + off.sp_part[s][f],
                                    sc.g_prob[f], res + 10 + 2 * f);
      break;
    case K_GSCORE:
      ground_score_body(bx, by, gx, sc.disp, sc.inv_K, ws + off.g_cand[s], B, h, w, rows, a.max_it, a.tol, dp,
                        reinterpret_cast<int*>(ws + off.g_counts[s]), reinterpret_cast<int*>(ws + off.g_cpart[s]));
      break;
    case K_GCOUNT:
      ground_count_body(by, t.gx2, reinterpret_cast<const int*>(ws + off.g_cpart[s]), B, a.max_it,
                        reinterpret_cast<int*>(ws + off.g_counts[s]));
      break;
    case K_SMFOLD:
      smooth_fold_body(ws + off.sm_part[s][k], B * nblk_sm, res + 2 * k);
      break;
    case K_DISPPRE:
    case K_DISPFIN4:
    case K_DISPFIN: {
      int normalised = -1;
#pragma unroll
      for (int e = 0; e < SMA_MAX_ENTRIES; ++e)
        if (sc.smooth[e].inp && sc.smooth[e].normalise) normalised = e;
      const bool nrm = normalised >= 0 && sc.smooth[normalised >= 0 ? normalised : 0].g_inp != nullptr, ground = sc.disp != nullptr;
      if (t.kind == K_DISPPRE)
        disp_pre_body(by, sc, nblk_sm, nrm, ground, nrm ? ws + off.mean[s] : nullptr, nrm ? ws + off.sm_part[s][normalised] : nullptr,
                      ground ? ws + off.g_cand[s] : nullptr, ground ? reinterpret_cast<const int*>(ws + off.g_cpart[s]) : nullptr, t.gx2, B, a.max_it,
                      ws + off.d_pre[s]);
      else if (t.kind == K_DISPFIN4)
        disp_finish4_body(bx, by, gx, sc, nrm, ground, ws + off.d_pre[s], nrm ? ws + off.sm_gtmp[s][normalised] : nullptr,
                          nrm ? sc.smooth[normalised].g_inp : nullptr, a.tol, a.max_depth, dp, ground ? ws + off.g_part[s] : nullptr);
      else
        disp_finish_body(bx, by, gx, sc, nrm, ground, ws + off.d_pre[s], nrm ? ws + off.sm_gtmp[s][normalised] : nullptr,
                         nrm ? sc.smooth[normalised].g_inp : nullptr, a.tol, a.max_depth, dp, ground ? ws + off.g_part[s] : nullptr);
      break;
    }
    case K_SPCOUNT2:
      sparsity_count2_body(bx, by, gx, sc, n, inv_total, ws + off.sp_part[s][0]);
      break;
    case K_SPGRAD2:
      sparsity_grad2_body(bx, by, gx, sc, B, n, inv_total, ws + off.sp_part[s][0], t.gx2, res);
      break;
    case K_IMGFOLD: {
      int normalised = -1;
#pragma unroll
      for (int e = 0; e < SMA_MAX_ENTRIES; ++e)
        if (sc.smooth[e].inp && sc.smooth[e].normalise) normalised = e;
      const bool nrm = normalised >= 0, ground = sc.disp != nullptr;
      image_fold_body(by, sc, off.fold[s], nrm, ground, nrm ? ws + off.mean[s] : nullptr, ground ? ws + off.g_cand[s] : nullptr,
                      ground ? reinterpret_cast<const int*>(ws + off.g_cpart[s]) : nullptr, t.gx2, B, a.max_it, ws + off.d_pre[s]);
      break;
    }
    case K_GFOLD: {
      // fixed-order fold of the hinge partials (a launch of its own: a last-workgroup-done counter costs one contended
      // device-scope atomic per workgroup -- 7 560 of them took longer than the hinge pass itself)
      __shared__ float red[RT_NT / 64];
      const float* partials = ws + off.g_part[s];
      float v[1] = {0.f};
      for (int i = threadIdx.x; i < ((n + RT_NT * FIN_PXT - 1) / (RT_NT * FIN_PXT)) * B; i += RT_NT) v[0] += partials[i];
      const float r = block_sum<1, RT_NT>(v, red);
      if (threadIdx.x == 0) res[14] = r;
      break;
    }
    default:
      break;
  }
}

constexpr int REG_STAGES = 5;
struct RegPlan {
  RegOffsets off;
  RegTasks stage[REG_STAGES];
  int blocks[REG_STAGES];
  size_t floats;
  int quad_nch;            // > 0: the smoothness of every scale runs in smooth_quad_kernel<quad_nch> (not as stage-2 tasks)
  int quad_blocks;
  SmoothQuadArgs quad;
  int score_blocks;        // > 0: the candidate scoring of every scale runs in ground_score_all_kernel (matrix pipe)
  ScoreArgs score;
  long long pre_all;       // float offset of the per-image records of all scales ([scale][image][PRE_STRIDE])
};

// does the smoothness of this launch qualify for smooth_quad_kernel?  Every scale that smooths anything must smooth the same number
// of channels (1 | 3 | 4 | 5: what the four phases produce with shared tensors) in rows of whole, 16-byte aligned quads.
static int quad_channels(const DDRegArgs& a) {
#ifdef DD_REG_NO_QUAD
  return 0;
#endif
  auto aligned = [](const void* q) { return (reinterpret_cast<unsigned long long>(q) & 15ull) == 0; };
  int common = 0;
  for (int s = 0; s < a.num_scales; ++s) {
    const DDRegScale& sc = a.scale[s];
    int nch = 0;
    bool ok = (sc.w % 4 == 0) && aligned(sc.img) && aligned(a.workspace);
    for (int k = 0; k < DD_REG_SMOOTH; ++k) {
      if (!sc.smooth[k].inp) continue;
      nch += sc.smooth[k].C;
      ok = ok && aligned(sc.smooth[k].inp) && aligned(sc.smooth[k].g_inp);
    }
    if (nch == 0) continue;
    if (!ok || (common && nch != common)) return 0;
    common = nch;
  }
  return (common == 1 || common == 3 || common == 4 || common == 5) ? common : 0;
}

static int reg_plan(const DDRegArgs& a, RegPlan& p) {
  size_t total = 0;
  auto take = [&](size_t nfloats) { const size_t o = total; total += (nfloats + 63) / 64 * 64; return (long long)o; };
  for (int st = 0; st < REG_STAGES; ++st) { p.stage[st].n = 0; p.blocks[st] = 0; }
  p.quad_nch = quad_channels(a);
  p.quad_blocks = 0;
  memset(&p.quad, 0, sizeof(p.quad));
  p.quad.num_scales = a.num_scales;
  p.quad.B = a.B;
  p.score_blocks = 0;
  memset(&p.score, 0, sizeof(p.score));
  p.score.num_scales = a.num_scales; p.score.B = a.B; p.score.max_it = a.max_it; p.score.tol = a.tol;
  p.score.dp = depth_params(a.min_depth, a.max_depth);
  auto add = [&](int st, int kind, int s, int idx, int gx, int gy, int gx2 = 0) -> int {
    RegTasks& T = p.stage[st];
    if (T.n >= REG_MAX_TASKS) return 1;
    RegTask& t = T.t[T.n++];
    t.first = p.blocks[st]; t.gx = gx; t.gx2 = gx2; t.kind = (short)kind; t.scale = (signed char)s; t.idx = (signed char)idx;
    p.blocks[st] += gx * gy;
    return 0;
  };
  int bad = 0;
  const long long pre_all = take((size_t)DD_MAX_SCALES * a.B * PRE_STRIDE);      // per-image records, scale-major, one block
  p.pre_all = pre_all;
  for (int s = 0; s < a.num_scales; ++s) {
    const DDRegScale& sc = a.scale[s];
    if (sc.h < 2 || sc.w < 2) return 1;
    p.off.post_part[s] = take((size_t)a.B * ((sc.h * sc.w + RT_NT * 4 - 1) / (RT_NT * 4)) * 8);
    p.off.d_pre[s] = pre_all + (long long)s * a.B * PRE_STRIDE;
    const int n = sc.h * sc.w;
    const int nblk_sm = (n + RT_NT * SMA_PXT - 1) / (RT_NT * SMA_PXT), nblk_spg = (n + RT_NT * SPG_PXT - 1) / (RT_NT * SPG_PXT),
              nblk_fin = (n + RT_NT * FIN_PXT - 1) / (RT_NT * FIN_PXT);
    int normalised = -1, nch = 0;
    for (int k = 0; k < DD_REG_SMOOTH; ++k) {
      const DDRegSmooth& sm = sc.smooth[k];
      if (!sm.inp) continue;
      if (!sc.img || sm.C < 1 || (sm.normalise && sm.C != 1)) return 1;
      nch += sm.C;
      if (sm.normalise) {
        if (normalised >= 0) return 1;                    // one mean buffer per scale
        normalised = k;
        p.off.mean[s] = take((size_t)a.B * MEAN_BPI);
        p.off.sm_gtmp[s][k] = take((size_t)a.B * n);
        bad |= add(0, K_MEAN, s, k, MEAN_BPI, a.B);
        // the finishing pass writes the ground hinge into the same gradient buffer
        if (sc.disp && sc.g_disp && sm.g_inp && sc.g_disp != sm.g_inp) return 1;
      }
      p.off.sm_part[s][k] = take((size_t)a.B * nblk_sm * 4);
      bad |= add(2, K_SMFOLD, s, k, 1, 1);
    }
    if (nch > SMA_MAX_CH) return 1;
    for (int k = 0; k < DD_REG_SMOOTH; ++k)
      if (sc.smooth[k].inp && sc.smooth[k].C > 3) return 1;        // the channel table of smooth_all_body
    p.quad.sc[s].first = p.quad_blocks;
    p.score.sc[s].first = p.score_blocks;
    if (nch > 0 && p.quad_nch > 0) {
      // the scale's channel table, resolved here: entries in ascending order, channels of an entry in order (smooth_all_body derives
      // the same table on the device)
      SmoothQuadScale& q = p.quad.sc[s];
      q.img = sc.img; q.h = sc.h; q.w = sc.w; q.gx = nblk_sm;
      q.mean = normalised >= 0 ? a.workspace + p.off.mean[s] : nullptr;
      int ch = 0;
      for (int k = 0; k < DD_REG_SMOOTH; ++k) {
        const DDRegSmooth& sm = sc.smooth[k];
        q.part[k] = sm.inp ? a.workspace + p.off.sm_part[s][k] : nullptr;
        if (!sm.inp) continue;
        const float cnt = static_cast<float>(a.B) * static_cast<float>(sm.C);
        const float wx = sm.weight / (cnt * sc.h * (sc.w - 1)), wy = sm.weight / (cnt * (sc.h - 1) * sc.w);
        for (int c = 0; c < sm.C; ++c, ++ch) {
          q.in[ch] = sm.inp + (size_t)c * n;
          q.bstride[ch] = sm.C * n;
          q.gp[ch] = sm.g_inp ? (sm.normalise ? a.workspace + p.off.sm_gtmp[s][k] : sm.g_inp + (size_t)c * n) : nullptr;
          q.wx[ch] = wx; q.wy[ch] = wy;
          q.ent[ch] = (signed char)k; q.nrm[ch] = (signed char)(sm.normalise != 0);
        }
      }
      q.npad = (nblk_sm * a.B + 7) / 8 * 8;
      p.quad_blocks += q.npad;
    } else if (nch > 0) {
      bad |= add(1, K_SMOOTHALL, s, nch, nblk_sm, a.B);
    }
    if ((normalised >= 0 && sc.smooth[normalised].g_inp) || sc.disp) {
      const int rows_g = sc.disp ? (int)(a.g_prior * (float)sc.h) : 0;
      const int score_rec = sc.disp ? (rows_g * sc.w + GS_SLABS * RT_NT - 1) / (GS_SLABS * RT_NT) : 0;
      p.off.d_pre[s] = pre_all + (long long)s * a.B * PRE_STRIDE;
      bad |= add(2, K_DISPPRE, s, 0, 1, a.B, score_rec);          // the image's scalars once (stage 3) ...
      bad |= add(3, K_DISPFIN, s, 0, nblk_fin, a.B);              // ... for the pixel pass (stage 4)
    }
    const bool shared_prob = sc.prob[0] && sc.prob[0] == sc.prob[1];
    for (int f = 0; f < DD_NUM_SRC; ++f) {
      if (!sc.prob[f]) continue;
      if (!sc.delta[f] || !sc.delta_sum[f]) return 1;
      p.off.sp_part[s][f] = take((size_t)a.B * (SP_BPI * 2 > ((n + SP_NT * SP2_PXT - 1) / (SP_NT * SP2_PXT)) * 4 ? SP_BPI * 2 : ((n + SP_NT * SP2_PXT - 1) / (SP_NT * SP2_PXT)) * 4));
      bad |= add(0, K_SPCOUNT, s, f, SP_BPI, a.B);
      if (!shared_prob) bad |= add(1, K_SPGRAD, s, f, nblk_spg, a.B);
    }
    if (shared_prob) bad |= add(1, K_SPGRAD, s, 2, nblk_spg, a.B);
    if (sc.disp) {
      if (!sc.inv_K || !sc.rand_idx || !sc.plane || a.max_it < 1 || a.max_it > GP_MAX_IT) return 1;
      const int rows = (int)(a.g_prior * (float)sc.h);
      if (rows < 1) return 1;
      p.off.g_cand[s] = take((size_t)a.B * a.max_it * 3);
      p.off.g_counts[s] = take((size_t)a.B * a.max_it);
      p.off.g_part[s] = take((size_t)a.B * nblk_fin);
      const int score_blocks = (rows * sc.w + GS_SLABS * RT_NT - 1) / (GS_SLABS * RT_NT);
      p.off.g_cpart[s] = take((size_t)a.B * score_blocks * a.max_it);
      bad |= add(0, K_GCAND, s, 0, (a.B * a.max_it + RT_NT - 1) / RT_NT, 1);
#ifdef DD_REG_SCORE_VALU
      bad |= add(1, K_GSCORE, s, 0, score_blocks, a.B);
#else
      {
        ScoreScale& q = p.score.sc[s];
        q.disp = sc.disp; q.inv_K = sc.inv_K; q.cand = a.workspace + p.off.g_cand[s];
        q.part = reinterpret_cast<int*>(a.workspace + p.off.g_cpart[s]);
        q.h = sc.h; q.w = sc.w; q.rows = rows; q.gx = score_blocks; q.first = p.score_blocks;
        p.score_blocks += score_blocks * a.B;
      }
#endif
      bad |= add(4, K_GFOLD, s, 0, 1, 1);
    }
  }
  p.floats = total;
  return bad;
}

}  // namespace dd

using namespace dd;

static inline int last_error() { return (int)hipGetLastError(); }
#define T_FULL(T) ((T).n + 3 >= REG_MAX_TASKS)

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
extern "C" int dd_ground_candidates(const float* disp, const float* inv_K, const int32_t* rand_idx, int B, int h, int w, int np_per_it,
                                    int max_it, float g_prior, float min_depth, float max_depth, float* cand, void* stream_) {
  if (!disp || !inv_K || !rand_idx || !cand || B < 1 || max_it < 1 || max_it > GP_MAX_IT) return (int)hipErrorInvalidValue;
  const int rows = (int)(g_prior * (float)h);
  if (rows < 1) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(ground_candidates_kernel, dim3((B * max_it + GP_NT - 1) / GP_NT), dim3(GP_NT), 0, stream, disp, inv_K, rand_idx, B, h, w, rows,
                     np_per_it, max_it, depth_params(min_depth, max_depth), cand, (int*)nullptr);
  return last_error();
}

extern "C" int dd_ground_select(const float* disp, const float* inv_K, const float* cand, int B, int h, int w, int max_it, float tol, float g_prior,
                                float min_depth, float max_depth, float weight, float* g_disp, int32_t* counts, float* plane, float* out,
                                float* workspace, void* stream_) {
  if (!disp || !inv_K || !cand || !counts || !plane || !out || !workspace || B < 1 || max_it < 1 || max_it > GP_MAX_IT) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int rows = (int)(g_prior * (float)h);
  if (rows < 1) return (int)hipErrorInvalidValue;
  const int n = h * w, ng = rows * w, nblk = (n + GP_NT - 1) / GP_NT;
  const DepthParams dp = depth_params(min_depth, max_depth);
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)B * max_it * sizeof(int), stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(ground_score_kernel, dim3((ng + GP_NT - 1) / GP_NT, B), dim3(GP_NT), 0, stream, disp, inv_K, cand, B, h, w, rows, max_it, tol, dp,
                     reinterpret_cast<int*>(counts));
  hipLaunchKernelGGL(ground_hinge_kernel, dim3(nblk, B), dim3(GP_NT), 0, stream, disp, inv_K, cand, reinterpret_cast<const int*>(counts), h, w, max_it,
                     tol, max_depth, dp, weight, g_disp, plane, workspace);
  hipLaunchKernelGGL(fold_kernel, dim3(1), dim3(256), 0, stream, workspace, B * nblk, out);
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

static int reg_run(const DDRegArgs* a, void* stream_, const DDAssembleArgs* asmb, float* loss, float* out) {
  if (!a || a->abi_version != DD_ABI_VERSION || a->B < 1 || a->num_scales < 1 || a->num_scales > DD_MAX_SCALES || !a->res || !a->workspace)
    return (int)hipErrorInvalidValue;
  RegPlan p;
  if (reg_plan(*a, p)) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
#ifdef DD_REG_DEBUG_SKIP
  // timing experiments only (variant build): DD_REG_SKIP = bit mask of task kinds whose workgroups return at once
  {
    const char* e = getenv("DD_REG_SKIP");
    const int mask = e ? atoi(e) : 0;
    for (int st = 0; st < REG_STAGES; ++st)
      for (int i = 0; i < p.stage[st].n; ++i)
        if (mask & (1 << p.stage[st].t[i].kind)) p.stage[st].t[i].kind = 99;
    if (mask & (1 << K_SMOOTHALL)) p.quad_blocks = 0;          // (the smoothness kernel of its own)
    if (mask & (1 << K_GSCORE)) p.score_blocks = 0;            // (the scoring kernel of its own)
  }
#endif
  // with an assembling request the last stage (the hinge fold) runs inside the assembling kernel
  for (int st = 0; st < (asmb ? REG_STAGES - 1 : REG_STAGES); ++st) {
    if (st == 1 && p.quad_nch > 0 && p.quad_blocks > 0) {
      // the smoothness pass of every scale, behind the per-image means of stage 1
      switch (p.quad_nch) {
        case 1: hipLaunchKernelGGL((smooth_quad_kernel<1>), dim3(p.quad_blocks), dim3(SM_NT), 0, stream, p.quad); break;
        case 3: hipLaunchKernelGGL((smooth_quad_kernel<3>), dim3(p.quad_blocks), dim3(SM_NT), 0, stream, p.quad); break;
        case 4: hipLaunchKernelGGL((smooth_quad_kernel<4>), dim3(p.quad_blocks), dim3(SM_NT), 0, stream, p.quad); break;
        default: hipLaunchKernelGGL((smooth_quad_kernel<5>), dim3(p.quad_blocks), dim3(SM_NT), 0, stream, p.quad); break;
      }
      const int e = last_error();
      if (e) return e;
    }
    if (st == 1 && p.score_blocks > 0) {
      // inlier counts of every candidate of every scale (the planes come from stage 1)
      hipLaunchKernelGGL(ground_score_all_kernel, dim3(p.score_blocks), dim3(GP_NT), 0, stream, p.score);
      const int e = last_error();
      if (e) return e;
    }
    if (p.blocks[st] == 0) continue;
    hipLaunchKernelGGL(reg_stage_kernel, dim3(p.blocks[st]), dim3(RT_NT), 0, stream, *a, p.off, p.stage[st]);
    const int e = last_error();
    if (e) return e;
  }
  if (asmb) {
    HingeFold hf;
    for (int s = 0; s < DD_MAX_SCALES; ++s) {
      const bool on = s < a->num_scales && a->scale[s].disp != nullptr;
      hf.part[s] = on ? a->workspace + p.off.g_part[s] : nullptr;
      hf.count[s] = on ? a->B * ((a->scale[s].h * a->scale[s].w + RT_NT * FIN_PXT - 1) / (RT_NT * FIN_PXT)) : 0;
    }
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, stream, a->res, hf, *asmb, loss, out);
    return last_error();
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// dd_fused_loss: does this (photo, regulariser) request take the five-launch pipeline of dd_fuse.h?  The gradient pass, both frames
// on shared tensors (or the rigid mode); every smoothness entry is one of the photometric kernel's own tensors (disparity, mean-
// normalised | flow | mask) accumulating into the photometric gradient buffer; the scale-0 pyramid level is the target image itself;
// rows of whole, 16-byte aligned quads at the scales >= 1.  grp_entry[s][g]: the DDRegScale.smooth index of group g (-1: off).
static bool fused_eligible(const DDPhotoArgs& pa, const DDRegArgs& ra, int grp_entry[DD_MAX_SCALES][3]) {
  if (!pa.want_grad || pa.num_scales != ra.num_scales || pa.B != ra.B) return false;
  if (pa.min_depth != ra.min_depth || pa.max_depth != ra.max_depth || ra.max_it > GP_MAX_IT || ra.max_it < 1) return false;
  if (pa.mode != DD_MODE_RIGID && (!frames_share_tensors(pa) || pa.automask)) return false;       // (the auto-mask belongs to the rigid phase)
  if (!pa.workspace || !ra.workspace || !ra.res) return false;
  auto aligned = [](const void* q) { return (reinterpret_cast<unsigned long long>(q) & 15ull) == 0; };
  if (!aligned(ra.workspace) || !aligned(pa.workspace)) return false;
  for (int s = 0; s < pa.num_scales; ++s) {
    const DDPhotoScale& ps = pa.scale[s];
    const DDRegScale& rs = ra.scale[s];
    if (rs.h != ps.h || rs.w != ps.w || rs.h < 2 || rs.w < 2) return false;
    for (int g = 0; g < 3; ++g) grp_entry[s][g] = -1;
    for (int k = 0; k < DD_REG_SMOOTH; ++k) {
      const DDRegSmooth& sm = rs.smooth[k];
      if (!sm.inp) continue;
      int g = -1;
      if (sm.normalise && sm.C == 1 && sm.inp == ps.disp && sm.g_inp == ps.g_disp) g = 0;
      else if (!sm.normalise && sm.C == 3 && pa.mode != DD_MODE_RIGID && sm.inp == ps.flow[0] && sm.g_inp == ps.g_flow[0]) g = 1;
      else if (!sm.normalise && sm.C == 1 && pa.mode == DD_MODE_FLOW_MASK && sm.inp == ps.mask[0] && sm.g_inp == ps.g_mask[0]) g = 2;
      if (g < 0 || grp_entry[s][g] >= 0 || !sm.g_inp || !rs.img || sm.weight == 0.f) return false;
      grp_entry[s][g] = k;
    }
    const bool any = grp_entry[s][0] >= 0 || grp_entry[s][1] >= 0 || grp_entry[s][2] >= 0;
    if (ps.shift == 0) {
      if (any && rs.img != pa.target) return false;              // the tile kernel takes the colours from its staged target
    } else {
      // the combine pass works on 16-byte quads whether it smooths or not
      if (ps.w % 4 != 0 || !aligned(ps.g_disp)) return false;
      if (pa.mode != DD_MODE_RIGID && (!aligned(ps.g_flow[0]) || ((size_t)ps.h * ps.w) % 4 != 0)) return false;
      if (pa.mode == DD_MODE_FLOW_MASK && !aligned(ps.g_mask[0])) return false;
      if (any && (!aligned(rs.img) || !aligned(ps.disp) || (pa.mode != DD_MODE_RIGID && !aligned(ps.flow[0])) ||
                  (pa.mode == DD_MODE_FLOW_MASK && !aligned(ps.mask[0]))))
        return false;
    }
    if (rs.disp && (rs.disp != ps.disp || rs.g_disp != ps.g_disp)) return false;
    // sparsity: both frames on one motion_prob tensor or two, as reg_plan takes them
  }
  return true;
}

// both frames on one motion_prob tensor, whole 16-byte aligned quads: the sparsity passes of this scale run as K_SPCOUNT2 / K_SPGRAD2
// (one pass for both frames, the gradient written with plain stores)
static bool sparsity_quads(const DDRegScale& rs) {
  auto al16 = [](const void* q) { return (reinterpret_cast<unsigned long long>(q) & 15ull) == 0; };
  const int n = rs.h * rs.w;
  return rs.prob[0] && rs.prob[0] == rs.prob[1] && rs.g_prob[0] == rs.g_prob[1] && n % 4 == 0 && rs.w % 4 == 0 && rs.delta[0] && rs.delta[1] &&
         al16(rs.delta[0]) && al16(rs.delta[1]) && al16(rs.prob[0]) && rs.g_prob[0] && al16(rs.g_prob[0]) && rs.delta_sum[0] && rs.delta_sum[1];
}

// part: 0 = all five launches, 1 = the tile kernel alone, 2 = the four behind it
// DD_FUSED_TRACE=1: which check or launch of dd_fused_loss failed (stderr)
static int fused_fail(int where, int code) {
  if (getenv("DD_FUSED_TRACE")) fprintf(stderr, "[dd_fused_loss] failed at check %d with code %d\n", where, code);
  return code;
}

static int fused_run(const DDPhotoArgs* pa, const DDRegArgs* ra, const DDAssembleArgs* asmb, float* loss, float* out, void* stream_, int part) {
  if (!pa || !ra || !asmb || !loss || !out || pa->abi_version != DD_ABI_VERSION || ra->abi_version != DD_ABI_VERSION) return fused_fail(1, (int)hipErrorInvalidValue);
  if (asmb->n < 0 || asmb->n > DD_MAX_RES || asmb->num_scales != ra->num_scales || ra->num_scales < 1 || ra->num_scales > DD_MAX_SCALES)
    return fused_fail(2, (int)hipErrorInvalidValue);
  int grp_entry[DD_MAX_SCALES][3];
  if (!fused_eligible(*pa, *ra, grp_entry)) return (int)hipErrorNotSupported;
  RegPlan p;
  if (reg_plan(*ra, p)) return fused_fail(3, (int)hipErrorInvalidValue);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int B = ra->B, S = ra->num_scales;
  const int tiles_x = (pa->W + TW - 1) / TW, tiles_y = (pa->H + TH - 1) / TH, tiles = tiles_x * tiles_y;
  float* ws = ra->workspace;

  // ---- 1: the tile kernel with the scale-0 smoothness ----
  FuseInfo fuse;
  memset(&fuse, 0, sizeof(fuse));
  for (int s = 0; s < S; ++s) {
    const DDRegScale& rs = ra->scale[s];
    for (int g = 0; g < 3; ++g) {
      const int k = grp_entry[s][g];
      if (k < 0) continue;
      const DDRegSmooth& sm = rs.smooth[k];
      const float cnt = static_cast<float>(B) * static_cast<float>(sm.C);
      fuse.sc[s].wx[g] = sm.weight / (cnt * rs.h * (rs.w - 1));
      fuse.sc[s].wy[g] = sm.weight / (cnt * (rs.h - 1) * rs.w);
      if (g == 0) fuse.sc[s].g_tmp = ws + p.off.sm_gtmp[s][k];
      fuse.on = 1;
    }
  }
  // the extra workgroup per (image, scale) of the tile kernel's launch: RANSAC candidates + disparity sums, photo-independent
  SideInfo side;
  memset(&side, 0, sizeof(side));
  side.np = ra->np_per_it; side.max_it = ra->max_it;
  for (int s = 0; s < S; ++s) {
    const DDRegScale& rs = ra->scale[s];
    if (rs.disp) {
      side.sc[s].inv_K = rs.inv_K; side.sc[s].rand_idx = rs.rand_idx; side.sc[s].cand = ws + p.off.g_cand[s];
      side.sc[s].rows = (int)(ra->g_prior * (float)rs.h);
      side.on = 1;
    }
    if (grp_entry[s][0] >= 0) { side.sc[s].mean_partial = ws + p.off.mean[s]; side.on = 1; }
  }
  if (part != 2) {
    const int e = launch_tile_fused(*pa, fuse, side, stream);
    if (e) return fused_fail(101, e);
  }
  if (part == 1) return 0;

  // ---- 2: footprint sums + smoothness | tile-record fold | scoring | disparity sums ----
  PostArgs post;
  memset(&post, 0, sizeof(post));
  post.num_scales = S;
  int blocks = 0;
  long long fp_off[DD_MAX_SCALES];
  footprint_floats(*pa, fp_off);
  const float* fp_base = pa->workspace + (size_t)tiles * B * S * DD_PARTIAL_STRIDE;
  const int nch = gradient_channels(*pa);
  post.comb.B = B; post.comb.tiles_x = tiles_x; post.comb.tiles_y = tiles_y;
  for (int s = 0; s < S; ++s) {
    const DDPhotoScale& ps = pa->scale[s];
    CombineScale& q = post.comb.sc[s];
    q.first = blocks;
    if (ps.shift == 0) continue;                 // gx stays 0: the tile kernel stored scale 0 itself
    const int n = ps.h * ps.w;
    q.img = ra->scale[s].img; q.fp = fp_base + fp_off[s]; q.h = ps.h; q.w = ps.w; q.shift = ps.shift;
    q.gx = (n + RT_NT_FUSED * 4 - 1) / (RT_NT_FUSED * 4);
    q.npad = (q.gx * B + 7) / 8 * 8;
    q.part = ws + p.off.post_part[s];
    q.g_tmp = fuse.sc[s].g_tmp;
    q.in[0] = ps.disp; q.gp[0] = ps.g_disp; q.bstride[0] = n;
    for (int c = 0; c < 3; ++c)
      if (nch >= 4) { q.in[1 + c] = ps.flow[0] + (size_t)c * n; q.gp[1 + c] = ps.g_flow[0] + (size_t)c * n; q.bstride[1 + c] = 3 * n; }
    if (nch >= 5) { q.in[4] = ps.mask[0]; q.gp[4] = ps.g_mask[0]; q.bstride[4] = n; }
    for (int g = 0; g < 3; ++g) { q.wx[g] = fuse.sc[s].wx[g]; q.wy[g] = fuse.sc[s].wy[g]; }
    blocks += q.npad;
  }
  post.first_fold = blocks;
  post.fold.partials = pa->workspace; post.fold.sums = pa->sums; post.fold.g_T[0] = pa->g_T[0]; post.fold.g_T[1] = pa->g_T[1];
  post.fold.S = S; post.fold.B = B; post.fold.tiles = tiles;
  blocks += S + B;
  post.first_score = blocks;
  post.score.num_scales = S; post.score.B = B; post.score.max_it = ra->max_it; post.score.np = ra->np_per_it; post.score.tol = ra->tol;
  post.score.dp = depth_params(ra->min_depth, ra->max_depth);
  int sblocks = 0;
  for (int s = 0; s < S; ++s) {
    post.score.sc[s] = p.score.sc[s];
    post.score.sc[s].first = sblocks;
    if (ra->scale[s].disp) {
      sblocks += p.score.sc[s].gx * B;            // (candidates: solved by the tile kernel's extra workgroups, read from p.score.sc[s].cand)
    } else {
      post.score.sc[s].gx = 0;
    }
  }
  blocks += sblocks;
  post.first_mean = blocks;
  // (the per-image disparity sums: tile kernel's extra workgroups; the mean task of this kernel stays for callers without them)
#ifdef DD_REG_DEBUG_SKIP
  // timing experiments only (variant build, scripts/post_task_costs.sh): DD_POST_SKIP = bit mask of post-kernel tasks whose workgroups
  // return at once (1 combine + smoothness, 2 tile-record fold, 4 scoring, 8 disparity sums); the results are wrong then
  {
    const char* ev = getenv("DD_POST_SKIP");
    const int mask = ev ? atoi(ev) : 0;
    if (mask & 1) for (int s = 0; s < S; ++s) post.comb.sc[s].gx = 0;
    if (mask & 2) post.fold.S = post.fold.B = 0, post.fold.tiles = 0;
    if (mask & 4) for (int s = 0; s < S; ++s) post.score.sc[s].gx = 0;
    if (mask & 8) for (int s = 0; s < S; ++s) post.mean.inp[s] = nullptr;
  }
#endif
  switch (nch) {
    case 1: hipLaunchKernelGGL((fused_post_kernel<1>), dim3(blocks), dim3(RT_NT_FUSED), 0, stream, post); break;
    case 4: hipLaunchKernelGGL((fused_post_kernel<4>), dim3(blocks), dim3(RT_NT_FUSED), 0, stream, post); break;
    case 5: hipLaunchKernelGGL((fused_post_kernel<5>), dim3(blocks), dim3(RT_NT_FUSED), 0, stream, post); break;
    default: return fused_fail(4, (int)hipErrorInvalidValue);
  }
  int e = last_error();
  if (e) return fused_fail(102, e);

  // ---- 3: static-pixel counts | per-image scalars;  4: sparsity gradient | disparity finish (tasks of reg_stage_kernel) ----
  RegTasks mid, fin;
  mid.n = fin.n = 0;
  int mid_blocks = 0, fin_blocks = 0;
  auto add = [](RegTasks& T, int& nb, int kind, int s, int idx, int gx, int gy, int gx2) {
    RegTask& t = T.t[T.n++];
    t.first = nb; t.gx = gx; t.gx2 = gx2; t.kind = (short)kind; t.scale = (signed char)s; t.idx = (signed char)idx;
    nb += gx * gy;
  };
  unsigned long long slots = 0ull;
  for (int s = 0; s < S; ++s) {
    const DDRegScale& rs = ra->scale[s];
    const DDPhotoScale& ps = pa->scale[s];
    const int n = rs.h * rs.w;
    const int nblk_spg = (n + RT_NT * SPG_PXT - 1) / (RT_NT * SPG_PXT), nblk_fin = (n + RT_NT * FIN_PXT - 1) / (RT_NT * FIN_PXT);
    const bool any = grp_entry[s][0] >= 0 || grp_entry[s][1] >= 0 || grp_entry[s][2] >= 0;
    for (int g = 0; g < 3; ++g) slots |= (unsigned long long)(grp_entry[s][g] >= 0 ? 2 * grp_entry[s][g] : 15) << (4 * (s * 3 + g));
    FoldSource& fs = p.off.fold[s];
    fs.rec = nullptr;
    if (any) {
      if (ps.shift == 0) {
        fs.rec = pa->workspace + (size_t)s * B * tiles * DD_PARTIAL_STRIDE; fs.count = tiles; fs.stride = DD_PARTIAL_STRIDE;
        fs.img_stride = tiles * DD_PARTIAL_STRIDE; fs.base = REC_SMOOTH;
      } else {
        fs.rec = ws + p.off.post_part[s]; fs.count = post.comb.sc[s].gx; fs.stride = 8; fs.img_stride = post.comb.sc[s].gx * 8; fs.base = 0;
      }
    }
    auto al16 = [](const void* q) { return (reinterpret_cast<unsigned long long>(q) & 15ull) == 0; };
    const bool quads = n % 4 == 0 && rs.w % 4 == 0;
    if (any || rs.disp) {
      if (T_FULL(mid) || T_FULL(fin)) return fused_fail(5, (int)hipErrorInvalidValue);
      add(mid, mid_blocks, K_IMGFOLD, s, 0, 1, B, rs.disp ? p.score.sc[s].gx : 0);
      const bool fin4 = quads && al16(ps.g_disp) && al16(ps.disp) && (grp_entry[s][0] < 0 || al16(fuse.sc[s].g_tmp));
      if ((grp_entry[s][0] >= 0) || rs.disp) add(fin, fin_blocks, fin4 ? K_DISPFIN4 : K_DISPFIN, s, 0, nblk_fin, B, 0);
    }
    if (sparsity_quads(rs)) {
      // both frames on one motion_prob tensor: one counting pass and one gradient pass for the two of them, plain stores
      if (T_FULL(mid) || T_FULL(fin)) return fused_fail(6, (int)hipErrorInvalidValue);
      const int gx2 = (n + SP_NT * SP2_PXT - 1) / (SP_NT * SP2_PXT);
      add(mid, mid_blocks, K_SPCOUNT2, s, 0, gx2, B, 0);
      add(fin, fin_blocks, K_SPGRAD2, s, 0, gx2, B, gx2);
    } else {
      const bool one_tensor = rs.prob[0] && rs.prob[0] == rs.prob[1];
      for (int f = 0; f < DD_NUM_SRC; ++f) {
        if (!rs.prob[f]) continue;
        if (T_FULL(mid) || T_FULL(fin)) return fused_fail(7, (int)hipErrorInvalidValue);
        add(mid, mid_blocks, K_SPCOUNT, s, f, SP_BPI, B, 0);
        if (!one_tensor) add(fin, fin_blocks, K_SPGRAD, s, f, nblk_spg, B, 0);
      }
      if (one_tensor) add(fin, fin_blocks, K_SPGRAD, s, 2, nblk_spg, B, 0);
    }
  }
  if (mid_blocks > 0) {
    hipLaunchKernelGGL(reg_stage_kernel, dim3(mid_blocks), dim3(RT_NT), 0, stream, *ra, p.off, mid);
    e = last_error();
    if (e) return fused_fail(103, e);
  }
  if (fin_blocks > 0) {
    hipLaunchKernelGGL(reg_stage_kernel, dim3(fin_blocks), dim3(RT_NT), 0, stream, *ra, p.off, fin);
    e = last_error();
    if (e) return fused_fail(104, e);
  }
  // ---- 5: the losses dict values ----
  HingeFold hf;
  for (int s = 0; s < DD_MAX_SCALES; ++s) {
    const bool on = s < S && ra->scale[s].disp != nullptr;
    hf.part[s] = on ? ws + p.off.g_part[s] : nullptr;
    hf.count[s] = on ? B * ((ra->scale[s].h * ra->scale[s].w + RT_NT * FIN_PXT - 1) / (RT_NT * FIN_PXT)) : 0;
  }
  ImageSums im;
  im.pre_all = ws + p.pre_all; im.slots = slots; im.B = B;
  hipLaunchKernelGGL(fused_finish_kernel, dim3(1), dim3(256), 0, stream, ra->res, hf, im, *asmb, loss, out);
  return last_error();
}

extern "C" int dd_fused_loss(const DDPhotoArgs* photo, const DDRegArgs* reg, const DDAssembleArgs* assemble, float* loss, float* out, void* stream) {
  return fused_run(photo, reg, assemble, loss, out, stream, 0);
}

extern "C" int dd_fused_loss_part(const DDPhotoArgs* photo, const DDRegArgs* reg, const DDAssembleArgs* assemble, float* loss, float* out,
                                  void* stream, int part) {
  if (part < 0 || part > 2) return (int)hipErrorInvalidValue;
  return fused_run(photo, reg, assemble, loss, out, stream, part);
}

extern "C" int dd_fused_loss_supported(const DDPhotoArgs* photo, const DDRegArgs* reg) {
  int grp_entry[DD_MAX_SCALES][3];
  if (!(photo && reg && fused_eligible(*photo, *reg, grp_entry))) return 0;
  // 2: every element of every gradient buffer the request names is written by exactly one plain store -- the caller need not zero
  // them (the photometric gradients always are; motion_prob's when its sparsity passes run on quads)
  for (int s = 0; s < reg->num_scales; ++s)
    if ((reg->scale[s].prob[0] || reg->scale[s].prob[1]) && !sparsity_quads(reg->scale[s])) return 1;
  return 2;
}

extern "C" int dd_reg_losses(const DDRegArgs* a, void* stream_) { return reg_run(a, stream_, nullptr, nullptr, nullptr); }

extern "C" int dd_reg_losses_finish(const DDRegArgs* a, const DDAssembleArgs* args, float* loss, float* out, void* stream_) {
  if (!args || !loss || !out || args->n < 0 || args->n > DD_MAX_RES || args->num_scales < 1 || args->num_scales > DD_MAX_SCALES || !a ||
      args->num_scales != a->num_scales)
    return (int)hipErrorInvalidValue;
  return reg_run(a, stream_, args, loss, out);
}
