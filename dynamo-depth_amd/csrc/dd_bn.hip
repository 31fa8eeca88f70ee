// Training-mode BatchNorm2d on channels-last tensors with the following activation (ReLU / GELU) and an optional residual
// add fused into the normalisation pass (reference: every `bn -> relu`, `bn(+identity) -> relu` of torchvision's BasicBlock used
// by networks/resnet_encoder.py:42-88, and `BNGELU` / `bn1` of networks/depth_encoder.py:137-148,194-199).
// Stock PyTorch-ROCm runs 3 MIOpen kernels per BN forward, 3 per backward, plus one element-wise kernel per activation /
// residual in each direction: ~570 + ~250 launches per training step, almost all at the ~5 us launch floor.  Here a BN(+act
// +residual) is 2 launches forward and 2 backward; every pass is a coalesced float4 stream over [rows, C].
// Sums: fp32 inside a chunk of rows, fp64 across lanes and chunks, fixed order -> run-to-run reproducible.
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_half.h"

namespace dd {

constexpr int BN_NT = 1024;       // 16 waves: enough loads in flight per CU for a one-block-per-CU grid to stream
constexpr int BN_MAX_C = 512;
constexpr int BN_MAX_CHUNKS = 256;
constexpr int BN_ROWS_PER_THREAD = 8;   // target rows per thread and block: short dependent chains, latency hidden by the unroll

enum { BN_ACT_NONE = 0, BN_ACT_RELU = 1, BN_ACT_GELU = 2 };

__device__ __forceinline__ float bn_gelu(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float bn_gelu_grad(float z) {
  return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.39894228040143268f * __expf(-0.5f * z * z);
}

struct BnGeom {
  int C4, lanes, threads;
};
static inline BnGeom bn_geom(int C) {
  BnGeom g;
  g.C4 = C >> 2;
  g.lanes = BN_NT / g.C4 < 1 ? 1 : BN_NT / g.C4;
  g.threads = g.lanes * g.C4;
  return g;
}

// ---- pass 1 (forward): per-chunk sum and sum of squares per channel ----------------------------------------------------
// partial[chunk][2][C].  thread = (row lane, 4 channels)
template <typename T>
__global__ __launch_bounds__(BN_NT) void bn_stats_kernel(const T* __restrict__ x, long long rows, int C, int lanes, int rows_per_chunk,
                                                          float* __restrict__ partial) {
  extern __shared__ float red[];                             // [lanes][2][C]
  const int C4 = C >> 2;
  const int lane = threadIdx.x / C4, c4 = threadIdx.x - lane * C4;
  const long long r0 = (long long)blockIdx.x * rows_per_chunk;
  long long r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll 4
  for (long long r = r0 + lane; r < r1; r += lanes) {
    const float4 v = IO<T>::load4(x, r * C4 + c4);
    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
    s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
  }
  reinterpret_cast<float4*>(red)[(lane * 2 + 0) * C4 + c4] = s1;
  reinterpret_cast<float4*>(red)[(lane * 2 + 1) * C4 + c4] = s2;
  __syncthreads();
  float* dst = partial + (size_t)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    double s = 0.0;
    for (int l = 0; l < lanes; ++l) s += (double)red[l * 2 * C + i];
    dst[i] = (float)s;
  }
}

// Folds partial[chunks][2][C] into per-channel totals (fp64, fixed order) with the whole block: thread (lane, c4) takes the
// chunks lane, lane+lanes, ... of its four channels (fp64), the lanes are then added in order (fp64).  scratch: [lanes][2][C] floats.
// On return (after the barrier inside) thread c < C reads its totals with bn_total().
__device__ __forceinline__ void bn_fold_block(const float* __restrict__ partial, int chunks, int C, int lanes, float* scratch) {
  const int C4 = C >> 2;
  const int lane = threadIdx.x / C4, c4 = threadIdx.x - lane * C4;
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  const float4* pv = reinterpret_cast<const float4*>(partial);
  for (int k = lane; k < chunks; k += lanes) {
    const float4 p1 = pv[(size_t)(k * 2 + 0) * C4 + c4], p2 = pv[(size_t)(k * 2 + 1) * C4 + c4];
    a[0] += (double)p1.x; a[1] += (double)p1.y; a[2] += (double)p1.z; a[3] += (double)p1.w;
    b[0] += (double)p2.x; b[1] += (double)p2.y; b[2] += (double)p2.z; b[3] += (double)p2.w;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    scratch[(size_t)(lane * 2 + 0) * C + c4 * 4 + j] = (float)a[j];
    scratch[(size_t)(lane * 2 + 1) * C + c4 * 4 + j] = (float)b[j];
  }
  __syncthreads();
}
__device__ __forceinline__ void bn_total(const float* scratch, int C, int lanes, int c, double& s1, double& s2) {
  s1 = 0.0; s2 = 0.0;
  for (int l = 0; l < lanes; ++l) {
    s1 += (double)scratch[(size_t)(l * 2 + 0) * C + c];
    s2 += (double)scratch[(size_t)(l * 2 + 1) * C + c];
  }
}

// ---- pass 2 (forward): finalise the statistics (every block, redundantly: cheaper than a third launch), normalise,
// add the residual, activate ---------------------------------------------------------------------------------------------
template <typename T, int ACT, bool RES>
__global__ __launch_bounds__(BN_NT) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, long long rows, int C,
                                                          int lanes, int chunks, const float* __restrict__ partial,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ save_mean, float* __restrict__ save_invstd, int rows_per_block,
                                                          T* __restrict__ out) {
  __shared__ __align__(16) float sc[BN_MAX_C];
  __shared__ __align__(16) float sh[BN_MAX_C];
  extern __shared__ float fold_scratch[];
  bn_fold_block(partial, chunks, C, lanes, fold_scratch);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s1, s2;
    bn_total(fold_scratch, C, lanes, c, s1, s2);
    const double n = (double)rows, inv_n = 1.0 / n;
    const double mean = s1 * inv_n;
    double var = s2 * inv_n - mean * mean;                 // fp64: the cancellation happens here
    if (var < 0.0) var = 0.0;
    const float invstd = 1.f / sqrtf((float)var + eps);     // fp32 like PyTorch's invstd
    const float k = gamma[c] * invstd;
    sc[c] = k;
    sh[c] = beta[c] - (float)mean * k;
    if (blockIdx.x == 0) {
      save_mean[c] = (float)mean;
      save_invstd[c] = invstd;
      if (running_mean) {
        const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
      }
    }
  }
  __syncthreads();
  const int C4 = C >> 2;
  const int lane = threadIdx.x / C4, c4 = threadIdx.x - lane * C4;
  const float4 k4 = reinterpret_cast<const float4*>(sc)[c4], b4 = reinterpret_cast<const float4*>(sh)[c4];
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
#pragma unroll 4
  for (long long r = r0 + lane; r < r1; r += lanes) {
    const float4 v = IO<T>::load4(x, r * C4 + c4);
    float4 y = make_float4(fmaf(v.x, k4.x, b4.x), fmaf(v.y, k4.y, b4.y), fmaf(v.z, k4.z, b4.z), fmaf(v.w, k4.w, b4.w));
    if (RES) {
      const float4 q = IO<T>::load4(res, r * C4 + c4);
      y.x += q.x; y.y += q.y; y.z += q.z; y.w += q.w;
    }
    if (ACT == BN_ACT_RELU) {
      y.x = y.x > 0.f ? y.x : 0.f; y.y = y.y > 0.f ? y.y : 0.f; y.z = y.z > 0.f ? y.z : 0.f; y.w = y.w > 0.f ? y.w : 0.f;
    } else if (ACT == BN_ACT_GELU) {
      y.x = bn_gelu(y.x); y.y = bn_gelu(y.y); y.z = bn_gelu(y.z); y.w = bn_gelu(y.w);
    }
    IO<T>::store4(out, r * C4 + c4, y);
  }
}

// gradient entering the normalisation: g through the activation (ReLU: mask from the saved output; GELU: derivative at the
// recomputed pre-activation)
template <int ACT>
__device__ __forceinline__ float4 bn_act_bwd(const float4 g, const float4 v, const float4 o, const float4 k4, const float4 b4) {
  float4 r = g;
  if (ACT == BN_ACT_RELU) {
    r.x = o.x > 0.f ? g.x : 0.f; r.y = o.y > 0.f ? g.y : 0.f; r.z = o.z > 0.f ? g.z : 0.f; r.w = o.w > 0.f ? g.w : 0.f;
  } else if (ACT == BN_ACT_GELU) {
    r.x = g.x * bn_gelu_grad(fmaf(v.x, k4.x, b4.x)); r.y = g.y * bn_gelu_grad(fmaf(v.y, k4.y, b4.y));
    r.z = g.z * bn_gelu_grad(fmaf(v.z, k4.z, b4.z)); r.w = g.w * bn_gelu_grad(fmaf(v.w, k4.w, b4.w));
  }
  return r;
}

// ---- pass 1 (backward): per-chunk sum of g' and of g' * xhat per channel ------------------------------------------------
// FROM_OUT: the ReLU mask is read from the saved output (needed when a residual was added in front of the activation); without a
// residual it is recomputed from x -- fmaf(x, k, b) with the forward's own k and b, the same bits -- and a quarter of the pass's reads goes away
template <typename T, int ACT, bool FROM_OUT>
__global__ __launch_bounds__(BN_NT) void bn_bwd_stats_kernel(const T* __restrict__ x, const T* __restrict__ g, const T* __restrict__ outp,
                                                              long long rows, int C, int lanes, int rows_per_chunk,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                              float* __restrict__ partial) {
  extern __shared__ float red[];
  const int C4 = C >> 2;
  const int lane = threadIdx.x / C4, c4 = threadIdx.x - lane * C4;
  const float4 m4 = reinterpret_cast<const float4*>(save_mean)[c4], i4 = reinterpret_cast<const float4*>(save_invstd)[c4];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[c4], be = reinterpret_cast<const float4*>(beta)[c4];
  const float4 k4 = make_float4(ga.x * i4.x, ga.y * i4.y, ga.z * i4.z, ga.w * i4.w);
  const float4 b4 = make_float4(be.x - m4.x * k4.x, be.y - m4.y * k4.y, be.z - m4.z * k4.z, be.w - m4.w * k4.w);
  const long long r0 = (long long)blockIdx.x * rows_per_chunk;
  long long r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
#pragma unroll 4
  for (long long r = r0 + lane; r < r1; r += lanes) {
    const float4 v = IO<T>::load4(x, r * C4 + c4);
    const float4 o = ACT != BN_ACT_RELU ? v : (FROM_OUT ? IO<T>::load4(outp, r * C4 + c4)
                                                        : make_float4(fmaf(v.x, k4.x, b4.x), fmaf(v.y, k4.y, b4.y), fmaf(v.z, k4.z, b4.z), fmaf(v.w, k4.w, b4.w)));
    const float4 gp = bn_act_bwd<ACT>(IO<T>::load4(g, r * C4 + c4), v, o, k4, b4);
    s1.x += gp.x; s1.y += gp.y; s1.z += gp.z; s1.w += gp.w;
    s2.x = fmaf(gp.x, (v.x - m4.x) * i4.x, s2.x); s2.y = fmaf(gp.y, (v.y - m4.y) * i4.y, s2.y);
    s2.z = fmaf(gp.z, (v.z - m4.z) * i4.z, s2.z); s2.w = fmaf(gp.w, (v.w - m4.w) * i4.w, s2.w);
  }
  reinterpret_cast<float4*>(red)[(lane * 2 + 0) * C4 + c4] = s1;
  reinterpret_cast<float4*>(red)[(lane * 2 + 1) * C4 + c4] = s2;
  __syncthreads();
  float* dst = partial + (size_t)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    double s = 0.0;
    for (int l = 0; l < lanes; ++l) s += (double)red[l * 2 * C + i];
    dst[i] = (float)s;
  }
}

// ---- pass 2 (backward): dx = gamma*invstd * (g' - mean(g') - xhat*mean(g' xhat)); dres = g'; block 0 writes dgamma/dbeta ---
template <typename T, int ACT, bool RES, bool FROM_OUT>
__global__ __launch_bounds__(BN_NT) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ g, const T* __restrict__ outp,
                                                              long long rows, int C, int lanes, int chunks, const float* __restrict__ partial,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                              int rows_per_block, T* __restrict__ gx, T* __restrict__ gres,
                                                              float* __restrict__ ggamma, float* __restrict__ gbeta) {
  __shared__ __align__(16) float mg[BN_MAX_C];               // mean of g'
  __shared__ __align__(16) float mgx[BN_MAX_C];              // mean of g' * xhat
  extern __shared__ float fold_scratch[];
  bn_fold_block(partial, chunks, C, lanes, fold_scratch);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double s1, s2;
    bn_total(fold_scratch, C, lanes, c, s1, s2);
    mg[c] = (float)(s1 / (double)rows);
    mgx[c] = (float)(s2 / (double)rows);
    if (blockIdx.x == 0) {
      gbeta[c] = (float)s1;
      ggamma[c] = (float)s2;
    }
  }
  __syncthreads();
  const int C4 = C >> 2;
  const int lane = threadIdx.x / C4, c4 = threadIdx.x - lane * C4;
  const float4 m4 = reinterpret_cast<const float4*>(save_mean)[c4], i4 = reinterpret_cast<const float4*>(save_invstd)[c4];
  const float4 ga = reinterpret_cast<const float4*>(gamma)[c4], be = reinterpret_cast<const float4*>(beta)[c4];
  const float4 k4 = make_float4(ga.x * i4.x, ga.y * i4.y, ga.z * i4.z, ga.w * i4.w);
  const float4 b4 = make_float4(be.x - m4.x * k4.x, be.y - m4.y * k4.y, be.z - m4.z * k4.z, be.w - m4.w * k4.w);
  const float4 a4 = reinterpret_cast<const float4*>(mg)[c4], c44 = reinterpret_cast<const float4*>(mgx)[c4];
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
#pragma unroll 4
  for (long long r = r0 + lane; r < r1; r += lanes) {
    const float4 v = IO<T>::load4(x, r * C4 + c4);
    const float4 o = ACT != BN_ACT_RELU ? v : (FROM_OUT ? IO<T>::load4(outp, r * C4 + c4)
                                                        : make_float4(fmaf(v.x, k4.x, b4.x), fmaf(v.y, k4.y, b4.y), fmaf(v.z, k4.z, b4.z), fmaf(v.w, k4.w, b4.w)));
    const float4 gp = bn_act_bwd<ACT>(IO<T>::load4(g, r * C4 + c4), v, o, k4, b4);
    float4 d;
    d.x = k4.x * (gp.x - a4.x - (v.x - m4.x) * i4.x * c44.x);
    d.y = k4.y * (gp.y - a4.y - (v.y - m4.y) * i4.y * c44.y);
    d.z = k4.z * (gp.z - a4.z - (v.z - m4.z) * i4.z * c44.z);
    d.w = k4.w * (gp.w - a4.w - (v.w - m4.w) * i4.w * c44.w);
    IO<T>::store4(gx, r * C4 + c4, d);
    if (RES) IO<T>::store4(gres, r * C4 + c4, gp);
  }
}

static inline bool bn_dims_ok(long long rows, int C) { return rows >= 1 && C >= 4 && (C & 3) == 0 && C <= BN_MAX_C; }

struct BnPlan {
  BnGeom g;
  int chunks, rows_per_chunk, blocks, rows_per_block;
};
static inline BnPlan bn_plan(long long rows, int C) {
  BnPlan p;
  p.g = bn_geom(C);
  const long long per_block = (long long)p.g.lanes * BN_ROWS_PER_THREAD;
  long long want = (rows + per_block - 1) / per_block;
  p.chunks = (int)(want < 1 ? 1 : (want > BN_MAX_CHUNKS ? BN_MAX_CHUNKS : want));     // large tensors: more rows per thread
  p.rows_per_chunk = (int)((rows + p.chunks - 1) / p.chunks);
  p.chunks = (int)((rows + p.rows_per_chunk - 1) / p.rows_per_chunk);
  long long nb = want > 1024 ? 1024 : (want < 1 ? 1 : want);
  p.rows_per_block = (int)((rows + nb - 1) / nb);
  p.blocks = (int)((rows + p.rows_per_block - 1) / p.rows_per_block);
  return p;
}

}  // namespace dd

using namespace dd;

extern "C" size_t dd_bn_workspace_bytes(int C) { return (size_t)BN_MAX_CHUNKS * 2 * C * sizeof(float); }

#define BN_DISPATCH(KERNEL, ...)                                                                             \
  do {                                                                                                       \
    if (act == BN_ACT_NONE) { if (has_res) KERNEL<T, BN_ACT_NONE, true> __VA_ARGS__; else KERNEL<T, BN_ACT_NONE, false> __VA_ARGS__; }  \
    else if (act == BN_ACT_RELU) { if (has_res) KERNEL<T, BN_ACT_RELU, true> __VA_ARGS__; else KERNEL<T, BN_ACT_RELU, false> __VA_ARGS__; } \
    else { if (has_res) KERNEL<T, BN_ACT_GELU, true> __VA_ARGS__; else KERNEL<T, BN_ACT_GELU, false> __VA_ARGS__; }        \
  } while (0)

template <typename T>
static void bn_fwd_launch(const void* x_, const void* residual_, long long rows, int C, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, float* save_mean, float* save_invstd, int act, void* out_, float* partial,
                          hipStream_t s) {
  const T* x = static_cast<const T*>(x_);
  const T* residual = static_cast<const T*>(residual_);
  T* out = static_cast<T*>(out_);
  const BnPlan p = bn_plan(rows, C);
  const size_t lds = (size_t)p.g.lanes * 2 * C * sizeof(float);
  hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(p.chunks), dim3(p.g.threads), lds, s, x, rows, C, p.g.lanes, p.rows_per_chunk, partial);
  const bool has_res = residual != nullptr;
  BN_DISPATCH(bn_apply_kernel, <<<dim3(p.blocks), dim3(p.g.threads), lds, s>>>(x, residual, rows, C, p.g.lanes, p.chunks, partial, gamma, beta, eps,
                                                                             momentum, running_mean, running_var, save_mean, save_invstd,
                                                                             p.rows_per_block, out));
}

template <typename T>
static void bn_bwd_launch(const void* x_, const void* g_, const void* out_, long long rows, int C, const float* gamma, const float* beta,
                          const float* save_mean, const float* save_invstd, int act, void* gx_, void* gres_, float* g_gamma, float* g_beta,
                          float* partial, hipStream_t s) {
  const T* x = static_cast<const T*>(x_);
  const T* g_out = static_cast<const T*>(g_);
  const T* out = static_cast<const T*>(out_);
  T* g_x = static_cast<T*>(gx_);
  T* g_residual = static_cast<T*>(gres_);
  const BnPlan p = bn_plan(rows, C);
  const size_t lds = (size_t)p.g.lanes * 2 * C * sizeof(float);
  const bool has_res = g_residual != nullptr;
  if (act == BN_ACT_NONE)
    hipLaunchKernelGGL((bn_bwd_stats_kernel<T, BN_ACT_NONE, false>), dim3(p.chunks), dim3(p.g.threads), lds, s, x, g_out, out, rows, C, p.g.lanes,
                       p.rows_per_chunk, gamma, beta, save_mean, save_invstd, partial);
  else if (act == BN_ACT_RELU && out != nullptr)
    hipLaunchKernelGGL((bn_bwd_stats_kernel<T, BN_ACT_RELU, true>), dim3(p.chunks), dim3(p.g.threads), lds, s, x, g_out, out, rows, C, p.g.lanes,
                       p.rows_per_chunk, gamma, beta, save_mean, save_invstd, partial);
  else if (act == BN_ACT_RELU)
    hipLaunchKernelGGL((bn_bwd_stats_kernel<T, BN_ACT_RELU, false>), dim3(p.chunks), dim3(p.g.threads), lds, s, x, g_out, out, rows, C, p.g.lanes,
                       p.rows_per_chunk, gamma, beta, save_mean, save_invstd, partial);
  else
    hipLaunchKernelGGL((bn_bwd_stats_kernel<T, BN_ACT_GELU, false>), dim3(p.chunks), dim3(p.g.threads), lds, s, x, g_out, out, rows, C, p.g.lanes,
                       p.rows_per_chunk, gamma, beta, save_mean, save_invstd, partial);
#define BN_BWD_APPLY(ACT_, RES_, FO_)                                                                                                       \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<T, ACT_, RES_, FO_>), dim3(p.blocks), dim3(p.g.threads), lds, s, x, g_out, out, rows, C, p.g.lanes, p.chunks, \
                     partial, gamma, beta, save_mean, save_invstd, p.rows_per_block, g_x, g_residual, g_gamma, g_beta)
  const bool from_out = act == BN_ACT_RELU && out != nullptr;
  if (act == BN_ACT_NONE) { if (has_res) BN_BWD_APPLY(BN_ACT_NONE, true, false); else BN_BWD_APPLY(BN_ACT_NONE, false, false); }
  else if (act == BN_ACT_GELU) { BN_BWD_APPLY(BN_ACT_GELU, false, false); }
  else if (from_out) { if (has_res) BN_BWD_APPLY(BN_ACT_RELU, true, true); else BN_BWD_APPLY(BN_ACT_RELU, false, true); }
  else { if (has_res) BN_BWD_APPLY(BN_ACT_RELU, true, false); else BN_BWD_APPLY(BN_ACT_RELU, false, false); }
#undef BN_BWD_APPLY
}

extern "C" int dd_bn_act_fwd_t(const void* x, const void* residual, long long rows, int C, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd, int act,
                               void* out, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !gamma || !beta || !save_mean || !save_invstd || !out || !workspace || !bn_dims_ok(rows, C) || act < 0 || act > 2 || dtype < 0 || dtype > 2)
    return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_bn_workspace_bytes(C) || (running_mean == nullptr) != (running_var == nullptr)) return (int)hipErrorInvalidValue;
  if (act == BN_ACT_GELU && residual) return (int)hipErrorInvalidValue;          // not a combination of the reference's networks
  DD_DISPATCH_DTYPE(dtype, bn_fwd_launch, x, residual, rows, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, act, out,
                    static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
  return (int)hipGetLastError();
}

extern "C" int dd_bn_act_bwd_t(const void* x, const void* g_out, const void* out, long long rows, int C, const float* gamma, const float* beta,
                               const float* save_mean, const float* save_invstd, int act, void* g_x, void* g_residual, float* g_gamma,
                               float* g_beta, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !g_out || !gamma || !beta || !save_mean || !save_invstd || !g_x || !g_gamma || !g_beta || !workspace || !bn_dims_ok(rows, C) ||
      act < 0 || act > 2 || (act == BN_ACT_GELU && g_residual) || dtype < 0 || dtype > 2)
    return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_bn_workspace_bytes(C)) return (int)hipErrorInvalidValue;
  DD_DISPATCH_DTYPE(dtype, bn_bwd_launch, x, g_out, out, rows, C, gamma, beta, save_mean, save_invstd, act, g_x, g_residual, g_gamma, g_beta,
                    static_cast<float*>(workspace), static_cast<hipStream_t>(stream));
  return (int)hipGetLastError();
}

extern "C" int dd_bn_act_fwd(const float* x, const float* residual, long long rows, int C, const float* gamma, const float* beta, float eps,
                             float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd, int act,
                             float* out, void* workspace, size_t workspace_bytes, void* stream) {
  return dd_bn_act_fwd_t(x, residual, rows, C, gamma, beta, eps, momentum, running_mean, running_var, save_mean, save_invstd, act, out, 0, workspace,
                         workspace_bytes, stream);
}

extern "C" int dd_bn_act_bwd(const float* x, const float* g_out, const float* out, long long rows, int C, const float* gamma, const float* beta,
                             const float* save_mean, const float* save_invstd, int act, float* g_x, float* g_residual, float* g_gamma,
                             float* g_beta, void* workspace, size_t workspace_bytes, void* stream) {
  return dd_bn_act_bwd_t(x, g_out, out, rows, C, gamma, beta, save_mean, save_invstd, act, g_x, g_residual, g_gamma, g_beta, 0, workspace,
                         workspace_bytes, stream);
}
