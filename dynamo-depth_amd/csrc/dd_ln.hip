// LayerNorm over the channel axis of channels-last tensors with few channels (LiteMono's LGFI blocks: C = 64 / 128 / 224,
// reference networks/depth_encoder.py:101-128 `LayerNorm(data_format="channels_last")`, applied at :241 and :252).
// ATen's row-per-block kernel needs ~100 us for a 23.6 MB tensor whose rows are 256 bytes long (12 x 48 x 160 rows of 64
// floats), and three kernels for the backward.  Here a row is a group of LP = 16/32/64 lanes of one wave (float4 per lane),
// row moments are xor-shuffle reductions inside the group, every pass is one coalesced stream, and the weight / bias
// gradients are fixed-order column sums (per-thread over its rows -> LDS across the block's rows -> one record per block
// -> fold).  Forward: 1 launch.  Backward: 2 launches.
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_half.h"

namespace dd {

constexpr int LN_NT = 256;
constexpr int LN_MAX_BLOCKS = 512;
constexpr int LN_ROWS_PER_THREAD = 8;

template <int LP>
__device__ __forceinline__ float ln_group_sum(float v) {
#pragma unroll
  for (int o = LP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// T: storage type of x / y (fp32, or fp16 / bf16 under autocast); statistics, affine parameters and all arithmetic fp32
template <int LP, typename T>
__global__ __launch_bounds__(LN_NT) void ln_fwd_kernel(const T* __restrict__ x, long long rows, int C, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, T* __restrict__ y,
                                                       float* __restrict__ mean, float* __restrict__ rstd) {
  constexpr int RPB = LN_NT / LP;                              // rows per block and pass
  const int C4 = C >> 2;
  const int lane = threadIdx.x % LP, rib = threadIdx.x / LP;
  const bool active = lane < C4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 ga = active ? reinterpret_cast<const float4*>(gamma)[lane] : zero;
  const float4 be = active ? reinterpret_cast<const float4*>(beta)[lane] : zero;
  const float inv_c = 1.f / static_cast<float>(C);
  for (long long r = (long long)blockIdx.x * RPB + rib; r < rows; r += (long long)gridDim.x * RPB) {
    const float4 v = active ? IO<T>::load4(x, r * C4 + lane) : zero;
    const float mu = ln_group_sum<LP>((v.x + v.y) + (v.z + v.w)) * inv_c;
    float4 d = make_float4(v.x - mu, v.y - mu, v.z - mu, v.w - mu);
    if (!active) d = zero;
    const float var = ln_group_sum<LP>((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * inv_c;
    const float rs = 1.f / sqrtf(var + eps);
    if (active)
      IO<T>::store4(y, r * C4 + lane, make_float4(fmaf(d.x * rs, ga.x, be.x), fmaf(d.y * rs, ga.y, be.y), fmaf(d.z * rs, ga.z, be.z),
                                                  fmaf(d.w * rs, ga.w, be.w)));
    if (lane == 0) {
      mean[r] = mu;
      rstd[r] = rs;
    }
  }
}

// dx = rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat)); per-block column sums of g*xhat and g
template <int LP, typename T>
__global__ __launch_bounds__(LN_NT) void ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, const float* __restrict__ gamma,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, long long rows, int C,
                                                       T* __restrict__ gx, float* __restrict__ partial) {
  constexpr int RPB = LN_NT / LP;
  __shared__ float4 red[2 * LN_NT];                            // [2][RPB][LP] float4
  const int C4 = C >> 2;
  const int lane = threadIdx.x % LP, rib = threadIdx.x / LP;
  const bool active = lane < C4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 ga = active ? reinterpret_cast<const float4*>(gamma)[lane] : zero;
  const float inv_c = 1.f / static_cast<float>(C);
  float4 dgam = zero, dbet = zero;
  for (long long r = (long long)blockIdx.x * RPB + rib; r < rows; r += (long long)gridDim.x * RPB) {
    const float4 v = active ? IO<T>::load4(x, r * C4 + lane) : zero;
    const float4 go = active ? IO<T>::load4(g, r * C4 + lane) : zero;
    const float mu = mean[r], rs = rstd[r];
    float4 xh = make_float4((v.x - mu) * rs, (v.y - mu) * rs, (v.z - mu) * rs, (v.w - mu) * rs);
    if (!active) xh = zero;
    const float4 gg = make_float4(go.x * ga.x, go.y * ga.y, go.z * ga.z, go.w * ga.w);
    const float a = ln_group_sum<LP>((gg.x + gg.y) + (gg.z + gg.w)) * inv_c;
    const float b = ln_group_sum<LP>((gg.x * xh.x + gg.y * xh.y) + (gg.z * xh.z + gg.w * xh.w)) * inv_c;
    if (active)
      IO<T>::store4(gx, r * C4 + lane, make_float4(rs * (gg.x - a - xh.x * b), rs * (gg.y - a - xh.y * b), rs * (gg.z - a - xh.z * b),
                                                   rs * (gg.w - a - xh.w * b)));
    dgam.x = fmaf(go.x, xh.x, dgam.x); dgam.y = fmaf(go.y, xh.y, dgam.y); dgam.z = fmaf(go.z, xh.z, dgam.z); dgam.w = fmaf(go.w, xh.w, dgam.w);
    dbet.x += go.x; dbet.y += go.y; dbet.z += go.z; dbet.w += go.w;
  }
  red[rib * LP + lane] = dgam;
  red[LN_NT + rib * LP + lane] = dbet;
  __syncthreads();
  // one record per block: [2][C]; thread (which, c) adds the block's RPB row groups in order
  float* dst = partial + (size_t)blockIdx.x * 2 * C;
  const float* redf = reinterpret_cast<const float*>(red);
  for (int i = threadIdx.x; i < 2 * C; i += LN_NT) {
    const int which = i / C, c = i - which * C;
    float s = 0.f;
    for (int k = 0; k < RPB; ++k) s += redf[(size_t)(which * LN_NT + k * LP + (c >> 2)) * 4 + (c & 3)];
    dst[i] = s;
  }
}

// out[i] = sum over blocks of partial[b][i], i < 2C: one 64-lane wave per output, fixed order inside a lane, shuffle tree across
__global__ __launch_bounds__(LN_NT) void ln_fold_kernel(const float* __restrict__ partial, int nblocks, int n, float* __restrict__ out) {
  const int o = blockIdx.x * (LN_NT / 64) + (threadIdx.x >> 6), l = threadIdx.x & 63;
  if (o >= n) return;
  float s = 0.f;
  for (int b = l; b < nblocks; b += 64) s += partial[(size_t)b * n + o];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (l == 0) out[o] = s;
}

static inline int ln_blocks(long long rows, int rpb) {
  long long want = (rows + (long long)rpb * LN_ROWS_PER_THREAD - 1) / ((long long)rpb * LN_ROWS_PER_THREAD);
  return (int)(want < 1 ? 1 : (want > LN_MAX_BLOCKS ? LN_MAX_BLOCKS : want));
}
static inline bool ln_dims_ok(long long rows, int C) { return rows >= 1 && C >= 4 && (C & 3) == 0 && C <= 256; }
static inline int ln_lp(int C) { return (C >> 2) <= 16 ? 16 : ((C >> 2) <= 32 ? 32 : 64); }

}  // namespace dd

using namespace dd;

extern "C" size_t dd_layer_norm_workspace_bytes(int C) { return (size_t)LN_MAX_BLOCKS * 2 * C * sizeof(float); }

template <typename T>
static void ln_fwd_launch(const void* x, long long rows, int C, const float* gamma, const float* beta, float eps, void* y, float* mean, float* rstd,
                          hipStream_t s) {
  const int lp = ln_lp(C);
  const int blocks = ln_blocks(rows, LN_NT / lp);
  const T* xt = static_cast<const T*>(x);
  T* yt = static_cast<T*>(y);
  if (lp == 16) hipLaunchKernelGGL((ln_fwd_kernel<16, T>), dim3(blocks), dim3(LN_NT), 0, s, xt, rows, C, gamma, beta, eps, yt, mean, rstd);
  else if (lp == 32) hipLaunchKernelGGL((ln_fwd_kernel<32, T>), dim3(blocks), dim3(LN_NT), 0, s, xt, rows, C, gamma, beta, eps, yt, mean, rstd);
  else hipLaunchKernelGGL((ln_fwd_kernel<64, T>), dim3(blocks), dim3(LN_NT), 0, s, xt, rows, C, gamma, beta, eps, yt, mean, rstd);
}

template <typename T>
static void ln_bwd_launch(const void* x, const void* g, const float* gamma, const float* mean, const float* rstd, long long rows, int C, void* gx,
                          float* partial, int blocks, hipStream_t s) {
  const int lp = ln_lp(C);
  const T* xt = static_cast<const T*>(x);
  const T* gt = static_cast<const T*>(g);
  T* gxt = static_cast<T*>(gx);
  if (lp == 16) hipLaunchKernelGGL((ln_bwd_kernel<16, T>), dim3(blocks), dim3(LN_NT), 0, s, xt, gt, gamma, mean, rstd, rows, C, gxt, partial);
  else if (lp == 32) hipLaunchKernelGGL((ln_bwd_kernel<32, T>), dim3(blocks), dim3(LN_NT), 0, s, xt, gt, gamma, mean, rstd, rows, C, gxt, partial);
  else hipLaunchKernelGGL((ln_bwd_kernel<64, T>), dim3(blocks), dim3(LN_NT), 0, s, xt, gt, gamma, mean, rstd, rows, C, gxt, partial);
}

extern "C" int dd_layer_norm_fwd_t(const void* x, long long rows, int C, const float* gamma, const float* beta, float eps, void* y, float* mean,
                                   float* rstd, int dtype, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || !ln_dims_ok(rows, C) || dtype < 0 || dtype > 2) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DD_DISPATCH_DTYPE(dtype, ln_fwd_launch, x, rows, C, gamma, beta, eps, y, mean, rstd, s);
  return (int)hipGetLastError();
}

extern "C" int dd_layer_norm_fwd(const float* x, long long rows, int C, const float* gamma, const float* beta, float eps, float* y,
                                 float* mean, float* rstd, void* stream) {
  return dd_layer_norm_fwd_t(x, rows, C, gamma, beta, eps, y, mean, rstd, 0, stream);
}

extern "C" int dd_layer_norm_bwd_t(const void* x, const void* g_out, const float* gamma, const float* mean, const float* rstd, long long rows, int C,
                                   void* g_x, float* g_gamma_beta, void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  if (!x || !g_out || !gamma || !mean || !rstd || !g_x || !g_gamma_beta || !workspace || !ln_dims_ok(rows, C) || dtype < 0 || dtype > 2)
    return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_layer_norm_workspace_bytes(C)) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int blocks = ln_blocks(rows, LN_NT / ln_lp(C));
  float* partial = static_cast<float*>(workspace);
  DD_DISPATCH_DTYPE(dtype, ln_bwd_launch, x, g_out, gamma, mean, rstd, rows, C, g_x, partial, blocks, s);
  const int n = 2 * C;
  hipLaunchKernelGGL(ln_fold_kernel, dim3((n + LN_NT / 64 - 1) / (LN_NT / 64)), dim3(LN_NT), 0, s, partial, blocks, n, g_gamma_beta);
  return (int)hipGetLastError();
}

extern "C" int dd_layer_norm_bwd(const float* x, const float* g_out, const float* gamma, const float* mean, const float* rstd, long long rows,
                                 int C, float* g_x, float* g_gamma_beta, void* workspace, size_t workspace_bytes, void* stream) {
  return dd_layer_norm_bwd_t(x, g_out, gamma, mean, rstd, rows, C, g_x, g_gamma_beta, workspace, workspace_bytes, 0, stream);
}

// ---- backward of the layer-scale residual of LiteMono's blocks: out = res + y * scale[b, c]  (scale = gamma * drop-path factor;
// reference networks/depth_encoder.py:219-226,266-274) ------------------------------------------------------------------------
// g_y = g * scale (written contiguous, ready for the Linear's GEMMs) and g_scale[b, c] = sum over the image of g * y, in one pass
// over g and y; ATen needs two broadcast multiplies, a reduction and (often) a layout copy.  Fixed-order sums.
namespace dd {

constexpr int LS_NT = 256;
constexpr int LS_MAX_CHUNKS = 64;                              // per image

template <typename T>
__global__ __launch_bounds__(LS_NT) void layer_scale_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y,
                                                                const float* __restrict__ scale, int rows, int C, int lanes,
                                                                int rows_per_chunk, T* __restrict__ gy, float* __restrict__ partial) {
  extern __shared__ float red[];                               // [lanes][C]
  const int C4 = C >> 2;
  const int lane = threadIdx.x / C4, c4 = threadIdx.x - lane * C4;
  const int b = blockIdx.y;
  const float4 sc = reinterpret_cast<const float4*>(scale + (size_t)b * C)[c4];
  const long long base = (long long)b * rows * C4;
  const int r0 = blockIdx.x * rows_per_chunk;
  int r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int r = r0 + lane; r < r1; r += lanes) {
    const long long i4 = base + (long long)r * C4 + c4;
    const float4 a = IO<T>::load4(g, i4), v = IO<T>::load4(y, i4);
    IO<T>::store4(gy, i4, make_float4(a.x * sc.x, a.y * sc.y, a.z * sc.z, a.w * sc.w));
    acc.x = fmaf(a.x, v.x, acc.x); acc.y = fmaf(a.y, v.y, acc.y); acc.z = fmaf(a.z, v.z, acc.z); acc.w = fmaf(a.w, v.w, acc.w);
  }
  reinterpret_cast<float4*>(red)[lane * C4 + c4] = acc;
  __syncthreads();
  float* dst = partial + ((size_t)b * gridDim.x + blockIdx.x) * C;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int l = 0; l < lanes; ++l) s += red[l * C + c];
    dst[c] = s;
  }
}

__global__ __launch_bounds__(LS_NT) void layer_scale_fold_kernel(const float* __restrict__ partial, int chunks, int C, float* __restrict__ gscale) {
  const int c = blockIdx.x * LS_NT + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < chunks; ++k) s += partial[((size_t)b * chunks + k) * C + c];
  gscale[(size_t)b * C + c] = s;
}

}  // namespace dd

// forward of the same block: out = res + y * scale[b, c] with fp32 arithmetic whatever the storage type (the layer-scale
// parameters start at 1e-6, below fp16's normal range: a half-precision addcmul would lose them)
namespace dd {
template <typename T>
__global__ __launch_bounds__(LS_NT) void layer_scale_fwd_kernel(const T* __restrict__ res, const T* __restrict__ y, const float* __restrict__ scale,
                                                                long long per_image4, int C4, long long total4, T* __restrict__ out) {
  const long long i = (long long)blockIdx.x * LS_NT + threadIdx.x;
  if (i >= total4) return;
  const int b = (int)(i / per_image4), c4 = (int)(i % C4);
  const float4 sc = reinterpret_cast<const float4*>(scale)[(long long)b * C4 + c4];
  const float4 r = IO<T>::load4(res, i), v = IO<T>::load4(y, i);
  IO<T>::store4(out, i, make_float4(fmaf(v.x, sc.x, r.x), fmaf(v.y, sc.y, r.y), fmaf(v.z, sc.z, r.z), fmaf(v.w, sc.w, r.w)));
}
template <typename T>
static void layer_scale_fwd_launch(const void* res, const void* y, const float* scale, long long per_image4, int C4, long long total4, void* out,
                                   hipStream_t s) {
  hipLaunchKernelGGL((layer_scale_fwd_kernel<T>), dim3((unsigned)((total4 + LS_NT - 1) / LS_NT)), dim3(LS_NT), 0, s, static_cast<const T*>(res),
                     static_cast<const T*>(y), scale, per_image4, C4, total4, static_cast<T*>(out));
}
}  // namespace dd

extern "C" int dd_layer_scale_fwd_t(const void* res, const void* y, const float* scale, int B, int rows, int C, void* out, int dtype, void* stream) {
  if (!res || !y || !scale || !out || B < 1 || rows < 1 || C < 4 || (C & 3) || dtype < 0 || dtype > 2) return (int)hipErrorInvalidValue;
  const int C4 = C >> 2;
  const long long per_image4 = (long long)rows * C4, total4 = per_image4 * B;
  DD_DISPATCH_DTYPE(dtype, dd::layer_scale_fwd_launch, res, y, scale, per_image4, C4, total4, out, static_cast<hipStream_t>(stream));
  return (int)hipGetLastError();
}

extern "C" size_t dd_layer_scale_workspace_bytes(int B, int C) { return (size_t)B * dd::LS_MAX_CHUNKS * C * sizeof(float); }

template <typename T>
static void layer_scale_launch(const void* g_out, const void* y, const float* scale, int rows, int C, int lanes, int per, void* g_y, float* partial,
                               dim3 grid, int threads, hipStream_t s) {
  hipLaunchKernelGGL((dd::layer_scale_bwd_kernel<T>), grid, dim3(threads), (size_t)lanes * C * sizeof(float), s, static_cast<const T*>(g_out),
                     static_cast<const T*>(y), scale, rows, C, lanes, per, static_cast<T*>(g_y), partial);
}

extern "C" int dd_layer_scale_bwd_t(const void* g_out, const void* y, const float* scale, int B, int rows, int C, void* g_y, float* g_scale,
                                    void* workspace, size_t workspace_bytes, int dtype, void* stream) {
  if (!g_out || !y || !scale || !g_y || !g_scale || !workspace || B < 1 || B > 65535 || rows < 1 || C < 4 || (C & 3) || C > 1024 || dtype < 0 || dtype > 2)
    return (int)hipErrorInvalidValue;
  if (workspace_bytes < dd_layer_scale_workspace_bytes(B, C)) return (int)hipErrorInvalidValue;
  const int C4 = C >> 2;
  int lanes = dd::LS_NT / C4;
  if (lanes < 1) lanes = 1;
  const int threads = lanes * C4;
  if (threads > 1024) return (int)hipErrorInvalidValue;
  int chunks = (rows + lanes * 8 - 1) / (lanes * 8);
  if (chunks > dd::LS_MAX_CHUNKS) chunks = dd::LS_MAX_CHUNKS;
  const int per = (rows + chunks - 1) / chunks;
  chunks = (rows + per - 1) / per;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(workspace);
  DD_DISPATCH_DTYPE(dtype, layer_scale_launch, g_out, y, scale, rows, C, lanes, per, g_y, partial, dim3(chunks, B), threads, s);
  hipLaunchKernelGGL(dd::layer_scale_fold_kernel, dim3((C + dd::LS_NT - 1) / dd::LS_NT, B), dim3(dd::LS_NT), 0, s, partial, chunks, C, g_scale);
  return (int)hipGetLastError();
}

extern "C" int dd_layer_scale_bwd(const float* g_out, const float* y, const float* scale, int B, int rows, int C, float* g_y, float* g_scale,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  return dd_layer_scale_bwd_t(g_out, y, scale, B, rows, C, g_y, g_scale, workspace, workspace_bytes, 0, stream);
}
