// dd_ops.hip -- the reference's tools.py operators one by one (forward and autograd backward), for callers
// that use the operator surface directly (eval scripts, vis_motion, user code) instead of the fused loss:
//   BackprojectDepth (tools.py:167-197)   Project3D (tools.py:200-224)   SSIM (tools.py:227-257)
//   disp_to_depth (tools.py:291-298)      transformation_from_parameters (networks/layers.py:7-82)
// Element-wise / 3x3-stencil kernels over NCHW fp32, one thread per pixel, coalesced along W.
#include <hip/hip_runtime.h>

#include "../../include/dynamo_hip.h"
#include "dd_math.h"
#include "dd_half.h"

namespace dd {

constexpr int OP_NT = 256;

__device__ __forceinline__ float wsum_ops(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- BackprojectDepth ------------------------------------------------------------------------------
__global__ __launch_bounds__(OP_NT) void backproject_kernel(const float* __restrict__ depth, const float* __restrict__ inv_K,
                                                             int h, int w, float* __restrict__ points) {
  const int b = blockIdx.y, n = h * w, p = blockIdx.x * OP_NT + threadIdx.x;
  if (p >= n) return;
  const float* A = inv_K + b * 16;
  const float x = static_cast<float>(p % w), y = static_cast<float>(p / w), d = depth[(size_t)b * n + p];
  float* out = points + (size_t)b * 4 * n + p;
#pragma unroll
  for (int i = 0; i < 3; ++i) out[(size_t)i * n] = d * (A[i * 4 + 0] * x + A[i * 4 + 1] * y + A[i * 4 + 2]);
  out[(size_t)3 * n] = 1.f;
}

__global__ __launch_bounds__(OP_NT) void backproject_bwd_kernel(const float* __restrict__ g_points, const float* __restrict__ inv_K,
                                                                 int h, int w, float* __restrict__ g_depth) {
  const int b = blockIdx.y, n = h * w, p = blockIdx.x * OP_NT + threadIdx.x;
  if (p >= n) return;
  const float* A = inv_K + b * 16;
  const float x = static_cast<float>(p % w), y = static_cast<float>(p / w);
  const float* g = g_points + (size_t)b * 4 * n + p;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) acc += g[(size_t)i * n] * (A[i * 4 + 0] * x + A[i * 4 + 1] * y + A[i * 4 + 2]);
  g_depth[(size_t)b * n + p] = acc;
}

// ---- Project3D -------------------------------------------------------------------------------------
template <bool HAS_T>
__global__ __launch_bounds__(OP_NT) void project_kernel(const float* __restrict__ points, const float* __restrict__ K,
                                                         const float* __restrict__ T, int h, int w, float eps,
                                                         float* __restrict__ pix, float* __restrict__ ego) {
  const int b = blockIdx.y, n = h * w, p = blockIdx.x * OP_NT + threadIdx.x;
  if (p >= n) return;
  const float* pp = points + (size_t)b * 4 * n + p;
  const float P[4] = {pp[0], pp[(size_t)n], pp[(size_t)2 * n], pp[(size_t)3 * n]};
  float Q[4];
  if (HAS_T) {
    const float* Tb = T + b * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) Q[i] = Tb[i * 4] * P[0] + Tb[i * 4 + 1] * P[1] + Tb[i * 4 + 2] * P[2] + Tb[i * 4 + 3] * P[3];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) Q[i] = P[i];
  }
  const float* Kb = K + b * 16;
  float c[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) c[j] = Kb[j * 4] * Q[0] + Kb[j * 4 + 1] * Q[1] + Kb[j * 4 + 2] * Q[2] + Kb[j * 4 + 3] * Q[3];
  const float inv = 1.f / (c[2] + eps);
  reinterpret_cast<float2*>(pix)[(size_t)b * n + p] = make_float2(grid_normalise(c[0] * inv, 1.f / static_cast<float>(w - 1)), grid_normalise(c[1] * inv, 1.f / static_cast<float>(h - 1)));
#pragma unroll
  for (int i = 0; i < 3; ++i) ego[((size_t)b * 3 + i) * n + p] = Q[i] - P[i];
}

template <bool HAS_T>
__global__ __launch_bounds__(OP_NT) void project_bwd_kernel(const float* __restrict__ points, const float* __restrict__ K,
                                                             const float* __restrict__ T, const float* __restrict__ g_pix,
                                                             const float* __restrict__ g_ego, int h, int w, float eps,
                                                             float* __restrict__ g_points, float* __restrict__ partials) {
  __shared__ float red[(OP_NT / 64) * 16];
  const int b = blockIdx.y, n = h * w, p = blockIdx.x * OP_NT + threadIdx.x;
  float gT[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) gT[k] = 0.f;
  if (p < n) {
    const float* pp = points + (size_t)b * 4 * n + p;
    const float P[4] = {pp[0], pp[(size_t)n], pp[(size_t)2 * n], pp[(size_t)3 * n]};
    float Q[4];
    const float* Tb = HAS_T ? T + b * 16 : nullptr;
    if (HAS_T) {
#pragma unroll
      for (int i = 0; i < 4; ++i) Q[i] = Tb[i * 4] * P[0] + Tb[i * 4 + 1] * P[1] + Tb[i * 4 + 2] * P[2] + Tb[i * 4 + 3] * P[3];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) Q[i] = P[i];
    }
    const float* Kb = K + b * 16;
    float c[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) c[j] = Kb[j * 4] * Q[0] + Kb[j * 4 + 1] * Q[1] + Kb[j * 4 + 2] * Q[2] + Kb[j * 4 + 3] * Q[3];
    const float inv = 1.f / (c[2] + eps);
    const float u = c[0] * inv, v = c[1] * inv;
    float gu = 0.f, gv = 0.f;
    if (g_pix) {
      const float2 g2 = reinterpret_cast<const float2*>(g_pix)[(size_t)b * n + p];
      gu = g2.x * 2.f / static_cast<float>(w - 1);
      gv = g2.y * 2.f / static_cast<float>(h - 1);
    }
    const float gc[3] = {gu * inv, gv * inv, -(gu * u + gv * v) * inv};
    float gQ[4], ge[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) gQ[k] = gc[0] * Kb[k] + gc[1] * Kb[4 + k] + gc[2] * Kb[8 + k];
    if (g_ego) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { ge[i] = g_ego[((size_t)b * 3 + i) * n + p]; gQ[i] += ge[i]; }
    }
    float gP[4];
    if (HAS_T) {
#pragma unroll
      for (int k = 0; k < 4; ++k) gP[k] = Tb[k] * gQ[0] + Tb[4 + k] * gQ[1] + Tb[8 + k] * gQ[2] + Tb[12 + k] * gQ[3];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) gT[i * 4 + k] = gQ[i] * P[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) gP[k] = gQ[k];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) gP[i] -= ge[i];
    float* go = g_points + (size_t)b * 4 * n + p;
#pragma unroll
    for (int k = 0; k < 4; ++k) go[(size_t)k * n] = gP[k];
  }
  if (HAS_T) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float r = wsum_ops(gT[k]);
      if (lane == 0) red[wave * 16 + k] = r;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
      float r = 0.f;
#pragma unroll
      for (int wv = 0; wv < OP_NT / 64; ++wv) r += red[wv * 16 + threadIdx.x];
      partials[((size_t)b * gridDim.x + blockIdx.x) * 16 + threadIdx.x] = r;
    }
  }
}

__global__ __launch_bounds__(256) void fold16_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ out) {
  __shared__ float red[4 * 16];
  const int b = blockIdx.x;
  float acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] += partials[((size_t)b * nblk + i) * 16 + k];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float r = wsum_ops(acc[k]);
    if (lane == 0) red[wave * 16 + k] = r;
  }
  __syncthreads();
  if (threadIdx.x < 16) out[b * 16 + threadIdx.x] = red[threadIdx.x] + red[16 + threadIdx.x] + red[32 + threadIdx.x] + red[48 + threadIdx.x];
}

// ---- SSIM ------------------------------------------------------------------------------------------
__device__ __forceinline__ SsimStats window_stats(const float* __restrict__ x, const float* __restrict__ y, int X, int Y, int W, int H) {
  SsimStats st = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = -1; j <= 1; ++j) {
    const int ry = dd_reflect(Y + j, H) * W;
#pragma unroll
    for (int i = -1; i <= 1; ++i) {
      const int q = ry + dd_reflect(X + i, W);
      const float xv = x[q], yv = y[q];
      st.sx += xv; st.sy += yv; st.sxx += xv * xv; st.syy += yv * yv; st.sxy += xv * yv;
    }
  }
  return st;
}

__global__ __launch_bounds__(OP_NT) void ssim_kernel(const float* __restrict__ x, const float* __restrict__ y, int H, int W,
                                                      float* __restrict__ out) {
  const int n = H * W, p = blockIdx.x * OP_NT + threadIdx.x;
  if (p >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  const SsimStats st = window_stats(x + base, y + base, p % W, p / W, W, H);
  out[base + p] = ssim_value(st, nullptr);
}

__global__ __launch_bounds__(OP_NT) void ssim_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          const float* __restrict__ g_out, int H, int W, float* __restrict__ g_x,
                                                          float* __restrict__ g_y) {
  const int n = H * W, p = blockIdx.x * OP_NT + threadIdx.x;
  if (p >= n) return;
  const size_t base = (size_t)blockIdx.y * n;
  const float* xb = x + base;
  const float* yb = y + base;
  const int X = p % W, Y = p / W;
  const float xv = xb[p], yv = yb[p];
  float gx = 0.f, gy = 0.f;
  for (int cy = Y - 1; cy <= Y + 1; ++cy) {
    if (cy < 0 || cy >= H) continue;
    const int my = reflect_multiplicity(cy, Y, H);
    for (int cx = X - 1; cx <= X + 1; ++cx) {
      if (cx < 0 || cx >= W) continue;
      const int mult = my * reflect_multiplicity(cx, X, W);
      if (mult == 0) continue;
      const float go = g_out[base + cy * W + cx] * static_cast<float>(mult) / 9.f;
      const SsimStats st = window_stats(xb, yb, cx, cy, W, H);
      SsimGrad sg;
      ssim_value(st, &sg);
      gx += go * (sg.dmu + 2.f * sg.dxx * xv + sg.dxy * yv);
      if (g_y) {
        const SsimStats sw = {st.sy, st.sx, st.syy, st.sxx, st.sxy};   // the formula is symmetric in (x, y)
        ssim_value(sw, &sg);
        gy += go * (sg.dmu + 2.f * sg.dxx * yv + sg.dxy * xv);
      }
    }
  }
  if (g_x) g_x[base + p] = gx;
  if (g_y) g_y[base + p] = gy;
}

// ---- disp_to_depth ---------------------------------------------------------------------------------
__global__ __launch_bounds__(OP_NT) void disp_to_depth_kernel(const float* __restrict__ disp, size_t n, DepthParams dp,
                                                               float* __restrict__ scaled, float* __restrict__ depth) {
  for (size_t i = (size_t)blockIdx.x * OP_NT + threadIdx.x; i < n; i += (size_t)gridDim.x * OP_NT) {
    const float s = dp.lo + dp.span * disp[i];
    if (scaled) scaled[i] = s;
    if (depth) depth[i] = 1.f / s;
  }
}

// ---- pose vector -> 4x4 ------------------------------------------------------------------------------
struct Rodrigues {
  float n[3], ca, sa, C, theta, R[9];
};

DD_HD void rodrigues(const float v[3], Rodrigues& r) {
  r.theta = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float inv = 1.f / (r.theta + 1e-7f);
  for (int i = 0; i < 3; ++i) r.n[i] = v[i] * inv;
  r.ca = cosf(r.theta);
  r.sa = sinf(r.theta);
  r.C = 1.f - r.ca;
  const float x = r.n[0], y = r.n[1], z = r.n[2];
  r.R[0] = x * x * r.C + r.ca;       r.R[1] = x * y * r.C - z * r.sa;   r.R[2] = z * x * r.C + y * r.sa;
  r.R[3] = x * y * r.C + z * r.sa;   r.R[4] = y * y * r.C + r.ca;       r.R[5] = y * z * r.C - x * r.sa;
  r.R[6] = z * x * r.C - y * r.sa;   r.R[7] = y * z * r.C + x * r.sa;   r.R[8] = z * z * r.C + r.ca;
}

__global__ void pose_matrix_kernel(const float* __restrict__ aa, const float* __restrict__ tr, int B, int invert, float* __restrict__ T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  Rodrigues r;
  const float v[3] = {aa[b * 3], aa[b * 3 + 1], aa[b * 3 + 2]};
  const float t[3] = {tr[b * 3], tr[b * 3 + 1], tr[b * 3 + 2]};
  rodrigues(v, r);
  float* M = T + b * 16;
  for (int i = 0; i < 3; ++i) {
    float last = invert ? 0.f : t[i];
    for (int j = 0; j < 3; ++j) {
      M[i * 4 + j] = invert ? r.R[j * 3 + i] : r.R[i * 3 + j];
      if (invert) last -= r.R[j * 3 + i] * t[j];
    }
    M[i * 4 + 3] = last;
  }
  M[12] = M[13] = M[14] = 0.f;
  M[15] = 1.f;
}

__global__ void pose_matrix_bwd_kernel(const float* __restrict__ aa, const float* __restrict__ tr, const float* __restrict__ g_T, int B,
                                       int invert, float* __restrict__ g_aa, float* __restrict__ g_tr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  Rodrigues r;
  const float v[3] = {aa[b * 3], aa[b * 3 + 1], aa[b * 3 + 2]};
  const float t[3] = {tr[b * 3], tr[b * 3 + 1], tr[b * 3 + 2]};
  rodrigues(v, r);
  const float* g = g_T + b * 16;
  float gR[9], gt[3];
  for (int i = 0; i < 3; ++i) {
    if (invert) {
      gt[i] = 0.f;
      for (int k = 0; k < 3; ++k) gt[i] -= g[k * 4 + 3] * r.R[i * 3 + k];    // M_k3 = -sum_i R_ik t_i
    } else {
      gt[i] = g[i * 4 + 3];
    }
    for (int j = 0; j < 3; ++j) {
      if (invert) gR[i * 3 + j] = g[j * 4 + i] - g[j * 4 + 3] * t[i];        // M_ji = R_ij ; M_j3 -= R_ij t_i
      else gR[i * 3 + j] = g[i * 4 + j];
    }
  }
  const float x = r.n[0], y = r.n[1], z = r.n[2];
  const float Kx[9] = {0.f, -z, y, z, 0.f, -x, -y, x, 0.f};
  float gth = 0.f;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      gth += gR[i * 3 + j] * (r.n[i] * r.n[j] * r.sa - (i == j ? r.sa : 0.f) + r.ca * Kx[i * 3 + j]);
  float gn[3];
  for (int k = 0; k < 3; ++k) {
    float acc = 0.f;
    for (int j = 0; j < 3; ++j) acc += (gR[k * 3 + j] + gR[j * 3 + k]) * r.n[j];
    gn[k] = r.C * acc;
  }
  gn[0] += r.sa * (gR[7] - gR[5]);
  gn[1] += r.sa * (gR[2] - gR[6]);
  gn[2] += r.sa * (gR[3] - gR[1]);
  const float inv = 1.f / (r.theta + 1e-7f);
  const float dot = gn[0] * v[0] + gn[1] * v[1] + gn[2] * v[2];
  for (int k = 0; k < 3; ++k) {
    const float dth = r.theta > 0.f ? v[k] / r.theta : 0.f;            // torch.norm backward: 0 at the origin
    g_aa[b * 3 + k] = gn[k] * inv - dot * inv * inv * dth + gth * dth;
    g_tr[b * 3 + k] = gt[k];
  }
}

// ---- per-channel sum of an NHWC (channels-last) tensor: the bias gradient of a convolution -----------------------------
// x is read as one flat, fully coalesced stream; a thread strides by a multiple of C, so it only ever sees one channel.
// (ATen's generic reduce_kernel needs 1.9 ms for a (12,9,192,640) channels-last gradient -- 53 MB; this is ~20 us.)
constexpr int CS_NT = 256;
constexpr int CS_MAX_BLOCKS = 2048;

template <typename T>
__global__ __launch_bounds__(CS_NT) void channel_sum_partial_kernel(const T* __restrict__ x, long long total, int C,
                                                                    long long stride /* multiple of C */, float* __restrict__ partials) {
  __shared__ float s_vals[CS_NT];
  const long long gid = (long long)blockIdx.x * CS_NT + threadIdx.x;
  float acc = 0.f;
  if (gid < stride)
    for (long long i = gid; i < total; i += stride) acc += IO<T>::load1(x, i);
  s_vals[threadIdx.x] = acc;
  __syncthreads();
  // thread t holds channel (blockIdx.x*CS_NT + t) % C; thread c < C folds the block's entries of channel c in index order
  if (threadIdx.x < C) {
    const int first = (int)(((threadIdx.x - (long long)blockIdx.x * CS_NT) % C + C) % C);
    float r = 0.f;
    for (int t = first; t < CS_NT; t += C) r += s_vals[t];
    partials[(size_t)blockIdx.x * C + threadIdx.x] = r;
  }
}

// wide matrices (C > CS_NT: the point-wise Linear bias gradients, C up to 1344): grid = (column blocks, row chunks), a thread owns
// one column of one chunk, four running sums in a fixed interleave
template <typename T>
__global__ __launch_bounds__(CS_NT) void channel_sum_wide_kernel(const T* __restrict__ x, long long rows, int C, int rows_per_chunk,
                                                                 float* __restrict__ partials) {
  const int c = blockIdx.x * CS_NT + threadIdx.x;
  if (c >= C) return;
  const long long r0 = (long long)blockIdx.y * rows_per_chunk;
  long long r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  long long r = r0;
  for (; r + 3 < r1; r += 4) {
    a0 += IO<T>::load1(x, r * C + c); a1 += IO<T>::load1(x, (r + 1) * C + c); a2 += IO<T>::load1(x, (r + 2) * C + c);
    a3 += IO<T>::load1(x, (r + 3) * C + c);
  }
  for (; r < r1; ++r) a0 += IO<T>::load1(x, r * C + c);
  partials[(size_t)blockIdx.y * C + c] = (a0 + a1) + (a2 + a3);
}

__global__ __launch_bounds__(CS_NT) void channel_sum_fold_kernel(const float* __restrict__ partials, int nblocks, int C, float* __restrict__ out) {
  __shared__ float red[CS_NT / 64];
  const int c = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += CS_NT) acc += partials[(size_t)i * C + c];
  acc = wsum_ops(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = red[0] + red[1] + red[2] + red[3];
}

// ---- ReflectionPad2d(1) on channels-last tensors (the pad in front of every decoder Conv3x3) ---------------------------
// ATen's reflection_pad2d returns an NCHW-contiguous tensor, which the NHWC conv then has to copy; forward and backward
// here stay channels-last and fully coalesced (consecutive threads walk C, then W).
constexpr int PAD_NT = 256;

// grid = (row chunks, padded rows, batch): one 32-bit division per element (64-bit div/mod chains made the first version
// slower than ATen + layout copy)
template <typename T>
__global__ __launch_bounds__(PAD_NT) void reflect_pad1_nhwc_kernel(const T* __restrict__ x, int H, int W, int C, T* __restrict__ out) {
  const int Wp = W + 2;
  const int yo = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * PAD_NT + threadIdx.x;          // position inside the padded row: xo*C + c
  if (j >= Wp * C) return;
  const int xo = j / C, c = j - xo * C;
  const int xi = dd_reflect(xo - 1, W), yi = dd_reflect(yo - 1, H);
  out[((size_t)b * (H + 2) + yo) * Wp * C + j] = x[(((size_t)b * H + yi) * W + xi) * C + c];
}

// adjoint: every input pixel gathers the padded positions that mirror onto it (1, 2 or 4 of them) -- no atomics
template <typename T>
__global__ __launch_bounds__(PAD_NT) void reflect_pad1_nhwc_bwd_kernel(const T* __restrict__ g, int H, int W, int C, T* __restrict__ gx) {
  const int Wp = W + 2, Hp = H + 2;
  const int yi = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * PAD_NT + threadIdx.x;          // position inside the row: xi*C + c
  if (j >= W * C) return;
  const int xi = j / C, c = j - xi * C;
  const int ys[2] = {yi + 1, yi == 1 ? 0 : (yi == H - 2 ? Hp - 1 : -1)};
  const int xs[2] = {xi + 1, xi == 1 ? 0 : (xi == W - 2 ? Wp - 1 : -1)};
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int d = 0; d < 2; ++d)
      if (ys[a] >= 0 && xs[d] >= 0) acc += IO<T>::load1(g, (((long long)b * Hp + ys[a]) * Wp + xs[d]) * C + c);
  IO<T>::store1(gx, ((long long)b * H + yi) * W * C + j, acc);
}

}  // namespace dd

using namespace dd;

static inline int ops_err() { return (int)hipGetLastError(); }
static inline dim3 pix_grid(int n, int B) { return dim3((n + OP_NT - 1) / OP_NT, B); }

extern "C" int dd_backproject(const float* depth, const float* inv_K, int B, int h, int w, float* points, void* stream) {
  if (!depth || !inv_K || !points) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(backproject_kernel, pix_grid(h * w, B), dim3(OP_NT), 0, static_cast<hipStream_t>(stream), depth, inv_K, h, w, points);
  return ops_err();
}

extern "C" int dd_backproject_bwd(const float* g_points, const float* inv_K, int B, int h, int w, float* g_depth, void* stream) {
  if (!g_points || !inv_K || !g_depth) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(backproject_bwd_kernel, pix_grid(h * w, B), dim3(OP_NT), 0, static_cast<hipStream_t>(stream), g_points, inv_K, h, w, g_depth);
  return ops_err();
}

extern "C" int dd_project3d(const float* points, const float* K, const float* T, int B, int h, int w, float eps, float* pix,
                            float* ego, void* stream) {
  if (!points || !K || !pix || !ego) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (T) hipLaunchKernelGGL((project_kernel<true>), pix_grid(h * w, B), dim3(OP_NT), 0, s, points, K, T, h, w, eps, pix, ego);
  else hipLaunchKernelGGL((project_kernel<false>), pix_grid(h * w, B), dim3(OP_NT), 0, s, points, K, T, h, w, eps, pix, ego);
  return ops_err();
}

extern "C" size_t dd_project3d_workspace_bytes(int B, int h, int w) {
  return (size_t)B * ((h * w + OP_NT - 1) / OP_NT) * 16 * sizeof(float);
}

extern "C" int dd_project3d_bwd(const float* points, const float* K, const float* T, const float* g_pix, const float* g_ego, int B,
                                int h, int w, float eps, float* g_points, float* g_T, float* workspace, void* stream) {
  if (!points || !K || !g_points || (T && (!g_T || !workspace))) return (int)hipErrorInvalidValue;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid = pix_grid(h * w, B);
  if (T) {
    hipLaunchKernelGGL((project_bwd_kernel<true>), grid, dim3(OP_NT), 0, s, points, K, T, g_pix, g_ego, h, w, eps, g_points, workspace);
    hipLaunchKernelGGL(fold16_kernel, dim3(B), dim3(256), 0, s, workspace, (int)grid.x, g_T);
  } else {
    hipLaunchKernelGGL((project_bwd_kernel<false>), grid, dim3(OP_NT), 0, s, points, K, T, g_pix, g_ego, h, w, eps, g_points, workspace);
  }
  return ops_err();
}

extern "C" int dd_ssim(const float* x, const float* y, int B, int C, int H, int W, float* out, void* stream) {
  if (!x || !y || !out || H < 2 || W < 2) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ssim_kernel, pix_grid(H * W, B * C), dim3(OP_NT), 0, static_cast<hipStream_t>(stream), x, y, H, W, out);
  return ops_err();
}

extern "C" int dd_ssim_bwd(const float* x, const float* y, const float* g_out, int B, int C, int H, int W, float* g_x, float* g_y,
                           void* stream) {
  if (!x || !y || !g_out || H < 2 || W < 2) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ssim_bwd_kernel, pix_grid(H * W, B * C), dim3(OP_NT), 0, static_cast<hipStream_t>(stream), x, y, g_out, H, W, g_x, g_y);
  return ops_err();
}

extern "C" int dd_disp_to_depth(const float* disp, size_t n, float min_depth, float max_depth, float* scaled, float* depth,
                                void* stream) {
  if (!disp) return (int)hipErrorInvalidValue;
  const int blocks = (int)((n + OP_NT - 1) / OP_NT < 4096 ? (n + OP_NT - 1) / OP_NT : 4096);
  hipLaunchKernelGGL(disp_to_depth_kernel, dim3(blocks > 0 ? blocks : 1), dim3(OP_NT), 0, static_cast<hipStream_t>(stream), disp, n,
                     depth_params(min_depth, max_depth), scaled, depth);
  return ops_err();
}

extern "C" int dd_pose_matrix(const float* axisangle, const float* translation, int B, int invert, float* T, void* stream) {
  if (!axisangle || !translation || !T) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(pose_matrix_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), axisangle, translation, B, invert, T);
  return ops_err();
}

extern "C" int dd_pose_matrix_bwd(const float* axisangle, const float* translation, const float* g_T, int B, int invert,
                                  float* g_axisangle, float* g_translation, void* stream) {
  if (!axisangle || !translation || !g_T || !g_axisangle || !g_translation) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(pose_matrix_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), axisangle, translation,
                     g_T, B, invert, g_axisangle, g_translation);
  return ops_err();
}

extern "C" size_t dd_channel_sum_workspace_bytes(int C) { return (size_t)CS_MAX_BLOCKS * C * sizeof(float); }

template <typename T>
static void channel_sum_launch(const void* x_, long long rows, int C, float* out, float* workspace, hipStream_t s) {
  const T* x = static_cast<const T*>(x_);
  if (C > CS_NT) {
    long long want = (rows + 31) / 32;
    const int chunks = (int)(want > CS_MAX_BLOCKS ? CS_MAX_BLOCKS : want);
    const int per = (int)((rows + chunks - 1) / chunks);
    hipLaunchKernelGGL(channel_sum_wide_kernel<T>, dim3((C + CS_NT - 1) / CS_NT, chunks), dim3(CS_NT), 0, s, x, rows, C, per, workspace);
    hipLaunchKernelGGL(channel_sum_fold_kernel, dim3(C), dim3(CS_NT), 0, s, workspace, chunks, C, out);
    return;
  }
  const long long total = rows * C;
  long long want = (total + CS_NT * 8 - 1) / (CS_NT * 8);           // >= 8 elements per thread
  int blocks = (int)(want < 1 ? 1 : (want > CS_MAX_BLOCKS ? CS_MAX_BLOCKS : want));
  long long stride = ((long long)blocks * CS_NT / C) * C;
  if (stride < C) stride = C;                                        // tiny tensors: C > threads in use
  hipLaunchKernelGGL(channel_sum_partial_kernel<T>, dim3(blocks), dim3(CS_NT), 0, s, x, total, C, stride, workspace);
  hipLaunchKernelGGL(channel_sum_fold_kernel, dim3(C), dim3(CS_NT), 0, s, workspace, blocks, C, out);
}

extern "C" int dd_channel_sum_nhwc_t(const void* x, long long rows, int C, float* out, int dtype, float* workspace, void* stream) {
  if (!x || !out || !workspace || rows < 1 || C < 1 || dtype < 0 || dtype > 2) return (int)hipErrorInvalidValue;
  DD_DISPATCH_DTYPE(dtype, channel_sum_launch, x, rows, C, out, workspace, static_cast<hipStream_t>(stream));
  return ops_err();
}

extern "C" int dd_channel_sum_nhwc(const float* x, long long rows, int C, float* out, float* workspace, void* stream) {
  return dd_channel_sum_nhwc_t(x, rows, C, out, 0, workspace, stream);
}

template <typename T>
static void reflect_pad_launch(const void* x, int B, int H, int W, int C, void* out, hipStream_t s) {
  hipLaunchKernelGGL(reflect_pad1_nhwc_kernel<T>, dim3(((W + 2) * C + PAD_NT - 1) / PAD_NT, H + 2, B), dim3(PAD_NT), 0, s, static_cast<const T*>(x), H, W, C,
                     static_cast<T*>(out));
}
template <typename T>
static void reflect_pad_bwd_launch(const void* g, int B, int H, int W, int C, void* gx, hipStream_t s) {
  hipLaunchKernelGGL(reflect_pad1_nhwc_bwd_kernel<T>, dim3((W * C + PAD_NT - 1) / PAD_NT, H, B), dim3(PAD_NT), 0, s, static_cast<const T*>(g), H, W, C,
                     static_cast<T*>(gx));
}

extern "C" int dd_reflect_pad1_nhwc_t(const void* x, int B, int H, int W, int C, void* out, int dtype, void* stream) {
  if (!x || !out || B < 1 || H < 4 || W < 4 || C < 1 || B > 65535 || H + 2 > 65535 || dtype < 0 || dtype > 2) return (int)hipErrorInvalidValue;
  DD_DISPATCH_DTYPE(dtype, reflect_pad_launch, x, B, H, W, C, out, static_cast<hipStream_t>(stream));
  return ops_err();
}

extern "C" int dd_reflect_pad1_nhwc_bwd_t(const void* g_out, int B, int H, int W, int C, void* g_x, int dtype, void* stream) {
  if (!g_out || !g_x || B < 1 || H < 4 || W < 4 || C < 1 || B > 65535 || H > 65535 || dtype < 0 || dtype > 2) return (int)hipErrorInvalidValue;
  DD_DISPATCH_DTYPE(dtype, reflect_pad_bwd_launch, g_out, B, H, W, C, g_x, static_cast<hipStream_t>(stream));
  return ops_err();
}

extern "C" int dd_reflect_pad1_nhwc(const float* x, int B, int H, int W, int C, float* out, void* stream) {
  return dd_reflect_pad1_nhwc_t(x, B, H, W, C, out, 0, stream);
}

extern "C" int dd_reflect_pad1_nhwc_bwd(const float* g_out, int B, int H, int W, int C, float* g_x, void* stream) {
  return dd_reflect_pad1_nhwc_bwd_t(g_out, B, H, W, C, g_x, 0, stream);
}
