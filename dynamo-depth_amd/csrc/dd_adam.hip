// dd_adam.hip -- the Adam update of ALL parameters of a training step in one launch (reference
// Trainer.py:150 `self.optim['optimizer'].step()` on torch.optim.Adam(params, lr), Trainer.py:492-497 -- betas (0.9, 0.999),
// eps 1e-8, no weight decay, no amsgrad).  torch's multi-tensor Adam covers the ~400 parameter tensors of the four networks (46 M
// floats) with a dozen launches that move 1.3 GB at ~1.4 TB/s; the update is pure streaming work -- read p, g, m, v, write p, m, v:
// 28 B per element -- and it runs ALONE at the end of the step (every backward graph has been joined), so its time is step time.
// Here: one launch (behind a one-thread-per-tensor prologue that advances the step counters and evaluates the bias corrections),
// a table of tensor records in device memory (the addresses are fixed: the step is replayed from graphs), a
// block -> (record, chunk) map, 16-byte accesses wherever the four arrays of a record are 16-byte aligned.
// The update rule, in torch's operation order (torch/optim/adam.py _single_tensor_adam / the fused kernel's adam_math):
//   g' = g / grad_scale (if given);  m = m + (1-b1)(g' - m);  v = b2 v + (1-b2) g' g';
//   p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps),   t = the record's step counter after its increment
// bias corrections in double from the float step counter, as torch computes them.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "../../include/dynamo_hip.h"

namespace dd {

constexpr int AD_NT = 256;
constexpr int AD_CHUNK = 4096;        // elements per workgroup: 4 x float4 per thread

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float b1, float b2, float one_m_b1, float one_m_b2, float step_size,
                                         float bc2_sqrt, float eps, float wd) {
  if (wd != 0.f) g = fmaf(wd, p, g);
  m = m + one_m_b1 * (g - m);
  v = b2 * v + one_m_b2 * g * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * m / denom;
}

// one thread per record: the step counter's increment (torch: _foreach_add_(steps, 1) in front of the update; a skipped step --
// *found_inf != 0 -- does not count) and the two bias corrections, in double from the float counter as torch computes them
__global__ __launch_bounds__(AD_NT) void adam_prep_kernel(const DDAdamRecord* __restrict__ recs, int n_records, double lr, double beta1, double beta2,
                                                          const float* __restrict__ found_inf, float2* __restrict__ aux) {
  const int i = blockIdx.x * AD_NT + threadIdx.x;
  if (i >= n_records) return;
  if (found_inf && *found_inf != 0.f) return;
  const float t_f = *recs[i].step + 1.f;
  *recs[i].step = t_f;
  const double t = (double)t_f;
  const float bc1 = (float)(1.0 - pow(beta1, t));
  aux[i] = make_float2((float)(lr / (double)bc1), (float)sqrt(1.0 - pow(beta2, t)));        // (step size, sqrt of bias correction 2)
}

__global__ __launch_bounds__(AD_NT) void adam_multi_kernel(const DDAdamRecord* __restrict__ recs, const int2* __restrict__ block_map,
                                                           const float2* __restrict__ aux, double beta1, double beta2, double eps_d, double wd_d,
                                                           const float* __restrict__ grad_scale, const float* __restrict__ found_inf) {
  if (found_inf && *found_inf != 0.f) return;           // GradScaler's skip: an overflowed step leaves parameters and moments alone
  const int2 bm = block_map[blockIdx.x];
  const DDAdamRecord r = recs[bm.x];
  const long long base = (long long)bm.y * AD_CHUNK;
  const long long n = r.n - base < AD_CHUNK ? r.n - base : AD_CHUNK;
  const float2 ax = aux[bm.x];
  const float step_size = ax.x, bc2_sqrt = ax.y;
  const float b1 = (float)beta1, b2 = (float)beta2, eps = (float)eps_d, wd = (float)wd_d;
  const float one_m_b1 = (float)(1.0 - beta1), one_m_b2 = (float)(1.0 - beta2);
  const float inv_scale = grad_scale ? 1.f / *grad_scale : 1.f;
  float* __restrict__ p = r.param + base;
  const float* __restrict__ g = r.grad + base;
  float* __restrict__ m = r.exp_avg + base;
  float* __restrict__ v = r.exp_avg_sq + base;
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
  if (vec) {
    const int n4 = (int)(n >> 2);
    float4 P[4], G[4], M[4], V[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = threadIdx.x + k * AD_NT;
      if (i < n4) {
        P[k] = reinterpret_cast<const float4*>(p)[i]; G[k] = reinterpret_cast<const float4*>(g)[i];
        M[k] = reinterpret_cast<const float4*>(m)[i]; V[k] = reinterpret_cast<const float4*>(v)[i];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = threadIdx.x + k * AD_NT;
      if (i < n4) {
        if (grad_scale) { G[k].x *= inv_scale; G[k].y *= inv_scale; G[k].z *= inv_scale; G[k].w *= inv_scale; }
        adam_one(P[k].x, G[k].x, M[k].x, V[k].x, b1, b2, one_m_b1, one_m_b2, step_size, bc2_sqrt, eps, wd);
        adam_one(P[k].y, G[k].y, M[k].y, V[k].y, b1, b2, one_m_b1, one_m_b2, step_size, bc2_sqrt, eps, wd);
        adam_one(P[k].z, G[k].z, M[k].z, V[k].z, b1, b2, one_m_b1, one_m_b2, step_size, bc2_sqrt, eps, wd);
        adam_one(P[k].w, G[k].w, M[k].w, V[k].w, b1, b2, one_m_b1, one_m_b2, step_size, bc2_sqrt, eps, wd);
        reinterpret_cast<float4*>(p)[i] = P[k]; reinterpret_cast<float4*>(m)[i] = M[k]; reinterpret_cast<float4*>(v)[i] = V[k];
      }
    }
    const int done = n4 << 2;                             // the 0..3 elements behind the last whole float4 of the tensor
    const int i = done + threadIdx.x;
    if (i < n) {
      float pp = p[i], gg = g[i] * inv_scale, mm = m[i], vv = v[i];
      adam_one(pp, gg, mm, vv, b1, b2, one_m_b1, one_m_b2, step_size, bc2_sqrt, eps, wd);
      p[i] = pp; m[i] = mm; v[i] = vv;
    }
  } else {
    for (int i = threadIdx.x; i < n; i += AD_NT) {
      float pp = p[i], gg = g[i] * inv_scale, mm = m[i], vv = v[i];
      adam_one(pp, gg, mm, vv, b1, b2, one_m_b1, one_m_b2, step_size, bc2_sqrt, eps, wd);
      p[i] = pp; m[i] = mm; v[i] = vv;
    }
  }
}

}  // namespace dd

extern "C" int dd_adam_chunk(void) { return dd::AD_CHUNK; }

extern "C" int dd_adam_multi(const DDAdamRecord* records, int n_records, const int* block_map, int n_blocks, void* aux, double lr, double beta1,
                             double beta2, double eps, double weight_decay, const float* grad_scale, const float* found_inf, void* stream) {
  if (n_blocks == 0 || n_records == 0) return 0;
  if (!records || !block_map || !aux || n_blocks < 0 || n_records < 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(dd::adam_prep_kernel, dim3((n_records + dd::AD_NT - 1) / dd::AD_NT), dim3(dd::AD_NT), 0, (hipStream_t)stream, records, n_records, lr,
                     beta1, beta2, found_inf, reinterpret_cast<float2*>(aux));
  hipLaunchKernelGGL(dd::adam_multi_kernel, dim3(n_blocks), dim3(dd::AD_NT), 0, (hipStream_t)stream, records, reinterpret_cast<const int2*>(block_map),
                     reinterpret_cast<const float2*>(aux), beta1, beta2, eps, weight_decay, grad_scale, found_inf);
  return (int)hipGetLastError();
}
