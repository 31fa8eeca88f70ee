// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950: the same number of wave-instructions, 8 independent accumulator
// chains per lane, 4 waves per SIMD.  Built and run by scripts/microbench/run_pk_rate.sh on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool PK>
__global__ __launch_bounds__(256) void burn(float* out, float a, float b, int iters) {
  f2 acc[8];
  for (int k = 0; k < 8; ++k) acc[k] = f2{(float)threadIdx.x + k, (float)k};
  const f2 va = {a, a * 1.0001f}, vb = {b, b * 0.9999f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 16; ++rep)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (PK) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(va), "v"(vb));
        } else {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[k].x) : "v"(va.x), "v"(vb.x));
        }
      }
  }
  float s = 0.f;
  for (int k = 0; k < 8; ++k) s += acc[k].x + acc[k].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 256 * 4 * sizeof(float));
  const int blocks = 256 * 4 * 4;      // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pk = 0; pk < 2; ++pk) {
    for (int w = 0; w < 2; ++w) {
      hipEventRecord(e0);
      if (pk) hipLaunchKernelGGL(burn<true>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
      else hipLaunchKernelGGL(burn<false>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double winstr = (double)blocks * 4 * iters * 128;      // wave-instructions
      const double per_simd = winstr / (256.0 * 4);
      if (w) printf("%s: %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz, %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ",
                    ms, ms * 1e-3 * 2.4e9 / per_simd, winstr * 64 * (pk ? 4 : 2) / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
