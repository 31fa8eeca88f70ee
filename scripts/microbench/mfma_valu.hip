// Can the matrix pipe take the 3x3 window sums of the photometric kernel off the VALU?  Three facts decide it:
// the rate of the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2), the rate of the packed fp32 VALU instruction the kernel uses
// today (v_pk_fma_f32), and whether a SIMD overlaps the two when half of its waves issue one kind and half the other.
//   vv : 4 waves per SIMD, all VALU        mm : 4 waves per SIMD, all MFMA        vm : 2 VALU waves + 2 MFMA waves per SIMD,
// every wave issuing the SAME number of wave-instructions as in its pure run -- if the pipes overlap, vm takes max(vv, mm) / 2
// of... (each kind has half the waves), i.e. vm ~ max(vv, mm)/2; if they serialise, vm ~ (vv + mm)/2.
// Built and run by scripts/microbench/run_mfma_valu.sh on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// kind per wave: 0 = VALU (v_pk_fma_f32, 8 chains), 1 = MFMA 16x16x4 f32 (4 chains), 2 = MFMA 32x32x2 f32 (2 chains)
__global__ __launch_bounds__(512) void burn(float* out, int iters, int kind_even, int kind_odd) {
  const int wave = threadIdx.x >> 6;
  const int kind = (wave & 4) ? kind_odd : kind_even;       // 8 waves per workgroup: waves w and w + 4 share SIMD w -- one of each kind per SIMD and workgroup
  float s = 0.f;
  if (kind == 0) {
    f2 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f2{(float)threadIdx.x + k, (float)k};
    const f2 va = {1.0001f, 1.0002f}, vb = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[k]) : "v"(va), "v"(vb));
    for (int k = 0; k < 8; ++k) s += acc[k].x + acc[k].y;
  } else if (kind == 1) {
    f4 acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = f4{(float)threadIdx.x, 1.f, 2.f, 3.f};
    const float a = 1.0001f, b = 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int rep = 0; rep < 32; ++rep)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
    for (int k = 0; k < 4; ++k) s += acc[k].x + acc[k].y + acc[k].z + acc[k].w;
  } else {
    f16v acc[2];
    for (int k = 0; k < 2; ++k)
      for (int j = 0; j < 16; ++j) acc[k][j] = (float)(threadIdx.x + j);
    const float a = 1.0001f, b = 0.5f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int rep = 0; rep < 64; ++rep)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
    for (int k = 0; k < 2; ++k)
      for (int j = 0; j < 16; ++j) s += acc[k][j];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 4 * 4 * 256 * sizeof(float));
  const int blocks = 256 * 2;          // 2 workgroups of 8 waves per CU: 4 waves per SIMD
  const int iters = 1000;              // 128 wave-instructions per iteration in every kind
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"v_pk_fma_f32", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x2_f32"};
  const double macs[3] = {128.0, 1024.0, 2048.0};          // multiply-adds per wave-instruction
  double t_pure[3] = {0, 0, 0};
  for (int pass = 0; pass < 5; ++pass) {
    const int ke = pass < 3 ? pass : 0, ko = pass < 3 ? pass : pass - 2;      // passes 3, 4: VALU on waves 0-3, MFMA kind 1 / 2 on waves 4-7
    float ms = 0.f;
    for (int w = 0; w < 2; ++w) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(burn, dim3(blocks), dim3(512), 0, 0, out, iters, ke, ko);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double winstr_per_simd = 4.0 * iters * 128;                  // 4 waves per SIMD
    if (pass < 3) {
      t_pure[pass] = ms;
      printf("%-24s alone : %.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz  %.0f MAC/cycle/SIMD  %.1f TFLOP/s\n", names[pass], ms,
             ms * 1e-3 * 2.4e9 / winstr_per_simd, macs[pass] * winstr_per_simd / (ms * 1e-3 * 2.4e9),
             2.0 * macs[pass] * winstr_per_simd * 1024 / (ms * 1e-3) / 1e12);
    } else {
      printf("half the waves %s + half %s: %.3f ms   (perfect overlap: %.3f ms, no overlap: %.3f ms)\n", names[0], names[ko], ms,
             0.5 * (t_pure[0] > t_pure[ko] ? t_pure[0] : t_pure[ko]), 0.5 * (t_pure[0] + t_pure[ko]));
    }
  }
  return 0;
}
