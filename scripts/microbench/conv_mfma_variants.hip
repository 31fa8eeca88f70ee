// Where does dd_conv3x3_mfma's time go?  The forward kernel with parts left out (DD_CM_EXP bits; results are wrong, times are not):
//   1 no per-step barrier   2 no fragment reads in the tap loop   4 no weight fetch / store   8 no MFMAs   16 no halo prefetch
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DDD_CM_EXP=<bits> conv_mfma_variants.hip -o conv_mfma_exp<bits>.bin
#include "../../dynamo-depth_amd/csrc/dd_conv_mfma.hip"
#include <cstdio>
#include <vector>
int main() {
  const int B = 12, C = 64, H = 96, W = 320;
  float *x, *y, *w, *bias; void* pack;
  const size_t n = (size_t)B * H * W * C;
  hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&w, C * C * 9 * 4); hipMalloc(&bias, C * 4);
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), C * C * 9 * 4, hipMemcpyHostToDevice);
  hipMemcpy(bias, h.data(), C * 4, hipMemcpyHostToDevice);
  const size_t pb = dd_conv3x3_mfma_pack_bytes(C, C);
  hipMalloc(&pack, pb);
  dd_conv3x3_mfma_pack(w, C * 9, 9, 3, 1, C, C, pack, nullptr, nullptr);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) dd_conv3x3_mfma(x, pack, bias, B, H, W, C, C, 1, y, nullptr);
  hipEventRecord(e0, nullptr);
  const int reps = 30;
  for (int i = 0; i < reps; ++i) dd_conv3x3_mfma(x, pack, bias, B, H, W, C, C, 1, y, nullptr);
  hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("DD_CM_EXP=%2d  forward 12x64x64x96x320: %.1f us  (%s)\n", DD_CM_EXP, ms / reps * 1e3, hipGetErrorString(hipGetLastError()));
  return 0;
}
