// Does ds_read_b128 work at 2-byte-aligned LDS addresses on gfx950 (unaligned DS access mode), and what does it cost?
// build: hipcc --offload-arch=gfx950 -O3 lds_unaligned.hip -o lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out, int shift_bytes, long long* cycles, int reps) {
  __shared__ __align__(16) unsigned short lds[64 * 64 + 64];
  for (int i = threadIdx.x; i < 64 * 64 + 64; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  unsigned addr = (unsigned)(size_t)lds + threadIdx.x * 48 + shift_bytes;     // 48-byte lane stride as in the convolution kernels
  uint4 v = make_uint4(0, 0, 0, 0), acc = make_uint4(0, 0, 0, 0);
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  long long t1 = clock64();
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
  if (threadIdx.x == 0) { cycles[0] = t1 - t0; out[256] = acc.x; }
}
int main() {
  unsigned* d; long long* c;
  hipMalloc(&d, 1028 * 4); hipMalloc(&c, 8);
  for (int shift = 0; shift <= 14; shift += 2) {
    probe<<<1, 64>>>(d, shift, c, 1000);
    std::vector<unsigned> h(1028); long long cyc;
    hipMemcpy(h.data(), d, 1028 * 4, hipMemcpyDeviceToHost); hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
    hipError_t e = hipDeviceSynchronize();
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int k = 0; k < 8; ++k) {
        unsigned expect = (unsigned)((l * 48 + shift) / 2 + k);
        unsigned got = (h[l * 4 + k / 2] >> (16 * (k & 1))) & 0xffff;
        if (got != expect) ++bad;
      }
    printf("shift %2d bytes: %s  mismatches %d  cycles per dependent read %.1f\n", shift, hipGetErrorString(e), bad, cyc / 1000.0);
  }
  return 0;
}
