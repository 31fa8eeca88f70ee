// dd_resize.hip -- the bicubic resize of frames that are not stored at the training resolution, on the device (SURVEY.md section
// 8(f) row 1, the stage between JPEG decode and ToTensor): the reference's loaders call transforms.Resize((H, W), BICUBIC) on PIL
// images (datasets/base_dataset.py:80,147), i.e. Pillow's Image.resize -- a separable two-pass convolution in 22-bit fixed point
// (src/libImaging/Resample.c, public): per output index a window of ceil(2 * max(scale, 1)) * 2 + 1 input pixels with Keys-cubic
// weights stretched by max(scale, 1) (the antialiasing), normalised and quantised on the host (hipops/resize.py builds the tables
// exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do); a horizontal pass, then a vertical pass, each rounding to
// uint8 with saturation.  Integer arithmetic throughout: the result is Pillow's, bit for bit (tests/test_resize_gpu.py against
// oracle/ref_resize.py and Pillow itself).
// HBM-bound byte work: one thread per output pixel (3 channels), taps read through L1/L2 (neighbouring outputs share most of
// them), coalesced 3-byte-per-thread stores.  ~0.3 ms for a 36-frame KITTI batch (1242x375 -> 640x192), on the prefetch stream.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/dynamo_hip.h"

namespace dd {

constexpr int RS_PRECISION = 32 - 8 - 2;
constexpr int RS_NT = 256;

__device__ __forceinline__ unsigned char rs_clip8(int acc) {
  const int v = (acc + (1 << (RS_PRECISION - 1))) >> RS_PRECISION;       // arithmetic shift: Pillow's clip8 on a signed int
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src (n, Hs, Ws, 3) -> tmp (n, Hs, W, 3)
__global__ __launch_bounds__(RS_NT) void resize_h_kernel(const unsigned char* __restrict__ src, int Hs, int Ws, int W, const int* __restrict__ bounds,
                                                         const int* __restrict__ kk, int ksize, const int* __restrict__ slot,
                                                         unsigned char* __restrict__ tmp) {
  const int x = blockIdx.x * RS_NT + threadIdx.x;
  const int y = blockIdx.y, img = blockIdx.z;
  if (x >= W) return;
  const int x0 = bounds[2 * x], cnt = bounds[2 * x + 1];
  const unsigned char* row = src + ((size_t)img * Hs + y) * Ws * 3;
  int a0 = 0, a1 = 0, a2 = 0;
  for (int k = 0; k < cnt; ++k) {
    const int c = kk[x * ksize + k];
    const unsigned char* p = row + (size_t)(x0 + k) * 3;
    a0 += c * (int)p[0]; a1 += c * (int)p[1]; a2 += c * (int)p[2];
  }
  unsigned char* o = tmp + (((size_t)(slot ? slot[img] : img) * Hs + y) * W + x) * 3;      // (slots only when this pass writes the result)
  o[0] = rs_clip8(a0); o[1] = rs_clip8(a1); o[2] = rs_clip8(a2);
}

// vertical pass: tmp (n, Hs, W, 3) -> dst (slot, H, W, 3)
__global__ __launch_bounds__(RS_NT) void resize_v_kernel(const unsigned char* __restrict__ tmp, int Hs, int W, int H, const int* __restrict__ bounds,
                                                         const int* __restrict__ kk, int ksize, const int* __restrict__ slot,
                                                         unsigned char* __restrict__ dst) {
  const int x = blockIdx.x * RS_NT + threadIdx.x;
  const int y = blockIdx.y, img = blockIdx.z;
  if (x >= W) return;
  const int y0 = bounds[2 * y], cnt = bounds[2 * y + 1];        // wave-uniform
  int a0 = 0, a1 = 0, a2 = 0;
  for (int k = 0; k < cnt; ++k) {
    const int c = kk[y * ksize + k];
    const unsigned char* p = tmp + (((size_t)img * Hs + (y0 + k)) * W + x) * 3;
    a0 += c * (int)p[0]; a1 += c * (int)p[1]; a2 += c * (int)p[2];
  }
  const int out_img = slot ? slot[img] : img;
  unsigned char* o = dst + (((size_t)out_img * H + y) * W + x) * 3;
  o[0] = rs_clip8(a0); o[1] = rs_clip8(a1); o[2] = rs_clip8(a2);
}

}  // namespace dd

extern "C" size_t dd_resize_workspace_bytes(int n_images, int Hs, int Ws, int H, int W) {
  (void)Ws; (void)H;
  return (size_t)n_images * Hs * W * 3;
}

extern "C" int dd_resize_bicubic(const unsigned char* src, int n_images, int Hs, int Ws, unsigned char* dst, const int32_t* dst_slot, int H, int W,
                                 const int32_t* h_bounds, const int32_t* h_coef, int h_ksize, const int32_t* v_bounds, const int32_t* v_coef, int v_ksize,
                                 void* workspace, size_t workspace_bytes, void* stream_) {
  using namespace dd;
  if (!src || !dst || n_images < 1 || Hs < 1 || Ws < 1 || H < 1 || W < 1) return (int)hipErrorInvalidValue;
  const bool do_h = Ws != W, do_v = Hs != H;                  // Pillow skips a pass whose size does not change
  if ((do_h && (!h_bounds || !h_coef || h_ksize < 1)) || (do_v && (!v_bounds || !v_coef || v_ksize < 1))) return (int)hipErrorInvalidValue;
  if (do_h && do_v && (!workspace || workspace_bytes < dd_resize_workspace_bytes(n_images, Hs, Ws, H, W))) return (int)hipErrorInvalidValue;
  if (n_images > 65535 || Hs > 65535 || H > 65535) return (int)hipErrorInvalidValue;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (!do_h && !do_v) {
    if (dst_slot) return (int)hipErrorInvalidValue;           // a plain copy: the caller keeps the frames where they are
    return (int)hipMemcpyAsync(dst, src, (size_t)n_images * H * W * 3, hipMemcpyDeviceToDevice, stream);
  }
  const unsigned char* vin = src;
  if (do_h) {
    // without a vertical pass the horizontal one writes the result itself (Hs == H)
    unsigned char* hout = do_v ? static_cast<unsigned char*>(workspace) : dst;
    hipLaunchKernelGGL(resize_h_kernel, dim3((W + RS_NT - 1) / RS_NT, Hs, n_images), dim3(RS_NT), 0, stream, src, Hs, Ws, W, h_bounds, h_coef, h_ksize,
                       do_v ? (const int*)nullptr : dst_slot, hout);
    vin = hout;
  }
  if (do_v)
    hipLaunchKernelGGL(resize_v_kernel, dim3((W + RS_NT - 1) / RS_NT, H, n_images), dim3(RS_NT), 0, stream, vin, Hs, W, H, v_bounds, v_coef, v_ksize, dst_slot, dst);
  return (int)hipGetLastError();
}
