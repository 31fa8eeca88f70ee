// dd_half.h -- element-type adapters for the network-side kernels: the tensors of an autocast (fp16 / bf16) forward are read and
// written in their own type, four elements per access, while every statistic, sum and intermediate stays fp32.
// DD_DTYPE_* are the `dtype` codes of the *_t entry points in include/dynamo_hip.h.
#pragma once

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

namespace dd {

template <typename T>
struct IO;

template <>
struct IO<float> {
  static __device__ __forceinline__ float4 load4(const float* p, long long i4) { return reinterpret_cast<const float4*>(p)[i4]; }
  static __device__ __forceinline__ void store4(float* p, long long i4, float4 v) { reinterpret_cast<float4*>(p)[i4] = v; }
  static __device__ __forceinline__ float load1(const float* p, long long i) { return p[i]; }
  static __device__ __forceinline__ void store1(float* p, long long i, float v) { p[i] = v; }
};

template <>
struct IO<__half> {
  static __device__ __forceinline__ float4 load4(const __half* p, long long i4) {
    const uint2 raw = reinterpret_cast<const uint2*>(p)[i4];
    const __half2 a = *reinterpret_cast<const __half2*>(&raw.x), b = *reinterpret_cast<const __half2*>(&raw.y);
    const float2 fa = __half22float2(a), fb = __half22float2(b);
    return make_float4(fa.x, fa.y, fb.x, fb.y);
  }
  static __device__ __forceinline__ void store4(__half* p, long long i4, float4 v) {
    const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
    uint2 raw;
    raw.x = *reinterpret_cast<const unsigned*>(&a);
    raw.y = *reinterpret_cast<const unsigned*>(&b);
    reinterpret_cast<uint2*>(p)[i4] = raw;
  }
  static __device__ __forceinline__ float load1(const __half* p, long long i) { return __half2float(p[i]); }
  static __device__ __forceinline__ void store1(__half* p, long long i, float v) { p[i] = __float2half(v); }
};

template <>
struct IO<__hip_bfloat16> {
  static __device__ __forceinline__ float4 load4(const __hip_bfloat16* p, long long i4) {
    const uint2 raw = reinterpret_cast<const uint2*>(p)[i4];
    return make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                       __uint_as_float(raw.y & 0xffff0000u));
  }
  static __device__ __forceinline__ unsigned rne(float f) {           // fp32 -> bf16 bits, round to nearest even (NaN stays NaN)
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  }
  static __device__ __forceinline__ void store4(__hip_bfloat16* p, long long i4, float4 v) {
    uint2 raw;
    raw.x = rne(v.x) | (rne(v.y) << 16);
    raw.y = rne(v.z) | (rne(v.w) << 16);
    reinterpret_cast<uint2*>(p)[i4] = raw;
  }
  static __device__ __forceinline__ float load1(const __hip_bfloat16* p, long long i) {
    return __uint_as_float(static_cast<unsigned>(reinterpret_cast<const unsigned short*>(p)[i]) << 16);
  }
  static __device__ __forceinline__ void store1(__hip_bfloat16* p, long long i, float v) {
    reinterpret_cast<unsigned short*>(p)[i] = static_cast<unsigned short>(rne(v));
  }
};

}  // namespace dd

// calls FN<T>(args...) for the element type selected by a DD_DTYPE_* code (0 fp32, 1 fp16, 2 bf16)
#define DD_DISPATCH_DTYPE(dtype, FN, ...)                              \
  do {                                                                 \
    if ((dtype) == 0) FN<float>(__VA_ARGS__);                          \
    else if ((dtype) == 1) FN<__half>(__VA_ARGS__);                    \
    else FN<__hip_bfloat16>(__VA_ARGS__);                              \
  } while (0)
