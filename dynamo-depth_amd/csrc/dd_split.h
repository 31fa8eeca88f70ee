// dd_split.h -- the exact three-way bf16 split of an fp32 operand that dd_conv_mfma.hip and dd_pw_gemm.hip feed to the bf16 matrix
// pipe (x = x1 + x2 + x3 with 8 + 8 + 8 significand bits; the arithmetic is restated in oracle/ref_split_bf16.py and pinned on the
// CPU by tests/test_split_bf16.py).  Private to csrc/.
#pragma once

#include <hip/hip_runtime.h>

namespace dd {
namespace cm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) float fl2;
typedef __attribute__((ext_vector_type(16))) float f16v;

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: round to nearest even, a in the low half
  return __builtin_bit_cast(unsigned, __builtin_convertvector(fl2{a, b}, bf2));
}
__device__ __forceinline__ float lo_f(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// two fp32 values -> their three bf16 pieces, packed pairwise
__device__ __forceinline__ void split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = pack_bf16(a, b);
  const float ra = a - lo_f(p1), rb = b - hi_f(p1);      // exact
  p2 = pack_bf16(ra, rb);
  p3 = pack_bf16(ra - lo_f(p2), rb - hi_f(p2));           // exact residual, representable in bf16
}

// the six partial products kept of (a1 + a2 + a3)(b1 + b2 + b3), the small ones first: piece of A, piece of B
constexpr int kPieceA[6] = {2, 0, 1, 1, 0, 0}, kPieceB[6] = {0, 2, 1, 0, 1, 0};

}  // namespace cm
}  // namespace dd
