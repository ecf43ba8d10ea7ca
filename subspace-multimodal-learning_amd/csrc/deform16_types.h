// Element-type plumbing of the 16-bit compute modes of the fused deformable attention (T = _Float16 or __bf16): conversions, the MFMA
// overloads, transposed LDS fragments, the fp16 score format with the dropout decision in its lowest bit.  Included inside the
// anonymous namespace of deform_attn16.hip and of cpb_regions.h (whose region kernels exist in an fp32-grade and a 16-bit form).
#pragma once
#include "deform_common.h"

namespace {

typedef unsigned short u16;

// ---- element-type plumbing: T = _Float16 or __bf16 ----
template <typename T> struct Vec8;
template <> struct Vec8<_Float16> { typedef half8 type; };
template <> struct Vec8<__bf16> { typedef bf16x8 type; };
__device__ __forceinline__ floatx16 mma(half8 a, half8 b, floatx16 c) { return mfma16(a, b, c); }
__device__ __forceinline__ floatx16 mma(bf16x8 a, bf16x8 b, floatx16 c) { return mfma16b(a, b, c); }
// two fp32 -> one 32-bit word of two T (round to nearest even; element 0 in the low half)
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b);
template <> __device__ __forceinline__ unsigned pack2<_Float16>(float a, float b) {
  const float2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, half2v));
}
template <> __device__ __forceinline__ unsigned pack2<__bf16>(float a, float b) {
  const float2v v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
template <typename T> __device__ __forceinline__ typename Vec8<T>::type cvt8(const float (&x)[8]) {
  const uint4v w = {pack2<T>(x[0], x[1]), pack2<T>(x[2], x[3]), pack2<T>(x[4], x[5]), pack2<T>(x[6], x[7])};
  return __builtin_bit_cast(typename Vec8<T>::type, w);
}
template <typename T> __device__ __forceinline__ uint2v pack4(const float4 v) {
  return (uint2v){pack2<T>(v.x, v.y), pack2<T>(v.z, v.w)};
}
// 16-bit pattern -> fp32
template <typename T> __device__ __forceinline__ float tof(unsigned u16bits);
template <> __device__ __forceinline__ float tof<__bf16>(unsigned u) { return __builtin_bit_cast(float, u << 16); }
template <> __device__ __forceinline__ float tof<_Float16>(unsigned u) { return (float)__builtin_bit_cast(_Float16, (u16)u); }
__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xFFFF0000u); }
// MFMA fragment of an operand stored k-major in LDS (two hardware-transposed reads, smml_common.h lds_frag_tr), any 16-bit type
template <typename T> __device__ __forceinline__ typename Vec8<T>::type frag_tr(const T* p0, const T* p1) {
  typedef short short4v __attribute__((ext_vector_type(4)));
  typedef short short8v __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) short4v lds_s4;
  const short4v r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)p0);
  const short4v r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)p1);
  const short8v r = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
  return __builtin_bit_cast(typename Vec8<T>::type, r);
}
// Stored scores are fp16 in BOTH modes: they are of forward range (softmax logits; clamped to +-60000 so that nothing rounds to inf)
// and fp16's 11 significant bits keep exp(score - lse) to 2^-11 |score| - eight times finer than bf16 at the same two bytes.
__device__ __forceinline__ unsigned pack_score(float a, float b) {
  return pack2<_Float16>(fminf(fmaxf(a, -60000.f), 60000.f), fminf(fmaxf(b, -60000.f), 60000.f));
}
__device__ __forceinline__ float score_of(unsigned u) { return tof<_Float16>(u); }
// The dropout keep decision REPLACES the lowest mantissa bit of a stored fp16 score (one v_and_or per pair; the fp32 path nudges the
// value by a zero-mean ulp instead - at fp16's 2^-11 the half-ulp this costs is below the rounding the score already carries)
__device__ __forceinline__ unsigned stash_keep16(unsigned u, bool keep) { return (u & 0xFFFEu) | (keep ? 1u : 0u); }
__device__ __forceinline__ unsigned stash_keep16x2(unsigned w, unsigned two_bits) {      // bit 0 -> low half, bit 1 -> high half
  return (w & 0xFFFEFFFEu) | (two_bits & 1u) | ((two_bits & 2u) << 15);
}


}  // namespace
