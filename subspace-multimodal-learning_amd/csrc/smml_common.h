// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of the multimodal-MIL attention path.
// Error convention of the C-ABI (include/smml.h): 0 = ok, negative = error, text via smml_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define SMML_OK 0
#define SMML_ERR_ARG -1
#define SMML_ERR_HIP -2

void smml_set_error(const char* fmt, ...);

#define SMML_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      smml_set_error(__VA_ARGS__);         \
      return SMML_ERR_ARG;                 \
    }                                      \
  } while (0)

#define SMML_LAUNCH_CHECK(name)                                                     \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      smml_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));         \
      return SMML_ERR_HIP;                                                          \
    }                                                                               \
  } while (0)

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: exact-fp32 matrix core op (k-ordered fmaf chain), 64 FLOP/clk/SIMD.
//   A operand: lane l holds A[i = l & 31][k = l >> 5];  B operand: lane l holds B[k = l >> 5][j = l & 31]
//   C/D:       lane l, register r holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
__device__ __forceinline__ floatx16 mfma32(float a, float b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// row of the 32x32 accumulator tile held in register r of a lane in half hf
__device__ __forceinline__ constexpr int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

#ifndef SMML_DPP_REDUCE
#define SMML_DPP_REDUCE 1   // cross-lane sums on the VALU (v_permlane32_swap / DPP) instead of ds_bpermute through LDS
#endif
// value of the other 32-lane half: v_permlane32_swap exchanges the upper half of its first operand with the
// lower half of its second, so with both operands = v the two results hold {lo, lo} and {hi, hi}
__device__ __forceinline__ void xhalf_pair(float v, float& lo, float& hi) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  lo = __uint_as_float(r[0]);
  hi = __uint_as_float(r[1]);
}
#if SMML_DPP_REDUCE
__device__ __forceinline__ float xhalf_sum(float v) { float a, b; xhalf_pair(v, a, b); return a + b; }
__device__ __forceinline__ float xhalf_max(float v) { float a, b; xhalf_pair(v, a, b); return fmaxf(a, b); }
// sum over the 64 lanes, valid in EVERY lane: inclusive row scan with DPP row_shr 1/2/4/8 (lane 15 of each
// 16-lane row holds the row sum), readlane of the four row sums
__device__ __forceinline__ float wave_sum(float v) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x111, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x112, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x114, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x118, 0xf, 0xf, false));
  const unsigned u = __float_as_uint(v);
  return __uint_as_float(__builtin_amdgcn_readlane(u, 15)) + __uint_as_float(__builtin_amdgcn_readlane(u, 31)) +
         __uint_as_float(__builtin_amdgcn_readlane(u, 47)) + __uint_as_float(__builtin_amdgcn_readlane(u, 63));
}
#else
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }
__device__ __forceinline__ float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#endif
// wave-local ordering of LDS traffic: LDS ops of one wave complete in issue order; this only
// stops the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// sign(d) * log(|d| + 1)  (continuous position bias input transform)
__device__ __forceinline__ float signed_log1p(float d) { return copysignf(logf(fabsf(d) + 1.0f), d); }
