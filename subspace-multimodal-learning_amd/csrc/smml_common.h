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

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }
__device__ __forceinline__ float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// wave-local ordering of LDS traffic: LDS ops of one wave complete in issue order; this only
// stops the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// sign(d) * log(|d| + 1)  (continuous position bias input transform)
__device__ __forceinline__ float signed_log1p(float d) { return copysignf(logf(fabsf(d) + 1.0f), d); }
