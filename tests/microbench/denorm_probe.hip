// Does v_mfma_f32_32x32x16_f16 / _bf16 keep fp16 / bf16 subnormal inputs?  (prints the products of 1.0 x 2^-20 etc.)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float tiny) {
  half8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
  const int lane = threadIdx.x;
  if (lane < 32) { a[0] = (_Float16)1.0f; b[0] = (_Float16)tiny; }   // k = 0 only: D[i][j] = 1 * tiny
  floatx16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  if (lane == 0) { out[0] = acc[0]; out[1] = (float)(_Float16)tiny; }
  bf16x8 c = {0, 0, 0, 0, 0, 0, 0, 0}, d = c;
  if (lane < 32) { c[0] = (__bf16)1.0f; d[0] = (__bf16)1e-39f; }
  floatx16 acc2 = {0};
  acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, d, acc2, 0, 0, 0);
  if (lane == 0) { out[2] = acc2[0]; out[3] = (float)(__bf16)1e-39f; }
}
int main() {
  float* d; hipMalloc(&d, 16);
  k<<<1, 64>>>(d, 9.5367431640625e-07f);   // 2^-20: an fp16 subnormal
  float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("f16 : 1 x 2^-20 (subnormal) -> %.9e   (operand as fp16: %.9e)  %s\n", h[0], h[1], h[0] == h[1] && h[0] != 0 ? "kept" : "FLUSHED");
  printf("bf16: 1 x 1e-39 (subnormal) -> %.9e   (operand as bf16: %.9e)  %s\n", h[2], h[3], h[2] == h[3] && h[2] != 0 ? "kept" : "FLUSHED");
  return 0;
}
