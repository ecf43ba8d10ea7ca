// Semantics check of ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4bf16) for an MFMA operand stored k-major:
// LDS image [16 k][32 m] bf16, value = 100 k + m.  Lane (i = l & 31, h = l >> 5) wants A[i][k = 8 h + j], j = 0..7.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) __bf16 img[16][32 + 32];   // row stride 128 B
  const int l = threadIdx.x;
  for (int e = l; e < 16 * 64; e += 64) img[e / 64][e % 64] = (__bf16)(float)((e / 64) * 100 + (e % 64));
  __syncthreads();
  const int h = l >> 5, mblk = (l >> 4) & 1, w = l & 15, q = w >> 2, p = w & 3;
  for (int t = 0; t < 2; ++t) {
    const __bf16* src = &img[8 * h + 4 * t + q][16 * mblk + 4 * p];
    bf16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)src);
    for (int j = 0; j < 4; ++j) out[l * 8 + 4 * t + j] = (float)r[j];
  }
}
int main() {
  float* d; hipMalloc(&d, 64 * 8 * 4);
  k<<<1, 64>>>(d);
  float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 8; ++j) {
      const float want = (8 * (l >> 5) + j) * 100 + (l & 31);
      if (h[l * 8 + j] != want) ++bad;
    }
  for (int l : {0, 1, 5, 17, 33, 63}) { printf("lane %2d:", l); for (int j = 0; j < 8; ++j) printf(" %5.0f", h[l * 8 + j]); printf("\n"); }
  printf("%s (%d mismatches): lane (i, h) receives A[i][8 h + j] from a [k][m] image\n", bad ? "MISMATCH" : "OK", bad);
  return 0;
}
