// Probe: issue behaviour of v_mfma_f32_32x32x2_f32 on gfx950 - dependent chains, waves per SIMD, VALU filler.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

template <int NACC, int FILL, bool BFROMVALU>
__global__ void probe(float* out, int iters, float x, float y) {
  floatx16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = floatx16{0};
  float a = x + threadIdx.x * 1e-3f, b = y, f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = x * i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float bb = b;
      if (BFROMVALU) bb = fmaxf(fmaf(a, f[s & 7], fmaf(b, f[(s + 1) & 7], y)), 0.f);
#pragma unroll
      for (int k = 0; k < FILL; ++k) f[k & 7] = fmaf(f[k & 7], 1.0001f, f[(k + 1) & 7]);
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = MF(a, bb, acc[i]);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int FILL, bool BV>
void run(const char* name, int threads, float* out) {
  const int iters = 2000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NACC, FILL, BV>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.f, 2.f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<NACC, FILL, BV>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf_per_simd = (double)iters * 16 * NACC * (threads / 256.0);
  const double tf = (double)blocks * (threads / 64) * iters * 16.0 * NACC * 4096 / (ms * 1e-3) / 1e12;
  printf("%-44s waves/SIMD %d  %8.3f ms  %6.1f TF  %6.1f ns per MFMA per SIMD (=%5.1f cyc @2.4GHz)\n", name, threads / 256, ms, tf,
         ms * 1e6 / mf_per_simd, ms * 1e6 / mf_per_simd * 2.4);
}

int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  run<1, 0, false>("1 dependent chain, no filler", 256, out);
  run<1, 0, false>("1 dependent chain, no filler", 512, out);
  run<4, 0, false>("4 independent accumulators", 256, out);
  run<1, 0, true>("1 chain, B operand from 3 VALU", 256, out);
  run<1, 0, true>("1 chain, B operand from 3 VALU", 512, out);
  run<1, 6, true>("1 chain, B from VALU + 6 filler fma", 256, out);
  run<1, 6, true>("1 chain, B from VALU + 6 filler fma", 512, out);
  run<1, 12, true>("1 chain, B from VALU + 12 filler fma", 256, out);
  run<1, 12, true>("1 chain, B from VALU + 12 filler fma", 512, out);
  run<2, 12, true>("2 chains, B from VALU + 12 filler fma", 256, out);
  return 0;
}
