// Do v_mfma_f32_32x32x16_f16 and ordinary VALU instructions overlap on gfx950, within one wave and across two waves of
// a SIMD?  Each loop trip issues 1 MFMA (dependent chain or 4 independent accumulators) and NV independent v_fma_f32.
// Reported: cycles per trip per wave (s_memtime) and per SIMD (wall clock / waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NV, int NACC, bool DO_MFMA>
__global__ __launch_bounds__(64) void probe(float* out, unsigned long long* cyc, int iters) {
  floatx16 acc[4];
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x + 2 * i)); }
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (DO_MFMA) {
        acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u % NACC], 0, 0, 0);
        asm volatile("" : "+a"(acc[u % NACC]));
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        x[v & 7] = x[v & 7] * 1.0001f + 0.001f;
        asm volatile("" : "+v"(x[v & 7]));
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 16; ++k) s += acc[i][k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NV, int NACC, bool DO_MFMA>
static void run(const char* name, int waves_per_simd) {
  float* out; unsigned long long* cyc;
  const int nblk = 1024 * waves_per_simd;
  hipMalloc(&out, (size_t)nblk * 64 * 4); hipMalloc(&cyc, 8);
  const int iters = 1000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NV, NACC, DO_MFMA><<<nblk, 64>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NV, NACC, DO_MFMA><<<nblk, 64>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double trips = (double)iters * 4;
  printf("%-30s NV=%2d acc=%d waves/SIMD=%d | per wave %6.1f cyc/trip | per SIMD %6.1f ns/trip (x2.2 = %6.1f cyc)\n", name, NV, NACC,
         waves_per_simd, (double)h / trips, ms * 1e6 / trips / waves_per_simd, ms * 1e6 / trips / waves_per_simd * 2.2);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0, 1, true>("mfma only, dependent", w);
    run<0, 4, true>("mfma only, 4 accumulators", w);
    run<8, 1, false>("valu only", w);
    run<16, 1, false>("valu only", w);
    run<4, 4, true>("mfma + valu", w);
    run<8, 4, true>("mfma + valu", w);
    run<8, 1, true>("mfma(dep) + valu", w);
    run<16, 4, true>("mfma + valu", w);
    run<16, 1, true>("mfma(dep) + valu", w);
    run<32, 4, true>("mfma + valu", w);
    run<32, 1, false>("valu only", w);
  }
  return 0;
}
