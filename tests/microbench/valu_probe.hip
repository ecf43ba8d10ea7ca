// VALU issue / dependent-latency probe for gfx950: cycles per v_fma_f32 (and v_pk_fma_f32) in one wave per SIMD,
// as 1, 2, 4 or 8 independent dependency chains.  Build: hipcc --offload-arch=gfx950 -O3 valu_probe.hip -o bin/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int CH, bool PK>
__global__ __launch_bounds__(64) void probe(float* out, unsigned long long* cyc, int iters) {
  float2v x[8];
  for (int i = 0; i < 8; ++i) x[i] = (float2v){(float)threadIdx.x * 0.001f + i, 1.0f + i};
  const float2v a = {1.0001f, 0.9999f}, b = {0.001f, -0.001f};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64 / CH; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (PK) x[c] = x[c] * a + b;
        else x[c][0] = x[c][0] * a[0] + b[0];
        asm volatile("" : "+v"(x[c]));
      }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CH, bool PK>
static void run(const char* name) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<CH, PK><<<1024, 64>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<CH, PK><<<1024, 64>>>(out, cyc, iters);     // 1024 waves = one per SIMD
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double n = (double)iters * 64;
  printf("%-34s chains=%d  %.2f ns / instr   (s_memtime ticks / instr %.3f)\n", name, CH, ms * 1e6 / n, (double)h / n);
}
int main() {
  run<1, false>("v_fma_f32 dependent"); run<2, false>("v_fma_f32"); run<4, false>("v_fma_f32"); run<8, false>("v_fma_f32");
  run<1, true>("v_pk_fma_f32 dependent"); run<2, true>("v_pk_fma_f32"); run<4, true>("v_pk_fma_f32"); run<8, true>("v_pk_fma_f32");
  return 0;
}
