// Does finer-grained wave-level parallelism recover the issue slots the position-bias kernels leave idle?  (r03)
// PMC of cpb_bwd / deform_attn_fwd at two 256-register waves per SIMD: vector issue ~48 % of SIMD cycles, matrix pipe busy
// 29-36 %, neither ~35 %; each wave is a serial chain  [MFMAs] -> [vector work on their results] -> [MFMAs] ...  and one wave
// alone issues a vector instruction only every ~5 cycles.  This probe runs that chain shape in two granularities of the SAME
// work per query:  BIG   = 32 queries per wave: MF x v_mfma_f32_32x32x16_f16 (32 cycles) + NV vector ops on 16 accumulators
//                  SMALL = 16 queries per wave: MF x v_mfma_f32_16x16x32_f16 (16 cycles) + NV / 2 vector ops on 8 accumulators
// at 1 / 2 / 4 / 8 resident waves per SIMD (occupancy forced through the dynamic LDS size).
// Reported: ns per 32-query trip per SIMD (wall) and the shader clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int MF, int NV, bool SMALL>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* clk, int iters) {
  extern __shared__ float dummy[];
  constexpr int NACC = SMALL ? 8 : 16;
  float x[NACC];
  for (int i = 0; i < NACC; ++i) x[i] = threadIdx.x * 0.001f + i;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * ((threadIdx.x * 7 + i) % 13) - 0.05f); b[i] = (_Float16)(0.02f * ((threadIdx.x + 3 * i) % 11) - 0.1f); }
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (SMALL) {
      floatx4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
      for (int m = 0; m < MF; m += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc1, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < NV / 2; ++v) {
        const int i = v & 7;
        x[i] = __builtin_fmaf(x[i], 0.999f, (v < 8) ? (i < 4 ? acc0[i & 3] : acc1[i & 3]) : 0.001f);
        asm volatile("" : "+v"(x[i]));
      }
    } else {
      floatx16 acc = {0};
#pragma unroll
      for (int m = 0; m < MF; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16((m & 1) ? b : a, (m & 1) ? a : b, acc, 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const int i = v & 15;
        x[i] = __builtin_fmaf(x[i], 0.999f, (v < 16) ? acc[i] : 0.001f);
        asm volatile("" : "+v"(x[i]));
      }
    }
    // the vector results become the next trip's matrix operand (dependency back into the MFMAs)
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(x[i % NACC] * 0.01f);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += x[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s + dummy[0] * 0.f;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MF, int NV, bool SMALL>
static void run(const char* name, int wps) {
  float* out; unsigned long long* clk;
  // waves per SIMD through the block shape: 256 threads = one wave on each of the CU's four SIMDs; one block per CU (LDS) up to
  // 1024 threads, two blocks of 1024 for 8 waves per SIMD
  const int threads = 256, blocks_per_cu = wps;       // 256 threads = one wave on each of the CU's four SIMDs; wps blocks per CU (LDS)
  const int nblk = 256 * blocks_per_cu * 6;          // six rounds of resident blocks
  hipMalloc(&out, (size_t)nblk * threads * 4); hipMalloc(&clk, 16);
  const int iters = 2000;
  const size_t lds = (size_t)(160 * 1024) / blocks_per_cu - (blocks_per_cu == 1 ? 65536 : 1024);   // exactly blocks_per_cu blocks fit
  hipFuncSetAttribute((const void*)probe<MF, NV, SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) probe<MF, NV, SMALL><<<nblk, threads, lds>>>(out, clk, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MF, NV, SMALL><<<nblk, threads, lds>>>(out, clk, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  // query-trips: BIG wave-trip = 32 queries, SMALL = 16
  const double qtrips32 = (double)nblk * (threads / 64) * iters * (SMALL ? 0.5 : 1.0);
  const double ns_per_trip_per_simd = ms * 1e6 / (qtrips32 / 1024.0);
  printf("%-6s MF=%2d NV=%3d waves/SIMD=%d | %7.1f ns per 32-query trip per SIMD | clock %.2f GHz | wave cycles per own trip %.0f\n", name, MF, NV, wps,
         ns_per_trip_per_simd, (double)h[0] / ((double)h[1] * 10.0) , (double)h[0] / iters);
  hipFree(out); hipFree(clk);
}
int main() {
  for (int wps = 1; wps <= 8; wps *= 2) {
    run<12, 264, false>("BIG", wps);
    run<12, 264, true>("SMALL", wps);
    run<10, 170, false>("BIG", wps);
    run<10, 170, true>("SMALL", wps);
  }
  return 0;
}
