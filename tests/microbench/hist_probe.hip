// LDS histogram / table-gather probe for gfx950 (round 5: sizing of the per-linear-region evaluation of the position-bias MLP).
//   hist: every lane adds 3 floats into an LDS table of R regions x 3 moments per iteration; region ids come in runs of L lanes
//         (consecutive queries of one key fall into the same linear region in runs of ~4); variants: float atomics, 64-bit integer
//         atomics, run heads only (wave-level segmented sums first), no atomics (loop overhead).
//   gather: every lane loads one 4-byte (or 8-byte) entry of a G x G table per iteration, lanes = 32 consecutive queries x 2 keys:
//         a strip of cells `stride` apart in one table row per key.
// Build: hipcc --offload-arch=gfx950 -O3 hist_probe.hip -o bin/hist_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// MODE 0: atomicAdd(float*) on LDS as hipcc compiles it; 1: ds_add_f32 by builtin; 2: 64-bit integer adds (fixed point);
// 3: segmented sums over runs of equal ids (DPP-free shuffle form), run heads add; 4: no adds; 5: ds_add_rtn_f32 (returning form)
template <int MODE>
__global__ __launch_bounds__(256, 2) void hist_kernel(float* out, int iters, int R, int L, int layout) {
  extern __shared__ float hist[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < R * 6; i += 256) hist[i] = 0.f;
  __syncthreads();
  float acc = 0.f;
  const unsigned wid = (blockIdx.x * 4 + wave) * 0x9E3779B9u;
  for (int it = 0; it < iters; ++it) {
    const unsigned run = (unsigned)(lane + (it & 3)) / (unsigned)L;
    const unsigned id = hash32(wid + it * 977u + run * 131u) % (unsigned)R;
    const float v0 = 1.0f + lane * 0.001f, v1 = v0 * 0.37f, v2 = v0 * -0.11f;
    if (MODE == 0) {
      if (layout == 0) { atomicAdd(&hist[id * 3], v0); atomicAdd(&hist[id * 3 + 1], v1); atomicAdd(&hist[id * 3 + 2], v2); }
      else { atomicAdd(&hist[id], v0); atomicAdd(&hist[R + id], v1); atomicAdd(&hist[2 * R + id], v2); }
    } else if (MODE == 1) {
      typedef __attribute__((address_space(3))) float lds_f;
      if (layout == 0) {
        __builtin_amdgcn_ds_faddf((lds_f*)&hist[id * 3], v0, 0, 0, false); __builtin_amdgcn_ds_faddf((lds_f*)&hist[id * 3 + 1], v1, 0, 0, false);
        __builtin_amdgcn_ds_faddf((lds_f*)&hist[id * 3 + 2], v2, 0, 0, false);
      } else {
        __builtin_amdgcn_ds_faddf((lds_f*)&hist[id], v0, 0, 0, false); __builtin_amdgcn_ds_faddf((lds_f*)&hist[R + id], v1, 0, 0, false);
        __builtin_amdgcn_ds_faddf((lds_f*)&hist[2 * R + id], v2, 0, 0, false);
      }
    } else if (MODE == 2) {
      unsigned long long* h64 = reinterpret_cast<unsigned long long*>(hist);
      const long long i0 = (long long)(v0 * 1048576.f), i1 = (long long)(v1 * 1048576.f), i2 = (long long)(v2 * 1048576.f);
      if (layout == 0) { atomicAdd(&h64[id * 3], (unsigned long long)i0); atomicAdd(&h64[id * 3 + 1], (unsigned long long)i1); atomicAdd(&h64[id * 3 + 2], (unsigned long long)i2); }
      else { atomicAdd(&h64[id], (unsigned long long)i0); atomicAdd(&h64[R + id], (unsigned long long)i1); atomicAdd(&h64[2 * R + id], (unsigned long long)i2); }
    } else if (MODE == 3) {
      // segmented inclusive suffix sums over runs of equal ids (log steps), heads (first lane of a run) add the run's sum
      float s0 = v0, s1 = v1, s2 = v2;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float t0 = __shfl_down(s0, o), t1 = __shfl_down(s1, o), t2 = __shfl_down(s2, o);
        const unsigned oid = __shfl_down(id, o);
        const bool same = (lane + o < 64) && (oid == id);
        s0 += same ? t0 : 0.f; s1 += same ? t1 : 0.f; s2 += same ? t2 : 0.f;
      }
      const unsigned pid = __shfl_up(id, 1);
      if (lane == 0 || pid != id) {
        if (layout == 0) { atomicAdd(&hist[id * 3], s0); atomicAdd(&hist[id * 3 + 1], s1); atomicAdd(&hist[id * 3 + 2], s2); }
        else { atomicAdd(&hist[id], s0); atomicAdd(&hist[R + id], s1); atomicAdd(&hist[2 * R + id], s2); }
      }
    } else if (MODE == 5) {
      acc += atomicAdd(&hist[id * 3], v0) + atomicAdd(&hist[id * 3 + 1], v1) + atomicAdd(&hist[id * 3 + 2], v2);
    } else {
      acc += v0 * (float)id + v1 + v2;
    }
  }
  __syncthreads();
  for (int i = tid; i < R * 3; i += 256) acc += hist[i];
  out[blockIdx.x * 256 + tid] = acc;
}

template <int MODE>
static void run_hist(const char* name, int R, int L, int layout) {
  float* out;
  const int blocks = 512, iters = 4000;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = (size_t)R * 6 * 4;
  hist_kernel<MODE><<<blocks, 256, lds>>>(out, 100, R, L, layout);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hist_kernel<MODE><<<blocks, 256, lds>>>(out, iters, R, L, layout);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double pairs = (double)blocks * 256 * iters;
  printf("hist %-28s R=%5d run=%2d layout=%d : %7.3f ms  %6.2f ps/pair   -> 4e8 pairs: %6.3f ms\n", name, R, L, layout, ms, ms * 1e9 / pairs,
         ms * 4e8 / pairs);
  hipFree(out);
}

// gather: lanes 0..31 = consecutive queries of key A, 32..63 of key B; entry index = row(key, it) * G + col0(key, it) + stride * c
template <typename T, int AHEAD>
__global__ __launch_bounds__(256, 2) void gather_kernel(const T* __restrict__ tab, unsigned* out, int iters, int G, int stride) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hf = lane >> 5;
  const unsigned wid = (blockIdx.x * 4 + wave) * 0x9E3779B9u;
  unsigned acc = 0;
  T pend[AHEAD];
  auto addr = [&](int it) {
    const unsigned h = hash32(wid + (it * 2 + hf) * 2654435761u);
    const unsigned row = h % (unsigned)G, col0 = (h >> 12) % (unsigned)(G - 32 * stride);
    return (size_t)row * G + col0 + (unsigned)(stride * c);
  };
#pragma unroll
  for (int a = 0; a < AHEAD; ++a) pend[a] = tab[addr(a)];
  for (int it = 0; it < iters; it += AHEAD) {
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) {
      const T v = pend[a];
      pend[a] = tab[addr(it + AHEAD + a)];
      acc += (unsigned)v * 3u + 1u;
    }
  }
  out[blockIdx.x * 256 + tid] = acc;
}

template <typename T, int AHEAD>
static void run_gather(int G, int stride) {
  T* tab; unsigned* out;
  const int blocks = 512, iters = 4000;
  const size_t n = (size_t)G * G + 65536;
  hipMalloc(&tab, n * sizeof(T)); hipMemset(tab, 1, n * sizeof(T));
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  gather_kernel<T, AHEAD><<<blocks, 256>>>(tab, out, 100, G, stride);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  gather_kernel<T, AHEAD><<<blocks, 256>>>(tab, out, iters, G, stride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double pairs = (double)blocks * 256 * iters;
  printf("gather %zu-byte entries G=%4d (%5.1f MB) stride=%d ahead=%d : %7.3f ms  %6.2f ps/pair   -> 4e8 pairs: %6.3f ms\n", sizeof(T), G,
         n * sizeof(T) / 1e6, stride, AHEAD, ms, ms * 1e9 / pairs, ms * 4e8 / pairs);
  hipFree(tab); hipFree(out);
}

int main() {
  for (int layout = 0; layout < 2; ++layout)
    for (int L : {1, 4, 64}) {
      run_hist<0>("atomicAdd(float)", 2048, L, layout);
      run_hist<1>("ds_faddf builtin", 2048, L, layout);
    }
  for (int L : {1, 4}) {
    run_hist<2>("u64 fixed point", 2048, L, 0);
    run_hist<3>("segmented sums + heads", 2048, L, 0);
    run_hist<5>("atomicAdd returning", 2048, L, 0);
  }
  run_hist<4>("no adds (loop only)", 2048, 4, 0);
  run_hist<0>("atomicAdd(float)", 512, 4, 0);
  run_hist<0>("atomicAdd(float)", 4096, 4, 0);
  run_hist<3>("segmented sums + heads", 4096, 4, 0);
  run_gather<unsigned, 2>(1024, 5);
  run_gather<unsigned, 4>(1024, 5);
  run_gather<unsigned long long, 2>(1024, 5);
  run_gather<unsigned short, 2>(1024, 5);
  run_gather<unsigned, 2>(2048, 10);
  run_gather<unsigned, 2>(1024, 1);
  return 0;
}
