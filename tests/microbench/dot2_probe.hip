// Checks on gfx950 that (1) v_dot2_f32_bf16(h_pk, {-1, 0} or {0, -1}, v) returns the exact residual v - float(h) of a bf16
// rounding, subnormals included, and (2) v_fma_f32 ... clamp with a large multiplier yields an exact 0.0 / 1.0 step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* v, float* r_dot, float* r_ref, float* st, const float* b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float2v pv = {v[2 * i], v[2 * i + 1]};
  const bf16x2 hh = __builtin_convertvector(pv, bf16x2);
  // constants opaque in VGPRs: as a folded inline constant {-1, 0} becomes "-1.0", which the instruction misreads
  unsigned c0, c1;
  asm("v_mov_b32 %0, 0x0000bf80" : "=v"(c0));
  asm("v_mov_b32 %0, 0xbf800000" : "=v"(c1));
  const bf16x2 n0 = __builtin_bit_cast(bf16x2, c0), n1 = __builtin_bit_cast(bf16x2, c1);
  // builtin, not inline asm: a DOT result needs 3 wait states before the next VALU read (compiler-inserted)
  const float a0 = __builtin_amdgcn_fdot2_f32_bf16(hh, n0, pv[0], false);
  const float a1 = __builtin_amdgcn_fdot2_f32_bf16(hh, n1, pv[1], false);
  r_dot[2 * i] = a0; r_dot[2 * i + 1] = a1;
  r_ref[2 * i] = pv[0] - (float)hh[0]; r_ref[2 * i + 1] = pv[1] - (float)hh[1];
  for (int t = 0; t < 2; ++t) {
    float big; asm("s_mov_b32 %0, 0x71800000" : "=s"(big));     // 2^100 in an SGPR so that the clamp modifier folds
    const float m = fminf(fmaxf(fmaf(pv[t], big, b[2 * i + t] * 0x1p100f), 0.f), 1.f);
    st[2 * i + t] = m;
  }
}
int main() {
  const int n = 1 << 22;
  std::vector<float> v(n), b(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    const int e = (i & 1023) < 16 ? -140 + (rand() % 20) : (rand() % 60) - 40;   // some subnormal inputs / residuals
    v[i] = ldexpf((float)rand() / RAND_MAX * 2.f - 1.f, e);
    b[i] = (i % 7 == 0) ? -v[i] : ((i % 7 == 1) ? -nextafterf(v[i], 1e30f) : ldexpf((float)rand() / RAND_MAX * 2.f - 1.f, (rand() % 40) - 30));
  }
  float *dv, *d1, *d2, *d3, *db;
  hipMalloc(&dv, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&d3, n * 4); hipMalloc(&db, n * 4);
  hipMemcpy(dv, v.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(dv, d1, d2, d3, db, n);
  std::vector<float> r1(n), r2(n), st(n);
  hipMemcpy(r1.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), d2, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(st.data(), d3, n * 4, hipMemcpyDeviceToHost);
  long bad = 0, badst = 0, frac = 0;
  for (int i = 0; i < n; ++i) {
    if (memcmp(&r1[i], &r2[i], 4) != 0 && !(r1[i] == 0.f && r2[i] == 0.f)) { if (bad++ < 5) printf("dot2 mismatch v=%a dot=%a ref=%a\n", v[i], r1[i], r2[i]); }
    const double s = (double)v[i] + (double)b[i];
    const float want = s > 0 ? 1.f : 0.f;
    if (st[i] != want) { if (st[i] > 0.f && st[i] < 1.f) ++frac; if (badst++ < 5) printf("step mismatch v=%a b=%a got=%a want=%a\n", v[i], b[i], st[i], want); }
  }
  printf("dot2 residual: %ld mismatches of %d; clamp step: %ld mismatches (%ld fractional)\n", bad, n, badst, frac);
  return 0;
}
