// Newton-Schulz pseudo-inverse iteration of the Nystrom block (models/NystromAttention.py:20-35, dup cmta_utils.py:144-159) as ONE host
// call per direction: the chain is 4 dependent batched m x m x m products per iteration forward and 8 + one elementwise update backward
// (6 iterations: 24 / 54 launches of ~17 us on 32 problems of 256^3).  Issued one by one from Python each launch costs ~25 us of host time
// (autograd Function + ctypes marshalling of 33 arguments) - more than the kernel - so the chain was host-bound even on its own stream.
// Here the host side is a C loop over smml_gemm_f32: ~3 us per launch.
//
//   z_{k+1} = 1/4 z_k (13 I - x z_k (15 I - x z_k (7 I - x z_k)))      evaluated as
//   xz = x z;  a = 7 xz - xz xz;  b = 15 xz - xz a;  z' = 3.25 z - 0.25 z b          (affine parts in the GEMM epilogues)
// backward (dz = gradient of z_{k+1}):
//   dzk = 3.25 dz - 0.25 dz b^T      db = -0.25 z^T dz         dxz = 15 db - db a^T        da = -xz^T db
//   dxz += 7 da - da xz^T - xz^T da  dx += dxz z^T             dz_k = dzk + x^T dxz
#include "smml_common.h"

extern "C" int smml_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* residual, int M, int N, int K,
                             long long sam, long long sak, long long sbk, long long sbn, long long ldc, long long ldr, int nb0, int nb1,
                             long long sa0, long long sa1, long long sb0, long long sb1, long long sc0, long long sc1, long long sbias0,
                             long long sbias1, int bias_mode, int rows_per_bias, long long bias_ld, int act, int splitk, int accumulate,
                             float alpha, float beta, void* stream);

namespace {

__global__ void axpy_kernel(float4* __restrict__ y, const float4* __restrict__ x, float a, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = y[i];
  const float4 u = x[i];
  v.x = fmaf(a, u.x, v.x); v.y = fmaf(a, u.y, v.y); v.z = fmaf(a, u.z, v.z); v.w = fmaf(a, u.w, v.w);
  y[i] = v;
}

// C = alpha op(A) op(B) + beta R over NB problems of m x m (row-major, contiguous); R may be C itself
int mm(const float* A, bool ta, const float* B, bool tb, float* C, const float* R, float alpha, float beta, int NB, int m, void* st) {
  const long long mm2 = (long long)m * m;
  return smml_gemm_f32(A, B, C, nullptr, R, m, m, m, ta ? 1 : m, ta ? m : 1, tb ? 1 : m, tb ? m : 1, m, m, NB, 1, mm2, 0, mm2, 0, mm2, 0,
                       0, 0, 0, 1, 0, 0, 1, 0, alpha, beta, st);
}

}  // namespace

#define SMML_TRY(call)      \
  do {                      \
    int rc_ = (call);       \
    if (rc_) return rc_;    \
  } while (0)

extern "C" {

// saved: [iters][4][NB, m, m] fp32 = (z_k, xz, a, b) of every iteration (slot [0][0] is not written: z_0 is the caller's z0);
// z_out [NB, m, m] = z_iters.  x, z0, saved, z_out must not overlap.
int smml_newton_schulz_fwd(const float* x, const float* z0, float* saved, float* z_out, int NB, int m, int iters, void* stream) {
  SMML_REQUIRE(x && z0 && saved && z_out, "smml_newton_schulz_fwd: null pointer");
  SMML_REQUIRE(NB > 0 && m > 0 && iters > 0, "smml_newton_schulz_fwd: bad sizes (NB=%d m=%d iters=%d)", NB, m, iters);
  const size_t per = (size_t)NB * m * m;
  for (int k = 0; k < iters; ++k) {
    float* slot = saved + (size_t)k * 4 * per;
    const float* z = k == 0 ? z0 : slot;
    float *xz = slot + per, *a = slot + 2 * per, *b = slot + 3 * per;
    float* zn = (k + 1 < iters) ? slot + 4 * per : z_out;
    SMML_TRY(mm(x, false, z, false, xz, nullptr, 1.f, 0.f, NB, m, stream));
    SMML_TRY(mm(xz, false, xz, false, a, xz, -1.f, 7.f, NB, m, stream));
    SMML_TRY(mm(xz, false, a, false, b, xz, -1.f, 15.f, NB, m, stream));
    SMML_TRY(mm(z, false, b, false, zn, z, -0.25f, 3.25f, NB, m, stream));
  }
  return SMML_OK;
}

// dz_in [NB, m, m]: gradient of z_iters.  dx, dz0 [NB, m, m] are overwritten.  scratch: 7 x [NB, m, m] floats.
int smml_newton_schulz_bwd(const float* x, const float* z0, const float* saved, const float* dz_in, float* dx, float* dz0, float* scratch,
                           int NB, int m, int iters, void* stream) {
  SMML_REQUIRE(x && z0 && saved && dz_in && dx && dz0 && scratch, "smml_newton_schulz_bwd: null pointer");
  SMML_REQUIRE(NB > 0 && m > 0 && iters > 0, "smml_newton_schulz_bwd: bad sizes (NB=%d m=%d iters=%d)", NB, m, iters);
  const size_t per = (size_t)NB * m * m;
  SMML_REQUIRE(per % 4 == 0, "smml_newton_schulz_bwd: NB m m must be a multiple of 4");
  float *pp[2] = {scratch, scratch + per}, *dzk = scratch + 2 * per, *db = scratch + 3 * per, *dxz = scratch + 4 * per,
        *da = scratch + 5 * per, *t = scratch + 6 * per;
  const float* dz = dz_in;
  for (int k = iters - 1; k >= 0; --k) {
    const float* slot = saved + (size_t)k * 4 * per;
    const float* z = k == 0 ? z0 : slot;
    const float *xz = slot + per, *a = slot + 2 * per, *b = slot + 3 * per;
    float* dzn = k == 0 ? dz0 : pp[k & 1];
    SMML_TRY(mm(dz, false, b, true, dzk, dz, -0.25f, 3.25f, NB, m, stream));
    SMML_TRY(mm(z, true, dz, false, db, nullptr, -0.25f, 0.f, NB, m, stream));
    SMML_TRY(mm(db, false, a, true, dxz, db, -1.f, 15.f, NB, m, stream));
    SMML_TRY(mm(xz, true, db, false, da, nullptr, -1.f, 0.f, NB, m, stream));
    SMML_TRY(mm(da, false, xz, true, t, dxz, -1.f, 1.f, NB, m, stream));
    SMML_TRY(mm(xz, true, da, false, t, t, -1.f, 1.f, NB, m, stream));
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((per / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<float4*>(t), reinterpret_cast<const float4*>(da), 7.f, per / 4);
    SMML_LAUNCH_CHECK("smml_newton_schulz_bwd/axpy");
    SMML_TRY(mm(t, false, z, true, dx, (k == iters - 1) ? nullptr : dx, 1.f, 1.f, NB, m, stream));
    SMML_TRY(mm(x, true, t, false, dzn, dzk, 1.f, 1.f, NB, m, stream));
    dz = dzn;
  }
  return SMML_OK;
}

}  // extern "C"
