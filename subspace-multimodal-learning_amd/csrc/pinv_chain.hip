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
#include <cstdlib>
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

// ------------------------------------------------------------------------------------------------
// One product of the chain for m = 256: C = alpha op(A) op(B) + beta R on NB problems, exact fp32 (v_mfma_f32_32x32x2_f32).
// The launch is small (32 x 16 tiles of 64 x 64, K = 256: two workgroups per CU) and sits in a chain of dependent launches, so what
// counts is its latency: every global load of the workgroup's two 64 x 256 operand panels (16 + 16 float4 per thread) and of the
// residual is issued up front - one exposed memory round trip instead of one per K tile - and the panels then pass through LDS in four
// K steps of 64 (double-buffered images, one barrier per step) while the MFMAs run.  Matrix time of a SIMD: 2 waves x 128 MFMAs x 64
// cycles = 7.8 us at 2.1 GHz, which is the floor of this decomposition (the generic 64-row tile of gemm.hip takes 17-21 us).
// A k-contiguous operand (op(A) = A, op(B) = B^T) is staged as [row][64 k + 4] and read as one float4 per lane and four MFMAs (lane
// half h takes k = 8 u + 4 h + j for MFMA j of group u); a row-contiguous one (A^T, B) as [k][64 rows] and read one float per MFMA
// with the same k assignment.  Workgroup ids are remapped so that the 16 tiles of a problem share one XCD's L2.
// ------------------------------------------------------------------------------------------------
constexpr int CM = 256, CT = 64, CKS = 64;     // matrix size, tile, K step
constexpr int KC_LDF = CKS + 4;                // floats per row of a k-contiguous image
constexpr int IMG = CT * KC_LDF;               // floats per image (the row-contiguous one, 64 x 64, fits too)

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void chain_mm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* C,
                                                          const float* R, float alpha, float beta, int NB) {
  __shared__ __attribute__((aligned(16))) float smem[2][2][IMG];      // [buffer][A | B]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntiles = NB * 16, id = blockIdx.x;
  const int xcd = id & 7, pos = id >> 3, q = ntiles >> 3, r8 = ntiles & 7;
  const int lin = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + pos;
  const int prob = lin >> 4, tile = lin & 15;
  const int i0 = (tile >> 2) * CT, n0 = (tile & 3) * CT;
  const size_t pb = (size_t)prob * CM * CM;
  A += pb; B += pb; C += pb;
  const int t16 = tid & 15, th = tid >> 4;                              // 16 lanes cover 256 contiguous bytes
  // every load of the two panels, then the residual
  floatx4 pa[16], pbv[16];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rr = th + 16 * j;
      pa[4 * s + j] = TA ? *reinterpret_cast<const floatx4*>(A + (size_t)(CKS * s + rr) * CM + i0 + 4 * t16)       // A stored [k][i]
                         : *reinterpret_cast<const floatx4*>(A + (size_t)(i0 + rr) * CM + CKS * s + 4 * t16);      // A stored [i][k]
      pbv[4 * s + j] = TB ? *reinterpret_cast<const floatx4*>(B + (size_t)(n0 + rr) * CM + CKS * s + 4 * t16)      // B stored [n][k]
                          : *reinterpret_cast<const floatx4*>(B + (size_t)(CKS * s + rr) * CM + n0 + 4 * t16);     // B stored [k][n]
    }
  const int col = n0 + wn * 32 + c, rowb = i0 + wm * 32;
  float rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rv[r] = R ? R[pb + (size_t)(rowb + acc_row(r, hf)) * CM + col] : 0.f;

  floatx16 acc = {0};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float* As = smem[s & 1][0];
    float* Bs = smem[s & 1][1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rr = th + 16 * j;
      *reinterpret_cast<floatx4*>(&As[TA ? (rr * CT + 4 * t16) : (rr * KC_LDF + 4 * t16)]) = pa[4 * s + j];
      *reinterpret_cast<floatx4*>(&Bs[TB ? (rr * KC_LDF + 4 * t16) : (rr * CT + 4 * t16)]) = pbv[4 * s + j];
    }
    __syncthreads();      // buffer (s & 1) was last read in step s - 2, which every wave left before the barrier of step s - 1
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float a[4], b[4];
      const int kq = 8 * u + 4 * hf;
      if (!TA) {
        const floatx4 v = *reinterpret_cast<const floatx4*>(&As[(wm * 32 + c) * KC_LDF + kq]);
        a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = As[(kq + j) * CT + wm * 32 + c];
      }
      if (TB) {
        const floatx4 v = *reinterpret_cast<const floatx4*>(&Bs[(wn * 32 + c) * KC_LDF + kq]);
        b[0] = v[0]; b[1] = v[1]; b[2] = v[2]; b[3] = v[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[(kq + j) * CT + wn * 32 + c];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = mfma32(a[j], b[j], acc);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) C[(size_t)(rowb + acc_row(r, hf)) * CM + col] = fmaf(beta, rv[r], alpha * acc[r]);
}

static int g_chain_fast = -1;      // -1: read SMML_CHAIN_FAST (default 1); 0: every product through smml_gemm_f32 (measurement / test switch)

// C = alpha op(A) op(B) + beta R over NB problems of m x m (row-major, contiguous); R may be C itself
int mm(const float* A, bool ta, const float* B, bool tb, float* C, const float* R, float alpha, float beta, int NB, int m, void* st) {
  if (g_chain_fast < 0) { const char* e = getenv("SMML_CHAIN_FAST"); g_chain_fast = e ? atoi(e) : 1; }
  auto al16 = [](const void* p) { return (((size_t)p) & 15) == 0; };
  if (g_chain_fast && m == CM && al16(A) && al16(B) && (long long)NB * 16 < (1LL << 31)) {
    dim3 grid((unsigned)(NB * 16)), block(256);
    hipStream_t s = (hipStream_t)st;
    if (ta && tb) hipLaunchKernelGGL((chain_mm_kernel<true, true>), grid, block, 0, s, A, B, C, R, alpha, beta, NB);
    else if (ta) hipLaunchKernelGGL((chain_mm_kernel<true, false>), grid, block, 0, s, A, B, C, R, alpha, beta, NB);
    else if (tb) hipLaunchKernelGGL((chain_mm_kernel<false, true>), grid, block, 0, s, A, B, C, R, alpha, beta, NB);
    else hipLaunchKernelGGL((chain_mm_kernel<false, false>), grid, block, 0, s, A, B, C, R, alpha, beta, NB);
    SMML_LAUNCH_CHECK("smml_newton_schulz/chain_mm");
    return SMML_OK;
  }
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

void smml_newton_schulz_set_fast(int on) { g_chain_fast = on; }

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
