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
#include <atomic>     // process-wide measurement switches (set once from the environment or a test hook): plain atomics, no launch state
#include "smml_common.h"

extern "C" int smml_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* residual, int M, int N, int K,
                             long long sam, long long sak, long long sbk, long long sbn, long long ldc, long long ldr, int nb0, int nb1,
                             long long sa0, long long sa1, long long sb0, long long sb1, long long sc0, long long sc1, long long sbias0,
                             long long sbias1, int bias_mode, int rows_per_bias, long long bias_ld, int act, int splitk, int accumulate,
                             float alpha, float beta, void* stream);

#define SMML_TRY(call)      \
  do {                      \
    int rc_ = (call);       \
    if (rc_) return rc_;    \
  } while (0)

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

// workgroup id -> contiguous range of work items per XCD
__device__ __forceinline__ int xcd_linear(int id, int nitems) {
  const int xcd = id & 7, pos = id >> 3, q = nitems >> 3, r8 = nitems & 7;
  return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + pos;
}

// one exact-fp32 product into `acc` (rows on the register axis, column on the lane) in two parts: every panel load issued (the caller
// requests its residual right after), then the K loop; smem: [buffer][A | B] images
struct F32Panels { floatx4 pa[16], pbv[16]; };                         // every load of a workgroup's two 64 x 256 operand panels
template <bool TA, bool TB>
__device__ __forceinline__ void f32_load(const float* __restrict__ A, const float* __restrict__ B, F32Panels& P, int i0, int n0) {
  const int tid = threadIdx.x, t16 = tid & 15, th = tid >> 4;          // 16 lanes cover 256 contiguous bytes
  floatx4 (&pa)[16] = P.pa;
  floatx4 (&pbv)[16] = P.pbv;
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
}
template <bool TA, bool TB>
__device__ __forceinline__ void f32_compute(const F32Panels& P, float (*smem)[2][IMG], floatx16& acc) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int t16 = tid & 15, th = tid >> 4;
  const floatx4 (&pa)[16] = P.pa;
  const floatx4 (&pbv)[16] = P.pbv;
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
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void chain_mm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* C,
                                                          const float* R, float alpha, float beta, int NB) {
  __shared__ __attribute__((aligned(16))) float smem[2][2][IMG];      // [buffer][A | B]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
  const int lin = xcd_linear(blockIdx.x, NB * 16);
  const int prob = lin >> 4, tile = lin & 15;
  const int i0 = (tile >> 2) * CT, n0 = (tile & 3) * CT;
  const size_t pb = (size_t)prob * CM * CM;
  const size_t col = pb + n0 + (wave & 1) * 32 + c;
  const int rowb = i0 + (wave >> 1) * 32;
  F32Panels P;
  f32_load<TA, TB>(A + pb, B + pb, P, i0, n0);
  // the residual, requested behind the panels; ONE branch around all sixteen loads (a select per element makes hipcc wait for each in turn)
  float rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rv[r] = 0.f;
  if (R) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = R[col + (size_t)(rowb + acc_row(r, hf)) * CM];
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] *= beta;
  }
  floatx16 acc = {0};
  f32_compute<TA, TB>(P, smem, acc);
#pragma unroll
  for (int r = 0; r < 16; ++r) C[col + (size_t)(rowb + acc_row(r, hf)) * CM] = fmaf(alpha, acc[r], rv[r]);
}

// two products of the backward chain in one launch, as chain_bf3_dual_kernel below (see there): j0 an A B^T product, j1 an A^T B one
struct FJob { const float* A; const float* B; float* C; const float* R; float alpha, beta; };
__global__ __launch_bounds__(256, 2) void chain_mm_dual_kernel(FJob j0, FJob j1, int fuse, int NB) {
  __shared__ __attribute__((aligned(16))) float smem[2][2][IMG];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
  const int lin = xcd_linear(blockIdx.x, fuse ? NB * 16 : NB * 32);
  const int prob = fuse ? (lin >> 4) : (lin >> 5), job = fuse ? 0 : ((lin >> 4) & 1), tile = lin & 15;
  const int i0 = (tile >> 2) * CT, n0 = (tile & 3) * CT;
  const size_t pb = (size_t)prob * CM * CM;
  const size_t col = pb + n0 + (wave & 1) * 32 + c;
  const int rowb = i0 + (wave >> 1) * 32;
  const FJob& jj = (fuse || job == 0) ? j0 : j1;
  F32Panels P;
  if (fuse || job == 0) f32_load<false, true>(j0.A + pb, j0.B + pb, P, i0, n0);      // workgroup-uniform
  else f32_load<true, false>(j1.A + pb, j1.B + pb, P, i0, n0);
  float rv[16], rw[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { rv[r] = 0.f; rw[r] = 0.f; }
  if (jj.R) {                                                            // one branch around each batch of sixteen loads
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = jj.R[col + (size_t)(rowb + acc_row(r, hf)) * CM];
  }
  if (fuse && j1.R) {
#pragma unroll
    for (int r = 0; r < 16; ++r) rw[r] = j1.R[col + (size_t)(rowb + acc_row(r, hf)) * CM];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) rv[r] = fmaf(j1.beta, rw[r], jj.beta * rv[r]);
  floatx16 acc = {0};
  if (fuse) {
    f32_compute<false, true>(P, smem, acc);
    f32_load<true, false>(j1.A + pb, j1.B + pb, P, i0, n0);
    __syncthreads();                                                     // the images are read to the end before the second product refills them
    f32_compute<true, false>(P, smem, acc);
  } else if (job == 0) {
    f32_compute<false, true>(P, smem, acc);
  } else {
    f32_compute<true, false>(P, smem, acc);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) jj.C[col + (size_t)(rowb + acc_row(r, hf)) * CM] = fmaf(jj.alpha, acc[r], rv[r]);
}

// ------------------------------------------------------------------------------------------------
// The same product on the 16-bit matrix pipe ("planes" form, m = 256) for the block's 16-bit compute mode: every matrix of the chain
// lives in memory as TWO bf16 planes [2][NB, m, m] - x = h + l to 2^-17 relative, fp32's exponent range - written once by the product
// that makes it (16 values per lane to split) and read as they stand by the products that consume it; three of the four cross products
// are kept (h h, h l, l h: <= 2^-16 |a||b| per product, i.e. ~100x finer than the bf16 attention products around the chain, and the
// forward iteration corrects its own errors).  A 32 x 32 x 16 block costs 3 MFMAs of 32 cycles instead of 8 fp32 MFMAs of 64 (1.5 us of
// matrix time per SIMD and product instead of 7.8) at the same bytes per element as fp32.  Measured on 32 problems (tests/bench_pinv_chain.py),
// a product of the chain takes 13-15 us as an exact-fp32 launch, of which ~5 us are the kernel boundary and ~8 the matrix pipe; a
// three-plane fp32-grade form (six products, 6 bytes per element) was built and dropped: 15-18 us - the launch is then bound by the
// 112 MB its 512 workgroups pull through L2, not by the MFMAs.
// Tile 64 x 64 per workgroup (2 x 2 waves), K in four steps of 64 through one LDS buffer; the loads of the first two steps are issued up
// front, those of step s + 2 right after step s has been written to LDS.  Images as in gemm.hip: k-contiguous operand [row][64 k + 8]
// (ds_read_b128 fragments), row-contiguous [k][64 rows + 32] (ds_read_b64_tr_b16).  The accumulator is transposed (rows on lanes, 4
// consecutive columns per register group) so that the epilogue reads R and writes C as 8-byte pieces per plane.
// ------------------------------------------------------------------------------------------------
constexpr int NPL = 2;                          // planes per matrix
// four fp32 -> two planes of four bf16: h = rn(v), l = rn(v - h)
__device__ __forceinline__ void split4_b2(const float4 v, uint2v& h, uint2v& l) {
  const float2v a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const float2v ra = bf16_residual2(a, ha), rb = bf16_residual2(b, hb);
  const bf16x2 la = __builtin_convertvector(ra, bf16x2), lb = __builtin_convertvector(rb, bf16x2);
  h = (uint2v){__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  l = (uint2v){__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}
constexpr int PK_LD = CKS + 8;                  // halves per row, k-contiguous image
constexpr int PR_LD = CT + 32;                  // halves per k-row, row-contiguous image
constexpr int PIMG = CT * PR_LD;                // halves per image (6144; the k-contiguous one needs 64 x 72 = 4608)

// the K loop of one product into `acc` (the transposed 32 x 32 block of this wave); smem: [A | B][plane] images, free on entry
struct PlaneRing { uint4v ra[2][2 * NPL], rb[2][2 * NPL]; };            // [ring slot][plane x 2 loads]
template <bool TA, bool TB>
__device__ __forceinline__ void planes_load_step(const __bf16* __restrict__ A, const __bf16* __restrict__ B, PlaneRing& Q, int s, int slot, size_t pb,
                                                 int i0, int n0, size_t plane) {
  const int tid = threadIdx.x, t8 = tid & 7, th = tid >> 3;              // 8 lanes cover 128 contiguous bytes
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = th + 32 * j;                                        // row (k-contiguous) or k (row-contiguous) within the step
      const __bf16* ap = A + p * plane + pb + (TA ? (size_t)(CKS * s + rr) * CM + i0 + 8 * t8 : (size_t)(i0 + rr) * CM + CKS * s + 8 * t8);
      const __bf16* bp = B + p * plane + pb + (TB ? (size_t)(n0 + rr) * CM + CKS * s + 8 * t8 : (size_t)(CKS * s + rr) * CM + n0 + 8 * t8);
      Q.ra[slot][2 * p + j] = *reinterpret_cast<const uint4v*>(ap);
      Q.rb[slot][2 * p + j] = *reinterpret_cast<const uint4v*>(bp);
    }
}
// the K loop; the loads of steps 0 and 1 are already in the ring (planes_load_step: the caller requests its residual behind them)
template <bool TA, bool TB>
__device__ __forceinline__ void planes_product(const __bf16* __restrict__ A, const __bf16* __restrict__ B, PlaneRing& Q, __bf16* smem, floatx16& acc,
                                               size_t pb, int i0, int n0, size_t plane) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int t8 = tid & 7, th = tid >> 3;
  uint4v (&ra)[2][2 * NPL] = Q.ra;
  uint4v (&rb)[2][2 * NPL] = Q.rb;
  auto load_step = [&](int s, int slot) { planes_load_step<TA, TB>(A, B, Q, s, slot, pb, i0, n0, plane); };
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int slot = s & 1;
    if (s) __syncthreads();                                              // the previous step's fragment reads are done
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rr = th + 32 * j;
        *reinterpret_cast<uint4v*>(&smem[p * PIMG + (TA ? rr * PR_LD : rr * PK_LD) + 8 * t8]) = ra[slot][2 * p + j];
        *reinterpret_cast<uint4v*>(&smem[(NPL + p) * PIMG + (TB ? rr * PK_LD : rr * PR_LD) + 8 * t8]) = rb[slot][2 * p + j];
      }
    __syncthreads();
    if (s + 2 < 4) load_step(s + 2, slot);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      bf16x8 af[NPL], bf[NPL];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        const __bf16* ia = &smem[p * PIMG];
        const __bf16* ib = &smem[(NPL + p) * PIMG];
        if (!TA) af[p] = *reinterpret_cast<const bf16x8*>(&ia[(wm * 32 + c) * PK_LD + 16 * kb + 8 * hf]);
        else {
          const __bf16* q0 = &ia[(16 * kb + 8 * hf + trq) * PR_LD + wm * 32 + trc];
          af[p] = lds_frag_tr(q0, q0 + 4 * PR_LD);
        }
        if (TB) bf[p] = *reinterpret_cast<const bf16x8*>(&ib[(wn * 32 + c) * PK_LD + 16 * kb + 8 * hf]);
        else {
          const __bf16* q0 = &ib[(16 * kb + 8 * hf + trq) * PR_LD + wn * 32 + trc];
          bf[p] = lds_frag_tr(q0, q0 + 4 * PR_LD);
        }
      }
      // transposed block: D^T[n, i] = sum_k opB(k, n) opA(i, k) - the B fragments are the MFMA's A operand
      acc = mfma16b(bf[1], af[0], acc);      // smallest terms first
      acc = mfma16b(bf[0], af[1], acc);
      acc = mfma16b(bf[0], af[0], acc);
    }
  }
}
// this lane's 16 values (row i, columns 8 g + 4 hf + j of the wave's block) of a planes matrix, times w, added to rv
__device__ __forceinline__ void planes_residual(const __bf16* __restrict__ R, float w, float (&rv)[16], size_t off0, size_t plane) {
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(R + p * plane + off0 + 8 * g);
#pragma unroll
      for (int i = 0; i < 4; ++i) rv[4 * g + i] = fmaf(w, (float)v[i], rv[4 * g + i]);     // h first, then l
    }
}
// C (planes, and its fp32 copy when Cf is given) = alpha acc + rv
__device__ __forceinline__ void planes_store(const floatx16& acc, const float (&rv)[16], __bf16* C, float* __restrict__ Cf, float alpha, size_t off0,
                                             size_t plane) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = make_float4(fmaf(alpha, acc[4 * g], rv[4 * g]), fmaf(alpha, acc[4 * g + 1], rv[4 * g + 1]), fmaf(alpha, acc[4 * g + 2], rv[4 * g + 2]),
                                 fmaf(alpha, acc[4 * g + 3], rv[4 * g + 3]));
    uint2v h, l;
    split4_b2(v, h, l);
    *reinterpret_cast<uint2v*>(C + off0 + 8 * g) = h;
    *reinterpret_cast<uint2v*>(C + plane + off0 + 8 * g) = l;
    if (Cf) *reinterpret_cast<float4*>(Cf + off0 + 8 * g) = v;
  }
}
template <bool TA, bool TB>
__global__ __launch_bounds__(256, 2) void chain_bf3_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* C,
                                                           const __bf16* R, float* __restrict__ Cf, float alpha, float beta, int NB,
                                                           size_t plane) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * NPL * PIMG];      // [A | B][plane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lin = xcd_linear(blockIdx.x, NB * 16);
  const int prob = lin >> 4, tile = lin & 15;
  const int i0 = (tile >> 2) * CT, n0 = (tile & 3) * CT;
  const size_t pb = (size_t)prob * CM * CM;
  const size_t off0 = pb + (size_t)(i0 + (wave >> 1) * 32 + (lane & 31)) * CM + n0 + (wave & 1) * 32 + 4 * (lane >> 5);
  PlaneRing Q;
  planes_load_step<TA, TB>(A, B, Q, 0, 0, pb, i0, n0, plane);
  planes_load_step<TA, TB>(A, B, Q, 1, 1, pb, i0, n0, plane);
  float rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rv[r] = 0.f;
  if (R) planes_residual(R, beta, rv, off0, plane);                      // requested behind the first two steps' operands
  floatx16 acc = {0};
  planes_product<TA, TB>(A, B, Q, smem, acc, pb, i0, n0, plane);
  planes_store(acc, rv, C, Cf, alpha, off0, plane);
}

// Two products of the backward chain in ONE launch - every pair the backward offers is an (A B^T, A^T B) pair:
//   fuse = 0: two independent results, C0 = alpha0 A0 B0^T + beta0 R0 and C1 = alpha1 A1^T B1 + beta1 R1 (2 x NB x 16 workgroups; the
//             two jobs of a problem sit next to each other so that they share one L2);
//   fuse = 1: one result, C0 = alpha0 (A0 B0^T + A1^T B1) + beta0 R0 + beta1 R1 (dxz + 7 da - da xz^T - xz^T da in one kernel).
// A launch boundary costs ~5 us against 5 - 7 us of work in a product: 9 launches per backward iteration become 4.
struct PJob { const __bf16* A; const __bf16* B; __bf16* C; const __bf16* R; float* Cf; float alpha, beta; };
__global__ __launch_bounds__(256, 2) void chain_bf3_dual_kernel(PJob j0, PJob j1, int fuse, int NB, size_t plane) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * NPL * PIMG];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lin = xcd_linear(blockIdx.x, fuse ? NB * 16 : NB * 32);
  const int prob = fuse ? (lin >> 4) : (lin >> 5), job = fuse ? 0 : ((lin >> 4) & 1), tile = lin & 15;
  const int i0 = (tile >> 2) * CT, n0 = (tile & 3) * CT;
  const size_t pb = (size_t)prob * CM * CM;
  const size_t off0 = pb + (size_t)(i0 + (wave >> 1) * 32 + (lane & 31)) * CM + n0 + (wave & 1) * 32 + 4 * (lane >> 5);
  float rv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rv[r] = 0.f;
  floatx16 acc = {0};
  PlaneRing Q;
  if (fuse) {
    planes_load_step<false, true>(j0.A, j0.B, Q, 0, 0, pb, i0, n0, plane);
    planes_load_step<false, true>(j0.A, j0.B, Q, 1, 1, pb, i0, n0, plane);
    if (j0.R) planes_residual(j0.R, j0.beta, rv, off0, plane);
    if (j1.R) planes_residual(j1.R, j1.beta, rv, off0, plane);
    planes_product<false, true>(j0.A, j0.B, Q, smem, acc, pb, i0, n0, plane);
    planes_load_step<true, false>(j1.A, j1.B, Q, 0, 0, pb, i0, n0, plane);
    planes_load_step<true, false>(j1.A, j1.B, Q, 1, 1, pb, i0, n0, plane);
    __syncthreads();                                                     // the images are read to the end before the second product refills them
    planes_product<true, false>(j1.A, j1.B, Q, smem, acc, pb, i0, n0, plane);
    planes_store(acc, rv, j0.C, j0.Cf, j0.alpha, off0, plane);
  } else if (job == 0) {                                                 // workgroup-uniform
    planes_load_step<false, true>(j0.A, j0.B, Q, 0, 0, pb, i0, n0, plane);
    planes_load_step<false, true>(j0.A, j0.B, Q, 1, 1, pb, i0, n0, plane);
    if (j0.R) planes_residual(j0.R, j0.beta, rv, off0, plane);
    planes_product<false, true>(j0.A, j0.B, Q, smem, acc, pb, i0, n0, plane);
    planes_store(acc, rv, j0.C, j0.Cf, j0.alpha, off0, plane);
  } else {
    planes_load_step<true, false>(j1.A, j1.B, Q, 0, 0, pb, i0, n0, plane);
    planes_load_step<true, false>(j1.A, j1.B, Q, 1, 1, pb, i0, n0, plane);
    if (j1.R) planes_residual(j1.R, j1.beta, rv, off0, plane);
    planes_product<true, false>(j1.A, j1.B, Q, smem, acc, pb, i0, n0, plane);
    planes_store(acc, rv, j1.C, j1.Cf, j1.alpha, off0, plane);
  }
}

// fp32 [n] -> two bf16 planes
__global__ void split_planes_kernel(const float4* __restrict__ x, __bf16* __restrict__ P, size_t n4, size_t plane) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  uint2v h, l;
  split4_b2(x[i], h, l);
  *reinterpret_cast<uint2v*>(P + 4 * i) = h;
  *reinterpret_cast<uint2v*>(P + plane + 4 * i) = l;
}
// planes form of mm(): C (and, when Cf is given, its fp32 copy) = alpha op(A) op(B) + beta R;  R may be C itself
int mm_p(const __bf16* A, bool ta, const __bf16* B, bool tb, __bf16* C, const __bf16* R, float* Cf, float alpha, float beta, int NB, void* st) {
  const size_t plane = (size_t)NB * CM * CM;
  dim3 grid((unsigned)(NB * 16)), block(256);
  hipStream_t s = (hipStream_t)st;
  if (ta && tb) hipLaunchKernelGGL((chain_bf3_kernel<true, true>), grid, block, 0, s, A, B, C, R, Cf, alpha, beta, NB, plane);
  else if (ta) hipLaunchKernelGGL((chain_bf3_kernel<true, false>), grid, block, 0, s, A, B, C, R, Cf, alpha, beta, NB, plane);
  else if (tb) hipLaunchKernelGGL((chain_bf3_kernel<false, true>), grid, block, 0, s, A, B, C, R, Cf, alpha, beta, NB, plane);
  else hipLaunchKernelGGL((chain_bf3_kernel<false, false>), grid, block, 0, s, A, B, C, R, Cf, alpha, beta, NB, plane);
  SMML_LAUNCH_CHECK("smml_newton_schulz/chain_bf3");
  return SMML_OK;
}
// two products in one launch (chain_bf3_dual_kernel): j0 is an A B^T product, j1 an A^T B one
int mm_p2(const PJob& j0, const PJob& j1, int fuse, int NB, void* st) {
  const size_t plane = (size_t)NB * CM * CM;
  dim3 grid((unsigned)(NB * (fuse ? 16 : 32))), block(256);
  hipLaunchKernelGGL(chain_bf3_dual_kernel, grid, block, 0, (hipStream_t)st, j0, j1, fuse, NB, plane);
  SMML_LAUNCH_CHECK("smml_newton_schulz/chain_bf3_dual");
  return SMML_OK;
}
int split_p(const float* x, __bf16* P, int NB, void* st) {
  const size_t plane = (size_t)NB * CM * CM;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((plane / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)st,
                     reinterpret_cast<const float4*>(x), P, plane / 4, plane);
  SMML_LAUNCH_CHECK("smml_newton_schulz/split");
  return SMML_OK;
}

// Which form a chain of m x m problems runs in: 0 every product through smml_gemm_f32 (any m), 1 the exact-fp32 chain kernel (m = 256),
// 2 the two-plane form on the 16-bit pipe (m = 256, only where the caller asked for reduced precision: the 16-bit compute mode).
// The switch (smml_newton_schulz_set_fast / SMML_CHAIN_FAST): 0 = form 0 everywhere, 1 = form 1 also where reduced precision was asked for,
// 2 (default) = as described.
static std::atomic<int> g_chain_fast{-1};
static int chain_form(int m, int reduced) {
  if (g_chain_fast < 0) { const char* e = getenv("SMML_CHAIN_FAST"); g_chain_fast = e ? atoi(e) : 2; }
  if (m != CM || g_chain_fast == 0) return 0;
  return (reduced && g_chain_fast == 2) ? 2 : 1;
}

// C = alpha op(A) op(B) + beta R over NB problems of m x m (row-major, contiguous); R may be C itself
int mm(const float* A, bool ta, const float* B, bool tb, float* C, const float* R, float alpha, float beta, int NB, int m, void* st) {
  auto al16 = [](const void* p) { return (((size_t)p) & 15) == 0; };
  if (chain_form(m, 0) && al16(A) && al16(B) && (long long)NB * 16 < (1LL << 31)) {
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

// exact-fp32 form: two products in one launch (chain_mm_dual_kernel)
static int mm2(const FJob& j0, const FJob& j1, int fuse, int NB, void* st) {
  dim3 grid((unsigned)(NB * (fuse ? 16 : 32))), block(256);
  hipLaunchKernelGGL(chain_mm_dual_kernel, grid, block, 0, (hipStream_t)st, j0, j1, fuse, NB);
  SMML_LAUNCH_CHECK("smml_newton_schulz/chain_mm_dual");
  return SMML_OK;
}

extern "C" {

void smml_newton_schulz_set_fast(int on) { g_chain_fast = on; }

// sizes (in floats) of the buffers the two entry points below need: they depend on the form the chain runs in
size_t smml_newton_schulz_saved_floats(int NB, int m, int iters, int reduced) {
  if (NB <= 0 || m <= 0 || iters <= 0) return 0;
  const size_t per = (size_t)NB * m * m;
  const int form = chain_form(m, reduced);
  return form == 2 ? (size_t)(3 + 4 * iters) * (NPL * per / 2) : (size_t)iters * 4 * per;
}
size_t smml_newton_schulz_scratch_floats(int NB, int m, int iters, int reduced) {
  if (NB <= 0 || m <= 0 || iters <= 0) return 0;
  const size_t per = (size_t)NB * m * m;
  const int form = chain_form(m, reduced);
  return form == 2 ? (size_t)10 * (NPL * per / 2) : (size_t)7 * per;
}

// planes form: saved = [x | z0 | (z_k, xz, a, b) per iteration | spare] as bf16 planes tensors of NPL NB m m halves each
static int ns_fwd_planes(const float* x, const float* z0, float* saved, float* z_out, int NB, int iters, void* stream) {
  const size_t P = (size_t)NPL * NB * CM * CM;
  __bf16* base = reinterpret_cast<__bf16*>(saved);
  __bf16 *xp = base, *z0p = base + P, *spare = base + (size_t)(2 + 4 * iters) * P;
  SMML_TRY(split_p(x, xp, NB, stream));
  SMML_TRY(split_p(z0, z0p, NB, stream));
  for (int k = 0; k < iters; ++k) {
    __bf16* slot = base + (size_t)(2 + 4 * k) * P;
    const __bf16* z = k == 0 ? z0p : slot;
    __bf16 *xz = slot + P, *a = slot + 2 * P, *b = slot + 3 * P;
    const bool last = k + 1 == iters;
    __bf16* zn = last ? spare : slot + 4 * P;
    SMML_TRY(mm_p(xp, false, z, false, xz, nullptr, nullptr, 1.f, 0.f, NB, stream));
    SMML_TRY(mm_p(xz, false, xz, false, a, xz, nullptr, -1.f, 7.f, NB, stream));
    SMML_TRY(mm_p(xz, false, a, false, b, xz, nullptr, -1.f, 15.f, NB, stream));
    SMML_TRY(mm_p(z, false, b, false, zn, z, last ? z_out : nullptr, -0.25f, 3.25f, NB, stream));
  }
  return SMML_OK;
}

static int ns_bwd_planes(const float* saved, const float* dz_in, float* dx, float* dz0, float* scratch, int NB, int iters, void* stream) {
  const size_t P = (size_t)NPL * NB * CM * CM, plane = P / NPL;
  const __bf16* base = reinterpret_cast<const __bf16*>(saved);
  const __bf16 *xp = base, *z0p = base + P;
  __bf16* sc = reinterpret_cast<__bf16*>(scratch);
  __bf16 *dzi = sc, *pp[2] = {sc + P, sc + 2 * P}, *dzk = sc + 3 * P, *db = sc + 4 * P, *dxz = sc + 5 * P, *da = sc + 6 * P, *t = sc + 7 * P,
         *dxp = sc + 8 * P, *dz0p = sc + 9 * P;
  SMML_TRY(split_p(dz_in, dzi, NB, stream));
  const __bf16* dz = dzi;
  for (int k = iters - 1; k >= 0; --k) {
    const __bf16* slot = base + (size_t)(2 + 4 * k) * P;
    const __bf16* z = k == 0 ? z0p : slot;
    const __bf16 *xz = slot + P, *a = slot + 2 * P, *b = slot + 3 * P;
    __bf16* dzn = k == 0 ? dz0p : pp[k & 1];
    // four launches per iteration: [dzk | db], [dxz | da], t (two products + two residuals fused), [dx | dz]
    SMML_TRY(mm_p2(PJob{dz, b, dzk, dz, nullptr, -0.25f, 3.25f}, PJob{z, dz, db, nullptr, nullptr, -0.25f, 0.f}, 0, NB, stream));
    SMML_TRY(mm_p2(PJob{db, a, dxz, db, nullptr, -1.f, 15.f}, PJob{xz, db, da, nullptr, nullptr, -1.f, 0.f}, 0, NB, stream));
    SMML_TRY(mm_p2(PJob{da, xz, t, dxz, nullptr, -1.f, 1.f}, PJob{xz, da, nullptr, da, nullptr, -1.f, 7.f}, 1, NB, stream));
    SMML_TRY(mm_p2(PJob{t, z, dxp, (k == iters - 1) ? nullptr : dxp, k == 0 ? dx : nullptr, 1.f, 1.f},
                   PJob{xp, t, dzn, dzk, k == 0 ? dz0 : nullptr, 1.f, 1.f}, 0, NB, stream));
    dz = dzn;
  }
  return SMML_OK;
}


// reduced != 0: the caller accepts 16-bit-mantissa products (the block's 16-bit compute mode).
// saved: smml_newton_schulz_saved_floats(NB, m, iters, reduced) floats - what the backward needs of every iteration ((z_k, xz, a, b) in fp32, or
// in the planes form also x and z0, as bf16 planes); the forward and the backward of one call must run in the same form (the mode switch is
// not to be changed between them).  z_out [NB, m, m] = z_iters.  x, z0, saved, z_out must not overlap.
int smml_newton_schulz_fwd(const float* x, const float* z0, float* saved, float* z_out, int NB, int m, int iters, int reduced, void* stream) {
  SMML_REQUIRE(x && z0 && saved && z_out, "smml_newton_schulz_fwd: null pointer");
  SMML_REQUIRE(NB > 0 && m > 0 && iters > 0, "smml_newton_schulz_fwd: bad sizes (NB=%d m=%d iters=%d)", NB, m, iters);
  auto al16 = [](const void* p) { return (((size_t)p) & 15) == 0; };
  SMML_REQUIRE(al16(x) && al16(z0) && al16(saved) && al16(z_out), "smml_newton_schulz_fwd: 16-byte aligned buffers needed");
  if (chain_form(m, reduced) == 2) return ns_fwd_planes(x, z0, saved, z_out, NB, iters, stream);
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

// dz_in [NB, m, m]: gradient of z_iters.  dx, dz0 [NB, m, m] are overwritten.  scratch: smml_newton_schulz_scratch_floats(NB, m, iters, reduced) floats.
int smml_newton_schulz_bwd(const float* x, const float* z0, const float* saved, const float* dz_in, float* dx, float* dz0, float* scratch,
                           int NB, int m, int iters, int reduced, void* stream) {
  SMML_REQUIRE(x && z0 && saved && dz_in && dx && dz0 && scratch, "smml_newton_schulz_bwd: null pointer");
  SMML_REQUIRE(NB > 0 && m > 0 && iters > 0, "smml_newton_schulz_bwd: bad sizes (NB=%d m=%d iters=%d)", NB, m, iters);
  auto al16 = [](const void* p) { return (((size_t)p) & 15) == 0; };
  SMML_REQUIRE(al16(x) && al16(z0) && al16(saved) && al16(dz_in) && al16(dx) && al16(dz0) && al16(scratch), "smml_newton_schulz_bwd: 16-byte aligned buffers needed");
  if (chain_form(m, reduced) == 2) return ns_bwd_planes(saved, dz_in, dx, dz0, scratch, NB, iters, stream);
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
    auto al16f = [](const void* p) { return (((size_t)p) & 15) == 0; };
    if (chain_form(m, 0) == 1 && al16f(x) && al16f(saved) && al16f(scratch) && al16f(dz_in) && al16f(dx) && al16f(dz0)) {
      // four launches per iteration: [dzk | db], [dxz | da], t (two products + two residuals fused), [dx | dz]
      // (the exact form is bound by the fp32 matrix pipe, 7.8 us per product: pairing buys it only the ~5 us launch boundaries)
      SMML_TRY(mm2(FJob{dz, b, dzk, dz, -0.25f, 3.25f}, FJob{z, dz, db, nullptr, -0.25f, 0.f}, 0, NB, stream));
      SMML_TRY(mm2(FJob{db, a, dxz, db, -1.f, 15.f}, FJob{xz, db, da, nullptr, -1.f, 0.f}, 0, NB, stream));
      SMML_TRY(mm2(FJob{da, xz, t, dxz, -1.f, 1.f}, FJob{xz, da, nullptr, da, -1.f, 7.f}, 1, NB, stream));
      SMML_TRY(mm2(FJob{t, z, dx, (k == iters - 1) ? nullptr : dx, 1.f, 1.f}, FJob{x, t, dzn, dzk, 1.f, 1.f}, 0, NB, stream));
      dz = dzn;
      continue;
    }
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
