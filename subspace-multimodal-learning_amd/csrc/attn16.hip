// Fused softmax attention on the 16-bit matrix pipe of gfx950 (v_mfma_f32_32x32x16_{bf16,f16}), forward and backward, for the two
// attention-shaped products of the Nystrom landmark block (models/NystromAttention.py:122-140; dup cmta_utils.py:246-264):
//
//   a1 side   softmax(q kl^T) W        queries = the n' tokens,      keys = the m landmarks, values W = z (attn3 v)   [m, d]
//   a3 side   softmax(ql k^T) v        queries = the m landmarks,    keys = the n' tokens,   values v                 [n', d]
//
// (out = (attn1 z)(attn3 v) of :140 is evaluated as attn1 (z (attn3 v)): the [n', m] x [m, m] product of the reference becomes an
// [m, m] x [m, d] one; the [n', m] and [m, n'] probability matrices are never written to HBM - forward keeps one log-sum-exp
// per query, backward recomputes the probabilities from it.)
//
// Storage is fp32 (q, k, v, out, gradients; head-major [BH, L, 64]) with operands converted to bf16 / fp16 when a tile is staged
// into LDS or a fragment is built in registers - or, for the LONG side of either product in bf16 mode, bf16 in memory at arbitrary
// (bag, head, row) strides: the n'-sized q / k / v are then read where the projection GEMM left them (token-major [b, n', 3, h, 64])
// and their gradients are written there, the output / its gradient are bf16 [b, n', h 64] as the output projection reads them; only
// the 256-row landmark-side tensors stay fp32 head-major.  Products accumulate in fp32.  This is the "16-bit compute" path of the
// block (BASELINE configs 2 / 4 / 5 name bf16 / fp16); the default fp32 path keeps the exact-fp32 GEMM composition.
// The softmax scale is applied to the fp32 scores (one FMA with the running maximum inside the exponential's argument), not to q.
//
// Orientation (as in deform_attn.hip): S^T = K Q^T puts keys on accumulator rows and queries on lanes, so the softmax
// reduction is in-lane (16 registers + one cross-half exchange) and the probabilities are the B operand of O^T = V^T P^T as
// they stand (converted pairwise to 16 bit; the k order of that fragment is acc_row(8 s + j, half), which the transposed
// LDS reads of V^T reproduce).  head dim 64; 32 queries per wave, 4 waves per workgroup, 32-key tiles, double-buffered
// LDS with the next tile's global loads in flight during the MFMAs.
#include <algorithm>
#include <atomic>     // process-wide measurement switches (set once from the environment or a test hook): plain atomics, no launch state
#include <cstdlib>
#include "smml_common.h"

namespace {

constexpr int AD = 64;        // head dim
constexpr int AQ = 32;        // queries per wave
constexpr int AW = 4;         // waves per workgroup
constexpr int AK = 32;        // keys per tile
constexpr int RLD = AD + 8;   // halves per row of a row-read image (144-byte rows: conflict-free ds_read_b128 of 32 rows)
constexpr int TLD = AD + 32;  // halves per row of a transposed-read image (192-byte rows, see deform_attn.hip)
constexpr float LOG2E_F = 1.4426950408889634f;

// layout of O / dO (floats): element (bh, query, d) at (bh / H) * bs + (bh % H) * hs + query * rs + d.  Head-major [BH, Lq, 64] is
// {H Lq 64, Lq 64, 64}; heads merged [B, Lq, H 64] - what the block's output projection consumes - is {Lq H 64, 64, H 64}.
struct OLayout { long long bs, hs, rs; int H; };
__device__ __forceinline__ size_t obase(const OLayout& L, int bh) { return (size_t)(bh / L.H) * L.bs + (size_t)(bh % L.H) * L.hs; }

template <typename T> struct Pipe;
template <> struct Pipe<__bf16> {
  typedef __bf16 x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 x2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ floatx16 mfma(x8 a, x8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Pipe<_Float16> {
  typedef _Float16 x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 x2 __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ floatx16 mfma(x8 a, x8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// 8 fp32 -> one 16-bit MFMA fragment (round to nearest even)
template <typename T>
__device__ __forceinline__ typename Pipe<T>::x8 pack8(const float (&x)[8]) {
  typedef typename Pipe<T>::x2 x2;
  uint4v w;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2v v = {x[2 * i], x[2 * i + 1]};
    w[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, x2));
  }
  return __builtin_bit_cast(typename Pipe<T>::x8, w);
}
template <typename T>
__device__ __forceinline__ uint2v pack4(const float4 v) {
  typedef typename Pipe<T>::x2 x2;
  const float2v a = {v.x, v.y}, b = {v.z, v.w};
  return (uint2v){__builtin_bit_cast(unsigned, __builtin_convertvector(a, x2)), __builtin_bit_cast(unsigned, __builtin_convertvector(b, x2))};
}
// MFMA fragment of an operand stored k-major ([k][rows]) in LDS: two hardware-transposed reads (see smml_common.h lds_frag_tr)
template <typename T>
__device__ __forceinline__ typename Pipe<T>::x8 frag_tr(const T* p0, const T* p1) {
  typedef short short4v __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) short4v lds_s4;
  const short4v r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)p0);
  const short4v r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)p1);
  typedef short short8v __attribute__((ext_vector_type(8)));
  const short8v r = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
  return __builtin_bit_cast(typename Pipe<T>::x8, r);
}
// this lane's 8 consecutive head-dim values of k-step s (d = 16 s + 8 half + j) of one row as a fragment: converted from fp32
// storage, or as they stand from 16-bit storage
template <typename T>
__device__ __forceinline__ typename Pipe<T>::x8 row_frag(const float* row, int s, int hf) {
  const float4 a = *reinterpret_cast<const float4*>(row + 16 * s + 8 * hf), b = *reinterpret_cast<const float4*>(row + 16 * s + 8 * hf + 4);
  const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  return pack8<T>(x);
}
template <typename T>
__device__ __forceinline__ typename Pipe<T>::x8 row_frag(const T* row, int s, int hf) {
  return *reinterpret_cast<const typename Pipe<T>::x8*>(row + 16 * s + 8 * hf);
}
// 8 consecutive values of a row as floats
__device__ __forceinline__ void load8f(const float* p, float (&x)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
}
template <typename T>
__device__ __forceinline__ void load8f(const T* p, float (&x)[8]) {
  const typename Pipe<T>::x8 v = *reinterpret_cast<const typename Pipe<T>::x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (float)v[i];
}
// 4 consecutive values: load as floats / store from floats
__device__ __forceinline__ float4 load4f(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <typename T>
__device__ __forceinline__ float4 load4f(const T* p) {
  typedef T x4 __attribute__((ext_vector_type(4)));
  const x4 v = *reinterpret_cast<const x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void store4f(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <typename T>
__device__ __forceinline__ void store4f(T* p, const float4 v) { *reinterpret_cast<uint2v*>(p) = pack4<T>(v); }

// Store this lane's 64 values of one output row - the two transposed accumulators o0 (d < 32) and o1 (d >= 32) of O^T[d, row]: lane half hf
// holds d = 32 blk + 8 rg + 4 hf + (0..3) in registers 4 rg .. 4 rg + 3 - as mul * o + add (add: the lane's own pieces of a residual / a running
// sum, zero when absent).  fp32 storage: 16-byte stores, the two halves of a lane pair write adjacent pieces (32 contiguous bytes per row).
// 16-bit storage: a piece is 8 bytes, and 16-byte fragments scattered over 32 rows would make every 32-byte sector a partial write; the two
// halves therefore exchange pieces first (v_permlane32_swap: half 0 keeps piece rg = 2 j and receives half 1's rg = 2 j, half 1 receives
// half 0's rg = 2 j + 1), so that each lane stores 16 contiguous bytes and a lane pair 32.
__device__ __forceinline__ void store_row(float* rowp, const floatx16& o0, const floatx16& o1, float mul, const float4 (&ra)[4],
                                          const float4 (&rb)[4], int hf) {
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int d = 8 * rg + 4 * hf;
    *reinterpret_cast<float4*>(rowp + d) = make_float4(fmaf(o0[4 * rg], mul, ra[rg].x), fmaf(o0[4 * rg + 1], mul, ra[rg].y),
                                                       fmaf(o0[4 * rg + 2], mul, ra[rg].z), fmaf(o0[4 * rg + 3], mul, ra[rg].w));
    *reinterpret_cast<float4*>(rowp + 32 + d) = make_float4(fmaf(o1[4 * rg], mul, rb[rg].x), fmaf(o1[4 * rg + 1], mul, rb[rg].y),
                                                            fmaf(o1[4 * rg + 2], mul, rb[rg].z), fmaf(o1[4 * rg + 3], mul, rb[rg].w));
  }
}
template <typename T>
__device__ __forceinline__ void store_row(T* rowp, const floatx16& o0, const floatx16& o1, float mul, const float4 (&ra)[4],
                                          const float4 (&rb)[4], int hf) {
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const floatx16& o = blk ? o1 : o0;
    uint2v pk[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float4 r = blk ? rb[rg] : ra[rg];
      pk[rg] = pack4<T>(make_float4(fmaf(o[4 * rg], mul, r.x), fmaf(o[4 * rg + 1], mul, r.y), fmaf(o[4 * rg + 2], mul, r.z), fmaf(o[4 * rg + 3], mul, r.w)));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const auto w0 = __builtin_amdgcn_permlane32_swap(pk[2 * j][0], pk[2 * j + 1][0], false, false);
      const auto w1 = __builtin_amdgcn_permlane32_swap(pk[2 * j][1], pk[2 * j + 1][1], false, false);
      *reinterpret_cast<uint4v*>(rowp + 32 * blk + 16 * j + 8 * hf) = (uint4v){w0[0], w1[0], w0[1], w1[1]};
    }
  }
}

// A wave's 32 query rows of 64 sixteen-bit values (128 bytes each) moved in FULL LINES: lane l takes the 16-byte chunk l & 7 of rows
// (l >> 3) + 8 i, i = 0..3 - eight lanes per row, eight whole rows per instruction - where the fragment-shaped access (each lane its own
// row, 16-32 bytes of it per instruction) touches 32 lines per instruction, 128 per tensor and wave.  The rows pass through a per-wave LDS
// image [32][RLD] to change between the two shapes.
template <typename T>
__device__ __forceinline__ void rows_load(const T* base, long long rs, int row0, int nrows, uint4v (&r)[4], int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const uint4v*>(base + (size_t)min(row0 + (lane >> 3) + 8 * i, nrows - 1) * rs + (lane & 7) * 8);
}
template <typename T>
__device__ __forceinline__ void rows_store(T* base, long long rs, int row0, int nrows, const uint4v (&r)[4], int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + (lane >> 3) + 8 * i;
    if (row < nrows) *reinterpret_cast<uint4v*>(base + (size_t)row * rs + (lane & 7) * 8) = r[i];
  }
}
template <typename T>
__device__ __forceinline__ void rows_to_lds(T* img, const uint4v (&r)[4], int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4v*>(&img[((lane >> 3) + 8 * i) * RLD + (lane & 7) * 8]) = r[i];
}
template <typename T>
__device__ __forceinline__ void rows_from_lds(const T* img, uint4v (&r)[4], int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const uint4v*>(&img[((lane >> 3) + 8 * i) * RLD + (lane & 7) * 8]);
}
template <typename A, typename B> struct same_type { static constexpr bool value = false; };
template <typename A> struct same_type<A, A> { static constexpr bool value = true; };

// One 32-row x 64 tile of a staged operand held by 256 threads between its global load and its LDS store.
//   fp32 storage: rows (tid >> 4) and (tid >> 4) + 16, 4 consecutive d each (two float4), converted when stored (8-byte LDS stores)
//   16-bit storage: row tid >> 3, 8 consecutive d (one 16-byte load, one 16-byte LDS store per image)
template <typename T, typename TS> struct TileRegs;
template <typename T> struct TileRegs<T, float> {
  float4 r[2];
  __device__ __forceinline__ void load(const float* base, long long rs, int row0, int nrows, int tid) {     // rows >= nrows read as zero
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = row0 + (tid >> 4) + 16 * i;
      r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nrows) r[i] = *reinterpret_cast<const float4*>(base + (size_t)row * rs + (tid & 15) * 4);
    }
  }
  __device__ __forceinline__ void store(T* img, int ld, int tid) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<uint2v*>(&img[((tid >> 4) + 16 * i) * ld + (tid & 15) * 4]) = pack4<T>(r[i]);
  }
  __device__ __forceinline__ void store2(T* img0, int ld0, T* img1, int ld1, int tid) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint2v pk = pack4<T>(r[i]);
      *reinterpret_cast<uint2v*>(&img0[((tid >> 4) + 16 * i) * ld0 + (tid & 15) * 4]) = pk;
      *reinterpret_cast<uint2v*>(&img1[((tid >> 4) + 16 * i) * ld1 + (tid & 15) * 4]) = pk;
    }
  }
};
template <typename T> struct TileRegs<T, T> {
  uint4v r;
  __device__ __forceinline__ void load(const T* base, long long rs, int row0, int nrows, int tid) {
    const int row = row0 + (tid >> 3);
    r = (uint4v){0u, 0u, 0u, 0u};
    if (row < nrows) r = *reinterpret_cast<const uint4v*>(base + (size_t)row * rs + (tid & 7) * 8);
  }
  __device__ __forceinline__ void store(T* img, int ld, int tid) const { *reinterpret_cast<uint4v*>(&img[(tid >> 3) * ld + (tid & 7) * 8]) = r; }
  __device__ __forceinline__ void store2(T* img0, int ld0, T* img1, int ld1, int tid) const {
    *reinterpret_cast<uint4v*>(&img0[(tid >> 3) * ld0 + (tid & 7) * 8]) = r;
    *reinterpret_cast<uint4v*>(&img1[(tid >> 3) * ld1 + (tid & 7) * 8]) = r;
  }
};

// ------------------------------------------------------------------------------------------------
// forward, queries-long form with 16-bit query-side storage, TWO 32-query blocks per wave: the [n', m] side of the
// Nystrom block has only m / 32 = 8 key tiles per wave, so what a wave pays once - its q / residual / output rows, the launch ramp - and what it
// pays per tile and wave - the K / V fragment reads from LDS, the barrier - weighs as much as the MFMAs.  With two query blocks a wave issues
// the same K and V fragments into two independent accumulator sets: LDS fragment reads, barriers and the staging of K / V per MFMA halve.
//   Q, O, RES: T at layouts ql / ol;  K, V: fp32 head-major [BH, Lk, 64] (the landmark side);  grid (ceil(Lq / 256), BH)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void attn16_fwd_q2_kernel(const T* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                               T* O, const T* RES, float* __restrict__ LSE2, int Lq, int Lk, float qscale,
                                                               OLayout ql, OLayout kl, OLayout ol) {
  typedef typename Pipe<T>::x8 x8;
  constexpr int NB = 2;                                  // query blocks per wave
  __shared__ __attribute__((aligned(16))) T Kr[2][AK * RLD];
  __shared__ __attribute__((aligned(16))) T Vt[2][AK * TLD];
  __shared__ __attribute__((aligned(16))) T Wr[AW * NB * AQ * RLD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int bh = blockIdx.y;
  const int q0 = blockIdx.x * (AQ * AW * NB) + wave * (AQ * NB);
  T* wr = Wr + wave * (NB * AQ * RLD);
  const float* Kb = K + obase(kl, bh);
  const float* Vb = V + obase(kl, bh);
  x8 qf[NB][4];
  {
    uint4v qraw[NB][4];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) rows_load(Q + obase(ql, bh), ql.rs, q0 + AQ * nb, Lq, qraw[nb], lane);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) rows_to_lds(wr + nb * AQ * RLD, qraw[nb], lane);
    wave_lds_fence();
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[nb][s] = *reinterpret_cast<const x8*>(&wr[(nb * AQ + c) * RLD + 16 * s + 8 * hf]);
    wave_lds_fence();
  }
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  TileRegs<T, float> kreg, vreg;
  kreg.load(Kb, kl.rs, 0, Lk, tid); vreg.load(Vb, kl.rs, 0, Lk, tid);
  floatx16 o0[NB], o1[NB];
  float m_run[NB], l_run[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) { o0[nb] = floatx16{0}; o1[nb] = floatx16{0}; m_run[nb] = -INFINITY; l_run[nb] = 0.f; }
  const int ntiles = (Lk + AK - 1) / AK;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int j0 = kt * AK, buf = kt & 1;
    const int nk = min(AK, Lk - j0);
    kreg.store(Kr[buf], RLD, tid); vreg.store(Vt[buf], TLD, tid);
    __syncthreads();        // buffer (kt & 1) was last read in iteration kt - 2: one barrier per tile is enough
    if (kt + 1 < ntiles) { kreg.load(Kb, kl.rs, j0 + AK, Lk, tid); vreg.load(Vb, kl.rs, j0 + AK, Lk, tid); }
    floatx16 s[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) s[nb] = floatx16{0};
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const x8 kf = *reinterpret_cast<const x8*>(&Kr[buf][c * RLD + 16 * st + 8 * hf]);          // one fragment read, two products
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) s[nb] = Pipe<T>::mfma(kf, qf[nb][st], s[nb]);
    }
    constexpr float LAZY = 8.f;                          // lazy reference point (attn16_fwd_kernel)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (nk < AK) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (acc_row(r, hf) >= nk) s[nb][r] = -INFINITY;
      }
      float tmax = s[nb][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[nb][r]);
      tmax = xhalf_max(tmax) * qscale;
      if (__builtin_amdgcn_ballot_w64(tmax > m_run[nb] + LAZY)) {      // wave-uniform
        const float m_new = fmaxf(m_run[nb], tmax);
        const float alpha = __builtin_amdgcn_exp2f(m_run[nb] - m_new);
        l_run[nb] *= alpha;
        m_run[nb] = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[nb][r] *= alpha; o1[nb][r] *= alpha; }
      }
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[nb][r] = __builtin_amdgcn_exp2f(fmaf(s[nb][r], qscale, -m_run[nb])); psum += s[nb][r]; }
      l_run[nb] += psum;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      x8 pb[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float p8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p8[j] = s[nb][8 * kb + j];
        pb[nb] = pack8<T>(p8);
      }
      const int ro = (16 * kb + 4 * hf + trq) * TLD + trc;
      const x8 v0 = frag_tr<T>(&Vt[buf][ro], &Vt[buf][ro + 8 * TLD]);
      const x8 v1 = frag_tr<T>(&Vt[buf][ro + 32], &Vt[buf][ro + 32 + 8 * TLD]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) { o0[nb] = Pipe<T>::mfma(v0, pb[nb], o0[nb]); o1[nb] = Pipe<T>::mfma(v1, pb[nb], o1[nb]); }
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int qb = q0 + AQ * nb;
    if (qb >= Lq) continue;                              // wave-uniform: nothing of this block is inside the bag
    T* wq = wr + nb * AQ * RLD;
    const float l = xhalf_sum(l_run[nb]);
    const float inv = 1.f / l;
    float4 ra[4], rb[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) { ra[rg] = make_float4(0.f, 0.f, 0.f, 0.f); rb[rg] = ra[rg]; }
    if (RES) {                                          // the residual rows: full lines -> this lane's pieces of its own row
      uint4v rraw[4];
      rows_load(RES + obase(ol, bh), ol.rs, qb, Lq, rraw, lane);
      rows_to_lds(wq, rraw, lane);
      wave_lds_fence();
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) { ra[rg] = load4f(&wq[c * RLD + 8 * rg + 4 * hf]); rb[rg] = load4f(&wq[c * RLD + 32 + 8 * rg + 4 * hf]); }
      wave_lds_fence();
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      store4f(&wq[c * RLD + 8 * rg + 4 * hf], make_float4(fmaf(o0[nb][4 * rg], inv, ra[rg].x), fmaf(o0[nb][4 * rg + 1], inv, ra[rg].y),
                                                          fmaf(o0[nb][4 * rg + 2], inv, ra[rg].z), fmaf(o0[nb][4 * rg + 3], inv, ra[rg].w)));
      store4f(&wq[c * RLD + 32 + 8 * rg + 4 * hf], make_float4(fmaf(o1[nb][4 * rg], inv, rb[rg].x), fmaf(o1[nb][4 * rg + 1], inv, rb[rg].y),
                                                               fmaf(o1[nb][4 * rg + 2], inv, rb[rg].z), fmaf(o1[nb][4 * rg + 3], inv, rb[rg].w)));
    }
    wave_lds_fence();
    uint4v oraw[4];
    rows_from_lds(wq, oraw, lane);
    rows_store(O + obase(ol, bh), ol.rs, qb, Lq, oraw, lane);
    if (qb + c < Lq && hf == 0) LSE2[(size_t)bh * Lq + qb + c] = m_run[nb] + __builtin_amdgcn_logf(l);     // v_log_f32 is log2
  }
}

// ------------------------------------------------------------------------------------------------
// forward: O = softmax(scale Q K^T) V (+ RES), LSE2 = log2-sum-exp2 of the scaled scores per query (base 2: what the backward needs)
//   Q, O, RES: storage TQ at layouts ql / ol;  K, V: storage TK at layout kl;  LSE2 [BH, Lq]     grid (ceil(Lq / 128), BH, key chunks)
//   RES may be O itself (the fp32 path accumulates into an output that already holds the residual)
// ------------------------------------------------------------------------------------------------
template <typename T, typename TQ, typename TK>
__global__ __launch_bounds__(256, 2) void attn16_fwd_kernel(const TQ* __restrict__ Q, const TK* __restrict__ K, const TK* __restrict__ V,
                                                            TQ* O, const TQ* RES, float* __restrict__ LSE2, int Lq, int Lk, float qscale,
                                                            OLayout ql, OLayout kl, OLayout ol, int chunk, float* __restrict__ Opart,
                                                            float* __restrict__ Lpart) {
  typedef typename Pipe<T>::x8 x8;
  __shared__ __attribute__((aligned(16))) T Kr[2][AK * RLD];
  __shared__ __attribute__((aligned(16))) T Vt[2][AK * TLD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int bh = blockIdx.y;
  const int q0 = blockIdx.x * (AQ * AW) + wave * AQ;
  const bool qvalid = (q0 + c) < Lq;
  const int qi = qvalid ? (q0 + c) : (Lq - 1);
  // key range of this workgroup: the whole row, or one of gridDim.z chunks (few queries, many keys - the [m, n'] side of the
  // Nystrom block has only m / 32 waves per head: the chunks' partial results are merged by attn16_merge_kernel)
  const int kbeg = blockIdx.z * chunk, kend = min(Lk, kbeg + chunk);
  const TK* Kb = K + obase(kl, bh) + (size_t)kbeg * kl.rs;
  const TK* Vb = V + obase(kl, bh) + (size_t)kbeg * kl.rs;
  Lk = kend - kbeg;                                   // from here on: keys of this chunk only

  constexpr bool QROWS = same_type<TQ, T>::value;        // 16-bit query-side storage: rows move in full lines through a per-wave image
  __shared__ __attribute__((aligned(16))) T Wr[QROWS ? AW * AQ * RLD : 8];
  T* wr = Wr + (QROWS ? wave * AQ * RLD : 0);
  x8 qf[4];
  const size_t oo = obase(ol, bh) + (size_t)qi * ol.rs;
  float4 ra[4], rb[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) { ra[rg] = make_float4(0.f, 0.f, 0.f, 0.f); rb[rg] = ra[rg]; }
  uint4v rraw[4];                                       // the residual rows in line shape (QROWS), requested before the key loop
  if constexpr (QROWS) {
    uint4v qraw[4];
    rows_load(reinterpret_cast<const T*>(Q) + obase(ql, bh), ql.rs, q0, Lq, qraw, lane);
    if (RES) rows_load(reinterpret_cast<const T*>(RES) + obase(ol, bh), ol.rs, q0, Lq, rraw, lane);
    rows_to_lds(wr, qraw, lane);
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const x8*>(&wr[c * RLD + 16 * s + 8 * hf]);
    wave_lds_fence();
  } else {
    const TQ* qrow = Q + obase(ql, bh) + (size_t)qi * ql.rs;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = row_frag<T>(qrow, s, hf);
    // the residual row is requested now (one batch of loads, consumed after the key loop): at the end it would be an exposed round trip
    if (RES && !Opart) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) { ra[rg] = load4f(RES + oo + 8 * rg + 4 * hf); rb[rg] = load4f(RES + oo + 32 + 8 * rg + 4 * hf); }
    }
  }
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  TileRegs<T, TK> kreg, vreg;
  kreg.load(Kb, kl.rs, 0, Lk, tid); vreg.load(Vb, kl.rs, 0, Lk, tid);
  floatx16 o0 = {0}, o1 = {0};
  float m_run = -INFINITY, l_run = 0.f;
  const int ntiles = (Lk + AK - 1) / AK;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int j0 = kt * AK, buf = kt & 1;
    const int nk = min(AK, Lk - j0);
    kreg.store(Kr[buf], RLD, tid); vreg.store(Vt[buf], TLD, tid);
    __syncthreads();        // buffer (kt & 1) was last read in iteration kt - 2: one barrier per tile is enough
    if (kt + 1 < ntiles) { kreg.load(Kb, kl.rs, j0 + AK, Lk, tid); vreg.load(Vb, kl.rs, j0 + AK, Lk, tid); }
    // S^T[key, query] = K Q^T (unscaled)
    floatx16 s = {0};
#pragma unroll
    for (int st = 0; st < 4; ++st) s = Pipe<T>::mfma(*reinterpret_cast<const x8*>(&Kr[buf][c * RLD + 16 * st + 8 * hf]), qf[st], s);
    if (nk < AK) {                                    // ragged last tile only (wave-uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (acc_row(r, hf) >= nk) s[r] = -INFINITY;
    }
    float tmax = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    tmax = xhalf_max(tmax) * qscale;
    // Lazy reference point: the running reference m_run only has to keep the exponentials in range, it need not be the maximum.  It
    // is moved (and the 32 output registers rescaled) only when some query's tile maximum exceeds it by more than 2^LAZY - after the
    // first tile that is rare, so most tiles skip the rescale; probabilities then reach at most 2^LAZY, far inside bf16 / fp16 / fp32
    // range (fp16: 2^8 x 256 keys of row sum is accumulated in fp32), and the log-sum-exp m_run + log2(l_run) is unaffected.
    constexpr float LAZY = 8.f;
    if (__builtin_amdgcn_ballot_w64(tmax > m_run + LAZY)) {      // wave-uniform
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], qscale, -m_run)); psum += s[r]; }
    l_run += psum;
    // O^T[d, query] += V^T P^T: the probabilities of k-step kb are accumulator registers 8 kb .. 8 kb + 7
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float p8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) p8[j] = s[8 * kb + j];
      const x8 pb = pack8<T>(p8);
      const int ro = (16 * kb + 4 * hf + trq) * TLD + trc;
      o0 = Pipe<T>::mfma(frag_tr<T>(&Vt[buf][ro], &Vt[buf][ro + 8 * TLD]), pb, o0);
      o1 = Pipe<T>::mfma(frag_tr<T>(&Vt[buf][ro + 32], &Vt[buf][ro + 32 + 8 * TLD]), pb, o1);
    }
  }
  l_run = xhalf_sum(l_run);
  const float inv = 1.f / l_run;
  if (Opart) {                                        // partial result of this chunk: normalised rows + their log-sum-exp
    if (qvalid) {
      const size_t row = ((size_t)blockIdx.z * gridDim.y + bh) * Lq + qi;
      float* op = Opart + row * AD;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 8 * rg + 4 * hf;
        *reinterpret_cast<float4*>(op + d) = make_float4(o0[4 * rg] * inv, o0[4 * rg + 1] * inv, o0[4 * rg + 2] * inv, o0[4 * rg + 3] * inv);
        *reinterpret_cast<float4*>(op + 32 + d) = make_float4(o1[4 * rg] * inv, o1[4 * rg + 1] * inv, o1[4 * rg + 2] * inv, o1[4 * rg + 3] * inv);
      }
      if (hf == 0) Lpart[row] = m_run + __builtin_amdgcn_logf(l_run);
    }
    return;
  }
  if constexpr (QROWS) {
    T* wq = wr;
    if (RES) {                                          // line shape -> this lane's pieces of its own row
      rows_to_lds(wq, rraw, lane);
      wave_lds_fence();
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) { ra[rg] = load4f(&wq[c * RLD + 8 * rg + 4 * hf]); rb[rg] = load4f(&wq[c * RLD + 32 + 8 * rg + 4 * hf]); }
      wave_lds_fence();
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      store4f(&wq[c * RLD + 8 * rg + 4 * hf], make_float4(fmaf(o0[4 * rg], inv, ra[rg].x), fmaf(o0[4 * rg + 1], inv, ra[rg].y),
                                                          fmaf(o0[4 * rg + 2], inv, ra[rg].z), fmaf(o0[4 * rg + 3], inv, ra[rg].w)));
      store4f(&wq[c * RLD + 32 + 8 * rg + 4 * hf], make_float4(fmaf(o1[4 * rg], inv, rb[rg].x), fmaf(o1[4 * rg + 1], inv, rb[rg].y),
                                                               fmaf(o1[4 * rg + 2], inv, rb[rg].z), fmaf(o1[4 * rg + 3], inv, rb[rg].w)));
    }
    wave_lds_fence();
    uint4v oraw[4];
    rows_from_lds(wq, oraw, lane);
    rows_store(reinterpret_cast<T*>(O) + obase(ol, bh), ol.rs, q0, Lq, oraw, lane);
    if (qvalid && hf == 0) LSE2[(size_t)bh * Lq + qi] = m_run + __builtin_amdgcn_logf(l_run);     // v_log_f32 is log2
  } else {
    if (qvalid) {
      store_row(O + oo, o0, o1, inv, ra, rb, hf);
      if (hf == 0) LSE2[(size_t)bh * Lq + qi] = m_run + __builtin_amdgcn_logf(l_run);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward for FEW KEYS (Lk <= 256: the [n', m] side of the Nystrom block, m = 256 landmarks): two-pass softmax, no online rescale.
// All keys' K and V are staged ONCE per workgroup (bf16 / fp16 images, 86 KB: one 8-wave workgroup per CU = two waves per SIMD);
// a wave keeps the whole score row block S^T [256 keys x 32 queries] in registers (8 accumulators), takes the exact row maximum,
// exponentiates once and feeds P^T straight into O^T = V^T P^T.  Against the online form this drops, per 32-key tile, the running
// maximum bookkeeping, the exp of the correction factor and the rescale of the 32 output registers: ~9 vector instructions per
// MFMA instead of ~24 (profiles/r02_nystrom16_pmc.txt).       grid (ceil(Lq / 256 / blocks per workgroup), BH), 512 threads, dynamic LDS: the staging is paid once per workgroup.
// ------------------------------------------------------------------------------------------------
constexpr int SK_MAX = 256;                       // keys
constexpr int SK_WAVES = 8;                       // waves per workgroup (32 queries each)
constexpr size_t SK_LDS = (size_t)SK_MAX * (RLD + TLD) * 2;   // bytes: K row image + V transposed-read image

template <typename T>
__global__ __launch_bounds__(512, 2) void attn16_fwd_fewkeys_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                    const float* __restrict__ V, float* __restrict__ O,
                                                                    float* __restrict__ LSE2, int Lq, int Lk, float qscale,
                                                                    OLayout ol, int accumulate, int blocks_per_wg) {
  typedef typename Pipe<T>::x8 x8;
  extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
  T* Kr = reinterpret_cast<T*>(sk_smem);                       // [256][RLD]
  T* Vt = Kr + SK_MAX * RLD;                                   // [256][TLD]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int bh = blockIdx.y;
  const int nblk = (Lq + AQ * SK_WAVES - 1) / (AQ * SK_WAVES);
  const int blk0 = blockIdx.x * blocks_per_wg, blk1 = min(nblk, blk0 + blocks_per_wg);    // this workgroup's 256-query blocks
  const float* Kb = K + (size_t)bh * Lk * AD;
  const float* Vb = V + (size_t)bh * Lk * AD;
  const float* Qb = Q + (size_t)bh * Lq * AD;
  // the first block's query rows are requested before the keys (they are consumed right after the staging barrier)
  float4 qraw[8];
  auto load_q = [&](int blk) {
    const int qi = min(blk * (AQ * SK_WAVES) + wave * AQ + c, Lq - 1);
    const float* qrow = Qb + (size_t)qi * AD + 8 * hf;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      qraw[2 * st] = *reinterpret_cast<const float4*>(qrow + 16 * st);
      qraw[2 * st + 1] = *reinterpret_cast<const float4*>(qrow + 16 * st + 4);
    }
  };
  load_q(blk0);
  // stage every key once per workgroup: thread -> key (tid >> 4) + 32 i, 4 consecutive d (rows past Lk are zero); the staging
  // is paid once for blocks_per_wg query blocks
  {
    // all sixteen loads are issued before the first use (clamped addresses, rows past Lk zeroed afterwards: a branch around each
    // load would make hipcc wait for every one of them in turn)
    const int skey = tid >> 4, sd4 = (tid & 15) * 4;
    float4 kreg[SK_MAX / 32], vreg[SK_MAX / 32];
#pragma unroll
    for (int i = 0; i < SK_MAX / 32; ++i) {
      const int key = min(skey + 32 * i, Lk - 1);
      kreg[i] = *reinterpret_cast<const float4*>(Kb + (size_t)key * AD + sd4);
      vreg[i] = *reinterpret_cast<const float4*>(Vb + (size_t)key * AD + sd4);
    }
#pragma unroll
    for (int i = 0; i < SK_MAX / 32; ++i) {
      const int key = skey + 32 * i;
      const float z = (key < Lk) ? 1.f : 0.f;
      *reinterpret_cast<uint2v*>(&Kr[key * RLD + sd4]) = pack4<T>(make_float4(kreg[i].x * z, kreg[i].y * z, kreg[i].z * z, kreg[i].w * z));
      *reinterpret_cast<uint2v*>(&Vt[key * TLD + sd4]) = pack4<T>(make_float4(vreg[i].x * z, vreg[i].y * z, vreg[i].z * z, vreg[i].w * z));
    }
  }
  __syncthreads();
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const int nt = (Lk + AK - 1) / AK;                  // <= 8 tiles of 32 keys
  for (int blk = blk0; blk < blk1; ++blk) {
    const int q0 = blk * (AQ * SK_WAVES) + wave * AQ;
    const bool qvalid = (q0 + c) < Lq;
    const int qi = qvalid ? (q0 + c) : (Lq - 1);
    x8 qf[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const float4 a = qraw[2 * st], b = qraw[2 * st + 1];
      const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      qf[st] = pack8<T>(x);
    }
    if (blk + 1 < blk1) load_q(blk + 1);              // next block's rows arrive during this block's MFMAs
    // pass 1: S^T = K Q^T (unscaled) for every tile, row maximum
    floatx16 s[SK_MAX / AK];
    float m = -INFINITY;
#pragma unroll
    for (int t = 0; t < SK_MAX / AK; ++t) {
      s[t] = floatx16{0};
      if (t < nt) {                                     // uniform
#pragma unroll
        for (int st = 0; st < 4; ++st)
          s[t] = Pipe<T>::mfma(*reinterpret_cast<const x8*>(&Kr[(32 * t + c) * RLD + 16 * st + 8 * hf]), qf[st], s[t]);
        if (32 * t + 32 > Lk) {                         // ragged last tile (uniform)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (32 * t + acc_row(r, hf) >= Lk) s[t][r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, s[t][r]);
      }
    }
    m = xhalf_max(m) * qscale;
    // pass 2: P^T = 2^(qscale S^T - m), row sum, O^T += V^T P^T
    floatx16 o0 = {0}, o1 = {0};
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < SK_MAX / AK; ++t) {
      if (t < nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], qscale, -m)); l += s[t][r]; }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          float p8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) p8[j] = s[t][8 * kb + j];
          const x8 pb = pack8<T>(p8);
          const int ro = (32 * t + 16 * kb + 4 * hf + trq) * TLD + trc;
          o0 = Pipe<T>::mfma(frag_tr<T>(&Vt[ro], &Vt[ro + 8 * TLD]), pb, o0);
          o1 = Pipe<T>::mfma(frag_tr<T>(&Vt[ro + 32], &Vt[ro + 32 + 8 * TLD]), pb, o1);
        }
      }
    }
    l = xhalf_sum(l);
    const float inv = 1.f / l;
    {
      // the residual (out already holds the depthwise convolution of v) is fetched as ONE batch of loads: with the load inside the
      // store loop hipcc waits for each (four dependent L2 round trips per block)
      float* op = O + obase(ol, bh) + (size_t)qi * ol.rs;
      float4 ra[4], rb[4];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) { ra[rg] = make_float4(0.f, 0.f, 0.f, 0.f); rb[rg] = ra[rg]; }
      if (accumulate) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          ra[rg] = *reinterpret_cast<const float4*>(op + 8 * rg + 4 * hf);
          rb[rg] = *reinterpret_cast<const float4*>(op + 32 + 8 * rg + 4 * hf);
        }
      }
      if (qvalid) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int d = 8 * rg + 4 * hf;
          *reinterpret_cast<float4*>(op + d) = make_float4(fmaf(o0[4 * rg], inv, ra[rg].x), fmaf(o0[4 * rg + 1], inv, ra[rg].y),
                                                           fmaf(o0[4 * rg + 2], inv, ra[rg].z), fmaf(o0[4 * rg + 3], inv, ra[rg].w));
          *reinterpret_cast<float4*>(op + 32 + d) = make_float4(fmaf(o1[4 * rg], inv, rb[rg].x), fmaf(o1[4 * rg + 1], inv, rb[rg].y),
                                                                fmaf(o1[4 * rg + 2], inv, rb[rg].z), fmaf(o1[4 * rg + 3], inv, rb[rg].w));
        }
        if (hf == 0) LSE2[(size_t)bh * Lq + qi] = m + __builtin_amdgcn_logf(l);
      }
    }
  }
}

// merges the chunks of a key-split forward: lse = log2 sum_s 2^lse_s, O = sum_s 2^(lse_s - lse) O_s   (thread = 4 head-dim values)
__global__ void attn16_merge_kernel(const float* __restrict__ Opart, const float* __restrict__ Lpart, float* __restrict__ O,
                                    float* __restrict__ LSE2, int BH, int Lq, int nsplit, OLayout ol, int accumulate) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t rows = (size_t)BH * Lq;
  if (i >= rows * (AD / 4)) return;
  const size_t row = i / (AD / 4);
  const int d4 = (int)(i % (AD / 4)) * 4;
  float m = -INFINITY;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, Lpart[(size_t)s * rows + row]);
  float den = 0.f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s = 0; s < nsplit; ++s) {
    const float w = __builtin_amdgcn_exp2f(Lpart[(size_t)s * rows + row] - m);
    const float4 o = *reinterpret_cast<const float4*>(Opart + ((size_t)s * rows + row) * AD + d4);
    den += w;
    acc.x = fmaf(w, o.x, acc.x); acc.y = fmaf(w, o.y, acc.y); acc.z = fmaf(w, o.z, acc.z); acc.w = fmaf(w, o.w, acc.w);
  }
  const float inv = 1.f / den;
  const int bh = (int)(row / Lq), qi = (int)(row % Lq);
  float* op = O + obase(ol, bh) + (size_t)qi * ol.rs + d4;
  float4 r = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  if (accumulate) { const float4 t = *reinterpret_cast<const float4*>(op); r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
  *reinterpret_cast<float4*>(op) = r;
  if (d4 == 0) LSE2[row] = m + __builtin_amdgcn_logf(den);
}

// ------------------------------------------------------------------------------------------------
// backward pass 1 (query owners): P^T recomputed from LSE2, dP^T = V dO^T, dS^T = P^T (dP^T - delta), dQ = scale dS K;
// also writes delta = rowsum(dO . (O - R)) [BH, Lq] for pass 2.          grid (ceil(Lq / 128), BH, key chunks)
//   Q, O, dO, R, dQ: storage TQ (layouts ql, ol, ol, ol, dql);  K, V: storage TK (layout kl);  key-split launches (fp32 storage only)
//   write their partial dQ to fp32 slabs [chunk][BH, Lq, 64] instead
// ------------------------------------------------------------------------------------------------
template <typename T, typename TQ, typename TK>
__global__ __launch_bounds__(256, 2) void attn16_bwd_dq_kernel(const TQ* __restrict__ Q, const TK* __restrict__ K, const TK* __restrict__ V,
                                                               const TQ* __restrict__ O, const TQ* __restrict__ dO,
                                                               const float* __restrict__ LSE2, TQ* __restrict__ dQ, float* __restrict__ dQslab,
                                                               float* __restrict__ DELTA, int Lq, int Lk, float qscale, float scale,
                                                               OLayout ql, OLayout kl, OLayout ol, OLayout dql, const TQ* __restrict__ R,
                                                               int chunk) {
  typedef typename Pipe<T>::x8 x8;
  __shared__ __attribute__((aligned(16))) T Kr[2][AK * RLD];
  __shared__ __attribute__((aligned(16))) T Kt[2][AK * TLD];
  __shared__ __attribute__((aligned(16))) T Vr[2][AK * RLD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int bh = blockIdx.y;
  const int q0 = blockIdx.x * (AQ * AW) + wave * AQ;
  const bool qvalid = (q0 + c) < Lq;
  const int qi = qvalid ? (q0 + c) : (Lq - 1);
  const int kbeg = blockIdx.z * chunk, kend = min(Lk, kbeg + chunk);
  const TK* Kb = K + obase(kl, bh) + (size_t)kbeg * kl.rs;
  const TK* Vb = V + obase(kl, bh) + (size_t)kbeg * kl.rs;
  Lk = kend - kbeg;
  constexpr bool QROWS = same_type<TQ, T>::value;        // 16-bit query-side storage: rows move in full lines through a per-wave image
  __shared__ __attribute__((aligned(16))) T Wr[QROWS ? AW * AQ * RLD : 8];
  __shared__ float Wd[QROWS ? AW * AQ : 1];
  T* wr = Wr + (QROWS ? wave * AQ * RLD : 0);
  x8 qf[4], dof[4];
  float delta = 0.f;
  if constexpr (QROWS) {
    uint4v qraw[4], graw[4], oraw[4], rraw[4];
    rows_load(reinterpret_cast<const T*>(Q) + obase(ql, bh), ql.rs, q0, Lq, qraw, lane);
    rows_load(reinterpret_cast<const T*>(dO) + obase(ol, bh), ol.rs, q0, Lq, graw, lane);
    rows_load(reinterpret_cast<const T*>(O) + obase(ol, bh), ol.rs, q0, Lq, oraw, lane);
    if (R) rows_load(reinterpret_cast<const T*>(R) + obase(ol, bh), ol.rs, q0, Lq, rraw, lane);
    // delta = rowsum(dO . (O - R)) in line shape: eight lanes hold one row, three exchanges add them up
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const x8 gv = __builtin_bit_cast(x8, graw[i]), ov = __builtin_bit_cast(x8, oraw[i]);
      float d = 0.f;
      if (R) {
        const x8 rv = __builtin_bit_cast(x8, rraw[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) d = fmaf((float)gv[e], (float)ov[e] - (float)rv[e], d);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) d = fmaf((float)gv[e], (float)ov[e], d);
      }
      d += __shfl_xor(d, 1); d += __shfl_xor(d, 2); d += __shfl_xor(d, 4);
      if ((lane & 7) == 0) Wd[wave * AQ + (lane >> 3) + 8 * i] = d;
    }
    rows_to_lds(wr, qraw, lane);
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const x8*>(&wr[c * RLD + 16 * s + 8 * hf]);
    delta = Wd[wave * AQ + c];
    wave_lds_fence();
    rows_to_lds(wr, graw, lane);
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 4; ++s) dof[s] = *reinterpret_cast<const x8*>(&wr[c * RLD + 16 * s + 8 * hf]);
    wave_lds_fence();
  } else {
    const TQ* qrow = Q + obase(ql, bh) + (size_t)qi * ql.rs;
    const size_t oo = obase(ol, bh) + (size_t)qi * ol.rs;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = row_frag<T>(qrow, s, hf);
      dof[s] = row_frag<T>(dO + oo, s, hf);
      float t[8], u[8];
      load8f(dO + oo + 16 * s + 8 * hf, t);
      load8f(O + oo + 16 * s + 8 * hf, u);
      if (R) {            // the forward added a residual to O: delta needs the attention output alone
        float rr[8];
        load8f(R + oo + 16 * s + 8 * hf, rr);
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] -= rr[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) delta = fmaf(t[i], u[i], delta);
    }
    delta = xhalf_sum(delta);
  }
  const float lse2 = LSE2[(size_t)bh * Lq + qi];
  if (qvalid && hf == 0 && blockIdx.z == 0) DELTA[(size_t)bh * Lq + qi] = delta;
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  TileRegs<T, TK> kreg, vreg;
  kreg.load(Kb, kl.rs, 0, Lk, tid); vreg.load(Vb, kl.rs, 0, Lk, tid);
  floatx16 dq0 = {0}, dq1 = {0};
  const int ntiles = (Lk + AK - 1) / AK;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int j0 = kt * AK, buf = kt & 1;
    const int nk = min(AK, Lk - j0);
    kreg.store2(Kr[buf], RLD, Kt[buf], TLD, tid);
    vreg.store(Vr[buf], RLD, tid);
    __syncthreads();
    if (kt + 1 < ntiles) { kreg.load(Kb, kl.rs, j0 + AK, Lk, tid); vreg.load(Vb, kl.rs, j0 + AK, Lk, tid); }
    floatx16 s = {0}, dp = {0};
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      s = Pipe<T>::mfma(*reinterpret_cast<const x8*>(&Kr[buf][c * RLD + 16 * st + 8 * hf]), qf[st], s);
      dp = Pipe<T>::mfma(*reinterpret_cast<const x8*>(&Vr[buf][c * RLD + 16 * st + 8 * hf]), dof[st], dp);
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ds[r] = __builtin_amdgcn_exp2f(fmaf(s[r], qscale, -lse2)) * (dp[r] - delta);
    if (nk < AK) {                                    // ragged last tile only (wave-uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (acc_row(r, hf) >= nk) ds[r] = 0.f;
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float d8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) d8[j] = ds[8 * kb + j];
      const x8 db = pack8<T>(d8);
      const int ro = (16 * kb + 4 * hf + trq) * TLD + trc;
      dq0 = Pipe<T>::mfma(frag_tr<T>(&Kt[buf][ro], &Kt[buf][ro + 8 * TLD]), db, dq0);
      dq1 = Pipe<T>::mfma(frag_tr<T>(&Kt[buf][ro + 32], &Kt[buf][ro + 32 + 8 * TLD]), db, dq1);
    }
  }
  if (qvalid) {
    if (dQslab) {
      float* qp = dQslab + (((size_t)blockIdx.z * gridDim.y + bh) * Lq + qi) * AD;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d = 8 * rg + 4 * hf;
        *reinterpret_cast<float4*>(qp + d) = make_float4(dq0[4 * rg] * scale, dq0[4 * rg + 1] * scale, dq0[4 * rg + 2] * scale, dq0[4 * rg + 3] * scale);
        *reinterpret_cast<float4*>(qp + 32 + d) = make_float4(dq1[4 * rg] * scale, dq1[4 * rg + 1] * scale, dq1[4 * rg + 2] * scale, dq1[4 * rg + 3] * scale);
      }
    } else {
      if constexpr (!QROWS) {
        float4 z4[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) z4[rg] = make_float4(0.f, 0.f, 0.f, 0.f);
        store_row(dQ + obase(dql, bh) + (size_t)qi * dql.rs, dq0, dq1, scale, z4, z4, hf);
      }
    }
  }
  if constexpr (QROWS) {
    if (!dQslab) {                                       // every lane takes part: pieces -> image -> full lines
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        store4f(&wr[c * RLD + 8 * rg + 4 * hf], make_float4(dq0[4 * rg] * scale, dq0[4 * rg + 1] * scale, dq0[4 * rg + 2] * scale, dq0[4 * rg + 3] * scale));
        store4f(&wr[c * RLD + 32 + 8 * rg + 4 * hf], make_float4(dq1[4 * rg] * scale, dq1[4 * rg + 1] * scale, dq1[4 * rg + 2] * scale, dq1[4 * rg + 3] * scale));
      }
      wave_lds_fence();
      uint4v oraw[4];
      rows_from_lds(wr, oraw, lane);
      rows_store(reinterpret_cast<T*>(dQ) + obase(dql, bh), dql.rs, q0, Lq, oraw, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward pass 2 (key owners): a wave owns 32 keys (key on the lane axis) and sweeps one slice of the query tiles:
//   S = Q K^T, P = exp2(qscale S - LSE2[q]), dP = dO V^T, dS = P (dP - delta[q]),  dV^T += dO^T P, dK^T += Q^T dS  (times scale at the end).
// fp32 key-side storage: partial sums of the query slices go to slabs [nparts][BH, Lk, 64] (nparts == 1: straight into dK / dV);
// 16-bit key-side storage (nparts == 1): dK is written, dV written or added to, at layout dkl.
//   grid (ceil(Lk / 128), nparts, BH)
// ------------------------------------------------------------------------------------------------
template <typename T, typename TQ, typename TK>
__global__ __launch_bounds__(256, 2) void attn16_bwd_dkv_kernel(const TQ* __restrict__ Q, const TK* __restrict__ K, const TK* __restrict__ V,
                                                                const TQ* __restrict__ dO, const float* __restrict__ LSE2,
                                                                const float* __restrict__ DELTA, TK* dKp, TK* dVp, int Lq, int Lk,
                                                                float qscale, float scale, int tiles_per_part, size_t part_stride, OLayout ql,
                                                                OLayout kl, OLayout ol, OLayout dkl, int dv_accumulate) {
  typedef typename Pipe<T>::x8 x8;
  __shared__ __attribute__((aligned(16))) T Qr[2][AQ * RLD];
  __shared__ __attribute__((aligned(16))) T Qt[2][AQ * TLD];
  __shared__ __attribute__((aligned(16))) T Dr[2][AQ * RLD];
  __shared__ __attribute__((aligned(16))) T Dt[2][AQ * TLD];
  __shared__ __attribute__((aligned(16))) float lsd[2][2][AQ];     // [buf][lse2 | delta][query]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int bh = blockIdx.z, part = blockIdx.y;
  const int j0 = blockIdx.x * (AK * AW) + wave * AK;
  const bool kvalid = (j0 + c) < Lk;
  const int key = min(j0 + c, Lk - 1);
  const bool wave_has_keys = j0 < Lk;
  constexpr bool KROWS = same_type<TK, T>::value;        // 16-bit key-side storage: this wave's 32 key rows move in full lines
  __shared__ __attribute__((aligned(16))) T Wr[KROWS ? AW * AK * RLD : 8];
  T* wr = Wr + (KROWS ? wave * AK * RLD : 0);
  x8 kf[4], vf[4];
  uint4v dvraw[4];                                      // the running dV rows in line shape (KROWS && dv_accumulate)
  if constexpr (KROWS) {
    uint4v kraw[4], vraw[4];
    rows_load(reinterpret_cast<const T*>(K) + obase(kl, bh), kl.rs, j0, Lk, kraw, lane);
    rows_load(reinterpret_cast<const T*>(V) + obase(kl, bh), kl.rs, j0, Lk, vraw, lane);
    if (dv_accumulate) rows_load(reinterpret_cast<const T*>(dVp) + obase(dkl, bh), dkl.rs, j0, Lk, dvraw, lane);
    rows_to_lds(wr, kraw, lane);
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const x8*>(&wr[c * RLD + 16 * s + 8 * hf]);
    wave_lds_fence();
    rows_to_lds(wr, vraw, lane);
    wave_lds_fence();
#pragma unroll
    for (int s = 0; s < 4; ++s) vf[s] = *reinterpret_cast<const x8*>(&wr[c * RLD + 16 * s + 8 * hf]);
    wave_lds_fence();
  } else {
    const size_t off = obase(kl, bh) + (size_t)key * kl.rs;
#pragma unroll
    for (int s = 0; s < 4; ++s) { kf[s] = row_frag<T>(K + off, s, hf); vf[s] = row_frag<T>(V + off, s, hf); }
  }
  const int nqt = (Lq + AQ - 1) / AQ;
  const int qt_begin = part * tiles_per_part, qt_end = min(qt_begin + tiles_per_part, nqt);
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const TQ* Qb = Q + obase(ql, bh);
  const TQ* dOb = dO + obase(ol, bh);
  TileRegs<T, TQ> qreg, dreg;
  float lreg = 0.f, ereg = 0.f;
  auto fetch = [&](int q0) {
    qreg.load(Qb, ql.rs, q0, Lq, tid);
    dreg.load(dOb, ol.rs, q0, Lq, tid);
    if (tid < AQ) {
      const int row = min(q0 + tid, Lq - 1);
      lreg = LSE2[(size_t)bh * Lq + row];
      ereg = DELTA[(size_t)bh * Lq + row];
    }
  };
  floatx16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
  // a running dV (16-bit key side: the residual convolution's gradient is already there) is requested now, not after the sweep
  float4 va[4], vb[4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) { va[rg] = make_float4(0.f, 0.f, 0.f, 0.f); vb[rg] = va[rg]; }
  if (!KROWS && dv_accumulate) {
    const TK* vp0 = dVp + (size_t)part * part_stride + obase(dkl, bh) + (size_t)key * dkl.rs;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) { va[rg] = load4f(vp0 + 8 * rg + 4 * hf); vb[rg] = load4f(vp0 + 32 + 8 * rg + 4 * hf); }
  }
  if (qt_begin < qt_end) fetch(qt_begin * AQ);
  for (int qt = qt_begin; qt < qt_end; ++qt) {
    const int q0 = qt * AQ, buf = (qt - qt_begin) & 1;
    qreg.store2(Qr[buf], RLD, Qt[buf], TLD, tid);
    dreg.store2(Dr[buf], RLD, Dt[buf], TLD, tid);
    if (tid < AQ) { lsd[buf][0][tid] = lreg; lsd[buf][1][tid] = ereg; }
    __syncthreads();
    if (qt + 1 < qt_end) fetch(q0 + AQ);
    if (wave_has_keys) {                                   // wave-uniform
      floatx16 s = {0}, dp = {0};
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        s = Pipe<T>::mfma(*reinterpret_cast<const x8*>(&Qr[buf][c * RLD + 16 * st + 8 * hf]), kf[st], s);
        dp = Pipe<T>::mfma(*reinterpret_cast<const x8*>(&Dr[buf][c * RLD + 16 * st + 8 * hf]), vf[st], dp);
      }
      float p[16], ds[16];
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 l4 = *reinterpret_cast<const float4*>(&lsd[buf][0][8 * rg + 4 * hf]);     // broadcast reads
        const float4 e4 = *reinterpret_cast<const float4*>(&lsd[buf][1][8 * rg + 4 * hf]);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * rg + i;
          p[r] = __builtin_amdgcn_exp2f(fmaf(s[r], qscale, -lv[i]));
          ds[r] = p[r] * (dp[r] - ev[i]);
        }
      }
      if (q0 + AQ > Lq || j0 + AK > Lk) {                  // ragged query tile or key block only (wave-uniform)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!(kvalid && (q0 + acc_row(r, hf)) < Lq)) { p[r] = 0.f; ds[r] = 0.f; }
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float p8[8], d8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { p8[j] = p[8 * kb + j]; d8[j] = ds[8 * kb + j]; }
        const x8 pb = pack8<T>(p8), db = pack8<T>(d8);
        const int ro = (16 * kb + 4 * hf + trq) * TLD + trc;
        dv0 = Pipe<T>::mfma(frag_tr<T>(&Dt[buf][ro], &Dt[buf][ro + 8 * TLD]), pb, dv0);
        dv1 = Pipe<T>::mfma(frag_tr<T>(&Dt[buf][ro + 32], &Dt[buf][ro + 32 + 8 * TLD]), pb, dv1);
        dk0 = Pipe<T>::mfma(frag_tr<T>(&Qt[buf][ro], &Qt[buf][ro + 8 * TLD]), db, dk0);
        dk1 = Pipe<T>::mfma(frag_tr<T>(&Qt[buf][ro + 32], &Qt[buf][ro + 32 + 8 * TLD]), db, dk1);
      }
    }
  }
  if constexpr (KROWS) {                                 // every lane takes part: pieces -> image -> full lines (no query slices on this form)
    if (dv_accumulate) {
      rows_to_lds(wr, dvraw, lane);
      wave_lds_fence();
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) { va[rg] = load4f(&wr[c * RLD + 8 * rg + 4 * hf]); vb[rg] = load4f(&wr[c * RLD + 32 + 8 * rg + 4 * hf]); }
      wave_lds_fence();
    }
    uint4v oraw[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      store4f(&wr[c * RLD + 8 * rg + 4 * hf], make_float4(dk0[4 * rg] * scale, dk0[4 * rg + 1] * scale, dk0[4 * rg + 2] * scale, dk0[4 * rg + 3] * scale));
      store4f(&wr[c * RLD + 32 + 8 * rg + 4 * hf], make_float4(dk1[4 * rg] * scale, dk1[4 * rg + 1] * scale, dk1[4 * rg + 2] * scale, dk1[4 * rg + 3] * scale));
    }
    wave_lds_fence();
    rows_from_lds(wr, oraw, lane);
    rows_store(reinterpret_cast<T*>(dKp) + obase(dkl, bh), dkl.rs, j0, Lk, oraw, lane);
    wave_lds_fence();
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      store4f(&wr[c * RLD + 8 * rg + 4 * hf], make_float4(dv0[4 * rg] + va[rg].x, dv0[4 * rg + 1] + va[rg].y, dv0[4 * rg + 2] + va[rg].z, dv0[4 * rg + 3] + va[rg].w));
      store4f(&wr[c * RLD + 32 + 8 * rg + 4 * hf], make_float4(dv1[4 * rg] + vb[rg].x, dv1[4 * rg + 1] + vb[rg].y, dv1[4 * rg + 2] + vb[rg].z, dv1[4 * rg + 3] + vb[rg].w));
    }
    wave_lds_fence();
    rows_from_lds(wr, oraw, lane);
    rows_store(reinterpret_cast<T*>(dVp) + obase(dkl, bh), dkl.rs, j0, Lk, oraw, lane);
  } else if (kvalid) {
    const size_t off = (size_t)part * part_stride + obase(dkl, bh) + (size_t)(j0 + c) * dkl.rs;
    TK* kp = dKp + off;
    TK* vp = dVp + off;
    float4 z4[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) z4[rg] = make_float4(0.f, 0.f, 0.f, 0.f);
    store_row(kp, dk0, dk1, scale, z4, z4, hf);
    store_row(vp, dv0, dv1, 1.f, va, vb, hf);
  }
}

// dK = sum_part dKp[part], dV = sum_part dVp[part] in a fixed order (n4 float4 elements per part)
__global__ void attn16_reduce_kernel(const float4* __restrict__ dKp, const float4* __restrict__ dVp, float4* __restrict__ dK,
                                     float4* __restrict__ dV, size_t n4, int nparts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 sk = dKp[i], sv = dVp[i];
  for (int p = 1; p < nparts; ++p) {
    const float4 a = dKp[(size_t)p * n4 + i], b = dVp[(size_t)p * n4 + i];
    sk.x += a.x; sk.y += a.y; sk.z += a.z; sk.w += a.w;
    sv.x += b.x; sv.y += b.y; sv.z += b.z; sv.w += b.w;
  }
  dK[i] = sk; dV[i] = sv;
}

// out = sum of nparts slabs of n4 float4 each, in a fixed order
__global__ void attn16_sum_kernel(const float4* __restrict__ parts, float4* __restrict__ out, size_t n4, int nparts) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 s = parts[i];
  for (int p = 1; p < nparts; ++p) { const float4 a = parts[(size_t)p * n4 + i]; s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
  out[i] = s;
}

// key chunks of the query-owner kernels (forward, pass 1): only when there are too few query tiles to fill the chip
static int attn16_ksplit(int BH, int Lq, int Lk, int* chunk) {
  const long waves = (long)BH * ((Lq + AQ - 1) / AQ);
  long ns = (2048 + waves - 1) / waves;
  const long maxs = (Lk + 255) / 256;                 // at least 256 keys per chunk
  if (ns > maxs) ns = maxs;
  if (ns > 16) ns = 16;
  if (ns < 1) ns = 1;
  int c = (int)((Lk + ns - 1) / ns);
  c = (c + AK - 1) / AK * AK;                         // whole 32-key tiles
  *chunk = c;
  return (int)((Lk + c - 1) / c);
}

// query slices of pass 2: enough workgroups to fill the chip a few times over, at most 32 slabs
static int attn16_parts(int BH, int Lq, int Lk) {
  const long base = (long)((Lk + AK * AW - 1) / (AK * AW)) * BH;
  const int nqt = (Lq + AQ - 1) / AQ;
  long parts = (1024 + base - 1) / base;
  if (parts < 1) parts = 1;
  if (parts > 32) parts = 32;
  if (parts > nqt) parts = nqt;
  return (int)parts;
}

}  // namespace

// forward with <= 256 keys: -1 = read SMML_ATTN16_FEWKEYS (default 0), 0 = the online-softmax kernel, 1 = the two-pass kernel where its
// grid fills the chip, 2 = wherever Lk <= 256 (tests).  OFF by default - measured at 4 x 8 heads x 10 240 queries x 256 keys
// (profiles/r03_nystrom16_notes.md): the two-pass kernel issues 13 vector instructions per MFMA instead of 24, and is SLOWER (69 vs 54
// us): with fp32 storage this launch moves 252 MB (q in, residual in, out out) - 4.7 TB/s at 54 us - i.e. the [n', m] side of the Nystrom
// block is HBM-bound, not issue-bound, and one 8-wave workgroup per CU (86 KB of LDS) keeps fewer bytes in flight than two 4-wave ones.
static std::atomic<int> g_fewkeys{-1};
static std::atomic<int> g_q2{1};              // queries-long bf16-storage forward: 0 one query block per wave, 1 (default) two blocks per wave where the
                                  // 256-query workgroups still cover the chip twice over, 2 always (smml_attn16_set_query_blocks)

extern "C" {

void smml_attn16_set_fewkeys(int mode) { g_fewkeys = mode; }
void smml_attn16_set_query_blocks(int mode) { g_q2 = mode; }

// scratch of smml_attn16_fwd_f32: the partial outputs + log-sum-exps of a key-split launch (0 when the launch is not split)
size_t smml_attn16_fwd_workspace_bytes(int BH, int Lq, int Lk) {
  if (BH <= 0 || Lq <= 0 || Lk <= 0) return 0;
  int chunk;
  const int ns = attn16_ksplit(BH, Lq, Lk, &chunk);
  return ns > 1 ? (size_t)ns * BH * Lq * (AD + 1) * sizeof(float) : 0;
}

// scratch of smml_attn16_bwd_f32: delta [BH, Lq] + the dQ slabs of a key-split pass 1 + the dK / dV slabs of the query slices
size_t smml_attn16_bwd_workspace_bytes(int BH, int Lq, int Lk) {
  if (BH <= 0 || Lq <= 0 || Lk <= 0) return 0;
  const int parts = attn16_parts(BH, Lq, Lk);
  int chunk;
  const int ns = attn16_ksplit(BH, Lq, Lk, &chunk);
  const size_t delta = ((size_t)BH * Lq + 3) & ~(size_t)3;
  const size_t qslabs = ns > 1 ? (size_t)ns * BH * Lq * AD : 0;
  const size_t slabs = parts > 1 ? (size_t)2 * parts * BH * Lk * AD : 0;
  return (delta + qslabs + slabs) * sizeof(float);
}

static int attn16_layout(const char* fn, int BH, int Lq, int heads_merged, OLayout* ol) {
  SMML_REQUIRE(heads_merged >= 0, "%s: heads_merged must be 0 (head-major output) or the number of heads", fn);
  if (heads_merged == 0) { *ol = OLayout{(long long)Lq * AD, 0, AD, 1}; return SMML_OK; }      // bh / 1 = bh
  SMML_REQUIRE(BH % heads_merged == 0, "%s: BH (%d) is not a multiple of the head count (%d)", fn, BH, heads_merged);
  *ol = OLayout{(long long)Lq * heads_merged * AD, AD, (long long)heads_merged * AD, heads_merged};
  return SMML_OK;
}
static OLayout head_major(int L) { return OLayout{(long long)L * AD, 0, AD, 1}; }

int smml_attn16_fwd_f32(const float* q, const float* k, const float* v, float* out, float* lse2, void* workspace,
                        size_t workspace_bytes, int BH, int Lq, int Lk, int D, float scale, int use_fp16, int heads_merged,
                        int accumulate, void* stream) {
  SMML_REQUIRE(q && k && v && out && lse2, "smml_attn16_fwd_f32: null pointer");
  SMML_REQUIRE(BH > 0 && BH <= 65535 && Lq > 0 && Lk > 0, "smml_attn16_fwd_f32: bad sizes (BH=%d Lq=%d Lk=%d)", BH, Lq, Lk);
  SMML_REQUIRE(D == AD, "smml_attn16_fwd_f32: head dim must be %d (got %d)", AD, D);
  OLayout ol;
  int rc = attn16_layout("smml_attn16_fwd_f32", BH, Lq, heads_merged, &ol);
  if (rc) return rc;
  int chunk;
  const int ns = attn16_ksplit(BH, Lq, Lk, &chunk);
  float *opart = nullptr, *lpart = nullptr;
  if (ns > 1) {
    SMML_REQUIRE(workspace && workspace_bytes >= smml_attn16_fwd_workspace_bytes(BH, Lq, Lk) && (reinterpret_cast<size_t>(workspace) & 15) == 0,
                 "smml_attn16_fwd_f32: this shape runs key-split and needs a 16-byte aligned workspace of %zu bytes",
                 smml_attn16_fwd_workspace_bytes(BH, Lq, Lk));
    opart = reinterpret_cast<float*>(workspace);
    lpart = opart + (size_t)ns * BH * Lq * AD;
  }
  hipStream_t st = (hipStream_t)stream;
  const float qscale = scale * LOG2E_F;
  // few keys, many queries (the [n', m] side of the Nystrom block): the two-pass kernel with all keys resident in LDS
  if (g_fewkeys < 0) { const char* e = getenv("SMML_ATTN16_FEWKEYS"); g_fewkeys = e ? atoi(e) : 0; }
  const bool fills = (long)BH * ((Lq + AQ * SK_WAVES - 1) / (AQ * SK_WAVES)) >= 128;      // enough 256-query workgroups for the chip
  if (ns == 1 && Lk <= SK_MAX && (g_fewkeys == 2 || (g_fewkeys == 1 && fills))) {
    // one workgroup stages the keys once and walks blocks_per_wg 256-query blocks: ~1.5 workgroups per CU in all
    const int nblk = (Lq + AQ * SK_WAVES - 1) / (AQ * SK_WAVES);
    const int bpw = (int)std::max(1L, std::min<long>(nblk, ((long)nblk * BH) / 384));
    dim3 gridf((nblk + bpw - 1) / bpw, BH), blockf(64 * SK_WAVES);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn16_fwd_fewkeys_kernel<_Float16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn16_fwd_fewkeys_kernel<__bf16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS);
      attr_set = true;
    }
    if (use_fp16) hipLaunchKernelGGL(attn16_fwd_fewkeys_kernel<_Float16>, gridf, blockf, SK_LDS, st, q, k, v, out, lse2, Lq, Lk, qscale, ol, accumulate, bpw);
    else hipLaunchKernelGGL(attn16_fwd_fewkeys_kernel<__bf16>, gridf, blockf, SK_LDS, st, q, k, v, out, lse2, Lq, Lk, qscale, ol, accumulate, bpw);
    SMML_LAUNCH_CHECK("smml_attn16_fwd_f32/fewkeys");
    return SMML_OK;
  }
  dim3 grid((Lq + AQ * AW - 1) / (AQ * AW), BH, ns), block(256);
  const OLayout ql = head_major(Lq), kl = head_major(Lk);
  const float* res = accumulate ? out : nullptr;      // the output buffer already holds the residual
  if (use_fp16) hipLaunchKernelGGL((attn16_fwd_kernel<_Float16, float, float>), grid, block, 0, st, q, k, v, out, res, lse2, Lq, Lk, qscale, ql, kl, ol, chunk, opart, lpart);
  else hipLaunchKernelGGL((attn16_fwd_kernel<__bf16, float, float>), grid, block, 0, st, q, k, v, out, res, lse2, Lq, Lk, qscale, ql, kl, ol, chunk, opart, lpart);
  SMML_LAUNCH_CHECK("smml_attn16_fwd_f32");
  if (ns > 1) {
    const size_t n = (size_t)BH * Lq * (AD / 4);
    hipLaunchKernelGGL(attn16_merge_kernel, dim3((unsigned)((n + 255) / 256)), block, 0, st, opart, lpart, out, lse2, BH, Lq, ns, ol, accumulate);
    SMML_LAUNCH_CHECK("smml_attn16_fwd_f32/merge");
  }
  return SMML_OK;
}

int smml_attn16_bwd_f32(const float* q, const float* k, const float* v, const float* out, const float* residual, const float* dout,
                        const float* lse2, float* dq, float* dk, float* dv, void* workspace, size_t workspace_bytes, int BH, int Lq,
                        int Lk, int D, float scale, int use_fp16, int heads_merged, void* stream) {
  SMML_REQUIRE(q && k && v && out && dout && lse2 && dq && dk && dv && workspace, "smml_attn16_bwd_f32: null pointer");
  SMML_REQUIRE(BH > 0 && BH <= 65535 && Lq > 0 && Lk > 0, "smml_attn16_bwd_f32: bad sizes (BH=%d Lq=%d Lk=%d)", BH, Lq, Lk);
  SMML_REQUIRE(D == AD, "smml_attn16_bwd_f32: head dim must be %d (got %d)", AD, D);
  SMML_REQUIRE(workspace_bytes >= smml_attn16_bwd_workspace_bytes(BH, Lq, Lk), "smml_attn16_bwd_f32: workspace too small");
  SMML_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "smml_attn16_bwd_f32: workspace must be 16-byte aligned");
  OLayout ol;
  int rc = attn16_layout("smml_attn16_bwd_f32", BH, Lq, heads_merged, &ol);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const float qscale = scale * LOG2E_F;
  int chunk;
  const int ns = attn16_ksplit(BH, Lq, Lk, &chunk);
  float* delta = reinterpret_cast<float*>(workspace);
  float* qslabs = delta + (((size_t)BH * Lq + 3) & ~(size_t)3);
  float* slabs = qslabs + (ns > 1 ? (size_t)ns * BH * Lq * AD : 0);
  dim3 block(256);
  dim3 gq((Lq + AQ * AW - 1) / (AQ * AW), BH, ns);
  const OLayout ql = head_major(Lq), kl = head_major(Lk);
  float* qs = ns > 1 ? qslabs : nullptr;
  if (use_fp16) hipLaunchKernelGGL((attn16_bwd_dq_kernel<_Float16, float, float>), gq, block, 0, st, q, k, v, out, dout, lse2, dq, qs, delta, Lq, Lk, qscale, scale, ql, kl, ol, ql, residual, chunk);
  else hipLaunchKernelGGL((attn16_bwd_dq_kernel<__bf16, float, float>), gq, block, 0, st, q, k, v, out, dout, lse2, dq, qs, delta, Lq, Lk, qscale, scale, ql, kl, ol, ql, residual, chunk);
  SMML_LAUNCH_CHECK("smml_attn16_bwd_f32/dq");
  if (ns > 1) {
    const size_t n4 = (size_t)BH * Lq * AD / 4;
    hipLaunchKernelGGL(attn16_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), block, 0, st, reinterpret_cast<const float4*>(qslabs),
                       reinterpret_cast<float4*>(dq), n4, ns);
    SMML_LAUNCH_CHECK("smml_attn16_bwd_f32/dq_sum");
  }
  const int parts = attn16_parts(BH, Lq, Lk);
  const int nqt = (Lq + AQ - 1) / AQ, tpp = (nqt + parts - 1) / parts;
  const size_t per = (size_t)BH * Lk * AD;
  float* dkp = parts > 1 ? slabs : dk;
  float* dvp = parts > 1 ? slabs + (size_t)parts * per : dv;
  dim3 gk((Lk + AK * AW - 1) / (AK * AW), parts, BH);
  if (use_fp16) hipLaunchKernelGGL((attn16_bwd_dkv_kernel<_Float16, float, float>), gk, block, 0, st, q, k, v, dout, lse2, delta, dkp, dvp, Lq, Lk, qscale, scale, tpp, per, ql, kl, ol, kl, 0);
  else hipLaunchKernelGGL((attn16_bwd_dkv_kernel<__bf16, float, float>), gk, block, 0, st, q, k, v, dout, lse2, delta, dkp, dvp, Lq, Lk, qscale, scale, tpp, per, ql, kl, ol, kl, 0);
  SMML_LAUNCH_CHECK("smml_attn16_bwd_f32/dkv");
  if (parts > 1) {
    const size_t n4 = per / 4;
    hipLaunchKernelGGL(attn16_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), block, 0, st, reinterpret_cast<const float4*>(dkp),
                       reinterpret_cast<const float4*>(dvp), reinterpret_cast<float4*>(dk), reinterpret_cast<float4*>(dv), n4, parts);
    SMML_LAUNCH_CHECK("smml_attn16_bwd_f32/reduce");
  }
  return SMML_OK;
}

// ---- bf16-storage forms: the LONG side of the product (long_side 0: keys / values, 1: queries) is bf16 in memory at strides
// (s_bs, s_hs, s_rs) per (bag, head, row), in elements; the short side is fp32 head-major [B H, L, 64].
//   long_side 0 ("a3": softmax(ql k^T) v):  q fp32, k / v bf16 strided, out fp32 [B H, Lq, 64]
//   long_side 1 ("a1": softmax(q kl^T) w):  q bf16 strided, k / v fp32, out = attention + residual in bf16 at strides (o_bs, o_hs, o_rs);
//                                           residual (same strides, may be NULL)
static int b16_check(const char* fn, int B, int H, int Lq, int Lk, int long_side, long long s_bs, long long s_hs, long long s_rs) {
  SMML_REQUIRE(B > 0 && H > 0 && (long long)B * H <= 65535 && Lq > 0 && Lk > 0, "%s: bad sizes (B=%d H=%d Lq=%d Lk=%d)", fn, B, H, Lq, Lk);
  SMML_REQUIRE(long_side == 0 || long_side == 1, "%s: long_side must be 0 (keys) or 1 (queries)", fn);
  SMML_REQUIRE((s_bs % 8) == 0 && (s_hs % 8) == 0 && (s_rs % 8) == 0 && s_rs >= AD, "%s: bf16 strides must be multiples of 8 elements (16-byte rows)", fn);
  return SMML_OK;
}
static bool al16p(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

int smml_attn16_fwd_b16(const void* q, const void* k, const void* v, void* out, const void* residual, float* lse2, void* workspace,
                        size_t workspace_bytes, int B, int H, int Lq, int Lk, float scale, int long_side, long long s_bs, long long s_hs,
                        long long s_rs, long long o_bs, long long o_hs, long long o_rs, void* stream) {
  SMML_REQUIRE(q && k && v && out && lse2, "smml_attn16_fwd_b16: null pointer");
  int rc = b16_check("smml_attn16_fwd_b16", B, H, Lq, Lk, long_side, s_bs, s_hs, s_rs);
  if (rc) return rc;
  const int BH = B * H;
  hipStream_t st = (hipStream_t)stream;
  const float qscale = scale * LOG2E_F;
  const OLayout sl{s_bs, s_hs, s_rs, H};
  dim3 block(256);
  if (long_side == 0) {
    SMML_REQUIRE(!residual, "smml_attn16_fwd_b16: no residual on the keys-long form");
    SMML_REQUIRE(al16p(k) && al16p(v), "smml_attn16_fwd_b16: k / v must be 16-byte aligned");
    int chunk;
    const int ns = attn16_ksplit(BH, Lq, Lk, &chunk);
    float *opart = nullptr, *lpart = nullptr;
    if (ns > 1) {
      SMML_REQUIRE(workspace && workspace_bytes >= smml_attn16_fwd_workspace_bytes(BH, Lq, Lk) && al16p(workspace),
                   "smml_attn16_fwd_b16: this shape runs key-split and needs a 16-byte aligned workspace of %zu bytes",
                   smml_attn16_fwd_workspace_bytes(BH, Lq, Lk));
      opart = reinterpret_cast<float*>(workspace);
      lpart = opart + (size_t)ns * BH * Lq * AD;
    }
    const OLayout ql = head_major(Lq);
    dim3 grid((Lq + AQ * AW - 1) / (AQ * AW), BH, ns);
    hipLaunchKernelGGL((attn16_fwd_kernel<__bf16, float, __bf16>), grid, block, 0, st, reinterpret_cast<const float*>(q),
                       reinterpret_cast<const __bf16*>(k), reinterpret_cast<const __bf16*>(v), reinterpret_cast<float*>(out),
                       (const float*)nullptr, lse2, Lq, Lk, qscale, ql, sl, ql, chunk, opart, lpart);
    SMML_LAUNCH_CHECK("smml_attn16_fwd_b16/keys");
    if (ns > 1) {
      const size_t n = (size_t)BH * Lq * (AD / 4);
      hipLaunchKernelGGL(attn16_merge_kernel, dim3((unsigned)((n + 255) / 256)), block, 0, st, opart, lpart, reinterpret_cast<float*>(out), lse2, BH, Lq, ns, ql, 0);
      SMML_LAUNCH_CHECK("smml_attn16_fwd_b16/merge");
    }
    return SMML_OK;
  }
  SMML_REQUIRE((o_bs % 8) == 0 && (o_hs % 8) == 0 && (o_rs % 8) == 0 && al16p(q) && al16p(out) && al16p(residual),
               "smml_attn16_fwd_b16: bf16 operands must be 16-byte aligned with strides that are multiples of 8");
  const OLayout ol{o_bs, o_hs, o_rs, H}, kl = head_major(Lk);
  // two query blocks per wave (attn16_fwd_q2_kernel) wherever its 256-query workgroups still cover the chip twice over; g_q2 = 0 / 2: never / always
  const long long wg2 = (long long)((Lq + 2 * AQ * AW - 1) / (2 * AQ * AW)) * BH;
  if (g_q2 == 2 || (g_q2 == 1 && wg2 >= 1024)) {
    dim3 grid2((Lq + 2 * AQ * AW - 1) / (2 * AQ * AW), BH, 1);
    hipLaunchKernelGGL(attn16_fwd_q2_kernel<__bf16>, grid2, block, 0, st, reinterpret_cast<const __bf16*>(q), reinterpret_cast<const float*>(k),
                       reinterpret_cast<const float*>(v), reinterpret_cast<__bf16*>(out), reinterpret_cast<const __bf16*>(residual), lse2, Lq, Lk,
                       qscale, sl, kl, ol);
    SMML_LAUNCH_CHECK("smml_attn16_fwd_b16/queries2");
    return SMML_OK;
  }
  dim3 grid((Lq + AQ * AW - 1) / (AQ * AW), BH, 1);
  hipLaunchKernelGGL((attn16_fwd_kernel<__bf16, __bf16, float>), grid, block, 0, st, reinterpret_cast<const __bf16*>(q),
                     reinterpret_cast<const float*>(k), reinterpret_cast<const float*>(v), reinterpret_cast<__bf16*>(out),
                     reinterpret_cast<const __bf16*>(residual), lse2, Lq, Lk, qscale, sl, kl, ol, Lk, (float*)nullptr, (float*)nullptr);
  SMML_LAUNCH_CHECK("smml_attn16_fwd_b16/queries");
  return SMML_OK;
}

// Backward of the forms above.  Gradients of the long side are bf16 at strides (g_bs, g_hs, g_rs) - they may live in another buffer than
// the operands (the gradient of the qkv buffer); gradients of the short side are fp32 head-major.
//   long_side 0: dout fp32 [B H, Lq, 64], out fp32; dq fp32; dk written, dv written or (dv_accumulate) added to
//   long_side 1: out, residual, dout bf16 at the o strides; dq bf16 written; dk, dv fp32
// workspace: smml_attn16_bwd_workspace_bytes(B H, Lq, Lk) bytes, 16-byte aligned.  The keys-long form never slices the queries (dk / dv go
// straight to their bf16 rows), the queries-long form never splits the keys.
int smml_attn16_bwd_b16(const void* q, const void* k, const void* v, const void* out, const void* residual, const void* dout,
                        const float* lse2, void* dq, void* dk, void* dv, void* workspace, size_t workspace_bytes, int B, int H, int Lq,
                        int Lk, float scale, int long_side, long long s_bs, long long s_hs, long long s_rs, long long o_bs, long long o_hs,
                        long long o_rs, long long g_bs, long long g_hs, long long g_rs, int dv_accumulate, void* stream) {
  SMML_REQUIRE(q && k && v && out && dout && lse2 && dq && dk && dv && workspace, "smml_attn16_bwd_b16: null pointer");
  int rc = b16_check("smml_attn16_bwd_b16", B, H, Lq, Lk, long_side, s_bs, s_hs, s_rs);
  if (rc) return rc;
  const int BH = B * H;
  SMML_REQUIRE(workspace_bytes >= smml_attn16_bwd_workspace_bytes(BH, Lq, Lk) && al16p(workspace), "smml_attn16_bwd_b16: workspace too small or misaligned");
  SMML_REQUIRE((g_bs % 8) == 0 && (g_hs % 8) == 0 && (g_rs % 8) == 0 && g_rs >= AD, "smml_attn16_bwd_b16: gradient strides must be multiples of 8");
  hipStream_t st = (hipStream_t)stream;
  const float qscale = scale * LOG2E_F;
  const OLayout sl{s_bs, s_hs, s_rs, H}, gl{g_bs, g_hs, g_rs, H};
  float* delta = reinterpret_cast<float*>(workspace);
  dim3 block(256);
  if (long_side == 0) {
    SMML_REQUIRE(!residual, "smml_attn16_bwd_b16: no residual on the keys-long form");
    SMML_REQUIRE(al16p(k) && al16p(v) && al16p(dk) && al16p(dv), "smml_attn16_bwd_b16: bf16 operands must be 16-byte aligned");
    int chunk;
    const int ns = attn16_ksplit(BH, Lq, Lk, &chunk);
    float* qslabs = delta + (((size_t)BH * Lq + 3) & ~(size_t)3);
    const OLayout ql = head_major(Lq);
    dim3 gq((Lq + AQ * AW - 1) / (AQ * AW), BH, ns);
    hipLaunchKernelGGL((attn16_bwd_dq_kernel<__bf16, float, __bf16>), gq, block, 0, st, reinterpret_cast<const float*>(q),
                       reinterpret_cast<const __bf16*>(k), reinterpret_cast<const __bf16*>(v), reinterpret_cast<const float*>(out),
                       reinterpret_cast<const float*>(dout), lse2, reinterpret_cast<float*>(dq), ns > 1 ? qslabs : (float*)nullptr, delta, Lq, Lk,
                       qscale, scale, ql, sl, ql, ql, (const float*)nullptr, chunk);
    SMML_LAUNCH_CHECK("smml_attn16_bwd_b16/keys dq");
    if (ns > 1) {
      const size_t n4 = (size_t)BH * Lq * AD / 4;
      hipLaunchKernelGGL(attn16_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), block, 0, st, reinterpret_cast<const float4*>(qslabs),
                         reinterpret_cast<float4*>(dq), n4, ns);
      SMML_LAUNCH_CHECK("smml_attn16_bwd_b16/dq_sum");
    }
    const int nqt = (Lq + AQ - 1) / AQ;
    dim3 gk((Lk + AK * AW - 1) / (AK * AW), 1, BH);
    hipLaunchKernelGGL((attn16_bwd_dkv_kernel<__bf16, float, __bf16>), gk, block, 0, st, reinterpret_cast<const float*>(q),
                       reinterpret_cast<const __bf16*>(k), reinterpret_cast<const __bf16*>(v), reinterpret_cast<const float*>(dout), lse2, delta,
                       reinterpret_cast<__bf16*>(dk), reinterpret_cast<__bf16*>(dv), Lq, Lk, qscale, scale, nqt, (size_t)0, ql, sl, ql, gl,
                       dv_accumulate);
    SMML_LAUNCH_CHECK("smml_attn16_bwd_b16/keys dkv");
    return SMML_OK;
  }
  SMML_REQUIRE((o_bs % 8) == 0 && (o_hs % 8) == 0 && (o_rs % 8) == 0 && al16p(q) && al16p(out) && al16p(residual) && al16p(dout) && al16p(dq),
               "smml_attn16_bwd_b16: bf16 operands must be 16-byte aligned with strides that are multiples of 8");
  const OLayout ol{o_bs, o_hs, o_rs, H}, kl = head_major(Lk);      // no key split on this form: the long side is the query side
  dim3 gq((Lq + AQ * AW - 1) / (AQ * AW), BH, 1);
  hipLaunchKernelGGL((attn16_bwd_dq_kernel<__bf16, __bf16, float>), gq, block, 0, st, reinterpret_cast<const __bf16*>(q),
                     reinterpret_cast<const float*>(k), reinterpret_cast<const float*>(v), reinterpret_cast<const __bf16*>(out),
                     reinterpret_cast<const __bf16*>(dout), lse2, reinterpret_cast<__bf16*>(dq), (float*)nullptr, delta, Lq, Lk, qscale, scale, sl,
                     kl, ol, gl, reinterpret_cast<const __bf16*>(residual), Lk);
  SMML_LAUNCH_CHECK("smml_attn16_bwd_b16/queries dq");
  const int parts = attn16_parts(BH, Lq, Lk);
  const int nqt = (Lq + AQ - 1) / AQ, tpp = (nqt + parts - 1) / parts;
  const size_t per = (size_t)BH * Lk * AD;
  float* slabs = delta + (((size_t)BH * Lq + 3) & ~(size_t)3);
  float* dkp = parts > 1 ? slabs : reinterpret_cast<float*>(dk);
  float* dvp = parts > 1 ? slabs + (size_t)parts * per : reinterpret_cast<float*>(dv);
  dim3 gk((Lk + AK * AW - 1) / (AK * AW), parts, BH);
  hipLaunchKernelGGL((attn16_bwd_dkv_kernel<__bf16, __bf16, float>), gk, block, 0, st, reinterpret_cast<const __bf16*>(q),
                     reinterpret_cast<const float*>(k), reinterpret_cast<const float*>(v), reinterpret_cast<const __bf16*>(dout), lse2, delta, dkp, dvp,
                     Lq, Lk, qscale, scale, tpp, per, sl, kl, ol, kl, 0);
  SMML_LAUNCH_CHECK("smml_attn16_bwd_b16/queries dkv");
  if (parts > 1) {
    const size_t n4 = per / 4;
    hipLaunchKernelGGL(attn16_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), block, 0, st, reinterpret_cast<const float4*>(dkp),
                       reinterpret_cast<const float4*>(dvp), reinterpret_cast<float4*>(dk), reinterpret_cast<float4*>(dv), n4, parts);
    SMML_LAUNCH_CHECK("smml_attn16_bwd_b16/reduce");
  }
  return SMML_OK;
}

}  // extern "C"
