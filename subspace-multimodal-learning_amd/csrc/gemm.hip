// Strided-batched fp32 GEMM on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32):
//   C[b0,b1][m,n] (+)= act(alpha * sum_k A[b0,b1](m,k) * B[b0,b1](k,n) + bias) + beta * residual
// Arbitrary element strides on A and B cover the NN / NT / TN products needed by the linear layers of
// the path, forward and backward:
//   _fc1, fusion_layer, to_q/to_k/to_v (grouped 1x1 convs = batch over groups), to_out, pooler.dense,
//   _fc2, multimodal_projection  (models/DeformCrossTransMIL.py:35-37,83,93-95; DeformableAttention2D.py:218-221)
//   Nystrom to_qkv / to_out and the landmark products (models/NystromAttention.py:86,122-140)
// Split-K (atomic accumulate into a zeroed C) serves the weight-gradient products whose reduction runs
// over all tokens.
// Tile: 128 x 64 x 16 per 256-thread workgroup; each wave owns 32 rows x 64 columns (two 32x32
// accumulators); operands are staged k-major in LDS so that the 32 lanes of a half-wave read
// consecutive banks.
#include "smml_common.h"
#include <atomic>     // process-wide measurement switches (set once from the environment or a test hook): plain atomics, no launch state

namespace {

constexpr int BM = 128, BN = 64, BK = 16;
constexpr int LDA = BM + 4, LDB = BN + 4;

struct GemmArgs {
  const float* A; const float* B; float* C; const float* bias; const float* residual;
  int M, N, K;
  long long sam, sak, sbk, sbn, ldc, ldr;
  int nb0, nb1;
  long long sa0, sa1, sb0, sb1, sc0, sc1, sbias0, sbias1;
  int bias_mode, rows_per_bias; long long bias_ld;
  int act, splitk; float alpha, beta;
  int atomic;    // accumulate into C with float atomics (split-K, or batch dims folded onto one C)
  int swap_xy;   // column tiles on grid.x (wide outputs: > 65535 column tiles)
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return tanhf(v);
  return v;
}

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ float As[BK][LDA];
  __shared__ float Bs[BK][LDB];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int zb = blockIdx.z / g.splitk, ks = blockIdx.z - zb * g.splitk;
  const int b0 = zb / g.nb1, b1 = zb - b0 * g.nb1;
  const float* A = g.A + b0 * g.sa0 + b1 * g.sa1;
  const float* B = g.B + b0 * g.sb0 + b1 * g.sb1;
  float* C = g.C + b0 * g.sc0 + b1 * g.sc1;
  const int m0 = (g.swap_xy ? blockIdx.y : blockIdx.x) * BM, n0 = (g.swap_xy ? blockIdx.x : blockIdx.y) * BN;
  // K range of this split, in whole BK tiles
  const int ktiles = (g.K + BK - 1) / BK;
  const int tps = (ktiles + g.splitk - 1) / g.splitk;
  const int kt0 = ks * tps, kt1 = min(ktiles, kt0 + tps);

  floatx16 acc0 = {0}, acc1 = {0};
  const bool a_kcontig = (g.sak == 1);
  const bool b_kcontig = (g.sbk == 1) && (g.sbn != 1);

  for (int kt = kt0; kt < kt1; ++kt) {
    const int k0 = kt * BK;
    __syncthreads();
    // ---- stage A tile [BK][BM] ----
    if (a_kcontig) {
      const int k = tid & 15;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = (tid >> 4) + 16 * i;
        const int gm = m0 + m, gk = k0 + k;
        As[k][m] = (gm < g.M && gk < g.K) ? A[gm * g.sam + gk] : 0.f;
      }
    } else {
      const int m = tid & 127;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = (tid >> 7) + 2 * i;
        const int gm = m0 + m, gk = k0 + k;
        As[k][m] = (gm < g.M && gk < g.K) ? A[gm * g.sam + gk * g.sak] : 0.f;
      }
    }
    // ---- stage B tile [BK][BN] ----
    if (b_kcontig) {
      const int k = tid & 15;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n = (tid >> 4) + 16 * i;
        const int gn = n0 + n, gk = k0 + k;
        Bs[k][n] = (gn < g.N && gk < g.K) ? B[gk + gn * g.sbn] : 0.f;
      }
    } else {
      const int n = tid & 63;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = (tid >> 6) + 4 * i;
        const int gn = n0 + n, gk = k0 + k;
        Bs[k][n] = (gn < g.N && gk < g.K) ? B[gk * g.sbk + gn * g.sbn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < BK / 2; ++st) {
      const float a = As[2 * st + hf][wave * 32 + c];
      acc0 = mfma32(a, Bs[2 * st + hf][c], acc0);
      acc1 = mfma32(a, Bs[2 * st + hf][32 + c], acc1);
    }
  }

  // ---- epilogue ----
  const float* bias = g.bias ? g.bias + b0 * g.sbias0 + b1 * g.sbias1 : nullptr;
  const float* res = g.residual ? g.residual + b0 * g.sc0 + b1 * g.sc1 : nullptr;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = n0 + 32 * t + c;
    if (n >= g.N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + acc_row(r, hf);
      if (m >= g.M) continue;
      float v = g.alpha * (t == 0 ? acc0[r] : acc1[r]);
      if (g.atomic) {
        if (bias && ks == 0) v += (g.bias_mode == 2) ? bias[(m / g.rows_per_bias) * g.bias_ld + n] : bias[n];
        atomicAdd(&C[m * g.ldc + n], v);
      } else {
        if (bias) v += (g.bias_mode == 2) ? bias[(m / g.rows_per_bias) * g.bias_ld + n] : bias[n];
        v = apply_act(v, g.act);
        if (res) v = fmaf(g.beta, res[m * g.ldr + n], v);
        C[m * g.ldc + n] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fast path: 128 x BN x 16 tile, 2 x 2 waves (64 x BN/2 per wave), float4 global loads staged through
// registers one K-tile ahead (issue early, write to LDS after the barrier), 16-byte LDS fragment reads when the
// operand is k-contiguous.  v_mfma_f32_32x32x2_f32 shares the SIMD's fp32 ALUs with ordinary VALU work
// (tests/microbench/mfma_probe.hip), so the inner loop is built for few vector / LDS instructions per MFMA:
// per K-tile a wave issues 8 x (BN/32) MFMAs against <= 8 fragment reads.
// Requirements (checked on the host): K % 16 == 0, 16-byte aligned operands with the unit stride on k or on
// the row index and all other strides multiples of 4 elements.
// ------------------------------------------------------------------------------------------------
template <int BN_, bool A_KC, bool B_KC, int FBM = 128>
__global__ __launch_bounds__(256, 2) void gemm_f32_fast_kernel(GemmArgs g) {
  constexpr int FBK = 16, NI = BN_ / 64, MI = FBM / 64;      // MI x NI 32x32 blocks per wave; FBM = 64: the small-problem tile
  constexpr int NA = FBM / 64;                               // float4 of A per thread per K-tile
  constexpr int A_LD = A_KC ? (FBK + 4) : (FBM + 4);
  constexpr int B_LD = B_KC ? (FBK + 4) : (BN_ + 4);
  __shared__ __attribute__((aligned(16))) float As[(A_KC ? FBM : FBK) * A_LD];
  __shared__ __attribute__((aligned(16))) float Bs[(B_KC ? BN_ : FBK) * B_LD];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int zb = blockIdx.z / g.splitk, ks = blockIdx.z - zb * g.splitk;
  const int b0 = zb / g.nb1, b1 = zb - b0 * g.nb1;
  const float* A = g.A + b0 * g.sa0 + b1 * g.sa1;
  const float* B = g.B + b0 * g.sb0 + b1 * g.sb1;
  float* C = g.C + b0 * g.sc0 + b1 * g.sc1;
  const int m0 = (g.swap_xy ? blockIdx.y : blockIdx.x) * FBM, n0 = (g.swap_xy ? blockIdx.x : blockIdx.y) * BN_;
  const int ktiles = (g.K + FBK - 1) / FBK;         // a K tail is zero-filled at load time
  const int tps = (ktiles + g.splitk - 1) / g.splitk;
  const int kt0 = ks * tps, kt1 = min(ktiles, kt0 + tps);

  // staging maps (2 float4 of A, NI float4 of B per thread).  Everything that does not depend on the K-tile - row / column
  // clamps, 64-bit row offsets, which of the three forms a row-contiguous float4 takes - is worked out once here: the
  // fp32 MFMA shares its ALUs with vector work, so address arithmetic inside the K loop costs matrix throughput.
  float4 ra[NA], rb[NI];
  const float* pa[NA];            // k-contiguous: row base (+ 4 kq); row-contiguous: column base (+ 4 mq), advanced by k * stride
  const float* pb[NI];
  int ka[NA], kb_[NI];            // the k index (within a tile) this thread stages
  int fa[NA], fb[NI];             // row-contiguous operands: 0 = nothing to read, 1 = one float4, 2 = ragged edge (scalar reads)
  constexpr int MQ = FBM / 4;                                // float4 per k-row of a row-contiguous A tile
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    if (A_KC) {
      const int row = (tid >> 2) + 64 * i, kq = tid & 3;
      pa[i] = A + (long long)min(m0 + row, g.M - 1) * g.sam + 4 * kq;
      ka[i] = 4 * kq; fa[i] = 1;
    } else {
      const int idx = tid + 256 * i, k = idx / MQ, mq = idx - k * MQ, gm = m0 + 4 * mq;
      pa[i] = A + min(gm, g.M - 1);
      ka[i] = k; fa[i] = (gm >= g.M) ? 0 : (gm + 3 < g.M ? 1 : 2);
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (B_KC) {
      const int row = (tid >> 2) + 64 * i, kq = tid & 3;
      pb[i] = B + (long long)min(n0 + row, g.N - 1) * g.sbn + 4 * kq;
      kb_[i] = 4 * kq; fb[i] = 1;
    } else {
      constexpr int NQ = BN_ / 4;                            // float4 per k-row
      const int idx = tid + 256 * i, k = idx / NQ, nq = idx - k * NQ, gn = n0 + 4 * nq;
      pb[i] = B + min(gn, g.N - 1);
      kb_[i] = k; fb[i] = (gn >= g.N) ? 0 : (gn + 3 < g.N ? 1 : 2);
    }
  }
  auto load_tile = [&](int kt) {
    const int k0 = kt * FBK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (A_KC) {
        ra[i] = (k0 + ka[i] < g.K) ? *reinterpret_cast<const float4*>(pa[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int k = k0 + ka[i];
        const float* p = pa[i] + (long long)min(k, g.K - 1) * g.sak;
        if (k >= g.K || fa[i] == 0) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);               // K tail / tile wider than the operand: nothing to read
        else if (fa[i] == 1) ra[i] = *reinterpret_cast<const float4*>(p);
        else { const int gm = m0 + 4 * ((tid + 256 * i) % MQ), last = g.M - 1 - gm;   // ragged edge: gm <= M - 1 < gm + 3
               ra[i] = make_float4(p[0], p[min(1, last)], p[min(2, last)], p[min(3, last)]); }
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (B_KC) {
        rb[i] = (k0 + kb_[i] < g.K) ? *reinterpret_cast<const float4*>(pb[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int k = k0 + kb_[i];
        const float* p = pb[i] + (long long)min(k, g.K - 1) * g.sbk;
        if (k >= g.K || fb[i] == 0) rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (fb[i] == 1) rb[i] = *reinterpret_cast<const float4*>(p);
        else { constexpr int NQ = BN_ / 4; const int idx = tid + 256 * i, gn = n0 + 4 * (idx - (idx / NQ) * NQ), last = g.N - 1 - gn;
               rb[i] = make_float4(p[0], p[min(1, last)], p[min(2, last)], p[min(3, last)]); }
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      if (A_KC) { const int row = (tid >> 2) + 64 * i, kq = tid & 3; *reinterpret_cast<float4*>(&As[row * A_LD + 4 * kq]) = ra[i]; }
      else { const int idx = tid + 256 * i, k = idx / MQ, mq = idx - k * MQ; *reinterpret_cast<float4*>(&As[k * A_LD + 4 * mq]) = ra[i]; }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      if (B_KC) { const int row = (tid >> 2) + 64 * i, kq = tid & 3; *reinterpret_cast<float4*>(&Bs[row * B_LD + 4 * kq]) = rb[i]; }
      else { constexpr int NQ = BN_ / 4; const int idx = tid + 256 * i, k = idx / NQ, nq = idx - k * NQ;
             *reinterpret_cast<float4*>(&Bs[k * B_LD + 4 * nq]) = rb[i]; }
    }
  };

  floatx16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = floatx16{0};

  if (kt0 < kt1) load_tile(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    __syncthreads();                 // previous tile's fragment reads are done
    store_tile();
    __syncthreads();
    if (kt + 1 < kt1) load_tile(kt + 1);   // in flight during the MFMAs below
    float af[MI][8], bf[NI][8];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int row = wm * (FBM / 2) + mi * 32 + c;
      if (A_KC) {
        const float4 x = *reinterpret_cast<const float4*>(&As[row * A_LD + 8 * hf]);
        const float4 y = *reinterpret_cast<const float4*>(&As[row * A_LD + 8 * hf + 4]);
        af[mi][0] = x.x; af[mi][1] = x.y; af[mi][2] = x.z; af[mi][3] = x.w;
        af[mi][4] = y.x; af[mi][5] = y.y; af[mi][6] = y.z; af[mi][7] = y.w;
      } else {
#pragma unroll
        for (int st = 0; st < 8; ++st) af[mi][st] = As[(8 * hf + st) * A_LD + row];
      }
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = wn * (BN_ / 2) + ni * 32 + c;
      if (B_KC) {
        const float4 x = *reinterpret_cast<const float4*>(&Bs[col * B_LD + 8 * hf]);
        const float4 y = *reinterpret_cast<const float4*>(&Bs[col * B_LD + 8 * hf + 4]);
        bf[ni][0] = x.x; bf[ni][1] = x.y; bf[ni][2] = x.z; bf[ni][3] = x.w;
        bf[ni][4] = y.x; bf[ni][5] = y.y; bf[ni][6] = y.z; bf[ni][7] = y.w;
      } else {
#pragma unroll
        for (int st = 0; st < 8; ++st) bf[ni][st] = Bs[(8 * hf + st) * B_LD + col];
      }
    }
#pragma unroll
    for (int st = 0; st < 8; ++st)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma32(af[mi][st], bf[ni][st], acc[mi][ni]);
  }

  const float* bias = g.bias ? g.bias + b0 * g.sbias0 + b1 * g.sbias1 : nullptr;
  const float* res = g.residual ? g.residual + b0 * g.sc0 + b1 * g.sc1 : nullptr;
  // Interior tiles of a non-atomic launch: no per-element bounds branches, and all 16 residual values of an accumulator are
  // loaded FIRST - with the branchy form below hipcc waits for each load before the next (16 dependent L2
  // round trips per accumulator: +7 us on a 20 us launch of the Nystrom pseudo-inverse products, tests/diag_smallgemm.py).
  if (!g.atomic && m0 + FBM <= g.M && n0 + BN_ <= g.N) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * (BN_ / 2) + ni * 32 + c;
        const int mb = m0 + wm * (FBM / 2) + mi * 32;
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = res ? res[(long long)(mb + acc_row(r, hf)) * g.ldr + n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + acc_row(r, hf);
          float v = g.alpha * acc[mi][ni][r];
          if (bias) v += (g.bias_mode == 2) ? bias[(m / g.rows_per_bias) * g.bias_ld + n] : bias[n];
          v = apply_act(v, g.act);
          C[(long long)m * g.ldc + n] = fmaf(g.beta, rv[r], v);
        }
      }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * (BN_ / 2) + ni * 32 + c;
      if (n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (FBM / 2) + mi * 32 + acc_row(r, hf);
        if (m >= g.M) continue;
        float v = g.alpha * acc[mi][ni][r];
        if (g.atomic) {
          if (bias && ks == 0) v += (g.bias_mode == 2) ? bias[(m / g.rows_per_bias) * g.bias_ld + n] : bias[n];
          atomicAdd(&C[m * g.ldc + n], v);
        } else {
          if (bias) v += (g.bias_mode == 2) ? bias[(m / g.rows_per_bias) * g.bias_ld + n] : bias[n];
          v = apply_act(v, g.act);
          if (res) v = fmaf(g.beta, res[m * g.ldr + n], v);
          C[m * g.ldc + n] = v;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// fp16 helpers of the single-term fp16 form: four fp32 -> four fp16 (RNE), and the f16 MFMA on fragments held as 16-bit blobs
__device__ __forceinline__ uint2v f16_pack4(const float4 v) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const float2v a = {v.x, v.y}, b = {v.z, v.w};
  return (uint2v){__builtin_bit_cast(unsigned, __builtin_convertvector(a, h2)), __builtin_bit_cast(unsigned, __builtin_convertvector(b, h2))};
}
__device__ __forceinline__ floatx16 mfma16h(bf16x8 a, bf16x8 b, floatx16 c) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// Split-bf16 path ("bf3"): the same fp32 product on the 16-bit matrix pipe.  Every operand element is split into three
// bf16 terms when its tile is staged (x = h + m + l to 2^-24, fp32's exponent range: no scaling, no overflow) and six of
// the nine cross products are kept (m m, l h, h l, m h, h m, h h; what is dropped is <= 2^-23 |a||b| per product, i.e.
// fp32-grade results).  A 32x32x16 block costs 6 MFMAs of 32 cycles instead of 8 fp32 MFMAs of 64: 2.7x fewer
// matrix-pipe cycles, and - since MFMA and vector time of a SIMD add up on gfx950 - that is what the GEMM's time is made
// of.  Tile 128 x BN x 32, 2 x 2 waves, float4 global loads staged through registers one K-tile ahead.
// LDS holds three bf16 planes per operand:
//   k-contiguous operand   [row][32 k + 8]   80-byte rows: the 8 k-values of an MFMA lane are one ds_read_b128
//   row-contiguous operand [k][rows + 32]    as it comes from memory (plain 8-byte stores); the MFMA fragment is gathered
//                                            by two ds_read_b64_tr_b16 (hardware transpose, tests/microbench/tr_probe.hip)
// ------------------------------------------------------------------------------------------------
// TERMS = 1 ("bf1"): only the leading bf16 term of each operand and one product - the plain bf16 matrix-pipe GEMM with fp32
// storage and fp32 accumulation (8 mantissa bits per operand element), used by the 16-bit compute mode of the Nystrom block.
// F16 (with TERMS = 1, "f1"): the single term is fp16 instead of bf16 - 11 operand mantissa bits, fp16's exponent range: for
// forward-range operands only (activations and weights of the fp16 compute modes; their gradient products use bf1).
template <int BN_, bool A_KC, bool B_KC, int TERMS = 3, int FBM = 128, bool F16 = false>
__global__ __launch_bounds__(256, 2) void gemm_bf3_kernel(GemmArgs g) {
  static_assert(!F16 || TERMS == 1, "the fp16 form is single-term");
  constexpr int FBK = 32, NI = BN_ / 64, MI = FBM / 64;        // MI x NI 32x32 blocks per wave; FBM = 64: the small-problem tile
  constexpr int A_LD = A_KC ? (FBK + 8) : (FBM + 32);          // halves per row of a plane
  constexpr int B_LD = B_KC ? (FBK + 8) : (BN_ + 32);
  constexpr int A_PLANE = (A_KC ? FBM : FBK) * A_LD, B_PLANE = (B_KC ? BN_ : FBK) * B_LD;
  constexpr int NA = FBM / 32, NB = BN_ / 32, MQ = FBM / 4;    // float4 per thread per K-tile; float4 per k-row of a row-contiguous A tile
  __shared__ __attribute__((aligned(16))) __bf16 As[TERMS * A_PLANE];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[TERMS * B_PLANE];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int zb = blockIdx.z / g.splitk, ks = blockIdx.z - zb * g.splitk;
  const int b0 = zb / g.nb1, b1 = zb - b0 * g.nb1;
  const float* A = g.A + b0 * g.sa0 + b1 * g.sa1;
  const float* B = g.B + b0 * g.sb0 + b1 * g.sb1;
  float* C = g.C + b0 * g.sc0 + b1 * g.sc1;
  const int m0 = (g.swap_xy ? blockIdx.y : blockIdx.x) * FBM, n0 = (g.swap_xy ? blockIdx.x : blockIdx.y) * BN_;
  const int ktiles = (g.K + FBK - 1) / FBK;                    // a K tail is zero-filled at load time
  const int tps = (ktiles + g.splitk - 1) / g.splitk;
  const int kt0 = ks * tps, kt1 = min(ktiles, kt0 + tps);

  float4 ra[NA], rb[NB];
  auto load_tile = [&](int kt) {
    const int k0 = kt * FBK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + 256 * i;
      if (A_KC) {
        const int row = idx >> 3, kq = idx & 7;
        const long long gm = min(m0 + row, g.M - 1);
        ra[i] = (k0 + 4 * kq < g.K) ? *reinterpret_cast<const float4*>(A + gm * g.sam + k0 + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        const int k = idx / MQ, mq = idx - k * MQ;
        const int gm = m0 + 4 * mq;
        const float* p = A + (long long)min(k0 + k, g.K - 1) * g.sak;
        if (k0 + k >= g.K) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (gm >= g.M) ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);            // tile wider than the operand: nothing to read
        else if (gm + 3 < g.M) ra[i] = *reinterpret_cast<const float4*>(p + gm);
        else ra[i] = make_float4(p[min(gm, g.M - 1)], p[min(gm + 1, g.M - 1)], p[min(gm + 2, g.M - 1)], p[min(gm + 3, g.M - 1)]);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + 256 * i;
      if (B_KC) {
        const int row = idx >> 3, kq = idx & 7;
        const long long gn = min(n0 + row, g.N - 1);
        rb[i] = (k0 + 4 * kq < g.K) ? *reinterpret_cast<const float4*>(B + gn * g.sbn + k0 + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        constexpr int NQ = BN_ / 4;
        const int k = idx / NQ, nq = idx - k * NQ;
        const int gn = n0 + 4 * nq;
        const float* p = B + (long long)min(k0 + k, g.K - 1) * g.sbk;
        if (k0 + k >= g.K) rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (gn >= g.N) rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (gn + 3 < g.N) rb[i] = *reinterpret_cast<const float4*>(p + gn);
        else rb[i] = make_float4(p[min(gn, g.N - 1)], p[min(gn + 1, g.N - 1)], p[min(gn + 2, g.N - 1)], p[min(gn + 3, g.N - 1)]);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + 256 * i;
      const int off = A_KC ? ((idx >> 3) * A_LD + 4 * (idx & 7)) : ((idx / MQ) * A_LD + 4 * (idx % MQ));
      uint2v h, m, l;
      if (F16) h = f16_pack4(ra[i]); else split4_bf3(ra[i], h, m, l);
      *reinterpret_cast<uint2v*>(&As[off]) = h;
      if (TERMS == 3) {
        *reinterpret_cast<uint2v*>(&As[A_PLANE + off]) = m;
        *reinterpret_cast<uint2v*>(&As[2 * A_PLANE + off]) = l;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + 256 * i;
      constexpr int NQ = BN_ / 4;
      const int off = B_KC ? ((idx >> 3) * B_LD + 4 * (idx & 7)) : ((idx / NQ) * B_LD + 4 * (idx % NQ));
      uint2v h, m, l;
      if (F16) h = f16_pack4(rb[i]); else split4_bf3(rb[i], h, m, l);
      *reinterpret_cast<uint2v*>(&Bs[off]) = h;
      if (TERMS == 3) {
        *reinterpret_cast<uint2v*>(&Bs[B_PLANE + off]) = m;
        *reinterpret_cast<uint2v*>(&Bs[2 * B_PLANE + off]) = l;
      }
    }
  };

  floatx16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = floatx16{0};

  // transposed-read lane map: lane 4 q + p of each 16-lane group addresses row q, columns 4 p .. 4 p + 3 of a 4 x 16 block
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  if (kt0 < kt1) load_tile(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    __syncthreads();                 // previous tile's fragment reads are done
    store_tile();
    __syncthreads();
    if (kt + 1 < kt1) load_tile(kt + 1);   // in flight during the MFMAs below
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      bf16x8 af[MI][3], bf[NI][3];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int rbase = wm * (FBM / 2) + mi * 32;
#pragma unroll
        for (int p = 0; p < TERMS; ++p) {
          if (A_KC) af[mi][p] = *reinterpret_cast<const bf16x8*>(&As[p * A_PLANE + (rbase + c) * A_LD + 16 * kb + 8 * hf]);
          else {
            const __bf16* q0 = &As[p * A_PLANE + (16 * kb + 8 * hf + trq) * A_LD + rbase + trc];
            af[mi][p] = lds_frag_tr(q0, q0 + 4 * A_LD);
          }
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int cbase = wn * (BN_ / 2) + ni * 32;
#pragma unroll
        for (int p = 0; p < TERMS; ++p) {
          if (B_KC) bf[ni][p] = *reinterpret_cast<const bf16x8*>(&Bs[p * B_PLANE + (cbase + c) * B_LD + 16 * kb + 8 * hf]);
          else {
            const __bf16* q0 = &Bs[p * B_PLANE + (16 * kb + 8 * hf + trq) * B_LD + cbase + trc];
            bf[ni][p] = lds_frag_tr(q0, q0 + 4 * B_LD);
          }
        }
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          floatx16 d = acc[mi][ni];
          if (TERMS == 3) {
            d = mfma16b(af[mi][1], bf[ni][1], d);     // smallest terms first
            d = mfma16b(af[mi][2], bf[ni][0], d);
            d = mfma16b(af[mi][0], bf[ni][2], d);
            d = mfma16b(af[mi][1], bf[ni][0], d);
            d = mfma16b(af[mi][0], bf[ni][1], d);
          }
          if (F16) d = mfma16h(af[mi][0], bf[ni][0], d);
          else d = mfma16b(af[mi][0], bf[ni][0], d);
          acc[mi][ni] = d;
        }
    }
  }

  const float* bias = g.bias ? g.bias + b0 * g.sbias0 + b1 * g.sbias1 : nullptr;
  const float* res = g.residual ? g.residual + b0 * g.sc0 + b1 * g.sc1 : nullptr;
  // Interior tiles of a non-atomic launch: no per-element bounds branches, and all 16 residual values of an accumulator are
  // loaded FIRST - with the branchy form below hipcc waits for each load before the next (16 dependent L2
  // round trips per accumulator: +7 us on a 20 us launch of the Nystrom pseudo-inverse products, tests/diag_smallgemm.py).
  if (!g.atomic && m0 + FBM <= g.M && n0 + BN_ <= g.N) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * (BN_ / 2) + ni * 32 + c;
        const int mb = m0 + wm * (FBM / 2) + mi * 32;
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = res ? res[(long long)(mb + acc_row(r, hf)) * g.ldr + n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + acc_row(r, hf);
          float v = g.alpha * acc[mi][ni][r];
          if (bias) v += (g.bias_mode == 2) ? bias[(m / g.rows_per_bias) * g.bias_ld + n] : bias[n];
          v = apply_act(v, g.act);
          C[(long long)m * g.ldc + n] = fmaf(g.beta, rv[r], v);
        }
      }
    return;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * (BN_ / 2) + ni * 32 + c;
      if (n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (FBM / 2) + mi * 32 + acc_row(r, hf);
        if (m >= g.M) continue;
        float v = g.alpha * acc[mi][ni][r];
        if (g.atomic) {
          if (bias && ks == 0) v += (g.bias_mode == 2) ? bias[(m / g.rows_per_bias) * g.bias_ld + n] : bias[n];
          atomicAdd(&C[m * g.ldc + n], v);
        } else {
          if (bias) v += (g.bias_mode == 2) ? bias[(m / g.rows_per_bias) * g.bias_ld + n] : bias[n];
          v = apply_act(v, g.act);
          if (res) v = fmaf(g.beta, res[m * g.ldr + n], v);
          C[m * g.ldc + n] = v;
        }
      }
    }
}

}  // namespace

static std::atomic<int> g_force_generic{0};   // test hook: route everything through the generic kernel
static std::atomic<int> g_mode{0};            // 0: automatic, 1: fp32-MFMA tiled kernel only, 2: split-bf16 kernel wherever it applies,
                                  // 3: single-term bf16 kernel wherever it applies (16-bit compute mode: 8-bit operand mantissas)
                                  // 4: single-term fp16 kernel (11-bit operand mantissas, fp16's range: forward-range operands only)
extern "C" void smml_gemm_force_generic(int on) { g_force_generic = on; }
extern "C" void smml_gemm_set_mode(int mode) { g_mode = mode; }
extern "C" int smml_gemm_get_mode(void) { return g_mode; }
static std::atomic<int> g_small_tile{0};      // 0: automatic, 1: never use the 64-row tile, 2: use it wherever it applies (measurement hooks)
extern "C" void smml_gemm_set_small_tile(int v) { g_small_tile = v; }

extern "C" int smml_gemm_f32(const float* A, const float* B, float* C, const float* bias, const float* residual,
                             int M, int N, int K, long long sam, long long sak, long long sbk, long long sbn,
                             long long ldc, long long ldr, int nb0, int nb1, long long sa0, long long sa1,
                             long long sb0, long long sb1, long long sc0, long long sc1, long long sbias0,
                             long long sbias1, int bias_mode, int rows_per_bias, long long bias_ld, int act,
                             int splitk, int accumulate, float alpha, float beta, void* stream) {
  SMML_REQUIRE(A && B && C, "smml_gemm_f32: null operand");
  SMML_REQUIRE(M > 0 && N > 0 && K > 0 && nb0 > 0 && nb1 > 0, "smml_gemm_f32: non-positive size (M=%d N=%d K=%d)", M, N, K);
  SMML_REQUIRE(splitk >= 1, "smml_gemm_f32: splitk must be >= 1");
  const int atomic = (splitk > 1 || accumulate) ? 1 : 0;
  SMML_REQUIRE(!(atomic && (act != 0 || residual)), "smml_gemm_f32: split-K / accumulate excludes activation/residual");
  SMML_REQUIRE(bias_mode >= 0 && bias_mode <= 2, "smml_gemm_f32: bad bias_mode %d", bias_mode);
  SMML_REQUIRE(bias_mode != 2 || rows_per_bias > 0, "smml_gemm_f32: rows_per_bias must be positive");
  SMML_REQUIRE(act >= 0 && act <= 2, "smml_gemm_f32: bad activation %d", act);
  const long long gz = (long long)nb0 * nb1 * splitk;
  const long long gy = (N + BN - 1) / BN, gx = (M + BM - 1) / BM;
  const int swap_xy = gy > 65535;
  SMML_REQUIRE(gz <= 65535 && (swap_xy ? gx : gy) <= 65535,
               "smml_gemm_f32: grid too large (batch*splitk=%lld, row tiles=%lld, col tiles=%lld)", gz, gx, gy);
  GemmArgs g{A, B, C, bias_mode ? bias : nullptr, residual, M, N, K, sam, sak, sbk, sbn, ldc, ldr, nb0, nb1,
             sa0, sa1, sb0, sb1, sc0, sc1, sbias0, sbias1, bias_mode, rows_per_bias > 0 ? rows_per_bias : 1,
             bias_ld, act, splitk, alpha, beta, atomic, swap_xy};
  SMML_REQUIRE(!bias_mode || bias, "smml_gemm_f32: bias_mode set but bias is null");
  // fast path: aligned operands with a unit stride on k (then K a multiple of 4: float4 loads along k) or on the row
  // index (any K); the K tail of the last 16-wide tile is zero-filled
  auto al4 = [](long long v) { return (v & 3) == 0; };
  auto al16 = [](const void* p) { return (((size_t)p) & 15) == 0; };
  const bool a_kc = (sak == 1), a_mc = (sam == 1 && sak != 1);
  const bool b_kc = (sbk == 1 && sbn != 1), b_nc = (sbn == 1);
  const bool a_ok = al16(A) && al4(sa0) && al4(sa1) && ((a_kc && al4(sam)) || (a_mc && al4(sak)));
  const bool b_ok = al16(B) && al4(sb0) && al4(sb1) && ((b_kc && al4(sbn)) || (b_nc && al4(sbk)));
  const bool k_ok = (!a_kc && !b_kc) || (K % 4) == 0;
  if (a_ok && b_ok && k_ok && !g_force_generic) {
    // 128-wide column tiles only when they still fill the chip a few times over (2 workgroups per CU resident)
    const int bn = (N > 64 && gx * ((N + 127) / 128) * gz >= 1024) ? 128 : 64;
    const long long fy = (N + bn - 1) / bn;
    // 64-row tiles for launches whose 128 x 64 tiling leaves the chip short of two workgroups per CU (the 32 x 256^3 products of
    // the Nystrom pseudo-inverse: 256 workgroups, one wave per SIMD, every global-load latency exposed): twice the workgroups,
    // half the work each
    const bool small = bn == 64 && M > 64 && (g_small_tile == 2 || (g_small_tile == 0 && gx * fy * gz < 512));
    const long long fx = small ? (M + 63) / 64 : gx;
    const int fswap = fy > 65535;
    SMML_REQUIRE((fswap ? fx : fy) <= 65535, "smml_gemm_f32: grid too large");
    g.swap_xy = fswap;
    dim3 grid((unsigned)(fswap ? fy : fx), (unsigned)(fswap ? fx : fy), (unsigned)gz);
    hipStream_t st = (hipStream_t)stream;
    // Measured: in isolation (tests/bench_gemm.py) the split-bf16 kernel beats the fp32 MFMA kernel wherever K >= 128 and the
    // tile is not a 64-column sliver (107 vs 92 TF on the 80 000 x 128 x 512 projections of the headline step, 150 vs 118
    // TF on the Nystrom qkv product, 173 vs 124 TF at 4096^3) - but inside the training step, A/B on one box
    // (SMML_GEMM_MODE=1 vs automatic with that wider rule), the step is not faster (17.2 / 17.5 vs 17.6 / 17.5 ms) and
    // the errors against the oracle grow 2x.  So the automatic choice keeps it for large square-ish products only.
    const bool bf3 = (g_mode == 2) || (g_mode == 0 && K >= 2048 && M >= 1024 && N >= 1024);
    const bool bf1 = (g_mode == 3), f1 = (g_mode == 4);
#define SMML_FAST(BNV, AK, BK2)                                                                   \
  do {                                                                                            \
    if (f1) hipLaunchKernelGGL((gemm_bf3_kernel<BNV, AK, BK2, 1, 128, true>), grid, dim3(256), 0, st, g); \
    else if (bf1) hipLaunchKernelGGL((gemm_bf3_kernel<BNV, AK, BK2, 1>), grid, dim3(256), 0, st, g);   \
    else if (bf3) hipLaunchKernelGGL((gemm_bf3_kernel<BNV, AK, BK2>), grid, dim3(256), 0, st, g); \
    else hipLaunchKernelGGL((gemm_f32_fast_kernel<BNV, AK, BK2>), grid, dim3(256), 0, st, g);     \
  } while (0)
#define SMML_SMALL(AK, BK2)                                                                          \
  do {                                                                                               \
    if (f1) hipLaunchKernelGGL((gemm_bf3_kernel<64, AK, BK2, 1, 64, true>), grid, dim3(256), 0, st, g);   \
    else if (bf1) hipLaunchKernelGGL((gemm_bf3_kernel<64, AK, BK2, 1, 64>), grid, dim3(256), 0, st, g);   \
    else if (bf3) hipLaunchKernelGGL((gemm_bf3_kernel<64, AK, BK2, 3, 64>), grid, dim3(256), 0, st, g); \
    else hipLaunchKernelGGL((gemm_f32_fast_kernel<64, AK, BK2, 64>), grid, dim3(256), 0, st, g);     \
  } while (0)
    if (small) {
      if (a_kc && b_kc) SMML_SMALL(true, true); else if (a_kc) SMML_SMALL(true, false);
      else if (b_kc) SMML_SMALL(false, true); else SMML_SMALL(false, false);
    } else if (bn == 128) {
      if (a_kc && b_kc) SMML_FAST(128, true, true); else if (a_kc) SMML_FAST(128, true, false);
      else if (b_kc) SMML_FAST(128, false, true); else SMML_FAST(128, false, false);
    } else {
      if (a_kc && b_kc) SMML_FAST(64, true, true); else if (a_kc) SMML_FAST(64, true, false);
      else if (b_kc) SMML_FAST(64, false, true); else SMML_FAST(64, false, false);
    }
#undef SMML_FAST
#undef SMML_SMALL
    SMML_LAUNCH_CHECK("smml_gemm_f32/fast");
    return SMML_OK;
  }
  dim3 grid((unsigned)(swap_xy ? gy : gx), (unsigned)(swap_xy ? gx : gy), (unsigned)gz);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
  SMML_LAUNCH_CHECK("smml_gemm_f32");
  return SMML_OK;
}
