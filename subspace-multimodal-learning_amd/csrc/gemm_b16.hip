// bf16-storage GEMM on the 16-bit matrix pipe of gfx950 (v_mfma_f32_32x32x16_bf16): both operands are bf16 IN MEMORY, products accumulate
// in fp32, the result is written as bf16 or fp32.  This is the projection GEMM of the Nystrom block's bf16 compute mode
// (models/NystromAttention.py:88 to_qkv, :147 to_out, and their backward): with bf16 bags (BASELINE configs 2 / 4) the activations never
// exist in fp32, so the GEMM moves half the bytes of the fp32-storage kernels in gemm.hip and spends no vector instructions on
// conversions (the single-term mode of gemm_bf3_kernel converts every operand element when its tile is staged: ~100 vector instructions
// per 8 MFMAs, which is what held it at ~370 TFLOP/s).
//
//   trans = 0 ("NT")   C[M, N] = A[M, K] . B[N, K]^T      both operands k-contiguous          (y = x W^T;  dx = dy (W^T)^T with W^T materialised)
//   trans = 1 ("TN")   C[M, N] = A[K, M]^T . B[K, N]      both operands row-contiguous, K outer (dW = dy^T x), split-K over gridDim.z
//
// Tile 128 x 128 x 64, 2 x 2 waves of 64 x 64 (2 x 2 accumulators of 32 x 32), global loads of 16 bytes staged through registers one
// K-tile ahead (written to LDS after the barrier that ends the previous tile's reads), LDS images as in gemm.hip:
//   k-contiguous operand   [row][64 k + 8]    144-byte rows: a lane's 8 k-values of an MFMA are one conflict-free ds_read_b128
//   row-contiguous operand [k][128 rows + 32] 320-byte rows as they come from memory; fragments by ds_read_b64_tr_b16
// Workgroup ids are remapped so that each XCD walks a contiguous range of tiles (column tiles fastest): the column tiles of one row panel
// share the panel through one L2 instead of eight.
#include <algorithm>
#include <atomic>     // process-wide measurement switches (set once from the environment or a test hook): plain atomics, no launch state
#include <cmath>
#include <cstdlib>
#include "smml_common.h"

namespace {

constexpr int GM = 128, GN = 128, GK = 64;
constexpr int KC_LD = GK + 8;          // halves per row, k-contiguous image
constexpr int RC_LD = GM + 32;         // halves per k-row, row-contiguous image (GM == GN)
constexpr int KC_PLANE = GM * KC_LD;   // 9216 halves = 18 KB
constexpr int RC_PLANE = GK * RC_LD;   // 10240 halves = 20 KB

struct B16Args {
  const __bf16* A; const __bf16* B; void* C; const float* bias;
  int M, N, K;
  long long lda, ldb, ldc;
  int splitk, tiles_m, tiles_n;
  int atomic;                    // add the result into C (split-K slices, batches that share one output)
  int slice_major, nb;           // 1-D grid in which the tiles of one (batch item, K slice) sit on ONE XCD (see the kernel)
  long long sa, sb, sc;          // batch strides in elements (gridDim.z problems; sc = 0 with split-K: the batches add up in one output)
};

#ifndef SMML_B16_MINBLOCKS
#define SMML_B16_MINBLOCKS 2
#endif
template <bool TN, bool OUT_BF16>
__global__ __launch_bounds__(256, SMML_B16_MINBLOCKS) void gemm_b16_kernel(B16Args g) {
  constexpr int PLANE = TN ? RC_PLANE : KC_PLANE;
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * PLANE];
  __bf16* As = smem;
  __bf16* Bs = smem + PLANE;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: hardware hands consecutive workgroup ids to the 8 XCDs in turn; id -> (xcd, position) -> a contiguous tile range
  const int ntiles = g.tiles_m * g.tiles_n;
  int tile, ks, bz;
  if (g.slice_major) {
    // Long reductions cut into slices (dW = dy^T x over 40 960 tokens): the ntiles workgroups of one slice walk the same K range in step,
    // so their working set is one K tile of each operand - if they share an L2.  Measured with the tiles of a slice spread over the 8
    // XCDs: 426 MB from HBM for 168 MB of operands (every XCD fetched every panel).  Here slice s lives on XCD s % 8: consecutive
    // workgroup ids go to the XCDs in turn, so id = 8 j + xcd runs tile j % ntiles of slice 8 (j / ntiles) + xcd.
    const int S = g.splitk * g.nb, id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int sl = (j / ntiles) * 8 + xcd;
    if (sl >= S) return;                                   // padding of the last group of eight slices (whole workgroup)
    tile = j - (j / ntiles) * ntiles;
    bz = sl / g.splitk; ks = sl - bz * g.splitk;
  } else {
    const int id = blockIdx.x;
    const int xcd = id & 7, pos = id >> 3, q = ntiles >> 3, r = ntiles & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    ks = blockIdx.y; bz = blockIdx.z;
  }
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * GM, n0 = tn * GN;
  g.A += (long long)bz * g.sa;
  g.B += (long long)bz * g.sb;
  g.C = OUT_BF16 ? (void*)(reinterpret_cast<__bf16*>(g.C) + (long long)bz * g.sc) : (void*)(reinterpret_cast<float*>(g.C) + (long long)bz * g.sc);
  const int ktiles = (g.K + GK - 1) / GK;
  const int tps = (ktiles + g.splitk - 1) / g.splitk;
  const int kt0 = ks * tps, kt1 = min(ktiles, kt0 + tps);

  uint4v ra[4], rb[4];
  const uint4v zero4 = {0u, 0u, 0u, 0u};
  auto load_tile = [&](int kt) {
    const int k0 = kt * GK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      if (!TN) {
        const int row = idx >> 3, kc = (idx & 7) * 8;          // 8 lanes cover one 128-byte row segment
        const long long am = min(m0 + row, g.M - 1), bn = min(n0 + row, g.N - 1);
        const bool ok = (k0 + kc) < g.K;                         // K is a multiple of 8: a chunk is whole or absent
        const int kk = ok ? (k0 + kc) : 0;
        const uint4v a = *reinterpret_cast<const uint4v*>(g.A + am * g.lda + kk);
        const uint4v b = *reinterpret_cast<const uint4v*>(g.B + bn * g.ldb + kk);
        ra[i] = ok ? a : zero4;
        rb[i] = ok ? b : zero4;
      } else {
        const int k = idx >> 4, rc = (idx & 15) * 8;             // 16 lanes cover one 256-byte k-row
        const bool kok = (k0 + k) < g.K;
        const long long kk = kok ? (k0 + k) : 0;
        const bool aok = kok && (m0 + rc) < g.M, bok = kok && (n0 + rc) < g.N;     // M, N multiples of 8
        const uint4v a = *reinterpret_cast<const uint4v*>(g.A + kk * g.lda + (aok ? (m0 + rc) : 0));
        const uint4v b = *reinterpret_cast<const uint4v*>(g.B + kk * g.ldb + (bok ? (n0 + rc) : 0));
        ra[i] = aok ? a : zero4;
        rb[i] = bok ? b : zero4;
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 256 * i;
      const int off = TN ? ((idx >> 4) * RC_LD + (idx & 15) * 8) : ((idx >> 3) * KC_LD + (idx & 7) * 8);
      *reinterpret_cast<uint4v*>(&As[off]) = ra[i];
      *reinterpret_cast<uint4v*>(&Bs[off]) = rb[i];
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = floatx16{0};
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  if (kt0 < kt1) load_tile(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    __syncthreads();                          // the previous tile's fragment reads are done
    store_tile();
    __syncthreads();
    if (kt + 1 < kt1) load_tile(kt + 1);      // in flight during the MFMAs below
#pragma unroll
    for (int kb = 0; kb < GK / 16; ++kb) {
      bf16x8 af[2], bf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int rbase = wm * 64 + mi * 32;
        if (!TN) af[mi] = *reinterpret_cast<const bf16x8*>(&As[(rbase + c) * KC_LD + 16 * kb + 8 * hf]);
        else {
          const __bf16* p = &As[(16 * kb + 8 * hf + trq) * RC_LD + rbase + trc];
          af[mi] = lds_frag_tr(p, p + 4 * RC_LD);
        }
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int cbase = wn * 64 + ni * 32;
        if (!TN) bf[ni] = *reinterpret_cast<const bf16x8*>(&Bs[(cbase + c) * KC_LD + 16 * kb + 8 * hf]);
        else {
          const __bf16* p = &Bs[(16 * kb + 8 * hf + trq) * RC_LD + cbase + trc];
          bf[ni] = lds_frag_tr(p, p + 4 * RC_LD);
        }
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)      // bf16 output: the transposed block (rows on lanes, 4 consecutive columns per register group)
          acc[mi][ni] = OUT_BF16 ? mfma16b(bf[ni], af[mi], acc[mi][ni]) : mfma16b(af[mi], bf[ni], acc[mi][ni]);
    }
  }

  if (OUT_BF16) {
    // Epilogue through LDS: each wave packs its 64 x 64 block into its own [64][72]-half image (rows on lanes: ds_write_b64 of 4 columns),
    // then stores whole 128-byte row segments (8 lanes x 16 bytes) - 8 store instructions per wave instead of 64 two-byte ones.
    __syncthreads();                                   // every wave is done with the operand images
    __bf16* img = smem + wave * (64 * KC_LD);          // 4 x 9216 bytes = the two operand planes of the NT form; TN planes are larger
    const int nb = n0 + wn * 64, mb = m0 + wm * 64;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int col = ni * 32 + 8 * gq + 4 * hf;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias && nb + col + 3 < g.N) bv = *reinterpret_cast<const float4*>(g.bias + nb + col);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const float2v lo = {acc[mi][ni][4 * gq] + bv.x, acc[mi][ni][4 * gq + 1] + bv.y};
          const float2v hi = {acc[mi][ni][4 * gq + 2] + bv.z, acc[mi][ni][4 * gq + 3] + bv.w};
          const uint2v pk = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2)),
                             __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2))};
          *reinterpret_cast<uint2v*>(&img[(mi * 32 + c) * KC_LD + col]) = pk;
        }
      }
    wave_lds_fence();
    __bf16* Cb = reinterpret_cast<__bf16*>(g.C);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = 8 * it + (lane >> 3), ch = (lane & 7) * 8;
      const uint4v v = *reinterpret_cast<const uint4v*>(&img[row * KC_LD + ch]);
      if (mb + row < g.M && nb + ch < g.N) *reinterpret_cast<uint4v*>(Cb + (long long)(mb + row) * g.ldc + nb + ch) = v;     // N % 8 == 0
    }
    return;
  }

  // epilogue: lane (c, hf) holds column n = .. + c, rows acc_row(r, hf) of each 32 x 32 block
  const bool interior = (m0 + GM <= g.M) && (n0 + GN <= g.N);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + c;
      const int mb = m0 + wm * 64 + mi * 32;
      if (!interior && n >= g.N) continue;
      const float bv = (g.bias && ks == 0 && (g.sc != 0 || bz == 0)) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + acc_row(r, hf);
        if (!interior && m >= g.M) continue;
        const float v = acc[mi][ni][r] + bv;
        if (g.atomic) atomicAdd(&reinterpret_cast<float*>(g.C)[(long long)m * g.ldc + n], v);
        else reinterpret_cast<float*>(g.C)[(long long)m * g.ldc + n] = v;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// The same product with a 256-row tile and eight waves (512 threads, one workgroup per CU): 256 x 256 (2 x 4 waves of 128 x 64) or
// 256 x 128 (4 x 2 waves of 64 x 64).  A 128 x 128 tile at full matrix rate asks the L2 -> CU path for 64 B/clk, all of it, and its
// row-contiguous (TN) form spends more LDS cycles on its transposed fragment reads than matrix cycles on its MFMAs; the 256 x 256 tile
// halves the bytes and the fragment reads per MFMA.  Same images, loads and epilogues as gemm_b16_kernel; the bf16 epilogue leaves
// through LDS 64 rows of a wave at a time.
// ------------------------------------------------------------------------------------------------
constexpr int LM = 256;
template <bool TN, bool OUT_BF16, int BN>
__global__ __launch_bounds__(512, 1) void gemm_b16_big_kernel(B16Args g) {
  constexpr int WM = (BN == 256) ? 2 : 4, WNW = 8 / WM;       // waves along M / N
  constexpr int TM = LM / WM, MI = TM / 32, NI = 2;            // wave tile TM x 64
  static_assert(BN / WNW == 64, "wave tiles are 64 columns wide");
  constexpr int A_LD = TN ? (LM + 32) : KC_LD, B_LD = TN ? (BN + 32) : KC_LD;
  constexpr int A_IMG = TN ? GK * A_LD : LM * A_LD, B_IMG = TN ? GK * B_LD : BN * B_LD;
  constexpr int EPI = 8 * 64 * KC_LD;                          // the bf16 epilogue: eight [64][72] images
  constexpr int SMEM = (A_IMG + B_IMG) > EPI ? (A_IMG + B_IMG) : EPI;
  __shared__ __attribute__((aligned(16))) __bf16 smem[SMEM];
  __bf16* As = smem;
  __bf16* Bs = smem + A_IMG;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wm = wave / WNW, wn = wave % WNW;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int id = blockIdx.x;
  const int xcd = id & 7, pos = id >> 3, q = ntiles >> 3, r = ntiles & 7;
  const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int m0 = tm * LM, n0 = tn * BN;
  const int ks = blockIdx.y;
  g.A += (long long)blockIdx.z * g.sa;
  g.B += (long long)blockIdx.z * g.sb;
  g.C = OUT_BF16 ? (void*)(reinterpret_cast<__bf16*>(g.C) + (long long)blockIdx.z * g.sc) : (void*)(reinterpret_cast<float*>(g.C) + (long long)blockIdx.z * g.sc);
  const int ktiles = (g.K + GK - 1) / GK;
  const int tps = (ktiles + g.splitk - 1) / g.splitk;
  const int kt0 = ks * tps, kt1 = min(ktiles, kt0 + tps);

  constexpr int NA = LM * GK / 8 / 512, NB = BN * GK / 8 / 512;      // 16-byte chunks per thread: 4 and 4 | 2
  uint4v ra[NA], rb[NB];
  const uint4v zero4 = {0u, 0u, 0u, 0u};
  auto load_tile = [&](int kt) {
    const int k0 = kt * GK;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + 512 * i;
      if (!TN) {
        const int row = idx >> 3, kc = (idx & 7) * 8;
        const long long am = min(m0 + row, g.M - 1);
        const bool ok = (k0 + kc) < g.K;
        const uint4v a = *reinterpret_cast<const uint4v*>(g.A + am * g.lda + (ok ? (k0 + kc) : 0));
        ra[i] = ok ? a : zero4;
      } else {
        const int k = idx >> 5, rc = (idx & 31) * 8;                  // 32 chunks per 256-row k-row
        const bool kok = (k0 + k) < g.K, aok = kok && (m0 + rc) < g.M;
        const uint4v a = *reinterpret_cast<const uint4v*>(g.A + (long long)(kok ? (k0 + k) : 0) * g.lda + (aok ? (m0 + rc) : 0));
        ra[i] = aok ? a : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + 512 * i;
      if (!TN) {
        const int row = idx >> 3, kc = (idx & 7) * 8;
        const long long bn = min(n0 + row, g.N - 1);
        const bool ok = (k0 + kc) < g.K;
        const uint4v b = *reinterpret_cast<const uint4v*>(g.B + bn * g.ldb + (ok ? (k0 + kc) : 0));
        rb[i] = ok ? b : zero4;
      } else {
        constexpr int CPR = BN / 8;                                   // chunks per k-row
        const int k = idx / CPR, rc = (idx % CPR) * 8;
        const bool kok = (k0 + k) < g.K, bok = kok && (n0 + rc) < g.N;
        const uint4v b = *reinterpret_cast<const uint4v*>(g.B + (long long)(kok ? (k0 + k) : 0) * g.ldb + (bok ? (n0 + rc) : 0));
        rb[i] = bok ? b : zero4;
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + 512 * i;
      *reinterpret_cast<uint4v*>(&As[TN ? ((idx >> 5) * A_LD + (idx & 31) * 8) : ((idx >> 3) * A_LD + (idx & 7) * 8)]) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = tid + 512 * i;
      constexpr int CPR = BN / 8;
      *reinterpret_cast<uint4v*>(&Bs[TN ? ((idx / CPR) * B_LD + (idx % CPR) * 8) : ((idx >> 3) * B_LD + (idx & 7) * 8)]) = rb[i];
    }
  };

  floatx16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = floatx16{0};
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

  if (kt0 < kt1) load_tile(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (kt + 1 < kt1) load_tile(kt + 1);
#pragma unroll
    for (int kb = 0; kb < GK / 16; ++kb) {
      bf16x8 af[MI], bf[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int rbase = wm * TM + mi * 32;
        if (!TN) af[mi] = *reinterpret_cast<const bf16x8*>(&As[(rbase + c) * A_LD + 16 * kb + 8 * hf]);
        else {
          const __bf16* p = &As[(16 * kb + 8 * hf + trq) * A_LD + rbase + trc];
          af[mi] = lds_frag_tr(p, p + 4 * A_LD);
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int cbase = wn * 64 + ni * 32;
        if (!TN) bf[ni] = *reinterpret_cast<const bf16x8*>(&Bs[(cbase + c) * B_LD + 16 * kb + 8 * hf]);
        else {
          const __bf16* p = &Bs[(16 * kb + 8 * hf + trq) * B_LD + cbase + trc];
          bf[ni] = lds_frag_tr(p, p + 4 * B_LD);
        }
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = OUT_BF16 ? mfma16b(bf[ni], af[mi], acc[mi][ni]) : mfma16b(af[mi], bf[ni], acc[mi][ni]);
    }
  }

  if (OUT_BF16) {
    __syncthreads();
    __bf16* img = smem + wave * (64 * KC_LD);
    const int nb = n0 + wn * 64;
    __bf16* Cb = reinterpret_cast<__bf16*>(g.C);
#pragma unroll
    for (int half = 0; half < MI / 2; ++half) {                       // 64 rows of the wave tile at a time
      const int mb = m0 + wm * TM + half * 64;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int col = ni * 32 + 8 * gq + 4 * hf;
          float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (g.bias && nb + col + 3 < g.N) bv = *reinterpret_cast<const float4*>(g.bias + nb + col);
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2) {
            const floatx16& a = acc[2 * half + m2][ni];
            const float2v lo = {a[4 * gq] + bv.x, a[4 * gq + 1] + bv.y};
            const float2v hi = {a[4 * gq + 2] + bv.z, a[4 * gq + 3] + bv.w};
            const uint2v pk = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2)),
                               __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2))};
            *reinterpret_cast<uint2v*>(&img[(m2 * 32 + c) * KC_LD + col]) = pk;
          }
        }
      wave_lds_fence();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = 8 * it + (lane >> 3), ch = (lane & 7) * 8;
        const uint4v v = *reinterpret_cast<const uint4v*>(&img[row * KC_LD + ch]);
        if (mb + row < g.M && nb + ch < g.N) *reinterpret_cast<uint4v*>(Cb + (long long)(mb + row) * g.ldc + nb + ch) = v;
      }
      wave_lds_fence();                                               // the image is rewritten by the next half
    }
    return;
  }
  const bool interior = (m0 + LM <= g.M) && (n0 + BN <= g.N);
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * 64 + ni * 32 + c;
      const int mb = m0 + wm * TM + mi * 32;
      if (!interior && n >= g.N) continue;
      const float bv = (g.bias && ks == 0 && (g.sc != 0 || blockIdx.z == 0)) ? g.bias[n] : 0.f;
#pragma unroll
      for (int r2 = 0; r2 < 16; ++r2) {
        const int m = mb + acc_row(r2, hf);
        if (!interior && m >= g.M) continue;
        const float v = acc[mi][ni][r2] + bv;
        if (g.atomic) atomicAdd(&reinterpret_cast<float*>(g.C)[(long long)m * g.ldc + n], v);
        else reinterpret_cast<float*>(g.C)[(long long)m * g.ldc + n] = v;
      }
    }
}

static std::atomic<int> g_b16_slice_major{1};   // measurement switch (smml_gemm_b16_set_slice_major): 0 = slices of a split reduction spread over the XCDs
static std::atomic<int> g_b16_tile{-1};      // -1: read SMML_B16_TILE; 0: automatic; 1: the 128 x 128 kernel only; 2: the 256-row kernel wherever it applies

}  // namespace

extern "C" void smml_gemm_b16_set_tile(int mode) { g_b16_tile = mode; }
extern "C" void smml_gemm_b16_set_slice_major(int on) { g_b16_slice_major = on; }

// C = A B^T (trans = 0: A [M, K], B [N, K], leading dimensions lda / ldb in elements) or C = A^T B (trans = 1: A [K, M], B [K, N]);
// A, B bf16; C bf16 (out_bf16 = 1, ldc in bf16 elements) or fp32; bias (fp32 [N], may be null) is added once.  splitk > 1: fp32 output only,
// the K range is cut into splitk slices whose partial products are ADDED to C atomically - the caller zeroes C first.
extern "C" int smml_gemm_b16_batched(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda,
                                     long long ldb, long long ldc, int trans, int out_bf16, int splitk, int nb, long long sa, long long sb,
                                     long long sc, void* stream);
extern "C" int smml_gemm_b16(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda,
                             long long ldb, long long ldc, int trans, int out_bf16, int splitk, void* stream) {
  return smml_gemm_b16_batched(A, B, C, bias, M, N, K, lda, ldb, ldc, trans, out_bf16, splitk, 1, 0, 0, 0, stream);
}

// nb problems of the same shape at element strides sa / sb / sc between them.  sc = 0 needs an fp32 output: the problems' products are then
// ADDED into the one (zeroed) output atomically, like split-K slices - dW = sum over bags of dy_bag^T x_bag with each bag's rows addressed in
// place.  This is how the front padding of a bag (models/NystromAttention.py:82) stays out of memory: a bag's n real rows are a batch item
// whose output lands pad rows into its n'-row block.
extern "C" int smml_gemm_b16_batched(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long long lda,
                                     long long ldb, long long ldc, int trans, int out_bf16, int splitk, int nb, long long sa, long long sb,
                                     long long sc, void* stream) {
  SMML_REQUIRE(A && B && C, "smml_gemm_b16: null operand");
  SMML_REQUIRE(M > 0 && N > 0 && K > 0, "smml_gemm_b16: non-positive size (M=%d N=%d K=%d)", M, N, K);
  SMML_REQUIRE(nb >= 1 && nb <= 65535 && (sa % 8) == 0 && (sb % 8) == 0 && (sc % 8) == 0, "smml_gemm_b16: bad batch (nb=%d; strides must be multiples of 8)", nb);
  SMML_REQUIRE(!(nb > 1 && sc == 0 && out_bf16), "smml_gemm_b16: batches that add up need an fp32 output");
  SMML_REQUIRE(splitk >= 0 && splitk <= 65535, "smml_gemm_b16: bad splitk %d", splitk);
  SMML_REQUIRE(!(splitk > 1 && out_bf16), "smml_gemm_b16: split-K accumulates in an fp32 output");
  auto al16 = [](const void* p) { return (((size_t)p) & 15) == 0; };
  SMML_REQUIRE(al16(A) && al16(B) && (lda % 8) == 0 && (ldb % 8) == 0, "smml_gemm_b16: operands must be 16-byte aligned with leading dimensions that are multiples of 8");
  if (!trans) SMML_REQUIRE((K % 8) == 0 && lda >= K && ldb >= K, "smml_gemm_b16: NT form needs K %% 8 == 0 and lda, ldb >= K (K=%d)", K);
  else SMML_REQUIRE((M % 8) == 0 && (N % 8) == 0 && lda >= M && ldb >= N, "smml_gemm_b16: TN form needs M, N %% 8 == 0 and lda >= M, ldb >= N (M=%d N=%d)", M, N);
  SMML_REQUIRE(ldc >= N, "smml_gemm_b16: ldc < N");
  if (out_bf16) SMML_REQUIRE((N % 8) == 0 && (ldc % 8) == 0 && al16(C), "smml_gemm_b16: a bf16 result needs N %% 8 == 0, ldc %% 8 == 0 and a 16-byte aligned C (N=%d)", N);
  if (g_b16_tile < 0) { const char* e = getenv("SMML_B16_TILE"); g_b16_tile = e ? atoi(e) : 0; }
  // Tile choice.  The 256-row tile runs one 512-thread workgroup per CU (256 slots), the 128 x 128 tile three 256-thread ones (768 slots);
  // what a launch gets out of either is (how full its last round of slots is) x (what the tile is worth per workgroup): measured
  // (profiles/r03_gemm_b16_times.txt) the big tile is worth 1.35 x on long reductions with k-contiguous operands (4096^3: 588 -> 826
  // TFLOP/s) and nothing on the Nystrom block's shapes - the K = 512 / 1536 projections are prologue and epilogue, and the dW products
  // (K = 40 960, split) gained from the split choice below, not from the tile (135 us either way; 178 with 21 slices of 128 x 128 tiles).
  const long long bigm = (M + LM - 1) / LM;
  const int big_bn = (N > 256 || N == 256) ? 256 : 128;
  const long long tiles_big = bigm * ((N + big_bn - 1) / big_bn);
  const long long tiles_small = ((long long)(M + GM - 1) / GM) * ((N + GN - 1) / GN);
  const int ktiles_all = (K + GK - 1) / GK;
  auto auto_split = [&](long long tiles, int slots) {                  // one round of slots, at least four K tiles per slice
    long long s = slots / std::max<long long>(1, tiles * nb);
    return (int)std::max<long long>(1, std::min<long long>(s, std::max(1, ktiles_all / 4)));
  };
  const bool can_big = M >= LM && N >= 128;
  bool big = false;
  if (g_b16_tile == 2) big = can_big;
  else if (g_b16_tile == 0 && can_big) {
    const int sb_ = splitk > 0 ? splitk : auto_split(tiles_big, 256), ss_ = splitk > 0 ? splitk : auto_split(tiles_small, 768);
    const double wb = (double)tiles_big * sb_ * nb, ws = (double)tiles_small * ss_ * nb;
    const double eff_b = wb / (std::ceil(wb / 256.0) * 256.0), eff_s = ws / (std::ceil(ws / 768.0) * 768.0);
    const double worth = (!trans && K >= 2048) ? 1.35 : 1.0;
    big = eff_b * worth > eff_s * 1.02;
  }
  if (splitk == 0) splitk = out_bf16 ? 1 : (big ? auto_split(tiles_big, 256) : auto_split(tiles_small, 768));
  if (big) {
    const long long tmb = bigm, tnb = (N + big_bn - 1) / big_bn;
    SMML_REQUIRE(tmb * tnb < (1LL << 31), "smml_gemm_b16: grid too large");
    const int atomic_b = (nb > 1 && sc == 0) ? 1 : 0;
    B16Args gb{reinterpret_cast<const __bf16*>(A), reinterpret_cast<const __bf16*>(B), C, bias, M, N, K, lda, ldb, ldc, splitk, (int)tmb, (int)tnb,
               (splitk > 1 || atomic_b) ? 1 : 0, 0, nb, sa, sb, sc};
    dim3 gridb((unsigned)(tmb * tnb), (unsigned)splitk, (unsigned)nb), blockb(512);
    hipStream_t stb = (hipStream_t)stream;
#define SMML_BIG(TNV, OB)                                                                                        \
  do {                                                                                                           \
    if (big_bn == 256) hipLaunchKernelGGL((gemm_b16_big_kernel<TNV, OB, 256>), gridb, blockb, 0, stb, gb);       \
    else hipLaunchKernelGGL((gemm_b16_big_kernel<TNV, OB, 128>), gridb, blockb, 0, stb, gb);                     \
  } while (0)
    if (trans) { if (out_bf16) SMML_BIG(true, true); else SMML_BIG(true, false); }
    else { if (out_bf16) SMML_BIG(false, true); else SMML_BIG(false, false); }
#undef SMML_BIG
    SMML_LAUNCH_CHECK("smml_gemm_b16/big");
    return SMML_OK;
  }
  const long long tm = (M + GM - 1) / GM, tn = (N + GN - 1) / GN;
  SMML_REQUIRE(tm * tn < (1LL << 31), "smml_gemm_b16: grid too large");
  // batches that add into one output take the atomic path like split-K slices: the kernel's "splitk > 1" test covers both
  const int atomic_batches = (nb > 1 && sc == 0) ? 1 : 0;
  // slices of a long reduction that add up in one output: one XCD per slice (see the kernel)
  const long long nslices = (long long)splitk * nb;
  const int slice_major = (g_b16_slice_major != 0 && !out_bf16 && (splitk > 1 || atomic_batches) && nslices >= 8 &&
                           ((nslices + 7) / 8) * 8 * tm * tn < (1LL << 31)) ? 1 : 0;
  B16Args g{reinterpret_cast<const __bf16*>(A), reinterpret_cast<const __bf16*>(B), C, bias, M, N, K, lda, ldb, ldc, splitk, (int)tm, (int)tn,
            (splitk > 1 || atomic_batches) ? 1 : 0, slice_major, nb, sa, sb, sc};
  dim3 grid((unsigned)(tm * tn), (unsigned)splitk, (unsigned)nb), block(256);
  if (slice_major) grid = dim3((unsigned)(((nslices + 7) / 8) * 8 * tm * tn), 1, 1);
  hipStream_t st = (hipStream_t)stream;
  if (trans) {
    if (out_bf16) hipLaunchKernelGGL((gemm_b16_kernel<true, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_b16_kernel<true, false>), grid, block, 0, st, g);
  } else {
    if (out_bf16) hipLaunchKernelGGL((gemm_b16_kernel<false, true>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_b16_kernel<false, false>), grid, block, 0, st, g);
  }
  SMML_LAUNCH_CHECK("smml_gemm_b16");
  return SMML_OK;
}
