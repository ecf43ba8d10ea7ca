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

}  // namespace

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
  dim3 grid((unsigned)(swap_xy ? gy : gx), (unsigned)(swap_xy ? gx : gy), (unsigned)gz);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, g);
  SMML_LAUNCH_CHECK("smml_gemm_f32");
  return SMML_OK;
}
