// 16-bit compute mode of the fused deformable cross-attention core (BASELINE config 4 names bf16; config 5 fp16): the same op
// sequence as deform_attn.hip -
//   models/DeformableAttention2D.py:284-312 (sim, rel_pos_bias, softmax, dropout, attn @ v) and :120-157 (CPB)
//   models/DeformableAttention1D.py:205-232 and :60-102
// with SINGLE-TERM 16-bit operands on the matrix pipe and 16-bit score / d-score storage, where deform_attn.hip spends two to six
// MFMAs per contraction block (and the vector instructions that split fp32 values into hi / mid / lo terms) on fp32-grade results.
//
// What is 16 bit (T = bf16 or fp16 for everything of forward range; gradient-range operands are always bf16 - fp32's exponents):
//   forward   q, k, v, the softmax'd probabilities and the hidden layer h1 of the position-bias MLP are rounded to T when they
//             become MFMA operands; W2 is one T term; the pre-softmax scores (incl. bias) are stored as fp16 [B, H, nst / 32, J, 32]
//             (both modes: forward range, 11 bits) and - in training - the forward's own softmax runs on the ROUNDED scores, so forward
//             and backward see one set of probabilities.  With dropout the keep decision rides in the stored score's lowest mantissa
//             bit (as in the fp32 path).
//   backward  K, V, Q, dO, P, dS as single bf16 terms; d scores stored as bf16; chain 2 of the position-bias backward with the
//             constant (W2 w3)^T as one fp16 term against the exact 0 / 1 mask operand; g = h1 . d bias as one bf16 term.
// What stays fp32: layer 1 of the position-bias MLP (the three-term bf16 product of deform_common.h - the SAME device function as
// the fp32 path, so the layer-1 ReLU decisions and their export are shared), every accumulator, the softmax statistics (running
// max, normaliser, log-sum-exp), delta = rowsum(dO . O), all parameter-gradient sums, q / k / v / out and their gradients in memory.
// Per (key, 32 queries) that is 3 + 8/32 MFMAs forward (7 + 24/32 in the fp32-grade path) and 8 backward (12).
#include "deform_common.h"
#include "deform16_types.h"
#include "cpb_regions.h"

// measurement knobs of this file (tests/build_variants.py)
#ifndef SMML16_FWD_WPS
#define SMML16_FWD_WPS 2      // waves per SIMD the forward is register-budgeted for
#endif
#ifndef SMML16_CPB_WPS
#define SMML16_CPB_WPS 2      // same, position-bias backward
#endif
#ifndef SMML16_FWD_PAIR
#define SMML16_FWD_PAIR 0     // 1: two keys per trip of the forward's position-bias loop (software pipelining left to the scheduler)
#endif
#ifndef SMML16_DP_MFMA
#define SMML16_DP_MFMA 1      // 1 (default; 0 = the fp32 path's vector form: measurement switch): d p = W1^T (m1 . d h1) of the position-bias backward (the d vs path) as one bf16 product on the matrix pipe (2 MFMAs
                              // + 8 conversions) instead of 16 packed FMAs + 8 LDS table reads: the kernel is vector-ISSUE bound (PMC: VALU issue
                              // cycles = kernel time, the matrix pipe 23 % busy), so an MFMA costs one issue slot.  A/B on one box, twice:
                              // 5.83 -> 5.50 and 5.91 -> 5.59 ms per launch (8 bags of 10 000 x 625 x 8 heads)
#endif
#ifndef SMML16_GATE_MUL
#define SMML16_GATE_MUL 0     // 1: the layer-1 ReLU gate of the backward as an exact 0 / 1 multiplier (clamped multiply) instead of compare + select
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// forward.  One wave = 32 queries on the lane axis (deform_attn.hip's mapping); K / V tiles are converted to T when staged
// (double-buffered: one barrier per tile, the next tile's loads in flight during the key loop), the wave's scaled Q tile lives in
// registers as four B fragments.
// ------------------------------------------------------------------------------------------------
// Tabulated position bias (TG > 0; section "table mode" at the end of this file): the bias MLP is a function of the 2 (1) signed-log
// offsets only - one function for every (bag, head of a group index, query, key) of a launch - so it is evaluated ONCE per launch on a
// TG^PD grid over [-pmax, pmax]^PD (by the caller: three small GEMMs) and every pair interpolates it (bi)linearly from LDS.
struct TabCfg {
  const float* tab;        // [o][TG^PD] fp32, point (i0, i1) at index i1 * TG + i0 <-> p = -pmax + i * h
  float invh, off, umax;   // u = p * invh + off (off = pmax * invh), clamped to [0, umax] (umax just below TG - 1)
};
template <int PD, int TG> constexpr int tab_cells() { return TG <= 0 ? 1 : (PD == 2 ? TG * TG : TG); }

template <int PDX, bool SAVE, typename T, int TG = 0>
__global__ __launch_bounds__(256, SMML16_FWD_WPS) void deform16_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ VS,
    const float* __restrict__ GQ, CpbParams cp, float* __restrict__ O, float* __restrict__ LSE, u16* __restrict__ LT,
    u16* __restrict__ MK, int N, int J, int H, int G, int NST, float scale, DropCfg dc_in, TabCfg tc = TabCfg{}) {
  constexpr int PD = PosCfg<PDX>::PD;
  constexpr bool RAW = PosCfg<PDX>::RAW;           // table modes (TG > 0) are built for the signed-log form only
  typedef typename Vec8<T>::type vec8;
  const DropCfg dc = drop_resolve(dc_in);
  __shared__ float tabl[tab_cells<PD, TG>()];                          // table mode: this head's bias table
  __shared__ __attribute__((aligned(16))) T Kp[2][KT * FRLD];          // K tile, row image (A operand of S^T)
  __shared__ __attribute__((aligned(16))) T Vp[2][KT * FTLD];          // V tile, read transposed (A operand of O^T)
  __shared__ float vsl[2][KT][2];                                      // sample positions of the tile's keys
  __shared__ float biasT[WAVES][KT][QT];                               // per-wave bias tile [key][query]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (QT * WAVES) + wave * QT;
  const int o = H / G, g = h / o, oi = h - g * o;
  const int HD = H * DH;
  const bool qvalid = (q0 + c) < N;
  const int qi = qvalid ? (q0 + c) : (N - 1);

  // scaled Q of this lane's query as the B operand of S^T = K . Q^T: k-step st holds d = 16 st + 8 hf + j
  vec8 qf[4];
  {
    const float* qp = Q + ((size_t)b * N + qi) * HD + h * DH + hf * 8;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const float4 t0 = *reinterpret_cast<const float4*>(qp + 16 * st), t1 = *reinterpret_cast<const float4*>(qp + 16 * st + 4);
      const float x8[8] = {t0.x * scale, t0.y * scale, t0.z * scale, t0.w * scale, t1.x * scale, t1.y * scale, t1.z * scale, t1.w * scale};
      qf[st] = cvt8<T>(x8);
    }
  }
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);      // transposed-read lane map
  const float gq0 = GQ[(size_t)qi * PD];
  const float gq1 = (PD == 2) ? GQ[(size_t)qi * PD + 1] : 0.f;

  // position-bias constants (layout notes: deform_attn.hip forward).  The chain runs on 2 h1 (relu2), so it delivers
  // 2 (W2 h1 + b2), and relu2 of that is 4 relu(.): b2 rides in doubled, w3 carries 1/4.
  float w3v[16];
  floatx16 b2acc, b1acc;
  bf16x8 a1;
  vec8 w2t[2];                       // W2 as ONE T term: lane (out = c, half hf), K-block kb, element j <-> in = acc_row(8 kb + j, hf)
  float b3h = 0.f;
  if constexpr (TG == 0) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int oc = acc_row(s, hf);
      b2acc[s] = 2.f * cp.b2[oc];
      w3v[s] = 0.25f * cp.w3[oi * CH + oc];
      b1acc[s] = cp.b1[oc];
    }
    a1 = cpb_l1_weights_q(cp.w1[c * PD], (PD == 2) ? cp.w1[c * PD + 1] : 0.f, hf);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[j] = cp.w2[c * CH + acc_row(8 * kb + j, hf)];
      w2t[kb] = cvt8<T>(wv);
    }
    b3h = (hf == 0) ? cp.b3[oi] : 0.f;
  } else {
    constexpr int NC = tab_cells<PD, TG>();
    const float* tsrc = tc.tab + (size_t)oi * NC;
    for (int i = tid; i < NC; i += 256) tabl[i] = tsrc[i];         // visible after the first tile's barrier
  }

  floatx16 oacc0 = {0}, oacc1 = {0};
  float m_run = -INFINITY, l_run = 0.f;
  const float* Kb = K + (size_t)b * J * HD + h * DH;
  const float* Vb = V + (size_t)b * J * HD + h * DH;
  const float* VSb = VS + (size_t)(b * G + g) * J * PD;
  u16* LTb = LT ? LT + ((size_t)(b * H + h) * NST + q0) * J : nullptr;              // this wave's [J][32] block of 16-bit scores
  u16* MKb = MK ? MK + (((size_t)(b * H + h) * NST + q0) * J) * 2 + hf * 32 : nullptr;   // [J][2][32] mask block, this lane half
  float big;                                                  // 2^100 in an SGPR (v_mul_f32 ... clamp takes no literal)
  asm("s_mov_b32 %0, 0x71800000" : "=s"(big));

  // staging map: thread -> keys (tid >> 4) and (tid >> 4) + 16, 4 consecutive d
  const int skey = tid >> 4, sd4 = (tid & 15) * 4;
  float4 kreg[2], vreg[2];
  float vsr0 = 0.f, vsr1 = 0.f;
  auto fetch = [&](int j0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = j0 + skey + 16 * i;
      kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f); vreg[i] = kreg[i];
      if (key < J) {
        kreg[i] = *reinterpret_cast<const float4*>(Kb + (size_t)key * HD + sd4);
        vreg[i] = *reinterpret_cast<const float4*>(Vb + (size_t)key * HD + sd4);
      }
    }
    if (tid < KT) {
      const int key = j0 + tid;
      vsr0 = (key < J) ? VSb[(size_t)key * PD] : 0.f;
      vsr1 = (PD == 2 && key < J) ? VSb[(size_t)key * PD + 1] : 0.f;
    }
  };
  fetch(0);

  const int ntiles = (J + KT - 1) / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int j0 = kt * KT, buf = kt & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = skey + 16 * i;
      *reinterpret_cast<uint2v*>(&Kp[buf][key * FRLD + sd4]) = pack4<T>(kreg[i]);
      *reinterpret_cast<uint2v*>(&Vp[buf][key * FTLD + sd4]) = pack4<T>(vreg[i]);
    }
    if (tid < KT) { vsl[buf][tid][0] = vsr0; vsl[buf][tid][1] = vsr1; }
    __syncthreads();        // buffer (kt & 1) was last read in iteration kt - 2, which every wave left before the previous barrier
    if (kt + 1 < ntiles) fetch(j0 + KT);

    // S^T[key, query] = K . (scale Q)^T
    floatx16 s = {0};
#pragma unroll
    for (int st = 0; st < 4; ++st)
      s = mma(*reinterpret_cast<const vec8*>(&Kp[buf][c * FRLD + 16 * st + 8 * hf]), qf[st], s);

    // continuous position bias: layer 1 (fp32-grade, deform_common.h) + one T-term layer 2 per key
    const int nk = min(KT, J - j0);
    // one key's chain: layer 1 (MFMA) -> ReLU, T -> layer 2 (two dependent MFMAs) -> ReLU, layer 3, mask bits.  SMML16_FWD_PAIR runs two
    // keys per trip so that the scheduler can fill one chain's MFMA latencies with the other's vector work (padded keys of a ragged tile
    // compute on zero positions; only their mask store is guarded)
    auto bias_chain = [&](int jj, bool store_mask) {
      const float p0 = pos_of<RAW>(gq0 - vsl[buf][jj][0]);
      const float p1 = (PD == 2) ? slog1p(gq1 - vsl[buf][jj][1]) : 0.f;
      floatx16 d = b2acc;
      const floatx16 xacc = cpb_layer1_q(a1, cpb_split_pos(p0, p1), hf, b1acc);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float hv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = relu2(xacc[8 * kb + j]);
        d = mma(w2t[kb], cvt8<T>(hv), d);
      }
      // layer 3: two packed-fp32 FMA chains; b3 rides in half 0's sum.  Mask bits as in the fp32 path (same bit layout).
      float2v ta = {b3h, 0.f}, tb = {0.f, 0.f};
      float mb0 = 0.f, mb1 = 0.f, mb2 = 0.f, mb3 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        const float2v ra = {relu2(d[r]), relu2(d[r + 1])}, rb = {relu2(d[r + 2]), relu2(d[r + 3])};
        ta = __builtin_elementwise_fma(ra, (float2v){w3v[r], w3v[r + 1]}, ta);
        tb = __builtin_elementwise_fma(rb, (float2v){w3v[r + 2], w3v[r + 3]}, tb);
        if (SAVE) {
          mb0 = fmaf(fminf(fmaxf(ra[0] * big, 0.f), 1.f), (float)(1u << ((13 + r) & 15)), mb0);
          mb1 = fmaf(fminf(fmaxf(ra[1] * big, 0.f), 1.f), (float)(1u << ((14 + r) & 15)), mb1);
          mb2 = fmaf(fminf(fmaxf(rb[0] * big, 0.f), 1.f), (float)(1u << ((15 + r) & 15)), mb2);
          mb3 = fmaf(fminf(fmaxf(rb[1] * big, 0.f), 1.f), (float)(1u << ((16 + r) & 15)), mb3);
        }
      }
      if (SAVE && store_mask) MKb[(size_t)(j0 + jj) * 64 + c] = (u16)(unsigned)((mb0 + mb1) + (mb2 + mb3));   // rows of padded query lanes exist
      ta += tb;
      biasT[wave][jj][c] = xhalf_sum(ta[0] + ta[1]);
    };
    if constexpr (TG > 0) {
      // table mode: the two lane halves take one key each (key jj + hf of the staged tile; a padded key computes on zero positions and
      // is masked below), four (two) LDS reads and one (bi)linear interpolation per pair
#pragma unroll 4
      for (int jj = 0; jj < nk; jj += 2) {
        const int key = jj + hf;
        const float u0 = __builtin_amdgcn_fmed3f(fmaf(slog1p(gq0 - vsl[buf][key][0]), tc.invh, tc.off), 0.f, tc.umax);
        const float f0 = __builtin_amdgcn_fractf(u0);
        if constexpr (PD == 2) {
          const float u1 = __builtin_amdgcn_fmed3f(fmaf(slog1p(gq1 - vsl[buf][key][1]), tc.invh, tc.off), 0.f, tc.umax);
          const float f1 = __builtin_amdgcn_fractf(u1);
          const int idx = __mul24((int)u1, TG) + (int)u0;
          const float t00 = tabl[idx], t10 = tabl[idx + 1], t01 = tabl[idx + TG], t11 = tabl[idx + TG + 1];
          const float lo = fmaf(f0, t10 - t00, t00), hi = fmaf(f0, t11 - t01, t01);
          biasT[wave][key][c] = fmaf(f1, hi - lo, lo);
        } else {
          const int idx = (int)u0;
          const float t0 = tabl[idx], t1 = tabl[idx + 1];
          biasT[wave][key][c] = fmaf(f0, t1 - t0, t0);
        }
      }
    } else {
#if SMML16_FWD_PAIR
      for (int jj = 0; jj < nk; jj += 2) {
        bias_chain(jj, true);
        bias_chain(jj + 1, jj + 1 < nk);          // jj + 1 <= 31: inside the staged tile
      }
#else
      for (int jj = 0; jj < nk; ++jj) bias_chain(jj, true);
#endif
    }
    wave_lds_fence();

    // bias add, key mask
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = acc_row(r, hf);
      s[r] = (key < nk) ? s[r] + biasT[wave][key][c] : -INFINITY;
    }
    unsigned keepbits = 0xFFFFu;
    if (dc.thresh) {
      const unsigned long long base2 = ((unsigned long long)(b * H + h) * N + qi) * ((J + 1) >> 1) + (j0 >> 1);
      keepbits = 0u;
#pragma unroll
      for (int r = 0; r < 16; r += 2) keepbits |= drop_keep2(dc, base2 + (acc_row(r, hf) >> 1)) << r;     // registers r, r + 1: keys 2 jp, 2 jp + 1
    }
    if (SAVE) {
      // the scores are rounded to T for storage and the forward's own softmax continues on the rounded (and, with dropout, stashed)
      // values: forward and backward agree on the probabilities.  Lanes past N write padding of their own tile.
      if (nk == KT) {                            // full tile (uniform): no per-key bounds
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          unsigned w = pack_score(s[r], s[r + 1]);
          if (dc.thresh) w = stash_keep16x2(w, keepbits >> r);
          const unsigned lo = w & 0xFFFFu, hi = w >> 16;
          LTb[(size_t)(j0 + acc_row(r, hf)) * 32 + c] = (u16)lo;
          LTb[(size_t)(j0 + acc_row(r + 1, hf)) * 32 + c] = (u16)hi;
          s[r] = score_of(lo); s[r + 1] = score_of(hi);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const unsigned w = pack_score(s[r], s[r + 1]);
          unsigned lo = w & 0xFFFFu, hi = w >> 16;
          const int k0 = acc_row(r, hf), k1 = acc_row(r + 1, hf);
          if (dc.thresh) {
            if (k0 < nk) lo = stash_keep16(lo, (keepbits >> r) & 1u);
            if (k1 < nk) hi = stash_keep16(hi, (keepbits >> (r + 1)) & 1u);
          }
          if (k0 < nk) { LTb[(size_t)(j0 + k0) * 32 + c] = (u16)lo; s[r] = score_of(lo); }
          if (k1 < nk) { LTb[(size_t)(j0 + k1) * 32 + c] = (u16)hi; s[r + 1] = score_of(hi); }
        }
      }
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    tmax = xhalf_max(tmax);
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = sexp(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = sexp(s[r] - m_new);
      psum += p;                                  // the normaliser sums the un-dropped probabilities
      s[r] = p;
    }
    if (dc.thresh) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= ((keepbits >> r) & 1u) ? dc.keep_scale : 0.f;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc0[r] *= alpha; oacc1[r] *= alpha; }

    // O^T[d, query] += V^T . P^T: accumulator registers 8 kb .. 8 kb + 7 of P^T are the B fragment of k-step kb
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float p8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) p8[j] = s[8 * kb + j];
      const vec8 pt = cvt8<T>(p8);
      const int ro = (16 * kb + 4 * hf + trq) * FTLD + trc;
      oacc0 = mma(frag_tr<T>(&Vp[buf][ro], &Vp[buf][ro + 8 * FTLD]), pt, oacc0);
      oacc1 = mma(frag_tr<T>(&Vp[buf][ro + 32], &Vp[buf][ro + 32 + 8 * FTLD]), pt, oacc1);
    }
  }

  l_run = xhalf_sum(l_run);
  const float inv = 1.f / l_run;
  if (qvalid) {
    float* op = O + ((size_t)b * N + qi) * HD + h * DH;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = 8 * rg + 4 * hf;
      *reinterpret_cast<float4*>(op + d) = make_float4(oacc0[4 * rg] * inv, oacc0[4 * rg + 1] * inv,
                                                       oacc0[4 * rg + 2] * inv, oacc0[4 * rg + 3] * inv);
      *reinterpret_cast<float4*>(op + 32 + d) = make_float4(oacc1[4 * rg] * inv, oacc1[4 * rg + 1] * inv,
                                                            oacc1[4 * rg + 2] * inv, oacc1[4 * rg + 3] * inv);
    }
    if (hf == 0) LSE[(size_t)(b * H + h) * N + qi] = m_run + logf(l_run);
  }
}

// ------------------------------------------------------------------------------------------------
// backward pass 1 (query owners): dS^T = P^T (dP^T - delta) -> bf16 d scores, dQ = scale * dS K.  dP = V dO^T runs in the forward's
// operand type T (fp16 mode: dO scaled per query by a power of two), dQ = dS K on bf16 terms.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void deform16_bwd_dq_kernel(
    const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ O, const float* __restrict__ dO,
    const float* __restrict__ LSE, const u16* __restrict__ LT, u16* __restrict__ dLT, float* __restrict__ dQ, int N, int J,
    int H, int NST, float scale, DropCfg dc_in, unsigned* __restrict__ AMAX = nullptr) {
  typedef typename Vec8<T>::type vec8;
  const DropCfg dc = drop_resolve(dc_in);
  float amax = 0.f;      // max |d scores| of this lane (region backward: scale of its fixed-point moment sums); AMAX may be null
  __shared__ __attribute__((aligned(16))) T Vp[2][KT * VBLD];          // V in the FORWARD's operand type (dP = V dO^T must see the V that made O)
  __shared__ __attribute__((aligned(16))) __bf16 Kp[2][KT * KBLD];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (QT * WAVES) + wave * QT;
  const int HD = H * DH;
  const bool qvalid = (q0 + c) < N;
  const int qi = qvalid ? (q0 + c) : (N - 1);

  // dO of this lane's query as the B operand of dP^T = V . dO^T (K-block kb holds d = 16 kb + 8 hf + j), rounded to T; in fp16 mode
  // the row is first scaled by a power of two 2^-e so that its largest element sits in [0.5, 1) (fp16 has 5 exponent bits: gradients
  // need a scale; per query it is exact and free - 2^e rides on the probabilities through the exponent bias below).
  // delta = rowsum(dO . O) is formed from the ROUNDED dO: dS = P (dP - delta) must cancel exactly where it does in exact arithmetic
  // (one key: dP = dO . V = dO . O), and sum_k dS_k stays at the rounding of the probabilities instead of that of dO (2^-9).
  vec8 dob[4];
  float delta = 0.f;
  int e2 = 0;
  {
    const size_t off = ((size_t)b * N + qi) * HD + h * DH + hf * 8;
    float4 t[8];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      t[2 * kb] = *reinterpret_cast<const float4*>(dO + off + 16 * kb); t[2 * kb + 1] = *reinterpret_cast<const float4*>(dO + off + 16 * kb + 4);
    }
    float inv = 1.f;
    if (sizeof(T) == 2 && !__is_same(T, __bf16)) {
      float amax = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) amax = fmaxf(fmaxf(amax, fmaxf(fabsf(t[i].x), fabsf(t[i].y))), fmaxf(fabsf(t[i].z), fabsf(t[i].w)));
      amax = xhalf_max(amax);
      const int eb = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFFu);          // biased exponent: amax in [2^(eb-127), 2^(eb-126))
      e2 = (amax > 0.f) ? min(max(eb - 126, -100), 100) : 0;
      inv = __builtin_bit_cast(float, (unsigned)(127 - e2) << 23);
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float4 t0 = t[2 * kb], t1 = t[2 * kb + 1];
      const float4 u0 = *reinterpret_cast<const float4*>(O + off + 16 * kb), u1 = *reinterpret_cast<const float4*>(O + off + 16 * kb + 4);
      const float x8[8] = {t0.x * inv, t0.y * inv, t0.z * inv, t0.w * inv, t1.x * inv, t1.y * inv, t1.z * inv, t1.w * inv};
      dob[kb] = cvt8<T>(x8);
      const uint4v w = __builtin_bit_cast(uint4v, dob[kb]);
      delta += tof<T>(w[0] & 0xFFFFu) * u0.x + tof<T>(w[0] >> 16) * u0.y + tof<T>(w[1] & 0xFFFFu) * u0.z + tof<T>(w[1] >> 16) * u0.w +
               tof<T>(w[2] & 0xFFFFu) * u1.x + tof<T>(w[2] >> 16) * u1.y + tof<T>(w[3] & 0xFFFFu) * u1.z + tof<T>(w[3] >> 16) * u1.w;
    }
  }
  delta = xhalf_sum(delta);                             // of the scaled row: dP below is scaled alike
#if SMML_FAST_MATH
  const float nl = prob_bias(LSE[(size_t)(b * H + h) * N + qi]) + (float)e2;                       // P 2^e = exp2(l log2e - lse log2e + e)
#else
  const float nl = prob_bias(LSE[(size_t)(b * H + h) * N + qi]) + (float)e2 * 0.6931471805599453f;
#endif

  floatx16 dq0 = {0}, dq1 = {0};
  const float* Kb = K + (size_t)b * J * HD + h * DH;
  const float* Vb = V + (size_t)b * J * HD + h * DH;
  const u16* LTb = LT + ((size_t)(b * H + h) * NST + q0) * J;
  u16* dLTb = dLT + ((size_t)(b * H + h) * NST + q0) * J;

  const int skey = tid >> 4, sd4 = (tid & 15) * 4;
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  float4 kreg[2], vreg[2];
  unsigned lt[16];
  auto fetch = [&](int j0, float4 (&kr)[2], float4 (&vr)[2], unsigned (&l)[16]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = j0 + skey + 16 * i;
      kr[i] = make_float4(0.f, 0.f, 0.f, 0.f); vr[i] = kr[i];
      if (key < J) {
        kr[i] = *reinterpret_cast<const float4*>(Kb + (size_t)key * HD + sd4);
        vr[i] = *reinterpret_cast<const float4*>(Vb + (size_t)key * HD + sd4);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = min(j0 + acc_row(r, hf), J - 1);     // clamped: always in bounds, masked at use
      l[r] = LTb[(size_t)key * 32 + c];
    }
  };
  fetch(0, kreg, vreg, lt);

  const int ntiles = (J + KT - 1) / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int j0 = kt * KT, buf = kt & 1;
    const int nk = min(KT, J - j0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = skey + 16 * i;
      *reinterpret_cast<uint2v*>(&Vp[buf][key * VBLD + sd4]) = pack4<T>(vreg[i]);
      *reinterpret_cast<uint2v*>(&Kp[buf][key * KBLD + sd4]) = pack4<__bf16>(kreg[i]);
    }
    __syncthreads();        // one barrier per tile (double buffer)
    unsigned ltc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ltc[r] = lt[r];
    if (kt + 1 < ntiles) fetch(j0 + KT, kreg, vreg, lt);

    // dP^T[key, query] = V . dO^T
    floatx16 dp = {0};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
      dp = mma(*reinterpret_cast<const vec8*>(&Vp[buf][c * VBLD + 16 * kb + 8 * hf]), dob[kb], dp);

    float ds[16];
    const bool interior = (nk == KT && q0 + QT <= N);      // uniform
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      float v2[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int key = acc_row(r + e, hf);
        float v = 0.f;
        if (interior || (key < nk && qvalid)) {
          const float p = prob_of(score_of(ltc[r + e]), nl);
          float dpr = dp[r + e];
          if (dc.thresh) dpr *= (ltc[r + e] & 1u) ? dc.keep_scale : 0.f;     // the forward's decision rides in the score's lowest bit
          v = p * (dpr - delta);
        }
        v2[e] = v;
        ds[r + e] = v;
        amax = fmaxf(amax, fabsf(v));
      }
      const unsigned w = pack2<__bf16>(v2[0], v2[1]);
      const int k0 = acc_row(r, hf), k1 = acc_row(r + 1, hf);
      if (interior || (k0 < nk && qvalid)) dLTb[(size_t)(j0 + k0) * 32 + c] = (u16)(w & 0xFFFFu);
      if (interior || (k1 < nk && qvalid)) dLTb[(size_t)(j0 + k1) * 32 + c] = (u16)(w >> 16);
    }
    // dQ^T[d, query] += K^T . dS^T
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float x8[8];
#pragma unroll
      for (int jx = 0; jx < 8; ++jx) x8[jx] = ds[8 * kb + jx];
      const bf16x8 sb = cvt8<__bf16>(x8);
      const int ro = (16 * kb + 4 * hf + trq) * KBLD + trc;
      dq0 = mfma16b(lds_frag_tr(&Kp[buf][ro], &Kp[buf][ro + 8 * KBLD]), sb, dq0);
      dq1 = mfma16b(lds_frag_tr(&Kp[buf][ro + 32], &Kp[buf][ro + 32 + 8 * KBLD]), sb, dq1);
    }
  }
  if (AMAX) {                                               // non-negative floats order like their bit patterns
    amax = wave_max_all(amax);
    if (lane == 0 && amax > 0.f) atomicMax(AMAX, __float_as_uint(amax));
  }
  if (qvalid) {
    float* qp = dQ + ((size_t)b * N + qi) * HD + h * DH;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = 8 * rg + 4 * hf;
      *reinterpret_cast<float4*>(qp + d) = make_float4(dq0[4 * rg] * scale, dq0[4 * rg + 1] * scale,
                                                       dq0[4 * rg + 2] * scale, dq0[4 * rg + 3] * scale);
      *reinterpret_cast<float4*>(qp + 32 + d) = make_float4(dq1[4 * rg] * scale, dq1[4 * rg + 1] * scale,
                                                            dq1[4 * rg + 2] * scale, dq1[4 * rg + 3] * scale);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward pass 2 (key owners): dV = P_dropped^T dO, dK = scale * dS^T Q (deform_attn.hip's mapping; single bf16 terms, the
// 16-bit scores / d scores are read as 8-byte runs of four consecutive queries).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void deform16_bwd_dkv_kernel(
    const float* __restrict__ Q, const float* __restrict__ dO, const float* __restrict__ LSE, const u16* __restrict__ LT,
    const u16* __restrict__ dLT, float* __restrict__ dKp, float* __restrict__ dVp, int N, int J, int H, int NST, int nkg,
    int tiles_per_part, int nparts, int Bn, DropCfg dc_in) {
  const DropCfg dc = drop_resolve(dc_in);
  __shared__ __attribute__((aligned(16))) __bf16 Qp[2][QT * QBLD];
  __shared__ __attribute__((aligned(16))) __bf16 dOp[2][QT * QBLD];
  __shared__ __attribute__((aligned(16))) float nls[2][QT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  // XCD-aware block order (deform_attn.hip): the key groups of one (query slice, head, bag) get ids that differ by multiples of 8
  const int nslices = nparts * H * Bn;
  const int chunk = blockIdx.x / (8 * nkg), rem = blockIdx.x - chunk * (8 * nkg);
  const int kg = rem >> 3, slice = chunk * 8 + (rem & 7);
  if (slice >= nslices) return;                         // padding of the last chunk (whole workgroup, before any barrier)
  const int part = slice % nparts, h = (slice / nparts) % H, b = slice / (nparts * H);
  const int j0 = kg * DKV_KEYS + wave * KT;
  const int HD = H * DH;
  const int nk = min(KT, J - j0);                       // <= 0: this wave has no keys (it still stages tiles)
  const bool kvalid = c < nk;
  const int key = min(j0 + c, J - 1);
  const u16* LTk = LT + (size_t)(b * H + h) * NST * J + (size_t)key * 32;
  const u16* dLTk = dLT + (size_t)(b * H + h) * NST * J + (size_t)key * 32;
  const float* LSEb = LSE + (size_t)(b * H + h) * N;

  const int nqt = (N + QT - 1) / QT;
  const int qt_begin = part * tiles_per_part, qt_end = min(qt_begin + tiles_per_part, nqt);
  const int srow = tid >> 4, sd4 = (tid & 15) * 4;
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  float4 qreg[2], doreg[2];
  uint2v ltr[4], dlr[4];
  float lsereg = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { ltr[i] = (uint2v){0u, 0u}; dlr[i] = ltr[i]; }
  qreg[0] = qreg[1] = doreg[0] = doreg[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int q0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int qrow = min(q0 + srow + 16 * i, N - 1);
      const size_t off = ((size_t)b * N + qrow) * HD + h * DH + sd4;
      qreg[i] = *reinterpret_cast<const float4*>(Q + off);
      doreg[i] = *reinterpret_cast<const float4*>(dO + off);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const size_t qq = (size_t)q0 * J + 8 * rg + 4 * hf;   // 4 consecutive queries of the tile: 8 bytes
      ltr[rg] = *reinterpret_cast<const uint2v*>(LTk + qq);
      dlr[rg] = *reinterpret_cast<const uint2v*>(dLTk + qq);
    }
    if (tid < QT) lsereg = LSEb[min(q0 + tid, N - 1)];
  };

  floatx16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
  if (qt_begin < qt_end) fetch(qt_begin * QT);
  for (int qt = qt_begin; qt < qt_end; ++qt) {
    const int q0 = qt * QT, buf = (qt - qt_begin) & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = (srow + 16 * i) * QBLD + sd4;
      *reinterpret_cast<uint2v*>(&Qp[buf][o]) = pack4<__bf16>(qreg[i]);
      *reinterpret_cast<uint2v*>(&dOp[buf][o]) = pack4<__bf16>(doreg[i]);
    }
    if (tid < QT) nls[buf][tid] = prob_bias(lsereg);
    __syncthreads();
    unsigned lv[16];
    float dsv[16], ls[16];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      lv[4 * rg + 0] = ltr[rg][0] & 0xFFFFu; lv[4 * rg + 1] = ltr[rg][0] >> 16;
      lv[4 * rg + 2] = ltr[rg][1] & 0xFFFFu; lv[4 * rg + 3] = ltr[rg][1] >> 16;
      dsv[4 * rg + 0] = bf_lo(dlr[rg][0]); dsv[4 * rg + 1] = bf_hi(dlr[rg][0]);
      dsv[4 * rg + 2] = bf_lo(dlr[rg][1]); dsv[4 * rg + 3] = bf_hi(dlr[rg][1]);
      const float4 t = *reinterpret_cast<const float4*>(&nls[buf][8 * rg + 4 * hf]);     // broadcast read
      ls[4 * rg + 0] = t.x; ls[4 * rg + 1] = t.y; ls[4 * rg + 2] = t.z; ls[4 * rg + 3] = t.w;
    }
    if (qt + 1 < qt_end) fetch(q0 + QT);
    if (nk > 0) {                                          // wave-uniform
      float p[16], ds[16];
      const bool interior = (nk == KT && q0 + QT <= N);    // uniform
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = interior || (kvalid && q0 + acc_row(r, hf) < N);
        float pv = ok ? prob_of(score_of(lv[r]), ls[r]) : 0.f;
        if (dc.thresh) pv *= (lv[r] & 1u) ? dc.keep_scale : 0.f;
        p[r] = pv;                                         // dV takes the dropped probabilities, dK the dS written by pass 1
        ds[r] = ok ? dsv[r] : 0.f;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float x8[8], y8[8];
#pragma unroll
        for (int jx = 0; jx < 8; ++jx) { x8[jx] = p[8 * kb + jx]; y8[jx] = ds[8 * kb + jx]; }
        const bf16x8 pb = cvt8<__bf16>(x8), sb = cvt8<__bf16>(y8);
        const int ro = (16 * kb + 4 * hf + trq) * QBLD + trc;
        dv0 = mfma16b(lds_frag_tr(&dOp[buf][ro], &dOp[buf][ro + 8 * QBLD]), pb, dv0);
        dk0 = mfma16b(lds_frag_tr(&Qp[buf][ro], &Qp[buf][ro + 8 * QBLD]), sb, dk0);
        dv1 = mfma16b(lds_frag_tr(&dOp[buf][ro + 32], &dOp[buf][ro + 32 + 8 * QBLD]), pb, dv1);
        dk1 = mfma16b(lds_frag_tr(&Qp[buf][ro + 32], &Qp[buf][ro + 32 + 8 * QBLD]), sb, dk1);
      }
    }
  }
  if (kvalid) {
    const size_t off = (((size_t)part * Bn + b) * J + (j0 + c)) * HD + h * DH;
    float* kp = dKp + off;
    float* vp = dVp + off;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = 8 * rg + 4 * hf;
      *reinterpret_cast<float4*>(kp + d) = make_float4(dk0[4 * rg], dk0[4 * rg + 1], dk0[4 * rg + 2], dk0[4 * rg + 3]);
      *reinterpret_cast<float4*>(kp + 32 + d) = make_float4(dk1[4 * rg], dk1[4 * rg + 1], dk1[4 * rg + 2], dk1[4 * rg + 3]);
      *reinterpret_cast<float4*>(vp + d) = make_float4(dv0[4 * rg], dv0[4 * rg + 1], dv0[4 * rg + 2], dv0[4 * rg + 3]);
      *reinterpret_cast<float4*>(vp + 32 + d) = make_float4(dv1[4 * rg], dv1[4 * rg + 1], dv1[4 * rg + 2], dv1[4 * rg + 3]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward of the continuous position bias, 16-bit mode: deform_attn.hip's cpb_bwd_kernel (both MFMA layouts, the forward's saved
// layer-2 ReLU bits rotated straight into an fp16 operand, per-workgroup slabs added in a fixed order) with d bias read from the
// bf16 d scores, chain 2 against ONE fp16 term of (W2 w3)^T and g = h1 . d bias as ONE bf16 term: 8 MFMAs per key
// (1 + 1 layer 1, 2 mask transposition, 2 chain 2, 2 dW2) instead of 12, and none of the residual arithmetic of the splits.
// ------------------------------------------------------------------------------------------------
// RECOMP (the forward took its bias from the table - smml_deform_attn_table_fwd - and saved no ReLU bits): layer 2 of the MLP is recomputed
// per key in the query-major layout, d = W2 relu(x1) + b2 as ONE bf16 term (2 MFMAs; b2 enters through the accumulator, read from LDS), and
// its sign pattern becomes the mask operand directly (exact 0 / 1 fp16 values; the constants then carry no slot scale).  EXPORT (tests): the
// decisions are also written out in the forward's bit layout (MKO), so that they can be imposed on the oracle.
// MSRC 2 (the default of table-forward calls): the layer-2 decisions come from a MASK TABLE instead - the sign pattern of W2 relu(W1 p + b1) + b2
// evaluated in fp32 at the centres of a 1024 x 1024 (16384 in 1-D) grid of cells over the same [-pmax, pmax] range (cpb_mask_table_kernel, 4 MB:
// L2-resident), one 16-bit gather per lane and key issued a key ahead.  A pair's decision can differ from its own pre-activation's sign only where
// a layer-2 kink crosses its cell, |x2| <= |grad x2| . 1.8e-3: the size of the decision noise the single-term bf16 product of the other two forms
// has anyway (tests/test_gpu_deform16.py asserts both against the same margin).
struct MaskTab {
  const u16* tab;          // [cells][2 lane halves], the forward's bit layout per half
  float invh, off, imax;   // cell index along an axis = clamp(p * invh + off, 0, imax) truncated
};
template <int PDX, int MSRC = 0, bool EXPORT = false>
__global__ __launch_bounds__(256, SMML16_CPB_WPS) void cpb16_bwd_kernel(
    const u16* __restrict__ dLT, const u16* __restrict__ MK, const float* __restrict__ VS, const float* __restrict__ GQ,
    CpbParams cp, float* __restrict__ slab, float* __restrict__ dvs_slab, int N, int J, int H, int G, int NST, u16* __restrict__ MKO = nullptr,
    MaskTab mt = MaskTab{}) {
  constexpr int PD = PosCfg<PDX>::PD;
  constexpr bool RAW = PosCfg<PDX>::RAW;           // raw offsets: with saved masks (MSRC 0) only - the table-forward modes are signed-log
  constexpr bool RECOMP = MSRC == 1;
  constexpr int MT_BITS = (PD == 2) ? 10 : 14;      // cells per axis of the mask table: 1024 (2-D), 16384 (1-D)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (QT * WAVES) + wave * QT;
  const int o = H / G, g = h / o, oi = h - g * o;
  const bool qvalid = (q0 + c) < N;
  const int qi = qvalid ? (q0 + c) : (N - 1);

  float* tab = smem;
  float* wbase = smem + CPB2_TAB;
  float* xq = wbase + wave * CPB2_WAVE_LDS;                 // [2][32]
  float2* stg = reinterpret_cast<float2*>(xq + CPB_XQ);     // [16 keys][65]
  const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float2* dvs_row = reinterpret_cast<float2*>(dvs_slab) + (size_t)(wg * WAVES + wave) * J;
  if (tid < 32) {                                           // tid = 16 half + r
    const int ch = acc_row(tid & 15, tid >> 4);
    if (RECOMP) {                                           // 2 b2 of the half's 16 output rows (the chain runs on 2 h1 = relu2)
      tab[(tid >> 4) * 32 + (tid & 15)] = 2.f * cp.b2[ch];
    } else {
      tab[(tid >> 4) * 32 + (tid & 15)] = cp.w1[ch * PD];
      tab[(tid >> 4) * 32 + 16 + (tid & 15)] = (PD == 2) ? cp.w1[ch * PD + 1] : 0.f;
    }
  }
  const float* tabh = tab + hf * 32;
  const float gq0 = GQ[(size_t)qi * PD];
  const float gq1 = (PD == 2) ? GQ[(size_t)qi * PD + 1] : 0.f;

  // layer 1 in both layouts (operand layouts: deform_attn.hip cpb_bwd_kernel)
  bf16x8 a1q, a1t;
  floatx16 b1acc;
  {
    const float wx = cp.w1[c * PD], wy = (PD == 2) ? cp.w1[c * PD + 1] : 0.f, bb = cp.b1[c];
    const __bf16 xh = (__bf16)wx; const float xr = wx - (float)xh; const __bf16 xm = (__bf16)xr;
    const __bf16 xl = (__bf16)(xr - (float)xm);
    const __bf16 yh = (__bf16)wy; const float yr = wy - (float)yh; const __bf16 ym = (__bf16)yr;
    const __bf16 yl = (__bf16)(yr - (float)ym);
    const __bf16 bh_ = (__bf16)bb; const float br = bb - (float)bh_; const __bf16 bm = (__bf16)br;
    const __bf16 bl_ = (__bf16)(br - (float)bm);
    const __bf16 z = (__bf16)0.f;
    a1q = cpb_l1_weights_q(wx, wy, hf);
    if (hf == 0) a1t = (bf16x8){xh, yh, xh, yh, xh, yh, bh_, bm};
    else a1t = (bf16x8){xm, ym, xm, ym, xl, yl, bl_, z};
#pragma unroll
    for (int s16 = 0; s16 < 16; ++s16) b1acc[s16] = cp.b1[acc_row(s16, hf)];
  }
  const float b2c = cp.b2[c];
  const float w3c = cp.w3[oi * CH + c];

  half8 w2t[2];                        // W2[out = ch(8 kb + j)][in = c] * w3[out] * slot scale * lift: A operand of chain 2, one fp16 term
  half8 idb[2];                        // scaled identity: mask . I = mask^T as exact 0.0 / 1.0
  float unlift2;
  {
    float amax = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) { const int ch = acc_row(s, hf); amax = fmaxf(amax, fabsf(cp.w2[ch * CH + c] * cp.w3[oi * CH + ch])); }
    const float lift2 = pow2_lift(wave_max_all(amax), 256.f, -8.f, 24.f);
    unlift2 = 1.f / lift2;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = acc_row(8 * kb + j, hf);
        const float sc = RECOMP ? 1.f : ((j & 1) ? 0.5f : 128.f);
        w2t[kb][j] = (_Float16)(cp.w2[ch * CH + c] * cp.w3[oi * CH + ch] * sc * lift2);
        idb[kb][j] = (ch == c) ? (_Float16)sc : (_Float16)0.0f;
      }
    }
  }

  bf16x8 w2f[2];                       // RECOMP: W2 as the A operand of layer 2 - lane (out = c, half hf), K-block kb, element j <-> in = acc_row(8 kb + j, hf)
  if (RECOMP) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[j] = cp.w2[c * CH + acc_row(8 * kb + j, hf)];
      w2f[kb] = cvt8<__bf16>(wv);
    }
  }
#if SMML16_DP_MFMA
  bf16x8 a1d[2];                       // W1^T as an A operand: row 0 = w1x, row 1 = w1y over k = hidden channel acc_row(8 kb + j, hf); other rows zero
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = acc_row(8 * kb + j, hf);
      const float wv = (c == 0) ? cp.w1[ch * PD] : ((c == 1 && PD == 2) ? cp.w1[ch * PD + 1] : 0.f);
      a1d[kb][j] = (__bf16)wv;
    }
#endif
  float big;
  asm("s_mov_b32 %0, 0x71800000" : "=s"(big));
  floatx16 e = {0};                    // sum_q mask[out, q] g[in, q]: rows = out, lane = in (times w3[out] at the end)
  float2v aw1x[8], aw1y[8], ab1[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) { aw1x[p] = (float2v){0.f, 0.f}; aw1y[p] = aw1x[p]; ab1[p] = aw1x[p]; }
  float ab3 = 0.f;
  float2v s2 = {0.f, 0.f};

  const float* VSb = VS + (size_t)(b * G + g) * J * PD;
  const u16* dLTb = dLT + ((size_t)(b * H + h) * NST + q0) * J;
  __syncthreads();
  float vx_n = VSb[0];
  float vy_n = (PD == 2) ? VSb[1] : 0.f;
  unsigned db_n = dLTb[c];
  const u16* MKb = (MSRC != 0) ? nullptr : MK + (((size_t)(b * H + h) * NST + q0) * J) * 2 + hf * 32;
  u16* MKOb = (MSRC != 0 && EXPORT) ? MKO + (((size_t)(b * H + h) * NST + q0) * J) * 2 + hf * 32 : nullptr;
  // mask table: cell of (query, key) -> this lane half's 16 decisions
  float p0_n = 0.f, p1_n = 0.f;                    // MSRC 2: the signed-log offsets of the NEXT key (computed for its lookup, reused as that key's p)
  auto mask_lookup = [&](float vxk, float vyk) -> unsigned {
    p0_n = slog1p(gq0 - vxk);
    const int i0 = (int)__builtin_amdgcn_fmed3f(fmaf(p0_n, mt.invh, mt.off), 0.f, mt.imax);
    int cell = i0;
    if (PD == 2) {
      p1_n = slog1p(gq1 - vyk);
      cell |= ((int)__builtin_amdgcn_fmed3f(fmaf(p1_n, mt.invh, mt.off), 0.f, mt.imax)) << MT_BITS;
    }
    return mt.tab[(size_t)cell * 2 + hf];
  };
  float vx_nn = 0.f, vy_nn = 0.f;                  // MSRC 2: sample position two keys ahead (the gather of key j + 1 needs key j + 1's position a key early)
  unsigned m16_n = (MSRC == 0) ? MKb[c] : 0u;
  if (MSRC == 2) {
    m16_n = mask_lookup(vx_n, vy_n);
    const int j1 = min(1, J - 1);
    vx_nn = VSb[(size_t)j1 * PD];
    vy_nn = (PD == 2) ? VSb[(size_t)j1 * PD + 1] : 0.f;
  }

  for (int j = 0; j < J; ++j) {
    const float vx = vx_n, vy = vy_n, dbias = qvalid ? tof<__bf16>(db_n) : 0.f;
    const unsigned m16 = m16_n;
    const float p0_c = p0_n, p1_c = p1_n;           // MSRC 2: this key's offsets, from its lookup an iteration ago
    {
      const int jn = min(j + 1, J - 1);
      if (MSRC == 2) {
        vx_n = vx_nn; vy_n = vy_nn;                 // key j + 1, loaded an iteration ago
        m16_n = mask_lookup(vx_n, vy_n);
        const int jnn = min(j + 2, J - 1);
        vx_nn = VSb[(size_t)jnn * PD];
        if (PD == 2) vy_nn = VSb[(size_t)jnn * PD + 1];
      } else {
        vx_n = VSb[(size_t)jn * PD];
        if (PD == 2) vy_n = VSb[(size_t)jn * PD + 1];
      }
      db_n = dLTb[(size_t)jn * 32 + c];
      if (MSRC == 0) m16_n = MKb[(size_t)jn * 64 + c];
    }
    float* xb = xq + (j & 1) * 32;
    xb[c] = dbias;
    const float d0 = gq0 - vx, d1 = gq1 - vy;
    const float p0 = (MSRC == 2) ? p0_c : pos_of<RAW>(d0);
    const float p1 = (PD == 2) ? ((MSRC == 2) ? p1_c : slog1p(d1)) : 0.f;

    floatx16 xacc, ht;
    {
      const PosTerms pt = cpb_split_pos(p0, p1);
      xacc = cpb_layer1_q(a1q, pt, hf, b1acc);
      const uint4v tw = {pt.hw, pt.mw, hf ? pt.hw : pt.lw, hf ? 0x00003F80u : 0x3F803F80u};
      ht = mfma16b(__builtin_bit_cast(bf16x8, tw), a1t, (floatx16){0});
    }
#if SMML16_GATE_MUL
    float on1[16];                      // exact 0.0 / 1.0: clamp(x 2^100)
#pragma unroll
    for (int r = 0; r < 16; ++r) on1[r] = fminf(fmaxf(xacc[r] * big, 0.f), 1.f);
#else
    bool on1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) on1[r] = xacc[r] > 0.f;
#endif

    half8 mk[2];
    if constexpr (RECOMP) {
      floatx16 d2;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const float4 t = *reinterpret_cast<const float4*>(tabh + 4 * rg);
        d2[4 * rg] = t.x; d2[4 * rg + 1] = t.y; d2[4 * rg + 2] = t.z; d2[4 * rg + 3] = t.w;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float hv[8];
#pragma unroll
        for (int jx = 0; jx < 8; ++jx) hv[jx] = relu2(xacc[8 * kb + jx]);
        d2 = mfma16b(w2f[kb], cvt8<__bf16>(hv), d2);
      }
      float mbits = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        uint4v mw;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int r = 8 * kb + 2 * p;
          const float m0 = fminf(fmaxf(d2[r] * big, 0.f), 1.f), m1 = fminf(fmaxf(d2[r + 1] * big, 0.f), 1.f);     // exact 0 / 1
          mw[p] = pack2<_Float16>(m0, m1);
          if (EXPORT) mbits = fmaf(m1, (float)(1u << ((14 + r) & 15)), fmaf(m0, (float)(1u << ((13 + r) & 15)), mbits));
        }
        mk[kb] = __builtin_bit_cast(half8, mw);
      }
      if (EXPORT) MKOb[(size_t)j * 64 + c] = (u16)(unsigned)mbits;
    } else {
      if (MSRC == 2 && EXPORT) MKOb[(size_t)j * 64 + c] = (u16)m16;
      const unsigned mm2 = m16 | (m16 << 16);
      uint4v w0, w1;
      w0[0] = mm2 & 0x40002000u;
      w0[1] = __builtin_amdgcn_alignbit(mm2, mm2, 2) & 0x40002000u;
      w0[2] = __builtin_amdgcn_alignbit(mm2, mm2, 4) & 0x40002000u;
      w0[3] = __builtin_amdgcn_alignbit(mm2, mm2, 6) & 0x40002000u;
      w1[0] = __builtin_amdgcn_alignbit(mm2, mm2, 8) & 0x40002000u;
      w1[1] = __builtin_amdgcn_alignbit(mm2, mm2, 10) & 0x40002000u;
      w1[2] = __builtin_amdgcn_alignbit(mm2, mm2, 12) & 0x40002000u;
      w1[3] = __builtin_amdgcn_alignbit(mm2, mm2, 14) & 0x40002000u;
      mk[0] = __builtin_bit_cast(half8, w0);
      mk[1] = __builtin_bit_cast(half8, w1);
    }
    floatx16 mtt = mfma16(mk[0], idb[0], (floatx16){0});
    mtt = mfma16(mk[1], idb[1], mtt);

    float dbq[16];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const float4 t = *reinterpret_cast<const float4*>(xb + 8 * rg + 4 * hf);          // broadcast reads
      dbq[4 * rg] = t.x; dbq[4 * rg + 1] = t.y; dbq[4 * rg + 2] = t.z; dbq[4 * rg + 3] = t.w;
    }
    bf16x8 am[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      uint4v amw;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int r = 8 * t + 2 * p;
        s2[0] = fmaf(mtt[r], dbq[r], s2[0]);
        s2[1] = fmaf(mtt[r + 1], dbq[r + 1], s2[1]);
        amw[p] = pack2<__bf16>(mtt[r], mtt[r + 1]);
      }
      am[t] = __builtin_bit_cast(bf16x8, amw);
    }
    // chain 2: dh1[in = ch(r)][query = c] = (W2 w3)^T mask, one fp16 term of the constant against the exact mask operand
    floatx16 dh = {0};
    dh = mfma16(w2t[0], mk[0], dh);
    dh = mfma16(w2t[1], mk[1], dh);
    ab3 += (hf == 0) ? dbias : 0.f;

    // dW2 += mask^T g, g = 2 h1^T . d bias as one bf16 term
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float g8[8];
#pragma unroll
      for (int jx = 0; jx < 8; ++jx) g8[jx] = relu2(ht[8 * t + jx]) * dbq[8 * t + jx];
      e = mfma16b(am[t], cvt8<__bf16>(g8), e);
    }

    // layer-1 backward, d vs
    {
      float2v dp0v = {0.f, 0.f}, dp1v = {0.f, 0.f};
      const float dbl = dbias * unlift2;
      const float p0i = p0 * dbl, p1i = p1 * dbl;
#if SMML16_DP_MFMA
      uint4v xw0, xw1;
#endif
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        float2v g1;
#if SMML16_GATE_MUL
        g1 = (float2v){dh[2 * p], dh[2 * p + 1]} * (float2v){on1[2 * p], on1[2 * p + 1]};
#else
        g1[0] = on1[2 * p] ? dh[2 * p] : 0.f;
        g1[1] = on1[2 * p + 1] ? dh[2 * p + 1] : 0.f;
#endif
        ab1[p] = g1 * (float2v){dbl, dbl} + ab1[p];
        aw1x[p] = g1 * (float2v){p0i, p0i} + aw1x[p];
        if (PD == 2) aw1y[p] = g1 * (float2v){p1i, p1i} + aw1y[p];
#if SMML16_DP_MFMA
        if (p < 4) xw0[p] = pack2<__bf16>(g1[0], g1[1]); else xw1[p - 4] = pack2<__bf16>(g1[0], g1[1]);
#else
        const float2 wx = *reinterpret_cast<const float2*>(tabh + 2 * p);
        dp0v = g1 * (float2v){wx.x, wx.y} + dp0v;
        if (PD == 2) {
          const float2 wy = *reinterpret_cast<const float2*>(tabh + 16 + 2 * p);
          dp1v = g1 * (float2v){wy.x, wy.y} + dp1v;
        }
#endif
      }
#if SMML16_DP_MFMA
      // d p[c][query] = sum_ch W1[ch][c] (m1 . d h1)[ch][query]: rows 0 / 1 of the product = registers 0 / 1 of the lanes of half 0 (complete sums:
      // the K dimension covers all 32 channels); half 1 holds rows 4 / 5 = 0
      floatx16 dpa = mfma16b(a1d[0], __builtin_bit_cast(bf16x8, xw0), (floatx16){0});
      dpa = mfma16b(a1d[1], __builtin_bit_cast(bf16x8, xw1), dpa);
      dp0v[0] = dpa[0]; dp1v[0] = dpa[1];
#endif
      float2 v;
      v.x = -(dp0v[0] + dp0v[1]) * dbl * dpos_of<RAW>(d0, big);
      v.y = (PD == 2) ? -(dp1v[0] + dp1v[1]) * dbl * (srcp(fabsf(d1) + 1.f) * fminf(fmaxf(fabsf(d1) * big, 0.f), 1.f)) : 0.f;
      stg[(j & (CPB2_STG_KEYS - 1)) * 65 + lane] = v;
    }
    if ((j & (CPB2_STG_KEYS - 1)) == CPB2_STG_KEYS - 1 || j == J - 1) {   // uniform: flush the staging tile
      asm volatile("" ::: "memory");
      const int kk = lane & 31, nrow = (j & (CPB2_STG_KEYS - 1)) + 1;
      float sx = 0.f, sy = 0.f;
      if (kk < nrow) {
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
          const float2 t = stg[kk * 65 + 32 * hf + i];
          sx += t.x; sy += t.y;
        }
      }
      sx = xhalf_sum(sx); sy = xhalf_sum(sy);
      if (hf == 0 && kk < nrow) dvs_row[(j & ~(CPB2_STG_KEYS - 1)) + kk] = make_float2(sx, (PD == 2) ? sy : 0.f);
      asm volatile("" ::: "memory");
    }
  }

  // workgroup reduction of the per-lane partials -> slab[wg] (fixed order, no atomics; layout: deform_common.h CPB_SLAB)
  __syncthreads();
  float* red = wbase + WAVES * CPB2_WAVE_LDS + wave * CPB_SLAB;
  for (int i = lane; i < CPB_SLAB; i += 64) red[i] = 0.f;
  {
    const float s2s = xhalf_sum(s2[0] + s2[1]);
    if (hf == 0) {
      red[1024 + 64 + 32 + c] = w3c * s2s;
      red[1024 + 64 + 32 + 32 + c] = b2c * s2s;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, hf);
    red[row * CH + c] = 0.5f * e[r] * cp.w3[oi * CH + row];
    float v;
    v = 0.5f * e[r] * cp.w2[row * CH + c];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (c == 0) red[1024 + 64 + 32 + 32 + row] += v;
    v = ab1[r >> 1][r & 1];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (c == 0) red[1024 + 64 + row] = v;
    v = aw1x[r >> 1][r & 1];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (c == 0) red[1024 + row * 2] = v;
    v = aw1y[r >> 1][r & 1];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (c == 0) red[1024 + row * 2 + 1] = v;
  }
  {
    float v = wave_sum(ab3);
    if (lane == 0) red[1024 + 64 + 32 + 32 + 32] = v;
  }
  __syncthreads();
  float* sl = slab + (size_t)wg * CPB_SLAB;
  const float* r0 = wbase + WAVES * CPB2_WAVE_LDS;
  for (int i = tid; i < CPB_SLAB; i += 256)
    sl[i] = (r0[i] + r0[CPB_SLAB + i]) + (r0[2 * CPB_SLAB + i] + r0[3 * CPB_SLAB + i]);
}

// ------------------------------------------------------------------------------------------------
// table mode, backward of the position bias.  bias(q, key) = interp(table, slog(gq - vs)), so
//   d table[cell] = sum over pairs of d bias . (interpolation weight of the cell)      - a histogram over the TG^PD cells, and
//   d vs[key]     = - sum over queries of d bias . (d interp / d p) / (|d| + 1)
// (the parameter gradients follow from d table through the three small GEMMs that built the table).  One LANE owns one KEY: it reads
// its key's row of a 32-query tile of d scores (64 contiguous bytes; the wave's 64 rows are one 4 KB block), walks the 32 queries with
// the query position in scalar registers, and keeps its d vs sum in registers - no cross-lane reduction anywhere.  The histogram lives
// in LDS (fp32 atomic adds: the one order-dependent sum of this mode) and is flushed to a per-workgroup slab.
// grid (nkb * S, H, B): key block kb = 64 keys, S query slices; the 8 waves of a workgroup take every 8th query tile of the slice.
// ------------------------------------------------------------------------------------------------
constexpr int TBW = 8;            // waves per workgroup of the table backward
template <int PD, int TG, bool HIST = true>
__global__ __launch_bounds__(64 * TBW, 1) void cpb_table_bwd_kernel(
    const u16* __restrict__ dLT, const float* __restrict__ VS, const float* __restrict__ GQ, TabCfg tc, float* __restrict__ hist_slab,
    float2* __restrict__ dvs_rows, int N, int J, int H, int G, int NST, int S, int tiles_per_slice) {
  constexpr int NC = tab_cells<PD, TG>();
  __shared__ float tabl[NC];
  __shared__ float hist[HIST ? NC : 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, h = blockIdx.y, kb = blockIdx.x / S, sl = blockIdx.x - kb * S;
  const int o = H / G, g = h / o, oi = h - g * o;
  {
    const float* tsrc = tc.tab + (size_t)oi * NC;
    for (int i = tid; i < NC; i += 64 * TBW) {
      tabl[i] = tsrc[i];
      if (HIST) hist[i] = 0.f;
    }
  }
  __syncthreads();
  const int key = kb * 64 + lane;
  const bool kvalid = key < J;
  const int keyc = kvalid ? key : (J - 1);
  const float* VSb = VS + ((size_t)(b * G + g) * J + keyc) * PD;
  const float vs0 = VSb[0], vs1 = (PD == 2) ? VSb[1] : 0.f;
  float acc0 = 0.f, acc1 = 0.f;
  int run_idx = 0;                                    // HIST: the lane's current cell and its corner sums
  float run_w[4] = {0.f, 0.f, 0.f, 0.f};
  const int ntq = (N + 31) / 32;
  const int t_end = min((sl + 1) * tiles_per_slice, ntq);
  for (int t = sl * tiles_per_slice + wave; t < t_end; t += TBW) {
    const int q0 = t * 32, nq = min(32, N - q0);
    const uint4v* rp = reinterpret_cast<const uint4v*>(dLT + (((size_t)(b * H + h) * NST + q0) * J + (size_t)keyc * 32));
    uint4v rw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rw[i] = kvalid ? rp[i] : (uint4v){0u, 0u, 0u, 0u};
    // the tile's query positions: one coalesced load (lane l holds float l of the tile's 32 x PD block), read back per query with
    // v_readlane - no memory wait inside the query loop, and no branch: queries past the bag's end run on a clamped position with d bias = 0
    const float gqv = GQ[min((size_t)q0 * PD + lane, (size_t)N * PD - 1)];
    // eight queries per trip of a four-trip loop (kept rolled: with the run-length branches of the histogram a fully unrolled body costs 176 spills)
#pragma unroll 1
    for (int g8 = 0; g8 < 4; ++g8) {
      const uint4v w4 = (g8 == 0) ? rw[0] : (g8 == 1) ? rw[1] : (g8 == 2) ? rw[2] : rw[3];
#pragma unroll
      for (int ii = 0; ii < 8; ++ii) {
        const int i = 8 * g8 + ii;
        const unsigned w = w4[ii >> 1];
        const float db = (i < nq) ? ((ii & 1) ? bf_hi(w) : bf_lo(w)) : 0.f;
        const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gqv), i * PD)) - vs0;
        const float a0 = fabsf(d0) + 1.0f;
        const float u0r = fmaf(copysignf(__builtin_amdgcn_logf(a0) * 0.6931471805599453f, d0), tc.invh, tc.off);
        const float u0 = __builtin_amdgcn_fmed3f(u0r, 0.f, tc.umax);
        const float f0 = __builtin_amdgcn_fractf(u0);
        if constexpr (PD == 2) {
          const float d1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gqv), i * PD + 1)) - vs1;
          const float a1 = fabsf(d1) + 1.0f;
          const float u1r = fmaf(copysignf(__builtin_amdgcn_logf(a1) * 0.6931471805599453f, d1), tc.invh, tc.off);
          const float u1 = __builtin_amdgcn_fmed3f(u1r, 0.f, tc.umax);
          const float f1 = __builtin_amdgcn_fractf(u1);
          const int idx = __mul24((int)u1, TG) + (int)u0;
          const float t00 = tabl[idx], t10 = tabl[idx + 1], t01 = tabl[idx + TG], t11 = tabl[idx + TG + 1];
          // slopes of the bilinear patch (zero where the position was clamped to the table's edge)
          const float gx0 = t10 - t00, gx1 = t11 - t01, gy0 = t01 - t00, gy1 = t11 - t10;
          float sx = fmaf(f1, gx1 - gx0, gx0), sy = fmaf(f0, gy1 - gy0, gy0);
          sx = (u0r == u0) ? sx : 0.f;
          sy = (u1r == u1) ? sy : 0.f;
          const float dbh = db * tc.invh;
          acc0 = fmaf(-dbh * sx, srcp(a0), acc0);
          acc1 = fmaf(-dbh * sy, srcp(a1), acc1);
          if (HIST) {
            // run-length accumulation: consecutive queries of a key mostly fall into the same cell (a lane keeps the four corner sums of its
            // current cell in registers and issues the four LDS atomics only when the cell changes - executed under the mask of the lanes that
            // moved: the atomics' cost goes with the active lanes)
            const float w1x = db * f0, w0x = db - w1x;
            const float h01 = w0x * f1, h11 = w1x * f1;
            if (idx != run_idx) {
              atomicAdd(&hist[run_idx], run_w[0]);
              atomicAdd(&hist[run_idx + 1], run_w[1]);
              atomicAdd(&hist[run_idx + TG], run_w[2]);
              atomicAdd(&hist[run_idx + TG + 1], run_w[3]);
              run_idx = idx;
              run_w[0] = run_w[1] = run_w[2] = run_w[3] = 0.f;
            }
            run_w[0] += w0x - h01; run_w[1] += w1x - h11; run_w[2] += h01; run_w[3] += h11;
          }
        } else {
          const int idx = (int)u0;
          const float t0 = tabl[idx], t1 = tabl[idx + 1];
          const float sx = (u0r == u0) ? (t1 - t0) : 0.f;
          acc0 = fmaf(-db * tc.invh * sx, srcp(a0), acc0);
          if (HIST) {           // run-length accumulation (see the 2-D branch); in 1-D a key's cell index is monotonic in the query index
            const float w1x = db * f0;
            if (idx != run_idx) {
              atomicAdd(&hist[run_idx], run_w[0]);
              atomicAdd(&hist[run_idx + 1], run_w[1]);
              run_idx = idx;
              run_w[0] = run_w[1] = 0.f;
            }
            run_w[0] += db - w1x; run_w[1] += w1x;
          }
        }
      }
    }
  }
  if (kvalid) dvs_rows[((size_t)((b * H + h) * S + sl) * TBW + wave) * J + key] = make_float2(acc0, acc1);
  if (HIST) {
    atomicAdd(&hist[run_idx], run_w[0]);              // the last run of every lane (zeros if it never started)
    atomicAdd(&hist[run_idx + 1], run_w[1]);
    if (PD == 2) {
      atomicAdd(&hist[run_idx + TG], run_w[2]);
      atomicAdd(&hist[run_idx + TG + 1], run_w[3]);
    }
    __syncthreads();
    float* slab = hist_slab + (size_t)((((size_t)b * H + h) * gridDim.x) + blockIdx.x) * NC;
    for (int i = tid; i < NC; i += 64 * TBW) slab[i] = hist[i];
  }
}

// ------------------------------------------------------------------------------------------------
// table mode, d table for queries on a REGULAR GRID (the 2-D module: query q = y * Ww + x sits at (X[x], Y[y])).  The bilinear weight
// of a pair factorises, w(c1, c0) = hat(c1; u1(y, key)) . hat(c0; u0(x, key)) with hat(c; u) = clamp(1 - |u - c|, 0, 1), and u0 depends on
// (x, key) only, u1 on (y, key) only, so the histogram of one key is a pair of small dense products on the matrix pipe,
//      d table[c1][c0] += sum_y hat1[c1][y] . ( sum_x DB_key[y][x] . hat0[c0][x] ),
// with DB_key the key's Hh x Ww sheet of d scores (bf16 as stored: exact operands) - no atomics anywhere (the LDS float atomics of
// cpb_table_bwd_kernel run at ~0.3 lane-operations per clock and CU: 8.8 ms per launch at the headline shape; this form: see DESIGN.md).
// The hat weights and the intermediate sheet are rounded to bf16 (8 bits; random-sign errors over >= 10^4 pairs per cell).
// One workgroup = (bag, head, chunk of keys); wave w owns the c0 block [32 w, 32 w + 32) of the 96 x 96 table and keeps its 96 x 32
// part of the sum in 48 accumulator registers across the chunk's keys; the second product takes the first one's accumulators as its
// B operand directly (register r <-> row y = acc_row(r, half): the A fragments of hat1 are read in the same order).
// Per key the workgroup stages the sheet (coalesced 16-byte pieces of the [J][32] tile rows; linear in q, so a row's 8 consecutive x
// of an A fragment are two 8-byte LDS reads when Ww % 4 == 0, eight 2-byte reads otherwise) and the key's 2 x 128 cell coordinates in LDS;
// the hat fragments are formed in registers from the coordinates when an MFMA needs them (a first version kept both hat tables in LDS:
// 52 KB per workgroup, two workgroups per CU, 0.85 ms; this form 0.53 ms).
// Requires Hh, Ww <= 128 (LDS: the sheet is padded to 32-row / 16-column blocks); other shapes take cpb_table_bwd_kernel's atomics.
// ------------------------------------------------------------------------------------------------
constexpr int TGW = 3;                       // waves per workgroup = 32-wide c0 blocks of the 96-point table
constexpr int TABLE_GRID_MAX = 128;          // largest grid side of the fast path
constexpr int TABLE_GRID_KEYS = 32;          // keys per workgroup
__host__ __device__ inline int table_sheet_halves(int Hh, int Ww) {     // LDS halves of one key's padded sheet
  return (((Hh + 31) / 32) * 32 - 1) * Ww + ((Ww + 15) / 16) * 16 + 8;
}
template <int TG, bool ALIGNED>
__global__ __launch_bounds__(64 * TGW) void cpb_table_grid_bwd_kernel(
    const u16* __restrict__ dLT, const float* __restrict__ VS, const float* __restrict__ GQ, TabCfg tc, float* __restrict__ hist_slab,
    int N, int J, int H, int G, int NST, int Hh, int Ww) {
  static_assert(TG == 32 * TGW, "one wave per 32 table columns");
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  __bf16* dbs = reinterpret_cast<__bf16*>(dyn_lds);                    // the key's sheet, linear in q (zero beyond N)
  __shared__ __attribute__((aligned(16))) float u0s[TABLE_GRID_MAX], u1s[TABLE_GRID_MAX];
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, h = blockIdx.y;
  const int o = H / G, g = h / o;
  const int k0 = blockIdx.x * TABLE_GRID_KEYS, k1 = min(k0 + TABLE_GRID_KEYS, J);
  const int nyb = (Hh + 31) / 32, nsx = (Ww + 15) / 16;
  const int sheet = table_sheet_halves(Hh, Ww);
  const int nchunk = ((N + 31) / 32) * 4;            // 16-byte pieces of the tile rows that hold queries
  for (int i = nchunk * 8 + tid; i < sheet; i += 64 * TGW) dbs[i] = (__bf16)0.f;     // never written again
  floatx16 acc2[3] = {{0}, {0}, {0}};
  const float* VSb = VS + (size_t)(b * G + g) * J * 2;
  const u16* Lb = dLT + (size_t)(b * H + h) * NST * J;
  // eight hat weights hat(row; u[p .. p + 7]) as one bf16 fragment, straight from the cell coordinates in LDS (two 16-byte broadcast reads):
  // the hat tables themselves are never materialised - 52 KB of LDS less per workgroup (two -> five workgroups per CU) and one barrier less per key
  auto hat8 = [&](const float* us, float rowf) -> bf16x8 {
    const float4 ua = *reinterpret_cast<const float4*>(us), ub = *reinterpret_cast<const float4*>(us + 4);
    const float hv[8] = {__builtin_amdgcn_fmed3f(1.f - fabsf(ua.x - rowf), 0.f, 1.f), __builtin_amdgcn_fmed3f(1.f - fabsf(ua.y - rowf), 0.f, 1.f),
                         __builtin_amdgcn_fmed3f(1.f - fabsf(ua.z - rowf), 0.f, 1.f), __builtin_amdgcn_fmed3f(1.f - fabsf(ua.w - rowf), 0.f, 1.f),
                         __builtin_amdgcn_fmed3f(1.f - fabsf(ub.x - rowf), 0.f, 1.f), __builtin_amdgcn_fmed3f(1.f - fabsf(ub.y - rowf), 0.f, 1.f),
                         __builtin_amdgcn_fmed3f(1.f - fabsf(ub.z - rowf), 0.f, 1.f), __builtin_amdgcn_fmed3f(1.f - fabsf(ub.w - rowf), 0.f, 1.f)};
    return cvt8<__bf16>(hv);
  };
  const float c0f = (float)(32 * wave + c);
  for (int key = k0; key < k1; ++key) {
    __syncthreads();                                 // the previous key's sheet and coordinates have been consumed
    const float vs0 = VSb[(size_t)key * 2], vs1 = VSb[(size_t)key * 2 + 1];
    for (int i = tid; i < 2 * TABLE_GRID_MAX; i += 64 * TGW) {
      const int a = i & (TABLE_GRID_MAX - 1);
      if (i < TABLE_GRID_MAX)          // x axis: query x of row 0
        u0s[a] = (a < Ww) ? __builtin_amdgcn_fmed3f(fmaf(slog1p(GQ[(size_t)a * 2] - vs0), tc.invh, tc.off), 0.f, tc.umax) : -10.f;
      else                             // y axis: first query of row y
        u1s[a] = (a < Hh) ? __builtin_amdgcn_fmed3f(fmaf(slog1p(GQ[(size_t)a * Ww * 2 + 1] - vs1), tc.invh, tc.off), 0.f, tc.umax) : -10.f;
    }
    for (int ch = tid; ch < nchunk; ch += 64 * TGW) {
      const int q = ch * 8;
      uint4v w = *reinterpret_cast<const uint4v*>(Lb + ((size_t)(q >> 5) * 32 * J + (size_t)key * 32 + (q & 31)));
      if (q + 8 > N) {                               // the bag's last tile: padded query lanes hold no d score
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (q + 2 * e >= N) w[e] = 0u;
          else if (q + 2 * e + 1 >= N) w[e] &= 0xFFFFu;
        }
      }
      *reinterpret_cast<uint4v*>(&dbs[q]) = w;
    }
    __syncthreads();
    // B fragments of the first product for this wave's 32 columns c0: hat0[c0][x], x = 16 st + 8 hf + j (the same for every row block)
    bf16x8 b1f[TABLE_GRID_MAX / 16];
#pragma unroll
    for (int st = 0; st < TABLE_GRID_MAX / 16; ++st)
      if (st < nsx) b1f[st] = hat8(u0s + 16 * st + 8 * hf, c0f);
    for (int yb = 0; yb < nyb; ++yb) {
      // first product: out1[y][c0] = sum_x DB[y][x] hat0[c0][x] for the 32 rows y of block yb (lane = y) and this wave's 32 columns c0
      floatx16 out1 = {0};
      const __bf16* ap = dbs + (32 * yb + c) * Ww + 8 * hf;
#pragma unroll
      for (int st = 0; st < TABLE_GRID_MAX / 16; ++st) {
        if (st < nsx) {
          bf16x8 a1;
          if (ALIGNED) {                             // Ww % 4 == 0: every row starts on an 8-byte boundary
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(ap + 16 * st), hi = *reinterpret_cast<const bf16x4*>(ap + 16 * st + 4);
            a1 = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) a1[j] = ap[16 * st + j];
          }
          out1 = mfma16b(a1, b1f[st], out1);
        }
      }
      // second product: d table[c1][c0] += sum_y hat1[c1][y] out1[y][c0]; A fragment of k-block kb: y = 32 yb + acc_row(8 kb + j, hf)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float o8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = out1[8 * kb + j];
        const bf16x8 b2 = cvt8<__bf16>(o8);
        const float* up = u1s + 32 * yb + 16 * kb + 4 * hf;
        const float4 ua = *reinterpret_cast<const float4*>(up), ub = *reinterpret_cast<const float4*>(up + 8);
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
          const float rowf = (float)(32 * cb + c);
          const float hv[8] = {__builtin_amdgcn_fmed3f(1.f - fabsf(ua.x - rowf), 0.f, 1.f), __builtin_amdgcn_fmed3f(1.f - fabsf(ua.y - rowf), 0.f, 1.f),
                               __builtin_amdgcn_fmed3f(1.f - fabsf(ua.z - rowf), 0.f, 1.f), __builtin_amdgcn_fmed3f(1.f - fabsf(ua.w - rowf), 0.f, 1.f),
                               __builtin_amdgcn_fmed3f(1.f - fabsf(ub.x - rowf), 0.f, 1.f), __builtin_amdgcn_fmed3f(1.f - fabsf(ub.y - rowf), 0.f, 1.f),
                               __builtin_amdgcn_fmed3f(1.f - fabsf(ub.z - rowf), 0.f, 1.f), __builtin_amdgcn_fmed3f(1.f - fabsf(ub.w - rowf), 0.f, 1.f)};
          acc2[cb] = mfma16b(cvt8<__bf16>(hv), b2, acc2[cb]);
        }
      }
    }
  }
  // this workgroup's partial table: lane <-> c0 = 32 wave + c, register r of block cb <-> c1 = 32 cb + acc_row(r, hf)
  float* slab = hist_slab + (size_t)((((size_t)b * H + h) * gridDim.x) + blockIdx.x) * (TG * TG) + 32 * wave + c;
#pragma unroll
  for (int cb = 0; cb < 3; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) slab[(size_t)(32 * cb + acc_row(r, hf)) * TG] = acc2[cb][r];
}

// d vs[(b, g)][j] = sum over the heads of the group and the S * TBW rows of each, in that fixed order
__global__ void dvs_table_reduce_kernel(const float2* __restrict__ rows, float* __restrict__ dVS, int Bn, int G, int H, int rows_per_head,
                                        int J, int PD) {
  const long long out = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (out >= (long long)Bn * G * J) return;
  const int j = (int)(out % J);
  const int bg = (int)(out / J), b = bg / G, g = bg - b * G, o = H / G;
  float2 s = make_float2(0.f, 0.f);
  for (int oi = 0; oi < o; ++oi) {
    const float2* p = rows + (size_t)(b * H + g * o + oi) * rows_per_head * J + j;
    for (int r = 0; r < rows_per_head; ++r) {
      const float2 v = p[(size_t)r * J];
      s.x += v.x; s.y += v.y;
    }
  }
  dVS[out * PD] = s.x;
  if (PD == 2) dVS[out * PD + 1] = s.y;
}

// d table[oi][cell] += sum over the workgroup slabs of heads h with h % o == oi (slabs ordered (b, h, block)); grid (cells / 256, chunks):
// fixed order inside a chunk, one fp32 atomic per chunk into the zero-initialised output
__global__ void table_hist_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dtab, int NC, int nwg, int wg_per_head,
                                         int H, int o, int chunk) {
  const int cell = blockIdx.x * blockDim.x + threadIdx.x;
  if (cell >= NC) return;
  const int w0 = blockIdx.y * chunk, w1 = min(w0 + chunk, nwg);
  float s0 = 0.f, s1 = 0.f;
  for (int w = w0; w < w1; ++w) {
    const int oi = ((w / wg_per_head) % H) % o;
    const float v = slab[(size_t)w * NC + cell];
    if (oi == 0) s0 += v; else s1 += v;
  }
  atomicAdd(&dtab[cell], s0);
  if (o > 1) atomicAdd(&dtab[NC + cell], s1);
}

// mask table of cpb16_bwd_kernel<PD, 2>: one thread per cell evaluates layers 1 and 2 of the MLP in plain fp32 at the cell's centre and packs the
// 32 layer-2 signs into the two lane-half words of the forward's bit layout (hidden channel acc_row(r, half) at bit (13 + r) % 16).
template <int PD>
__global__ __launch_bounds__(256) void cpb_mask_table_kernel(CpbParams cp, u16* __restrict__ tab, int cells_per_axis, float pmax) {
  __shared__ float w1s[CH * 2], b1s[CH], w2s[CH * CH], b2s[CH];
  for (int i = threadIdx.x; i < CH * CH; i += 256) w2s[i] = cp.w2[i];
  if (threadIdx.x < CH) {
    w1s[threadIdx.x * 2] = cp.w1[threadIdx.x * PD];
    w1s[threadIdx.x * 2 + 1] = (PD == 2) ? cp.w1[threadIdx.x * PD + 1] : 0.f;
    b1s[threadIdx.x] = cp.b1[threadIdx.x];
    b2s[threadIdx.x] = cp.b2[threadIdx.x];
  }
  __syncthreads();
  const long long cell = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long ncell = (PD == 2) ? (long long)cells_per_axis * cells_per_axis : cells_per_axis;
  if (cell >= ncell) return;
  const float hcell = 2.f * pmax / (float)cells_per_axis;
  const int i0 = (int)(cell % cells_per_axis), i1 = (int)(cell / cells_per_axis);
  const float p0 = -pmax + ((float)i0 + 0.5f) * hcell, p1 = (PD == 2) ? -pmax + ((float)i1 + 0.5f) * hcell : 0.f;
  float h1[CH];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) h1[ch] = fmaxf(fmaf(w1s[2 * ch], p0, fmaf(w1s[2 * ch + 1], p1, b1s[ch])), 0.f);
  unsigned words[2] = {0u, 0u};
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int out = acc_row(r, half);
      float x2 = b2s[out];
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) x2 = fmaf(w2s[out * CH + ch], h1[ch], x2);
      words[half] |= (x2 > 0.f ? 1u : 0u) << ((13 + r) & 15);
    }
  reinterpret_cast<unsigned*>(tab)[cell] = words[0] | (words[1] << 16);
}

int check16(const char* fn, int B, int N, int J, int H, int G, int posdim, int dtype) {
  SMML_REQUIRE(B > 0 && N > 0 && J > 0 && H > 0 && G > 0, "%s: non-positive dimension", fn);
  SMML_REQUIRE(H % G == 0, "%s: heads (%d) must be divisible by offset groups (%d)", fn, H, G);
  SMML_REQUIRE(H / G <= 2, "%s: at most 2 heads per offset group are supported (got %d)", fn, H / G);
  SMML_REQUIRE(posdim == 1 || posdim == 2, "%s: posdim must be 1 or 2 (got %d)", fn, posdim);
  SMML_REQUIRE(deform_dims_ok(B, N, J, H), "%s: B, H <= 65535, N <= 2^26, J <= 2^22 (got B %d N %d J %d H %d)", fn, B, N, J, H);
  SMML_REQUIRE(dtype == 0 || dtype == 1, "%s: dtype must be 0 (bf16) or 1 (fp16), got %d", fn, dtype);
  return SMML_OK;
}

template <typename T>
void launch_fwd16(dim3 grid, hipStream_t st, bool save, int posdim, const float* q, const float* k, const float* v, const float* vs,
                  const float* gq, CpbParams cp, float* out, float* lse, u16* lt, u16* mk, int N, int J, int H, int G, int nst,
                  float scale, DropCfg dc, const SmmlDeformOpts* opts) {
  dim3 block(256);
  if (posdim == 2 && save)
    hipLaunchKernelGGL((deform16_fwd_kernel<2, true, T>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, lt, mk, N, J, H, G, nst, scale, dc);
  else if (posdim == 2)
    hipLaunchKernelGGL((deform16_fwd_kernel<2, false, T>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, lt, mk, N, J, H, G, nst, scale, dc);
  else if (pdx_of(posdim, opts) == 3 && save)              // 1-D, raw offsets (cpb_log_distance = False)
    hipLaunchKernelGGL((deform16_fwd_kernel<3, true, T>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, lt, mk, N, J, H, G, nst, scale, dc);
  else if (pdx_of(posdim, opts) == 3)
    hipLaunchKernelGGL((deform16_fwd_kernel<3, false, T>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, lt, mk, N, J, H, G, nst, scale, dc);
  else if (save)
    hipLaunchKernelGGL((deform16_fwd_kernel<1, true, T>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, lt, mk, N, J, H, G, nst, scale, dc);
  else
    hipLaunchKernelGGL((deform16_fwd_kernel<1, false, T>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, lt, mk, N, J, H, G, nst, scale, dc);
}

constexpr int TABLE_G2 = 96;      // grid points per axis of the 2-D table (36 KB in LDS: two forward workgroups per CU)
constexpr int TABLE_G1 = 1024;    // points of the 1-D table

template <typename T>
void launch_fwd_table(dim3 grid, hipStream_t st, bool save, int posdim, const float* q, const float* k, const float* v, const float* vs,
                      const float* gq, float* out, float* lse, u16* lt, int N, int J, int H, int G, int nst, float scale, DropCfg dc,
                      TabCfg tc) {
  dim3 block(256);
  const CpbParams cp{};
  u16* mk = nullptr;
  if (posdim == 2 && save)
    hipLaunchKernelGGL((deform16_fwd_kernel<2, true, T, TABLE_G2>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, lt, mk, N, J, H, G, nst, scale, dc, tc);
  else if (posdim == 2)
    hipLaunchKernelGGL((deform16_fwd_kernel<2, false, T, TABLE_G2>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, lt, mk, N, J, H, G, nst, scale, dc, tc);
  else if (save)
    hipLaunchKernelGGL((deform16_fwd_kernel<1, true, T, TABLE_G1>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, lt, mk, N, J, H, G, nst, scale, dc, tc);
  else
    hipLaunchKernelGGL((deform16_fwd_kernel<1, false, T, TABLE_G1>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, lt, mk, N, J, H, G, nst, scale, dc, tc);
}

TabCfg make_tab(const float* table, int tg, float pmax) {
  TabCfg tc;
  tc.tab = table;
  tc.invh = (float)((double)(tg - 1) / (2.0 * (double)pmax));
  tc.off = (float)((double)pmax * (double)(tg - 1) / (2.0 * (double)pmax));
  tc.umax = (float)(tg - 1) - 1.0f / 1024.0f;
  return tc;
}
int table_slices(int B, int N, int J, int H) {
  const int nkb = (J + 63) / 64, ntq = (N + 31) / 32;
  const long base = (long)B * H * nkb;
  long S = (1280 + base / 2) / base;
  if (S > ntq / TBW) S = ntq / TBW;
  if (S < 1) S = 1;
  return (int)S;
}
struct TableWorkspace { size_t rows, slab, total; };      // float offsets behind bwd_workspace(...).total
TableWorkspace table_workspace(int B, int N, int J, int H, int cells) {
  TableWorkspace w;
  const size_t S = table_slices(B, N, J, H), nkb = (J + 63) / 64;
  w.rows = (bwd_workspace(B, N, J, H).total + 3) & ~(size_t)3;
  w.slab = w.rows + (size_t)B * H * S * TBW * J * 2;
  size_t nslab = nkb * S;
  const size_t kc = (J + TABLE_GRID_KEYS - 1) / TABLE_GRID_KEYS;        // the grid fast path's workgroups per (bag, head)
  if (kc > nslab) nslab = kc;
  w.total = w.slab + (size_t)B * H * nslab * cells;
  return w;
}
int check_table(const char* fn, int posdim, int table_g, float pmax, const SmmlDeformOpts* opts) {
  SMML_REQUIRE(pdx_of(posdim, opts) != 3, "%s: the table modes are built for the signed-log position transform (log_distance) only", fn);
  SMML_REQUIRE(table_g == (posdim == 2 ? TABLE_G2 : TABLE_G1), "%s: the table kernels are built for %d grid points per axis with posdim %d (got %d)",
               fn, posdim == 2 ? TABLE_G2 : TABLE_G1, posdim, table_g);
  SMML_REQUIRE(pmax > 0.f, "%s: table_pmax must be positive", fn);
  return SMML_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C-ABI (include/smml.h)
// ------------------------------------------------------------------------------------------------
extern "C" {

int smml_deform_attn_nst(int N);
size_t smml_deform_attn_bwd_workspace_bytes(int B, int N, int J, int H);

int smml_deform_attn16_fwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* w1,
                           const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, float* out,
                           float* lse, unsigned short* logits16, unsigned short* relu_masks, int B, int N, int J, int H, int G,
                           int posdim, float scale, float dropout_p, unsigned long long dropout_seed, int dtype, void* ev_start,
                           void* ev_stop, void* stream, const SmmlDeformOpts* opts) {
  int rc = check16("smml_deform_attn16_fwd", B, N, J, H, G, posdim, dtype);
  if (rc) return rc;
  SMML_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "smml_deform_attn16_fwd: dropout_p must be in [0, 1)");
  SMML_REQUIRE(q && k && v && vs && gq && w1 && b1 && w2 && b2 && w3 && b3 && out && lse, "smml_deform_attn16_fwd: null pointer");
  SMML_REQUIRE((logits16 == nullptr) == (relu_masks == nullptr),
               "smml_deform_attn16_fwd: logits16 and relu_masks are saved together (training) or not at all");
  const DropCfg dc = make_drop(dropout_p, dropout_seed, opts);
  CpbParams cp{w1, b1, w2, b2, w3, b3};
  dim3 grid((N + QT * WAVES - 1) / (QT * WAVES), H, B);
  const int nst = smml_deform_attn_nst(N);
  hipStream_t st = (hipStream_t)stream;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
  if (dtype == 1) launch_fwd16<_Float16>(grid, st, relu_masks != nullptr, posdim, q, k, v, vs, gq, cp, out, lse, logits16, relu_masks, N, J, H, G, nst, scale, dc, opts);
  else launch_fwd16<__bf16>(grid, st, relu_masks != nullptr, posdim, q, k, v, vs, gq, cp, out, lse, logits16, relu_masks, N, J, H, G, nst, scale, dc, opts);
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
  SMML_LAUNCH_CHECK("smml_deform_attn16_fwd");
  return SMML_OK;
}

int smml_deform_attn16_bwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* w1,
                           const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, const float* out,
                           const float* dout, const float* lse, const unsigned short* logits16, const unsigned short* relu_masks,
                           unsigned short* dlogits16, float* dq, float* dk, float* dv, float* dvs, float* dw1, float* db1,
                           float* dw2, float* db2, float* dw3, float* db3, void* workspace, size_t workspace_bytes, int B, int N,
                           int J, int H, int G, int posdim, float scale, float dropout_p, unsigned long long dropout_seed,
                           int dtype, void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts) {
  int rc = check16("smml_deform_attn16_bwd", B, N, J, H, G, posdim, dtype);
  if (rc) return rc;
  SMML_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "smml_deform_attn16_bwd: dropout_p must be in [0, 1)");
  SMML_REQUIRE(q && k && v && vs && gq && w1 && b1 && w2 && b2 && w3 && b3 && out && dout && lse && logits16 &&
                   dlogits16 && dq && dk && dv && dvs && dw1 && db1 && dw2 && db2 && dw3 && db3 && workspace,
               "smml_deform_attn16_bwd: null pointer");        // relu_masks may be null: layer 2 is then recomputed (table-forward calls)
  SMML_REQUIRE(workspace_bytes >= smml_deform_attn_bwd_workspace_bytes(B, N, J, H),
               "smml_deform_attn16_bwd: workspace too small (%zu < %zu)", workspace_bytes,
               smml_deform_attn_bwd_workspace_bytes(B, N, J, H));
  SMML_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "smml_deform_attn16_bwd: workspace must be 16-byte aligned");
  const DropCfg dc = make_drop(dropout_p, dropout_seed, opts);
  CpbParams cp{w1, b1, w2, b2, w3, b3};
  hipStream_t st = (hipStream_t)stream;
  const int nst = smml_deform_attn_nst(N);
  const int qtiles = (N + QT * WAVES - 1) / (QT * WAVES);
  dim3 block(256);
  const BwdWorkspace wsl = bwd_workspace(B, N, J, H);
  float* wsf = reinterpret_cast<float*>(workspace);
  // pass 1: d scores (bf16), dQ
  if (dtype == 1)
    hipLaunchKernelGGL(deform16_bwd_dq_kernel<_Float16>, dim3(qtiles, H, B), block, 0, st, k, v, out, dout, lse, logits16, dlogits16, dq, N, J, H, nst, scale, dc);
  else
    hipLaunchKernelGGL(deform16_bwd_dq_kernel<__bf16>, dim3(qtiles, H, B), block, 0, st, k, v, out, dout, lse, logits16, dlogits16, dq, N, J, H, nst, scale, dc);
  SMML_LAUNCH_CHECK("smml_deform_attn16_bwd/dq");
  // pass 2: dK, dV (query-sliced partial sums, then a fixed-order reduction)
  {
    const int nkg = (J + DKV_KEYS - 1) / DKV_KEYS, nqt = (N + QT - 1) / QT;
    const int parts = dkv_parts(B, N, J, H), tpp = (nqt + parts - 1) / parts;
    const int nslices = parts * H * B;
    const dim3 gk(((nslices + 7) / 8) * 8 * nkg);
    hipLaunchKernelGGL(deform16_bwd_dkv_kernel, gk, block, 0, st, q, dout, lse, logits16, dlogits16, wsf + wsl.dkp, wsf + wsl.dvp, N, J, H, nst, nkg, tpp, parts, B, dc);
    SMML_LAUNCH_CHECK("smml_deform_attn16_bwd/dkv");
    const size_t n4 = (size_t)B * J * H * DH / 4;
    hipLaunchKernelGGL(dkv_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), block, 0, st,
                       reinterpret_cast<const float4*>(wsf + wsl.dkp), reinterpret_cast<const float4*>(wsf + wsl.dvp),
                       reinterpret_cast<float4*>(dk), reinterpret_cast<float4*>(dv), n4, parts, scale);
    SMML_LAUNCH_CHECK("smml_deform_attn16_bwd/dkv_reduce");
  }
  // pass 3: position-bias MLP backward
  {
    float* slab = wsf;
    const size_t lds = ((size_t)CPB2_TAB + WAVES * CPB2_WAVE_LDS + WAVES * CPB_SLAB) * sizeof(float);
    if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);   // brackets the position-bias backward kernel only
    u16* mko = opts ? opts->export_masks : nullptr;
    const dim3 gc(qtiles, H, B);
    if (relu_masks) {
      if (posdim == 2)
        hipLaunchKernelGGL((cpb16_bwd_kernel<2>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko);
      else if (pdx_of(posdim, opts) == 3)
        hipLaunchKernelGGL((cpb16_bwd_kernel<3>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko);
      else
        hipLaunchKernelGGL((cpb16_bwd_kernel<1>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko);
    } else if (opts && opts->mask_table) {          // decisions from the mask table
      const int cells = posdim == 2 ? 1024 : 16384;
      MaskTab mt;
      mt.tab = opts->mask_table;
      mt.invh = (float)((double)cells / (2.0 * (double)opts->mask_table_pmax));
      mt.off = 0.5f * (float)cells;
      mt.imax = (float)cells - 0.5f;
      if (posdim == 2 && mko)
        hipLaunchKernelGGL((cpb16_bwd_kernel<2, 2, true>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko, mt);
      else if (posdim == 2)
        hipLaunchKernelGGL((cpb16_bwd_kernel<2, 2, false>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko, mt);
      else if (mko)
        hipLaunchKernelGGL((cpb16_bwd_kernel<1, 2, true>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko, mt);
      else
        hipLaunchKernelGGL((cpb16_bwd_kernel<1, 2, false>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko, mt);
    } else if (mko) {                   // layer 2 recomputed per pair
      if (posdim == 2)
        hipLaunchKernelGGL((cpb16_bwd_kernel<2, 1, true>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko);
      else
        hipLaunchKernelGGL((cpb16_bwd_kernel<1, 1, true>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko);
    } else {
      if (posdim == 2)
        hipLaunchKernelGGL((cpb16_bwd_kernel<2, 1, false>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko);
      else
        hipLaunchKernelGGL((cpb16_bwd_kernel<1, 1, false>), gc, block, lds, st, dlogits16, relu_masks, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst, mko);
    }
    if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
    SMML_LAUNCH_CHECK("smml_deform_attn16_bwd/cpb");
    const int nwg = qtiles * H * B;
    {
      const long long threads = (long long)B * G * J * 4;
      hipLaunchKernelGGL(dvs_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                         reinterpret_cast<const float2*>(wsf + wsl.dvs), dvs, B, G, H, qtiles, J, posdim);
    }
    const int nchunks = min(CPB_RED_CHUNKS, nwg), chunk = (nwg + nchunks - 1) / nchunks;
    hipLaunchKernelGGL(cpb_partial_kernel, dim3((CPB_SLAB + 63) / 64, nchunks), dim3(256), 0, st, slab, nwg, H / G, qtiles, H, chunk, wsf + wsl.partial);
    hipLaunchKernelGGL(cpb_final_kernel, dim3((CPB_SLAB + 255) / 256), dim3(256), 0, st, wsf + wsl.partial, nchunks, H / G, dw1, db1, dw2, db2, dw3, db3, posdim);
    SMML_LAUNCH_CHECK("smml_deform_attn16_bwd/reduce");
  }
  return SMML_OK;
}

// ---- position bias per linear region (cpb_regions.h) in the 16-bit compute modes: the bias is the fp32 lookup of the fp32-grade path, the
// attention core runs on single-term T operands with fp16 scores / bf16 d scores (2-D signed-log offsets, one head per offset group)
int smml_deform_attn16_region_fwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* w1,
                                  const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, const void* tables,
                                  float* out, float* lse, unsigned short* logits16, unsigned short* region_ids, int B, int N, int J, int H,
                                  float scale, float dropout_p, unsigned long long dropout_seed, int dtype, void* ev_start, void* ev_stop,
                                  void* stream, const SmmlDeformOpts* opts) {
  int rc = check_region("smml_deform_attn16_region_fwd", B, N, J, H);
  if (rc) return rc;
  SMML_REQUIRE(dtype == 0 || dtype == 1, "smml_deform_attn16_region_fwd: dtype must be 0 (bf16) or 1 (fp16), got %d", dtype);
  SMML_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "smml_deform_attn16_region_fwd: dropout_p must be in [0, 1)");
  SMML_REQUIRE(q && k && v && vs && gq && w1 && b1 && w2 && b2 && w3 && b3 && tables && out && lse, "smml_deform_attn16_region_fwd: null pointer");
  SMML_REQUIRE((logits16 == nullptr) == (region_ids == nullptr),
               "smml_deform_attn16_region_fwd: logits16 and region_ids are saved together (training) or not at all");
  const DropCfg dc = make_drop(dropout_p, dropout_seed, opts);
  const int lcap = (opts && opts->region_lds_cap > 0) ? (opts->region_lds_cap < RG_LCAP ? opts->region_lds_cap : RG_LCAP) : RG_LCAP;
  CpbParams cp{w1, b1, w2, b2, w3, b3};
  const RegionView rv = region_view(const_cast<void*>(tables));
  dim3 grid((N + QT * WAVES - 1) / (QT * WAVES), H, B), block(256);
  const int nst = smml_deform_attn_nst(N);
  hipStream_t st = (hipStream_t)stream;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
  if (dtype == 1) {
    if (region_ids)
      hipLaunchKernelGGL((deform_region_fwd_kernel<true, _Float16>), grid, block, 0, st, q, k, v, vs, gq, cp, rv, out, lse, logits16, region_ids, N, J, H, nst, scale, dc, lcap);
    else
      hipLaunchKernelGGL((deform_region_fwd_kernel<false, _Float16>), grid, block, 0, st, q, k, v, vs, gq, cp, rv, out, lse, logits16, region_ids, N, J, H, nst, scale, dc, lcap);
  } else {
    if (region_ids)
      hipLaunchKernelGGL((deform_region_fwd_kernel<true, __bf16>), grid, block, 0, st, q, k, v, vs, gq, cp, rv, out, lse, logits16, region_ids, N, J, H, nst, scale, dc, lcap);
    else
      hipLaunchKernelGGL((deform_region_fwd_kernel<false, __bf16>), grid, block, 0, st, q, k, v, vs, gq, cp, rv, out, lse, logits16, region_ids, N, J, H, nst, scale, dc, lcap);
  }
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
  SMML_LAUNCH_CHECK("smml_deform_attn16_region_fwd");
  return SMML_OK;
}

/* workspace: smml_deform_attn_region_bwd_workspace_bytes (the fp32-grade region backward's), 256-byte aligned */
int smml_deform_attn16_region_bwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* w1,
                                  const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, const void* tables,
                                  const float* out, const float* dout, const float* lse, const unsigned short* logits16,
                                  const unsigned short* region_ids, unsigned short* dlogits16, float* dq, float* dk, float* dv, float* dvs,
                                  float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, void* workspace,
                                  size_t workspace_bytes, int B, int N, int J, int H, float scale, float dropout_p,
                                  unsigned long long dropout_seed, int dtype, void* ev_start, void* ev_stop, void* stream,
                                  const SmmlDeformOpts* opts) {
  int rc = check_region("smml_deform_attn16_region_bwd", B, N, J, H);
  if (rc) return rc;
  SMML_REQUIRE(dtype == 0 || dtype == 1, "smml_deform_attn16_region_bwd: dtype must be 0 (bf16) or 1 (fp16), got %d", dtype);
  SMML_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "smml_deform_attn16_region_bwd: dropout_p must be in [0, 1)");
  SMML_REQUIRE(q && k && v && vs && gq && w1 && b1 && w2 && b2 && w3 && b3 && tables && out && dout && lse && logits16 && region_ids &&
                   dlogits16 && dq && dk && dv && dvs && dw1 && db1 && dw2 && db2 && dw3 && db3 && workspace,
               "smml_deform_attn16_region_bwd: null pointer");
  const RegionBwdPlan pl = region_bwd_plan(B, N, J, H);
  SMML_REQUIRE(workspace_bytes >= pl.total, "smml_deform_attn16_region_bwd: workspace too small (%zu < %zu)", workspace_bytes, pl.total);
  SMML_REQUIRE((reinterpret_cast<size_t>(workspace) & 255) == 0, "smml_deform_attn16_region_bwd: workspace must be 256-byte aligned");
  SMML_REQUIRE(pl.wpk >= 1, "smml_deform_attn16_region_bwd: too many keys (%d)", J);
  const DropCfg dc = make_drop(dropout_p, dropout_seed, opts);
  const int lcap = (opts && opts->region_lds_cap > 0) ? (opts->region_lds_cap < RG_LCAP ? opts->region_lds_cap : RG_LCAP) : RG_LCAP;
  CpbParams cp{w1, b1, w2, b2, w3, b3};
  hipStream_t st = (hipStream_t)stream;
  const int nst = smml_deform_attn_nst(N);
  const int qtiles = (N + QT * WAVES - 1) / (QT * WAVES);
  dim3 block(256);
  const BwdWorkspace wsl = bwd_workspace(B, N, J, H);
  float* wsf = reinterpret_cast<float*>(workspace);
  char* wsb = reinterpret_cast<char*>(workspace);
  unsigned* amax = reinterpret_cast<unsigned*>(wsb + pl.amax);
  (void)hipMemsetAsync(wsb + pl.amax, 0, pl.dvs - pl.amax, st);        // amax | hist | grad are contiguous
  // pass 1: d scores (bf16), dQ, max |d scores|
  if (dtype == 1)
    hipLaunchKernelGGL(deform16_bwd_dq_kernel<_Float16>, dim3(qtiles, H, B), block, 0, st, k, v, out, dout, lse, logits16, dlogits16, dq, N, J, H, nst, scale, dc, amax);
  else
    hipLaunchKernelGGL(deform16_bwd_dq_kernel<__bf16>, dim3(qtiles, H, B), block, 0, st, k, v, out, dout, lse, logits16, dlogits16, dq, N, J, H, nst, scale, dc, amax);
  SMML_LAUNCH_CHECK("smml_deform_attn16_region_bwd/dq");
  // pass 2: dK, dV (query-sliced partial sums, then a fixed-order reduction)
  {
    const int nkg = (J + DKV_KEYS - 1) / DKV_KEYS, nqt = (N + QT - 1) / QT;
    const int parts = dkv_parts(B, N, J, H), tpp = (nqt + parts - 1) / parts;
    const int nslices = parts * H * B;
    const dim3 gk(((nslices + 7) / 8) * 8 * nkg);
    hipLaunchKernelGGL(deform16_bwd_dkv_kernel, gk, block, 0, st, q, dout, lse, logits16, dlogits16, wsf + wsl.dkp, wsf + wsl.dvp, N, J, H, nst, nkg, tpp, parts, B, dc);
    SMML_LAUNCH_CHECK("smml_deform_attn16_region_bwd/dkv");
    const size_t n4 = (size_t)B * J * H * DH / 4;
    hipLaunchKernelGGL(dkv_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), block, 0, st,
                       reinterpret_cast<const float4*>(wsf + wsl.dkp), reinterpret_cast<const float4*>(wsf + wsl.dvp),
                       reinterpret_cast<float4*>(dk), reinterpret_cast<float4*>(dv), n4, parts, scale);
    SMML_LAUNCH_CHECK("smml_deform_attn16_region_bwd/dkv_reduce");
  }
  // pass 3: position bias per region on the bf16 d scores
  return region_bias_bwd_launch<u16>("smml_deform_attn16_region_bwd", dlogits16, region_ids, vs, gq, cp, tables, wsb, pl, B, N, J, H, nst, lcap, dvs,
                                     dw1, db1, dw2, db2, dw3, db3, ev_start, ev_stop, st);
}

// mask table of the table-forward backward (include/smml.h): cells per axis, and the kernel that fills one
int smml_cpb_mask_table_cells(int posdim) { return posdim == 2 ? 1024 : 16384; }
int smml_cpb_mask_table(const float* w1, const float* b1, const float* w2, const float* b2, unsigned short* table, int posdim, float pmax,
                        void* stream) {
  SMML_REQUIRE(w1 && b1 && w2 && b2 && table, "smml_cpb_mask_table: null pointer");
  SMML_REQUIRE(posdim == 1 || posdim == 2, "smml_cpb_mask_table: posdim must be 1 or 2 (got %d)", posdim);
  SMML_REQUIRE(pmax > 0.f, "smml_cpb_mask_table: pmax must be positive");
  CpbParams cp{w1, b1, w2, b2, nullptr, nullptr};
  const int cpa = smml_cpb_mask_table_cells(posdim);
  const long long ncell = posdim == 2 ? (long long)cpa * cpa : cpa;
  const dim3 grid((unsigned)((ncell + 255) / 256));
  if (posdim == 2) hipLaunchKernelGGL(cpb_mask_table_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, cp, table, cpa, pmax);
  else hipLaunchKernelGGL(cpb_mask_table_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, cp, table, cpa, pmax);
  SMML_LAUNCH_CHECK("smml_cpb_mask_table");
  return SMML_OK;
}
// ---- table mode (tabulated position bias): same contract as the two entry points above with `table` [o, table_g^posdim] in place of
// the six MLP tensors and d table in place of their gradients
int smml_deform_attn_table_points(int posdim) { return posdim == 2 ? TABLE_G2 : TABLE_G1; }

size_t smml_deform_attn_table_bwd_workspace_bytes(int B, int N, int J, int H, int posdim) {
  if (!deform_dims_ok(B, N, J, H)) return 0;
  return table_workspace(B, N, J, H, posdim == 2 ? TABLE_G2 * TABLE_G2 : TABLE_G1).total * sizeof(float);
}

int smml_deform_attn_table_fwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* table,
                               float* out, float* lse, unsigned short* logits16, int B, int N, int J, int H, int G, int posdim,
                               int table_g, float table_pmax, float scale, float dropout_p, unsigned long long dropout_seed, int dtype,
                               void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts) {
  int rc = check16("smml_deform_attn_table_fwd", B, N, J, H, G, posdim, dtype);
  if (rc) return rc;
  rc = check_table("smml_deform_attn_table_fwd", posdim, table_g, table_pmax, opts);
  if (rc) return rc;
  SMML_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "smml_deform_attn_table_fwd: dropout_p must be in [0, 1)");
  SMML_REQUIRE(q && k && v && vs && gq && table && out && lse, "smml_deform_attn_table_fwd: null pointer");
  const DropCfg dc = make_drop(dropout_p, dropout_seed, opts);
  const TabCfg tc = make_tab(table, table_g, table_pmax);
  dim3 grid((N + QT * WAVES - 1) / (QT * WAVES), H, B);
  const int nst = smml_deform_attn_nst(N);
  hipStream_t st = (hipStream_t)stream;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
  if (dtype == 1) launch_fwd_table<_Float16>(grid, st, logits16 != nullptr, posdim, q, k, v, vs, gq, out, lse, logits16, N, J, H, G, nst, scale, dc, tc);
  else launch_fwd_table<__bf16>(grid, st, logits16 != nullptr, posdim, q, k, v, vs, gq, out, lse, logits16, N, J, H, G, nst, scale, dc, tc);
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
  SMML_LAUNCH_CHECK("smml_deform_attn_table_fwd");
  return SMML_OK;
}

int smml_deform_attn_table_bwd(const float* q, const float* k, const float* v, const float* vs, const float* gq, const float* table,
                               const float* out, const float* dout, const float* lse, const unsigned short* logits16,
                               unsigned short* dlogits16, float* dq, float* dk, float* dv, float* dvs, float* dtable, void* workspace,
                               size_t workspace_bytes, int B, int N, int J, int H, int G, int posdim, int table_g, float table_pmax,
                               int grid_h, int grid_w, float scale, float dropout_p, unsigned long long dropout_seed, int dtype,
                               void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts) {
  int rc = check16("smml_deform_attn_table_bwd", B, N, J, H, G, posdim, dtype);
  if (rc) return rc;
  rc = check_table("smml_deform_attn_table_bwd", posdim, table_g, table_pmax, opts);
  if (rc) return rc;
  SMML_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "smml_deform_attn_table_bwd: dropout_p must be in [0, 1)");
  SMML_REQUIRE(q && k && v && vs && gq && table && out && dout && lse && logits16 && dlogits16 && dq && dk && dv && dvs && dtable && workspace,
               "smml_deform_attn_table_bwd: null pointer");
  const size_t need = smml_deform_attn_table_bwd_workspace_bytes(B, N, J, H, posdim);
  SMML_REQUIRE(workspace_bytes >= need, "smml_deform_attn_table_bwd: workspace too small (%zu < %zu)", workspace_bytes, need);
  SMML_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "smml_deform_attn_table_bwd: workspace must be 16-byte aligned");
  const DropCfg dc = make_drop(dropout_p, dropout_seed, opts);
  const TabCfg tc = make_tab(table, table_g, table_pmax);
  hipStream_t st = (hipStream_t)stream;
  const int nst = smml_deform_attn_nst(N);
  const int qtiles = (N + QT * WAVES - 1) / (QT * WAVES);
  dim3 block(256);
  const BwdWorkspace wsl = bwd_workspace(B, N, J, H);
  const int cells = posdim == 2 ? TABLE_G2 * TABLE_G2 : TABLE_G1;
  const TableWorkspace tw = table_workspace(B, N, J, H, cells);
  float* wsf = reinterpret_cast<float*>(workspace);
  const int o = H / G;
  (void)hipMemsetAsync(dtable, 0, (size_t)o * cells * sizeof(float), st);
  // pass 1: d scores (bf16), dQ
  if (dtype == 1)
    hipLaunchKernelGGL(deform16_bwd_dq_kernel<_Float16>, dim3(qtiles, H, B), block, 0, st, k, v, out, dout, lse, logits16, dlogits16, dq, N, J, H, nst, scale, dc);
  else
    hipLaunchKernelGGL(deform16_bwd_dq_kernel<__bf16>, dim3(qtiles, H, B), block, 0, st, k, v, out, dout, lse, logits16, dlogits16, dq, N, J, H, nst, scale, dc);
  SMML_LAUNCH_CHECK("smml_deform_attn_table_bwd/dq");
  // pass 2: dK, dV
  {
    const int nkg = (J + DKV_KEYS - 1) / DKV_KEYS, nqt = (N + QT - 1) / QT;
    const int parts = dkv_parts(B, N, J, H), tpp = (nqt + parts - 1) / parts;
    const int nslices = parts * H * B;
    const dim3 gk(((nslices + 7) / 8) * 8 * nkg);
    hipLaunchKernelGGL(deform16_bwd_dkv_kernel, gk, block, 0, st, q, dout, lse, logits16, dlogits16, wsf + wsl.dkp, wsf + wsl.dvp, N, J, H, nst, nkg, tpp, parts, B, dc);
    SMML_LAUNCH_CHECK("smml_deform_attn_table_bwd/dkv");
    const size_t n4 = (size_t)B * J * H * DH / 4;
    hipLaunchKernelGGL(dkv_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), block, 0, st,
                       reinterpret_cast<const float4*>(wsf + wsl.dkp), reinterpret_cast<const float4*>(wsf + wsl.dvp),
                       reinterpret_cast<float4*>(dk), reinterpret_cast<float4*>(dv), n4, parts, scale);
    SMML_LAUNCH_CHECK("smml_deform_attn_table_bwd/dkv_reduce");
  }
  // pass 3: d table (histogram of d bias over the table cells) and d vs
  {
    const int S = table_slices(B, N, J, H), nkb = (J + 63) / 64, ntq = (N + 31) / 32, tps = (ntq + S - 1) / S;
    float* slab = wsf + tw.slab;
    float2* rows = reinterpret_cast<float2*>(wsf + tw.rows);
    const dim3 gt(nkb * S, H, B), bt(64 * TBW);
    if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);   // brackets the position-bias backward kernel only
    // queries on a regular grid (grid_h x grid_w = N, both <= 128): d table on the matrix pipe, d vs from the per-pair kernel without its histogram
    const bool on_grid = posdim == 2 && grid_h > 0 && grid_w > 0 && (long long)grid_h * grid_w == N && grid_h <= TABLE_GRID_MAX && grid_w <= TABLE_GRID_MAX;
    int wg_per_head = nkb * S;
    if (on_grid) {
      hipLaunchKernelGGL((cpb_table_bwd_kernel<2, TABLE_G2, false>), gt, bt, 0, st, dlogits16, vs, gq, tc, slab, rows, N, J, H, G, nst, S, tps);
      wg_per_head = (J + TABLE_GRID_KEYS - 1) / TABLE_GRID_KEYS;
      const int need_h = max(table_sheet_halves(grid_h, grid_w), ((N + 31) / 32) * 32);
      const size_t dyn = (((size_t)need_h * 2 + 15) / 16) * 16;
      static size_t dyn_set = 0;
      if (dyn > dyn_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cpb_table_grid_bwd_kernel<TABLE_G2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cpb_table_grid_bwd_kernel<TABLE_G2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        dyn_set = dyn;
      }
      if (grid_w % 4 == 0)
        hipLaunchKernelGGL((cpb_table_grid_bwd_kernel<TABLE_G2, true>), dim3(wg_per_head, H, B), dim3(64 * TGW), dyn, st, dlogits16, vs, gq, tc, slab, N, J, H, G, nst, grid_h, grid_w);
      else
        hipLaunchKernelGGL((cpb_table_grid_bwd_kernel<TABLE_G2, false>), dim3(wg_per_head, H, B), dim3(64 * TGW), dyn, st, dlogits16, vs, gq, tc, slab, N, J, H, G, nst, grid_h, grid_w);
    } else if (posdim == 2)
      hipLaunchKernelGGL((cpb_table_bwd_kernel<2, TABLE_G2>), gt, bt, 0, st, dlogits16, vs, gq, tc, slab, rows, N, J, H, G, nst, S, tps);
    else
      hipLaunchKernelGGL((cpb_table_bwd_kernel<1, TABLE_G1>), gt, bt, 0, st, dlogits16, vs, gq, tc, slab, rows, N, J, H, G, nst, S, tps);
    if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
    SMML_LAUNCH_CHECK("smml_deform_attn_table_bwd/cpb");
    const long long outs = (long long)B * G * J;
    hipLaunchKernelGGL(dvs_table_reduce_kernel, dim3((unsigned)((outs + 255) / 256)), dim3(256), 0, st, rows, dvs, B, G, H, S * TBW, J, posdim);
    const int nwg = wg_per_head * H * B, nchunks = min(32, nwg), chunk = (nwg + nchunks - 1) / nchunks;
    hipLaunchKernelGGL(table_hist_reduce_kernel, dim3((cells + 255) / 256, nchunks), dim3(256), 0, st, slab, dtable, cells, nwg, wg_per_head, H, o, chunk);
    SMML_LAUNCH_CHECK("smml_deform_attn_table_bwd/reduce");
  }
  return SMML_OK;
}

}  // extern "C"
