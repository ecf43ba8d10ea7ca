// Fused deformable cross-attention core for gfx950 (MI355X): QK^T + continuous position bias (CPB)
// + softmax + PV, forward and backward; fp32 results, the position-bias MLP as split products on the 16-bit matrix pipe.
//
// Replaces the op sequence of the reference at
//   models/DeformableAttention2D.py:284-312 (sim, rel_pos_bias, softmax, attn @ v) and :120-157 (CPB)
//   models/DeformableAttention1D.py:205-232 and :60-102
// without ever materialising the [pairs, 32] hidden activations of the CPB MLP (2.95 GB each at
// the reference's B = 8).
//
// Data layout (all fp32, token-major, HBM resident):
//   q [B, N, H*64]   k, v [B, J, H*64]   vs [B*G, J, PD]  (normalised sample positions)
//   gq [N, PD]       (normalised query grid)             out [B, N, H*64]   lse [B, H, N]
//   logits_t / dlogits_t [B, H, J, NST]  (scores incl. bias, key-major so that a wave's 32 queries
//   are contiguous; NST = N rounded up to 32)            PD = 2 (2-D module) or 1 (1-D module)
//
// Matrix-core mapping (one wave = 32 queries on the lane axis):
//   S^T[key,query]  = K . Q^T            v_mfma_f32_32x32x2_f32; A = K tile from LDS, B = Q tile from LDS
//   h1[ch,query]    = relu(W1 p + b1)    two v_mfma_f32_32x32x16_bf16 (every factor in three bf16 terms)
//   D[out,query]    = W2 . h1 + b2       split-fp16 product on v_mfma_f32_32x32x16_f16, W2 (hi / mid / lo) in VGPRs,
//                                        h1 (hi / lo) converted straight into the B operand
//   O^T[d,query]   += V^T . P^T          v_mfma_f32_32x32x2_f32; the softmax'd accumulator registers are the B operand
//                                        as they stand (the sum runs over the accumulator's row index)
// The 32x32 CPB layer is 2048 of the 2240 flop per (query, key) pair.  On gfx950 MFMA time and vector time of a SIMD add
// up (tests/microbench/overlap_probe.hip), so the kernels minimise both instruction counts; results are fp32-grade
// (DESIGN.md section 4 for the error bounds and the measurements behind them).
#include "deform_common.h"
#include "cpb_regions.h"

namespace {


// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int PDX, bool SAVE>
__global__ __launch_bounds__(256, SMML_FWD_WPS) void deform_attn_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ VS, const float* __restrict__ GQ, CpbParams cp, float* __restrict__ O,
    float* __restrict__ LSE, float* __restrict__ LT, unsigned short* __restrict__ MK, int N, int J, int H, int G, int NST,
    float scale, DropCfg dc_in) {
  constexpr int PD = PosCfg<PDX>::PD;
  constexpr bool RAW = PosCfg<PDX>::RAW;
  const DropCfg dc = drop_resolve(dc_in);
#if SMML_FWD_QK16
  __shared__ __attribute__((aligned(16))) _Float16 Kp[2][KT * FRLD];         // K tile, fp16 hi / lo planes, row image (A operand of S^T)
  __shared__ __attribute__((aligned(16))) _Float16 Vp[2][KT * FTLD];         // V tile, hi / lo planes, read transposed (A operand of O^T)
  __shared__ __attribute__((aligned(16))) _Float16 Qp[WAVES][2][QT * FRLD];  // per-wave scaled Q tile, hi / lo planes, row image
#else
  __shared__ float Ks[DH][KT + 1];           // K tile, d-major (A operand of S^T)
  __shared__ float Vs[KT][DH];               // V tile, key-major (A operand of O^T)
  __shared__ float Qs[WAVES][DH][QT];        // per-wave scaled Q tile, d-major
#endif
  __shared__ float vsl[KT][2];               // sample positions of the tile's keys
  __shared__ float biasT[WAVES][KT][QT];     // per-wave bias tile [key][query]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (QT * WAVES) + wave * QT;
  const int o = H / G, g = h / o, oi = h - g * o;
  const int HD = H * DH;
  const bool qvalid = (q0 + c) < N;
  const int qi = qvalid ? (q0 + c) : (N - 1);

  // Q tile of this wave, pre-multiplied by the softmax scale, parked in LDS d-major (B operand of S^T):
  // lane (query c, half hf) owns d = 32 hf .. 32 hf + 31
  {
    const float4* qp = reinterpret_cast<const float4*>(Q + ((size_t)b * N + qi) * HD + h * DH + hf * 32);
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      const float4 t = qp[s4];
#if SMML_FWD_QK16
      uint2v hi, lo;
      split4_h2(make_float4(t.x * scale, t.y * scale, t.z * scale, t.w * scale), hi, lo);
      *reinterpret_cast<uint2v*>(&Qp[wave][0][c * FRLD + 32 * hf + 4 * s4]) = hi;
      *reinterpret_cast<uint2v*>(&Qp[wave][1][c * FRLD + 32 * hf + 4 * s4]) = lo;
#else
      Qs[wave][32 * hf + 4 * s4 + 0][c] = t.x * scale; Qs[wave][32 * hf + 4 * s4 + 1][c] = t.y * scale;
      Qs[wave][32 * hf + 4 * s4 + 2][c] = t.z * scale; Qs[wave][32 * hf + 4 * s4 + 3][c] = t.w * scale;
#endif
    }
  }
#if SMML_FWD_QK16
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);      // transposed-read lane map
#endif
  const float gq0 = GQ[(size_t)qi * PD];
  const float gq1 = (PD == 2) ? GQ[(size_t)qi * PD + 1] : 0.f;

  // CPB constants: b2 / w3 in accumulator layout (VGPRs)
  float w3v[16];
  floatx16 b2acc;                               // b2 in accumulator layout: the chain's initial C operand
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int oc = acc_row(s, hf);              // output channel held in accumulator register s
    b2acc[s] = 2.f * cp.b2[oc];                 // the chain runs on 2 h1 (relu2), so it delivers 2 (W2 h1 + b2) ...
    w3v[s] = 0.25f * cp.w3[oi * CH + oc];       // ... and relu2 of that is 4 relu(.)
  }
  // Layer 1 runs on the matrix pipe as ONE bf16 MFMA: x[ch][q] = w1x[ch] p0[q] + w1y[ch] p1[q] + b1[ch] with the weights
  // and the positions each split into three bf16 terms (h + m + l = the fp32 value to 2^-24, fp32's exponent range) and
  // b1 riding in as the C operand (exact).  Eight of the nine cross products (only l l, <= 2^-32, is left out) fill the
  // 16 K slots; the B operand is the three packed conversion results as they stand plus one select:
  //   B, both halves:  p0_h p1_h  p0_m p1_m  p0_l p1_l   and in slots 6, 7:  p0_h p1_h (half 0) / p0_m p1_m (half 1)
  //   A, half 0:       x_h  y_h   x_h  y_h   x_h  y_h    x_l  y_l
  //   A, half 1:       x_m  y_m   x_m  y_m   x_m  y_m    x_l  y_l
  // The result arrives in accumulator layout, so operand slot (K-block kb, element j) of the 32x32 layer carries hidden
  // channel acc_row(8 kb + j, hf).
  bf16x8 a1;
  floatx16 b1acc;
  {
    a1 = cpb_l1_weights_q(cp.w1[c * PD], (PD == 2) ? cp.w1[c * PD + 1] : 0.f, hf);
#pragma unroll
    for (int s = 0; s < 16; ++s) b1acc[s] = cp.b1[acc_row(s, hf)];
  }
  // W2 as the A operand of the fp16 form: lane (out = c, half hf), K-block kb, element j <-> in = acc_row(8 kb + j, hf)
  half8 w2h[2], w2m[2], w2l[2];
  {
    float amax = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) amax = fmaxf(amax, fabsf(cp.w2[c * CH + acc_row(s, hf)]));
    // W2 2^k with the largest element in (2^12, 2^13]: hi / mid fp16 terms normal for elements down to 2^-16 of the largest, lo
    // down to 2^-5 (below that its absolute error is 2^-37 of the largest element: irrelevant);
    // the chain then delivers 2^k 2 (W2 h1 + b2): b2 rides in scaled, w3 carries 2^-k (powers of two: exact)
    const float lift = SMML_LIFT_FWD ? pow2_lift(wave_max_all(amax), 8192.f, -8.f, 24.f) : 1.f, unlift = 1.f / lift;
#pragma unroll
    for (int s = 0; s < 16; ++s) { b2acc[s] *= lift; w3v[s] *= unlift; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float wv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[j] = cp.w2[c * CH + acc_row(8 * kb + j, hf)] * lift;
      split8_3(wv, w2h[kb], w2m[kb], w2l[kb]);
    }
  }
  const float b3h = (hf == 0) ? cp.b3[oi] : 0.f;

  floatx16 oacc0 = {0}, oacc1 = {0};
  float m_run = -INFINITY, l_run = 0.f;
  const float* Kb = K + (size_t)b * J * HD + h * DH;
  const float* Vb = V + (size_t)b * J * HD + h * DH;
  const float* VSb = VS + (size_t)(b * G + g) * J * PD;
  // score-shaped tensors (logits_t, dlogits_t, relu_masks) are stored per 32-query tile: [B, H, NST / 32, J, 32] - a wave streams its
  // own contiguous [J][32] block in the forward, the dq pass and the position-bias backward, and the dkv pass reads 32 keys x 128 B
  // = one contiguous 4 KB run per query tile (a key-major [J][NST] row layout made every one of these a 40-KB-strided gather)
  float* LTb = LT ? LT + ((size_t)(b * H + h) * NST + q0) * J : nullptr;              // this wave's [J][32] block
  // layer-2 ReLU masks for the backward (training only): 16 bits per lane and key, element r (hidden channel
  // acc_row(r, hf)) at bit (13 + r) % 16 - the position from which the backward rotates it straight into an fp16 operand
  unsigned short* MKb = MK ? MK + (((size_t)(b * H + h) * NST + q0) * J) * 2 + hf * 32 : nullptr;   // [J][2][32] block, this lane half
  float big;                                                  // 2^100 in an SGPR (v_mul_f32 ... clamp takes no literal)
  asm("s_mov_b32 %0, 0x71800000" : "=s"(big));

  const int ntiles = (J + KT - 1) / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int j0 = kt * KT;
    __syncthreads();
    // cooperative tile load: 32 keys x 64 d, two float4 per thread per tensor
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = (tid >> 4) + 16 * i, d4 = (tid & 15) * 4;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (j0 + key < J) {
        kv = *reinterpret_cast<const float4*>(Kb + (size_t)(j0 + key) * HD + d4);
        vv = *reinterpret_cast<const float4*>(Vb + (size_t)(j0 + key) * HD + d4);
      }
#if SMML_FWD_QK16
      uint2v hi, lo;
      split4_h2(kv, hi, lo);
      *reinterpret_cast<uint2v*>(&Kp[0][key * FRLD + d4]) = hi; *reinterpret_cast<uint2v*>(&Kp[1][key * FRLD + d4]) = lo;
      split4_h2(vv, hi, lo);
      *reinterpret_cast<uint2v*>(&Vp[0][key * FTLD + d4]) = hi; *reinterpret_cast<uint2v*>(&Vp[1][key * FTLD + d4]) = lo;
#else
      Ks[d4 + 0][key] = kv.x; Ks[d4 + 1][key] = kv.y; Ks[d4 + 2][key] = kv.z; Ks[d4 + 3][key] = kv.w;
      *reinterpret_cast<float4*>(&Vs[key][d4]) = vv;
#endif
    }
    if (tid < KT) {
      const int key = j0 + tid;
      vsl[tid][0] = (key < J) ? VSb[(size_t)key * PD] : 0.f;
      vsl[tid][1] = (PD == 2 && key < J) ? VSb[(size_t)key * PD + 1] : 0.f;
    }
    __syncthreads();

    // S^T[key, query] = K . (scale Q)^T
    floatx16 s = {0};
#if SMML_FWD_QK16
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int o = c * FRLD + 16 * st + 8 * hf;
      const half8 kh = *reinterpret_cast<const half8*>(&Kp[0][o]), kl = *reinterpret_cast<const half8*>(&Kp[1][o]);
      const half8 qh = *reinterpret_cast<const half8*>(&Qp[wave][0][o]), ql = *reinterpret_cast<const half8*>(&Qp[wave][1][o]);
      s = mfma16(kl, qh, s);
      s = mfma16(kh, ql, s);
      s = mfma16(kh, qh, s);
    }
#else
#pragma unroll
    for (int st = 0; st < 32; ++st) s = mfma32(Ks[32 * hf + st][c], Qs[wave][32 * hf + st][c], s);
#endif

    // continuous position bias: one MFMA chain per key.  On gfx950 v_mfma_f32_32x32x2_f32 runs at the fp32
    // vector rate and does NOT overlap VALU work of the same SIMD (tests/microbench/mfma_probe.hip: every
    // VALU instruction between two of these MFMAs adds its full issue time), so the loop is written for the
    // fewest vector instructions: b2 rides in as the chain's initial accumulator, no register copies.
    const int nk = min(KT, J - j0);
    auto bias_chain = [&](int jj, bool store_mask) {
      const float p0 = pos_of<RAW>(gq0 - vsl[jj][0]);
      const float p1 = (PD == 2) ? slog1p(gq1 - vsl[jj][1]) : 0.f;
      floatx16 d = b2acc;
      // layer 1 on the matrix pipe, ReLU, fp16 hi / lo split, five MFMAs per K-block
      const floatx16 xacc = cpb_layer1_q(a1, cpb_split_pos(p0, p1), hf, b1acc);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float hv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = relu2(xacc[8 * kb + j]);
        half8 bh, bl;
        split8(hv, bh, bl);
        d = mfma16_split(w2h[kb], w2m[kb], w2l[kb], bh, bl, d);
      }
      // layer 3: two packed-fp32 FMA chains (v_pk_fma_f32 does two channels per issue); b3 rides in half 0's sum
      float2v ta = {b3h, 0.f}, tb = {0.f, 0.f};
      float mb0 = 0.f, mb1 = 0.f, mb2 = 0.f, mb3 = 0.f;   // the 16 mask bits, summed as exact powers of two (four chains)
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        const float2v ra = {relu2(d[r]), relu2(d[r + 1])}, rb = {relu2(d[r + 2]), relu2(d[r + 3])};
        ta = __builtin_elementwise_fma(ra, (float2v){w3v[r], w3v[r + 1]}, ta);
        tb = __builtin_elementwise_fma(rb, (float2v){w3v[r + 2], w3v[r + 3]}, tb);
        if (SAVE) {                         // [d > 0] = clamp(relu2(d) 2^100): exact 0.0 / 1.0, fast-class instructions only
          mb0 = fmaf(fminf(fmaxf(ra[0] * big, 0.f), 1.f), (float)(1u << ((13 + r) & 15)), mb0);
          mb1 = fmaf(fminf(fmaxf(ra[1] * big, 0.f), 1.f), (float)(1u << ((14 + r) & 15)), mb1);
          mb2 = fmaf(fminf(fmaxf(rb[0] * big, 0.f), 1.f), (float)(1u << ((15 + r) & 15)), mb2);
          mb3 = fmaf(fminf(fmaxf(rb[1] * big, 0.f), 1.f), (float)(1u << ((16 + r) & 15)), mb3);
        }
      }
      if (SAVE && store_mask) MKb[(size_t)(j0 + jj) * 64 + c] = (unsigned short)(unsigned)((mb0 + mb1) + (mb2 + mb3));   // rows of padded query lanes exist
      ta += tb;
      biasT[wave][jj][c] = xhalf_sum(ta[0] + ta[1]);   // both halves store the same sum: no exec masking in the loop
    };
#if SMML_FWD_PAIR
    for (int jj = 0; jj < nk; jj += 2) {              // two keys per trip: one chain's MFMA latencies under the other's vector work
      bias_chain(jj, true);
      bias_chain(jj + 1, jj + 1 < nk);                // jj + 1 <= 31: inside the staged tile (zero positions past the last key)
    }
#else
    for (int jj = 0; jj < nk; ++jj) bias_chain(jj, true);
#endif
    wave_lds_fence();

    // bias add, key mask, online softmax
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = acc_row(r, hf);
      const float sv = (key < nk) ? s[r] + biasT[wave][key][c] : -INFINITY;
      s[r] = sv;
      tmax = fmaxf(tmax, sv);
    }
    unsigned keepbits = 0xFFFFu;              // dropout decisions of this lane's 16 keys (bit r)
    if (dc.thresh) {
      const unsigned long long base2 = ((unsigned long long)(b * H + h) * N + qi) * ((J + 1) >> 1) + (j0 >> 1);
      keepbits = 0u;
#pragma unroll
      for (int r = 0; r < 16; r += 2) keepbits |= drop_keep2(dc, base2 + (acc_row(r, hf) >> 1)) << r;     // registers r, r + 1: keys 2 jp, 2 jp + 1
    }
    if (SAVE) {                               // rows are padded to whole workgroup tiles: lanes past N write padding
      if (dc.thresh) {
        tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (acc_row(r, hf) < nk) s[r] = stash_keep(s[r], (keepbits >> r) & 1u);      // finite scores only
          tmax = fmaxf(tmax, s[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = acc_row(r, hf);
        if (key < nk) LTb[(size_t)(j0 + key) * 32 + c] = s[r];
      }
    }
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
    if (dc.thresh) {                              // dropped / rescaled probabilities feed P.V only
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] *= ((keepbits >> r) & 1u) ? dc.keep_scale : 0.f;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc0[r] *= alpha; oacc1[r] *= alpha; }

    // O^T[d, query] += V^T . P^T   (accumulator registers of P^T are the B operand as they stand)
#if SMML_FWD_QK16
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {            // accumulator registers 8 kb .. 8 kb + 7 of P^T are the B fragment of k-step kb
      float p8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) p8[j] = s[8 * kb + j];
      half8 ph, pl;
      split8(p8, ph, pl);
      const int ro = (16 * kb + 4 * hf + trq) * FTLD + trc;
      const half8 vh0 = lds_frag_tr_h(&Vp[0][ro], &Vp[0][ro + 8 * FTLD]), vl0 = lds_frag_tr_h(&Vp[1][ro], &Vp[1][ro + 8 * FTLD]);
      const half8 vh1 = lds_frag_tr_h(&Vp[0][ro + 32], &Vp[0][ro + 32 + 8 * FTLD]), vl1 = lds_frag_tr_h(&Vp[1][ro + 32], &Vp[1][ro + 32 + 8 * FTLD]);
      oacc0 = mfma16(vl0, ph, oacc0); oacc0 = mfma16(vh0, pl, oacc0); oacc0 = mfma16(vh0, ph, oacc0);
      oacc1 = mfma16(vl1, ph, oacc1); oacc1 = mfma16(vh1, pl, oacc1); oacc1 = mfma16(vh1, ph, oacc1);
    }
#else
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = acc_row(r, hf);
      oacc0 = mfma32(Vs[key][c], s[r], oacc0);
      oacc1 = mfma32(Vs[key][32 + c], s[r], oacc1);
    }
#endif
    wave_lds_fence();
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
// backward pass 1 (query owners): dS^T = P^T (dP^T - delta), dQ = scale * dS K
//   reads logits_t, writes dlogits_t (same layout) and dq.  Both contractions run on the 16-bit matrix pipe as
//   three-term bf16 products (fp32-grade, 48 MFMAs of 32 cycles per 32-key tile instead of 64 of 64): K / V tiles are
//   split when they are staged (three bf16 planes each, double-buffered, the next tile's K, V and logits loads in flight
//   during the MFMAs; one barrier per tile), dO once per wave, dS^T per tile.
//   V planes [key][72]: the dP^T A operand (lane = key, 8 consecutive d) is one ds_read_b128.
//   K planes [key][96]: dQ^T needs K^T (lane = d, 8 keys): two ds_read_b64_tr_b16 per fragment; 192-byte rows keep the
//   four key rows of a transposed read on disjoint banks.
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256, 2) void deform_attn_bwd_dq_kernel(
    const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ O,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ LT,
    float* __restrict__ dLT, float* __restrict__ dQ, float* __restrict__ RHO, unsigned* __restrict__ AMAX, int N, int J, int H, int NST,
    float scale, DropCfg dc_in) {
  const DropCfg dc = drop_resolve(dc_in);
  __shared__ __attribute__((aligned(16))) __bf16 Vp[2][3][KT * VBLD];
  __shared__ __attribute__((aligned(16))) __bf16 Kp[2][3][KT * KBLD];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (QT * WAVES) + wave * QT;
  const int HD = H * DH;
  const bool qvalid = (q0 + c) < N;
  const int qi = qvalid ? (q0 + c) : (N - 1);

  // dO of this lane's query as the B operand of dP^T = V . dO^T: K-block kb holds d = 16 kb + 8 hf + j
  bf16x8 doh[4], dom[4], dol[4];
  float delta = 0.f;
  {
    const size_t off = ((size_t)b * N + qi) * HD + h * DH + hf * 8;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float4 t0 = *reinterpret_cast<const float4*>(dO + off + 16 * kb), t1 = *reinterpret_cast<const float4*>(dO + off + 16 * kb + 4);
      const float4 u0 = *reinterpret_cast<const float4*>(O + off + 16 * kb), u1 = *reinterpret_cast<const float4*>(O + off + 16 * kb + 4);
      delta += t0.x * u0.x + t0.y * u0.y + t0.z * u0.z + t0.w * u0.w + t1.x * u1.x + t1.y * u1.y + t1.z * u1.z + t1.w * u1.w;
      const float x8[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
      split8_bf3(x8, doh[kb], dom[kb], dol[kb]);
    }
  }
  delta = xhalf_sum(delta);
  const float nl = prob_bias(LSE[(size_t)(b * H + h) * N + qi]);

  floatx16 dq0 = {0}, dq1 = {0};
  float amax = 0.f;      // max |d scores| of this lane (region backward: scale of its fixed-point moment sums); AMAX may be null
  float rho = 0.f;       // sum over keys of this query's d scores: 0 in exact arithmetic, ~1e-7 |dO||O| with delta = rowsum(dO . O)
  const float* Kb = K + (size_t)b * J * HD + h * DH;
  const float* Vb = V + (size_t)b * J * HD + h * DH;
  const float* LTb = LT + ((size_t)(b * H + h) * NST + q0) * J;          // this wave's [J][32] blocks (layout: forward kernel)
  float* dLTb = dLT + ((size_t)(b * H + h) * NST + q0) * J;

  // staging map: thread -> keys (tid >> 4) and (tid >> 4) + 16, 4 consecutive d
  const int skey = tid >> 4, sd4 = (tid & 15) * 4;
  // transposed-read lane map (lane 4 q + p of a 16-lane group: row q, columns 4 p .. 4 p + 3)
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  float4 kreg[2], vreg[2];
  float lt[16];
  auto fetch = [&](int j0, float4 (&kr)[2], float4 (&vr)[2], float (&l)[16]) {
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
#if SMML_DELTA_EXACT
  // Measurement variant: delta = sum_k P_k dP_k from the SAME dP products the d scores are made of, in a first sweep
  // over the keys, instead of rowsum(dO . O).  A systematic relative error of the dP products (the matrix pipe truncates the aligned
  // products of a block toward zero, tests/microbench/mfma_round_probe.hip) then scales dS as a whole instead of surviving the
  // cancellation dP - delta, where it is amplified by |dP| / |dP - delta|.
  {
    float dsum = 0.f;
    for (int kt = 0; kt < ntiles; ++kt) {
      const int j0 = kt * KT, buf = kt & 1;
      const int nk = min(KT, J - j0);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int key = skey + 16 * i;
        uint2v hh, mm, ll;
        split4_bf3(vreg[i], hh, mm, ll);
        *reinterpret_cast<uint2v*>(&Vp[buf][0][key * VBLD + sd4]) = hh;
        *reinterpret_cast<uint2v*>(&Vp[buf][1][key * VBLD + sd4]) = mm;
        *reinterpret_cast<uint2v*>(&Vp[buf][2][key * VBLD + sd4]) = ll;
      }
      __syncthreads();
      float ltc[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) ltc[r] = lt[r];
      if (kt + 1 < ntiles) fetch(j0 + KT, kreg, vreg, lt);
      floatx16 dp = {0};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int o = c * VBLD + 16 * kb + 8 * hf;
        dp = bwd_prod<3>(*reinterpret_cast<const bf16x8*>(&Vp[buf][0][o]), *reinterpret_cast<const bf16x8*>(&Vp[buf][1][o]),
                         *reinterpret_cast<const bf16x8*>(&Vp[buf][2][o]), doh[kb], dom[kb], dol[kb], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (acc_row(r, hf) < nk) {
          float dpr = dp[r];
          if (dc.thresh) dpr *= stashed_factor(ltc[r], dc.keep_scale);
          dsum = fmaf(prob_of(ltc[r], nl), dpr, dsum);
        }
      }
    }
    delta = xhalf_sum(dsum);
    __syncthreads();
    fetch(0, kreg, vreg, lt);
  }
#endif
  for (int kt = 0; kt < ntiles; ++kt) {
    const int j0 = kt * KT, buf = kt & 1;
    const int nk = min(KT, J - j0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = skey + 16 * i;
      uint2v hh, mm, ll;
      split4_bf3(vreg[i], hh, mm, ll);
      *reinterpret_cast<uint2v*>(&Vp[buf][0][key * VBLD + sd4]) = hh;
      *reinterpret_cast<uint2v*>(&Vp[buf][1][key * VBLD + sd4]) = mm;
      if (SMML_BWD_TERMS == 3) *reinterpret_cast<uint2v*>(&Vp[buf][2][key * VBLD + sd4]) = ll;
      split4_bf3(kreg[i], hh, mm, ll);
      *reinterpret_cast<uint2v*>(&Kp[buf][0][key * KBLD + sd4]) = hh;
      *reinterpret_cast<uint2v*>(&Kp[buf][1][key * KBLD + sd4]) = mm;
      if (SMML_DQ_OUT_TERMS == 3) *reinterpret_cast<uint2v*>(&Kp[buf][2][key * KBLD + sd4]) = ll;
    }
    __syncthreads();        // buffer (kt & 1) was last read in iteration kt - 2, which every wave left before this barrier's
                            // predecessor: one barrier per tile is enough
    float ltc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ltc[r] = lt[r];
    if (kt + 1 < ntiles) fetch(j0 + KT, kreg, vreg, lt);

    // dP^T[key, query] = V . dO^T
    floatx16 dp = {0};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int o = c * VBLD + 16 * kb + 8 * hf;
      const bf16x8 vh = *reinterpret_cast<const bf16x8*>(&Vp[buf][0][o]);
      const bf16x8 vm = *reinterpret_cast<const bf16x8*>(&Vp[buf][1][o]);
      const bf16x8 vl = *reinterpret_cast<const bf16x8*>(&Vp[buf][2][o]);
#if SMML_BWD_EXP != 1
      dp = bwd_prod<SMML_BWD_TERMS>(vh, vm, vl, doh[kb], dom[kb], dol[kb], dp);
#else
      dp[kb] += __builtin_bit_cast(float, (unsigned)vh[0] << 16) + __builtin_bit_cast(float, (unsigned)vm[1] << 16) + __builtin_bit_cast(float, (unsigned)vl[2] << 16);
#endif
    }

    float ds[16];
    if (nk == KT && q0 + QT <= N) {                        // interior tile (uniform): no bounds branches around the stores
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = prob_of(ltc[r], nl);
        float dpr = dp[r];
        if (dc.thresh) dpr *= stashed_factor(ltc[r], dc.keep_scale);          // the forward's decision rides in the score's lowest bit
        const float v = p * (dpr - delta);
#if !SMML_EXP_NODLT
        dLTb[(size_t)(j0 + acc_row(r, hf)) * 32 + c] = v;
#endif
        ds[r] = v;
        rho += v;
        amax = fmaxf(amax, fabsf(v));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = acc_row(r, hf);
        float v = 0.f;
        if (key < nk && qvalid) {
          const float p = prob_of(ltc[r], nl);
          float dpr = dp[r];
          if (dc.thresh) dpr *= stashed_factor(ltc[r], dc.keep_scale);
          v = p * (dpr - delta);
          dLTb[(size_t)(j0 + key) * 32 + c] = v;
        }
        ds[r] = v;
        rho += v;
        amax = fmaxf(amax, fabsf(v));
      }
    }
    // dQ^T[d, query] += K^T . dS^T: accumulator register 8 kb + j of dS^T is element j of K-block kb (keys 16 kb + 4 hf + 0..3
    // and 16 kb + 8 + 4 hf + 0..3), the K^T fragment is gathered for the same keys by two transposed reads
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float x8[8];
#pragma unroll
      for (int jx = 0; jx < 8; ++jx) x8[jx] = ds[8 * kb + jx];
      bf16x8 sh, sm, sl;
      split8_bf3(x8, sh, sm, sl);
      const int ro = (16 * kb + 4 * hf + trq) * KBLD + trc;
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const bf16x8 kh = lds_frag_tr(&Kp[buf][0][ro + 32 * db], &Kp[buf][0][ro + 32 * db + 8 * KBLD]);
        const bf16x8 km = lds_frag_tr(&Kp[buf][1][ro + 32 * db], &Kp[buf][1][ro + 32 * db + 8 * KBLD]);
        const bf16x8 kl = (SMML_DQ_OUT_TERMS == 3) ? lds_frag_tr(&Kp[buf][2][ro + 32 * db], &Kp[buf][2][ro + 32 * db + 8 * KBLD]) : km;
#if SMML_BWD_EXP != 2
        if (db == 0) dq0 = bwd_prod<SMML_DQ_OUT_TERMS>(kh, km, kl, sh, sm, sl, dq0);
        else dq1 = bwd_prod<SMML_DQ_OUT_TERMS>(kh, km, kl, sh, sm, sl, dq1);
#else
        if (db == 0) dq0[kb] += __builtin_bit_cast(float, (unsigned)kh[0] << 16) * __builtin_bit_cast(float, (unsigned)sh[0] << 16) + __builtin_bit_cast(float, (unsigned)km[1] << 16) * __builtin_bit_cast(float, (unsigned)sm[1] << 16);
        else dq1[kb] += __builtin_bit_cast(float, (unsigned)kh[2] << 16) * __builtin_bit_cast(float, (unsigned)sh[2] << 16) + __builtin_bit_cast(float, (unsigned)km[3] << 16) * __builtin_bit_cast(float, (unsigned)sm[3] << 16);
#endif
      }
    }
  }
  rho = xhalf_sum(rho);
  if (qvalid && hf == 0) RHO[(size_t)(b * H + h) * N + qi] = rho;
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
// backward pass 2 (key owners): dV = P_dropped^T dO, dK = scale * dS^T Q.  A workgroup owns 128 keys (one 32-key
// tile per wave) and one slice of the query tiles; its four waves consume the same Q / dO tile from LDS (staged
// once per workgroup as three bf16 planes each, double-buffered, the next tile's global loads in flight during the
// MFMAs).  Both contractions are three-term bf16 products on the 16-bit matrix pipe (48 MFMAs per tile): the A operands
// Q^T / dO^T (lane = d, 8 queries) are gathered from the row-major planes by ds_read_b64_tr_b16, the B operands P and
// dS are split from registers.  The partial sums of the query slices go to slabs [nparts][B, J, H*64] that
// dkv_reduce_kernel adds up in a fixed order (no atomics: run-to-run identical results).
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256, 2) void deform_attn_bwd_dkv_kernel(
    const float* __restrict__ Q, const float* __restrict__ dO, const float* __restrict__ LSE,
    const float* __restrict__ LT, const float* __restrict__ dLT, float* __restrict__ dKp,
    float* __restrict__ dVp, int N, int J, int H, int NST, int nkg, int tiles_per_part, int nparts, int Bn, DropCfg dc_in) {
  const DropCfg dc = drop_resolve(dc_in);
  __shared__ __attribute__((aligned(16))) __bf16 Qp[2][3][QT * QBLD];
  __shared__ __attribute__((aligned(16))) __bf16 dOp[2][3][QT * QBLD];
  __shared__ __attribute__((aligned(16))) float nls[2][QT];   // -lse (times log2 e on the fast path) of the tile's queries
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  // XCD-aware block order: workgroups go to the 8 XCDs (each with its own L2) round-robin by linear id.  The nkg key groups
  // of one (query slice, head, bag) read the same Q / dO tiles, so they are given ids that differ by multiples of 8:
  // id = 8 nkg * chunk + 8 kg + x  <->  slice = 8 chunk + x.
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
  const float* LTk = LT + (size_t)(b * H + h) * NST * J + (size_t)key * 32;      // + query tile * J * 32 + query within the tile
  const float* dLTk = dLT + (size_t)(b * H + h) * NST * J + (size_t)key * 32;
  const float* LSEb = LSE + (size_t)(b * H + h) * N;

  const int nqt = (N + QT - 1) / QT;
  const int qt_begin = part * tiles_per_part, qt_end = min(qt_begin + tiles_per_part, nqt);

  // staging map: thread -> rows (tid >> 4) and (tid >> 4) + 16 of the 32-query tile, 4 consecutive d
  const int srow = tid >> 4, sd4 = (tid & 15) * 4;
  // transposed-read lane map (lane 4 q + p of a 16-lane group: row q, columns 4 p .. 4 p + 3)
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  float4 qreg[2], doreg[2], ltr[4], dlr[4];
  float lsereg = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { ltr[i] = make_float4(0.f, 0.f, 0.f, 0.f); dlr[i] = ltr[i]; qreg[i >> 1] = ltr[i]; doreg[i >> 1] = ltr[i]; }
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
      const size_t qq = (size_t)q0 * J + 8 * rg + 4 * hf;   // tile q0 / 32 -> (q0 / 32) J 32 floats; 4 consecutive queries of the tile
      ltr[rg] = *reinterpret_cast<const float4*>(LTk + qq);
#if SMML_EXP_NODLT
      dlr[rg] = ltr[rg];
#else
      dlr[rg] = *reinterpret_cast<const float4*>(dLTk + qq);
#endif
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
      uint2v hh, mm, ll;
      split4_bf3(qreg[i], hh, mm, ll);
      *reinterpret_cast<uint2v*>(&Qp[buf][0][o]) = hh;
      *reinterpret_cast<uint2v*>(&Qp[buf][1][o]) = mm;
      if (SMML_DKV_TERMS == 3) *reinterpret_cast<uint2v*>(&Qp[buf][2][o]) = ll;
      split4_bf3(doreg[i], hh, mm, ll);
      *reinterpret_cast<uint2v*>(&dOp[buf][0][o]) = hh;
      *reinterpret_cast<uint2v*>(&dOp[buf][1][o]) = mm;
      if (SMML_DKV_TERMS == 3) *reinterpret_cast<uint2v*>(&dOp[buf][2][o]) = ll;
    }
    if (tid < QT) nls[buf][tid] = prob_bias(lsereg);
    __syncthreads();        // one barrier per tile (double buffer, see pass 1)
    float lv[16], dsv[16], ls[16];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      lv[4 * rg + 0] = ltr[rg].x; lv[4 * rg + 1] = ltr[rg].y; lv[4 * rg + 2] = ltr[rg].z; lv[4 * rg + 3] = ltr[rg].w;
      dsv[4 * rg + 0] = dlr[rg].x; dsv[4 * rg + 1] = dlr[rg].y; dsv[4 * rg + 2] = dlr[rg].z; dsv[4 * rg + 3] = dlr[rg].w;
      const float4 t = *reinterpret_cast<const float4*>(&nls[buf][8 * rg + 4 * hf]);     // broadcast read
      ls[4 * rg + 0] = t.x; ls[4 * rg + 1] = t.y; ls[4 * rg + 2] = t.z; ls[4 * rg + 3] = t.w;
    }
    if (qt + 1 < qt_end) fetch(q0 + QT);
    if (nk > 0) {                                          // wave-uniform
      // P[query, key] and dS[query, key] with the query on the accumulator-row axis
      float p[16], ds[16];
      if (nk == KT && q0 + QT <= N) {                      // interior tile (uniform): no bounds selects
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float pv = prob_of(lv[r], ls[r]);
          if (dc.thresh) pv *= stashed_factor(lv[r], dc.keep_scale);
          p[r] = pv; ds[r] = dsv[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = q0 + acc_row(r, hf);
          const bool ok = kvalid && qq < N;
          float pv = ok ? prob_of(lv[r], ls[r]) : 0.f;
          if (dc.thresh && ok) pv *= stashed_factor(lv[r], dc.keep_scale);
          p[r] = pv;                                       // dV takes the dropped probabilities, dK the dS written by pass 1
          ds[r] = ok ? dsv[r] : 0.f;
        }
      }
      // dV^T[d, key] += dO^T . P ;  dK^T[d, key] += Q^T . dS.  Register 8 kb + j of p / ds is element j of K-block kb (queries
      // 16 kb + 4 hf + 0..3 and 16 kb + 8 + 4 hf + 0..3); the A fragments are gathered for the same queries
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float x8[8], y8[8];
#pragma unroll
        for (int jx = 0; jx < 8; ++jx) { x8[jx] = p[8 * kb + jx]; y8[jx] = ds[8 * kb + jx]; }
        bf16x8 ph, pm, pl, sh, sm, sl;
        split8_bf3(x8, ph, pm, pl);
        split8_bf3(y8, sh, sm, sl);
        const int ro = (16 * kb + 4 * hf + trq) * QBLD + trc;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int o = ro + 32 * db;
          const bf16x8 ah = lds_frag_tr(&dOp[buf][0][o], &dOp[buf][0][o + 8 * QBLD]);
          const bf16x8 am = lds_frag_tr(&dOp[buf][1][o], &dOp[buf][1][o + 8 * QBLD]);
          const bf16x8 al = (SMML_DKV_TERMS == 3) ? lds_frag_tr(&dOp[buf][2][o], &dOp[buf][2][o + 8 * QBLD]) : am;
          const bf16x8 qh = lds_frag_tr(&Qp[buf][0][o], &Qp[buf][0][o + 8 * QBLD]);
          const bf16x8 qm = lds_frag_tr(&Qp[buf][1][o], &Qp[buf][1][o + 8 * QBLD]);
          const bf16x8 ql = (SMML_DKV_TERMS == 3) ? lds_frag_tr(&Qp[buf][2][o], &Qp[buf][2][o + 8 * QBLD]) : qm;
#if SMML_BWD_EXP != 3
          if (db == 0) { dv0 = bwd_prod<SMML_DKV_TERMS>(ah, am, al, ph, pm, pl, dv0); dk0 = bwd_prod<SMML_DKV_TERMS>(qh, qm, ql, sh, sm, sl, dk0); }
          else { dv1 = bwd_prod<SMML_DKV_TERMS>(ah, am, al, ph, pm, pl, dv1); dk1 = bwd_prod<SMML_DKV_TERMS>(qh, qm, ql, sh, sm, sl, dk1); }
#else
          if (db == 0) { dv0[kb] += __builtin_bit_cast(float, (unsigned)ah[0] << 16) * __builtin_bit_cast(float, (unsigned)ph[0] << 16) + __builtin_bit_cast(float, (unsigned)am[1] << 16); dk0[kb] += __builtin_bit_cast(float, (unsigned)qh[0] << 16) * __builtin_bit_cast(float, (unsigned)sh[0] << 16) + __builtin_bit_cast(float, (unsigned)qm[1] << 16); }
          else { dv1[kb] += __builtin_bit_cast(float, (unsigned)ah[2] << 16) * __builtin_bit_cast(float, (unsigned)ph[2] << 16) + __builtin_bit_cast(float, (unsigned)am[3] << 16); dk1[kb] += __builtin_bit_cast(float, (unsigned)qh[2] << 16) * __builtin_bit_cast(float, (unsigned)sh[2] << 16) + __builtin_bit_cast(float, (unsigned)qm[3] << 16); }
#endif
        }
      }
    }
  }
  // accumulators hold [d = acc_row(r, hf) (+32)][key = c]: each lane writes its key's 4-float runs
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
// backward of the continuous position bias: given dS^T (= d bias) and the layer-2 ReLU decisions the forward saved,
// accumulate dW1, db1, dW2, db2, dW3, db3 and d vs per (key, 32 queries).  Every contraction runs on the 16-bit matrix
// pipe; the two register layouts an MFMA can deliver are both used so that no operand is transposed through LDS:
//
//   "query-major"   lane = query, registers = channels        "channel-major"  lane = channel, registers = queries
//   x1 = W1 p + b1 (one bf16 MFMA; layer-1 masks)              x1^T: the same product with A and B exchanged -> h1^T
//   mask2: the forward's bits, rotated into an fp16 operand     mask2^T = mask2 . (scaled I): exact 0.0 / 1.0
//   dh1 = (W2 w3)^T mask2  (chain 2, exact mask operand)        db2: one scalar pair per lane (and with e, dW3)
//   layer-1 backward, d vs                                     dW2 = w3 . mask2^T g,  g = h1 . d bias (bf16 x 2)
//
// Layer 2 itself is NOT recomputed (no W2 h1 product, no fp16 split of h1): its only use in the backward is the ReLU
// mask, which costs the forward 16 bits per lane and key (1.6 GB per 8 bags of 10 000 x 625 x 8 heads).
//
// What measurement says about gfx950 (tests/microbench/{valu,overlap,mfma,valu_mix}_probe.hip): two waves per SIMD are
// needed to keep the vector unit issuing; an MFMA - the 16-bit forms included - occupies the SIMD for its duration, it
// does not run beside other waves' vector work; vector instructions come in a fast and a slow issue class.  So the
// kernel is written for two resident waves (256 registers each) and for the fewest MFMAs + slow-class instructions per
// key: 12 MFMAs (1 + 1 + 2 + 4 + 4) + about 230 vector instructions.
// Per-lane partial sums are reduced per workgroup into a slab [numWG][CPB_SLAB] that two small kernels add up in a
// fixed order (deterministic).   slab layout: dW2[1024] | dW1[32*2] | db1[32] | db2[32] | dW3[32] | db3[1]  (+pad)
// ------------------------------------------------------------------------------------------------

template <int PDX>
__global__ __launch_bounds__(256, 2) void cpb_bwd_kernel(
    const float* __restrict__ dLT, const unsigned short* __restrict__ MK, const float* __restrict__ LT,
    const float* __restrict__ LSE, const float* __restrict__ RHO, const float* __restrict__ VS,
    const float* __restrict__ GQ, CpbParams cp, float* __restrict__ slab, float* __restrict__ dvs_slab, int N, int J, int H,
    int G, int NST) {
  constexpr int PD = PosCfg<PDX>::PD;
  constexpr bool RAW = PosCfg<PDX>::RAW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // layout: tab[2][2][16] | per wave: xq[2][32], stg[16][65] float2 | red[CPB_SLAB]
  // d vs: every wave writes the sums over its 32 queries, 16 keys at a time, to its own slab row [wg * WAVES + wave][J][2];
  // dvs_reduce_kernel adds the rows of a (bag, group) in a fixed order (no atomics: run-to-run identical, no limit on J)
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
    tab[(tid >> 4) * 32 + (tid & 15)] = cp.w1[ch * PD];
    tab[(tid >> 4) * 32 + 16 + (tid & 15)] = (PD == 2) ? cp.w1[ch * PD + 1] : 0.f;
  }
  const float* tabh = tab + hf * 32;

  const float gq0 = GQ[(size_t)qi * PD];
  const float gq1 = (PD == 2) ? GQ[(size_t)qi * PD + 1] : 0.f;

  // Layer 1, x[ch][q] = w1x[ch] p0[q] + w1y[ch] p1[q] + b1[ch], runs on the matrix pipe in both layouts, one bf16 MFMA
  // each, with weights and positions split into three bf16 terms (h + m + l = the fp32 value to 2^-24):
  //  * query-major (lane = query), exactly the forward's product - eight of the nine cross terms in the 16 K slots
  //    (slots 6, 7 of the position operand: p_h in lane half 0, p_m in half 1), b1 exact in the C operand - so the
  //    layer-1 ReLU masks of the backward are the forward's, bit for bit;
  //  * channel-major (x^T: queries in the registers, channel c in the lane; feeds only g = h1 . d bias, which is kept
  //    to 16 bits) with the operands exchanged: six cross terms + b1 in three terms in the 16 K slots (what is left
  //    out, m l + l m + l l, is <= 2^-23 of a term):
  //      positions, half 0:  p0_h p1_h p0_m p1_m p0_l p1_l 1 1     half 1:  p0_h p1_h p0_m p1_m p0_h p1_h 1 0
  //      constants, half 0:  x_h  y_h  x_h  y_h  x_h  y_h  b_h b_m  half 1:  x_m  y_m  x_m  y_m  x_l  y_l  b_l 0
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

  // The layer-2 ReLU mask comes from the forward (16 bits per lane and key, element r at bit (13 + r) % 16).  Rotated by
  // 2 p and masked with 0x40002000 the word holds elements 2 p and 2 p + 1 as the fp16 pair {2^-7 or 0, 2.0 or 0} (single
  // bits 13 and 30): an MFMA operand without any conversion.  The constant operands carry the inverse scales per K slot
  // (128 for even slots, 0.5 for odd ones - powers of two, exact):
  half8 w2th[2], w2tm[2], w2tl[2];     // W2[out = ch(8 kb + j)][in = c] * w3[out] * slot scale * lift: A operand of chain 2
  half8 idb[2];                        // scaled identity: mask (operand layout, lane = query) . I = mask^T as exact 0.0 / 1.0
  float unlift2;                       // 2^-k of the chain-2 lift (pow2_lift): multiplies d bias where d h1 is consumed
  {
    float amax = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) { const int ch = acc_row(s, hf); amax = fmaxf(amax, fabsf(cp.w2[ch * CH + c] * cp.w3[oi * CH + ch])); }
    // largest |W2 w3| 2^k in (128, 256]: the even slots (x 128) top out at 2^15 < 65504, the odd ones (x 0.5) at 128 - both terms
    // of the two-term split stay normal fp16 numbers (>= 2^-14) for elements down to 2^-10 of the largest
    const float lift2 = SMML_LIFT_BWD ? pow2_lift(wave_max_all(amax), 256.f, -8.f, 24.f) : 1.f;
    unlift2 = 1.f / lift2;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = acc_row(8 * kb + j, hf);
        const float sc = (j & 1) ? 0.5f : 128.f;
        t[j] = cp.w2[ch * CH + c] * cp.w3[oi * CH + ch] * sc * lift2;
        idb[kb][j] = (ch == c) ? (_Float16)sc : (_Float16)0.0f;
      }
      split8_3(t, w2th[kb], w2tm[kb], w2tl[kb]);
    }
  }

  float big;                           // 2^100 in an SGPR (the clamp-multiply takes no literal)
  asm("s_mov_b32 %0, 0x71800000" : "=s"(big));
  floatx16 e = {0};                    // sum_q mask[out, q] g[in, q]: rows = out, lane = in (times w3[out] at the end)
  float2v aw1x[8], aw1y[8], ab1[8];    // channel pairs (registers 2 p, 2 p + 1)
#pragma unroll
  for (int p = 0; p < 8; ++p) { aw1x[p] = (float2v){0.f, 0.f}; aw1y[p] = aw1x[p]; ab1[p] = aw1x[p]; }
  float ab3 = 0.f;
  float2v s2 = {0.f, 0.f};             // sum mask . d bias of out = c (db2; dW3 follows from it and e at the end)

  const float* VSb = VS + (size_t)(b * G + g) * J * PD;
  const float* dLTb = dLT + ((size_t)(b * H + h) * NST + q0) * J;        // this wave's [J][32] block (layout: forward kernel)
  __syncthreads();
  float vx_n = VSb[0];
  float vy_n = (PD == 2) ? VSb[1] : 0.f;
  float db_n = dLTb[c];                                 // lanes past the bag end read padding of their own tile and are zeroed
#if SMML_DELTA_FIX
  // The d scores of a fused softmax backward use delta = rowsum(dO . O), which leaves sum_k dS_k = rho != 0 at the 1e-7
  // level per query (in exact arithmetic delta = sum_k P_k dP_k and the row sums vanish).  The sums below multiply d bias
  // by near-constant factors, which amplifies exactly that component, so the row is re-centred here: d bias_k - P_k rho.
  const float* LTb = LT + ((size_t)(b * H + h) * NST + q0) * J;
  const float nl = prob_bias(LSE[(size_t)(b * H + h) * N + qi]);
  const float nrho = -RHO[(size_t)(b * H + h) * N + qi];
  float lt_n = LTb[c];
#endif
  const unsigned short* MKb = MK + (((size_t)(b * H + h) * NST + q0) * J) * 2 + hf * 32;
  unsigned m16_n = MKb[c];

  for (int j = 0; j < J; ++j) {
#if SMML_DELTA_FIX
    const float vx = vx_n, vy = vy_n, dbias = qvalid ? fmaf(prob_of(lt_n, nl), nrho, db_n) : 0.f;
#else
    const float vx = vx_n, vy = vy_n, dbias = qvalid ? db_n : 0.f;
#endif
    const unsigned m16 = m16_n;
    {
      const int jn = min(j + 1, J - 1);                     // branch-free prefetch of the next key's operands
      vx_n = VSb[(size_t)jn * PD];
      if (PD == 2) vy_n = VSb[(size_t)jn * PD + 1];
      db_n = dLTb[(size_t)jn * 32 + c];
      m16_n = MKb[(size_t)jn * 64 + c];
#if SMML_DELTA_FIX
      lt_n = LTb[(size_t)jn * 32 + c];
#endif
    }
    float* xb = xq + (j & 1) * 32;
    xb[c] = dbias;                                          // for the channel-major stage (both halves store the same value)
    const float d0 = gq0 - vx, d1 = gq1 - vy;
    const float p0 = pos_of<RAW>(d0);
    const float p1 = (PD == 2) ? slog1p(d1) : 0.f;

    // ---- layer 1 on the matrix pipe, in both layouts ----
    floatx16 xacc, ht;
    {
      const PosTerms pt = cpb_split_pos(p0, p1);
      const unsigned hw = pt.hw, mw = pt.mw, lw = pt.lw;
      xacc = cpb_layer1_q(a1q, pt, hf, b1acc);
      const uint4v tw = {hw, mw, hf ? hw : lw, hf ? 0x00003F80u : 0x3F803F80u};
      ht = mfma16b(__builtin_bit_cast(bf16x8, tw), a1t, (floatx16){0});
    }
    bool on1[16];                       // layer-1 ReLU masks: live to the end of the trip as lane masks in SGPRs
#pragma unroll
    for (int r = 0; r < 16; ++r) on1[r] = xacc[r] > 0.f;

    // ---- layer-2 mask: the forward's bits -> query-major fp16 operand (lane = query, K slot = out channel) ----
    half8 mk[2];
    {
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
    // ---- the matrix pipe transposes it: mask . (scaled I) = mask^T[query = acc_row(r, hf)][out = c] as exact 0.0 / 1.0 ----
    floatx16 mtt = mfma16(mk[0], idb[0], (floatx16){0});
    mtt = mfma16(mk[1], idb[1], mtt);

    // ---- channel-major stage, part 1: db2 partial sums and the exact 0 / 1 bf16 mask operand of the dW2 product ----
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
        const float2v mf = {mtt[r], mtt[r + 1]};
        s2[0] = fmaf(mf[0], dbq[r], s2[0]);
        s2[1] = fmaf(mf[1], dbq[r + 1], s2[1]);
        amw[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(mf, bf16x2));
      }
      am[t] = __builtin_bit_cast(bf16x8, amw);
    }
    // ---- chain 2: dh1[in = ch(r)][query = c] = (W2 w3)^T mask.  With an exact mask operand and the constant split into
    //      fp16 terms (SMML_CHAIN2_TERMS), two MFMAs per K-block give the column sums to the constant's 22 bits (three:
    //      exactly); the lane's d bias multiplies its column afterwards (no gradient scaling needed) ----
    floatx16 dh = {0};
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#if SMML_CHAIN2_TERMS == 3
      dh = mfma16(w2tl[kb], mk[kb], dh);
#endif
      dh = mfma16(w2tm[kb], mk[kb], dh);
      dh = mfma16(w2th[kb], mk[kb], dh);
    }
    ab3 += (hf == 0) ? dbias : 0.f;

    // ---- channel-major stage, part 2: dW2 += mask^T g with g = h1 . d bias in three bf16 terms ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float g8[8];
#pragma unroll
      for (int jx = 0; jx < 8; ++jx) g8[jx] = relu2(ht[8 * t + jx]) * dbq[8 * t + jx];   // 2 h1^T . d bias
#if SMML_G_TERMS == 3
      bf16x8 g1, g2, g3;
      split8_bf3(g8, g1, g2, g3);
      e = mfma16b(am[t], g3, e);
      e = mfma16b(am[t], g2, e);
      e = mfma16b(am[t], g1, e);
#elif SMML_G_TERMS == 1
      // one bf16 term (measurement switch, NOT accurate enough): 8 mantissa bits per summand.  The gradient is a sum of
      // random-sign terms (|sum| ~ sqrt(pairs) rms), so the relative error of the sum is the per-term rounding, 2^-9 / sqrt(3):
      // measured dW2 5.4e-3 vs 2.7e-4 with two terms (profiles/r02_split_terms.txt) - though 0.75 ms per step faster
      uint4v g1w;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float2v v = {g8[2 * i], g8[2 * i + 1]};
        g1w[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
      }
      e = mfma16b(am[t], __builtin_bit_cast(bf16x8, g1w), e);
#else
      // two bf16 terms: 16 mantissa bits per summand (<= 2^-17 relative, unbiased round-to-nearest) against the exact mask
      uint4v g1w, g2w;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float2v v = {g8[2 * i], g8[2 * i + 1]};
        const bf16x2 hh = __builtin_convertvector(v, bf16x2);
        const float2v r1 = {v[0] - (float)hh[0], v[1] - (float)hh[1]};
        const bf16x2 mm = __builtin_convertvector(r1, bf16x2);
        g1w[i] = __builtin_bit_cast(unsigned, hh); g2w[i] = __builtin_bit_cast(unsigned, mm);
      }
      e = mfma16b(am[t], __builtin_bit_cast(bf16x8, g2w), e);
      e = mfma16b(am[t], __builtin_bit_cast(bf16x8, g1w), e);
#endif
    }

    // ---- layer-1 backward, d vs ----
    {
      float2v dp0v = {0.f, 0.f}, dp1v = {0.f, 0.f};
      const float dbl = dbias * unlift2;                    // d h1 arrives lifted by 2^k (chain-2 constants): undone here, once per key
      const float p0i = p0 * dbl, p1i = p1 * dbl;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        float2v g1;
        g1[0] = on1[2 * p] ? dh[2 * p] : 0.f;
        g1[1] = on1[2 * p + 1] ? dh[2 * p + 1] : 0.f;
        const float2 wx = *reinterpret_cast<const float2*>(tabh + 2 * p);               // broadcast reads
        ab1[p] = g1 * (float2v){dbl, dbl} + ab1[p];
        aw1x[p] = g1 * (float2v){p0i, p0i} + aw1x[p];
        dp0v = g1 * (float2v){wx.x, wx.y} + dp0v;
        if (PD == 2) {
          const float2 wy = *reinterpret_cast<const float2*>(tabh + 16 + 2 * p);
          aw1y[p] = g1 * (float2v){p1i, p1i} + aw1y[p];
          dp1v = g1 * (float2v){wy.x, wy.y} + dp1v;
        }
      }
      // d/dx sign(x) log(|x| + 1) = 1 / (|x| + 1) - except that autograd's convention |x|' = 0, sign' = 0 at x == 0 makes the
      // reference's gradient vanish where a query sits exactly on a sample position (DeformableAttention2D.py:148 through
      // torch.sign / torch.abs).  Kept: [x != 0] as clamp(|x| 2^100), one fast-class multiply (subnormal distances aside).
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

  // ---- workgroup reduction of the per-lane partials -> slab[wg]: every wave fills its own copy of the slab in LDS with plain
  //      stores (one writer per address; LDS operations of a wave execute in program order), the copies are added in a fixed
  //      order - no float atomics, so the parameter gradients are run-to-run identical ----
  __syncthreads();
  float* red = wbase + WAVES * CPB2_WAVE_LDS + wave * CPB_SLAB;   // this wave's [CPB_SLAB]
  for (int i = lane; i < CPB_SLAB; i += 64) red[i] = 0.f;
  {
    const float s2s = xhalf_sum(s2[0] + s2[1]);                       // both lane halves: sum mask . d bias of out = c
    if (hf == 0) {
      red[1024 + 64 + 32 + c] = w3c * s2s;                            // db2[c] = w3[c] sum mask . d bias
      // dW3[out] = sum relu(D + b2) . d bias = sum_in W2[out][in] e[out][in] + b2[out] sum mask . d bias: nothing of it
      // has to be accumulated per key (the first term is added row by row below)
      red[1024 + 64 + 32 + 32 + c] = b2c * s2s;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, hf);
    red[row * CH + c] = 0.5f * e[r] * cp.w3[oi * CH + row];           // dW2[out = row][in = c]  (e holds 2 x the sum)
    float v;
    v = 0.5f * e[r] * cp.w2[row * CH + c];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (c == 0) red[1024 + 64 + 32 + 32 + row] += v;                   // after the store above (same wave: in order)
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


// Decision export (tests only): the layer-1 ReLU decisions [x1 > 0] of the position-bias MLP exactly as the forward and the
// backward evaluate them, in the bit layout of the saved layer-2 masks (hidden channel acc_row(r, half) at bit (13 + r) % 16):
// masks [B * G, J, 2, NST] uint16.  Parity tests impose these decisions (and the saved layer-2 bits) on the fp64 oracle, so that
// a gradient comparison no longer depends on which way a rounding-level tie of a pre-activation fell.
template <int PDX>
__global__ __launch_bounds__(256) void relu1_masks_kernel(const float* __restrict__ VS, const float* __restrict__ GQ,
                                                          const float* __restrict__ w1, const float* __restrict__ b1,
                                                          unsigned short* __restrict__ MK, int N, int J, int G, int NST) {
  constexpr int PD = PosCfg<PDX>::PD;
  constexpr bool RAW = PosCfg<PDX>::RAW;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int b = blockIdx.z, g = blockIdx.y;
  const int q0 = blockIdx.x * (QT * WAVES) + wave * QT;
  const int qi = min(q0 + c, N - 1);
  const float gq0 = GQ[(size_t)qi * PD];
  const float gq1 = (PD == 2) ? GQ[(size_t)qi * PD + 1] : 0.f;
  const bf16x8 a1 = cpb_l1_weights_q(w1[c * PD], (PD == 2) ? w1[c * PD + 1] : 0.f, hf);
  floatx16 b1acc;
#pragma unroll
  for (int s = 0; s < 16; ++s) b1acc[s] = b1[acc_row(s, hf)];
  const float* VSb = VS + (size_t)(b * G + g) * J * PD;
  unsigned short* MKb = MK + ((size_t)(b * G + g) * J * 2 + hf) * NST;
  for (int j = 0; j < J; ++j) {
    const float p0 = pos_of<RAW>(gq0 - VSb[(size_t)j * PD]);
    const float p1 = (PD == 2) ? slog1p(gq1 - VSb[(size_t)j * PD + 1]) : 0.f;
    const floatx16 xacc = cpb_layer1_q(a1, cpb_split_pos(p0, p1), hf, b1acc);
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m |= (xacc[r] > 0.f) ? (1u << ((13 + r) & 15)) : 0u;
    MKb[(size_t)j * 2 * NST + q0 + c] = (unsigned short)m;      // rows are padded to whole workgroup tiles
  }
}


// keep-mask a launch with (dropout_p, dropout_seed) uses, as 0 / 1 floats [B, H, N, J] (tests only)
__global__ void drop_mask_kernel(float* __restrict__ mask, unsigned long long total, int J, DropCfg dc_in) {
  const DropCfg dc = drop_resolve(dc_in);
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const unsigned long long row = i / (unsigned long long)J;
  const int j = (int)(i - row * (unsigned long long)J);
  mask[i] = (dc.thresh == 0 || ((drop_keep2(dc, row * (unsigned long long)((J + 1) >> 1) + (j >> 1)) >> (j & 1)) & 1u)) ? 1.f : 0.f;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

// row stride of the key-major score / mask tensors: whole 128-query workgroup tiles, so that the forward stores its rows
// without per-lane bounds checks (columns >= N are padding nobody reads)
int smml_deform_attn_nst(int N) { return (N <= 0 || N > SMML_MAX_QUERIES) ? 0 : (N + QT * WAVES - 1) / (QT * WAVES) * (QT * WAVES); }

int smml_deform_attn_dropout_mask_f32(float* mask, int B, int N, int J, int H, float dropout_p, unsigned long long dropout_seed,
                                      void* stream, const SmmlDeformOpts* opts) {
  SMML_REQUIRE(mask && deform_dims_ok(B, N, J, H) && dropout_p >= 0.f && dropout_p < 1.f,
               "smml_deform_attn_dropout_mask_f32: bad argument");
  const unsigned long long total = (unsigned long long)B * H * N * J;
  hipLaunchKernelGGL(drop_mask_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mask, total, J,
                     make_drop(dropout_p, dropout_seed, opts));
  SMML_LAUNCH_CHECK("smml_deform_attn_dropout_mask_f32");
  return SMML_OK;
}

int smml_deform_attn_relu1_masks(const float* vs, const float* gq, const float* w1, const float* b1, unsigned short* masks, int B,
                                 int N, int J, int G, int posdim, void* stream, const SmmlDeformOpts* opts) {
  SMML_REQUIRE(vs && gq && w1 && b1 && masks && deform_dims_ok(B, N, J, G) && (posdim == 1 || posdim == 2),
               "smml_deform_attn_relu1_masks: bad argument");
  dim3 grid((N + QT * WAVES - 1) / (QT * WAVES), G, B), block(256);
  const int nst = smml_deform_attn_nst(N);
  if (posdim == 2)
    hipLaunchKernelGGL(relu1_masks_kernel<2>, grid, block, 0, (hipStream_t)stream, vs, gq, w1, b1, masks, N, J, G, nst);
  else if (pdx_of(posdim, opts) == 3)
    hipLaunchKernelGGL(relu1_masks_kernel<3>, grid, block, 0, (hipStream_t)stream, vs, gq, w1, b1, masks, N, J, G, nst);
  else
    hipLaunchKernelGGL(relu1_masks_kernel<1>, grid, block, 0, (hipStream_t)stream, vs, gq, w1, b1, masks, N, J, G, nst);
  SMML_LAUNCH_CHECK("smml_deform_attn_relu1_masks");
  return SMML_OK;
}

size_t smml_deform_attn_bwd_workspace_bytes(int B, int N, int J, int H) {
  if (!deform_dims_ok(B, N, J, H)) return 0;
  return bwd_workspace(B, N, J, H).total * sizeof(float);
}

static int check_common(const char* fn, int B, int N, int J, int H, int G, int posdim) {
  SMML_REQUIRE(B > 0 && N > 0 && J > 0 && H > 0 && G > 0, "%s: non-positive dimension", fn);
  SMML_REQUIRE(H % G == 0, "%s: heads (%d) must be divisible by offset groups (%d)", fn, H, G);
  SMML_REQUIRE(H / G <= 2, "%s: at most 2 heads per offset group are supported (got %d)", fn, H / G);
  SMML_REQUIRE(posdim == 1 || posdim == 2, "%s: posdim must be 1 or 2 (got %d)", fn, posdim);
  SMML_REQUIRE(deform_dims_ok(B, N, J, H), "%s: B, H <= 65535, N <= 2^26, J <= 2^22 (got B %d N %d J %d H %d)", fn, B, N, J, H);
  return SMML_OK;
}

int smml_deform_attn_fwd_f32(const float* q, const float* k, const float* v, const float* vs, const float* gq,
                             const float* w1, const float* b1, const float* w2, const float* b2,
                             const float* w3, const float* b3, float* out, float* lse, float* logits_t,
                             unsigned short* relu_masks, int B, int N, int J, int H, int G, int posdim, float scale,
                             float dropout_p, unsigned long long dropout_seed, void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts) {
  int rc = check_common("smml_deform_attn_fwd_f32", B, N, J, H, G, posdim);
  SMML_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "smml_deform_attn_fwd_f32: dropout_p must be in [0, 1)");
  const DropCfg dc = make_drop(dropout_p, dropout_seed, opts);
  if (rc) return rc;
  SMML_REQUIRE(q && k && v && vs && gq && w1 && b1 && w2 && b2 && w3 && b3 && out && lse,
               "smml_deform_attn_fwd_f32: null pointer");
  CpbParams cp{w1, b1, w2, b2, w3, b3};
  dim3 grid((N + QT * WAVES - 1) / (QT * WAVES), H, B), block(256);
  const int nst = smml_deform_attn_nst(N);
  hipStream_t st = (hipStream_t)stream;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
  SMML_REQUIRE((logits_t == nullptr) == (relu_masks == nullptr),
               "smml_deform_attn_fwd_f32: logits_t and relu_masks are saved together (training) or not at all");
  if (posdim == 2 && relu_masks)
    hipLaunchKernelGGL((deform_attn_fwd_kernel<2, true>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, logits_t,
                       relu_masks, N, J, H, G, nst, scale, dc);
  else if (posdim == 2)
    hipLaunchKernelGGL((deform_attn_fwd_kernel<2, false>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, logits_t,
                       relu_masks, N, J, H, G, nst, scale, dc);
  else if (pdx_of(posdim, opts) == 3 && relu_masks)         // 1-D, raw offsets (cpb_log_distance = False)
    hipLaunchKernelGGL((deform_attn_fwd_kernel<3, true>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, logits_t,
                       relu_masks, N, J, H, G, nst, scale, dc);
  else if (pdx_of(posdim, opts) == 3)
    hipLaunchKernelGGL((deform_attn_fwd_kernel<3, false>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, logits_t,
                       relu_masks, N, J, H, G, nst, scale, dc);
  else if (relu_masks)
    hipLaunchKernelGGL((deform_attn_fwd_kernel<1, true>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, logits_t,
                       relu_masks, N, J, H, G, nst, scale, dc);
  else
    hipLaunchKernelGGL((deform_attn_fwd_kernel<1, false>), grid, block, 0, st, q, k, v, vs, gq, cp, out, lse, logits_t,
                       relu_masks, N, J, H, G, nst, scale, dc);
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
  SMML_LAUNCH_CHECK("smml_deform_attn_fwd_f32");
  return SMML_OK;
}

int smml_deform_attn_bwd_f32(const float* q, const float* k, const float* v, const float* vs, const float* gq,
                             const float* w1, const float* b1, const float* w2, const float* b2,
                             const float* w3, const float* b3, const float* out, const float* dout,
                             const float* lse, const float* logits_t, const unsigned short* relu_masks,
                             float* dlogits_t, float* dq, float* dk,
                             float* dv, float* dvs, float* dw1, float* db1, float* dw2, float* db2, float* dw3,
                             float* db3, void* workspace, size_t workspace_bytes, int B, int N, int J, int H,
                             int G, int posdim, float scale, float dropout_p, unsigned long long dropout_seed,
                             void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts) {
  int rc = check_common("smml_deform_attn_bwd_f32", B, N, J, H, G, posdim);
  SMML_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "smml_deform_attn_bwd_f32: dropout_p must be in [0, 1)");
  const DropCfg dc = make_drop(dropout_p, dropout_seed, opts);
  if (rc) return rc;
  SMML_REQUIRE(q && k && v && vs && gq && w1 && b1 && w2 && b2 && w3 && b3 && out && dout && lse && logits_t &&
                   relu_masks && dlogits_t && dq && dk && dv && dvs && dw1 && db1 && dw2 && db2 && dw3 && db3 && workspace,
               "smml_deform_attn_bwd_f32: null pointer");
  SMML_REQUIRE(workspace_bytes >= smml_deform_attn_bwd_workspace_bytes(B, N, J, H),
               "smml_deform_attn_bwd_f32: workspace too small (%zu < %zu)", workspace_bytes,
               smml_deform_attn_bwd_workspace_bytes(B, N, J, H));
  SMML_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "smml_deform_attn_bwd_f32: workspace must be 16-byte aligned");
  CpbParams cp{w1, b1, w2, b2, w3, b3};
  hipStream_t st = (hipStream_t)stream;
  const int nst = smml_deform_attn_nst(N);
  const int qtiles = (N + QT * WAVES - 1) / (QT * WAVES);
  dim3 block(256);
  // pass 1: dS^T, dQ
  const BwdWorkspace wsl = bwd_workspace(B, N, J, H);
  float* wsf = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(deform_attn_bwd_dq_kernel, dim3(qtiles, H, B), block, 0, st, k, v, out, dout, lse, logits_t,
                     dlogits_t, dq, wsf + wsl.rho, (unsigned*)nullptr, N, J, H, nst, scale, dc);
  SMML_LAUNCH_CHECK("smml_deform_attn_bwd_f32/dq");
  // pass 2: dK, dV (query-sliced partial sums, then a fixed-order reduction)
  {
    const int nkg = (J + DKV_KEYS - 1) / DKV_KEYS, nqt = (N + QT - 1) / QT;
    const int parts = dkv_parts(B, N, J, H), tpp = (nqt + parts - 1) / parts;
    const int nslices = parts * H * B;
    hipLaunchKernelGGL(deform_attn_bwd_dkv_kernel, dim3(((nslices + 7) / 8) * 8 * nkg), block, 0, st, q, dout, lse,
                       logits_t, dlogits_t, wsf + wsl.dkp, wsf + wsl.dvp, N, J, H, nst, nkg, tpp, parts, B, dc);
    SMML_LAUNCH_CHECK("smml_deform_attn_bwd_f32/dkv");
    const size_t n4 = (size_t)B * J * H * DH / 4;
    hipLaunchKernelGGL(dkv_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), block, 0, st,
                       reinterpret_cast<const float4*>(wsf + wsl.dkp), reinterpret_cast<const float4*>(wsf + wsl.dvp),
                       reinterpret_cast<float4*>(dk), reinterpret_cast<float4*>(dv), n4, parts, scale);
    SMML_LAUNCH_CHECK("smml_deform_attn_bwd_f32/dkv_reduce");
  }
  // pass 3: position-bias MLP backward
  {
    float* slab = (float*)workspace;
    const size_t lds = ((size_t)CPB2_TAB + WAVES * CPB2_WAVE_LDS + WAVES * CPB_SLAB) * sizeof(float);
    if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);   // brackets the position-bias backward kernel only
    if (posdim == 2)
      hipLaunchKernelGGL(cpb_bwd_kernel<2>, dim3(qtiles, H, B), block, lds, st, dlogits_t, relu_masks, logits_t, lse,
                         wsf + wsl.rho, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst);
    else if (pdx_of(posdim, opts) == 3)
      hipLaunchKernelGGL(cpb_bwd_kernel<3>, dim3(qtiles, H, B), block, lds, st, dlogits_t, relu_masks, logits_t, lse,
                         wsf + wsl.rho, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst);
    else
      hipLaunchKernelGGL(cpb_bwd_kernel<1>, dim3(qtiles, H, B), block, lds, st, dlogits_t, relu_masks, logits_t, lse,
                         wsf + wsl.rho, vs, gq, cp, slab, wsf + wsl.dvs, N, J, H, G, nst);
    if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
    SMML_LAUNCH_CHECK("smml_deform_attn_bwd_f32/cpb");
    const int nwg = qtiles * H * B;
    {
      const long long threads = (long long)B * G * J * 4;
      hipLaunchKernelGGL(dvs_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                         reinterpret_cast<const float2*>(wsf + wsl.dvs), dvs, B, G, H, qtiles, J, posdim);
    }
    const int nchunks = min(CPB_RED_CHUNKS, nwg), chunk = (nwg + nchunks - 1) / nchunks;
    hipLaunchKernelGGL(cpb_partial_kernel, dim3((CPB_SLAB + 63) / 64, nchunks), dim3(256), 0, st, slab, nwg, H / G, qtiles,
                       H, chunk, wsf + wsl.partial);
    hipLaunchKernelGGL(cpb_final_kernel, dim3((CPB_SLAB + 255) / 256), dim3(256), 0, st, wsf + wsl.partial, nchunks, H / G,
                       dw1, db1, dw2, db2, dw3, db3, posdim);
    SMML_LAUNCH_CHECK("smml_deform_attn_bwd_f32/reduce");
  }
  return SMML_OK;
}


// ------------------------------------------------------------------------------------------------
// position bias per linear region (cpb_regions.h): exact, 2-D signed-log offsets, one head per offset group
// ------------------------------------------------------------------------------------------------
size_t smml_cpb_regions_bytes(void) { return region_layout().total; }

int smml_cpb_regions_build(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                           float pmax, void* tables, size_t tables_bytes, void* stream) {
  SMML_REQUIRE(w1 && b1 && w2 && b2 && w3 && b3 && tables, "smml_cpb_regions_build: null pointer");
  SMML_REQUIRE(pmax > 0.f && pmax < 16.f, "smml_cpb_regions_build: pmax must be in (0, 16) (got %g)", (double)pmax);
  SMML_REQUIRE(tables_bytes >= region_layout().total, "smml_cpb_regions_build: table buffer too small (%zu < %zu)", tables_bytes,
               region_layout().total);
  SMML_REQUIRE((reinterpret_cast<size_t>(tables) & 255) == 0, "smml_cpb_regions_build: table buffer must be 256-byte aligned");
  region_build_launch(CpbParams{w1, b1, w2, b2, w3, b3}, pmax, tables, (hipStream_t)stream);
  SMML_LAUNCH_CHECK("smml_cpb_regions_build");
  return SMML_OK;
}

int smml_deform_attn_region_fwd_f32(const float* q, const float* k, const float* v, const float* vs, const float* gq,
                                    const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                    const float* b3, const void* tables, float* out, float* lse, float* logits_t,
                                    unsigned short* region_ids, int B, int N, int J, int H, float scale, float dropout_p,
                                    unsigned long long dropout_seed, void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts) {
  int rc = check_region("smml_deform_attn_region_fwd_f32", B, N, J, H);
  if (rc) return rc;
  SMML_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "smml_deform_attn_region_fwd_f32: dropout_p must be in [0, 1)");
  SMML_REQUIRE(q && k && v && vs && gq && w1 && b1 && w2 && b2 && w3 && b3 && tables && out && lse,
               "smml_deform_attn_region_fwd_f32: null pointer");
  SMML_REQUIRE((logits_t == nullptr) == (region_ids == nullptr),
               "smml_deform_attn_region_fwd_f32: logits_t and region_ids are saved together (training) or not at all");
  const DropCfg dc = make_drop(dropout_p, dropout_seed, opts);
  const int lcap = (opts && opts->region_lds_cap > 0) ? (opts->region_lds_cap < RG_LCAP ? opts->region_lds_cap : RG_LCAP) : RG_LCAP;
  CpbParams cp{w1, b1, w2, b2, w3, b3};
  const RegionView rv = region_view(const_cast<void*>(tables));
  dim3 grid((N + QT * WAVES - 1) / (QT * WAVES), H, B), block(256);
  const int nst = smml_deform_attn_nst(N);
  hipStream_t st = (hipStream_t)stream;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
  if (region_ids)
    hipLaunchKernelGGL(deform_region_fwd_kernel<true>, grid, block, 0, st, q, k, v, vs, gq, cp, rv, out, lse, logits_t, region_ids, N, J,
                       H, nst, scale, dc, lcap);
  else
    hipLaunchKernelGGL(deform_region_fwd_kernel<false>, grid, block, 0, st, q, k, v, vs, gq, cp, rv, out, lse, logits_t, region_ids, N, J,
                       H, nst, scale, dc, lcap);
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
  SMML_LAUNCH_CHECK("smml_deform_attn_region_fwd_f32");
  return SMML_OK;
}

size_t smml_deform_attn_region_bwd_workspace_bytes(int B, int N, int J, int H) {
  if (!deform_dims_ok(B, N, J, H) || J > RG_MAX_KEYS) return 0;
  return region_bwd_plan(B, N, J, H).total;
}

int smml_deform_attn_region_bwd_f32(const float* q, const float* k, const float* v, const float* vs, const float* gq,
                                    const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                                    const float* b3, const void* tables, const float* out, const float* dout, const float* lse,
                                    const float* logits_t, const unsigned short* region_ids, float* dlogits_t, float* dq, float* dk,
                                    float* dv, float* dvs, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3,
                                    void* workspace, size_t workspace_bytes, int B, int N, int J, int H, float scale, float dropout_p,
                                    unsigned long long dropout_seed, void* ev_start, void* ev_stop, void* stream, const SmmlDeformOpts* opts) {
  int rc = check_region("smml_deform_attn_region_bwd_f32", B, N, J, H);
  if (rc) return rc;
  SMML_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "smml_deform_attn_region_bwd_f32: dropout_p must be in [0, 1)");
  SMML_REQUIRE(q && k && v && vs && gq && w1 && b1 && w2 && b2 && w3 && b3 && tables && out && dout && lse && logits_t && region_ids &&
                   dlogits_t && dq && dk && dv && dvs && dw1 && db1 && dw2 && db2 && dw3 && db3 && workspace,
               "smml_deform_attn_region_bwd_f32: null pointer");
  const RegionBwdPlan pl = region_bwd_plan(B, N, J, H);
  SMML_REQUIRE(workspace_bytes >= pl.total, "smml_deform_attn_region_bwd_f32: workspace too small (%zu < %zu)", workspace_bytes, pl.total);
  SMML_REQUIRE((reinterpret_cast<size_t>(workspace) & 255) == 0, "smml_deform_attn_region_bwd_f32: workspace must be 256-byte aligned");
  SMML_REQUIRE(pl.wpk >= 1, "smml_deform_attn_region_bwd_f32: too many keys (%d)", J);
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
  // the accumulators of this launch: amax | hist | grad are contiguous
  (void)hipMemsetAsync(wsb + pl.amax, 0, pl.dvs - pl.amax, st);
  // pass 1: dS^T, dQ, max |dS|
  hipLaunchKernelGGL(deform_attn_bwd_dq_kernel, dim3(qtiles, H, B), block, 0, st, k, v, out, dout, lse, logits_t, dlogits_t, dq,
                     wsf + wsl.rho, amax, N, J, H, nst, scale, dc);
  SMML_LAUNCH_CHECK("smml_deform_attn_region_bwd_f32/dq");
  // pass 2: dK, dV
  {
    const int nkg = (J + DKV_KEYS - 1) / DKV_KEYS, nqt = (N + QT - 1) / QT;
    const int parts = dkv_parts(B, N, J, H), tpp = (nqt + parts - 1) / parts;
    const int nslices = parts * H * B;
    hipLaunchKernelGGL(deform_attn_bwd_dkv_kernel, dim3(((nslices + 7) / 8) * 8 * nkg), block, 0, st, q, dout, lse, logits_t, dlogits_t,
                       wsf + wsl.dkp, wsf + wsl.dvp, N, J, H, nst, nkg, tpp, parts, B, dc);
    SMML_LAUNCH_CHECK("smml_deform_attn_region_bwd_f32/dkv");
    const size_t n4 = (size_t)B * J * H * DH / 4;
    hipLaunchKernelGGL(dkv_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), block, 0, st, reinterpret_cast<const float4*>(wsf + wsl.dkp),
                       reinterpret_cast<const float4*>(wsf + wsl.dvp), reinterpret_cast<float4*>(dk), reinterpret_cast<float4*>(dv), n4, parts,
                       scale);
    SMML_LAUNCH_CHECK("smml_deform_attn_region_bwd_f32/dkv_reduce");
  }
  // pass 3: position bias - d vs per pair, region moments, then the dense pass to the six parameter gradients
  rc = region_bias_bwd_launch<float>("smml_deform_attn_region_bwd_f32", dlogits_t, region_ids, vs, gq, cp, tables, wsb, pl, B, N, J, H, nst, lcap, dvs,
                                     dw1, db1, dw2, db2, dw3, db3, ev_start, ev_stop, st);
  if (rc) return rc;
  return SMML_OK;
}

}  // extern "C"
