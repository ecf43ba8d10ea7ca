// Shared pieces of the fused deformable cross-attention kernels (deform_attn.hip: fp32-grade split products; deform_attn16.hip:
// the 16-bit compute mode): tuning knobs, fast-math helpers, the counter-based dropout mask, layer 1 of the position-bias MLP on
// the matrix pipe (one device function for the forward, the backward and the decision export of BOTH modes, so their layer-1 ReLU
// decisions are the same bits), the fixed-order slab reducers and the workspace layout of the backward.
// Reference: models/DeformableAttention2D.py:120-157,284-312; models/DeformableAttention1D.py:60-102,205-232.
#pragma once
#include "smml_common.h"

// tuning knobs (defaults = the configuration measured fastest on MI355X; tests/microbench sweeps them)
#ifndef SMML_FWD_WPS
#define SMML_FWD_WPS 2      // waves per SIMD the forward kernel is register-budgeted for
#endif
#ifndef SMML_SPLIT_TERMS
#define SMML_SPLIT_TERMS 3    // products kept of W h = (wh + wm + wl)(hh + hl) in the 32x32 layer of the forward:
                              //   3 = wh hh + wh hl + wm hh (<= 2^-21 |w||h| dropped), 4 adds wm hl (W2 and h both to 22 bits: <= 2^-23
                              //   dropped), 5 adds wl hh (W2 to 33 bits).  The layer's VALUE is fp32-grade with 3; its SIGN is the ReLU mask the
                              //   backward consumes, and with 3 terms a few more rounding-level ties (|pre-activation| ~ 1e-7) fall the other
                              //   way than in an fp64 evaluation (profiles/r02_split_terms.txt: 8-9 of 2.3e8 decisions for every variant,
                              //   torch fp32: 6).  Round 2 shipped 4 terms because its gradient-level gate compared gradients ACROSS such
                              //   flips (one flipped unit moves dW1 by ~4e-4 of its norm); round 3's parity tests impose the kernels' own
                              //   decisions on the oracle (tests/helpers.py), under which 3 and 4 terms give the same errors
                              //   (gpurun_out/parity_report_{default,s3}.tsv) - and 3 is two MFMAs per key cheaper (-0.5 ms per 8-bag step).
#endif
#ifndef SMML_DELTA_FIX
#define SMML_DELTA_FIX 0      // 1: re-centre the rows of d bias in the position-bias backward (d bias_k - P_k sum_k d bias_k).
                              // A fused softmax backward leaves sum_k dS_k != 0 at the 1e-7 level (delta = rowsum(dO . O));
                              // sums that weight d bias with near-constant factors amplify it.  Measured (tests/diag_gterms.py):
                              // dW3 4x and, with every ReLU unit active, dW2 / db 3-4x closer to fp64; everything else
                              // unchanged; costs 0.3-0.45 ms of the 17.5 ms step (one more score read + an exp per pair).
#endif
#ifndef SMML_FWD_PAIR
#define SMML_FWD_PAIR 0       // 1: two keys per trip of the forward's position-bias loop (measurement variant)
#endif
#ifndef SMML_DELTA_EXACT
#define SMML_DELTA_EXACT 0    // 1: the dq pass forms delta = sum_k P_k dP_k from its own dP products in a first sweep over the keys (measurement
                              // variant, deform_attn.hip; + ~0.4 ms per 8-bag step).  Default: delta = rowsum(dO . O).
#endif
#ifndef SMML_CHAIN2_TERMS
#define SMML_CHAIN2_TERMS 2   // fp16 terms of the constant (W2 w3)^T in d h1 = (W2 w3)^T mask.  The mask operand is exact, so the
                              // only error is the constant's: 2 terms = 22 bits, a fixed relative perturbation <= 2^-23 of
                              // each (W2 w3)[out][in] - below what rounding d bias w3 and the 32-term fp32 dot cost the
                              // unfused evaluation.  3: the constant to 2^-33.
#endif
#ifndef SMML_G_TERMS
#define SMML_G_TERMS 2        // bf16 terms of g = h1 . d bias in the dW2 product of the position-bias backward.  2: every
                              // summand carries 16 mantissa bits (<= 2^-17 relative, round-to-nearest, unbiased) against an
                              // exact 0 / 1 mask operand, fp32 accumulation - the error of dW2 against an fp64 evaluation
                              // is unchanged to three digits vs 3 terms (tests/diag_gterms.py: it is set by ReLU mask flips
                              // and by delta = rowsum(dO . O)), and the kernel is 10 % faster.  3: fp32-grade summands.
#endif
#ifndef SMML_BWD_EXP
#define SMML_BWD_EXP 0        // measurement variants of the dq / dkv passes (wrong results): 1 no dP products, 2 no dQ products, 3 no dK / dV products
#endif
#ifndef SMML_BWD_TERMS
#define SMML_BWD_TERMS 3      // bf16 terms per operand in the dq pass: 3 = fp32-grade (six products per block), 2 = 16-bit operands (hi + mid,
                              // three products: measurement switch - d scores then carry 2^-17 errors, which the position-bias gradients
                              // (dW3) and the single-key case (dS = 0 exactly) do not pass the parity gate with, profiles/r02_split_terms.txt)
#endif
#ifndef SMML_DQ_OUT_TERMS
#define SMML_DQ_OUT_TERMS 2   // the dQ = dS K product of the dq pass (a plain output, like dK / dV): two terms; the d scores themselves
                              // (dP = V dO^T, SMML_BWD_TERMS) keep three
#endif
#ifndef SMML_DKV_TERMS
#define SMML_DKV_TERMS 2      // the dkv pass: hi + mid bf16 terms (16 operand bits, three products).  dK and dV are plain sums of products -
                              // nothing downstream recomputes from them, unlike the d scores of the dq pass - and land 6e-6 from fp64
                              // (l2; fp32 operands: 5e-7), inside the 1e-4 gate of every parity test; once the dropout hashes were out
                              // of the pass (r03) the third term's 24 MFMAs per tile were its longest pole: -0.33 ms per 8-bag step, A/B
                              // on one box.  3 = fp32-grade operands.
#endif
#ifndef SMML_FWD_QK16
#define SMML_FWD_QK16 1       // forward QK^T / PV on the 16-bit matrix pipe: every operand as fp16 hi + lo (RNE, 22 bits), three of the
                              // four cross products (hi hi, hi lo, lo hi; <= 2^-22 dropped) - 24 MFMAs of 32 cycles per 32-key tile
                              // instead of 64 fp32 MFMAs of 64 cycles (which run at the vector rate and share the ALUs).  0: fp32 MFMA.
#endif
#ifndef SMML_FMA_MIX
#define SMML_FMA_MIX 1          // residual of the fp16 split by v_fma_mix_f32 (one instruction per value instead of convert + subtract)
#endif
#ifndef SMML_FAST_MATH
#define SMML_FAST_MATH 1    // 1: hardware log2/exp2/rcp approximations (1 ulp) instead of the libm-accurate forms
#endif

namespace {

#if SMML_FAST_MATH
// |d| + 1 >= 1 is never subnormal: the raw v_log_f32 (log2) needs none of __logf's range handling
#ifndef SMML_RAW_LOG
#define SMML_RAW_LOG 1
#endif
__device__ __forceinline__ float slog1p(float d) {
#if SMML_RAW_LOG
  return copysignf(__builtin_amdgcn_logf(fabsf(d) + 1.0f) * 0.6931471805599453f, d);
#else
  return copysignf(__logf(fabsf(d) + 1.0f), d);
#endif
}
__device__ __forceinline__ float sexp(float x) { return __expf(x); }
__device__ __forceinline__ float srcp(float x) { return __builtin_amdgcn_rcpf(x); }
#else
__device__ __forceinline__ float slog1p(float d) { return signed_log1p(d); }
__device__ __forceinline__ float sexp(float x) { return expf(x); }
__device__ __forceinline__ float srcp(float x) { return 1.0f / x; }
#endif
// 2 relu(x) = x + |x|, exact.  On gfx950 v_add_f32 (with its free |.| modifier) is in the fast issue class (~1.1 ns per
// instruction per SIMD with two resident waves) while v_max_f32 is in the slow one (~2.0 ns) -
// tests/microbench/valu_mix_probe.hip; the factor 2 is folded into the constants downstream (powers of two: exact).
__device__ __forceinline__ float relu2(float x) { return x + __builtin_fabsf(x); }

// Position transform of the bias MLP's input.  The kernels' position-dimension template argument PDX is 1 or 2 (signed-log offsets, the
// reference's default) or 3 = one dimension with the RAW offset (DeformableAttention1D.py:92 with cpb_log_distance = False; the 2-D module has
// no such switch): PD = dimensions, RAW = no log.  A template value, so the default path's instruction stream does not change.
template <int PDX> struct PosCfg {
  static constexpr int PD = (PDX == 3) ? 1 : PDX;
  static constexpr bool RAW = (PDX == 3);
};
template <bool RAW> __device__ __forceinline__ float pos_of(float d) { return RAW ? d : slog1p(d); }
// d pos / d d (times the [d != 0] factor autograd's sign() gives the log form); `big` = 2^100
template <bool RAW> __device__ __forceinline__ float dpos_of(float d, float big) {
  return RAW ? 1.f : srcp(fabsf(d) + 1.f) * fminf(fmaxf(fabsf(d) * big, 0.f), 1.f);
}
constexpr int DH = 64;       // head dim (fixed: dim_head = 64 in both reference modules)
constexpr int CH = 32;       // CPB hidden width = dim // 4 with dim = 128
constexpr int QT = 32;       // queries per wave
constexpr int WAVES = 4;     // waves per workgroup
constexpr int KT = 32;       // keys per tile

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
// v_mfma_f32_32x32x16_f16: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 8h + j], B[k = 8h + j][col r], j = 0..7;
// C/D layout as for the f32 form.  32 cycles per instruction on the matrix pipe, concurrent with VALU work.
__device__ __forceinline__ floatx16 mfma16(half8 a, half8 b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// split 8 fp32 values into fp16 hi + fp16 lo, both round-to-nearest: x - hi is exact in fp32, so hi + lo carries
// ~23 mantissa bits (residual <= 2^-24 |x|)
__device__ __forceinline__ void split8(const float (&x)[8], half8& hi, half8& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2v v = {x[2 * i], x[2 * i + 1]};
    const half2v h = __builtin_convertvector(v, half2v);
#if SMML_FMA_MIX
    // x - float(hi) in one mixed-precision fma per value (v_fma_mix_f32 reads the fp16 half directly)
    const unsigned hp = __builtin_bit_cast(unsigned, h);
    float2v r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(hp), "v"(v[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(hp), "v"(v[1]));
#else
    const float2v r = {x[2 * i] - (float)h[0], x[2 * i + 1] - (float)h[1]};
#endif
    const half2v l = __builtin_convertvector(r, half2v);
    hi[2 * i] = h[0]; hi[2 * i + 1] = h[1];
    lo[2 * i] = l[0]; lo[2 * i + 1] = l[1];
  }
}
// three-term split of a constant operand: hi + mid + lo reproduces the fp32 value exactly (33 bits)
__device__ __forceinline__ void split8_3(const float (&x)[8], half8& hi, half8& mid, half8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 h = (_Float16)x[i];
    const float r1 = x[i] - (float)h;
    const _Float16 m = (_Float16)r1;
    const _Float16 l = (_Float16)(r1 - (float)m);
    hi[i] = h; mid[i] = m; lo[i] = l;
  }
}
// four fp32 -> fp16 hi / lo planes (8 bytes each), both round-to-nearest
__device__ __forceinline__ void split4_h2(const float4 v, uint2v& hi, uint2v& lo) {
  const float2v a = {v.x, v.y}, b = {v.z, v.w};
  const half2v ha = __builtin_convertvector(a, half2v), hb = __builtin_convertvector(b, half2v);
  const float2v ra = {a[0] - (float)ha[0], a[1] - (float)ha[1]}, rb = {b[0] - (float)hb[0], b[1] - (float)hb[1]};
  const half2v la = __builtin_convertvector(ra, half2v), lb = __builtin_convertvector(rb, half2v);
  hi = (uint2v){__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  lo = (uint2v){__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}
// fp16 MFMA fragment of an operand stored k-major in LDS: two hardware-transposed reads (see smml_common.h lds_frag_tr)
__device__ __forceinline__ half8 lds_frag_tr_h(const _Float16* p0, const _Float16* p1) {
  typedef short short4v __attribute__((ext_vector_type(4)));
  typedef short short8v __attribute__((ext_vector_type(8)));
  typedef __attribute__((address_space(3))) short4v lds_s4;
  const short4v r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)p0);
  const short4v r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)p1);
  const short8v r = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
  return __builtin_bit_cast(half8, r);
}
constexpr int FRLD = 64 + 8;    // halves per row of a row-read fp16 image (144-byte rows)
constexpr int FTLD = 64 + 32;   // halves per row of a transposed-read fp16 image (192-byte rows)

// D += W . h with W = wh + wm + wl (exact) and h = bh + bl: all products down to 2^-22 of the leading one
__device__ __forceinline__ floatx16 mfma16_split(half8 wh, half8 wm, half8 wl, half8 bh, half8 bl, floatx16 d) {
#if SMML_SPLIT_TERMS == 5
  d = mfma16(wl, bh, d);
#endif
#if SMML_SPLIT_TERMS >= 4
  d = mfma16(wm, bl, d);        // 4: W2 to 22 bits (hi + mid; <= 2^-23 |w| dropped, below fp32's own product rounding), h to 22 bits
#endif
  d = mfma16(wm, bh, d);
  d = mfma16(wh, bl, d);
  return mfma16(wh, bh, d);
}

// Attention dropout (nn.Dropout on the softmax'd probabilities, DeformableAttention2D.py:309): a counter-based
// keep decision per (b, h, query, key) from a 64-bit seed - the same element gets the same decision in the
// forward and in both backward passes, no mask is stored.  keep_scale = 1 / (1 - p); thresh = p * 2^32.
struct DropCfg {
  unsigned long long seed;
  unsigned thresh;      // 16-bit threshold p * 2^16: keep iff the element's 16-bit half of its pair's hash >= thresh; 0 disables dropout
  float keep_scale;
  const unsigned long long* seed_dev;   // optional device-resident offset added to `seed` when the kernel runs (a launch captured in
                                        // a hipGraph bakes `seed` in; the offset lets every replay draw a new mask), or nullptr
};
// 64-bit finaliser (splitmix64, Steele / Lea / Flood): every input bit reaches every output bit
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// the seed a launch uses: one uniform load and two finalisers per wave.  The replay offset must NOT simply be added to the seed:
// drop_hash mixes idx + seed, so an offset that grows by k per replay would make the mask of replay r the mask of replay 0
// shifted by k r elements along the key axis.  The offset is hashed into a fresh 64-bit key instead.
__device__ __forceinline__ DropCfg drop_resolve(DropCfg dc) {
  if (dc.thresh && dc.seed_dev) dc.seed = mix64(dc.seed ^ mix64(*dc.seed_dev));
  dc.seed_dev = nullptr;
  return dc;
}
// counter-based keep decision: the 64-bit counter (element index + seed) is folded to 32 bits and run through a 32-bit
// avalanche mixer (two 32-bit multiplies; "lowbias32", C. Wellons).  For fewer than 2^32 elements the fold is injective, so
// no two elements of a launch share a mixer input.  A 64-bit splitmix here cost eight 32-bit multiplies per element and
// 0.5 ms of the 17 ms step (three kernels evaluate it per (query, key) pair).
__device__ __forceinline__ unsigned drop_hash(unsigned long long seed, unsigned long long idx) {
  const unsigned long long z = idx + seed;
  const unsigned hi = (unsigned)(z >> 32);
  unsigned x = (unsigned)z ^ ((hi << 13) | (hi >> 19)) ^ 0x9E3779B9u;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// One hash serves the TWO keys 2 jp, 2 jp + 1 of a (bag, head, query) row: its low / high 16 bits are compared with the 16-bit threshold
// p 2^16 (the drop rate is p to 2^-16; keep_scale uses the rate actually applied).  Pair index = row * ceil(J / 2) + jp.  Halves the
// mixer work of the forward (the backward passes read the decision from the saved score); -> bit 0: key 2 jp, bit 1: key 2 jp + 1
// the same decision pair from z = seed + pair index formed by the caller (one 64-bit add per tile, compile-time steps between its pairs)
__device__ __forceinline__ unsigned drop_keep2_z(const DropCfg& dc, unsigned long long z) {
  const unsigned h = drop_hash(0ull, z);
  return ((h & 0xFFFFu) >= dc.thresh ? 1u : 0u) | ((h >> 16) >= dc.thresh ? 2u : 0u);
}
__device__ __forceinline__ unsigned drop_keep2(const DropCfg& dc, unsigned long long pair_idx) {
  const unsigned h = drop_hash(dc.seed, pair_idx);
  return ((h & 0xFFFFu) >= dc.thresh ? 1u : 0u) | ((h >> 16) >= dc.thresh ? 2u : 0u);
}
// In training with dropout the forward stashes each element's keep decision in the LOWEST MANTISSA BIT of the score it saves for the
// backward (logits_t): both backward passes read those scores anyway and get the decision for free - the hash (two 32-bit multiplies
// + 64-bit index arithmetic per element) was a third of the dq / dkv passes' vector work.  The score moves by at most one ulp; the
// direction comes from its second-lowest bit, so the move is zero-mean whatever the decision, and the forward's own softmax uses the
// stashed value: forward and backward see the same probabilities.
__device__ __forceinline__ float stash_keep(float x, bool keep) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const unsigned flip = (u ^ (keep ? 1u : 0u)) & 1u;                    // lowest bit differs from the decision
  const int dir = (u & 0x7FFFFFFEu) ? (int)(u & 2u) - 1 : 1;            // +1 / -1: independent of the decision (never below +-0)
  return __builtin_bit_cast(float, u + (unsigned)(flip ? dir : 0));
}
// one-instruction form (the region forward is bound by its vector instructions): the lowest bit is REPLACED by the decision - the score
// moves by at most one ulp toward zero (2^-24 relative, against the 1e-4 parity gate); the backward passes read the bit the same way
__device__ __forceinline__ float stash_keep_trunc(float x, unsigned keep) {
  return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, x) & 0xFFFFFFFEu) | keep);
}
__device__ __forceinline__ float stashed_factor(float x, float keep_scale) {
  return (__builtin_bit_cast(unsigned, x) & 1u) ? keep_scale : 0.f;
}

// max over the 64 lanes (prologue use only)
__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
// Power-of-two lift of a constant MFMA operand that is split into fp16 terms: hi + mid (+ lo) only carries 22 (33) bits
// while the residual terms stay NORMAL fp16 numbers (>= 2^-14); for weights of size 0.01 .. 0.1 the second term is already
// subnormal (spacing 2^-24 absolute) and the split degrades to ~2^-20 relative.  Scaling the whole operand by 2^k so that its
// largest element sits near `target` (a power of two; exact) keeps every kept term normal; the consumer undoes the scale.
#ifndef SMML_LIFT_FWD
#define SMML_LIFT_FWD 0       // lift of W2 in the forward's layer-2 product.  OFF: measured (profiles/r02_split_terms.txt) it makes the saved
                              // ReLU masks flip MORE often against fp64 (dW1 1.06e-3 instead of 2.07e-4 with 5 terms) although every
                              // term is more precise - not understood; the reference's init (N(0, 1/sqrt(32)) = 0.18) keeps the second
                              // term of the unlifted split normal anyway
#endif
#ifndef SMML_LIFT_BWD
#define SMML_LIFT_BWD 1       // lift of the chain-2 constants (W2 w3)^T in the backward: d vs 90 x closer to fp64 when every unit is active
#endif
__device__ __forceinline__ float pow2_lift(float amax, float target, float lo, float hi) {
  if (!(amax > 0.f)) return 1.f;
  const float k = floorf(log2f(target / amax));
  return ldexpf(1.f, (int)fminf(fmaxf(k, lo), hi));      // an exact power of two (exp2f is the 1-ulp hardware approximation)
}

// one 32x32x16 block of a backward contraction from split operands (TERMS bf16 terms per operand: 3 = six products, 2 = three)
template <int TERMS>
__device__ __forceinline__ floatx16 bwd_prod(bf16x8 ah, bf16x8 am, bf16x8 al, bf16x8 bh, bf16x8 bm, bf16x8 bl, floatx16 d) {
  if (TERMS == 3) return mfma16b_x6(ah, am, al, bh, bm, bl, d);
  d = mfma16b(am, bh, d);
  d = mfma16b(ah, bm, d);
  return mfma16b(ah, bh, d);
}

struct CpbParams {
  const float* w1;  // [32, PD]
  const float* b1;  // [32]
  const float* w2;  // [32, 32]
  const float* b2;  // [32]
  const float* w3;  // [o, 32]
  const float* b3;  // [o]
};

// Layer 1 of the position-bias MLP on the matrix pipe, query-major (lane = query, accumulator rows = hidden channels):
// x[ch][q] = w1x[ch] p0[q] + w1y[ch] p1[q] + b1[ch] as ONE bf16 MFMA with weights and positions in three bf16 terms each.
// The forward, the position-bias backward and the decision export (relu1_masks_kernel) all go through these two functions,
// so the layer-1 ReLU decisions of the three are the same bits.
__device__ __forceinline__ bf16x8 cpb_l1_weights_q(float wx, float wy, int hf) {
  const __bf16 xh = (__bf16)wx; const float xr = wx - (float)xh; const __bf16 xm = (__bf16)xr;
  const __bf16 xl = (__bf16)(xr - (float)xm);
  const __bf16 yh = (__bf16)wy; const float yr = wy - (float)yh; const __bf16 ym = (__bf16)yr;
  const __bf16 yl = (__bf16)(yr - (float)ym);
  return hf == 0 ? (bf16x8){xh, yh, xh, yh, xh, yh, xl, yl} : (bf16x8){xm, ym, xm, ym, xm, ym, xl, yl};
}
struct PosTerms { unsigned hw, mw, lw; };       // {p0, p1} as three packed bf16 pairs (h + m + l = the fp32 values to 2^-24)
__device__ __forceinline__ PosTerms cpb_split_pos(float p0, float p1) {
  const float2v pv = {p0, p1};
  const bf16x2 hh = __builtin_convertvector(pv, bf16x2);
  const float2v r1 = bf16_residual2(pv, hh);
  const bf16x2 mm = __builtin_convertvector(r1, bf16x2);
  const float2v r2 = bf16_residual2(r1, mm);
  const bf16x2 ll = __builtin_convertvector(r2, bf16x2);
  return PosTerms{__builtin_bit_cast(unsigned, hh), __builtin_bit_cast(unsigned, mm), __builtin_bit_cast(unsigned, ll)};
}
__device__ __forceinline__ floatx16 cpb_layer1_q(bf16x8 a1, const PosTerms& t, int hf, floatx16 b1acc) {
  const uint4v bw = {t.hw, t.mw, t.lw, hf ? t.mw : t.hw};
  return mfma16b(a1, __builtin_bit_cast(bf16x8, bw), b1acc);
}

constexpr int VBLD = DH + 8;    // halves per row of a V plane
constexpr int KBLD = DH + 32;   // halves per row of a K plane
constexpr float LOG2E = 1.4426950408889634f;

#if SMML_FAST_MATH
// exp(l - lse) as one fma + v_exp_f32: nl = -lse * log2(e)
__device__ __forceinline__ float prob_of(float l, float nl) { return __builtin_amdgcn_exp2f(fmaf(l, LOG2E, nl)); }
__device__ __forceinline__ float prob_bias(float lse) { return -lse * LOG2E; }
#else
__device__ __forceinline__ float prob_of(float l, float nl) { return expf(l + nl); }
__device__ __forceinline__ float prob_bias(float lse) { return -lse; }
#endif

constexpr int QBLD = DH + 32;  // halves per row of a Q / dO plane: 192-byte rows (four rows of a transposed read on disjoint banks)
constexpr int DKV_KEYS = KT * WAVES;

constexpr int CPB_SLAB = 1024 + 64 + 32 + 32 + 32 + 8;   // 1192 floats
constexpr int CPB_XQ = 2 * 32;                            // double-buffered d bias of the wave's 32 queries
// Register budget (256 per wave, two waves per SIMD):
//   * layer 1 itself runs on the matrix pipe (K = 16 bf16 products with every factor in three terms): two 4-register
//     constant operands + b1 in accumulator layout instead of 48 per-channel constants;
//   * no software pipeline inside the wave (the sibling wave is the pipeline): nothing of the previous key is alive;
//   * the layer-1 weights of the d vs product are read from a 256-byte LDS table when they are needed.
constexpr int CPB2_STG_KEYS = 16;                                   // d vs staging rows per wave
constexpr int CPB2_WAVE_LDS = CPB_XQ + 2 * CPB2_STG_KEYS * 65;      // floats
constexpr int CPB2_TAB = 2 * 2 * 16;                                // {w1x, w1y} of ch(r) for both lane halves

// dK = scale * sum_part dKp[part], dV = sum_part dVp[part]   (n4 float4 elements per part)
__global__ void dkv_reduce_kernel(const float4* __restrict__ dKp, const float4* __restrict__ dVp, float4* __restrict__ dK,
                                  float4* __restrict__ dV, size_t n4, int nparts, float scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 sk = dKp[i], sv = dVp[i];
  for (int p = 1; p < nparts; ++p) {
    const float4 a = dKp[(size_t)p * n4 + i], bq = dVp[(size_t)p * n4 + i];
    sk.x += a.x; sk.y += a.y; sk.z += a.z; sk.w += a.w;
    sv.x += bq.x; sv.y += bq.y; sv.z += bq.z; sv.w += bq.w;
  }
  dK[i] = make_float4(sk.x * scale, sk.y * scale, sk.z * scale, sk.w * scale);
  dV[i] = sv;
}

// dVS[(b, g)][j][0..PD) = sum over the heads of the group, the query tiles and the four waves of a workgroup of the slab rows, in
// that fixed order: four lanes per output (one per wave slot) walk the o * qtiles workgroups, then combine.
__global__ __launch_bounds__(256) void dvs_reduce_kernel(const float2* __restrict__ rows, float* __restrict__ dVS, int Bn, int G,
                                                         int H, int qtiles, int J, int PD) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int w = (int)(t & 3);
  const long long out = t >> 2;                       // (b * G + g) * J + j
  const bool ok = out < (long long)Bn * G * J;
  float2 s = make_float2(0.f, 0.f);
  if (ok) {
    const int j = (int)(out % J);
    const int bg = (int)(out / J), b = bg / G, g = bg - b * G, o = H / G;
    for (int oi = 0; oi < o; ++oi) {
      const float2* p = rows + ((size_t)((b * H + g * o + oi) * qtiles) * WAVES + w) * J + j;
#pragma unroll 8
      for (int x = 0; x < qtiles; ++x) {
        const float2 v = p[(size_t)x * WAVES * J];
        s.x += v.x; s.y += v.y;
      }
    }
  }
  // lanes 4 k .. 4 k + 3 hold the four wave slots of one output: (w0 + w1) + (w2 + w3)
  s.x += __shfl_xor(s.x, 1); s.y += __shfl_xor(s.y, 1);
  s.x += __shfl_xor(s.x, 2); s.y += __shfl_xor(s.y, 2);
  if (ok && w == 0) {
    dVS[out * PD] = s.x;
    if (PD == 2) dVS[out * PD + 1] = s.y;
  }
}

// sums the per-workgroup slabs in two deterministic stages.  Stage 1: block (x, y) adds the slabs of chunk y for the 64
// outputs of column block x, separately for the two output rows oi = h % o that dW3 / db3 distinguish.
constexpr int CPB_RED_CHUNKS = 64;
__global__ void cpb_partial_kernel(const float* __restrict__ slab, int nwg, int o, int wg_per_head, int H, int chunk,
                                   float* __restrict__ part_out) {
  // slabs are ordered (b, h, qtile)
  const int k = blockIdx.x * 64 + (threadIdx.x & 63);
  const int part = threadIdx.x >> 6, nparts = blockDim.x >> 6;
  const int w0 = blockIdx.y * chunk, w1 = min(w0 + chunk, nwg);
  __shared__ float acc[4][64][2];
  float s0 = 0.f, s1 = 0.f;     // s0: rows with oi == 0 (or shared), s1: oi == 1
  if (k < CPB_SLAB) {
    for (int w = w0 + part; w < w1; w += nparts) {
      const int hh = (w / wg_per_head) % H;
      const int oi = hh % o;
      const float v = slab[(size_t)w * CPB_SLAB + k];
      if (oi == 0) s0 += v; else if (oi == 1) s1 += v;
    }
  }
  acc[part][threadIdx.x & 63][0] = s0;
  acc[part][threadIdx.x & 63][1] = s1;
  __syncthreads();
  if (part == 0 && k < CPB_SLAB) {
    float t0 = 0.f, t1 = 0.f;
    for (int p = 0; p < nparts; ++p) { t0 += acc[p][threadIdx.x][0]; t1 += acc[p][threadIdx.x][1]; }
    part_out[((size_t)blockIdx.y * CPB_SLAB + k) * 2 + 0] = t0;
    part_out[((size_t)blockIdx.y * CPB_SLAB + k) * 2 + 1] = t1;
  }
}
// Stage 2: one thread per output adds the chunk partials and scatters into the parameter gradients.
__global__ void cpb_final_kernel(const float* __restrict__ part_in, int nchunks, int o, float* __restrict__ dW1,
                                 float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2,
                                 float* __restrict__ dW3, float* __restrict__ db3, int PD) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= CPB_SLAB) return;
  float t0 = 0.f, t1 = 0.f;
  for (int p = 0; p < nchunks; ++p) {
    t0 += part_in[((size_t)p * CPB_SLAB + k) * 2 + 0];
    t1 += part_in[((size_t)p * CPB_SLAB + k) * 2 + 1];
  }
  if (k < 1024) dW2[k] = t0 + t1;
  else if (k < 1024 + 64) {
    const int ch = (k - 1024) >> 1, comp = (k - 1024) & 1;
    if (comp < PD) dW1[ch * PD + comp] = t0 + t1;
  } else if (k < 1024 + 96) db1[k - 1088] = t0 + t1;
  else if (k < 1024 + 128) db2[k - 1120] = t0 + t1;
  else if (k < 1024 + 160) {
    dW3[k - 1152] = t0;
    if (o > 1) dW3[CH + k - 1152] = t1;
  } else if (k == 1024 + 160) {
    db3[0] = t0;
    if (o > 1) db3[1] = t1;
  }
}

// query slices of backward pass 2: enough workgroups to fill the chip a few times over, at most 16 slabs
static int dkv_parts(int B, int N, int J, int H) {
  const int nkg = (J + DKV_KEYS - 1) / DKV_KEYS, nqt = (N + QT - 1) / QT;
  const long base = (long)nkg * H * B;
  long parts = (1280 + base / 2) / base;
  if (parts < 1) parts = 1;
  if (parts > 16) parts = 16;
  if (parts > nqt) parts = nqt;
  return (int)parts;
}
// workspace layout (floats): [CPB slabs nwg * CPB_SLAB][stage-1 partials CHUNKS * CPB_SLAB * 2]
//                            [dK slabs parts * B*J*H*64][dV slabs parts * B*J*H*64]
struct BwdWorkspace {
  size_t slab, partial, dkp, dvp, rho, dvs, total;   // float offsets / total floats
};
static BwdWorkspace bwd_workspace(int B, int N, int J, int H) {
  BwdWorkspace w;
  const size_t nwg = (size_t)B * H * ((N + QT * WAVES - 1) / (QT * WAVES));
  const size_t kv = (size_t)dkv_parts(B, N, J, H) * B * J * H * DH;
  w.slab = 0;
  w.partial = nwg * CPB_SLAB;
  w.dkp = (w.partial + (size_t)CPB_RED_CHUNKS * CPB_SLAB * 2 + 3) & ~(size_t)3;
  w.dvp = w.dkp + kv;
  w.rho = w.dvp + kv;                      // [B, H, N] row sums of d scores (SMML_DELTA_FIX)
  w.dvs = w.rho + (((size_t)B * H * N + 3) & ~(size_t)3);     // [nwg * WAVES][J][2] d vs rows of the position-bias backward
  w.total = w.dvs + nwg * WAVES * (size_t)J * 2;
  return w;
}

// Sizes the fused attention families accept: grid dimensions B, H <= 65535, N <= 2^26 queries, J <= 2^22 keys - every index and size
// expression of the launch geometry then stays inside 63 bits (and the int ones inside 31); larger values are refused, not wrapped
// (tests/test_host_sanitizers.py walks the entry points under UBSan).
constexpr int SMML_MAX_QUERIES = 1 << 26, SMML_MAX_KEYS = 1 << 22;
bool deform_dims_ok(int B, int N, int J, int H) {
  return B > 0 && N > 0 && J > 0 && H > 0 && B <= 65535 && H <= 65535 && N <= SMML_MAX_QUERIES && J <= SMML_MAX_KEYS;
}
// position-dimension template value of a launch (PosCfg): 2 | 1 | 3 = one dimension, raw offsets (opts->raw_distance)
int pdx_of(int posdim, const SmmlDeformOpts* opts) { return posdim == 2 ? 2 : ((opts && opts->raw_distance) ? 3 : 1); }
// opts->seed_offset is read into the launch's DropCfg and dereferenced on the device only
DropCfg make_drop(float p, unsigned long long seed, const SmmlDeformOpts* opts) {
  DropCfg dc;
  dc.seed = seed;
  dc.seed_dev = opts ? opts->seed_offset : nullptr;
  dc.thresh = (p > 0.f) ? (unsigned)fmax(1.0, (double)p * 65536.0) : 0u;
  dc.keep_scale = (p > 0.f) ? (float)(65536.0 / (65536.0 - (double)dc.thresh)) : 1.0f;
  return dc;
}

}  // namespace
