// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of the multimodal-MIL attention path.
// Error convention of the C-ABI (include/smml.h): 0 = ok, negative = error, text via smml_last_error().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define SMML_OK 0
#define SMML_ERR_ARG -1
#define SMML_ERR_HIP -2

void smml_set_error(const char* fmt, ...);

#ifndef SMML_DEFORM_OPTS_DEFINED
#define SMML_DEFORM_OPTS_DEFINED
/* Optional behaviour of ONE fused deformable-attention launch (trailing `opts` argument of the entry points below; NULL = all defaults).
 * Passed by the caller with every call: the library keeps no per-thread or global launch state. */
typedef struct SmmlDeformOpts {
  const unsigned long long* seed_offset; /* device-resident offset hashed into dropout_seed when the kernel RUNS (a launch captured in a hipGraph
                                            bakes dropout_seed in; the offset lets every replay draw a new mask), or NULL */
  int raw_distance;                      /* 1: posdim-1 launches feed the bias MLP the raw offset gq - vs (DeformableAttention1D.py:92,
                                            cpb_log_distance = False); 0: its signed log (the reference's default; posdim 2 has no such switch) */
  const unsigned short* mask_table;      /* smml_deform_attn16_bwd with relu_masks == NULL: layer-2 ReLU decisions from this table (filled by
                                            smml_cpb_mask_table with mask_table_pmax), or NULL = recompute layer 2 per pair */
  float mask_table_pmax;
  unsigned short* export_masks;          /* tests: smml_deform_attn16_bwd with relu_masks == NULL also writes the decisions it used here
                                            ([B, H, nst / 32, J, 2, 32] u16, the forward's layout), or NULL */
  int region_lds_cap;                    /* tests: the region entry points keep only regions with an id below this in LDS and take the others from
                                            global memory (the path of parameter sets with more than 2048 linear regions); 0 = the default (2048) */
} SmmlDeformOpts;
#endif

#define SMML_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      smml_set_error(__VA_ARGS__);         \
      return SMML_ERR_ARG;                 \
    }                                      \
  } while (0)

#define SMML_LAUNCH_CHECK(name)                                                     \
  do {                                                                              \
    hipError_t e_ = hipGetLastError();                                              \
    if (e_ != hipSuccess) {                                                         \
      smml_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));         \
      return SMML_ERR_HIP;                                                          \
    }                                                                               \
  } while (0)

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: exact-fp32 matrix core op (k-ordered fmaf chain), 64 FLOP/clk/SIMD.
//   A operand: lane l holds A[i = l & 31][k = l >> 5];  B operand: lane l holds B[k = l >> 5][j = l & 31]
//   C/D:       lane l, register r holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
__device__ __forceinline__ floatx16 mfma32(float a, float b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// row of the 32x32 accumulator tile held in register r of a lane in half hf
__device__ __forceinline__ constexpr int acc_row(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// ---- 16-bit matrix pipe helpers (v_mfma_f32_32x32x16_bf16) and the three-term bf16 split of fp32 data ----
//   A operand: lane l holds A[i = l & 31][k = 8 (l >> 5) + j], B operand: B[k = 8 (l >> 5) + j][j' = l & 31], j = 0..7;
//   C/D as for the f32 form.  x = h + m + l with three round-to-nearest bf16 terms reproduces fp32 x to 2^-24 with fp32's
//   exponent range; the six products h h, h m, m h, h l, l h, m m of two such splits give a product to <= 2^-23.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ floatx16 mfma16b(bf16x8 a, bf16x8 b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// all six kept products of a (ah, am, al) x (bh, bm, bl) block, smallest terms first
__device__ __forceinline__ floatx16 mfma16b_x6(bf16x8 ah, bf16x8 am, bf16x8 al, bf16x8 bh, bf16x8 bm, bf16x8 bl, floatx16 d) {
  d = mfma16b(am, bm, d);
  d = mfma16b(al, bh, d);
  d = mfma16b(ah, bl, d);
  d = mfma16b(am, bh, d);
  d = mfma16b(ah, bm, d);
  return mfma16b(ah, bh, d);
}
// four fp32 -> three planes of four bf16 (8 bytes each)
#ifndef SMML_DOT2_SPLIT
#define SMML_DOT2_SPLIT 0   // exact but not faster: fewer instructions, yet every DOT result costs 3 wait states (tests/microbench/README.md)
#endif
// v - float(h) for a packed bf16 pair h = rn(v): exact (the residual of a rounding fits fp32).  v_dot2_f32_bf16 against
// the constant pairs {-1, 0} / {0, -1} reads the packed halves directly - one instruction per value instead of unpacking
// (shift / and) and subtracting; exactness incl. subnormals checked by tests/microbench/dot2_probe.hip.  Through the
// builtin, never inline asm: a DOT result needs 3 wait states before another VALU instruction reads it, which only the
// compiler's hazard recognizer inserts.
__device__ __forceinline__ float2v bf16_residual2(const float2v v, const bf16x2 h) {
#if SMML_DOT2_SPLIT
  // the constant pairs are kept opaque in VGPRs: folded to an inline constant, {-1, 0} is encoded as "-1.0", which
  // the instruction does not read as a bf16 pair (dot2_probe.hip caught it)
  unsigned c0, c1;
  asm("v_mov_b32 %0, 0x0000bf80" : "=v"(c0));
  asm("v_mov_b32 %0, 0xbf800000" : "=v"(c1));
  return (float2v){__builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, c0), v[0], false),
                   __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, c1), v[1], false)};
#else
  return (float2v){v[0] - (float)h[0], v[1] - (float)h[1]};
#endif
}
__device__ __forceinline__ void split4_bf3(const float4 v, uint2v& h, uint2v& m, uint2v& l) {
  const float2v a = {v.x, v.y}, b = {v.z, v.w};
  const bf16x2 ha = __builtin_convertvector(a, bf16x2), hb = __builtin_convertvector(b, bf16x2);
  const float2v ra = bf16_residual2(a, ha), rb = bf16_residual2(b, hb);
  const bf16x2 ma = __builtin_convertvector(ra, bf16x2), mb = __builtin_convertvector(rb, bf16x2);
  const float2v sa = bf16_residual2(ra, ma), sb = bf16_residual2(rb, mb);
  const bf16x2 la = __builtin_convertvector(sa, bf16x2), lb = __builtin_convertvector(sb, bf16x2);
  h = (uint2v){__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  m = (uint2v){__builtin_bit_cast(unsigned, ma), __builtin_bit_cast(unsigned, mb)};
  l = (uint2v){__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}
// eight fp32 -> three bf16x8 operands
__device__ __forceinline__ void split8_bf3(const float (&x)[8], bf16x8& a, bf16x8& b, bf16x8& c) {
  uint2v h0, m0, l0, h1, m1, l1;
  split4_bf3(make_float4(x[0], x[1], x[2], x[3]), h0, m0, l0);
  split4_bf3(make_float4(x[4], x[5], x[6], x[7]), h1, m1, l1);
  a = __builtin_bit_cast(bf16x8, (uint4v){h0[0], h0[1], h1[0], h1[1]});
  b = __builtin_bit_cast(bf16x8, (uint4v){m0[0], m0[1], m1[0], m1[1]});
  c = __builtin_bit_cast(bf16x8, (uint4v){l0[0], l0[1], l1[0], l1[1]});
}
// MFMA fragment of an operand stored k-major in LDS ([k][rows], 16-bit): two hardware-transposed reads
// (ds_read_b64_tr_b16; lane 4 q + p of each 16-lane group addresses row q, columns 4 p .. 4 p + 3 of a 4 x 16 block and
// receives its own column of the 4 rows - tests/microbench/tr_probe.hip).  p0 / p1: this lane's addresses in the two 4-row blocks.
__device__ __forceinline__ bf16x8 lds_frag_tr(const __bf16* p0, const __bf16* p1) {
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const bf16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
  const bf16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p1);
  return (bf16x8){r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
}

#ifndef SMML_DPP_REDUCE
#define SMML_DPP_REDUCE 1   // cross-lane sums on the VALU (v_permlane32_swap / DPP) instead of ds_bpermute through LDS
#endif
// value of the other 32-lane half: v_permlane32_swap exchanges the upper half of its first operand with the
// lower half of its second, so with both operands = v the two results hold {lo, lo} and {hi, hi}
__device__ __forceinline__ void xhalf_pair(float v, float& lo, float& hi) {
  const unsigned u = __float_as_uint(v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  lo = __uint_as_float(r[0]);
  hi = __uint_as_float(r[1]);
}
#if SMML_DPP_REDUCE
__device__ __forceinline__ float xhalf_sum(float v) { float a, b; xhalf_pair(v, a, b); return a + b; }
__device__ __forceinline__ float xhalf_max(float v) { float a, b; xhalf_pair(v, a, b); return fmaxf(a, b); }
// sum over the 64 lanes, valid in EVERY lane: inclusive row scan with DPP row_shr 1/2/4/8 (lane 15 of each
// 16-lane row holds the row sum), readlane of the four row sums
__device__ __forceinline__ float wave_sum(float v) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x111, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x112, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x114, 0xf, 0xf, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x118, 0xf, 0xf, false));
  const unsigned u = __float_as_uint(v);
  return __uint_as_float(__builtin_amdgcn_readlane(u, 15)) + __uint_as_float(__builtin_amdgcn_readlane(u, 31)) +
         __uint_as_float(__builtin_amdgcn_readlane(u, 47)) + __uint_as_float(__builtin_amdgcn_readlane(u, 63));
}
#else
__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }
__device__ __forceinline__ float xhalf_max(float v) { return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#endif
// wave-local ordering of LDS traffic: LDS ops of one wave complete in issue order; this only
// stops the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// sign(d) * log(|d| + 1)  (continuous position bias input transform)
__device__ __forceinline__ float signed_log1p(float d) { return copysignf(logf(fabsf(d) + 1.0f), d); }
