// bf16-storage forms of the n'-sized elementwise / stencil kernels of the Nystrom block's bf16 compute mode: landmark means, their
// backward, and the depthwise residual convolution (models/NystromAttention.py:62-66,102-118,144-145), all reading q / k / v where the
// projection GEMM wrote them - the token-major bf16 buffer [b, n', 3, h, D] - and writing gradients into a buffer of the same layout.
// Arithmetic is fp32; only loads and stores are bf16 (8 or 16 bytes per lane).
#include "smml_common.h"

namespace {

__device__ __forceinline__ void ld8(const __bf16* p, float (&x)[8]) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = (float)v[i];
}
__device__ __forceinline__ void st8(__bf16* p, const float (&x)[8]) {
  uint4v w;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float2v v = {x[2 * i], x[2 * i + 1]};
    w[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  }
  *reinterpret_cast<uint4v*>(p) = w;
}
__device__ __forceinline__ float4 ld4(const __bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(__bf16* p, const float4 v) {
  const float2v a = {v.x, v.y}, b = {v.z, v.w};
  *reinterpret_cast<uint2v*>(p) = (uint2v){__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2)),
                                           __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2))};
}

// qmean / kmean [b, h, j, d] = 1/l sum_{i < l} qkv[b, j l + i, which, h, d]     which = 0 (q), 1 (k);  C = h D channels per part
// grid (m, B), C / 4 threads (2 C channels, 8 per thread)
__global__ void segment_mean_b16_kernel(const __bf16* __restrict__ qkv, float* __restrict__ qmean, float* __restrict__ kmean, int B, int n, int l,
                                        int Hh, int D) {
  const int j = blockIdx.x, b = blockIdx.y, C = Hh * D, m = n / l;
  const int ch = threadIdx.x * 8;                       // 0 .. 2 C - 8
  const __bf16* p = qkv + ((size_t)b * n + (size_t)j * l) * (3 * C) + ch;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int i = 0;
  for (; i + 4 <= l; i += 4) {                          // four rows in flight
    float x0[8], x1[8], x2[8], x3[8];
    ld8(p + (size_t)i * 3 * C, x0); ld8(p + (size_t)(i + 1) * 3 * C, x1); ld8(p + (size_t)(i + 2) * 3 * C, x2); ld8(p + (size_t)(i + 3) * 3 * C, x3);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (x0[e] + x1[e]) + (x2[e] + x3[e]);
  }
  for (; i < l; ++i) {
    float x0[8];
    ld8(p + (size_t)i * 3 * C, x0);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += x0[e];
  }
  const int which = ch / C, cc = ch - which * C, h = cc / D, d = cc - h * D;
  float* o = (which ? kmean : qmean) + (((size_t)b * Hh + h) * m + j) * D + d;
  const float inv = 1.f / (float)l;
  *reinterpret_cast<float4*>(o) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
  *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv);
}

// dqkv[b, t, which, h, d] += (dqmean | dkmean)[b, h, t / l, d] / l    for which = 0, 1 (the q and k parts of the gradient buffer)
__global__ void segment_mean_bwd_add_b16_kernel(__bf16* __restrict__ dqkv, const float* __restrict__ dqmean, const float* __restrict__ dkmean, int B,
                                                int n, int l, int Hh, int D) {
  const int C = Hh * D, m = n / l, per_row = 2 * C / 8;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * n * per_row) return;
  const int ch = (int)(idx % per_row) * 8;
  const size_t row = idx / per_row;                     // b n + t
  const int t = (int)(row % n), b = (int)(row / n);
  const int which = ch / C, cc = ch - which * C, h = cc / D, d = cc - h * D;
  const float* g = (which ? dkmean : dqmean) + (((size_t)b * Hh + h) * m + t / l) * D + d;
  const float4 g0 = *reinterpret_cast<const float4*>(g), g1 = *reinterpret_cast<const float4*>(g + 4);
  __bf16* p = dqkv + row * (3 * C) + ch;
  float x[8];
  ld8(p, x);
  const float inv = 1.f / (float)l;
  x[0] = fmaf(g0.x, inv, x[0]); x[1] = fmaf(g0.y, inv, x[1]); x[2] = fmaf(g0.z, inv, x[2]); x[3] = fmaf(g0.w, inv, x[3]);
  x[4] = fmaf(g1.x, inv, x[4]); x[5] = fmaf(g1.y, inv, x[5]); x[6] = fmaf(g1.z, inv, x[6]); x[7] = fmaf(g1.w, inv, x[7]);
  st8(p, x);
}

// Strip form of the residual convolution (33 taps), as resconv_strip_kernel of nystrom.hip: one thread owns 4 consecutive d and a strip
// of 16 consecutive tokens, walks the 48 input rows the strip meets once and feeds every output a row belongs to from registers.
//   MODE 0: out[b, t, h, d] = sum_k w[h, k]      in[b, t + k - 16, h, d]
//   MODE 1: out[b, t, h, d] = sum_k w[h, 32 - k] in[b, t + k - 16, h, d]      (gradient with respect to the input)
// in / out element (b, t, h, d) at b * bs + t * rs + h * D + d of their (bf16) buffers.
constexpr int RB_T = 16, RB_KW = 33, RB_HALF = 16;
template <int MODE>
__global__ __launch_bounds__(256) void resconv_b16_kernel(const __bf16* __restrict__ in, const float* __restrict__ w, __bf16* __restrict__ out,
                                                          int B, int Hh, int n, int D, long long i_bs, long long i_rs, long long o_bs,
                                                          long long o_rs) {
  __shared__ float ws[64 * RB_KW];
  for (int i = threadIdx.x; i < Hh * RB_KW; i += 256) {
    const int h = i / RB_KW, k = i - h * RB_KW;
    ws[i] = w[h * RB_KW + (MODE ? RB_KW - 1 - k : k)];
  }
  __syncthreads();
  const int d4n = D >> 2, strips = (n + RB_T - 1) / RB_T;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Hh * strips * d4n) return;
  // d fastest, then head, then strip: consecutive lanes read consecutive bytes of a token row (all heads: h D contiguous elements)
  const int d4 = (int)(idx % d4n);
  const int h = (int)((idx / d4n) % Hh);
  const int st = (int)((idx / ((long long)d4n * Hh)) % strips);
  const int b = (int)(idx / ((long long)d4n * Hh * strips));
  const int t0 = st * RB_T;
  const __bf16* ip = in + (long long)b * i_bs + h * D + d4 * 4;
  float wk[RB_KW];
#pragma unroll
  for (int k = 0; k < RB_KW; ++k) wk[k] = ws[h * RB_KW + k];
  float4 acc[RB_T];
#pragma unroll
  for (int i = 0; i < RB_T; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < RB_T + RB_KW - 1; ++j) {
    // branch-free: a clamped row is loaded and zeroed (a branch around each load makes hipcc wait for every one in turn)
    const int tj = t0 - RB_HALF + j;
    const float ok = (tj >= 0 && tj < n) ? 1.f : 0.f;
    float4 x = ld4(ip + (long long)min(max(tj, 0), n - 1) * i_rs);
    x.x *= ok; x.y *= ok; x.z *= ok; x.w *= ok;
#pragma unroll
    for (int i = 0; i < RB_T; ++i) {
      const int k = j - i;
      if (k >= 0 && k < RB_KW) {
        acc[i].x = fmaf(wk[k], x.x, acc[i].x); acc[i].y = fmaf(wk[k], x.y, acc[i].y);
        acc[i].z = fmaf(wk[k], x.z, acc[i].z); acc[i].w = fmaf(wk[k], x.w, acc[i].w);
      }
    }
  }
  __bf16* op = out + (long long)b * o_bs + h * D + d4 * 4;
#pragma unroll
  for (int i = 0; i < RB_T; ++i)
    if (t0 + i < n) st4(op + (long long)(t0 + i) * o_rs, acc[i]);
}

// dw[h, k] += sum_{b, t, d} dout[b, t, h, d] v[b, t + k - 16, h, d]: a strip's 16 dout rows sit in registers, every v row is read once.
// A block is (b, h, 256 / (D / 4) strips side by side x RB_SPT in a row); 33 sums reduced by shuffles, LDS, then 33 atomics per block.
constexpr int RB_SPT = 2;
__global__ __launch_bounds__(256) void resconv_b16_wgrad_kernel(const __bf16* __restrict__ dout, const __bf16* __restrict__ v, float* __restrict__ dw,
                                                                int B, int Hh, int n, int D, long long g_bs, long long g_rs, long long v_bs,
                                                                long long v_rs) {
  __shared__ float red[RB_KW];
  const int tid = threadIdx.x, d4n = D >> 2;
  const int h = blockIdx.y, b = blockIdx.z;
  if (tid < RB_KW) red[tid] = 0.f;
  __syncthreads();
  const int per = 256 / d4n;
  const int d4 = tid % d4n, sl = tid / d4n;
  const __bf16* vb = v + (long long)b * v_bs + h * D + d4 * 4;
  const __bf16* db = dout + (long long)b * g_bs + h * D + d4 * 4;
  float s[RB_KW];
#pragma unroll
  for (int k = 0; k < RB_KW; ++k) s[k] = 0.f;
  for (int rep = 0; rep < RB_SPT; ++rep) {
    const int t0 = ((blockIdx.x * RB_SPT + rep) * per + sl) * RB_T;
    if (t0 >= n) break;
    float4 g[RB_T];
#pragma unroll
    for (int i = 0; i < RB_T; ++i) {
      const float ok = (t0 + i < n) ? 1.f : 0.f;
      g[i] = ld4(db + (long long)min(t0 + i, n - 1) * g_rs);
      g[i].x *= ok; g[i].y *= ok; g[i].z *= ok; g[i].w *= ok;
    }
#pragma unroll
    for (int j = 0; j < RB_T + RB_KW - 1; ++j) {
      const int tj = t0 - RB_HALF + j;
      const float ok = (tj >= 0 && tj < n) ? 1.f : 0.f;
      float4 x = ld4(vb + (long long)min(max(tj, 0), n - 1) * v_rs);
      x.x *= ok; x.y *= ok; x.z *= ok; x.w *= ok;
#pragma unroll
      for (int i = 0; i < RB_T; ++i) {
        const int k = j - i;
        if (k >= 0 && k < RB_KW) s[k] = fmaf(g[i].x, x.x, fmaf(g[i].y, x.y, fmaf(g[i].z, x.z, fmaf(g[i].w, x.w, s[k]))));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < RB_KW; ++k) {
    float t = s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((tid & 63) == 0) atomicAdd(&red[k], t);
  }
  __syncthreads();
  if (tid < RB_KW) atomicAdd(&dw[h * RB_KW + tid], red[tid]);
}

bool al16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

// qkv bf16 [B, n, 3, H, D] (n = m l) -> qmean, kmean fp32 [B, H, m, D] (landmark means of q and of k)
int smml_segment_mean_b16(const void* qkv, float* qmean, float* kmean, int B, int n, int l, int H, int D, void* stream) {
  SMML_REQUIRE(qkv && qmean && kmean && al16(qkv) && al16(qmean) && al16(kmean), "smml_segment_mean_b16: null or misaligned pointer");
  SMML_REQUIRE(B > 0 && B <= 65535 && n > 0 && l > 0 && n % l == 0 && H > 0 && D > 0 && D % 8 == 0, "smml_segment_mean_b16: bad sizes (n=%d l=%d H=%d D=%d)", n, l, H, D);
  const int threads = 2 * H * D / 8;
  SMML_REQUIRE(threads <= 1024, "smml_segment_mean_b16: H D too large (%d)", H * D);
  hipLaunchKernelGGL(segment_mean_b16_kernel, dim3(n / l, B), dim3(threads), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(qkv), qmean, kmean, B, n, l, H, D);
  SMML_LAUNCH_CHECK("smml_segment_mean_b16");
  return SMML_OK;
}

// the backward of the above, ADDED to the q and k parts of dqkv bf16 [B, n, 3, H, D]
int smml_segment_mean_bwd_add_b16(void* dqkv, const float* dqmean, const float* dkmean, int B, int n, int l, int H, int D, void* stream) {
  SMML_REQUIRE(dqkv && dqmean && dkmean && al16(dqkv) && al16(dqmean) && al16(dkmean), "smml_segment_mean_bwd_add_b16: null or misaligned pointer");
  SMML_REQUIRE(B > 0 && n > 0 && l > 0 && n % l == 0 && H > 0 && D > 0 && D % 8 == 0, "smml_segment_mean_bwd_add_b16: bad sizes");
  const size_t total = (size_t)B * n * (2 * H * D / 8);
  hipLaunchKernelGGL(segment_mean_bwd_add_b16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<__bf16*>(dqkv), dqmean, dkmean, B, n, l, H, D);
  SMML_LAUNCH_CHECK("smml_segment_mean_bwd_add_b16");
  return SMML_OK;
}

// Depthwise residual convolution over tokens, 33 taps, bf16 in / out; element (b, t, h, d) of in at b i_bs + t i_rs + h D + d (elements),
// of out at b o_bs + t o_rs + h D + d.  flip = 0: out = sum_k w[h, k] in[t + k - 16] (forward);  1: taps reversed (gradient w.r.t. in).
int smml_resconv_b16(const void* in, const float* w, void* out, int B, int H, int n, int D, int KW, long long i_bs, long long i_rs,
                     long long o_bs, long long o_rs, int flip, void* stream) {
  SMML_REQUIRE(in && w && out, "smml_resconv_b16: null pointer");
  SMML_REQUIRE(KW == RB_KW, "smml_resconv_b16: the bf16 form is built for %d taps (got %d)", RB_KW, KW);
  SMML_REQUIRE(B > 0 && H > 0 && H <= 64 && n > 0 && D > 0 && D % 4 == 0, "smml_resconv_b16: bad sizes");
  SMML_REQUIRE((i_bs % 4) == 0 && (i_rs % 4) == 0 && (o_bs % 4) == 0 && (o_rs % 4) == 0 && (reinterpret_cast<size_t>(in) & 7) == 0 &&
               (reinterpret_cast<size_t>(out) & 7) == 0, "smml_resconv_b16: 8-byte aligned rows needed");
  const long long total = (long long)B * H * ((n + RB_T - 1) / RB_T) * (D / 4);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (flip) hipLaunchKernelGGL(resconv_b16_kernel<1>, grid, block, 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(in), w, reinterpret_cast<__bf16*>(out), B, H, n, D, i_bs, i_rs, o_bs, o_rs);
  else hipLaunchKernelGGL(resconv_b16_kernel<0>, grid, block, 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(in), w, reinterpret_cast<__bf16*>(out), B, H, n, D, i_bs, i_rs, o_bs, o_rs);
  SMML_LAUNCH_CHECK("smml_resconv_b16");
  return SMML_OK;
}

// dw[h, k] += sum dout[b, t, h, d] v[b, t + k - 16, h, d]   (dw fp32 [H, 33], accumulated: the caller zeroes it)
int smml_resconv_wgrad_b16(const void* dout, const void* v, float* dw, int B, int H, int n, int D, int KW, long long g_bs, long long g_rs,
                           long long v_bs, long long v_rs, void* stream) {
  SMML_REQUIRE(dout && v && dw, "smml_resconv_wgrad_b16: null pointer");
  SMML_REQUIRE(KW == RB_KW, "smml_resconv_wgrad_b16: built for %d taps (got %d)", RB_KW, KW);
  SMML_REQUIRE(B > 0 && B <= 65535 && H > 0 && H <= 65535 && n > 0 && D >= 4 && D <= 64 && (D & (D - 1)) == 0, "smml_resconv_wgrad_b16: bad sizes (D must be a power of two <= 64)");
  SMML_REQUIRE((g_bs % 4) == 0 && (g_rs % 4) == 0 && (v_bs % 4) == 0 && (v_rs % 4) == 0 && (reinterpret_cast<size_t>(dout) & 7) == 0 &&
               (reinterpret_cast<size_t>(v) & 7) == 0, "smml_resconv_wgrad_b16: 8-byte aligned rows needed");
  const int per = 256 / (D / 4);
  const int strips = (n + RB_T - 1) / RB_T;
  dim3 grid((strips + per * RB_SPT - 1) / (per * RB_SPT), H, B);
  hipLaunchKernelGGL(resconv_b16_wgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const __bf16*>(dout),
                     reinterpret_cast<const __bf16*>(v), dw, B, H, n, D, g_bs, g_rs, v_bs, v_rs);
  SMML_LAUNCH_CHECK("smml_resconv_wgrad_b16");
  return SMML_OK;
}

}  // extern "C"
