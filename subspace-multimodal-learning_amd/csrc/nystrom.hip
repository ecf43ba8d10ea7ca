// HBM-bound pieces of the Nystrom landmark self-attention block for gfx950 (the dense contractions run in
// gemm.hip on the matrix cores):
//   row softmax forward / backward           models/NystromAttention.py:137 (attn1, attn2, attn3; attn3's rows are n' long)
//   landmark segment means (+ backward tile) :102-118   (forward = colsum_kernel over [B*h*m, l, d]; backward = tile_rows)
//   depthwise residual conv along tokens     :72,144-145 (33 taps, per head, no bias), forward / data / weight gradients
//   PPEG depthwise 7x7 + 5x5 + 3x3 + identity models/mil.py:192-206 as ONE merged 7x7 depthwise pass, channel-last
// Layouts: scores [rows, L]; v [B, h, n', d]; conv output merged [B, n', h*d]; PPEG maps [B, H, W, C].
#include "smml_common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// one 256-thread block per row; rows of any length (three passes, the row stays L2/MALL resident)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int L) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const float* xr = x + row * L;
  float* yr = y + row * L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m = -INFINITY;
  for (int i = tid; i < L; i += 256) m = fmaxf(m, xr[i]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = tid; i < L; i += 256) s += expf(xr[i] - m);
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  for (int i = tid; i < L; i += 256) yr[i] = expf(xr[i] - m) * inv;
}

// dx = y * (dy - sum(dy * y))
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int L) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const float* yr = y + row * L;
  const float* dr = dy + row * L;
  float* xr = dx + row * L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f;
  for (int i = tid; i < L; i += 256) s = fmaf(yr[i], dr[i], s);
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float dot = red[0] + red[1] + red[2] + red[3];
  for (int i = tid; i < L; i += 256) xr[i] = yr[i] * (dr[i] - dot);
}

// short rows (L <= 1024): one wave per row, the row lives in registers
template <int NV>
__global__ __launch_bounds__(256) void softmax_fwd_wave_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               long long rows, int L) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * L;
  float v[NV];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < L) ? xr[c] : -INFINITY;
    m = fmaxf(m, v[i]);
  }
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) { v[i] = expf(v[i] - m); s += v[i]; }
  const float inv = 1.f / wave_sum(s);
  float* yr = y + row * L;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < L) yr[c] = v[i] * inv;
  }
}

// dst[b, r, c] = scale * src[b, c]  (backward of a mean over r)
__global__ void tile_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long long nb, int R, int C,
                                 float scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = nb * R * C;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long b = i / ((long long)R * C);
  dst[i] = src[b * C + c] * scale;
}

// out[b, t, h*D + d] (+)= sum_k w[h, k] * v[b, h, t + k - KW/2, d]     (zero padding), float4 over d
__global__ __launch_bounds__(256) void resconv_fwd_kernel(const float* __restrict__ v, const float* __restrict__ w,
                                                          float* __restrict__ out, int B, int Hh, int n, int D, int KW) {
  const int d4n = D >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * Hh * n * d4n;
  if (idx >= total) return;
  const int d4 = (int)(idx % d4n);
  const int t = (int)((idx / d4n) % n);
  const int h = (int)((idx / ((long long)d4n * n)) % Hh);
  const int b = (int)(idx / ((long long)d4n * n * Hh));
  const float* vb = v + (((long long)b * Hh + h) * n) * D + d4 * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int half = KW / 2;
  for (int k = 0; k < KW; ++k) {
    const int tt = t + k - half;
    if (tt < 0 || tt >= n) continue;
    const float wk = w[h * KW + k];
    const float4 x = *reinterpret_cast<const float4*>(vb + (long long)tt * D);
    acc.x = fmaf(wk, x.x, acc.x); acc.y = fmaf(wk, x.y, acc.y); acc.z = fmaf(wk, x.z, acc.z); acc.w = fmaf(wk, x.w, acc.w);
  }
  *reinterpret_cast<float4*>(out + ((long long)b * n + t) * (Hh * D) + h * D + d4 * 4) = acc;
}

// dv[b, h, t, d] = sum_k w[h, k] * dout[b, t - k + KW/2, h*D + d]
__global__ __launch_bounds__(256) void resconv_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                               float* __restrict__ dv, int B, int Hh, int n, int D, int KW) {
  const int d4n = D >> 2;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * Hh * n * d4n;
  if (idx >= total) return;
  const int d4 = (int)(idx % d4n);
  const int t = (int)((idx / d4n) % n);
  const int h = (int)((idx / ((long long)d4n * n)) % Hh);
  const int b = (int)(idx / ((long long)d4n * n * Hh));
  const float* db = dout + ((long long)b * n) * (Hh * D) + h * D + d4 * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int half = KW / 2;
  for (int k = 0; k < KW; ++k) {
    const int tt = t - k + half;
    if (tt < 0 || tt >= n) continue;
    const float wk = w[h * KW + k];
    const float4 x = *reinterpret_cast<const float4*>(db + (long long)tt * (Hh * D));
    acc.x = fmaf(wk, x.x, acc.x); acc.y = fmaf(wk, x.y, acc.y); acc.z = fmaf(wk, x.z, acc.z); acc.w = fmaf(wk, x.w, acc.w);
  }
  *reinterpret_cast<float4*>(dv + (((long long)b * Hh + h) * n + t) * D + d4 * 4) = acc;
}

// dw[h, k] += sum_{b, t, d} dout[b, t, h*D + d] * v[b, h, t + k - KW/2, d]; one block per (h, b, 64 tokens): the dout tile
// [64][D] and the v rows it meets [64 + KW - 1][D] are staged in LDS once, thread (tap = tid / 8, sub = tid % 8) owns one tap
// and every 8th float4 of d (float4 LDS reads; D % 4 == 0, D <= 64, KW <= 64)
constexpr int RCW_T = 64;
__global__ __launch_bounds__(256) void resconv_bwd_weight_kernel(const float* __restrict__ dout, const float* __restrict__ v,
                                                                 float* __restrict__ dw, int B, int Hh, int n, int D, int KW) {
  extern __shared__ __attribute__((aligned(16))) float rcw[];
  const int RCW_LD = D + 4, DQ = D >> 2;
  float* ds = rcw;                              // [RCW_T][RCW_LD]
  float* vs = rcw + RCW_T * RCW_LD;             // [RCW_T + KW - 1][RCW_LD]
  float* red = vs + (RCW_T + KW - 1) * RCW_LD;  // [64]
  const int h = blockIdx.y, b = blockIdx.z, t0 = blockIdx.x * RCW_T;
  const int tid = threadIdx.x, half = KW / 2;
  const float* vb = v + (((long long)b * Hh + h) * n) * D;
  const float* db = dout + ((long long)b * n) * (Hh * D) + h * D;
  if (tid < 64) red[tid] = 0.f;
  for (int i = tid; i < RCW_T * DQ; i += 256) {                 // DQ float4 per row
    const int r = i / DQ, q4 = (i - r * DQ) * 4, t = t0 + r;
    *reinterpret_cast<float4*>(&ds[r * RCW_LD + q4]) =
        (t < n) ? *reinterpret_cast<const float4*>(db + (long long)t * (Hh * D) + q4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int i = tid; i < (RCW_T + KW - 1) * DQ; i += 256) {
    const int r = i / DQ, q4 = (i - r * DQ) * 4, t = t0 + r - half;
    *reinterpret_cast<float4*>(&vs[r * RCW_LD + q4]) =
        (t >= 0 && t < n) ? *reinterpret_cast<const float4*>(vb + (long long)t * D + q4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int sub = tid & 7;
  for (int k = tid >> 3; k < KW; k += 32) {
    float s = 0.f;
#pragma unroll 4
    for (int r = 0; r < RCW_T; ++r) {
      for (int q = sub; q < DQ; q += 8) {
        const float4 a = *reinterpret_cast<const float4*>(&ds[r * RCW_LD + 4 * q]);
        const float4 w = *reinterpret_cast<const float4*>(&vs[(r + k) * RCW_LD + 4 * q]);
        s = fmaf(a.x, w.x, s); s = fmaf(a.y, w.y, s); s = fmaf(a.z, w.z, s); s = fmaf(a.w, w.w, s);
      }
    }
    atomicAdd(&red[k], s);
  }
  __syncthreads();
  if (tid < KW) atomicAdd(&dw[h * KW + tid], red[tid]);
}

// ------------------------------------------------------------------------------------------------
// Strip forms of the three residual-convolution kernels for the reference's 33 taps (residual_conv_kernel = 33,
// NystromAttention.py:62-66).  One thread owns one float4 of d and a strip of RCS_T = 16 consecutive tokens; it walks the
// 48 input rows its strip meets ONCE, and every row feeds all the outputs it belongs to from registers (fully unrolled:
// 16 x 33 float4 FMAs) - 3 row loads per output instead of 33 through the cache.  The kernels are then bound by the one
// read and one write of the tensor.  Lanes of a wave hold 16 consecutive float4 of a row (256 contiguous bytes for D = 64)
// for several strips.
//   MODE 0: out[b, t, h D + d]  = sum_k w[h, k]      v[b, h, t + k - 16, d]      (v head-major in, merged out)
//   MODE 1: dv[b, h, t, d]      = sum_k w[h, 32 - k] dout[b, t + k - 16, h D + d] (merged in, head-major out)
// ------------------------------------------------------------------------------------------------
constexpr int RCS_T = 16, RCS_KW = 33, RCS_HALF = 16;
template <int MODE>
__global__ __launch_bounds__(256) void resconv_strip_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                            float* __restrict__ out, int B, int Hh, int n, int D) {
  __shared__ float ws[64 * RCS_KW];                        // all heads' taps (Hh <= 64), flipped for the data gradient
  for (int i = threadIdx.x; i < Hh * RCS_KW; i += 256) {
    const int h = i / RCS_KW, k = i - h * RCS_KW;
    ws[i] = w[h * RCS_KW + (MODE ? RCS_KW - 1 - k : k)];
  }
  __syncthreads();
  const int d4n = D >> 2, strips = (n + RCS_T - 1) / RCS_T;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * Hh * strips * d4n) return;
  const int d4 = (int)(idx % d4n);
  const int st = (int)((idx / d4n) % strips);
  const int h = (int)((idx / ((long long)d4n * strips)) % Hh);
  const int b = (int)(idx / ((long long)d4n * strips * Hh));
  const int t0 = st * RCS_T;
  const long long hm = (((long long)b * Hh + h) * n) * D + d4 * 4, hs = D;                  // head-major base / row stride
  const long long mm = ((long long)b * n) * ((long long)Hh * D) + h * D + d4 * 4, ms = (long long)Hh * D;   // merged
  const float* ip = in + (MODE ? mm : hm);
  const long long is = MODE ? ms : hs;
  float wk[RCS_KW];
#pragma unroll
  for (int k = 0; k < RCS_KW; ++k) wk[k] = ws[h * RCS_KW + k];
  float4 acc[RCS_T];
#pragma unroll
  for (int i = 0; i < RCS_T; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < RCS_T + RCS_KW - 1; ++j) {
    const int tj = t0 - RCS_HALF + j;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tj >= 0 && tj < n) x = *reinterpret_cast<const float4*>(ip + (long long)tj * is);
#pragma unroll
    for (int i = 0; i < RCS_T; ++i) {
      const int k = j - i;                                  // tap that row j carries into output i
      if (k >= 0 && k < RCS_KW) {
        acc[i].x = fmaf(wk[k], x.x, acc[i].x); acc[i].y = fmaf(wk[k], x.y, acc[i].y);
        acc[i].z = fmaf(wk[k], x.z, acc[i].z); acc[i].w = fmaf(wk[k], x.w, acc[i].w);
      }
    }
  }
  float* op = out + (MODE ? hm : mm);
  const long long os = MODE ? hs : ms;
#pragma unroll
  for (int i = 0; i < RCS_T; ++i)
    if (t0 + i < n) *reinterpret_cast<float4*>(op + (long long)(t0 + i) * os) = acc[i];
}

// dw[h, k] += sum_{b, t, d} dout[b, t, h D + d] v[b, h, t + k - 16, d]: the same walk with the roles exchanged - the strip's 16
// dout rows sit in registers, every v row is read once and meets the taps it realises.  A block is (b, h, 256 RCS_SPT tokens):
// 16 float4 columns x 16 strips side by side, RCS_SPT strips one after the other per thread; the 33 sums are reduced over the
// wave by shuffles, over the block in LDS, and leave as 33 atomics per block (n = 10 240: 20 blocks per (b, h)).
constexpr int RCS_SPT = 2;
__global__ __launch_bounds__(256) void resconv_strip_wgrad_kernel(const float* __restrict__ dout, const float* __restrict__ v,
                                                                  float* __restrict__ dw, int B, int Hh, int n, int D) {
  __shared__ float red[RCS_KW];
  const int tid = threadIdx.x, d4n = D >> 2;
  const int h = blockIdx.y, b = blockIdx.z;
  if (tid < RCS_KW) red[tid] = 0.f;
  __syncthreads();
  const int per = 256 / d4n;                               // strips side by side in the block (d4n divides 256: D in {4, .., 64} powers of two)
  const int d4 = tid % d4n, sl = tid / d4n;
  const float* vb = v + (((long long)b * Hh + h) * n) * D + d4 * 4;
  const float* db = dout + ((long long)b * n) * ((long long)Hh * D) + h * D + d4 * 4;
  const long long ms = (long long)Hh * D;
  float s[RCS_KW];
#pragma unroll
  for (int k = 0; k < RCS_KW; ++k) s[k] = 0.f;
  for (int rep = 0; rep < RCS_SPT; ++rep) {
    const int t0 = ((blockIdx.x * RCS_SPT + rep) * per + sl) * RCS_T;
    if (t0 >= n) break;
    float4 g[RCS_T];
#pragma unroll
    for (int i = 0; i < RCS_T; ++i)
      g[i] = (t0 + i < n) ? *reinterpret_cast<const float4*>(db + (long long)(t0 + i) * ms) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < RCS_T + RCS_KW - 1; ++j) {
      const int tj = t0 - RCS_HALF + j;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tj >= 0 && tj < n) x = *reinterpret_cast<const float4*>(vb + (long long)tj * D);
#pragma unroll
      for (int i = 0; i < RCS_T; ++i) {
        const int k = j - i;
        if (k >= 0 && k < RCS_KW) s[k] = fmaf(g[i].x, x.x, fmaf(g[i].y, x.y, fmaf(g[i].z, x.z, fmaf(g[i].w, x.w, s[k]))));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < RCS_KW; ++k) {
    float t = s[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((tid & 63) == 0) atomicAdd(&red[k], t);
  }
  __syncthreads();
  if (tid < RCS_KW) atomicAdd(&dw[h * RCS_KW + tid], red[tid]);
}

// PPEG: y[b, y, x, c] = bias[c] + sum_{ky,kx} wm[c, ky, kx] * x[b, y+ky-3, x+kx-3, c]  (wm = merged 7x7 incl. identity)
__global__ __launch_bounds__(256) void dw7_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wm,
                                                      const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                      int W, int C, int flip) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * H * W * C;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const int px = (int)((idx / C) % W);
  const int py = (int)((idx / ((long long)C * W)) % H);
  const int b = (int)(idx / ((long long)C * W * H));
  float acc = bias ? bias[c] : 0.f;
  const float* xb = x + (long long)b * H * W * C + c;
#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
    const int yy = py + ky - 3;
    if (yy < 0 || yy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const int xx = px + kx - 3;
      if (xx < 0 || xx >= W) continue;
      const int wi = flip ? (6 - ky) * 7 + (6 - kx) : ky * 7 + kx;
      acc = fmaf(wm[c * 49 + wi], xb[((long long)yy * W + xx) * C], acc);
    }
  }
  y[idx] = acc;
}

// dwm[c, ky, kx] += sum_{b,y,x} dy[b,y,x,c] * x[b, y+ky-3, x+kx-3, c]; db[c] += sum dy
// block = 64 channels x 4 pixel lanes, grid-stride over pixels; 49 + 1 register accumulators per thread
__global__ __launch_bounds__(256) void dw7_bwd_weight_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ dwm, float* __restrict__ db, int B, int H,
                                                             int W, int C) {
  const int c = blockIdx.y * 64 + (threadIdx.x & 63);
  const int pl = threadIdx.x >> 6;
  float acc[49];
#pragma unroll
  for (int i = 0; i < 49; ++i) acc[i] = 0.f;
  float ab = 0.f;
  const long long npix = (long long)B * H * W;
  if (c < C) {
    for (long long p = (long long)blockIdx.x * 4 + pl; p < npix; p += (long long)gridDim.x * 4) {
      const int px = (int)(p % W), py = (int)((p / W) % H);
      const long long b = p / ((long long)W * H);
      const float g = dy[p * C + c];
      ab += g;
      const float* xb = x + b * H * W * C + c;
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        const int yy = py + ky - 3;
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          const int xx = px + kx - 3;
          if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc[ky * 7 + kx] = fmaf(g, xb[((long long)yy * W + xx) * C], acc[ky * 7 + kx]);
        }
      }
    }
  }
  __shared__ float red[64 * 50];
  for (int i = threadIdx.x; i < 64 * 50; i += 256) red[i] = 0.f;
  __syncthreads();
  const int cl = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 49; ++i) atomicAdd(&red[cl * 50 + i], acc[i]);
  atomicAdd(&red[cl * 50 + 49], ab);
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 50; i += 256) {
    const int cc = blockIdx.y * 64 + i / 50, k = i % 50;
    if (cc < C) {
      if (k < 49) atomicAdd(&dwm[cc * 49 + k], red[i]);
      else atomicAdd(&db[cc], red[i]);
    }
  }
}

}  // namespace

extern "C" {

int smml_softmax_fwd_f32(const float* x, float* y, long long rows, int L, void* stream) {
  SMML_REQUIRE(x && y && rows > 0 && L > 0, "smml_softmax_fwd_f32: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (L <= 256) {
    hipLaunchKernelGGL(softmax_fwd_wave_kernel<4>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, y, rows, L);
  } else if (L <= 1024) {
    hipLaunchKernelGGL(softmax_fwd_wave_kernel<16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, y, rows, L);
  } else {
    SMML_REQUIRE(rows <= 2147483647LL, "smml_softmax_fwd_f32: too many rows");
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)rows), dim3(256), 0, st, x, y, L);
  }
  SMML_LAUNCH_CHECK("smml_softmax_fwd_f32");
  return SMML_OK;
}

int smml_softmax_bwd_f32(const float* y, const float* dy, float* dx, long long rows, int L, void* stream) {
  SMML_REQUIRE(y && dy && dx && rows > 0 && L > 0 && rows <= 2147483647LL, "smml_softmax_bwd_f32: bad argument");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, y, dy, dx, L);
  SMML_LAUNCH_CHECK("smml_softmax_bwd_f32");
  return SMML_OK;
}

int smml_tile_rows_f32(const float* src, float* dst, long long nb, int R, int C, float scale, void* stream) {
  SMML_REQUIRE(src && dst && nb > 0 && R > 0 && C > 0, "smml_tile_rows_f32: bad argument");
  const long long total = nb * R * C;
  hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, nb,
                     R, C, scale);
  SMML_LAUNCH_CHECK("smml_tile_rows_f32");
  return SMML_OK;
}

int smml_resconv_fwd_f32(const float* v, const float* w, float* out_merged, int B, int H, int n, int D, int KW,
                         void* stream) {
  SMML_REQUIRE(v && w && out_merged && B > 0 && H > 0 && n > 0 && D > 0 && D % 4 == 0 && KW > 0 && KW % 2 == 1,
               "smml_resconv_fwd_f32: bad argument (D %% 4 == 0, odd kernel)");
  const long long total = (long long)B * H * n * (D / 4);
  if (KW == RCS_KW && H <= 64) {                              // the reference's 33 taps: strip kernel
    const long long threads = (long long)B * H * ((n + RCS_T - 1) / RCS_T) * (D / 4);
    hipLaunchKernelGGL(resconv_strip_kernel<0>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, w,
                       out_merged, B, H, n, D);
  } else {
    hipLaunchKernelGGL(resconv_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v, w,
                       out_merged, B, H, n, D, KW);
  }
  SMML_LAUNCH_CHECK("smml_resconv_fwd_f32");
  return SMML_OK;
}

// dv overwritten; dw accumulated into
int smml_resconv_bwd_f32(const float* dout_merged, const float* v, const float* w, float* dv, float* dw, int B, int H,
                         int n, int D, int KW, void* stream) {
  SMML_REQUIRE(dout_merged && v && w && dv && dw && D % 4 == 0 && KW % 2 == 1 && KW <= 64,
               "smml_resconv_bwd_f32: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)B * H * n * (D / 4);
  const bool strip = KW == RCS_KW && H <= 64 && (D == 64 || D == 32 || D == 16 || D == 8 || D == 4) && B <= 65535 && H <= 65535;
  if (strip) {
    const long long threads = (long long)B * H * ((n + RCS_T - 1) / RCS_T) * (D / 4);
    hipLaunchKernelGGL(resconv_strip_kernel<1>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, dout_merged, w, dv, B,
                       H, n, D);
    SMML_LAUNCH_CHECK("smml_resconv_bwd_f32/data");
    const int tokens_per_block = RCS_SPT * (256 / (D / 4)) * RCS_T;
    hipLaunchKernelGGL(resconv_strip_wgrad_kernel, dim3((n + tokens_per_block - 1) / tokens_per_block, H, B), dim3(256), 0, st,
                       dout_merged, v, dw, B, H, n, D);
    SMML_LAUNCH_CHECK("smml_resconv_bwd_f32/weight");
    return SMML_OK;
  }
  hipLaunchKernelGGL(resconv_bwd_data_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dout_merged, w, dv, B,
                     H, n, D, KW);
  SMML_LAUNCH_CHECK("smml_resconv_bwd_f32/data");
  SMML_REQUIRE((D % 4) == 0 && D <= 64 && KW <= 64,
               "smml_resconv_bwd_f32: the weight-gradient kernel needs dim_head % 4 == 0, dim_head <= 64 and <= 64 taps");
  const size_t lds = ((size_t)(2 * RCW_T + KW - 1) * (D + 4) + 64) * sizeof(float);
  hipLaunchKernelGGL(resconv_bwd_weight_kernel, dim3((n + RCW_T - 1) / RCW_T, H, B), dim3(256), lds, st, dout_merged, v, dw,
                     B, H, n, D, KW);
  SMML_LAUNCH_CHECK("smml_resconv_bwd_f32/weight");
  return SMML_OK;
}

// merged depthwise 7x7 on channel-last maps; flip = 1 applies the 180-degree rotated kernel (data gradient)
int smml_dwconv7_fwd_f32(const float* x, const float* wm, const float* bias, float* y, int B, int H, int W, int C, int flip,
                         void* stream) {
  SMML_REQUIRE(x && wm && y && B > 0 && H > 0 && W > 0 && C > 0, "smml_dwconv7_fwd_f32: bad argument");
  const long long total = (long long)B * H * W * C;
  hipLaunchKernelGGL(dw7_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, wm, bias, y,
                     B, H, W, C, flip);
  SMML_LAUNCH_CHECK("smml_dwconv7_fwd_f32");
  return SMML_OK;
}

// dwm [C, 49] and db [C] accumulated into
int smml_dwconv7_bwd_weight_f32(const float* x, const float* dy, float* dwm, float* db, int B, int H, int W, int C,
                                void* stream) {
  SMML_REQUIRE(x && dy && dwm && db && B > 0 && H > 0 && W > 0 && C > 0, "smml_dwconv7_bwd_weight_f32: bad argument");
  const long long npix = (long long)B * H * W;
  const int gx = (int)((npix + 3) / 4 < 256 ? (npix + 3) / 4 : 256);
  hipLaunchKernelGGL(dw7_bwd_weight_kernel, dim3(gx, (C + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, dy, dwm, db, B, H,
                     W, C);
  SMML_LAUNCH_CHECK("smml_dwconv7_bwd_weight_f32");
  return SMML_OK;
}

}  // extern "C"
