// Offset network + deformable sampling for gfx950 (HBM / gather bound, no matrix-core work):
//   offsets  : depthwise strided conv -> GELU(erf) -> 1x1 (dg -> PD, no bias) -> tanh -> * offset_scale,
//              vgrid = meshgrid + offsets, vs = 2 vgrid / max(t - 1, 1) - 1
//              (models/DeformableAttention2D.py:207-213,255-266; DeformableAttention1D.py:139-146,176-183)
//   sampling : F.grid_sample(bilinear, zeros, align_corners=False) of the grouped path stream at vs
//              (DeformableAttention2D.py:268-274; DeformableAttention1D.py:36-43,185-190 incl. its
//              degenerate-axis behaviour: the 1-D module samples a [H = n, W = 1] map at (x = vs, y = 0))
// Layouts: q [B, Hh, Ww, G*dg] and x [B, Hh, Ww, G*cg] token-major (channel-last): a wave reads one
// pixel's channels as one contiguous row; vgrid [(B G), PD, th, tw] as the reference returns it;
// vs [(B G), J, PD]; kv [B, J, G*cg].
// The integer path (pixel coordinate, floor, corner indices, in-bounds masks) is evaluated with one
// rounding per operation (__f*_rn, no FMA contraction) so that it is bit-exact against the oracle.
#include "smml_common.h"

namespace {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
// normalised coordinate -> pixel coordinate, align_corners = False: ((v + 1) * size - 1) / 2
__device__ __forceinline__ float unnormalize(float v, int size) {
  return __fdiv_rn(__fsub_rn(__fmul_rn(__fadd_rn(v, 1.f), (float)size), 1.f), 2.f);
}
__device__ __forceinline__ float normalize_pos(float vg, int den) {
  return __fsub_rn(__fdiv_rn(__fmul_rn(2.f, vg), (float)den), 1.f);
}

struct Corners {
  int x0, y0;
  float fx, fy;       // fractional parts (weights of the +1 corners)
  bool mx0, mx1, my0, my1;
};
__device__ __forceinline__ Corners corners_of(float vx, float vy, int W, int H) {
  Corners k;
  const float ix = unnormalize(vx, W), iy = unnormalize(vy, H);
  const float x0f = floorf(ix), y0f = floorf(iy);
  k.x0 = (int)x0f; k.y0 = (int)y0f;
  k.fx = __fsub_rn(ix, x0f); k.fy = __fsub_rn(iy, y0f);
  k.mx0 = k.x0 >= 0 && k.x0 < W; k.mx1 = (k.x0 + 1) >= 0 && (k.x0 + 1) < W;
  k.my0 = k.y0 >= 0 && k.y0 < H; k.my1 = (k.y0 + 1) >= 0 && (k.y0 + 1) < H;
  return k;
}

// ---------------------------------------------------------------------------------------------
// offsets forward: one wave per output point (bg, ty, tx); lane = channel (CPL channels per lane)
// ---------------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(256) void offsets_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ w0, const float* __restrict__ b0,
    const float* __restrict__ w2, float* __restrict__ vgrid, float* __restrict__ vs, int B, int Hh, int Ww, int G,
    int kh, int kw, int rh, int rw, int ph, int pw, int th, int tw, int PD, float offset_scale) {
  // depthwise weights transposed to [tap][channel] in LDS: the per-tap read of a wave is then one contiguous row instead
  // of a cache line per lane
  extern __shared__ float w0s[];
  const int dg = 64 * CPL, inner = G * dg, KK = kh * kw;
  for (int i = threadIdx.x; i < dg * KK; i += blockDim.x) {
    const int ch = i / KK, t = i - ch * KK;
    w0s[t * dg + ch] = w0[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int J = th * tw, npts = B * G * J;
  if (wid >= npts) return;
  const int bg = wid / J, j = wid - bg * J, b = bg / G, g = bg - b * G;
  const int ty = j / tw, tx = j - ty * tw;
  float o0 = 0.f, o1 = 0.f;
#pragma unroll
  for (int u = 0; u < CPL; ++u) {
    const int ch = lane + 64 * u;
    float acc = b0[ch];
    for (int ky = 0; ky < kh; ++ky) {
      const int iy = ty * rh - ph + ky;
      if (iy < 0 || iy >= Hh) continue;
      for (int kx = 0; kx < kw; ++kx) {
        const int ix = tx * rw - pw + kx;
        if (ix < 0 || ix >= Ww) continue;
        acc = fmaf(w0s[(ky * kw + kx) * dg + ch], q[(((size_t)b * Hh + iy) * Ww + ix) * inner + g * dg + ch], acc);
      }
    }
    const float ge = gelu_erf(acc);
    o0 = fmaf(ge, w2[ch], o0);
    if (PD == 2) o1 = fmaf(ge, w2[dg + ch], o1);
  }
  o0 = wave_sum(o0);
  if (PD == 2) o1 = wave_sum(o1);
  if (lane == 0) {
    if (PD == 2) {
      const float vg0 = __fadd_rn((float)tx, tanhf(o0) * offset_scale);
      const float vg1 = __fadd_rn((float)ty, tanhf(o1) * offset_scale);
      vgrid[((size_t)bg * 2 + 0) * J + j] = vg0;
      vgrid[((size_t)bg * 2 + 1) * J + j] = vg1;
      // normalize_grid quirk (DeformableAttention2D.py:100-108): channel 0 / (rows - 1), channel 1 / (cols - 1)
      vs[((size_t)bg * J + j) * 2 + 0] = normalize_pos(vg0, max(th - 1, 1));
      vs[((size_t)bg * J + j) * 2 + 1] = normalize_pos(vg1, max(tw - 1, 1));
    } else {
      const float vg0 = __fadd_rn((float)tx, tanhf(o0) * offset_scale);
      vgrid[(size_t)bg * J + j] = vg0;
      vs[(size_t)bg * J + j] = normalize_pos(vg0, max(tw - 1, 1));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// offsets backward: d vgrid (direct) + d vs  ->  dq, dw0, db0, dw2, in three deterministic passes (no global atomics):
//   1. point pass: each wave walks sampled points with a grid stride, recomputes the forward, stores the gradient
//      dy[point][channel] of the depthwise conv's output and keeps its weight-gradient partials in registers; the
//      workgroup's partials are summed in LDS and written to a slab [workgroup][nred]
//   2. gather pass: every q element sums dy . w0 over the (at most ceil(k / r)^2 = 4) windows that cover it
//      (the scatter form needed 2304 float atomics per point: 92 M per step at B = 8, L2-atomic bound)
//   3. slab reduction in a fixed order
// ---------------------------------------------------------------------------------------------
template <int CPL, int KH, int KW>
__global__ __launch_bounds__(256) void offsets_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ w0, const float* __restrict__ b0,
    const float* __restrict__ w2, const float* __restrict__ dvgrid, const float* __restrict__ dvs,
    float* __restrict__ dyb, float* __restrict__ slab, int B, int Hh,
    int Ww, int G, int rh, int rw, int ph, int pw, int th, int tw, int PD, float offset_scale) {
  const int lane = threadIdx.x & 63;
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  const int J = th * tw, npts = B * G * J;
  const int dg = 64 * CPL, inner = G * dg;
  float aw0[CPL][KH * KW], ab0[CPL], aw2[CPL][2];
#pragma unroll
  for (int u = 0; u < CPL; ++u) {
    ab0[u] = 0.f; aw2[u][0] = 0.f; aw2[u][1] = 0.f;
#pragma unroll
    for (int t = 0; t < KH * KW; ++t) aw0[u][t] = 0.f;
  }
  for (int wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); wid < npts; wid += nwaves) {
    const int bg = wid / J, j = wid - bg * J, b = bg / G, g = bg - b * G;
    const int ty = j / tw, tx = j - ty * tw;
    // recompute the forward
    float y[CPL], ge[CPL];
    float qv[CPL][KH * KW];                  // the window's taps: read once, used by the recompute and by the weight gradient
    float o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
      const int ch = lane + 64 * u;
      float acc = b0[ch];
#pragma unroll
      for (int ky = 0; ky < KH; ++ky) {
        const int iy = ty * rh - ph + ky;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
          const int ix = tx * rw - pw + kx;
          const bool in = iy >= 0 && iy < Hh && ix >= 0 && ix < Ww;
          qv[u][ky * KW + kx] = in ? q[(((size_t)b * Hh + iy) * Ww + ix) * inner + g * dg + ch] : 0.f;
          acc = fmaf(w0[(ch * KH + ky) * KW + kx], qv[u][ky * KW + kx], acc);
        }
      }
      y[u] = acc;
      ge[u] = gelu_erf(acc);
      o0 = fmaf(ge[u], w2[ch], o0);
      if (PD == 2) o1 = fmaf(ge[u], w2[dg + ch], o1);
    }
    o0 = wave_sum(o0);
    if (PD == 2) o1 = wave_sum(o1);
    // upstream gradient of vgrid = direct part + 2 / max(t - 1, 1) * d vs
    float g0, g1 = 0.f;
    if (PD == 2) {
      g0 = (dvgrid ? dvgrid[((size_t)bg * 2 + 0) * J + j] : 0.f) +
           (dvs ? dvs[((size_t)bg * J + j) * 2 + 0] * (2.f / (float)max(th - 1, 1)) : 0.f);
      g1 = (dvgrid ? dvgrid[((size_t)bg * 2 + 1) * J + j] : 0.f) +
           (dvs ? dvs[((size_t)bg * J + j) * 2 + 1] * (2.f / (float)max(tw - 1, 1)) : 0.f);
    } else {
      g0 = (dvgrid ? dvgrid[(size_t)bg * J + j] : 0.f) +
           (dvs ? dvs[(size_t)bg * J + j] * (2.f / (float)max(tw - 1, 1)) : 0.f);
    }
    const float t0 = tanhf(o0), t1 = tanhf(o1);
    const float ds0 = g0 * offset_scale * (1.f - t0 * t0);
    const float ds1 = (PD == 2) ? g1 * offset_scale * (1.f - t1 * t1) : 0.f;
#pragma unroll
    for (int u = 0; u < CPL; ++u) {
      const int ch = lane + 64 * u;
      aw2[u][0] = fmaf(ds0, ge[u], aw2[u][0]);
      float dge = ds0 * w2[ch];
      if (PD == 2) {
        aw2[u][1] = fmaf(ds1, ge[u], aw2[u][1]);
        dge = fmaf(ds1, w2[dg + ch], dge);
      }
      const float dy = dge * gelu_erf_grad(y[u]);
      dyb[(size_t)wid * dg + ch] = dy;
      ab0[u] += dy;
#pragma unroll
      for (int t = 0; t < KH * KW; ++t) aw0[u][t] = fmaf(dy, qv[u][t], aw0[u][t]);   // taps outside the map are 0
    }
  }
  // weight-gradient partials: sum the workgroup's waves in LDS, then one contiguous slab row per workgroup
  __shared__ float red[128 * (KH * KW + 3)];
  constexpr int KK = KH * KW;
  const int nred = dg * (KK + 1 + 2);
  for (int i = threadIdx.x; i < nred; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < CPL; ++u) {
    const int ch = lane + 64 * u;
#pragma unroll
    for (int t = 0; t < KK; ++t) atomicAdd(&red[ch * KK + t], aw0[u][t]);
    atomicAdd(&red[dg * KK + ch], ab0[u]);
    atomicAdd(&red[dg * (KK + 1) + ch], aw2[u][0]);
    if (PD == 2) atomicAdd(&red[dg * (KK + 2) + ch], aw2[u][1]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nred; i += blockDim.x) slab[(size_t)blockIdx.x * nred + i] = red[i];
}

// dq[b, y, x, g*dg + ch] = sum over the windows (ty, tx) covering (y, x) of dy[(b, g, ty, tx)][ch] * w0[ch][ky][kx];
// one block row per (b, y), a thread per (x, 4 channels): float4 loads of dy, float4 store of dq
// ACC: dq += (the buffer already holds the gradient q received from the attention core: one pass instead of this kernel's store + an
// elementwise add of two 164 MB tensors per 8-bag step)
template <bool ACC>
__global__ __launch_bounds__(256) void offsets_bwd_gather_kernel(const float* __restrict__ dyb, const float* __restrict__ w0,
                                                                 float* __restrict__ dq, int Hh, int Ww, int G, int dg, int KH,
                                                                 int KW, int rh, int rw, int ph, int pw, int th, int tw) {
  extern __shared__ __attribute__((aligned(16))) float wT[];   // w0 transposed to [tap][channel]: one float4 per window
  const int KKs = KH * KW;
  for (int i = threadIdx.x; i < dg * KKs; i += blockDim.x) {
    const int chs = i / KKs, tap = i - chs * KKs;
    wT[tap * dg + chs] = w0[i];
  }
  __syncthreads();
  const int inner = G * dg, iq = inner >> 2;
  const int y = blockIdx.y % Hh, b = blockIdx.y / Hh;
  const int ty1 = min(th - 1, (y + ph) / rh), ty0 = max(0, (y + ph - (KH - 1) + rh - 1) / rh);
  // a block walks a slice of one (b, y) row of the map: the transposed weights are staged once per slice
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < Ww * iq; t += gridDim.x * blockDim.x) {
    const int x = t / iq, c4 = (t - x * iq) * 4, g = c4 / dg, ch = c4 - g * dg;
    const int tx1 = min(tw - 1, (x + pw) / rw), tx0 = max(0, (x + pw - (KW - 1) + rw - 1) / rw);
    float4* dst = reinterpret_cast<float4*>(dq + (((size_t)b * Hh + y) * Ww + x) * inner + c4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ACC) acc = *dst;
    for (int ty = ty0; ty <= ty1; ++ty) {
      const int ky = y + ph - ty * rh;
      for (int tx = tx0; tx <= tx1; ++tx) {
        const int kx = x + pw - tx * rw;
        const float4 d = *reinterpret_cast<const float4*>(dyb + ((size_t)(b * G + g) * th * tw + ty * tw + tx) * dg + ch);
        const float4 w = *reinterpret_cast<const float4*>(wT + (ky * KW + kx) * dg + ch);
        acc.x = fmaf(d.x, w.x, acc.x); acc.y = fmaf(d.y, w.y, acc.y);
        acc.z = fmaf(d.z, w.z, acc.z); acc.w = fmaf(d.w, w.w, acc.w);
      }
    }
    *dst = acc;
  }
}

// column sums of a slab [nrows][nred] in two fixed-order stages: part[p][i] = sum of rows p*chunk .. ; then the final sum
// scattered into dw0 | db0 | dw2
constexpr int OFF_RED_PARTS = 32;
__global__ void offsets_bwd_partial_kernel(const float* __restrict__ slab, int nrows, int nred, int chunk,
                                           float* __restrict__ part) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nred) return;
  const int r0 = blockIdx.y * chunk, r1 = min(nrows, r0 + chunk);
  float v = 0.f;
  for (int k = r0; k < r1; ++k) v += slab[(size_t)k * nred + i];
  part[(size_t)blockIdx.y * nred + i] = v;
}
__global__ void offsets_bwd_reduce_kernel(const float* __restrict__ part, int nparts, int nred, int dg, int KK, int PD,
                                          float* __restrict__ dw0, float* __restrict__ db0, float* __restrict__ dw2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nred) return;
  float v = 0.f;
  for (int k = 0; k < nparts; ++k) v += part[(size_t)k * nred + i];
  if (i < dg * KK) dw0[i] = v;
  else if (i < dg * (KK + 1)) db0[i - dg * KK] = v;
  else if (i < dg * (KK + 1 + PD)) dw2[i - dg * (KK + 1)] = v;
}

// ---------------------------------------------------------------------------------------------
// bilinear sampling forward: thread per (b, j, channel); cg consecutive lanes share a sample point
// ---------------------------------------------------------------------------------------------
__global__ void sample_fwd_kernel(const float* __restrict__ x, const float* __restrict__ vs, float* __restrict__ kv,
                                  int B, int Hh, int Ww, int G, int cg, int J, int PD) {
  const int C = G * cg;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * J * C) return;
  const int ch = (int)(idx % C);
  const int j = (int)((idx / C) % J);
  const int b = (int)(idx / ((size_t)C * J));
  const int g = ch / cg;
  const float* p = vs + ((size_t)(b * G + g) * J + j) * PD;
  const float vx = p[0], vy = (PD == 2) ? p[1] : 0.f;
  const Corners k = corners_of(vx, vy, Ww, Hh);
  const float* xb = x + (size_t)b * Hh * Ww * C + ch;
  const float v00 = (k.mx0 && k.my0) ? xb[((size_t)k.y0 * Ww + k.x0) * C] : 0.f;
  const float v10 = (k.mx1 && k.my0) ? xb[((size_t)k.y0 * Ww + k.x0 + 1) * C] : 0.f;
  const float v01 = (k.mx0 && k.my1) ? xb[((size_t)(k.y0 + 1) * Ww + k.x0) * C] : 0.f;
  const float v11 = (k.mx1 && k.my1) ? xb[((size_t)(k.y0 + 1) * Ww + k.x0 + 1) * C] : 0.f;
  const float wx1 = k.fx, wx0 = 1.f - k.fx, wy1 = k.fy, wy0 = 1.f - k.fy;
  kv[idx] = v00 * (wx0 * wy0) + v10 * (wx1 * wy0) + v01 * (wx0 * wy1) + v11 * (wx1 * wy1);
}

// backward: dx (atomic scatter-add into a zeroed / accumulating buffer), dvs += (unique owner per point)
template <int CG>
__global__ void sample_bwd_kernel(const float* __restrict__ x, const float* __restrict__ vs,
                                  const float* __restrict__ dkv, float* __restrict__ dx, float* __restrict__ dvs,
                                  int B, int Hh, int Ww, int G, int J, int PD) {
  const int C = G * CG;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = idx < (size_t)B * J * C;
  const size_t id = live ? idx : 0;
  const int ch = (int)(id % C);
  const int j = (int)((id / C) % J);
  const int b = (int)(id / ((size_t)C * J));
  const int g = ch / CG;
  const float* p = vs + ((size_t)(b * G + g) * J + j) * PD;
  const float vx = p[0], vy = (PD == 2) ? p[1] : 0.f;
  const Corners k = corners_of(vx, vy, Ww, Hh);
  const size_t base = (size_t)b * Hh * Ww * C + ch;
  const size_t i00 = base + ((size_t)k.y0 * Ww + k.x0) * C, i10 = i00 + C;
  const size_t i01 = i00 + (size_t)Ww * C, i11 = i01 + C;
  const bool m00 = k.mx0 && k.my0, m10 = k.mx1 && k.my0, m01 = k.mx0 && k.my1, m11 = k.mx1 && k.my1;
  const float go = live ? dkv[id] : 0.f;
  const float wx1 = k.fx, wx0 = 1.f - k.fx, wy1 = k.fy, wy0 = 1.f - k.fy;
  const float v00 = m00 ? x[i00] : 0.f, v10 = m10 ? x[i10] : 0.f, v01 = m01 ? x[i01] : 0.f, v11 = m11 ? x[i11] : 0.f;
  if (live) {
    if (m00) atomicAdd(&dx[i00], go * (wx0 * wy0));
    if (m10) atomicAdd(&dx[i10], go * (wx1 * wy0));
    if (m01) atomicAdd(&dx[i01], go * (wx0 * wy1));
    if (m11) atomicAdd(&dx[i11], go * (wx1 * wy1));
  }
  // d/d(ix), d/d(iy) of the interpolant, then chain through ix = ((v + 1) W - 1) / 2
  float gx = go * ((v10 - v00) * wy0 + (v11 - v01) * wy1) * (0.5f * (float)Ww);
  float gy = go * ((v01 - v00) * wx0 + (v11 - v10) * wx1) * (0.5f * (float)Hh);
#pragma unroll
  for (int o = CG / 2; o > 0; o >>= 1) {
    gx += __shfl_xor(gx, o);
    gy += __shfl_xor(gy, o);
  }
  if (live && (ch % CG) == 0) {
    float* d = dvs + ((size_t)(b * G + g) * J + j) * PD;
    d[0] += gx;
    if (PD == 2) d[1] += gy;
  }
}

__global__ void corners_kernel(const float* __restrict__ vs, int* __restrict__ cx, int* __restrict__ cy,
                               unsigned char* __restrict__ cm, int n, int Hh, int Ww, int PD) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float vx = vs[(size_t)i * PD], vy = (PD == 2) ? vs[(size_t)i * PD + 1] : 0.f;
  const Corners k = corners_of(vx, vy, Ww, Hh);
  cx[4 * i + 0] = k.x0; cx[4 * i + 1] = k.x0 + 1; cx[4 * i + 2] = k.x0; cx[4 * i + 3] = k.x0 + 1;
  cy[4 * i + 0] = k.y0; cy[4 * i + 1] = k.y0; cy[4 * i + 2] = k.y0 + 1; cy[4 * i + 3] = k.y0 + 1;
  cm[4 * i + 0] = k.mx0 && k.my0; cm[4 * i + 1] = k.mx1 && k.my0;
  cm[4 * i + 2] = k.mx0 && k.my1; cm[4 * i + 3] = k.mx1 && k.my1;
}

}  // namespace

extern "C" {

static int offsets_geometry(const char* fn, int Hh, int Ww, int dg, int ks, int r, int posdim, int* kh, int* rh,
                            int* ph, int* th, int* tw) {
  SMML_REQUIRE(dg == 64 || dg == 128, "%s: channels per offset group must be 64 or 128 (got %d)", fn, dg);
  SMML_REQUIRE(posdim == 1 || posdim == 2, "%s: posdim must be 1 or 2", fn);
  SMML_REQUIRE(posdim == 2 || Hh == 1, "%s: the 1-D module expects Hh == 1", fn);
  SMML_REQUIRE(ks >= r && (ks - r) % 2 == 0, "%s: kernel %d / stride %d unsupported", fn, ks, r);
  const int pad = (ks - r) / 2;
  *tw = (Ww + 2 * pad - ks) / r + 1;
  if (posdim == 2) { *kh = ks; *rh = r; *ph = pad; *th = (Hh + 2 * pad - ks) / r + 1; }
  else { *kh = 1; *rh = 1; *ph = 0; *th = 1; }
  SMML_REQUIRE(*th > 0 && *tw > 0, "%s: token grid too small for the offset conv", fn);
  return SMML_OK;
}

// out-length of the strided offset conv along one axis of size s (0 if it does not fit)
int smml_offsets_out_len(int s, int ks, int r) {
  if (s <= 0 || ks <= 0 || r <= 0 || ks < r) return 0;
  const int pad = (ks - r) / 2;
  const long long span = (long long)s + 2 * pad - ks;
  return span < 0 ? 0 : (int)(span / r + 1);
}

int smml_offsets_fwd_f32(const float* q, const float* w0, const float* b0, const float* w2, float* vgrid, float* vs,
                         int B, int Hh, int Ww, int G, int dg, int ks, int r, int posdim, float offset_scale,
                         void* stream) {
  SMML_REQUIRE(q && w0 && b0 && w2 && vgrid && vs, "smml_offsets_fwd_f32: null pointer");
  int kh, rh, ph, th, tw;
  int rc = offsets_geometry("smml_offsets_fwd_f32", Hh, Ww, dg, ks, r, posdim, &kh, &rh, &ph, &th, &tw);
  if (rc) return rc;
  const int pw = (ks - r) / 2;
  const int npts = B * G * th * tw;
  dim3 grid((npts + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)dg * kh * ks * sizeof(float);
  if (dg == 64)
    hipLaunchKernelGGL(offsets_fwd_kernel<1>, grid, block, lds, st, q, w0, b0, w2, vgrid, vs, B, Hh, Ww, G, kh, ks, rh,
                       r, ph, pw, th, tw, posdim, offset_scale);
  else
    hipLaunchKernelGGL(offsets_fwd_kernel<2>, grid, block, lds, st, q, w0, b0, w2, vgrid, vs, B, Hh, Ww, G, kh, ks, rh,
                       r, ph, pw, th, tw, posdim, offset_scale);
  SMML_LAUNCH_CHECK("smml_offsets_fwd_f32");
  return SMML_OK;
}

static int offsets_bwd_blocks(int npts) { return min((npts + 3) / 4, 2048); }

// scratch of the backward: dy [points][dg] + weight-gradient slabs [workgroups][dg * (kh*ks + 3)]
size_t smml_offsets_bwd_workspace_bytes(int B, int Hh, int Ww, int G, int dg, int ks, int r, int posdim) {
  int kh, rh, ph, th, tw;
  if (offsets_geometry("smml_offsets_bwd_workspace_bytes", Hh, Ww, dg, ks, r, posdim, &kh, &rh, &ph, &th, &tw)) return 0;
  const size_t npts = (size_t)B * G * th * tw;
  return (npts * dg + (size_t)(offsets_bwd_blocks((int)npts) + OFF_RED_PARTS) * dg * (kh * ks + 3)) * sizeof(float);
}

// dq, dw0, db0, dw2 are overwritten (no atomics, run-to-run identical).
int smml_offsets_bwd_f32(const float* q, const float* w0, const float* b0, const float* w2, const float* dvgrid,
                         const float* dvs, float* dq, float* dw0, float* db0, float* dw2, void* workspace,
                         size_t workspace_bytes, int B, int Hh, int Ww, int G, int dg, int ks, int r, int posdim,
                         float offset_scale, int accumulate_dq, void* stream) {
  SMML_REQUIRE(q && w0 && b0 && w2 && dq && dw0 && db0 && dw2 && workspace, "smml_offsets_bwd_f32: null pointer");
  SMML_REQUIRE(dvgrid || dvs, "smml_offsets_bwd_f32: no upstream gradient");
  SMML_REQUIRE(ks == 6, "smml_offsets_bwd_f32: only offset_kernel_size = 6 is instantiated (got %d)", ks);
  int kh, rh, ph, th, tw;
  int rc = offsets_geometry("smml_offsets_bwd_f32", Hh, Ww, dg, ks, r, posdim, &kh, &rh, &ph, &th, &tw);
  if (rc) return rc;
  SMML_REQUIRE(workspace_bytes >= smml_offsets_bwd_workspace_bytes(B, Hh, Ww, G, dg, ks, r, posdim),
               "smml_offsets_bwd_f32: workspace too small");
  const int pw = (ks - r) / 2;
  const int npts = B * G * th * tw;
  const int nblk = offsets_bwd_blocks(npts);
  const int nred = dg * (kh * ks + 3);
  float* dyb = (float*)workspace;
  float* slab = dyb + (size_t)npts * dg;
  dim3 grid(nblk), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (posdim == 2 && dg == 64)
    hipLaunchKernelGGL((offsets_bwd_kernel<1, 6, 6>), grid, block, 0, st, q, w0, b0, w2, dvgrid, dvs, dyb, slab,
                       B, Hh, Ww, G, rh, r, ph, pw, th, tw, posdim, offset_scale);
  else if (posdim == 2 && dg == 128)
    hipLaunchKernelGGL((offsets_bwd_kernel<2, 6, 6>), grid, block, 0, st, q, w0, b0, w2, dvgrid, dvs, dyb, slab,
                       B, Hh, Ww, G, rh, r, ph, pw, th, tw, posdim, offset_scale);
  else if (posdim == 1 && dg == 64)
    hipLaunchKernelGGL((offsets_bwd_kernel<1, 1, 6>), grid, block, 0, st, q, w0, b0, w2, dvgrid, dvs, dyb, slab,
                       B, Hh, Ww, G, rh, r, ph, pw, th, tw, posdim, offset_scale);
  else
    hipLaunchKernelGGL((offsets_bwd_kernel<2, 1, 6>), grid, block, 0, st, q, w0, b0, w2, dvgrid, dvs, dyb, slab,
                       B, Hh, Ww, G, rh, r, ph, pw, th, tw, posdim, offset_scale);
  SMML_LAUNCH_CHECK("smml_offsets_bwd_f32/points");
  SMML_REQUIRE((dg % 4) == 0, "smml_offsets_bwd_f32: channels per group must be a multiple of 4");
  // grid.x: slices of a (b, y) row; few enough that the weight staging is amortised, enough rows x slices to fill the chip
  const int row_threads = Ww * (G * dg / 4);
  const int slices = max(1, min((row_threads + 255) / 256, (4096 + B * Hh - 1) / (B * Hh)));
  if (accumulate_dq)   // dq already holds the gradient q received from its other consumer (the fused attention core): add, do not overwrite
    hipLaunchKernelGGL(offsets_bwd_gather_kernel<true>, dim3(slices, B * Hh), block, (size_t)dg * kh * ks * sizeof(float), st, dyb, w0, dq, Hh,
                       Ww, G, dg, kh, ks, rh, r, ph, pw, th, tw);
  else
    hipLaunchKernelGGL(offsets_bwd_gather_kernel<false>, dim3(slices, B * Hh), block, (size_t)dg * kh * ks * sizeof(float), st, dyb, w0, dq, Hh,
                       Ww, G, dg, kh, ks, rh, r, ph, pw, th, tw);
  SMML_LAUNCH_CHECK("smml_offsets_bwd_f32/gather");
  float* part = slab + (size_t)nblk * nred;
  const int chunk = (nblk + OFF_RED_PARTS - 1) / OFF_RED_PARTS;
  hipLaunchKernelGGL(offsets_bwd_partial_kernel, dim3((nred + 255) / 256, OFF_RED_PARTS), block, 0, st, slab, nblk, nred, chunk,
                     part);
  hipLaunchKernelGGL(offsets_bwd_reduce_kernel, dim3((nred + 255) / 256), block, 0, st, part, OFF_RED_PARTS, nred, dg, kh * ks,
                     posdim, dw0, db0, dw2);
  SMML_LAUNCH_CHECK("smml_offsets_bwd_f32/reduce");
  return SMML_OK;
}

int smml_bilinear_sample_fwd_f32(const float* x, const float* vs, float* kv, int B, int Hh, int Ww, int G, int cg,
                                 int J, int posdim, void* stream) {
  SMML_REQUIRE(x && vs && kv, "smml_bilinear_sample_fwd_f32: null pointer");
  SMML_REQUIRE(posdim == 1 || posdim == 2, "smml_bilinear_sample_fwd_f32: posdim must be 1 or 2");
  SMML_REQUIRE(B > 0 && Hh > 0 && Ww > 0 && G > 0 && cg > 0 && J > 0, "smml_bilinear_sample_fwd_f32: bad size");
  const size_t n = (size_t)B * J * G * cg;
  hipLaunchKernelGGL(sample_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, vs, kv,
                     B, Hh, Ww, G, cg, J, posdim);
  SMML_LAUNCH_CHECK("smml_bilinear_sample_fwd_f32");
  return SMML_OK;
}

// dx is accumulated into (atomics); dvs is accumulated into (+=, one owner per sample point).
int smml_bilinear_sample_bwd_f32(const float* x, const float* vs, const float* dkv, float* dx, float* dvs, int B,
                                 int Hh, int Ww, int G, int cg, int J, int posdim, void* stream) {
  SMML_REQUIRE(x && vs && dkv && dx && dvs, "smml_bilinear_sample_bwd_f32: null pointer");
  SMML_REQUIRE(posdim == 1 || posdim == 2, "smml_bilinear_sample_bwd_f32: posdim must be 1 or 2");
  SMML_REQUIRE(cg == 16 || cg == 32, "smml_bilinear_sample_bwd_f32: channels per group must be 16 or 32 (got %d)", cg);
  const size_t n = (size_t)B * J * G * cg;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (cg == 16)
    hipLaunchKernelGGL(sample_bwd_kernel<16>, grid, block, 0, (hipStream_t)stream, x, vs, dkv, dx, dvs, B, Hh, Ww, G, J,
                       posdim);
  else
    hipLaunchKernelGGL(sample_bwd_kernel<32>, grid, block, 0, (hipStream_t)stream, x, vs, dkv, dx, dvs, B, Hh, Ww, G, J,
                       posdim);
  SMML_LAUNCH_CHECK("smml_bilinear_sample_bwd_f32");
  return SMML_OK;
}

// integer path only: corner indices [n, 4] (x, y) and in-bounds masks [n, 4] for n sample points
int smml_bilinear_corners_f32(const float* vs, int* cx, int* cy, unsigned char* cm, int n, int Hh, int Ww, int posdim,
                              void* stream) {
  SMML_REQUIRE(vs && cx && cy && cm && n > 0, "smml_bilinear_corners_f32: bad argument");
  hipLaunchKernelGGL(corners_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, vs, cx, cy, cm, n, Hh, Ww,
                     posdim);
  SMML_LAUNCH_CHECK("smml_bilinear_corners_f32");
  return SMML_OK;
}

}  // extern "C"
