// Row-wise / column-wise HBM-bound pieces of the path for gfx950 (wavefront reductions, no matrix cores):
//   LayerNorm forward / backward          (models/DeformCrossTransMIL.py:44,71,75,90,144; mil.py:175,187)
//   column sums / means over tokens       (Pooler mean, DeformCrossTransMIL.py:193; bias gradients)
//   ReLU backward                         (_fc1, DeformCrossTransMIL.py:83)
// One wave per row, 16-byte accesses when the row length allows it (C % 256 == 0 uses float4 per lane,
// otherwise scalar with a 64-lane stride); partial column sums leave a workgroup through float atomics.
#include <algorithm>
#include "smml_common.h"

namespace {

constexpr int MAXV = 16;   // per-lane elements: supports C <= 1024

// y = (x - mean) * rstd * gamma + beta ; saves mean / rstd per row
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, long long R,
                                                            int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const float* xr = x + row * C;
  float v[MAXV];
  float s = 0.f;
  const int nv = (C + 63) >> 6;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (i < nv) {
      const int c = lane + 64 * i;
      v[i] = (c < C) ? xr[c] : 0.f;
      s += v[i];
    }
  }
  const float mu = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (i < nv) {
      const int c = lane + 64 * i;
      const float d = (c < C) ? v[i] - mu : 0.f;
      ss += d * d;
    }
  }
  const float rs = rsqrtf(wave_sum(ss) / (float)C + eps);
  float* yr = y + row * C;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (i < nv) {
      const int c = lane + 64 * i;
      if (c < C) yr[c] = (v[i] - mu) * rs * gamma[c] + beta[c];
    }
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// dx (+)= rstd * (g dy - mean(g dy) - xhat * mean(g dy xhat)); dgamma += sum dy xhat; dbeta += sum dy
// dy row for x row i is dy[(i / rows_per_dy) * C ...] * dy_scale  (rows_per_dy > 1: broadcast rows, e.g. the
// gradient of a mean over tokens).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            long long R, int C, long long rows_per_dy, float dy_scale,
                                                            int accumulate_dx) {
  __shared__ float red[2][4][64 * MAXV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nv = (C + 63) >> 6;
  float ag[MAXV], ab[MAXV], gm[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    ag[i] = 0.f; ab[i] = 0.f;
    const int c = lane + 64 * i;
    gm[i] = (i < nv && c < C) ? gamma[c] : 0.f;
  }
  for (long long row = (long long)blockIdx.x * 4 + wave; row < R; row += (long long)gridDim.x * 4) {
    const float* xr = x + row * C;
    const float* dr = dy + (row / rows_per_dy) * C;
    const float mu = mean[row], rs = rstd[row];
    float xh[MAXV], gd[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (i < nv) {
        const int c = lane + 64 * i;
        const bool ok = c < C;
        const float d = ok ? dr[c] * dy_scale : 0.f;
        xh[i] = ok ? (xr[c] - mu) * rs : 0.f;
        gd[i] = d * gm[i];
        s1 += gd[i];
        s2 += gd[i] * xh[i];
        ag[i] += d * xh[i];
        ab[i] += d;
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
    float* dxr = dx + row * C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (i < nv) {
        const int c = lane + 64 * i;
        if (c < C) {
          const float v = rs * (gd[i] - s1 - xh[i] * s2);
          dxr[c] = accumulate_dx ? dxr[c] + v : v;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (i < nv) { red[0][wave][lane + 64 * i] = ag[i]; red[1][wave][lane + 64 * i] = ab[i]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const float g = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    const float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
    atomicAdd(&dgamma[c], g);
    atomicAdd(&dbeta[c], b);
  }
}

// sum over the 32 lanes of a half-wave (both halves independently)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// C = 128 (the path's token width): a half-wave per row, one float4 per lane, two row pairs per trip - four rows of
// loads in flight per wave instead of one (the generic kernel is latency-bound at 128 columns)
__global__ __launch_bounds__(256) void layernorm_bwd128_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ dx,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               long long R, long long rows_per_dy, float dy_scale,
                                                               int accumulate_dx) {
  constexpr int C = 128;
  __shared__ float red[2][8][C];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31, half = lane >> 5;
  const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * hl);
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
  const long long stride = (long long)gridDim.x * 16;          // rows per trip of the whole grid: 4 waves x 2 halves x 2
  for (long long r0 = ((long long)blockIdx.x * 4 + wave) * 4 + half; r0 < R; r0 += stride) {
    float4 xv[2], dv[2];
    float mu[2], rs[2];
    bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long long row = r0 + 2 * u;
      ok[u] = row < R;
      const long long rr = ok[u] ? row : R - 1;
      xv[u] = *reinterpret_cast<const float4*>(x + rr * C + 4 * hl);
      dv[u] = *reinterpret_cast<const float4*>(dy + (rr / rows_per_dy) * C + 4 * hl);
      mu[u] = mean[rr]; rs[u] = rstd[rr];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float sc = ok[u] ? dy_scale : 0.f;
      const float4 d = make_float4(dv[u].x * sc, dv[u].y * sc, dv[u].z * sc, dv[u].w * sc);
      const float4 xh = make_float4((xv[u].x - mu[u]) * rs[u], (xv[u].y - mu[u]) * rs[u], (xv[u].z - mu[u]) * rs[u],
                                    (xv[u].w - mu[u]) * rs[u]);
      const float4 gd = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
      float s1 = (gd.x + gd.y) + (gd.z + gd.w);
      float s2 = (gd.x * xh.x + gd.y * xh.y) + (gd.z * xh.z + gd.w * xh.w);
      ag.x += d.x * xh.x; ag.y += d.y * xh.y; ag.z += d.z * xh.z; ag.w += d.w * xh.w;
      ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
      s1 = half_sum(s1) * (1.f / C);
      s2 = half_sum(s2) * (1.f / C);
      if (ok[u]) {
        float4* dxr = reinterpret_cast<float4*>(dx + (r0 + 2 * u) * C + 4 * hl);
        float4 v = make_float4(rs[u] * (gd.x - s1 - xh.x * s2), rs[u] * (gd.y - s1 - xh.y * s2),
                               rs[u] * (gd.z - s1 - xh.z * s2), rs[u] * (gd.w - s1 - xh.w * s2));
        if (accumulate_dx) { const float4 o = *dxr; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *dxr = v;
      }
    }
  }
  *reinterpret_cast<float4*>(&red[0][wave * 2 + half][4 * hl]) = ag;
  *reinterpret_cast<float4*>(&red[1][wave * 2 + half][4 * hl]) = ab;
  __syncthreads();
  if (threadIdx.x < C) {
    float g = 0.f, bsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { g += red[0][i][threadIdx.x]; bsum += red[1][i][threadIdx.x]; }
    atomicAdd(&dgamma[threadIdx.x], g);
    atomicAdd(&dbeta[threadIdx.x], bsum);
  }
}

// out[b, c] += scale * sum_{r in chunk} x[b, r, c]   (x [nb, R, C] contiguous; out zeroed by the caller)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long long R,
                                                     int C, float scale, int rows_per_block) {
  __shared__ float red[256];
  const int b = blockIdx.y;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(R, r0 + rows_per_block);
  // thread -> (column group, row lane): C <= 256 columns handled per pass
  for (int cb = 0; cb < C; cb += 256) {
    const int ncol = min(256, C - cb);
    // rows interleaved over 256 / ncol_pow2 row lanes
    int cp = 1;
    while (cp < ncol) cp <<= 1;
    const int rl = 256 / cp;
    const int col = threadIdx.x % cp, rlane = threadIdx.x / cp;
    float s = 0.f;
    if (col < ncol && rlane < rl) {
      const float* xb = x + ((long long)b * R) * C + cb + col;
      for (long long r = r0 + rlane; r < r1; r += rl) s += xb[r * C];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < cp && threadIdx.x < ncol) {
      float t = 0.f;
      for (int k = 0; k < rl; ++k) t += red[k * cp + threadIdx.x];
      atomicAdd(&out[(long long)b * C + cb + threadIdx.x], t * scale);
    }
    __syncthreads();
  }
}

// the same for C % 4 == 0 and C <= 1024 (every bias gradient of the path: C = 128 / 512): 16-byte loads, a thread owns four columns and
// every (256 / (C / 4))-th row of its block's chunk, eight loads in flight per thread (the scalar form above read 41 MB in 26 us
// = 1.6 TB/s; this one is HBM-paced)
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ x, float* __restrict__ out, long long R, int C,
                                                      float scale, int rows_per_block) {
  __shared__ float4 red[256];
  const int b = blockIdx.y;
  const int cq = C >> 2;                               // float4 columns (<= 256)
  const int rl = 256 / cq;                             // row lanes
  const int col = threadIdx.x % cq, rlane = threadIdx.x / cq;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (rlane < rl) {
    const float4* xb = reinterpret_cast<const float4*>(x + ((long long)b * R) * C) + col;
    long long r = r0 + rlane;
    for (; r + 7LL * rl < r1; r += 8LL * rl) {
      const float4 a0 = xb[r * cq], a1 = xb[(r + rl) * cq], a2 = xb[(r + 2LL * rl) * cq], a3 = xb[(r + 3LL * rl) * cq];
      const float4 a4 = xb[(r + 4LL * rl) * cq], a5 = xb[(r + 5LL * rl) * cq], a6 = xb[(r + 6LL * rl) * cq], a7 = xb[(r + 7LL * rl) * cq];
      s0.x += a0.x + a4.x; s0.y += a0.y + a4.y; s0.z += a0.z + a4.z; s0.w += a0.w + a4.w;
      s1.x += a1.x + a5.x; s1.y += a1.y + a5.y; s1.z += a1.z + a5.z; s1.w += a1.w + a5.w;
      s2.x += a2.x + a6.x; s2.y += a2.y + a6.y; s2.z += a2.z + a6.z; s2.w += a2.w + a6.w;
      s3.x += a3.x + a7.x; s3.y += a3.y + a7.y; s3.z += a3.z + a7.z; s3.w += a3.w + a7.w;
    }
    for (; r < r1; r += rl) { const float4 a = xb[r * cq]; s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; }
  }
  red[threadIdx.x] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                                 (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  if (threadIdx.x < cq) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < rl; ++k) { const float4 v = red[k * cq + threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    float* o = out + (long long)b * C + 4 * threadIdx.x;
    atomicAdd(o, t.x * scale); atomicAdd(o + 1, t.y * scale); atomicAdd(o + 2, t.z * scale); atomicAdd(o + 3, t.w * scale);
  }
}

__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                long long n) {
  const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 d = *reinterpret_cast<const float4*>(dy + i), v = *reinterpret_cast<const float4*>(y + i);
    *reinterpret_cast<float4*>(dx + i) =
        make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f);
  } else {
    for (long long k = i; k < n; ++k) dx[k] = y[k] > 0.f ? dy[k] : 0.f;
  }
}


// OrthogonalLoss (models/cmta_utils.py:1217-1228): one wave per sample, five |cosine similarities| of the
// Tail of BatchLoss (utils/loss.py:26-40) after the two Gram products: S = G_o / ||G_o||_row, V = mean_g G_v[g] / ||G_v[g]||_row,
// L = (S - V)^2 / N - and its backward - on [R, R] / [nv, R, R] matrices with R = batch x world <= 64: one workgroup, one thread per row.
// (In torch this tail is ~30 launch-bound ATen kernels per loss and direction: norm, div, stack, mean, sub, pow, div and their autograd twins.)
//   dL given:  dS = 2 (S - V) dL / N;  d G_o row i = (dS_i - S_i (S_i . dS_i)) / ||G_o,i||;  the same through each G_v[g] with -dS / nv
__global__ __launch_bounds__(64) void batchloss_tail_kernel(const float* __restrict__ Go, const float* __restrict__ Gv,
                                                            const float* __restrict__ dL, float* __restrict__ L,
                                                            float* __restrict__ dGo, float* __restrict__ dGv, int R, int nv, float Nf) {
  const int i = threadIdx.x;
  if (i >= R) return;
  const float* go = Go + (size_t)i * R;
  float no = 0.f;
  for (int j = 0; j < R; ++j) no = fmaf(go[j], go[j], no);
  no = sqrtf(no);
  float nvn[16];
  for (int g = 0; g < nv; ++g) {
    const float* gv = Gv + ((size_t)g * R + i) * R;
    float t = 0.f;
    for (int j = 0; j < R; ++j) t = fmaf(gv[j], gv[j], t);
    nvn[g] = sqrtf(t);
  }
  const float invn = 1.f / (float)nv;
  if (!dL) {
    for (int j = 0; j < R; ++j) {
      float v = 0.f;
      for (int g = 0; g < nv; ++g) v += Gv[((size_t)g * R + i) * R + j] / nvn[g];
      const float d = go[j] / no - v * invn;
      L[(size_t)i * R + j] = d * d / Nf;
    }
    return;
  }
  // backward: dS row, its projection on S, then the same per vgrid group
  float sdot = 0.f;
  float vdot[16];
  for (int g = 0; g < nv; ++g) vdot[g] = 0.f;
  for (int j = 0; j < R; ++j) {
    float v = 0.f;
    for (int g = 0; g < nv; ++g) v += Gv[((size_t)g * R + i) * R + j] / nvn[g];
    const float s = go[j] / no;
    const float ds = 2.f * (s - v * invn) * dL[(size_t)i * R + j] / Nf;
    sdot = fmaf(s, ds, sdot);
    for (int g = 0; g < nv; ++g) vdot[g] = fmaf(Gv[((size_t)g * R + i) * R + j] / nvn[g], -ds * invn, vdot[g]);
  }
  for (int j = 0; j < R; ++j) {
    float v = 0.f;
    for (int g = 0; g < nv; ++g) v += Gv[((size_t)g * R + i) * R + j] / nvn[g];
    const float s = go[j] / no;
    const float ds = 2.f * (s - v * invn) * dL[(size_t)i * R + j] / Nf;
    dGo[(size_t)i * R + j] = (ds - s * sdot) / no;
    for (int g = 0; g < nv; ++g) {
      const float y = Gv[((size_t)g * R + i) * R + j] / nvn[g];
      dGv[((size_t)g * R + i) * R + j] = (-ds * invn - y * vdot[g]) / nvn[g];
    }
  }
}

// rows P, P_hat, G, G_hat [B, D] by wavefront reductions:
//   loss = (1 - |cos(sg P, Ph)|) + (1 - |cos(sg G, Gh)|) + gamma (|cos(P, G)| + |cos(sg P, Gh)| + |cos(sg G, Ph)|)
// (sg = stop-gradient / .detach()).  Backward in the same kernel family: d cos(a,b)/da = b/(|a||b|) - cos a/|a|^2.
__global__ __launch_bounds__(64) void orth_loss_kernel(const float* __restrict__ P, const float* __restrict__ Ph,
                                                       const float* __restrict__ G, const float* __restrict__ Gh,
                                                       const float* __restrict__ dloss, float* __restrict__ loss,
                                                       float* __restrict__ dP, float* __restrict__ dPh, float* __restrict__ dG,
                                                       float* __restrict__ dGh, int D, float gamma, float eps) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float *p = P + (long long)b * D, *ph = Ph + (long long)b * D, *g = G + (long long)b * D, *gh = Gh + (long long)b * D;
  float pp = 0, hh = 0, gg = 0, kk = 0, p_ph = 0, g_gh = 0, p_g = 0, p_gh = 0, g_ph = 0;
  for (int i = lane; i < D; i += 64) {
    const float a = p[i], c = ph[i], e = g[i], f = gh[i];
    pp = fmaf(a, a, pp); hh = fmaf(c, c, hh); gg = fmaf(e, e, gg); kk = fmaf(f, f, kk);
    p_ph = fmaf(a, c, p_ph); g_gh = fmaf(e, f, g_gh); p_g = fmaf(a, e, p_g); p_gh = fmaf(a, f, p_gh); g_ph = fmaf(e, c, g_ph);
  }
  pp = wave_sum(pp); hh = wave_sum(hh); gg = wave_sum(gg); kk = wave_sum(kk);
  p_ph = wave_sum(p_ph); g_gh = wave_sum(g_gh); p_g = wave_sum(p_g); p_gh = wave_sum(p_gh); g_ph = wave_sum(g_ph);
  const float np = fmaxf(sqrtf(pp), eps), nh = fmaxf(sqrtf(hh), eps), ng = fmaxf(sqrtf(gg), eps), nk = fmaxf(sqrtf(kk), eps);
  const float c1 = p_ph / (np * nh), c2 = g_gh / (ng * nk), c3 = p_g / (np * ng), c4 = p_gh / (np * nk), c5 = g_ph / (ng * nh);
  if (loss && lane == 0) loss[b] = (1.f - fabsf(c1)) + (1.f - fabsf(c2)) + gamma * (fabsf(c3) + fabsf(c4) + fabsf(c5));
  if (!dloss) return;
  const float go = dloss[b];
  const float s1 = -copysignf(1.f, c1) * go, s2 = -copysignf(1.f, c2) * go;                       // pos pairs
  const float s3 = gamma * copysignf(1.f, c3) * go, s4 = gamma * copysignf(1.f, c4) * go, s5 = gamma * copysignf(1.f, c5) * go;
  for (int i = lane; i < D; i += 64) {
    const float a = p[i], c = ph[i], e = g[i], f = gh[i];
    // d cos(x, y) / dy = x / (|x||y|) - cos * y / |y|^2
    dPh[(long long)b * D + i] = s1 * (a / (np * nh) - c1 * c / (nh * nh)) + s5 * (e / (ng * nh) - c5 * c / (nh * nh));
    dGh[(long long)b * D + i] = s2 * (e / (ng * nk) - c2 * f / (nk * nk)) + s4 * (a / (np * nk) - c4 * f / (nk * nk));
    dP[(long long)b * D + i] = s3 * (e / (np * ng) - c3 * a / (np * np));                       // only cos(P, G) carries grad to P
    dG[(long long)b * D + i] = s3 * (a / (np * ng) - c3 * e / (ng * ng));
  }
}

}  // namespace

extern "C" {

int smml_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                           long long R, int C, float eps, void* stream) {
  SMML_REQUIRE(x && gamma && beta && y && mean && rstd, "smml_layernorm_fwd_f32: null pointer");
  SMML_REQUIRE(R > 0 && C > 0 && C <= 64 * MAXV, "smml_layernorm_fwd_f32: need 0 < C <= %d (got %d)", 64 * MAXV, C);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                     beta, y, mean, rstd, R, C, eps);
  SMML_LAUNCH_CHECK("smml_layernorm_fwd_f32");
  return SMML_OK;
}

// dgamma / dbeta are accumulated into (callers zero them once; the shared LayerNorm of the two streams adds twice)
int smml_layernorm_bwd_f32(const float* x, const float* dy, const float* gamma, const float* mean, const float* rstd,
                           float* dx, float* dgamma, float* dbeta, long long R, int C, long long rows_per_dy,
                           float dy_scale, int accumulate_dx, void* stream) {
  SMML_REQUIRE(x && dy && gamma && mean && rstd && dx && dgamma && dbeta, "smml_layernorm_bwd_f32: null pointer");
  SMML_REQUIRE(R > 0 && C > 0 && C <= 64 * MAXV, "smml_layernorm_bwd_f32: need 0 < C <= %d (got %d)", 64 * MAXV, C);
  SMML_REQUIRE(rows_per_dy >= 1, "smml_layernorm_bwd_f32: rows_per_dy must be >= 1");
  const long long nblk = (R + 3) / 4;
  const bool aligned = ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(dy) | reinterpret_cast<size_t>(dx) |
                         reinterpret_cast<size_t>(gamma)) & 15) == 0;
  if (C == 128 && aligned) {
    const long long nb16 = (R + 15) / 16;
    hipLaunchKernelGGL(layernorm_bwd128_kernel, dim3((unsigned)(nb16 < 768 ? nb16 : 768)), dim3(256), 0, (hipStream_t)stream,
                       x, dy, gamma, mean, rstd, dx, dgamma, dbeta, R, rows_per_dy, dy_scale, accumulate_dx);
  } else
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)(nblk < 1024 ? nblk : 1024)), dim3(256), 0, (hipStream_t)stream,
                     x, dy, gamma, mean, rstd, dx, dgamma, dbeta, R, C, rows_per_dy, dy_scale, accumulate_dx);
  SMML_LAUNCH_CHECK("smml_layernorm_bwd_f32");
  return SMML_OK;
}

// out [nb, C] must be zeroed by the caller; out[b, c] += scale * sum_r x[b, r, c]
int smml_colsum_f32(const float* x, float* out, int nb, long long R, int C, float scale, void* stream) {
  SMML_REQUIRE(x && out && nb > 0 && R > 0 && C > 0, "smml_colsum_f32: bad argument");
  SMML_REQUIRE(nb <= 65535, "smml_colsum_f32: too many batches");
  if ((C & 3) == 0 && C <= 1024 && 256 % (C >> 2) == 0 && ((reinterpret_cast<size_t>(x) & 15) == 0)) {
    // enough blocks to fill the chip a few times, at least 64 rows per row lane
    const int rl = 256 / (C >> 2);
    long long rpb = std::max<long long>(64LL * rl, (R * nb + 2047) / 2048);
    rpb = (rpb + rl - 1) / rl * rl;
    const long long nb4 = (R + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum4_kernel, dim3((unsigned)nb4, (unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, out, R, C, scale, (int)rpb);
    SMML_LAUNCH_CHECK("smml_colsum_f32/4");
    return SMML_OK;
  }
  int rows_per_block = 256;
  long long nblk = (R + rows_per_block - 1) / rows_per_block;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)nblk, (unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, out, R, C,
                     scale, rows_per_block);
  SMML_LAUNCH_CHECK("smml_colsum_f32");
  return SMML_OK;
}

// loss [B] (nullable) and, when dloss [B] is given, the four gradients [B, D]
int smml_orth_loss_f32(const float* P, const float* Ph, const float* G, const float* Gh, const float* dloss, float* loss,
                       float* dP, float* dPh, float* dG, float* dGh, int B, int D, float gamma, void* stream) {
  SMML_REQUIRE(P && Ph && G && Gh && B > 0 && D > 0, "smml_orth_loss_f32: bad argument");
  SMML_REQUIRE(loss || dloss, "smml_orth_loss_f32: nothing to compute");
  SMML_REQUIRE(!dloss || (dP && dPh && dG && dGh), "smml_orth_loss_f32: gradient outputs missing");
  hipLaunchKernelGGL(orth_loss_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, P, Ph, G, Gh, dloss, loss, dP, dPh, dG,
                     dGh, D, gamma, 1e-8f);
  SMML_LAUNCH_CHECK("smml_orth_loss_f32");
  return SMML_OK;
}

// BatchLoss tail (utils/loss.py:26-40): go [R, R], gv [nv, R, R] -> loss [R, R] (dloss == NULL) or dgo, dgv (dloss given); R <= 64, nv <= 16
int smml_batchloss_tail_f32(const float* go, const float* gv, const float* dloss, float* loss, float* dgo, float* dgv, int R, int nv,
                            float n_total, void* stream) {
  SMML_REQUIRE(go && gv && R > 0 && R <= 64 && nv > 0 && nv <= 16 && n_total > 0.f, "smml_batchloss_tail_f32: need 0 < R <= 64, 0 < nv <= 16");
  SMML_REQUIRE(dloss ? (dgo && dgv) : (loss != nullptr), "smml_batchloss_tail_f32: outputs missing");
  hipLaunchKernelGGL(batchloss_tail_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, go, gv, dloss, loss, dgo, dgv, R, nv, n_total);
  SMML_LAUNCH_CHECK("smml_batchloss_tail_f32");
  return SMML_OK;
}

int smml_relu_bwd_f32(const float* dy, const float* y, float* dx, long long n, void* stream) {
  SMML_REQUIRE(dy && y && dx && n > 0, "smml_relu_bwd_f32: bad argument");
  const long long nthr = (n + 3) / 4;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
  SMML_LAUNCH_CHECK("smml_relu_bwd_f32");
  return SMML_OK;
}

}  // extern "C"
