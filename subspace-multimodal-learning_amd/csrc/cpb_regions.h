// Exact evaluation of the continuous position bias (CPB) per LINEAR REGION of its MLP (round 5).
//
// The bias MLP of the reference (models/DeformableAttention2D.py:129-152: Linear(2, 32) - ReLU - Linear(32, 32) - ReLU - Linear(32, 1)
// on the signed-log offsets p of :148) is piecewise AFFINE in p: inside a region of the p plane on which the 64 ReLU decisions (D1, D2)
// are constant,  bias(p) = a_r . p + c_r  with  a_r = w3^T D2 W2 D1 W1,  c_r = w3^T D2 (W2 D1 b1 + b2) + b3.  One launch evaluates ONE
// such function at 4e8 points (8 bags x 8 heads x 10 000 queries x 625 keys); the arrangement of its 32 + 32 kink curves over the
// reachable square [-pmax, pmax]^2 has a few thousand regions.  So, per call:
//   build  (this file, ~6 small kernels, fp64): a 1024 x 1024 cell table over the square; a cell holds
//            kind 0  the region it lies in (no kink crosses it: decided EXACTLY from the ReLU patterns of its four corners - layer 1 is
//                    affine, and layer 2 is affine on a cell no layer-1 kink crosses),
//            kind 1  a record {kink line alpha . p + beta, region on its negative side, region on its positive side} when exactly one
//                    kink crosses it (a layer-1 line, or the kink of one layer-2 unit inside one layer-1 region: a straight line),
//            kind 2  a pointer to an 8 x 8 block of sub-cells of the same format when several kinks cross it (one level only),
//            kind 3  "evaluate the MLP": border cells and sub-cells that several kinks still cross (~3e-5 of the pairs);
//          the regions' (a, c) and ReLU patterns in dense tables (ids = rank of the 64-bit pattern: run-to-run identical).
//   forward (deform_attn.hip, region kernels): per pair two signed logs, one 4-byte gather (L2-resident table), for kind 1 one more
//          16-byte gather + 2 FMAs + a select, one 16-byte LDS read of (a, c), 2 FMAs - instead of 7 MFMAs + ~140 vector instructions
//          per (key, 32 queries); the region id of every pair is saved (2 bytes) in place of the 4 bytes of layer-2 ReLU bits.
//   backward: per pair the three moments  d bias . (1, p0, p1)  go to the region's 64-bit FIXED-POINT accumulators in LDS (integer
//          ds_add_u64: 12 x the rate of float LDS atomics on gfx950, tests/microbench/hist_probe.hip, and order-independent, so the
//          parameter gradients stay run-to-run identical); all six parameter gradients are LINEAR in the ~2 000 x 3 region moments
//          (a dense fp64 pass of a few microseconds);  d vs = - d bias . a_r . slog'(d) per pair.
// Nothing is approximated: every pair is evaluated on its own linear piece (or, for kind 3, by the MLP itself), and a decision can
// differ from an fp64 evaluation of the reference's formula only where the pre-activation is within fp32 rounding of zero.
#pragma once
#include "deform_common.h"

namespace {

constexpr int RG_G = 1024;               // level-0 cells per axis
constexpr int RG_SUB = 8;                // sub-cells per axis of a refined cell
constexpr int RG_SUBCAP = 16384;         // refined cells (8 x 8 entries each)
constexpr int RG_EDGECAP = 1 << 18;      // single-kink records
constexpr int RG_CANDCAP = 1 << 18;      // cells whose single kink is a layer-1 line, awaiting their check
constexpr int RG_HASH = 16384;           // hash slots of the region patterns (< 2^16: a slot fits the 16-bit fields of a record)
constexpr int RG_RCAP = 4096;            // regions with a dense id; patterns beyond it fall to kind 3
constexpr int RG_LCAP = 2048;            // regions whose (a, c) / moments live in LDS (ids above it: global memory)
constexpr unsigned RG_NONE = 0xFFFFu;    // "no region id": evaluate the MLP
constexpr unsigned long long RG_EMPTY = ~0ull;
constexpr unsigned RG_K_REGION = 0u << 30, RG_K_EDGE = 1u << 30, RG_K_SUB = 2u << 30, RG_K_MLP = 3u << 30, RG_PAYLOAD = 0x3FFFFFFFu;

struct RegionHeader {                    // first 256 bytes of the table buffer
  unsigned n_sub, n_edge, n_cand, n_regions, n_keys, overflow;   // overflow: bit 0 sub-blocks, 1 records, 2 candidates, 3 hash, 4 regions
  float pmax, cs, co, pad;               // cell of p: floor(p * cs + co)
  unsigned stats[8];
};

// byte offsets inside the table buffer (all multiples of 256)
struct RegionLayout {
  size_t hdr, reg, pat, t0, t1, edge, hkey, hid, sublist, cand, klist, corners, wd, total;
};
__host__ __device__ inline RegionLayout region_layout() {
  RegionLayout l;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  l.hdr = take(256);
  l.reg = take((size_t)RG_RCAP * 16);                       // float4 {a0, a1, c, 0} per region
  l.pat = take((size_t)RG_RCAP * 8);                        // D1 | D2 << 32
  l.t0 = take((size_t)RG_G * RG_G * 4);
  l.t1 = take((size_t)RG_SUBCAP * RG_SUB * RG_SUB * 4);
  l.edge = take((size_t)RG_EDGECAP * 16);                   // float4 {alpha0, alpha1, beta, bits(neg | pos << 16)}
  l.hkey = take((size_t)RG_HASH * 8);
  l.hid = take((size_t)RG_HASH * 4);
  l.sublist = take((size_t)RG_SUBCAP * 4);
  l.cand = take((size_t)RG_CANDCAP * 4);
  l.klist = take((size_t)RG_HASH * 8 + (size_t)RG_HASH * 4);   // compacted keys, then their slots
  l.corners = take((size_t)(RG_G + 1) * (RG_G + 1) * 8);
  l.wd = take(1200 * 8);                                    // the MLP's parameters in fp64
  l.total = o;
  return l;
}
struct RegionTables {                    // device pointers into one table buffer
  RegionHeader* hdr; float4* reg; unsigned long long* pat; unsigned* t0; unsigned* t1; float4* edge; unsigned long long* hkey;
  unsigned* hid; unsigned* sublist; unsigned* cand; unsigned long long* klist; unsigned* kslot; uint2* corners; double* wd;
};
inline RegionTables region_tables(void* base) {
  const RegionLayout l = region_layout();
  char* b = reinterpret_cast<char*>(base);
  RegionTables t;
  t.hdr = reinterpret_cast<RegionHeader*>(b + l.hdr); t.reg = reinterpret_cast<float4*>(b + l.reg);
  t.pat = reinterpret_cast<unsigned long long*>(b + l.pat); t.t0 = reinterpret_cast<unsigned*>(b + l.t0);
  t.t1 = reinterpret_cast<unsigned*>(b + l.t1); t.edge = reinterpret_cast<float4*>(b + l.edge);
  t.hkey = reinterpret_cast<unsigned long long*>(b + l.hkey); t.hid = reinterpret_cast<unsigned*>(b + l.hid);
  t.sublist = reinterpret_cast<unsigned*>(b + l.sublist); t.cand = reinterpret_cast<unsigned*>(b + l.cand);
  t.klist = reinterpret_cast<unsigned long long*>(b + l.klist); t.kslot = reinterpret_cast<unsigned*>(b + l.klist + (size_t)RG_HASH * 8);
  t.corners = reinterpret_cast<uint2*>(b + l.corners); t.wd = reinterpret_cast<double*>(b + l.wd);
  return t;
}

// fp64 copy of the parameters: w1 [32][2] at 0, b1 at 64, w2 [32][32] at 96, b2 at 1120, w3 at 1152, b3 at 1184
constexpr int WD_W1 = 0, WD_B1 = 64, WD_W2 = 96, WD_B2 = 1120, WD_W3 = 1152, WD_B3 = 1184, WD_N = 1185;

__global__ void region_prep_kernel(CpbParams cp, float pmax, RegionTables t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 64) t.wd[WD_W1 + i] = (double)cp.w1[i];
  else if (i < 96) t.wd[i] = (double)cp.b1[i - 64];
  else if (i < 1120) t.wd[i] = (double)cp.w2[i - 96];
  else if (i < 1152) t.wd[i] = (double)cp.b2[i - 1120];
  else if (i < 1184) t.wd[i] = (double)cp.w3[i - 1152];
  else if (i == 1184) t.wd[i] = (double)cp.b3[0];
  if (i == 0) {
    RegionHeader h;
    h.n_sub = h.n_edge = h.n_cand = h.n_regions = h.n_keys = h.overflow = 0;
    h.pmax = pmax; h.cs = (float)((double)RG_G / (2.0 * (double)pmax)); h.co = (float)(RG_G / 2); h.pad = 0.f;
    for (int k = 0; k < 8; ++k) h.stats[k] = 0;
    *t.hdr = h;
  }
  for (int s = i; s < RG_HASH; s += gridDim.x * blockDim.x) t.hkey[s] = RG_EMPTY;
}

// ReLU patterns of the MLP at p in fp64, "natural" evaluation (every unit decides by its own pre-activation).  wd: uniform address
// -> scalar loads; one thread evaluates one point (1088 fp64 FMAs).
__device__ __forceinline__ uint2 region_eval(const double* __restrict__ wd, double p0, double p1) {
  double h1[CH];
  unsigned d1 = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const double x = fma(wd[WD_W1 + 2 * i], p0, fma(wd[WD_W1 + 2 * i + 1], p1, wd[WD_B1 + i]));
    d1 |= (x > 0.0) ? (1u << i) : 0u;
    h1[i] = x > 0.0 ? x : 0.0;
  }
  unsigned d2 = 0;
#pragma unroll 4
  for (int o = 0; o < CH; ++o) {
    double x = wd[WD_B2 + o];
#pragma unroll
    for (int i = 0; i < CH; ++i) x = fma(wd[WD_W2 + o * CH + i], h1[i], x);
    d2 |= (x > 0.0) ? (1u << o) : 0u;
  }
  return make_uint2(d1, d2);
}

__global__ __launch_bounds__(256) void region_corners_kernel(RegionTables t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr int GP = RG_G + 1;
  if (i >= GP * GP) return;
  const int iy = i / GP, ix = i - iy * GP;
  const double pm = (double)t.hdr->pmax, h = 2.0 * pm / RG_G;
  t.corners[i] = region_eval(t.wd, -pm + ix * h, -pm + iy * h);
}

__device__ __forceinline__ unsigned region_hash_insert(unsigned long long* __restrict__ hkey, unsigned long long key) {
  if (key == RG_EMPTY) return RG_NONE;              // the all-ones pattern doubles as the empty marker: such pairs take the MLP path
  unsigned slot = (unsigned)(mix64(key) & (RG_HASH - 1));
  for (int probe = 0; probe < RG_HASH; ++probe) {
    const unsigned long long prev = atomicCAS(&hkey[slot], RG_EMPTY, key);
    if (prev == RG_EMPTY || prev == key) return slot;
    slot = (slot + 1) & (RG_HASH - 1);
  }
  return RG_NONE;
}

// record of a cell crossed by the kink of layer-2 unit o while every layer-1 unit keeps its sign d1: inside the cell
// x2_o(p) = alpha . p + beta with alpha = sum_i W2[o][i] d1_i W1[i], beta = sum_i W2[o][i] d1_i b1[i] + b2[o]
__device__ __forceinline__ float4 region_edge_l2(const double* __restrict__ wd, unsigned d1, unsigned d2, int o, unsigned long long* hkey) {
  double a0 = 0.0, a1 = 0.0, be = wd[WD_B2 + o];
  for (int i = 0; i < CH; ++i)
    if ((d1 >> i) & 1u) {
      const double w = wd[WD_W2 + o * CH + i];
      a0 = fma(w, wd[WD_W1 + 2 * i], a0); a1 = fma(w, wd[WD_W1 + 2 * i + 1], a1); be = fma(w, wd[WD_B1 + i], be);
    }
  const unsigned bit = 1u << o;
  const unsigned neg = region_hash_insert(hkey, (unsigned long long)d1 | ((unsigned long long)(d2 & ~bit) << 32));
  const unsigned pos = region_hash_insert(hkey, (unsigned long long)d1 | ((unsigned long long)(d2 | bit) << 32));
  return make_float4((float)a0, (float)a1, (float)be, __uint_as_float(neg | (pos << 16)));
}
__device__ __forceinline__ float4 region_edge_l1(const double* __restrict__ wd, unsigned d1, unsigned d2, int u, unsigned long long* hkey) {
  const unsigned bit = 1u << u;
  const unsigned neg = region_hash_insert(hkey, (unsigned long long)(d1 & ~bit) | ((unsigned long long)d2 << 32));
  const unsigned pos = region_hash_insert(hkey, (unsigned long long)(d1 | bit) | ((unsigned long long)d2 << 32));
  return make_float4((float)wd[WD_W1 + 2 * u], (float)wd[WD_W1 + 2 * u + 1], (float)wd[WD_B1 + u], __uint_as_float(neg | (pos << 16)));
}
__device__ __forceinline__ unsigned region_new_edge(RegionTables t, float4 rec) {
  const unsigned idx = atomicAdd(&t.hdr->n_edge, 1u);
  if (idx >= (unsigned)RG_EDGECAP) { atomicOr(&t.hdr->overflow, 2u); return RG_K_MLP; }
  t.edge[idx] = rec;
  return RG_K_EDGE | idx;
}

// Classification of one cell from the patterns of its four corners.  -> final entry, or RG_K_SUB without payload = "several kinks"
// (the caller refines or gives up), or 0xFFFFFFFF = "one layer-1 line crosses it, layer-2 signs equal at the corners": to be checked
// at the two points where the line leaves the cell (the only places a layer-2 unit could still change sign inside it).
constexpr unsigned RG_PENDING = 0xFFFFFFFFu;
__device__ __forceinline__ unsigned region_classify(RegionTables t, uint2 c00, uint2 c10, uint2 c01, uint2 c11) {
  const unsigned m1 = (c00.x ^ c10.x) | (c00.x ^ c01.x) | (c00.x ^ c11.x);
  const unsigned m2 = (c00.y ^ c10.y) | (c00.y ^ c01.y) | (c00.y ^ c11.y);
  if (m1 == 0u && m2 == 0u) {
    const unsigned slot = region_hash_insert(t.hkey, (unsigned long long)c00.x | ((unsigned long long)c00.y << 32));
    return slot == RG_NONE ? RG_K_MLP : (RG_K_REGION | slot);
  }
  if (m1 == 0u && __popc(m2) == 1) return region_new_edge(t, region_edge_l2(t.wd, c00.x, c00.y, __ffs(m2) - 1, t.hkey));
  if (m2 == 0u && __popc(m1) == 1) return RG_PENDING;
  return RG_K_SUB;
}

// the check of a RG_PENDING cell [x0, x0 + h] x [y0, y0 + h] crossed by layer-1 line u: layer-2 patterns where the line leaves it
__device__ __forceinline__ bool region_line_cell_ok(const double* __restrict__ wd, int u, unsigned d2, double x0, double y0, double h) {
  const double wx = wd[WD_W1 + 2 * u], wy = wd[WD_W1 + 2 * u + 1], bb = wd[WD_B1 + u];
  const double cx[4] = {x0, x0 + h, x0 + h, x0}, cy[4] = {y0, y0, y0 + h, y0 + h};     // corners in cyclic order
  double v[4];
  for (int k = 0; k < 4; ++k) v[k] = fma(wx, cx[k], fma(wy, cy[k], bb));
  bool ok = true;
  for (int k = 0; k < 4; ++k) {
    const int n = (k + 1) & 3;
    if ((v[k] > 0.0) != (v[n] > 0.0)) {
      const double s = v[k] / (v[k] - v[n]);
      const uint2 m = region_eval(wd, cx[k] + s * (cx[n] - cx[k]), cy[k] + s * (cy[n] - cy[k]));
      ok = ok && (m.y == d2);
    }
  }
  return ok;
}

__device__ __forceinline__ unsigned region_new_sub(RegionTables t, unsigned cell) {
  const unsigned idx = atomicAdd(&t.hdr->n_sub, 1u);
  if (idx >= (unsigned)RG_SUBCAP) { atomicOr(&t.hdr->overflow, 1u); return RG_K_MLP; }
  t.sublist[idx] = cell;
  return RG_K_SUB | idx;
}

__global__ __launch_bounds__(256) void region_classify0_kernel(RegionTables t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= RG_G * RG_G) return;
  const int iy = i / RG_G, ix = i - iy * RG_G;
  constexpr int GP = RG_G + 1;
  unsigned e;
  if (ix == 0 || iy == 0 || ix == RG_G - 1 || iy == RG_G - 1) e = RG_K_MLP;      // the border also takes every point outside the square
  else {
    e = region_classify(t, t.corners[iy * GP + ix], t.corners[iy * GP + ix + 1], t.corners[(iy + 1) * GP + ix], t.corners[(iy + 1) * GP + ix + 1]);
    if (e == RG_PENDING) {
      const unsigned idx = atomicAdd(&t.hdr->n_cand, 1u);
      if (idx < (unsigned)RG_CANDCAP) { t.cand[idx] = (unsigned)i; e = RG_K_MLP; }      // region_cand0_kernel writes the final entry
      else { atomicOr(&t.hdr->overflow, 4u); e = region_new_sub(t, (unsigned)i); }
    } else if (e == RG_K_SUB) e = region_new_sub(t, (unsigned)i);
  }
  t.t0[i] = e;
}

__global__ __launch_bounds__(256) void region_cand0_kernel(RegionTables t) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= min(t.hdr->n_cand, (unsigned)RG_CANDCAP)) return;
  const unsigned i = t.cand[k];
  const int iy = i / RG_G, ix = i - iy * RG_G;
  constexpr int GP = RG_G + 1;
  const uint2 c00 = t.corners[iy * GP + ix], c10 = t.corners[iy * GP + ix + 1], c01 = t.corners[(iy + 1) * GP + ix];
  const unsigned m1 = (c00.x ^ c10.x) | (c00.x ^ c01.x) | (c00.x ^ t.corners[(iy + 1) * GP + ix + 1].x);
  const int u = __ffs(m1) - 1;
  const double pm = (double)t.hdr->pmax, h = 2.0 * pm / RG_G;
  unsigned e;
  if (region_line_cell_ok(t.wd, u, c00.y, -pm + ix * h, -pm + iy * h, h)) e = region_new_edge(t, region_edge_l1(t.wd, c00.x, c00.y, u, t.hkey));
  else e = region_new_sub(t, i);
  t.t0[i] = e;
}

// level 1: one workgroup per refined cell, 81 sub-corners, 64 sub-cells
__global__ __launch_bounds__(128) void region_sub_kernel(RegionTables t) {
  __shared__ uint2 sc[(RG_SUB + 1) * (RG_SUB + 1)];
  const unsigned nsub = min(t.hdr->n_sub, (unsigned)RG_SUBCAP);
  const double pm = (double)t.hdr->pmax, h = 2.0 * pm / RG_G, hs = h / RG_SUB;
  for (unsigned blk = blockIdx.x; blk < nsub; blk += gridDim.x) {
    const unsigned cell = t.sublist[blk];
    const int iy = cell / RG_G, ix = cell - iy * RG_G;
    const double x0 = -pm + ix * h, y0 = -pm + iy * h;
    __syncthreads();
    if (threadIdx.x < (RG_SUB + 1) * (RG_SUB + 1)) {
      const int sy = threadIdx.x / (RG_SUB + 1), sx = threadIdx.x - sy * (RG_SUB + 1);
      sc[threadIdx.x] = region_eval(t.wd, x0 + sx * hs, y0 + sy * hs);
    }
    __syncthreads();
    if (threadIdx.x < RG_SUB * RG_SUB) {
      const int sy = threadIdx.x / RG_SUB, sx = threadIdx.x - sy * RG_SUB;
      const uint2 c00 = sc[sy * (RG_SUB + 1) + sx], c10 = sc[sy * (RG_SUB + 1) + sx + 1], c01 = sc[(sy + 1) * (RG_SUB + 1) + sx],
                  c11 = sc[(sy + 1) * (RG_SUB + 1) + sx + 1];
      unsigned e = region_classify(t, c00, c10, c01, c11);
      if (e == RG_PENDING) {
        const unsigned m1 = (c00.x ^ c10.x) | (c00.x ^ c01.x) | (c00.x ^ c11.x);
        const int u = __ffs(m1) - 1;
        e = region_line_cell_ok(t.wd, u, c00.y, x0 + sx * hs, y0 + sy * hs, hs) ? region_new_edge(t, region_edge_l1(t.wd, c00.x, c00.y, u, t.hkey))
                                                                                   : RG_K_MLP;
      } else if (e == RG_K_SUB) e = RG_K_MLP;
      t.t1[(size_t)blk * (RG_SUB * RG_SUB) + threadIdx.x] = e;
    }
  }
}

// dense region ids = rank of the pattern among the patterns in use (independent of insertion order)
__global__ __launch_bounds__(256) void region_compact_kernel(RegionTables t) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= RG_HASH) return;
  const unsigned long long k = t.hkey[s];
  t.hid[s] = RG_NONE;
  if (k != RG_EMPTY) {
    const unsigned idx = atomicAdd(&t.hdr->n_keys, 1u);
    t.klist[idx] = k; t.kslot[idx] = (unsigned)s;
  }
}
__global__ __launch_bounds__(256) void region_rank_kernel(RegionTables t) {
  const unsigned n = t.hdr->n_keys;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { t.hdr->n_regions = min(n, (unsigned)RG_RCAP); if (n > (unsigned)RG_RCAP) atomicOr(&t.hdr->overflow, 16u); }
  if (i >= n) return;
  const unsigned long long k = t.klist[i];
  unsigned rank = 0;
  for (unsigned j = 0; j < n; ++j) rank += (t.klist[j] < k) ? 1u : 0u;
  if (rank >= (unsigned)RG_RCAP) return;                                     // hid stays RG_NONE: the MLP path
  t.hid[t.kslot[i]] = rank;
  t.pat[rank] = k;
  // (a, c) of the region: c1_i = d1_i sum_o W2[o][i] w3[o] d2_o;  a = sum_i c1_i W1[i];  c = sum_i c1_i b1[i] + sum_o d2_o w3[o] b2[o] + b3
  const double* __restrict__ wd = t.wd;
  const unsigned d1 = (unsigned)k, d2 = (unsigned)(k >> 32);
  double a0 = 0.0, a1 = 0.0, c = wd[WD_B3];
  for (int o = 0; o < CH; ++o)
    if ((d2 >> o) & 1u) c = fma(wd[WD_W3 + o], wd[WD_B2 + o], c);
  for (int i2 = 0; i2 < CH; ++i2) {
    if (!((d1 >> i2) & 1u)) continue;
    double c1 = 0.0;
    for (int o = 0; o < CH; ++o)
      if ((d2 >> o) & 1u) c1 = fma(wd[WD_W2 + o * CH + i2], wd[WD_W3 + o], c1);
    a0 = fma(c1, wd[WD_W1 + 2 * i2], a0); a1 = fma(c1, wd[WD_W1 + 2 * i2 + 1], a1); c = fma(c1, wd[WD_B1 + i2], c);
  }
  t.reg[rank] = make_float4((float)a0, (float)a1, (float)c, 0.f);
}
// hash slots -> dense ids in the cell tables and the records
__global__ __launch_bounds__(256) void region_remap_kernel(RegionTables t) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n0 = (size_t)RG_G * RG_G, n1 = (size_t)min(t.hdr->n_sub, (unsigned)RG_SUBCAP) * (RG_SUB * RG_SUB);
  const size_t n2 = min(t.hdr->n_edge, (unsigned)RG_EDGECAP);
  if (i < n0 + n1) {
    unsigned* p = i < n0 ? &t.t0[i] : &t.t1[i - n0];
    const unsigned e = *p;
    if ((e >> 30) == 0u) {
      const unsigned id = t.hid[e & RG_PAYLOAD];
      *p = id == RG_NONE ? RG_K_MLP : (RG_K_REGION | id);
    }
  } else if (i < n0 + n1 + n2) {
    float4 r = t.edge[i - n0 - n1];
    const unsigned b = __float_as_uint(r.w), neg = b & 0xFFFFu, pos = b >> 16;
    const unsigned ineg = neg == RG_NONE ? RG_NONE : t.hid[neg], ipos = pos == RG_NONE ? RG_NONE : t.hid[pos];
    r.w = __uint_as_float(ineg | (ipos << 16));
    t.edge[i - n0 - n1] = r;
  }
}

static int region_build_launch(CpbParams cp, float pmax, void* tables, hipStream_t st) {
  const RegionTables t = region_tables(tables);
  hipLaunchKernelGGL(region_prep_kernel, dim3(64), dim3(256), 0, st, cp, pmax, t);
  constexpr int GP = RG_G + 1;
  hipLaunchKernelGGL(region_corners_kernel, dim3((GP * GP + 255) / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_classify0_kernel, dim3(RG_G * RG_G / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_cand0_kernel, dim3(RG_CANDCAP / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_sub_kernel, dim3(4096), dim3(128), 0, st, t);
  hipLaunchKernelGGL(region_compact_kernel, dim3(RG_HASH / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_rank_kernel, dim3(RG_HASH / 256), dim3(256), 0, st, t);
  const size_t nmax = (size_t)RG_G * RG_G + (size_t)RG_SUBCAP * RG_SUB * RG_SUB + RG_EDGECAP;
  hipLaunchKernelGGL(region_remap_kernel, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, st, t);
  return 0;
}


// ------------------------------------------------------------------------------------------------
// per-pair pieces shared by the region kernels
// ------------------------------------------------------------------------------------------------
struct RegionView {                      // what a region kernel reads of the tables
  const RegionHeader* hdr; const unsigned* t0; const unsigned* t1; const float4* edge; const float4* reg;
};
inline RegionView region_view(void* base) {
  const RegionTables t = region_tables(base);
  return RegionView{t.hdr, t.t0, t.t1, t.edge, t.reg};
}
__device__ __forceinline__ int region_cell(float u) { return min(max((int)u, 0), RG_G - 1); }
// first-level entry address of p (cell coordinates in cx, cy; u = p cs + co)
__device__ __forceinline__ unsigned region_cell_index(float u0, float u1, int& cx, int& cy) {
  cx = region_cell(u0); cy = region_cell(u1);
  return (unsigned)(cy * RG_G + cx);
}
// entry -> region id (RG_NONE: evaluate the MLP).  e: first-level entry of the pair's cell.
__device__ __forceinline__ unsigned region_resolve(const RegionView& rv, unsigned e, float p0, float p1, float u0, float u1, int cx, int cy) {
  if ((e >> 30) == 2u) {                                    // refined cell: the sub-cell's entry
    const int sx = min(max((int)((u0 - (float)cx) * (float)RG_SUB), 0), RG_SUB - 1);
    const int sy = min(max((int)((u1 - (float)cy) * (float)RG_SUB), 0), RG_SUB - 1);
    e = rv.t1[(size_t)(e & RG_PAYLOAD) * (RG_SUB * RG_SUB) + sy * RG_SUB + sx];
  }
  unsigned rid = e & 0xFFFFu;
  if ((e >> 30) == 1u) {                                    // one kink crosses the cell: which side
    const float4 r = rv.edge[e & RG_PAYLOAD];
    const float g = fmaf(r.x, p0, fmaf(r.y, p1, r.z));
    const unsigned b = __float_as_uint(r.w);
    rid = g > 0.f ? (b >> 16) : (b & 0xFFFFu);
  }
  return (e >> 30) == 3u ? RG_NONE : rid;
}

// The MLP itself for ONE pair, evaluated by the whole wave (kind 3 pairs: ~3e-5 of all): lane c = l & 31 owns hidden unit c of both
// layers (the two lane halves compute the same values).  w2t: W2 transposed in LDS, w2t[i * 32 + o] = W2[o][i].
struct CoopMlp {
  float w1x, w1y, b1, b2, w3, b3;        // of unit c
  const float* w2t;                      // LDS [32][32], in-major
  const float* w2r;                      // LDS [32][32], out-major (W2 as stored), backward only
};
__device__ __forceinline__ float coop_sum32(float v) {       // sum over the 32 lanes of a half (the halves hold equal values)
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// -> bias (uniform); h1 / m2 of unit c are left in the references for the backward
__device__ __forceinline__ float coop_mlp_fwd(const CoopMlp& m, float p0, float p1, int c, float& h1, bool& on2) {
  const float x1 = fmaf(m.w1x, p0, fmaf(m.w1y, p1, m.b1));
  h1 = x1 > 0.f ? x1 : 0.f;
  float x2 = m.b2;
#pragma unroll 8
  for (int i = 0; i < CH; ++i) x2 = fmaf(m.w2t[i * CH + c], __shfl(h1, i), x2);
  on2 = x2 > 0.f;
  return coop_sum32(on2 ? x2 * m.w3 : 0.f) + m.b3;
}

// ------------------------------------------------------------------------------------------------
// forward, position bias per linear region (PD = 2, signed-log offsets, one head per offset group)
// ------------------------------------------------------------------------------------------------
template <bool SAVE>
__global__ __launch_bounds__(256, 2) void deform_region_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ VS,
    const float* __restrict__ GQ, CpbParams cp, RegionView rv, float* __restrict__ O, float* __restrict__ LSE, float* __restrict__ LT,
    unsigned short* __restrict__ RID, int N, int J, int H, int NST, float scale, DropCfg dc_in) {
  const DropCfg dc = drop_resolve(dc_in);
  __shared__ __attribute__((aligned(16))) _Float16 Kp[2][KT * FRLD];         // K tile, fp16 hi / lo planes, row image (A operand of S^T)
  __shared__ __attribute__((aligned(16))) _Float16 Vp[2][KT * FTLD];         // V tile, hi / lo planes, read transposed (A operand of O^T)
  __shared__ __attribute__((aligned(16))) float4 regl[RG_LCAP];              // (a0, a1, c) of the regions
  __shared__ float w2t[CH * CH];                                             // W2 transposed (the MLP path)
  __shared__ float vsl[KT][2];                                               // sample positions of the tile's keys

  const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (QT * WAVES) + wave * QT;
  const int HD = H * DH;
  const bool qvalid = (q0 + c) < N;
  const int qi = qvalid ? (q0 + c) : (N - 1);

  {
    const int nreg = min((int)rv.hdr->n_regions, RG_LCAP);
    for (int i = tid; i < nreg; i += 256) regl[i] = rv.reg[i];
    for (int i = tid; i < CH * CH; i += 256) w2t[(i & 31) * CH + (i >> 5)] = cp.w2[i];      // i = o * 32 + in
  }
  const float cs = rv.hdr->cs, co = rv.hdr->co;
  const CoopMlp mlp{cp.w1[c * 2], cp.w1[c * 2 + 1], cp.b1[c], cp.b2[c], cp.w3[c], cp.b3[0], w2t, nullptr};

  // scaled Q of this lane's query as the B operand of S^T = K . Q^T: k-step st holds d = 16 st + 8 hf + j, fp16 hi / lo
  half8 qh[4], ql[4];
  {
    const float* qp = Q + ((size_t)b * N + qi) * HD + h * DH + hf * 8;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const float4 t0 = *reinterpret_cast<const float4*>(qp + 16 * st), t1 = *reinterpret_cast<const float4*>(qp + 16 * st + 4);
      const float x8[8] = {t0.x * scale, t0.y * scale, t0.z * scale, t0.w * scale, t1.x * scale, t1.y * scale, t1.z * scale, t1.w * scale};
      split8(x8, qh[st], ql[st]);
    }
  }
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);      // transposed-read lane map
  const float gq0 = GQ[(size_t)qi * 2], gq1 = GQ[(size_t)qi * 2 + 1];

  floatx16 oacc0 = {0}, oacc1 = {0};
  float m_run = -INFINITY, l_run = 0.f;
  const float* Kb = K + (size_t)b * J * HD + h * DH;
  const float* Vb = V + (size_t)b * J * HD + h * DH;
  const float* VSb = VS + (size_t)(b * H + h) * J * 2;                       // one head per offset group: group = head
  float* LTb = LT ? LT + ((size_t)(b * H + h) * NST + q0) * J : nullptr;    // this wave's [J][32] block (layout: deform_attn_fwd_kernel)
  unsigned short* RIDb = RID ? RID + ((size_t)(b * H + h) * NST + q0) * J : nullptr;

  const int ntiles = (J + KT - 1) / KT;
  for (int kt = 0; kt < ntiles; ++kt) {
    const int j0 = kt * KT;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = (tid >> 4) + 16 * i, d4 = (tid & 15) * 4;
      float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
      if (j0 + key < J) {
        kv = *reinterpret_cast<const float4*>(Kb + (size_t)(j0 + key) * HD + d4);
        vv = *reinterpret_cast<const float4*>(Vb + (size_t)(j0 + key) * HD + d4);
      }
      uint2v hi, lo;
      split4_h2(kv, hi, lo);
      *reinterpret_cast<uint2v*>(&Kp[0][key * FRLD + d4]) = hi; *reinterpret_cast<uint2v*>(&Kp[1][key * FRLD + d4]) = lo;
      split4_h2(vv, hi, lo);
      *reinterpret_cast<uint2v*>(&Vp[0][key * FTLD + d4]) = hi; *reinterpret_cast<uint2v*>(&Vp[1][key * FTLD + d4]) = lo;
    }
    if (tid < KT) {
      const int key = min(j0 + tid, J - 1);
      vsl[tid][0] = VSb[(size_t)key * 2];
      vsl[tid][1] = VSb[(size_t)key * 2 + 1];
    }
    __syncthreads();

    // position bias of this lane's 16 (key, query) pairs: register r <-> key acc_row(r, hf), in two groups of eight - first the
    // cell entries of a group (independent gathers, all in flight; the S^T products run under the first group's), then their resolution
    const int nk = min(KT, J - j0);
    floatx16 s = {0};
    float tmax = -INFINITY;
#pragma unroll
    for (int g8 = 0; g8 < 2; ++g8) {
      float p0[8], p1[8];
      unsigned ent[8];
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const float2 vv = *reinterpret_cast<const float2*>(&vsl[acc_row(8 * g8 + r8, hf)][0]);
        p0[r8] = slog1p(gq0 - vv.x);
        p1[r8] = slog1p(gq1 - vv.y);
        int cx, cy;
        ent[r8] = rv.t0[region_cell_index(fmaf(p0[r8], cs, co), fmaf(p1[r8], cs, co), cx, cy)];
      }
      if (g8 == 0) {
        // S^T[key, query] = K . (scale Q)^T
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          const int o = c * FRLD + 16 * st + 8 * hf;
          const half8 kh = *reinterpret_cast<const half8*>(&Kp[0][o]), kl = *reinterpret_cast<const half8*>(&Kp[1][o]);
          s = mfma16(kl, qh[st], s);
          s = mfma16(kh, ql[st], s);
          s = mfma16(kh, qh[st], s);
        }
      }
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int r = 8 * g8 + r8;
        const float u0 = fmaf(p0[r8], cs, co), u1 = fmaf(p1[r8], cs, co);
        int cx, cy;
        region_cell_index(u0, u1, cx, cy);
        const unsigned id = region_resolve(rv, ent[r8], p0[r8], p1[r8], u0, u1, cx, cy);
        float bias = 0.f;
        if (id < (unsigned)RG_LCAP) {
          const float4 ac = regl[id];
          bias = fmaf(ac.x, p0[r8], fmaf(ac.y, p1[r8], ac.z));
        } else if (id != RG_NONE) {
          const float4 ac = rv.reg[id];
          bias = fmaf(ac.x, p0[r8], fmaf(ac.y, p1[r8], ac.z));
        }
        // pairs without a region: the MLP itself, one pair at a time, by the whole wave
        unsigned long long todo = __ballot(id == RG_NONE);
        while (todo) {
          const int l = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          float h1; bool on2;
          const float v = coop_mlp_fwd(mlp, __shfl(p0[r8], l), __shfl(p1[r8], l), c, h1, on2);
          if (lane == l) bias = v;
        }
        const bool kin = acc_row(r, hf) < nk;
        if (SAVE && kin) RIDb[(size_t)(j0 + acc_row(r, hf)) * 32 + c] = (unsigned short)id;
        const float sv = kin ? s[r] + bias : -INFINITY;
        s[r] = sv;
        tmax = fmaxf(tmax, sv);
      }
    }
    unsigned keepbits = 0xFFFFu;              // dropout decisions of this lane's 16 keys (bit r)
    if (dc.thresh) {
      const unsigned long long base2 = ((unsigned long long)(b * H + h) * N + qi) * ((J + 1) >> 1) + (j0 >> 1);
      keepbits = 0u;
#pragma unroll
      for (int r = 0; r < 16; r += 2) keepbits |= drop_keep2(dc, base2 + (acc_row(r, hf) >> 1)) << r;
    }
    if (SAVE) {                               // rows are padded to whole workgroup tiles: lanes past N write padding
      if (dc.thresh) {
        tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (acc_row(r, hf) < nk) s[r] = stash_keep(s[r], (keepbits >> r) & 1u);
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
      psum += p;
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

    // O^T[d, query] += V^T . P^T
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
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
  }

  l_run = xhalf_sum(l_run);
  const float inv = 1.f / l_run;
  if (qvalid) {
    float* op = O + ((size_t)b * N + qi) * HD + h * DH;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = 8 * rg + 4 * hf;
      *reinterpret_cast<float4*>(op + d) = make_float4(oacc0[4 * rg] * inv, oacc0[4 * rg + 1] * inv, oacc0[4 * rg + 2] * inv, oacc0[4 * rg + 3] * inv);
      *reinterpret_cast<float4*>(op + 32 + d) = make_float4(oacc1[4 * rg] * inv, oacc1[4 * rg + 1] * inv, oacc1[4 * rg + 2] * inv, oacc1[4 * rg + 3] * inv);
    }
    if (hf == 0) LSE[(size_t)(b * H + h) * N + qi] = m_run + logf(l_run);
  }
}


// ------------------------------------------------------------------------------------------------
// backward of the position bias per linear region: d vs per pair, region moments of d bias in 64-bit fixed point
// ------------------------------------------------------------------------------------------------
// value -> fixed point with scale S (a power of two): round-to-nearest through the 1.5 * 2^52 trick (|v S| < 2^51)
__device__ __forceinline__ long long region_fix(float v, double S) {
  return __double_as_longlong(fma((double)v, S, 6755399441055744.0)) - 0x4338000000000000ll;
}
struct RegionScale { double S; int e; };
// amax_bits: bit pattern of max |d scores| of the launch (written by the dq pass).  S = 2^(kbits - e) with amax < 2^e.
__device__ __forceinline__ RegionScale region_scale(unsigned amax_bits, int kbits) {
  RegionScale r;
  r.e = amax_bits ? (int)((amax_bits >> 23) & 0xFFu) - 126 : 0;
  r.S = __longlong_as_double((long long)(1023 + kbits - r.e) << 52);
  return r;
}
constexpr int RG_GRAD = CPB_SLAB;        // dW2[1024] | dW1[32 * 2] | db1[32] | db2[32] | dW3[32] | db3[1]

struct RegionBwdLds {                    // dynamic LDS of cpb_region_bwd_kernel
  unsigned long long hist[RG_LCAP * 3];
  unsigned long long grad[RG_GRAD];
  float2 reg2[RG_LCAP];
  float w2t[CH * CH], w2r[CH * CH];
  float2 dvs[16][64];                    // per-wave d vs of its key block (combined in a fixed order at the end)
};

// grid (chunks, H, B); block = 64 * nkb * wpk threads: wave w owns key block w % nkb (64 keys, lane = key) and every wpk-th query
// tile (32 queries) of the chunk.  For each of its tiles a lane reads its key's 32 d scores and 32 region ids (one 128-byte and one
// 64-byte row) and walks the queries: no cross-lane sums for d vs, three integer LDS atomics per pair for the moments.
__global__ __launch_bounds__(1024) void cpb_region_bwd_kernel(
    const float* __restrict__ dLT, const unsigned short* __restrict__ RID, const float* __restrict__ VS, const float* __restrict__ GQ,
    CpbParams cp, RegionView rv, const unsigned* __restrict__ AMAX, unsigned long long* __restrict__ HIST, unsigned long long* __restrict__ GRAD,
    float* __restrict__ dvs_slab, int N, int J, int H, int NST, int nkb, int wpk, int tiles_per_chunk, int kbits, int shift) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RegionBwdLds& L = *reinterpret_cast<RegionBwdLds*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, hf = lane >> 5;
  const int b = blockIdx.z, h = blockIdx.y, chunk = blockIdx.x;
  const int nthreads = blockDim.x;
  {
    const int nreg = min((int)rv.hdr->n_regions, RG_LCAP);
    for (int i = tid; i < RG_LCAP * 3; i += nthreads) L.hist[i] = 0ull;
    for (int i = tid; i < RG_GRAD; i += nthreads) L.grad[i] = 0ull;
    for (int i = tid; i < nreg; i += nthreads) { const float4 r = rv.reg[i]; L.reg2[i] = make_float2(r.x, r.y); }
    for (int i = tid; i < CH * CH; i += nthreads) { const float w = cp.w2[i]; L.w2r[i] = w; L.w2t[(i & 31) * CH + (i >> 5)] = w; }
  }
  __syncthreads();
  const RegionScale sc = region_scale(*AMAX, kbits);
  const CoopMlp mlp{cp.w1[c * 2], cp.w1[c * 2 + 1], cp.b1[c], cp.b2[c], cp.w3[c], cp.b3[0], L.w2t, L.w2r};
  float big;
  asm("s_mov_b32 %0, 0x71800000" : "=s"(big));

  const int kb = wave % nkb, tslot = wave / nkb;                   // waves beyond nkb * wpk do not exist (block size)
  const int key = kb * 64 + lane;
  const bool kvalid = key < J;
  const int keyc = min(key, J - 1);
  const float vs0 = VS[((size_t)(b * H + h) * J + keyc) * 2], vs1 = VS[((size_t)(b * H + h) * J + keyc) * 2 + 1];
  const int ntq = (N + QT - 1) / QT;
  const int t_begin = chunk * tiles_per_chunk, t_end = min(t_begin + tiles_per_chunk, ntq);
  float dv0 = 0.f, dv1 = 0.f;

  for (int tile = t_begin + tslot; tile < t_end; tile += wpk) {
    const int q0 = tile * QT;
    const int nq = min(QT, N - q0);
    const size_t row = ((size_t)(b * H + h) * NST + q0) * J + (size_t)keyc * 32;     // [B, H, nst / 32, J, 32]: this key's 32 queries of the tile
    float dbr[32];
    unsigned ridw[16];
    {
      const float4* dp = reinterpret_cast<const float4*>(dLT + row);
      const uint4* rp = reinterpret_cast<const uint4*>(RID + row);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float4 t = dp[i]; dbr[4 * i] = t.x; dbr[4 * i + 1] = t.y; dbr[4 * i + 2] = t.z; dbr[4 * i + 3] = t.w; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { const uint4 t = rp[i]; ridw[4 * i] = t.x; ridw[4 * i + 1] = t.y; ridw[4 * i + 2] = t.z; ridw[4 * i + 3] = t.w; }
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      if (q < nq) {                                                 // uniform
        const float gx = GQ[(size_t)(q0 + q) * 2], gy = GQ[(size_t)(q0 + q) * 2 + 1];
        const float d0 = gx - vs0, d1 = gy - vs1;
        const float p0 = slog1p(d0), p1 = slog1p(d1);
        const float s0 = dpos_of<false>(d0, big), s1 = dpos_of<false>(d1, big);
        const unsigned id = kvalid ? ((ridw[q >> 1] >> (16 * (q & 1))) & 0xFFFFu) : 0u;
        const float dbv = kvalid ? dbr[q] : 0.f;
        float a0 = 0.f, a1 = 0.f;
        if (id < (unsigned)RG_LCAP) {
          const float2 a = L.reg2[id];
          a0 = a.x; a1 = a.y;
          atomicAdd(&L.hist[id * 3], (unsigned long long)region_fix(dbv, sc.S));
          atomicAdd(&L.hist[id * 3 + 1], (unsigned long long)region_fix(dbv * p0, sc.S));
          atomicAdd(&L.hist[id * 3 + 2], (unsigned long long)region_fix(dbv * p1, sc.S));
        } else if (id != RG_NONE) {                                 // a region beyond the LDS-resident ones: global memory
          const float4 a = rv.reg[id];
          a0 = a.x; a1 = a.y;
          const double Sg = __longlong_as_double(__double_as_longlong(sc.S) - ((long long)shift << 52));
          atomicAdd(&HIST[id * 3], (unsigned long long)region_fix(dbv, Sg));
          atomicAdd(&HIST[id * 3 + 1], (unsigned long long)region_fix(dbv * p0, Sg));
          atomicAdd(&HIST[id * 3 + 2], (unsigned long long)region_fix(dbv * p1, Sg));
        }
        // pairs without a region: the MLP's own backward for that pair, by the whole wave
        unsigned long long todo = __ballot(kvalid && id == RG_NONE);
        while (todo) {
          const int l = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          const float pp0 = __shfl(p0, l), pp1 = __shfl(p1, l), dbl = __shfl(dbv, l);
          float h1; bool on2;
          (void)coop_mlp_fwd(mlp, pp0, pp1, c, h1, on2);
          // x2 of unit c again (the forward helper returns only its sign): cheap next to the rest
          float x2 = mlp.b2;
#pragma unroll 8
          for (int i = 0; i < CH; ++i) x2 = fmaf(mlp.w2t[i * CH + c], __shfl(h1, i), x2);
          const float g2 = on2 ? mlp.w3 : 0.f;
          float c1 = 0.f;
#pragma unroll 8
          for (int o = 0; o < CH; ++o) c1 = fmaf(mlp.w2r[o * CH + c], __shfl(g2, o), c1);
          c1 = h1 > 0.f ? c1 : 0.f;
          const float dp0 = coop_sum32(c1 * mlp.w1x), dp1 = coop_sum32(c1 * mlp.w1y);
          if (lane == l) { a0 = dp0; a1 = dp1; }
          for (int ii = 0; ii < 16; ++ii) {                         // dW2[out = c][in = 16 hf + ii]
            const int i = 16 * hf + ii;
            const float hi = __shfl(h1, i);
            atomicAdd(&L.grad[c * CH + i], (unsigned long long)region_fix(dbl * g2 * hi, sc.S));
          }
          if (hf == 0) {
            atomicAdd(&L.grad[1024 + 2 * c], (unsigned long long)region_fix(dbl * c1 * pp0, sc.S));
            atomicAdd(&L.grad[1024 + 2 * c + 1], (unsigned long long)region_fix(dbl * c1 * pp1, sc.S));
            atomicAdd(&L.grad[1024 + 64 + c], (unsigned long long)region_fix(dbl * c1, sc.S));
            atomicAdd(&L.grad[1024 + 96 + c], (unsigned long long)region_fix(dbl * g2, sc.S));
            atomicAdd(&L.grad[1024 + 128 + c], (unsigned long long)region_fix(on2 ? dbl * x2 : 0.f, sc.S));
            if (c == 0) atomicAdd(&L.grad[1024 + 160], (unsigned long long)region_fix(dbl, sc.S));
          }
        }
        dv0 = fmaf(-dbv * a0, s0, dv0);
        dv1 = fmaf(-dbv * a1, s1, dv1);
      }
    }
  }
  L.dvs[wave][lane] = make_float2(dv0, dv1);
  __syncthreads();
  // d vs of this chunk: the wpk waves of a key block in a fixed order -> slab [chunk][b, h][J]
  if (wave < nkb && kvalid) {
    float2 sum = L.dvs[wave][lane];
    for (int s2 = 1; s2 < wpk; ++s2) { const float2 t = L.dvs[wave + s2 * nkb][lane]; sum.x += t.x; sum.y += t.y; }
    reinterpret_cast<float2*>(dvs_slab)[((size_t)chunk * gridDim.z * gridDim.y + (size_t)(b * H + h)) * J + key] = sum;
  }
  // moments and direct gradient sums -> global 64-bit accumulators (coarser scale: 2^-shift, rounded)
  const long long half = shift > 0 ? (1ll << (shift - 1)) : 0ll;
  for (int i = tid; i < RG_LCAP * 3; i += nthreads) {
    const long long v = (long long)L.hist[i];
    if (v != 0ll) atomicAdd(&HIST[i], (unsigned long long)((v + half) >> shift));
  }
  for (int i = tid; i < RG_GRAD; i += nthreads) {
    const long long v = (long long)L.grad[i];
    if (v != 0ll) atomicAdd(&GRAD[i], (unsigned long long)((v + half) >> shift));
  }
}

// d vs [(b, h), J, 2] = sum of the chunk slabs in a fixed order
__global__ __launch_bounds__(256) void region_dvs_reduce_kernel(const float2* __restrict__ slab, float2* __restrict__ dVS, size_t n, int chunks) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float2 s = slab[i];
  for (int k = 1; k < chunks; ++k) { const float2 t = slab[(size_t)k * n + i]; s.x += t.x; s.y += t.y; }
  dVS[i] = s;
}

// The six parameter gradients from the region moments M_r = sum d bias (1, p0, p1) - all of them are linear in M:
//   X1_i = W1[i] . (M1, M2) + b1_i M0,  H1_i = d1_i X1_i;   X2_o = W2[o] . H1 + b2_o M0;   G2_o = d2_o w3_o;   C1_i = d1_i sum_o W2[o][i] G2_o
//   dW3_o += d2_o X2_o;  db3 += M0;  db2_o += G2_o M0;  dW2[o][i] += G2_o H1_i;  db1_i += C1_i M0;  dW1[i] += C1_i (M1, M2)
// Stage 1: one workgroup per 32 regions -> partial sums [groups][RG_GRAD] (fp64); stage 2 adds the groups in order.
constexpr int RG_FIN = 32;
__global__ __launch_bounds__(256) void region_final1_kernel(RegionTables t, const unsigned long long* __restrict__ HIST, double* __restrict__ part) {
  __shared__ double M[RG_FIN][3], H1[RG_FIN][CH], X2[RG_FIN][CH], C1[RG_FIN][CH];
  __shared__ unsigned D1[RG_FIN], D2[RG_FIN];
  const int tid = threadIdx.x, r0 = blockIdx.x * RG_FIN;
  const int nreg = (int)t.hdr->n_regions;
  const double* __restrict__ wd = t.wd;
  if (tid < RG_FIN * 3) {
    const int r = r0 + tid / 3;
    M[tid / 3][tid % 3] = r < nreg ? (double)(long long)HIST[(size_t)r * 3 + tid % 3] : 0.0;
  }
  if (tid < RG_FIN) {
    const int r = r0 + tid;
    const unsigned long long k = r < nreg ? t.pat[r] : 0ull;
    D1[tid] = (unsigned)k; D2[tid] = (unsigned)(k >> 32);
  }
  __syncthreads();
  for (int x = tid; x < RG_FIN * CH; x += 256) {
    const int r = x >> 5, i = x & 31;
    const double x1 = wd[WD_W1 + 2 * i] * M[r][1] + wd[WD_W1 + 2 * i + 1] * M[r][2] + wd[WD_B1 + i] * M[r][0];
    H1[r][i] = ((D1[r] >> i) & 1u) ? x1 : 0.0;
  }
  __syncthreads();
  for (int x = tid; x < RG_FIN * CH; x += 256) {
    const int r = x >> 5, o = x & 31;
    double v = wd[WD_B2 + o] * M[r][0];
    for (int i = 0; i < CH; ++i) v = fma(wd[WD_W2 + o * CH + i], H1[r][i], v);
    X2[r][o] = v;
    double cc = 0.0;                                               // C1 of unit i = o
    for (int oo = 0; oo < CH; ++oo)
      if ((D2[r] >> oo) & 1u) cc = fma(wd[WD_W2 + oo * CH + o], wd[WD_W3 + oo], cc);
    C1[r][o] = ((D1[r] >> o) & 1u) ? cc : 0.0;
  }
  __syncthreads();
  double* out = part + (size_t)blockIdx.x * RG_GRAD;
  for (int k = tid; k < RG_GRAD; k += 256) {
    double v = 0.0;
    if (k < 1024) {
      const int o = k >> 5, i = k & 31;
      for (int r = 0; r < RG_FIN; ++r) if ((D2[r] >> o) & 1u) v += H1[r][i];
      v *= wd[WD_W3 + o];
    } else if (k < 1024 + 64) {
      const int i = (k - 1024) >> 1, comp = (k - 1024) & 1;
      for (int r = 0; r < RG_FIN; ++r) v = fma(C1[r][i], M[r][1 + comp], v);
    } else if (k < 1024 + 96) {
      const int i = k - 1088;
      for (int r = 0; r < RG_FIN; ++r) v = fma(C1[r][i], M[r][0], v);
    } else if (k < 1024 + 128) {
      const int o = k - 1120;
      for (int r = 0; r < RG_FIN; ++r) if ((D2[r] >> o) & 1u) v += M[r][0];
      v *= wd[WD_W3 + o];
    } else if (k < 1024 + 160) {
      const int o = k - 1152;
      for (int r = 0; r < RG_FIN; ++r) if ((D2[r] >> o) & 1u) v += X2[r][o];
    } else if (k == 1024 + 160) {
      for (int r = 0; r < RG_FIN; ++r) v += M[r][0];
    }
    out[k] = v;
  }
}
__global__ __launch_bounds__(256) void region_final2_kernel(const double* __restrict__ part, int groups, const unsigned long long* __restrict__ GRAD,
                                                            const unsigned* __restrict__ AMAX, int kbits_global, float* __restrict__ dW1,
                                                            float* __restrict__ db1, float* __restrict__ dW2, float* __restrict__ db2,
                                                            float* __restrict__ dW3, float* __restrict__ db3) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= RG_GRAD) return;
  double v = (double)(long long)GRAD[k];
  for (int g = 0; g < groups; ++g) v += part[(size_t)g * RG_GRAD + k];
  const RegionScale sc = region_scale(*AMAX, kbits_global);
  const float r = (float)(v / sc.S);
  if (k < 1024) dW2[k] = r;
  else if (k < 1024 + 64) dW1[k - 1024] = r;
  else if (k < 1024 + 96) db1[k - 1088] = r;
  else if (k < 1024 + 128) db2[k - 1120] = r;
  else if (k < 1024 + 160) dW3[k - 1152] = r;
  else if (k == 1024 + 160) db3[0] = r;
}

// decisions of a pair for the parity tests: (D1, D2) of its region as two 32-bit words (tests only; pairs without a region: 0, flag)
}  // namespace
