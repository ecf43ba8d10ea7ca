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
#include <type_traits>
#include "deform_common.h"
#include "deform16_types.h"

namespace {

#ifndef SMML_RGN_EXP
#define SMML_RGN_EXP 0                   // measurement variants of the region forward (tests/build_variants.py): 1 no cell gather, 2 no record /
#endif                                   // sub-cell resolution, 3 no score store, 4 no region-id store, 5 no signed logs, 6 no LDS (a, c) read
constexpr int RG_MAX_KEYS = 16384;       // keys per (bag, head) the region entry points accept
constexpr int RG_G = 1024;               // level-0 cells per axis
constexpr int RG_SUB = 8;                // sub-cells per axis of a refined cell
constexpr int RG_SUBC = (RG_SUB + 1) * (RG_SUB + 1);
constexpr int RG_SUBCAP = 16384;         // refined cells (8 x 8 entries each)
constexpr int RG_EDGES = 1 << 15;        // hash slots of the single-kink records (one per pair of adjacent regions)
constexpr unsigned RG_E_SUB0 = 4096u, RG_E_EDGE0 = RG_E_SUB0 + 16384u;   // 16-bit cell codes: [0, 4096) region id | [4096, 20480) refined cell | [20480, 53248) record slot | 0xFFFF evaluate the MLP
constexpr int RG_CANDCAP = 1 << 18;      // cells whose single kink is a layer-1 line, awaiting their check
constexpr int RG_HASH = 16384;           // hash slots of the region patterns (< 2^16: a slot fits the 16-bit fields of a record)
constexpr int RG_RCAP = 4096;            // regions with a dense id; patterns beyond it fall to kind 3
constexpr int RG_LCAP = 2048;            // regions whose (a, c) / moments live in LDS (ids above it: global memory)
constexpr unsigned RG_NONE = 0xFFFFu;    // "no region id": evaluate the MLP
constexpr unsigned long long RG_EMPTY = ~0ull;
constexpr unsigned RG_K_REGION = 0u << 30, RG_K_EDGE = 1u << 30, RG_K_SUB = 2u << 30, RG_K_MLP = 3u << 30, RG_PAYLOAD = 0x3FFFFFFFu;

struct RegionHeader {                    // first 256 bytes of the table buffer
  unsigned n_sub, n_edge, n_cand, n_regions, n_keys, overflow;   // overflow: bit 0 refined cells, 1 records, 2 candidates, 3 hash, 4 regions
  unsigned n_cand1, pad0;                  // candidates among the sub-cells
  float pmax, cs, co, pad;               // cell of p: floor(p * cs + co)
  unsigned stats[6];
};

// byte offsets inside the table buffer (all multiples of 256)
struct RegionLayout {
  size_t hdr, reg, pat, t0h, t1h, edge, t0, t1, ekey, hkey, hid, sublist, cand, klist, corners, subcorners, wd, total;
};
__host__ __device__ inline RegionLayout region_layout() {
  RegionLayout l;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  l.hdr = take(256);
  l.reg = take((size_t)RG_RCAP * 16);                       // float4 {a0, a1, c, 0} per region
  l.pat = take((size_t)RG_RCAP * 8);                        // D1 | D2 << 32
  // what the attention kernels read: 16-bit codes of the cells / sub-cells (2 + 2 MB, of which ~2.7 MB are touched) and the records
  l.t0h = take((size_t)RG_G * RG_G * 2);
  l.t1h = take((size_t)RG_SUBCAP * RG_SUB * RG_SUB * 2);
  l.edge = take((size_t)RG_EDGES * 16);                     // float4 {alpha0, alpha1, beta, bits(neg | pos << 16)} at the record's hash slot
  // build scratch
  l.t0 = take((size_t)RG_G * RG_G * 4);                     // kind << 30 | payload, regions as hash slots until the remap pass
  l.t1 = take((size_t)RG_SUBCAP * RG_SUB * RG_SUB * 4);
  l.ekey = take((size_t)RG_EDGES * 4);                      // neg | pos << 16 (region hash slots) of the record in that slot
  l.hkey = take((size_t)RG_HASH * 8);
  l.hid = take((size_t)RG_HASH * 4);
  l.sublist = take((size_t)RG_SUBCAP * 4);
  l.cand = take((size_t)RG_CANDCAP * 4 * 2);                 // level 0, level 1
  l.klist = take((size_t)RG_HASH * 8 + (size_t)RG_HASH * 4);   // compacted keys, then their slots
  l.corners = take((size_t)(RG_G + 1) * (RG_G + 1) * 8);
  l.subcorners = take((size_t)RG_SUBCAP * RG_SUBC * 8);        // ReLU patterns at the 9 x 9 corners of every refined cell
  l.wd = take(1200 * 8);                                    // the MLP's parameters in fp64
  l.total = o;
  return l;
}
struct RegionTables {                    // device pointers into one table buffer
  RegionHeader* hdr; float4* reg; unsigned long long* pat; unsigned short* t0h; unsigned short* t1h; float4* edge; unsigned* t0; unsigned* t1;
  unsigned* ekey; unsigned long long* hkey;
  unsigned* hid; unsigned* sublist; unsigned* cand; unsigned long long* klist; unsigned* kslot; uint2* corners; uint2* subcorners; double* wd;
};
inline RegionTables region_tables(void* base) {
  const RegionLayout l = region_layout();
  char* b = reinterpret_cast<char*>(base);
  RegionTables t;
  t.hdr = reinterpret_cast<RegionHeader*>(b + l.hdr); t.reg = reinterpret_cast<float4*>(b + l.reg);
  t.pat = reinterpret_cast<unsigned long long*>(b + l.pat); t.t0 = reinterpret_cast<unsigned*>(b + l.t0);
  t.t1 = reinterpret_cast<unsigned*>(b + l.t1); t.edge = reinterpret_cast<float4*>(b + l.edge);
  t.t0h = reinterpret_cast<unsigned short*>(b + l.t0h); t.t1h = reinterpret_cast<unsigned short*>(b + l.t1h); t.ekey = reinterpret_cast<unsigned*>(b + l.ekey);
  t.hkey = reinterpret_cast<unsigned long long*>(b + l.hkey); t.hid = reinterpret_cast<unsigned*>(b + l.hid);
  t.sublist = reinterpret_cast<unsigned*>(b + l.sublist); t.cand = reinterpret_cast<unsigned*>(b + l.cand);
  t.klist = reinterpret_cast<unsigned long long*>(b + l.klist); t.kslot = reinterpret_cast<unsigned*>(b + l.klist + (size_t)RG_HASH * 8);
  t.corners = reinterpret_cast<uint2*>(b + l.corners); t.subcorners = reinterpret_cast<uint2*>(b + l.subcorners);
  t.wd = reinterpret_cast<double*>(b + l.wd);
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
    h.n_sub = h.n_edge = h.n_cand = h.n_regions = h.n_keys = h.overflow = h.n_cand1 = h.pad0 = 0;
    h.pmax = pmax; h.cs = (float)((double)RG_G / (2.0 * (double)pmax)); h.co = (float)(RG_G / 2); h.pad = 0.f;
    for (int k = 0; k < 6; ++k) h.stats[k] = 0;
    *t.hdr = h;
  }
  for (int s = i; s < RG_HASH; s += gridDim.x * blockDim.x) t.hkey[s] = RG_EMPTY;
  for (int s = i; s < RG_EDGES; s += gridDim.x * blockDim.x) t.ekey[s] = 0xFFFFFFFFu;
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

// slot of `key` in the pattern hash (inserted if absent).  A million cells name ~2 000 patterns: the slot is READ first (an L2 read,
// past the CU's L1, which is not coherent) and the compare-and-swap is issued only on an empty slot.
__device__ __forceinline__ unsigned region_hash_insert(unsigned long long* __restrict__ hkey, unsigned long long key) {
  if (key == RG_EMPTY) return RG_NONE;              // the all-ones pattern doubles as the empty marker: such pairs take the MLP path
  unsigned slot = (unsigned)(mix64(key) & (RG_HASH - 1));
  for (int probe = 0; probe < RG_HASH; ++probe) {
    unsigned long long cur = __hip_atomic_load(&hkey[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == RG_EMPTY) {
      cur = atomicCAS(&hkey[slot], RG_EMPTY, key);
      if (cur == RG_EMPTY) return slot;
    }
    if (cur == key) return slot;
    slot = (slot + 1) & (RG_HASH - 1);
  }
  return RG_NONE;
}
// the same for a wave whose neighbouring lanes mostly hold the same key: the first lane of every run inserts, the others then find it
__device__ __forceinline__ unsigned region_hash_insert_runs(unsigned long long* __restrict__ hkey, unsigned long long key, bool active) {
  const unsigned long long prev = __shfl_up(key, 1);
  const bool prev_active = __shfl_up((int)active, 1) != 0;
  const bool leader = active && ((threadIdx.x & 63) == 0 || !prev_active || prev != key);
  unsigned slot = RG_NONE;
  if (leader) slot = region_hash_insert(hkey, key);
  if (active && !leader) slot = region_hash_insert(hkey, key);
  return slot;
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
// one slot of a list per lane that wants one: a single atomic per wave (ballot + prefix count) instead of one per lane on one counter.
// Every lane of the wave must call it.
__device__ __forceinline__ unsigned wave_alloc(unsigned* counter, bool want) {
  const unsigned long long m = __ballot(want);
  if (m == 0ull) return 0u;
  const int lane = threadIdx.x & 63, first = __ffsll((long long)m) - 1;
  unsigned base = 0u;
  if (lane == first) base = atomicAdd(counter, (unsigned)__popcll(m));
  base = __shfl(base, first);
  return base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
}
// The record of a one-kink cell, shared by every cell the same kink segment crosses: two adjacent regions meet in exactly one kink, so
// the pair (negative side, positive side) names the record - a hash of 32 K slots, the slot is the record's 15-bit id.  Every cell of the
// segment computes the same record and may write it (identical values).  `want` lanes get an entry (callable by any subset of lanes).
__device__ __forceinline__ unsigned region_new_edge(RegionTables t, bool want, float4 rec) {
  if (!want) return RG_K_MLP;
  const unsigned key = __float_as_uint(rec.w);
  if ((key & 0xFFFFu) == RG_NONE || (key >> 16) == RG_NONE) return RG_K_MLP;       // a side without a hash slot (hash full)
  unsigned slot = (unsigned)(mix64((unsigned long long)key) & (RG_EDGES - 1));
  for (int probe = 0; probe < RG_EDGES; ++probe) {
    unsigned cur = __hip_atomic_load(&t.ekey[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0xFFFFFFFFu) {
      cur = atomicCAS(&t.ekey[slot], 0xFFFFFFFFu, key);
      if (cur == 0xFFFFFFFFu) { atomicAdd(&t.hdr->n_edge, 1u); cur = key; }
    }
    if (cur == key) { t.edge[slot] = rec; return RG_K_EDGE | slot; }
    slot = (slot + 1) & (RG_EDGES - 1);
  }
  atomicOr(&t.hdr->overflow, 2u);
  return RG_K_MLP;
}
// appends a refined cell (wave-wide; `want` lanes get an entry)
__device__ __forceinline__ unsigned region_new_sub(RegionTables t, bool want, unsigned cell) {
  const unsigned idx = wave_alloc(&t.hdr->n_sub, want);
  if (!want) return RG_K_MLP;
  if (idx >= (unsigned)RG_SUBCAP) { atomicOr(&t.hdr->overflow, 1u); return RG_K_MLP; }
  t.sublist[idx] = cell;
  return RG_K_SUB | idx;
}

// Classification of one cell from the patterns of its four corners (every lane of the wave calls it; `valid` lanes get an entry).
// -> final entry, or RG_K_SUB without payload = "several kinks" (the caller refines or gives up), or RG_PENDING = "one layer-1 line
// crosses it, layer-2 signs equal at the corners": to be checked at the two points where the line leaves the cell (the only places a
// layer-2 unit could still change sign inside it).
constexpr unsigned RG_PENDING = 0xFFFFFFFFu;
__device__ __forceinline__ unsigned region_classify(RegionTables t, bool valid, uint2 c00, uint2 c10, uint2 c01, uint2 c11) {
  const unsigned m1 = (c00.x ^ c10.x) | (c00.x ^ c01.x) | (c00.x ^ c11.x);
  const unsigned m2 = (c00.y ^ c10.y) | (c00.y ^ c01.y) | (c00.y ^ c11.y);
  const bool plain = valid && m1 == 0u && m2 == 0u;
  const unsigned slot = region_hash_insert_runs(t.hkey, (unsigned long long)c00.x | ((unsigned long long)c00.y << 32), plain);
  const bool kink2 = valid && m1 == 0u && __popc(m2) == 1;
  float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
  if (kink2) rec = region_edge_l2(t.wd, c00.x, c00.y, __ffs(m2) - 1, t.hkey);
  const unsigned e_edge = region_new_edge(t, kink2, rec);
  if (plain) return slot == RG_NONE ? RG_K_MLP : (RG_K_REGION | slot);
  if (!valid) return RG_K_MLP;
  if (kink2) return e_edge;
  if (m2 == 0u && __popc(m1) == 1) return RG_PENDING;
  return RG_K_SUB;
}

// the check of a RG_PENDING cell [x0, x0 + h] x [y0, y0 + h] crossed by layer-1 line u: layer-2 patterns where the line leaves it
__device__ __forceinline__ bool region_line_cell_ok(const double* __restrict__ wd, int u, unsigned d2, double x0, double y0, double h) {
  const double wx = wd[WD_W1 + 2 * u], wy = wd[WD_W1 + 2 * u + 1], bb = wd[WD_B1 + u];
  const double cx[4] = {x0, x0 + h, x0 + h, x0}, cy[4] = {y0, y0, y0 + h, y0 + h};     // corners in cyclic order
  double v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = fma(wx, cx[k], fma(wy, cy[k], bb));
  double px[2] = {x0, x0}, py[2] = {y0, y0};
  int n = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int nx = (k + 1) & 3;
    if ((v[k] > 0.0) != (v[nx] > 0.0)) {
      const double s = v[k] / (v[k] - v[nx]);
      const double qx = cx[k] + s * (cx[nx] - cx[k]), qy = cy[k] + s * (cy[nx] - cy[k]);
      if (n == 0) { px[0] = qx; py[0] = qy; } else { px[1] = qx; py[1] = qy; }
      ++n;
    }
  }
  if (n != 2) return false;                           // a line through a corner: let the caller refine / evaluate
  const uint2 ma = region_eval(wd, px[0], py[0]);
  const uint2 mb = region_eval(wd, px[1], py[1]);
  return ma.y == d2 && mb.y == d2;
}

__global__ __launch_bounds__(256) void region_classify0_kernel(RegionTables t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;            // the grid covers the cells exactly
  const int iy = i / RG_G, ix = i - iy * RG_G;
  constexpr int GP = RG_G + 1;
  const bool inner = !(ix == 0 || iy == 0 || ix == RG_G - 1 || iy == RG_G - 1);   // the border also takes every point outside the square
  unsigned e = region_classify(t, inner, t.corners[iy * GP + ix], t.corners[iy * GP + ix + 1], t.corners[(iy + 1) * GP + ix],
                               t.corners[(iy + 1) * GP + ix + 1]);
  const bool pending = e == RG_PENDING;
  const unsigned cidx = wave_alloc(&t.hdr->n_cand, pending);
  bool refine = e == RG_K_SUB;
  if (pending) {
    if (cidx < (unsigned)RG_CANDCAP) { t.cand[cidx] = (unsigned)i; e = RG_K_MLP; }     // region_cand_kernel<0> writes the final entry
    else { atomicOr(&t.hdr->overflow, 4u); refine = true; }
  }
  const unsigned e_sub = region_new_sub(t, refine, (unsigned)i);
  t.t0[i] = refine ? e_sub : e;
}

// level 1: the 9 x 9 corners of every refined cell (one thread per corner), then its 8 x 8 sub-cells (one thread per sub-cell)
__global__ __launch_bounds__(256) void region_subcorners_kernel(RegionTables t) {
  const unsigned nsub = min(t.hdr->n_sub, (unsigned)RG_SUBCAP);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsub * RG_SUBC) return;
  const unsigned blk = i / RG_SUBC, k = i - blk * RG_SUBC;
  const unsigned cell = t.sublist[blk];
  const int iy = cell / RG_G, ix = cell - iy * RG_G, sy = k / (RG_SUB + 1), sx = k - sy * (RG_SUB + 1);
  const double pm = (double)t.hdr->pmax, h = 2.0 * pm / RG_G, hs = h / RG_SUB;
  t.subcorners[i] = region_eval(t.wd, -pm + ix * h + sx * hs, -pm + iy * h + sy * hs);
}
__global__ __launch_bounds__(256) void region_classify1_kernel(RegionTables t) {
  const unsigned nsub = min(t.hdr->n_sub, (unsigned)RG_SUBCAP);
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < nsub * (RG_SUB * RG_SUB);
  const unsigned ic = valid ? i : 0u;
  const unsigned blk = ic / (RG_SUB * RG_SUB), k = ic - blk * (RG_SUB * RG_SUB);
  const int sy = k / RG_SUB, sx = k - sy * RG_SUB;
  const uint2* sc = t.subcorners + (size_t)blk * RG_SUBC;
  uint2 z = make_uint2(0u, 0u);
  const uint2 c00 = nsub ? sc[sy * (RG_SUB + 1) + sx] : z, c10 = nsub ? sc[sy * (RG_SUB + 1) + sx + 1] : z,
              c01 = nsub ? sc[(sy + 1) * (RG_SUB + 1) + sx] : z, c11 = nsub ? sc[(sy + 1) * (RG_SUB + 1) + sx + 1] : z;
  unsigned e = region_classify(t, valid, c00, c10, c01, c11);
  const bool pending = valid && e == RG_PENDING;
  const unsigned cidx = wave_alloc(&t.hdr->n_cand1, pending);
  if (!valid) return;
  if (pending) {
    if (cidx < (unsigned)RG_CANDCAP) t.cand[RG_CANDCAP + cidx] = i;
    else atomicOr(&t.hdr->overflow, 4u);
    e = RG_K_MLP;                                                   // region_cand_kernel<1> writes the final entry
  } else if (e == RG_K_SUB) e = RG_K_MLP;
  t.t1[i] = e;
}

// the cells one layer-1 line crosses: edge record if no layer-2 unit changes sign where the line leaves the cell, else refine
// (level 0) or evaluate the MLP (level 1)
template <int LEVEL>
__global__ __launch_bounds__(256) void region_cand_kernel(RegionTables t) {
  const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned ncand = min(LEVEL ? t.hdr->n_cand1 : t.hdr->n_cand, (unsigned)RG_CANDCAP);
  if (blockIdx.x * blockDim.x >= ncand) return;                     // whole workgroup
  const bool valid = k < ncand;
  const unsigned i = t.cand[LEVEL * RG_CANDCAP + (valid ? k : 0u)];
  const double pm = (double)t.hdr->pmax, h = 2.0 * pm / RG_G, hs = h / RG_SUB;
  uint2 c00, c10, c01, c11;
  double x0, y0, hh;
  if (LEVEL == 0) {
    const int iy = i / RG_G, ix = i - iy * RG_G;
    constexpr int GP = RG_G + 1;
    c00 = t.corners[iy * GP + ix]; c10 = t.corners[iy * GP + ix + 1]; c01 = t.corners[(iy + 1) * GP + ix]; c11 = t.corners[(iy + 1) * GP + ix + 1];
    x0 = -pm + ix * h; y0 = -pm + iy * h; hh = h;
  } else {
    const unsigned blk = i / (RG_SUB * RG_SUB), kk = i - blk * (RG_SUB * RG_SUB);
    const int sy = kk / RG_SUB, sx = kk - sy * RG_SUB;
    const uint2* sc = t.subcorners + (size_t)blk * RG_SUBC;
    c00 = sc[sy * (RG_SUB + 1) + sx]; c10 = sc[sy * (RG_SUB + 1) + sx + 1]; c01 = sc[(sy + 1) * (RG_SUB + 1) + sx]; c11 = sc[(sy + 1) * (RG_SUB + 1) + sx + 1];
    const unsigned cell = t.sublist[blk];
    const int iy = cell / RG_G, ix = cell - iy * RG_G;
    x0 = -pm + ix * h + sx * hs; y0 = -pm + iy * h + sy * hs; hh = hs;
  }
  const unsigned m1 = (c00.x ^ c10.x) | (c00.x ^ c01.x) | (c00.x ^ c11.x);
  const int u = __ffs(m1) - 1;
  const bool ok = valid && region_line_cell_ok(t.wd, u, c00.y, x0, y0, hh);
  float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) rec = region_edge_l1(t.wd, c00.x, c00.y, u, t.hkey);
  const unsigned e_edge = region_new_edge(t, ok, rec);
  const unsigned e_sub = LEVEL == 0 ? region_new_sub(t, valid && !ok, i) : RG_K_MLP;
  if (valid) (LEVEL == 0 ? t.t0 : t.t1)[i] = ok ? e_edge : e_sub;
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
  __shared__ unsigned long long ks[2048];
  const unsigned n = t.hdr->n_keys;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { t.hdr->n_regions = min(n, (unsigned)RG_RCAP); if (n > (unsigned)RG_RCAP) atomicOr(&t.hdr->overflow, 16u); }
  if (blockIdx.x * blockDim.x >= n) return;                                  // whole workgroup
  const unsigned long long k = i < n ? t.klist[i] : 0ull;
  unsigned rank = 0;
  for (unsigned base = 0; base < n; base += 2048) {
    __syncthreads();
    for (unsigned j = threadIdx.x; j < 2048 && base + j < n; j += 256) ks[j] = t.klist[base + j];
    __syncthreads();
    const unsigned m = min(2048u, n - base);
    for (unsigned j = 0; j < m; ++j) rank += (ks[j] < k) ? 1u : 0u;
  }
  if (i >= n || rank >= (unsigned)RG_RCAP) return;                           // hid stays RG_NONE: the MLP path
  t.hid[t.kslot[i]] = rank;
  t.pat[rank] = k;
}
// (a, c) of every region, one wave per region (lane = hidden unit i; the two halves compute the same):
//   c1_i = d1_i sum_o W2[o][i] w3[o] d2_o;  a = sum_i c1_i W1[i];  c = sum_i c1_i b1[i] + sum_o d2_o w3[o] b2[o] + b3
__global__ __launch_bounds__(256) void region_coef_kernel(RegionTables t) {
  const unsigned r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= t.hdr->n_regions) return;
  const int i = threadIdx.x & 31;
  const double* __restrict__ wd = t.wd;
  const unsigned long long k = t.pat[r];
  const unsigned d1 = (unsigned)k, d2 = (unsigned)(k >> 32);
  double c1 = 0.0;
  for (int o = 0; o < CH; ++o)
    if ((d2 >> o) & 1u) c1 = fma(wd[WD_W2 + o * CH + i], wd[WD_W3 + o], c1);
  c1 = ((d1 >> i) & 1u) ? c1 : 0.0;
  double a0 = c1 * wd[WD_W1 + 2 * i], a1 = c1 * wd[WD_W1 + 2 * i + 1];
  double c = fma(c1, wd[WD_B1 + i], ((d2 >> i) & 1u) ? wd[WD_W3 + i] * wd[WD_B2 + i] : 0.0);
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) { a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); c += __shfl_xor(c, off); }
  if ((threadIdx.x & 63) == 0) t.reg[r] = make_float4((float)a0, (float)a1, (float)(c + wd[WD_B3]), 0.f);
}
// hash slots -> dense ids: the 16-bit cell codes the attention kernels read, and the region fields of the records
__device__ __forceinline__ unsigned short region_code(const RegionTables& t, unsigned e) {
  const unsigned kind = e >> 30, pay = e & RG_PAYLOAD;
  if (kind == 0u) return (unsigned short)t.hid[pay];                         // dense id, or RG_NONE = 0xFFFF
  if (kind == 1u) return (unsigned short)(RG_E_EDGE0 + pay);
  if (kind == 2u) return (unsigned short)(RG_E_SUB0 + pay);
  return (unsigned short)RG_NONE;
}
__global__ __launch_bounds__(256) void region_remap_kernel(RegionTables t) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n0 = (size_t)RG_G * RG_G, n1 = (size_t)min(t.hdr->n_sub, (unsigned)RG_SUBCAP) * (RG_SUB * RG_SUB);
  if (i < n0) t.t0h[i] = region_code(t, t.t0[i]);
  else if (i < n0 + n1) t.t1h[i - n0] = region_code(t, t.t1[i - n0]);
  else if (i < n0 + n1 + RG_EDGES) {
    const size_t k = i - n0 - n1;
    const unsigned b = t.ekey[k];
    if (b != 0xFFFFFFFFu) {
      float4 r = t.edge[k];
      r.w = __uint_as_float(t.hid[b & 0xFFFFu] | (t.hid[b >> 16] << 16));
      t.edge[k] = r;
    }
  }
}

static int region_build_launch(CpbParams cp, float pmax, void* tables, hipStream_t st) {
  const RegionTables t = region_tables(tables);
  hipLaunchKernelGGL(region_prep_kernel, dim3(64), dim3(256), 0, st, cp, pmax, t);
  constexpr int GP = RG_G + 1;
  hipLaunchKernelGGL(region_corners_kernel, dim3((GP * GP + 255) / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_classify0_kernel, dim3(RG_G * RG_G / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_cand_kernel<0>, dim3(RG_CANDCAP / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_subcorners_kernel, dim3((RG_SUBCAP * RG_SUBC + 255) / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_classify1_kernel, dim3(RG_SUBCAP * RG_SUB * RG_SUB / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_cand_kernel<1>, dim3(RG_CANDCAP / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_compact_kernel, dim3(RG_HASH / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_rank_kernel, dim3(RG_HASH / 256), dim3(256), 0, st, t);
  hipLaunchKernelGGL(region_coef_kernel, dim3(RG_RCAP / 4), dim3(256), 0, st, t);
  const size_t nmax = (size_t)RG_G * RG_G + (size_t)RG_SUBCAP * RG_SUB * RG_SUB + RG_EDGES;
  hipLaunchKernelGGL(region_remap_kernel, dim3((unsigned)((nmax + 255) / 256)), dim3(256), 0, st, t);
  return 0;
}


// ------------------------------------------------------------------------------------------------
// per-pair pieces shared by the region kernels
// ------------------------------------------------------------------------------------------------
struct RegionView {                      // what a region kernel reads of the tables
  const RegionHeader* hdr; const unsigned short* t0; const unsigned short* t1; const float4* edge; const float4* reg;
};
inline RegionView region_view(void* base) {
  const RegionTables t = region_tables(base);
  return RegionView{t.hdr, t.t0h, t.t1h, t.edge, t.reg};
}
__device__ __forceinline__ int region_cell(float u) { return min(max((int)u, 0), RG_G - 1); }
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL load (vmcnt(0)), which would
// drain the table gathers and the K / V prefetch the forward keeps in flight across its barriers
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// first-level entry address of p (cell coordinates in cx, cy; u = p cs + co)
__device__ __forceinline__ unsigned region_cell_index(float u0, float u1, int& cx, int& cy) {
  cx = region_cell(u0); cy = region_cell(u1);
  return (unsigned)(cy * RG_G + cx);
}
// entry -> region id (RG_NONE: evaluate the MLP).  e: first-level entry of the pair's cell.  The two common kinds (region: 92 %
// of the pairs, one kink: 8 %) take no branch - every lane reads a kink record (record 0 for the lanes that need none) and selects;
// refined cells (0.7 %) branch to their sub-cell's entry.
__device__ __forceinline__ unsigned region_side(const float4 r, float p0, float p1) {
  const float g = fmaf(r.x, p0, fmaf(r.y, p1, r.z));
  const unsigned b = __float_as_uint(r.w);
  return g > 0.f ? (b >> 16) : (b & 0xFFFFu);
}
__device__ __forceinline__ unsigned region_resolve(const RegionView& rv, unsigned e, float p0, float p1, float u0, float u1, int cx, int cy) {
  const unsigned es = e - RG_E_EDGE0;                       // record slot if e codes a record
  unsigned rid = e;                                         // region ids and 0xFFFF (= RG_NONE) pass through
  if (es < (unsigned)RG_EDGES) rid = region_side(rv.edge[es], p0, p1);    // ~8 % of the lanes: the others issue no request
  if (e - RG_E_SUB0 < RG_E_EDGE0 - RG_E_SUB0) {             // refined cell: the sub-cell's code
    const int sx = min(max((int)((u0 - (float)cx) * (float)RG_SUB), 0), RG_SUB - 1);
    const int sy = min(max((int)((u1 - (float)cy) * (float)RG_SUB), 0), RG_SUB - 1);
    const unsigned e1 = rv.t1[(size_t)(e - RG_E_SUB0) * (RG_SUB * RG_SUB) + sy * RG_SUB + sx];
    const unsigned es1 = e1 - RG_E_EDGE0;
    rid = e1;
    if (es1 < (unsigned)RG_EDGES) rid = region_side(rv.edge[es1], p0, p1);
  }
  return rid;
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
// T = float: the fp32-grade core (fp16 hi / lo split products, fp32 scores saved).  T = __bf16 / _Float16: the 16-bit compute mode of
// deform_attn16.hip (single-term T operands, scores saved as fp16 and the forward's own softmax continued on the ROUNDED scores) - the
// position bias is the same fp32 lookup in every mode.
template <typename T> struct RegionScore { typedef u16 type; };
template <> struct RegionScore<float> { typedef float type; };
template <bool SAVE, typename T = float>
__global__ __launch_bounds__(256, 2) void deform_region_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ VS,
    const float* __restrict__ GQ, CpbParams cp, RegionView rv, float* __restrict__ O, float* __restrict__ LSE,
    typename RegionScore<T>::type* __restrict__ LT, unsigned short* __restrict__ RID, int N, int J, int H, int NST, float scale, DropCfg dc_in,
    int lcap) {
  constexpr bool F32 = std::is_same<T, float>::value;
  typedef typename std::conditional<F32, _Float16, T>::type T16;        // element type of the single-term operands (unused for F32)
  typedef typename Vec8<T16>::type vec8;
  const DropCfg dc = drop_resolve(dc_in);
  __shared__ __attribute__((aligned(16))) _Float16 Kp[2][KT * FRLD];         // K tile, fp16 hi / lo planes, row image (A operand of S^T)
  __shared__ __attribute__((aligned(16))) _Float16 Vp[2][KT * FTLD];         // V tile, hi / lo planes, read transposed (A operand of O^T)
  __shared__ __attribute__((aligned(16))) float4 regl[RG_LCAP];              // (a0, a1, c) of the regions
  __shared__ float w2t[CH * CH];                                             // W2 transposed (the MLP path)
  __shared__ float vsl[2][KT][2];                                            // sample positions of this tile's and the next tile's keys
  __shared__ __attribute__((aligned(16))) unsigned short ridl[WAVES][KT][QT + 8];   // per wave: the tile's region ids [key][query] (80-byte rows)

  const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (QT * WAVES) + wave * QT;
  const int HD = H * DH;
  const bool qvalid = (q0 + c) < N;
  const int qi = qvalid ? (q0 + c) : (N - 1);
  const float* VSb = VS + (size_t)(b * H + h) * J * 2;                       // one head per offset group: group = head

  {
    const int nreg = min((int)rv.hdr->n_regions, lcap);       // lcap <= RG_LCAP: regions resident in LDS (tests lower it)
    for (int i = tid; i < nreg; i += 256) regl[i] = rv.reg[i];
    for (int i = tid; i < CH * CH; i += 256) w2t[(i & 31) * CH + (i >> 5)] = cp.w2[i];      // i = o * 32 + in
    if (tid < KT) {
      const int key = min(tid, J - 1);
      vsl[0][tid][0] = VSb[(size_t)key * 2];
      vsl[0][tid][1] = VSb[(size_t)key * 2 + 1];
    }
  }
  __syncthreads();
  const float cs = rv.hdr->cs, co = rv.hdr->co;
  const CoopMlp mlp{cp.w1[c * 2], cp.w1[c * 2 + 1], cp.b1[c], cp.b2[c], cp.w3[c], cp.b3[0], w2t, nullptr};

  // scaled Q of this lane's query as the B operand of S^T = K . Q^T: k-step st holds d = 16 st + 8 hf + j, fp16 hi / lo
  half8 qh[4], ql[4];
  vec8 qf[4];
  {
    const float* qp = Q + ((size_t)b * N + qi) * HD + h * DH + hf * 8;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const float4 t0 = *reinterpret_cast<const float4*>(qp + 16 * st), t1 = *reinterpret_cast<const float4*>(qp + 16 * st + 4);
      const float x8[8] = {t0.x * scale, t0.y * scale, t0.z * scale, t0.w * scale, t1.x * scale, t1.y * scale, t1.z * scale, t1.w * scale};
      if constexpr (F32) split8(x8, qh[st], ql[st]);
      else qf[st] = cvt8<T16>(x8);
    }
  }
  T16* const K16 = reinterpret_cast<T16*>(&Kp[0][0]);                  // 16-bit modes: one plane each
  T16* const V16 = reinterpret_cast<T16*>(&Vp[0][0]);
  const int trq = (lane & 15) >> 2, trc = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);      // transposed-read lane map
  const float gq0 = GQ[(size_t)qi * 2], gq1 = GQ[(size_t)qi * 2 + 1];

  floatx16 oacc0 = {0}, oacc1 = {0};
  float m_run = -INFINITY, l_run = 0.f;
  const unsigned long long drop_row = dc.seed + ((unsigned long long)(b * H + h) * N + qi) * ((J + 1) >> 1);   // dropout counter of the row's first pair
  const float* Kb = K + (size_t)b * J * HD + h * DH;
  const float* Vb = V + (size_t)b * J * HD + h * DH;
  typename RegionScore<T>::type* LTb = LT ? LT + ((size_t)(b * H + h) * NST + q0) * J : nullptr;    // this wave's [J][32] block (layout: deform_attn_fwd_kernel)
  unsigned short* RIDb = RID ? RID + ((size_t)(b * H + h) * NST + q0) * J : nullptr;

  // K / V rows of a tile travel global -> registers (one tile ahead) -> fp16 hi / lo images in LDS
  float4 kreg[2], vreg[2];
  float2 vsn = make_float2(0.f, 0.f);
  auto fetch_kv = [&](int jn) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = jn + (tid >> 4) + 16 * i, d4 = (tid & 15) * 4;
      kreg[i] = make_float4(0.f, 0.f, 0.f, 0.f); vreg[i] = kreg[i];
      if (key < J) {
        kreg[i] = *reinterpret_cast<const float4*>(Kb + (size_t)key * HD + d4);
        vreg[i] = *reinterpret_cast<const float4*>(Vb + (size_t)key * HD + d4);
      }
    }
    if (tid < KT) vsn = *reinterpret_cast<const float2*>(VSb + (size_t)min(jn + KT + tid, J - 1) * 2);
  };
  fetch_kv(0);
  const int ntiles = (J + KT - 1) / KT;
  // Position bias of this lane's 16 (key, query) pairs of a tile, register r <-> key acc_row(r, hf).  The lookup is a chain of up to three
  // dependent gathers (cell code -> sub-cell code of a refined cell -> kink record); each link is issued a program phase ahead of its
  // use so that its L2 round trip passes behind other work, in STAGES - all gathers of a link together (a dependent gather inside a
  // per-pair branch exposes one round trip per pair: measured 2 ms of this kernel's first 3.2):
  //   step 1  (signed-log offsets, the 16 cell codes) for tile kt + 1 behind tile kt's score stores - in flight during the softmax;
  //   step 2a (sub-cell codes of the refined cells) for tile kt + 1 behind tile kt's softmax - in flight during the P V products;
  //   step 2b (kink records), four pairs at a time and double-buffered: the first group at the top of the tile - in flight during the
  //           staging of K / V, the barriers and the S^T products -, group g + 1 while group g is consumed.
  float p0[16], p1[16];
  unsigned ent[16];
  auto step1 = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float2 vv = *reinterpret_cast<const float2*>(&vsl[buf][acc_row(r, hf)][0]);
#if SMML_RGN_EXP == 5
      p0[r] = (gq0 - vv.x) * 0.5f; p1[r] = (gq1 - vv.y) * 0.5f;
#else
      p0[r] = slog1p(gq0 - vv.x);
      p1[r] = slog1p(gq1 - vv.y);
#endif
      int cx, cy;
#if SMML_RGN_EXP == 1
      ent[r] = region_cell_index(fmaf(p0[r], cs, co), fmaf(p1[r], cs, co), cx, cy) & 1023u;
#else
      ent[r] = rv.t0[region_cell_index(fmaf(p0[r], cs, co), fmaf(p1[r], cs, co), cx, cy)];
#endif
    }
  };
  // Step 2a: refined cells (0.7 % of the pairs) take their sub-cell's code, loaded into the cell code's own register (a third of the
  // (wave, pair) steps have a lane that takes one; none is consumed here).
  auto step2a = [&]() {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned e = ent[r];
#if SMML_RGN_EXP != 2 && SMML_RGN_EXP != 7
      if (e - RG_E_SUB0 < RG_E_EDGE0 - RG_E_SUB0) {
        // sub-cell of p: the low bits of floor(RG_SUB u) clamped to the grid - the same cell as clamp(floor(RG_SUB (u - cell)), 0, RG_SUB - 1)
        // of the table build (RG_SUB u is exact: a power-of-two multiple; inside the grid u - cell is exact too, outside both clamp to
        // the border sub-cell), in half the instructions
        const int X = min(max((int)fmaf(p0[r], cs * (float)RG_SUB, co * (float)RG_SUB), 0), RG_G * RG_SUB - 1);
        const int Y = min(max((int)fmaf(p1[r], cs * (float)RG_SUB, co * (float)RG_SUB), 0), RG_G * RG_SUB - 1);
        ent[r] = rv.t1[(e - RG_E_SUB0) * (RG_SUB * RG_SUB) + (Y & (RG_SUB - 1)) * RG_SUB + (X & (RG_SUB - 1))];
      }
#endif
    }
  };
  // Step 2b: the records of the one-kink cells (8 % of the pairs) - every lane reads one per pair (record 0 where it needs none: one
  // address, no traffic), four pairs at a time, double-buffered.
  float4 rec[2][4];
  auto rec_issue = [&](int g, float4 (&dst)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned es = ent[4 * g + i] - RG_E_EDGE0;
#if SMML_RGN_EXP == 8
      dst[i] = make_float4(__uint_as_float(es), 1.f, 0.5f, __uint_as_float(0x00010002u));
#else
      dst[i] = rv.edge[es < (unsigned)RG_EDGES ? es : 0u];
#endif
    }
  };
  step1(0);
  step2a();
  for (int kt = 0; kt < ntiles; ++kt) {
    const int j0 = kt * KT;
    rec_issue(0, rec[0]);                      // (the sub-cell codes were requested in front of the previous tile's P V products)
    __builtin_amdgcn_sched_barrier(0);
    lds_barrier();                                     // every wave is done with the previous tile's K / V images
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = (tid >> 4) + 16 * i, d4 = (tid & 15) * 4;
      if constexpr (F32) {
        uint2v hi, lo;
        split4_h2(kreg[i], hi, lo);
        *reinterpret_cast<uint2v*>(&Kp[0][key * FRLD + d4]) = hi; *reinterpret_cast<uint2v*>(&Kp[1][key * FRLD + d4]) = lo;
        split4_h2(vreg[i], hi, lo);
        *reinterpret_cast<uint2v*>(&Vp[0][key * FTLD + d4]) = hi; *reinterpret_cast<uint2v*>(&Vp[1][key * FTLD + d4]) = lo;
      } else {
        *reinterpret_cast<uint2v*>(&K16[key * FRLD + d4]) = pack4<T16>(kreg[i]);
        *reinterpret_cast<uint2v*>(&V16[key * FTLD + d4]) = pack4<T16>(vreg[i]);
      }
    }
    if (tid < KT) { vsl[(kt + 1) & 1][tid][0] = vsn.x; vsl[(kt + 1) & 1][tid][1] = vsn.y; }     // the NEXT tile's sample positions
    lds_barrier();

    // S^T[key, query] = K . (scale Q)^T
    floatx16 s = {0};
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int o = c * FRLD + 16 * st + 8 * hf;
      if constexpr (F32) {
        const half8 kh = *reinterpret_cast<const half8*>(&Kp[0][o]), kl = *reinterpret_cast<const half8*>(&Kp[1][o]);
        s = mfma16(kl, qh[st], s);
        s = mfma16(kh, ql[st], s);
        s = mfma16(kh, qh[st], s);
      } else {
        s = mma(*reinterpret_cast<const vec8*>(&K16[o]), qf[st], s);
      }
    }

    const int nk = min(KT, J - j0);
    unsigned spec = 0u;                        // bit r: pair r has no LDS-resident region (no region at all, or one beyond RG_LCAP)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g < 3) rec_issue(g + 1, rec[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g + i;
        const unsigned e = ent[r];
#if SMML_RGN_EXP == 2
        const unsigned id = e & 1023u;
#else
        // side of the kink for every lane (record 0 where the code is no record), merged by a bit mask: some lane of the wave needs the
        // record in 99 % of the steps, so a branch around the three multiply-adds never skips them - and the compiler would move the
        // record's load into that branch
        const unsigned em = (e - RG_E_EDGE0) < (unsigned)RG_EDGES ? 0xFFFFFFFFu : 0u;
        const unsigned id = (region_side(rec[g & 1][i], p0[r], p1[r]) & em) | (e & ~em);    // region ids and 0xFFFF pass through
#endif
        // (a, c) of the region from LDS, no branch.  Ids beyond the LDS-resident regions and "no region" (~1e-4 of the pairs) read some
        // entry and add nothing here; `spec` tells the wave afterwards whether it has such a pair at all
        const bool inl = id < (unsigned)lcap;
#if SMML_RGN_EXP == 6
        const float4 ac = make_float4(__uint_as_float(id), 0.5f, 0.25f, 0.f);
#else
        const float4 ac = regl[id & (unsigned)(RG_LCAP - 1)];
#endif
        const float bias = fmaf(ac.x, p0[r], fmaf(ac.y, p1[r], ac.z));
        spec |= (!inl && acc_row(r, hf) < nk) ? (1u << r) : 0u;
        ridl[wave][acc_row(r, hf)][c] = (unsigned short)id;
        s[r] += inl ? bias : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    unsigned nonemask = 0u;                    // bit r: pair r has no region (evaluates the MLP below)
    if (__ballot(spec != 0u)) {                // a tenth of the (wave, tile) steps: which pairs, from the wave's id image
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned idr = ridl[wave][acc_row(r, hf)][c];
        const bool kin = acc_row(r, hf) < nk;
        if (idr == RG_NONE) nonemask |= kin ? (1u << r) : 0u;
        else if (idr >= (unsigned)lcap) {      // a region beyond the LDS-resident ones (none for up to RG_LCAP regions): from global memory
          const float4 ac = rv.reg[idr];
          s[r] += fmaf(ac.x, p0[r], fmaf(ac.y, p1[r], ac.z));
        }
      }
    }
    if (SAVE) {
      // the tile's region ids leave as whole rows: [key][32 queries] is 64 bytes per key - a lane takes the 16 ids of its (key, half of
      // the queries) from the wave's LDS image and stores them with two 16-byte stores (instead of 16 two-byte stores per lane)
      wave_lds_fence();
      const int key = lane >> 1, qh16 = (lane & 1) * 16;
      const uint4 w0 = *reinterpret_cast<const uint4*>(&ridl[wave][key][qh16]), w1 = *reinterpret_cast<const uint4*>(&ridl[wave][key][qh16 + 8]);
      if (key < nk && SMML_RGN_EXP != 4) {
        uint4* dst = reinterpret_cast<uint4*>(RIDb + (size_t)(j0 + key) * 32 + qh16);
        dst[0] = w0;
        dst[1] = w1;
      }
      wave_lds_fence();
    }
    // pairs without a region (~1e-4 of all): the MLP itself, one pair at a time, by the whole wave - outside the unrolled loops
    for (unsigned long long todo = __ballot(nonemask != 0u); todo; todo &= todo - 1) {
      const int l = __ffsll((long long)todo) - 1;
      const float g0 = __shfl(gq0, l), g1 = __shfl(gq1, l);
      for (unsigned m = (unsigned)__shfl((int)nonemask, l); m; m &= m - 1) {
        const int r = __ffs((int)m) - 1;
        const float2 vv = *reinterpret_cast<const float2*>(&vsl[kt & 1][acc_row(r, l >> 5)][0]);
        float h1; bool on2;
        const float v = coop_mlp_fwd(mlp, slog1p(g0 - vv.x), slog1p(g1 - vv.y), c, h1, on2);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) s[rr] += (rr == r && lane == l) ? v : 0.f;
      }
    }
    if (nk < KT) {                            // last tile (uniform): keys past J take no part
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = acc_row(r, hf) < nk ? s[r] : -INFINITY;
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    unsigned keepbits = 0xFFFFu;              // dropout decisions of this lane's 16 keys (bit r)
    if (dc.thresh) {
      // z of the lane's first pair of the tile, pinned in a register pair: the eight pairs are compile-time steps from it (left to itself
      // the compiler hoists eight loop-invariant 64-bit sums out of the tile loop and spills them)
      unsigned long long z0 = drop_row + (unsigned)((j0 >> 1) + 2 * hf);
      asm volatile("" : "+v"(z0));
      keepbits = 0u;
#pragma unroll
      for (int r = 0; r < 16; r += 2) keepbits |= drop_keep2_z(dc, z0 + (unsigned)(acc_row(r, 0) >> 1)) << r;
    }
    if constexpr (SAVE && F32) {              // rows are padded to whole workgroup tiles: lanes past N write padding
      if (dc.thresh) {
        tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (nk == KT || acc_row(r, hf) < nk) s[r] = stash_keep_trunc(s[r], (keepbits >> r) & 1u);      // finite scores only
          tmax = fmaxf(tmax, s[r]);
        }
      }
      if (nk == KT) {                         // interior tile (uniform): no bounds branches around the stores
#pragma unroll
        for (int r = 0; r < 16; ++r) if (SMML_RGN_EXP != 3) LTb[(size_t)(j0 + acc_row(r, hf)) * 32 + c] = s[r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = acc_row(r, hf);
          if (key < nk && SMML_RGN_EXP != 3) LTb[(size_t)(j0 + key) * 32 + c] = s[r];
        }
      }
    }
    if constexpr (SAVE && !F32) {
      // 16-bit modes (deform16_fwd_kernel's rule): the scores are rounded to fp16 for storage, the keep decision replaces the stored
      // score's lowest bit, and the forward's own softmax continues on the stored values - forward and backward agree on the
      // probabilities
      if (nk == KT) {
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
      tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    }
    // The next tile's K / V rows (and the sample positions of the tile after it) and its cell codes: in flight during the softmax, the
    // P V products and the top of the next tile.  Vector-memory operations - loads AND stores - retire in the order of issue, and a wait
    // counts the operations behind the one it waits for:
    //   * issued in front of the lookup stages the prefetch would be waited for with the gathers (behind branches the compiler has to
    //     assume the fewest operations in between; measured: 0.8 ms of this kernel);
    //   * issued in front of the score stores, the stores would be the youngest operations at the top of the next tile and its first
    //     wait (for the cell codes) would be a wait for their completion as well (measured: 0.1 ms).
    fetch_kv(j0 + KT);
    if (kt + 1 < ntiles) step1((kt + 1) & 1);  // (the next tile's sample positions were staged above)
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
    if (kt + 1 < ntiles) step2a();             // the next tile's sub-cell codes: in flight during the P V products
    __builtin_amdgcn_sched_barrier(0);

    // O^T[d, query] += V^T . P^T
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float p8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) p8[j] = s[8 * kb + j];
      const int ro = (16 * kb + 4 * hf + trq) * FTLD + trc;
      if constexpr (F32) {
        half8 ph, pl;
        split8(p8, ph, pl);
        const half8 vh0 = lds_frag_tr_h(&Vp[0][ro], &Vp[0][ro + 8 * FTLD]), vl0 = lds_frag_tr_h(&Vp[1][ro], &Vp[1][ro + 8 * FTLD]);
        const half8 vh1 = lds_frag_tr_h(&Vp[0][ro + 32], &Vp[0][ro + 32 + 8 * FTLD]), vl1 = lds_frag_tr_h(&Vp[1][ro + 32], &Vp[1][ro + 32 + 8 * FTLD]);
        oacc0 = mfma16(vl0, ph, oacc0); oacc0 = mfma16(vh0, pl, oacc0); oacc0 = mfma16(vh0, ph, oacc0);
        oacc1 = mfma16(vl1, ph, oacc1); oacc1 = mfma16(vh1, pl, oacc1); oacc1 = mfma16(vh1, ph, oacc1);
      } else {
        const vec8 pt = cvt8<T16>(p8);
        oacc0 = mma(frag_tr<T16>(&V16[ro], &V16[ro + 8 * FTLD]), pt, oacc0);
        oacc1 = mma(frag_tr<T16>(&V16[ro + 32], &V16[ro + 32 + 8 * FTLD]), pt, oacc1);
      }
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

// grid (chunks * key groups, H, B); block = 64 * nkbg * wpk threads (<= 768), nkbg = key blocks of one group (all nkb = ceil(J / 64) of
// them up to 768 keys; more keys: groups of <= 12 key blocks, each group its own workgroups): wave w of group g owns the keys
// lane * nkb + g * nkbg + (w % nkbg) (lane = key; keys of one
// wave are nkb apart, so that its lanes fall into different regions: no same-address serialisation of the LDS adds) and every wpk-th
// query tile (32 queries) of the chunk.  For each of its tiles a lane reads its key's 32 d scores and 32 region ids (one 128-byte and
// one 64-byte row, the next tile's rows and query positions in flight meanwhile) and walks the queries: no cross-lane sums for d vs; the
// three moments of a RUN of queries in the same region (~4 on the query grid) and the run's two d vs sums are kept in registers and go to
// the region's LDS accumulators with three 64-bit integer adds when the region changes.  The query is the same for all lanes: its position
// comes from a lane of the tile's position register into scalar registers, the second coordinate's log / reciprocal only when it changes.
// DS = float: the fp32-grade core's d scores; DS = u16: the bf16 d scores of the 16-bit modes (deform16_bwd_dq_kernel).
template <typename DS>
__global__ __launch_bounds__(768) void cpb_region_bwd_kernel(
    const DS* __restrict__ dLT, const unsigned short* __restrict__ RID, const float* __restrict__ VS, const float* __restrict__ GQ,
    CpbParams cp, RegionView rv, const unsigned* __restrict__ AMAX, unsigned long long* __restrict__ HIST, unsigned long long* __restrict__ GRAD,
    float* __restrict__ dvs_slab, int N, int J, int H, int NST, int nkb, int nkbg, int chunks, int wpk, int tiles_per_chunk, int kbits, int shift,
    int lcap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RegionBwdLds& L = *reinterpret_cast<RegionBwdLds*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, hf = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform in a scalar register: tile indices, bounds and the query positions stay scalar
  const int b = blockIdx.z, h = blockIdx.y, chunk = blockIdx.x % chunks, grp = blockIdx.x / chunks;
  const int nthreads = blockDim.x;
  {
    const int nreg = min((int)rv.hdr->n_regions, lcap);       // lcap <= RG_LCAP: regions with LDS accumulators (tests lower it)
    for (int i = tid; i < RG_LCAP * 3; i += nthreads) L.hist[i] = 0ull;
    for (int i = tid; i < RG_GRAD; i += nthreads) L.grad[i] = 0ull;
    for (int i = tid; i < nreg; i += nthreads) { const float4 r = rv.reg[i]; L.reg2[i] = make_float2(r.x, r.y); }
    for (int i = tid; i < CH * CH; i += nthreads) { const float w = cp.w2[i]; L.w2r[i] = w; L.w2t[(i & 31) * CH + (i >> 5)] = w; }
  }
  __syncthreads();
  const RegionScale sc = region_scale(*AMAX, kbits);
  const double Sg = __longlong_as_double(__double_as_longlong(sc.S) - ((long long)shift << 52));     // scale of the global accumulators
  const CoopMlp mlp{cp.w1[c * 2], cp.w1[c * 2 + 1], cp.b1[c], cp.b2[c], cp.w3[c], cp.b3[0], L.w2t, L.w2r};
  float big;
  asm("s_mov_b32 %0, 0x71800000" : "=s"(big));

  // (the division runs on the vector unit: back to scalar registers)
  const int kb = grp * nkbg + __builtin_amdgcn_readfirstlane(wave % nkbg), tslot = __builtin_amdgcn_readfirstlane(wave / nkbg);   // waves beyond nkbg * wpk do not exist (block size)
  const int key = lane * nkb + kb;
  const bool kvalid = key < J && kb < nkb;                          // (the last key group may have key blocks to spare: kb >= nkb would alias lane + 1's keys)
  const int keyc = min(key, J - 1);
  const float vs0 = VS[((size_t)(b * H + h) * J + keyc) * 2], vs1 = VS[((size_t)(b * H + h) * J + keyc) * 2 + 1];
  const int ntq = (N + QT - 1) / QT;
  const int t_begin = chunk * tiles_per_chunk, t_end = min(t_begin + tiles_per_chunk, ntq);
  float dv0 = 0.f, dv1 = 0.f;
  // The current run of this lane: its region, the three moment sums of d bias and the two sums of d bias * d p / d offset (d vs of a run is
  // -slope of the region * that sum: the slope is read once per run, not per pair).  cur >= RG_NONE: no run (start, pair without a region).
  unsigned cur = ~0u;
  float r0 = 0.f, r1 = 0.f, r2 = 0.f, u0 = 0.f, u1 = 0.f;
  typedef __attribute__((address_space(3))) unsigned long long lds_u64;
  typedef __attribute__((address_space(1))) unsigned long long glb_u64;
  lds_u64* const hist_l = (lds_u64*)L.hist;
  glb_u64* const hist_g = (glb_u64*)HIST;
  auto flush = [&]() {
    if (cur < (unsigned)lcap) {
      const float2 a = L.reg2[cur];
      dv0 = fmaf(-a.x, u0, dv0); dv1 = fmaf(-a.y, u1, dv1);
      lds_u64* hp = hist_l + cur * 3;
      __hip_atomic_fetch_add(hp, (unsigned long long)region_fix(r0, sc.S), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(hp + 1, (unsigned long long)region_fix(r1, sc.S), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(hp + 2, (unsigned long long)region_fix(r2, sc.S), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (cur < (unsigned)RG_NONE) {                           // a region beyond the LDS-resident ones: global memory
      const float4 a = rv.reg[cur];
      dv0 = fmaf(-a.x, u0, dv0); dv1 = fmaf(-a.y, u1, dv1);
      glb_u64* hp = hist_g + cur * 3;
      __hip_atomic_fetch_add(hp, (unsigned long long)region_fix(r0, Sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(hp + 1, (unsigned long long)region_fix(r1, Sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(hp + 2, (unsigned long long)region_fix(r2, Sg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  constexpr bool DS32 = std::is_same<DS, float>::value;
  float4 dbn[DS32 ? 8 : 4];                                         // (16-bit d scores: 8 per 16-byte word)
  uint4 ridn[4];
  float2 gqn;                                                       // lane c: position of query c of the tile (one coalesced load per tile)
  auto fetch = [&](int tile) {
    const size_t row = ((size_t)(b * H + h) * NST + (size_t)tile * QT) * J + (size_t)keyc * 32;   // [B, H, nst / 32, J, 32]: this key's 32 queries of the tile
    const float4* dp = reinterpret_cast<const float4*>(dLT + row);
    const uint4* rp = reinterpret_cast<const uint4*>(RID + row);
#pragma unroll
    for (int i = 0; i < (DS32 ? 8 : 4); ++i) dbn[i] = dp[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) ridn[i] = rp[i];
    gqn = reinterpret_cast<const float2*>(GQ)[min(tile * QT + c, N - 1)];
  };
  if (t_begin + tslot < t_end) fetch(t_begin + tslot);
  for (int tile = t_begin + tslot; tile < t_end; tile += wpk) {
    const int q0 = __builtin_amdgcn_readfirstlane(tile) * QT;        // (the compiler keeps the loop counter in a vector register)
    const int nq = min(QT, N - q0);
    float dbr[32];
    unsigned ridw[16];
    if constexpr (DS32) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { dbr[4 * i] = dbn[i].x; dbr[4 * i + 1] = dbn[i].y; dbr[4 * i + 2] = dbn[i].z; dbr[4 * i + 3] = dbn[i].w; }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned w4[4] = {__float_as_uint(dbn[i].x), __float_as_uint(dbn[i].y), __float_as_uint(dbn[i].z), __float_as_uint(dbn[i].w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) { dbr[8 * i + 2 * j] = bf_lo(w4[j]); dbr[8 * i + 2 * j + 1] = bf_hi(w4[j]); }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { ridw[4 * i] = ridn[i].x; ridw[4 * i + 1] = ridn[i].y; ridw[4 * i + 2] = ridn[i].z; ridw[4 * i + 3] = ridn[i].w; }
    if (!kvalid) {                                                  // lanes beyond the last key: no region, nothing accumulated
#pragma unroll
      for (int i = 0; i < 16; ++i) ridw[i] = 0xFFFFFFFFu;
    }
    const int gqx = __float_as_int(gqn.x), gqy = __float_as_int(gqn.y);
    if (tile + wpk < t_end) fetch(tile + wpk);                      // the next tile's rows, in flight during this tile's 32 queries
    unsigned nonemask = 0u;                                         // bit q: this key's pair with query q0 + q has no region
    int gy_prev = 0;
    float p1 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      if (q < nq) {                                                 // scalar
        const int gxb = __builtin_amdgcn_readlane(gqx, q), gyb = __builtin_amdgcn_readlane(gqy, q);
        const float d0 = __int_as_float(gxb) - vs0;
        const float p0 = slog1p(d0), s0 = dpos_of<false>(d0, big);
        if (q == 0 || gyb != gy_prev) {                             // scalar: the second coordinate changes once per row of the query grid
          const float d1 = __int_as_float(gyb) - vs1;
          p1 = slog1p(d1); s1 = dpos_of<false>(d1, big);
          gy_prev = gyb;
        }
        const unsigned id = (ridw[q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
        const float dbv = dbr[q];
        if (id != cur) {                                            // the run ends: its sums go to the region's accumulators
          flush();
          if (id == RG_NONE) { nonemask |= 1u << q; cur = ~0u; }    // (the next pair starts a run whatever its id)
          else cur = id;
          r0 = 0.f; r1 = 0.f; r2 = 0.f; u0 = 0.f; u1 = 0.f;
        }
        r0 += dbv; r1 = fmaf(dbv, p0, r1); r2 = fmaf(dbv, p1, r2);
        u0 = fmaf(dbv, s0, u0); u1 = fmaf(dbv, s1, u1);
      }
    }
    if (!kvalid) nonemask = 0u;
    // pairs without a region (~1e-4 of all): the MLP's own backward for that pair, by the whole wave - outside the unrolled loop
    for (unsigned long long todo = __ballot(nonemask != 0u); todo; todo &= todo - 1) {
      const int l = __ffsll((long long)todo) - 1;
      const float v0 = __shfl(vs0, l), v1 = __shfl(vs1, l);
      const int kl = __shfl(keyc, l);
      const DS* drow = dLT + ((size_t)(b * H + h) * NST + (size_t)q0) * J + (size_t)kl * 32;
      for (unsigned m = (unsigned)__shfl((int)nonemask, l); m; m &= m - 1) {
        const int q = __ffs((int)m) - 1;
        const float d0 = GQ[(size_t)(q0 + q) * 2] - v0, d1 = GQ[(size_t)(q0 + q) * 2 + 1] - v1;
        const float pp0 = slog1p(d0), pp1 = slog1p(d1);
        float dbl;
        if constexpr (DS32) dbl = drow[q];
        else dbl = tof<__bf16>((unsigned)drow[q]);
        float h1; bool on2;
        (void)coop_mlp_fwd(mlp, pp0, pp1, c, h1, on2);
        float x2 = mlp.b2;                                          // x2 of unit c again (the forward helper returns only its sign)
#pragma unroll 8
        for (int i = 0; i < CH; ++i) x2 = fmaf(mlp.w2t[i * CH + c], __shfl(h1, i), x2);
        const float g2 = on2 ? mlp.w3 : 0.f;
        float c1 = 0.f;
#pragma unroll 8
        for (int o = 0; o < CH; ++o) c1 = fmaf(mlp.w2r[o * CH + c], __shfl(g2, o), c1);
        c1 = h1 > 0.f ? c1 : 0.f;
        const float dp0 = coop_sum32(c1 * mlp.w1x), dp1 = coop_sum32(c1 * mlp.w1y);
        if (lane == l) {
          dv0 = fmaf(-dbl * dp0, dpos_of<false>(d0, big), dv0);
          dv1 = fmaf(-dbl * dp1, dpos_of<false>(d1, big), dv1);
        }
        for (int ii = 0; ii < 16; ++ii) {                           // dW2[out = c][in = 16 hf + ii]
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
    }
  }
  flush();
  L.dvs[wave][lane] = make_float2(dv0, dv1);
  __syncthreads();
  // d vs of this chunk: the wpk waves of a key block in a fixed order -> slab [chunk][b, h][J]
  if (wave < nkbg && kvalid) {
    float2 sum = L.dvs[wave][lane];
    for (int s2 = 1; s2 < wpk; ++s2) { const float2 t = L.dvs[wave + s2 * nkbg][lane]; sum.x += t.x; sum.y += t.y; }
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
// ------------------------------------------------------------------------------------------------
// host side shared by the fp32-grade (deform_attn.hip) and the 16-bit (deform_attn16.hip) region entry points
// ------------------------------------------------------------------------------------------------
static int ceil_log2_u64(unsigned long long x) { int k = 0; while ((1ull << k) < x && k < 63) ++k; return k; }
struct RegionBwdPlan {
  int chunks, tiles_per_chunk, nkb, nkbg, ngrp, wpk, kbits, shift;
  size_t amax, hist, grad, dvs, part, total;       // byte offsets behind the dq / dkv workspace
};
static RegionBwdPlan region_bwd_plan(int B, int N, int J, int H) {
  RegionBwdPlan p;
  const int ntq = (N + QT - 1) / QT;
  p.chunks = (512 + B * H - 1) / (B * H);
  if (p.chunks < 1) p.chunks = 1;
  if (p.chunks > ntq) p.chunks = ntq;
  p.tiles_per_chunk = (ntq + p.chunks - 1) / p.chunks;
  p.chunks = (ntq + p.tiles_per_chunk - 1) / p.tiles_per_chunk;
  p.nkb = (J + 63) / 64;
  p.ngrp = (p.nkb + 11) / 12;                  // at most 12 waves per workgroup (three per SIMD at <= 168 registers): more than 768 keys
  p.nkbg = (p.nkb + p.ngrp - 1) / p.ngrp;      // are split into groups of key blocks, each group with workgroups of its own
  p.wpk = 12 / p.nkbg;
  if (p.wpk > p.tiles_per_chunk) p.wpk = p.tiles_per_chunk;
  // fixed point: |d bias (1, p0, p1)| <= 4 amax (|p| <= log(1 + |d|) < 4 for any reachable offset); a workgroup adds at most
  // tiles_per_chunk 32 J values into an LDS accumulator, the launch at most B H N J into a global one - both stay below 2^62
  const int kl = 60 - ceil_log2_u64((unsigned long long)p.tiles_per_chunk * QT * J) - 2;
  const int kg = 60 - ceil_log2_u64((unsigned long long)B * H * N * J) - 2;
  p.kbits = kl < 38 ? kl : 38;
  const int kgl = kg < p.kbits ? kg : p.kbits;
  p.shift = p.kbits - kgl;
  size_t o = (bwd_workspace(B, N, J, H).total * sizeof(float) + 255) & ~(size_t)255;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  p.amax = take(256);
  p.hist = take((size_t)RG_RCAP * 3 * 8);
  p.grad = take((size_t)RG_GRAD * 8);
  p.dvs = take((size_t)p.chunks * B * H * J * 2 * sizeof(float));
  p.part = take((size_t)(RG_RCAP / RG_FIN) * RG_GRAD * 8);
  p.total = o;
  return p;
}

static int check_region(const char* fn, int B, int N, int J, int H) {
  SMML_REQUIRE(B > 0 && N > 0 && J > 0 && H > 0, "%s: non-positive dimension", fn);
  SMML_REQUIRE(deform_dims_ok(B, N, J, H), "%s: B, H <= 65535, N <= 2^26, J <= 2^22 (got B %d N %d J %d H %d)", fn, B, N, J, H);
  SMML_REQUIRE(J <= RG_MAX_KEYS, "%s: the region kernels take at most %d keys (got %d)", fn, RG_MAX_KEYS, J);
  return SMML_OK;
}


// pass 3 of a region backward: d vs per pair, region moments (cpb_region_bwd_kernel<DS>), then the dense pass to the six parameter gradients.
// wsb: the call's workspace (bytes), pl: its plan; amax | hist | grad were zeroed and amax filled by the dq pass of the caller.
template <typename DS>
static int region_bias_bwd_launch(const char* fn, const DS* dlogits, const unsigned short* region_ids, const float* vs, const float* gq, CpbParams cp,
                                  const void* tables, char* wsb, const RegionBwdPlan& pl, int B, int N, int J, int H, int nst, int lcap, float* dvs,
                                  float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, void* ev_start, void* ev_stop,
                                  hipStream_t st) {
  const RegionTables rt = region_tables(const_cast<void*>(tables));
  const RegionView rv = region_view(const_cast<void*>(tables));
  unsigned* amax = reinterpret_cast<unsigned*>(wsb + pl.amax);
  unsigned long long* hist = reinterpret_cast<unsigned long long*>(wsb + pl.hist);
  unsigned long long* grad = reinterpret_cast<unsigned long long*>(wsb + pl.grad);
  float* dvs_slab = reinterpret_cast<float*>(wsb + pl.dvs);
  double* part = reinterpret_cast<double*>(wsb + pl.part);
  {   // 89 KB of dynamic LDS: above the 64 KB a kernel gets without asking (a host-side attribute of the function: cheap, idempotent)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cpb_region_bwd_kernel<DS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)sizeof(RegionBwdLds));
    SMML_REQUIRE(e == hipSuccess, "%s: hipFuncSetAttribute failed: %s", fn, hipGetErrorString(e));
  }
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, st);
  hipLaunchKernelGGL(cpb_region_bwd_kernel<DS>, dim3(pl.chunks * pl.ngrp, H, B), dim3(64 * pl.nkbg * pl.wpk), sizeof(RegionBwdLds), st, dlogits,
                     region_ids, vs, gq, cp, rv, amax, hist, grad, dvs_slab, N, J, H, nst, pl.nkb, pl.nkbg, pl.chunks, pl.wpk, pl.tiles_per_chunk,
                     pl.kbits, pl.shift, lcap);
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, st);
  SMML_LAUNCH_CHECK(fn);
  const size_t n = (size_t)B * H * J;
  hipLaunchKernelGGL(region_dvs_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float2*>(dvs_slab),
                     reinterpret_cast<float2*>(dvs), n, pl.chunks);
  const int groups = RG_RCAP / RG_FIN;
  hipLaunchKernelGGL(region_final1_kernel, dim3(groups), dim3(256), 0, st, rt, hist, part);
  hipLaunchKernelGGL(region_final2_kernel, dim3((RG_GRAD + 255) / 256), dim3(256), 0, st, part, groups, grad, amax, pl.kbits - pl.shift, dw1, db1,
                     dw2, db2, dw3, db3);
  SMML_LAUNCH_CHECK(fn);
  return SMML_OK;
}

}  // namespace
