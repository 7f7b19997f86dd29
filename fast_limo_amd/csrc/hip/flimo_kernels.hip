// fast_limo_amd/csrc/hip/flimo_kernels.hip  -- gfx950 (MI355X) only.
//
// Hand-written HIP kernels of the fast_LIMO registration hot path:
//   match_kernel<L>   Mapper::match + match_plane + Plane + Match + calculate_H for one scan
//                     point per group of L lanes (reference Modules/Mapper.cpp:59-114,
//                     Objects/Octree.hpp:526-599, Objects/Plane.cpp:23-114, Objects/Match.cpp:23-28,
//                     Modules/Localizer.cpp:537-577)
//   knn_kernel<L>     octree::Octree::knn for a batch of world-frame queries (Octree.hpp:526-555)
//   reduce_kernel     h_x^T h_x, h_x^T h and the match count with v_mfma_f64_16x16x4_f64
//                     (esekfom.hpp:1723,1727)
//   cap_kernel        MAX_NUM_MATCHES "first M matches" rule (Localizer.cpp:539)
//   deskew_kernel     Localizer::deskewPointCloud loop (Localizer.cpp:822-843)
//   transform_kernel  pcl::transformPointCloud(pc2match, state.get_RT()) (Localizer.cpp:361)
//
// The map is a uniform grid (flimo_types.h: GridView).  A query visits the 3x3x3 block of cells
// around it as 9 contiguous point ranges (x-adjacent cells are adjacent in memory), keeps a
// register-resident sorted best-5, and proves exactness with a conservative "ball inside the
// visited block" test; otherwise it widens ring by ring (shell only, pruned by box distance).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <float.h>
#include <limits.h>
#include <stdlib.h>
#include <algorithm>
#include "flimo_types.h"
#include "flimo_math.h"
#include "flimo_kernels.h"
#include "flimo_chain.h"
#include "flimo_ieskf.h"

#pragma clang fp contract(off)

namespace flimo {

// ------------------------------------------------------------------------------------------
// best-5 list: ascending by (squared distance, map index).  Empty slots: (FLT_MAX, INT_MAX).
// ------------------------------------------------------------------------------------------
FLIMO_DEV bool pair_less(float d0, int i0, float d1, int i1) { return (d0 < d1) || (d0 == d1 && i0 < i1); }

template <int K>
FLIMO_DEV void topk_insert(float (&bd)[K], int (&bi)[K], float d, int id) {
  if (!pair_less(d, id, bd[K - 1], bi[K - 1])) return;
  bd[K - 1] = d;
  bi[K - 1] = id;
#pragma unroll
  for (int s = K - 1; s > 0; s--) {
    const bool sw = pair_less(bd[s], bi[s], bd[s - 1], bi[s - 1]);
    const float td = bd[s - 1];
    const int ti = bi[s - 1];
    bd[s - 1] = sw ? bd[s] : td;
    bi[s - 1] = sw ? bi[s] : ti;
    bd[s] = sw ? td : bd[s];
    bi[s] = sw ? ti : bi[s];
  }
}

// examine the points of one contiguous range with the L lanes of the query group
template <int L, int K>
FLIMO_DEV void scan_range(const float4* __restrict__ pts, uint32_t lo, uint32_t hi, int sub, float gx, float gy,
                          float gz, float (&bd)[K], int (&bi)[K], int& cand) {
  uint32_t j = lo + (uint32_t)sub;
  // two loads in flight per lane
  for (; j + L < hi; j += 2 * L) {
    const float4 p0 = pts[j];
    const float4 p1 = pts[j + L];
    const float d0 = sqdist3(gx, gy, gz, p0.x, p0.y, p0.z);
    const float d1 = sqdist3(gx, gy, gz, p1.x, p1.y, p1.z);
    topk_insert<K>(bd, bi, d0, (int)j);
    topk_insert<K>(bd, bi, d1, (int)(j + L));
    cand += 2;
  }
  if (j < hi) {
    const float4 p0 = pts[j];
    const float d0 = sqdist3(gx, gy, gz, p0.x, p0.y, p0.z);
    topk_insert<K>(bd, bi, d0, (int)j);
    cand += 1;
  }
}

// merge the private lists of the L lanes of a group; afterwards every lane holds the merged list
template <int L, int K>
FLIMO_DEV void merge_group(float (&bd)[K], int (&bi)[K]) {
#pragma unroll
  for (int off = 1; off < L; off <<= 1) {
    float od[K];
    int oi[K];
#pragma unroll
    for (int s = 0; s < K; s++) {
      od[s] = __shfl_xor(bd[s], off, 64);
      oi[s] = __shfl_xor(bi[s], off, 64);
    }
#pragma unroll
    for (int s = 0; s < K; s++) topk_insert<K>(bd, bi, od[s], oi[s]);
  }
}

// distance (in cell units) from the query to the slab of cells at offset d along one axis;
// r = fractional position of the query inside its own cell, in [0,1]
FLIMO_DEV float slab_dist(int d, float r) {
  return d == 0 ? 0.f : (d > 0 ? ((float)d - r) : (r + (float)(-d - 1)));
}

template <int K = 5>
struct KnnResultK {
  float bd[K];
  int bi[K];
  int cand;      // candidates examined by this lane
  bool exact;    // the K entries are provably the exact K-NN
};
typedef KnnResultK<5> KnnResult;

// visit one row (fixed dy,dz) of the shell (r_prev, r] around cell (cx,cy,cz)
template <int L, int K>
FLIMO_DEV void visit_row(const GridView& G, int cx, int cy, int cz, int dy, int dz, int r_prev, int r, float rx,
                         float ry, float rz, float margin, float cell2, float bound, int sub, float gx, float gy,
                         float gz, KnnResultK<K>& R) {
  const float a = fmaxf(slab_dist(dy, ry) - margin, 0.f), b = fmaxf(slab_dist(dz, rz) - margin, 0.f);
  const float yz2 = a * a + b * b;
  if (yz2 * cell2 >= fminf(bound, R.bd[K - 1])) return;   // the whole row is farther than the current 5th best
  if (max(abs(dy), abs(dz)) > r_prev) {
    const int x0 = max(cx - r, 0), x1 = min(cx + r, G.nx - 1);
    if (x0 <= x1) {
      uint32_t lo, hi;
      grid_row_range(G, G.dir, cy + dy, cz + dz, x0 * G.xs, (x1 + 1) * G.xs, lo, hi);
      scan_range<L, K>(G.pts, lo, hi, sub, gx, gy, gz, R.bd, R.bi, R.cand);
    }
  } else {
    // only the cells beyond the block already visited: [cx-r, cx-r_prev-1] and [cx+r_prev+1, cx+r]
    {
      const int x0 = max(cx - r, 0), x1 = min(cx - r_prev - 1, G.nx - 1);
      const float sx = fmaxf(slab_dist(-(r_prev + 1), rx) - margin, 0.f);
      if (x0 <= x1 && (sx * sx + yz2) * cell2 < fminf(bound, R.bd[K - 1])) {
        uint32_t lo, hi;
        grid_row_range(G, G.dir, cy + dy, cz + dz, x0 * G.xs, (x1 + 1) * G.xs, lo, hi);
        scan_range<L, K>(G.pts, lo, hi, sub, gx, gy, gz, R.bd, R.bi, R.cand);
      }
    }
    {
      const int x0 = max(cx + r_prev + 1, 0), x1 = min(cx + r, G.nx - 1);
      const float sx = fmaxf(slab_dist(r_prev + 1, rx) - margin, 0.f);
      if (x0 <= x1 && (sx * sx + yz2) * cell2 < fminf(bound, R.bd[K - 1])) {
        uint32_t lo, hi;
        grid_row_range(G, G.dir, cy + dy, cz + dz, x0 * G.xs, (x1 + 1) * G.xs, lo, hi);
        scan_range<L, K>(G.pts, lo, hi, sub, gx, gy, gz, R.bd, R.bi, R.cand);
      }
    }
  }
}

// Exact 5-NN of (gx,gy,gz) over the grid; the L lanes of a group cooperate (sub = lane % L).
// max_ring bounds the widening: if the search stops at max_ring without `exact`, then the true
// 5th squared distance is >= ((max_ring + edge - margin) * cell)^2 >= (max_ring*cell - margin*cell)^2.
template <int L, int K = 5>
FLIMO_DEV void knn_search(const GridView& G, float gx, float gy, float gz, int sub, int max_ring, KnnResultK<K>& R) {
#pragma unroll
  for (int s = 0; s < K; s++) { R.bd[s] = FLT_MAX; R.bi[s] = INT_MAX; }
  R.cand = 0;
  R.exact = false;

  // query cell (same float expression as the map build: floor((p - o) * inv_cell))
  const float fx = (gx - G.ox) * G.inv_cell, fy = (gy - G.oy) * G.inv_cell, fz = (gz - G.oz) * G.inv_cell;
  if (!(fx == fx) || !(fy == fy) || !(fz == fz)) return;   // NaN query: no neighbours
  const float lim = 1.0e9f;
  const float flx = floorf(fminf(fmaxf(fx, -lim), lim));
  const float fly = floorf(fminf(fmaxf(fy, -lim), lim));
  const float flz = floorf(fminf(fmaxf(fz, -lim), lim));
  const int cx = (int)flx - G.six, cy = (int)fly - G.siy, cz = (int)flz - G.siz;      // (GridView: the origin is fixed, the grid's corner is a cell shift)
  const float rx = fminf(fmaxf(fx - flx, 0.f), 1.f), ry = fminf(fmaxf(fy - fly, 0.f), 1.f),
              rz = fminf(fmaxf(fz - flz, 0.f), 1.f);

  const int maxdim = grid_maxdim(G);
  const float margin = 1.0e-3f + 4.0e-7f * (float)maxdim;   // cell units; covers the rounding of the cell map
  const float cell2 = G.cell * G.cell;
  const float edge = fminf(fminf(fminf(rx, 1.f - rx), fminf(ry, 1.f - ry)), fminf(rz, 1.f - rz));

  // first ring that can reach the grid at all
  int r0;
  {
    const int ox_ = cx < 0 ? -cx : (cx >= G.nx ? cx - G.nx + 1 : 0);
    const int oy_ = cy < 0 ? -cy : (cy >= G.ny ? cy - G.ny + 1 : 0);
    const int oz_ = cz < 0 ? -cz : (cz >= G.nz ? cz - G.nz + 1 : 0);
    r0 = max(1, max(ox_, max(oy_, oz_)));
  }
  if (r0 > max_ring) return;   // nothing within the gate distance

  int r_prev = -1;
  int r = r0;
  float bound = FLT_MAX;       // merged 5th-best of the group so far (valid pruning bound)
  for (;;) {
    if (r == 1) {
      // ---- common case: the 3x3x3 block as 9 contiguous ranges.  All 18 range bounds are loaded
      //      before any point so their latencies overlap. ----
      uint32_t lo[9], hi[9];
      const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.nx - 1);
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const int dy = (t % 3 == 0) ? 0 : ((t % 3 == 1) ? -1 : 1);
        const int dz = (t / 3 == 0) ? 0 : ((t / 3 == 1) ? -1 : 1);
        const int yy = cy + dy, zz = cz + dz;
        const bool in = (yy >= 0) && (yy < G.ny) && (zz >= 0) && (zz < G.nz) && (x0 <= x1);
        lo[t] = 0u; hi[t] = 0u;
        if (in) grid_row_range(G, G.dir, yy, zz, x0 * G.xs, (x1 + 1) * G.xs, lo[t], hi[t]);
      }
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const int dy = (t % 3 == 0) ? 0 : ((t % 3 == 1) ? -1 : 1);
        const int dz = (t / 3 == 0) ? 0 : ((t / 3 == 1) ? -1 : 1);
        const float a = fmaxf(slab_dist(dy, ry) - margin, 0.f), b = fmaxf(slab_dist(dz, rz) - margin, 0.f);
        if ((a * a + b * b) * cell2 >= R.bd[K - 1]) continue;
        scan_range<L, K>(G.pts, lo[t], hi[t], sub, gx, gy, gz, R.bd, R.bi, R.cand);
      }
    } else {
      // ---- shell (r_prev, r] (rare): rows in ascending order, clipped to the grid ----
      const int dz0 = max(-r, -cz), dz1 = min(r, G.nz - 1 - cz);
      const int dy0 = max(-r, -cy), dy1 = min(r, G.ny - 1 - cy);
      for (int dz = dz0; dz <= dz1; dz++)
        for (int dy = dy0; dy <= dy1; dy++)
          visit_row<L, K>(G, cx, cy, cz, dy, dz, r_prev, r, rx, ry, rz, margin, cell2, bound, sub, gx, gy, gz, R);
    }
    // ---- merge the group, test exactness ----
    if (L > 1) merge_group<L, K>(R.bd, R.bi);
    const float rg = ((float)r + edge - margin) * G.cell;   // every unvisited point is >= rg away
    const bool have5 = R.bi[K - 1] != INT_MAX;
    const bool covers = (cx - r <= 0) && (cx + r >= G.nx - 1) && (cy - r <= 0) && (cy + r >= G.ny - 1) &&
                        (cz - r <= 0) && (cz + r >= G.nz - 1);
    if (covers || (have5 && R.bd[K - 1] <= rg * rg * (1.f - 1.0e-6f))) { R.exact = true; break; }
    if (r >= max_ring) break;
    // next ring: straight to the radius that proves exactness once 5 candidates are known, else double
    int rn = 2 * r;
    if (have5) {
      bound = R.bd[K - 1];
      const float need = fl_sqrt(R.bd[K - 1]) * G.inv_cell * (1.f + 4.0e-6f) - edge + margin;
      rn = max(r + 1, (int)ceilf(fminf(need, 1.0e9f)));
    }
    rn = min(rn, max_ring);
    // keep the merged list on lane 0 of the group only (no duplicates at the next merge)
    if (L > 1 && sub != 0) {
#pragma unroll
      for (int s = 0; s < K; s++) { R.bd[s] = FLT_MAX; R.bi[s] = INT_MAX; }
    }
    r_prev = r;
    r = rn;
  }
}

// ------------------------------------------------------------------------------------------
// Fast path of the per-pass k-NN: L lanes per scan point, 3x3x3 block only.
//   * the 18 range bounds of the 9 rows are loaded first, then ALL candidate points of the block
//     are addressed through one flattened index space (slot s -> row t, offset), so every lane
//     issues its loads back to back (up to SLOTS in flight) regardless of how the candidates are
//     spread over the rows;
//   * each lane keeps a sorted private best-5 of 64-bit keys (float bits of d2 << 32 | map index:
//     the same total order as (d2, index)); the group result is extracted with five rounds of a
//     shuffle min-reduction;
//   * a query whose 5-ball is not provably inside the block goes to a worklist that
//     widen_kernel finishes with the general ring search.
// Queries are processed in the spatial (Morton) order of `scan_sorted`; w of each scan point holds
// its original index.  Block b works on chunk (b % 8) * (nblocks / 8) + b / 8 so that each XCD
// (observed dispatch: block b -> XCD b % 8) sees one contiguous spatial range and its L2 keeps
// that part of the map.
// ------------------------------------------------------------------------------------------
typedef unsigned long long u64;
#define KEY_EMPTY 0xffffffffffffffffull

// Developer-only phase stamps (built with -DFLIMO_TRACE into libflimo_hip_trace.so, never shipped):
// thread 0 of every block stores the 100 MHz wall clock after draining its outstanding memory operations.
#ifdef FLIMO_TRACE
__device__ unsigned long long g_trace[2][16384 * 8];
#if FLIMO_TRACE == 2
// light form (round 6): the stamps are kept in shared memory and written out once, at the launch's end (TRACE_FLUSH) -- a stamp
// stored to global memory makes the NEXT stamp's s_waitcnt wait for that store's acknowledgement (0.4 - 1 us per phase)
__shared__ unsigned long long s_trace_lds[2][8];
#define TRACE(k, slot)                                                                                  \
  do {                                                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
    if (threadIdx.x == 0) s_trace_lds[k][slot] = wall_clock64();                                        \
  } while (0)
#define TRACE_FLUSH()                                                                                   \
  do {                                                                                                  \
    if (threadIdx.x == 0 && blockIdx.x < 16384)                                                         \
      for (int k_ = 0; k_ < 2; k_++) for (int s_ = 0; s_ < 8; s_++) g_trace[k_][blockIdx.x * 8 + s_] = s_trace_lds[k_][s_]; \
  } while (0)
#else
#define TRACE(k, slot)                                                                                  \
  do {                                                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
    if (threadIdx.x == 0 && blockIdx.x < 16384) g_trace[k][blockIdx.x * 8 + (slot)] = wall_clock64();  \
  } while (0)
#define TRACE_FLUSH() do {} while (0)
#endif
extern "C" int flimo_trace_read(int k, unsigned long long* out, size_t n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), n * sizeof(unsigned long long), (size_t)k * 16384 * 8 * sizeof(unsigned long long),
                                  hipMemcpyDeviceToHost);
}
#else
#define TRACE(k, slot) do {} while (0)
#define TRACE_FLUSH() do {} while (0)
#endif

FLIMO_DEV u64 make_key(float d, uint32_t idx) { return ((u64)__float_as_uint(d) << 32) | (u64)idx; }

FLIMO_DEV void key5_insert(u64 (&k)[5], u64 key) {
  if (key >= k[4]) return;
  k[4] = key;
#pragma unroll
  for (int s = 4; s > 0; s--) {
    const u64 a = k[s - 1], b = k[s];
    const bool sw = b < a;
    k[s - 1] = sw ? b : a;
    k[s] = sw ? a : b;
  }
}

// Block -> chunk of the Morton-sorted scan.  The dispatcher places block b on XCD b % 8 (observed); XCD x works on
// stripes of c_stripe consecutive chunks, stripes dealt round-robin over the XCDs: neighbouring queries (same map
// cells) stay on one XCD's L2 while every XCD gets its share of the cheap (outside the map) and the expensive
// (far, cache-cold) parts of the scan.  c_stripe = 0: one contiguous eighth per XCD.
__constant__ int c_stripe = 8;
void set_xcd_stripe(int stripe) { (void)hipMemcpyToSymbol(HIP_SYMBOL(c_stripe), &stripe, sizeof(int)); }

__device__ __forceinline__ int xcd_chunk(int b, int nb) {
  // nb is a multiple of 8
  const int x = b & 7, k = b >> 3, per = nb >> 3;
  const int S = c_stripe;
  if (S <= 0 || S >= per) return x * per + k;
  const int full = (per / S) * S;                 // the last partial stripe of every XCD falls back to contiguous
  if (k >= full) return 8 * full + x * (per - full) + (k - full);
  return ((k / S) * 8 + x) * S + (k % S);
}

struct NbrRec {      // 32 bytes per query (sorted order)
  int32_t idx[5];
  int32_t flag;      // 1: exact 5-NN present, 0: no valid neighbourhood, 2: pending (worklist)
  int32_t d5_bits;   // float bits of the 5th squared distance (bound for the next pass of the same scan)
  int32_t d5_valid;  // 1 when flag == 1
};

// Sorted private best-5 of one lane.  A (distance, position) pair is the 64-bit key (float bits << 32 | position);
// distances are non-negative finite floats (+inf = empty), so the key read as an IEEE double is a positive finite
// double (the float's exponent field lands inside the double's 11-bit exponent, below 0x7ff) and doubles of one sign
// order exactly like their bit patterns: v_min_f64 / v_max_f64 compare-and-select whole keys in ONE instruction each
// (fp64 denormals are always preserved on gfx9, min/max return an operand unchanged).  Insertion into the sorted list is
// a 9-instruction min/max ladder, no compares, no payload selects.
struct U3 { uint32_t a, b, c; };
struct __attribute__((aligned(8))) Seg2 { uint32_t x0, y0, x1, y1; };   // two neighbouring entries of a row of GridView::segs (one 16-byte load)
__device__ __forceinline__ double key_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double key_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double key_make(float d, uint32_t id) { return __hiloint2double((int)__float_as_uint(d), (int)id); }
__device__ __forceinline__ void best5_insert(double (&k)[5], double x) {
  double t = x;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const double lo = key_min(k[i], t);
    t = key_max(k[i], t);
    k[i] = lo;
  }
  k[4] = key_min(k[4], t);
}
// The same with a sixth entry: the best candidate that did NOT make the five.  Its distance tells whether the choice of the five
// hinges on an exact float32 tie (d5 == d6) -- the one case in which the reference's answer depends on its octree's visiting
// order (tie_kernel below restores that order).
__device__ __forceinline__ void best6_insert(double (&k)[6], double x) {
  double t = x;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    const double lo = key_min(k[i], t);
    t = key_max(k[i], t);
    k[i] = lo;
  }
  k[5] = key_min(k[5], t);
}
#define KEY_NONE 0x7f800000ffffffffull     // (+inf, -1): an empty slot of the merged list

// Minimum of a key over an aligned group of Gl lanes (2, 4, ..., 64), result in every lane of the group.  Within a row of 16
// lanes the exchange is DPP (quad permutes, half-row and row mirrors: a couple of cycles each, no LDS crossbar round trip as with
// ds_bpermute); across rows the four row minima are read with v_readlane.  A 5-round extraction over 64 lanes costs ~0.3 us
// this way instead of ~1.5 us of dependent ds_bpermute pairs.
template <int CTRL>
__device__ __forceinline__ double key_dpp(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int l2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  const int h2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(h2, l2);
}
__device__ __forceinline__ double key_group_min(double md, int Gl, int lane) {
  md = key_min(md, key_dpp<0xB1>(md));                        // quad_perm [1,0,3,2]
  if (Gl >= 4) md = key_min(md, key_dpp<0x4E>(md));           // quad_perm [2,3,0,1]
  if (Gl >= 8) md = key_min(md, key_dpp<0x141>(md));          // row_half_mirror
  if (Gl >= 16) md = key_min(md, key_dpp<0x140>(md));         // row_mirror
  if (Gl >= 32) {
    const int lo = __double2loint(md), hi = __double2hiint(md);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    const double a = key_min(r0, r1), b = key_min(r2, r3);
    md = (Gl >= 64) ? key_min(a, b) : (lane < 32 ? a : b);
  }
  return md;
}

// Two ascending 6-lists of unique keys -> the six smallest, ascending: min(a[i], b[5 - i]) are the six smallest of the twelve
// (bitonic merge), a 12-exchange network (depth 5) sorts them.
__device__ __forceinline__ void key_merge6(const double (&a)[6], const double (&b)[6], double (&c)[6]) {
#pragma unroll
  for (int i = 0; i < 6; i++) c[i] = key_min(a[i], b[5 - i]);
#define FLIMO_CE(i, j) { const double lo_ = key_min(c[i], c[j]); c[j] = key_max(c[i], c[j]); c[i] = lo_; }
  FLIMO_CE(0, 5) FLIMO_CE(1, 3) FLIMO_CE(2, 4)
  FLIMO_CE(1, 2) FLIMO_CE(3, 4)
  FLIMO_CE(0, 3) FLIMO_CE(2, 5)
  FLIMO_CE(0, 1) FLIMO_CE(2, 3) FLIMO_CE(4, 5)
  FLIMO_CE(1, 2) FLIMO_CE(3, 4)
#undef FLIMO_CE
}
// the lists of the two lanes of a query (L = 2) -> the pair's list in both lanes: one DPP quad permute per half key (all twelve
// independent) instead of six dependent extraction rounds; keys are unique, so both lanes arrive at the same list
__device__ __forceinline__ void key_pair_merge6(const double (&a)[6], double (&c)[6]) {
  double b[6];
#pragma unroll
  for (int i = 0; i < 6; i++) b[i] = key_dpp<0xB1>(a[i]);
  key_merge6(a, b, c);
}

// N candidate slots of one lane: stream positions s0, s0 + L, ...; dead slots (>= total) re-read the last position and
// insert +inf.  All N loads are issued back to back (the empty asm consumes every loaded value at once).
// PAYW: the key's payload is the point's w (the fine level's copies carry their position in the main map there) instead of the
// position in `pts`
template <int L, int N, bool PAYW = false, int NSEG = 9>
__device__ __forceinline__ void knn5_body(const float4* __restrict__ pts, uint32_t s0, uint32_t total, uint32_t last,
                                          const uint32_t (&off)[NSEG + 1], const uint32_t (&dl)[NSEG], float gx, float gy, float gz,
                                          double (&k5)[6]) {
  float4 pt[N];
  uint32_t id[N];
#pragma unroll
  for (int u = 0; u < N; u++) {
    const uint32_t s = min(s0 + u * L, last);
    uint32_t delta = dl[0];
#pragma unroll
    for (int t = 1; t < NSEG; t++) delta = (s >= off[t]) ? dl[t] : delta;
    id[u] = s + delta;
    pt[u] = pts[id[u]];
  }
  if constexpr (N == 8)
    asm volatile("" : "+v"(pt[0].x), "+v"(pt[0].y), "+v"(pt[0].z), "+v"(pt[1].x), "+v"(pt[1].y), "+v"(pt[1].z),
                      "+v"(pt[2].x), "+v"(pt[2].y), "+v"(pt[2].z), "+v"(pt[3].x), "+v"(pt[3].y), "+v"(pt[3].z),
                      "+v"(pt[4].x), "+v"(pt[4].y), "+v"(pt[4].z), "+v"(pt[5].x), "+v"(pt[5].y), "+v"(pt[5].z),
                      "+v"(pt[6].x), "+v"(pt[6].y), "+v"(pt[6].z), "+v"(pt[7].x), "+v"(pt[7].y), "+v"(pt[7].z));
  else if constexpr (N == 4)
    asm volatile("" : "+v"(pt[0].x), "+v"(pt[0].y), "+v"(pt[0].z), "+v"(pt[1].x), "+v"(pt[1].y), "+v"(pt[1].z),
                      "+v"(pt[2].x), "+v"(pt[2].y), "+v"(pt[2].z), "+v"(pt[3].x), "+v"(pt[3].y), "+v"(pt[3].z));
  else if constexpr (N == 2)
    asm volatile("" : "+v"(pt[0].x), "+v"(pt[0].y), "+v"(pt[0].z), "+v"(pt[1].x), "+v"(pt[1].y), "+v"(pt[1].z));
#pragma unroll
  for (int u = 0; u < N; u++) {
    const bool live = s0 + u * L < total;
    const float d = sqdist3(gx, gy, gz, pt[u].x, pt[u].y, pt[u].z);
    best6_insert(k5, key_make(live ? d : INFINITY, live ? (PAYW ? __float_as_uint(pt[u].w) : id[u]) : 0xffffffffu));
  }
}


// ------------------------------------------------------------------------------------------
// In-kernel widening ("tail") of the per-pass k-NN: the queries of a wave whose 5-ball is not provably inside their
// 3x3x3 block are finished by the SAME wave right after its fast path -- no worklist, no second dispatch, no
// cross-wave protocol.  The wave's F pending queries share its 64 lanes: Gl = 64 / pow2ceil(F) lanes per query
// (64 lanes for a lone straggler of a converged pass, 2 lanes each when a whole wave of far-off points of the first pass is
// pending), so the work of a wave is spread over all of its lanes whatever F is.  Per query and ring r (2 or 3):
//   * the (2r+1)^2 rows of the ring-r block are dealt to the group's lanes; a row (and the cells at its two ends) that
//     cannot hold one of the five nearest points -- farther than the bound known from the 3x3x3 block's own 5th
//     distance and from the previous pass of the same scan -- is dropped (exact, same rule as the fast path);
//   * every lane walks the candidates of its own rows (bounds of four rows per round trip, eight candidate loads in flight),
//     private best-5, min-extraction across the group (DPP); the block is searched afresh (the bound, not the list, is what
//     the 3x3x3 pass hands over);
//   * exactness test as everywhere: the 5-ball inside the searched block, else the next ring, never beyond max_ring.
// ------------------------------------------------------------------------------------------
struct FitIdx { unsigned char raw[FIT_LIVE_PAD]; };   // live sum k -> index into the wave's 256 raw MFMA accumulators
// TieList (flimo_kernels.h): list + counter of THIS pass; the pass's last reduction block re-arms the counter the NEXT pass will
// use (two counters, alternating: the host may still read this pass's count for tie_kernel)
// exactly tied float32 distances among the five, or between the 5th and the best of the rest
__device__ __forceinline__ bool key_has_tie(const u64 (&best)[5], u64 sixth) {
  const uint32_t d0 = (uint32_t)(best[0] >> 32), d1 = (uint32_t)(best[1] >> 32), d2 = (uint32_t)(best[2] >> 32),
                 d3 = (uint32_t)(best[3] >> 32), d4 = (uint32_t)(best[4] >> 32), d5 = (uint32_t)(sixth >> 32);
  return (d0 == d1) | (d1 == d2) | (d2 == d3) | (d3 == d4) | (d4 == d5);
}
constexpr int TAIL_MAX_RING = 3;
constexpr int KNN_FAR_RING = 4;                  // flimo_knn: rings the ring search tries before the tiles' best-first search takes over
constexpr uint32_t PROBE_MIN_OWN = 6;           // first pass: an own cell with fewer points gives no useful bound
struct __align__(16) WaveLds {                  // one per wave of the block
  float tile[16 * 65];                          // fused pass: the wave's rows, [col][row] with stride 65
  double acc[256];                              // fused pass: the wave's raw f64 MFMA accumulators
  int src[64];                                  // tail: lane of the wave's i-th pending query
  int res[64][8];                               // fused pass: the tail hands a finished query back to its own lane (5 indices, flag)
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void knn5_tail(const GridView& G, const uint16_t* dir, int max_ring, NbrRec* __restrict__ nbr, bool pending, int p,
                                          float gx, float gy, float gz, uint32_t hint_bits, float b2, WaveLds& S,
                                          int* __restrict__ straggler_count, unsigned long long* __restrict__ cand_total,
                                          bool keep_res, const TieList& tl) {
  const int lane = threadIdx.x & 63;
  const u64 B = __ballot(pending);
  const int F = __popcll(B);
  if (F == 0) return;                                         // wave-uniform
  if (lane == 0) atomicAdd(straggler_count, F);               // statistics only (published with the pass result)
#ifdef FLIMO_TRACE
  const unsigned long long tr_t0 = wall_clock64();
  int tr_iters = 0;
#endif
  if (pending) S.src[__popcll(B & ((1ull << lane) - 1ull))] = lane;
  wave_lds_sync();
  int ngroups = 1;
  while (ngroups < F && ngroups < 32) ngroups <<= 1;          // pow2ceil(F), at most 32 (>= 2 lanes per query)
  const int Gl = 64 / ngroups;
  const int lg = __ffs(Gl) - 1;
  const int grp = lane >> lg, sub = lane & (Gl - 1);
  const int maxdim = grid_maxdim(G);
  const float margin = 1.0e-3f + 4.0e-7f * (float)maxdim;
  const double none = __longlong_as_double((long long)KEY_NONE);
  int cand = 0;
  for (int base = 0; base < F; base += ngroups) {
    const int qi = base + grp;
    bool active = qi < F;
    const int src = S.src[active ? qi : 0];
    const float qx = __shfl(gx, src, 64), qy = __shfl(gy, src, 64), qz = __shfl(gz, src, 64);
    const int qp = __shfl(p, src, 64);
    const float hint = __uint_as_float((uint32_t)__shfl((int)hint_bits, src, 64));   // 5th sq. distance inside the 3x3x3 block (+inf: none)
    float bnd2 = __shfl(b2, src, 64);                          // pruning bound, cell units squared (+inf: none)
    const float fx = (qx - G.ox) * G.inv_cell, fy = (qy - G.oy) * G.inv_cell, fz = (qz - G.oz) * G.inv_cell;
    const float flx = floorf(fminf(fmaxf(fx, -1.0e9f), 1.0e9f)), fly = floorf(fminf(fmaxf(fy, -1.0e9f), 1.0e9f)),
                flz = floorf(fminf(fmaxf(fz, -1.0e9f), 1.0e9f));
    const int cx = (int)flx - G.six, cy = (int)fly - G.siy, cz = (int)flz - G.siz;      // (GridView: the origin is fixed, the grid's corner is a cell shift)
    const float rx = fminf(fmaxf(fx - flx, 0.f), 1.f), ry = fminf(fmaxf(fy - fly, 0.f), 1.f),
                rz = fminf(fmaxf(fz - flz, 0.f), 1.f);
    const float edge = fminf(fminf(fminf(rx, 1.f - rx), fminf(ry, 1.f - ry)), fminf(rz, 1.f - rz));
    int r = 2;
    {
      const int ox_ = cx < 0 ? -cx : (cx >= G.nx ? cx - G.nx + 1 : 0);
      const int oy_ = cy < 0 ? -cy : (cy >= G.ny ? cy - G.ny + 1 : 0);
      const int oz_ = cz < 0 ? -cz : (cz >= G.nz ? cz - G.nz + 1 : 0);
      r = max(r, max(ox_, max(oy_, oz_)));                     // first ring that reaches the grid at all
      if (hint >= 0.f && hint < INFINITY) {                    // an upper bound of the true 5th distance: its ring, its ball
        const float need = fl_sqrt(hint) * G.inv_cell * (1.f + 4.0e-6f) - edge + margin;
        r = max(r, (int)ceilf(fminf(need, 1.0e9f)));
        const float rc = (fl_sqrt(hint) * (1.f + 1.0e-5f) + 1.0e-6f) * G.inv_cell;
        bnd2 = fminf(bnd2, rc * rc * (1.f + 1.0e-5f));
      }
      r = min(r, max_ring);
    }
    u64 best[5] = {KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE};
    u64 sixth = KEY_NONE;
    int flag = 0;
    for (;;) {
      if (!__any(active)) break;
#ifdef FLIMO_TRACE
      tr_iters++;
#endif
      if (active) {
        // Every lane of the group owns rows of the ring-r block (row j of lane sub: j = sub + i * Gl) and walks its own rows'
        // candidates: bounds of four rows in one round trip, then up to eight candidate loads in flight per row.  No
        // flattening, no tables: with 64 lanes per query (the lone straggler of a converged pass) a lane owns at most one
        // row and the chain is two round trips (bounds, candidates) and the extraction.
        const int side = 2 * r + 1, rows = side * side;
        // lanes per row: a lone straggler has the whole wave -- two lanes share each of the 25 rows of a ring-2 block
        const int lr = (Gl == 64 && rows <= 32) ? 2 : 1, nslots = Gl / lr;
        const int slot = sub / lr, part = sub - slot * lr;
        double k5[6] = {none, none, none, none, none, none};
        for (int jb = slot; jb < rows; jb += 4 * nslots) {
          uint32_t lo4[4], hi4[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int j = jb + u * nslots;
            lo4[u] = 0u; hi4[u] = 0u;
            if (j < rows) {
              const int jz = j / side, jy = j - jz * side;
              const int dy = jy - r, dz = jz - r;
              const int yy = cy + dy, zz = cz + dz;
              const float a = fmaxf(slab_dist(dy, ry) - margin, 0.f), b = fmaxf(slab_dist(dz, rz) - margin, 0.f);
              const float dyz2 = a * a + b * b;
              if (dyz2 <= bnd2 && yy >= 0 && yy < G.ny && zz >= 0 && zz < G.nz) {
                // cells of the row the bound's ball can reach: offset +d is (d - rx) away, offset -d is (rx + d - 1) away
                const float xr = fl_sqrt(fmaxf(bnd2 - dyz2, 0.f)) * (1.f + 1.0e-6f) + margin + 1.0e-4f;
                const int dr = (int)fminf(floorf(fminf(xr + rx, 1.0e6f)), (float)r);
                const int dl = (int)fminf(floorf(fminf(xr + (1.f - rx), 1.0e6f)), (float)r);
                const int x0 = max(cx - dl, 0), x1 = min(cx + dr, G.nx - 1);
                if (x0 <= x1) {
                  grid_row_range(G, dir, yy, zz, x0 * G.xs, (x1 + 1) * G.xs, lo4[u], hi4[u]);
                }
              }
            }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t hi = hi4[u];
            for (uint32_t i0 = lo4[u] + (uint32_t)part; i0 < hi; i0 += 8u * (uint32_t)lr) {
              float4 q[8];
#pragma unroll
              for (int w = 0; w < 8; w++) q[w] = G.pts[min(i0 + (uint32_t)(w * lr), hi - 1u)];
              asm volatile("" : "+v"(q[0].x), "+v"(q[0].y), "+v"(q[0].z), "+v"(q[1].x), "+v"(q[1].y), "+v"(q[1].z),
                                "+v"(q[2].x), "+v"(q[2].y), "+v"(q[2].z), "+v"(q[3].x), "+v"(q[3].y), "+v"(q[3].z),
                                "+v"(q[4].x), "+v"(q[4].y), "+v"(q[4].z), "+v"(q[5].x), "+v"(q[5].y), "+v"(q[5].z),
                                "+v"(q[6].x), "+v"(q[6].y), "+v"(q[6].z), "+v"(q[7].x), "+v"(q[7].y), "+v"(q[7].z));
#pragma unroll
              for (int w = 0; w < 8; w++) {
                const uint32_t ii = i0 + (uint32_t)(w * lr);
                const bool live = ii < hi;
                const float d = sqdist3(qx, qy, qz, q[w].x, q[w].y, q[w].z);
                best6_insert(k5, key_make(live ? d : INFINITY, live ? ii : 0xffffffffu));
              }
            }
            if (part == 0) cand += (int)(hi - lo4[u]);
          }
        }
        u64 mine[6];
#pragma unroll
        for (int i = 0; i < 6; i++) mine[i] = (u64)__double_as_longlong(k5[i]);
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const double md = key_group_min(__longlong_as_double((long long)mine[0]), Gl, lane);
          const u64 m = (u64)__double_as_longlong(md);
          if (k < 5) best[k] = m; else sixth = m;
          if (mine[0] == m) { mine[0] = mine[1]; mine[1] = mine[2]; mine[2] = mine[3]; mine[3] = mine[4]; mine[4] = mine[5]; mine[5] = KEY_NONE; }
        }
        const float rg = ((float)r + edge - margin) * G.cell;
        const float d5 = __uint_as_float((uint32_t)(best[4] >> 32));
        const bool have5 = d5 < INFINITY;
        const bool covers = (cx - r <= 0) && (cx + r >= G.nx - 1) && (cy - r <= 0) && (cy + r >= G.ny - 1) &&
                            (cz - r <= 0) && (cz + r >= G.nz - 1);
        if (have5 && (covers || d5 <= rg * rg * (1.f - 1.0e-6f))) { flag = 1; active = false; }
        else if (covers || r >= max_ring) { flag = 0; active = false; }
        else {
          int rn = r + 1;
          if (have5) {
            const float need = fl_sqrt(d5) * G.inv_cell * (1.f + 4.0e-6f) - edge + margin;
            rn = max(r + 1, (int)ceilf(fminf(need, 1.0e9f)));
            const float rc = (fl_sqrt(d5) * (1.f + 1.0e-5f) + 1.0e-6f) * G.inv_cell;
            bnd2 = fminf(bnd2, rc * rc * (1.f + 1.0e-5f));
          }
          r = min(rn, max_ring);
        }
      }
    }
    if (qi < F && sub == 0) {
      const bool tie = flag == 1 && key_has_tie(best, sixth);
      int4 a, b;
      a.x = (int)(uint32_t)best[0]; a.y = (int)(uint32_t)best[1]; a.z = (int)(uint32_t)best[2]; a.w = (int)(uint32_t)best[3];
      b.x = (int)(uint32_t)best[4]; b.y = flag; b.z = (int)(uint32_t)(best[4] >> 32); b.w = ((flag == 1) ? 1 : 0) | (tie ? 2 : 0);
      int4* o = reinterpret_cast<int4*>(&nbr[qp]);
      o[0] = a;
      o[1] = b;
      if (tie && tl.list) { const unsigned slot = atomicAdd(tl.count, 1u); if (slot < tl.cap) tl.list[slot] = qp; }
      if (keep_res) {
        int4* rr = reinterpret_cast<int4*>(S.res[src]);
        rr[0] = a;
        rr[1] = b;
      }
    }
  }
  if (cand_total && cand) atomicAdd(cand_total, (unsigned long long)cand);
#if defined(FLIMO_TRACE) && FLIMO_TRACE != 2
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0 && blockIdx.x < 16384)      // developer statistics of the tail: duration (10 ns units), stragglers of the wave, ring iterations
    g_trace[0][blockIdx.x * 8 + 7] = ((wall_clock64() - tr_t0) & 0xffffull) | ((unsigned long long)F << 16) | ((unsigned long long)tr_iters << 24) | (1ull << 40);
#endif
}

// ------------------------------------------------------------------------------------------
// deskew (Localizer.cpp:822-843) with State::update (State.cpp:76-119)
// ------------------------------------------------------------------------------------------
struct DevFrame {
  float p[3], q[4], v[3], g[3], w[3], a[3], bg[3], ba[3];
  float pad;
  double time;
};

// Eigen::Quaternionf::toRotationMatrix
FLIMO_DEV void quat_to_rot(float qx, float qy, float qz, float qw, float (&R)[9]) {
  const float tx = 2.f * qx, ty = 2.f * qy, tz = 2.f * qz;
  const float twx = tx * qw, twy = ty * qw, twz = tz * qw;
  const float txx = tx * qx, txy = ty * qx, txz = tz * qx;
  const float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0] = 1.f - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.f - (txx + tyy);
}

// One point: p = raw LiDAR-frame point (w = original index), tk = its absolute stamp; returns the point in the body frame at the
// scan's end (w unchanged).  The per-pass k-NN kernel of a scan's FIRST pass calls it too (the deskew rides on that launch).
__device__ __forceinline__ float4 deskew_point(const float4 p, const double tk, const DevFrame* __restrict__ frames, int nf,
                                               const float* __restrict__ mats /* [0..15] lidar2baselink_T, [16..31] last_state.get_RT_inv() */) {
  // binary_search_tailored (Algorithms.hpp:25-38)
  int low = 0, high = nf - 1;
  while (high >= low) {
    const int mid = (low + high) / 2;
    if (frames[mid].time > tk) high = mid - 1; else low = mid + 1;
  }
  const int i_f = high < 0 ? 0 : high;
  const DevFrame F = frames[i_f];
  // State::update(tk)
  const double dt = tk - F.time;
  const float wx = F.w[0] - F.bg[0], wy = F.w[1] - F.bg[1], wz = F.w[2] - F.bg[2];
  const float w_norm = fl_sqrt(sum3(wx * wx, wy * wy, wz * wz));
  float Rm[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
  if ((double)w_norm > 1.e-7) {
    const float r0 = fl_div(wx, w_norm), r1 = fl_div(wy, w_norm), r2 = fl_div(wz, w_norm);
    const float K[9] = {0.f, -r2, r1, r2, 0.f, -r0, -r1, r0, 0.f};
    const float r_ang = (float)((double)w_norm * dt);
    float s, cs;
    libm_sincosf(r_ang, s, cs);                        // std::sin / std::cos of a float, as the host's libm rounds them
    const float c = (float)(1.0 - (double)cs);
    float cK[9];
#pragma unroll
    for (int i = 0; i < 9; i++) cK[i] = c * K[i];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const float kk = sum3(cK[i * 3 + 0] * K[0 * 3 + j], cK[i * 3 + 1] * K[1 * 3 + j], cK[i * 3 + 2] * K[2 * 3 + j]);
        Rm[i * 3 + j] = Rm[i * 3 + j] + (s * K[i * 3 + j] + kk);
      }
  }
  // a0 = q._transformVector(a - ba) + g   (only needed for p)
  const float qx = F.q[0], qy = F.q[1], qz = F.q[2], qw = F.q[3];
  const float ax = F.a[0] - F.ba[0], ay = F.a[1] - F.ba[1], az = F.a[2] - F.ba[2];
  float ux, uy, uz;
  cross3(qx, qy, qz, ax, ay, az, ux, uy, uz);
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  float cx_, cy_, cz_;
  cross3(qx, qy, qz, ux, uy, uz, cx_, cy_, cz_);
  float a0x = (ax + qw * ux) + cx_, a0y = (ay + qw * uy) + cy_, a0z = (az + qw * uz) + cz_;
  a0x = a0x + F.g[0]; a0y = a0y + F.g[1]; a0z = a0z + F.g[2];
  // q *= Quaternionf(R)
  float ux_, uy_, uz_, uw_;
  {
    float tr = sum3(Rm[0], Rm[4], Rm[8]);            // trace() = diagonal().sum(): Eigen's 3-coefficient redux c0 + (c1 + c2)
    if (tr > 0.f) {
      tr = fl_sqrt(tr + 1.0f);
      uw_ = 0.5f * tr;
      tr = fl_div(0.5f, tr);
      ux_ = (Rm[7] - Rm[5]) * tr;
      uy_ = (Rm[2] - Rm[6]) * tr;
      uz_ = (Rm[3] - Rm[1]) * tr;
    } else {
      // i = the largest diagonal entry, j = (i+1)%3, k = (j+1)%3 (Eigen's quaternion-from-matrix); the three cases written out
      // so that no matrix entry is addressed by a run-time index (registers, no scratch)
      int i = 0;
      if (Rm[4] > Rm[0]) i = 1;
      if (Rm[8] > (i == 0 ? Rm[0] : Rm[4])) i = 2;
      if (i == 0) {
        float tq = fl_sqrt(Rm[0] - Rm[4] - Rm[8] + 1.0f);
        ux_ = 0.5f * tq;
        tq = fl_div(0.5f, tq);
        uw_ = (Rm[7] - Rm[5]) * tq;
        uy_ = (Rm[3] + Rm[1]) * tq;
        uz_ = (Rm[6] + Rm[2]) * tq;
      } else if (i == 1) {
        float tq = fl_sqrt(Rm[4] - Rm[8] - Rm[0] + 1.0f);
        uy_ = 0.5f * tq;
        tq = fl_div(0.5f, tq);
        uw_ = (Rm[2] - Rm[6]) * tq;
        uz_ = (Rm[7] + Rm[5]) * tq;
        ux_ = (Rm[1] + Rm[3]) * tq;
      } else {
        float tq = fl_sqrt(Rm[8] - Rm[0] - Rm[4] + 1.0f);
        uz_ = 0.5f * tq;
        tq = fl_div(0.5f, tq);
        uw_ = (Rm[3] - Rm[1]) * tq;
        ux_ = (Rm[2] + Rm[6]) * tq;
        uy_ = (Rm[5] + Rm[7]) * tq;
      }
    }
  }
  const float nqw = qw * uw_ - qx * ux_ - qy * uy_ - qz * uz_;
  const float nqx = qw * ux_ + qx * uw_ + qy * uz_ - qz * uy_;
  const float nqy = qw * uy_ + qy * uw_ + qz * ux_ - qx * uz_;
  const float nqz = qw * uz_ + qz * uw_ + qx * uy_ - qy * ux_;
  // p += v*dt + 0.5*a0*dt*dt
  const float fdt = (float)dt;
  const float px = F.p[0] + (fdt * F.v[0] + fdt * (fdt * (0.5f * a0x)));
  const float py = F.p[1] + (fdt * F.v[1] + fdt * (fdt * (0.5f * a0y)));
  const float pz = F.p[2] + (fdt * F.v[2] + fdt * (fdt * (0.5f * a0z)));
  // T = X0.get_RT() * lidar2baselink_T  (4x4 * 4x4, columns accumulated left to right)
  float R0[9];
  quat_to_rot(nqx, nqy, nqz, nqw, R0);
  const float X[16] = {R0[0], R0[1], R0[2], px, R0[3], R0[4], R0[5], py, R0[6], R0[7], R0[8], pz, 0.f, 0.f, 0.f, 1.f};
  float T[16];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float acc = X[i * 4 + 0] * mats[0 * 4 + j];
      acc = acc + X[i * 4 + 1] * mats[1 * 4 + j];
      acc = acc + X[i * 4 + 2] * mats[2 * 4 + j];
      acc = acc + X[i * 4 + 3] * mats[3 * 4 + j];
      T[i * 4 + j] = acc;
    }
  // world = T * [p,1]; the 4th component (T row 3) is carried like the reference does
  float wx_ = ((T[0] * p.x + T[1] * p.y) + T[2] * p.z) + T[3] * 1.f;
  float wy_ = ((T[4] * p.x + T[5] * p.y) + T[6] * p.z) + T[7] * 1.f;
  float wz_ = ((T[8] * p.x + T[9] * p.y) + T[10] * p.z) + T[11] * 1.f;
  float ww_ = ((T[12] * p.x + T[13] * p.y) + T[14] * p.z) + T[15] * 1.f;
  const float* Li = mats + 16;
  const float ox_ = ((Li[0] * wx_ + Li[1] * wy_) + Li[2] * wz_) + Li[3] * ww_;
  const float oy_ = ((Li[4] * wx_ + Li[5] * wy_) + Li[6] * wz_) + Li[7] * ww_;
  const float oz_ = ((Li[8] * wx_ + Li[9] * wy_) + Li[10] * wz_) + Li[11] * ww_;
  return make_float4(ox_, oy_, oz_, p.w);
}

__global__ __launch_bounds__(256) void deskew_kernel(const float4* __restrict__ in, const double* __restrict__ t,
                                                     int n, const DevFrame* __restrict__ frames, int nf,
                                                     const float* __restrict__ mats, float4* __restrict__ out_sorted,
                                                     float4* __restrict__ out_orig, double t_offset) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  // (sweep reference +- point time) + offset, as the host forms it (Localizer.cpp:741-805)
  const float4 o = deskew_point(in[k], t[k] + t_offset, frames, nf, mats);
  out_sorted[k] = o;                                            // w carries the original index
  out_orig[__float_as_uint(o.w)] = make_float4(o.x, o.y, o.z, 1.0f);
}

// FUSE: the whole measurement pass in ONE launch (fast path of flimo_match_reduce): every wave goes on from its queries'
// neighbours (fast path + tail) to their plane fit, residual and H row and to the H^T H reduction (fit_reduce_publish below).
struct FuseArgs {
  // second level (crowded regions): 0 = none; 1 = this launch follows a fine pre-pass and takes over the queries it settled
  // (record flag 4); 2 = this launch IS the fine pre-pass (G = the fine grid; a query is settled when its five are proven
  // inside its fine 3x3x3 block and that block lies inside the region the fine grid copies completely: cell in [qlo, qhi])
  int fine_mode;
  int qlo[3], qhi[3];
  TieList tl;
  MatchParams mp;
  FitIdx idx;
  double* partials;
  double2* granules;
  unsigned int* ticket;
  unsigned long long seq;
  DeskewArgs dk;       // dk.on: first pass of a scan, the deskew rides on this launch
  BookView book;       // book.node_c: exact distance ties are settled inside this launch, the reference's way (tie_repair_wave)
  ChainCtl ch;         // ch.S: a pass of a chained update (flimo_chain.h): the launch has one extra workgroup, and the workgroup
                       // that completes it goes on with the filter's algebra
  int spread;          // one-launch pass of a small scan: only every 2^spread-th query slot is taken (see knn5_pass)
};
template <int ROWS>
__device__ __forceinline__ void fit_reduce_publish(const float (&v)[16], bool owns_row, int row, float* sr, double* sa0, double* sa1,
                                                   double* sa2, double* sa3, unsigned int* s_last, const FitIdx& idx,
                                                   double* __restrict__ partials, double2* __restrict__ out_granules,
                                                   unsigned int* __restrict__ ticket, int* __restrict__ wl_count,
                                                   unsigned long long seq, const TieList& tl, int blk = -1, int nblk = -1);
FLIMO_DEV void fit_row(const GridView& G, const PoseMats& P, const MatchParams& mp, const int (&ids)[5], float gx, float gy, float gz,
                       float (&v)[16]);

// Exact float32 distance ties, settled where the rows are built (every reducing launch): the wave takes its tied queries one at a
// time -- every candidate with d <= d5 inside the proven block, ordered the reference's way by walking the device copy of its
// octree (tie_select_wave, further down) -- and the owner lane goes on with the reference's five; the record is rewritten, its tie
// bit cleared.  No list, no second launch, no host round trip.  Wave-wide; TS: the wave's own shared memory.
struct TieLds;
__device__ bool tie_select_wave(const GridView& G, const BookView& B, float qx, float qy, float qz, float dk, int k, TieLds& S,
                                uint32_t (&out_pos)[8], float (&out_d)[8]);
__device__ __forceinline__ void tie_repair_wave(const GridView& G, const BookView& B, NbrRec* __restrict__ nbr, bool mine_tied, int p,
                                                float gx, float gy, float gz, uint32_t d5bits, int (&ids)[5], TieLds& TS) {
  const int lane = threadIdx.x & 63;
  unsigned long long tm = __ballot(mine_tied);
  while (tm) {                                                 // wave-uniform
    const int tlane = __ffsll((long long)tm) - 1;
    tm &= tm - 1ull;
    const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gx), tlane));
    const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gy), tlane));
    const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gz), tlane));
    const float dk = __int_as_float(__builtin_amdgcn_readlane((int)d5bits, tlane));
    uint32_t pos[8];
    float d[8];
#pragma unroll
    for (int s_ = 0; s_ < 8; s_++) { pos[s_] = 0u; d[s_] = 0.f; }
    wave_lds_sync();
    const bool ok = tie_select_wave(G, B, qx, qy, qz, dk, 5, TS, pos, d);      // (the result is lane 0's)
    int got[5];
#pragma unroll
    for (int s_ = 0; s_ < 5; s_++) got[s_] = __builtin_amdgcn_readlane((int)pos[s_], 0);
    if (ok && lane == tlane) {
#pragma unroll
      for (int s_ = 0; s_ < 5; s_++) ids[s_] = got[s_];
      int4* rec = reinterpret_cast<int4*>(&nbr[p]);
      rec[0] = make_int4(got[0], got[1], got[2], got[3]);
      rec[1] = make_int4(got[4], 1, (int)d5bits, 1);           // same 5th distance; the tie is settled
      if (B.settled) atomicAdd(B.settled, 1ull);               // (statistics)
    }
    wave_lds_sync();
  }
}

#ifdef FLIMO_WPE
#define KNN_WPE __attribute__((amdgpu_waves_per_eu(FLIMO_WPE, FLIMO_WPE)))
#else
#define KNN_WPE
#endif
// The directory of the index (GridView) in the workgroup's shared memory: at most GRID_DIR_MAX 16-bit entries = 512 x 16 bytes, two
// 16-byte loads per thread (the host pads the directory to a multiple of 16 bytes).  Fetch and store are separate so that a launch
// can have the loads in flight while it fetches its query; the store is followed by a barrier (every thread of the workgroup).
struct DirRegs { uint4 v[GRID_DIR_MAX / 8 / 256]; };
__device__ __forceinline__ DirRegs grid_dir_fetch(const GridView& G) {
  const int n16 = (G.ntx * G.nty * G.ntz + 7) >> 3;
  const uint4* src = reinterpret_cast<const uint4*>(G.dir);
  DirRegs r;
#pragma unroll
  for (int k = 0; k < GRID_DIR_MAX / 8 / 256; k++)
    r.v[k] = (int)threadIdx.x + 256 * k < n16 ? src[threadIdx.x + 256 * k] : make_uint4(0u, 0u, 0u, 0u);
  return r;
}
__device__ __forceinline__ void grid_dir_store(uint16_t* s_dir, const DirRegs& r) {
  uint4* dst = reinterpret_cast<uint4*>(s_dir);
#pragma unroll
  for (int k = 0; k < GRID_DIR_MAX / 8 / 256; k++) dst[threadIdx.x + 256 * k] = r.v[k];
  __syncthreads();
}
// The pass's body, shared by the two kernels below: a host-driven pass gets its pose constants by value (kernel arguments), a
// chained pass (flimo_chain.h) reads them from the device filter's head -- same code, the constants come from another address.
template <int L, int SLOTS, bool FUSE, bool FINE = false>
__device__ __forceinline__ void knn5_pass(const GridView& G, uint16_t* dir, const DirRegs* dir_pending, const float4* __restrict__ scan_sorted, int n,
                                          const PoseMats& P, int max_ring, NbrRec* __restrict__ nbr,
                                          int* __restrict__ wl, int* __restrict__ wl_count,
                                          unsigned long long* __restrict__ cand_total, const float* __restrict__ prev_RT, int prev_valid,
                                          unsigned prev_probe_min, int tail, const FuseArgs& fa) {
  constexpr int QPB = 256 / L;          // queries per block; SLOTS = candidate loads in flight per lane
  const TieList tl = fa.tl;
  __shared__ WaveLds s_w[4];
  __shared__ unsigned int s_last;
  static_assert(sizeof(WaveLds) * 4 >= (size_t)IESKF_LDS_BYTES, "the filter's algebra runs in this workgroup's shared memory");
  // a chained pass's reducing launch has one workgroup more than the scan needs: it does the filter's measurement-independent half
  int nb = (int)gridDim.x;
  if constexpr (FUSE) {
    if (fa.ch.S) {
      nb -= 1;
      if ((int)blockIdx.x == nb) {
        double* big = reinterpret_cast<double*>(s_w);
        ik_extra_block(fa.ch, big, (int)threadIdx.x);
        return;
      }
    }
  }
  const int chunk = xcd_chunk(blockIdx.x, nb);
  // A small scan is spread over 2^spread times as many workgroups: only every 2^spread-th query slot of a wave is taken.  The fast
  // path is a chain of dependent round trips whatever the wave holds, and the tail shares a wave's 64 lanes among its pending
  // queries -- 2 lanes each in a full wave of 32, 16 each when the wave holds 4: a 2 000-point scan over a sparse map (every
  // query needs the rings 2 and 3) takes a fifth of the time.
  const int slot = chunk * QPB + threadIdx.x / L;
  const int p = slot >> fa.spread;
  const int sub = threadIdx.x % L;
  const bool in_range = p < n && (slot & ((1 << fa.spread) - 1)) == 0;          // no early exit: the tail below is a wave-wide phase
#if defined(FLIMO_TRACE) && FLIMO_TRACE == 2
  if (threadIdx.x < 16) s_trace_lds[threadIdx.x >> 3][threadIdx.x & 7] = 0ull;
  __syncthreads();
#endif
  TRACE(0, 0);

  float4 sp;
  if (!FINE && fa.dk.on) {
    // first pass of a scan: the deskew of this query's raw point (Localizer.cpp:822-843) instead of a dispatch of its own; the
    // result is what deskew_kernel would have stored, and is stored for the later passes, the map insert and the clouds
    const DevFrame* dk_frames = static_cast<const DevFrame*>(fa.dk.frames);
    const float* dk_mats = fa.dk.mats;
    const bool staged = fa.dk.stage_words > 0;                 // (launch-uniform)
    if (staged) {
      // the IMU frames came straight from the host (stores into fine-grained device memory, no copy launch ahead of this pass):
      // one look past the caches per workgroup, into shared memory nobody uses before the tail / the fit
      unsigned int* dst = reinterpret_cast<unsigned int*>(s_w);
      const unsigned int* src = reinterpret_cast<const unsigned int*>(fa.dk.frames);
      for (int i = (int)threadIdx.x; i < fa.dk.stage_words; i += 256)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __syncthreads();
      dk_frames = reinterpret_cast<const DevFrame*>(dst);
      dk_mats = reinterpret_cast<const float*>(dst) + (fa.dk.mats - static_cast<const float*>(fa.dk.frames));
    }
    sp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in_range) {
      sp = deskew_point(fa.dk.raw[p], fa.dk.t[p] + fa.dk.t_offset, dk_frames, fa.dk.nf, dk_mats);
      if (sub == 0) {
        fa.dk.out_sorted[p] = sp;
        fa.dk.out_orig[__float_as_uint(sp.w)] = make_float4(sp.x, sp.y, sp.z, 1.0f);
      }
    }
    if (staged) __syncthreads();                               // every wave is done with the frames before any of them reuses s_w
  } else {
    sp = scan_sorted[in_range ? p : 0];
  }
  // the query's record of the previous pass (its bound) travels with the point: issued here, it shares the point's round trip --
  // behind the directory's barrier (where the compiler leaves it) it was a round trip of its own in every workgroup's chain
  const bool want_rec = (prev_valid || (!FINE && fa.fine_mode == 1)) && in_range;
  int4 rec_early = make_int4(0, 0, 0, 0);
  if (want_rec) rec_early = reinterpret_cast<const int4*>(&nbr[p])[1];       // idx[4], flag, d5 bits, "d5 valid"
  // (the directory's loads were issued before the query's: into shared memory now, behind a barrier)
  if (dir_pending) grid_dir_store(dir, *dir_pending);
  float gx, gy, gz;
  xform4(P.RT, sp.x, sp.y, sp.z, gx, gy, gz);
  TRACE(0, 1);

  const float fx = (gx - G.ox) * G.inv_cell, fy = (gy - G.oy) * G.inv_cell, fz = (gz - G.oz) * G.inv_cell;
  // Upper bound of this pass's 5th-neighbour distance from the previous pass of the same scan: the five old
  // neighbours are still in the map, and the query moved by |g - g_old|, so d5 <= sqrt(d5_old) + |g - g_old|.
  // Cells farther than that cannot hold any of the five nearest points and are skipped (exactly, no heuristic).
  float b2 = INFINITY;                               // bound, squared, in cell units
  bool resolved = false;                             // settled by the fine pre-pass of this pass: nothing to search
  int4 pa_res = make_int4(0, 0, 0, 0), pb_res = make_int4(0, 0, 0, 0);
  if (want_rec) {
    const int4 pb = rec_early;
    // (a record settled by the fine pre-pass of THIS pass carries flag 4 and the pass number in the upper bits of w: a stale or
    //  never-written record cannot be mistaken for one)
    if (!FINE && fa.fine_mode == 1 && pb.y == 4 && ((uint32_t)pb.w >> 8) == (uint32_t)(fa.seq & 0xffffffull)) {
      resolved = true;
      pb_res = pb;
      pa_res = reinterpret_cast<const int4*>(&nbr[p])[0];
    } else if (prev_valid && pb.y == 1 && (pb.w & 1)) {
      float ox_, oy_, oz_;
      xform4(prev_RT, sp.x, sp.y, sp.z, ox_, oy_, oz_);
      const float ex = gx - ox_, ey = gy - oy_, ez = gz - oz_;
      const float moved = fl_sqrt(sum3(ex * ex, ey * ey, ez * ez));
      const float rc = ((fl_sqrt(__int_as_float(pb.z)) + moved) * (1.f + 1.0e-5f) + 1.0e-6f) * G.inv_cell;
      b2 = rc * rc * (1.f + 1.0e-5f);
    }
  }
  int flag = 0;
  u64 best[5] = {KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE};
  u64 sixth = KEY_NONE;
  bool tie = false;
  int cand = 0;
  bool tie_listed = false;
  int qcx = 0, qcy = 0, qcz = 0;
  if (resolved) {
    // the five (positions in the main map), their 5th distance and the tie bit as the pre-pass left them
    best[0] = (u64)(uint32_t)pa_res.x; best[1] = (u64)(uint32_t)pa_res.y; best[2] = (u64)(uint32_t)pa_res.z; best[3] = (u64)(uint32_t)pa_res.w;
    best[4] = ((u64)(uint32_t)pb_res.z << 32) | (u64)(uint32_t)pb_res.x;
    flag = 1;
    tie = (pb_res.w & 2) != 0;
    tie_listed = true;
  } else if (in_range && (fx == fx) && (fy == fy) && (fz == fz)) {
    const float lim = 1.0e9f;
    const float flx = floorf(fminf(fmaxf(fx, -lim), lim)), fly = floorf(fminf(fmaxf(fy, -lim), lim)),
                flz = floorf(fminf(fmaxf(fz, -lim), lim));
    const int cx = (int)flx - G.six, cy = (int)fly - G.siy, cz = (int)flz - G.siz;      // (GridView: the origin is fixed, the grid's corner is a cell shift)
    qcx = cx; qcy = cy; qcz = cz;
    const int ox_ = cx < 0 ? -cx : (cx >= G.nx ? cx - G.nx + 1 : 0);
    const int oy_ = cy < 0 ? -cy : (cy >= G.ny ? cy - G.ny + 1 : 0);
    const int oz_ = cz < 0 ? -cz : (cz >= G.nz ? cz - G.nz + 1 : 0);
    const int r0 = max(1, max(ox_, max(oy_, oz_)));
    if (r0 > max_ring) {
      flag = 0;                       // nothing within the gate distance
    } else if (r0 > 1) {
      flag = 2;                       // outside the grid but within reach: general search
    } else {
      // ---- range bounds of the 9 rows: nine 16-byte loads from the segment table (GridView) -- the two neighbouring entries that
      //      hold the rows' x range -- and three 12-byte loads of the rows' starts, all in one round trip.  The table is kept at
      //      FINE column resolution along x (G.xs columns per cell): without a bound the range is the three cells (columns
      //      (cx-1) xs .. (cx+2) xs), with the bound of the previous pass only the columns its ball can reach --
      //      |x_point - x_query| <= sqrt(b2) in cell units, widened by the rounding margin -- clipped to the three cells (the
      //      exactness proof below is about the 3x3x3 block).  The same columns for all nine rows; rows the ball cannot reach
      //      at all are dropped as before.  At most 3 xs <= 8 columns: the range's two ends lie in the entry of its first column
      //      or in the next one. ----
      // (32-bit table indices: the host keeps the table below 2^32 entries)
      const float rx = fminf(fmaxf(fx - flx, 0.f), 1.f), ry = fminf(fmaxf(fy - fly, 0.f), 1.f),
                  rz = fminf(fmaxf(fz - flz, 0.f), 1.f);
      const int maxdim = grid_maxdim(G);
      const float margin = 1.0e-3f + 4.0e-7f * (float)maxdim;
      int c0 = (cx - 1) * G.xs, c1 = (cx + 2) * G.xs;          // first column, one past the last
      if (prev_valid && b2 < 1.0e6f) {
        const float rb_ = fl_sqrt(b2) + margin;                 // reach along x in cell units (b2 is already inflated)
        const float fxs = (float)G.xs;
        c0 = max(c0, (int)floorf((fx - rb_) * fxs) - G.six * G.xs);
        c1 = min(c1, (int)floorf((fx + rb_) * fxs) - G.six * G.xs + 1);
      }
      c0 = min(max(c0, 0), G.nxf);
      c1 = min(max(c1, c0), G.nxf);
      const uint32_t py0 = (uint32_t)(cy + GRID_PAD - 1), pz0 = (uint32_t)(cz + GRID_PAD - 1);      // padded row (cy - 1, cz - 1)
      const uint32_t sg0 = (uint32_t)c0 >> 3;                   // entry of the range's first column
      const uint32_t tx0 = sg0 >> G.ts;
      // the nine rows' entries: tile (directory in shared memory) -> the two neighbouring entries of the row inside the tile
      uint32_t eidx[9];                  // (32-bit: the host keeps the pool below 2^32 entries)
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const uint32_t py_ = py0 + (uint32_t)(t % 3), pz_ = pz0 + (uint32_t)(t / 3);
        eidx[t] = (uint32_t)grid_entry_index(G, (uint32_t)dir[grid_dir_index(G, py_, pz_, sg0)], py_, pz_, sg0);
      }
      // First pass of a scan (no bound from a previous pass): a query whose OWN cell is crowded (raw sweeps inserted into the
      // map leave cells with tens to hundreds of points) first walks that cell alone; its 5th distance there is an upper bound
      // of the true one, and the other 26 cells are then walked only as far as that ball reaches (rows and end cells it cannot
      // reach are dropped, exactly as with the bound of a previous pass).
      const bool probe_on = L == 2 && !prev_valid && prev_probe_min != 0u;          // wave-uniform
      // (the entries hold POSITIONS in pts: rows do not move when other rows grow -- GridView -- so nothing has to be added; pad
      //  rows and tiles that do not exist read zeros: an empty range)
      // conservative distances (cell units) to the neighbouring rows; without a bound every row is in
      const float yd[3] = {fmaxf(ry - margin, 0.f), 0.f, fmaxf(1.f - ry - margin, 0.f)};
      const float zd[3] = {fmaxf(rz - margin, 0.f), 0.f, fmaxf(1.f - rz - margin, 0.f)};
      const float yd2[3] = {yd[0] * yd[0], 0.f, yd[2] * yd[2]}, zd2[3] = {zd[0] * zd[0], 0.f, zd[2] * zd[2]};
      U3 rbl[3], rbh[3];
      Seg2 own_e;                       // the centre row's two entries: the probe's own cell / column is read off them
      {
        Seg2 e[9];
#pragma unroll
        for (int dz = 0; dz < 3; dz++) {
#pragma unroll
          for (int k = 0; k < 3; k++) {
            e[3 * dz + k] = *reinterpret_cast<const Seg2*>(G.tiles + eidx[3 * dz + k]);
          }
        }
        own_e = e[4];
        const uint32_t k0 = (uint32_t)c0 & 7u, k1 = (uint32_t)c1 & 7u;
        const bool next = ((uint32_t)c1 >> 3) != sg0;           // the range's end lies in the second entry
        // (one test for all eighteen counts: escapes are segments of crowded cells)
        uint32_t lo9[9], hi9[9], esc = 0u;
#pragma unroll
        for (int t = 0; t < 9; t++) {
          const uint32_t hx = next ? e[t].x1 : e[t].x0, hy = next ? e[t].y1 : e[t].y0;
          esc |= e[t].x0 | hx;
          lo9[t] = seg_count_plain(e[t].x0, e[t].y0, k0);
          hi9[t] = seg_count_plain(hx, hy, k1);
        }
        if (__builtin_expect((int)esc < 0, 0)) {
#pragma unroll
          for (int t = 0; t < 9; t++) {
            lo9[t] = seg_count(e[t].x0, e[t].y0, k0, G.ovf);
            hi9[t] = seg_count(next ? e[t].x1 : e[t].x0, next ? e[t].y1 : e[t].y0, k1, G.ovf);
          }
        }
#pragma unroll
        for (int dz = 0; dz < 3; dz++) {
          rbl[dz].a = lo9[3 * dz]; rbl[dz].b = lo9[3 * dz + 1]; rbl[dz].c = lo9[3 * dz + 2];
          rbh[dz].a = hi9[3 * dz]; rbh[dz].b = hi9[3 * dz + 1]; rbh[dz].c = hi9[3 * dz + 2];
        }
      }
      uint32_t off[10], dl[9];          // dl[t] = lo[t] - off[t]: stream position -> map position
      off[0] = 0;
#pragma unroll
      for (int dz = 0; dz < 3; dz++) {
        const uint32_t sl[3] = {rbl[dz].a, rbl[dz].b, rbl[dz].c}, sh[3] = {rbh[dz].a, rbh[dz].b, rbh[dz].c};
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int t = 3 * dz + k;
          const bool row = !prev_valid || (yd2[k] + zd2[dz] <= b2);
          dl[t] = sl[k] - off[t];
          off[t + 1] = off[t] + (row ? sh[k] - sl[k] : 0u);
        }
      }
      // probing pays when the block is heavy (crowded region: every lane of the wave probes, so nobody waits for a neighbour's
      // full walk) and the own cell can give a bound at all.  Its bounds are read off the centre row's entries, which are here already:
      // they hold the own cell's range and let the second walk be clipped to cells.
      const bool heavy_block = probe_on && off[9] >= prev_probe_min && cx >= 0 && cx < G.nx;
      // position of column `col` (within the loaded range's two entries) in the centre row
      auto centre_pos = [&](int col) -> uint32_t {
        const bool nx_ = ((uint32_t)col >> 3) != sg0;
        return seg_count(nx_ ? own_e.x1 : own_e.x0, nx_ ? own_e.y1 : own_e.y0, (uint32_t)col & 7u, G.ovf);
      };
      uint32_t own_a = rbl[1].b, own_b = rbh[1].b;                  // the own cell's range (heavy blocks only; no load: the entries are here)
      if (heavy_block) {
        own_a = centre_pos(cx * G.xs);
        own_b = centre_pos((cx + 1) * G.xs);
      }
      // The probe walks the query's own COLUMN (the table's x resolution: half a cell by default) when that alone can give a bound,
      // its own cell otherwise: in a crowded cell the column holds a fraction of the points and its 5th distance is as good.
      uint32_t lo_own = own_a, hi_own = own_b;
      if (heavy_block && G.xs > 1 && hi_own - lo_own >= 2u * PROBE_MIN_OWN) {
        const int col = min(max((int)floorf(fx * (float)G.xs) - G.six * G.xs, cx * G.xs), (cx + 1) * G.xs - 1);
        const uint32_t ca = (col == cx * G.xs) ? lo_own : centre_pos(col);
        const uint32_t cb = (col + 1 == (cx + 1) * G.xs) ? hi_own : centre_pos(col + 1);
        if (cb - ca >= PROBE_MIN_OWN) { lo_own = ca; hi_own = cb; }
      }
      const uint32_t n_own = hi_own - lo_own;
      const bool two = heavy_block && n_own >= PROBE_MIN_OWN;        // the same in both lanes of the pair
      if (two) {                        // first walk: the own column / cell only (one segment)
        dl[0] = lo_own;
#pragma unroll
        for (int t = 1; t < 10; t++) off[t] = n_own;
      }
      // A wave in which SOME queries probe walks twice (probe, then the rest of the ball) while the others would sit out the second
      // walk after a full first one: those walk the first half of their stream now and the second half then -- both walks are about
      // half as long for every lane of the wave (the two halves are disjoint point sets, merged like the probe's)
      const bool split_walk = L == 2 && probe_on && __any(two) && !two;
      const uint32_t half = split_walk ? (off[9] + 1u) / 2u : 0u;
      const uint32_t total = split_walk ? half : off[9];
      TRACE(0, 2);
      // ---- flattened candidate stream, branch-free body ----
      const double none = __longlong_as_double((long long)KEY_NONE);
      double k5[6] = {none, none, none, none, none, none};
      {
        const uint32_t last = total - 1u;
        // full bodies while more than half of a body's slots are live for this lane, then one half body
        uint32_t s0 = (uint32_t)sub;
        for (; s0 + (SLOTS / 2) * L < total; s0 += SLOTS * L)
          knn5_body<L, SLOTS, FINE>(G.pts, s0, total, last, off, dl, gx, gy, gz, k5);
        if (s0 < total) knn5_body<L, (SLOTS >= 2 ? SLOTS / 2 : 1), FINE>(G.pts, s0, total, last, off, dl, gx, gy, gz, k5);
      }
      cand = total > (uint32_t)sub ? (int)((total - (uint32_t)sub + L - 1) / L) : 0;
#pragma unroll
      for (int i = 0; i < 5; i++) best[i] = (u64)__double_as_longlong(k5[i]);
      sixth = (u64)__double_as_longlong(k5[5]);
      TRACE(0, 3);
#if defined(FLIMO_TRACE) && FLIMO_TRACE != 2
      if (blockIdx.x < 16384) {   // developer statistics: accumulated block candidates (all passes), CU id
        if (sub == 0) atomicAdd(&g_trace[0][blockIdx.x * 8 + 6], (unsigned long long)total);
        if (threadIdx.x == 0) g_trace[0][blockIdx.x * 8 + 7] = __smid();
      }
#endif
      // ---- group result: the five, and the best of the rest ----
      if constexpr (L == 2) {
        double c[6];
        key_pair_merge6(k5, c);
        if (__any(two)) {
          // second walk of the queries that probed: the rest of the probe's ball, laid out exactly like a pass that HAS a bound
          // (the rows the ball reaches, the columns it reaches -- one more round trip for the two x planes that bound them) minus
          // the probed range.  Ten segments: eight rows, and the centre row on both sides of the probed range.
          uint32_t off2[11], dl2[10];
          off2[0] = 0;
          if (two) {
            const float d5p = __uint_as_float((uint32_t)((u64)__double_as_longlong(c[4]) >> 32));      // n_own >= 5: finite
            const float rc = (fl_sqrt(d5p) * (1.f + 1.0e-5f) + 1.0e-6f) * G.inv_cell;
            const float bb = rc * rc * (1.f + 1.0e-5f);                                              // bound, cell units squared
            int h0 = (cx - 1) * G.xs, h1 = (cx + 2) * G.xs;
            if (bb < 1.0e6f) {
              const float rb_ = fl_sqrt(bb) + margin;
              const float fxs = (float)G.xs;
              h0 = max(h0, (int)floorf((fx - rb_) * fxs) - G.six * G.xs);
              h1 = min(h1, (int)floorf((fx + rb_) * fxs) - G.six * G.xs + 1);
            }
            h0 = min(max(h0, 0), G.nxf);
            h1 = min(max(h1, h0), G.nxf);
            // (h0 .. h1 lie inside the first walk's columns: the same two entries per row, read again -- they are not kept in
            //  registers across the first walk)
            U3 hl[3], hh[3];
            {
              const uint32_t k0 = (uint32_t)h0 & 7u, k1 = (uint32_t)h1 & 7u;
              const bool n0 = ((uint32_t)h0 >> 3) != sg0, n1 = ((uint32_t)h1 >> 3) != sg0;
              Seg2 e[9];
#pragma unroll
              for (int t = 0; t < 9; t++)
                e[t] = *reinterpret_cast<const Seg2*>(G.tiles + eidx[t]);
              uint32_t lo9[9], hi9[9];
#pragma unroll
              for (int t = 0; t < 9; t++) {
                lo9[t] = seg_count(n0 ? e[t].x1 : e[t].x0, n0 ? e[t].y1 : e[t].y0, k0, G.ovf);
                hi9[t] = seg_count(n1 ? e[t].x1 : e[t].x0, n1 ? e[t].y1 : e[t].y0, k1, G.ovf);
              }
#pragma unroll
              for (int dz = 0; dz < 3; dz++) {
                hl[dz].a = lo9[3 * dz]; hl[dz].b = lo9[3 * dz + 1]; hl[dz].c = lo9[3 * dz + 2];
                hh[dz].a = hi9[3 * dz]; hh[dz].b = hi9[3 * dz + 1]; hh[dz].c = hi9[3 * dz + 2];
              }
            }
#pragma unroll
            for (int dz = 0; dz < 3; dz++) {
              const uint32_t sl_[3] = {hl[dz].a, hl[dz].b, hl[dz].c}, sh_[3] = {hh[dz].a, hh[dz].b, hh[dz].c};
#pragma unroll
              for (int k = 0; k < 3; k++) {
                const bool row = yd2[k] + zd2[dz] <= bb;
                if (dz == 1 && k == 1) {
                  // centre row: [sl, probed range) and (probed range, sh), each clipped to the ball's columns (a probed CELL may
                  // stick out of them; what sticks out has been walked already)
                  const uint32_t le = min(max(lo_own, sl_[k]), sh_[k]), rs = min(max(hi_own, sl_[k]), sh_[k]);
                  dl2[4] = sl_[k] - off2[4];
                  off2[5] = off2[4] + (le - sl_[k]);
                  dl2[5] = rs - off2[5];
                  off2[6] = off2[5] + (sh_[k] - rs);
                } else {
                  const int t = 3 * dz + k + ((3 * dz + k) > 4 ? 1 : 0);
                  dl2[t] = sl_[k] - off2[t];
                  off2[t + 1] = off2[t] + (row ? sh_[k] - sl_[k] : 0u);
                }
              }
            }
          } else if (split_walk) {
            // second half of this query's stream: positions [half, off[9]) of the nine segments, shifted down by `half`
#pragma unroll
            for (int t = 0; t < 9; t++) { off2[t + 1] = off[t + 1] > half ? off[t + 1] - half : 0u; dl2[t] = dl[t] + half; }
            off2[10] = off2[9]; dl2[9] = 0u;
          } else {
#pragma unroll
            for (int t = 0; t < 10; t++) { off2[t + 1] = 0u; dl2[t] = 0u; }
          }
          const uint32_t total2 = off2[10];
          double kb[6] = {none, none, none, none, none, none};
          {
            const uint32_t last = total2 - 1u;
            uint32_t s0 = (uint32_t)sub;
            for (; s0 + (SLOTS / 2) * L < total2; s0 += SLOTS * L)
              knn5_body<L, SLOTS, FINE>(G.pts, s0, total2, last, off2, dl2, gx, gy, gz, kb);
            if (s0 < total2) knn5_body<L, (SLOTS >= 2 ? SLOTS / 2 : 1), FINE>(G.pts, s0, total2, last, off2, dl2, gx, gy, gz, kb);
          }
          cand += total2 > (uint32_t)sub ? (int)((total2 - (uint32_t)sub + L - 1) / L) : 0;
          double cb[6], cc[6];
          key_pair_merge6(kb, cb);
          key_merge6(c, cb, cc);                       // own cell + rest of the ball / the two halves: disjoint point sets, unique keys
          if (two || split_walk) {
#pragma unroll
            for (int i = 0; i < 6; i++) c[i] = cc[i];
          }
        }
#pragma unroll
        for (int i = 0; i < 5; i++) best[i] = (u64)__double_as_longlong(c[i]);
        sixth = (u64)__double_as_longlong(c[5]);
      } else if (L > 1) {
        u64 mine[6];
#pragma unroll
        for (int i = 0; i < 5; i++) mine[i] = best[i];
        mine[5] = sixth;
#pragma unroll
        for (int r = 0; r < 6; r++) {
          double md = __longlong_as_double((long long)mine[0]);
#pragma unroll
          for (int o = 1; o < L; o <<= 1) md = key_min(md, __shfl_xor(md, o, 64));
          const u64 m = (u64)__double_as_longlong(md);
          if (r < 5) best[r] = m; else sixth = m;
          if (mine[0] == m) { mine[0] = mine[1]; mine[1] = mine[2]; mine[2] = mine[3]; mine[3] = mine[4]; mine[4] = mine[5]; mine[5] = KEY_NONE; }
        }
      }
      // ---- exactness: the 5-ball must lie inside the 3x3x3 block ----
      const float edge = fminf(fminf(fminf(rx, 1.f - rx), fminf(ry, 1.f - ry)), fminf(rz, 1.f - rz));
      const float rg = (1.f + edge - margin) * G.cell;
      const float d5 = __uint_as_float((uint32_t)(best[4] >> 32));
      const bool have5 = d5 < INFINITY;
      const bool covers = (cx - 1 <= 0) && (cx + 1 >= G.nx - 1) && (cy - 1 <= 0) && (cy + 1 >= G.ny - 1) &&
                          (cz - 1 <= 0) && (cz + 1 >= G.nz - 1);
      if (have5 && (covers || d5 <= rg * rg * (1.f - 1.0e-6f))) {
        flag = 1;
        // exactly tied float32 distances among the five or between the 5th and the 6th: the reference's choice / order follows
        // its octree's visiting order (resolved by tie_kernel)
        tie = key_has_tie(best, sixth);
      }
      else if (covers) flag = 0;                 // the whole map holds fewer than 5 points
      else flag = (max_ring > 1) ? 2 : 0;
    }
  }
  if (cand_total) {
    int c = cand;
#pragma unroll
    for (int o = 1; o < L; o <<= 1) c += __shfl_xor(c, o, 64);
    if (sub == 0 && in_range) atomicAdd(cand_total, (unsigned long long)c);
  }
  if constexpr (FINE) {
    // fine pre-pass: settle what is proven inside the completely copied region, leave everything else to the main launch
    const bool inside = qcx >= fa.qlo[0] && qcx <= fa.qhi[0] && qcy >= fa.qlo[1] && qcy <= fa.qhi[1] && qcz >= fa.qlo[2] && qcz <= fa.qhi[2];
    if (in_range && sub == 0 && flag == 1 && inside) {
      int4 a, b;
      a.x = (int)(uint32_t)best[0]; a.y = (int)(uint32_t)best[1]; a.z = (int)(uint32_t)best[2]; a.w = (int)(uint32_t)best[3];
      b.x = (int)(uint32_t)best[4]; b.y = 4; b.z = (int)(uint32_t)(best[4] >> 32);
      b.w = (int)(1u | (tie ? 2u : 0u) | ((uint32_t)(fa.seq & 0xffffffull) << 8));
      int4* o = reinterpret_cast<int4*>(&nbr[p]);
      o[0] = a;
      o[1] = b;
      if (tie && tl.list) { const unsigned slot = atomicAdd(tl.count, 1u); if (slot < tl.cap) tl.list[slot] = p; }
    }
    return;
  }
  TRACE(0, 4);
  // pending queries (flag 2) are finished right here by this wave when the gate needs at most TAIL_MAX_RING rings
  // (`tail`); otherwise (wider gates, the developer's crowded-block hand-over) they go to the worklist kernels
  const bool pend_tail = tail && in_range && flag == 2;
  if (in_range && sub == 0 && !pend_tail) {
    int4 a, b;
    a.x = (int)(uint32_t)best[0]; a.y = (int)(uint32_t)best[1]; a.z = (int)(uint32_t)best[2]; a.w = (int)(uint32_t)best[3];
    b.x = (int)(uint32_t)best[4]; b.y = min(flag, 2); b.z = (int)(uint32_t)(best[4] >> 32); b.w = ((flag == 1) ? 1 : 0) | (tie ? 2 : 0);   // d5 for the next pass; bit 1: tie
    if (tie && !tie_listed && tl.list) { const unsigned slot = atomicAdd(tl.count, 1u); if (slot < tl.cap) tl.list[slot] = p; }
    int4* o = reinterpret_cast<int4*>(&nbr[p]);
    o[0] = a;
    o[1] = b;
    if (flag >= 2) {
      // worklist entry = everything the widening wave needs to start (no dependent loads on its side):
      // query index, world position, 5th squared distance found inside the 3x3x3 block (+inf: none)
      const int slot = atomicAdd(wl_count, 1);
      int4* e = reinterpret_cast<int4*>(wl) + 2 * (size_t)slot;
      e[0] = make_int4(p, __float_as_int(gx), __float_as_int(gy), __float_as_int(gz));
      e[1] = make_int4((int)(uint32_t)(best[4] >> 32), 2, __float_as_int(b2), 0);
    }
  }
  WaveLds& W = s_w[threadIdx.x >> 6];
  if (tail) knn5_tail(G, dir, max_ring, nbr, pend_tail && sub == 0, p, gx, gy, gz, (uint32_t)(best[4] >> 32), b2, W, wl_count, cand_total, FUSE, tl);
  TRACE(0, 5);
  if constexpr (FUSE) {
    // ---- fit + reduction of this wave's queries (one row per query, computed by the pair's first lane) ----
    const int lane = threadIdx.x & 63;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = 0.f;
    wave_lds_sync();                                           // the tail's hand-backs are visible; its tables are free
    int ids[5] = {(int)(uint32_t)best[0], (int)(uint32_t)best[1], (int)(uint32_t)best[2], (int)(uint32_t)best[3], (int)(uint32_t)best[4]};
    int fl = flag;
    bool tied = tie;
    uint32_t d5b = (uint32_t)(best[4] >> 32);
    if (sub == 0 && in_range && pend_tail) {
      const int4* rr = reinterpret_cast<const int4*>(W.res[lane]);
      const int4 a = rr[0], b = rr[1];
      ids[0] = a.x; ids[1] = a.y; ids[2] = a.z; ids[3] = a.w; ids[4] = b.x;
      fl = b.y;
      tied = (b.w & 2) != 0;
      d5b = (uint32_t)b.z;
    }
    if (fa.book.node_c)
      tie_repair_wave(G, fa.book, nbr, sub == 0 && in_range && fl == 1 && tied, p, gx, gy, gz, d5b, ids, *reinterpret_cast<TieLds*>(W.tile));
    TRACE(1, 1);
    if (sub == 0 && in_range) {
      if (fl == 1 && __float_as_uint(sp.w) < (uint32_t)fa.mp.n_queries) fit_row(G, P, fa.mp, ids, gx, gy, gz, v);
    }
    TRACE(1, 3);
    wave_lds_sync();                                           // every lane is done with W.res before the tile overwrites the tables
    fit_reduce_publish<64 / L>(v, sub == 0, lane / L, W.tile, s_w[0].acc, s_w[1].acc, s_w[2].acc, s_w[3].acc, &s_last, fa.idx,
                               fa.partials, fa.granules, fa.ticket, wl_count, fa.seq, tl, (int)blockIdx.x, nb);
  }
  TRACE_FLUSH();
}

template <int L, int SLOTS, bool FUSE, bool FINE = false>
__global__ __launch_bounds__(256) KNN_WPE void knn5_kernel(GridView G, const float4* __restrict__ scan_sorted, int n,
                                                   PoseMats P, int max_ring, NbrRec* __restrict__ nbr,
                                                   int* __restrict__ wl, int* __restrict__ wl_count,
                                                   unsigned long long* __restrict__ cand_total, PrevPass prev, int tail,
                                                   FuseArgs fa) {
  __shared__ __align__(16) uint16_t s_dir[GRID_DIR_MAX];
  const DirRegs dr = grid_dir_fetch(G);
  // the pose constants go through shared memory (like a chained pass's, which reads them from the filter's head): 66 scalars that
  // would otherwise sit in registers from the first instruction to the fit -- the barrier behind the directory's store covers them
  __shared__ PoseMats s_pose;
  static_assert(sizeof(PoseMats) % 4 == 0 && sizeof(PoseMats) / 4 <= 256, "one word per thread");
  if (threadIdx.x < sizeof(PoseMats) / 4)
    reinterpret_cast<uint32_t*>(&s_pose)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&P)[threadIdx.x];
  knn5_pass<L, SLOTS, FUSE, FINE>(G, s_dir, &dr, scan_sorted, n, s_pose, max_ring, nbr, wl, wl_count, cand_total, prev.RT, prev.valid, prev.probe_min, tail, fa);
}
// How a launch of a chained pass gets the filter's head: a copy in the workgroup's shared memory (one word per thread), which the pass
// reads its constants from.  wait_epoch == 0: the head was stored by an earlier launch on this stream (the algebra as a launch of
// its own; a later launch of a pass whose first one waited): plain loads.  Otherwise a resident workgroup publishes it while this
// launch is already placed: one thread polls head.epoch in device memory (s_sleep between looks, bounded by the wall clock), then
// the workgroup reads the head past the caches.  nullptr: the chain has ended (or the wait ran out): leave.
__device__ __forceinline__ const ChainHead* chain_enter(const ChainHead* __restrict__ H, unsigned int wait_epoch, unsigned int end_code) {
  constexpr int NW = (int)(sizeof(ChainHead) / 4);
  __shared__ unsigned int s_head[NW];
  __shared__ int s_go;
  if (wait_epoch != 0u) {
    if (threadIdx.x == 0) {
      const unsigned long long t0 = wall_clock64();
      int go = 0;
      // (the algebra takes a few microseconds from the moment the last pass delivered: hundreds of workgroups looking at one word
      //  every 60 ns would stand in its way -- a first look, a nap of 1.5 us (the host's algebra of a pipelined loop takes two),
      //  then a look every quarter of a microsecond)
      // Whether the launch runs or leaves is ONE decision for all of its workgroups.  Every workgroup looks at the host's word and
      // goes the moment it sees its constants published (no atomics, nothing between the publish and the start); only workgroup 0
      // may declare the wait over, and it does so in two steps through the head's decision word: "pending", a grace period longer
      // than a store takes to become visible, then -- after another look at the host's word -- "leave", or "go" when the publish
      // arrived in between.  A workgroup that finds "pending" waits for the verdict; one that finds "leave" leaves even if it
      // also sees the publish (the host reads the same word and launches the pass again).  So a host thread descheduled around its
      // publish can no longer leave half a launch running and half of it gone (tickets never completed, sums that cannot come).
      // The word holds the wait's own number: go = number, pending = number | bit 31, leave = number without bit 30 | bit 31.
      unsigned int* dec = const_cast<unsigned int*>(&H->decision);
      const unsigned int v_go = wait_epoch, v_pend = wait_epoch | 0x80000000u, v_no = (wait_epoch & 0x3fffffffu) | 0x80000000u;
      for (int look = 0;; look++) {
        const unsigned int d = __hip_atomic_load(dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned int e = __hip_atomic_load(&H->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (written by the HOST)
        if (d == v_no) break;
        if (d == v_pend) { __builtin_amdgcn_s_sleep(8); continue; }
        if (e == wait_epoch || d == v_go) { go = 1; break; }
        if (e == end_code) break;                              // (told to leave by the host: every workgroup reads the same)
        const unsigned long long waited = wall_clock64() - t0;                                                // 100 MHz
        if (blockIdx.x == 0 && waited > (unsigned long long)CH_POLL_MS * 100000ull) {
          __hip_atomic_store(dec, v_pend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          for (int nap = 0; nap < 4; nap++) __builtin_amdgcn_s_sleep(127);                                   // >= 10 us
          const unsigned int e2 = __hip_atomic_load(&H->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          go = e2 == wait_epoch ? 1 : 0;
          __hip_atomic_store(dec, go ? v_go : v_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        if (waited > 4ull * (unsigned long long)CH_POLL_MS * 100000ull) break;      // (backstop: workgroup 0 decides long before)
        if (look == 0) __builtin_amdgcn_s_sleep(48);
        else __builtin_amdgcn_s_sleep(8);
      }
      s_go = go;
    }
    __syncthreads();
    if (!s_go) return nullptr;
    if ((int)threadIdx.x < NW)
      s_head[threadIdx.x] = __hip_atomic_load(reinterpret_cast<const unsigned int*>(H) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    if ((int)threadIdx.x < NW) s_head[threadIdx.x] = reinterpret_cast<const unsigned int*>(H)[threadIdx.x];
  }
  __syncthreads();
  const ChainHead* Lh = reinterpret_cast<const ChainHead*>(s_head);
  return Lh->status != 0 ? nullptr : Lh;
}
// The same pass inside a chain: pose constants and the bound's reference pose from the device filter (written by the algebra
// queued before this launch, or by the resident one while this launch waits); nothing to do once the chain has ended.
template <int L, int SLOTS, bool FUSE, bool FINE = false>
__global__ __launch_bounds__(256) KNN_WPE void knn5_chain_kernel(GridView G, const float4* __restrict__ scan_sorted, int n,
                                                   const ChainHead* __restrict__ H, int max_ring, NbrRec* __restrict__ nbr,
                                                   int* __restrict__ wl, int* __restrict__ wl_count,
                                                   unsigned long long* __restrict__ cand_total, int prev_valid, unsigned probe_min, int tail,
                                                   FuseArgs fa, unsigned int wait_epoch) {
  // (the directory of the map's index is copied while the launch may still be waiting for its pose: the map does not change
  //  between the passes of a scan)
  __shared__ __align__(16) uint16_t s_dir[GRID_DIR_MAX];
  {
    const DirRegs dr = grid_dir_fetch(G);
    grid_dir_store(s_dir, dr);
  }
  const ChainHead* Lh = chain_enter(H, wait_epoch, fa.ch.end_code);
  if (!Lh) return;
  knn5_pass<L, SLOTS, FUSE, FINE>(G, s_dir, nullptr, scan_sorted, n, Lh->pose, max_ring, nbr, wl, wl_count, cand_total, Lh->prev_RT, prev_valid, probe_min, tail, fa);
}

// Widening for the worklist (rare): ONE WAVE per query.  The (2r+1)^2 <= 49 rows of the ring-r block
// are owned by one lane each (range bounds fetched in a single round trip), a wave prefix sum
// flattens their candidates, every lane scans a strided share (four loads in flight) and the best 5 are
// extracted with wave-wide min reductions.  r starts at the ring the fast path's own 5th distance asks for
// (at least 2) and jumps to the ring that proves exactness, never beyond max_ring (<= 3 here; the host
// falls back to the general kernel for larger gates).
// (blk / nblk: this block's number among the launch's blocks)
__device__ __forceinline__ void widen_body(const GridView& G, int max_ring, NbrRec* __restrict__ nbr,
                                           const int* __restrict__ wl, const int* __restrict__ wl_count,
                                           unsigned long long* __restrict__ cand_total, int first_ring, const TieList& tl,
                                           int blk, int nblk, uint32_t (*s_off)[65], uint32_t (*s_lo)[64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int maxdim = grid_maxdim(G);
  const float margin = 1.0e-3f + 4.0e-7f * (float)maxdim;
  // the first entry of this wave is fetched together with the count (one round trip); slots beyond the count hold
  // stale entries of earlier passes and are never used
  int w = blk * 4 + wave;
  const int4* entries = reinterpret_cast<const int4*>(wl);
  int4 e0 = entries[2 * (size_t)w], e1 = entries[2 * (size_t)w + 1];
  const int count = *wl_count;
  for (; w < count; w += nblk * 4) {
    if (w != blk * 4 + wave) { e0 = entries[2 * (size_t)w]; e1 = entries[2 * (size_t)w + 1]; }
    const int p = e0.x;
    const float gx = __int_as_float(e0.y), gy = __int_as_float(e0.z), gz = __int_as_float(e0.w);
    const int hint_bits = e1.x;                        // 5th squared distance inside the 3x3x3 block (+inf: none)
    const float fx = (gx - G.ox) * G.inv_cell, fy = (gy - G.oy) * G.inv_cell, fz = (gz - G.oz) * G.inv_cell;
    const float flx = floorf(fminf(fmaxf(fx, -1.0e9f), 1.0e9f)), fly = floorf(fminf(fmaxf(fy, -1.0e9f), 1.0e9f)),
                flz = floorf(fminf(fmaxf(fz, -1.0e9f), 1.0e9f));
    const int cx = (int)flx - G.six, cy = (int)fly - G.siy, cz = (int)flz - G.siz;      // (GridView: the origin is fixed, the grid's corner is a cell shift)
    const float rx = fminf(fmaxf(fx - flx, 0.f), 1.f), ry = fminf(fmaxf(fy - fly, 0.f), 1.f),
                rz = fminf(fmaxf(fz - flz, 0.f), 1.f);
    const float edge = fminf(fminf(fminf(rx, 1.f - rx), fminf(ry, 1.f - ry)), fminf(rz, 1.f - rz));
    u64 best[5] = {KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE};
    u64 sixth = KEY_NONE;
    int flag = 0;
    int cand = 0;
    int r = first_ring;                                // no hint: straight to the gate's ring (one search instead of two)
    // The ball that must hold the five (cell units squared, +inf: none known): the bound of the previous pass of this scan that the
    // k-NN launch handed over, and the 3x3x3 block's own 5th distance.  Rows the ball cannot reach are not walked and the others
    // only over the cells it reaches -- exactly as the in-kernel tail does (round 4 walked whole ring-2/3 blocks here: 56-117 KB
    // per query against 1 KB of candidates that can matter, profiles/r04/pmc_hbm_regime.json).
    float bnd2 = __int_as_float(e1.z);
    if (!(bnd2 >= 0.f)) bnd2 = INFINITY;
    {
      const float hint = __int_as_float(hint_bits);
      if (hint >= 0.f && hint < INFINITY) {            // an upper bound of the true 5th distance: go straight to its ring
        const float need = fl_sqrt(hint) * G.inv_cell * (1.f + 4.0e-6f) - edge + margin;
        r = max(2, (int)ceilf(fminf(need, 1.0e9f)));
        const float rc = (fl_sqrt(hint) * (1.f + 1.0e-5f) + 1.0e-6f) * G.inv_cell;
        bnd2 = fminf(bnd2, rc * rc * (1.f + 1.0e-5f));
      }
      r = min(r, max_ring);
    }
    for (;;) {
      const int side = 2 * r + 1;
      // one row per lane
      uint32_t lo = 0, len = 0;
      if (lane < side * side) {
        const int dy = (lane % side) - r, dz = (lane / side) - r;
        const int yy = cy + dy, zz = cz + dz;
        const float a_ = fmaxf(slab_dist(dy, ry) - margin, 0.f), b_ = fmaxf(slab_dist(dz, rz) - margin, 0.f);
        const float dyz2 = a_ * a_ + b_ * b_;
        int x0 = max(cx - r, 0), x1 = min(cx + r, G.nx - 1);
        if (bnd2 < 1.0e18f) {
          // cells of the row the ball can reach: offset +d is (d - rx) away, offset -d is (rx + d - 1) away
          const float xr = fl_sqrt(fmaxf(bnd2 - dyz2, 0.f)) * (1.f + 1.0e-6f) + margin + 1.0e-4f;
          const int dr = (int)fminf(floorf(fminf(xr + rx, 1.0e6f)), (float)r);
          const int dl = (int)fminf(floorf(fminf(xr + (1.f - rx), 1.0e6f)), (float)r);
          x0 = max(cx - dl, 0); x1 = min(cx + dr, G.nx - 1);
        }
        if (dyz2 <= bnd2 && yy >= 0 && yy < G.ny && zz >= 0 && zz < G.nz && x0 <= x1) {
          uint32_t hi_;
          grid_row_range(G, G.dir, yy, zz, x0 * G.xs, (x1 + 1) * G.xs, lo, hi_);
          len = hi_ - lo;
        }
      }
      // inclusive prefix sum over the wave
      uint32_t inc = len;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(inc, o, 64);
        if (lane >= o) inc += v;
      }
      const uint32_t total = __shfl(inc, 63, 64);
      s_off[wave][lane + 1] = inc;
      if (lane == 0) s_off[wave][0] = 0;
      s_lo[wave][lane] = lo;
      __builtin_amdgcn_wave_barrier();
      // strided share of the flattened candidates; rows ascend in memory, so a lane meets ascending map positions
      const double none = __longlong_as_double((long long)KEY_NONE);
      double k5[6] = {none, none, none, none, none, none};
      const uint32_t last = total - 1u;
      for (uint32_t s0 = (uint32_t)lane; s0 < total; s0 += 4 * 64) {
        float4 q[4];
        uint32_t id[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t s = min(s0 + 64u * u, last);
          int a = 0;                                   // row a with off[a] <= s < off[a + 1]
#pragma unroll
          for (int step = 32; step >= 1; step >>= 1) a = (s_off[wave][a + step] <= s) ? a + step : a;
          id[u] = s_lo[wave][a] + (s - s_off[wave][a]);
          q[u] = G.pts[id[u]];
        }
        asm volatile("" : "+v"(q[0].x), "+v"(q[0].y), "+v"(q[0].z), "+v"(q[1].x), "+v"(q[1].y), "+v"(q[1].z),
                          "+v"(q[2].x), "+v"(q[2].y), "+v"(q[2].z), "+v"(q[3].x), "+v"(q[3].y), "+v"(q[3].z));
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const bool live = s0 + 64u * u < total;
          const float d = sqdist3(gx, gy, gz, q[u].x, q[u].y, q[u].z);
          best6_insert(k5, key_make(live ? d : INFINITY, live ? id[u] : 0xffffffffu));
        }
      }
      cand += total > (uint32_t)lane ? (int)((total - (uint32_t)lane + 63u) / 64u) : 0;
      __builtin_amdgcn_wave_barrier();
      u64 mine[6];
#pragma unroll
      for (int i = 0; i < 6; i++) mine[i] = (u64)__double_as_longlong(k5[i]);
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const double md = key_group_min(__longlong_as_double((long long)mine[0]), 64, lane);
        const u64 m = (u64)__double_as_longlong(md);
        if (k < 5) best[k] = m; else sixth = m;
        if (mine[0] == m) { mine[0] = mine[1]; mine[1] = mine[2]; mine[2] = mine[3]; mine[3] = mine[4]; mine[4] = mine[5]; mine[5] = KEY_NONE; }
      }
      const float rg = ((float)r + edge - margin) * G.cell;
      const float d5 = __uint_as_float((uint32_t)(best[4] >> 32));
      const bool have5 = d5 < INFINITY;
      const bool covers = (cx - r <= 0) && (cx + r >= G.nx - 1) && (cy - r <= 0) && (cy + r >= G.ny - 1) &&
                          (cz - r <= 0) && (cz + r >= G.nz - 1);
      if (have5 && (covers || d5 <= rg * rg * (1.f - 1.0e-6f))) { flag = 1; break; }
      if (covers || r >= max_ring) { flag = 0; break; }
      int rn = r + 1;
      if (have5) {
        const float need = fl_sqrt(d5) * G.inv_cell * (1.f + 4.0e-6f) - edge + margin;
        rn = max(r + 1, (int)ceilf(fminf(need, 1.0e9f)));
        const float rc = (fl_sqrt(d5) * (1.f + 1.0e-5f) + 1.0e-6f) * G.inv_cell;
        bnd2 = fminf(bnd2, rc * rc * (1.f + 1.0e-5f));
      }
      r = min(rn, max_ring);
    }
    if (cand_total) {
      int c = cand;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) c += __shfl_xor(c, o, 64);
      if (lane == 0) atomicAdd(cand_total, (unsigned long long)c);
    }
    if (lane == 0) {
      int4 a, b;
      a.x = (int)(uint32_t)best[0]; a.y = (int)(uint32_t)best[1]; a.z = (int)(uint32_t)best[2]; a.w = (int)(uint32_t)best[3];
      const bool tie = flag == 1 && key_has_tie(best, sixth);
      b.x = (int)(uint32_t)best[4]; b.y = flag; b.z = (int)(uint32_t)(best[4] >> 32); b.w = ((flag == 1) ? 1 : 0) | (tie ? 2 : 0);
      int4* o = reinterpret_cast<int4*>(&nbr[p]);
      o[0] = a;
      o[1] = b;
      if (tie && tl.list) { const unsigned slot = atomicAdd(tl.count, 1u); if (slot < tl.cap) tl.list[slot] = p; }
    }
  }
}
__global__ __launch_bounds__(256) void widen_kernel(GridView G, const float4* __restrict__ scan_sorted, PoseMats P,
                                                    int max_ring, NbrRec* __restrict__ nbr, const int* __restrict__ wl,
                                                    const int* __restrict__ wl_count,
                                                    unsigned long long* __restrict__ cand_total, int first_ring, TieList tl,
                                                    const ChainHead* __restrict__ H) {
  __shared__ uint32_t s_off[4][65];
  __shared__ uint32_t s_lo[4][64];
  if (H && H->status != 0) return;                   // a chained pass after the chain has ended
  widen_body(G, max_ring, nbr, wl, wl_count, cand_total, first_ring, tl, (int)blockIdx.x, (int)gridDim.x, s_off, s_lo);
}

// general ring search for the worklist when the gate needs more than 3 rings (unusual configs)
__global__ __launch_bounds__(256) void widen_general_kernel(GridView G, const float4* __restrict__ scan_sorted,
                                                            PoseMats P, int max_ring, NbrRec* __restrict__ nbr,
                                                            const int* __restrict__ wl, const int* __restrict__ wl_count, TieList tl) {
  constexpr int L = 16;
  const int count = *wl_count;
  const int sub = threadIdx.x % L;
  for (int w = blockIdx.x * (256 / L) + threadIdx.x / L; w < count; w += gridDim.x * (256 / L)) {
    const int p = reinterpret_cast<const int4*>(wl)[2 * (size_t)w].x;
    const float4 sp = scan_sorted[p];
    float gx, gy, gz;
    xform4(P.RT, sp.x, sp.y, sp.z, gx, gy, gz);
    KnnResult R;
    knn_search<L>(G, gx, gy, gz, sub, max_ring, R);
    if (sub == 0) {
      const bool ok = R.exact && (R.bi[4] != INT_MAX);
      int4 a, b;
      a.x = R.bi[0]; a.y = R.bi[1]; a.z = R.bi[2]; a.w = R.bi[3];
      // the ring search does not track the sixth candidate: every query it settles is listed for tie_kernel, which re-derives the
      // five in the reference's visiting order (the same five when no distances are tied)
      b.x = R.bi[4]; b.y = ok ? 1 : 0; b.z = ok ? __float_as_int(R.bd[4]) : 0; b.w = ok ? 3 : 0;
      int4* o = reinterpret_cast<int4*>(&nbr[p]);
      o[0] = a;
      o[1] = b;
      if (ok && tl.list) { const unsigned slot = atomicAdd(tl.count, 1u); if (slot < tl.cap) tl.list[slot] = p; }
    }
  }
}

// ------------------------------------------------------------------------------------------
// fit kernel: one lane per scan point (sorted order).  Plane gates + 5x3 QR + plane_eval + Match +
// calculate_H row, then H^T H / H^T h / count of the block with the f64 matrix core.
//   RECS: also write the 64-byte record at the point's ORIGINAL index (debug / MAX_NUM_MATCHES path)
//   DBG : also write the debug side record
// ------------------------------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));


template <bool RECS, bool DBG, int FIT_THREADS>
__global__ __launch_bounds__(FIT_THREADS) void fit_kernel(GridView G, const float4* __restrict__ scan_sorted, int n,
                                                          const NbrRec* __restrict__ nbr, PoseMats P, MatchParams mp,
                                                          double* __restrict__ partials, Rec16* __restrict__ recs,
                                                          RecDbg* __restrict__ dbg, double* __restrict__ out256,
                                                          unsigned int* __restrict__ ticket, int* __restrict__ wl_count,
                                                          unsigned long long seq) {
  constexpr int FIT_WAVES = FIT_THREADS / 64;
  __shared__ float s_rec[FIT_WAVES][16 * 65];       // per wave: [col][row] with stride 65
  __shared__ double s_acc[FIT_WAVES][256];
  __shared__ unsigned int s_last;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int p = chunk * FIT_THREADS + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  float v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = 0.f;
  float n4[4] = {0.f, 0.f, 0.f, 0.f};
  float gx = 0.f, gy = 0.f, gz = 0.f;
  float sq[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  int ids[5] = {-1, -1, -1, -1, -1};
  int flag = 0;
  uint32_t orig = 0;
  bool valid = false;
  TRACE(1, 0);
  if (p < n) {
    const float4 sp = scan_sorted[p];
    orig = __float_as_uint(sp.w);
    xform4(P.RT, sp.x, sp.y, sp.z, gx, gy, gz);
    const int4* nb = reinterpret_cast<const int4*>(&nbr[p]);
    const int4 a = nb[0], b = nb[1];
    ids[0] = a.x; ids[1] = a.y; ids[2] = a.z; ids[3] = a.w; ids[4] = b.x;
    flag = b.y;
    valid = (flag == 1) && (orig < (uint32_t)mp.n_queries);
    TRACE(1, 1);
    if (valid) {
      float px[5], py[5], pz[5];
#pragma unroll
      for (int s = 0; s < 5; s++) {
        const float4 q = G.pts[ids[s]];
        px[s] = q.x; py[s] = q.y; pz[s] = q.z;
        sq[s] = sqdist3(gx, gy, gz, q.x, q.y, q.z);
      }
      TRACE(1, 2);
      // Plane gates (Plane.cpp:23-31): 5 neighbours, 5th SQUARED distance < MAX_DIST_PLANE
      valid = (double)sq[4] < mp.max_dist_plane_d;            // float squared distance vs the double threshold (Plane.cpp:47)
      if (valid) {
        plane_fit5(px, py, pz, n4);
        valid = plane_eval5(n4, px, py, pz, mp.plane_threshold);
      }
      if (valid) {
        const float dist = n4[0] * gx + n4[1] * gy + n4[2] * gz + n4[3];   // Match::Match (Plane.cpp:50-52)
        float row[12];
        h_row(P, gx, gy, gz, n4, mp.estimate_extrinsics, row);            // calculate_H (Localizer.cpp:546-572)
#pragma unroll
        for (int i = 0; i < 12; i++) v[i] = row[i];
        v[12] = -dist;
        v[13] = 1.f;
      }
    }
    if (RECS && orig < (uint32_t)mp.n_queries) {
      float4* out = reinterpret_cast<float4*>(&recs[orig]);
      out[0] = make_float4(v[0], v[1], v[2], v[3]);
      out[1] = make_float4(v[4], v[5], v[6], v[7]);
      out[2] = make_float4(v[8], v[9], v[10], v[11]);
      out[3] = make_float4(v[12], v[13], 0.f, 0.f);
      if (DBG) {
        RecDbg d;
#pragma unroll
        for (int i = 0; i < 4; i++) d.n[i] = valid ? n4[i] : 0.f;
        d.p_global[0] = gx; d.p_global[1] = gy; d.p_global[2] = gz;
        int cnt = 0;
#pragma unroll
        for (int s = 0; s < 5; s++) {
          const bool has = (flag == 1);
          cnt += has ? 1 : 0;
          d.sqd[s] = has ? sq[s] : 0.f;
          d.nbr[s] = has ? (int32_t)__float_as_uint(G.pts[ids[s]].w) : -1;      // (insertion index: flimo_map_points' order)
        }
        d.n_nbr = cnt;
        d.cand = 0;
        d.pad = 0;
        dbg[orig] = d;
      }
    }
  }
  TRACE(1, 3);
  if (partials == nullptr) return;      // records-only launch (MAX_NUM_MATCHES path): capreduce_kernel reduces the records
  // ---- block reduction: D += X^T X with X = the 64 rows of this wave, 4 rows per MFMA ----
  float* sr = s_rec[wave];
#pragma unroll
  for (int c = 0; c < 16; c++) sr[c * 65 + lane] = v[c];
  __builtin_amdgcn_wave_barrier();
  __syncthreads();
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int col = lane & 15, sub4 = lane >> 4;
#pragma unroll
  for (int s = 0; s < 16; s++) {
    const double a = (double)sr[col * 65 + 4 * s + sub4];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
  }
  double* sa = s_acc[wave];
  sa[lane * 4 + 0] = acc[0]; sa[lane * 4 + 1] = acc[1]; sa[lane * 4 + 2] = acc[2]; sa[lane * 4 + 3] = acc[3];
  __syncthreads();
  if (threadIdx.x < 256) {
    const int t = threadIdx.x;
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < FIT_WAVES; w++) r += s_acc[w][t];          // fixed order
    // written through to the agent-coherent level (no dirty L2 line is left behind for the ticket to flush)
    __hip_atomic_store(&partials[(size_t)blockIdx.x * 256 + t], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- grid reduction in FIT_GROUPS independent groups (blocks b, b + 8, b + 16, ...): the last block of a
  //      group to arrive sums the group's partials in block order and writes them, followed by the pass
  //      number, to its 264-double slot of out256 (mapped host memory on the fast path); the host adds the
  //      FIT_GROUPS slots in slot order.  Agent-scope release / acquire around the ticket (guide G16). ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TRACE(1, 4);
  const int group = blockIdx.x & (FIT_GROUPS - 1);
  const int nb_g = (int)(gridDim.x / FIT_GROUPS);                  // gridDim.x is a multiple of FIT_GROUPS
  if (threadIdx.x == 0) {
    // every partial of this block is already performed at agent scope (write-through stores, vmcnt(0), barrier)
    const unsigned int old = __hip_atomic_fetch_add(ticket + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (old == (unsigned int)nb_g - 1u) ? 1u : 0u;
  }
  __syncthreads();
  TRACE(1, 5);
  if (s_last) {
    // 256 accumulator slots x (FIT_THREADS / 256) slices of the group's block list; up to 32 loads in flight;
    // agent-scope atomic loads read past this XCD's L2
    constexpr int PARTS = FIT_THREADS / 256;
    const int t = threadIdx.x & 255, part = threadIdx.x >> 8;
    const int per = (nb_g + PARTS - 1) / PARTS;
    const int k0 = part * per, k1 = min(nb_g, k0 + per);
    const double* base = partials + (size_t)group * 256 + t;
    const size_t stride = (size_t)FIT_GROUPS * 256;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int k = k0;
    for (; k + 31 < k1; k += 32) {
      double v[32];
#pragma unroll
      for (int u = 0; u < 32; u++) v[u] = __hip_atomic_load(base + (size_t)(k + u) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 32; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
    }
    for (; k + 3 < k1; k += 4) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = __hip_atomic_load(base + (size_t)(k + u) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
    }
    for (; k < k1; k++) s0 += __hip_atomic_load(base + (size_t)k * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_acc[part][t] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    TRACE(1, 6);
    double* out = out256 + (size_t)group * FIT_SLOT;
    if (part == 0) {
      double r = 0.0;
#pragma unroll
      for (int q = 0; q < PARTS; q++) r += s_acc[q][t];
      __hip_atomic_store(&out[t], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // publish: out256 may live in mapped host memory; the host spins on word 256 of every slot.  The value
    // stores are complete (vmcnt(0)) before the barrier, the pass number is stored after it.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(out + 256), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      ticket[group] = 0u;                                          // ready for the next pass (visible at kernel end)
      if (group == 0) *wl_count = 0;
    }
    TRACE(1, 7);
  }
}

// ------------------------------------------------------------------------------------------
// fit2 kernel: the per-pass fast path of the fit stage (no records, no caps).  Same per-point routines as fit_kernel
// (bit-identical rows); what differs is how the launch is laid out around the two things that bound it:
//   * a point's chain (neighbour record -> 5 gathered map points -> gates -> 5x3 QR with its correctly rounded
//     divisions and square roots -> plane test -> H row) is ~1700 DEPENDENT instructions: one full wave per SIMD has
//     nothing to overlap them with.  PPW points per wave (64 by default; with 32 or 16 lanes PPW..63 idle in the point stage) put
//     64 / PPW waves on every SIMD, which interleave their chains;
//   * the grid reduction's hand-offs: only the 91 sums the filter reads (upper triangle of H^T H, H^T h, M) leave a block,
//     the last block of each of the FIT_GROUPS groups adds its group's partials in block order (two halves, fixed
//     order: bit-reproducible) and publishes 16-byte {sum, pass number} granules to mapped host memory -- data and
//     "ready" travel together, so neither an acknowledged write nor a separate flag sits on the host's critical path.
// X = [H | h | valid]: D = X^T X on the f64 matrix core gives H^T H, H^T h and M at once (4 rows per MFMA).
// ------------------------------------------------------------------------------------------

// Plane gates + 5x3 QR + plane test + Match + calculate_H row of one query with its five neighbours (ascending distance):
// v = [H row (12) | h = -dist | 1] when every gate passes, untouched (zeros) otherwise.
FLIMO_DEV void fit_row(const GridView& G, const PoseMats& P, const MatchParams& mp, const int (&ids)[5], float gx, float gy, float gz,
                       float (&v)[16]) {
  float px[5], py[5], pz[5], sq4 = 0.f;
#pragma unroll
  for (int s = 0; s < 5; s++) {
    const float4 q = G.pts[ids[s]];
    px[s] = q.x; py[s] = q.y; pz[s] = q.z;
    if (s == 4) sq4 = sqdist3(gx, gy, gz, q.x, q.y, q.z);
  }
  // Plane gates (Plane.cpp:23-31,45-48): 5 neighbours, 5th SQUARED distance (float) < MAX_DIST_PLANE (double)
  if (!((double)sq4 < mp.max_dist_plane_d)) return;
  float n4[4];
  plane_fit5(px, py, pz, n4);
  if (!plane_eval5(n4, px, py, pz, mp.plane_threshold)) return;
  const float dist = n4[0] * gx + n4[1] * gy + n4[2] * gz + n4[3];   // Match::Match (Plane.cpp:50-52)
  float row[12];
  h_row(P, gx, gy, gz, n4, mp.estimate_extrinsics, row);            // calculate_H (Localizer.cpp:546-572)
#pragma unroll
  for (int i = 0; i < 12; i++) v[i] = row[i];
  v[12] = -dist;
  v[13] = 1.f;
}

// Block-wide part of a pass: the wave's ROWS rows X = [H | h | valid] -> D = X^T X on the f64 matrix core (4 rows per MFMA),
// the 91 sums the filter reads -> per-block partial (written through) -> ticket -> the last block of each of the FIT_GROUPS
// groups adds its group's partials in block order (two halves, fixed order: bit-reproducible) and publishes 16-byte
// {sum, pass number} granules to mapped host memory: data and "ready" travel together, so neither an acknowledged write
// nor a separate flag sits on the host's critical path.  Must be reached by every thread of the block.
template <int ROWS>
__device__ __forceinline__ void fit_reduce_publish(const float (&v)[16], bool owns_row, int row, float* sr, double* sa0, double* sa1,
                                                   double* sa2, double* sa3, unsigned int* s_last, const FitIdx& idx,
                                                   double* __restrict__ partials, double2* __restrict__ out_granules,
                                                   unsigned int* __restrict__ ticket, int* __restrict__ wl_count,
                                                   unsigned long long seq, const TieList& tl, int blk, int nblk) {
  // blk / nblk: this block's number among the launch's nblk fit blocks (-1: the whole launch consists of them)
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  const int fb = blk < 0 ? (int)blockIdx.x : blk, fnb = nblk < 0 ? (int)gridDim.x : nblk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (owns_row) {
#pragma unroll
    for (int c = 0; c < 16; c++) sr[c * 65 + row] = v[c];
  }
  wave_lds_sync();
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int col = lane & 15, sub4 = lane >> 4;
#pragma unroll
  for (int s = 0; s < ROWS / 4; s++) {
    const double a = (double)sr[col * 65 + 4 * s + sub4];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
  }
  double* sa = wave == 0 ? sa0 : (wave == 1 ? sa1 : (wave == 2 ? sa2 : sa3));
  sa[lane * 4 + 0] = acc[0]; sa[lane * 4 + 1] = acc[1]; sa[lane * 4 + 2] = acc[2]; sa[lane * 4 + 3] = acc[3];
  __syncthreads();
  if (threadIdx.x < FIT_LIVE) {
    const int t = idx.raw[threadIdx.x];
    const double r = ((sa0[t] + sa1[t]) + sa2[t]) + sa3[t];          // fixed order
    // written through to the agent-coherent level (no dirty L2 line is left behind for the ticket to flush)
    __hip_atomic_store(&partials[(size_t)fb * FIT_LIVE_PAD + threadIdx.x], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  TRACE(1, 4);
  const int group = fb & (FIT_GROUPS - 1);
  const int nb_g = fnb / FIT_GROUPS;                               // the number of fit blocks is a multiple of FIT_GROUPS
  if (threadIdx.x == 0) {
    // every partial of this block is already performed at agent scope (write-through stores, vmcnt(0), barrier)
    const unsigned int old = __hip_atomic_fetch_add(ticket + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_last = (old == (unsigned int)nb_g - 1u) ? 1u : 0u;
  }
  __syncthreads();
  TRACE(1, 5);
  if (*s_last) {
    // Launch-wide ticket (slot FIT_GROUPS): the group whose last block arrives here LAST knows that every block of EVERY group has
    // taken its group ticket, i.e. finished its k-NN / tail phase (in the one-launch pass the groups run concurrently: the last
    // block of group 0 alone cannot know that).  Only that block reads the two counters of the pass and re-arms them.  Taken now,
    // so that the round trip overlaps the loads of the group's partials.
    bool launch_last = false;
    if (threadIdx.x == 0)
      launch_last = __hip_atomic_fetch_add(ticket + FIT_GROUPS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned int)FIT_GROUPS - 1u;
    // Five slices of the group's block list x 48 column PAIRS: 16-byte loads, a slice's thirteen (at 64 blocks per group) in flight
    // at once -- one round trip, where round 5's 32 eight-byte loads per thread went out in two to three batches behind their
    // address arithmetic (1.5 us of the launch's tail).  Agent-scope loads (sc1) read past this XCD's L2.  Fixed slices, fixed
    // order inside a slice and of the slices: bit-reproducible.
    const int c2 = (int)threadIdx.x % (FIT_LIVE_PAD / 2), sl = (int)threadIdx.x / (FIT_LIVE_PAD / 2);      // (threads 240 .. 255 have no slice)
    const int per = (nb_g + 4) / 5;
    const int k0 = sl * per, k1 = sl < 5 ? min(nb_g, k0 + per) : 0;
    v2d_t a0 = {0.0, 0.0}, a1 = {0.0, 0.0};
    {
      const double* base = partials + (size_t)group * FIT_LIVE_PAD + 2 * c2;
      const size_t stride = (size_t)FIT_GROUPS * FIT_LIVE_PAD;
      for (int k = k0; k < k1; k += 16) {
        v2d_t w[16];
#pragma unroll
        for (int u = 0; u < 16; u++) {
          const double* q = base + (size_t)min(k + u, k1 - 1) * stride;      // (dead slots read the slice's last partial again)
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[u]) : "v"(q));
        }
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]), "+v"(w[9]),
                       "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
          if (k + u < k1) a0 += w[u];
          if (k + u + 1 < k1) a1 += w[u + 1];
        }
      }
    }
    __syncthreads();                                               // sa0 .. sa2 are free (every thread read its block sums above)
    if (sl < 5) {
      double* dstb = sl < 2 ? sa0 + sl * FIT_LIVE_PAD : (sl < 4 ? sa1 + (sl - 2) * FIT_LIVE_PAD : sa2);
      dstb[2 * c2] = a0.x + a1.x;
      dstb[2 * c2 + 1] = a0.y + a1.y;
    }
    __syncthreads();
    TRACE(1, 6);
    if (threadIdx.x < FIT_LIVE) {
      // one 16-byte store per sum: {value, pass number}; the host accepts a slot when all of its tags carry this pass
      // (s_nop 1 after every such store: the two wait states of the VMEM-store-data hazard on gfx940+, which the compiler cannot
      //  insert for inline assembly -- the next VALU write of g's registers could otherwise reach the store)
      v2d_t g;
      g.x = ((sa0[threadIdx.x] + sa0[FIT_LIVE_PAD + threadIdx.x]) + (sa1[threadIdx.x] + sa1[FIT_LIVE_PAD + threadIdx.x])) + sa2[threadIdx.x];
      g.y = __longlong_as_double((long long)seq);
      double2* o = out_granules + (size_t)group * FIT_LIVE_PAD + threadIdx.x;
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(o), "v"(g) : "memory");
    }
    if (threadIdx.x == 0) {
      __hip_atomic_store(ticket + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next pass
      if (launch_last) {
        // the launch's two counters, published in slot 0 whichever group this is (the host waits for these two granules first):
        // granule FIT_LIVE = queries of this pass that needed more than their 3x3x3 block, granule FIT_LIVE + 1 = queries whose five
        // hinge on an exact distance tie (the host then runs tie_kernel and the fit again).  Then everything is re-armed.
        v2d_t g;
        g.x = (double)__hip_atomic_load(wl_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g.y = __longlong_as_double((long long)seq);
        double2* o = out_granules + FIT_LIVE;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(o), "v"(g) : "memory");
        g.x = tl.count ? (double)__hip_atomic_load(tl.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        o = out_granules + FIT_LIVE + 1;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(o), "v"(g) : "memory");
        __hip_atomic_store(ticket + FIT_GROUPS, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(wl_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tl.count_next) __hip_atomic_store(tl.count_next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    TRACE(1, 7);
  }
}

// ------------------------------------------------------------------------------------------
// fit2 kernel: the fit stage as its own dispatch when the pass is not fused (first pass of a scan with a poor prior, other
// lanes-per-query settings): neighbour records -> rows (fit_row) -> fit_reduce_publish.  PPW points per wave.
// ------------------------------------------------------------------------------------------
template <int PPW>
__device__ __forceinline__ void fit2_pass(const GridView& G, const float4* __restrict__ scan_sorted, int n,
                                          const NbrRec* __restrict__ nbr, const PoseMats& P, const MatchParams& mp, const FitIdx& idx,
                                          double* __restrict__ partials, double2* __restrict__ out_granules,
                                          unsigned int* __restrict__ ticket, int* __restrict__ wl_count,
                                          unsigned long long seq, const TieList& tl, const ChainCtl& ch, const BookView& book) {
  __shared__ __align__(16) float s_rec[4][16 * 65];       // per wave: [col][row] with stride 65
  __shared__ double s_acc[4][256];
  __shared__ unsigned int s_last;
  static_assert(sizeof(float) * 4 * 16 * 65 >= (size_t)IESKF_LDS_BYTES, "the filter's algebra runs in this workgroup's shared memory");
  int nb = (int)gridDim.x;
  if (ch.S) {
    nb -= 1;
    if ((int)blockIdx.x == nb) {
      double* big = reinterpret_cast<double*>(&s_rec[0][0]);
      ik_extra_block(ch, big, (int)threadIdx.x);
      return;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = xcd_chunk(blockIdx.x, nb);
  const int p = (chunk * 4 + wave) * PPW + lane;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = 0.f;
  TRACE(1, 0);
  {
    const bool mine = lane < PPW && p < n;
    const float4 sp = scan_sorted[mine ? p : 0];
    const int4* nb = reinterpret_cast<const int4*>(&nbr[mine ? p : 0]);
    const int4 a = nb[0], b = nb[1];
    float gx, gy, gz;
    xform4(P.RT, sp.x, sp.y, sp.z, gx, gy, gz);
    int ids[5] = {a.x, a.y, a.z, a.w, b.x};
    TRACE(1, 1);
    if (book.node_c)
      tie_repair_wave(G, book, const_cast<NbrRec*>(nbr), mine && b.y == 1 && (b.w & 2) != 0, p, gx, gy, gz, (uint32_t)b.z, ids,
                      *reinterpret_cast<TieLds*>(&s_rec[wave][0]));
    if (mine && b.y == 1 && __float_as_uint(sp.w) < (uint32_t)mp.n_queries) fit_row(G, P, mp, ids, gx, gy, gz, v);
  }
  TRACE(1, 3);
  fit_reduce_publish<PPW>(v, lane < PPW, lane, s_rec[wave], s_acc[0], s_acc[1], s_acc[2], s_acc[3], &s_last, idx, partials,
                          out_granules, ticket, wl_count, seq, tl, (int)blockIdx.x, nb);
}

template <int PPW>
__global__ __launch_bounds__(256) void fit2_kernel(GridView G, const float4* __restrict__ scan_sorted, int n,
                                                   const NbrRec* __restrict__ nbr, PoseMats P, MatchParams mp, FitIdx idx,
                                                   double* __restrict__ partials, double2* __restrict__ out_granules,
                                                   unsigned int* __restrict__ ticket, int* __restrict__ wl_count,
                                                   unsigned long long seq, TieList tl, ChainCtl ch, BookView book) {
  fit2_pass<PPW>(G, scan_sorted, n, nbr, P, mp, idx, partials, out_granules, ticket, wl_count, seq, tl, ch, book);
}
template <int PPW>
__global__ __launch_bounds__(256) void fit2_chain_kernel(GridView G, const float4* __restrict__ scan_sorted, int n,
                                                   const NbrRec* __restrict__ nbr, const ChainHead* __restrict__ H, MatchParams mp, FitIdx idx,
                                                   double* __restrict__ partials, double2* __restrict__ out_granules,
                                                   unsigned int* __restrict__ ticket, int* __restrict__ wl_count,
                                                   unsigned long long seq, TieList tl, ChainCtl ch, BookView book) {
  if (H->status != 0) return;
  fit2_pass<PPW>(G, scan_sorted, n, nbr, H->pose, mp, idx, partials, out_granules, ticket, wl_count, seq, tl, ch, book);
}

// ------------------------------------------------------------------------------------------
// The reference's order among EXACTLY tied distances.  Octree::knn (Objects/Octree.hpp:526-599) keeps, of two candidates at the
// same float32 squared distance, the one its depth-first recursion meets first (Heap::addPoint rejects `dist >= worst`,
// :72-87): at every octant it descends into the child holding the query first (mortonCode, :269-275), then into the others in
// the order of `ordered_indices[morton]` (:144-153,585-596); inside a leaf the points are visited in insertion order
// (createOctant / updateOctant append in batch order, :303-432).  So the reference's five = the first five of the candidates
// sorted by (distance, visiting order).  The k-NN kernels above choose by (distance, position in the cell-sorted map) and flag
// a query whose five -- or their order -- hinge on such a tie (key_has_tie); tie_kernel below re-derives the five of a flagged
// query: every candidate with d <= d5 inside the proven block is collected, ties are ordered by walking the device copy of the
// octree (the insert book: the reference's own cubes, flimo_gbook.hip) from the root to the node where the two candidates'
// paths part.  A few queries per scan on measured data; every query of a lattice.
// ------------------------------------------------------------------------------------------
__constant__ unsigned char c_ordered_indices[8][7] = {
    {1, 2, 4, 3, 5, 6, 7}, {0, 3, 5, 2, 4, 7, 6}, {0, 3, 6, 1, 4, 7, 5}, {1, 2, 7, 0, 5, 6, 4},
    {0, 5, 6, 1, 2, 7, 3}, {1, 4, 7, 0, 3, 6, 2}, {2, 4, 7, 0, 3, 5, 1}, {3, 5, 6, 1, 2, 4, 0}};

__device__ __forceinline__ int book_morton(float x, float y, float z, const float4& c) {
  return (x > c.x ? 1 : 0) | (y > c.y ? 2 : 0) | (z > c.z ? 4 : 0);
}
__device__ __forceinline__ int visit_pos(int m, int c) {      // place of child c in the visiting order of a node whose query child is m
  if (c == m) return 0;
  int pos = 8;
#pragma unroll
  for (int i = 0; i < 7; i++) pos = (c_ordered_indices[m][i] == c) ? 1 + i : pos;
  return pos;
}
// does Octree::knn meet stored point a (raw insertion index ra) before stored point b?
__device__ bool visited_before(const BookView& B, float ax, float ay, float az, uint32_t ra, float bx, float by, float bz, uint32_t rb,
                               float qx, float qy, float qz) {
  int node = B.root;
  for (int depth = 0; depth < 64 && node >= 0; depth++) {
    if (B.node_cnt[node] >= 0) break;                         // same leaf: insertion order
    const float4 c = B.node_c[node];
    const int ca = book_morton(ax, ay, az, c), cb = book_morton(bx, by, bz, c);
    if (ca != cb) {
      const int m = book_morton(qx, qy, qz, c);
      return visit_pos(m, ca) < visit_pos(m, cb);
    }
    node = B.node_child[(size_t)node * 8 + ca];
  }
  return ra < rb;
}

struct TieLds { float4 pt[64]; float d[64]; uint32_t pos[64]; unsigned int cnt; };

// One wave: the first k of {candidates with d <= dk inside the ring-r block around q} in the reference's order -> out_pos / out_d
// (lane 0).  Returns false when the ball holds more than 64 such candidates (nothing is changed then).
__device__ bool tie_select_wave(const GridView& G, const BookView& B, float qx, float qy, float qz, float dk, int k, TieLds& S,
                                uint32_t (&out_pos)[8], float (&out_d)[8]) {
  const int lane = threadIdx.x & 63;
  const int maxdim = grid_maxdim(G);
  const float margin = 1.0e-3f + 4.0e-7f * (float)maxdim;
  const float fx = (qx - G.ox) * G.inv_cell, fy = (qy - G.oy) * G.inv_cell, fz = (qz - G.oz) * G.inv_cell;
  const float flx = floorf(fminf(fmaxf(fx, -1.0e9f), 1.0e9f)), fly = floorf(fminf(fmaxf(fy, -1.0e9f), 1.0e9f)),
              flz = floorf(fminf(fmaxf(fz, -1.0e9f), 1.0e9f));
  const int cx = (int)flx - G.six, cy = (int)fly - G.siy, cz = (int)flz - G.siz;      // (GridView: the origin is fixed, the grid's corner is a cell shift)
  const float rx = fminf(fmaxf(fx - flx, 0.f), 1.f), ry = fminf(fmaxf(fy - fly, 0.f), 1.f), rz = fminf(fmaxf(fz - flz, 0.f), 1.f);
  const float edge = fminf(fminf(fminf(rx, 1.f - rx), fminf(ry, 1.f - ry)), fminf(rz, 1.f - rz));
  // smallest ring whose block provably holds the ball of radius sqrt(dk) (the same bound the searches use, turned around)
  const float need = fl_sqrt(dk) * G.inv_cell * (1.f + 4.0e-6f) - edge + margin;
  int r = max(1, (int)ceilf(fminf(need, 1.0e9f)));
  {
    const float rg = ((float)r + edge - margin) * G.cell;
    if (!(dk <= rg * rg * (1.f - 1.0e-6f))) r++;
  }
  if (r > 64) return false;                                    // (a gate of more than 32 m at the default cell edge)
  if (lane == 0) S.cnt = 0u;
  wave_lds_sync();
  const int side = 2 * r + 1, rows = side * side;
  const float rc = (fl_sqrt(dk) * (1.f + 1.0e-5f) + 1.0e-6f) * G.inv_cell;
  const float bnd2 = rc * rc * (1.f + 1.0e-5f);
  for (int jb = 0; jb < rows; jb += 64) {                       // one row per lane, 64 rows per round (49 rows up to ring 3)
    const int j = jb + lane;
    if (j < rows) {
      const int jz = j / side, jy = j - jz * side;
      const int dy = jy - r, dz = jz - r;
      const int yy = cy + dy, zz = cz + dz;
      const float a = fmaxf(slab_dist(dy, ry) - margin, 0.f), b = fmaxf(slab_dist(dz, rz) - margin, 0.f);
      const float dyz2 = a * a + b * b;
      if (dyz2 <= bnd2 && yy >= 0 && yy < G.ny && zz >= 0 && zz < G.nz) {
        const float xr = fl_sqrt(fmaxf(bnd2 - dyz2, 0.f)) * (1.f + 1.0e-6f) + margin + 1.0e-4f;
        const int dr = (int)fminf(floorf(fminf(xr + rx, 1.0e6f)), (float)r);
        const int dl = (int)fminf(floorf(fminf(xr + (1.f - rx), 1.0e6f)), (float)r);
        const int x0 = max(cx - dl, 0), x1 = min(cx + dr, G.nx - 1);
        if (x0 <= x1) {
          uint32_t lo, hi;
          grid_row_range(G, G.dir, yy, zz, x0 * G.xs, (x1 + 1) * G.xs, lo, hi);
          for (uint32_t i = lo; i < hi; i++) {
            const float4 p = G.pts[i];
            const float d = sqdist3(qx, qy, qz, p.x, p.y, p.z);
            if (d <= dk) {
              const unsigned slot = atomicAdd(&S.cnt, 1u);
              if (slot < 64u) { S.pt[slot] = p; S.d[slot] = d; S.pos[slot] = i; }
            }
          }
        }
      }
    }
  }
  wave_lds_sync();
  const unsigned n = S.cnt;
  if (n > 64u || (int)n < k) return false;
  if (lane == 0) {
    // selection sort of the first k by (distance, visiting order); n is a handful
    for (int s = 0; s < k; s++) {
      int best = s;
      for (unsigned j = (unsigned)s + 1; j < n; j++) {
        const float dj = S.d[j], db = S.d[best];
        bool less = dj < db;
        if (dj == db) {
          const float4 a = S.pt[j], b = S.pt[best];
          less = visited_before(B, a.x, a.y, a.z, __float_as_uint(a.w), b.x, b.y, b.z, __float_as_uint(b.w), qx, qy, qz);
        }
        if (less) best = (int)j;
      }
      const float4 tp = S.pt[s]; const float td = S.d[s]; const uint32_t tq = S.pos[s];
      S.pt[s] = S.pt[best]; S.d[s] = S.d[best]; S.pos[s] = S.pos[best];
      S.pt[best] = tp; S.d[best] = td; S.pos[best] = tq;
      out_pos[s] = S.pos[s];
      out_d[s] = S.d[s];
    }
  }
  return true;
}

// per-pass form: the queries of the tie list (one wave each); rewrites their neighbour records
__global__ __launch_bounds__(256) void tie_kernel(GridView G, BookView B, const float4* __restrict__ scan_sorted, PoseMats P,
                                                  NbrRec* __restrict__ nbr, const int* __restrict__ list,
                                                  const unsigned int* __restrict__ count, unsigned int cap) {
  __shared__ TieLds s_t[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned n = min(*count, cap);
  for (unsigned e = blockIdx.x * 4 + wave; e < n; e += gridDim.x * 4) {
    const int p = list[e];
    const float4 sp = scan_sorted[p];
    float gx, gy, gz;
    xform4(P.RT, sp.x, sp.y, sp.z, gx, gy, gz);
    int4* rec = reinterpret_cast<int4*>(&nbr[p]);
    const int4 b = rec[1];
    uint32_t pos[8];
    float d[8];
    const bool ok = tie_select_wave(G, B, gx, gy, gz, __int_as_float(b.z), 5, s_t[wave], pos, d);
    if (ok && lane == 0) {
      rec[0] = make_int4((int)pos[0], (int)pos[1], (int)pos[2], (int)pos[3]);
      rec[1] = make_int4((int)pos[4], b.y, b.z, 1);            // same 5th distance; the tie is settled
    }
    wave_lds_sync();
  }
}

// standalone form (flimo_knn): every query, after knn_kernel
__global__ __launch_bounds__(256) void knn_tie_kernel(GridView G, BookView B, const float* __restrict__ qxyz, int nq, int k,
                                                      int32_t* __restrict__ idx, float* __restrict__ sqd, const int32_t* __restrict__ cnt) {
  __shared__ TieLds s_t[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = blockIdx.x * 4 + wave; q < nq; q += gridDim.x * 4) {
    if (cnt[q] != k) continue;                                 // fewer than k points in reach, or not proven exact
    const float dk = sqd[(size_t)q * k + k - 1];
    bool tied = false;                                         // cheap filter: a tie inside the k is visible; one at the boundary is not
    uint32_t pos[8];
    float d[8];
    (void)tied;
    const bool ok = tie_select_wave(G, B, qxyz[3 * q], qxyz[3 * q + 1], qxyz[3 * q + 2], dk, k, s_t[wave], pos, d);
    if (ok && lane == 0) {
      // (what leaves the library: the point's insertion index -- flimo_map_points' order -- not its place in the sorted array)
      for (int s = 0; s < k; s++) { idx[(size_t)q * k + s] = (int32_t)__float_as_uint(G.pts[pos[s]].w); sqd[(size_t)q * k + s] = d[s]; }
    }
    wave_lds_sync();
  }
}

void launch_tie(hipStream_t st, const GridView& G, const BookView& B, const float4* scan_sorted, const PoseMats& P, void* nbr,
                const TieList& tl) {
  hipLaunchKernelGGL(tie_kernel, dim3(64), dim3(256), 0, st, G, B, scan_sorted, P, (NbrRec*)nbr, tl.list, tl.count, tl.cap);
}
void launch_knn_tie(hipStream_t st, const GridView& G, const BookView& B, const float* qxyz, int nq, int k, int32_t* idx, float* sqd,
                    const int32_t* cnt) {
  if (nq <= 0) return;
  const int blocks = min(2048, (nq + 3) / 4);
  hipLaunchKernelGGL(knn_tie_kernel, dim3(blocks), dim3(256), 0, st, G, B, qxyz, nq, k, idx, sqd, cnt);
}

// ------------------------------------------------------------------------------------------
// General NUM_MATCH_POINTS (3..8, Mapper.cpp:106-109, Plane.cpp:41-43): the reference's plumbing accepts any k; every shipped
// configuration uses 5, which the kernels above are specialised for.  Other values take this slower, general pass: exact k-NN
// by the ring search (4 lanes per query), M x 3 plane fit (plane_fit_m), records at the points' original indices, then the
// record reduction (cap_kernel / reduce_kernel).
// ------------------------------------------------------------------------------------------
struct NbrRecK { int32_t idx[8]; int32_t flag; int32_t pad[3]; };    // 48 bytes per query (sorted order)

template <int K>
__global__ __launch_bounds__(256) void knnk_kernel(GridView G, const float4* __restrict__ scan_sorted, int n, PoseMats P,
                                                   int max_ring, NbrRecK* __restrict__ out) {
  constexpr int L = 4;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int p = tid / L, sub = tid % L;
  if (p >= n) return;                       // the four lanes of a query leave together
  const float4 sp = scan_sorted[p];
  float gx, gy, gz;
  xform4(P.RT, sp.x, sp.y, sp.z, gx, gy, gz);
  KnnResultK<K> R;
  knn_search<L, K>(G, gx, gy, gz, sub, max_ring, R);
  if (sub == 0) {
    NbrRecK o;
#pragma unroll
    for (int s = 0; s < 8; s++) o.idx[s] = s < K ? R.bi[s] : -1;
    o.flag = (R.exact && R.bi[K - 1] != INT_MAX) ? 1 : 0;
    o.pad[0] = o.pad[1] = o.pad[2] = 0;
    out[p] = o;
  }
}

template <int K>
__global__ __launch_bounds__(256) void fitk_kernel(GridView G, const float4* __restrict__ scan_sorted, int n,
                                                   const NbrRecK* __restrict__ nbr, PoseMats P, MatchParams mp,
                                                   Rec16* __restrict__ recs, RecDbg* __restrict__ dbg) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float4 sp = scan_sorted[p];
  const uint32_t orig = __float_as_uint(sp.w);
  if (orig >= (uint32_t)mp.n_queries) return;
  float gx, gy, gz;
  xform4(P.RT, sp.x, sp.y, sp.z, gx, gy, gz);
  const NbrRecK nb = nbr[p];
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = 0.f;
  float n4[4] = {0.f, 0.f, 0.f, 0.f};
  float px[K], py[K], pz[K], sq[K];
  bool valid = nb.flag == 1;
  if (valid) {
#pragma unroll
    for (int s = 0; s < K; s++) {
      const float4 q = G.pts[nb.idx[s]];
      px[s] = q.x; py[s] = q.y; pz[s] = q.z;
      sq[s] = sqdist3(gx, gy, gz, q.x, q.y, q.z);
    }
    valid = (double)sq[K - 1] < mp.max_dist_plane_d;
    if (valid) {
      plane_fit_m<K>(px, py, pz, n4);
      valid = plane_eval_m<K>(n4, px, py, pz, mp.plane_threshold);
    }
    if (valid) {
      const float dist = n4[0] * gx + n4[1] * gy + n4[2] * gz + n4[3];
      float row[12];
      h_row(P, gx, gy, gz, n4, mp.estimate_extrinsics, row);
#pragma unroll
      for (int i = 0; i < 12; i++) v[i] = row[i];
      v[12] = -dist;
      v[13] = 1.f;
    }
  }
  float4* o = reinterpret_cast<float4*>(&recs[orig]);
  o[0] = make_float4(v[0], v[1], v[2], v[3]);
  o[1] = make_float4(v[4], v[5], v[6], v[7]);
  o[2] = make_float4(v[8], v[9], v[10], v[11]);
  o[3] = make_float4(v[12], v[13], 0.f, 0.f);
  if (dbg) {
    RecDbg d;
#pragma unroll
    for (int i = 0; i < 4; i++) d.n[i] = valid ? n4[i] : 0.f;
    d.p_global[0] = gx; d.p_global[1] = gy; d.p_global[2] = gz;
    const bool has = nb.flag == 1;
#pragma unroll
    for (int s = 0; s < 5; s++) {            // the debug record shows the first five neighbours
      d.sqd[s] = (has && s < K) ? sq[s < K ? s : 0] : 0.f;
      d.nbr[s] = (has && s < K) ? (int32_t)__float_as_uint(G.pts[nb.idx[s]].w) : -1;
    }
    d.n_nbr = has ? K : 0;
    d.cand = 0;
    d.pad = 0;
    dbg[orig] = d;
  }
}

// the general pass settles EVERY query's order the reference's way (it does not track the (K+1)-th candidate, so it cannot tell
// which queries hinge on a tie): one wave per query, first K of the candidates within the K-th distance in visiting order
template <int K>
__global__ __launch_bounds__(256) void tiek_kernel(GridView G, BookView B, const float4* __restrict__ scan_sorted, int n, PoseMats P,
                                                   NbrRecK* __restrict__ nbr) {
  __shared__ TieLds s_t[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p = blockIdx.x * 4 + wave; p < n; p += gridDim.x * 4) {
    if (nbr[p].flag != 1) continue;                            // wave-uniform
    const float4 sp = scan_sorted[p];
    float gx, gy, gz;
    xform4(P.RT, sp.x, sp.y, sp.z, gx, gy, gz);
    const float4 far = G.pts[nbr[p].idx[K - 1]];
    const float dk = sqdist3(gx, gy, gz, far.x, far.y, far.z);
    uint32_t pos[8];
    float d[8];
    const bool ok = tie_select_wave(G, B, gx, gy, gz, dk, K, s_t[wave], pos, d);
    if (ok && lane == 0) {
#pragma unroll
      for (int s2 = 0; s2 < K; s2++) nbr[p].idx[s2] = (int32_t)pos[s2];
    }
    wave_lds_sync();
  }
}

size_t nbrk_rec_size() { return sizeof(NbrRecK); }
template <int K>
static void launch_match_k_K(hipStream_t st, const GridView& G, const float4* scan_sorted, int n, const PoseMats& P,
                             const MatchParams& mp, void* nbrk, Rec16* recs, RecDbg* dbg, const BookView* book) {
  const long long threads = (long long)n * 4;
  hipLaunchKernelGGL((knnk_kernel<K>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, G, scan_sorted, n, P, mp.max_ring, (NbrRecK*)nbrk);
  if (book) hipLaunchKernelGGL((tiek_kernel<K>), dim3((unsigned)std::min(8192, (n + 3) / 4)), dim3(256), 0, st, G, *book, scan_sorted, n, P, (NbrRecK*)nbrk);
  hipLaunchKernelGGL((fitk_kernel<K>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, G, scan_sorted, n, (const NbrRecK*)nbrk, P, mp, recs, dbg);
}
bool launch_match_k(hipStream_t st, int k, const GridView& G, const float4* scan_sorted, int n, const PoseMats& P,
                    const MatchParams& mp, void* nbrk, Rec16* recs, RecDbg* dbg, const BookView* book) {
  if (n <= 0) return true;
  switch (k) {
    case 3: launch_match_k_K<3>(st, G, scan_sorted, n, P, mp, nbrk, recs, dbg, book); return true;
    case 4: launch_match_k_K<4>(st, G, scan_sorted, n, P, mp, nbrk, recs, dbg, book); return true;
    case 5: launch_match_k_K<5>(st, G, scan_sorted, n, P, mp, nbrk, recs, dbg, book); return true;
    case 6: launch_match_k_K<6>(st, G, scan_sorted, n, P, mp, nbrk, recs, dbg, book); return true;
    case 7: launch_match_k_K<7>(st, G, scan_sorted, n, P, mp, nbrk, recs, dbg, book); return true;
    case 8: launch_match_k_K<8>(st, G, scan_sorted, n, P, mp, nbrk, recs, dbg, book); return true;
    default: return false;
  }
}

// ------------------------------------------------------------------------------------------
// standalone exact kNN (Octree::knn boundary)
// ------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256) void knn_kernel(GridView G, const float* __restrict__ qxyz, int nq, int k,
                                                  int max_ring, int32_t* __restrict__ idx, float* __restrict__ sqd,
                                                  int32_t* __restrict__ cnt) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = tid / L;
  const int sub = tid % L;
  if (q >= nq) return;
  KnnResult R;
  knn_search<L>(G, qxyz[3 * q], qxyz[3 * q + 1], qxyz[3 * q + 2], sub, max_ring, R);
  if (sub == 0) {
    int c = 0;
    for (int s = 0; s < k; s++) {
      const bool has = R.bi[s] != INT_MAX;
      c += has ? 1 : 0;
      idx[(size_t)q * k + s] = has ? (int32_t)__float_as_uint(G.pts[R.bi[s]].w) : -1;      // (insertion index: flimo_map_points' order)
      sqd[(size_t)q * k + s] = has ? R.bd[s] : 0.f;
    }
    cnt[q] = R.exact ? c : -c - 1;
  }
}

// ------------------------------------------------------------------------------------------
// Octree::knn answers from ANYWHERE at bounded cost (Octree.hpp:526-555: a descent, then siblings pruned by the heap's worst
// distance).  The ring search above does too near the map, but a query far from every point would walk (2r+1)^2 rows per ring --
// minutes on a sparse grid of kilometres.  knn_kernel is therefore run with a bound of KNN_FAR_RING rings (flimo_knn), and what
// it could not prove is finished here, ONE WAVE PER QUERY, over the index's own coarse level: the directory of tiles that exist
// (at most GRID_DIR_MAX).  Tiles are visited in ascending order of their box's distance to the query -- a lower bound of the
// distance of every point inside -- until the next tile is farther than the k-th best: best-first search, exact.  Inside a tile
// the rows are dealt to the lanes; a row (and the cells of it) the k-th best's ball cannot reach is not looked at.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_far_kernel(GridView G, const float* __restrict__ qxyz, int nq, int k,
                                                      int32_t* __restrict__ idx, float* __restrict__ sqd, int32_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int maxdim = grid_maxdim(G);
  const double none = __longlong_as_double((long long)KEY_NONE);
  const int ndir = G.ntx * G.nty * G.ntz;
  const int cells_per_xtile = max(1, (8 << G.ts) / G.xs);
  for (int q = blockIdx.x * 4 + wave; q < nq; q += gridDim.x * 4) {
    if (cnt[q] >= 0) continue;                                   // proven by the ring search (wave-uniform)
    const float gx = qxyz[3 * q], gy = qxyz[3 * q + 1], gz = qxyz[3 * q + 2];
    const float fx = (gx - G.ox) * G.inv_cell, fy = (gy - G.oy) * G.inv_cell, fz = (gz - G.oz) * G.inv_cell;
    if (!(fx == fx) || !(fy == fy) || !(fz == fz)) {             // NaN query: no neighbours
      if (lane == 0) { cnt[q] = 0; for (int s = 0; s < k; s++) { idx[(size_t)q * k + s] = -1; sqd[(size_t)q * k + s] = 0.f; } }
      continue;
    }
    const float flx = floorf(fminf(fmaxf(fx, -1.0e9f), 1.0e9f)), fly = floorf(fminf(fmaxf(fy, -1.0e9f), 1.0e9f)),
                flz = floorf(fminf(fmaxf(fz, -1.0e9f), 1.0e9f));
    const int cx = (int)flx - G.six, cy = (int)fly - G.siy, cz = (int)flz - G.siz;
    const float rx = fminf(fmaxf(fx - flx, 0.f), 1.f), ry = fminf(fmaxf(fy - fly, 0.f), 1.f), rz = fminf(fmaxf(fz - flz, 0.f), 1.f);
    // the query in grid cell units (a query a million cells away: its own coordinate's rounding joins the margin)
    const float qcx = (float)cx + rx, qcy = (float)cy + ry, qcz = (float)cz + rz;
    const float margin = 1.0e-3f + 4.0e-7f * fmaxf((float)maxdim, fmaxf(fmaxf(fabsf(qcx), fabsf(qcy)), fabsf(qcz)));
    double k5[6] = {none, none, none, none, none, none};          // this lane's own candidates (every point is seen by one lane, once)
    float bnd2 = INFINITY;                                        // ball of the k-th best so far, cell units squared, inflated
    float last_d = -1.f;
    int last_i = -1;
    for (;;) {
      // ---- the nearest tile not visited yet: (distance, directory index) in ascending order ----
      float best_d = INFINITY;
      int best_i = INT_MAX;
      for (int i = lane; i < ndir; i += 64) {
        if (G.dir[i] == 0) continue;
        const int tx = i % G.ntx, tyz = i / G.ntx, ty_ = tyz % G.nty, tz_ = tyz / G.nty;
        const float x0 = (float)(tx * cells_per_xtile), x1 = (float)((tx + 1) * cells_per_xtile);
        const float y0 = (float)((ty_ << G.ty) - GRID_PAD), y1 = (float)(((ty_ + 1) << G.ty) - GRID_PAD);
        const float z0 = (float)((tz_ << G.tz) - GRID_PAD), z1 = (float)(((tz_ + 1) << G.tz) - GRID_PAD);
        const float ax = fmaxf(fmaxf(x0 - qcx, qcx - x1) - margin, 0.f), ay = fmaxf(fmaxf(y0 - qcy, qcy - y1) - margin, 0.f),
                    az = fmaxf(fmaxf(z0 - qcz, qcz - z1) - margin, 0.f);
        const float d = (ax * ax + ay * ay + az * az) * (1.f - 1.0e-6f);
        const bool after = d > last_d || (d == last_d && i > last_i);
        if (after && (d < best_d || (d == best_d && i < best_i))) { best_d = d; best_i = i; }
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float od = __shfl_xor(best_d, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (od < best_d || (od == best_d && oi < best_i)) { best_d = od; best_i = oi; }
      }
      if (best_i == INT_MAX) break;                              // every tile has been visited
      if (best_d >= bnd2) break;                                 // no point of it -- or of any later one -- can be among the k
      last_d = best_d; last_i = best_i;
      // ---- its rows, one per lane and round ----
      const int tx = best_i % G.ntx, tyz = best_i / G.ntx, ty_ = tyz % G.nty, tz_ = tyz / G.nty;
      const int xt0 = tx * cells_per_xtile, xt1 = min((tx + 1) * cells_per_xtile, G.nx) - 1;
      const int nrows = 1 << (G.ty + G.tz);
      for (int jb = 0; jb < nrows; jb += 64) {
        const int j = jb + lane;
        const int yy = (ty_ << G.ty) + (j & ((1 << G.ty) - 1)) - GRID_PAD, zz = (tz_ << G.tz) + (j >> G.ty) - GRID_PAD;
        if (j < nrows && yy >= 0 && yy < G.ny && zz >= 0 && zz < G.nz && xt0 <= xt1) {
          const float a = fmaxf(fmaxf((float)yy - qcy, qcy - (float)(yy + 1)) - margin, 0.f),
                      b = fmaxf(fmaxf((float)zz - qcz, qcz - (float)(zz + 1)) - margin, 0.f);
          const float dyz2 = a * a + b * b;
          if (dyz2 <= bnd2) {
            int x0 = xt0, x1 = xt1;
            if (bnd2 < 1.0e18f) {
              // cells of the row the ball reaches: |x - qcx| <= xr, widened
              const float xr = fl_sqrt(fmaxf(bnd2 - dyz2, 0.f)) * (1.f + 1.0e-6f) + margin + 1.0e-4f;
              x0 = max(x0, (int)floorf(fmaxf(qcx - xr, -1.0e9f)));
              x1 = min(x1, (int)floorf(fminf(qcx + xr, 1.0e9f)));
            }
            if (x0 <= x1) {
              uint32_t lo, hi;
              grid_row_range(G, G.dir, yy, zz, x0 * G.xs, (x1 + 1) * G.xs, lo, hi);
              for (uint32_t i = lo; i < hi; i++) {
                const float4 p = G.pts[i];
                best6_insert(k5, key_make(sqdist3(gx, gy, gz, p.x, p.y, p.z), i));
              }
            }
          }
        }
        // the ball shrinks as soon as ANY lane holds k candidates (its k-th is an upper bound of the true one)
        const float mine_k = __uint_as_float((uint32_t)((u64)__double_as_longlong(k5[k - 1]) >> 32));
        float mk = mine_k;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) mk = fminf(mk, __shfl_xor(mk, o, 64));
        if (mk < INFINITY) {
          const float rc = (fl_sqrt(mk) * (1.f + 1.0e-5f) + 1.0e-6f) * G.inv_cell;
          bnd2 = fminf(bnd2, rc * rc * (1.f + 1.0e-5f));
        }
      }
      // ---- the k-th best over the whole wave (the lanes keep their lists) ----
      u64 mine[6];
#pragma unroll
      for (int i = 0; i < 6; i++) mine[i] = (u64)__double_as_longlong(k5[i]);
      u64 kth = KEY_NONE;
      for (int r = 0; r < k; r++) {
        const u64 m = (u64)__double_as_longlong(key_group_min(__longlong_as_double((long long)mine[0]), 64, lane));
        kth = m;
        if (mine[0] == m) { mine[0] = mine[1]; mine[1] = mine[2]; mine[2] = mine[3]; mine[3] = mine[4]; mine[4] = mine[5]; mine[5] = KEY_NONE; }
      }
      const float dk = __uint_as_float((uint32_t)(kth >> 32));
      if (dk < INFINITY) {
        const float rc = (fl_sqrt(dk) * (1.f + 1.0e-5f) + 1.0e-6f) * G.inv_cell;
        bnd2 = fminf(bnd2, rc * rc * (1.f + 1.0e-5f));
      }
    }
    // ---- the k nearest, ascending by (distance, position) ----
    u64 mine[6];
#pragma unroll
    for (int i = 0; i < 6; i++) mine[i] = (u64)__double_as_longlong(k5[i]);
    int found = 0;
    for (int r = 0; r < k; r++) {
      const u64 m = (u64)__double_as_longlong(key_group_min(__longlong_as_double((long long)mine[0]), 64, lane));
      if (mine[0] == m) { mine[0] = mine[1]; mine[1] = mine[2]; mine[2] = mine[3]; mine[3] = mine[4]; mine[4] = mine[5]; mine[5] = KEY_NONE; }
      const bool has = m != KEY_NONE;
      found += has ? 1 : 0;
      if (lane == 0) {
        idx[(size_t)q * k + r] = has ? (int32_t)__float_as_uint(G.pts[(uint32_t)m].w) : -1;      // (insertion index, as knn_kernel)
        sqd[(size_t)q * k + r] = has ? __uint_as_float((uint32_t)(m >> 32)) : 0.f;
      }
    }
    if (lane == 0) cnt[q] = found;
  }
}

// ------------------------------------------------------------------------------------------
// MAX_NUM_MATCHES: keep only the first `cap` valid records in scan order (single block)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void cap_kernel(Rec16* __restrict__ recs, int n, int cap) {
  __shared__ int s_cnt[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int b = t * per, e = min(n, b + per);
  int c = 0;
  for (int i = b; i < e; i++) c += (recs[i].v[13] != 0.f) ? 1 : 0;
  s_cnt[t] = c;
  __syncthreads();
  // inclusive scan (Hillis-Steele)
  for (int off = 1; off < 1024; off <<= 1) {
    int v = (t >= off) ? s_cnt[t - off] : 0;
    __syncthreads();
    s_cnt[t] += v;
    __syncthreads();
  }
  int rank = s_cnt[t] - c;   // exclusive prefix
  for (int i = b; i < e; i++) {
    if (recs[i].v[13] != 0.f) {
      if (rank >= cap) {
        float4* o = reinterpret_cast<float4*>(&recs[i]);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        o[0] = z; o[1] = z; o[2] = z; o[3] = z;
      }
      rank++;
    }
  }
}

// ------------------------------------------------------------------------------------------
// HTH / HTh / M reduction with the f64 matrix core: D += X^T X where the 16 columns of X are
// [H(12) | h | valid | 0 | 0]; so D[0:12,0:12] = H^T H, D[0:12,12] = H^T h, D[13,13] = M.
// One wave accumulates a strided set of 4-record groups; raw accumulators (4 doubles per lane)
// go to partials[wave][lane*4 + r]; reduce_final sums them in a fixed order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void reduce_kernel(const Rec16* __restrict__ recs, int n, double* __restrict__ partials) {
  const int lane = threadIdx.x;
  const int w = blockIdx.x;
  const int nw = gridDim.x;
  const int col = lane & 15, sub = lane >> 4;
  const int groups = (n + 3) >> 2;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const float* base = reinterpret_cast<const float*>(recs);
  constexpr int U = 8;   // loads in flight
  for (int g0 = w; g0 < groups; g0 += nw * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int g = g0 + u * nw;
      const int q = 4 * g + sub;
      v[u] = (g < groups && q < n) ? base[(size_t)q * 16 + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const double a = (double)v[u];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
  }
  double* o = partials + (size_t)w * 256 + lane * 4;
  o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
}

__global__ __launch_bounds__(1024) void reduce_final_kernel(const double* __restrict__ partials, int nw,
                                                            double* __restrict__ out) {
  // 256 accumulator slots x 4 slices of the partial list; fixed summation order
  __shared__ double s[4][256];
  const int t = threadIdx.x & 255, part = threadIdx.x >> 8;
  const int per = (nw + 3) >> 2;
  const int b = part * per, e = min(nw, b + per);
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int w = b;
  for (; w + 15 < e; w += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) v[u] = partials[(size_t)(w + u) * 256 + t];
#pragma unroll
    for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
  }
  for (; w < e; w++) s0 += partials[(size_t)w * 256 + t];
  s[part][t] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part == 0) out[t] = (s[0][t] + s[1][t]) + (s[2][t] + s[3][t]);
}

// ------------------------------------------------------------------------------------------
// MAX_NUM_MATCHES path in ONE launch: rank the valid records in scan order (block-wide scan), reduce the first
// `cap` of them with the f64 matrix core and publish sums + pass number to slot 0 of the (host-mapped) output.
// One block: the cap only binds together with MAX_NUM_PC2MATCH-sized inputs (10^4 records, 640 KB).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void capreduce_kernel(const Rec16* __restrict__ recs, int n, int cap,
                                                         double* __restrict__ out256, int* __restrict__ wl_count,
                                                         unsigned long long seq) {
  __shared__ int s_cnt[1024];
  __shared__ float s_rec[16][16 * 65];
  __shared__ double s_acc[16][256];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int per = (n + 1023) / 1024;
  const int b = t * per, e = min(n, b + per);
  const float* base = reinterpret_cast<const float*>(recs);
  int c = 0;
  for (int i = b; i < e; i++) c += (base[(size_t)i * 16 + 13] != 0.f) ? 1 : 0;
  s_cnt[t] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {              // inclusive scan (Hillis-Steele)
    const int v = (t >= off) ? s_cnt[t - off] : 0;
    __syncthreads();
    s_cnt[t] += v;
    __syncthreads();
  }
  int rank = s_cnt[t] - c;                                // valid records before this thread's chunk
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int col = lane & 15, sub4 = lane >> 4;
  float* sr = s_rec[wave];
  for (int k = 0; k < per; k++) {                         // `per` is block-uniform: every wave runs the same trip count
    const int i = b + k;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0;
    if (i < e) {
      const float4* rp = reinterpret_cast<const float4*>(&recs[i]);
      const float4 q3 = rp[3];
      if (q3.y != 0.f) {                                  // v[13]: valid
        if (rank < cap) { r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = q3; }
        rank++;
      }
    }
    if (__ballot(r3.y != 0.f) == 0ull) continue;           // no row of this wave is in: nothing to add (wave-uniform)
    const float v[16] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w};
#pragma unroll
    for (int cc = 0; cc < 16; cc++) sr[cc * 65 + lane] = v[cc];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const double a = (double)sr[col * 65 + 4 * q + sub4];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  double* sa = s_acc[wave];
  sa[lane * 4 + 0] = acc[0]; sa[lane * 4 + 1] = acc[1]; sa[lane * 4 + 2] = acc[2]; sa[lane * 4 + 3] = acc[3];
  __syncthreads();
  if (t < 256) {
    double r = 0.0;
#pragma unroll
    for (int w = 0; w < 16; w++) r += s_acc[w][t];        // fixed order
    __hip_atomic_store(&out256[t], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t == 0) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(out256 + 256), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *wl_count = 0;                                        // ready for the next pass
  }
}

// calibration of the MFMA operand/result layout: D = A*B with A[i][0]=1, A[i][1]=i, B[0][j]=j,
// B[1][j]=16  =>  D[i][j] = j + 16*i.  Operands are placed under the assumed layout
// (lane -> row/col = lane%16, k = lane/16); the host decodes raw[lane*4+r] -> (i,j).
__global__ void mfma_layout_kernel(double* __restrict__ raw) {
  const int lane = threadIdx.x;
  const int rc = lane & 15, k = lane >> 4;
  const double a = (k == 0) ? 1.0 : (k == 1 ? (double)rc : 0.0);
  const double b = (k == 0) ? (double)rc : (k == 1 ? 16.0 : 0.0);
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  raw[lane * 4 + 0] = acc[0]; raw[lane * 4 + 1] = acc[1]; raw[lane * 4 + 2] = acc[2]; raw[lane * 4 + 3] = acc[3];
}

// What a host-driven pass pays on THIS host beyond its kernels: a one-thread launch that stores a 16-byte {value, tag} granule to
// mapped host memory the way a pass publishes its sums (flimo_ctx_create times launch -> granule seen, flimo_capi.hip)
// (flimo_capi.hip: host_store_probe) the word at `src`, read past the caches, as the value of a {value, tag} granule in host memory
__global__ void word_probe_kernel(const unsigned int* __restrict__ src, double2* __restrict__ out, unsigned long long tag) {
  const unsigned int w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  v2d_t g;
  g.x = (double)w;
  g.y = __longlong_as_double((long long)tag);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(out), "v"(g) : "memory");
}
__global__ void rtt_probe_kernel(double2* __restrict__ out, unsigned long long tag) {
  typedef double v2d_t __attribute__((ext_vector_type(2)));
  v2d_t g;
  g.x = 1.0;
  g.y = __longlong_as_double((long long)tag);
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(out), "v"(g) : "memory");
}
void launch_rtt_probe(hipStream_t st, void* out_granule, unsigned long long tag) {
  hipLaunchKernelGGL(rtt_probe_kernel, dim3(1), dim3(1), 0, st, (double2*)out_granule, tag);
}

// pcl::transformPointCloud (PCL 1.10 SSE2 Transformer::se3): c0*x + (c1*y + (c2*z + c3))
__global__ __launch_bounds__(256) void transform_kernel(const float4* __restrict__ in, int n, PoseMats P,
                                                        float4* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float4 p = in[k];
  const float* M = P.RT;
  const float x = M[0] * p.x + (M[1] * p.y + (M[2] * p.z + M[3]));
  const float y = M[4] * p.x + (M[5] * p.y + (M[6] * p.z + M[7]));
  const float z = M[8] * p.x + (M[9] * p.y + (M[10] * p.z + M[11]));
  out[k] = make_float4(x, y, z, p.w);
}

// ------------------------------------------------------------------------------------------
// host-callable launchers
// ------------------------------------------------------------------------------------------
static inline int round_up8(int x) { return (x + 7) & ~7; }

template <int L>
static void launch_knn5_L(hipStream_t st, const GridView& G, const float4* scan_sorted, int n, const PoseMats& P,
                          int max_ring, void* nbr, int* wl, int* wl_count, unsigned long long* cand, const PrevPass& prev,
                          int tail, hipEvent_t e0, hipEvent_t e1, const FuseArgs* fuse, const TieList* tlp, int after_fine = 0,
                          unsigned long long seq = 0ull, const DeskewArgs* dk = nullptr, const ChainHead* chain = nullptr,
                          unsigned int wait_epoch = 0u, unsigned int end_code = 0u) {
  const int qpb = 256 / L;
  const int blocks = round_up8((n + qpb - 1) / qpb);
  constexpr int slots = 8;          // candidate loads in flight per lane
  if constexpr (L == 2) {
    if (fuse) {      // the whole pass in one launch (blocks is a multiple of 8 = FIT_GROUPS; a chained pass has one workgroup more)
      const int grid = fused_blocks(n) + (fuse->ch.S ? 1 : 0);
      if (chain)
        hipExtLaunchKernelGGL((knn5_chain_kernel<2, 8, true>), dim3(grid), dim3(256), 0, st, e0, e1, 0, G, scan_sorted, n, chain, max_ring, (NbrRec*)nbr, wl, wl_count, cand, prev.valid, prev.probe_min, 1, *fuse, wait_epoch);
      else
      hipExtLaunchKernelGGL((knn5_kernel<2, 8, true>), dim3(grid), dim3(256), 0, st, e0, e1, 0, G, scan_sorted, n, P, max_ring, (NbrRec*)nbr, wl, wl_count, cand, prev, 1, *fuse);
      return;
    }
  }
  FuseArgs nofuse{};
  if (tlp) nofuse.tl = *tlp;
  nofuse.fine_mode = after_fine ? 1 : 0;
  nofuse.seq = seq;
  if (dk) nofuse.dk = *dk;
  nofuse.ch.end_code = end_code;
  // e0 / e1 (optional) are attached to the dispatch itself: they read the kernel's own begin / end
  // timestamps, without the extra barrier packets of hipEventRecord
  if constexpr (L == 2) {
    if (chain) {
      hipExtLaunchKernelGGL((knn5_chain_kernel<2, 8, false>), dim3(blocks), dim3(256), 0, st, e0, e1, 0, G, scan_sorted, n, chain, max_ring, (NbrRec*)nbr, wl, wl_count, cand, prev.valid, prev.probe_min, tail, nofuse, wait_epoch);
      return;
    }
  }
  hipExtLaunchKernelGGL((knn5_kernel<L, slots, false>), dim3(blocks), dim3(256), 0, st, e0, e1, 0, G, scan_sorted, n, P, max_ring, (NbrRec*)nbr, wl, wl_count, cand, prev, tail, nofuse);
}

void launch_knn5(hipStream_t st, int lanes_per_query, const GridView& G, const float4* scan_sorted, int n,
                 const PoseMats& P, int max_ring, void* nbr, int* wl, int* wl_count, unsigned long long* cand,
                 const PrevPass& prev, int tail, hipEvent_t e0, hipEvent_t e1, const FuseArgs* fuse, const TieList* tlp, int after_fine,
                 unsigned long long seq, const DeskewArgs* dk, const ChainHead* chain, unsigned int wait_epoch, unsigned int end_code) {
  if (n <= 0) return;
  if (max_ring < 2 || max_ring > TAIL_MAX_RING) tail = 0;
  if (chain) { launch_knn5_L<2>(st, G, scan_sorted, n, P, max_ring, nbr, wl, wl_count, cand, prev, tail, e0, e1, fuse, tlp, after_fine, seq, dk, chain, wait_epoch, end_code); return; }   // (two lanes per query only)
  // (two lanes per query: the other lane counts of rounds 1-4 were A/B variants nothing selected)
  (void)lanes_per_query;
  launch_knn5_L<2>(st, G, scan_sorted, n, P, max_ring, nbr, wl, wl_count, cand, prev, tail, e0, e1, fuse, tlp, after_fine, seq, dk);
}

void launch_widen(hipStream_t st, const GridView& G, const float4* scan_sorted, const PoseMats& P, int max_ring, void* nbr,
                  int* wl, int* wl_count, unsigned long long* cand, hipEvent_t e0, hipEvent_t e1, const TieList* tlp, const ChainHead* chain) {
  TieList tl{};
  if (tlp) tl = *tlp;
  if (max_ring <= 1) return;
  // 2048 blocks = 8 waves per SIMD: the wave-per-query search is a latency chain (the worklist has 8192 slots of prefetch slack);
  // entries without a hint go straight to the gate's ring (measured at 6.6 k pending queries: 19.3 -> 16.3 us)
  if (max_ring <= 3)
    hipExtLaunchKernelGGL(widen_kernel, dim3(2048), dim3(256), 0, st, e0, e1, 0, G, scan_sorted, P, max_ring, (NbrRec*)nbr, wl, wl_count, cand, max_ring, tl, chain);
  else
    hipLaunchKernelGGL(widen_general_kernel, dim3(256), dim3(256), 0, st, G, scan_sorted, P, max_ring, (NbrRec*)nbr, wl, wl_count, tl);
}

int fit_blocks(int n) { const int b = (n + 255) / 256; return (b + FIT_GROUPS - 1) / FIT_GROUPS * FIT_GROUPS; }   // a multiple of FIT_GROUPS (and of 8)

void launch_fit(hipStream_t st, const GridView& G, const float4* scan_sorted, int n, const void* nbr, const PoseMats& P,
                const MatchParams& mp, double* partials, Rec16* recs, RecDbg* dbg, double* out256, unsigned int* ticket,
                int* wl_count, unsigned long long seq) {
  if (n <= 0) return;
  const int blocks = fit_blocks(n);
  if (recs && dbg)
    hipLaunchKernelGGL((fit_kernel<true, true, 256>), dim3(blocks), dim3(256), 0, st, G, scan_sorted, n, (const NbrRec*)nbr, P, mp, partials, recs, dbg, out256, ticket, wl_count, seq);
  else if (recs)
    hipLaunchKernelGGL((fit_kernel<true, false, 256>), dim3(blocks), dim3(256), 0, st, G, scan_sorted, n, (const NbrRec*)nbr, P, mp, partials, recs, dbg, out256, ticket, wl_count, seq);
  // (the records are what this path is for: the per-pass fast path is launch_fit2 / the one-launch pass)
}

// fit2: 64 points per wave (measured: 64 -> 10.7 us, 32 -> 13.0, 16 -> 20.0: the QR is VALU-issue bound, not latency bound)
int fit2_blocks(int n) { const int b = (n + 255) / 256; return (b + FIT_GROUPS - 1) / FIT_GROUPS * FIT_GROUPS; }

void launch_fit2(hipStream_t st, const GridView& G, const float4* scan_sorted, int n, const void* nbr, const PoseMats& P,
                 const MatchParams& mp, const unsigned char* live_idx, double* partials, void* out_granules, unsigned int* ticket,
                 int* wl_count, unsigned long long seq, hipEvent_t e0, hipEvent_t e1, const TieList* tlp, const ChainHead* chain,
                 const ChainCtl* ctl, const BookView* bookp) {
  if (n <= 0) return;
  BookView book{};
  if (bookp) book = *bookp;
  TieList tl{};
  if (tlp) tl = *tlp;
  FitIdx idx;
  for (int i = 0; i < FIT_LIVE_PAD; i++) idx.raw[i] = i < FIT_LIVE ? live_idx[i] : 0;
  ChainCtl ch{};
  if (ctl) ch = *ctl;
  const int grid = fit2_blocks(n) + (ch.S ? 1 : 0);
  if (chain) {
    hipExtLaunchKernelGGL((fit2_chain_kernel<64>), dim3(grid), dim3(256), 0, st, e0, e1, 0, G, scan_sorted, n, (const NbrRec*)nbr, chain, mp, idx, partials, (double2*)out_granules, ticket, wl_count, seq, tl, ch, book);
    return;
  }
  hipExtLaunchKernelGGL((fit2_kernel<64>), dim3(grid), dim3(256), 0, st, e0, e1, 0, G, scan_sorted, n, (const NbrRec*)nbr, P, mp, idx, partials, (double2*)out_granules, ticket, wl_count, seq, tl, ch, book);
}

// (a small scan is spread: see knn5_pass)
int fused_spread(int n) { int sh = 0; while (sh < 3 && ((long long)n << (sh + 1)) <= FUSED_SPREAD_SLOTS) sh++; return sh; }
int fused_blocks(int n) { return round_up8((int)((((long long)n << fused_spread(n)) + 127) / 128)); }
// fine pre-pass over the second-level grid (crowded regions): settles the queries whose five are proven inside their fine 3x3x3
// block; their records get flag 4, which the main launch of the same pass (fine_mode 1) takes over
void launch_knn5_fine(hipStream_t st, const GridView& Gf, const float4* scan_sorted, int n, const PoseMats& P, void* nbr,
                      const PrevPass& prev, const int qlo[3], const int qhi[3], const TieList* tlp, unsigned long long seq, const ChainHead* chain,
                      unsigned int wait_epoch, unsigned int end_code) {
  if (n <= 0) return;
  FuseArgs fa{};
  fa.ch.end_code = end_code;
  fa.fine_mode = 2;
  fa.seq = seq;
  for (int a = 0; a < 3; a++) { fa.qlo[a] = qlo[a]; fa.qhi[a] = qhi[a]; }
  if (tlp) fa.tl = *tlp;
  PrevPass pv = prev;
  if (chain) {
    hipLaunchKernelGGL((knn5_chain_kernel<2, 8, false, true>), dim3(round_up8((n + 127) / 128)), dim3(256), 0, st, Gf, scan_sorted, n, chain, 1,
                       (NbrRec*)nbr, (int*)nullptr, (int*)nullptr, (unsigned long long*)nullptr, pv.valid, pv.probe_min, 0, fa, wait_epoch);
    return;
  }
  hipLaunchKernelGGL((knn5_kernel<2, 8, false, true>), dim3(round_up8((n + 127) / 128)), dim3(256), 0, st, Gf, scan_sorted, n, P, 1,
                     (NbrRec*)nbr, (int*)nullptr, (int*)nullptr, (unsigned long long*)nullptr, pv, 0, fa);
}
void launch_match_fused(hipStream_t st, const GridView& G, const float4* scan_sorted, int n, const PoseMats& P, const MatchParams& mp,
                        void* nbr, int* wl, int* wl_count, unsigned long long* cand, const PrevPass& prev,
                        const unsigned char* live_idx, double* partials, void* out_granules, unsigned int* ticket,
                        unsigned long long seq, hipEvent_t e0, hipEvent_t e1, const TieList* tlp, int after_fine, const DeskewArgs* dk,
                        const ChainHead* chain, const ChainCtl* ctl, const BookView* bookp, unsigned int wait_epoch) {
  if (n <= 0) return;
  FuseArgs fa{};
  if (ctl) fa.ch = *ctl;
  if (bookp) fa.book = *bookp;
  if (dk) fa.dk = *dk;
  fa.fine_mode = after_fine ? 1 : 0;
  fa.tl = TieList{};
  if (tlp) fa.tl = *tlp;
  fa.mp = mp;
  for (int i = 0; i < FIT_LIVE_PAD; i++) fa.idx.raw[i] = i < FIT_LIVE ? live_idx[i] : 0;
  fa.partials = partials; fa.granules = (double2*)out_granules; fa.ticket = ticket; fa.seq = seq;
  fa.spread = fused_spread(n);
  launch_knn5_L<2>(st, G, scan_sorted, n, P, mp.max_ring, nbr, wl, wl_count, cand, prev, 1, e0, e1, &fa, nullptr, 0, 0ull, nullptr, chain, wait_epoch, fa.ch.end_code);
}

size_t nbr_rec_size() { return sizeof(NbrRec); }
size_t wl_entry_size() { return 2 * sizeof(int4); }

void launch_knn(hipStream_t st, const GridView& G, const float* qxyz, int nq, int k, int max_ring, int32_t* idx,
                float* sqd, int32_t* cnt) {
  const int L = 4;
  const long long threads = (long long)nq * L;
  const int blocks = (int)((threads + 255) / 256);
  if (blocks == 0) return;
  // no gate (Octree::knn's answer from anywhere): rings near the map, then the best-first search over the tiles for what is left
  const bool anywhere = max_ring > KNN_FAR_RING * 1024;
  hipLaunchKernelGGL((knn_kernel<4>), dim3(blocks), dim3(256), 0, st, G, qxyz, nq, k, anywhere ? KNN_FAR_RING : max_ring, idx, sqd, cnt);
  if (anywhere) hipLaunchKernelGGL(knn_far_kernel, dim3(std::min(4096, (nq + 3) / 4)), dim3(256), 0, st, G, qxyz, nq, k, idx, sqd, cnt);
}

void launch_cap(hipStream_t st, Rec16* recs, int n, int cap) {
  hipLaunchKernelGGL(cap_kernel, dim3(1), dim3(1024), 0, st, recs, n, cap);
}

void launch_capreduce(hipStream_t st, const Rec16* recs, int n, int cap, double* out256, int* wl_count, unsigned long long seq) {
  hipLaunchKernelGGL(capreduce_kernel, dim3(1), dim3(1024), 0, st, recs, n, cap, out256, wl_count, seq);
}

void launch_reduce(hipStream_t st, const Rec16* recs, int n, int nwaves, double* partials, double* out256) {
  hipLaunchKernelGGL(reduce_kernel, dim3(nwaves), dim3(64), 0, st, recs, n, partials);
  hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(1024), 0, st, partials, nwaves, out256);
}

void launch_word_probe(hipStream_t st, const unsigned int* src, void* out_granule, unsigned long long tag) {
  hipLaunchKernelGGL(word_probe_kernel, dim3(1), dim3(1), 0, st, src, (double2*)out_granule, tag);
}
void launch_mfma_layout(hipStream_t st, double* raw256) {
  hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, st, raw256);
}

void launch_deskew(hipStream_t st, const float4* in, const double* t, int n, const void* frames, int nf,
                   const float* mats32, float4* out_sorted, float4* out_orig, double t_offset) {
  const int blocks = (n + 255) / 256;
  if (blocks == 0) return;
  hipLaunchKernelGGL(deskew_kernel, dim3(blocks), dim3(256), 0, st, in, t, n, (const DevFrame*)frames, nf, mats32,
                     out_sorted, out_orig, t_offset);
}

void launch_transform(hipStream_t st, const float4* in, int n, const PoseMats& P, float4* out) {
  const int blocks = (n + 255) / 256;
  if (blocks == 0) return;
  hipLaunchKernelGGL(transform_kernel, dim3(blocks), dim3(256), 0, st, in, n, P, out);
}

size_t dev_frame_size() { return sizeof(DevFrame); }

}  // namespace flimo
