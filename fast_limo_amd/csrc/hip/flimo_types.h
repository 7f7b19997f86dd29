// fast_limo_amd/csrc/hip/flimo_types.h
// Plain structs shared by the HIP translation units of libflimo_hip.so (gfx950 only).
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace flimo {

// GPU-resident map = uniform grid over the map's bounding box.  Points are stored as float4
// (xyz + original insertion index bits in w), sorted by linear cell id with x fastest, so the
// three x-adjacent cells of a row are ONE contiguous range of `pts`.
//
// The index (round 5) is ONE table of 8-byte entries, one per SEGMENT of 8 fine x columns of a row (row = one (y, z) line of cells
// along x; rows are padded by two empty rows on both sides of y and z, so that a 3x3 block of rows around any row of the grid is
// addressable):
//     entry.x = number of the row's points in columns below the segment's first         (bit 31: the entry is an escape, below)
//     entry.y = eight 4-bit counts, one per column of the segment (column k in bits 4k .. 4k+3)
// "points of row r in columns < c" is entry.x + the sum of the nibbles below c & 7: one v_bfe + one v_dot8_u32_u4.  A segment with
// a column of more than 15 points (crowded maps: raw sweeps inserted under the sensor) is an ESCAPE: entry.y is the index of eight
// cumulative 32-bit counts in `ovf` (one more dependent load, only there).  One byte of index per fine column instead of the
// eight of rounds 3-5a (a row-major and a y-fastest table of 32-bit positions): 1.6x the map's bytes at 20 M points where
// those were 12x.  Every count is RELATIVE TO THE ROW's first point; the rows' absolute starts are a table of their own (a point
// merged into the map shifts every later row -- the small table -- and changes entries only inside its own row).
struct GridView {
  const float4* pts;           // [n_pts]  sorted by (z, y, fine x column)
  const uint2* segs;           // [(nz+4) * (ny+4)][nseg]  padded row (y, z) -> row (z+2) * (ny+4) + (y+2); pad rows: all zero
  const uint32_t* ovf;         // [..][8]  escapes: points of the segment in columns below k, k = 0..7
  const uint32_t* row_start;   // [(nz+4) * (ny+4)]  the padded row's first point in pts (pad rows: 0)
  float ox, oy, oz;            // min corner of cell (0,0,0)
  float inv_cell;              // 1 / cell edge
  float cell;                  // cell edge [m]
  int nx, ny, nz;
  uint32_t n_pts;
  // x-slabs: the table is kept at a FINER resolution along x (xs columns per cell, nxf = nx * xs columns per row; points are
  // sorted by (z, y, fine column)): cell (x, y, z) starts at column x * xs.  Geometry (rings, exactness proofs) stays in whole
  // cells; the fast path uses the fine columns to cut a row down to the columns its bound's ball can reach.
  int xs, nxf;
  int nxs;                     // stride of the 32-bit sort key of a column: key = (z * ny + y) * nxs + column, nxs = nxf + 1
  int nseg;                    // entries per row: (nxf >> 3) + 2 (column nxf -- the row's length -- has an entry, and one more
                               // so that the 16-byte load of two neighbouring entries stays inside the row)
};
constexpr int GRID_PAD = 2;    // empty rows around the grid in y and z
constexpr size_t grid_nseg(int nxf) { return ((size_t)nxf >> 3) + 2; }
constexpr size_t grid_prows(int ny, int nz) { return ((size_t)ny + 2 * GRID_PAD) * ((size_t)nz + 2 * GRID_PAD); }

#if defined(__HIPCC__)
// points in the columns below column k (0..7) of a segment
__device__ __forceinline__ uint32_t seg_count(uint32_t ex, uint32_t ey, uint32_t k, const uint32_t* __restrict__ ovf) {
  if (__builtin_expect((int)ex < 0, 0)) return (ex & 0x7fffffffu) + ovf[(size_t)ey * 8u + k];
  return __builtin_amdgcn_udot8(ey, 0x11111111u & ((1u << (4u * k)) - 1u), ex, false);
}
// the same without the escape test (the caller has looked at bit 31 of the entries it uses)
__device__ __forceinline__ uint32_t seg_count_plain(uint32_t ex, uint32_t ey, uint32_t k) {
  return __builtin_amdgcn_udot8(ey, 0x11111111u & ((1u << (4u * k)) - 1u), ex, false);
}
__device__ __forceinline__ uint32_t grid_prow(const GridView& G, int y, int z) {
  return (uint32_t)(z + GRID_PAD) * (uint32_t)(G.ny + 2 * GRID_PAD) + (uint32_t)(y + GRID_PAD);
}
// points of padded row `prow` in columns < col (col in 0 .. nxf)
__device__ __forceinline__ uint32_t grid_count(const GridView& G, uint32_t prow, int col) {
  const uint2 e = G.segs[(size_t)prow * (size_t)G.nseg + (size_t)(col >> 3)];
  return seg_count(e.x, e.y, (uint32_t)col & 7u, G.ovf);
}
// [lo, hi) = positions in pts of the points of row (y, z) in columns [col0, col1)
__device__ __forceinline__ void grid_row_range(const GridView& G, int y, int z, int col0, int col1, uint32_t& lo, uint32_t& hi) {
  const uint32_t prow = grid_prow(G, y, z);
  const uint32_t rs = G.row_start[prow];
  const uint2* row = G.segs + (size_t)prow * (size_t)G.nseg;
  const uint2 e0 = row[col0 >> 3], e1 = row[col1 >> 3];
  lo = rs + seg_count(e0.x, e0.y, (uint32_t)col0 & 7u, G.ovf);
  hi = rs + seg_count(e1.x, e1.y, (uint32_t)col1 & 7u, G.ovf);
}
#endif

// Previous pass of the SAME scan (same sorted scan, same neighbour records): its body -> world matrix lets the k-NN
// fast path bound how far every query moved; valid = 0 disables the pruning (first pass of a scan).
struct PrevPass {
  float RT[16];
  int valid;
  unsigned probe_min;   // first pass: a query whose 3x3x3 block holds at least this many candidates walks its own cell first, for a bound (0: never)
};

// Per-pass pose constants, computed on the host exactly as the reference does
// (State(x).get_RT() etc., reference Objects/State.cpp:38-55,136-172 and
// Modules/Localizer.cpp:549-555) and passed by value as a kernel argument.
struct PoseMats {
  float RT[16];       // body -> world            (Mapper.cpp:71-72)
  float RT_inv[16];   // world -> body (IMU)      (Localizer.cpp:549)
  float TLI_inv[16];  // body -> lidar            (Localizer.cpp:550)
  float R_inv[9];     // s.rot^-1 as float        (Localizer.cpp:554)
  float RLI_inv[9];   // s.offset_R_L_I^-1        (Localizer.cpp:555)
};

struct MatchParams {
  double max_dist_plane_d; // MAX_DIST_PLANE as configured: the gate compares the FLOAT 5th squared distance, widened, with
                           // this double (Plane.cpp:45-48)
  float plane_threshold;   // Plane.cpp:110
  int estimate_extrinsics;
  int n_queries;           // min(N, MAX_NUM_PC2MATCH)
  int max_ring;            // search rings needed to honour the MAX_DIST_PLANE gate exactly
};

// Input filters of Localizer::updatePointCloud (Localizer.cpp:262-302) + the per-point stamp of deskewPointCloud (:741-805)
struct FilterParams {
  int crop;                // CropBox, negative: points strictly inside the box are removed (:57-59,269-270)
  float mn[3], mx[3];
  int dist;                // min distance filter (:295-297)
  float min_dist;          // compared with the float norm
  int rate_on, rate;       // every rate-th of the points that survive NaN removal + crop box
  int kind;                // time union view: 0 OUSTER u32 ns, 1 VELODYNE f32 s, 2 HESAI f64 s, 3 LIVOX f64 ns
  int eos;                 // end_of_sweep
  double sweep_ref;        // sweep reference time
  int fov;                 // FoV filter (:873-876): fabs(atan2(y, x)) < fov_angle on the rate filter's survivors
  float fov_angle;
};

// 64-byte per-query record consumed by the HTH reducer: H row, h, valid flag.
struct alignas(16) Rec16 {
  float v[16];   // [0..11] H row, [12] h = -dist, [13] valid (1/0), [14..15] 0
};

// Debug side-record (only written when the context has debug records enabled).
struct alignas(16) RecDbg {
  float n[4];
  float p_global[3];
  int32_t n_nbr;
  float sqd[5];
  int32_t nbr[5];
  int32_t cand;      // candidates examined
  int32_t pad;
};

}  // namespace flimo
