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
// The index (round 5) has two levels.
//  * SEGMENT ENTRIES, 8 bytes per segment of 8 fine x columns of a row (row = one (y, z) line of cells along x):
//        entry.x = POSITION in pts of the row's first point in the segment's columns or beyond       (bit 31: escape, below)
//        entry.y = eight 4-bit counts, one per column of the segment (column k in bits 4k .. 4k+3)
//    "position of the first point in columns >= c" is entry.x + the sum of the nibbles below c & 7: one v_bfe + one v_dot8_u32_u4.
//    (Positions, not counts: a row does not move when another row grows -- "rows with room", below -- so an entry is rewritten
//    only with its own row, and the fast path adds nothing to it.)  A segment with a
//    column of more than 15 points (crowded maps: raw sweeps inserted under the sensor) is an ESCAPE: entry.y is the index of
//    eight cumulative 32-bit counts in `ovf` (one more dependent load, only there).  One byte of index per fine column instead of
//    the eight of rounds 3-5a (a row-major and a y-fastest table of 32-bit positions).
//  * TILES: the entries live in tiles of 2^ts segments x 2^ty rows x 2^tz layers, and only tiles that hold a point exist; `dir`
//    maps a tile's coordinates to its number in the pool (0: the shared all-zero tile: "no points here").  The directory is at most
//    GRID_DIR_MAX 16-bit entries (a layout doubles the tile until half of that is enough: room for the grid to grow): the k-NN pass keeps it in shared memory, so the way to an
//    entry is still ONE round trip to memory.  A row of a tile has 2^ts + 1 entries: the last one continues into the first
//    segment of the next tile along x (same prefix base), so that the 16-byte load of two neighbouring entries never leaves the
//    tile -- a tile whose first segment holds a point makes its left neighbour exist.
//  * `xstart[pz][tx][py]` (y fastest, rows padded by two on both sides of y and z): position in pts of the first point of row
//    (y, z) that lies in x-tile tx or beyond.  A point merged into the map shifts every later row -- this small table -- and
//    changes entries only inside its own row.
// Padded coordinates everywhere below: py = y + GRID_PAD, pz = z + GRID_PAD.
struct GridView {
  const float4* pts;           // [n_pts]  sorted by (z, y, fine x column)
  const uint2* tiles;          // [1 + present tiles][2^tz][2^ty][2^ts + 1]
  const uint16_t* dir;         // [ntz][nty][ntx]
  const uint32_t* ovf;         // [..][8]  escapes: points of the segment in columns below k, k = 0..7
  const uint32_t* xstart;      // [nz + 4][ntx][ny + 4]
  float ox, oy, oz;            // the FIXED origin of the map's cells (set at the first layout): a point p lies in cell
                               // floor((p - o) * inv_cell) - si.  A grid that grows moves its corner by whole cells (si), never
                               // the origin: a stored point's cell -- and with it the sorted order, the tiles, the entries --
                               // is the same under every later geometry (an integer subtraction behind the floor, no rounding)
  int six, siy, siz;           // cell (0, 0, 0) of the grid is cell (six, siy, siz) of the origin's lattice
  float inv_cell;              // 1 / cell edge
  float cell;                  // cell edge [m]
  int nx, ny, nz;
  uint32_t n_pts;
  // x-slabs: the table is kept at a FINER resolution along x (xs columns per cell, nxf = nx * xs columns per row; points are
  // sorted by (z, y, fine column)): cell (x, y, z) starts at column x * xs.  Geometry (rings, exactness proofs) stays in whole
  // cells; the fast path uses the fine columns to cut a row down to the columns its bound's ball can reach.
  int xs, nxf;
  int nxs;                     // stride of the 32-bit sort key of a column: key = (z * ny + y) * nxs + column, nxs = nxf + 1
  int ts, ty, tz;              // log2 of the tile's extent in segments / rows / layers
  int ntx, nty, ntz;           // directory extent
};
constexpr int GRID_PAD = 2;    // empty rows around the grid in y and z
constexpr int GRID_DIR_MAX = 4096;     // entries of the directory (8 KB of the pass's shared memory; 16 KB cost 0.7 us per pass)
constexpr int GRID_DIR_BUILD = 2048;   // ... a layout uses at most: a grid that grows keeps its tile shape until the directory is full
struct TileShape { int ts, ty, tz, ntx, nty, ntz; };
// directory extent of a grid of nxf columns x ny x nz rows under a tile shape
inline void grid_tile_extent(TileShape& t, int nxf, int ny, int nz) {
  const int nseg = (nxf >> 3) + 1;                         // segments 0 .. nxf >> 3 (column nxf -- a row's length -- has an entry)
  t.ntx = (nseg + (1 << t.ts) - 1) >> t.ts;
  t.nty = (ny + 2 * GRID_PAD + (1 << t.ty) - 1) >> t.ty;
  t.ntz = (nz + 2 * GRID_PAD + (1 << t.tz) - 1) >> t.tz;
}
// tile shape a LAYOUT chooses: from 32 segments (256 columns) x 32 rows x 8 layers, doubled (z, y, x in turn) until the directory
// has at most GRID_DIR_BUILD entries
inline TileShape grid_tile_shape(int nxf, int ny, int nz) {
  TileShape t{5, 5, 3, 0, 0, 0};
  for (int turn = 0;; turn++) {
    grid_tile_extent(t, nxf, ny, nz);
    if ((long long)t.ntx * t.nty * t.ntz <= GRID_DIR_BUILD) return t;
    if (turn % 3 == 0) t.tz++; else if (turn % 3 == 1) t.ty++; else t.ts++;
  }
}
constexpr size_t grid_tile_entries(int ts, int ty, int tz) { return (((size_t)1 << ts) + 1) << (ty + tz); }
constexpr size_t grid_xstart_size(int ny, int nz, int ntx) { return ((size_t)ny + 2 * GRID_PAD) * ((size_t)nz + 2 * GRID_PAD) * (size_t)ntx; }

#if defined(__HIPCC__)
// what the rounding margin of the cell map scales with: the largest cell coordinate in the origin's lattice
__device__ __forceinline__ int grid_maxdim(const GridView& G) {
  return max(G.nx + abs(G.six), max(G.ny + abs(G.siy), G.nz + abs(G.siz)));
}
// points in the columns below column k (0..7) of a segment
__device__ __forceinline__ uint32_t seg_count(uint32_t ex, uint32_t ey, uint32_t k, const uint32_t* __restrict__ ovf) {
  if (__builtin_expect((int)ex < 0, 0)) return (ex & 0x7fffffffu) + ovf[(size_t)ey * 8u + k];
  return __builtin_amdgcn_udot8(ey, 0x11111111u & ((1u << (4u * k)) - 1u), ex, false);
}
// the same without the escape test (the caller has looked at bit 31 of the entries it uses)
__device__ __forceinline__ uint32_t seg_count_plain(uint32_t ex, uint32_t ey, uint32_t k) {
  return __builtin_amdgcn_udot8(ey, 0x11111111u & ((1u << (4u * k)) - 1u), ex, false);
}
// index of the directory entry of the tile of padded row (py, pz), segment sg
__device__ __forceinline__ uint32_t grid_dir_index(const GridView& G, uint32_t py, uint32_t pz, uint32_t sg) {
  return __umul24(__umul24(pz >> G.tz, (uint32_t)G.nty) + (py >> G.ty), (uint32_t)G.ntx) + (sg >> G.ts);      // (24-bit multiplies: full rate)
}
// index in `tiles` of the entry of padded row (py, pz), segment sg, given its tile's number (a tile's row: 2^ts + 1 entries)
// (32-bit: the host keeps the pool below 2^32 entries)
__device__ __forceinline__ uint32_t grid_entry_index(const GridView& G, uint32_t tile, uint32_t py, uint32_t pz, uint32_t sg) {
  const uint32_t zl = pz & ((1u << G.tz) - 1u), yl = py & ((1u << G.ty) - 1u), sl = sg & ((1u << G.ts) - 1u);
  const uint32_t row = (tile << (G.ty + G.tz)) + (zl << G.ty) + yl;
  return (row << G.ts) + row + sl;
}
__device__ __forceinline__ uint32_t grid_xstart_index(const GridView& G, uint32_t py, uint32_t pz, uint32_t tx) {
  return (pz * (uint32_t)G.ntx + tx) * (uint32_t)(G.ny + 2 * GRID_PAD) + py;
}
// position in pts of the first point of row (y, z) in column col or beyond (col in 0 .. nxf); dir: G.dir or a copy of it
__device__ __forceinline__ uint32_t grid_pos(const GridView& G, const uint16_t* dir, int y, int z, int col) {
  const uint32_t py = (uint32_t)(y + GRID_PAD), pz = (uint32_t)(z + GRID_PAD), sg = (uint32_t)col >> 3;
  // (entries hold positions; where no tile exists -- no point of any row -- the position is the row's `xstart` of that x-tile.  No
  //  branch: tile 0 is all zero, so that the loads of two calls overlap)
  const uint32_t tile = dir[grid_dir_index(G, py, pz, sg)];
  const uint32_t base = G.xstart[grid_xstart_index(G, py, pz, sg >> G.ts)];
  const uint2 e = G.tiles[grid_entry_index(G, tile, py, pz, sg)];
  return ((tile && e.x) ? 0u : base) + seg_count(e.x, e.y, (uint32_t)col & 7u, G.ovf);      // (an entry that reads 0 was never written: a tile that is new, a row that has nothing in it)
}
// [lo, hi) = positions in pts of the points of row (y, z) in columns [col0, col1).  Both ends in ONE tile (the usual case: a ring
// search's range is a few cells): the entries' positions as they are -- a row that has nothing in the tile reads zeros at both ends,
// an empty range -- and nothing else is loaded.  Two tiles, or an escape: the careful way (xstart where a tile is missing or has
// nothing of the row).
__device__ __forceinline__ void grid_row_range(const GridView& G, const uint16_t* dir, int y, int z, int col0, int col1, uint32_t& lo, uint32_t& hi) {
  const uint32_t py = (uint32_t)(y + GRID_PAD), pz = (uint32_t)(z + GRID_PAD), s0 = (uint32_t)col0 >> 3, s1 = (uint32_t)col1 >> 3;
  const uint32_t d0 = grid_dir_index(G, py, pz, s0), d1 = grid_dir_index(G, py, pz, s1);
  const uint32_t t0 = dir[d0], t1 = dir[d1];
  const uint2 e0 = G.tiles[grid_entry_index(G, t0, py, pz, s0)], e1 = G.tiles[grid_entry_index(G, t1, py, pz, s1)];
  lo = seg_count_plain(e0.x, e0.y, (uint32_t)col0 & 7u);
  hi = seg_count_plain(e1.x, e1.y, (uint32_t)col1 & 7u);
  if (__builtin_expect(d0 != d1 || (int)(e0.x | e1.x) < 0, 0)) {
    const uint32_t b0 = G.xstart[grid_xstart_index(G, py, pz, s0 >> G.ts)], b1 = G.xstart[grid_xstart_index(G, py, pz, s1 >> G.ts)];
    lo = ((t0 && e0.x) ? 0u : b0) + seg_count(e0.x, e0.y, (uint32_t)col0 & 7u, G.ovf);
    hi = ((t1 && e1.x) ? 0u : b1) + seg_count(e1.x, e1.y, (uint32_t)col1 & 7u, G.ovf);
  }
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
