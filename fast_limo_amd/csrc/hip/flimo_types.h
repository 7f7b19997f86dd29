// fast_limo_amd/csrc/hip/flimo_types.h
// Plain structs shared by the HIP translation units of libflimo_hip.so (gfx950 only).
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace flimo {

// GPU-resident map = uniform grid over the map's bounding box.  Points are stored as float4
// (xyz + original insertion index bits in w), sorted by linear cell id with x fastest, so the
// three x-adjacent cells of a row are ONE contiguous range of `pts`.
struct GridView {
  const float4* pts;           // [n_pts]  sorted by (z, y, fine x column)
  // Both tables hold positions RELATIVE TO THEIR ROW's first point (row = one (y, z) line of cells along x); the rows' absolute
  // starts are a table of their own.  A point merged into the map shifts every later row by one -- that is the small table --
  // and changes relative entries only inside its own row: an insert rewrites the rows it touches, not the index (round 4 kept
  // absolute positions: 2.9 ms of table rewriting per 256k-point insert at 20M points for 0.3 ms of everything else).
  const uint32_t* cell_start;  // [ny*nz][nxs]  x fastest; entry (row, xf): points of the row in columns < xf; entry xf = nxf: the row's length
  const uint32_t* row_start;   // [ny*nz + 1]   row r = z*ny + y starts at pts[row_start[r]]; the last entry is n_pts
  const uint32_t* row_table;   // [(nxf+1)][nz+4][ny+4], y fastest, two empty cells of padding on both sides of y and z:
                               // row_table[xf][z+2][y+2] = cell_start entry (row (y,z), xf).
                               // The 3 y-neighbours of a row bound are 12 contiguous bytes (fast path of the k-NN).
  const uint32_t* row_start_t; // [nz+4][ny+4], the same padding: row_start of row (y, z) where the fast path looks its rows up (pads: 0)
  float ox, oy, oz;            // min corner of cell (0,0,0)
  float inv_cell;              // 1 / cell edge
  float cell;                  // cell edge [m]
  int nx, ny, nz;
  uint32_t n_pts;
  // x-slabs: both tables are kept at a FINER resolution along x (xs columns per cell, nxf = nx * xs columns per row; points are
  // sorted by (z, y, fine column)): cell (x, y, z) starts at column x * xs.  Geometry (rings, exactness proofs) stays in whole
  // cells; the fast path uses the fine columns to cut a row down to the columns its bound's ball can reach.
  int xs, nxf;
  int nxs;                     // entries per row of cell_start: nxf + 1
};

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
