// fast_limo_amd/csrc/hip/flimo_map.hip  -- gfx950 (MI355X) only.
//
// Build of the GPU-resident map index: a uniform grid over the bounding box of the stored points.
// Replaces the pointer octree of the reference as the k-NN acceleration structure
// (Objects/Octree.hpp:282-338 createOctant); which points are STORED is decided by the caller
// (flimo_capi.cpp implements the reference's insert rule).  Steps, all on the context stream:
//   1. bbox      : min/max reduction (wave shuffles + one atomic per wave)
//   2. cell keys : key = (floor((p - o) * inv_cell) - corner) linearised with x fastest, at fine-column resolution along x
//   3. sort      : stable LSD radix sort of (key, index) pairs (rocPRIM) -- keeps insertion order inside a column, so the
//                  device order is deterministic
//   4. rows      : per row its key range and its room, an exclusive sum of the rooms, every point to its row's place (float4,
//                  w keeps the insertion index)
//   5. index     : tiles marked and numbered, one workgroup per row turns the row's columns into its entries
//                  (GridView, flimo_types.h)
// An insert afterwards touches only the rows it adds to (rows_insert_kernel); a map that outgrows its grid has the grid grown
// around it (index_regrid).
#include <hip/hip_runtime.h>
#include <chrono>
#include <immintrin.h>
#include "flimo_prims.h"
#include <float.h>
#include "flimo_types.h"
#include "flimo_math.h"
#include "flimo_kernels.h"
#include "flimo_gbook.h"

namespace flimo {

// order-preserving float <-> uint mapping for atomic min/max
__device__ __forceinline__ unsigned f2o(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
static inline float o2f_host(unsigned o) {
  unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

__global__ __launch_bounds__(256) void bbox_kernel(const float4* __restrict__ pts, size_t n, unsigned* __restrict__ box) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 p = pts[i];
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      atomicMin(&box[a], f2o(mn[a]));
      atomicMax(&box[3 + a], f2o(mx[a]));
    }
  }
}

// column key of a point: (z, y, fine x column).  The fine column is floor(t * xs) of the SAME rounded t = (p.x - ox) * inv_cell whose
// floor is the cell (xs is a power of two: the product is exact), so columns nest in cells exactly.
// (the geometry a cell computation needs: GridView's, without its tables)
struct CellGeo { float ox, oy, oz, inv_cell; int nx, ny, nz, xs, six, siy, siz; };
static inline CellGeo cell_geo(const GridView& G) { return CellGeo{G.ox, G.oy, G.oz, G.inv_cell, G.nx, G.ny, G.nz, G.xs, G.six, G.siy, G.siz}; }
// (64-bit: a grid of kilometres has more than 2^32 fine columns; the sort looks only at the bits the grid needs)
typedef unsigned long long ckey_t;
__device__ __forceinline__ ckey_t column_key(const float4& p, const CellGeo& c) {
  const float tx = (p.x - c.ox) * c.inv_cell;     // identical float expression to the query side (flimo_kernels.hip: knn_search)
  int cx = (int)floorf(fminf(fmaxf(tx * (float)c.xs, -1.0e9f), 1.0e9f)) - c.six * c.xs;      // (the grid's corner: a whole-cell shift behind the floor)
  int cy = (int)floorf((p.y - c.oy) * c.inv_cell) - c.siy;
  int cz = (int)floorf((p.z - c.oz) * c.inv_cell) - c.siz;
  cx = min(max(cx, 0), c.nx * c.xs - 1);
  cy = min(max(cy, 0), c.ny - 1);
  cz = min(max(cz, 0), c.nz - 1);
  return ((ckey_t)cz * (ckey_t)c.ny + (ckey_t)cy) * ((ckey_t)c.nx * c.xs + 1ull) + (ckey_t)cx;      // (a row of columns: nxf of them + its end entry)
}
__global__ __launch_bounds__(256) void cellkey_kernel(const float4* __restrict__ pts, size_t n, CellGeo c,
                                                      ckey_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = column_key(pts[i], c);
  vals[i] = (uint32_t)i;
}

struct MailArgs { const uint32_t* src[6]; int n[6]; int dst[6]; int parts; unsigned* rearm; uint32_t* zero; int zero_n; uint32_t tag; };
__global__ __launch_bounds__(64) void mail_kernel(MailArgs a, uint32_t* __restrict__ mail) {
  for (int k = 0; k < a.parts; k++)
    if ((int)threadIdx.x < a.n[k]) mail[a.dst[k] + threadIdx.x] = a.src[k][threadIdx.x];
  __threadfence_system();
  __syncthreads();                                   // (one wave: orders the copies before the resets below)
  if (a.rearm && threadIdx.x < 6) a.rearm[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
  if (a.zero && (int)threadIdx.x < a.zero_n) a.zero[threadIdx.x] = 0u;
  // the words are on their way to host memory; the tag follows them (mail_wait spins on it: no stream wait for a few words)
  if (threadIdx.x == 0) __hip_atomic_store(&mail[MAIL_TAG], a.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t ensure_mail(MapBuildScratch& S) {
  hipError_t e;
  if (!S.bbox) {
    if ((e = hipMalloc(&S.bbox, 6 * sizeof(unsigned))) != hipSuccess) return e;
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    if ((e = hipMemcpy(S.bbox, init, sizeof(init), hipMemcpyHostToDevice)) != hipSuccess) return e;
  }
  if (!S.mail_host) {
    if ((e = hipHostMalloc((void**)&S.mail_host, MAIL_WORDS * sizeof(uint32_t), hipHostMallocMapped)) != hipSuccess) return e;
    memset(S.mail_host, 0, MAIL_WORDS * sizeof(uint32_t));
    if ((e = hipHostGetDevicePointer((void**)&S.mail_dev, S.mail_host, 0)) != hipSuccess) return e;
  }
  return hipSuccess;
}
hipError_t mail_words(hipStream_t st, MapBuildScratch& S, const MailPart* parts, int nparts, bool rearm_bbox, void* zero, int zero_n) {
  hipError_t e = ensure_mail(S);
  if (e != hipSuccess) return e;
  MailArgs a{};
  a.parts = nparts;
  for (int k = 0; k < nparts && k < 6; k++) { a.src[k] = (const uint32_t*)parts[k].src; a.n[k] = parts[k].n; a.dst[k] = parts[k].dst; }
  a.rearm = rearm_bbox ? (unsigned*)S.bbox : nullptr;
  a.zero = (uint32_t*)zero; a.zero_n = zero_n;
  if (++S.mail_seq == 0u) S.mail_seq = 1u;
  a.tag = S.mail_seq;
  hipLaunchKernelGGL(mail_kernel, dim3(1), dim3(64), 0, st, a, S.mail_dev);
  return hipGetLastError();
}
// Waits for the words of the last mail_words (every launch queued before it on the stream has completed by then): spins on the tag
// in mapped memory; a wait that runs out (2 s) falls back to the stream.
hipError_t mail_wait(hipStream_t st, MapBuildScratch& S) {
  const volatile uint32_t* m = S.mail_host;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned long long spins = 0;; spins++) {
    if (m[MAIL_TAG] == S.mail_seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return hipSuccess; }
    _mm_pause();
    if ((spins & 0xfffull) == 0xfffull && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) break;
  }
  return hipStreamSynchronize(st);
}
// the box the last bbox reduction left in S.bbox -> host (ordered-uint words), re-armed; ends synchronised
static hipError_t fetch_bbox(hipStream_t st, MapBuildScratch& S, unsigned ob[6]) {
  const MailPart part{S.bbox, 6, MAIL_BBOX};
  hipError_t e = mail_words(st, S, &part, 1, true);
  if (e == hipSuccess) e = mail_wait(st, S);
  if (e != hipSuccess) {
    // the re-arming rode on the mail kernel that did not run: put the empty box back by hand, so the next reduction starts from it
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    (void)hipMemcpy(S.bbox, init, sizeof(init), hipMemcpyHostToDevice);
    return e;
  }
  for (int i = 0; i < 6; i++) ob[i] = S.mail_host[MAIL_BBOX + i];
  return hipSuccess;
}

static hipError_t ensure_scratch(MapBuildScratch& S, size_t n) {
  hipError_t e;
  if ((e = ensure_mail(S)) != hipSuccess) return e;
  if (n > S.cap_pts) {
    if (S.keys_in) { hipFree(S.keys_in); hipFree(S.keys_out); hipFree(S.vals_in); hipFree(S.vals_out); }
    const size_t cap = n + n / 4 + 1024;
    if ((e = hipMalloc(&S.keys_in, cap * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&S.keys_out, cap * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&S.vals_in, cap * 4)) != hipSuccess) return e;
    if ((e = hipMalloc(&S.vals_out, cap * 4)) != hipSuccess) return e;
    S.cap_pts = cap;
  }
  return hipSuccess;
}

hipError_t map_bbox(hipStream_t st, const float4* pts, size_t n, MapBuildScratch& S, float bbox_host[6]) {
  hipError_t e = ensure_scratch(S, 0);
  if (e != hipSuccess) return e;
  const int blocks = (int)std::min<size_t>((n + 255) / 256, 2048);
  if (blocks > 0) hipLaunchKernelGGL(bbox_kernel, dim3(blocks), dim3(256), 0, st, pts, n, (unsigned*)S.bbox);
  unsigned out[6];
  if ((e = fetch_bbox(st, S, out)) != hipSuccess) return e;
  for (int i = 0; i < 6; i++) bbox_host[i] = o2f_host(out[i]);
  return hipSuccess;
}

// ---- the index (GridView, flimo_types.h): directory, tiles of segment entries, xstart ----------------------------------------
__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ keys, uint32_t lo, uint32_t hi, uint32_t key) {
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// The same by a whole wave (every lane gets the result): 64 probes per round instead of one -- 3 dependent loads for 64k keys
// instead of 16.
__device__ __forceinline__ uint32_t wave_lower_bound_u32(const uint32_t* __restrict__ keys, uint32_t k, uint32_t key) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t lo = 0u, hi = k;                            // the answer lies in [lo, hi]
  while (hi - lo > 64u) {
    const uint32_t step = (hi - lo + 63u) / 64u;
    const uint32_t idx = lo + lane * step + (step - 1u);            // last element of this lane's segment
    const bool below = idx < hi && keys[idx] < key;                  // monotone over the lanes (sorted keys)
    const uint32_t c = (uint32_t)__popcll(__ballot(below));
    const uint32_t nlo = min(lo + c * step, hi);
    hi = min(nlo + (step - 1u), hi);
    lo = nlo;
  }
  const uint32_t idx = lo + lane;
  const bool below = idx < hi && keys[idx] < key;
  return lo + (uint32_t)__popcll(__ballot(below));
}

__device__ __forceinline__ uint32_t lower_bound_k(const ckey_t* __restrict__ keys, uint32_t lo, uint32_t hi, ckey_t key) {
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (keys[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ uint32_t wave_lower_bound_k(const ckey_t* __restrict__ keys, uint32_t k, ckey_t key) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t lo = 0u, hi = k;                            // the answer lies in [lo, hi]
  while (hi - lo > 64u) {
    const uint32_t step = (hi - lo + 63u) / 64u;
    const uint32_t idx = lo + lane * step + (step - 1u);            // last element of this lane's segment
    const bool below = idx < hi && keys[idx] < key;                  // monotone over the lanes (sorted keys)
    const uint32_t c = (uint32_t)__popcll(__ballot(below));
    const uint32_t nlo = min(lo + c * step, hi);
    hi = min(nlo + (step - 1u), hi);
    lo = nlo;
  }
  const uint32_t idx = lo + lane;
  const bool below = idx < hi && keys[idx] < key;
  return lo + (uint32_t)__popcll(__ballot(below));
}
// geometry of the index as the builders need it
struct TabGeo { int nxs, ny, nz, ts, ty, tz, ntx, nty, ntz; };
struct SegTab { uint2* tiles; const uint16_t* dir; uint32_t* ovf; uint32_t* ovf_count; uint32_t ovf_cap; uint32_t* xstart; TabGeo g; uint32_t* full_mail; };      // full_mail: word of mapped host memory that says "lay the index out afresh" (an insert), or null
__device__ __forceinline__ uint32_t tab_dir_index(const TabGeo& g, uint32_t py, uint32_t pz, uint32_t sg) {
  return ((pz >> g.tz) * (uint32_t)g.nty + (py >> g.ty)) * (uint32_t)g.ntx + (sg >> g.ts);
}
// which tiles a batch of sorted column keys needs: the tile of each key's segment, and -- for a key in the FIRST segment of a
// tile -- the tile to its left (whose last entry continues into that segment)
__global__ __launch_bounds__(256) void tiles_mark_kernel(const ckey_t* __restrict__ keys, uint32_t n, TabGeo g, uint32_t* __restrict__ need) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const ckey_t key = keys[i];
  const uint32_t r = (uint32_t)(key / (ckey_t)g.nxs), sg = (uint32_t)(key - (ckey_t)r * (ckey_t)g.nxs) >> 3;
  const uint32_t py = r % (uint32_t)g.ny + GRID_PAD, pz = r / (uint32_t)g.ny + GRID_PAD;
  const uint32_t d = tab_dir_index(g, py, pz, sg);
  need[d] = 1u;
  if ((sg & ((1u << g.ts) - 1u)) == 0u && sg != 0u) need[d - 1u] = 1u;
}
// needed tiles that do not exist yet take the next numbers, in directory order (one workgroup: a scan over at most GRID_DIR_MAX
// entries); counters[0] = next free number, counters[1] = 1 when the pool ran out (those tiles stay absent: the host lays the
// index out afresh before anything reads it), mail[0..1] = the same two words for the host
__global__ __launch_bounds__(256) void tiles_number_kernel(uint16_t* __restrict__ dir, uint32_t* __restrict__ need, int ndir, uint32_t cap_tiles,
                                                           uint32_t* __restrict__ counters, uint32_t* __restrict__ mail, int fresh) {
  __shared__ uint32_t s_scan[256];
  const int t = (int)threadIdx.x;
  constexpr int PER = GRID_DIR_MAX / 256;
  uint32_t mine[PER], cnt = 0u;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = t * PER + k;
    const bool want = i < ndir && need[i] != 0u && (fresh || dir[i] == 0);
    mine[k] = want ? 1u : 0u;
    cnt += mine[k];
    if (i < ndir) { need[i] = 0u; if (fresh && !want) dir[i] = 0; }
  }
  s_scan[t] = cnt;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const uint32_t v = (t >= o) ? s_scan[t - o] : 0u;
    __syncthreads();
    s_scan[t] += v;
    __syncthreads();
  }
  const uint32_t first = fresh ? 1u : counters[0];
  uint32_t num = first + s_scan[t] - cnt;
  bool over = false;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    if (mine[k]) {
      if (num < cap_tiles && num < 65536u) dir[t * PER + k] = (uint16_t)num; else over = true;
      num++;
    }
  }
  __syncthreads();
  if (t == 255) { counters[0] = min(first + s_scan[255], cap_tiles); if (mail) mail[0] = first + s_scan[255]; }
  if (over) { counters[1] = 1u; if (mail) mail[1] = 1u; }
}

// The entries of ONE row from its points' columns: a histogram of the row's columns in shared memory (windows of SEG_WIN
// columns; a row's points are sorted by column), the segments' prefix sums by a scan over the workgroup, then one entry per
// segment -- eight nibbles, or an escape (eight cumulative counts in `ovf`) when a column holds more than 15 points -- written
// into the tiles that exist, counted from the row's first point in the tile; xstart gets the position of that point.  An escape
// keeps the slot its entry already had (the map only grows: a segment that escaped stays one), a new one takes the next free slot.
// cols(i) = column of the row's i-th point; start = position of the row's first point in pts.
constexpr int SEG_WIN = 4096;                                      // columns per window: 512 segments, two per thread
template <typename ColOf>
__device__ __forceinline__ void row_entries(const SegTab& T, uint32_t py, uint32_t pz, uint32_t start, uint32_t len, ColOf cols,
                                            uint32_t* s_cnt /*[SEG_WIN + 8]*/, uint32_t* s_pre /*[513]*/, uint32_t* s_scan /*[256]*/) {
  const TabGeo& g = T.g;
  const int t = (int)threadIdx.x;
  const uint32_t TS = 1u << g.ts;
  const int nseg = g.ntx << g.ts;                                  // segments of the directory's x extent
  const uint32_t xrow = (pz * (uint32_t)g.ntx) * (uint32_t)(g.ny + 2 * GRID_PAD) + py;      // xstart index of x-tile 0
  uint32_t carry = 0u;                                             // points in the columns below the window
  for (int w0 = 0; w0 < nseg * 8; w0 += SEG_WIN) {
    // (eight columns more than the window: the segment behind it, whose nibbles the closing entry of a tile that ends with the
    //  window carries)
    for (int i = t; i < SEG_WIN + 8; i += 256) s_cnt[i] = 0u;
    __syncthreads();
    for (uint32_t i0 = (uint32_t)t; i0 < len; i0 += 1024u) {       // (four columns per thread and round: their loads overlap)
      int c[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const uint32_t i = i0 + 256u * (uint32_t)u; c[u] = i < len ? cols(i) - w0 : -1; }
#pragma unroll
      for (int u = 0; u < 4; u++) if (c[u] >= 0 && c[u] < SEG_WIN + 8) atomicAdd(&s_cnt[c[u]], 1u);
    }
    __syncthreads();
    uint32_t tot[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      uint32_t a = 0u;
#pragma unroll
      for (int k = 0; k < 8; k++) a += s_cnt[(2 * t + h) * 8 + k];
      tot[h] = a;
    }
    s_scan[t] = tot[0] + tot[1];
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const uint32_t v = (t >= o) ? s_scan[t - o] : 0u;
      __syncthreads();
      s_scan[t] += v;
      __syncthreads();
    }
    const uint32_t excl = s_scan[t] - (tot[0] + tot[1]);
    const uint32_t win_total = s_scan[255];
    s_pre[2 * t] = carry + excl;                                   // points of the row below segment (w0 / 8) + 2 t
    s_pre[2 * t + 1] = carry + excl + tot[0];
    if (t == 255) s_pre[512] = carry + win_total;
    __syncthreads();
    // entry of window segment i (0 .. 512) at `slot` of tile `tile`: the position of the row's first point in its columns or beyond
    auto put = [&](int i, uint32_t tile, uint32_t slot) {
      uint2* dst = T.tiles + ((((size_t)tile << (g.ty + g.tz)) + (size_t)(((pz & ((1u << g.tz) - 1u)) << g.ty) + (py & ((1u << g.ty) - 1u)))) * (TS + 1u) + slot);
      const uint32_t pre = start + s_pre[i];
      uint32_t nib = 0u;
      bool big = false;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t v = s_cnt[i * 8 + k];
        big = big || v > 15u;
        nib |= (v & 15u) << (4 * k);
      }
      if (!big) { *dst = make_uint2(pre, nib); return; }
      const uint2 old = *dst;
      const uint32_t slot_o = ((int)old.x < 0) ? old.y : atomicAdd(T.ovf_count, 1u);
      if (slot_o < T.ovf_cap) {            // (the pool holds one slot per 16 points of the point buffer's capacity -- 2 bytes per point; a segment at a tile's edge takes two, its own entry and the left tile's closing one, and moved rows never hand slots back: when it does run out, the insert says so below)
        uint32_t a = 0u;
#pragma unroll
        for (int k = 0; k < 8; k++) { T.ovf[(size_t)slot_o * 8u + k] = a; a += s_cnt[i * 8 + k]; }
        *dst = make_uint2(pre | 0x80000000u, slot_o);
      } else if (T.full_mail) {
        *T.full_mail = 1u;                 // out of escape slots: the entry stays stale -- the host lays the index out afresh before anything reads it
      }
    };
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int i = 2 * t + h, sg = (w0 >> 3) + i;
      if (sg < nseg) {
        const uint32_t sl = (uint32_t)sg & (TS - 1u);
        if (sl == 0u) T.xstart[xrow + (uint32_t)(sg >> g.ts) * (uint32_t)(g.ny + 2 * GRID_PAD)] = start + s_pre[i];
        const uint32_t tile = T.dir[tab_dir_index(g, py, pz, (uint32_t)sg)];
        if (tile) {
          put(i, tile, sl);
          if (sl == TS - 1u) put(i + 1, tile, TS);               // the closing entry: the segment behind the tile
        }
      }
    }
    carry += win_total;
    __syncthreads();
  }
}
// Full build.  The rows are laid out behind each other in (z, y) order, each with ROOM behind its last point when the map is one
// that receives inserts (`slack`: half its length, at least two points; an empty row has none and goes to the end of the array
// with its first points) -- a packed layout would send every row the next insert touches to the end of the array at once.
//   rows_len_kernel   : per row, where its keys begin in the sorted keys and how much room it gets
//   (exclusive sum of the rooms: the rows' first positions)
//   rows_place_kernel : sorted point i -> its row's first position + its rank in the row
//   rows_build_kernel : one workgroup per row: xstart and the entries
__global__ __launch_bounds__(256) void rows_len_kernel(const ckey_t* __restrict__ keys, uint32_t n, uint32_t nrows, int nxs, int slack,
                                                       uint32_t* __restrict__ row_lo /*[nrows + 1]*/, uint32_t* __restrict__ room) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const uint32_t lo = lower_bound_k(keys, 0u, n, (ckey_t)r * (ckey_t)nxs), hi = lower_bound_k(keys, lo, n, ((ckey_t)r + 1ull) * (ckey_t)nxs);
  const uint32_t len = hi - lo;
  row_lo[r] = lo;
  if (r + 1u == nrows) row_lo[nrows] = n;
  room[r] = (slack && len) ? len + max(2u, len >> 1) : len;
}
__global__ __launch_bounds__(256) void rows_place_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ perm, const ckey_t* __restrict__ keys,
                                                         uint32_t n, int nxs, const uint32_t* __restrict__ row_lo, const uint32_t* __restrict__ row_off,
                                                         float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = (uint32_t)(keys[i] / (ckey_t)nxs);
  out[row_off[r] + 1u + (i - row_lo[r])] = in[perm[i]];      // (+1: position 0 of the array is nobody's -- an entry that reads 0 was never written)
}
__global__ __launch_bounds__(256) void rows_build_kernel(SegTab T, uint32_t* __restrict__ rowcap, uint32_t* __restrict__ tail, const ckey_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ row_lo, const uint32_t* __restrict__ room,
                                                         const uint32_t* __restrict__ row_off, uint32_t nrows) {
  __shared__ uint32_t s_cnt[SEG_WIN + 8], s_pre[513], s_scan[256];
  const TabGeo& g = T.g;
  const uint32_t r = blockIdx.x;
  const ckey_t first = (ckey_t)r * (ckey_t)g.nxs;
  const uint32_t lo = row_lo[r], len = row_lo[r + 1u] - lo, cap = room[r], start = row_off[r] + 1u;
  const uint32_t py = r % (uint32_t)g.ny + GRID_PAD, pz = r / (uint32_t)g.ny + GRID_PAD;
  if (threadIdx.x == 0) {
    rowcap[pz * (uint32_t)(g.ny + 2 * GRID_PAD) + py] = cap;
    if (r + 1u == nrows) tail[0] = start + cap;                    // first free position behind the last row
  }
  if (len == 0u) {
    // an empty row: no entries (the pool was cleared); its xstarts say where it would begin (it has no room there: its first
    // points take it to the end of the array)
    const uint32_t xrow = (pz * (uint32_t)g.ntx) * (uint32_t)(g.ny + 2 * GRID_PAD) + py;
    for (int tx = (int)threadIdx.x; tx < g.ntx; tx += 256) T.xstart[xrow + (uint32_t)tx * (uint32_t)(g.ny + 2 * GRID_PAD)] = start;
    return;
  }
  const ckey_t* rk = keys + lo;
  row_entries(T, py, pz, start, len, [&](uint32_t i) { return (int)(rk[i] - first); }, s_cnt, s_pre, s_scan);
}
static hipError_t ensure_keys(MapBuildScratch& S, size_t n) {
  if (n <= S.ck_cap) return hipSuccess;
  if (S.ck_in) { (void)hipFree(S.ck_in); (void)hipFree(S.ck_out); }
  S.ck_in = S.ck_out = nullptr; S.ck_cap = 0;
  const size_t cap = n + n / 4 + 1024;
  hipError_t e;
  if ((e = hipMalloc(&S.ck_in, cap * sizeof(unsigned long long))) != hipSuccess) return e;
  if ((e = hipMalloc(&S.ck_out, cap * sizeof(unsigned long long))) != hipSuccess) return e;
  S.ck_cap = cap;
  return hipSuccess;
}
// (column keys S.ck_in -> S.ck_out, their points' numbers S.vals_in -> S.vals_out; `bits`: the key bits the grid uses)
static hipError_t sort_keys(hipStream_t st, size_t n, int bits, MapBuildScratch& S) {
  size_t tmp_bytes = 0;
  hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, S.ck_in, S.ck_out, S.vals_in, S.vals_out, n, 0, (unsigned)bits, st);
  if (e != hipSuccess) return e;
  if (tmp_bytes > S.cub_tmp_bytes) {
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;   // (an earlier launch may still be using cub_tmp)
    if (S.cub_tmp) (void)hipFree(S.cub_tmp);
    S.cub_tmp = nullptr; S.cub_tmp_bytes = 0;
    if ((e = hipMalloc(&S.cub_tmp, tmp_bytes + 1024)) != hipSuccess) return e;
    S.cub_tmp_bytes = tmp_bytes + 1024;
  }
  return rocprim::radix_sort_pairs(S.cub_tmp, tmp_bytes, S.ck_in, S.ck_out, S.vals_in, S.vals_out, n, 0, (unsigned)bits, st);
}
void index_view(const IndexTables& T, GridView& G) {
  TileShape t = T.shape;                                          // (the shape the index was laid out with: a grown grid keeps it)
  grid_tile_extent(t, G.nxf, G.ny, G.nz);
  G.tiles = T.tiles; G.dir = T.dir; G.ovf = T.ovf; G.xstart = T.xstart;
  G.ts = t.ts; G.ty = t.ty; G.tz = t.tz; G.ntx = t.ntx; G.nty = t.nty; G.ntz = t.ntz;
}
bool index_merge_overflow(const MapBuildScratch& S) { return S.mail_host && (S.mail_host[MAIL_TILES + 3] != 0u || S.mail_host[MAIL_ROWS] != 0u); }
void index_free(IndexTables& T) {
  (void)hipFree(T.tiles); (void)hipFree(T.dir); (void)hipFree(T.need); (void)hipFree(T.ovf); (void)hipFree(T.xstart); (void)hipFree(T.rowcap); (void)hipFree(T.rowoff);      // (counters and tail live behind `need`)
  (void)hipFree(T.xstart_alt); (void)hipFree(T.rowcap_alt); (void)hipFree(T.dir_alt);
  T = IndexTables{};
}
template <typename P>
static hipError_t grow(P*& p, size_t& cap, size_t need, size_t slack) {
  if (need <= cap) return hipSuccess;
  if (p) (void)hipFree(p);
  p = nullptr; cap = 0;
  hipError_t e = hipMalloc(&p, (need + slack) * sizeof(P));
  if (e == hipSuccess) cap = need + slack;
  return e;
}
static TabGeo tab_geo(int nx, int ny, int nz, int xs, const TileShape& ts) {
  return TabGeo{nx * xs + 1, ny, nz, ts.ts, ts.ty, ts.tz, ts.ntx, ts.nty, ts.ntz};
}
hipError_t map_build_grid(hipStream_t st, const float4* pts_in, size_t n, float4* pts_out, size_t out_cap, bool slack, IndexTables& T, size_t pts_cap,
                          const GridView& geo, MapBuildScratch& S) {
  const int nx = geo.nx, ny = geo.ny, nz = geo.nz, xs = geo.xs;
  const int nxf = nx * xs, nxs = nxf + 1;
  const size_t nrows = (size_t)ny * nz, ncols = nrows * (size_t)nxs;
  if (out_cap < n + 1) return hipErrorInvalidValue;                // (position 0 is nobody's: n points take n + 1 places)
  const TileShape shape = grid_tile_shape(nxf, ny, nz);
  T.shape = shape;
  const TabGeo g = tab_geo(nx, ny, nz, xs, shape);
  const int ndir = shape.ntx * shape.nty * shape.ntz;
  hipError_t e = ensure_scratch(S, std::max(n, nrows + 1));       // (the rows' key ranges and rooms live in the sort's input buffers)
  if (e != hipSuccess) return e;
  if ((e = ensure_mail(S)) != hipSuccess) return e;
  if ((e = ensure_keys(S, n)) != hipSuccess) return e;
  if (!T.dir) {
    if ((e = hipMalloc(&T.dir, (GRID_DIR_MAX + 8) * sizeof(uint16_t))) != hipSuccess) return e;
    // (need, counters and tail are one allocation: one clear per layout)
    if ((e = hipMalloc(&T.need, (GRID_DIR_MAX + 8) * sizeof(uint32_t))) != hipSuccess) return e;
    T.counters = T.need + GRID_DIR_MAX;
    T.tail = T.counters + 4;
  }
  {
    size_t slots = T.ovf_cap;
    if ((e = grow(T.ovf, slots, (pts_cap / 16 + 64) * 8, 0)) != hipSuccess) return e;
    T.ovf_cap = slots;
    if ((e = grow(T.xstart, T.xstart_cap, grid_xstart_size(ny, nz, shape.ntx), grid_xstart_size(ny, nz, shape.ntx) / 2)) != hipSuccess) return e;
  }
  const int blocks = (int)((n + 255) / 256);
  if (blocks > 0)
    hipLaunchKernelGGL(cellkey_kernel, dim3(blocks), dim3(256), 0, st, pts_in, n, cell_geo(geo), S.ck_in, S.vals_in);
  int bits = 1;                                                    // number of key bits actually used
  while (bits < 64 && ((size_t)1 << bits) < ncols) bits++;
  if (n > 0 && (e = sort_keys(st, n, bits, S)) != hipSuccess) return e;
  // the rows' places: key range and room per row, an exclusive sum of the rooms, every point to its row's place + its rank
  // (slack: only as far as the output array has room for it -- half a row's length and two points per row at least)
  const bool roomy = slack && (n + n / 2 + 2 * std::min(n, nrows) + 64 <= out_cap);
  uint32_t* row_lo = S.keys_in;                                    // [nrows + 1]
  uint32_t* room = S.vals_in;                                      // [nrows]
  {
    hipLaunchKernelGGL(rows_len_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, st, S.ck_out, (uint32_t)n, (uint32_t)nrows, nxs, roomy ? 1 : 0,
                       row_lo, room);
    size_t off_cap = T.rowoff_cap;
    if ((e = grow(T.rowoff, off_cap, nrows, nrows / 2)) != hipSuccess) return e;
    T.rowoff_cap = off_cap;
    size_t scan_bytes = 0;
    if ((e = exclusive_sum(nullptr, scan_bytes, room, T.rowoff, nrows, st)) != hipSuccess) return e;
    if (scan_bytes > S.cub_tmp_bytes) {
      if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
      if (S.cub_tmp) (void)hipFree(S.cub_tmp);
      S.cub_tmp = nullptr; S.cub_tmp_bytes = 0;
      if ((e = hipMalloc(&S.cub_tmp, scan_bytes + 1024)) != hipSuccess) return e;
      S.cub_tmp_bytes = scan_bytes + 1024;
    }
    if ((e = exclusive_sum(S.cub_tmp, scan_bytes, room, T.rowoff, nrows, st)) != hipSuccess) return e;
    if (n > 0) hipLaunchKernelGGL(rows_place_kernel, dim3(blocks), dim3(256), 0, st, pts_in, S.vals_out, S.ck_out, (uint32_t)n, nxs, row_lo, T.rowoff, pts_out);
  }
  // which tiles exist, and their numbers (directory order); the host sizes the pool by their count
  const size_t te = grid_tile_entries(shape.ts, shape.ty, shape.tz);
  // a grid whose every tile fits in 32 MB (the second level over a crowded region) takes them all: nothing to count, no wait
  const bool all_tiles = !slack && (size_t)(ndir + 1) * te * sizeof(uint2) <= ((size_t)32 << 20);
  if ((e = hipMemsetAsync(T.need, 0, (GRID_DIR_MAX + 8) * sizeof(uint32_t), st)) != hipSuccess) return e;      // (need, counters, tail)
  if (all_tiles && (e = hipMemsetAsync(T.need, 0x01, (size_t)ndir * sizeof(uint32_t), st)) != hipSuccess) return e;      // (every tile is needed)
  if ((e = hipMemsetAsync(T.dir, 0, (GRID_DIR_MAX + 8) * sizeof(uint16_t), st)) != hipSuccess) return e;
  if (n > 0 && !all_tiles) hipLaunchKernelGGL(tiles_mark_kernel, dim3(blocks), dim3(256), 0, st, S.ck_out, (uint32_t)n, g, T.need);
  S.mail_host[MAIL_TILES] = 0u; S.mail_host[MAIL_TILES + 1] = 0u;
  hipLaunchKernelGGL(tiles_number_kernel, dim3(1), dim3(256), 0, st, T.dir, T.need, ndir, 65536u, T.counters, S.mail_dev + MAIL_TILES, 1);
  size_t ntiles = (size_t)ndir + 1;
  if (!all_tiles) {
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    ntiles = S.mail_host[MAIL_TILES];                              // the zero tile + the tiles that exist
    if (S.mail_host[MAIL_TILES + 1] || ntiles == 0) return hipErrorOutOfMemory;      // (more than 65535 tiles: the directory has at most 4096 entries)
  }
  {
    // the pool in TILES of this shape; room for the map to grow into (a merge that runs out lays the index out afresh)
    size_t cap_entries = T.tiles_cap_entries;
    if ((e = grow(T.tiles, cap_entries, (ntiles + std::max<size_t>(64, ntiles / 2)) * te, 0)) != hipSuccess) return e;
    T.tiles_cap_entries = cap_entries;
  }
  if ((ntiles + std::max<size_t>(64, ntiles / 2)) * te >= ((size_t)1 << 32)) return hipErrorOutOfMemory;      // (32-bit entry indices)
  T.cap_tiles = (uint32_t)std::min<size_t>(std::min<size_t>(T.tiles_cap_entries, ((size_t)1 << 32) - 1) / te, 65536);
  if ((e = hipMemsetAsync(T.tiles, 0, (size_t)T.cap_tiles * te * sizeof(uint2), st)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(T.xstart, 0, grid_xstart_size(ny, nz, shape.ntx) * sizeof(uint32_t), st)) != hipSuccess) return e;
  const SegTab tab{T.tiles, T.dir, T.ovf, T.counters + 2, (uint32_t)(T.ovf_cap / 8), T.xstart, g, nullptr};
  {
    size_t rc = T.rowcap_cap;
    if ((e = grow(T.rowcap, rc, ((size_t)ny + 2 * GRID_PAD) * ((size_t)nz + 2 * GRID_PAD), ((size_t)ny + 2 * GRID_PAD) * ((size_t)nz + 2 * GRID_PAD) / 2)) != hipSuccess) return e;
    T.rowcap_cap = rc;
  }
  if ((e = hipMemsetAsync(T.rowcap, 0, ((size_t)ny + 2 * GRID_PAD) * ((size_t)nz + 2 * GRID_PAD) * sizeof(uint32_t), st)) != hipSuccess) return e;
  hipLaunchKernelGGL(rows_build_kernel, dim3((unsigned)nrows), dim3(256), 0, st, tab, T.rowcap, T.tail, S.ck_out, row_lo, room, T.rowoff, (uint32_t)nrows);
  T.tiles_used = (uint32_t)ntiles;
  return hipGetLastError();
}

// ---- incremental update of the cell-sorted map ------------------------------------------------
// The map only grows (the insert rule drops incoming points, never stored ones), so while the grid geometry still covers the map
// the k points appended since the last build are put into their rows instead of sorting everything again: the new points are
// sorted by column key (k is a scan, not the map) and every row that receives some is rewritten -- exactly the rows a stable sort
// of (stored points..., new points...) by column key gives.
// One row's points are one contiguous run of `pts`, but the rows are NOT packed behind each other (round 5): `xstart` says where a
// row begins, `rowcap` how many points fit before the next one.  A build leaves room behind every row's last point; an insert touches only
// the rows that receive points: a row whose new points fit is merged in place (old points move up inside the row, from the back),
// a row that outgrows its room moves to the end of the array with room to double.  No pass over the stored points, no shift of
// any other row.  What a row leaves behind is garbage until the next full build (which the insert asks for when the array is full).
// Order inside a row: by column, stored points before new ones, new ones in their batch order -- exactly what a stable sort of
// (stored..., new...) by column key gives, i.e. what map_build_grid produces for the same points.
__device__ __forceinline__ int point_column(const float4* __restrict__ p, const CellGeo& c, ckey_t first) {
  // (points this workgroup has just written: read past the vector L1)
  const unsigned* w = reinterpret_cast<const unsigned*>(p);
  float4 q;
  q.x = __uint_as_float(__hip_atomic_load(w + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  q.y = __uint_as_float(__hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  q.z = __uint_as_float(__hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  q.w = 0.f;
  return (int)(column_key(q, c) - first);
}
// tail[0] = first free position behind the last row, tail[1] = 1: the array is full (nothing was written for that row; the host
// lays the map out afresh), tail[2] = rows moved (statistics)
// the rows a batch of sorted keys touches: (row, index of its first key), one entry per run of keys of one row; tail[3] counts them
__global__ __launch_bounds__(256) void rows_touched_kernel(const ckey_t* __restrict__ nkeys, uint32_t k, int nxs, uint32_t* __restrict__ rows,
                                                           uint32_t* __restrict__ firsts, uint32_t* __restrict__ count) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  const uint32_t r = (uint32_t)(nkeys[j] / (ckey_t)nxs);
  if (j != 0u && (uint32_t)(nkeys[j - 1u] / (ckey_t)nxs) == r) return;
  const uint32_t slot = atomicAdd(count, 1u);
  rows[slot] = r; firsts[slot] = j;
}
// (one workgroup per touched row, a fixed number of workgroups walking the list)
__global__ __launch_bounds__(256) void rows_insert_kernel(SegTab T, uint32_t* __restrict__ rowcap, uint32_t* __restrict__ tail, uint32_t* __restrict__ full_mail, uint32_t pts_capacity,
                                                          float4* __restrict__ pts, const ckey_t* __restrict__ nkeys, uint32_t k,
                                                          const uint32_t* __restrict__ rows, const uint32_t* __restrict__ firsts,
                                                          const float4* __restrict__ new_pts, const uint32_t* __restrict__ nperm, CellGeo cg) {
  __shared__ uint32_t s_cnt[SEG_WIN + 8], s_pre[513], s_scan[256], s_hi, s_dest;
  const TabGeo& g = T.g;
  const int t = (int)threadIdx.x;
  const uint32_t ntouched = tail[3];
 for (uint32_t item = blockIdx.x; item < ntouched; item += gridDim.x) {
  __syncthreads();                                                 // (the last row's shared memory is done with)
  const uint32_t r = rows[item], lo = firsts[item];
  const ckey_t first = (ckey_t)r * (ckey_t)g.nxs;
  if (t < 64) { const uint32_t v = wave_lower_bound_k(nkeys, k, first + (ckey_t)g.nxs); if (t == 0) s_hi = v; }
  __syncthreads();
  const uint32_t hi = s_hi;
  const uint32_t py = r % (uint32_t)g.ny + GRID_PAD, pz = r / (uint32_t)g.ny + GRID_PAD;
  const uint32_t stride = (uint32_t)(g.ny + 2 * GRID_PAD);
  const uint32_t prow = pz * stride + py;
  const uint32_t xrow = (pz * (uint32_t)g.ntx) * stride + py;
  const uint32_t TS = 1u << g.ts;
  // the row before the insert: first point, points below a column (from the old entries: they stay until row_entries rewrites them)
  const uint32_t start_old = T.xstart[xrow];
  auto below_old = [&](uint32_t col) -> uint32_t {                  // stored points of the row in columns < col
    const uint32_t sg = col >> 3;
    const uint32_t tile = T.dir[tab_dir_index(g, py, pz, sg)];
    const uint32_t xs_ = T.xstart[xrow + (sg >> g.ts) * stride];
    const uint2 e = T.tiles[(((size_t)tile << (g.ty + g.tz)) + (size_t)(((pz & ((1u << g.tz) - 1u)) << g.ty) + (py & ((1u << g.ty) - 1u)))) * (TS + 1u) + (sg & (TS - 1u))];
    // (entries hold positions; no tile, or a tile that came into being for this batch and has nothing of this row yet -- an entry
    //  that reads 0: no position is 0 -- : the row's xstart there)
    return ((tile && e.x) ? 0u : xs_) + seg_count(e.x, e.y, col & 7u, T.ovf) - start_old;
  };
  const uint32_t len_old = below_old((uint32_t)(g.nxs - 1)), add = hi - lo, need = len_old + add;
  if (t == 0) {
    uint32_t dest = start_old;
    if (need > rowcap[prow]) {                                     // the row moves behind everything, with room to double
      const uint32_t ncap = 2u * need + 8u;
      const uint32_t d = atomicAdd(&tail[0], ncap);
      if (d > pts_capacity || ncap > pts_capacity - d) { atomicExch(&tail[1], 1u); *full_mail = 1u; dest = 0xffffffffu; }
      else { dest = d; rowcap[prow] = ncap; atomicAdd(&tail[2], 1u); }
    }
    s_dest = dest;
  }
  __syncthreads();
  const uint32_t dest = s_dest;
  if (dest == 0xffffffffu) continue;                               // (block-uniform)
  const float4* src = pts + start_old;
  float4* dst = pts + dest;
  // stored points: point i goes to i + #new points in columns below its own.  From the back, a chunk at a time: read, barrier,
  // write -- a point only ever moves up, by no less than the one before it, so a chunk's writes land on positions already read.
  // (a row merged where it is: the points in columns below the first new one stay; the row's new keys, when they are few, are
  //  searched in shared memory -- s_pre is free until row_entries)
  const uint32_t i_stay = (dest == start_old) ? below_old((uint32_t)(nkeys[lo] - first)) : 0u;
  const bool keys_lds = add <= 512u;
  if (keys_lds) for (uint32_t j = (uint32_t)t; j < add; j += 256u) s_pre[j] = (uint32_t)(nkeys[lo + j] - first);      // (the row's new COLUMNS)
  __syncthreads();
  for (int c0 = (int)((len_old + 1023u) / 1024u) * 1024 - 1024; c0 >= 0 && (uint32_t)(c0 + 1024) > i_stay; c0 -= 1024) {
    // (1024 points a round, four per thread: read all, barrier, write all -- a point only ever moves up, by no less than the one
    //  before it, so a round's writes land on positions this round or an earlier one has read)
    float4 p[4];
    uint32_t sh[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t i = (uint32_t)(c0 + t + 256 * u);
      p[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < len_old && i >= i_stay) p[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t i = (uint32_t)(c0 + t + 256 * u);
      sh[u] = 0u;
      if (i < len_old && i >= i_stay) {
        const ckey_t key = column_key(p[u], cg);
        const uint32_t colk = (uint32_t)(key - first);
        if (keys_lds) {
          uint32_t a = 0u, b = add;
          while (a < b) { const uint32_t m = (a + b) >> 1; if (s_pre[m] < colk) a = m + 1u; else b = m; }
          sh[u] = a;
        } else {
          sh[u] = lower_bound_k(nkeys, lo, hi, key) - lo;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t i = (uint32_t)(c0 + t + 256 * u);
      if (i < len_old && i >= i_stay && (sh[u] != 0u || dest != start_old)) dst[i + sh[u]] = p[u];
    }
    __syncthreads();
  }
  // new points: the j-th of the row goes behind the stored points of its column
  for (uint32_t j = lo + (uint32_t)t; j < hi; j += 256u) {
    const uint32_t col = (uint32_t)(nkeys[j] - first);
    dst[(j - lo) + below_old(col + 1u)] = new_pts[nperm[j]];
  }
  __syncthreads();                                                 // (everybody is done with the old entries, every point is in place)
  row_entries(T, py, pz, dest, need,
              [&](uint32_t i) { return point_column(dst + i, cg, first); }, s_cnt, s_pre, s_scan);
 }
}
hipError_t map_merge_grid(hipStream_t st, float4* sorted, size_t sorted_cap, const float4* new_pts, size_t k,
                          IndexTables& T, const GridView& geo, MapBuildScratch& S) {
  if (k == 0) return hipSuccess;
  const int nx = geo.nx, ny = geo.ny, nz = geo.nz, xs = geo.xs;
  const int nxs = nx * xs + 1;
  const size_t nrows = (size_t)ny * nz, ncols = nrows * (size_t)nxs;
  TileShape shape = T.shape;
  grid_tile_extent(shape, nx * xs, ny, nz);
  const TabGeo g = tab_geo(nx, ny, nz, xs, shape);
  hipError_t e = ensure_scratch(S, k);
  if (e != hipSuccess) return e;
  if ((e = ensure_keys(S, k)) != hipSuccess) return e;
  const int kb = (int)((k + 255) / 256);
  hipLaunchKernelGGL(cellkey_kernel, dim3(kb), dim3(256), 0, st, new_pts, k, cell_geo(geo), S.ck_in, S.vals_in);
  int bits = 1;
  while (bits < 64 && ((size_t)1 << bits) < ncols) bits++;
  if ((e = sort_keys(st, k, bits, S)) != hipSuccess) return e;
  // tiles the new points need and the map did not have take the next numbers of the pool (cleared when it was laid out)
  hipLaunchKernelGGL(tiles_mark_kernel, dim3(kb), dim3(256), 0, st, S.ck_out, (uint32_t)k, g, T.need);
  hipLaunchKernelGGL(tiles_number_kernel, dim3(1), dim3(256), 0, st, T.dir, T.need, shape.ntx * shape.nty * shape.ntz, T.cap_tiles, T.counters,
                     S.mail_dev + MAIL_TILES + 2, 0);
  const SegTab tab{T.tiles, T.dir, T.ovf, T.counters + 2, (uint32_t)(T.ovf_cap / 8), T.xstart, g, S.mail_dev + MAIL_ROWS};
  // the rows that receive points (the sort's input buffers are free again: they take the list)
  if ((e = hipMemsetAsync(T.tail + 3, 0, sizeof(uint32_t), st)) != hipSuccess) return e;
  hipLaunchKernelGGL(rows_touched_kernel, dim3(kb), dim3(256), 0, st, S.ck_out, (uint32_t)k, nxs, S.keys_in, S.vals_in, T.tail + 3);
  const unsigned walkers = (unsigned)std::min<size_t>(std::min<size_t>(k, nrows), 8192);
  hipLaunchKernelGGL(rows_insert_kernel, dim3(walkers), dim3(256), 0, st, tab, T.rowcap, T.tail, S.mail_dev + MAIL_ROWS, (uint32_t)std::min<size_t>(sorted_cap, 0x7fffffffull),
                     sorted, S.ck_out, (uint32_t)k, S.keys_in, S.vals_in, new_pts, S.vals_out, cell_geo(geo));
  return hipGetLastError();
}

// ---- second level over crowded regions ----------------------------------------------------------------------------------
// Raw sweeps inserted into the map leave cells right under the sensor with hundreds to thousands of points (the insert rule
// keeps the whole first batch that lands in a leaf).  A query there has its five neighbours within centimetres but would walk
// every point of its 3x3x3 cells.  The map therefore keeps a SECOND grid with a quarter of the cell edge over the box around the
// crowded cells, holding a copy of every map point inside that box (w = position in the main sorted map, so that neighbour
// ids and tie-breaks are the main map's); the pass asks it first (fine pre-pass, flimo_kernels.hip).
// Crowded cells (more than `threshold` points) are LISTED once each (a bit per cell marks the listed ones); the host keeps the
// list and boxes the part of it around the sensor.  crowded_list_all looks at every cell (new geometry: bits and list start
// empty), crowded_list_points only at the cells of k freshly merged points (cells gain points nowhere else).
__device__ __forceinline__ void crowded_append(uint32_t cell, int x, int y, int z, uint32_t* __restrict__ bits, int4* __restrict__ list,
                                               uint32_t cap, uint32_t* __restrict__ count) {
  const uint32_t m = 1u << (cell & 31u);
  if (atomicOr(&bits[cell >> 5], m) & m) return;
  const uint32_t slot = atomicAdd(count, 1u);
  if (slot < cap) list[slot] = make_int4(x, y, z, 0);
}
__global__ __launch_bounds__(256) void crowded_all_kernel(GridView G, uint32_t threshold, uint32_t* __restrict__ bits, int4* __restrict__ list,
                                                          uint32_t cap, uint32_t* __restrict__ count) {
  const size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t ncells = (size_t)G.nx * G.ny * G.nz;
  if (c >= ncells) return;
  const int x = (int)(c % (size_t)G.nx);
  const size_t row = c / (size_t)G.nx;
  const int y = (int)(row % (size_t)G.ny), z = (int)(row / (size_t)G.ny);
  uint32_t lo, hi;
  grid_row_range(G, G.dir, y, z, x * G.xs, (x + 1) * G.xs, lo, hi);
  if (hi - lo <= threshold) return;
  crowded_append((uint32_t)c, x, y, z, bits, list, cap, count);
}
__global__ __launch_bounds__(256) void crowded_points_kernel(const float4* __restrict__ pts, size_t k, GridView G,
                                                             uint32_t threshold, uint32_t* __restrict__ bits, int4* __restrict__ list,
                                                             uint32_t cap, uint32_t* __restrict__ count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= k) return;
  const ckey_t col = column_key(pts[i], CellGeo{G.ox, G.oy, G.oz, G.inv_cell, G.nx, G.ny, G.nz, G.xs, G.six, G.siy, G.siz});
  const uint32_t row = (uint32_t)(col / (ckey_t)G.nxs), xf = (uint32_t)(col - (ckey_t)row * (ckey_t)G.nxs);
  const int x = (int)(xf / (uint32_t)G.xs), y = (int)(row % (uint32_t)G.ny), z = (int)(row / (uint32_t)G.ny);
  uint32_t lo, hi;
  grid_row_range(G, G.dir, y, z, x * G.xs, (x + 1) * G.xs, lo, hi);
  if (hi - lo <= threshold) return;
  crowded_append(row * (uint32_t)G.nx + (uint32_t)x, x, y, z, bits, list, cap, count);
}
// the bits of a list of crowded cells under another grid geometry (a grid that grew: index_regrid), count = the list's length
__global__ __launch_bounds__(256) void crowded_relist_kernel(const int4* __restrict__ list, uint32_t m, int nx, int ny, uint32_t* __restrict__ bits,
                                                             uint32_t* __restrict__ count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *count = m;
  if (i >= m) return;
  const int4 q = list[i];
  const uint32_t cell = ((uint32_t)q.z * (uint32_t)ny + (uint32_t)q.y) * (uint32_t)nx + (uint32_t)q.x;
  atomicOr(&bits[cell >> 5], 1u << (cell & 31u));
}
hipError_t crowded_relist(hipStream_t st, const int4* list, uint32_t m, int nx, int ny, int nz, uint32_t* bits, uint32_t* count_dev) {
  hipError_t e = hipMemsetAsync(bits, 0, (((size_t)nx * ny * nz + 31) / 32) * sizeof(uint32_t), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(crowded_relist_kernel, dim3((m + 255) / 256 + 1), dim3(256), 0, st, list, m, nx, ny, bits, count_dev);
  return hipGetLastError();
}
// *count_host = entries listed so far (may exceed cap: the list is then incomplete)
hipError_t crowded_list_all(hipStream_t st, const GridView& G, uint32_t threshold, uint32_t* bits,
                            int4* list, uint32_t cap, uint32_t* count_dev, uint32_t* count_host, MapBuildScratch& S) {
  const size_t ncells = (size_t)G.nx * G.ny * G.nz;
  hipError_t e;
  if ((e = hipMemsetAsync(bits, 0, ((ncells + 31) / 32) * sizeof(uint32_t), st)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(count_dev, 0, sizeof(uint32_t), st)) != hipSuccess) return e;
  hipLaunchKernelGGL(crowded_all_kernel, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, st, G, threshold, bits, list, cap, count_dev);
  const MailPart part{count_dev, 1, MAIL_CROWD};
  if ((e = mail_words(st, S, &part, 1)) != hipSuccess) return e;
  if ((e = mail_wait(st, S)) != hipSuccess) return e;
  *count_host = S.mail_host[MAIL_CROWD];
  return hipSuccess;
}
hipError_t crowded_list_points(hipStream_t st, const float4* pts, size_t k, const GridView& G, uint32_t threshold, uint32_t* bits, int4* list,
                               uint32_t cap, uint32_t* count_dev, uint32_t* count_host, MapBuildScratch& S) {
  hipError_t e;
  if (k > 0)
    hipLaunchKernelGGL(crowded_points_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, pts, k, G, threshold, bits, list, cap, count_dev);
  const MailPart part{count_dev, 1, MAIL_CROWD};
  if ((e = mail_words(st, S, &part, 1)) != hipSuccess) return e;
  if ((e = mail_wait(st, S)) != hipSuccess) return e;
  *count_host = S.mail_host[MAIL_CROWD];
  return hipSuccess;
}
// Copy of the map points of a box of cells [c0, c1] (inclusive, already clipped to the grid), w = position in the main sorted
// map.  The map is sorted by (z, y, x column), so the box is (y1-y0+1)(z1-z0+1) contiguous ranges read off the index:
// counting and copying cost what the box holds, not what the map holds.  Two steps, so that the caller can size its
// buffers (or give up) once it knows the count.
__global__ __launch_bounds__(256) void boxrows_count_kernel(GridView G, int x0, int x1, int y0, int nyb, int z0, int nrows,
                                                            uint32_t* __restrict__ cnt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  uint32_t lo, hi;
  grid_row_range(G, G.dir, y0 + r % nyb, z0 + r / nyb, x0 * G.xs, (x1 + 1) * G.xs, lo, hi);
  cnt[r] = hi - lo;
}
__global__ __launch_bounds__(64) void boxrows_total_kernel(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, int nrows,
                                                           uint32_t* __restrict__ total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *total = off[nrows - 1] + cnt[nrows - 1];
}
__global__ __launch_bounds__(64) void boxrows_copy_kernel(GridView G, int x0, int x1, int y0, int nyb, int z0, const uint32_t* __restrict__ cnt,
                                                          const uint32_t* __restrict__ off, float4* __restrict__ out) {
  const int r = blockIdx.x;
  uint32_t a, hi;
  grid_row_range(G, G.dir, y0 + r % nyb, z0 + r / nyb, x0 * G.xs, (x1 + 1) * G.xs, a, hi);
  const uint32_t n = cnt[r], o = off[r];
  for (uint32_t i = threadIdx.x; i < n; i += 64) {
    const float4 p = G.pts[a + i];
    out[o + i] = make_float4(p.x, p.y, p.z, __uint_as_float(a + i));
  }
}
hipError_t map_box_count(hipStream_t st, const GridView& G, const int c0[3], const int c1[3],
                         uint32_t* count_dev, uint32_t* count_host, MapBuildScratch& S) {
  *count_host = 0;
  const int nyb = c1[1] - c0[1] + 1, nzb = c1[2] - c0[2] + 1;
  if (nyb <= 0 || nzb <= 0 || c1[0] < c0[0]) return hipSuccess;
  const int nrows = nyb * nzb;
  hipError_t e = ensure_scratch(S, (size_t)nrows);
  if (e != hipSuccess) return e;
  size_t scan_bytes = 0;
  if ((e = exclusive_sum(nullptr, scan_bytes, S.keys_in, S.vals_in, (size_t)nrows, st)) != hipSuccess) return e;
  if (scan_bytes > S.cub_tmp_bytes) {
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    if (S.cub_tmp) (void)hipFree(S.cub_tmp);
    if ((e = hipMalloc(&S.cub_tmp, scan_bytes + 1024)) != hipSuccess) return e;
    S.cub_tmp_bytes = scan_bytes + 1024;
  }
  hipLaunchKernelGGL(boxrows_count_kernel, dim3((nrows + 255) / 256), dim3(256), 0, st, G, c0[0], c1[0], c0[1], nyb, c0[2], nrows, S.keys_in);
  if ((e = exclusive_sum(S.cub_tmp, scan_bytes, S.keys_in, S.vals_in, (size_t)nrows, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(boxrows_total_kernel, dim3(1), dim3(64), 0, st, S.keys_in, S.vals_in, nrows, S.mail_dev + MAIL_BOXCOUNT);
  (void)count_dev;
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  *count_host = S.mail_host[MAIL_BOXCOUNT];
  return hipSuccess;
}
// second step: the counts / offsets of map_box_count (same box, nothing else used the scratch in between) -> the copies
hipError_t map_box_copy(hipStream_t st, const GridView& G, const int c0[3], const int c1[3], float4* out, MapBuildScratch& S) {
  const int nyb = c1[1] - c0[1] + 1, nzb = c1[2] - c0[2] + 1;
  if (nyb <= 0 || nzb <= 0 || c1[0] < c0[0]) return hipSuccess;
  hipLaunchKernelGGL(boxrows_copy_kernel, dim3(nyb * nzb), dim3(64), 0, st, G, c0[0], c1[0], c0[1], nyb, c0[2], S.keys_in, S.vals_in, out);
  return hipGetLastError();
}

// The device's atan2f (flimo_math.h: libm_atan2f, fdlibm's routine as glibc up to 2.40 evaluates it) on n argument pairs: the host
// compares with ITS libm once per context before the FoV filter may run on the device (flimo_capi.hip: fov_selfcheck)
__global__ void atan2f_probe_kernel(const float2* __restrict__ yx, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = libm_atan2f(yx[i].x, yx[i].y);
}
hipError_t atan2f_probe(hipStream_t st, const float* yx_host, int n, float* out_host) {
  float2* d_in = nullptr; float* d_out = nullptr;
  hipError_t e;
  if ((e = hipMalloc(&d_in, (size_t)n * sizeof(float2))) != hipSuccess) return e;
  if ((e = hipMalloc(&d_out, (size_t)n * sizeof(float))) != hipSuccess) { (void)hipFree(d_in); return e; }
  e = hipMemcpyAsync(d_in, yx_host, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) { hipLaunchKernelGGL(atan2f_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_in, n, d_out); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(out_host, d_out, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_in); (void)hipFree(d_out);
  return e;
}

// ---- a grid that grows without a re-sort ---------------------------------------------------------------------------------------
// The origin of the cells is fixed (GridView): a larger grid is the same lattice with another corner (six, siy, siz) and other
// extents.  When the corner moves by whole tiles and the tile shape stays, nothing a tile holds changes -- its counts are relative
// to its own row segments -- and the sorted rows stay where they are: only the small tables are laid out afresh.
//   regrid_rows_kernel : xstart and rowcap of every row of the new grid (an old row: its values, its row start / end for x-tiles
//                        the old grid did not have; a new row: empty, no room -- its first points take it to the array's end)
//   regrid_dir_kernel  : the directory (an old tile keeps its number)
__global__ __launch_bounds__(256) void regrid_rows_kernel(GridView O, GridView N, uint32_t* __restrict__ xstart_new, uint32_t* __restrict__ rowcap_new,
                                                          const uint32_t* __restrict__ rowcap_old) {
  const uint32_t sy = (uint32_t)(N.ny + 2 * GRID_PAD), sz = (uint32_t)(N.nz + 2 * GRID_PAD);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sy * sz) return;
  const uint32_t py = i % sy, pz = i / sy;
  // the same row in the old grid (cell of the origin's lattice = y + siy)
  const int yo = (int)py - GRID_PAD + N.siy - O.siy, zo = (int)pz - GRID_PAD + N.siz - O.siz;
  const bool had = yo >= 0 && yo < O.ny && zo >= 0 && zo < O.nz;
  const int dtx = ((N.six - O.six) * N.xs) >> (3 + N.ts);          // x-tile of the old grid = x-tile of the new one + dtx (whole tiles)
  uint32_t first = 0u, last = 0u;
  if (had) { first = grid_pos(O, O.dir, yo, zo, 0); last = grid_pos(O, O.dir, yo, zo, O.nxf); }
  rowcap_new[pz * sy + py] = had ? rowcap_old[(uint32_t)(zo + GRID_PAD) * (uint32_t)(O.ny + 2 * GRID_PAD) + (uint32_t)(yo + GRID_PAD)] : 0u;
  for (int tx = 0; tx < N.ntx; tx++) {
    uint32_t v = 0u;
    if (had) {
      const int to = tx + dtx;
      v = to < 0 ? first : (to >= O.ntx ? last : O.xstart[grid_xstart_index(O, (uint32_t)(yo + GRID_PAD), (uint32_t)(zo + GRID_PAD), (uint32_t)to)]);
    }
    xstart_new[(pz * (uint32_t)N.ntx + (uint32_t)tx) * sy + py] = v;
  }
}
__global__ __launch_bounds__(256) void regrid_dir_kernel(GridView O, GridView N, uint16_t* __restrict__ dir_new) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N.ntx * N.nty * N.ntz) return;
  const int tx = i % N.ntx, ty = (i / N.ntx) % N.nty, tz = i / (N.ntx * N.nty);
  const int to_x = tx + (((N.six - O.six) * N.xs) >> (3 + N.ts)), to_y = ty + ((N.siy - O.siy) >> N.ty), to_z = tz + ((N.siz - O.siz) >> N.tz);
  uint16_t v = 0;
  if (to_x >= 0 && to_x < O.ntx && to_y >= 0 && to_y < O.nty && to_z >= 0 && to_z < O.ntz) v = O.dir[(to_z * O.nty + to_y) * O.ntx + to_x];
  dir_new[i] = v;
}
// N: the new geometry (extents, shifts; same origin, cell, column factor and tile shape as O, corner moved by whole tiles).  The
// tables of T are replaced, the tiles, the escapes and the points stay; N gets its views.
hipError_t index_regrid(hipStream_t st, IndexTables& T, const GridView& O, GridView& N) {
  TileShape sh = T.shape;                                          // the grid keeps its tile shape while the directory has room
  grid_tile_extent(sh, N.nxf, N.ny, N.nz);
  if ((long long)sh.ntx * sh.nty * sh.ntz > GRID_DIR_MAX) return hipErrorInvalidValue;
  N.ts = sh.ts; N.ty = sh.ty; N.tz = sh.tz; N.ntx = sh.ntx; N.nty = sh.nty; N.ntz = sh.ntz;
  const size_t nrowsp = ((size_t)N.ny + 2 * GRID_PAD) * ((size_t)N.nz + 2 * GRID_PAD), nx = grid_xstart_size(N.ny, N.nz, N.ntx);
  // the new tables go into the index's second set of buffers (kept between calls: no allocation, no wait on the way), then the
  // two sets change places
  hipError_t e;
  if ((e = grow(T.xstart_alt, T.xstart_alt_cap, nx, nx / 2)) != hipSuccess) return e;
  if ((e = grow(T.rowcap_alt, T.rowcap_alt_cap, nrowsp, nrowsp / 2)) != hipSuccess) return e;
  if (!T.dir_alt && (e = hipMalloc(&T.dir_alt, (GRID_DIR_MAX + 8) * sizeof(uint16_t))) != hipSuccess) return e;
  if ((e = hipMemsetAsync(T.dir_alt, 0, (GRID_DIR_MAX + 8) * sizeof(uint16_t), st)) != hipSuccess) return e;
  hipLaunchKernelGGL(regrid_rows_kernel, dim3((unsigned)((nrowsp + 255) / 256)), dim3(256), 0, st, O, N, T.xstart_alt, T.rowcap_alt, T.rowcap);
  hipLaunchKernelGGL(regrid_dir_kernel, dim3((unsigned)((sh.ntx * sh.nty * sh.ntz + 255) / 256)), dim3(256), 0, st, O, N, T.dir_alt);
  std::swap(T.xstart, T.xstart_alt); std::swap(T.xstart_cap, T.xstart_alt_cap);
  std::swap(T.rowcap, T.rowcap_alt); std::swap(T.rowcap_cap, T.rowcap_alt_cap);
  std::swap(T.dir, T.dir_alt);
  N.tiles = T.tiles; N.dir = T.dir; N.ovf = T.ovf; N.xstart = T.xstart;
  return hipGetLastError();
}

// more room in the pool of tiles (same shape): the tiles move to a larger array, the new ones are cleared.  Ends with the stream
// waited for (the old array is freed).
hipError_t index_grow_pool(hipStream_t st, IndexTables& T, uint32_t cap_tiles_new) {
  const size_t te = grid_tile_entries(T.shape.ts, T.shape.ty, T.shape.tz);
  cap_tiles_new = (uint32_t)std::min<size_t>(std::min<size_t>(cap_tiles_new, 65536), (((size_t)1 << 32) - 1) / te);
  if (cap_tiles_new <= T.cap_tiles) return hipSuccess;
  uint2* np = nullptr;
  hipError_t e = hipMalloc(&np, (size_t)cap_tiles_new * te * sizeof(uint2));
  if (e != hipSuccess) return e;
  if ((e = hipMemcpyAsync(np, T.tiles, (size_t)T.cap_tiles * te * sizeof(uint2), hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(np + (size_t)T.cap_tiles * te, 0, (size_t)(cap_tiles_new - T.cap_tiles) * te * sizeof(uint2), st)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  (void)hipFree(T.tiles);
  T.tiles = np;
  T.tiles_cap_entries = (size_t)cap_tiles_new * te;
  T.cap_tiles = cap_tiles_new;
  return hipSuccess;
}

// ---- debug: two indices of the same geometry say the same (flimo_map_grid_selfcheck) -- compared by MEANING, column by column:
//      escapes take their slots in the order the workgroups arrive ----
__global__ __launch_bounds__(256) void index_compare_kernel(GridView A, GridView B, unsigned long long* __restrict__ diff) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)A.nxf + 1;
  if (i >= (size_t)A.ny * A.nz * per) return;
  const int row = (int)(i / per), col = (int)(i % per);
  const int y = row % A.ny, z = row / A.ny;
  // (relative to the row's first point: the rows of a maintained map are not packed behind each other)
  if (grid_pos(A, A.dir, y, z, col) - grid_pos(A, A.dir, y, z, 0) != grid_pos(B, B.dir, y, z, col) - grid_pos(B, B.dir, y, z, 0)) {
#ifdef FLIMO_SELFCHECK_DEBUG
    printf("index_compare: row (y %d, z %d) col %d: ref %u - %u, live %u - %u\n", y, z, col, grid_pos(A, A.dir, y, z, col), grid_pos(A, A.dir, y, z, 0),
           grid_pos(B, B.dir, y, z, col), grid_pos(B, B.dir, y, z, 0));
#endif
    atomicAdd(diff, 1ull);
  }
  // ... and the way the k-NN pass's fast path reads a range (flimo_kernels.hip: knn5_pass): both ends from the TWO NEIGHBOURING
  // ENTRIES of the tile of the range's first column, no xstart -- a tile that does not exist, or a row that has nothing in it, must
  // read as an empty range there exactly when the range IS empty.  Every range of one column and of three cells that starts here.
  const uint32_t py = (uint32_t)(y + GRID_PAD), pz = (uint32_t)(z + GRID_PAD);
  const int widths[2] = {1, 3 * B.xs};
  for (int wi = 0; wi < 2; wi++) {
    const int c0 = col, c1 = min(col + widths[wi], B.nxf);
    if (c1 <= c0) continue;
    const uint32_t sg0 = (uint32_t)c0 >> 3;
    const uint32_t tile = B.dir[grid_dir_index(B, py, pz, sg0)];
    const uint2* e = B.tiles + grid_entry_index(B, tile, py, pz, sg0);
    const uint2 e0 = e[0], e1 = e[1];
    const bool next = ((uint32_t)c1 >> 3) != sg0;
    const uint32_t lo = seg_count(e0.x, e0.y, (uint32_t)c0 & 7u, B.ovf);
    const uint32_t hi = seg_count(next ? e1.x : e0.x, next ? e1.y : e0.y, (uint32_t)c1 & 7u, B.ovf);
    const uint32_t want = grid_pos(A, A.dir, y, z, c1) - grid_pos(A, A.dir, y, z, c0);
    if (hi - lo != want || (want != 0u && lo != grid_pos(B, B.dir, y, z, c0))) {
#ifdef FLIMO_SELFCHECK_DEBUG
      printf("index_compare: row (y %d, z %d) columns [%d, %d): the pass would read %u .. %u, the range holds %u points from %u\n", y, z, c0, c1, lo, hi, want,
             grid_pos(B, B.dir, y, z, c0));
#endif
      atomicAdd(diff, 1ull);
    }
  }
}
// ... and hold the same points, row by row (the rows of a maintained map are not packed behind each other)
__global__ __launch_bounds__(64) void rows_compare_kernel(GridView A, GridView B, unsigned long long* __restrict__ diff) {
  const int row = (int)blockIdx.x;
  const int y = row % A.ny, z = row / A.ny;
  const uint32_t a0 = grid_pos(A, A.dir, y, z, 0), a1 = grid_pos(A, A.dir, y, z, A.nxf);
  const uint32_t b0 = grid_pos(B, B.dir, y, z, 0), b1 = grid_pos(B, B.dir, y, z, B.nxf);
  if (a1 - a0 != b1 - b0) {
#ifdef FLIMO_SELFCHECK_DEBUG
    if (threadIdx.x == 0) printf("rows_compare: row (y %d, z %d): ref len %u live len %u\n", y, z, a1 - a0, b1 - b0);
#endif
    if (threadIdx.x == 0) atomicAdd(diff, 1ull);
    return;
  }
  unsigned bad = 0;
  for (uint32_t i = threadIdx.x; i < a1 - a0; i += 64) {
    const float4 p = A.pts[a0 + i], q = B.pts[b0 + i];
    const unsigned b_ = (__float_as_uint(p.x) != __float_as_uint(q.x)) + (__float_as_uint(p.y) != __float_as_uint(q.y)) +
           (__float_as_uint(p.z) != __float_as_uint(q.z)) + (__float_as_uint(p.w) != __float_as_uint(q.w));
#ifdef FLIMO_SELFCHECK_DEBUG
    if (b_) printf("rows_compare: row (y %d, z %d) len %u point %u: ref (%g %g %g w %u) live (%g %g %g w %u) a0 %u b0 %u\n", y, z, a1 - a0, i, p.x, p.y, p.z,
                   __float_as_uint(p.w), q.x, q.y, q.z, __float_as_uint(q.w), a0, b0);
#endif
    bad += b_;
  }
  if (bad) atomicAdd(diff, (unsigned long long)bad);
}
hipError_t index_compare(hipStream_t st, const GridView& A, const GridView& B, unsigned long long* diff_dev) {
  hipLaunchKernelGGL(rows_compare_kernel, dim3((unsigned)((size_t)A.ny * A.nz)), dim3(64), 0, st, A, B, diff_dev);
  const size_t n = (size_t)A.ny * A.nz * ((size_t)A.nxf + 1);
  hipLaunchKernelGGL(index_compare_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, A, B, diff_dev);
  return hipGetLastError();
}

// ---- spatial ordering of the scan ----------------------------------------------------------
__device__ __forceinline__ uint32_t spread10(uint32_t v) {   // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ __launch_bounds__(256) void mortonkey_kernel(const float4* __restrict__ pts, size_t n,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  // body-frame coordinates quantised to 0.5 m, +-256 m range (clamped); NaNs go to the end
  const float qx = fminf(fmaxf(p.x * 2.0f + 512.f, 0.f), 1023.f);
  const float qy = fminf(fmaxf(p.y * 2.0f + 512.f, 0.f), 1023.f);
  const float qz = fminf(fmaxf(p.z * 2.0f + 512.f, 0.f), 1023.f);
  const bool nan = !(p.x == p.x) || !(p.y == p.y) || !(p.z == p.z);
  const uint32_t k = spread10((uint32_t)qx) | (spread10((uint32_t)qy) << 1) | (spread10((uint32_t)qz) << 2);
  keys[i] = nan ? 0x3fffffffu : k;
  vals[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void gather_scan_kernel(const float4* __restrict__ in, const uint32_t* __restrict__ perm,
                                                          size_t n, float4* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t src = perm[i];
  float4 p = in[src];
  p.w = __uint_as_float(src);
  out[i] = p;
}

__global__ __launch_bounds__(256) void gather_f64_kernel(const double* __restrict__ in, const uint32_t* __restrict__ perm,
                                                         size_t n, double* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[perm[i]];
}

hipError_t sort_scan(hipStream_t st, const float4* in, size_t n, float4* out, MapBuildScratch& S, const double* t_in,
                     double* t_out) {
  if (n == 0) return hipSuccess;
  hipError_t e = ensure_scratch(S, n);
  if (e != hipSuccess) return e;
  const int blocks = (int)((n + 255) / 256);
  hipLaunchKernelGGL(mortonkey_kernel, dim3(blocks), dim3(256), 0, st, in, n, S.keys_in, S.vals_in);
  size_t tmp_bytes = 0;
  e = sort_pairs_u32(nullptr, tmp_bytes, S.keys_in, S.keys_out, S.vals_in, S.vals_out, (int)n, 0, 30, st);
  if (e != hipSuccess) return e;
  if (tmp_bytes > S.cub_tmp_bytes) {
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    if (S.cub_tmp) (void)hipFree(S.cub_tmp);
    if ((e = hipMalloc(&S.cub_tmp, tmp_bytes + 1024)) != hipSuccess) return e;
    S.cub_tmp_bytes = tmp_bytes + 1024;
  }
  e = sort_pairs_u32(S.cub_tmp, tmp_bytes, S.keys_in, S.keys_out, S.vals_in, S.vals_out, (int)n, 0, 30, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(gather_scan_kernel, dim3(blocks), dim3(256), 0, st, in, S.vals_out, n, out);
  if (t_in && t_out) hipLaunchKernelGGL(gather_f64_kernel, dim3(blocks), dim3(256), 0, st, t_in, S.vals_out, n, t_out);
  return hipGetLastError();
}

// The sweep left in the order it has (a voxel filter follows and re-orders the scan anyway: the Morton order of the raw sweep would
// serve nobody): out[i] = (xyz, w = i), the same layout sort_scan produces.
__global__ __launch_bounds__(256) void index_scan_kernel(const float4* __restrict__ in, size_t n, float4* __restrict__ out,
                                                         const double* __restrict__ t_in, double* __restrict__ t_out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = in[i];
  out[i] = make_float4(p.x, p.y, p.z, __uint_as_float((uint32_t)i));
  if (t_in) t_out[i] = t_in[i];
}
hipError_t index_scan(hipStream_t st, const float4* in, size_t n, float4* out, const double* t_in, double* t_out) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(index_scan_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, out, t_in, t_out);
  return hipGetLastError();
}

// ---- input filters of a raw sweep on the device (Localizer.cpp:262-302): NaN removal -> crop box -> every rate-th survivor ->
//      min distance, order preserved; per-point stamp (:741-805).  Records are the reference's 32-byte PointType. -----------------
struct Raw32 { float x, y, z, w, intensity; uint32_t pad; uint32_t u0, u1; };   // u0/u1: the 8-byte time union
__device__ __forceinline__ bool filt_alive(const Raw32& p, const FilterParams& F) {
  const bool finite = isfinite(p.x) & isfinite(p.y) & isfinite(p.z);
  const bool outside = (p.x < F.mn[0]) | (p.y < F.mn[1]) | (p.z < F.mn[2]) | (p.x > F.mx[0]) | (p.y > F.mx[1]) | (p.z > F.mx[2]);
  return finite & (!F.crop | outside);
}
// The filters of a sweep in ONE launch.  A workgroup takes a tile of FILT_TILE consecutive points (tiles are handed out in order by
// a ticket, so a tile's predecessors are always running or done); a point's rank among the survivors of NaN removal + crop box (the
// rate filter's "every rate-th") and its output position are the tile's own counts (wave ballots; eight coalesced rows of 64 points
// per wave) plus the sums over all earlier tiles, which every tile publishes the moment it knows its own count -- {count, launch
// number} in one 64-bit word -- and reads from all its predecessors at once, 64 per round trip (two chained waits per tile, a few
// microseconds for the whole sweep instead of eight dispatches).
// out[pos] = (xyz, w = pos), t_out[pos] = the point's stamp without the sweep offset, ext[0] = extreme ordered key (max; min when
// sorting descending, stored complemented), ext[1] = kept count, ext[2] = a kept stamp is NaN
constexpr int FILT_ROWS = 8;
constexpr int FILT_TILE = 256 * FILT_ROWS;
__device__ __forceinline__ unsigned int filt_lookback(const unsigned long long* __restrict__ desc, int tile, unsigned int epoch, unsigned int* s_sum) {
  // exclusive sum of the earlier tiles' counts (first wave; the block picks it up from *s_sum)
  if (threadIdx.x < 64) {
    unsigned int sum = 0;
    for (int base = 0; base < tile; base += 64) {
      const int t = base + (int)threadIdx.x;
      unsigned int c = 0;
      if (t < tile) {
        unsigned long long d;
        do { d = __hip_atomic_load(&desc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while ((unsigned int)(d >> 32) != epoch);
        c = (unsigned int)d;
      }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) c += __shfl_xor(c, o, 64);
      sum += c;
    }
    if (threadIdx.x == 0) *s_sum = sum;
  }
  __syncthreads();
  return *s_sum;
}
// REC = 32: the reference's PointType records; REC = 16: {x, y, z, the 32-bit time word} (OUSTER / VELODYNE stamps), what a caller
// that stages the upload itself may pack the sweep into -- half the bytes over PCIe
template <int REC>
__global__ __launch_bounds__(256) void filt_onepass_kernel(const void* __restrict__ in_, size_t n, FilterParams F, unsigned int* __restrict__ ticket,
                                                           unsigned int ticket_base, unsigned long long* __restrict__ desc_alive,
                                                           unsigned long long* __restrict__ desc_kept, unsigned int epoch,
                                                           float4* __restrict__ out, double* __restrict__ t_out,
                                                           unsigned long long* __restrict__ ext, unsigned long long* __restrict__ key_out,
                                                           unsigned int* __restrict__ done, unsigned int done_base,
                                                           unsigned long long* __restrict__ mail) {
  __shared__ unsigned int s_tile, s_sum[2], s_cnt[2][4];
  __shared__ unsigned long long s_max[4];
  __shared__ int s_nan[4];
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
  __syncthreads();
  const int tile = (int)s_tile;
  if (tile < 0 || tile >= (int)gridDim.x) return;             // (a ticket out of step with the host's count -- an earlier launch that never ran: no tile of this launch; the caller sees a count that was never written)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const size_t base = (size_t)tile * FILT_TILE + (size_t)wave * (64 * FILT_ROWS);
  // ---- the tile's points (row r of this wave: points base + 64 r + lane), NaN removal + crop box ----
  float px[FILT_ROWS], py[FILT_ROWS], pz[FILT_ROWS];
  uint32_t u0[FILT_ROWS], u1[FILT_ROWS];
  unsigned long long alive[FILT_ROWS];
  unsigned int n_alive = 0;
#pragma unroll
  for (int r = 0; r < FILT_ROWS; r++) {
    const size_t i = base + (size_t)r * 64 + lane;
    Raw32 p;
    p.x = p.y = p.z = 0.f; p.u0 = p.u1 = 0u;
    const bool inb = i < n;
    if (inb) {
      if constexpr (REC == 32) {
        const Raw32* in = static_cast<const Raw32*>(in_);
        const float4 a = reinterpret_cast<const float4*>(in + i)[0];
        const uint4 b = reinterpret_cast<const uint4*>(in + i)[1];
        p.x = a.x; p.y = a.y; p.z = a.z; p.u0 = b.z; p.u1 = b.w;
      } else {
        const float4 a = static_cast<const float4*>(in_)[i];
        p.x = a.x; p.y = a.y; p.z = a.z; p.u0 = __float_as_uint(a.w);
      }
    }
    px[r] = p.x; py[r] = p.y; pz[r] = p.z; u0[r] = p.u0; u1[r] = p.u1;
    alive[r] = __ballot(inb && filt_alive(p, F));
    n_alive += (unsigned int)__popcll(alive[r]);
  }
  if (lane == 0) s_cnt[0][wave] = n_alive;
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(&desc_alive[tile], ((unsigned long long)epoch << 32) | (unsigned long long)(s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3]),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- rank among the survivors -> every rate-th, then FoV and min distance ----
  unsigned int rank0 = 0;
  if (F.rate_on) {
    rank0 = filt_lookback(desc_alive, tile, epoch, &s_sum[0]);
    for (int w = 0; w < wave; w++) rank0 += s_cnt[0][w];
  }
  unsigned long long keep[FILT_ROWS];
  unsigned int n_keep = 0;
#pragma unroll
  for (int r = 0; r < FILT_ROWS; r++) {
    bool k = (alive[r] >> lane) & 1ull;
    if (k && F.rate_on) k = ((rank0 + (unsigned int)__popcll(alive[r] & lt)) % (uint32_t)F.rate) == 0u;
    if (k && F.fov) k = __builtin_fabsf(libm_atan2f(py[r], px[r])) < F.fov_angle;       // std::atan2 of two floats, as the host's libm rounds it
    if (k && F.dist) k = __builtin_sqrtf(px[r] * px[r] + (py[r] * py[r] + pz[r] * pz[r])) > F.min_dist;
    keep[r] = __ballot(k);
    n_keep += (unsigned int)__popcll(keep[r]);
    rank0 += (unsigned int)__popcll(alive[r]);
  }
  if (lane == 0) s_cnt[1][wave] = n_keep;
  __syncthreads();
  const unsigned int tile_keep = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
  if (threadIdx.x == 0)
    __hip_atomic_store(&desc_kept[tile], ((unsigned long long)epoch << 32) | (unsigned long long)tile_keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned int pos0 = filt_lookback(desc_kept, tile, epoch, &s_sum[1]);
  if ((size_t)(tile + 1) * FILT_TILE >= n && threadIdx.x == 0)                                                        // the last tile: kept count
    __hip_atomic_store(&ext[1], (unsigned long long)(pos0 + tile_keep), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int w = 0; w < wave; w++) pos0 += s_cnt[1][w];
  // ---- compaction: position, stamp, ordered key ----
  unsigned long long mine = 0ull;                              // extreme ordered key of this lane's points (0: none)
  bool any_nan = false;
  const bool desc = F.eos && F.kind <= 1;
#pragma unroll
  for (int r = 0; r < FILT_ROWS; r++) {
    if ((keep[r] >> lane) & 1ull) {
      const uint32_t o = pos0 + (unsigned int)__popcll(keep[r] & lt);
      out[o] = make_float4(px[r], py[r], pz[r], __uint_as_float(o));
      double t;
      unsigned long long key;
      bool nan = false;
      if (F.kind == 0) {
        const float tf = (float)u0[r] * 1e-9f;
        t = F.eos ? F.sweep_ref - (double)tf : F.sweep_ref + (double)tf;
        key = (unsigned long long)u0[r];
      } else if (F.kind == 1) {
        float v = __uint_as_float(u0[r]);
        t = F.eos ? F.sweep_ref - (double)v : F.sweep_ref + (double)v;
        nan = v != v;
        v += 0.0f;
        const uint32_t b = __float_as_uint(v);
        key = (unsigned long long)((b & 0x80000000u) ? ~b : (b | 0x80000000u));
      } else {
        double v = __longlong_as_double((long long)(((unsigned long long)u1[r] << 32) | (unsigned long long)u0[r]));
        t = (F.kind == 2) ? v : v * (double)1e-9f;
        nan = v != v;
        v += 0.0;
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        key = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
      }
      t_out[o] = t;
      const unsigned long long ok = desc ? ~key : key;        // ascending in this key = the order of the reference's time sort
      if (key_out) key_out[o] = ok;
      if (!nan) mine = ok > mine ? ok : mine;
      any_nan |= nan;
    }
    pos0 += (unsigned int)__popcll(keep[r]);
  }
  // block maximum of the keys, block OR of the NaN marks: one atomic per block
  unsigned long long m = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned long long v = __shfl_xor(m, o, 64); m = v > m ? v : m; }
  const int wave_nan = __any(any_nan) ? 1 : 0;
  if (lane == 0) { s_max[wave] = m; s_nan[wave] = wave_nan; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long bm = s_max[0];
    for (int w = 1; w < 4; w++) bm = s_max[w] > bm ? s_max[w] : bm;
    if (bm) atomicMax(&ext[0], bm);
    if (s_nan[0] | s_nan[1] | s_nan[2] | s_nan[3]) atomicOr(&ext[2], 1ull);
    // The tile that finishes LAST hands the three results to the host: 16-byte {value, launch number} granules in mapped memory
    // (data and "ready" in one store -- the host spins on the tags instead of paying a copy and a stream wait).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this tile's contributions are performed
    const unsigned int arrived = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - done_base;
    if (arrived == gridDim.x - 1u) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        typedef unsigned long long v2u_t __attribute__((ext_vector_type(2)));
        v2u_t g;
        g.x = __hip_atomic_load(&ext[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g.y = (unsigned long long)epoch;
        unsigned long long* o = mail + 2 * k;
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(o), "v"(g) : "memory");
      }
    }
  }
}

hipError_t filter_raw_scan(hipStream_t st, const void* raw32_dev, size_t n, const FilterParams& F, float4* out, double* t_out,
                           unsigned long long* ext_dev, MapBuildScratch& S, unsigned long long* key_out, int rec_bytes) {
  if (n == 0) return hipSuccess;
  const size_t tiles = (n + FILT_TILE - 1) / FILT_TILE;
  hipError_t e;
  if (tiles > S.filt_tiles_cap) {
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    if (S.filt_desc) (void)hipFree(S.filt_desc);
    S.filt_desc = nullptr;
    const size_t cap = tiles + tiles / 2 + 64;
    if ((e = hipMalloc(&S.filt_desc, (2 * cap + 1) * sizeof(unsigned long long))) != hipSuccess) return e;
    if ((e = hipMemsetAsync(S.filt_desc, 0, (2 * cap + 1) * sizeof(unsigned long long), st)) != hipSuccess) return e;   // launch number 0: never valid
    S.filt_tiles_cap = cap;
    S.filt_epoch = 0;
    S.filt_ticket_base = 0;
  }
  if (!S.filt_mail_host) {
    if ((e = hipHostMalloc((void**)&S.filt_mail_host, 8 * sizeof(unsigned long long), hipHostMallocMapped)) != hipSuccess) return e;
    memset(S.filt_mail_host, 0, 8 * sizeof(unsigned long long));
    if ((e = hipHostGetDevicePointer((void**)&S.filt_mail_dev, S.filt_mail_host, 0)) != hipSuccess) return e;
  }
  if ((e = hipMemsetAsync(ext_dev, 0, 4 * sizeof(unsigned long long), st)) != hipSuccess) return e;
  unsigned long long* desc_alive = S.filt_desc;
  unsigned long long* desc_kept = S.filt_desc + S.filt_tiles_cap;
  unsigned int* ticket = reinterpret_cast<unsigned int*>(S.filt_desc + 2 * S.filt_tiles_cap);
  unsigned int* done = ticket + 1;                              // (the same 8-byte word: tiles started / tiles finished, both counted up for ever)
  S.filt_epoch++;
  if (S.filt_epoch == 0u) S.filt_epoch = 1u;                   // (after 2^32 launches the words are 2^32 launches stale: never equal)
  if (rec_bytes == 16)
    hipLaunchKernelGGL(filt_onepass_kernel<16>, dim3((unsigned)tiles), dim3(256), 0, st, raw32_dev, n, F, ticket,
                       S.filt_ticket_base, desc_alive, desc_kept, S.filt_epoch, out, t_out, ext_dev, key_out, done, S.filt_ticket_base, S.filt_mail_dev);
  else
    hipLaunchKernelGGL(filt_onepass_kernel<32>, dim3((unsigned)tiles), dim3(256), 0, st, raw32_dev, n, F, ticket,
                       S.filt_ticket_base, desc_alive, desc_kept, S.filt_epoch, out, t_out, ext_dev, key_out, done, S.filt_ticket_base, S.filt_mail_dev);
  S.filt_ticket_base += (unsigned int)tiles;                   // (wraps with the counter)
  e = hipGetLastError();
  if (e != hipSuccess) S.filt_tiles_cap = 0;                   // the launch did not happen: ticket and count are out of step -- start over next time
  return e;
}

// The three results of the last filter_raw_scan on the host: spins on the granules' tags (the kernel's last tile stores them to
// mapped memory); a wait that runs out falls back to a copy behind the stream.  ext3: extreme key, kept count, NaN mark.
hipError_t filter_raw_scan_result(hipStream_t st, MapBuildScratch& S, const unsigned long long* ext_dev, int timeout_ms, unsigned long long ext3[3]) {
  const volatile unsigned long long* m = S.filt_mail_host;
  const unsigned long long tag = (unsigned long long)S.filt_epoch;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned long long spins = 0;; spins++) {
    if (m[1] == tag && m[3] == tag && m[5] == tag) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      ext3[0] = m[0]; ext3[1] = m[2]; ext3[2] = m[4];
      return hipSuccess;
    }
    _mm_pause();
    if ((spins & 0xfffull) == 0xfffull &&
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > (double)(timeout_ms > 0 ? timeout_ms : 2000)) break;
  }
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  return hipMemcpy(ext3, ext_dev, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}

// ---- the reference's time order of a sweep on the device (Localizer.cpp:789-790: std::partial_sort_copy of the whole cloud by
//      stamp).  With pairwise different stamps the sorted order is unique, whatever algorithm produces it: a stable radix sort of
//      (ordered stamp key, position) gives it; ext[3] reports whether two kept stamps are equal -- then the order among them is
//      the library's heap moves' and only the host routine reproduces it (fast_limo.cpp: time_order). -----------------------------
__global__ __launch_bounds__(256) void iota_kernel(uint32_t* __restrict__ v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void tied_keys_kernel(const unsigned long long* __restrict__ sorted, size_t n, unsigned long long* __restrict__ ext) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool tie = (i + 1 < n) && sorted[i] == sorted[i + 1];
  if (__ballot(tie) && (threadIdx.x & 63) == 0) atomicOr(&ext[3], 1ull);
}
__global__ __launch_bounds__(256) void gather_time_order_kernel(const float4* __restrict__ in, const double* __restrict__ t_in,
                                                                const uint32_t* __restrict__ perm, size_t n, float4* __restrict__ out,
                                                                double* __restrict__ t_out, const unsigned long long* __restrict__ ext,
                                                                unsigned long long* __restrict__ mail, unsigned long long tag) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    // "two kept stamps are equal" (tied_keys_kernel, the launch before this one) goes to the host as a {value, tag} granule
    typedef unsigned long long v2u_t __attribute__((ext_vector_type(2)));
    v2u_t g;
    g.x = ext[3];
    g.y = tag;
    unsigned long long* o = mail + 6;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(o), "v"(g) : "memory");
  }
  if (i >= n) return;
  const uint32_t j = perm[i];
  const float4 p = in[j];
  out[i] = make_float4(p.x, p.y, p.z, __uint_as_float((uint32_t)i));
  t_out[i] = t_in[j];
}
hipError_t time_order_raw(hipStream_t st, const float4* pts, const double* t, size_t n, const unsigned long long* keys,
                          unsigned long long* keys_sorted, float4* pts_out, double* t_out, uint32_t* perm_out,
                          unsigned long long* ext_dev, MapBuildScratch& S) {
  if (n == 0) return hipSuccess;
  hipError_t e = ensure_scratch(S, n);
  if (e != hipSuccess) return e;
  const int blocks = (int)((n + 255) / 256);
  hipLaunchKernelGGL(iota_kernel, dim3(blocks), dim3(256), 0, st, S.vals_in, n);
  size_t tmp_bytes = 0;
  e = sort_pairs_u64(nullptr, tmp_bytes, keys, keys_sorted, S.vals_in, perm_out, n, st);
  if (e != hipSuccess) return e;
  if (tmp_bytes > S.cub_tmp_bytes) {
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    if (S.cub_tmp) (void)hipFree(S.cub_tmp);
    if ((e = hipMalloc(&S.cub_tmp, tmp_bytes + 1024)) != hipSuccess) return e;
    S.cub_tmp_bytes = tmp_bytes + 1024;
  }
  e = sort_pairs_u64(S.cub_tmp, tmp_bytes, keys, keys_sorted, S.vals_in, perm_out, n, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(tied_keys_kernel, dim3(blocks), dim3(256), 0, st, keys_sorted, n, ext_dev);
  hipLaunchKernelGGL(gather_time_order_kernel, dim3(blocks), dim3(256), 0, st, pts, t, perm_out, n, pts_out, t_out, ext_dev, S.filt_mail_dev,
                     (unsigned long long)S.filt_epoch);
  return hipGetLastError();
}

// "two kept stamps are equal" of the last time_order_raw (same way home as filter_raw_scan_result)
hipError_t time_order_raw_tied(hipStream_t st, MapBuildScratch& S, const unsigned long long* ext_dev, int timeout_ms, bool* tied) {
  const volatile unsigned long long* m = S.filt_mail_host;
  const unsigned long long tag = (unsigned long long)S.filt_epoch;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned long long spins = 0;; spins++) {
    if (m[7] == tag) { __atomic_thread_fence(__ATOMIC_ACQUIRE); *tied = m[6] != 0ull; return hipSuccess; }
    _mm_pause();
    if ((spins & 0xfffull) == 0xfffull &&
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > (double)(timeout_ms > 0 ? timeout_ms : 2000)) break;
  }
  hipError_t e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  unsigned long long v = 0;
  e = hipMemcpy(&v, ext_dev + 3, sizeof(v), hipMemcpyDeviceToHost);
  *tied = v != 0ull;
  return e;
}

// ---- voxel-grid down-sampling of the scan (pcl::VoxelGrid, reference Localizer.cpp:313-321) -----
// One output point per occupied voxel = centroid of its points (float sums in ascending point order),
// output in ascending linear voxel index  i + j*div_x + k*div_x*div_y  with
// i = floor(x * inv_leaf) - floor(min_x * inv_leaf)  (PCL 1.10 filters/impl/voxel_grid.hpp).
__global__ __launch_bounds__(256) void bbox_finite_kernel(const float4* __restrict__ pts, size_t n, unsigned* __restrict__ box) {
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 p = pts[i];
    if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) continue;
    mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
    mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int a = 0; a < 3; a++) {
      mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
      mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
    }
  // one set of atomics per block (same-address atomics cost ~70 ns each on this part)
  __shared__ float s_mn[4][3], s_mx[4][3];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int a = 0; a < 3; a++) { s_mn[wave][a] = mn[a]; s_mx[wave][a] = mx[a]; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    const float lo = fminf(fminf(s_mn[0][a], s_mn[1][a]), fminf(s_mn[2][a], s_mn[3][a]));
    const float hi = fmaxf(fmaxf(s_mx[0][a], s_mx[1][a]), fmaxf(s_mx[2][a], s_mx[3][a]));
    atomicMin(&box[a], f2o(lo));
    atomicMax(&box[3 + a], f2o(hi));
  }
}

// The lattice comes from the box the reduction before this launch left in device memory (ordered-uint words: no host round trip
// for it): min_b = floor(min * inv), div_b = floor(max * inv) - min_b + 1, the host's expressions.  flags[0] = 1: no finite point;
// flags[1] = 1: the lattice would overflow an int (PCL then returns its input) -- keys are not valid then, the host looks at the
// flags when it reads the count.
__global__ __launch_bounds__(256) void voxelkey_kernel(const float4* __restrict__ pts, size_t n, float inv, const unsigned* __restrict__ box,
                                                       uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ flags) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned b0 = box[0];
  int mb0 = 0, mb1 = 0, mb2 = 0, mul1 = 1, mul2 = 1;
  bool bad = b0 == 0xffffffffu;
  if (!bad) {
    mb0 = (int)floorf(o2f(box[0]) * inv); mb1 = (int)floorf(o2f(box[1]) * inv); mb2 = (int)floorf(o2f(box[2]) * inv);
    const int d0 = (int)floorf(o2f(box[3]) * inv) - mb0 + 1, d1 = (int)floorf(o2f(box[4]) * inv) - mb1 + 1, d2 = (int)floorf(o2f(box[5]) * inv) - mb2 + 1;
    const long long cells = (long long)d0 * d1 * d2;
    if (cells > 2147483647ll) { bad = true; if (i == 0) flags[1] = 1u; }
    mul1 = d0; mul2 = d0 * d1;
  } else if (i == 0) {
    flags[0] = 1u;
  }
  if (i >= n) return;
  const float4 p = pts[i];
  uint32_t k = 0xffffffffu;                                 // non-finite points sort to the end and are skipped
  if (!bad && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
    const int i0 = (int)(floorf(p.x * inv) - (float)mb0);
    const int i1 = (int)(floorf(p.y * inv) - (float)mb1);
    const int i2 = (int)(floorf(p.z * inv) - (float)mb2);
    k = (uint32_t)(i0 + i1 * mul1 + i2 * mul2);
  }
  keys[i] = k;
  vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void voxelhead_kernel(const uint32_t* __restrict__ keys, size_t n, uint32_t* __restrict__ head) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = keys[i];
  head[i] = (k != 0xffffffffu && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void voxelcentroid_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ keys,
                                                            const uint32_t* __restrict__ perm, const uint32_t* __restrict__ head,
                                                            const uint32_t* __restrict__ pos, size_t n, float4* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  const uint32_t k = keys[i];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  uint32_t cnt = 0;
  // stable sort: ascending original index.  Eight keys / positions / points per round trip (a voxel of the reference's shipped
  // configuration holds about sixteen); the sums keep their order
  for (size_t j = i; j < n; j += 8) {
    uint32_t kk[8], pp[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const size_t q = j + u < n ? j + u : n - 1; kk[u] = j + u < n ? keys[q] : ~k; pp[u] = perm[q]; }
    float4 pt[8];
#pragma unroll
    for (int u = 0; u < 8; u++) pt[u] = pts[pp[u]];
    bool more = true;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      more = more && kk[u] == k;
      if (more) { sx += pt[u].x; sy += pt[u].y; sz += pt[u].z; cnt++; }
    }
    if (!more) break;
  }
  const float nf = (float)cnt;
  out[pos[i]] = make_float4(sx / nf, sy / nf, sz / nf, 1.0f);
}

hipError_t batch_bbox(hipStream_t st, const float4* pts, size_t n, MapBuildScratch& S, float bb[6], bool* any) {
  *any = false;
  if (n == 0) return hipSuccess;
  hipError_t e = ensure_scratch(S, 0);
  if (e != hipSuccess) return e;
  const int blocks = (int)std::min<size_t>((n + 1023) / 1024, 128);
  hipLaunchKernelGGL(bbox_finite_kernel, dim3(blocks), dim3(256), 0, st, pts, n, (unsigned*)S.bbox);
  unsigned ob[6];
  if ((e = fetch_bbox(st, S, ob)) != hipSuccess) return e;
  if (ob[0] == 0xffffffffu) return hipSuccess;
  for (int i = 0; i < 6; i++) bb[i] = o2f_host(ob[i]);
  *any = true;
  return hipSuccess;
}

// in -> out (may not alias).  *n_out receives the voxel count; returns hipErrorInvalidValue through
// *passthrough = true when the voxel lattice would overflow an int (PCL then returns the input).
hipError_t voxel_grid(hipStream_t st, const float4* in, size_t n, float leaf, float4* out, size_t* n_out, bool* passthrough,
                      MapBuildScratch& S) {
  *n_out = 0;
  *passthrough = false;
  if (n == 0) return hipSuccess;
  hipError_t e = ensure_scratch(S, n);
  if (e != hipSuccess) return e;
  const int blocks = (int)((n + 255) / 256);
  hipLaunchKernelGGL(bbox_finite_kernel, dim3(std::max(1, std::min(blocks / 4, 128))), dim3(256), 0, st, in, n, (unsigned*)S.bbox);
  const float inv = 1.0f / leaf;
  // (the box stays on the device: the key kernel derives the lattice from it; two flag words ride home with the count)
  uint32_t* flags = S.mail_dev + MAIL_VOXEL + 2;
  S.mail_host[MAIL_VOXEL + 2] = S.mail_host[MAIL_VOXEL + 3] = 0u;      // (mapped memory; nothing in flight writes these words)
  __atomic_thread_fence(__ATOMIC_RELEASE);
  hipLaunchKernelGGL(voxelkey_kernel, dim3(blocks), dim3(256), 0, st, in, n, inv, (const unsigned*)S.bbox, S.keys_in, S.vals_in, flags);
  size_t tmp_bytes = 0;
  e = sort_pairs_u32(nullptr, tmp_bytes, S.keys_in, S.keys_out, S.vals_in, S.vals_out, (int)n, 0, 32, st);
  if (e != hipSuccess) return e;
  size_t scan_bytes = 0;
  e = exclusive_sum(nullptr, scan_bytes, S.keys_in, S.vals_in, (int)n, st);
  if (e != hipSuccess) return e;
  const size_t need = std::max(tmp_bytes, scan_bytes);
  if (need > S.cub_tmp_bytes) {
    if (S.cub_tmp) (void)hipFree(S.cub_tmp);
    if ((e = hipMalloc(&S.cub_tmp, need + 1024)) != hipSuccess) return e;
    S.cub_tmp_bytes = need + 1024;
  }
  e = sort_pairs_u32(S.cub_tmp, tmp_bytes, S.keys_in, S.keys_out, S.vals_in, S.vals_out, (int)n, 0, 32, st);
  if (e != hipSuccess) return e;
  // keys_in := head flags, vals_in := exclusive scan of the flags (output slot of each run)
  hipLaunchKernelGGL(voxelhead_kernel, dim3(blocks), dim3(256), 0, st, S.keys_out, n, S.keys_in);
  e = exclusive_sum(S.cub_tmp, scan_bytes, S.keys_in, S.vals_in, (int)n, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(voxelcentroid_kernel, dim3(blocks), dim3(256), 0, st, in, S.keys_out, S.vals_out, S.keys_in, S.vals_in, n, out);
  const MailPart parts[2] = {{S.vals_in + (n - 1), 1, MAIL_VOXEL}, {S.keys_in + (n - 1), 1, MAIL_VOXEL + 1}};
  if ((e = mail_words(st, S, parts, 2, true)) != hipSuccess) return e;              // (+ the box re-armed for the next reduction)
  if ((e = mail_wait(st, S)) != hipSuccess) {
    const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    (void)hipMemcpy(S.bbox, init, sizeof(init), hipMemcpyHostToDevice);
    return e;
  }
  if (S.mail_host[MAIL_VOXEL + 2]) return hipSuccess;        // no finite point
  if (S.mail_host[MAIL_VOXEL + 3]) { *passthrough = true; return hipSuccess; }
  *n_out = (size_t)S.mail_host[MAIL_VOXEL] + S.mail_host[MAIL_VOXEL + 1];
  return hipGetLastError();
}

void map_scratch_free(MapBuildScratch& S) {
  if (S.cub_tmp) hipFree(S.cub_tmp);
  if (S.keys_in) { hipFree(S.keys_in); hipFree(S.keys_out); hipFree(S.vals_in); hipFree(S.vals_out); }
  if (S.ck_in) { (void)hipFree(S.ck_in); (void)hipFree(S.ck_out); S.ck_in = S.ck_out = nullptr; S.ck_cap = 0; }
  if (S.bbox) hipFree(S.bbox);
  if (S.filt_desc) hipFree(S.filt_desc);
  if (S.filt_mail_host) hipHostFree(S.filt_mail_host);
  if (S.mail_host) hipHostFree(S.mail_host);
  S = MapBuildScratch();
}

}  // namespace flimo
