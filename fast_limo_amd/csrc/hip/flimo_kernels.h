// fast_limo_amd/csrc/hip/flimo_kernels.h -- host-callable launchers of the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "flimo_types.h"

namespace flimo {
struct FuseArgs;
struct TieList;
struct BookView;
struct ChainCtl;    // flimo_chain.h: the filter's algebra inside the pass's reducing launch
struct ChainHead;   // flimo_chain.h: a launch given one reads its pose constants from the device filter and leaves at once when the chain has ended
// The deskew of a scan's raw points riding on the first pass's k-NN launch (instead of a dispatch of its own): arguments of
// launch_deskew, `on` = 1 when they are valid
struct DeskewArgs {
  const float4* raw; const double* t; const void* frames; int nf; const float* mats; double t_offset;
  float4* out_sorted; float4* out_orig; int on;
  int stage_words;     // > 0: frames + matrices (this many 4-byte words from `frames`) were stored by the HOST into fine-grained device
                       // memory -- no copy launch; the riding pass reads them once per workgroup, past the caches, into shared memory
};

// flimo_kernels.hip
// per pass: k-NN (fast path + worklist widening), then fit + in-block reduction, then the final sum
void launch_knn5(hipStream_t st, int lanes_per_query, const GridView& G, const float4* scan_sorted, int n,
                 const PoseMats& P, int max_ring, void* nbr, int* wl, int* wl_count, unsigned long long* cand,
                 const PrevPass& prev, int tail, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, const struct FuseArgs* fuse = nullptr,
                 const TieList* ties = nullptr, int after_fine = 0, unsigned long long seq = 0ull, const DeskewArgs* deskew = nullptr,
                 const ChainHead* chain = nullptr, unsigned int wait_epoch = 0u, unsigned int end_code = 0u);
// (wait_epoch != 0: the launch's workgroups wait for the resident algebra to publish the chain's head under that number -- the FIRST
//  launch of a chained pass; end_code: the value that says the chain has ended; flimo_chain.h)
// fine pre-pass over the second-level grid of crowded regions (see flimo_map.hip); launches that follow it pass after_fine = 1
void launch_knn5_fine(hipStream_t st, const GridView& Gf, const float4* scan_sorted, int n, const PoseMats& P, void* nbr,
                      const PrevPass& prev, const int qlo[3], const int qhi[3], const TieList* ties, unsigned long long seq,
                      const ChainHead* chain = nullptr, unsigned int wait_epoch = 0u, unsigned int end_code = 0u);
// tail != 0: queries that need more than the 3x3x3 block are finished inside the k-NN launch itself (gates of at most 3
// rings; launch_knn5 clears the flag otherwise) and launch_widen has nothing to do
void launch_widen(hipStream_t st, const GridView& G, const float4* scan_sorted, const PoseMats& P, int max_ring, void* nbr,
                  int* wl, int* wl_count, unsigned long long* cand, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr,
                  const TieList* ties = nullptr, const ChainHead* chain = nullptr);
int fit_blocks(int n);
void set_xcd_stripe(int stripe);   // block -> scan chunk mapping of the per-pass kernels (see xcd_chunk)
// the fit kernel reduces in FIT_GROUPS groups; slot g of its output = 256 sums + the pass number (FIT_SLOT doubles).
// `ticket` arrays hold FIT_GROUPS + 1 counters: one per group and one launch-wide (fit_reduce_publish)
constexpr int FIT_GROUPS = 8;
constexpr int FIT_SLOT = 264;
// fit + in-block MFMA reduction + grid reduction by the last block: out256[0..255] receives the raw
// 16x16 accumulator and out256[256] (as u64) the pass number `seq` (system-scope release), `ticket` and `wl_count` are reset for the next pass
void launch_fit(hipStream_t st, const GridView& G, const float4* scan_sorted, int n, const void* nbr, const PoseMats& P,
                const MatchParams& mp, double* partials, Rec16* recs, RecDbg* dbg, double* out256, unsigned int* ticket,
                int* wl_count, unsigned long long seq);
// fit2: the per-pass fast path (no records, no caps).  Only the FIT_LIVE sums the filter reads leave a block (upper triangle
// of H^T H, H^T h, M; live_idx[k] = index of sum k among the wave's 256 raw MFMA accumulators); the last block of each
// group publishes FIT_LIVE_PAD 16-byte granules {sum, pass number as bits} to out_granules[group] (mapped host memory).
constexpr int FIT_LIVE = 91;
constexpr int FIT_LIVE_PAD = 96;
int fit2_blocks(int n);
void launch_fit2(hipStream_t st, const GridView& G, const float4* scan_sorted, int n, const void* nbr, const PoseMats& P,
                 const MatchParams& mp, const unsigned char* live_idx, double* partials, void* out_granules, unsigned int* ticket,
                 int* wl_count, unsigned long long seq, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, const TieList* ties = nullptr,
                 const ChainHead* chain = nullptr, const ChainCtl* ctl = nullptr, const BookView* book = nullptr);
// The whole measurement pass in ONE launch (k-NN fast path + in-kernel tail + fit + reduction + publish; two lanes per query, gates of
// 2..3 rings): partials needs fused_blocks(n) * FIT_LIVE_PAD doubles; results arrive as launch_fit2's granules.
constexpr int FUSED_SPREAD_SLOTS = 32768;     // scans below this many query slots are spread over more workgroups
int fused_spread(int n);
int fused_blocks(int n);
void launch_match_fused(hipStream_t st, const GridView& G, const float4* scan_sorted, int n, const PoseMats& P, const MatchParams& mp,
                        void* nbr, int* wl, int* wl_count, unsigned long long* cand, const PrevPass& prev,
                        const unsigned char* live_idx, double* partials, void* out_granules, unsigned int* ticket,
                        unsigned long long seq, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, const TieList* ties = nullptr,
                        int after_fine = 0, const DeskewArgs* deskew = nullptr, const ChainHead* chain = nullptr,
                        const ChainCtl* ctl = nullptr, const BookView* book = nullptr, unsigned int wait_epoch = 0u);
// The reference's choice among exactly tied distances (first met by Octree::knn's recursion): BookView = the device copy of the
// octree (insert book), TieList = the queries a pass flagged.  launch_tie rewrites their neighbour records; launch_knn_tie does the
// same for the output of launch_knn.
struct BookView { const float4* node_c; const int* node_child; const int* node_cnt; int root; unsigned long long* settled; };   // settled (optional): queries whose ties were settled inside a reducing launch
struct TieList { int* list; unsigned int* count; unsigned int cap; unsigned int* count_next; };
void launch_tie(hipStream_t st, const GridView& G, const BookView& B, const float4* scan_sorted, const PoseMats& P, void* nbr,
                const TieList& tl);
void launch_knn_tie(hipStream_t st, const GridView& G, const BookView& B, const float* qxyz, int nq, int k, int32_t* idx, float* sqd,
                    const int32_t* cnt);
// General NUM_MATCH_POINTS (3..8): exact k-NN by the ring search + M x 3 plane fit, records written at the original indices
// (reduce them with launch_cap / launch_reduce).  nbrk: n * nbrk_rec_size() bytes of scratch.  false: k out of range.
size_t nbrk_rec_size();
bool launch_match_k(hipStream_t st, int k, const GridView& G, const float4* scan_sorted, int n, const PoseMats& P,
                    const MatchParams& mp, void* nbrk, Rec16* recs, RecDbg* dbg, const BookView* book = nullptr);   // book: ties the reference's way
size_t nbr_rec_size();
size_t wl_entry_size();   // bytes per worklist entry (query index, world position, 5th-distance hint)
void launch_knn(hipStream_t st, const GridView& G, const float* qxyz, int nq, int k, int max_ring, int32_t* idx,
                float* sqd, int32_t* cnt);
void launch_cap(hipStream_t st, Rec16* recs, int n, int cap);
// MAX_NUM_MATCHES path in one launch: rank valid records in scan order, reduce the first `cap`, publish to slot 0 + pass number
void launch_capreduce(hipStream_t st, const Rec16* recs, int n, int cap, double* out256, int* wl_count, unsigned long long seq);
void launch_reduce(hipStream_t st, const Rec16* recs, int n, int nwaves, double* partials, double* out256);
void launch_mfma_layout(hipStream_t st, double* raw256);
void launch_word_probe(hipStream_t st, const unsigned int* src, void* out_granule, unsigned long long tag);   // the 32-bit word at src (read past the caches) as a granule
void launch_rtt_probe(hipStream_t st, void* out_granule, unsigned long long tag);   // one 16-byte {1.0, tag} granule to mapped host memory
// in/t are in the (Morton-)sorted order with the original index in in[k].w; writes the deskewed point
// to out_sorted[k] (w = original index) and to out_orig[original index]
void launch_deskew(hipStream_t st, const float4* in, const double* t, int n, const void* frames, int nf,
                   const float* mats32, float4* out_sorted, float4* out_orig, double t_offset = 0.0);   // stamp of point k = t[k] + t_offset
void launch_transform(hipStream_t st, const float4* in, int n, const PoseMats& P, float4* out);
size_t dev_frame_size();

// flimo_map.hip
struct MapBuildScratch {
  void* cub_tmp = nullptr;
  size_t cub_tmp_bytes = 0;
  uint32_t* keys_in = nullptr;
  uint32_t* keys_out = nullptr;
  uint32_t* vals_in = nullptr;
  uint32_t* vals_out = nullptr;
  size_t cap_pts = 0;
  unsigned long long *ck_in = nullptr, *ck_out = nullptr;      // 64-bit column keys of the index builds / inserts (flimo_map.hip)
  size_t ck_cap = 0;
  float* bbox = nullptr;   // 6 floats on device (min xyz, max xyz) as ordered ints; armed (empty box) whenever no reduction is running
  // "mail": 64 words of pinned host memory the device writes into (mail_words); the host reads them after synchronising the
  // stream.  Replaces the 4-byte device-to-host copies (each a staged, blocking copy) of counts and boxes.
  uint32_t* mail_host = nullptr;
  uint32_t* mail_dev = nullptr;
  uint32_t mail_seq = 0;
  // the one-launch input filter (filter_raw_scan): per tile {count, launch number} of two sums, then the tile ticket
  unsigned long long* filt_desc = nullptr;
  size_t filt_tiles_cap = 0;
  unsigned int filt_epoch = 0, filt_ticket_base = 0;
  unsigned long long* filt_mail_host = nullptr;   // mapped: four {value, launch number} granules (extreme key, kept count, NaN mark; tied stamps)
  unsigned long long* filt_mail_dev = nullptr;
};
// slots of the mail words
enum MailSlot { MAIL_BOOK = 0 /* 6 */, MAIL_BOOK_END = 8 /* 2 */, MAIL_CROWD = 12, MAIL_BOXCOUNT = 13, MAIL_BBOX = 16 /* 6 */, MAIL_VOXEL = 24 /* 4 */, MAIL_TILES = 28 /* build: tiles, overflow; merge: tiles, overflow */, MAIL_ROWS = 32 /* an insert found the point array full */,
                MAIL_TAG = 62 /* number of the last mail_words, written behind its words */, MAIL_WORDS = 64 };
struct MailPart { const void* src; int n; int dst; };
// queues ONE small kernel that copies up to 6 runs of words into the mail slots; `rearm_bbox`: S.bbox is reset to the empty box
// after it has been copied; `zero` / `zero_n`: words set to 0 after the copy (counters for the next round).  No synchronisation.
hipError_t mail_words(hipStream_t st, MapBuildScratch& S, const MailPart* parts, int nparts, bool rearm_bbox = false, void* zero = nullptr,
                      int zero_n = 0);
hipError_t ensure_mail(MapBuildScratch& S);
hipError_t mail_wait(hipStream_t st, MapBuildScratch& S);   // the words of the last mail_words are in S.mail_host (spins on their tag)

// min/max of n float4 points (NaN-free) -> host bbox[6]
hipError_t map_bbox(hipStream_t st, const float4* pts, size_t n, MapBuildScratch& S, float bbox_host[6]);
// Spatial (Morton) sort of the scan: out[i] = (xyz of in[perm[i]], w = bit pattern of perm[i]).
// Optionally permutes a per-point double array (times) the same way.
hipError_t sort_scan(hipStream_t st, const float4* in, size_t n, float4* out, MapBuildScratch& S,
                     const double* t_in = nullptr, double* t_out = nullptr);
// the same layout without re-ordering (out[i] = (xyz, w = i)): for a sweep a voxel filter re-orders anyway
hipError_t index_scan(hipStream_t st, const float4* in, size_t n, float4* out, const double* t_in, double* t_out);
// The index of a grid (GridView, flimo_types.h) as the host owns it: the pool of tiles (tile 0: all zero), the directory, the
// escape pool (one slot per 16 points of the point buffer's capacity is always enough) and xstart.  map_build_grid sizes and
// (re)allocates all of it; map_merge_grid takes tiles from the pool's room and reports when it ran out (index_merge_overflow).
struct IndexTables {
  uint2* tiles = nullptr; size_t tiles_cap_entries = 0;
  uint32_t cap_tiles = 0, tiles_used = 0;                 // tiles of the current shape the pool holds / numbers taken at the last build
  TileShape shape{5, 5, 3, 0, 0, 0};                      // the tile shape of the last layout (ts, ty, tz; the extents follow the grid)
  uint16_t* dir = nullptr;                                // GRID_DIR_MAX entries
  uint32_t* need = nullptr;                               // GRID_DIR_MAX words: tiles a batch of new points needs
  uint32_t* counters = nullptr;                           // [0] next free tile number, [1] a merge ran out of tiles, [2] escape slots taken
  uint32_t* ovf = nullptr; size_t ovf_cap = 0;            // words (8 per slot)
  uint32_t* xstart = nullptr; size_t xstart_cap = 0;
  uint32_t* rowcap = nullptr; size_t rowcap_cap = 0;      // [(nz+4)(ny+4)] points that fit from a row's first one to the next row's
  uint32_t* rowoff = nullptr; size_t rowoff_cap = 0;      // build scratch: the rows' first positions
  uint32_t* xstart_alt = nullptr; size_t xstart_alt_cap = 0;      // the second set of the small tables: a grid that grows
  uint32_t* rowcap_alt = nullptr; size_t rowcap_alt_cap = 0;      // (index_regrid) writes it and the two sets change places
  uint16_t* dir_alt = nullptr;
  uint32_t* tail = nullptr;                               // [0] first free position of the point array, [1] an insert found it full, [2] rows moved
};
void index_free(IndexTables& T);
// Sorts `pts_in` by (z, y, fine x column) into the rows of `pts_out` (room for out_cap points) and builds the index (ends with the
// stream waited for once: the pool is sized by the number of tiles the points need).  slack: the rows keep room behind their last
// point (a map that receives inserts) as far as out_cap allows; otherwise they are packed.  pts_cap: capacity of the point buffer
// the index is for (sizes the escape pool).
// geo: the grid's geometry (origin, cell, extents, column factor, cell shifts; its table pointers are not looked at).
hipError_t map_build_grid(hipStream_t st, const float4* pts_in, size_t n, float4* pts_out, size_t out_cap, bool slack, IndexTables& T, size_t pts_cap,
                          const GridView& geo, MapBuildScratch& S);
// A grid that grows without a re-sort: N = the new geometry (extents and cell shifts; origin, cell, column factor as O; corner moved
// by whole tiles of O's shape).  hipErrorInvalidValue when the larger grid needs another tile shape (the caller builds afresh).
hipError_t index_regrid(hipStream_t st, IndexTables& T, const GridView& O, GridView& N);
// more room in the pool of tiles before an insert needs it (the tiles move; the caller refreshes its views with index_view)
hipError_t index_grow_pool(hipStream_t st, IndexTables& T, uint32_t cap_tiles_new);
// fills the table pointers and the tile shape of a GridView whose geometry (nx, ny, nz, xs, nxf) is set
void index_view(const IndexTables& T, GridView& G);
// after the stream has been waited for: did a merge since the last build run out of tiles (the index is then incomplete)?
bool index_merge_overflow(const MapBuildScratch& S);
// debug: *diff_dev += the number of (row, column) pairs at which two indices of the same geometry differ
hipError_t index_compare(hipStream_t st, const GridView& A, const GridView& B, unsigned long long* diff_dev);
// Puts the k points appended since the last build into the rows of the cell-sorted array IN PLACE (same geometry): a row whose
// new points fit is merged where it is, a row that outgrows its room moves to the end of the array; the index of the rows that
// received points is rebuilt from those rows.  O(rows + touched rows x row length): no pass over the stored points.  A full array
// or tile pool is reported after the stream has been waited for (index_merge_overflow): the caller lays the map out afresh.
hipError_t map_merge_grid(hipStream_t st, float4* sorted, size_t sorted_cap, const float4* new_pts, size_t k,
                          IndexTables& T, const GridView& geo, MapBuildScratch& S);
// Input filters of a raw sweep (32-byte PointType records already on the device): NaN removal, crop box, every rate-th survivor,
// min distance; order preserved.  out[k] = (xyz, w = k), t_out[k] = stamp without the sweep offset; ext_dev[4] = {extreme ordered
// stamp key (complemented when the sweep is sorted descending), kept count, "a kept stamp is NaN", "two kept stamps are equal"
// (time_order_raw)}; key_out[k] (optional) = ordered stamp key, ascending = the order of the reference's time sort.
hipError_t filter_raw_scan(hipStream_t st, const void* raw32_dev, size_t n, const FilterParams& F, float4* out, double* t_out,
                           unsigned long long* ext_dev, MapBuildScratch& S, unsigned long long* key_out = nullptr, int rec_bytes = 32);
// (rec_bytes 16: records {x, y, z, 32-bit time word} instead of the reference's 32-byte PointType -- time kinds 0 and 1)
// (the three host-side results of the last filter_raw_scan -- extreme key, kept count, NaN mark -- without a copy and a stream wait)
hipError_t filter_raw_scan_result(hipStream_t st, MapBuildScratch& S, const unsigned long long* ext_dev, int timeout_ms, unsigned long long ext3[3]);
hipError_t time_order_raw_tied(hipStream_t st, MapBuildScratch& S, const unsigned long long* ext_dev, int timeout_ms, bool* tied);
// The kept points in the reference's time order (unique when no two stamps are equal; ext_dev[3] = 1 reports equal stamps): stable
// radix sort of the ordered stamp keys filter_raw_scan wrote, pts_out[i] = (xyz of pts[perm_out[i]], w = i), t_out likewise.
hipError_t time_order_raw(hipStream_t st, const float4* pts, const double* t, size_t n, const unsigned long long* keys,
                          unsigned long long* keys_sorted, float4* pts_out, double* t_out, uint32_t* perm_out,
                          unsigned long long* ext_dev, MapBuildScratch& S);
// Second level over crowded regions: box (cell coordinates, inclusive) around the cells holding more than `threshold` points
// (box_host[6] = their number; box_dev: 7 ints of device scratch), and the copy of the map points inside a box of metres
// (w = position in the main sorted map).
hipError_t crowded_list_all(hipStream_t st, const GridView& G, uint32_t threshold, uint32_t* bits,
                            int4* list, uint32_t cap, uint32_t* count_dev, uint32_t* count_host, MapBuildScratch& S);
hipError_t crowded_list_points(hipStream_t st, const float4* pts, size_t k, const GridView& G, uint32_t threshold, uint32_t* bits, int4* list,
                               uint32_t cap, uint32_t* count_dev, uint32_t* count_host, MapBuildScratch& S);
hipError_t crowded_relist(hipStream_t st, const int4* list, uint32_t m, int nx, int ny, int nz, uint32_t* bits, uint32_t* count_dev);
hipError_t map_box_count(hipStream_t st, const GridView& G, const int c0[3], const int c1[3],
                         uint32_t* count_dev, uint32_t* count_host, MapBuildScratch& S);
hipError_t map_box_copy(hipStream_t st, const GridView& G, const int c0[3], const int c1[3], float4* out, MapBuildScratch& S);
hipError_t atan2f_probe(hipStream_t st, const float* yx_host, int n, float* out_host);   // the device's atan2f on n (y, x) pairs
// pcl::VoxelGrid on device points: out gets one centroid per occupied voxel in ascending voxel index
hipError_t voxel_grid(hipStream_t st, const float4* in, size_t n, float leaf, float4* out, size_t* n_out, bool* passthrough,
                      MapBuildScratch& S);
void map_scratch_free(MapBuildScratch& S);

}  // namespace flimo
