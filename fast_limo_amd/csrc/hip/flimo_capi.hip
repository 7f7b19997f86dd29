// fast_limo_amd/csrc/hip/flimo_capi.hip -- C ABI (include/flimo_c.h) over the gfx950 kernels.
//
// Host-side glue only: context / buffer management, the float32 pose matrices the reference builds
// per pass (Objects/State.cpp:38-55,136-172; Modules/Localizer.cpp:549-555), launch sequencing and
// the decode of the MFMA accumulator layout.  No compute fallback lives here: without a HIP
// device every entry point fails with FLIMO_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <string>
#include <chrono>
#include <thread>
#include <array>
#include <vector>
#include <algorithm>
#include <immintrin.h>
#include "../../../include/flimo_c.h"
#include "../../../include/flimo_dev.h"
#include "flimo_types.h"
#include "flimo_kernels.h"
#include "flimo_math.h"
#include "flimo_pose.h"
#include "flimo_chain.h"
#include "flimo_ieskf.h"
#include "flimo_insert.h"
#include "flimo_gbook.h"

#pragma clang fp contract(off)

using namespace flimo;

struct flimo_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  std::string err;
  // config
  flimo_map_cfg map_cfg{0.2f, 2, 1, 0.5f};
  int timing = 0;                  // 0 off, 1 k-NN kernel only (events ride on its dispatch), 2 every stage
  int tail_max_env = 0;            // developer: FLIMO_TAIL_MAX overrides the straggler count up to which a launch finishes its own
  int timing_stride = 1;           // level 1: time every n-th pass only (sampling keeps the perturbation small)
  bool debug_recs = false;
  // map
  float4* d_map_raw = nullptr;     // insertion order
  float4* d_map_sorted = nullptr;  // cell order
  float gbox[6] = {0, 0, 0, 0, 0, 0};   // box the grid geometry was laid out for (the map box plus slack on the sides that grew)
  bool have_gbox = false;
  bool force_full = false;         // the next index update lays the grid out afresh (cell size changed)
  bool full_rebuild = false;       // FLIMO_FULL_REBUILD=1: sort the whole map on every insert (A/B of the merge)
  uint64_t grid_merges = 0, grid_builds = 0, grid_regrids = 0, index_overflows = 0, pool_grows = 0;
  bool have_origin = false;        // the origin of the map's cells is set (GridView: it stays; a grid that grows moves its corner by whole cells)
  size_t map_n = 0, map_cap = 0, sorted_cap = 0;
  bool sorted_follows = false;     // the raw buffer grew: the cell-sorted copy (3x its capacity) and the escape pool have to follow
  IndexTables idx;                 // the index of the main grid (GridView, flimo_types.h): tiles, directory, escapes, xstart
  GridView grid{};
  bool grid_valid = false;
  double map_last_time = -1.0;
  float bb[6] = {3.4e38f, 3.4e38f, 3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f};   // bounding box of the stored points
  MapBuildScratch scratch;
  InsertBook* book = nullptr;      // host statement of the insert rule (flimo_insert.h): A/B checks and flimo_insert_rule_replay only
  GBook gbook;                     // the same tree on the device: every later batch is decided there
  float4* d_batch = nullptr;       // staging for host-supplied later batches
  size_t batch_cap = 0;
  // scan
  float4* d_scan = nullptr;        // pc2match (body frame), caller order
  float4* d_scan_sorted = nullptr; // the same points in Morton order, w = original index
  void* d_nbr = nullptr;           // per-query neighbour records (sorted order)
  int* d_wl = nullptr;             // worklist of queries that need the general ring search
  int* d_wl_count = nullptr;
  void* d_raw32 = nullptr;         // unfiltered sweep as 32-byte PointType records (flimo_raw_scan_filter_set)
  size_t raw32_cap = 0;
  unsigned long long* d_filt_ext = nullptr;
  unsigned long long* h_filt_ext = nullptr;   // pinned
  double resident_t_offset = 0.0;  // sweep offset of the resident raw scan's stamps (0: already contained in them)
  unsigned long long* d_tkey[2] = {nullptr, nullptr};   // ordered stamp keys of the kept points / the same sorted (device time order)
  uint32_t* d_tperm = nullptr;     // time rank -> position among the kept points (arrival order)
  double* d_t_tmp = nullptr;
  size_t tkey_cap = 0;
  bool raw_time_ordered = false;   // the resident raw scan is in the reference's time order (stamps pairwise different)
  float4* d_scan_raw = nullptr;    // raw lidar-frame points for deskew (caller order)
  float4* d_raw_sorted = nullptr;  // the same in Morton order, w = original index
  double* d_t_sorted = nullptr;
  size_t raw_sorted_cap = 0;       // capacity of the two above
  float4* d_scan_world = nullptr;
  double* d_scan_t = nullptr;
  size_t scan_n = 0, scan_cap = 0, raw_n = 0;
  size_t order_n = 0;              // points of the last input stage run HERE (d_tperm; stays when the sweep is handed over)
  size_t sorted_n = 0;             // points in d_scan_sorted: scan_n, or the MAX_NUM_PC2MATCH prefix once a pass asked for it
  void* d_frames = nullptr;
  size_t frames_cap = 0;
  char* d_frames_fg = nullptr;              // two slots of FRAMES_FG_SLOT bytes in fine-grained device memory the host stores into (large-BAR devices):
                                            // the IMU frames of a deskew without a copy launch (DeskewArgs::stage_words)
  void* h_frames[2] = {nullptr, nullptr};   // pinned staging of the IMU frames, alternating: the deskew call does not wait
  size_t h_frames_cap = 0;
  int frames_slot = 0, async_deskews = 0;   // copies possibly still in flight since the stream was last known idle
  // A deskew that has been set up (frames uploaded) but not run: the scan's first k-NN launch runs it on the way (DeskewArgs),
  // anything else that reads d_scan / d_scan_sorted first runs the stand-alone kernel (flush_deskew)
  DeskewArgs deskew_args{};
  size_t deskew_n = 0;
  bool deskew_pending = false;
  // per pass
  Rec16* d_recs = nullptr;
  RecDbg* d_dbg = nullptr;
  size_t rec_cap = 0;
  int last_nq = 0;
  int reduce_waves = 256;
  double* d_partials = nullptr;    // records-mode reduction partials [reduce_waves][256]
  double* d_fit_partials = nullptr;  // per fit-block partials [fit_blocks][256]
  size_t fit_partials_cap = 0;
  double* d_out256 = nullptr;
  double* h_out256 = nullptr;      // pinned + mapped: the last fit block writes the result straight to host memory
  double* d_out256_host = nullptr; // device alias of h_out256
  unsigned int* d_ticket = nullptr;
  // fit2 (per-pass fast path): per-block partials of the FIT_LIVE sums, granule slots in mapped host memory
  double* d_fit2_partials = nullptr;
  size_t fit2_partials_cap = 0;
  double* h_granules = nullptr;    // pinned + mapped: [FIT_GROUPS][FIT_LIVE_PAD] x {sum, pass number}
  double* d_granules_host = nullptr;
  unsigned char live_idx[FIT_LIVE_PAD];   // live sum k -> raw MFMA accumulator index (from the calibrated layout)
  bool fuse = true;                // FLIMO_FUSE=0: k-NN and fit stay separate dispatches in every pass (A/B checks)
  unsigned long long fused_passes = 0;
  unsigned long long pass_seq = 0;  // last pass number published by the fit kernel
  unsigned long long* d_cand = nullptr;
  unsigned long long* h_cand = nullptr;  // pinned
  double last_cand_per_query = 0.0;
  int mfma_idx[16][16];            // (i,j) -> raw index
  // staging
  void* h_stage = nullptr;         // pinned
  void* h_clouds = nullptr;        // pinned: the two clouds of flimo_scan_clouds
  void (*overlap_fn)(void*) = nullptr;   // flimo_match_reduce_overlap: host work of the caller to run while the pass is in flight
  void* overlap_arg = nullptr;
  size_t clouds_cap = 0;
  size_t stage_cap = 0;
  // timing
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [0,1] k-NN / fused dispatch, [2,3] fit, [4,5] widening
  // deferred reading (flimo_set_timing_deferred): each timed pass takes the next set of a ring and nobody reads it until the totals
  // are asked for -- reading a pass's events right behind it costs the host tens of microseconds per pass, and a GPU that idles
  // through those lowers its clocks on some boxes (the same kernels then read 10-15 % long)
  static constexpr int EV_RING = 64;
  hipEvent_t ev_ring[EV_RING][6];
  unsigned char ev_kind[EV_RING];   // bit 0: one-launch pass, bit 1: fit2 layout, bit 2: the widening was timed
  int ev_pending = 0;
  bool ev_ring_made = false, timing_deferred = false;
  float last_knn_ms = 0.f, last_widen_ms = 0.f, last_fit_ms = 0.f;
  int* h_wl_count = nullptr;   // pinned
  int last_widen_count = 0;
  double tot_knn_ms = 0, tot_widen_ms = 0, tot_fit_ms = 0;
  long long tot_passes = 0, tot_queries = 0;
  // level-1 totals by kind of timed pass: [0] one-launch passes: total ms, count; [1] separate dispatches: k-NN ms, widening ms, fit ms, count
  double split_fused_ms = 0, split_knn_ms = 0, split_widen_ms = 0, split_fit_ms = 0;
  long long split_fused_n = 0, split_sep_n = 0;
  flimo_match_cfg last_cfg{};
  PoseMats last_P{};
  MatchParams last_mp{};
  int last_n_all = 0;
  bool recs_valid = false, dbg_valid = false;
  PrevPass prev{};                 // previous pass of the same resident scan (k-NN pruning bound); valid = 0 after any scan change
  bool prune = true;               // FLIMO_PRUNE=0 disables the bound (A/B checks)
  unsigned probe_min = 96;         // FLIMO_PROBE=<n>: first pass, a query with >= n candidates in its 3x3x3 block walks its own cell first for a bound (0: off)
  int stragglers_hist[4] = {1 << 30, 0, 0, 0};   // the same per pass position within a scan (0 = first pass .. 3 = fourth and later), last scan that reported
  int pass_in_scan = 0;
  int last_stragglers = -1;        // the same of the last pass (-1: not reported by that pass's path)
  // second level over crowded regions (flimo_map.hip): a grid with a quarter of the cell edge over the box around the cells near the
  // sensor that hold more than fine_threshold points, with a copy of every map point inside it
  bool fine_on = true;             // FLIMO_FINE=0 switches it off (A/B checks)
  unsigned fine_threshold = 64;    // FLIMO_FINE_THRESHOLD
  int fine_div = 4;                // FLIMO_FINE_DIV: fine cells per cell edge (power of two)
  unsigned fine_min_points = 32768;// FLIMO_FINE_MIN_POINTS: smaller crowded regions are not worth the extra dispatch (measured: 1M map, 7k points: +4 us)
  bool fine_valid = false;
  bool crowd_box_valid = false;    // the crowded-cell list (device bits + list, host copy) belongs to the current geometry
  GridView fine{};
  int fine_qlo[3] = {0, 0, 0}, fine_qhi[3] = {-1, -1, -1};
  float4 *d_fine_tmp = nullptr, *d_fine_pts = nullptr;
  size_t fine_pts_cap = 0;
  IndexTables fine_idx;
  uint32_t* d_fine_count = nullptr;
  void* d_crowd_list = nullptr;        // int4 (x, y, z, -) of every crowded cell of the current geometry, listed once
  uint32_t* d_crowd_count = nullptr;
  uint32_t* d_crowd_bits = nullptr;    // one bit per cell: listed
  size_t crowd_bits_cap = 0;
  uint32_t crowd_listed = 0;           // entries of the device list the host has fetched
  std::vector<std::array<int, 3>> crowd_cells;
  float fine_center[3] = {0.f, 0.f, 0.f};   // sensor position of the last inserted scan (world)
  bool have_fine_center = false;
  float fine_radius = 24.f;            // FLIMO_FINE_RADIUS [m]: crowded cells farther from the sensor (xy) stay out of the region
  uint64_t fine_builds = 0, fine_passes = 0;
  int xslabs = 2;                  // fine columns per cell along x (1, 2, 4, 8; power of two): a pass that has the
                                   // previous pass's bound walks only the half cells its ball reaches.  Measured (round 3, 1M map):
                                   // clean map no change (21.8 -> 22.1 us), after 50 raw 64k sweeps the later passes 30.9 -> 23.2 us
                                   // (1.05x the clean map's); inserts unchanged at 2 (0.28 ms), +14 % at 4; both tables twice the size
                                   // (the layout falls back to fewer columns where 32-bit indices would not do)
  int* d_tie_list = nullptr;       // queries of a pass whose five hinge on an exact distance tie (capacity = scan capacity)
  unsigned int* d_tie_count = nullptr;   // [2]: alternating by pass number (the pass's reduction re-arms the next one's)
  size_t tie_cap = 0;
  bool ties = true;                // FLIMO_TIES=0: leave ties to the position rule (A/B checks)
  unsigned long long tie_redos = 0, tie_queries = 0;
  unsigned long long* d_tie_settled = nullptr;   // queries whose ties were settled inside a reducing launch (tie_repair_wave), counted on the device
  void* d_nbrk = nullptr;          // neighbour records of the general pass
  size_t nbrk_cap = 0;
  int wait_timeout_ms = 2000;      // wall-clock bound of the wait for a pass's result (flimo_set_wait_timeout_ms)
  hipEvent_t adopt_ev = nullptr;   // flimo_scan_adopt: orders the two contexts' streams around the hand-over copies
  hipEvent_t timeout_ev = nullptr; // recorded behind the launches of a pass / chain whose wait ran out
  bool timeout_pending = false;    // ... and not yet seen complete: no new pass is queued on top of it (check_abandoned)
  // the whole iterated update enqueued at once (flimo_chain.h, flimo_update_chain)
  ChainState* d_chain = nullptr;         // device filter state
  void* d_chain_gran = nullptr;          // device copy of the granule slots: a chained pass publishes its sums here
  ChainPrior* h_chain_prior = nullptr;   // mapped: the prior of the scan's update (read by the first algebra kernel)
  ChainPrior* d_chain_prior = nullptr;   // its device alias
  double* h_chain_res = nullptr;         // mapped: CH_RES result granules {value, tag}
  void* d_chain_res = nullptr;
  double* h_chain_log = nullptr;         // mapped: CH_MAX_PASSES x CH_LOGN log granules
  void* d_chain_log = nullptr;
  unsigned long long chain_tag = 0x4000000000000000ull;   // tag of the last chain (own number space)
  // Host loop, pipelined (FLIMO_PIPELINE=0 switches it off): while a one-launch pass runs, the NEXT pass of the same update is already
  // queued behind it -- a chained-pass kernel whose workgroups wait, placed on the GPU, for their constants in `d_pipe_head`
  // (fine-grained device memory the host stores into through the PCIe BAR).  The next flimo_match_reduce then publishes the pose
  // instead of launching: no doorbell -> dispatch -> kernel start on the iteration's critical path.  A pass nobody asks for (the
  // update converged) is told to leave by the next call on the context (ctx_enter).
  struct Prelaunch {
    bool active = false;
    unsigned long long seq = 0;      // the pass number it will publish under
    size_t nq = 0;
    int n_all = 0, pos = 0;
    flimo_match_cfg cfg{};
    uint64_t grid_version = 0;
    unsigned int end_code = 0;
    double t_launch = 0.0;
  } pre;
  bool pipeline = false;                 // flimo_set_pass_pipeline (FLIMO_PIPELINE=0/1 presets it and wins)
  bool pipeline_env = false;
  ChainHead* d_pipe_head = nullptr;      // fine-grained device memory (host-writable); nullptr: not available on this system
  unsigned int pipe_tag = 0;
  bool pipe_last_hint = false;           // flimo_pass_pipeline_last: the next pass is its update's last -- nothing is queued behind it
  unsigned long long pipe_published = 0, pipe_cancelled = 0;   // statistics
  unsigned long long pipe_aged = 0, pipe_left = 0;             // passes found too old to be published to / that left before the publish reached them
  bool row_slack = true;                  // FLIMO_ROW_SLACK=0: a full layout packs the rows (A/B of the room behind every row)
  bool test_tight_array = false;          // FLIMO_TEST_TIGHT_ARRAY (tests): the cell-sorted array gets 4096 points of room instead of twice the map
  int test_publish_delay_ms = 0;         // FLIMO_TEST_PUBLISH_DELAY_MS (tests): a sleep between the age check of a waiting pass and the publish
  PrevPass prev_before{};                // `prev` as the pass in flight was given it (a pass that has to be launched a second time)
  // Which way the iterated update runs: the chain costs about 11 us per pass on top of the pass's kernels whatever the host (the
  // algebra launch and two dispatch boundaries); the host loop costs this host's launch -> result round trip + 2-3 us of algebra --
  // 9 us on a fast host, 15-19 us on a slow one (BENCH_r03: 5 497 scans/s where the builder's box gave 7 102).  The round trip is
  // measured once at context creation (launch_rtt_us) and REPORTED; update_mode 0 and 1 = host loop, 2 = chain (the caller's choice).
  int update_mode = 0;                   // FLIMO_HOST_UPDATE=1 -> 1, FLIMO_HOST_UPDATE=0 -> 2, unset -> 0 (flimo_set_update_mode)
  double launch_rtt_us = 0.0;            // launch -> granule seen (lower quartile) of a one-thread kernel on this host
  bool host_update = false;              // (the choice in force) flimo_update_chain always declines (the host loop runs the update; A/B)
  hipEvent_t chain_ev[CH_MAX_PASSES][8]; // per pass: [0,1] first launch, [2,3] fit launch, [4,5] algebra launch, [6,7] widening launch (lazy)
  bool chain_ev_made = false;
  double chain_alg_ms = 0;
  long long chain_alg_n = 0, chains_run = 0, chains_back = 0, chains_declined = 0;
  int fov_check = 0, fov_check_bad = 0;   // device atan2f vs this host's libm: 0 not checked yet, 1 equal, -1 different (the FoV filter is then declined)
  bool tail = true;                // FLIMO_TAIL=0: pending queries go to the worklist + widen_kernel dispatch instead of being finished inside the k-NN launch (A/B checks)
};

constexpr size_t FRAMES_FG_SLOT = 8192;             // bytes per slot of flimo_ctx::d_frames_fg (about 70 IMU frames)
static int timing_drain(flimo_ctx* c);            // reads the timed passes nobody has read yet (deferred reading)
static inline void ctx_enter(flimo_ctx* c);       // hipSetDevice + a pass queued ahead of the filter's algebra is told to leave (see cancel_prelaunch)
static int fail(flimo_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}
#define HIPCHK(c, call)                                                                          \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess) return fail(c, FLIMO_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

static int ensure_stage(flimo_ctx* c, size_t bytes) {
  if (bytes <= c->stage_cap) return FLIMO_OK;
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  c->h_stage = nullptr;
  c->stage_cap = 0;
  size_t cap = bytes + bytes / 4 + 4096;
  HIPCHK(c, hipHostMalloc(&c->h_stage, cap, hipHostMallocDefault));
  c->stage_cap = cap;
  return FLIMO_OK;
}

template <typename T>
static int ensure_dev(flimo_ctx* c, T*& p, size_t& cap, size_t need, bool keep, size_t keep_n) {
  if (need <= cap) return FLIMO_OK;
  size_t ncap = need + need / 4 + 1024;
  T* np = nullptr;
  HIPCHK(c, hipMalloc(&np, ncap * sizeof(T)));
  if (keep && p && keep_n) {
    HIPCHK(c, hipMemcpyAsync(np, p, keep_n * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (p) (void)hipFree(p);
  p = np;
  cap = ncap;
  return FLIMO_OK;
}

// (the float32 pose algebra of a pass -- State casts, get_RT / get_RT_inv / get_extr_RT_inv, the conjugate rotations of
//  calculate_H -- lives in flimo_pose.h: the host and the device filter form the same constants from the same code)

// ---- developer switches: every environment variable this library reads, in one place ------------------------------------------
// None of them changes a result (the parity tests run under them); they exist for A/B measurements, fault isolation and the tests
// that need a path forced.  Round 5 removed the ones nothing exercised (lanes per query, forced tail / general-k, x columns, XCD
// stripe, fine divisor / radius, host insert, eager deskew, the frames' BAR switch) together with the code paths they selected.
//   FLIMO_FUSE=0                  k-NN and fit stay separate dispatches in every pass (default: the whole pass is one launch)
//   FLIMO_TAIL=0                  pending queries always go to the worklist + widening dispatch instead of the in-kernel tail
//   FLIMO_PRUNE=0                 no pruning by the previous pass's bound
//   FLIMO_PROBE=<n>               first pass: a query with >= n candidates in its 3x3x3 block walks its own cell first (96; 0: off)
//   FLIMO_TIES=0                  exact distance ties keep the position rule (default: the reference's first-met rule)
//   FLIMO_FINE=0, FLIMO_FINE_THRESHOLD=<points per cell, 64>, FLIMO_FINE_MIN_POINTS=<32768>   second-level grid over crowded regions
//   FLIMO_FULL_REBUILD=1          the index is sorted from scratch on every insert (default: merged)
//   FLIMO_HOST_UPDATE=<1|0>       the iterated update runs as a host loop over single passes (the default) / as a chain queued at once
//                                 (flimo_update_chain)
//   FLIMO_PIPELINE=0              host loop: no pass is queued ahead of the filter's algebra (default: the next one-launch pass waits on the
//                                 GPU for its pose, which the host stores into device memory)
//   FLIMO_NO_BAR=1                behave like a system that does not map device memory for the host (no pipelined loop, staged IMU frames)
//   FLIMO_ROW_SLACK=0             a full layout packs the rows of the cell-sorted array (default: half a row's length of room behind each)
//   FLIMO_TEST_TIGHT_ARRAY=1      (tests) the cell-sorted point array has room for 4096 more points only: inserts find it full and the
//                                 map is laid out afresh (the path a long drive takes when the array fills up)
//   FLIMO_TEST_PUBLISH_DELAY_MS   (tests) a sleep between the age check of a waiting pass and the publish of its pose
//   FLIMO_PROF_PASS / FLIMO_PROF_INSERT   host-side timing prints (stderr)
// (csrc/host: FLIMO_REFERENCE_SOLVE=1 literal two-inverse gain, FLIMO_NO_FRONT_CTX=1 input stage on the main context,
//  FLIMO_PROF_DESKEW / FLIMO_PROF_CLOUDS timing prints; bench.py: FLIMO_BENCH_*.)
static void load_dev_switches(flimo_ctx* c) {
  auto env_int = [](const char* name, int& out) { const char* e = getenv(name); if (!e) return false; out = atoi(e); return true; };
  int v = 0;
  if (env_int("FLIMO_FUSE", v)) c->fuse = v != 0;
  if (env_int("FLIMO_TAIL", v)) c->tail = v != 0;
  if (env_int("FLIMO_PRUNE", v)) c->prune = v != 0;
  if (env_int("FLIMO_PROBE", v) && v >= 0) c->probe_min = (unsigned)v;
  if (env_int("FLIMO_TIES", v)) c->ties = v != 0;
  if (env_int("FLIMO_FINE", v)) c->fine_on = v != 0;
  if (env_int("FLIMO_FINE_THRESHOLD", v) && v > 0) c->fine_threshold = (unsigned)v;
  if (env_int("FLIMO_FINE_MIN_POINTS", v) && v >= 0) c->fine_min_points = (unsigned)v;
  if (env_int("FLIMO_FULL_REBUILD", v)) c->full_rebuild = v != 0;
  if (env_int("FLIMO_HOST_UPDATE", v)) c->update_mode = v != 0 ? 1 : 2;
  if (env_int("FLIMO_PIPELINE", v)) { c->pipeline = v != 0; c->pipeline_env = true; }
  if (env_int("FLIMO_TEST_PUBLISH_DELAY_MS", v) && v > 0) c->test_publish_delay_ms = v;
  if (env_int("FLIMO_TEST_TIGHT_ARRAY", v)) c->test_tight_array = v != 0;
  if (env_int("FLIMO_ROW_SLACK", v)) c->row_slack = v != 0;
  if (env_int("FLIMO_TAIL_MAX", v) && v > 0) c->tail_max_env = v;
}

// Does the GPU see what the HOST stores into this allocation?  The host writes a pattern into the head's epoch word (a plain store
// through the BAR, as publish_prelaunch does), a one-thread kernel reads the word past the caches and hands it back in a granule
// of mapped host memory; twice, with different patterns.  A mapping that is missing altogether would fault at the first store --
// the large-BAR attribute is what stands for its presence (hipPointerGetAttributes reports such memory as plain device memory);
// FLIMO_NO_BAR=1 switches the path off on a system where that is not enough.
static bool host_store_probe(flimo_ctx* c, ChainHead* head) {
  volatile unsigned long long* tagp = reinterpret_cast<volatile unsigned long long*>(c->h_chain_res) + 1;
  for (unsigned int round = 0; round < 2; round++) {
    const unsigned int pattern = 0x5a17c0deu ^ (round * 0x01010101u);
    __atomic_store_n(&head->epoch, pattern, __ATOMIC_RELEASE);
    _mm_sfence();
    const unsigned long long tag = 0x7200000000000000ull + round;
    launch_word_probe(c->stream, &head->epoch, c->d_chain_res, tag);
    if (hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (*tagp != tag) return false;
    const double seen = c->h_chain_res[0];
    if ((unsigned int)llround(seen) != pattern) return false;
  }
  __atomic_store_n(&head->epoch, 0u, __ATOMIC_RELEASE);
  _mm_sfence();
  memset(c->h_chain_res, 0, 2 * sizeof(double));
  return true;
}

// ---- context ----------------------------------------------------------------------------------
extern "C" const char* flimo_version(void) { return "fast_limo_amd 0.1.0 (gfx950)"; }

extern "C" const char* flimo_last_error(const flimo_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int flimo_ctx_create(int device, flimo_ctx** out) {
  if (!out) return FLIMO_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return FLIMO_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev) return FLIMO_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return FLIMO_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return FLIMO_ERR_NO_DEVICE;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return FLIMO_ERR_NO_DEVICE;   // kernels are gfx950 only
  flimo_ctx* c = new flimo_ctx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return FLIMO_ERR_HIP; }
  for (int i = 0; i < 6; i++) (void)hipEventCreate(&c->ev[i]);
  bool ok = hipMalloc(&c->d_partials, (size_t)c->reduce_waves * 256 * sizeof(double)) == hipSuccess &&
            hipMalloc(&c->d_out256, FIT_GROUPS * FIT_SLOT * sizeof(double)) == hipSuccess &&
            hipHostMalloc((void**)&c->h_out256, FIT_GROUPS * FIT_SLOT * sizeof(double), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**)&c->d_out256_host, c->h_out256, 0) == hipSuccess &&
            hipHostMalloc((void**)&c->h_granules, FIT_GROUPS * FIT_LIVE_PAD * 2 * sizeof(double), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**)&c->d_granules_host, c->h_granules, 0) == hipSuccess &&
            hipMalloc(&c->d_ticket, (FIT_GROUPS + 2) * sizeof(unsigned int)) == hipSuccess &&      // one per group + one launch-wide + the chained passes' third
            hipMemset(c->d_ticket, 0, (FIT_GROUPS + 2) * sizeof(unsigned int)) == hipSuccess &&
            hipMalloc(&c->d_cand, sizeof(unsigned long long)) == hipSuccess &&
            hipMalloc(&c->d_tie_count, 2 * sizeof(unsigned int)) == hipSuccess &&
            hipMemset(c->d_tie_count, 0, 2 * sizeof(unsigned int)) == hipSuccess &&
            hipMalloc((void**)&c->d_chain, chain_state_size()) == hipSuccess &&
            hipMemset(c->d_chain, 0, chain_state_size()) == hipSuccess &&
            hipMalloc(&c->d_chain_gran, FIT_GROUPS * FIT_LIVE_PAD * 2 * sizeof(double)) == hipSuccess &&
            hipMemset(c->d_chain_gran, 0, FIT_GROUPS * FIT_LIVE_PAD * 2 * sizeof(double)) == hipSuccess &&
            hipHostMalloc((void**)&c->h_chain_prior, sizeof(ChainPrior), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**)&c->d_chain_prior, c->h_chain_prior, 0) == hipSuccess &&
            hipHostMalloc((void**)&c->h_chain_res, (size_t)CH_RES * 2 * sizeof(double), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&c->d_chain_res, c->h_chain_res, 0) == hipSuccess &&
            hipHostMalloc((void**)&c->h_chain_log, (size_t)CH_MAX_PASSES * CH_LOGN * 2 * sizeof(double), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(&c->d_chain_log, c->h_chain_log, 0) == hipSuccess &&
            hipMalloc((void**)&c->d_tie_settled, sizeof(unsigned long long)) == hipSuccess &&
            hipMemset(c->d_tie_settled, 0, sizeof(unsigned long long)) == hipSuccess &&
            hipMalloc(&c->d_wl_count, sizeof(int)) == hipSuccess &&
            hipMemset(c->d_wl_count, 0, sizeof(int)) == hipSuccess &&
            hipHostMalloc((void**)&c->h_wl_count, sizeof(int), hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void**)&c->h_cand, sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess;
  if (!ok) { flimo_ctx_destroy(c); return FLIMO_ERR_HIP; }
  memset(c->h_chain_res, 0, (size_t)CH_RES * 2 * sizeof(double));
  memset(c->h_chain_log, 0, (size_t)CH_MAX_PASSES * CH_LOGN * 2 * sizeof(double));
  memset(c->h_out256, 0, FIT_GROUPS * FIT_SLOT * sizeof(double));
  memset(c->h_granules, 0, FIT_GROUPS * FIT_LIVE_PAD * 2 * sizeof(double));
  // calibrate the v_mfma_f64_16x16x4_f64 accumulator layout: D[i][j] = j + 16 i
  launch_mfma_layout(c->stream, c->d_out256);
  if (hipMemcpyAsync(c->h_out256, c->d_out256, 256 * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
      hipStreamSynchronize(c->stream) != hipSuccess) { flimo_ctx_destroy(c); return FLIMO_ERR_HIP; }
  bool seen[256] = {false};
  for (int r = 0; r < 256; r++) {
    const double v = c->h_out256[r];
    const int code = (int)llround(v);
    if (code < 0 || code > 255 || fabs(v - code) > 1e-9 || seen[code]) { flimo_ctx_destroy(c); return FLIMO_ERR_HIP; }
    seen[code] = true;
    c->mfma_idx[code / 16][code % 16] = r;
  }
  {
    // the sums the filter reads: upper triangle of H^T H (78), H^T h (12), M (1)
    int k = 0;
    for (int i = 0; i < 12; i++) for (int j = i; j < 12; j++) c->live_idx[k++] = (unsigned char)c->mfma_idx[i][j];
    for (int i = 0; i < 12; i++) c->live_idx[k++] = (unsigned char)c->mfma_idx[i][12];
    c->live_idx[k++] = (unsigned char)c->mfma_idx[13][13];
    for (; k < FIT_LIVE_PAD; k++) c->live_idx[k] = 0;
  }
  c->book = insert_book_create();
  load_dev_switches(c);
  {
    // the pipelined host loop's head: device memory the HOST stores into (fine-grained; reached through the PCIe BAR).  Not every
    // system maps it: without it the host loop launches every pass when its pose is known, as before.
    void* p = nullptr;
    int large_bar = 0;
    (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device);      // (host stores into device memory need the whole of it behind the BAR)
    if (getenv("FLIMO_NO_BAR") != nullptr) large_bar = 0;      // (stands in for a system that does not map device memory for the host)
    if (large_bar && hipExtMallocWithFlags(&p, sizeof(ChainHead), hipDeviceMallocFinegrained) == hipSuccess && p &&
        hipMemset(p, 0, sizeof(ChainHead)) == hipSuccess && hipDeviceSynchronize() == hipSuccess && host_store_probe(c, static_cast<ChainHead*>(p))) {
      c->d_pipe_head = static_cast<ChainHead*>(p);
      void* q = nullptr;
      if (hipExtMallocWithFlags(&q, 2 * FRAMES_FG_SLOT, hipDeviceMallocFinegrained) == hipSuccess && q)
        c->d_frames_fg = static_cast<char*>(q);
      else (void)hipGetLastError();
    } else {
      if (p) (void)hipFree(p);
      (void)hipGetLastError();
      c->d_pipe_head = nullptr;
    }
  }
  {
    // launch -> result round trip of this host (lower quartile of 32 after 8 warm-ups: a property of the host, not of what else
    // runs at the moment): a one-thread kernel stores a granule to mapped memory, the host spins on its tag -- what every
    // host-driven pass pays beyond its kernels
    std::vector<double> rt;
    volatile unsigned long long* tagp = reinterpret_cast<volatile unsigned long long*>(c->h_chain_res) + 1;
    for (int i = 0; i < 40; i++) {
      const unsigned long long tag = 0x7100000000000000ull + (unsigned long long)i;
      const auto t0 = std::chrono::steady_clock::now();
      launch_rtt_probe(c->stream, c->d_chain_res, tag);
      unsigned long long spins = 0;
      while (*tagp != tag && ++spins < 200000000ull) _mm_pause();
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (i >= 8) rt.push_back(us);
    }
    (void)hipStreamSynchronize(c->stream);
    std::sort(rt.begin(), rt.end());
    c->launch_rtt_us = rt[rt.size() / 4];
    memset(c->h_chain_res, 0, 2 * sizeof(double));
    c->host_update = c->update_mode != 2;          // (the round trip is reported, not acted on: see auto_host_update)
  }
  *out = c;
  return FLIMO_OK;
}

extern "C" void flimo_ctx_destroy(flimo_ctx* c) {
  if (!c) return;
  ctx_enter(c);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  (void)hipFree(c->d_map_raw); (void)hipFree(c->d_map_sorted); index_free(c->idx); index_free(c->fine_idx);
  (void)hipFree(c->d_scan_sorted); (void)hipFree(c->d_nbr); (void)hipFree(c->d_wl); (void)hipFree(c->d_wl_count);
  (void)hipFree(c->d_fit_partials); (void)hipFree(c->d_raw_sorted); (void)hipFree(c->d_t_sorted);
  (void)hipFree(c->d_scan); (void)hipFree(c->d_scan_raw); (void)hipFree(c->d_scan_world); (void)hipFree(c->d_scan_t);
  (void)hipFree(c->d_frames); (void)hipFree(c->d_recs); (void)hipFree(c->d_dbg);
  (void)hipFree(c->d_fine_tmp); (void)hipFree(c->d_fine_pts); 
  (void)hipFree(c->d_fine_count); (void)hipFree(c->d_crowd_list); (void)hipFree(c->d_crowd_count); (void)hipFree(c->d_crowd_bits);
  (void)hipFree(c->d_tkey[0]); (void)hipFree(c->d_tkey[1]); (void)hipFree(c->d_tperm); (void)hipFree(c->d_t_tmp);
  (void)hipFree(c->d_raw32); (void)hipFree(c->d_filt_ext); (void)hipFree(c->d_nbrk); (void)hipFree(c->d_tie_list); (void)hipFree(c->d_tie_count);
  if (c->h_filt_ext) (void)hipHostFree(c->h_filt_ext);
  (void)hipFree(c->d_partials); (void)hipFree(c->d_out256); (void)hipFree(c->d_cand); (void)hipFree(c->d_ticket);
  if (c->h_out256) (void)hipHostFree(c->h_out256);
  if (c->h_granules) (void)hipHostFree(c->h_granules);
  (void)hipFree(c->d_fit2_partials);
  if (c->h_cand) (void)hipHostFree(c->h_cand);
  (void)hipFree(c->d_chain); (void)hipFree(c->d_chain_gran); (void)hipFree(c->d_tie_settled);
  if (c->h_chain_prior) (void)hipHostFree(c->h_chain_prior);
  if (c->h_chain_res) (void)hipHostFree(c->h_chain_res);
  if (c->h_chain_log) (void)hipHostFree(c->h_chain_log);
  if (c->chain_ev_made) for (int i = 0; i < CH_MAX_PASSES; i++) for (int k = 0; k < 8; k++) (void)hipEventDestroy(c->chain_ev[i][k]);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->h_clouds) (void)hipHostFree(c->h_clouds);
  for (int k = 0; k < 2; k++) if (c->h_frames[k]) (void)hipHostFree(c->h_frames[k]);
  map_scratch_free(c->scratch);
  for (int i = 0; i < 6; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->ev_ring_made) for (int r = 0; r < flimo_ctx::EV_RING; r++) for (int i = 0; i < 6; i++) (void)hipEventDestroy(c->ev_ring[r][i]);
  if (c->timeout_ev) (void)hipEventDestroy(c->timeout_ev);
  if (c->adopt_ev) (void)hipEventDestroy(c->adopt_ev);
  if (c->h_wl_count) (void)hipHostFree(c->h_wl_count);
  if (c->d_pipe_head) (void)hipFree(c->d_pipe_head);
  if (c->d_frames_fg) (void)hipFree(c->d_frames_fg);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->book) insert_book_destroy(c->book);
  c->gbook.release();
  (void)hipFree(c->d_batch);
  delete c;
}

// ---- map --------------------------------------------------------------------------------------
extern "C" int flimo_map_config(flimo_ctx* c, const flimo_map_cfg* cfg) {
  if (!c || !cfg) return FLIMO_ERR_INVALID;
  if (!(cfg->min_extent > 0.f)) return fail(c, FLIMO_ERR_INVALID, "min_extent must be > 0");
  const float cell_before = c->map_cfg.cell_size;
  c->map_cfg = *cfg;
  if (!(c->map_cfg.cell_size > 0.f)) c->map_cfg.cell_size = 0.5f;
  if (c->map_cfg.cell_size != cell_before) { c->force_full = true; c->have_gbox = false; c->have_origin = false; }   // takes effect at the next index update
  insert_book_config(c->book, cfg->min_extent, cfg->downsample != 0);
  return FLIMO_OK;
}

extern "C" int flimo_map_clear(flimo_ctx* c) {
  if (!c) return FLIMO_ERR_INVALID;
  c->map_n = 0;
  c->grid_valid = false;
  c->fine_valid = false;
  c->have_gbox = false;
  c->have_origin = false;
  c->map_last_time = -1.0;
  c->bb[0] = c->bb[1] = c->bb[2] = 3.4e38f; c->bb[3] = c->bb[4] = c->bb[5] = -3.4e38f;
  c->crowd_box_valid = false;
  c->have_fine_center = false;
  insert_book_clear(c->book);
  c->gbook.active = false;
  c->prev.valid = 0;               // the pruning bound only survives map ADDITIONS (distances can only shrink)
  return FLIMO_OK;
}

extern "C" size_t flimo_map_size(const flimo_ctx* c) { return c ? c->map_n : 0; }
extern "C" double flimo_map_last_time(const flimo_ctx* c) { return c ? c->map_last_time : -1.0; }

// Does the current grid geometry hold every point of box `bb` with the margins the kernels assume (lowest cell index
// >= 0, highest <= n-2)?  Same float expressions as the kernels.
// Along x the lowest cell is kept out of the FIRST SEGMENT (8 fine columns) of the row: a point in a tile's first segment makes the
// tile to its left exist and carry the row's closing entry (GridView) -- x-tile 0 has no left neighbour, and when the corner later
// moves down by whole tiles (index_regrid) the old x-tile 0 gets one.  With the first segment of x-tile 0 always empty, every row
// that has points in a tile's first segment got them through tiles_mark_kernel / row_entries, which write that closing entry.
static inline int grid_low_x_cells(int xs) { return (8 + xs - 1) / xs; }
static bool grid_side_low_ok(const GridView& g, const float* bb, int a) {
  const float o[3] = {g.ox, g.oy, g.oz};
  const int si[3] = {g.six, g.siy, g.siz};
  const int low = a == 0 ? grid_low_x_cells(g.xs) : 0;
  if ((int)floorf((bb[a] - 0.5f * g.cell - o[a]) * g.inv_cell) - si[a] < low) return false;
  if ((int)floorf((bb[a] - o[a]) * g.inv_cell) - si[a] < low) return false;
  return true;
}
static bool grid_covers(const GridView& g, const float* bb) {
  const float o[3] = {g.ox, g.oy, g.oz};
  const int n[3] = {g.nx, g.ny, g.nz}, si[3] = {g.six, g.siy, g.siz};
  for (int a = 0; a < 3; a++) {
    if (!grid_side_low_ok(g, bb, a)) return false;
    if ((int)floorf((bb[3 + a] - o[a]) * g.inv_cell) - si[a] > n[a] - 2) return false;
  }
  return true;
}
// Second level: (re)built after every index update.  Cheap when no cell is crowded (a look at the cells of the new points).
// `relayout`: the geometry is new -> every cell is looked at; otherwise only the cells of the n_new points merged since the last look.
static int update_fine_grid(flimo_ctx* c, bool relayout, const float4* new_pts = nullptr, size_t n_new = 0) {
  c->fine_valid = false;
  if (!c->fine_on || !c->grid_valid || c->map_n == 0) return FLIMO_OK;
  const GridView& g = c->grid;
  if ((double)g.nx * (double)g.ny * (double)g.nz > 2.0e9) return FLIMO_OK;      // (a bit per cell: not for a grid of many kilometres)
  constexpr uint32_t CROWD_CAP = 1u << 20;
  if (!c->d_crowd_list) {
    HIPCHK(c, hipMalloc(&c->d_crowd_list, (size_t)CROWD_CAP * sizeof(int4)));
    HIPCHK(c, hipMalloc(&c->d_crowd_count, sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->d_fine_count, sizeof(uint32_t)));
  }
  static const bool prof = getenv("FLIMO_PROF_INSERT") != nullptr;     // developer timing of the stages (each ends synchronised)
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tp0 = prof ? now() : 0.0;
  // 1. the list of crowded cells (device: a bit per cell + append list; host: their centres in world coordinates)
  uint32_t listed = 0;
  const size_t ncells_main = (size_t)g.nx * g.ny * g.nz;
  if (relayout || !c->crowd_box_valid || !new_pts) {
    const size_t words = (ncells_main + 31) / 32;
    if (words > c->crowd_bits_cap) {
      (void)hipFree(c->d_crowd_bits);
      c->d_crowd_bits = nullptr; c->crowd_bits_cap = 0;
      HIPCHK(c, hipMalloc(&c->d_crowd_bits, (words + words / 4 + 64) * sizeof(uint32_t)));
      c->crowd_bits_cap = words + words / 4 + 64;
    }
    c->crowd_cells.clear();
    c->crowd_listed = 0;
    HIPCHK(c, crowded_list_all(c->stream, g, c->fine_threshold, c->d_crowd_bits, (int4*)c->d_crowd_list, CROWD_CAP,
                               c->d_crowd_count, &listed, c->scratch));
    c->crowd_box_valid = true;
  } else {
    // same geometry as at the last look: only the cells of the points merged since can have become crowded
    HIPCHK(c, crowded_list_points(c->stream, new_pts, n_new, g,
                                  c->fine_threshold, c->d_crowd_bits, (int4*)c->d_crowd_list, CROWD_CAP, c->d_crowd_count, &listed, c->scratch));
  }
  if (listed > CROWD_CAP) return FLIMO_OK;                      // crowded all over: no region to speak of
  if (listed > c->crowd_listed) {
    const size_t k = listed - c->crowd_listed;
    std::vector<int> buf(4 * k);
    HIPCHK(c, hipMemcpy(buf.data(), (const int4*)c->d_crowd_list + c->crowd_listed, k * sizeof(int4), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < k; i++) c->crowd_cells.push_back({buf[4 * i], buf[4 * i + 1], buf[4 * i + 2]});
    c->crowd_listed = listed;
  }
  if (c->crowd_cells.empty()) return FLIMO_OK;
  // 2. the box of the crowded cells around the sensor (the pose of the last inserted scan): that is where the next scans' dense
  //    near-range queries land; crowded cells left behind along the path, or a lone one on a far wall, do not stretch it
  int box[7] = {INT_MAX, INT_MAX, INT_MAX, -1, -1, -1, 0};
  {
    const float R = c->fine_radius * g.inv_cell;                // cells
    const float scx = (c->fine_center[0] - g.ox) * g.inv_cell - (float)g.six, scy = (c->fine_center[1] - g.oy) * g.inv_cell - (float)g.siy;
    for (const auto& q : c->crowd_cells) {
      if (c->have_fine_center) {
        const float dx = (float)q[0] + 0.5f - scx, dy = (float)q[1] + 0.5f - scy;
        if (dx * dx + dy * dy > R * R) continue;
      }
      for (int a = 0; a < 3; a++) { box[a] = std::min(box[a], q[a]); box[3 + a] = std::max(box[3 + a], q[a]); }
      box[6]++;
    }
  }
  if (box[6] <= 0) return FLIMO_OK;
  // region copied completely: the crowded cells and one cell around them
  float lo[3], hi[3];
  const float o[3] = {g.ox, g.oy, g.oz};
  const int si3[3] = {g.six, g.siy, g.siz};
  for (int a = 0; a < 3; a++) { lo[a] = o[a] + (float)(box[a] - 1 + si3[a]) * g.cell; hi[a] = o[a] + (float)(box[3 + a] + 2 + si3[a]) * g.cell; }
  const float cf = g.cell / (float)c->fine_div, inv_f = 1.0f / cf;
  const float of[3] = {lo[0] - cf, lo[1] - cf, lo[2] - cf};             // one fine cell of margin below the region
  int nf[3];
  for (int a = 0; a < 3; a++) nf[a] = (int)floorf((hi[a] - of[a]) * inv_f) + 3;
  const double ncf = (double)nf[0] * nf[1] * nf[2];
  if (ncf > 6.4e7) return FLIMO_OK;      // crowded all over: not a region
  // the copies: the region's cells (clipped to the grid) are contiguous ranges of the cell-sorted map
  const int gdim[3] = {g.nx, g.ny, g.nz};
  int c0[3], c1[3];
  for (int a = 0; a < 3; a++) { c0[a] = std::max(box[a] - 1, 0); c1[a] = std::min(box[3 + a] + 1, gdim[a] - 1); }
  uint32_t m = 0;
  HIPCHK(c, map_box_count(c->stream, g, c0, c1, c->d_fine_count, &m, c->scratch));
  if (m == 0 || m < c->fine_min_points) return FLIMO_OK;
  const double tp1 = prof ? now() : 0.0;
  if (m > c->fine_pts_cap) {
    (void)hipFree(c->d_fine_tmp); (void)hipFree(c->d_fine_pts);
    c->d_fine_tmp = c->d_fine_pts = nullptr; c->fine_pts_cap = 0;
    const size_t cap = (size_t)m + m / 2 + 4096;
    HIPCHK(c, hipMalloc(&c->d_fine_tmp, cap * sizeof(float4)));
    HIPCHK(c, hipMalloc(&c->d_fine_pts, cap * sizeof(float4)));
    c->fine_pts_cap = cap;
  }
  HIPCHK(c, map_box_copy(c->stream, g, c0, c1, c->d_fine_tmp, c->scratch));
  {
    GridView fg{};
    fg.ox = of[0]; fg.oy = of[1]; fg.oz = of[2]; fg.inv_cell = inv_f; fg.cell = cf;
    fg.nx = nf[0]; fg.ny = nf[1]; fg.nz = nf[2]; fg.xs = 1; fg.nxf = nf[0]; fg.nxs = nf[0] + 1;
    HIPCHK(c, map_build_grid(c->stream, c->d_fine_tmp, m, c->d_fine_pts, c->fine_pts_cap, false, c->fine_idx, c->fine_pts_cap, fg, c->scratch));
  }
  if (prof) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    fprintf(stderr, "[flimo fine] crowded cells + count %.1f us, copy + sort + tables of %d x %d x %d cells %.1f us (%u points)\n",
            (tp1 - tp0) * 1e6, nf[0], nf[1], nf[2], (now() - tp1) * 1e6, m);
  }
  GridView& f = c->fine;
  f.pts = c->d_fine_pts;
  f.ox = of[0]; f.oy = of[1]; f.oz = of[2]; f.six = f.siy = f.siz = 0;
  f.inv_cell = inv_f; f.cell = cf;
  f.nx = nf[0]; f.ny = nf[1]; f.nz = nf[2];
  f.n_pts = m; f.xs = 1; f.nxf = nf[0]; f.nxs = nf[0] + 1;
  index_view(c->fine_idx, f);
  // a query may be settled here when its fine 3x3x3 block lies inside [lo, hi): fine cell 1 starts at lo, so the query's own cell
  // is >= 2; on the upper side one more cell of safety against the rounding of (hi - of) * inv_f
  for (int a = 0; a < 3; a++) {
    c->fine_qlo[a] = 2;
    c->fine_qhi[a] = (int)floorf((hi[a] - of[a]) * inv_f) - 3;
  }
  c->fine_valid = c->fine_qhi[0] >= 2 && c->fine_qhi[1] >= 2 && c->fine_qhi[2] >= 2;
  if (c->fine_valid) c->fine_builds++;
  return FLIMO_OK;
}

// Brings the cell-sorted copy of the map and its tables up to date with d_map_raw[0 .. map_n).
//  * insert in place: the geometry still covers the map box and only points were appended since the last build -> they go into
//    their rows (map_merge_grid: only the rows that receive points are touched);
//  * the map outgrew the geometry -> the grid grows (same cell origin, the corner moves by whole tiles: index_regrid), then the
//    same insert -- nothing is sorted;
//  * full layout: first build, another tile shape or cell size, a full point array or tile pool -> a sort of the whole map.
// (the k points appended since the index was last brought up to date go into their rows: map_merge_grid)
static int merge_appended(flimo_ctx* c) {
  const size_t n_old = c->grid.n_pts, k = c->map_n - n_old;
  if (k == 0) { c->grid_valid = true; return FLIMO_OK; }
  c->grid_valid = false;
  // room in the tile pool for what this insert may need (the count of tiles in use is the last insert's, read behind its wait):
  // an eighth of the tiles in use, 32 at least (a layout leaves half of them, 64 at least) -- a pool that runs out anyway has the
  // map laid out afresh
  {
    const uint32_t used = std::max(c->idx.tiles_used, c->scratch.mail_host ? c->scratch.mail_host[MAIL_TILES + 2] : 0u);
    const uint32_t want_free = std::max(32u, used / 8);
    if (c->idx.cap_tiles < used + want_free) {
      HIPCHK(c, index_grow_pool(c->stream, c->idx, used + std::max(128u, used / 2)));
      index_view(c->idx, c->grid);
      c->pool_grows++;
    }
  }
  HIPCHK(c, map_merge_grid(c->stream, c->d_map_sorted, c->sorted_cap, c->d_map_raw + n_old, k, c->idx, c->grid, c->scratch));
  // (no wait here: everything that reads the index is queued behind this on the same stream; map_add_device ends synchronised
  //  and looks whether the point array or the tile pool ran out)
  c->grid.n_pts = (uint32_t)c->map_n;
  c->grid_valid = true;
  c->grid_merges++;
  return update_fine_grid(c, false, c->d_map_raw + n_old, k);
}
static int rebuild_grid(flimo_ctx* c) {
  ctx_enter(c);
  if (c->map_n == 0) { c->grid_valid = false; return FLIMO_OK; }
  const float* bb = c->bb;    // tracked on the host while points are appended (no reduction kernel)
  if (c->sorted_follows) {
    // the raw buffer grew (by a quarter each time): the cell-sorted copy moves into a larger array as it is -- its rows stay where
    // they are -- and the escape pool grows with it; nothing is sorted
    c->sorted_follows = false;
    if (c->d_map_sorted && c->grid_valid && !c->test_tight_array) {
      const size_t ncap = std::min<size_t>(3 * c->map_cap + 65536, 0x7fffffffull);
      float4* np = nullptr;
      HIPCHK(c, hipMalloc(&np, ncap * sizeof(float4)));
      HIPCHK(c, hipMemcpyAsync(np, c->d_map_sorted, std::min(ncap, c->sorted_cap) * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
      const size_t ovf_words = (c->map_cap / 16 + 64) * 8;
      uint32_t* no = nullptr;
      if (ovf_words > c->idx.ovf_cap) {
        HIPCHK(c, hipMalloc(&no, ovf_words * sizeof(uint32_t)));
        HIPCHK(c, hipMemcpyAsync(no, c->idx.ovf, c->idx.ovf_cap * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
      }
      HIPCHK(c, hipStreamSynchronize(c->stream));
      (void)hipFree(c->d_map_sorted);
      c->d_map_sorted = np; c->sorted_cap = ncap;
      if (no) { (void)hipFree(c->idx.ovf); c->idx.ovf = no; c->idx.ovf_cap = ovf_words; }
      c->grid.pts = c->d_map_sorted;
      index_view(c->idx, c->grid);
    } else if (c->d_map_sorted) {
      (void)hipFree(c->d_map_sorted); c->d_map_sorted = nullptr; c->grid_valid = false;
    }
  }
  const bool index_live = c->grid_valid && c->d_map_sorted && !c->full_rebuild && !c->force_full && c->grid.n_pts > 0 &&
                          c->map_n >= c->grid.n_pts && (c->map_n - c->grid.n_pts) <= c->grid.n_pts;
  if (index_live && grid_covers(c->grid, bb)) return merge_appended(c);
  c->grid_valid = false;
  float cell = c->map_cfg.cell_size > 0.f ? c->map_cfg.cell_size : 0.5f;
  // slack: a side the map has grown beyond since the last layout moves out by max(8 cells, 1/8 of the extent)
  float W[6];
  for (int a = 0; a < 3; a++) { W[a] = bb[a]; W[3 + a] = bb[3 + a]; }
  // (the low x side keeps the first segment of x-tile 0 empty: grid_covers)
  const float low_x_pad = (float)(grid_low_x_cells(1) + 2) * cell;      // (whatever column factor the layout ends up with)
  if (!c->have_gbox) {
    // first layout: a little room on every side (8 cells horizontally, 4 vertically), so that the first scans inserted into a
    // pre-built map -- their noise alone pokes through an exact box -- are merged instead of re-sorting the whole map
    for (int a = 0; a < 3; a++) {
      float pad = (a < 2 ? 8.0f : 4.0f) * cell;
      if (a == 0) pad = std::max(pad, low_x_pad);
      W[a] = bb[a] - pad; W[3 + a] = bb[3 + a] + pad;
    }
  } else {
    for (int a = 0; a < 3; a++) {
      const float pad = std::max(8.0f * cell, 0.125f * (bb[3 + a] - bb[a]));
      // (a low side moves when the box has passed it -- or, with the geometry it was laid out for, no longer keeps its margin)
      const bool low_moves = bb[a] < c->gbox[a] || (cell == c->grid.cell && !grid_side_low_ok(c->grid, bb, a));
      W[a] = low_moves ? bb[a] - (a == 0 ? std::max(pad, low_x_pad) : pad) : c->gbox[a];
      W[3 + a] = (bb[3 + a] > c->gbox[3 + a]) ? bb[3 + a] + pad : c->gbox[3 + a];
    }
  }
  // ---- the map outgrew its grid, the index is up to date: the grid grows, nothing is sorted.  The origin of the cells is fixed
  //      (GridView); the corner moves down by whole tiles, the extents follow; the rows stay where they are and new rows start
  //      empty (index_regrid).  Not when the larger grid needs another tile shape or 32-bit column keys no longer do. ----
  if (index_live && c->have_origin && cell == c->grid.cell) {
    const GridView O = c->grid;
    GridView N = O;
    const float o[3] = {O.ox, O.oy, O.oz};
    const int si_old[3] = {O.six, O.siy, O.siz}, n_old[3] = {O.nx, O.ny, O.nz};
    const int T[3] = {std::max(1, (8 << O.ts) / O.xs), 1 << O.ty, 1 << O.tz};      // cells per tile
    int si[3], n[3];
    for (int a = 0; a < 3; a++) {
      const int lo = (int)floorf((W[a] - 0.5f * cell - o[a]) * O.inv_cell), hi = (int)floorf((W[3 + a] - o[a]) * O.inv_cell);
      si[a] = si_old[a];
      if (lo < si_old[a]) si[a] = si_old[a] - ((si_old[a] - lo + T[a] - 1) / T[a]) * T[a];
      n[a] = std::max(hi - si[a] + 2, n_old[a] + (si_old[a] - si[a]));
    }
    N.six = si[0]; N.siy = si[1]; N.siz = si[2];
    N.nx = n[0]; N.ny = n[1]; N.nz = n[2];
    N.nxf = N.nx * N.xs; N.nxs = N.nxf + 1;
    // (column keys are 64-bit; what stays 32-bit: a row's number and a column's number within its row)
    const bool fits = (double)N.ny * (double)N.nz < 4.0e9 && (double)N.nx * N.xs < 2.0e9;
    hipError_t e = fits ? index_regrid(c->stream, c->idx, O, N) : hipErrorInvalidValue;
    if (e == hipSuccess) {
      // the crowded cells listed so far (second level): the same cells under the new corner
      const int d[3] = {si_old[0] - si[0], si_old[1] - si[1], si_old[2] - si[2]};
      for (auto& q : c->crowd_cells) for (int a = 0; a < 3; a++) q[a] += d[a];
      if (c->crowd_box_valid && c->d_crowd_list && c->crowd_cells.size() == c->crowd_listed) {
        // ... and the device's bit per cell, re-made from the list (a look at every cell of a 20M-point map's grid is milliseconds)
        const size_t words = ((size_t)N.nx * N.ny * N.nz + 31) / 32;
        if (words > c->crowd_bits_cap) {
          (void)hipFree(c->d_crowd_bits);
          c->d_crowd_bits = nullptr; c->crowd_bits_cap = 0;
          HIPCHK(c, hipMalloc(&c->d_crowd_bits, (words + words / 4 + 64) * sizeof(uint32_t)));
          c->crowd_bits_cap = words + words / 4 + 64;
        }
        std::vector<int> buf(4 * c->crowd_cells.size() + 4);
        for (size_t i = 0; i < c->crowd_cells.size(); i++) { buf[4 * i] = c->crowd_cells[i][0]; buf[4 * i + 1] = c->crowd_cells[i][1]; buf[4 * i + 2] = c->crowd_cells[i][2]; buf[4 * i + 3] = 0; }
        if (!c->crowd_cells.empty())
          HIPCHK(c, hipMemcpyAsync(c->d_crowd_list, buf.data(), c->crowd_cells.size() * sizeof(int4), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, crowded_relist(c->stream, (const int4*)c->d_crowd_list, (uint32_t)c->crowd_cells.size(), N.nx, N.ny, N.nz, c->d_crowd_bits, c->d_crowd_count));
        HIPCHK(c, hipStreamSynchronize(c->stream));              // (buf goes out of scope)
      } else {
        c->crowd_box_valid = false;                                // (every cell is looked at again at the next index update)
      }
      c->grid = N;
      for (int a = 0; a < 6; a++) c->gbox[a] = W[a];
      c->grid_regrids++;
      return merge_appended(c);
    }
    if (e != hipErrorInvalidValue) HIPCHK(c, e);
  }
  if (getenv("FLIMO_PROF_INSERT"))
    fprintf(stderr, "[flimo index] full layout: sorted %p full_rebuild %d force %d n_pts %u map_n %zu  box [%g %g %g | %g %g %g] grid o (%g %g %g) n (%d %d %d)\n",
            (void*)c->d_map_sorted, (int)c->full_rebuild, (int)c->force_full, c->grid.n_pts, c->map_n,
            bb[0], bb[1], bb[2], bb[3], bb[4], bb[5], c->grid.ox, c->grid.oy, c->grid.oz, c->grid.nx, c->grid.ny, c->grid.nz);
  // ---- full layout: sort everything.  The origin stays once it is set (same cell size); the corner is a cell shift. ----
  GridView G{};
  int xs = c->xslabs;
  auto layout = [&](const float* box) {
    G.cell = cell; G.inv_cell = 1.0f / cell;
    if (c->have_origin && cell == c->grid.cell) { G.ox = c->grid.ox; G.oy = c->grid.oy; G.oz = c->grid.oz; }
    else { G.ox = box[0] - 0.5f * cell; G.oy = box[1] - 0.5f * cell; G.oz = box[2] - 0.5f * cell; }
    // same float expression as the kernels: floor((p - o) * inv) - shift
    G.six = (int)floorf((box[0] - 0.5f * cell - G.ox) * G.inv_cell);
    G.siy = (int)floorf((box[1] - 0.5f * cell - G.oy) * G.inv_cell);
    G.siz = (int)floorf((box[2] - 0.5f * cell - G.oz) * G.inv_cell);
    G.nx = (int)floorf((box[3] - G.ox) * G.inv_cell) - G.six + 2;
    G.ny = (int)floorf((box[4] - G.oy) * G.inv_cell) - G.siy + 2;
    G.nz = (int)floorf((box[5] - G.oz) * G.inv_cell) - G.siz + 2;
    // column keys are 64-bit (round 5: a grid of kilometres keeps its cell); what stays 32-bit is a row's number and a column's
    // number within its row -- beyond that (a box of thousands of kilometres) the fine x columns are given up, then the cell grows
    for (xs = c->xslabs; xs >= 1; xs >>= 1)
      if ((double)G.ny * (double)G.nz < 4.0e9 && (double)G.nx * xs < 2.0e9) return true;
    xs = 1;
    return false;
  };
  if (!layout(W)) {
    for (int a = 0; a < 6; a++) W[a] = bb[a];
    W[0] = bb[0] - 10.0f * cell;
    while (!layout(W)) { cell *= 2.0f; W[0] = bb[0] - 10.0f * cell; }   // (rows and columns-per-row within 32 bits; the first segment of x-tile 0 stays empty)
  }
  G.xs = xs; G.nxf = G.nx * xs; G.nxs = G.nxf + 1;
  for (int a = 0; a < 6; a++) c->gbox[a] = W[a];
  c->have_gbox = true;
  if (!c->d_map_sorted) {
    // the rows of the cell-sorted copy are not packed (a build leaves half a row's length of room behind every row, an insert
    // moves a row that outgrows its room to the end): three times the raw capacity (freed whenever that grows); an insert that
    // finds it full has the map laid out afresh
    c->sorted_cap = std::min<size_t>(3 * c->map_cap + 65536, 0x7fffffffull);      // (positions are 31-bit: bit 31 of an entry marks an escape)
    HIPCHK(c, hipMalloc(&c->d_map_sorted, c->sorted_cap * sizeof(float4)));
  }
  if (c->test_tight_array) {
    (void)hipFree(c->d_map_sorted); c->d_map_sorted = nullptr;
    c->sorted_cap = c->map_n + 4096;
    HIPCHK(c, hipMalloc(&c->d_map_sorted, c->sorted_cap * sizeof(float4)));
  }
  HIPCHK(c, map_build_grid(c->stream, c->d_map_raw, c->map_n, c->d_map_sorted, c->sorted_cap, c->row_slack, c->idx, c->map_cap, G, c->scratch));
  c->scratch.mail_host[MAIL_TILES + 2] = c->scratch.mail_host[MAIL_TILES + 3] = c->scratch.mail_host[MAIL_ROWS] = 0u;      // (the merges' words: tiles taken, "ran out", "array full")
  HIPCHK(c, hipStreamSynchronize(c->stream));
  G.pts = c->d_map_sorted;
  c->grid = G;
  index_view(c->idx, c->grid);
  c->grid.n_pts = (uint32_t)c->map_n;
  c->grid_valid = true;
  c->have_origin = true;
  c->force_full = false;
  c->grid_builds++;
  return update_fine_grid(c, true);
}

// Debug: sort the whole map again with the CURRENT geometry into temporary buffers and compare the result with the
// incrementally maintained index, by meaning (every row's points in order, every row's position at every column).
// stats = {inserts in place, full layouts} so far.
extern "C" int flimo_map_grid_selfcheck(flimo_ctx* c, uint64_t* mismatches, uint64_t stats[2]) {
  if (!c || !mismatches) return FLIMO_ERR_INVALID;
  if (stats) { stats[0] = c->grid_merges; stats[1] = c->grid_builds; }
  *mismatches = 0;
  if (!c->grid_valid) return FLIMO_OK;
  ctx_enter(c);
  const GridView& g = c->grid;
  const size_t n = c->map_n;
  if (g.n_pts != n) { *mismatches = 1; return FLIMO_OK; }
  struct Tmp {
    float4* pts = nullptr; unsigned long long* diff = nullptr; IndexTables idx;
    ~Tmp() { (void)hipFree(pts); (void)hipFree(diff); index_free(idx); }
  } t;
  HIPCHK(c, hipMalloc(&t.pts, (n + 1) * sizeof(float4)));      // (position 0 of a cell-sorted array is nobody's)
  HIPCHK(c, hipMalloc(&t.diff, sizeof(unsigned long long)));
  HIPCHK(c, hipMemsetAsync(t.diff, 0, sizeof(unsigned long long), c->stream));
  HIPCHK(c, map_build_grid(c->stream, c->d_map_raw, n, t.pts, n + 1, false, t.idx, n, g, c->scratch));
  // by meaning: every row holds the same points in the same order, every row's position at every column agrees with its own
  // array (the rows of the maintained copy are not packed; escapes and tiles take their numbers in arrival order)
  GridView ref = g;
  ref.pts = t.pts;
  index_view(t.idx, ref);
  HIPCHK(c, index_compare(c->stream, ref, g, t.diff));
  unsigned long long d = 0;
  HIPCHK(c, hipMemcpyAsync(&d, t.diff, sizeof(d), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *mismatches += d;
  return FLIMO_OK;
}

// append `n` packed float4 points (host, NaN-free, already filtered by the insert rule)
static int map_append_host(flimo_ctx* c, const float4* pts, size_t n) {
  if (n == 0) return FLIMO_OK;
  if (c->map_n + n > 0x7fff0000ull) return fail(c, FLIMO_ERR_TOO_LARGE, "map would exceed 2^31 points");
  const size_t old_cap = c->map_cap;
  int rc = ensure_dev(c, c->d_map_raw, c->map_cap, c->map_n + n, true, c->map_n);
  if (rc) return rc;
  if (c->map_cap != old_cap && c->d_map_sorted) c->sorted_follows = true;      // (the cell-sorted copy and the escape pool follow at the next index update: rebuild_grid)
  HIPCHK(c, hipMemcpyAsync(c->d_map_raw + c->map_n, pts, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->map_n += n;
  return FLIMO_OK;
}

// Octree::update for a batch that already lives on the device (m points, NaNs allowed): the device
// book decides keep / drop, the kept points are appended in batch order, the grid is rebuilt.
static int map_add_device(flimo_ctx* c, const float4* d_pts, size_t m, double stamp) {
  if (m == 0) return FLIMO_OK;
  if (c->map_n + m > 0x7fff0000ull) return fail(c, FLIMO_ERR_TOO_LARGE, "map would exceed 2^31 points");
  static const bool prof = getenv("FLIMO_PROF_INSERT") != nullptr;     // developer timing of the insert stages
  auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = prof ? now() : 0.0;
  float bb[6];
  bool any = false;
  HIPCHK(c, batch_bbox(c->stream, d_pts, m, c->scratch, bb, &any));
  const double t1 = prof ? now() : 0.0;
  double t2 = t1, t3 = t1;
  if (any) {
    const size_t old_cap = c->map_cap;
    int rc = ensure_dev(c, c->d_map_raw, c->map_cap, c->map_n + m, true, c->map_n);
    if (rc) return rc;
    if (c->map_cap != old_cap && c->d_map_sorted) c->sorted_follows = true;      // (the cell-sorted copy and the escape pool follow at the next index update: rebuild_grid)
    int kept = 0;
    if (!c->gbook.active)        // first batch: Octree::initialize on the device
      HIPCHK(c, c->gbook.init(c->stream, d_pts, (int)m, bb, c->d_map_raw, &kept, c->map_cfg.min_extent, c->map_cfg.downsample != 0,
                              c->scratch));
    else
      HIPCHK(c, c->gbook.update(c->stream, d_pts, (int)m, bb, c->d_map_raw, (int)c->map_n, &kept, c->scratch));
    t2 = prof ? now() : 0.0;
    c->map_n += (size_t)kept;
    if (kept > 0) {
      // the kept points lie inside the batch box: a superset box only makes the dense grid a little larger
      for (int a = 0; a < 3; a++) { if (bb[a] < c->bb[a]) c->bb[a] = bb[a]; if (bb[3 + a] > c->bb[3 + a]) c->bb[3 + a] = bb[3 + a]; }
    }
    // also when nothing was kept: a batch that made the raw buffer grow released the sorted copy, and a map without its
    // index would answer every later pass with "no matches"
    if (kept > 0 || (!c->grid_valid && c->map_n > 0)) {
      rc = rebuild_grid(c);
      if (rc) return rc;
    }
    // ends synchronised: the book's node count (left in the mail words) is taken over behind the same wait
    HIPCHK(c, c->gbook.finish(c->stream, c->scratch));
    if (c->grid_valid && index_merge_overflow(c->scratch)) {
      // the merge needed more tiles than the pool had room for: lay the index out afresh (sized by what the map needs now)
      c->force_full = true;
      c->index_overflows++;
      rc = rebuild_grid(c);
      if (rc) return rc;
    }
    c->idx.tiles_used = std::max(c->idx.tiles_used, c->scratch.mail_host[MAIL_TILES + 2]);
    t3 = prof ? now() : 0.0;
  }
  if (prof) fprintf(stderr, "[flimo insert] bbox %.0f us, book %.0f us, grid %.0f us (batch %zu, map %zu)\n", t1 - t0, t2 - t1, t3 - t2, m, c->map_n);
  c->map_last_time = stamp;
  return FLIMO_OK;
}

// hand the host-built tree of the first batch to the device book
static int gbook_import(flimo_ctx* c) {
  std::vector<float> c4;
  std::vector<int> child, cnt;
  int root = -1;
  insert_book_export(c->book, c4, child, cnt, &root);
  if (root < 0) return FLIMO_OK;
  HIPCHK(c, c->gbook.import_host(c->stream, c4, child, cnt, root, c->d_map_raw, (int)c->map_n, c->map_cfg.min_extent,
                                 c->map_cfg.downsample != 0));
  insert_book_clear(c->book);      // the device copy is the book from here on
  return FLIMO_OK;
}

extern "C" int flimo_map_add(flimo_ctx* c, const float* xyz, size_t n, size_t stride_bytes, double stamp) {
  if (!c) return FLIMO_ERR_INVALID;
  if (n < 1) return FLIMO_OK;                         // Mapper::add: `if(pc->points.size() < 1) return;`
  if (!xyz || stride_bytes < 12) return fail(c, FLIMO_ERR_INVALID, "bad xyz/stride");
  ctx_enter(c);
  int rc = ensure_stage(c, n * sizeof(float4));
  if (rc) return rc;
  float4* st = (float4*)c->h_stage;
  // Octree::initialize / Octree::update on the device (flimo_gbook.hip)
  const unsigned char* b = (const unsigned char*)xyz;
  for (size_t i = 0; i < n; i++) {
    const float* p = (const float*)(b + i * stride_bytes);
    st[i].x = p[0]; st[i].y = p[1]; st[i].z = p[2]; st[i].w = 0.f;
  }
  rc = ensure_dev(c, c->d_batch, c->batch_cap, n, false, 0);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->d_batch, st, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
  return map_add_device(c, c->d_batch, n, stamp);
}

extern "C" int flimo_map_points(flimo_ctx* c, float* out, size_t cap, size_t* n) {
  if (!c || !n) return FLIMO_ERR_INVALID;
  *n = c->map_n;
  if (!out || cap == 0 || c->map_n == 0) return FLIMO_OK;
  if (!c->grid_valid) { int rc0 = rebuild_grid(c); if (rc0) return rc0; }
  if (!c->grid_valid) return fail(c, FLIMO_ERR_NOMAP, "map index not built");
  ctx_enter(c);
  const size_t m = std::min(cap, c->map_n);
  int rc = ensure_stage(c, m * sizeof(float4));
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->h_stage, c->d_map_raw, m * sizeof(float4), hipMemcpyDeviceToHost, c->stream));      // (insertion order)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const float4* s = (const float4*)c->h_stage;
  for (size_t i = 0; i < m; i++) { out[3 * i] = s[i].x; out[3 * i + 1] = s[i].y; out[3 * i + 2] = s[i].z; }
  return FLIMO_OK;
}

// ---- kNN --------------------------------------------------------------------------------------
extern "C" int flimo_knn(flimo_ctx* c, const float* q, size_t nq, int k, int32_t* idx, float* sqd, int32_t* cnt) {
  if (!c || !q || !idx || !sqd || !cnt) return FLIMO_ERR_INVALID;
  if (k < 1 || k > 5) return fail(c, FLIMO_ERR_UNSUPPORTED, "k must be in 1..5");
  if (nq == 0) return FLIMO_OK;
  if (!c->grid_valid && c->map_n > 0) { int rc0 = rebuild_grid(c); if (rc0) return rc0; }
  if (!c->grid_valid) {                       // Octree::knn with root_ == nullptr returns nothing
    for (size_t i = 0; i < nq; i++) cnt[i] = 0;
    for (size_t i = 0; i < nq * (size_t)k; i++) { idx[i] = -1; sqd[i] = 0.f; }
    return FLIMO_OK;
  }
  ctx_enter(c);
  // scratch of this call, released on every exit path
  struct Scratch {
    float* q = nullptr; int32_t* idx = nullptr; float* sqd = nullptr; int32_t* cnt = nullptr;
    ~Scratch() { (void)hipFree(q); (void)hipFree(idx); (void)hipFree(sqd); (void)hipFree(cnt); }
  } d;
  HIPCHK(c, hipMalloc(&d.q, nq * 3 * sizeof(float)));
  HIPCHK(c, hipMalloc(&d.idx, nq * k * sizeof(int32_t)));
  HIPCHK(c, hipMalloc(&d.sqd, nq * k * sizeof(float)));
  HIPCHK(c, hipMalloc(&d.cnt, nq * sizeof(int32_t)));
  HIPCHK(c, hipMemcpyAsync(d.q, q, nq * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  // (no gate like a pass's MAX_DIST_PLANE: Octree::knn answers from anywhere.  A few rings of cells near the map, then -- round 6 --
  //  the best-first search over the tiles that exist (knn_far_kernel): a query hundreds of metres from every point of a sparse map
  //  of kilometres costs a look at the directory and at the nearest tiles, not (2r+1)^2 row lookups per ring)
  launch_knn(c->stream, c->grid, d.q, (int)nq, k, 1 << 29, d.idx, d.sqd, d.cnt);
  if (c->ties && c->gbook.active) {    // exactly tied distances: the reference's first-met choice (device copy of its octree)
    const BookView book{c->gbook.node_c, c->gbook.node_child, c->gbook.node_cnt, c->gbook.root, c->d_tie_settled};
    launch_knn_tie(c->stream, c->grid, book, d.q, (int)nq, k, d.idx, d.sqd, d.cnt);
  }
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(idx, d.idx, nq * k * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(sqd, d.sqd, nq * k * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(cnt, d.cnt, nq * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return FLIMO_OK;
}

// ---- scan -------------------------------------------------------------------------------------
// the Morton-ordered raw sweep (deskew's input) has a capacity of its own: flimo_scan_adopt exchanges it between two contexts
static int ensure_raw_sorted(flimo_ctx* c, size_t n) {
  if (n <= c->raw_sorted_cap) return FLIMO_OK;
  const size_t cap = n + n / 4 + 1024;
  float4* rs = nullptr;
  double* ts = nullptr;
  HIPCHK(c, hipMalloc(&rs, cap * sizeof(float4)));
  HIPCHK(c, hipMalloc(&ts, cap * sizeof(double)));
  (void)hipFree(c->d_raw_sorted); (void)hipFree(c->d_t_sorted);
  c->d_raw_sorted = rs; c->d_t_sorted = ts; c->raw_sorted_cap = cap;
  return FLIMO_OK;
}
static int ensure_scan(flimo_ctx* c, size_t n) {
  { const int rc = ensure_raw_sorted(c, n); if (rc) return rc; }
  if (n <= c->scan_cap) return FLIMO_OK;
  const size_t cap = n + n / 4 + 1024;
  float4 *a = nullptr, *b = nullptr, *w = nullptr, *so = nullptr;
  double* t = nullptr;
  void* nb = nullptr;
  int* wl = nullptr;
  double* fp = nullptr;
  const size_t fpn = (size_t)fit_blocks((int)cap) * 256;
  {
    int* tlst = nullptr;
    HIPCHK(c, hipMalloc(&tlst, cap * sizeof(int)));
    (void)hipFree(c->d_tie_list);
    c->d_tie_list = tlst; c->tie_cap = cap;
  }
  {
    // (a one-launch pass spreads a small scan over up to FUSED_SPREAD_SLOTS query slots)
    const size_t f2n = (size_t)std::max(std::max(fit2_blocks((int)cap), fused_blocks((int)cap)), FUSED_SPREAD_SLOTS / 128 + 8) * FIT_LIVE_PAD;
    double* f2 = nullptr;
    HIPCHK(c, hipMalloc(&f2, f2n * sizeof(double)));
    (void)hipFree(c->d_fit2_partials);
    c->d_fit2_partials = f2; c->fit2_partials_cap = f2n;
  }
  HIPCHK(c, hipMalloc(&so, cap * sizeof(float4)));
  HIPCHK(c, hipMalloc(&nb, cap * nbr_rec_size()));
  HIPCHK(c, hipMemsetAsync(nb, 0, cap * nbr_rec_size(), c->stream));      // flag 0 everywhere: no record is ever read uninitialised
  HIPCHK(c, hipMalloc(&wl, (cap + 8192) * wl_entry_size()));   // + slack: every widening wave prefetches its first slot
  HIPCHK(c, hipMalloc(&fp, fpn * sizeof(double)));
  (void)hipFree(c->d_scan_sorted); (void)hipFree(c->d_nbr); (void)hipFree(c->d_wl); (void)hipFree(c->d_fit_partials);
  c->d_scan_sorted = so; c->d_nbr = nb; c->d_wl = wl; c->d_fit_partials = fp; c->fit_partials_cap = fpn;
  HIPCHK(c, hipMalloc(&a, cap * sizeof(float4)));
  HIPCHK(c, hipMalloc(&b, cap * sizeof(float4)));
  HIPCHK(c, hipMalloc(&w, cap * sizeof(float4)));
  HIPCHK(c, hipMalloc(&t, cap * sizeof(double)));
  (void)hipFree(c->d_scan); (void)hipFree(c->d_scan_raw); (void)hipFree(c->d_scan_world); (void)hipFree(c->d_scan_t);
  c->d_scan = a; c->d_scan_raw = b; c->d_scan_world = w; c->d_scan_t = t;
  c->scan_cap = cap;
  c->scan_n = 0; c->sorted_n = 0; c->raw_n = 0; c->prev.valid = 0;
  return FLIMO_OK;
}

static int ensure_recs(flimo_ctx* c, size_t n) {
  if (n <= c->rec_cap) return FLIMO_OK;
  const size_t cap = n + n / 4 + 1024;
  (void)hipFree(c->d_recs); (void)hipFree(c->d_dbg);
  c->d_recs = nullptr; c->d_dbg = nullptr; c->rec_cap = 0;
  HIPCHK(c, hipMalloc(&c->d_recs, cap * sizeof(Rec16)));
  HIPCHK(c, hipMalloc(&c->d_dbg, cap * sizeof(RecDbg)));
  c->rec_cap = cap;
  return FLIMO_OK;
}

static int upload_points(flimo_ctx* c, const float* xyz, size_t n, size_t stride_bytes, float4* dst) {
  int rc = ensure_stage(c, n * sizeof(float4));
  if (rc) return rc;
  float4* st = (float4*)c->h_stage;
  const unsigned char* b = (const unsigned char*)xyz;
  for (size_t i = 0; i < n; i++) {
    const float* p = (const float*)(b + i * stride_bytes);
    st[i].x = p[0]; st[i].y = p[1]; st[i].z = p[2]; st[i].w = (stride_bytes >= 16) ? p[3] : 0.f;
  }
  HIPCHK(c, hipMemcpyAsync(dst, st, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return FLIMO_OK;
}

// Runs a deskew that is still pending (see flimo_ctx::deskew_pending) as a dispatch of its own.
static int flush_deskew(flimo_ctx* c) {
  if (!c->deskew_pending) return FLIMO_OK;
  c->deskew_pending = false;
  const DeskewArgs& a = c->deskew_args;
  launch_deskew(c->stream, a.raw, a.t, (int)c->deskew_n, a.frames, a.nf, a.mats, a.out_sorted, a.out_orig, a.t_offset);
  HIPCHK(c, hipGetLastError());
  return FLIMO_OK;
}

extern "C" int flimo_scan_set(flimo_ctx* c, const float* xyz, size_t n, size_t stride_bytes) {
  if (!c) return FLIMO_ERR_INVALID;
  if (n > 0 && (!xyz || stride_bytes < 12)) return fail(c, FLIMO_ERR_INVALID, "bad xyz/stride");
  if (n > 0x7fff0000ull) return fail(c, FLIMO_ERR_TOO_LARGE, "scan too large");
  ctx_enter(c);
  c->deskew_pending = false;                 // the scan it belonged to is replaced
  int rc = ensure_scan(c, n);
  if (rc) return rc;
  if (n) {
    rc = upload_points(c, xyz, n, stride_bytes, c->d_scan);
    if (rc) return rc;
    HIPCHK(c, sort_scan(c->stream, c->d_scan, n, c->d_scan_sorted, c->scratch));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->scan_n = n; c->sorted_n = n; c->prev.valid = 0;
  return FLIMO_OK;
}

extern "C" size_t flimo_scan_size(const flimo_ctx* c) { return c ? c->scan_n : 0; }

static int download_xyz(flimo_ctx* c, const float4* src, size_t m, float* out) {
  int rc = ensure_stage(c, m * sizeof(float4));
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(c->h_stage, src, m * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const float4* s = (const float4*)c->h_stage;
  for (size_t i = 0; i < m; i++) { out[3 * i] = s[i].x; out[3 * i + 1] = s[i].y; out[3 * i + 2] = s[i].z; }
  return FLIMO_OK;
}

extern "C" int flimo_scan_get(flimo_ctx* c, float* out, size_t cap, size_t* n) {
  if (!c || !n) return FLIMO_ERR_INVALID;
  *n = c->scan_n;
  if (!out || cap == 0 || c->scan_n == 0) return FLIMO_OK;
  ctx_enter(c);
  { const int rcf = flush_deskew(c); if (rcf) return rcf; }
  return download_xyz(c, c->d_scan, std::min(cap, c->scan_n), out);
}

// ---- voxel filter on the resident scan (Localizer.cpp:313-321) ---------------------------------
extern "C" int flimo_scan_voxel_filter(flimo_ctx* c, float leaf, size_t* n_out) {
  if (!c || !(leaf > 0.f)) return FLIMO_ERR_INVALID;
  if (n_out) *n_out = c->scan_n;
  if (c->scan_n == 0) return FLIMO_OK;
  ctx_enter(c);
  size_t m = 0;
  bool pass = false;
  { const int rcf = flush_deskew(c); if (rcf) return rcf; }
  // d_scan_world is free scratch at this point of the scan life cycle
  HIPCHK(c, voxel_grid(c->stream, c->d_scan, c->scan_n, leaf, c->d_scan_world, &m, &pass, c->scratch));
  if (!pass) {
    std::swap(c->d_scan, c->d_scan_world);
    c->scan_n = m; c->sorted_n = m; c->prev.valid = 0;
  }
  if (c->scan_n) HIPCHK(c, sort_scan(c->stream, c->d_scan, c->scan_n, c->d_scan_sorted, c->scratch));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_out) *n_out = c->scan_n;
  return FLIMO_OK;
}

// ---- deskew -----------------------------------------------------------------------------------
extern "C" int flimo_raw_scan_set(flimo_ctx* c, const float* xyz, size_t n, size_t stride_bytes, const double* t) {
  if (!c) return FLIMO_ERR_INVALID;
  if (n > 0 && (!xyz || !t || stride_bytes < 12)) return fail(c, FLIMO_ERR_INVALID, "bad xyz/t/stride");
  ctx_enter(c);
  c->deskew_pending = false;                 // a deskew never run belonged to the scan this one replaces
  int rc = ensure_scan(c, n);
  if (rc) return rc;
  if (n) {
    rc = upload_points(c, xyz, n, stride_bytes, c->d_scan_raw);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_scan_t, t, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    // spatial order is fixed here, once per scan: deskew is a near-rigid motion, so the Morton order of
    // the raw LiDAR-frame points stays spatially coherent for the per-pass kernels
    HIPCHK(c, sort_scan(c->stream, c->d_scan_raw, n, c->d_raw_sorted, c->scratch, c->d_scan_t, c->d_t_sorted));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  c->raw_n = n; c->order_n = n;
  c->resident_t_offset = 0.0;
  return FLIMO_OK;
}

extern "C" int flimo_upload_stage(flimo_ctx* c, size_t bytes, void** host_ptr) {
  if (!c || !host_ptr) return FLIMO_ERR_INVALID;
  *host_ptr = nullptr;
  ctx_enter(c);
  const int rc = ensure_stage(c, bytes);
  if (rc) return rc;
  *host_ptr = c->h_stage;
  return FLIMO_OK;
}

extern "C" int flimo_raw_scan_filter_set(flimo_ctx* c, const void* points32, size_t n, const flimo_filter_cfg* cfg, size_t* n_kept,
                                         double* last_stamp, int* nan_stamp) {
  int tied = 0;
  return flimo_raw_scan_filter_order_set(c, points32, n, cfg, 0, n_kept, last_stamp, nan_stamp, &tied);
}

extern "C" int flimo_raw_scan_filter_order_set(flimo_ctx* c, const void* points32, size_t n, const flimo_filter_cfg* cfg, int time_order,
                                               size_t* n_kept, double* last_stamp, int* nan_stamp, int* tied) {
  if (!c || !cfg || !n_kept || !last_stamp || !nan_stamp || !tied || (n > 0 && !points32)) return FLIMO_ERR_INVALID;
  if (cfg->time_kind < 0 || cfg->time_kind > 3) return fail(c, FLIMO_ERR_INVALID, "time_kind must be 0..3");
  if (cfg->rate_active && cfg->rate_value < 1) return fail(c, FLIMO_ERR_INVALID, "rate_value must be >= 1");
  if (n > 0x7fff0000ull) return fail(c, FLIMO_ERR_TOO_LARGE, "scan too large");
  *n_kept = 0; *last_stamp = 0.0; *nan_stamp = 0; *tied = 0;
  c->deskew_pending = false;                 // a deskew never run belonged to the scan this one replaces
  ctx_enter(c);
  int rc = ensure_scan(c, n);
  if (rc) return rc;
  c->raw_n = 0; c->order_n = 0; c->resident_t_offset = 0.0; c->raw_time_ordered = false;
  if (n == 0) return FLIMO_OK;
  if (n > c->raw32_cap) {
    (void)hipFree(c->d_raw32);
    c->d_raw32 = nullptr; c->raw32_cap = 0;
    const size_t cap = n + n / 4 + 1024;
    HIPCHK(c, hipMalloc(&c->d_raw32, cap * 32));
    c->raw32_cap = cap;
  }
  if (!c->d_filt_ext) {
    HIPCHK(c, hipMalloc(&c->d_filt_ext, 4 * sizeof(unsigned long long)));
    HIPCHK(c, hipHostMalloc((void**)&c->h_filt_ext, 4 * sizeof(unsigned long long), hipHostMallocDefault));
  }
  if (time_order && n > c->tkey_cap) {
    (void)hipFree(c->d_tkey[0]); (void)hipFree(c->d_tkey[1]); (void)hipFree(c->d_tperm); (void)hipFree(c->d_t_tmp);
    c->d_tkey[0] = c->d_tkey[1] = nullptr; c->d_tperm = nullptr; c->d_t_tmp = nullptr; c->tkey_cap = 0;
    const size_t cap = c->scan_cap;                           // (>= n after ensure_scan)
    HIPCHK(c, hipMalloc(&c->d_tkey[0], cap * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&c->d_tkey[1], cap * sizeof(unsigned long long)));
    HIPCHK(c, hipMalloc(&c->d_tperm, cap * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc(&c->d_t_tmp, cap * sizeof(double)));
    c->tkey_cap = cap;
  }
  // time_order bit 2: the records are 16 bytes -- x, y, z and the 32-bit time word (OUSTER's t, VELODYNE's time)
  const size_t rec = (time_order & 4) ? 16 : 32;
  if (rec == 16 && cfg->time_kind > 1) return fail(c, FLIMO_ERR_INVALID, "16-byte records carry a 32-bit time word: time_kind 0 or 1");
  const bool staged = points32 == c->h_stage && n * rec <= c->stage_cap;     // the caller filled flimo_upload_stage's buffer itself
  if (!staged) {
    rc = ensure_stage(c, n * rec);
    if (rc) return rc;
  }
  if (staged) {
    HIPCHK(c, hipMemcpyAsync(c->d_raw32, c->h_stage, n * rec, hipMemcpyHostToDevice, c->stream));
  } else {
    // pageable caller memory -> pinned stage -> HBM in chunks: the DMA of a chunk runs while the next one is staged
    const size_t total = n * rec, chunk = 512u << 10;
    for (size_t o = 0; o < total; o += chunk) {
      const size_t len = std::min(chunk, total - o);
      memcpy((char*)c->h_stage + o, (const char*)points32 + o, len);
      HIPCHK(c, hipMemcpyAsync((char*)c->d_raw32 + o, (const char*)c->h_stage + o, len, hipMemcpyHostToDevice, c->stream));
    }
  }
  FilterParams F;
  F.crop = cfg->crop_active ? 1 : 0;
  for (int a = 0; a < 3; a++) { F.mn[a] = cfg->crop_active ? cfg->crop_min[a] : 0.f; F.mx[a] = cfg->crop_active ? cfg->crop_max[a] : 0.f; }
  F.dist = cfg->dist_active ? 1 : 0; F.min_dist = cfg->min_dist;
  F.rate_on = cfg->rate_active ? 1 : 0; F.rate = cfg->rate_value;
  F.kind = cfg->time_kind; F.eos = cfg->end_of_sweep ? 1 : 0; F.sweep_ref = cfg->sweep_ref_time;
  if (cfg->fov_active) {
    // The FoV verdict is |atan2f(y, x)| < angle with the HOST libm's rounding (Localizer.cpp:873-876).  The device restates glibc's
    // fdlibm routine (up to 2.40); a libm that rounds atan2f differently (glibc >= 2.41: CORE-MATH) would keep another set at the
    // FoV's edge than the host path and the reference on that host do.  Checked once per context: 8 192 argument pairs (signed
    // zeros, denormals, huge ratios, the octant boundaries, random bit patterns); on any difference the device declines and the
    // caller's host front end filters the sweep.
    if (c->fov_check == 0) {
      std::vector<float> yx, got;
      auto push = [&](float y, float x) { yx.push_back(y); yx.push_back(x); };
      const float sp[] = {0.f, -0.f, 1.f, -1.f, 1e-38f, -1e-38f, 1e-45f, 3.4e38f, -3.4e38f, 0.4375f, 0.6875f, 1.1875f, 2.4375f, 1e-10f, 1e10f, 0.5f, 2.f};
      for (float y : sp) for (float x : sp) push(y, x);
      uint64_t st = 0x9E3779B97F4A7C15ull;
      auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
      while (yx.size() < 2 * 8192) {
        uint32_t a = rnd(), b = rnd();
        float y, x;
        if (yx.size() < 2 * 4096) { y = ((int)(a % 200001u) - 100000) * 1e-3f; x = ((int)(b % 200001u) - 100000) * 1e-3f; }     // sensor-like coordinates
        else { memcpy(&y, &a, 4); memcpy(&x, &b, 4); if (!(y == y) || !(x == x)) continue; }                                        // any finite or infinite pattern
        push(y, x);
      }
      const int npair = (int)(yx.size() / 2);
      got.resize(npair);
      HIPCHK(c, atan2f_probe(c->stream, yx.data(), npair, got.data()));
      int bad = 0;
      for (int i = 0; i < npair; i++) {
        const float ref = std::atan2(yx[2 * i], yx[2 * i + 1]);
        if (memcmp(&ref, &got[i], 4) != 0 && !(ref != ref && got[i] != got[i])) bad++;
      }
      c->fov_check = bad == 0 ? 1 : -1;
      c->fov_check_bad = bad;
    }
    if (c->fov_check < 0)
      return fail(c, FLIMO_ERR_UNSUPPORTED, "this host's atan2f differs from the device's restatement on %d of 8192 argument pairs: the FoV filter stays on the host front end", c->fov_check_bad);
  }
  F.fov = cfg->fov_active ? 1 : 0; F.fov_angle = cfg->fov_angle;
  const bool keep_order = (time_order & 2) != 0;      // bit 1: no spatial order (a voxel filter follows and re-orders the scan)
  const bool ties_as_they_come = (time_order & 8) != 0;      // bit 3: equal stamps keep their arrival order (see below)
  time_order &= 1;
  HIPCHK(c, filter_raw_scan(c->stream, c->d_raw32, n, F, c->d_scan_raw, c->d_scan_t, c->d_filt_ext, c->scratch, time_order ? c->d_tkey[0] : nullptr, (int)rec));
  HIPCHK(c, filter_raw_scan_result(c->stream, c->scratch, c->d_filt_ext, c->wait_timeout_ms, c->h_filt_ext));
  const size_t m = (size_t)c->h_filt_ext[1];
  *n_kept = m;
  *nan_stamp = c->h_filt_ext[2] ? 1 : 0;
  if (m == 0 || *nan_stamp) return FLIMO_OK;
  // stamp of the point the reference's sort puts last, from its key (the same float / double expressions as the kernel)
  const bool desc = cfg->end_of_sweep && cfg->time_kind <= 1;
  unsigned long long key = c->h_filt_ext[0];
  if (desc) key = ~key;
  double t;
  if (cfg->time_kind == 0) {
    const float tf = (float)(uint32_t)key * 1e-9f;
    t = cfg->end_of_sweep ? cfg->sweep_ref_time - (double)tf : cfg->sweep_ref_time + (double)tf;
  } else if (cfg->time_kind == 1) {
    uint32_t b = (uint32_t)key;
    b = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b;
    float v;
    memcpy(&v, &b, 4);
    t = cfg->end_of_sweep ? cfg->sweep_ref_time - (double)v : cfg->sweep_ref_time + (double)v;
  } else {
    unsigned long long b = key;
    b = (b & 0x8000000000000000ull) ? (b & 0x7fffffffffffffffull) : ~b;
    double v;
    memcpy(&v, &b, 8);
    t = (cfg->time_kind == 2) ? v : v * (double)1e-9f;
  }
  *last_stamp = t;
  if (time_order) {
    // the kept points into the reference's time order (d_scan_world / d_t_tmp are free at this point of the scan's life)
    HIPCHK(c, time_order_raw(c->stream, c->d_scan_raw, c->d_scan_t, m, c->d_tkey[0], c->d_tkey[1], c->d_scan_world, c->d_t_tmp, c->d_tperm,
                             c->d_filt_ext, c->scratch));
    bool is_tied = false;
    HIPCHK(c, time_order_raw_tied(c->stream, c->scratch, c->d_filt_ext, c->wait_timeout_ms, &is_tied));
    // Equal stamps (a spinning sensor: all rings of a column share one): the reference's std::partial_sort_copy leaves them in the
    // order its heap happens to produce, which only the host routine reproduces move for move.  That order is observable only as
    // ulp-level voxel centroids and as which of several equally stamped points a cap cuts off -- far below north_star's 1e-4 m.
    // A caller that says so (bit 3) gets the stable order of the radix sort -- arrival order among equal stamps -- and the sweep
    // stays on the device; without the bit the call reports `tied` and leaves the sweep to the caller's host routine.
    if (is_tied) { *tied = 1; if (!ties_as_they_come) return FLIMO_OK; }
    std::swap(c->d_scan_raw, c->d_scan_world);
    std::swap(c->d_scan_t, c->d_t_tmp);
    c->raw_time_ordered = true;
  }
  if (keep_order) HIPCHK(c, index_scan(c->stream, c->d_scan_raw, m, c->d_raw_sorted, c->d_scan_t, c->d_t_sorted));
  else HIPCHK(c, sort_scan(c->stream, c->d_scan_raw, m, c->d_raw_sorted, c->scratch, c->d_scan_t, c->d_t_sorted));
  c->raw_n = m; c->order_n = m;
  return FLIMO_OK;
}

// The resident raw sweep of `src` becomes `dst`'s (both on one GPU): the Morton-ordered points and stamps that
// flimo_raw_scan_filter_order_set left in src -- the deskew's input, the only part of a raw sweep read after the input stage --
// change owner (the two contexts exchange the buffers: no copy), dst's stream is ordered behind src's queued work.  Lets a caller run
// the input stage of sweep k + 1 on a context of its own while dst's stream still carries sweep k's map insert.
extern "C" int flimo_scan_adopt(flimo_ctx* dst, flimo_ctx* src) {
  if (!dst || !src || dst == src) return FLIMO_ERR_INVALID;
  if (dst->device != src->device) return fail(dst, FLIMO_ERR_INVALID, "flimo_scan_adopt: the two contexts are on different devices");
  ctx_enter(dst);
  const size_t m = src->raw_n;
  dst->deskew_pending = false;
  { const int rc = ensure_scan(dst, m); if (rc) return rc; }
  dst->raw_n = 0; dst->order_n = 0; dst->resident_t_offset = 0.0; dst->raw_time_ordered = false;
  if (m == 0) return FLIMO_OK;
  if (!src->adopt_ev) HIPCHK(dst, hipEventCreateWithFlags(&src->adopt_ev, hipEventDisableTiming));
  if (!dst->adopt_ev) HIPCHK(dst, hipEventCreateWithFlags(&dst->adopt_ev, hipEventDisableTiming));
  // dst reads what src's queued sort writes; src's next sweep writes what dst's queued work may still read
  HIPCHK(dst, hipEventRecord(src->adopt_ev, src->stream));
  HIPCHK(dst, hipStreamWaitEvent(dst->stream, src->adopt_ev, 0));
  HIPCHK(dst, hipEventRecord(dst->adopt_ev, dst->stream));
  HIPCHK(dst, hipStreamWaitEvent(src->stream, dst->adopt_ev, 0));
  std::swap(dst->d_raw_sorted, src->d_raw_sorted);
  std::swap(dst->d_t_sorted, src->d_t_sorted);
  std::swap(dst->raw_sorted_cap, src->raw_sorted_cap);
  dst->raw_n = m;
  dst->resident_t_offset = src->resident_t_offset;
  dst->raw_time_ordered = src->raw_time_ordered;
  src->raw_n = 0;                                   // src keeps the time order (flimo_raw_scan_order), not the sweep
  return FLIMO_OK;
}

// time rank -> position among the kept points of the last flimo_raw_scan_filter_order_set (identity when the sweep was left in
// arrival order)
extern "C" int flimo_raw_scan_order(flimo_ctx* c, uint32_t* order_out, size_t cap, size_t* n) {
  if (!c || !n) return FLIMO_ERR_INVALID;
  *n = c->order_n;
  if (!order_out || cap == 0 || c->order_n == 0) return FLIMO_OK;
  const size_t m = std::min(cap, c->order_n);
  if (!c->raw_time_ordered) { for (size_t i = 0; i < m; i++) order_out[i] = (uint32_t)i; return FLIMO_OK; }
  ctx_enter(c);
  HIPCHK(c, hipMemcpyAsync(order_out, c->d_tperm, m * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return FLIMO_OK;
}

extern "C" int flimo_deskew_resident(flimo_ctx* c, const flimo_frame* frames, size_t nf, const float L2B[16],
                                     const double last_x26[26]) {
  return flimo_deskew_resident_offset(c, frames, nf, L2B, last_x26, c ? c->resident_t_offset : 0.0);
}

extern "C" int flimo_deskew_resident_offset(flimo_ctx* c, const flimo_frame* frames, size_t nf, const float L2B[16],
                                            const double last_x26[26], double t_offset) {
  if (!c || !frames || nf == 0 || !L2B || !last_x26) return FLIMO_ERR_INVALID;
  c->resident_t_offset = t_offset;        // a later re-registration of the same resident sweep (benchmark) uses the same stamps
  ctx_enter(c);
  static_assert(sizeof(flimo_frame) == 112, "flimo_frame layout");
  if (dev_frame_size() != sizeof(flimo_frame)) return fail(c, FLIMO_ERR_INVALID, "frame layout mismatch");
  const size_t n = c->raw_n;
  if (n == 0) { c->scan_n = 0; c->sorted_n = 0; c->prev.valid = 0; return FLIMO_OK; }
  const size_t fbytes = nf * sizeof(flimo_frame);
  const size_t total = fbytes + 32 * sizeof(float);
  if (c->d_frames_fg && total <= FRAMES_FG_SLOT) {
    // The host stores frames + matrices straight into (fine-grained) device memory: no copy launch and no dispatch boundary ahead of
    // the pass the deskew rides on.  Two slots, alternating: a pass that still reads the last sweep's is not disturbed.
    alignas(64) char buf[FRAMES_FG_SLOT];
    memcpy(buf, frames, fbytes);
    float* m = reinterpret_cast<float*>(buf + fbytes);
    memcpy(m, L2B, 16 * sizeof(float));
    {
      const float p[3] = {(float)last_x26[0], (float)last_x26[1], (float)last_x26[2]};
      const float q[4] = {(float)last_x26[3], (float)last_x26[4], (float)last_x26[5], (float)last_x26[6]};
      se3_inv_from(q, p, m + 16);                     // last_state.get_RT_inv()
    }
    char* dst = c->d_frames_fg + (size_t)c->frames_slot * FRAMES_FG_SLOT;
    c->frames_slot ^= 1;
    memcpy(dst, buf, (total + 7) & ~(size_t)7);
    _mm_sfence();
    c->deskew_args = DeskewArgs{c->d_raw_sorted, c->d_t_sorted, dst, (int)nf, reinterpret_cast<const float*>(dst + fbytes),
                                t_offset, c->d_scan_sorted, c->d_scan, 1, (int)(total / 4)};
    c->deskew_n = n;
    c->deskew_pending = true;
    c->scan_n = n; c->sorted_n = n; c->prev.valid = 0;
    return FLIMO_OK;
  }
  // frames + the two 4x4 matrices go through one pinned staging copy
  if (total > c->frames_cap) {
    (void)hipFree(c->d_frames);
    c->d_frames = nullptr;
    HIPCHK(c, hipMalloc(&c->d_frames, total * 2));
    c->frames_cap = total * 2;
  }
  if (total > c->h_frames_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->async_deskews = 0;
    for (int k = 0; k < 2; k++) { if (c->h_frames[k]) (void)hipHostFree(c->h_frames[k]); c->h_frames[k] = nullptr; }
    c->h_frames_cap = 0;
    for (int k = 0; k < 2; k++) HIPCHK(c, hipHostMalloc(&c->h_frames[k], total * 2, hipHostMallocDefault));
    c->h_frames_cap = total * 2;
  }
  if (c->async_deskews >= 2) {                       // both staging slots may still be read by queued copies
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->async_deskews = 0;
  }
  char* hs = (char*)c->h_frames[c->frames_slot];
  c->frames_slot ^= 1;
  memcpy(hs, frames, fbytes);
  float* m = (float*)(hs + fbytes);
  memcpy(m, L2B, 16 * sizeof(float));
  {
    const float p[3] = {(float)last_x26[0], (float)last_x26[1], (float)last_x26[2]};
    const float q[4] = {(float)last_x26[3], (float)last_x26[4], (float)last_x26[5], (float)last_x26[6]};
    se3_inv_from(q, p, m + 16);                       // last_state.get_RT_inv()
  }
  // stream-ordered: everything that consumes the deskewed scan is queued behind this on the same stream
  HIPCHK(c, hipMemcpyAsync(c->d_frames, hs, total, hipMemcpyHostToDevice, c->stream));
  c->deskew_args = DeskewArgs{c->d_raw_sorted, c->d_t_sorted, c->d_frames, (int)nf, (const float*)((const char*)c->d_frames + fbytes),
                              t_offset, c->d_scan_sorted, c->d_scan, 1, 0};
  c->deskew_n = n;
  c->deskew_pending = true;
  c->async_deskews++;
  c->scan_n = n; c->sorted_n = n; c->prev.valid = 0;
  return FLIMO_OK;
}

extern "C" int flimo_deskew(flimo_ctx* c, const float* xyz, size_t n, size_t stride_bytes, const double* t,
                            const flimo_frame* frames, size_t nf, const float L2B[16], const double last_x26[26]) {
  int rc = flimo_raw_scan_set(c, xyz, n, stride_bytes, t);
  if (rc) return rc;
  return flimo_deskew_resident(c, frames, nf, L2B, last_x26);
}

// ---- measurement pass -------------------------------------------------------------------------
extern "C" int flimo_set_timing(flimo_ctx* c, int level) {
  if (!c) return FLIMO_ERR_INVALID;
  c->timing = level > 0 ? 1 : 0;          // (level 2 -- events around every stage, synchronous -- went with the A/B kernels it timed)
  if (c->timing == 1 && !c->chain_ev_made) {      // the chained update's per-pass events: made here, not inside a timed update
    ctx_enter(c);
    for (int i = 0; i < CH_MAX_PASSES; i++) for (int k = 0; k < 8; k++) HIPCHK(c, hipEventCreate(&c->chain_ev[i][k]));
    c->chain_ev_made = true;
  }
  return FLIMO_OK;
}
extern "C" int flimo_set_timing_deferred(flimo_ctx* c, int on) {
  if (!c) return FLIMO_ERR_INVALID;
  ctx_enter(c);
  if (on && !c->ev_ring_made) {
    for (int r = 0; r < flimo_ctx::EV_RING; r++) for (int i = 0; i < 6; i++) HIPCHK(c, hipEventCreate(&c->ev_ring[r][i]));
    c->ev_ring_made = true;
  }
  if (!on && c->ev_pending) { const int rc = timing_drain(c); if (rc) return rc; }
  c->timing_deferred = on != 0;
  return FLIMO_OK;
}
extern "C" unsigned long long flimo_pass_count(const flimo_ctx* c) { return c ? c->pass_seq : 0ull; }
extern "C" unsigned long long flimo_fused_pass_count(const flimo_ctx* c) { return c ? c->fused_passes : 0ull; }
extern "C" int flimo_fine_stats(const flimo_ctx* c, unsigned long long out[4]) {
  if (!c || !out) return FLIMO_ERR_INVALID;
  out[0] = c->fine_valid ? 1 : 0; out[1] = c->fine_valid ? c->fine.n_pts : 0; out[2] = c->fine_builds; out[3] = c->fine_passes;
  return FLIMO_OK;
}
extern "C" int flimo_map_index_bytes(const flimo_ctx* c, uint64_t out[6]) {
  if (!c || !out) return FLIMO_ERR_INVALID;
  out[0] = (uint64_t)c->map_n * sizeof(float4);
  // (the tiles that exist + the zero tile, the directory, xstart, the rows' room and first positions, the escape pool as allocated)
  auto tables = [](const GridView& g, const IndexTables& T) {
    const uint64_t rows = ((uint64_t)g.ny + 2 * GRID_PAD) * ((uint64_t)g.nz + 2 * GRID_PAD);
    return (uint64_t)T.tiles_used * grid_tile_entries(g.ts, g.ty, g.tz) * 8ull + (uint64_t)GRID_DIR_MAX * 2ull +
           (uint64_t)(grid_xstart_size(g.ny, g.nz, g.ntx) + T.ovf_cap) * 4ull + rows * 4ull + (uint64_t)g.ny * g.nz * 4ull;
  };
  // the cell-sorted array AS ALLOCATED: three times the raw buffer's capacity, its rows keep room behind their last point
  out[5] = c->d_map_sorted ? (uint64_t)c->sorted_cap * sizeof(float4) : 0ull;
  out[1] = c->grid_valid ? tables(c->grid, c->idx) : 0ull;
  out[2] = c->fine_valid ? (uint64_t)c->fine.n_pts * sizeof(float4) + tables(c->fine, c->fine_idx) : 0ull;
  out[3] = c->grid_valid ? c->idx.tiles_used : 0ull;
  out[4] = c->index_overflows;
  return FLIMO_OK;
}
extern "C" int flimo_tie_stats(flimo_ctx* c, unsigned long long out[2]) {
  if (!c || !out) return FLIMO_ERR_INVALID;
  out[0] = c->tie_redos; out[1] = c->tie_queries;
  // + the queries settled inside the reducing launches (counted on the device).  NOT a passive getter: the context is entered (a
  // pass queued ahead of its pose is told to leave) and the stream drained for the read -- the owner's thread only.
  unsigned long long dev = 0;
  ctx_enter(c);
  if (c->d_tie_settled && hipMemcpy(&dev, c->d_tie_settled, sizeof(dev), hipMemcpyDeviceToHost) == hipSuccess) out[1] += dev;
  return FLIMO_OK;
}
extern "C" int flimo_set_timing_stride(flimo_ctx* c, int every) { if (!c || every < 1) return FLIMO_ERR_INVALID; c->timing_stride = every; return FLIMO_OK; }
extern "C" int flimo_set_debug_records(flimo_ctx* c, int on) { if (!c) return FLIMO_ERR_INVALID; c->debug_recs = on != 0; return FLIMO_OK; }
extern "C" int flimo_last_kernel_ms(const flimo_ctx* c, float* knn_ms, float* widen_ms, float* fit_ms) {
  if (!c) return FLIMO_ERR_INVALID;
  if (knn_ms) *knn_ms = c->last_knn_ms;
  if (widen_ms) *widen_ms = c->last_widen_ms;
  if (fit_ms) *fit_ms = c->last_fit_ms;
  return FLIMO_OK;
}
extern "C" int flimo_last_widen_count(const flimo_ctx* c) { return c ? c->last_widen_count : 0; }
extern "C" int flimo_last_stragglers(const flimo_ctx* c) { return c ? c->last_stragglers : -1; }
extern "C" int flimo_stragglers_by_pass(const flimo_ctx* c, int out[4]) {
  if (!c || !out) return FLIMO_ERR_INVALID;
  for (int i = 0; i < 4; i++) out[i] = c->stragglers_hist[i];
  return FLIMO_OK;
}
extern "C" int flimo_timing_totals(flimo_ctx* c, double* knn_ms, double* widen_ms, double* fit_ms, long long* passes,
                                   long long* queries, int reset) {
  if (!c) return FLIMO_ERR_INVALID;
  if (c->ev_pending) { ctx_enter(c); const int rc = timing_drain(c); if (rc) return rc; }
  if (knn_ms) *knn_ms = c->tot_knn_ms;
  if (widen_ms) *widen_ms = c->tot_widen_ms;
  if (fit_ms) *fit_ms = c->tot_fit_ms;
  if (passes) *passes = c->tot_passes;
  if (queries) *queries = c->tot_queries;
  if (reset) { c->tot_knn_ms = c->tot_widen_ms = c->tot_fit_ms = 0.0; c->tot_passes = 0; c->tot_queries = 0; }
  return FLIMO_OK;
}
// developer / benchmark A/B: negative leaves a switch as it is
extern "C" int flimo_set_path_switches(flimo_ctx* c, int tail, int fuse) {
  if (!c) return FLIMO_ERR_INVALID;
  if (tail >= 0) c->tail = tail != 0;
  if (fuse >= 0) c->fuse = fuse != 0;
  return FLIMO_OK;
}
extern "C" int flimo_timing_split(flimo_ctx* c, double out[6], int reset) {
  if (!c || !out) return FLIMO_ERR_INVALID;
  if (c->ev_pending) { ctx_enter(c); const int rc = timing_drain(c); if (rc) return rc; }
  out[0] = c->split_fused_ms; out[1] = (double)c->split_fused_n;
  out[2] = c->split_knn_ms; out[3] = c->split_widen_ms; out[4] = c->split_fit_ms; out[5] = (double)c->split_sep_n;
  if (reset) { c->split_fused_ms = c->split_knn_ms = c->split_widen_ms = c->split_fit_ms = 0.0; c->split_fused_n = c->split_sep_n = 0; }
  return FLIMO_OK;
}
extern "C" double flimo_last_candidates_per_query(const flimo_ctx* c) { return c ? c->last_cand_per_query : 0.0; }

static int gate_rings(const flimo_ctx* c, double max_dist_plane) {
  // smallest ring r with ((r - margin) * cell)^2 >= MAX_DIST_PLANE: beyond it close_enough() fails anyway
  const double cell = c->grid.cell;
  const int maxdim = std::max(c->grid.nx, std::max(c->grid.ny, c->grid.nz));
  const double margin = 1.0e-3 + 4.0e-7 * maxdim;
  const double need = sqrt(std::max(max_dist_plane, 0.0)) * (1.0 + 1e-5) / cell + margin;
  int r = (int)ceil(need);
  return std::max(r, 1);
}

// Waits until every granule of the pass `want` has arrived in mapped host memory (spinning: a pass lasts tens of microseconds).
// Bounded by wall clock (FLIMO_DEFAULT_WAIT_MS, flimo_set_wait_timeout_ms): a launch that never publishes -- a device fault, a
// hung kernel -- ends the call with an error instead of hanging the caller, which holds the filter's mutex.
// A wait that ran out leaves its launches in flight: they still own the pass buffers (neighbour records, tickets, the chain's prior
// in mapped memory).  An event behind them tells when they are gone; until then every new pass is refused with the same error.
static int abandon_wait(flimo_ctx* c, const char* what, unsigned long long id) {
  if (!c->timeout_ev) (void)hipEventCreateWithFlags(&c->timeout_ev, hipEventDisableTiming);
  if (c->timeout_ev && hipEventRecord(c->timeout_ev, c->stream) == hipSuccess) c->timeout_pending = true;
  c->prev.valid = 0;
  return fail(c, FLIMO_ERR_TIMEOUT, "%s %llu did not publish its result within %d ms (kernels still running)", what, id, c->wait_timeout_ms);
}
static inline bool same_match_cfg(const flimo_match_cfg& a, const flimo_match_cfg& b) {      // (field by field: the struct has padding)
  return a.NUM_MATCH_POINTS == b.NUM_MATCH_POINTS && a.MAX_NUM_MATCHES == b.MAX_NUM_MATCHES && a.MAX_NUM_PC2MATCH == b.MAX_NUM_PC2MATCH &&
         a.MAX_DIST_PLANE == b.MAX_DIST_PLANE && a.PLANE_THRESHOLD == b.PLANE_THRESHOLD && a.estimate_extrinsics == b.estimate_extrinsics;
}
// ---- pipelined host loop: the pass queued ahead of the algebra --------------------------------------------------------------
static inline double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// the waiting pass is told to leave (its workgroups poll head.epoch); cheap: one posted store when there is one, nothing otherwise
static inline void cancel_prelaunch(flimo_ctx* c) {
  if (!c->pre.active) return;
  c->pre.active = false;
  c->pipe_cancelled++;
  __atomic_store_n(&c->d_pipe_head->epoch, c->pre.end_code, __ATOMIC_RELEASE);
  _mm_sfence();
}
// every entry point that queues work on the context's stream passes here first: nothing may line up behind a pass that waits
static inline void ctx_enter(flimo_ctx* c) {
  (void)hipSetDevice(c->device);
  cancel_prelaunch(c);
}
// the constants of the waiting pass: pose, the last pass's pose (its pruning bound's reference), then -- behind a store fence --
// the word its workgroups poll
static inline void publish_prelaunch(flimo_ctx* c, const PoseMats& P, const float prev_RT[16], unsigned long long seq) {
  ChainHead h;
  memset(&h, 0, sizeof(h));
  h.pose = P;
  memcpy(h.prev_RT, prev_RT, sizeof(h.prev_RT));
  h.status = 0;
  // (everything but the epoch word, in 8-byte stores; then the epoch)
  const size_t words8 = offsetof(ChainHead, status) / 8;
  static_assert(offsetof(ChainHead, status) % 8 == 0 && offsetof(ChainHead, epoch) == offsetof(ChainHead, status) + 4, "ChainHead layout");
  volatile unsigned long long* dst = reinterpret_cast<volatile unsigned long long*>(c->d_pipe_head);
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&h);
  for (size_t i = 0; i < words8; i++) dst[i] = src[i];
  __atomic_store_n(&c->d_pipe_head->status, 0, __ATOMIC_RELAXED);
  _mm_sfence();
  __atomic_store_n(&c->d_pipe_head->epoch, ch_epoch_of(seq), __ATOMIC_RELEASE);
  _mm_sfence();
  c->pre.active = false;
  c->pipe_published++;
}

// Which layout runs is the caller's choice, never a measurement's: the two layouts agree to 1e-15 per pass, not bit for bit (libm
// against the device's sin / cos / acos), so a choice made from a launch round trip measured at context creation -- as round 4
// did, with a threshold inside the spread of the hosts it ran on -- made the filter's bits depend on timing noise.  Mode 0 is the
// host loop (pipelined where the device maps fine-grained memory); the chain runs when it is asked for (flimo_set_update_mode(2),
// FLIMO_HOST_UPDATE=0): a host whose launch round trip is beyond ~16 us is better off with it.
static inline bool auto_host_update(const flimo_ctx*) { return true; }

static int check_abandoned(flimo_ctx* c) {
  if (!c->timeout_pending) return FLIMO_OK;
  hipError_t q = hipEventQuery(c->timeout_ev);
  if (q == hipErrorNotReady) return fail(c, FLIMO_ERR_TIMEOUT, "the launches of an earlier pass whose wait ran out are still running");
  c->timeout_pending = false;
  if (q != hipSuccess) return fail(c, FLIMO_ERR_HIP, "an abandoned pass failed: %s", hipGetErrorString(q));
  return FLIMO_OK;
}
static bool tags_complete(const flimo_ctx* c, unsigned long long want) {
  const volatile unsigned long long* t0 = reinterpret_cast<const volatile unsigned long long*>(c->h_granules);
  if (t0[2 * FIT_LIVE + 1] != want || t0[2 * (FIT_LIVE + 1) + 1] != want) return false;
  for (int g = 0; g < FIT_GROUPS; g++) {
    const volatile unsigned long long* tags = reinterpret_cast<const volatile unsigned long long*>(c->h_granules + (size_t)g * FIT_LIVE_PAD * 2);
    for (int k = 0; k < FIT_LIVE; k++)
      if (tags[2 * k + 1] != want) return false;
  }
  return true;
}
// left_ms >= 0: the pass was published to a launch that had been waiting for it (pipelined host loop).  Such a launch leaves as a whole
// when its wait ran out before the publish reached it (chain_enter's decision word) -- nothing of the pass has run then, no ticket
// was touched: FLIMO_PASS_LEFT tells the caller to launch the pass the usual way.  Its own bound is a little over the launch's.
constexpr int FLIMO_PASS_LEFT = 1000;
static int wait_tags(flimo_ctx* c, unsigned long long want, int left_ms = -1) {
  const volatile unsigned long long* t0 = reinterpret_cast<const volatile unsigned long long*>(c->h_granules);
  unsigned long long spins = 0;
  double deadline = 0.0;
  auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  if (c->wait_timeout_ms == 0) return abandon_wait(c, "pass", want);      // "do not wait at all" (the error path's test hook)
  const int bound_ms = left_ms >= 0 ? std::min(left_ms, c->wait_timeout_ms) : c->wait_timeout_ms;
  for (;;) {
    if (t0[2 * (FIT_LIVE + 1) + 1] == want && tags_complete(c, want)) return FLIMO_OK;
    _mm_pause();
    if ((++spins & 0x3fffull) != 0) continue;            // look at the clock every 16k polls (about 0.1 ms)
    // (a pass published to a waiting launch: the launch's own verdict, read through the BAR -- "left" ends the wait at once)
    if (left_ms >= 0 && c->d_pipe_head &&
        __atomic_load_n(&c->d_pipe_head->decision, __ATOMIC_ACQUIRE) == ((ch_epoch_of(want) & 0x3fffffffu) | 0x80000000u)) return FLIMO_PASS_LEFT;
    const double t = now_s();
    if (deadline == 0.0) { deadline = t + 1e-3 * (double)bound_ms; continue; }
    if (t < deadline) continue;
    // out of time: what does the stream say?
    const hipError_t q = hipStreamQuery(c->stream);
    if (q == hipErrorNotReady && left_ms >= 0 && bound_ms < c->wait_timeout_ms) { left_ms = -1; deadline = t + 1e-3 * (double)(c->wait_timeout_ms - bound_ms); continue; }   // (still running: the usual bound)
    if (q == hipErrorNotReady) return abandon_wait(c, "pass", want);
    if (q != hipSuccess) return fail(c, FLIMO_ERR_HIP, "pass %llu failed: %s", want, hipGetErrorString(q));
    if (tags_complete(c, want)) return FLIMO_OK;          // arrived while we looked
    if (left_ms >= 0) return FLIMO_PASS_LEFT;
    // the stream is idle and the result never arrived: a ticket was left behind by an aborted launch.  Re-arm and report.
    (void)hipMemsetAsync(c->d_ticket, 0, (FIT_GROUPS + 2) * sizeof(unsigned int), c->stream);
    (void)hipMemsetAsync(c->d_wl_count, 0, sizeof(int), c->stream);
    (void)hipMemsetAsync(c->d_tie_count, 0, 2 * sizeof(unsigned int), c->stream);
    (void)hipStreamSynchronize(c->stream);
    c->prev.valid = 0;
    return fail(c, FLIMO_ERR_HIP, "pass %llu completed without publishing its result (reduction tickets re-armed)", want);
  }
}

extern "C" int flimo_set_wait_timeout_ms(flimo_ctx* c, int ms) {
  if (!c || ms < 0) return FLIMO_ERR_INVALID;
  c->wait_timeout_ms = ms;
  return FLIMO_OK;
}

extern "C" int flimo_match_reduce(flimo_ctx* c, const double x26[26], const flimo_match_cfg* cfg, double HTH[144], double HTh[12], int* M);
extern "C" int flimo_match_reduce_overlap(flimo_ctx* c, const double x26[26], const flimo_match_cfg* cfg, double HTH[144], double HTh[12],
                                          int* M, void (*while_in_flight)(void*), void* arg) {
  if (!c) return FLIMO_ERR_INVALID;
  c->overlap_fn = while_in_flight;
  c->overlap_arg = arg;
  const int rc = flimo_match_reduce(c, x26, cfg, HTH, HTh, M);
  c->overlap_fn = nullptr;                            // (not reached on this path: the caller runs it)
  return rc;
}

static inline void run_overlap(flimo_ctx* c) {
  if (!c->overlap_fn) return;
  void (*fn)(void*) = c->overlap_fn;
  c->overlap_fn = nullptr;
  fn(c->overlap_arg);
}

// One measurement pass = Mapper::match + calculate_H + H^T H for the scan at pose x26, in three steps (round 6: the 380-line function
// of rounds 2-5 cut where its phases part):
//   pass_plan    -- what this call is: which points take part, the pose constants, which layout the pass runs in (one launch /
//                   k-NN + widening + fit / records), whether the launch that waits on the GPU is this call's;
//   pass_launch  -- publishes the pose to the waiting launch or queues the pass's launches, then queues the NEXT pass of the
//                   same update behind them (pipelined host loop);
//   pass_collect -- waits for the sums (granules in mapped host memory), settles exact distance ties, decodes H^T H, H^T h, M.
// NUM_MATCH_POINTS other than 5 takes pass_general_k (records + record reduction, synchronous).
struct PassPlan {
  const flimo_match_cfg* cfg = nullptr;
  PoseMats P;
  MatchParams mp;
  size_t nq = 0;                   // points that take part (the first MAX_NUM_PC2MATCH of pc2match)
  int n_all = 0;                   // resident query set (== nq, or the whole scan when no cap binds)
  bool general_k = false;
  bool use_pre = false, was_pre = false;      // the pass queued ahead of this call is this call's / was published to
  bool last_of_update = false;     // the caller said so (flimo_pass_pipeline_last): no pass is queued behind this one
  uint64_t grid_version = 0;
  const DeskewArgs* dkp = nullptr;            // a pending deskew rides on this pass's k-NN launch
  bool cap_binds = false, want_recs = false, want_count = false, fused_cap = false, use_fit2 = false;
  int tlev = 0;                    // timing level of THIS pass
  hipEvent_t* ev = nullptr;        // its events (the context's set, or the next of the ring when reading is deferred)
  int tail_max = 0;
  bool tail = false, fused = false, after_fine = false, widen_timed = false;
  unsigned long long seq = 0;
  bool ties_on = false, inline_ties = false;
  TieList tl{};
  BookView book{};
  const BookView* bookp = nullptr;
  double tp0 = 0, tpa = 0, tpb = 0, tpc = 0, tp1 = 0;      // developer timing of the host side (FLIMO_PROF_PASS)
};
static const bool g_prof_pass = getenv("FLIMO_PROF_PASS") != nullptr;        // developer timing of the host side of a pass
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// rc > 0: nothing to do (no map, no points): the zeroed outputs stand and the call returns FLIMO_OK
constexpr int PASS_NOTHING = 2000;
static int pass_plan(flimo_ctx* c, const double x26[26], const flimo_match_cfg* cfg, PassPlan& pl) {
  pl.cfg = cfg;
  pl.general_k = cfg->NUM_MATCH_POINTS != 5;
  pl.last_of_update = c->pipe_last_hint;
  c->pipe_last_hint = false;
  // (a pass queued ahead of this call -- pipelined host loop -- is either this call's, decided below before anything is queued, or
  //  told to leave: every early way out of this function cancels it)
  struct PreGuard { flimo_ctx* c; bool decided = false; ~PreGuard() { if (!decided) cancel_prelaunch(c); } } pre_guard{c};
  { const int rca = check_abandoned(c); if (rca) return rca; }
  c->last_nq = 0;
  c->last_cfg = *cfg;
  c->recs_valid = c->dbg_valid = false;
  if (!c->grid_valid && c->map_n > 0) { int rc0 = rebuild_grid(c); if (rc0) return rc0; }   // never a stored map without its index
  if (!c->grid_valid || c->map_n == 0) return PASS_NOTHING;      // Mapper::match: `if(not this->exists()) return matches;`
  size_t nq = c->scan_n;
  if (cfg->MAX_NUM_PC2MATCH >= 0 && nq > (size_t)cfg->MAX_NUM_PC2MATCH) nq = (size_t)cfg->MAX_NUM_PC2MATCH;
  if (nq == 0) return PASS_NOTHING;
  pl.nq = nq;
  (void)hipSetDevice(c->device);            // (not ctx_enter: a waiting pass may be THIS call's)
  // Is the pass that waits on the GPU this call's?  Same scan, same settings, same map index, the next pass number, nothing else
  // to queue first (no re-sort, no pending deskew), and not so old that its workgroups may have given up (CH_POLL_MS).
  pl.grid_version = c->grid_builds * 0x100000000ull + c->grid_merges;
  bool use_pre = c->pre.active && !pl.general_k && c->prev.valid && c->pre.nq == nq && c->pre.seq == c->pass_seq + 1 &&
                 same_match_cfg(c->pre.cfg, *cfg) && c->pre.grid_version == pl.grid_version && !c->deskew_pending &&
                 !(nq < c->sorted_n || (c->sorted_n < c->scan_n && nq > c->sorted_n)) && c->pre.n_all == (int)c->sorted_n &&
                 !c->debug_recs && !(cfg->MAX_NUM_MATCHES >= 0 && (size_t)cfg->MAX_NUM_MATCHES < nq);
  if (use_pre && !(wall_s() - c->pre.t_launch < 0.25e-3 * (double)CH_POLL_MS)) { use_pre = false; c->pipe_aged++; }      // (this call's, but too old)
  if (!use_pre) cancel_prelaunch(c);
  pre_guard.decided = true;
  int rc = ensure_recs(c, nq);
  if (rc) { cancel_prelaunch(c); return rc; }

  pl.tp0 = g_prof_pass ? now_us() : 0.0;
  pose_from_x26(x26, pl.P);
  MatchParams& mp = pl.mp;
  mp.max_dist_plane_d = cfg->MAX_DIST_PLANE;          // the gate compares the float sq. distance with the DOUBLE threshold (Plane.cpp:47)
  mp.plane_threshold = (float)cfg->PLANE_THRESHOLD;
  mp.estimate_extrinsics = cfg->estimate_extrinsics ? 1 : 0;
  mp.n_queries = (int)nq;
  mp.max_ring = gate_rings(c, cfg->MAX_DIST_PLANE);

  // Only the first MAX_NUM_PC2MATCH points (pc2match order) take part (Mapper.cpp:63-69): when that is a strict prefix,
  // the Morton-sorted query set is rebuilt once per scan from that prefix alone.
  if (nq < c->sorted_n || (c->sorted_n < c->scan_n && nq > c->sorted_n)) {
    const size_t want = nq;
    { const int rcf = flush_deskew(c); if (rcf) return rcf; }
    HIPCHK(c, sort_scan(c->stream, c->d_scan, want, c->d_scan_sorted, c->scratch));
    c->sorted_n = want;
    c->prev.valid = 0;
  }
  pl.n_all = (int)c->sorted_n;
  // A pending deskew rides on this pass's k-NN launch when that launch covers the whole scan and is the first to read it
  const bool ride = c->deskew_pending && !pl.general_k && c->deskew_n == (size_t)pl.n_all &&
                    !(c->fine_valid && mp.max_ring >= 1);     // (a fine pre-pass reads the scan first)
  if (!ride) { const int rcf = flush_deskew(c); if (rcf) return rcf; }
  pl.dkp = ride ? &c->deskew_args : nullptr;
  c->deskew_pending = false;
  pl.cap_binds = cfg->MAX_NUM_MATCHES >= 0 && (size_t)cfg->MAX_NUM_MATCHES < nq;
  if (pl.general_k) return FLIMO_OK;
  pl.want_recs = c->debug_recs || pl.cap_binds;
  if (c->debug_recs) HIPCHK(c, hipMemsetAsync(c->d_cand, 0, sizeof(unsigned long long), c->stream));
  pl.want_count = c->debug_recs;
  pl.tpa = g_prof_pass ? now_us() : 0.0;
  // effective level of THIS pass (level 1 may sample every timing_stride-th pass)
  pl.tlev = (c->timing == 1 && c->timing_stride > 1 && ((c->pass_seq + 1) % (unsigned long long)c->timing_stride) != 0) ? 0 : c->timing;
  pl.ev = (c->timing_deferred && c->ev_ring_made) ? c->ev_ring[c->ev_pending] : c->ev;
  // level 1: the two events ride on the k-NN dispatch itself (kernel begin / end, no extra packets)
  // the k-NN launch finishes its own stragglers (in-kernel tail) for gates of up to 3 rings; the worklist + widening dispatch
  // remains for wider gates and for the developer switch FLIMO_TAIL=0
  // First pass of a scan (no bound from a previous pass): with a poor prior whole waves of far-off points are pending at once,
  // and a wave finishing 32 such queries two lanes each is one long chain -- those are better spread over the chip by the
  // worklist dispatch.  With a good prior (the usual case in a sequence) the first pass has a handful of stragglers like any
  // other: the count every pass publishes with its result decides for the next scan.
  const bool first_pass = !c->prev.valid;
  c->pass_in_scan = first_pass ? 0 : std::min(c->pass_in_scan + 1, 3);
  // later passes likewise, by the count the pass at the same position of the last scan published: a sparse far range (256k-point
  // sweeps over a 900 m map) keeps thousands of points beyond their 3x3x3 block in every pass
  // The bound grows with the scan: what hurts is a wave whose queries are ALL pending (one long chain), and a launch of n
  // queries spreads 1/64 of them over its waves a handful at a time
  pl.tail_max = c->tail_max_env > 0 ? c->tail_max_env : std::max(1024, pl.n_all / 64);
  const bool tail_here = first_pass ? (c->stragglers_hist[0] <= pl.tail_max)
                                    : (c->stragglers_hist[c->pass_in_scan] <= pl.tail_max);
  pl.tail = c->tail && tail_here && mp.max_ring >= 2 && mp.max_ring <= 3;
  c->prev.probe_min = c->probe_min;
  // One launch for the whole pass (k-NN + tail + fit + reduction) whenever the tail applies, no records are wanted and the
  // k-NN runs with its default two lanes per query
  pl.fused = pl.tail && c->fuse && !pl.want_recs;
  pl.seq = ++c->pass_seq;
  // exact distance ties: the reference's choice needs the octree's visiting order, i.e. the device insert book
  pl.ties_on = c->ties && c->gbook.active;
  pl.tl = TieList{};
  pl.tl.count_next = c->d_tie_count + ((pl.seq + 1) & 1);            // re-armed by this pass's reduction for the next pass
  // The per-pass fast paths (one launch / k-NN + widening + fit2) settle ties where they build the rows (tie_repair_wave): no list,
  // the count they publish stays 0.  The records / caps / debug path lists them for tie_kernel as before.
  pl.inline_ties = pl.ties_on && !pl.want_recs;
  if (pl.ties_on && !pl.inline_ties) { pl.tl.list = c->d_tie_list; pl.tl.count = c->d_tie_count + (pl.seq & 1); pl.tl.cap = (unsigned)c->tie_cap; }
  pl.book = BookView{c->gbook.node_c, c->gbook.node_child, c->gbook.node_cnt, c->gbook.root, c->d_tie_settled};
  pl.bookp = pl.inline_ties ? &pl.book : nullptr;
  // crowded regions first: the fine pre-pass settles the queries whose five lie within centimetres (second-level grid)
  pl.after_fine = c->fine_valid && mp.max_ring >= 1;
  if (use_pre && !(pl.fused && !pl.after_fine && pl.tlev == 0)) {
    // (what was queued ahead is a one-launch pass without a fine pre-pass and without timing events: anything else -- a straggler
    //  count that changed the layout, a sampled pass -- is launched the usual way)
    cancel_prelaunch(c);
    use_pre = false;
  }
  if (use_pre) {
    // (the clock once more, right before the store: the pose maths above, a page fault, a descheduled thread -- a pass whose
    //  workgroups may be giving up is told to leave instead, and launched the usual way below)
    const int test_delay_ms = c->test_publish_delay_ms;
    if (wall_s() - c->pre.t_launch >= 0.25e-3 * (double)CH_POLL_MS) { cancel_prelaunch(c); use_pre = false; c->pipe_aged++; }
    else if (test_delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(test_delay_ms));      // (tests: the window the decision word closes)
  }
  pl.use_pre = pl.was_pre = use_pre;
  pl.widen_timed = !pl.tail && pl.tlev == 1 && mp.max_ring >= 2;
  // MAX_NUM_MATCHES path: the fit kernel only writes the records (no reduction), one fused kernel ranks them in scan
  // order, reduces the first MAX_NUM_MATCHES and publishes to slot 0
  pl.fused_cap = pl.cap_binds && !c->debug_recs;
  pl.use_fit2 = !pl.want_recs;                             // the per-pass fast path (granule results)
  return FLIMO_OK;
}

// any NUM_MATCH_POINTS: exact k-NN by the ring search, M x 3 plane fit, records, record reduction (slow, general pass)
static int pass_general_k(flimo_ctx* c, const PassPlan& pl, double HTH[144], double HTh[12], int* M) {
  const flimo_match_cfg* cfg = pl.cfg;
  const int n_all = pl.n_all;
  const size_t nq = pl.nq;
  if ((size_t)n_all > c->nbrk_cap) {
    (void)hipFree(c->d_nbrk);
    c->d_nbrk = nullptr; c->nbrk_cap = 0;
    const size_t cap = (size_t)n_all + (size_t)n_all / 4 + 1024;
    HIPCHK(c, hipMalloc(&c->d_nbrk, cap * nbrk_rec_size()));
    c->nbrk_cap = cap;
  }
  c->prev.valid = 0;                                 // the 5-NN records of this scan (pruning bound) are not maintained here
  const BookView bookk{c->gbook.node_c, c->gbook.node_child, c->gbook.node_cnt, c->gbook.root, nullptr};
  if (!launch_match_k(c->stream, cfg->NUM_MATCH_POINTS, c->grid, c->d_scan_sorted, n_all, pl.P, pl.mp, c->d_nbrk, c->d_recs, c->d_dbg,
                      (c->ties && c->gbook.active) ? &bookk : nullptr))
    return fail(c, FLIMO_ERR_UNSUPPORTED, "NUM_MATCH_POINTS out of range");
  if (pl.cap_binds) launch_cap(c->stream, c->d_recs, (int)nq, cfg->MAX_NUM_MATCHES);
  launch_reduce(c->stream, c->d_recs, (int)nq, c->reduce_waves, c->d_partials, c->d_out256_host);
  // the general pass settles ties for every query (tiek_kernel) and does not use the two alternating tie counters of the
  // 5-NN passes; it keeps them armed for whichever 5-NN pass comes next
  HIPCHK(c, hipMemsetAsync(c->d_tie_count, 0, 2 * sizeof(unsigned int), c->stream));
  HIPCHK(c, hipGetLastError());
  run_overlap(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  ++c->pass_seq;
  for (int i = 0; i < 12; i++) {
    for (int j = 0; j < 12; j++) HTH[i * 12 + j] = c->h_out256[c->mfma_idx[i][j]];
    HTh[i] = c->h_out256[c->mfma_idx[i][12]];
  }
  *M = (int)llround(c->h_out256[c->mfma_idx[13][13]]);
  c->last_nq = (int)nq;
  c->last_P = pl.P; c->last_mp = pl.mp; c->last_n_all = n_all;
  c->recs_valid = true; c->dbg_valid = true;
  c->last_stragglers = -1;
  c->async_deskews = 0;
  return FLIMO_OK;
}

static int pass_launch(flimo_ctx* c, PassPlan& pl) {
  const flimo_match_cfg* cfg = pl.cfg;
  const PoseMats& P = pl.P;
  const MatchParams& mp = pl.mp;
  const int n_all = pl.n_all;
  const size_t nq = pl.nq;
  const unsigned long long seq = pl.seq;
  const int tlev = pl.tlev;
  if (pl.use_pre) {
    publish_prelaunch(c, P, c->prev.RT, seq);
    c->fused_passes++;
  } else if (pl.after_fine) {
    launch_knn5_fine(c->stream, c->fine, c->d_scan_sorted, n_all, P, c->d_nbr, c->prev, c->fine_qlo, c->fine_qhi, &pl.tl, seq);
    c->fine_passes++;
  }
  if (pl.use_pre) {
    // its launch is on the GPU already
  } else if (pl.fused) {
    launch_match_fused(c->stream, c->grid, c->d_scan_sorted, n_all, P, mp, c->d_nbr, c->d_wl, c->d_wl_count, nullptr, c->prev,
                       c->live_idx, c->d_fit2_partials, c->d_granules_host, c->d_ticket, seq, tlev == 1 ? pl.ev[0] : nullptr,
                       tlev == 1 ? pl.ev[1] : nullptr, &pl.tl, pl.after_fine ? 1 : 0, pl.dkp, nullptr, nullptr, pl.bookp);
    c->fused_passes++;
  } else
  launch_knn5(c->stream, 2, c->grid, c->d_scan_sorted, n_all, P, mp.max_ring, c->d_nbr, c->d_wl,
              c->d_wl_count, c->debug_recs ? c->d_cand : nullptr, c->prev, pl.tail ? 1 : 0, tlev == 1 ? pl.ev[0] : nullptr,
              tlev == 1 ? pl.ev[1] : nullptr, nullptr, &pl.tl, pl.after_fine ? 1 : 0, seq, pl.dkp);
  c->prev_before = c->prev;
  if (c->prune) { memcpy(c->prev.RT, P.RT, sizeof(c->prev.RT)); c->prev.valid = 1; }   // the records now belong to this pose
  pl.tpb = g_prof_pass ? now_us() : 0.0;
  // A separate-dispatch pass is three launches: k-NN, widening of the worklist (one wave per pending query, dealt out over the whole
  // chip), fit + reduction.  (Rounds 3's widen_fit_kernel ran the last two as one launch, its fit workgroups polling records its
  // widening workgroups were still writing: 3 us per step bought with a forward-progress assumption and relaxed cross-XCD reads --
  // retired in round 4 by design.)
  if (!pl.tail)
    launch_widen(c->stream, c->grid, c->d_scan_sorted, P, mp.max_ring, c->d_nbr, c->d_wl, c->d_wl_count,
                 c->debug_recs ? c->d_cand : nullptr, pl.widen_timed ? pl.ev[4] : nullptr, pl.widen_timed ? pl.ev[5] : nullptr, &pl.tl);
  pl.tpc = g_prof_pass ? now_us() : 0.0;
  if (pl.want_count) HIPCHK(c, hipMemcpyAsync(c->h_wl_count, c->d_wl_count, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  // fit + reductions; the last block writes the 16x16 accumulator to mapped host memory, publishes the
  // pass number and re-arms the ticket and the worklist counter
  if (pl.fused) {
    // the fit and the reduction ran inside the k-NN launch
  } else if (pl.use_fit2)
    launch_fit2(c->stream, c->grid, c->d_scan_sorted, n_all, c->d_nbr, P, mp, c->live_idx, c->d_fit2_partials, c->d_granules_host,
                c->d_ticket, c->d_wl_count, seq, tlev == 1 ? pl.ev[2] : nullptr, tlev == 1 ? pl.ev[3] : nullptr, &pl.tl, nullptr, nullptr, pl.bookp);
  else {
    // records / caps / debug / timing level 2 (synchronous): settle the ties before the rows are built, re-arm both counters after
    HIPCHK(c, hipGetLastError());
    if (pl.ties_on) { launch_tie(c->stream, c->grid, pl.book, c->d_scan_sorted, P, c->d_nbr, pl.tl); HIPCHK(c, hipGetLastError()); }
    launch_fit(c->stream, c->grid, c->d_scan_sorted, n_all, c->d_nbr, P, mp, pl.fused_cap ? nullptr : c->d_fit_partials,
               pl.want_recs ? c->d_recs : nullptr, c->debug_recs ? c->d_dbg : nullptr, pl.cap_binds ? c->d_out256 : c->d_out256_host,
               c->d_ticket, c->d_wl_count, seq);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemsetAsync(c->d_tie_count, 0, 2 * sizeof(unsigned int), c->stream));
  }
  if (pl.fused_cap) {
    launch_capreduce(c->stream, c->d_recs, (int)nq, cfg->MAX_NUM_MATCHES, c->d_out256_host, c->d_wl_count, seq);
  } else if (pl.cap_binds) {
    launch_cap(c->stream, c->d_recs, (int)nq, cfg->MAX_NUM_MATCHES);
    launch_reduce(c->stream, c->d_recs, (int)nq, c->reduce_waves, c->d_partials, c->d_out256_host);
  }
  HIPCHK(c, hipGetLastError());
  pl.tp1 = g_prof_pass ? now_us() : 0.0;
  if (c->debug_recs) HIPCHK(c, hipMemcpyAsync(c->h_cand, c->d_cand, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  // ---- pipelined host loop: the NEXT pass of this update is queued now, behind this one.  Its workgroups are placed when this pass
  //      ends and wait for their constants in device memory; the next call publishes them instead of launching (use_pre above).
  //      Only the usual case is queued ahead: a one-launch pass (by the straggler count its position published in the last scan),
  //      no fine pre-pass, no timing events, no records. ----
  if (c->pipeline && !pl.last_of_update && c->d_pipe_head && c->prune && pl.use_fit2 && !pl.want_count && c->tail && c->fuse &&
      mp.max_ring >= 2 && mp.max_ring <= 3 && !c->fine_valid && pl.inline_ties == pl.ties_on) {
    const unsigned long long nseq = seq + 1;
    const int ntlev = (c->timing == 1 && c->timing_stride > 1 && (nseq % (unsigned long long)c->timing_stride) != 0) ? 0 : c->timing;
    const int npos = std::min(c->pass_in_scan + 1, 3);
    if (ntlev == 0 && c->stragglers_hist[npos] <= pl.tail_max) {
      TieList tln{};
      tln.count_next = c->d_tie_count + ((nseq + 1) & 1);
      ChainCtl pc{};
      pc.end_code = 0x80000000u | (++c->pipe_tag & 0x7fffffffu);
      PrevPass pv = c->prev;
      pv.valid = 1;                                            // (its reference pose comes from the head)
      launch_match_fused(c->stream, c->grid, c->d_scan_sorted, n_all, P, mp, c->d_nbr, c->d_wl, c->d_wl_count, nullptr, pv,
                         c->live_idx, c->d_fit2_partials, c->d_granules_host, c->d_ticket, nseq, nullptr, nullptr, &tln, 0, nullptr,
                         c->d_pipe_head, &pc, pl.bookp, ch_epoch_of(nseq));
      HIPCHK(c, hipGetLastError());
      c->pre.active = true; c->pre.seq = nseq; c->pre.nq = nq; c->pre.n_all = n_all; c->pre.pos = npos; c->pre.cfg = *cfg;
      c->pre.grid_version = pl.grid_version; c->pre.end_code = pc.end_code; c->pre.t_launch = wall_s();
    }
  }
  return FLIMO_OK;
}

// Level-1 timing: read one timed pass's events (kernel begin / end stamps of its dispatches) into the totals.
static int timing_read(flimo_ctx* c, hipEvent_t* ev, int kind) {
  const bool fused = kind & 1, use_fit2 = kind & 2, widen_timed = kind & 4;
  if (hipEventElapsedTime(&c->last_knn_ms, ev[0], ev[1]) != hipSuccess) {      // not marked complete yet: wait for it
    HIPCHK(c, hipEventSynchronize(ev[1]));
    (void)hipEventElapsedTime(&c->last_knn_ms, ev[0], ev[1]);
  }
  c->tot_knn_ms += c->last_knn_ms;
  if (fused) {
    c->last_fit_ms = 0.f; c->last_widen_ms = 0.f;            // one dispatch: everything is in the k-NN figure
    c->split_fused_ms += c->last_knn_ms; c->split_fused_n++;
  } else if (use_fit2) {
    c->last_widen_ms = 0.f;
    if (widen_timed) {
      if (hipEventElapsedTime(&c->last_widen_ms, ev[4], ev[5]) != hipSuccess) c->last_widen_ms = 0.f;
      c->tot_widen_ms += c->last_widen_ms;
    }
    // the fit dispatch carries its own pair of events (kernel begin / end); the host saw the granules, the kernel's
    // end-of-dispatch signal may still be a moment away
    if (hipEventElapsedTime(&c->last_fit_ms, ev[2], ev[3]) != hipSuccess) {
      HIPCHK(c, hipEventSynchronize(ev[3]));
      (void)hipEventElapsedTime(&c->last_fit_ms, ev[2], ev[3]);
    }
    c->tot_fit_ms += c->last_fit_ms;
    c->split_knn_ms += c->last_knn_ms; c->split_widen_ms += c->last_widen_ms; c->split_fit_ms += c->last_fit_ms; c->split_sep_n++;
  }
  return FLIMO_OK;
}
// the passes whose events nobody has read yet (deferred reading), oldest first
static int timing_drain(flimo_ctx* c) {
  const int n = c->ev_pending;
  c->ev_pending = 0;
  for (int i = 0; i < n; i++) { const int rc = timing_read(c, c->ev_ring[i], c->ev_kind[i]); if (rc) return rc; }
  return FLIMO_OK;
}

static int pass_collect(flimo_ctx* c, PassPlan& pl, double HTH[144], double HTh[12], int* M) {
  const PoseMats& P = pl.P;
  const MatchParams& mp = pl.mp;
  const int n_all = pl.n_all;
  const size_t nq = pl.nq;
  const unsigned long long seq = pl.seq;
  const int tlev = pl.tlev;
  double acc[256];
  if (pl.use_fit2) {
    // low-latency completion: every sum arrives as a 16-byte granule {value, pass number}; a group's slot is complete when
    // all of its tags carry this pass (the last granule stored is polled, then all are checked)
    auto wait_granules = [&](unsigned long long want, int left_ms = -1) -> int {
      // slot 0 carries the launch's two counters (stored last, by the block that finishes the launch); then every group's sums
      const int rcw = wait_tags(c, want, left_ms);
      if (rcw) return rcw;
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      return FLIMO_OK;
    };
    run_overlap(c);                                    // the caller's own work, beside the launch
    {
      int rcw = wait_granules(seq, pl.was_pre ? CH_POLL_MS + 25 : -1);
      if (rcw == FLIMO_PASS_LEFT) {
        // the launch that waited for this pass left before the pose reached it, as a whole: nothing ran.  The pass queued behind it for
        // the NEXT iteration is told to leave too, and this pass is launched the usual way (same number, same buffers).
        cancel_prelaunch(c);
        c->pipe_left++;
        launch_match_fused(c->stream, c->grid, c->d_scan_sorted, n_all, P, mp, c->d_nbr, c->d_wl, c->d_wl_count, nullptr, c->prev_before,
                           c->live_idx, c->d_fit2_partials, c->d_granules_host, c->d_ticket, seq, nullptr, nullptr, &pl.tl, 0, nullptr, nullptr,
                           nullptr, pl.bookp);
        HIPCHK(c, hipGetLastError());
        rcw = wait_granules(seq);
      }
      if (rcw) return rcw;
    }
    c->last_stragglers = (int)llround(c->h_granules[2 * FIT_LIVE]);
    c->stragglers_hist[c->pass_in_scan] = c->last_stragglers;
    const long n_ties = (long)llround(c->h_granules[2 * (FIT_LIVE + 1)]);
    if (n_ties > 0 && pl.ties_on) {
      // A few queries' five hinge on an exact float32 distance tie: settle them the reference's way (tie_kernel), then build the
      // rows and the sums again (one more fit dispatch).  Roughly once per 65k-point scan on measured data.
      c->tie_redos++; c->tie_queries += (unsigned long long)n_ties;
      launch_tie(c->stream, c->grid, pl.book, c->d_scan_sorted, P, c->d_nbr, pl.tl);
      const unsigned long long seq2 = ++c->pass_seq;
      TieList tl2{};
      tl2.count_next = c->d_tie_count + ((seq2 + 1) & 1);      // == this pass's counter: consumed by tie_kernel just above
      launch_fit2(c->stream, c->grid, c->d_scan_sorted, n_all, c->d_nbr, P, mp, c->live_idx, c->d_fit2_partials, c->d_granules_host,
                  c->d_ticket, c->d_wl_count, seq2, nullptr, nullptr, &tl2);
      HIPCHK(c, hipGetLastError());
      const int rcw = wait_granules(seq2);
      if (rcw) return rcw;
    }
    // slot sums in slot order, then the full 16x16 raw layout the decode below reads
    double live[FIT_LIVE];
    for (int k = 0; k < FIT_LIVE; k++) {
      double r = c->h_granules[2 * k];
      for (int g = 1; g < FIT_GROUPS; g++) r += c->h_granules[((size_t)g * FIT_LIVE_PAD + k) * 2];
      live[k] = r;
    }
    for (int t = 0; t < 256; t++) acc[t] = 0.0;
    {
      int k = 0;
      for (int i = 0; i < 12; i++) for (int j = i; j < 12; j++) { acc[c->mfma_idx[i][j]] = live[k]; acc[c->mfma_idx[j][i]] = live[k]; k++; }
      for (int i = 0; i < 12; i++) acc[c->mfma_idx[i][12]] = live[k++];
      acc[c->mfma_idx[13][13]] = live[k++];
    }
  } else if (!c->debug_recs) {
    // low-latency completion: spin on the pass number every reduction group publishes to host memory (one slot on
    // the MAX_NUM_MATCHES path)
    unsigned long long spins = 0;
    for (int g = 0; g < (pl.cap_binds ? 1 : FIT_GROUPS); g++) {
      volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->h_out256 + (size_t)g * FIT_SLOT + 256);
      while (*flag != seq) {
        _mm_pause();
        if (++spins > 40000000ull) {                             // also surfaces launch errors
          HIPCHK(c, hipStreamSynchronize(c->stream));
          if (*flag != seq) return fail(c, FLIMO_ERR_HIP, "pass %llu completed without publishing its result", seq);
          break;
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    // the fit kernel has published, so the k-NN dispatch (earlier on the same stream) and its events are complete:
    // no hipEventSynchronize (its blocking wait costs ~40 us per timed pass)
  } else {
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (tlev) {
    const int kind = (pl.fused ? 1 : 0) | (pl.use_fit2 ? 2 : 0) | (pl.widen_timed ? 4 : 0);
    if (pl.ev != c->ev) {                                         // deferred: the events stay unread (timing_drain)
      c->ev_kind[c->ev_pending++] = (unsigned char)kind;
      if (c->ev_pending == flimo_ctx::EV_RING) { const int rc = timing_drain(c); if (rc) return rc; }
    } else {
      const int rc = timing_read(c, c->ev, kind);
      if (rc) return rc;
    }
    c->tot_passes++; c->tot_queries += n_all;
  }
  if (g_prof_pass) {
    static double acc_launch = 0, acc_wait = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0; static long cnt = 0;
    const double tp2 = now_us();
    acc_launch += pl.tp1 - pl.tp0; acc_wait += tp2 - pl.tp1;
    a0 += pl.tpa - pl.tp0; a1 += pl.tpb - pl.tpa; a2 += pl.tpc - pl.tpb; a3 += pl.tp1 - pl.tpc;
    if (++cnt % 200 == 0) {
      fprintf(stderr, "[flimo pass] host: pose+launches %.2f us (prep %.2f, knn launch %.2f, widen launch %.2f, fit launch %.2f), wait for result %.2f us (mean of 200)\n",
              acc_launch / 200, a0 / 200, a1 / 200, a2 / 200, a3 / 200, acc_wait / 200);
      acc_launch = acc_wait = a0 = a1 = a2 = a3 = 0;
    }
  }
  c->async_deskews = 0;            // the pass completed: the stream is idle
  if (pl.want_count) c->last_widen_count = *c->h_wl_count;
  if (c->debug_recs) c->last_cand_per_query = (double)(*c->h_cand) / (double)nq;
  // final sum over the reduction groups in slot order (the records path delivers one slot)
  if (!pl.use_fit2) {
    c->last_stragglers = -1;
    const int slots = pl.cap_binds ? 1 : FIT_GROUPS;
    for (int t = 0; t < 256; t++) {
      double r = c->h_out256[t];
      for (int g = 1; g < slots; g++) r += c->h_out256[(size_t)g * FIT_SLOT + t];
      acc[t] = r;
    }
  }
  for (int i = 0; i < 12; i++) {
    for (int j = 0; j < 12; j++) HTH[i * 12 + j] = acc[c->mfma_idx[i][j]];
    HTh[i] = acc[c->mfma_idx[i][12]];
  }
  *M = (int)llround(acc[c->mfma_idx[13][13]]);
  c->last_nq = (int)nq;
  c->last_P = P; c->last_mp = mp; c->last_n_all = n_all;
  c->recs_valid = pl.want_recs && !pl.fused_cap;      // the fused path leaves the records un-capped: a fetch re-materialises them
  c->dbg_valid = c->debug_recs;
  return FLIMO_OK;
}

extern "C" int flimo_match_reduce(flimo_ctx* c, const double x26[26], const flimo_match_cfg* cfg, double HTH[144],
                                  double HTh[12], int* M) {
  if (!c || !x26 || !cfg || !HTH || !HTh || !M) return FLIMO_ERR_INVALID;
  if (cfg->NUM_MATCH_POINTS < 3 || cfg->NUM_MATCH_POINTS > 8)
    return fail(c, FLIMO_ERR_UNSUPPORTED, "NUM_MATCH_POINTS must be in 3..8 (a plane needs 3 points; the neighbour records hold 8)");
  for (int i = 0; i < 144; i++) HTH[i] = 0.0;
  for (int i = 0; i < 12; i++) HTh[i] = 0.0;
  *M = 0;
  PassPlan pl;
  int rc = pass_plan(c, x26, cfg, pl);
  if (rc == PASS_NOTHING) return FLIMO_OK;
  if (rc) return rc;
  if (pl.general_k) return pass_general_k(c, pl, HTH, HTh, M);
  if ((rc = pass_launch(c, pl)) != FLIMO_OK) return rc;
  return pass_collect(c, pl, HTH, HTh, M);
}

// ---- the whole iterated update enqueued at once (flimo_chain.h) ---------------------------------------------------------------
// Layout of a pass inside the chain = the layout flimo_match_reduce gives the pass at the same position: one launch (k-NN + tail +
// fit + reduction) when the last scan's pass at that position published few stragglers, else the k-NN launch followed by widening
// + fit in one launch (or widening, then fit); a fine pre-pass before either when a crowded region is active.  Pass 0 gets its pose
// constants as kernel arguments (the host knows x), later passes read them from the device filter.
extern "C" int flimo_set_update_mode(flimo_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return FLIMO_ERR_INVALID;
  c->update_mode = mode;
  c->host_update = mode == 1 || (mode == 0 && auto_host_update(c));
  return FLIMO_OK;
}
extern "C" int flimo_set_pass_pipeline(flimo_ctx* c, int on) {
  if (!c) return FLIMO_ERR_INVALID;
  cancel_prelaunch(c);
  if (!c->pipeline_env) c->pipeline = on != 0;
  if (c->update_mode == 0) c->host_update = auto_host_update(c);
  return FLIMO_OK;
}
extern "C" int flimo_pass_pipeline_end(flimo_ctx* c) {
  if (!c) return FLIMO_ERR_INVALID;
  cancel_prelaunch(c);
  return FLIMO_OK;
}
extern "C" int flimo_pass_pipeline_last(flimo_ctx* c) {
  if (!c) return FLIMO_ERR_INVALID;
  c->pipe_last_hint = true;                            // (consumed by the next flimo_match_reduce)
  return FLIMO_OK;
}
extern "C" int flimo_pass_pipeline_stats(const flimo_ctx* c, unsigned long long out[4]) {
  if (!c || !out) return FLIMO_ERR_INVALID;
  out[0] = c->pipe_published; out[1] = c->pipe_cancelled; out[2] = c->pipe_aged; out[3] = c->pipe_left;
  return FLIMO_OK;
}
extern "C" int flimo_update_mode(const flimo_ctx* c, int* chained, double* launch_rtt_us) {
  if (!c) return FLIMO_ERR_INVALID;
  if (chained) *chained = c->host_update ? 0 : 1;
  if (launch_rtt_us) *launch_rtt_us = c->launch_rtt_us;
  return FLIMO_OK;
}
extern "C" int flimo_chain_stats(flimo_ctx* c, double out[5], int reset) {
  if (!c || !out) return FLIMO_ERR_INVALID;
  out[0] = c->chain_alg_ms; out[1] = (double)c->chain_alg_n; out[2] = (double)c->chains_run; out[3] = (double)c->chains_back;
  out[4] = (double)c->chains_declined;
  if (reset) { c->chain_alg_ms = 0; c->chain_alg_n = 0; c->chains_run = c->chains_back = c->chains_declined = 0; }
  return FLIMO_OK;
}

extern "C" int flimo_update_chain(flimo_ctx* c, const flimo_match_cfg* cfg, flimo_chain_io* io) {
  if (!c || !cfg || !io) return FLIMO_ERR_INVALID;
  io->status = FLIMO_CHAIN_DECLINED; io->reason = 0; io->passes = 0; io->it_next = -1; io->t = 0; io->meas_valid = 0; io->meas_M = 0;
  auto decline = [&]() { c->chains_declined++; return FLIMO_OK; };
  { const int rca = check_abandoned(c); if (rca) return rca; }
  if (c->host_update) return decline();
  if (cfg->NUM_MATCH_POINTS != 5) return decline();
  const int n_pass = io->max_iter + 1;
  if (io->max_iter < 0 || n_pass > CH_MAX_PASSES) return decline();
  if (c->debug_recs || !c->tail || !c->fuse) return decline();
  if (!c->grid_valid && c->map_n > 0) { int rc0 = rebuild_grid(c); if (rc0) return rc0; }
  if (!c->grid_valid || c->map_n == 0) return decline();      // Mapper::match returns no matches: the host loop handles M = 0
  size_t nq = c->scan_n;
  if (cfg->MAX_NUM_PC2MATCH >= 0 && nq > (size_t)cfg->MAX_NUM_PC2MATCH) nq = (size_t)cfg->MAX_NUM_PC2MATCH;
  if (nq == 0) return decline();
  if (cfg->MAX_NUM_MATCHES >= 0 && (size_t)cfg->MAX_NUM_MATCHES < nq) return decline();      // caps need the records
  ctx_enter(c);
  { int rc = ensure_recs(c, nq); if (rc) return rc; }
  MatchParams mp;
  mp.max_dist_plane_d = cfg->MAX_DIST_PLANE;
  mp.plane_threshold = (float)cfg->PLANE_THRESHOLD;
  mp.estimate_extrinsics = cfg->estimate_extrinsics ? 1 : 0;
  mp.n_queries = (int)nq;
  mp.max_ring = gate_rings(c, cfg->MAX_DIST_PLANE);
  if (mp.max_ring < 2 || mp.max_ring > 3) return decline();
  c->last_nq = 0;
  c->last_cfg = *cfg;
  c->recs_valid = c->dbg_valid = false;
  if (nq < c->sorted_n || (c->sorted_n < c->scan_n && nq > c->sorted_n)) {
    { const int rcf = flush_deskew(c); if (rcf) return rcf; }
    HIPCHK(c, sort_scan(c->stream, c->d_scan, nq, c->d_scan_sorted, c->scratch));
    c->sorted_n = nq;
    c->prev.valid = 0;
  }
  const int n_all = (int)c->sorted_n;
  const bool after_fine = c->fine_valid && mp.max_ring >= 1;
  const bool ride = c->deskew_pending && c->deskew_n == (size_t)n_all && !after_fine;
  if (!ride) { const int rcf = flush_deskew(c); if (rcf) return rcf; }
  const DeskewArgs* dkp = ride ? &c->deskew_args : nullptr;
  c->deskew_pending = false;

  // the prior, where the first pass's extra workgroup reads it; the measurement-independent half of iteration -1 with it
  // (x == x_prop: no transcendental function is evaluated, host and device agree bit for bit)
  PoseMats P0;
  pose_from_x26(io->x26, P0);
  ChainPrior& pr = *c->h_chain_prior;
  memcpy(pr.x, io->x26, sizeof(pr.x));
  memcpy(pr.P, io->P, sizeof(pr.P));
  memcpy(pr.limit, io->limits, sizeof(pr.limit));
  pr.R = io->R; pr.D = io->D; pr.max_iter = io->max_iter; pr.pad = 0;
  ik_pre_serial(io->x26, io->x26, io->P, io->R, pr.dxn, pr.AG, pr.AG + 144);
  __atomic_thread_fence(__ATOMIC_RELEASE);

  const unsigned long long tag = ++c->chain_tag;
  const unsigned long long seq0 = c->pass_seq;
  const bool ties_on = c->ties && c->gbook.active;
  const BookView book{c->gbook.node_c, c->gbook.node_child, c->gbook.node_cnt, c->gbook.root, c->d_tie_settled};
  const BookView* bookp = ties_on ? &book : nullptr;
  const ChainHead* head = reinterpret_cast<const ChainHead*>(c->d_chain);
  const int tail_max = std::max(1024, n_all / 64);
  struct Plan { int pos; bool fused, combined, timed; };
  Plan plan[CH_MAX_PASSES];
  int pos = c->pass_in_scan;
  bool prev_valid = c->prev.valid != 0;
  c->prev.probe_min = c->probe_min;
  ChainCtl ctl{};
  ctl.S = c->d_chain;
  ctl.gran = reinterpret_cast<double2*>(c->d_chain_gran);
  ctl.res = reinterpret_cast<double2*>(c->d_chain_res);
  ctl.log = io->want_log ? reinterpret_cast<double2*>(c->d_chain_log) : nullptr;
  ctl.tag = tag;
  ctl.end_code = 0x80000000u | (unsigned int)(tag & 0x7fffffffull);
  for (int i = 0; i < n_pass; i++) {
    const unsigned long long seq = seq0 + 1 + (unsigned long long)i;
    const bool first_pass = !prev_valid;
    pos = first_pass ? 0 : std::min(pos + 1, 3);
    const bool tail_here = first_pass ? (c->stragglers_hist[0] <= tail_max)
                                      : (c->stragglers_hist[pos] <= tail_max);
    const bool fused = tail_here;
    const bool combined = false;
    const bool timed = c->timing == 1 && (c->timing_stride <= 1 || (seq % (unsigned long long)c->timing_stride) == 0);
    plan[i] = Plan{pos, fused, combined, timed};
    hipEvent_t* ev = timed ? c->chain_ev[i] : nullptr;
    const ChainHead* ch = i == 0 ? nullptr : head;
    ctl.prior = i == 0 ? c->d_chain_prior : nullptr;
    PrevPass pv = c->prev;
    pv.valid = prev_valid ? 1 : 0;            // (pass 0: the context's own bound, if any; later passes: RT comes from the device filter)
    TieList tl{};                                // (ties are settled inside the reducing launches: nothing is listed)
    tl.count_next = c->d_tie_count + ((seq + 1) & 1);
    unsigned int wait_epoch = 0u;                 // (the algebra launch queued before this pass has stored its constants)
    if (after_fine) {
      launch_knn5_fine(c->stream, c->fine, c->d_scan_sorted, n_all, P0, c->d_nbr, pv, c->fine_qlo, c->fine_qhi, &tl, seq, ch, wait_epoch, ctl.end_code);
      wait_epoch = 0u;
      c->fine_passes++;
    }
    const DeskewArgs* dk = i == 0 ? dkp : nullptr;
    if (fused) {
      launch_match_fused(c->stream, c->grid, c->d_scan_sorted, n_all, P0, mp, c->d_nbr, c->d_wl, c->d_wl_count, nullptr, pv, c->live_idx,
                         c->d_fit2_partials, c->d_chain_gran, c->d_ticket, seq, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr, &tl,
                         after_fine ? 1 : 0, dk, ch, &ctl, bookp, wait_epoch);
    } else {
      launch_knn5(c->stream, 2, c->grid, c->d_scan_sorted, n_all, P0, mp.max_ring, c->d_nbr, c->d_wl, c->d_wl_count, nullptr, pv, 0,
                  ev ? ev[0] : nullptr, ev ? ev[1] : nullptr, nullptr, &tl, after_fine ? 1 : 0, seq, dk, ch, wait_epoch, ctl.end_code);
      {
        launch_widen(c->stream, c->grid, c->d_scan_sorted, P0, mp.max_ring, c->d_nbr, c->d_wl, c->d_wl_count, nullptr, ev ? ev[6] : nullptr, ev ? ev[7] : nullptr, &tl, ch);
        launch_fit2(c->stream, c->grid, c->d_scan_sorted, n_all, c->d_nbr, P0, mp, c->live_idx, c->d_fit2_partials, c->d_chain_gran,
                    c->d_ticket, c->d_wl_count, seq, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr, &tl, ch, &ctl, bookp);
      }
    }
    launch_ieskf(c->stream, ctl, seq, i == 0 ? P0.RT : nullptr, ev ? ev[4] : nullptr, ev ? ev[5] : nullptr);
    prev_valid = c->prune;
  }
  HIPCHK(c, hipGetLastError());
  c->chains_run++;

  // ---- one wait: the head of the result (stored last), then every granule ----
  const volatile unsigned long long* rt = reinterpret_cast<const volatile unsigned long long*>(c->h_chain_res);
  auto tag_at = [&](int slot) { return rt[2 * slot + 1] == tag; };
  auto val_at = [&](int slot) { return c->h_chain_res[2 * slot]; };
  auto complete = [&]() {
    for (int k = 0; k < CH_RES; k++) if (!tag_at(k)) return false;
    return true;
  };
  {
    // (whatever ends the wait early: the context's pass count stays consistent with the tie counters' parity only if every queued
    //  pass is assumed to run; a chain that is abandoned or fails resets the pass buffers' bookkeeping instead)
    if (c->wait_timeout_ms == 0) { c->pass_seq = seq0 + (unsigned long long)n_pass; return abandon_wait(c, "update chain", tag - 0x4000000000000000ull); }
    unsigned long long spins = 0;
    double deadline = 0.0;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (;;) {
      if (tag_at(CH_STATUS) && complete()) break;
      _mm_pause();
      if ((++spins & 0x3fffull) != 0) continue;
      const double t = now_s();
      if (deadline == 0.0) { deadline = t + 1e-3 * (double)c->wait_timeout_ms; continue; }
      if (t < deadline) continue;
      const hipError_t q = hipStreamQuery(c->stream);
      c->prev.valid = 0;
      if (q == hipErrorNotReady) { c->pass_seq = seq0 + (unsigned long long)n_pass; return abandon_wait(c, "update chain", tag - 0x4000000000000000ull); }
      if (q != hipSuccess) return fail(c, FLIMO_ERR_HIP, "the update chain failed: %s", hipGetErrorString(q));
      if (complete()) break;
      (void)hipMemsetAsync(c->d_ticket, 0, (FIT_GROUPS + 2) * sizeof(unsigned int), c->stream);
      (void)hipMemsetAsync(c->d_wl_count, 0, sizeof(int), c->stream);
      (void)hipMemsetAsync(c->d_tie_count, 0, 2 * sizeof(unsigned int), c->stream);
      (void)hipStreamSynchronize(c->stream);
      return fail(c, FLIMO_ERR_HIP, "the update chain completed without publishing its result (reduction tickets re-armed)");
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  io->status = (int)llround(val_at(CH_STATUS));
  io->reason = (int)llround(val_at(CH_BAIL));
  io->passes = (int)llround(val_at(CH_PASSES));
  io->it_next = (int)llround(val_at(CH_IT));
  io->t = (int)llround(val_at(CH_T));
  for (int k = 0; k < 26; k++) io->x26_out[k] = val_at(CH_X + k);
  if (io->reason == CH_R_FINAL || io->reason == CH_R_DEGENERATE) {
    io->meas_valid = 1;
    int k = 0;
    for (int i = 0; i < 12; i++) for (int j = i; j < 12; j++) { const double v = val_at(CH_SUMS + k++); io->meas_HTH[i * 12 + j] = v; io->meas_HTH[j * 12 + i] = v; }
    for (int i = 0; i < 12; i++) io->meas_HTh[i] = val_at(CH_SUMS + k++);
    io->meas_M = (int)llround(val_at(CH_SUMS + k));
  }
  // passes whose measurement ran: the completed iterations, and the one that was handed back
  const int executed = std::min(n_pass, io->passes + 1);
  for (int i = 0; i < executed; i++) {
    flimo_chain_pass& L = io->log[i];
    L.M = (int)llround(val_at(CH_PASSINFO + 3 * i)); L.stragglers = (int)llround(val_at(CH_PASSINFO + 3 * i + 1));
    L.ties = (int)llround(val_at(CH_PASSINFO + 3 * i + 2));
    c->stragglers_hist[plan[i].pos] = L.stragglers;
    c->last_stragglers = L.stragglers;
    c->pass_in_scan = plan[i].pos;
    if (plan[i].fused) c->fused_passes++;
    if (io->want_log && i < io->passes) {
      const volatile unsigned long long* lt = reinterpret_cast<const volatile unsigned long long*>(c->h_chain_log) + (size_t)i * CH_LOGN * 2;
      const double* lv = c->h_chain_log + (size_t)i * CH_LOGN * 2;
      unsigned long long spins = 0;
      for (int k = 0; k < CH_LOGN; k++) {
        while (lt[2 * k + 1] != tag) { _mm_pause(); if (++spins > 400000000ull) return fail(c, FLIMO_ERR_HIP, "the update chain's log did not arrive"); }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      for (int k = 0; k < 144; k++) L.HTH[k] = lv[2 * k];
      for (int k = 0; k < 12; k++) L.HTh[k] = lv[2 * (144 + k)];
      for (int k = 0; k < 23; k++) L.dx[k] = lv[2 * (156 + k)];
      for (int k = 0; k < 26; k++) L.x_after[k] = lv[2 * (179 + k)];
    }
  }
  if (io->reason != CH_R_FINAL) c->chains_back++;
  // what the context knows about its last pass (fetches, the next pass's bound): the handed-back iteration's, at x26_out
  c->pass_seq = seq0 + (unsigned long long)executed;
  if (io->reason == CH_R_FAILED) {
    c->prev.valid = 0;
    (void)hipStreamSynchronize(c->stream);
    (void)hipMemsetAsync(c->d_ticket, 0, (FIT_GROUPS + 2) * sizeof(unsigned int), c->stream);
    (void)hipStreamSynchronize(c->stream);
    return fail(c, FLIMO_ERR_HIP, "a pass of the update chain did not publish its sums");
  }
  PoseMats Pl;
  pose_from_x26(io->x26_out, Pl);
  if (c->prune && executed > 0) { memcpy(c->prev.RT, Pl.RT, sizeof(c->prev.RT)); c->prev.valid = 1; }
  c->last_nq = (int)nq;
  c->last_P = Pl; c->last_mp = mp; c->last_n_all = n_all;
  c->async_deskews = 0;
  // timing (level 1): the launches' own begin / end stamps
  for (int i = 0; i < executed; i++) {
    if (!plan[i].timed) continue;
    hipEvent_t* ev = c->chain_ev[i];
    auto elapsed = [&](hipEvent_t a, hipEvent_t b) {
      float m = 0.f;
      if (hipEventElapsedTime(&m, a, b) != hipSuccess) { (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&m, a, b); }
      return m;
    };
    const float ms = elapsed(ev[0], ev[1]);
    c->last_knn_ms = ms; c->tot_knn_ms += ms;
    if (plan[i].fused) { c->split_fused_ms += ms; c->split_fused_n++; c->last_fit_ms = c->last_widen_ms = 0.f; }
    else {
      const float ms2 = elapsed(ev[2], ev[3]), msw = elapsed(ev[6], ev[7]);
      c->split_knn_ms += ms; c->split_fit_ms += ms2; c->split_widen_ms += msw; c->split_sep_n++; c->tot_fit_ms += ms2; c->tot_widen_ms += msw;
      c->last_fit_ms = ms2; c->last_widen_ms = msw;
    }
    c->chain_alg_ms += elapsed(ev[4], ev[5]); c->chain_alg_n++;
    c->tot_passes++; c->tot_queries += n_all;
  }
  return FLIMO_OK;
}

// The fast pass keeps the per-point records on chip; the fetch entry points re-run the fit kernel in
// record mode for the same pass when somebody asks for them (debug, or the M < 23 branch).
static int materialize_recs(flimo_ctx* c, bool need_dbg) {
  if (c->last_nq == 0) return FLIMO_OK;
  if (c->recs_valid && (!need_dbg || c->dbg_valid)) return FLIMO_OK;
  launch_fit(c->stream, c->grid, c->d_scan_sorted, c->last_n_all, c->d_nbr, c->last_P, c->last_mp, c->d_fit_partials,
             c->d_recs, need_dbg ? c->d_dbg : nullptr, c->d_out256, c->d_ticket, c->d_wl_count, 0ull);
  const flimo_match_cfg& cfg = c->last_cfg;
  if (cfg.MAX_NUM_MATCHES >= 0 && cfg.MAX_NUM_MATCHES < c->last_nq) launch_cap(c->stream, c->d_recs, c->last_nq, cfg.MAX_NUM_MATCHES);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->recs_valid = true;
  c->dbg_valid = c->dbg_valid || need_dbg;
  return FLIMO_OK;
}

extern "C" int flimo_match_fetch(flimo_ctx* c, flimo_match_rec* out, size_t cap, size_t* n) {
  if (!c || !n) return FLIMO_ERR_INVALID;
  *n = (size_t)c->last_nq;
  if (!out || cap == 0 || c->last_nq == 0) return FLIMO_OK;
  ctx_enter(c);
  { int rc = materialize_recs(c, true); if (rc) return rc; }
  const size_t m = std::min(cap, (size_t)c->last_nq);
  std::vector<Rec16> r(m);
  std::vector<RecDbg> d(m);
  HIPCHK(c, hipMemcpyAsync(r.data(), c->d_recs, m * sizeof(Rec16), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(d.data(), c->d_dbg, m * sizeof(RecDbg), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t i = 0; i < m; i++) {
    flimo_match_rec& o = out[i];
    memset(&o, 0, sizeof(o));
    for (int k = 0; k < 12; k++) o.H[k] = r[i].v[k];
    o.h = r[i].v[12];
    o.valid = r[i].v[13];
    {
      for (int k = 0; k < 4; k++) o.n[k] = d[i].n[k];
      for (int k = 0; k < 3; k++) o.p_global[k] = d[i].p_global[k];
      for (int k = 0; k < 5; k++) { o.sqd[k] = d[i].sqd[k]; o.nbr[k] = d[i].nbr[k]; }
      o.n_nbr = d[i].n_nbr;
    }
  }
  return FLIMO_OK;
}

extern "C" int flimo_match_fetch_H(flimo_ctx* c, double* H, double* h, size_t cap_rows, size_t* M) {
  if (!c || !M) return FLIMO_ERR_INVALID;
  *M = 0;
  if (c->last_nq == 0) return FLIMO_OK;
  ctx_enter(c);
  { int rc = materialize_recs(c, false); if (rc) return rc; }
  const size_t m = (size_t)c->last_nq;
  std::vector<Rec16> r(m);
  HIPCHK(c, hipMemcpyAsync(r.data(), c->d_recs, m * sizeof(Rec16), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  size_t k = 0;
  for (size_t i = 0; i < m; i++) {
    if (r[i].v[13] == 0.f) continue;
    if (H && h && k < cap_rows) {
      for (int j = 0; j < 12; j++) H[k * 12 + j] = (double)r[i].v[j];
      h[k] = (double)r[i].v[12];
    }
    k++;
  }
  *M = k;
  return FLIMO_OK;
}

// ---- path exit --------------------------------------------------------------------------------
extern "C" int flimo_scan_to_world(flimo_ctx* c, const double x26[26], float* out, size_t cap) {
  if (!c || !x26) return FLIMO_ERR_INVALID;
  if (c->scan_n == 0) return FLIMO_OK;
  ctx_enter(c);
  PoseMats P;
  pose_from_x26(x26, P);
  { const int rcf = flush_deskew(c); if (rcf) return rcf; }
  launch_transform(c->stream, c->d_scan, (int)c->scan_n, P, c->d_scan_world);
  HIPCHK(c, hipGetLastError());
  if (out && cap) return download_xyz(c, c->d_scan_world, std::min(cap, c->scan_n), out);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return FLIMO_OK;
}

extern "C" int flimo_scan_clouds(flimo_ctx* c, const double x26[26], const float** body, const float** world, size_t* n) {
  if (!c || !x26 || !body || !world || !n) return FLIMO_ERR_INVALID;
  *body = *world = nullptr;
  *n = c->scan_n;
  if (c->scan_n == 0) return FLIMO_OK;
  ctx_enter(c);
  const size_t bytes = c->scan_n * sizeof(float4);
  if (2 * bytes > c->clouds_cap) {
    if (c->h_clouds) (void)hipHostFree(c->h_clouds);
    c->h_clouds = nullptr; c->clouds_cap = 0;
    const size_t cap = 2 * bytes + bytes / 2 + 4096;
    HIPCHK(c, hipHostMalloc(&c->h_clouds, cap, hipHostMallocDefault));
    c->clouds_cap = cap;
  }
  PoseMats P;
  pose_from_x26(x26, P);
  { const int rcf = flush_deskew(c); if (rcf) return rcf; }
  HIPCHK(c, hipMemcpyAsync(c->h_clouds, c->d_scan, bytes, hipMemcpyDeviceToHost, c->stream));
  launch_transform(c->stream, c->d_scan, (int)c->scan_n, P, c->d_scan_world);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync((char*)c->h_clouds + bytes, c->d_scan_world, bytes, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *body = (const float*)c->h_clouds;
  *world = (const float*)((const char*)c->h_clouds + bytes);
  return FLIMO_OK;
}

extern "C" int flimo_map_add_scan(flimo_ctx* c, const double x26[26], double stamp) {
  if (!c || !x26) return FLIMO_ERR_INVALID;
  if (c->scan_n == 0) return FLIMO_OK;
  for (int a = 0; a < 3; a++) c->fine_center[a] = (float)x26[a];       // the second level follows the sensor (update_fine_grid)
  c->have_fine_center = true;
  // resident path: transform, decide, append and re-index on the device
  ctx_enter(c);
  PoseMats P;
  pose_from_x26(x26, P);
  { const int rcf = flush_deskew(c); if (rcf) return rcf; }
  launch_transform(c->stream, c->d_scan, (int)c->scan_n, P, c->d_scan_world);      // no host wait: the insert's first read-back follows
  HIPCHK(c, hipGetLastError());
  return map_add_device(c, c->d_scan_world, c->scan_n, stamp);
}

// ---- host-side evaluation of the SAME plane routines the fit kernel runs (flimo_math.h is __host__ __device__):
//      backs the fast_limo::Plane object of the host C++ mirror; not used on the hot path -----------------------
extern "C" void flimo_plane_fit5_host(const float xyz[15], float n_out[4]) {
  float px[5], py[5], pz[5];
  for (int j = 0; j < 5; j++) { px[j] = xyz[3 * j]; py[j] = xyz[3 * j + 1]; pz[j] = xyz[3 * j + 2]; }
  float n[4];
  flimo::plane_fit5(px, py, pz, n);
  for (int i = 0; i < 4; i++) n_out[i] = n[i];
}
extern "C" int flimo_plane_eval5_host(const float n_in[4], const float xyz[15], float threshold) {
  float px[5], py[5], pz[5], n[4];
  for (int j = 0; j < 5; j++) { px[j] = xyz[3 * j]; py[j] = xyz[3 * j + 1]; pz[j] = xyz[3 * j + 2]; }
  for (int i = 0; i < 4; i++) n[i] = n_in[i];
  return flimo::plane_eval5(n, px, py, pz, threshold) ? 1 : 0;
}

// Localizer::calculate_H (Localizer.cpp:537-577) on the host with the fit kernel's own row routine: H is M x 12
// row-major float64 (the float32 row values, widened), h[i] = -dist[i].
extern "C" int flimo_calculate_H_host(const double x26[26], const float* p_global, const float* n, const float* dist, size_t M,
                                      int estimate_extrinsics, double* H, double* h) {
  if (!x26 || (M && (!p_global || !n || !dist || !H || !h))) return FLIMO_ERR_INVALID;
  PoseMats P;
  pose_from_x26(x26, P);
  for (size_t i = 0; i < M; i++) {
    const float n4[4] = {n[4 * i], n[4 * i + 1], n[4 * i + 2], n[4 * i + 3]};
    float row[12];
    flimo::h_row(P, p_global[3 * i], p_global[3 * i + 1], p_global[3 * i + 2], n4, estimate_extrinsics, row);
    for (int j = 0; j < 12; j++) H[i * 12 + j] = (double)row[j];
    h[i] = (double)(-dist[i]);
  }
  return FLIMO_OK;
}

// ---- host-only replay of the reference's insert rule (no GPU needed) --------------------------
extern "C" int flimo_insert_rule_replay(float min_extent, int downsample, const float* xyz, const size_t* batch_sizes,
                                        size_t n_batches, unsigned char* keep, size_t* stored) {
  if (!xyz || !batch_sizes || !keep) return FLIMO_ERR_INVALID;
  InsertBook* b = insert_book_create();
  insert_book_config(b, min_extent, downsample != 0);
  size_t off = 0;
  for (size_t k = 0; k < n_batches; k++) {
    insert_book_update(b, xyz + 3 * off, batch_sizes[k], keep + off);
    off += batch_sizes[k];
  }
  if (stored) *stored = insert_book_size(b);
  insert_book_destroy(b);
  return FLIMO_OK;
}
