/* include/flimo_c.h -- C ABI of the MI355X-native fast_LIMO registration hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each entry point
 * names the reference interface it replaces (paths relative to the fast_LIMO tree,
 * include/fast_limo/... unless stated).  INTEGRATION.md shows how the reference's
 * Mapper / Localizer / h_share_model call these.
 *
 * Conventions
 *   - every function returns FLIMO_OK (0) or a negative error code; flimo_last_error() gives text.
 *     Nothing throws or exits across this boundary (the reference prints and returns,
 *     Modules/Localizer.cpp:249-260,379-380).
 *   - one flimo_ctx owns one GPU's map, scan scratch and HIP stream; it is not re-entrant (the
 *     reference serialises the update under mtx_ikfom, Modules/Localizer.cpp:326-353).
 *   - the library fails loudly (FLIMO_ERR_NO_DEVICE) when no gfx950 device is present: there is
 *     no CPU fallback.
 *   - x26 = flat state_ikfom (IKFoM/use-ikfom.hpp:12-21):
 *       pos[3] rot(x,y,z,w) offset_R_L_I(x,y,z,w) offset_T_L_I[3] vel[3] bg[3] ba[3] grav[3]
 */
#ifndef FLIMO_C_H
#define FLIMO_C_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FLIMO_OK 0
#define FLIMO_ERR_NO_DEVICE (-1)
#define FLIMO_ERR_INVALID (-2)
#define FLIMO_ERR_HIP (-3)
#define FLIMO_ERR_NOMAP (-4)
#define FLIMO_ERR_TOO_LARGE (-5)
#define FLIMO_ERR_UNSUPPORTED (-6)
#define FLIMO_ERR_TIMEOUT (-7)      /* a pass did not publish its result within the wait bound (flimo_set_wait_timeout_ms) */

typedef struct flimo_ctx flimo_ctx;

/* Config::iKFoM::Mapping::Octree (Utils/Config.hpp:64-68) + the GPU grid knob. */
typedef struct flimo_map_cfg {
  float min_extent;  /* Octree/min_extent (default 0.2) */
  int bucket_size;   /* accepted and ignored: the reference setter is a no-op, effective 32
                        (Objects/Octree.hpp:155,178-180) */
  int downsample;    /* Octree/downsampling */
  float cell_size;   /* GPU hash-grid cell edge [m]; <= 0 selects the default (0.5 m) */
} flimo_map_cfg;

/* Config::iKFoM::Mapping (Utils/Config.hpp:58-63) + ikfom.estimate_extrinsics */
typedef struct flimo_match_cfg {
  int NUM_MATCH_POINTS;    /* k in 3..8; 5 (every shipped configuration) runs the specialised kernels, other values a general pass */
  int MAX_NUM_MATCHES;     /* Modules/Localizer.cpp:539 */
  int MAX_NUM_PC2MATCH;    /* Modules/Mapper.cpp:63 */
  double MAX_DIST_PLANE;   /* Objects/Plane.cpp:47 (compared with a SQUARED distance) */
  double PLANE_THRESHOLD;  /* Objects/Plane.cpp:73,110 */
  int estimate_extrinsics; /* Modules/Localizer.cpp:569 */
} flimo_match_cfg;

/* One per scan point: what Mapper::match_plane produced (debug / parity surface; replaces
 * Localizer::get_matches(), Modules/Localizer.cpp:139-141,575-576). */
typedef struct flimo_match_rec {
  float H[12];       /* calculate_H row [n, A, B, C] (B, C zero unless estimate_extrinsics) */
  float h;           /* -dist */
  float valid;       /* 1.0f if the plane passed all gates (Match::lisanAlGaib) else 0.0f */
  float n[4];        /* plane n_ABCD */
  float p_global[3]; /* scan point in the world frame */
  float sqd[5];      /* ascending squared distances of the 5 neighbours */
  int32_t nbr[5];    /* indices into the map's insertion order (see flimo_map_points) or -1 */
  int32_t n_nbr;
} flimo_match_rec;

/* IMU frame handed to the deskew kernel: fast_limo::State (Objects/State.hpp:22-48) */
typedef struct flimo_frame {
  float p[3], q[4] /* x,y,z,w */, v[3], g[3], w[3], a[3], bg[3], ba[3];
  double time;
} flimo_frame;

/* ---- context ---- */
int flimo_ctx_create(int device, flimo_ctx** out);
void flimo_ctx_destroy(flimo_ctx* ctx);
const char* flimo_last_error(const flimo_ctx* ctx);
const char* flimo_version(void);

/* ---- map: replaces fast_limo::Mapper::{set_config,add,exists,size} (Modules/Mapper.cpp:38-57,88-96)
 *      and octree::Octree::{initialize,update} (Objects/Octree.hpp:282-432) ---- */
int flimo_map_config(flimo_ctx* ctx, const flimo_map_cfg* cfg);
/* xyz: n points, stride_bytes between consecutive points (>= 12), host memory.  NaN points are
 * dropped (Octree::processPoints, Objects/Octree.hpp:243-244). */
int flimo_map_add(flimo_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, double stamp);
int flimo_map_clear(flimo_ctx* ctx);
size_t flimo_map_size(const flimo_ctx* ctx);
double flimo_map_last_time(const flimo_ctx* ctx);
/* copies the stored points (insertion order: what neighbour indices refer to) as packed xyz; *n receives the total count
 * (Octree::getData, Objects/Octree.hpp:198-215) */
int flimo_map_points(flimo_ctx* ctx, float* xyz_out, size_t cap, size_t* n);

/* ---- exact k-NN: replaces octree::Octree::knn (Objects/Octree.hpp:526-555) for a batch ----
 * q_xyz packed [nq][3] host; outputs host: idx [nq][k] (the map's insertion order, -1 padded),
 * sqd [nq][k] ascending squared distances (0 padded), cnt [nq].  k <= 5.  No gate: like Octree::knn the call answers from
 * anywhere -- rings of cells near the map, then a best-first search over the index's tiles (nearest tile first, until the next
 * one is farther than the k-th best): a query kilometres from every point costs a look at the tile directory, not at the
 * empty space in between. */
int flimo_knn(flimo_ctx* ctx, const float* q_xyz, size_t nq, int k, int32_t* idx, float* sqd, int32_t* cnt);

/* ---- scan: pc2match of the reference (Modules/Localizer.hpp:36) ---- */
int flimo_scan_set(flimo_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes);
size_t flimo_scan_size(const flimo_ctx* ctx);
/* copy the resident scan back (packed xyz) */
int flimo_scan_get(flimo_ctx* ctx, float* xyz_out, size_t cap, size_t* n);

/* ---- voxel-grid filter on the resident scan: replaces pcl::VoxelGrid in Localizer::updatePointCloud
 *      (Modules/Localizer.cpp:313-321): centroid per occupied voxel, ascending voxel index ---- */
int flimo_scan_voxel_filter(flimo_ctx* ctx, float leaf_size, size_t* n_out);

/* ---- deskew: replaces the OpenMP loop of Localizer::deskewPointCloud
 *      (Modules/Localizer.cpp:820-843) incl. State::update (Objects/State.cpp:76-119) and
 *      binary_search_tailored (Utils/Algorithms.hpp:25-38).  Input: time-sorted LiDAR-frame points
 *      (xyz, stride) with per-point absolute times t[i] (= extract_point_time + offset, double);
 *      frames sorted by time; lidar2baselink_T row-major 4x4; last_x26 = _iKFoM.get_x().
 *      The result (body frame at scan end) becomes the resident scan (pc2match). ---- */
int flimo_deskew(flimo_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, const double* t,
                 const flimo_frame* frames, size_t n_frames, const float lidar2baselink_T[16],
                 const double last_x26[26]);
/* the same in two steps, so that the raw scan can be made resident in HBM ahead of time */
int flimo_raw_scan_set(flimo_ctx* ctx, const float* xyz, size_t n, size_t stride_bytes, const double* t);
int flimo_deskew_resident(flimo_ctx* ctx, const flimo_frame* frames, size_t n_frames,
                          const float lidar2baselink_T[16], const double last_x26[26]);

/* ---- input filters + stamps of a raw sweep on the GPU: replaces removeNaNFromPointCloud, the negative CropBox and the rate /
 *      min-distance filters of Localizer::updatePointCloud (Modules/Localizer.cpp:262-302) and the per-point stamp of
 *      deskewPointCloud (:741-805) for sweeps that may reach the GPU in arrival order.  points32: n records in the reference's
 *      32-byte PointType layout (Common.hpp:100-113), host memory.  The kept points (order preserved) become the resident raw
 *      scan; *n_kept their number; *last_stamp = stamp (without the sweep offset) of the point the reference's time sort puts
 *      last; *nan_stamp = 1 when a kept stamp is NaN (the caller then takes the host path).  The FoV filter (:873-876) runs here
 *      too (fov_active): atan2 of two floats as glibc's fdlibm routine evaluates it, checked once per context against THIS host's
 *      libm on a set of argument pairs (a host whose atan2f rounds differently declines the filter: the call then reports
 *      FLIMO_ERR_UNSUPPORTED and the caller filters on the host).  Follow with flimo_deskew_resident_offset. ---- */
typedef struct flimo_filter_cfg {
  int crop_active;  float crop_min[3], crop_max[3];
  int dist_active;  float min_dist;
  int rate_active;  int rate_value;
  int time_kind;    /* 0 OUSTER (uint32 t, ns), 1 VELODYNE (float time, s), 2 HESAI (double timestamp, s), 3 LIVOX (double, ns) */
  int end_of_sweep;
  double sweep_ref_time;
  int fov_active;   float fov_angle;   /* FoV filter (Localizer.cpp:873-876): fabs(atan2(y, x)) < fov_angle, atan2 as the host's libm rounds it */
} flimo_filter_cfg;
int flimo_raw_scan_filter_set(flimo_ctx* ctx, const void* points32, size_t n, const flimo_filter_cfg* cfg, size_t* n_kept,
                              double* last_stamp, int* nan_stamp);
/* Optional: the context's pinned upload buffer (>= bytes).  A caller that copies the sweep's records into it -- from several threads
 * if it likes -- and passes THAT pointer as points32 to the next flimo_raw_scan_filter_set / _order_set saves the call's own
 * single-threaded staging copy (pcl clouds live in pageable memory).  Valid until the next call on the context that stages data. */
int flimo_upload_stage(flimo_ctx* ctx, size_t bytes, void** host_ptr);
/* The same, and with time_order != 0 the kept points are put into the order of the reference's time sort (std::partial_sort_copy by
 * stamp, Localizer.cpp:789-790) on the device: the order MAX_NUM_PC2MATCH / MAX_NUM_MATCHES ("the first N of pc2match") and the
 * voxel grid's float sums are defined in.  That order is unique -- a stable radix sort gives it -- when no two kept stamps are
 * equal; with equal stamps it is the library's heap moves', *tied = 1 is returned, nothing is made resident, and the caller takes
 * the host routine.  flimo_raw_scan_order: time rank -> position among the kept points (for the clouds handed back to callers).
 * time_order bit 1 (value 2): the sweep is NOT put into the spatial order the per-pass kernels like -- for a caller that runs
 * flimo_scan_voxel_filter right after the deskew, which re-orders the scan anyway.
 * time_order bit 2 (value 4): points32 holds 16-byte records {float x, y, z; 32-bit time word (PointType offset 24: OUSTER's uint32 t,
 * VELODYNE's float time)} instead of PointType records: time_kind 0 or 1 only.  A caller that stages the upload itself
 * (flimo_upload_stage) packs the sweep while it copies it and halves the bytes over PCIe.
 * time_order bit 3 (value 8): equal stamps keep their ARRIVAL order (the radix sort is stable) and the sweep is made resident all
 * the same; *tied = 1 still says that there were some.  Among equal stamps the reference's order is whatever its heap sort leaves
 * -- observable only as ulp-level voxel centroids and as which of several equally stamped points a cap cuts off. */
int flimo_raw_scan_filter_order_set(flimo_ctx* ctx, const void* points32, size_t n, const flimo_filter_cfg* cfg, int time_order,
                                    size_t* n_kept, double* last_stamp, int* nan_stamp, int* tied);
int flimo_raw_scan_order(flimo_ctx* ctx, uint32_t* order_out, size_t cap, size_t* n);
/* The resident raw sweep of `src` (what flimo_raw_scan_filter_set / _order_set left there) becomes `dst`'s: the two contexts exchange
 * the buffers of the Morton-ordered points and stamps (no copy) and dst's stream is ordered behind src's queued work.  Both contexts
 * on one GPU.  The input stage of a sweep -- upload, filters, stamps, time order (Localizer.cpp:262-302,741-805) -- does not read
 * the map: a caller runs it on a context of its own while dst's stream still carries the previous sweep's Mapper::add, then hands
 * the sweep over.  src keeps the time order (flimo_raw_scan_order) and no sweep. */
int flimo_scan_adopt(flimo_ctx* dst, flimo_ctx* src);
/* flimo_deskew_resident with the sweep's time offset added to every resident stamp (Localizer.cpp:795-800) */
int flimo_deskew_resident_offset(flimo_ctx* ctx, const flimo_frame* frames, size_t n_frames, const float lidar2baselink_T[16],
                                 const double last_x26[26], double t_offset);

/* ---- one measurement pass: replaces IKFoM::h_share_model (IKFoM/use-ikfom.cpp:10-31) =
 *      Mapper::match (Modules/Mapper.cpp:59-86) + Localizer::calculate_H
 *      (Modules/Localizer.cpp:537-577) + the h_x^T h_x / h_x^T h products of
 *      esekf::update_iterated_dyn_share_modified (esekfom.hpp:1723,1727).
 *      HTH row-major 12x12, HTh 12, *M = number of matches used (after both caps). ---- */
int flimo_match_reduce(flimo_ctx* ctx, const double x26[26], const flimo_match_cfg* cfg, double HTH[144],
                       double HTh[12], int* M);
/* The same pass, with host work of the caller to do WHILE the GPU runs it: `while_in_flight(arg)` is called at most once, after
 * the pass's launches are queued and before the call waits for their result.  In esekf::update_iterated_dyn_share_modified the
 * part of an iteration that does not depend on the measurement (x boxminus x_propagated, the re-projection of P through the
 * manifold Jacobians, esekfom.hpp:1652-1697) is such work.  On paths that do not wait (errors, an empty map) the function is not
 * called: the caller checks and runs it itself. */
int flimo_match_reduce_overlap(flimo_ctx* ctx, const double x26[26], const flimo_match_cfg* cfg, double HTH[144], double HTh[12], int* M,
                               void (*while_in_flight)(void*), void* arg);
/* ---- the iterated update of the resident scan, enqueued at once: replaces the loop of esekf::update_iterated_dyn_share_modified
 *      (IKFoM_toolkit/esekfom/esekfom.hpp:1620-1823) up to the iteration whose covariance update is due.  Every outer iteration =
 *      the pass above (h_share_model) + the 23-dof algebra of :1652-1760, run INSIDE the pass's reducing launch: one extra workgroup
 *      does the half that does not depend on the measurement beside the pass, the workgroup that completes the launch goes on from
 *      the sums to the gain, the step, boxplus and the convergence test and leaves the next pass's pose in device memory; passes
 *      queued behind the end of the chain leave at once.  No host round trip and no launch between the passes.
 *      The loop always comes back to the caller (status FLIMO_CHAIN_HANDED_BACK) at iteration it_next with counter t, state x26_out:
 *        reason FLIMO_CHAIN_FINAL       that iteration ends the loop (:1764: converged twice, or the last one): its pass has run,
 *                                       meas_* hold its sums -- the caller runs :1652-1820 for it without another pass;
 *        reason FLIMO_CHAIN_DEGENERATE  H^T H needs the eigen-decomposition of :1736-1744 (or the solve met a zero pivot): meas_* valid;
 *        reason FLIMO_CHAIN_FEW         M < 23 (:1701-1709 needs the dense rows): the caller runs the iteration through
 *                                       flimo_match_reduce / flimo_match_fetch_H;
 *        reason FLIMO_CHAIN_TIES        exact float32 distance ties to settle the reference's way: likewise;
 *      or not at all: FLIMO_CHAIN_DECLINED, nothing was run (records / caps / debug / timing level 2 / NUM_MATCH_POINTS != 5 / gates
 *      wider than 3 rings / more than FLIMO_CHAIN_MAX_PASSES iterations / FLIMO_HOST_UPDATE=1): the caller runs its loop.
 *      The gain is the matrix-inversion-lemma form of :1722-1729 (one 12x12 solve), as in csrc/host/flimo_ikfom.cpp. ---- */
#define FLIMO_CHAIN_MAX_PASSES 12
#define FLIMO_CHAIN_DECLINED 0
#define FLIMO_CHAIN_HANDED_BACK 2
#define FLIMO_CHAIN_FEW 1
#define FLIMO_CHAIN_TIES 2
#define FLIMO_CHAIN_DEGENERATE 3
#define FLIMO_CHAIN_FINAL 5
typedef struct flimo_chain_pass {
  int M, stragglers, ties;       /* matches of the pass, queries that needed more than their 3x3x3 block, queries on an exact tie */
  double HTH[144], HTh[12];      /* want_log only: the pass's sums, */
  double dx[23], x_after[26];    /* the un-projected step dx_ (:1733) and the state after boxplus (:1747) */
} flimo_chain_pass;
typedef struct flimo_chain_io {
  /* in */
  double x26[26];                /* x_ at entry (= x_propagated) */
  double P[529];                 /* P_ at entry, 23 x 23 row-major */
  double limits[23];             /* esekf::limit (:1757-1763) */
  double R, D;                   /* measurement noise, degeneracy threshold (Localizer.cpp:333: 0.001, 5.0) */
  int max_iter;                  /* maximum_iter (MAX_NUM_ITERS): iterations -1 .. max_iter - 1 */
  int want_log;                  /* also fill log[].HTH / HTh / dx / x_after */
  /* out */
  int status, reason;
  int passes;                    /* outer iterations completed on the device */
  int it_next, t;                /* loop variables to resume with (it_next = -1 + passes) */
  double x26_out[26];            /* x_ at that iteration */
  int meas_valid, meas_M;        /* that iteration's pass: usable sums (reasons FINAL, DEGENERATE) */
  double meas_HTH[144], meas_HTh[12];
  flimo_chain_pass log[FLIMO_CHAIN_MAX_PASSES];   /* entries 0 .. passes - 1 (+ the handed-back iteration's M / stragglers / ties) */
} flimo_chain_io;
int flimo_update_chain(flimo_ctx* ctx, const flimo_match_cfg* cfg, flimo_chain_io* io);
/* Which way a caller's update runs: 0 (default) and 1 = flimo_update_chain declines and the caller's host loop runs the update pass
 * by pass; 2 = the chain runs.  The choice is the caller's, never a measurement's: the two layouts agree to 1e-15 per pass but not
 * bit for bit, so a choice made from timing (round 4 chose by the launch round trip measured at context creation) made the
 * filter's bits depend on the host.  A host whose launch -> result round trip (reported below) is beyond ~16 us gains from mode 2.
 * FLIMO_HOST_UPDATE=1 / 0 preset 1 / 2.
 * flimo_update_mode: *chained = 1 when flimo_update_chain will run, *launch_rtt_us = the measured round trip. */
/* Host loop, pipelined.  With the switch on, a flimo_match_reduce that ran a one-launch pass queues the NEXT pass of the same update
 * right behind it: a kernel whose workgroups are placed on the GPU when the current pass ends and wait there for their pose.  The next
 * flimo_match_reduce (same scan, same settings) stores the pose into device memory instead of launching: the doorbell -> dispatch ->
 * kernel start of a launch leave the iteration's critical path (4-5 us per pass).  What it asks of the caller: say when the update is
 * over (flimo_pass_pipeline_end, right after the loop of esekfom.hpp:1652-1820) -- a pass nobody asks for is also told to leave by the
 * next call on the context and gives up by itself after 50 ms, but until then a device-wide synchronisation anywhere in the process
 * waits for it.  Off by default for that reason; fast_limo::Localizer switches it on and makes the call.  FLIMO_PIPELINE=0/1 presets it.
 */
int flimo_set_pass_pipeline(flimo_ctx* ctx, int on);
int flimo_pass_pipeline_end(flimo_ctx* ctx);
/* Optional hint of the same loop: the NEXT flimo_match_reduce is the last pass its update can run (the loop's i == maximum_iter - 1,
 * esekfom.hpp:1634) -- nothing is queued behind it.  Without the hint one launch per update is queued for nothing and told to leave. */
int flimo_pass_pipeline_last(flimo_ctx* ctx);
int flimo_set_update_mode(flimo_ctx* ctx, int mode);
int flimo_update_mode(const flimo_ctx* ctx, int* chained, double* launch_rtt_us);
/* per-point records of the last flimo_match_reduce (first min(N, MAX_NUM_PC2MATCH) points) */
/* also write the per-point debug part of flimo_match_rec (plane, neighbours, candidate counts) */
int flimo_set_debug_records(flimo_ctx* ctx, int on);
int flimo_match_fetch(flimo_ctx* ctx, flimo_match_rec* out, size_t cap, size_t* n);
/* dense H (M x 12 row-major, compacted in scan order, capped) and h of the last pass: needed by the
 * M < 23 branch of the update (esekfom.hpp:1701-1709) */
int flimo_match_fetch_H(flimo_ctx* ctx, double* H, double* h, size_t cap_rows, size_t* M);

/* ---- path exit: pcl::transformPointCloud(pc2match, state.get_RT()) + Mapper::add
 *      (Modules/Localizer.cpp:361-377).  world_xyz_out may be NULL. ---- */
int flimo_scan_to_world(flimo_ctx* ctx, const double x26[26], float* world_xyz_out, size_t cap);
/* Both clouds the caller of Localizer::updatePointCloud may ask for (pc2match: body frame; final_scan: world frame of pose x26,
 * Localizer.cpp:361-371) in ONE round trip: packed float4 records (x, y, z, unused) in pinned memory owned by the context, valid
 * until the next flimo_scan_clouds on it.  *n = points in each. */
int flimo_scan_clouds(flimo_ctx* ctx, const double x26[26], const float** body_xyzw, const float** world_xyzw, size_t* n);
int flimo_map_add_scan(flimo_ctx* ctx, const double x26[26], double stamp);

/* Wall-clock bound (milliseconds, default 2000) of the wait for a pass's result inside flimo_match_reduce: the reference's
 * Mapper::match (Modules/Mapper.cpp:59-86) cannot hang, a GPU launch can -- when the bound expires the call returns
 * FLIMO_ERR_TIMEOUT (kernel still running) or FLIMO_ERR_HIP (stream idle, nothing published) instead of blocking its caller,
 * which holds the filter's mutex (Localizer.cpp:326-353). */
int flimo_set_wait_timeout_ms(flimo_ctx* ctx, int ms);

#ifdef __cplusplus
}
#endif
#endif /* FLIMO_C_H */
