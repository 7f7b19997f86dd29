/* include/flimo_localizer_c.h -- C wrapper over the host C++ library (libfast_limo.so), i.e. over
 * fast_limo::Localizer / fast_limo::Mapper with the reference's API (Modules/Localizer.hpp:138-209,
 * Modules/Mapper.hpp:48-71).  It exists so that non-C++ callers (the Python tests and bench.py, or a
 * ROS-free replay tool) can drive the same objects the ROS wrapper would (src/main.cpp:14-95).
 * The hot-path boundary itself is include/flimo_c.h. */
#ifndef FLIMO_LOCALIZER_C_H
#define FLIMO_LOCALIZER_C_H
#include <stddef.h>
#include "flimo_c.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct flimo_loc flimo_loc;

/* hot-path subset of fast_limo::Config (Utils/Config.hpp:23-95); defaults: src/main.cpp:101-168 */
typedef struct flimo_loc_cfg {
  int NUM_MATCH_POINTS, MAX_NUM_MATCHES, MAX_NUM_PC2MATCH;
  int bucket_size;
  double MAX_DIST_PLANE, PLANE_THRESHOLD;
  float min_extent;
  int downsampling;
  int MAX_NUM_ITERS;
  int estimate_extrinsics;
  double LIMITS[23];
  double cov_gyro, cov_acc, cov_bias_gyro, cov_bias_acc;
  int time_offset, end_of_sweep, num_threads;
  float imu2baselink_t[3], imu2baselink_R[9];
  float lidar2baselink_t[3], lidar2baselink_R[9];
  float accel_bias[3], gyro_bias[3], imu_sm[9];
  /* filters (Config::Filters) */
  int voxel_active; float leaf_size;
  int crop_active; float cropBoxMin[3], cropBoxMax[3];
  int dist_active; double min_dist;
  int rate_active; int rate_value;
  int fov_active; float fov_angle;
  int sensor_type;
  /* IMU stand-still calibration (Config flags, Modules/Localizer.cpp:411-509) */
  int gravity_align, calibrate_accel, calibrate_gyro;
  double imu_calib_time;
  /* MI355X additions */
  int gpu_device;
  float gpu_cell_size;
  int debug;                   /* Config::debug (`debug` in the config yaml): keep original_scan / deskewed / matches for the caller */
} flimo_loc_cfg;

int    flimo_loc_create(const flimo_loc_cfg* cfg, flimo_loc** out);   /* Localizer::init */
void   flimo_loc_destroy(flimo_loc* L);
/* The Mapper's GPU context.  The map insert that ends a scan (reference Localizer.cpp:361-377) runs on a worker thread
 * and may still be in flight when flimo_loc_update_pointcloud* returns; this call (like every flimo_loc_* call that
 * touches the map) waits for it first.  Fetch the handle again after each scan rather than caching it. */
flimo_ctx* flimo_loc_ctx(flimo_loc* L);
void   flimo_loc_sync(flimo_loc* L);                                  /* wait for a running map insert */
void   flimo_loc_set_async_insert(flimo_loc* L, int on);              /* default on; FLIMO_SYNC_INSERT=1 turns it off */
/* default on: when no cap can bind and the voxel grid is off, the order of std::partial_sort_copy (Localizer.cpp:789-790) is not
 * observable by the registration; the GPU then gets the sweep in arrival order and the permutation is only computed for the
 * clouds handed back to the caller (set_flags download_clouds), while the GPU works.  off: always sort first. */
void   flimo_loc_set_lazy_time_order(flimo_loc* L, int on);
/* default on: NaN removal, crop box, rate and min-distance filters and the per-point stamps run on the GPU
 * (flimo_raw_scan_filter_set) whenever the arrival-order path applies and no host copies of the clouds are requested */
void   flimo_loc_set_gpu_filters(flimo_loc* L, int on);
/* default off.  A sweep whose stamps are not pairwise different (every spinning sensor: all rings of a column share one) and whose
 * time order is observable (MAX_NUM_PC2MATCH / MAX_NUM_MATCHES can bind, or the voxel grid is on): off = the device's stable
 * order, arrival order among equal stamps -- observable only as ulp-level voxel centroids and as which of several equally stamped
 * points a cap cuts off; on = the order std::partial_sort_copy leaves them in (Localizer.cpp:789-790), reproduced move for move
 * by the host front end (bit-exact against the reference's library call, 1.5 ms per 64k-point sweep). */
void   flimo_loc_set_exact_tied_order(flimo_loc* L, int on);
int    flimo_loc_last_sweep_tied(const flimo_loc* L);      /* 1: the last sweep of the device front end had equal stamps */
/* how long updatePointCloud waits for the IMU stream to reach the end of the sweep (Localizer::propagatedFromTimeRange,
 * Localizer.cpp:855-871).  The reference waits on its condition variable without bound, and so does fast_limo::Localizer used
 * through its C++ header (seconds < 0).  Handles made by flimo_loc_create start with 1 s, because their callers usually feed IMU
 * and sweeps from ONE thread, where an unbounded wait could never be satisfied. */
void   flimo_loc_set_propagation_wait(flimo_loc* L, double seconds);
double flimo_loc_last_insert_seconds(flimo_loc* L);                   /* duration of the last insert (waits for it) */
int    flimo_loc_update_imu(flimo_loc* L, double stamp, const float ang_vel[3], const float lin_accel[3]);
/* n samples in arrival order, one updateIMU each (a binding whose per-call cost matters -- ctypes: 25 us -- hands over the samples
 * between two sweeps at once) */
int    flimo_loc_update_imu_n(flimo_loc* L, size_t n, const double* stamps, const float* ang_vel3, const float* lin_accel3);
/* A recorded drive replayed at full speed from native code: before sweep k every IMU sample with stamp <= imu_until[k] goes to
 * updateIMU, then the sweep (PointType records, as flimo_loc_update_pointcloud_points takes them) to updatePointCloud.
 * status_out[k]: that sweep's status; seconds_out[k] (or NULL): when its call returned, since the start of the replay. */
int    flimo_loc_replay(flimo_loc* L, size_t n_sweeps, const void* const* sweeps32, const size_t* n_points, const double* sweep_stamps,
                        const double* imu_until, size_t n_imu, const double* imu_stamps, const float* ang_vel3, const float* lin_accel3,
                        int* status_out, double* seconds_out);
/* pts5: n x (x y z intensity time[s since sweep reference]).  Returns Localizer status:
 * 0 ok, 1 null iteration, <0 early return */
int    flimo_loc_update_pointcloud(flimo_loc* L, const float* pts5, size_t n, double stamp);
/* The same with points in the reference's PointType layout (Common.hpp:100-113): float x, y, z, w; float intensity;
 * 4 bytes of padding; 8-byte time union {uint32 t (OUSTER ns) | float time (VELODYNE s) | double timestamp
 * (HESAI s, LIVOX ns)} -- the view that is read follows flimo_loc_cfg.sensor_type. */
int    flimo_loc_update_pointcloud_points(flimo_loc* L, const void* points32, size_t n, double stamp);
int    flimo_loc_map_add(flimo_loc* L, const float* xyz, size_t n, double stamp);   /* Mapper::add */
size_t flimo_loc_map_size(flimo_loc* L);
void   flimo_loc_get_x(flimo_loc* L, double x26[26]);
void   flimo_loc_set_x(flimo_loc* L, const double x26[26]);
void   flimo_loc_get_P(flimo_loc* L, double P[529]);
void   flimo_loc_set_P(flimo_loc* L, const double P[529]);
void   flimo_loc_set_flags(flimo_loc* L, int add_to_map, int download_clouds, int keep_log);
int    flimo_loc_num_passes(flimo_loc* L);
void   flimo_loc_get_pass(flimo_loc* L, int i, int* M, double* HTH, double* HTh, double* dx, double* x_after);
size_t flimo_loc_get_pc2match(flimo_loc* L, float* xyz_out, size_t cap);
size_t flimo_loc_get_final_scan(flimo_loc* L, float* xyz_out, size_t cap);
void   flimo_loc_get_stage_times(flimo_loc* L, double t[4]);
void   flimo_loc_get_pose_cov(flimo_loc* L, double cov36[36]);       /* getPoseCovariance */
/* host-side profile of register_resident: seconds in deskew call, whole update, flimo_match_reduce; passes */
void   flimo_loc_host_profile(flimo_loc* L, double out[4], int reset);
/* benchmark step: restore the prior (x26, P) and re-register the resident raw scan
 * (GPU deskew + iterated update) */
int    flimo_loc_register_resident(flimo_loc* L, const double x26_prior[26], const double P_prior[529]);
/* fast_limo::State::update (State.cpp:76-119) on a flat state p3 q4(xyzw) v3 g3 w3 a3 bg3 ba3 -- for unit tests */
void   flimo_host_state_update(float s[25], double time, double t);
/* fast_limo::Plane + Match object API in isolation (Plane.cpp:23-31, Match.cpp:23-28): returns good_fit(), the normal
 * (zeros when not a plane) and Match(p_global, ., plane).dist -- for unit tests */
/* Order in which Localizer::deskewPointCloud's std::partial_sort_copy (Localizer.cpp:789-790) leaves a sweep, ties included.
 * kind: 0 uint32 (OUSTER), 1 float (VELODYNE), 2 double (HESAI/LIVOX); use_library=1 runs the library call itself (tests). */
int    flimo_host_time_order(const void* keys, int kind, size_t n, int descending, int use_library, uint32_t* order_out);
int    flimo_host_plane(const float* xyz, const float* sqd, int n, int num_match_points, double max_dist_plane,
                        double plane_threshold, const float p_global[3], float n_out[4], float* dist_out);
/* IESKF algebra in isolation with a fixed measurement (H [M][12], h [M]) -- for unit tests */
int    flimo_eskf_update_fixed(double x26[26], double P[529], const double* H, const double* h, int M, int max_iters,
                               const double limits[23], double R, double D, int* n_passes);
int    flimo_eskf_predict(double x26[26], double P[529], double dt, const double Qdiag[12], const double acc[3],
                          const double gyro[3]);
/* The filter's restatement of Eigen::EigenSolver<Matrix<double,6,6>> (IKFoM_toolkit/esekfom/esekfom.hpp:1736-1738, degeneracy
 * handling): A row-major; eigenvalues in the solver's order (real, imaginary part), real parts of the normalised eigenvectors as
 * the columns of V (row-major) -- for unit tests */
void   flimo_host_eigen_solver6(const double A[36], double wr[6], double wi[6], double V[36]);

#ifdef __cplusplus
}
#endif
#endif
