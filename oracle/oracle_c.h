/* oracle/oracle_c.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * C entry points of the CPU restatement ("oracle") of fast_LIMO's per-scan registration
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so.  PARITY UNPINNED: the reference ships no tests or golden vectors and cannot
 * be compiled in this image (Eigen3 / PCL / Boost absent), see DESIGN.md.
 */
#ifndef FLIMO_ORACLE_C_H
#define FLIMO_ORACLE_C_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors the hot-path subset of fast_limo::Config (reference Utils/Config.hpp:23-95). */
typedef struct oracle_cfg {
  int NUM_MATCH_POINTS, MAX_NUM_MATCHES, MAX_NUM_PC2MATCH;
  int bucket_size;            /* no effect, as in the reference (Octree.hpp:178-180) */
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
  int gravity_align, calibrate_accel, calibrate_gyro;
  double imu_calib_time;
  int voxel_active;
  float leaf_size;
  int sensor_type;            /* 0 OUSTER, 1 VELODYNE, 2 HESAI, 3 LIVOX (Common.hpp:82) */
  int crop_active;
  float crop_min[3], crop_max[3];
  int dist_active;
  double min_dist;
  int rate_active, rate_value;
  int fov_active;
  float fov_angle;
} oracle_cfg;

/* per scan-point record of Mapper::match (before compaction) */
typedef struct oracle_match_rec {
  float p_global[3];
  float n[4];
  float dist;
  int is_plane;
  int n_nbr;
  float nbr[5][3];
  float sqd[5];
} oracle_match_rec;

/* State::update (Objects/State.cpp:76-119): s = p[3] q[4](xyzw) v[3] g[3] w[3] a[3] bg[3] ba[3] (25 floats), propagated
 * from `time` to `t`; writes p, q, v back */
void   oracle_state_update(float s[25], double time, double t);

/* ---- octree (reference Objects/Octree.hpp) ---- */
void*  oracle_octree_create(float min_extent, int downsample);
void   oracle_octree_destroy(void* t);
void   oracle_octree_update(void* t, const float* xyz, size_t n);   /* initialize on first call */
size_t oracle_octree_size(void* t);
size_t oracle_octree_points(void* t, float* xyz_out, size_t cap);
/* batch knn: nbr [nq][k][3], sqd [nq][k], cnt [nq]; returns total leaf-point distance evaluations */
long long oracle_octree_knn(void* t, const float* q_xyz, size_t nq, int k, float* nbr, float* sqd, int* cnt,
                            int num_threads);

/* ---- plane fit (reference Objects/Plane.cpp) ---- */
void oracle_plane_fit(const float* nbr_xyz, const float* sqd, int n_nbr, int k, double max_dist_plane,
                      double plane_threshold, float n_out[4], int* is_plane);

/* ---- pcl::VoxelGrid restatement: in n x 3, out cap x 3; returns the number of voxels ---- */
size_t oracle_voxel_grid(const float* xyz, size_t n, float leaf, float* out, size_t cap);

/* ---- state helpers ---- */
/* x26 = pos3 rot(xyzw) offR(xyzw) offT3 vel3 bg3 ba3 grav3.  Outputs State(x).get_RT(), get_RT_inv(),
 * get_extr_RT_inv() row-major 4x4, and R_inv / I_R_L_inv (Localizer.cpp:554-555) row-major 3x3. */
void oracle_pose_mats(const double x26[26], float RT[16], float RT_inv[16], float TLI_inv[16], float R_inv[9],
                      float RLI_inv[9]);
void oracle_state_boxplus(double x26[26], const double dx[23]);
void oracle_state_boxminus(const double a26[26], const double b26[26], double out[23]);

/* ---- Mapper::match + calculate_H against a given octree ---- */
/* scan: n points, stride 3 floats.  recs (optional, n entries).  H [n][12], h [n] are filled for the
 * first *M rows (compacted, capped).  Returns evals. */
long long oracle_match_H(void* octree, const oracle_cfg* cfg, const double x26[26], const float* scan_xyz,
                         size_t n, oracle_match_rec* recs, double* H, double* h, int* M);

/* ---- full Localizer ---- */
void*  oracle_loc_create(const oracle_cfg* cfg);
void   oracle_loc_destroy(void* L);
void   oracle_loc_update_imu(void* L, double stamp, const float ang_vel[3], const float lin_accel[3]);
/* pts: n x 5 floats (x y z intensity time).  returns 0 ok / 1 null iteration / <0 early return */
int    oracle_loc_update_pointcloud(void* L, const float* pts5, size_t n, double stamp, int add_to_map);
/* points in the reference's 32-byte PointType layout: float x,y,z,w; float intensity; 4 bytes pad; 8-byte time union */
int    oracle_loc_update_pointcloud_points(void* L, const void* pts32, size_t n, double stamp, int add_to_map);
void   oracle_loc_map_add(void* L, const float* xyz, size_t n, double stamp);
size_t oracle_loc_map_size(void* L);
void   oracle_loc_get_x(void* L, double x26[26]);
void   oracle_loc_set_x(void* L, const double x26[26]);
void   oracle_loc_get_P(void* L, double P[23 * 23]);
void   oracle_loc_set_P(void* L, const double P[23 * 23]);
int    oracle_loc_num_iters(void* L);
/* per-pass log: M, HTH[144], HTh[12], dx[23], x_after[26] */
void   oracle_loc_get_iter(void* L, int i, int* M, double* HTH, double* HTh, double* dx, double* x_after);
size_t oracle_loc_get_pc2match(void* L, float* xyz_out, size_t cap);
size_t oracle_loc_get_final_scan(void* L, float* xyz_out, size_t cap);
/* timings of the last updatePointCloud: deskew (incl. sort), update, map add, time sort alone [s];
 * evals/queries of the last match */
void   oracle_loc_get_stats(void* L, double t[6], long long* evals, long long* queries);
/* deskew only (Localizer.cpp:733-853); out n x 3 (body frame at scan end); returns count or -1 */
long long oracle_loc_deskew(void* L, const float* pts5, size_t n, double stamp, float* out_xyz);
/* run only the IESKF update (no deskew): pc2match := xyz (n x 3) */
int    oracle_loc_update_only(void* L, const float* xyz, size_t n);

/* ---- IESKF algebra in isolation (esekfom.hpp:1620-1823) with a caller-supplied, state-independent
 * measurement (H [M][12], h [M]) -- used to cross-check the manifold algebra against numpy. ---- */
void oracle_eskf_update_fixed(double x26[26], double P[23 * 23], const double* H, const double* h, int M,
                              int max_iters, const double limits[23], double R, double D, int* n_passes);
void oracle_eigen_solver6(const double A[36], double wr[6], double wi[6], double V[36]);
void oracle_eskf_predict(double x26[26], double P[23 * 23], double dt, const double Qdiag[12], const double acc[3],
                         const double gyro[3]);

#ifdef __cplusplus
}
#endif
#endif
