// fast_limo_amd/csrc/host/capi_host.cpp -- C wrapper (include/flimo_localizer_c.h) over the host
// C++ Localizer / Mapper.
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include "../../../include/flimo_localizer_c.h"
#include "fast_limo/Modules/Localizer.hpp"
#include "fast_limo/Objects/Plane.hpp"
#include "flimo_ikfom.hpp"

using namespace fast_limo;

struct flimo_loc {
  std::unique_ptr<Mapper> map;
  std::unique_ptr<Localizer> loc;
  pcl::PointCloud<PointType>::Ptr in_pc;     // the cloud the last sweep came in with: its storage is reused when nobody kept it
};

static Config to_config(const flimo_loc_cfg* c) {
  Config cfg = Config::defaults();
  cfg.ikfom.mapping.NUM_MATCH_POINTS = c->NUM_MATCH_POINTS;
  cfg.ikfom.mapping.MAX_NUM_MATCHES = c->MAX_NUM_MATCHES;
  cfg.ikfom.mapping.MAX_NUM_PC2MATCH = c->MAX_NUM_PC2MATCH;
  cfg.ikfom.mapping.MAX_DIST_PLANE = c->MAX_DIST_PLANE;
  cfg.ikfom.mapping.PLANE_THRESHOLD = c->PLANE_THRESHOLD;
  cfg.ikfom.mapping.octree.bucket_size = c->bucket_size;
  cfg.ikfom.mapping.octree.min_extent = c->min_extent;
  cfg.ikfom.mapping.octree.downsampling = c->downsampling != 0;
  cfg.ikfom.MAX_NUM_ITERS = c->MAX_NUM_ITERS;
  cfg.ikfom.estimate_extrinsics = c->estimate_extrinsics != 0;
  cfg.ikfom.LIMITS.assign(c->LIMITS, c->LIMITS + 23);
  cfg.ikfom.cov_gyro = c->cov_gyro; cfg.ikfom.cov_acc = c->cov_acc;
  cfg.ikfom.cov_bias_gyro = c->cov_bias_gyro; cfg.ikfom.cov_bias_acc = c->cov_bias_acc;
  cfg.time_offset = c->time_offset != 0;
  cfg.end_of_sweep = c->end_of_sweep != 0;
  cfg.num_threads = c->num_threads;
  cfg.extrinsics.imu2baselink_t.assign(c->imu2baselink_t, c->imu2baselink_t + 3);
  cfg.extrinsics.imu2baselink_R.assign(c->imu2baselink_R, c->imu2baselink_R + 9);
  cfg.extrinsics.lidar2baselink_t.assign(c->lidar2baselink_t, c->lidar2baselink_t + 3);
  cfg.extrinsics.lidar2baselink_R.assign(c->lidar2baselink_R, c->lidar2baselink_R + 9);
  cfg.intrinsics.accel_bias.assign(c->accel_bias, c->accel_bias + 3);
  cfg.intrinsics.gyro_bias.assign(c->gyro_bias, c->gyro_bias + 3);
  cfg.intrinsics.imu_sm.assign(c->imu_sm, c->imu_sm + 9);
  cfg.filters.crop_active = c->crop_active != 0;
  cfg.filters.cropBoxMin.assign(c->cropBoxMin, c->cropBoxMin + 3);
  cfg.filters.cropBoxMax.assign(c->cropBoxMax, c->cropBoxMax + 3);
  cfg.filters.voxel_active = c->voxel_active != 0;
  cfg.filters.leafSize = {c->leaf_size, c->leaf_size, c->leaf_size};
  cfg.filters.dist_active = c->dist_active != 0;
  cfg.filters.min_dist = c->min_dist;
  cfg.filters.rate_active = c->rate_active != 0;
  cfg.filters.rate_value = c->rate_value;
  cfg.filters.fov_active = c->fov_active != 0;
  cfg.filters.fov_angle = c->fov_angle;
  cfg.sensor_type = c->sensor_type;
  cfg.gravity_align = c->gravity_align != 0;
  cfg.calibrate_accel = c->calibrate_accel != 0;
  cfg.calibrate_gyro = c->calibrate_gyro != 0;
  cfg.imu_calib_time = c->imu_calib_time;
  cfg.debug = c->debug != 0;
  cfg.verbose = false;
  cfg.gpu_device = c->gpu_device;
  cfg.gpu_cell_size = c->gpu_cell_size;
  return cfg;
}

extern "C" {

int flimo_loc_create(const flimo_loc_cfg* cfg, flimo_loc** out) {
  if (!cfg || !out) return FLIMO_ERR_INVALID;
  *out = nullptr;
  std::unique_ptr<flimo_loc> L(new flimo_loc());
  L->map.reset(new Mapper(cfg->gpu_device));
  L->loc.reset(new Localizer(L->map.get()));
  L->loc->propagation_wait_s = 1.0;            // single-threaded callers (flimo_localizer_c.h); the C++ class itself waits like the reference
  Config c = to_config(cfg);
  L->loc->init(c);
  L->loc->filter().reference_solve = getenv("FLIMO_REFERENCE_SOLVE") != nullptr;
  if (!L->map->ctx()) return FLIMO_ERR_NO_DEVICE;      // loud failure: no CPU fallback
  *out = L.release();
  return FLIMO_OK;
}
void flimo_loc_destroy(flimo_loc* L) { delete L; }
flimo_ctx* flimo_loc_ctx(flimo_loc* L) { return L ? L->map->ctx() : nullptr; }

int flimo_loc_update_imu(flimo_loc* L, double stamp, const float w[3], const float a[3]) {
  if (!L) return FLIMO_ERR_INVALID;
  IMUmeas m;
  m.stamp = stamp;
  m.dt = 0.0;
  m.ang_vel = Eigen::Vector3f(w[0], w[1], w[2]);
  m.lin_accel = Eigen::Vector3f(a[0], a[1], a[2]);
  L->loc->updateIMU(m);
  return FLIMO_OK;
}
int flimo_loc_update_imu_n(flimo_loc* L, size_t n, const double* stamps, const float* w3, const float* a3) {
  if (!L || (n && (!stamps || !w3 || !a3))) return FLIMO_ERR_INVALID;
  for (size_t i = 0; i < n; i++) {
    const int rc = flimo_loc_update_imu(L, stamps[i], w3 + 3 * i, a3 + 3 * i);
    if (rc != FLIMO_OK) return rc;
  }
  return FLIMO_OK;
}
int flimo_loc_update_pointcloud(flimo_loc* L, const float* pts5, size_t n, double stamp) {
  if (!L) return FLIMO_ERR_INVALID;
  auto pc = std::make_shared<pcl::PointCloud<PointType>>();
  pc->points.resize(n);
  for (size_t i = 0; i < n; i++) {
    PointType& p = pc->points[i];
    p.x = pts5[5 * i]; p.y = pts5[5 * i + 1]; p.z = pts5[5 * i + 2]; p.intensity = pts5[5 * i + 3];
    p.timestamp = 0.0;
    p.time = pts5[5 * i + 4];
  }
  L->loc->updatePointCloud(pc, stamp);
  return L->loc->last_status();
}
int flimo_loc_update_pointcloud_points(flimo_loc* L, const void* pts32, size_t n, double stamp) {
  if (!L || (!pts32 && n)) return FLIMO_ERR_INVALID;
  static_assert(sizeof(PointType) == 32, "PointType must keep the reference's 32-byte layout");
  if (L->loc->updatePointCloudView(static_cast<const PointType*>(pts32), n, stamp)) return L->loc->last_status();
  // (a fresh cloud per sweep costs its pages' first touch -- about as much as the copy; the library filters the cloud in place
  //  and keeps no pointer to it, so the last sweep's storage is free again unless the wrapper's user took it)
  pcl::PointCloud<PointType>::Ptr pc = (L->in_pc && L->in_pc.use_count() == 1) ? L->in_pc : std::make_shared<pcl::PointCloud<PointType>>();
  pc->points.resize(n);
  if (n) std::memcpy(static_cast<void*>(&pc->points[0]), pts32, n * sizeof(PointType));
  L->in_pc = pc;
  L->loc->updatePointCloud(pc, stamp);
  return L->loc->last_status();
}
// A recorded drive replayed at full speed from native code: before sweep k every IMU sample with stamp <= imu_until[k] is handed to
// updateIMU, then the sweep to updatePointCloud -- what a C++ driver replaying a bag does, without a binding's per-call cost
// between the sweeps.  status_out[k]: the sweep's status; seconds_out[k] (optional): when its call returned, since the start.
int flimo_loc_replay(flimo_loc* L, size_t n_sweeps, const void* const* sweeps32, const size_t* n_points, const double* sweep_stamps,
                     const double* imu_until, size_t n_imu, const double* imu_stamps, const float* w3, const float* a3,
                     int* status_out, double* seconds_out) {
  if (!L || (n_sweeps && (!sweeps32 || !n_points || !sweep_stamps || !imu_until || !status_out))) return FLIMO_ERR_INVALID;
  if (n_imu && (!imu_stamps || !w3 || !a3)) return FLIMO_ERR_INVALID;
  const auto t0 = std::chrono::steady_clock::now();
  size_t i = 0;
  for (size_t k = 0; k < n_sweeps; k++) {
    for (; i < n_imu && imu_stamps[i] <= imu_until[k]; i++) flimo_loc_update_imu(L, imu_stamps[i], w3 + 3 * i, a3 + 3 * i);
    status_out[k] = flimo_loc_update_pointcloud_points(L, sweeps32[k], n_points[k], sweep_stamps[k]);
    if (seconds_out) seconds_out[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return FLIMO_OK;
}
// fast_limo::State::update (State.cpp:76-119) on a flat state, for unit tests: p3 q4(xyzw) v3 g3 w3 a3 bg3 ba3
void flimo_host_state_update(float s[25], double time, double t) {
  State X;
  X.p = Eigen::Vector3f(s[0], s[1], s[2]);
  X.q = Eigen::Quaternionf(s[6], s[3], s[4], s[5]);
  X.v = Eigen::Vector3f(s[7], s[8], s[9]);
  X.g = Eigen::Vector3f(s[10], s[11], s[12]);
  X.w = Eigen::Vector3f(s[13], s[14], s[15]);
  X.a = Eigen::Vector3f(s[16], s[17], s[18]);
  X.b.gyro = Eigen::Vector3f(s[19], s[20], s[21]);
  X.b.accel = Eigen::Vector3f(s[22], s[23], s[24]);
  X.time = time;
  X.update(t);
  for (int i = 0; i < 3; i++) { s[i] = X.p(i); s[7 + i] = X.v(i); }
  s[3] = X.q.x(); s[4] = X.q.y(); s[5] = X.q.z(); s[6] = X.q.w();
}
// time order of a sweep (Localizer.cpp:789-790) for tests: use_library = 1 runs std::partial_sort_copy itself
int flimo_host_time_order(const void* keys, int kind, size_t n, int descending, int use_library, uint32_t* order_out) {
  if (!keys || !order_out || kind < 0 || kind > 2) return -1;
  std::vector<uint32_t> order;
  fast_limo::time_order(keys, kind, n, descending != 0, use_library != 0, order);
  for (size_t i = 0; i < n; i++) order_out[i] = order[i];
  return 0;
}
// fast_limo::Plane / Match object API (reference Objects/Plane.cpp:23-31, Match.cpp:23-28) for tests
int flimo_host_plane(const float* xyz, const float* sqd, int n, int num_match_points, double max_dist_plane,
                     double plane_threshold, const float p_global[3], float n_out[4], float* dist_out) {
  Config::iKFoM::Mapping cfg;
  cfg.NUM_MATCH_POINTS = num_match_points;
  cfg.MAX_DIST_PLANE = max_dist_plane;
  cfg.PLANE_THRESHOLD = plane_threshold;
  MapPoints pts;
  std::vector<float> d;
  for (int i = 0; i < n; i++) { pts.push_back(pcl::PointXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])); d.push_back(sqd[i]); }
  Plane pl(pts, d, &cfg);
  const Eigen::Vector4f nv = pl.get_normal();
  for (int i = 0; i < 4; i++) n_out[i] = pl.good_fit() ? nv(i) : 0.f;
  if (pl.good_fit() && p_global && dist_out) {
    Match m(Eigen::Vector3f(p_global[0], p_global[1], p_global[2]), Eigen::Vector3f(0.f, 0.f, 0.f), pl);
    *dist_out = m.dist;
  }
  return pl.good_fit() ? 1 : 0;
}
int flimo_loc_map_add(flimo_loc* L, const float* xyz, size_t n, double stamp) {
  if (!L) return FLIMO_ERR_INVALID;
  if (!L->map->ctx()) return FLIMO_ERR_NO_DEVICE;
  return flimo_map_add(L->map->ctx(), xyz, n, 12, stamp);
}
size_t flimo_loc_map_size(flimo_loc* L) { return L ? (size_t)L->map->size() : 0; }
void flimo_loc_get_x(flimo_loc* L, double x26[26]) { L->loc->filter().get_x().to_flat(x26); }
void flimo_loc_set_x(flimo_loc* L, const double x26[26]) { flimo_host::StateIkfom s; s.from_flat(x26); L->loc->filter().change_x(s); }
void flimo_loc_get_P(flimo_loc* L, double P[529]) { std::memcpy(P, &L->loc->filter().get_P().a[0][0], sizeof(double) * 529); }
void flimo_loc_set_P(flimo_loc* L, const double P[529]) {
  flimo_host::Esekf::Cov C;
  std::memcpy(&C.a[0][0], P, sizeof(double) * 529);
  L->loc->filter().change_P(C);
}
void flimo_loc_set_flags(flimo_loc* L, int add_to_map, int download_clouds, int keep_log) {
  L->loc->add_to_map = add_to_map != 0;
  L->loc->download_clouds = download_clouds != 0;
  L->loc->filter().keep_log = keep_log != 0;
}
void flimo_loc_set_lazy_time_order(flimo_loc* L, int on) { if (L) L->loc->lazy_time_order = on != 0; }
void flimo_loc_set_gpu_filters(flimo_loc* L, int on) { if (L) L->loc->gpu_filters = on != 0; }
void flimo_loc_set_exact_tied_order(flimo_loc* L, int on) { if (L) L->loc->exact_tied_order = on != 0; }
int flimo_loc_last_sweep_tied(const flimo_loc* L) { return (L && L->loc->last_sweep_tied()) ? 1 : 0; }
void flimo_loc_set_propagation_wait(flimo_loc* L, double seconds) { if (L) L->loc->propagation_wait_s = seconds; }
// the map insert that ends a scan runs on the Mapper's worker thread (Mapper::add_scan): wait for it / switch it off
void flimo_loc_sync(flimo_loc* L) { if (L) L->map->sync(); }
void flimo_loc_set_async_insert(flimo_loc* L, int on) { if (L) L->map->set_async(on != 0); }
double flimo_loc_last_insert_seconds(flimo_loc* L) { return L ? L->map->last_insert_seconds() : 0.0; }
int flimo_loc_num_passes(flimo_loc* L) { return (int)L->loc->filter().log.size(); }
void flimo_loc_get_pass(flimo_loc* L, int i, int* M, double* HTH, double* HTh, double* dx, double* x_after) {
  const flimo_host::PassLog& g = L->loc->filter().log[i];
  *M = g.M;
  std::memcpy(HTH, g.HTH, sizeof(g.HTH));
  std::memcpy(HTh, g.HTh, sizeof(g.HTh));
  std::memcpy(dx, g.dx, sizeof(g.dx));
  std::memcpy(x_after, g.x_after, sizeof(g.x_after));
}
static size_t copy_cloud(const pcl::PointCloud<PointType>& pc, float* out, size_t cap) {
  const size_t n = pc.points.size() < cap ? pc.points.size() : cap;
  for (size_t i = 0; i < n; i++) { out[3 * i] = pc.points[i].x; out[3 * i + 1] = pc.points[i].y; out[3 * i + 2] = pc.points[i].z; }
  return pc.points.size();
}
size_t flimo_loc_get_pc2match(flimo_loc* L, float* out, size_t cap) { return copy_cloud(*L->loc->get_pc2match_pointcloud(), out, cap); }
size_t flimo_loc_get_final_scan(flimo_loc* L, float* out, size_t cap) { return copy_cloud(*L->loc->get_pointcloud(), out, cap); }
void flimo_loc_get_stage_times(flimo_loc* L, double t[4]) { L->loc->get_stage_times(t); }
void flimo_loc_get_pose_cov(flimo_loc* L, double cov36[36]) {
  std::vector<double> c = L->loc->getPoseCovariance();
  std::memcpy(cov36, c.data(), sizeof(double) * 36);
}
void flimo_loc_host_profile(flimo_loc* L, double out[4], int reset) {
  for (int i = 0; i < 4; i++) { out[i] = L->loc->prof_[i]; if (reset) L->loc->prof_[i] = 0.0; }
}
int flimo_loc_register_resident(flimo_loc* L, const double x26_prior[26], const double P_prior[529]) {
  if (!L) return FLIMO_ERR_INVALID;
  return L->loc->registerResident(x26_prior, P_prior);
}

void flimo_host_eigen_solver6(const double A[36], double wr[6], double wi[6], double V[36]) {
  flimo_host::Mat<6, 6> a, v;
  std::memcpy(&a.a[0][0], A, sizeof(double) * 36);
  flimo_host::eigen_solver6(a, wr, wi, v);
  std::memcpy(V, &v.a[0][0], sizeof(double) * 36);
}

int flimo_eskf_update_fixed(double x26[26], double P[529], const double* H, const double* h, int M, int max_iters,
                            const double limits[23], double R, double D, int* n_passes) {
  flimo_host::Esekf f;
  flimo_host::StateIkfom s;
  s.from_flat(x26);
  f.change_x(s);
  flimo_host::Esekf::Cov C;
  std::memcpy(&C.a[0][0], P, sizeof(double) * 529);
  f.change_P(C);
  f.init(max_iters, limits);
  f.keep_log = true;
  f.reference_solve = getenv("FLIMO_REFERENCE_SOLVE") != nullptr;
  f.h_reduced = [&](const flimo_host::StateIkfom&, flimo_host::ReducedMeas& out) {
    out.M = M;
    for (int i = 0; i < 144; i++) out.HTH[i] = 0.0;
    for (int i = 0; i < 12; i++) out.HTh[i] = 0.0;
    for (int m = 0; m < M; m++)
      for (int i = 0; i < 12; i++) {
        for (int j = 0; j < 12; j++) out.HTH[i * 12 + j] += H[(size_t)m * 12 + i] * H[(size_t)m * 12 + j];
        out.HTh[i] += H[(size_t)m * 12 + i] * h[m];
      }
  };
  f.h_dense = [&](flimo_host::DenseMeas& dm) { dm.H.assign(H, H + (size_t)M * 12); dm.h.assign(h, h + M); };
  f.update_iterated_dyn_share_modified(R, D);
  f.get_x().to_flat(x26);
  std::memcpy(P, &f.get_P().a[0][0], sizeof(double) * 529);
  if (n_passes) *n_passes = (int)f.log.size();
  return FLIMO_OK;
}
int flimo_eskf_predict(double x26[26], double P[529], double dt, const double Qd[12], const double acc[3], const double gyro[3]) {
  flimo_host::Esekf f;
  flimo_host::StateIkfom s;
  s.from_flat(x26);
  f.change_x(s);
  flimo_host::Esekf::Cov C;
  std::memcpy(&C.a[0][0], P, sizeof(double) * 529);
  f.change_P(C);
  flimo_host::Mat<12, 12> Q = flimo_host::Mat<12, 12>::zero();
  for (int i = 0; i < 12; i++) Q(i, i) = Qd[i];
  flimo_host::InputIkfom in;
  for (int i = 0; i < 3; i++) { in.acc(i, 0) = acc[i]; in.gyro(i, 0) = gyro[i]; }
  f.predict(dt, Q, in);
  f.get_x().to_flat(x26);
  std::memcpy(P, &f.get_P().a[0][0], sizeof(double) * 529);
  return FLIMO_OK;
}

}  // extern "C"
