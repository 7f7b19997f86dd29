// oracle/oracle_c.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// C entry points over the restatement headers (rl_*.h).  See oracle_c.h.
#include <cstring>
#include "oracle_c.h"
#include "rl_localizer.h"

using namespace oracle;

static LocCfg to_loc_cfg(const oracle_cfg* c) {
  LocCfg L;
  L.mapping.NUM_MATCH_POINTS = c->NUM_MATCH_POINTS;
  L.mapping.MAX_NUM_MATCHES = c->MAX_NUM_MATCHES;
  L.mapping.MAX_NUM_PC2MATCH = c->MAX_NUM_PC2MATCH;
  L.mapping.MAX_DIST_PLANE = c->MAX_DIST_PLANE;
  L.mapping.PLANE_THRESHOLD = c->PLANE_THRESHOLD;
  L.mapping.bucket_size = c->bucket_size;
  L.mapping.min_extent = c->min_extent;
  L.mapping.downsampling = c->downsampling != 0;
  L.MAX_NUM_ITERS = c->MAX_NUM_ITERS;
  L.estimate_extrinsics = c->estimate_extrinsics != 0;
  for (int i = 0; i < NDOF; i++) L.LIMITS[i] = c->LIMITS[i];
  L.cov_gyro = c->cov_gyro; L.cov_acc = c->cov_acc; L.cov_bias_gyro = c->cov_bias_gyro; L.cov_bias_acc = c->cov_bias_acc;
  L.time_offset = c->time_offset != 0;
  L.end_of_sweep = c->end_of_sweep != 0;
  L.num_threads = c->num_threads;
  for (int i = 0; i < 3; i++) {
    L.imu2baselink_t[i] = c->imu2baselink_t[i];
    L.lidar2baselink_t[i] = c->lidar2baselink_t[i];
    L.accel_bias[i] = c->accel_bias[i];
    L.gyro_bias[i] = c->gyro_bias[i];
  }
  for (int i = 0; i < 9; i++) {
    L.imu2baselink_R[i] = c->imu2baselink_R[i];
    L.lidar2baselink_R[i] = c->lidar2baselink_R[i];
    L.imu_sm[i] = c->imu_sm[i];
  }
  L.gravity_align = c->gravity_align != 0;
  L.calibrate_accel = c->calibrate_accel != 0;
  L.calibrate_gyro = c->calibrate_gyro != 0;
  L.imu_calib_time = c->imu_calib_time;
  L.voxel_active = c->voxel_active != 0;
  L.leaf_size = c->leaf_size;
  L.sensor_type = c->sensor_type;
  L.crop_active = c->crop_active != 0;
  for (int i = 0; i < 3; i++) { L.crop_min[i] = c->crop_min[i]; L.crop_max[i] = c->crop_max[i]; }
  L.dist_active = c->dist_active != 0;
  L.min_dist = c->min_dist;
  L.rate_active = c->rate_active != 0;
  L.rate_value = c->rate_value;
  L.fov_active = c->fov_active != 0;
  L.fov_angle = c->fov_angle;
  return L;
}

static void fill_rec(const MatchRec& m, oracle_match_rec* r) {
  r->p_global[0] = m.p_global.x; r->p_global[1] = m.p_global.y; r->p_global[2] = m.p_global.z;
  for (int i = 0; i < 4; i++) r->n[i] = m.n[i];
  r->dist = m.dist;
  r->is_plane = m.is_plane ? 1 : 0;
  r->n_nbr = m.n_nbr;
  for (int j = 0; j < 5; j++) {
    if (j < m.n_nbr) { r->nbr[j][0] = m.nbr[j].x; r->nbr[j][1] = m.nbr[j].y; r->nbr[j][2] = m.nbr[j].z; r->sqd[j] = m.sqd[j]; }
    else { r->nbr[j][0] = r->nbr[j][1] = r->nbr[j][2] = 0.f; r->sqd[j] = 0.f; }
  }
}

extern "C" {

void oracle_state_update(float s[25], double time, double t) {
  State X;
  X.p = V3f(s[0], s[1], s[2]);
  X.q = Quatf(s[6], s[3], s[4], s[5]);
  X.v = V3f(s[7], s[8], s[9]);
  X.g = V3f(s[10], s[11], s[12]);
  X.w = V3f(s[13], s[14], s[15]);
  X.a = V3f(s[16], s[17], s[18]);
  X.bgyro = V3f(s[19], s[20], s[21]);
  X.baccel = V3f(s[22], s[23], s[24]);
  X.time = time;
  X.update(t);
  s[0] = X.p.x; s[1] = X.p.y; s[2] = X.p.z;
  s[3] = X.q.x; s[4] = X.q.y; s[5] = X.q.z; s[6] = X.q.w;
  s[7] = X.v.x; s[8] = X.v.y; s[9] = X.v.z;
}

void* oracle_octree_create(float min_extent, int downsample) {
  Octree* t = new Octree();
  t->setMinExtent(min_extent);
  t->setDownsample(downsample != 0);
  return t;
}
void oracle_octree_destroy(void* t) { delete (Octree*)t; }
void oracle_octree_update(void* t, const float* xyz, size_t n) { ((Octree*)t)->update(xyz, n, 3); }
size_t oracle_octree_size(void* t) { return ((Octree*)t)->size(); }
size_t oracle_octree_points(void* t, float* out, size_t cap) {
  Octree* o = (Octree*)t;
  std::vector<V3f> pts;
  o->get_points(o->root_, pts);
  size_t n = pts.size() < cap ? pts.size() : cap;
  for (size_t i = 0; i < n; i++) { out[3 * i] = pts[i].x; out[3 * i + 1] = pts[i].y; out[3 * i + 2] = pts[i].z; }
  return pts.size();
}
long long oracle_octree_knn(void* t, const float* q, size_t nq, int k, float* nbr, float* sqd, int* cnt, int nthreads) {
  Octree* o = (Octree*)t;
  long long total = 0;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) reduction(+ : total) schedule(static)
  for (long long i = 0; i < (long long)nq; i++) {
    V3f nb[64];
    float sd[64];
    long long ev = 0;
    int c = o->knn(V3f(q[3 * i], q[3 * i + 1], q[3 * i + 2]), k, nb, sd, &ev);
    total += ev;
    cnt[i] = c;
    for (int j = 0; j < k; j++) {
      size_t b = ((size_t)i * k + j);
      if (j < c) { nbr[3 * b] = nb[j].x; nbr[3 * b + 1] = nb[j].y; nbr[3 * b + 2] = nb[j].z; sqd[b] = sd[j]; }
      else { nbr[3 * b] = nbr[3 * b + 1] = nbr[3 * b + 2] = 0.f; sqd[b] = 0.f; }
    }
  }
  return total;
}

void oracle_plane_fit(const float* nbr_xyz, const float* sqd, int n_nbr, int k, double max_dist_plane,
                      double plane_threshold, float n_out[4], int* is_plane) {
  MappingCfg cfg;
  cfg.NUM_MATCH_POINTS = k;
  cfg.MAX_DIST_PLANE = max_dist_plane;
  cfg.PLANE_THRESHOLD = plane_threshold;
  V3f nb[16];
  for (int j = 0; j < n_nbr && j < 16; j++) nb[j] = V3f(nbr_xyz[3 * j], nbr_xyz[3 * j + 1], nbr_xyz[3 * j + 2]);
  bool ok = false;
  plane_from_neighbors(n_nbr, nb, sqd, cfg, n_out, ok);
  *is_plane = ok ? 1 : 0;
}

size_t oracle_voxel_grid(const float* xyz, size_t n, float leaf, float* out, size_t cap) {
  std::vector<Pt> in(n);
  for (size_t i = 0; i < n; i++) { in[i].x = xyz[3 * i]; in[i].y = xyz[3 * i + 1]; in[i].z = xyz[3 * i + 2]; in[i].intensity = 0; in[i].time = 0; }
  std::vector<Pt> o = voxel_grid(in, leaf);
  for (size_t i = 0; i < o.size() && i < cap; i++) { out[3 * i] = o[i].x; out[3 * i + 1] = o[i].y; out[3 * i + 2] = o[i].z; }
  return o.size();
}

void oracle_pose_mats(const double x26[26], float RT[16], float RT_inv[16], float TLI_inv[16], float R_inv[9],
                      float RLI_inv[9]) {
  StateIkfom s;
  s.from_flat(x26);
  State S(s);
  M4f a = S.get_RT(), b = S.get_RT_inv(), c = S.get_extr_RT_inv();
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { RT[i * 4 + j] = a.m[i][j]; RT_inv[i * 4 + j] = b.m[i][j]; TLI_inv[i * 4 + j] = c.m[i][j]; }
  double Rd[3][3], Ld[3][3];
  quat_to_rot<double>(s.rot.conjugate(), Rd);
  quat_to_rot<double>(s.offset_R_L_I.conjugate(), Ld);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R_inv[i * 3 + j] = (float)Rd[i][j]; RLI_inv[i * 3 + j] = (float)Ld[i][j]; }
}
void oracle_state_boxplus(double x26[26], const double dx[23]) {
  StateIkfom s;
  s.from_flat(x26);
  s.boxplus(dx);
  s.to_flat(x26);
}
void oracle_state_boxminus(const double a26[26], const double b26[26], double out[23]) {
  StateIkfom a, b;
  a.from_flat(a26);
  b.from_flat(b26);
  a.boxminus(out, b);
}

long long oracle_match_H(void* octree, const oracle_cfg* cfg, const double x26[26], const float* scan_xyz, size_t n,
                         oracle_match_rec* recs, double* H, double* h, int* M) {
  LocCfg L = to_loc_cfg(cfg);
  // borrow the caller's octree without copying: build a Mapper view
  Mapper map;
  map.config = L.mapping;
  int nt = omp_get_max_threads();
  if (nt > L.num_threads) nt = L.num_threads;
  map.num_threads_ = nt < 1 ? 1 : nt;
  Octree* o = (Octree*)octree;
  std::swap(map.octree_.root_, o->root_);
  std::swap(map.octree_.num_points_, o->num_points_);
  map.octree_.min_extent_ = o->min_extent_;
  map.octree_.downsample_ = o->downsample_;
  StateIkfom s;
  s.from_flat(x26);
  std::vector<Pt> pc(n);
  for (size_t i = 0; i < n; i++) { pc[i].x = scan_xyz[3 * i]; pc[i].y = scan_xyz[3 * i + 1]; pc[i].z = scan_xyz[3 * i + 2]; pc[i].intensity = 0; pc[i].time = 0; }
  std::vector<MatchRec> all;
  std::vector<MatchRec> chosen = map.match(State(s), pc, &all);
  MeasOut out;
  calculate_H(s, chosen, L, map.num_threads_, out);
  if (recs) for (size_t i = 0; i < all.size(); i++) fill_rec(all[i], &recs[i]);
  if (recs) for (size_t i = all.size(); i < n; i++) { MatchRec e; e.n_nbr = 0; e.is_plane = false; e.dist = 0; e.n[0] = e.n[1] = e.n[2] = e.n[3] = 0; fill_rec(e, &recs[i]); }
  *M = out.M;
  for (int i = 0; i < out.M; i++) { for (int j = 0; j < 12; j++) H[(size_t)i * 12 + j] = out.h_x[(size_t)i * 12 + j]; h[i] = out.h[i]; }
  long long ev = map.last_evals;
  std::swap(map.octree_.root_, o->root_);
  std::swap(map.octree_.num_points_, o->num_points_);
  return ev;
}

void* oracle_loc_create(const oracle_cfg* cfg) {
  Localizer* L = new Localizer();
  L->init(to_loc_cfg(cfg));
  return L;
}
void oracle_loc_destroy(void* L) { delete (Localizer*)L; }
void oracle_loc_update_imu(void* Lp, double stamp, const float w[3], const float a[3]) {
  Localizer* L = (Localizer*)Lp;
  IMUmeas m;
  m.stamp = stamp;
  m.ang_vel = V3f(w[0], w[1], w[2]);
  m.lin_accel = V3f(a[0], a[1], a[2]);
  L->updateIMU(m);
}
static std::vector<Pt> to_pts5(const float* p, size_t n) {
  std::vector<Pt> v(n);
  for (size_t i = 0; i < n; i++) { v[i].x = p[5 * i]; v[i].y = p[5 * i + 1]; v[i].z = p[5 * i + 2]; v[i].intensity = p[5 * i + 3]; v[i].time = p[5 * i + 4]; }
  return v;
}
int oracle_loc_update_pointcloud_points(void* Lp, const void* pts32, size_t n, double stamp, int add_to_map) {
  Localizer* L = (Localizer*)Lp;
  std::vector<Pt> v(n);
  const unsigned char* b = (const unsigned char*)pts32;
  for (size_t i = 0; i < n; i++) {
    const unsigned char* q = b + 32 * i;
    std::memcpy(&v[i].x, q, 4); std::memcpy(&v[i].y, q + 4, 4); std::memcpy(&v[i].z, q + 8, 4);
    std::memcpy(&v[i].intensity, q + 16, 4);
    std::memcpy(&v[i].t, q + 24, 4);            // the three views of the union
    std::memcpy(&v[i].time, q + 24, 4);
    std::memcpy(&v[i].timestamp, q + 24, 8);
  }
  return L->updatePointCloud(v, stamp, add_to_map != 0);
}
int oracle_loc_update_pointcloud(void* Lp, const float* pts5, size_t n, double stamp, int add_to_map) {
  Localizer* L = (Localizer*)Lp;
  return L->updatePointCloud(to_pts5(pts5, n), stamp, add_to_map != 0);
}
void oracle_loc_map_add(void* Lp, const float* xyz, size_t n, double stamp) { ((Localizer*)Lp)->map.add(xyz, n, 3, stamp); }
size_t oracle_loc_map_size(void* Lp) { return (size_t)((Localizer*)Lp)->map.size(); }
void oracle_loc_get_x(void* Lp, double x[26]) { ((Localizer*)Lp)->ikfom.x_.to_flat(x); }
void oracle_loc_set_x(void* Lp, const double x[26]) { ((Localizer*)Lp)->ikfom.x_.from_flat(x); }
void oracle_loc_get_P(void* Lp, double P[529]) { std::memcpy(P, ((Localizer*)Lp)->ikfom.P_, sizeof(double) * 529); }
void oracle_loc_set_P(void* Lp, const double P[529]) { std::memcpy(((Localizer*)Lp)->ikfom.P_, P, sizeof(double) * 529); }
int oracle_loc_num_iters(void* Lp) { return (int)((Localizer*)Lp)->ikfom.log.size(); }
void oracle_loc_get_iter(void* Lp, int i, int* M, double* HTH, double* HTh, double* dx, double* x_after) {
  const IterLog& g = ((Localizer*)Lp)->ikfom.log[i];
  *M = g.M;
  std::memcpy(HTH, g.HTH, sizeof(g.HTH));
  std::memcpy(HTh, g.HTh, sizeof(g.HTh));
  std::memcpy(dx, g.dx, sizeof(g.dx));
  std::memcpy(x_after, g.x_after, sizeof(g.x_after));
}
static size_t copy_xyz(const std::vector<Pt>& v, float* out, size_t cap) {
  size_t n = v.size() < cap ? v.size() : cap;
  for (size_t i = 0; i < n; i++) { out[3 * i] = v[i].x; out[3 * i + 1] = v[i].y; out[3 * i + 2] = v[i].z; }
  return v.size();
}
size_t oracle_loc_get_pc2match(void* Lp, float* out, size_t cap) { return copy_xyz(((Localizer*)Lp)->pc2match, out, cap); }
size_t oracle_loc_get_final_scan(void* Lp, float* out, size_t cap) { return copy_xyz(((Localizer*)Lp)->final_scan, out, cap); }
void oracle_loc_get_stats(void* Lp, double t[6], long long* evals, long long* queries) {
  Localizer* L = (Localizer*)Lp;
  t[0] = L->t_deskew; t[1] = L->t_update; t[2] = L->t_mapadd; t[3] = L->t_sort; t[4] = L->t_match; t[5] = L->t_hrows;
  *evals = L->map.sum_evals;       // over every pass of the last updatePointCloud
  *queries = L->map.sum_queries;
}
long long oracle_loc_deskew(void* Lp, const float* pts5, size_t n, double stamp, float* out_xyz) {
  Localizer* L = (Localizer*)Lp;
  std::vector<Pt> out;
  if (!L->deskew(to_pts5(pts5, n), stamp, out)) return -1;
  copy_xyz(out, out_xyz, out.size());
  return (long long)out.size();
}
int oracle_loc_update_only(void* Lp, const float* xyz, size_t n) {
  Localizer* L = (Localizer*)Lp;
  L->pc2match.resize(n);
  for (size_t i = 0; i < n; i++) { L->pc2match[i].x = xyz[3 * i]; L->pc2match[i].y = xyz[3 * i + 1]; L->pc2match[i].z = xyz[3 * i + 2]; L->pc2match[i].intensity = 0; L->pc2match[i].time = 0; }
  if (n <= 1) return 1;
  L->ikfom.update_iterated_dyn_share_modified(0.001, 5.0);
  return 0;
}

void oracle_eskf_update_fixed(double x26[26], double P[529], const double* H, const double* h, int M, int max_iters,
                              const double limits[23], double R, double D, int* n_passes) {
  Esekf f;
  f.x_.from_flat(x26);
  std::memcpy(f.P_, P, sizeof(double) * 529);
  f.maximum_iter = max_iters;
  for (int i = 0; i < NDOF; i++) f.limit[i] = limits[i];
  f.h_dyn_share = [&](const StateIkfom&, MeasOut& out) {
    out.M = M;
    out.h_x.assign(H, H + (size_t)M * 12);
    out.h.assign(h, h + M);
  };
  f.update_iterated_dyn_share_modified(R, D);
  f.x_.to_flat(x26);
  std::memcpy(P, f.P_, sizeof(double) * 529);
  if (n_passes) *n_passes = (int)f.log.size();
}
// Eigen::EigenSolver<Matrix<double,6,6>> restated (rl_linalg.h): eigenvalues in the solver's order, real parts of the normalised
// eigenvectors as columns (row-major V)
void oracle_eigen_solver6(const double A[36], double wr[6], double wi[6], double V[36]) {
  double Vm[6][6];
  eigen_solver6(A, wr, wi, Vm);
  for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) V[i * 6 + j] = Vm[i][j];
}
void oracle_eskf_predict(double x26[26], double P[529], double dt, const double Qd[12], const double acc[3], const double gyro[3]) {
  Esekf f;
  f.x_.from_flat(x26);
  std::memcpy(f.P_, P, sizeof(double) * 529);
  double Q[12][12];
  for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) Q[i][j] = (i == j) ? Qd[i] : 0.0;
  InputIkfom in;
  for (int i = 0; i < 3; i++) { in.acc[i] = acc[i]; in.gyro[i] = gyro[i]; }
  f.predict(dt, Q, in);
  f.x_.to_flat(x26);
  std::memcpy(P, f.P_, sizeof(double) * 529);
}

}  // extern "C"
