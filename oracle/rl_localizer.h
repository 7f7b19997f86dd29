// oracle/rl_localizer.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the fast_limo::Localizer / Mapper / State / Plane / Match pieces on the
// per-scan registration hot path.  Reference files (all under /root/reference/include):
//   fast_limo/Objects/State.cpp:38-55 (ctor from state_ikfom), :76-119 (update), :136-172 (get_RT*)
//   fast_limo/Objects/Plane.cpp:23-31,41-48,80-114
//   fast_limo/Objects/Match.cpp:23-32
//   fast_limo/Modules/Mapper.cpp:59-86 (match), :88-96 (add), :100-114 (match_plane)
//   fast_limo/Modules/Localizer.cpp:245-399 (updatePointCloud), :401-531 (updateIMU, calibrated
//       branch only), :537-577 (calculate_H), :583-608 (propagateImu), :660-694 (init_iKFoM*),
//       :696-731 (imu2baselink), :733-853 (deskewPointCloud), :855-915 (integrateImu,
//       propagatedFromTimeRange)
//   fast_limo/Utils/Algorithms.hpp:25-38 (binary_search_tailored)
//   IKFoM/use-ikfom.cpp:10-31 (h_share_model)
//   fast_limo/Modules/Localizer.cpp:262-302 (NaN removal, negative crop box, distance / rate / FoV filters),
//       :873-876 (isInRange), :745-781 (per-sensor time decoding: OUSTER / VELODYNE / HESAI / LIVOX)
// Out of scope here: debug board, ROS I/O.
// PARITY UNPINNED: the reference has no tests and cannot be built here (Eigen/PCL/Boost absent).
#pragma once
#include <vector>
#include <deque>
#include <algorithm>
#include <functional>
#include <cstdint>
#include <cstdio>
#include <cfloat>
#include <climits>
#include <omp.h>
#include "rl_linalg.h"
#include "rl_octree.h"
#include "rl_ikfom.h"

namespace oracle {

// PointType (Common.hpp:100-113; 32-byte AoS with a time UNION in the reference).  The three views of the union
// are kept as separate members here; the C entry point fills the one the configured sensor reads.
struct Pt {
  float x, y, z, intensity;
  float time = 0.f;          // VELODYNE: s since the sweep reference
  uint32_t t = 0;            // OUSTER: ns since the sweep reference
  double timestamp = 0.0;    // HESAI: absolute s; LIVOX: absolute ns
};
enum SensorType { OUSTER = 0, VELODYNE = 1, HESAI = 2, LIVOX = 3, UNKNOWN = 4 };   // Common.hpp:82

struct MappingCfg {                 // Config::iKFoM::Mapping  (Utils/Config.hpp:58-69)
  int NUM_MATCH_POINTS = 5;
  int MAX_NUM_MATCHES = 2000;
  int MAX_NUM_PC2MATCH = 10000;
  double MAX_DIST_PLANE = 2.0;
  double PLANE_THRESHOLD = 5.e-2;
  int bucket_size = 2;
  float min_extent = 0.2f;
  bool downsampling = true;
};

struct LocCfg {                     // the subset of fast_limo::Config the hot path reads
  MappingCfg mapping;
  int MAX_NUM_ITERS = 3;
  double LIMITS[NDOF];
  bool estimate_extrinsics = true;
  double cov_gyro = 6.e-4, cov_acc = 1.e-2, cov_bias_gyro = 1.e-5, cov_bias_acc = 3.e-4;
  bool time_offset = true;
  bool end_of_sweep = false;
  int num_threads = 10;
  float imu2baselink_t[3] = {0, 0, 0};
  float imu2baselink_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};     // YAML list, row-major
  float lidar2baselink_t[3] = {0, 0, 0};
  float lidar2baselink_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  float accel_bias[3] = {0, 0, 0};
  float gyro_bias[3] = {0, 0, 0};
  float imu_sm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  bool gravity_align = false, calibrate_accel = false, calibrate_gyro = false;
  double imu_calib_time = 3.0;
  bool voxel_active = false;
  float leaf_size = 0.25f;
  int sensor_type = VELODYNE;
  bool crop_active = false;                 // Config::filters (Utils/Config.hpp:40-56)
  float crop_min[3] = {-1.f, -1.f, -1.f}, crop_max[3] = {1.f, 1.f, 1.f};
  bool dist_active = false;
  double min_dist = 4.0;
  bool rate_active = false;
  int rate_value = 4;
  bool fov_active = false;
  float fov_angle = 3.14159265f;
  LocCfg() { for (int i = 0; i < NDOF; i++) LIMITS[i] = 1e-3; }
};

// ---- fast_limo::State -------------------------------------------------------------------
struct State {
  V3f p, v, g, w, a, pLI, bgyro, baccel;
  Quatf q, qLI;
  double time = 0.0;
  State() {}
  explicit State(const StateIkfom& s) {                          // State.cpp:38-55
    q = Quatf((float)s.rot.w, (float)s.rot.x, (float)s.rot.y, (float)s.rot.z);
    p = V3f((float)s.pos[0], (float)s.pos[1], (float)s.pos[2]);
    v = V3f((float)s.vel[0], (float)s.vel[1], (float)s.vel[2]);
    g = V3f((float)s.grav.vec[0], (float)s.grav.vec[1], (float)s.grav.vec[2]);
    bgyro = V3f((float)s.bg[0], (float)s.bg[1], (float)s.bg[2]);
    baccel = V3f((float)s.ba[0], (float)s.ba[1], (float)s.ba[2]);
    qLI = Quatf((float)s.offset_R_L_I.w, (float)s.offset_R_L_I.x, (float)s.offset_R_L_I.y, (float)s.offset_R_L_I.z);
    pLI = V3f((float)s.offset_T_L_I[0], (float)s.offset_T_L_I[1], (float)s.offset_T_L_I[2]);
  }
  State(const StateIkfom& s, double t, const V3f& a_, const V3f& w_) : State(s) { time = t; a = a_; w = w_; }

  void update(double t) {                                        // State.cpp:76-119
    double dt = t - time;
    V3f wv = w - bgyro;
    float w_norm = norm3(wv);
    M3f R = M3f::identity();
    if (w_norm > 1.e-7) {
      V3f r = wv / w_norm;
      M3f K;
      K.m[0][0] = 0.f;   K.m[0][1] = -r.z; K.m[0][2] = r.y;
      K.m[1][0] = r.z;   K.m[1][1] = 0.f;  K.m[1][2] = -r.x;
      K.m[2][0] = -r.y;  K.m[2][1] = r.x;  K.m[2][2] = 0.f;
      float r_ang = (float)(w_norm * dt);
      float s = std::sin(r_ang);
      float c = (float)(1.0 - std::cos(r_ang));                  // double scalar converted to float
      M3f cK;                                                    // `(1-cos)*K*K` == ((c*K)*K)
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cK.m[i][j] = c * K.m[i][j];
      M3f cKK = mul(cK, K);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R.m[i][j] = R.m[i][j] + (s * K.m[i][j] + cKK.m[i][j]);
    }
    V3f a0 = quat_rotate(q, a - baccel);
    a0 = a0 + g;
    Quatf q_update = rot_to_quat<float>(R.m);
    q = qmul(q, q_update);
    // p += v*dt + 0.5*a0*dt*dt  : double scalars are converted to float before scaling
    float fdt = (float)dt;
    {
      // Eigen: v*dt + ((0.5*a0)*dt)*dt with every double scalar converted to float first
      V3f t1 = 0.5f * a0;
      V3f t2 = fdt * t1;
      V3f t3 = fdt * t2;
      V3f t0 = fdt * v;
      p = p + (t0 + t3);
    }
    v = v + fdt * a0;
  }
  M4f get_RT() const {                                           // State.cpp:136-143
    M4f T = M4f::identity();
    M3f R = quat_to_M3f(q);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T.m[i][j] = R.m[i][j];
    T.m[0][3] = p.x; T.m[1][3] = p.y; T.m[2][3] = p.z;
    return T;
  }
  static M4f inv_from(const Quatf& qq, const V3f& pp) {          // State.cpp:145-153 / :164-172
    M4f T = M4f::identity();
    M3f R = quat_to_M3f(qq);
    M3f Rt = R.transpose();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T.m[i][j] = Rt.m[i][j];
    M3f nRt;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) nRt.m[i][j] = -Rt.m[i][j];
    V3f t = mul(nRt, pp);
    T.m[0][3] = t.x; T.m[1][3] = t.y; T.m[2][3] = t.z;
    return T;
  }
  M4f get_RT_inv() const { return inv_from(q, p); }
  M4f get_extr_RT() const {
    M4f T = M4f::identity();
    M3f R = quat_to_M3f(qLI);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T.m[i][j] = R.m[i][j];
    T.m[0][3] = pLI.x; T.m[1][3] = pLI.y; T.m[2][3] = pLI.z;
    return T;
  }
  M4f get_extr_RT_inv() const { return inv_from(qLI, pLI); }
};

// ---- pcl::VoxelGrid (Localizer.cpp:313-321; PCL 1.10 filters/impl/voxel_grid.hpp, restated from its
// documented algorithm -- PCL is not available here): one output point per occupied voxel = centroid
// of its points (float accumulation), output ordered by ascending linear voxel index
// idx = i + j*div_x + k*div_x*div_y.  PCL sorts (idx, point) pairs with std::sort, which leaves the
// order INSIDE a voxel unspecified; the restatement fixes it to ascending point index.
inline std::vector<Pt> voxel_grid(const std::vector<Pt>& in, float leaf) {
  std::vector<Pt> out;
  if (in.empty()) return out;
  const float inv = 1.0f / leaf;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const Pt& p : in) {
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
    mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
  }
  int min_b[3], max_b[3], div_b[3];
  for (int a = 0; a < 3; a++) {
    min_b[a] = (int)std::floor(mn[a] * inv);
    max_b[a] = (int)std::floor(mx[a] * inv);
    div_b[a] = max_b[a] - min_b[a] + 1;
  }
  const long long cells = (long long)div_b[0] * div_b[1] * div_b[2];
  if (cells > (long long)INT_MAX) return in;          // PCL warns and returns the input unchanged
  const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
  std::vector<std::pair<int, int>> iv;
  iv.reserve(in.size());
  for (int i = 0; i < (int)in.size(); i++) {
    const Pt& p = in[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    const int i0 = (int)(std::floor(p.x * inv) - (float)min_b[0]);
    const int i1 = (int)(std::floor(p.y * inv) - (float)min_b[1]);
    const int i2 = (int)(std::floor(p.z * inv) - (float)min_b[2]);
    iv.push_back(std::make_pair(i0 + i1 * mul1 + i2 * mul2, i));
  }
  std::stable_sort(iv.begin(), iv.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
  size_t f = 0;
  while (f < iv.size()) {
    size_t l = f + 1;
    while (l < iv.size() && iv[l].first == iv[f].first) l++;
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    for (size_t k = f; k < l; k++) { const Pt& p = in[iv[k].second]; sx += p.x; sy += p.y; sz += p.z; si += p.intensity; }
    const float n = (float)(l - f);
    Pt o;
    o.x = sx / n; o.y = sy / n; o.z = sz / n; o.intensity = si / n; o.time = 0.f;
    out.push_back(o);
    f = l;
  }
  return out;
}

// ---- Plane + Match ----------------------------------------------------------------------
struct MatchRec {
  V3f p_global, p_local;
  float n[4];         // plane n_ABCD
  float dist;
  bool is_plane;
  int n_nbr;
  V3f nbr[8];
  float sqd[8];
};

// Plane::Plane (Plane.cpp:23-31): gates, estimate_plane (:80-105), plane_eval (:107-114)
inline void plane_from_neighbors(int n_nbr, const V3f* nbr, const float* sqd, const MappingCfg& cfg,
                                 float n_out[4], bool& is_plane) {
  is_plane = false;
  n_out[0] = n_out[1] = n_out[2] = n_out[3] = 0.f;
  if (!(n_nbr >= cfg.NUM_MATCH_POINTS)) return;                          // enough_points :41-43
  if (n_nbr < 1) return;
  if (!((double)sqd[n_nbr - 1] < cfg.MAX_DIST_PLANE)) return;            // close_enough :45-48 (sq. dist vs metres)
  float A[16 * 3], b[16];
  for (int j = 0; j < n_nbr; j++) {
    A[j * 3 + 0] = nbr[j].x; A[j * 3 + 1] = nbr[j].y; A[j * 3 + 2] = nbr[j].z;
    b[j] = -1.0f;
  }
  float nv[3];
  colpiv_qr_solve_nx3(n_nbr, A, b, nv);
  float n = std::sqrt(sum3(nv[0] * nv[0], nv[1] * nv[1], nv[2] * nv[2]));   // normvec.norm()
  n_out[0] = nv[0] / n;
  n_out[1] = nv[1] / n;
  n_out[2] = nv[2] / n;
  n_out[3] = (float)(1.0 / n);                                           // `1.0 / n` in double, stored float
  const float thres = (float)cfg.PLANE_THRESHOLD;                         // const float& binding at :69-73
  bool ok = true;
  for (int j = 0; j < n_nbr; j++) {
    float res = n_out[0] * nbr[j].x + n_out[1] * nbr[j].y + n_out[2] * nbr[j].z + n_out[3];
    if (std::fabs(res) > thres) { ok = false; break; }
  }
  is_plane = ok;
}

struct Mapper {
  Octree octree_;
  MappingCfg config;
  double last_map_time = -1.0;
  int num_threads_ = 1;
  long long last_evals = 0;        // instrumentation: distance evaluations in the last match()
  long long last_queries = 0;
  long long sum_evals = 0, sum_queries = 0;   // since the last reset (Localizer::updatePointCloud entry)

  void set_config(const MappingCfg& c) {                          // Mapper.cpp:38-45
    config = c;
    octree_.setBucketSize(c.bucket_size);
    octree_.setDownsample(c.downsampling);
    octree_.setMinExtent(c.min_extent);
  }
  bool exists() const { return octree_.num_points_ > 0; }
  int size() const { return (int)octree_.num_points_; }

  void add(const float* xyz, size_t n, size_t stride_f, double time) {   // Mapper.cpp:88-96
    if (n < 1) return;
    if (!exists()) octree_.initialize(xyz, n, stride_f);
    else octree_.update(xyz, n, stride_f);
    last_map_time = time;
  }

  MatchRec match_plane(const V4f& p, const V4f& p_local, long long& evals) const {   // Mapper.cpp:100-114
    MatchRec m;
    V3f q(p.v[0], p.v[1], p.v[2]);
    m.n_nbr = octree_.knn(q, config.NUM_MATCH_POINTS, m.nbr, m.sqd, &evals);
    plane_from_neighbors(m.n_nbr, m.nbr, m.sqd, config, m.n, m.is_plane);
    m.p_global = q;
    m.p_local = V3f(p_local.v[0], p_local.v[1], p_local.v[2]);
    // Match::Match (Match.cpp:23-28): dist2plane(p_global) (Plane.cpp:50-52)
    m.dist = m.n[0] * q.x + m.n[1] * q.y + m.n[2] * q.z + m.n[3];
    return m;
  }

  // Mapper::match (Mapper.cpp:59-86).  `all` (optional) receives every per-point record.
  std::vector<MatchRec> match(const State& s, const std::vector<Pt>& pc, std::vector<MatchRec>* all = nullptr) {
    std::vector<MatchRec> chosen;
    if (!exists()) return chosen;
    size_t N = pc.size();
    size_t N0 = (N > (size_t)config.MAX_NUM_PC2MATCH) ? N - (size_t)config.MAX_NUM_PC2MATCH : 0;
    size_t cnt = N - N0;
    std::vector<MatchRec> init(cnt);
    M4f RT = s.get_RT();
    long long evals_total = 0;
#pragma omp parallel for num_threads(num_threads_) reduction(+ : evals_total)
    for (long long i = 0; i < (long long)cnt; i++) {
      V4f bl(pc[i].x, pc[i].y, pc[i].z, 1.f);
      V4f gp = mul(RT, bl);
      long long ev = 0;
      init[i] = match_plane(gp, bl, ev);
      evals_total += ev;
    }
    last_evals = evals_total;
    last_queries = (long long)cnt;
    sum_evals += evals_total;          // accumulated over the passes of one update
    sum_queries += (long long)cnt;
    for (size_t j = 0; j < init.size(); j++)
      if (init[j].is_plane) chosen.push_back(init[j]);
    if (all) *all = init;
    return chosen;
  }
};

// Localizer::calculate_H (Localizer.cpp:537-577).  H is N x 12 row-major, h has N entries.
inline void calculate_H(const StateIkfom& s, const std::vector<MatchRec>& matches, const LocCfg& cfg,
                        int num_threads, MeasOut& out) {
  int N = ((int)matches.size() > cfg.mapping.MAX_NUM_MATCHES) ? cfg.mapping.MAX_NUM_MATCHES : (int)matches.size();
  out.M = N;
  out.h_x.assign((size_t)N * 12, 0.0);
  out.h.assign((size_t)N, 0.0);
  State S(s);
  M4f RTinv = S.get_RT_inv();
  M4f Einv = S.get_extr_RT_inv();
  double Rd[3][3], Ld[3][3];
  quat_to_rot<double>(s.rot.conjugate(), Rd);
  quat_to_rot<double>(s.offset_R_L_I.conjugate(), Ld);
  M3f R_inv, I_R_L_inv;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R_inv.m[i][j] = (float)Rd[i][j]; I_R_L_inv.m[i][j] = (float)Ld[i][j]; }
#pragma omp parallel for num_threads(num_threads)
  for (int i = 0; i < N; i++) {
    const MatchRec& m = matches[i];
    V4f g4(m.p_global.x, m.p_global.y, m.p_global.z, 1.0f);
    V4f p4_imu = mul(RTinv, g4);
    V4f p4_lidar = mul(Einv, p4_imu);
    V3f p_lidar(p4_lidar.v[0], p4_lidar.v[1], p4_lidar.v[2]);
    V3f p_imu(p4_imu.v[0], p4_imu.v[1], p4_imu.v[2]);
    V3f n(m.n[0], m.n[1], m.n[2]);
    V3f C = mul(R_inv, n);
    V3f B = cross3(p_lidar, mul(I_R_L_inv, C));
    V3f A = cross3(p_imu, C);
    double* row = &out.h_x[(size_t)i * 12];
    row[0] = n.x; row[1] = n.y; row[2] = n.z; row[3] = A.x; row[4] = A.y; row[5] = A.z;
    if (cfg.estimate_extrinsics) { row[6] = B.x; row[7] = B.y; row[8] = B.z; row[9] = C.x; row[10] = C.y; row[11] = C.z; }
    out.h[i] = -m.dist;
  }
}

struct IMUmeas {                    // Common.hpp:126-132
  double stamp = 0, dt = 0;
  V3f ang_vel, lin_accel;
  Quatf q;
};

// algorithms::binary_search_tailored (Utils/Algorithms.hpp:25-38)
inline int binary_search_tailored(const std::vector<State>& v, double t) {
  int high, mid, low;
  low = 0; high = (int)v.size() - 1;
  while (high >= low) {
    mid = (low + high) / 2;
    (v[mid].time > t) ? high = mid - 1 : low = mid + 1;
  }
  if (high < 0) return 0;
  return high;
}

struct Localizer {
  LocCfg config;
  Mapper map;
  Esekf ikfom;
  State state, last_state;
  int num_threads_ = 1;
  bool imu_calibrated_ = false;
  double gravity_ = 9.81;                                        // Localizer.cpp:25
  double scan_stamp = 0.0, prev_scan_stamp = 0.0, imu_stamp = 0.0, prev_imu_stamp = 0.0;
  double last_propagate_time_ = -1.0;
  M3f imu_accel_sm_;
  struct SE3 { V3f t; M3f R; };
  SE3 imu2baselink_, lidar2baselink_;
  M4f imu2baselink_T, lidar2baselink_T;
  IMUmeas last_imu;
  std::deque<IMUmeas> imu_buffer;          // front = newest (boost::circular_buffer push_front, cap 2000)
  std::deque<State> propagated_buffer;
  std::vector<Pt> pc2match;                // body frame @ Xt2
  std::vector<Pt> final_scan;              // world frame
  double first_imu_stamp = 0.0, imu_calib_time_ = 3.0;
  int calib_n = 0;
  V3f calib_gyro, calib_accel;
  bool have_prev_ang = false;
  V3f ang_vel_cg_prev;
  // instrumentation
  int last_null_iteration = 0;
  double t_deskew = 0, t_update = 0, t_mapadd = 0, t_sort = 0;
  double t_match = 0, t_hrows = 0;   // inside t_update: Mapper::match (k-NN + plane fit) and calculate_H, summed over the passes

  void init(const LocCfg& cfg) {                                 // Localizer.cpp:35-117
    config = cfg;
    num_threads_ = omp_get_max_threads();
    if (num_threads_ > config.num_threads) num_threads_ = config.num_threads;
    map.num_threads_ = num_threads_ < 1 ? 1 : num_threads_;
    map.set_config(config.mapping);
    // init_iKFoM :660-670
    ikfom.maximum_iter = config.MAX_NUM_ITERS;
    for (int i = 0; i < NDOF; i++) ikfom.limit[i] = config.LIMITS[i];
    ikfom.h_dyn_share = [this](const StateIkfom& x, MeasOut& out) { this->h_share_model(x, out); };
    // intrinsics :67-69 -- Eigen::Map<Matrix3f> over the flat list is COLUMN-major
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) imu_accel_sm_.m[i][j] = config.imu_sm[j * 3 + i];
    state.baccel = V3f(config.accel_bias[0], config.accel_bias[1], config.accel_bias[2]);
    state.bgyro = V3f(config.gyro_bias[0], config.gyro_bias[1], config.gyro_bias[2]);
    // extrinsics :72-86 -- column-major map, then transposed => stored R == list read row-major
    imu2baselink_.t = V3f(config.imu2baselink_t[0], config.imu2baselink_t[1], config.imu2baselink_t[2]);
    lidar2baselink_.t = V3f(config.lidar2baselink_t[0], config.lidar2baselink_t[1], config.lidar2baselink_t[2]);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      imu2baselink_.R.m[i][j] = config.imu2baselink_R[i * 3 + j];
      lidar2baselink_.R.m[i][j] = config.lidar2baselink_R[i * 3 + j];
    }
    imu2baselink_T = M4f::identity();
    lidar2baselink_T = M4f::identity();
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      imu2baselink_T.m[i][j] = imu2baselink_.R.m[i][j];
      lidar2baselink_T.m[i][j] = lidar2baselink_.R.m[i][j];
    }
    imu2baselink_T.m[0][3] = imu2baselink_.t.x; imu2baselink_T.m[1][3] = imu2baselink_.t.y; imu2baselink_T.m[2][3] = imu2baselink_.t.z;
    lidar2baselink_T.m[0][3] = lidar2baselink_.t.x; lidar2baselink_T.m[1][3] = lidar2baselink_.t.y; lidar2baselink_T.m[2][3] = lidar2baselink_.t.z;
    if (!(config.gravity_align || config.calibrate_accel || config.calibrate_gyro)) {   // :92-95
      imu_calibrated_ = true;
      init_iKFoM_state();
    }
    imu_calib_time_ = config.imu_calib_time;
  }

  void init_iKFoM_state() {                                      // Localizer.cpp:672-694
    StateIkfom s = ikfom.x_;
    s.rot = Quatd((double)state.q.w, (double)state.q.x, (double)state.q.y, (double)state.q.z);
    s.pos[0] = state.p.x; s.pos[1] = state.p.y; s.pos[2] = state.p.z;
    s.grav = S2g(0., 0., -gravity_);
    s.bg[0] = state.bgyro.x; s.bg[1] = state.bgyro.y; s.bg[2] = state.bgyro.z;
    s.ba[0] = state.baccel.x; s.ba[1] = state.baccel.y; s.ba[2] = state.baccel.z;
    double Rd[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rd[i][j] = (double)lidar2baselink_.R.m[i][j];
    s.offset_R_L_I = rot_to_quat<double>(Rd);
    s.offset_T_L_I[0] = lidar2baselink_.t.x; s.offset_T_L_I[1] = lidar2baselink_.t.y; s.offset_T_L_I[2] = lidar2baselink_.t.z;
    ikfom.x_ = s;
    for (int i = 0; i < NDOF; i++) for (int j = 0; j < NDOF; j++) ikfom.P_[i][j] = (i == j) ? 1.0 : 0.0;
    for (int i = 6; i < 12; i++) ikfom.P_[i][i] = 0.000001;
    for (int i = 15; i < 18; i++) ikfom.P_[i][i] = 0.00001;
    for (int i = 18; i < 21; i++) ikfom.P_[i][i] = 0.0001;
    ikfom.P_[21][21] = ikfom.P_[22][22] = 0.000001;
  }

  // IKFoM::h_share_model (use-ikfom.cpp:10-31)
  void h_share_model(const StateIkfom& x, MeasOut& out) {
    const double ta = omp_get_wtime();
    std::vector<MatchRec> matches = map.match(State(x), pc2match);
    const double tb = omp_get_wtime();
    calculate_H(x, matches, config, num_threads_ < 1 ? 1 : num_threads_, out);
    t_match += tb - ta;
    t_hrows += omp_get_wtime() - tb;
  }

  IMUmeas imu2baselink(const IMUmeas& imu) {                     // Localizer.cpp:696-731
    IMUmeas o;
    double dt = imu.stamp - prev_imu_stamp;
    if ((dt == 0.) || (dt > 0.1)) dt = 1.0 / 200.0;
    V3f ang_vel_cg = mul(imu2baselink_.R, imu.ang_vel);
    if (!have_prev_ang) { ang_vel_cg_prev = ang_vel_cg; have_prev_ang = true; }   // function-local static
    V3f lin_accel_cg = mul(imu2baselink_.R, imu.lin_accel);
    V3f nt(-imu2baselink_.t.x, -imu2baselink_.t.y, -imu2baselink_.t.z);
    V3f dw = (ang_vel_cg - ang_vel_cg_prev) / (float)dt;
    lin_accel_cg = (lin_accel_cg + cross3(dw, nt)) + cross3(ang_vel_cg, cross3(ang_vel_cg, nt));
    ang_vel_cg_prev = ang_vel_cg;
    o.ang_vel = ang_vel_cg;
    o.lin_accel = lin_accel_cg;
    o.dt = dt;
    o.stamp = imu.stamp;
    Quatf q = rot_to_quat<float>(imu2baselink_.R.m);
    float qn = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    q.x /= qn; q.y /= qn; q.z /= qn; q.w /= qn;
    o.q = qmul(q, imu.q);
    prev_imu_stamp = imu.stamp;
    return o;
  }

  void propagateImu(const IMUmeas& imu) {                        // Localizer.cpp:583-608
    InputIkfom in;
    in.acc[0] = imu.lin_accel.x; in.acc[1] = imu.lin_accel.y; in.acc[2] = imu.lin_accel.z;
    in.gyro[0] = imu.ang_vel.x; in.gyro[1] = imu.ang_vel.y; in.gyro[2] = imu.ang_vel.z;
    double Q[12][12];
    for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) Q[i][j] = 0.0;
    for (int i = 0; i < 3; i++) { Q[i][i] = config.cov_gyro; Q[3 + i][3 + i] = config.cov_acc; Q[6 + i][6 + i] = config.cov_bias_gyro; Q[9 + i][9 + i] = config.cov_bias_acc; }
    ikfom.predict(imu.dt, Q, in);
    propagated_buffer.push_front(State(ikfom.x_, imu.stamp, imu.lin_accel, imu.ang_vel));
    if (propagated_buffer.size() > 2000) propagated_buffer.pop_back();
    last_propagate_time_ = imu.stamp;
  }

  void updateIMU(const IMUmeas& raw) {                           // Localizer.cpp:401-531 (calibrated branch)
    imu_stamp = raw.stamp;
    IMUmeas imu = imu2baselink(raw);
    if (first_imu_stamp == 0.0) first_imu_stamp = imu.stamp;
    if (!imu_calibrated_) {                                        // Localizer.cpp:411-493
      if ((imu.stamp - first_imu_stamp) < imu_calib_time_) {
        calib_n++;
        calib_gyro = calib_gyro + imu.ang_vel;
        calib_accel = calib_accel + imu.lin_accel;
        return;
      }
      V3f gyro_avg = calib_gyro / (float)calib_n, accel_avg = calib_accel / (float)calib_n;
      V3f grav_vec(0.f, 0.f, (float)gravity_);
      state.q = imu.q;
      if (config.gravity_align) {
        V3f d = accel_avg - state.baccel;
        grav_vec = std::fabs((float)gravity_) * (d / norm3(d));          // normalized() * abs(g)
        // Quaternionf::FromTwoVectors(grav_vec, (0,0,g))
        V3f v0 = grav_vec / norm3(grav_vec);
        V3f zz(0.f, 0.f, (float)gravity_);
        V3f v1 = zz / norm3(zz);
        float c = dot3(v1, v0);
        if (!(c < -1.0f + 1e-5f)) {
          V3f axis = cross3(v0, v1);
          float sq = std::sqrt((1.0f + c) * 2.0f);
          float invs = 1.0f / sq;
          state.q = Quatf(sq * 0.5f, axis.x * invs, axis.y * invs, axis.z * invs);
        }
        state.g = grav_vec;
      }
      if (config.calibrate_accel) state.baccel = accel_avg - grav_vec;
      if (config.calibrate_gyro) state.bgyro = gyro_avg;
      float qn = std::sqrt(state.q.x * state.q.x + state.q.y * state.q.y + state.q.z * state.q.z + state.q.w * state.q.w);
      state.q.x /= qn; state.q.y /= qn; state.q.z /= qn; state.q.w /= qn;
      init_iKFoM_state();
      imu_calibrated_ = true;
      return;
    }
    V3f sm_a = mul(imu_accel_sm_, imu.lin_accel);
    imu.lin_accel = sm_a - state.baccel;
    imu.ang_vel = imu.ang_vel - state.bgyro;
    last_imu = imu;
    imu_buffer.push_front(imu);
    if (imu_buffer.size() > 2000) imu_buffer.pop_back();
    propagateImu(imu);
  }

  // Localizer::propagatedFromTimeRange + integrateImu (Localizer.cpp:855-915); no blocking wait here:
  // returns false instead of waiting on cv_prop_stamp.
  bool frames_from_range(double start_time, double end_time, std::vector<State>& frames) {
    frames.clear();
    if (propagated_buffer.empty() || propagated_buffer.front().time < end_time) return false;
    size_t it = 0, n = propagated_buffer.size();
    size_t last = it;
    it++;
    while (it != n && propagated_buffer[it].time >= end_time) { last = it; it++; }
    while (it != n && propagated_buffer[it].time >= start_time) it++;
    if (it == n) return false;
    it++;
    // reverse iteration: from element (it-1) down to element `last` (exclusive of last? see below)
    // begin = reverse_iterator(prop_it) -> points at element it-1 ; end = reverse_iterator(last_prop_it)
    // -> one before element `last` in reverse order, i.e. iteration covers it-1, it-2, ..., last.
    for (size_t k = it; k-- > last;) frames.push_back(propagated_buffer[k]);
    return true;
  }

  // Localizer::deskewPointCloud (Localizer.cpp:733-853), VELODYNE time base.
  bool deskew(const std::vector<Pt>& pc, double start_time, std::vector<Pt>& out, std::vector<Pt>* world = nullptr) {
    out.clear();
    if (pc.empty()) return false;
    double sweep_ref_time = start_time;
    const bool eos = config.end_of_sweep;
    // Localizer.cpp:745-781: comparator and time decoding per sensor.  `pt.t * 1e-9f` is a float product
    // (uint32 converted to float), `pt.timestamp * 1e-9f` a double product.
    std::function<bool(const Pt&, const Pt&)> cmp;
    std::function<double(const Pt&)> extract;
    if (config.sensor_type == OUSTER) {
      cmp = [eos](const Pt& a, const Pt& b) { return eos ? a.t > b.t : a.t < b.t; };
      extract = [sweep_ref_time, eos](const Pt& p) { return eos ? sweep_ref_time - p.t * 1e-9f : sweep_ref_time + p.t * 1e-9f; };
    } else if (config.sensor_type == VELODYNE) {
      cmp = [eos](const Pt& a, const Pt& b) { return eos ? a.time > b.time : a.time < b.time; };
      extract = [sweep_ref_time, eos](const Pt& p) { return eos ? sweep_ref_time - p.time : sweep_ref_time + p.time; };
    } else if (config.sensor_type == HESAI) {
      cmp = [](const Pt& a, const Pt& b) { return a.timestamp < b.timestamp; };
      extract = [](const Pt& p) { return p.timestamp; };
    } else if (config.sensor_type == LIVOX) {
      cmp = [](const Pt& a, const Pt& b) { return a.timestamp < b.timestamp; };
      extract = [](const Pt& p) { return p.timestamp * 1e-9f; };
    } else {
      return false;                         // "LiDAR sensor type unknown or not specified" (:776-781)
    }
    std::vector<Pt> sorted(pc.size());
    const double ts0 = omp_get_wtime();
    std::partial_sort_copy(pc.begin(), pc.end(), sorted.begin(), sorted.end(), cmp);   // :789-790
    t_sort = omp_get_wtime() - ts0;
    double offset = 0.0;
    if (config.time_offset) {
      offset = imu_stamp - extract(sorted.back()) - 1.e-4;
      if (offset > 0.0) offset = 0.0;
    }
    scan_stamp = extract(sorted.back()) + offset;
    std::vector<State> frames;
    if (!frames_from_range(prev_scan_stamp, scan_stamp, frames) || frames.empty()) return false;
    last_state = State(ikfom.x_);
    M4f last_inv = last_state.get_RT_inv();
    out.resize(sorted.size());
    if (world) world->resize(sorted.size());
    int nt = num_threads_ < 1 ? 1 : num_threads_;
#pragma omp parallel for num_threads(nt)
    for (long long k = 0; k < (long long)sorted.size(); k++) {
      double tk = extract(sorted[k]) + offset;
      int i_f = binary_search_tailored(frames, tk);
      State X0 = frames[i_f];
      X0.update(tk);
      M4f T = mul(X0.get_RT(), lidar2baselink_T);
      V4f pt(sorted[k].x, sorted[k].y, sorted[k].z, 1.f);
      V4f pw = mul(T, pt);
      V4f p2 = mul(last_inv, pw);
      Pt o = sorted[k];
      o.x = p2.v[0]; o.y = p2.v[1]; o.z = p2.v[2];
      out[k] = o;
      if (world) { Pt wq = sorted[k]; wq.x = pw.v[0]; wq.y = pw.v[1]; wq.z = pw.v[2]; (*world)[k] = wq; }
    }
    return true;
  }

  // Localizer::updatePointCloud (Localizer.cpp:245-399), filters off.  Returns 0 ok, 1 null iteration,
  // <0 early return.
  int updatePointCloud(const std::vector<Pt>& raw, double time_stamp, bool add_to_map = true) {
    last_null_iteration = 0;
    map.sum_evals = 0; map.sum_queries = 0;
    if (raw.empty()) return -1;
    if (!imu_calibrated_) return -2;
    if (imu_buffer.empty()) return -3;
    std::vector<Pt> finite;
    finite.reserve(raw.size());
    for (const auto& p : raw) {                                  // removeNaNFromPointCloud :263-265
      if (std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z)) finite.push_back(p);
    }
    if (config.crop_active) {                                    // pcl::CropBox, setNegative(true) (:57-59,268-271):
      std::vector<Pt> kept;                                      // a point is "inside" unless it is beyond a face
      kept.reserve(finite.size());
      for (const auto& p : finite) {
        const bool outside = p.x < config.crop_min[0] || p.y < config.crop_min[1] || p.z < config.crop_min[2] ||
                             p.x > config.crop_max[0] || p.y > config.crop_max[1] || p.z > config.crop_max[2];
        if (outside) kept.push_back(p);
      }
      finite.swap(kept);
    }
    // distance / rate / FoV filters over the INDEXED cloud (:274-302); the index is the position after the crop
    std::vector<Pt> input;
    input.reserve(finite.size());
    {
      const float min_dist = (float)config.min_dist;             // `static float min_dist = static_cast<float>(...)`
      for (size_t i = 0; i < finite.size(); i++) {
        const Pt& p = finite[i];
        bool keep = true;
        if (config.dist_active) keep = keep && (norm3(V3f(p.x, p.y, p.z)) > min_dist);
        if (config.rate_active) keep = keep && ((long)i % config.rate_value == 0);
        if (config.fov_active) keep = keep && (std::fabs(std::atan2(p.y, p.x)) < config.fov_angle);   // isInRange :873-876
        if (keep) input.push_back(p);
      }
    }
    double t0 = omp_get_wtime();
    std::vector<Pt> deskewed;
    deskew(input, time_stamp, deskewed);
    double t1 = omp_get_wtime();
    t_deskew = t1 - t0;
    if (config.voxel_active) pc2match = voxel_grid(deskewed, config.leaf_size);   // :313-321
    else pc2match = deskewed;
    int rc = 0;
    if (pc2match.size() > 1) {
      t_match = t_hrows = 0.0;
      ikfom.update_iterated_dyn_share_modified(0.001, 5.0);      // :333
      double t2 = omp_get_wtime();
      t_update = t2 - t1;
      State corrected(ikfom.x_);
      state = corrected;
      state.w = last_imu.ang_vel;
      state.a = last_imu.lin_accel;
      lidar2baselink_T = state.get_extr_RT();                    // :356
      M4f RT = state.get_RT();
      final_scan.resize(pc2match.size());
      for (size_t i = 0; i < pc2match.size(); i++) {             // pcl::transformPointCloud :361-371
        // PCL 1.10 detail::Transformer::se3 (SSE2 path): c0*x + (c1*y + (c2*z + c3))
        const Pt& s = pc2match[i];
        Pt o = s;
        o.x = RT.m[0][0] * s.x + (RT.m[0][1] * s.y + (RT.m[0][2] * s.z + RT.m[0][3]));
        o.y = RT.m[1][0] * s.x + (RT.m[1][1] * s.y + (RT.m[1][2] * s.z + RT.m[1][3]));
        o.z = RT.m[2][0] * s.x + (RT.m[2][1] * s.y + (RT.m[2][2] * s.z + RT.m[2][3]));
        final_scan[i] = o;
      }
      if (add_to_map) {
        map.add(&final_scan[0].x, final_scan.size(), sizeof(Pt) / sizeof(float), scan_stamp);   // :377
        t_mapadd = omp_get_wtime() - t2;
      }
    } else {
      last_null_iteration = 1;
      rc = 1;
    }
    prev_scan_stamp = scan_stamp;
    return rc;
  }
};

}  // namespace oracle
