// fast_limo_amd/csrc/host/fast_limo.cpp
// Host side of the MI355X-native fast_limo library: State, Mapper and Localizer with the
// reference's public API, driving the HIP hot path through the C ABI of include/flimo_c.h.
// Reference files mirrored: include/fast_limo/Objects/State.cpp, Modules/Mapper.cpp,
// Modules/Localizer.cpp (file:line cited per function).  Compile with -ffp-contract=off: the
// float32 pose algebra must round like the reference (no FMA).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <mutex>
#include <thread>

#include "../../../include/flimo_c.h"
#include "../../../include/flimo_dev.h"      // flimo_calculate_H_host (Localizer::calculate_H of the mirror)
#include "fast_limo/Modules/Localizer.hpp"
#include "fast_limo/Modules/Mapper.hpp"
#include "flimo_ikfom.hpp"

using namespace fast_limo;
using flimo_host::Esekf;
using flimo_host::StateIkfom;

// ---------------------------------------------------------------------------------------------
// float32 helpers (Eigen evaluation order: 3-term reductions are c0 + (c1 + c2))
// ---------------------------------------------------------------------------------------------
namespace {
inline float s3(float a, float b, float c) { return a + (b + c); }

void quat_to_R(const Eigen::Quaternionf& q, float R[9]) {   // Eigen::Quaternionf::toRotationMatrix
  const float tx = 2.f * q.x(), ty = 2.f * q.y(), tz = 2.f * q.z();
  const float twx = tx * q.w(), twy = ty * q.w(), twz = tz * q.w();
  const float txx = tx * q.x(), txy = ty * q.x(), txz = tz * q.x();
  const float tyy = ty * q.y(), tyz = tz * q.y(), tzz = tz * q.z();
  R[0] = 1.f - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.f - (txx + tyy);
}
Eigen::Matrix4f se3(const Eigen::Quaternionf& q, const Eigen::Vector3f& p) {
  float R[9];
  quat_to_R(q, R);
  Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) T(i, j) = R[i * 3 + j]; T(i, 3) = p(i); }
  return T;
}
Eigen::Matrix4f se3_inv(const Eigen::Quaternionf& q, const Eigen::Vector3f& p) {
  float R[9];
  quat_to_R(q, R);
  Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) T(i, j) = R[j * 3 + i];
    T(i, 3) = s3((-R[0 * 3 + i]) * p(0), (-R[1 * 3 + i]) * p(1), (-R[2 * 3 + i]) * p(2));
  }
  return T;
}
Eigen::Vector3f mat3_mul(const Eigen::Matrix3f& A, const Eigen::Vector3f& v) {
  return Eigen::Vector3f(s3(A(0, 0) * v(0), A(0, 1) * v(1), A(0, 2) * v(2)), s3(A(1, 0) * v(0), A(1, 1) * v(1), A(1, 2) * v(2)),
                         s3(A(2, 0) * v(0), A(2, 1) * v(1), A(2, 2) * v(2)));
}
Eigen::Vector3f cross(const Eigen::Vector3f& a, const Eigen::Vector3f& b) {
  return Eigen::Vector3f(a(1) * b(2) - a(2) * b(1), a(2) * b(0) - a(0) * b(2), a(0) * b(1) - a(1) * b(0));
}
Eigen::Quaternionf quat_from_R(const Eigen::Matrix3f& m) {   // Eigen Quaternion(Matrix3f)
  float t = m(0, 0) + (m(1, 1) + m(2, 2));
  float x, y, z, w;
  if (t > 0.f) {
    t = std::sqrt(t + 1.0f);
    w = 0.5f * t;
    t = 0.5f / t;
    x = (m(2, 1) - m(1, 2)) * t;
    y = (m(0, 2) - m(2, 0)) * t;
    z = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0f);
    float c[3];
    c[i] = 0.5f * t;
    t = 0.5f / t;
    w = (m(k, j) - m(j, k)) * t;
    c[j] = (m(j, i) + m(i, j)) * t;
    c[k] = (m(k, i) + m(i, k)) * t;
    x = c[0]; y = c[1]; z = c[2];
  }
  return Eigen::Quaternionf(w, x, y, z);
}
Eigen::Quaternionf quat_mul(const Eigen::Quaternionf& a, const Eigen::Quaternionf& b) {
  return Eigen::Quaternionf(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                            a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                            a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                            a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
}
double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// Config defaults (reference src/main.cpp:101-168)
// ---------------------------------------------------------------------------------------------
Config Config::defaults() {
  Config c;
  c.topics.lidar = "/velodyne_points";
  c.topics.imu = "/EL/Sensors/vectornav/IMU";
  c.num_threads = 10;
  c.sensor_type = 1;
  c.debug = true;
  c.verbose = true;
  c.ikfom.estimate_extrinsics = true;
  c.time_offset = true;
  c.end_of_sweep = false;
  c.gravity_align = true;
  c.calibrate_accel = true;
  c.calibrate_gyro = true;
  c.imu_calib_time = 3.0;
  c.extrinsics.imu2baselink_t = {0.f, 0.f, 0.f};
  c.extrinsics.imu2baselink_R = std::vector<float>(9, 0.f);
  c.extrinsics.lidar2baselink_t = {0.f, 0.f, 0.f};
  c.extrinsics.lidar2baselink_R = std::vector<float>(9, 0.f);
  c.intrinsics.accel_bias = {0.f, 0.f, 0.f};
  c.intrinsics.gyro_bias = {0.f, 0.f, 0.f};
  c.intrinsics.imu_sm = std::vector<float>(9, 0.f);
  c.filters.crop_active = true;
  c.filters.cropBoxMin = {-1.f, -1.f, -1.f};
  c.filters.cropBoxMax = {1.f, 1.f, 1.f};
  c.filters.voxel_active = true;
  c.filters.leafSize = {0.25f, 0.25f, 0.25f};
  c.filters.dist_active = false;
  c.filters.min_dist = 4.0;
  c.filters.rate_active = false;
  c.filters.rate_value = 4;
  c.filters.fov_active = false;
  c.filters.fov_angle = (float)(360.0f * M_PI / 360.0);
  c.ikfom.mapping.NUM_MATCH_POINTS = 5;
  c.ikfom.mapping.MAX_NUM_MATCHES = 2000;
  c.ikfom.mapping.MAX_NUM_PC2MATCH = 10000;
  c.ikfom.mapping.MAX_DIST_PLANE = 2.0;
  c.ikfom.mapping.PLANE_THRESHOLD = 5.e-2;
  c.ikfom.mapping.octree.bucket_size = 2;
  c.ikfom.mapping.octree.min_extent = 0.2f;
  c.ikfom.mapping.octree.downsampling = true;
  c.ikfom.MAX_NUM_ITERS = 3;
  c.ikfom.cov_gyro = 6.e-4;
  c.ikfom.cov_acc = 1.e-2;
  c.ikfom.cov_bias_gyro = 1.e-5;
  c.ikfom.cov_bias_acc = 3.e-4;
  c.ikfom.LIMITS = std::vector<double>(23, 1.e-3);
  return c;
}

// ---------------------------------------------------------------------------------------------
// State
// ---------------------------------------------------------------------------------------------
State::State() : time(0.0) {}
State::State(const StateIkfom& s) : time(0.0) {                    // State.cpp:38-55
  q = Eigen::Quaternionf((float)s.rot.w, (float)s.rot.x, (float)s.rot.y, (float)s.rot.z);
  p = Eigen::Vector3f((float)s.pos(0, 0), (float)s.pos(1, 0), (float)s.pos(2, 0));
  v = Eigen::Vector3f((float)s.vel(0, 0), (float)s.vel(1, 0), (float)s.vel(2, 0));
  g = Eigen::Vector3f((float)s.grav.vec(0, 0), (float)s.grav.vec(1, 0), (float)s.grav.vec(2, 0));
  b.gyro = Eigen::Vector3f((float)s.bg(0, 0), (float)s.bg(1, 0), (float)s.bg(2, 0));
  b.accel = Eigen::Vector3f((float)s.ba(0, 0), (float)s.ba(1, 0), (float)s.ba(2, 0));
  qLI = Eigen::Quaternionf((float)s.offset_R_L_I.w, (float)s.offset_R_L_I.x, (float)s.offset_R_L_I.y, (float)s.offset_R_L_I.z);
  pLI = Eigen::Vector3f((float)s.offset_T_L_I(0, 0), (float)s.offset_T_L_I(1, 0), (float)s.offset_T_L_I(2, 0));
}
State::State(const StateIkfom& s, double t) : State(s) { time = t; }
State::State(const StateIkfom& s, double t, Eigen::Vector3f a_, Eigen::Vector3f w_) : State(s, t) { a = a_; w = w_; }
State::State(Eigen::Matrix4f& T) : State() {                          // State.cpp:68-74
  Eigen::Matrix3f R;
  for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R(i, j) = T(i, j); p(i) = T(i, 3); }
  q = quat_from_R(R);
}

void State::operator+=(const State& s) {                              // State.cpp:121-134
  q = quat_mul(q, s.q);
  for (int i = 0; i < 3; i++) { p(i) += s.p(i); v(i) += s.v(i); w(i) += s.w(i); }
  b.gyro = s.b.gyro; b.accel = s.b.accel;
  qLI = s.qLI; pLI = s.pLI;
  g = s.g;
}

void State::update(double t) {                                        // State.cpp:76-119, float32 in the reference's order
  const double dt = t - time;
  const Eigen::Vector3f wv(w(0) - b.gyro(0), w(1) - b.gyro(1), w(2) - b.gyro(2));
  const float w_norm = std::sqrt(s3(wv(0) * wv(0), wv(1) * wv(1), wv(2) * wv(2)));
  Eigen::Matrix3f R = Eigen::Matrix3f::Identity();
  if (w_norm > 1.e-7) {
    const float r0 = wv(0) / w_norm, r1 = wv(1) / w_norm, r2 = wv(2) / w_norm;
    const float K[9] = {0.f, -r2, r1, r2, 0.f, -r0, -r1, r0, 0.f};
    const float r_ang = (float)(w_norm * dt);
    const float sn = std::sin(r_ang);
    const float cs = (float)(1.0 - std::cos(r_ang));                 // double scalar, converted to the float expression type
    float cK[9];
    for (int i = 0; i < 9; i++) cK[i] = cs * K[i];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        const float kk = s3(cK[i * 3 + 0] * K[0 * 3 + j], cK[i * 3 + 1] * K[1 * 3 + j], cK[i * 3 + 2] * K[2 * 3 + j]);
        R(i, j) = R(i, j) + (sn * K[i * 3 + j] + kk);
      }
  }
  // a0 = q._transformVector(a - ba) + g  (Eigen: v + w*uv + q.vec x uv with uv = 2 q.vec x v)
  const Eigen::Vector3f av(a(0) - b.accel(0), a(1) - b.accel(1), a(2) - b.accel(2));
  const Eigen::Vector3f qv(q.x(), q.y(), q.z());
  Eigen::Vector3f uv = cross(qv, av);
  for (int i = 0; i < 3; i++) uv(i) = uv(i) + uv(i);
  const Eigen::Vector3f c2 = cross(qv, uv);
  Eigen::Vector3f a0;
  for (int i = 0; i < 3; i++) a0(i) = (av(i) + q.w() * uv(i) + c2(i)) + g(i);
  q = quat_mul(q, quat_from_R(R));
  // p += v*dt + 0.5*a0*dt*dt, v += a0*dt: Eigen converts every double scalar to the float expression type first
  const float fdt = (float)dt;
  for (int i = 0; i < 3; i++) {
    const float t3 = fdt * (fdt * (0.5f * a0(i)));
    p(i) = p(i) + (fdt * v(i) + t3);
    v(i) = v(i) + fdt * a0(i);
  }
}

Eigen::Matrix4f State::get_RT() const { return se3(q, p); }
Eigen::Matrix4f State::get_RT_inv() const { return se3_inv(q, p); }
Eigen::Matrix4f State::get_extr_RT() const { return se3(qLI, pLI); }
Eigen::Matrix4f State::get_extr_RT_inv() const { return se3_inv(qLI, pLI); }

// ---------------------------------------------------------------------------------------------
// Mapper
// ---------------------------------------------------------------------------------------------
Mapper::Mapper() : num_threads_(1), ctx_(nullptr), front_(nullptr), device_(0), cell_size_(0.f), async_(std::getenv("FLIMO_SYNC_INSERT") == nullptr) {
  config.NUM_MATCH_POINTS = 5;                                    // Mapper.cpp:23-31
  config.MAX_NUM_MATCHES = 2000;
  config.MAX_NUM_PC2MATCH = 10000;
  config.MAX_DIST_PLANE = 2.0;
  config.PLANE_THRESHOLD = 5.e-2;
  config.octree.bucket_size = 2;
  config.octree.min_extent = 0.2f;
  config.octree.downsampling = true;
}
Mapper::Mapper(int device) : Mapper() { device_ = device; }
Mapper::~Mapper() {
  sync();
  if (worker_.joinable()) {
    { std::lock_guard<std::mutex> lk(wm_); quit_ = true; }
    wcv_.notify_all();
    worker_.join();
  }
  if (front_) flimo_ctx_destroy(front_);
  if (ctx_) flimo_ctx_destroy(ctx_);
}
// ---- asynchronous path exit -------------------------------------------------------------------------------------------
// An insert takes a few hundred microseconds; waking a sleeping thread costs tens of them on either side of the hand-over.
// Both sides therefore poll the flag for a bounded time before they fall back to the condition variable.
static inline void spin_pause() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}
void Mapper::sync() {
  if (!worker_.joinable()) return;
  const double t0 = now_s();
  while (busy_.load(std::memory_order_acquire)) {
    if (now_s() - t0 > 2.0e-3) {
      std::unique_lock<std::mutex> lk(wm_);
      wcv_.wait(lk, [this] { return !busy_.load(); });
      return;
    }
    spin_pause();
  }
}
void Mapper::run_insert(const double x26[26], double stamp) {
  const double t0 = now_s();
  const int rc = flimo_map_add_scan(ctx_, x26, stamp);
  if (rc != FLIMO_OK) std::cout << "FAST_LIMO::map insert failed: " << flimo_last_error(ctx_) << "\n";
  insert_seconds_ = now_s() - t0;
}
void Mapper::worker_main() {
  std::unique_lock<std::mutex> lk(wm_);
  for (;;) {
    if (!busy_.load() && !quit_.load()) {
      // the next job of a back-to-back sequence arrives within a millisecond: poll for it before going to sleep
      lk.unlock();
      const double t0 = now_s();
      while (!busy_.load(std::memory_order_acquire) && !quit_.load(std::memory_order_acquire) && now_s() - t0 < 1.5e-3) spin_pause();
      lk.lock();
    }
    wcv_.wait(lk, [this] { return busy_.load() || quit_.load(); });
    if (quit_) return;
    lk.unlock();
    run_insert(job_x_, job_stamp_);                 // the only user of ctx_ while busy_ is set
    lk.lock();
    busy_ = false;
    wcv_.notify_all();
  }
}
void Mapper::add_scan(const double x26[26], double stamp) {
  if (!ctx_) return;
  sync();
  handoff_time_ = now_s();
  if (!async_) { run_insert(x26, stamp); return; }
  if (!worker_.joinable()) worker_ = std::thread(&Mapper::worker_main, this);
  {
    std::lock_guard<std::mutex> lk(wm_);
    std::memcpy(job_x_, x26, sizeof(job_x_));
    job_stamp_ = stamp;
    busy_ = true;
  }
  wcv_.notify_all();
}

bool Mapper::attach(int device, float cell_size) {
  sync();
  if (ctx_) return true;
  device_ = device;
  cell_size_ = cell_size;
  const int rc = flimo_ctx_create(device, &ctx_);
  if (rc != FLIMO_OK) {
    ctx_ = nullptr;
    err_ = "flimo_ctx_create failed (" + std::to_string(rc) + "): no gfx950 device or HIP error";
    std::cout << "FAST_LIMO::FATAL ERROR: " << err_ << "\n";
    return false;
  }
  flimo_map_cfg mc{config.octree.min_extent, config.octree.bucket_size, config.octree.downsampling ? 1 : 0, cell_size_};
  flimo_map_config(ctx_, &mc);
  // The host loop of the iterated update queues the next pass ahead of the filter's algebra (flimo_set_pass_pipeline): this library
  // tells the context when an update is over (Esekf::h_update_end), which is what the switch asks of its caller.
  if (std::getenv("FLIMO_PIPELINE") == nullptr) (void)flimo_set_pass_pipeline(ctx_, 1);
  // the input stage's own context (see front_ctx()); a sequential insert leaves nothing to overlap with
  if (std::getenv("FLIMO_NO_FRONT_CTX") == nullptr && flimo_ctx_create(device, &front_) != FLIMO_OK) front_ = nullptr;
  return true;
}
flimo_ctx* Mapper::front_ctx() { return async_ ? front_ : nullptr; }
void Mapper::set_num_threads(int n) { if (n >= 1) num_threads_ = n; }
void Mapper::set_config(const Config::iKFoM::Mapping& cfg) {       // Mapper.cpp:38-45
  config = cfg;
  sync();
  if (ctx_) {
    flimo_map_cfg mc{config.octree.min_extent, config.octree.bucket_size, config.octree.downsampling ? 1 : 0, cell_size_};
    flimo_map_config(ctx_, &mc);
  }
}
bool Mapper::exists() { sync(); return ctx_ && flimo_map_size(ctx_) > 0; }
int Mapper::size() { sync(); return ctx_ ? (int)flimo_map_size(ctx_) : 0; }
double Mapper::last_time() { sync(); return ctx_ ? flimo_map_last_time(ctx_) : -1.0; }

void Mapper::add(pcl::PointCloud<PointType>::Ptr& pc, double time) {   // Mapper.cpp:88-96
  if (!pc || pc->points.size() < 1) return;
  sync();
  if (!ctx_ && !attach(device_, cell_size_)) return;
  const int rc = flimo_map_add(ctx_, &pc->points[0].x, pc->points.size(), sizeof(PointType), time);
  if (rc != FLIMO_OK) std::cout << "FAST_LIMO::Mapper::add failed: " << flimo_last_error(ctx_) << "\n";
}

Matches Mapper::match(State s, pcl::PointCloud<PointType>::Ptr& pc) {   // Mapper.cpp:59-86
  Matches chosen;
  if (!exists() || !pc) return matches;
  // make `pc` the resident scan, run one pass at `s`, and read the per-point records back
  flimo_scan_set(ctx_, &pc->points[0].x, pc->points.size(), sizeof(PointType));
  double x26[26] = {0};
  x26[0] = s.p(0); x26[1] = s.p(1); x26[2] = s.p(2);
  x26[3] = s.q.x(); x26[4] = s.q.y(); x26[5] = s.q.z(); x26[6] = s.q.w();
  x26[7] = s.qLI.x(); x26[8] = s.qLI.y(); x26[9] = s.qLI.z(); x26[10] = s.qLI.w();
  x26[11] = s.pLI(0); x26[12] = s.pLI(1); x26[13] = s.pLI(2);
  flimo_match_cfg mc{config.NUM_MATCH_POINTS, (int)pc->points.size(), config.MAX_NUM_PC2MATCH, config.MAX_DIST_PLANE,
                     config.PLANE_THRESHOLD, 1};
  double HTH[144], HTh[12];
  int M = 0;
  flimo_set_debug_records(ctx_, 1);
  const int rc = flimo_match_reduce(ctx_, x26, &mc, HTH, HTh, &M);
  if (rc == FLIMO_OK) {
    size_t n = 0;
    flimo_match_fetch(ctx_, nullptr, 0, &n);
    std::vector<flimo_match_rec> recs(n);
    flimo_match_fetch(ctx_, recs.data(), n, &n);
    for (size_t i = 0; i < n; i++) {
      if (recs[i].valid == 0.f) continue;
      const Plane plane(Eigen::Vector4f(recs[i].n[0], recs[i].n[1], recs[i].n[2], recs[i].n[3]), true, &config);
      Match m(Eigen::Vector3f(recs[i].p_global[0], recs[i].p_global[1], recs[i].p_global[2]),
              Eigen::Vector3f(pc->points[i].x, pc->points[i].y, pc->points[i].z), plane);
      m.dist = -recs[i].h;       // the kernel's own value (identical expression, Match.cpp:27)
      chosen.push_back(m);
    }
  }
  flimo_set_debug_records(ctx_, 0);
  matches = chosen;
  return chosen;
}

// ---------------------------------------------------------------------------------------------
// Helper threads of a Localizer: the host-only part of handing the clouds back (filters on the host copy, time order, assembling
// two PCL clouds) runs beside the GPU work instead of after it.  Each worker has its own queue (a task can be pinned to a worker:
// the time-order routine keeps a per-thread memo); wait() returns when everything queued has run.
// ---------------------------------------------------------------------------------------------
namespace flimo_host {
class Helpers {
 public:
  explicit Helpers(int n) {
    for (int i = 0; i < n; i++) w_.emplace_back(new Worker());
    for (auto& w : w_) w->th = std::thread([this, p = w.get()] { loop(*p); });
  }
  ~Helpers() {
    for (auto& w : w_) { { std::lock_guard<std::mutex> g(w->m); w->stop = true; } w->cv.notify_one(); }
    for (auto& w : w_) w->th.join();
  }
  int size() const { return (int)w_.size(); }
  void run(int worker, std::function<void()> f) {
    Worker& w = *w_[(size_t)worker % w_.size()];
    pending_.fetch_add(1, std::memory_order_relaxed);
    { std::lock_guard<std::mutex> g(w.m); w.q.push_back(std::move(f)); w.queued.fetch_add(1, std::memory_order_release); }
    w.cv.notify_one();
  }
  void wait() {
    // short spin first: the tasks are tens to hundreds of microseconds, a sleeping waiter's wake-up costs as much again
    for (int k = 0; k < 20000 && pending_.load(std::memory_order_acquire) != 0; k++) std::this_thread::yield();
    std::unique_lock<std::mutex> g(dm_);
    dcv_.wait(g, [this] { return pending_.load(std::memory_order_acquire) == 0; });
  }
 private:
  struct Worker {
    std::thread th; std::mutex m; std::condition_variable cv; std::deque<std::function<void()>> q; bool stop = false;
    std::atomic<int> queued{0};
  };
  void loop(Worker& w) {
    bool worked = false;
    for (;;) {
      std::function<void()> f;
      if (worked) {
        // the tasks of one sweep follow each other within a few hundred microseconds (filters ... assembly): poll that long
        // before sleeping -- a sleeping thread's wake-up costs as much as the task
        const auto t0 = std::chrono::steady_clock::now();
        while (w.queued.load(std::memory_order_acquire) == 0 &&
               std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(1000)) {}
      }
      {
        std::unique_lock<std::mutex> g(w.m);
        w.cv.wait(g, [&] { return w.stop || !w.q.empty(); });
        if (w.q.empty()) return;
        f = std::move(w.q.front());
        w.q.pop_front();
        w.queued.fetch_sub(1, std::memory_order_relaxed);
      }
      f();
      worked = true;
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> g(dm_); dcv_.notify_all(); }
    }
  }
  std::vector<std::unique_ptr<Worker>> w_;
  std::atomic<int> pending_{0};
  std::mutex dm_;
  std::condition_variable dcv_;
};
}  // namespace flimo_host

// ---------------------------------------------------------------------------------------------
// Localizer
// ---------------------------------------------------------------------------------------------
Localizer::Localizer() : Localizer(&Mapper::getInstance()) { own_map_ = false; }
Localizer::Localizer(Mapper* map)
    : map_(map), own_map_(false), ikfom_(new Esekf()), sensor(SensorType::UNKNOWN), scan_stamp(0.0),
      prev_scan_stamp(0.0), imu_stamp(0.0), prev_imu_stamp(0.0), first_imu_stamp(0.0), last_propagate_time_(-1.0),
      imu_calib_time_(3.0), gravity_(9.81f), imu_calibrated_(false), num_threads_(1), have_prev_ang_(false),
      last_status_(0), cpu_time(0.f), cpu_max_time(0.f), cpu_mean_time(0.f), scans_timed_(0) {
  original_scan = fast_limo::make_shared<pcl::PointCloud<PointType>>();
  deskewed_scan = fast_limo::make_shared<pcl::PointCloud<PointType>>();
  pc2match = fast_limo::make_shared<pcl::PointCloud<PointType>>();
  final_raw_scan = fast_limo::make_shared<pcl::PointCloud<PointType>>();
  final_scan = fast_limo::make_shared<pcl::PointCloud<PointType>>();
  for (int i = 0; i < 4; i++) stage_t_[i] = 0.0;
  last_imu.stamp = 0; last_imu.dt = 0;
}
Localizer::~Localizer() { helpers_.reset(); delete ikfom_; }

void Localizer::init(Config& cfg) {                                // Localizer.cpp:35-117
  config = cfg;
  num_threads_ = config.num_threads < 1 ? 1 : config.num_threads;
  map_->set_num_threads(num_threads_);
  map_->set_config(config.ikfom.mapping);
  map_->attach(config.gpu_device, config.gpu_cell_size);
  init_iKFoM();
  set_sensor_type((uint8_t)config.sensor_type);
  // intrinsics (:67-69): Eigen::Map<Matrix3f> over the flat list is COLUMN-major
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) imu_accel_sm_(i, j) = config.intrinsics.imu_sm[j * 3 + i];
  state.b.accel = Eigen::Vector3f(config.intrinsics.accel_bias[0], config.intrinsics.accel_bias[1], config.intrinsics.accel_bias[2]);
  state.b.gyro = Eigen::Vector3f(config.intrinsics.gyro_bias[0], config.intrinsics.gyro_bias[1], config.intrinsics.gyro_bias[2]);
  // extrinsics (:72-86): column-major map then transposed => the YAML list read row-major
  extr.imu2baselink.t = Eigen::Vector3f(config.extrinsics.imu2baselink_t[0], config.extrinsics.imu2baselink_t[1], config.extrinsics.imu2baselink_t[2]);
  extr.lidar2baselink.t = Eigen::Vector3f(config.extrinsics.lidar2baselink_t[0], config.extrinsics.lidar2baselink_t[1], config.extrinsics.lidar2baselink_t[2]);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    extr.imu2baselink.R(i, j) = config.extrinsics.imu2baselink_R[i * 3 + j];
    extr.lidar2baselink.R(i, j) = config.extrinsics.lidar2baselink_R[i * 3 + j];
  }
  extr.imu2baselink_T = Eigen::Matrix4f::Identity();
  extr.lidar2baselink_T = Eigen::Matrix4f::Identity();
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) { extr.imu2baselink_T(i, j) = extr.imu2baselink.R(i, j); extr.lidar2baselink_T(i, j) = extr.lidar2baselink.R(i, j); }
    extr.imu2baselink_T(i, 3) = extr.imu2baselink.t(i);
    extr.lidar2baselink_T(i, 3) = extr.lidar2baselink.t(i);
  }
  if (!(config.gravity_align || config.calibrate_accel || config.calibrate_gyro)) {   // :92-95
    imu_calibrated_ = true;
    init_iKFoM_state();
  }
  imu_calib_time_ = config.imu_calib_time;
}

void Localizer::init_iKFoM() {                                     // Localizer.cpp:660-670
  ikfom_->init(config.ikfom.MAX_NUM_ITERS, config.ikfom.LIMITS.data());
  // IKFoM::h_share_model (use-ikfom.cpp:10-31) with the reduced seam
  ikfom_->h_reduced_overlap = [this](const StateIkfom& x, flimo_host::ReducedMeas& out, const std::function<void()>& in_flight) {
    double x26[26];
    x.to_flat(x26);
    const Config::iKFoM::Mapping& m = config.ikfom.mapping;
    flimo_match_cfg mc{m.NUM_MATCH_POINTS, m.MAX_NUM_MATCHES, m.MAX_NUM_PC2MATCH, m.MAX_DIST_PLANE, m.PLANE_THRESHOLD,
                       config.ikfom.estimate_extrinsics ? 1 : 0};
    out.M = 0;
    for (int i = 0; i < 144; i++) out.HTH[i] = 0.0;
    for (int i = 0; i < 12; i++) out.HTh[i] = 0.0;
    flimo_ctx* c = map_->ctx();
    if (!c) return;
    const double tm0 = now_s();
    const int rc = flimo_match_reduce_overlap(c, x26, &mc, out.HTH, out.HTh, &out.M,
                                              [](void* f) { (*static_cast<const std::function<void()>*>(f))(); },
                                              const_cast<std::function<void()>*>(&in_flight));
    prof_[2] += now_s() - tm0;
    prof_[3] += 1.0;
    if (rc != FLIMO_OK) {
      // the reference's plug-in cannot fail (Mapper.cpp:59-86); a GPU pass can (timeout, HIP error): the update is abandoned
      std::cout << "FAST_LIMO::match_reduce failed: " << flimo_last_error(c) << "\n";
      out.M = 0;
      ikfom_->failed = true;
    }
  };
  // the whole update enqueued at once (flimo_update_chain): esekfom.hpp:1620-1823 without a host round trip per iteration
  ikfom_->device_chain = [this](const double x26[26], const Esekf::Cov& P, const double* limits, double R, double D, int max_iter,
                                flimo_host::ChainResult& out) {
    out.status = 0;
    flimo_ctx* c = map_->ctx();
    if (!c) return;
    if (!chain_io_) chain_io_.reset(new flimo_chain_io);
    flimo_chain_io& io = *chain_io_;
    std::memcpy(io.x26, x26, sizeof(io.x26));
    std::memcpy(io.P, &P.a[0][0], sizeof(io.P));
    std::memcpy(io.limits, limits, sizeof(io.limits));
    io.R = R; io.D = D; io.max_iter = max_iter;
    io.want_log = ikfom_->keep_log ? 1 : 0;
    const Config::iKFoM::Mapping& m = config.ikfom.mapping;
    flimo_match_cfg mc{m.NUM_MATCH_POINTS, m.MAX_NUM_MATCHES, m.MAX_NUM_PC2MATCH, m.MAX_DIST_PLANE, m.PLANE_THRESHOLD,
                       config.ikfom.estimate_extrinsics ? 1 : 0};
    const double tm0 = now_s();
    const int rc = flimo_update_chain(c, &mc, &io);
    prof_[2] += now_s() - tm0;
    if (rc != FLIMO_OK) {
      std::cout << "FAST_LIMO::update_chain failed: " << flimo_last_error(c) << "\n";
      ikfom_->failed = true;
      return;
    }
    out.status = io.status; out.passes = io.passes; out.it_next = io.it_next; out.t = io.t;
    if (io.status == FLIMO_CHAIN_DECLINED) return;
    prof_[3] += (double)io.passes;
    std::memcpy(out.x, io.x26_out, sizeof(out.x));
    out.have_meas = io.meas_valid != 0;
    if (out.have_meas) {
      out.meas.M = io.meas_M;
      std::memcpy(out.meas.HTH, io.meas_HTH, sizeof(out.meas.HTH));
      std::memcpy(out.meas.HTh, io.meas_HTh, sizeof(out.meas.HTh));
      prof_[3] += 1.0;                                                // (its pass ran on the device too)
    }
    if (ikfom_->keep_log)
      for (int i = 0; i < io.passes; i++) {
        flimo_host::PassLog lg;
        lg.M = io.log[i].M;
        std::memcpy(lg.HTH, io.log[i].HTH, sizeof(lg.HTH));
        std::memcpy(lg.HTh, io.log[i].HTh, sizeof(lg.HTh));
        std::memcpy(lg.dx, io.log[i].dx, sizeof(lg.dx));
        std::memcpy(lg.x_after, io.log[i].x_after, sizeof(lg.x_after));
        ikfom_->log.push_back(lg);
      }
  };
  ikfom_->h_update_end = [this]() {
    flimo_ctx* c = map_->ctx();
    if (c) (void)flimo_pass_pipeline_end(c);
  };
  ikfom_->h_last_iteration = [this]() {
    flimo_ctx* c = map_->ctx();
    if (c) (void)flimo_pass_pipeline_last(c);
  };
  ikfom_->h_dense = [this](flimo_host::DenseMeas& dm) {
    flimo_ctx* c = map_->ctx();
    size_t M = 0;
    dm.H.clear(); dm.h.clear();
    if (!c) return;
    flimo_match_fetch_H(c, nullptr, nullptr, 0, &M);
    dm.H.assign(M * 12, 0.0);
    dm.h.assign(M, 0.0);
    if (M) flimo_match_fetch_H(c, dm.H.data(), dm.h.data(), M, &M);
  };
}

void Localizer::init_iKFoM_state() {                               // Localizer.cpp:672-694
  StateIkfom s = ikfom_->get_x();
  s.rot.x = state.q.x(); s.rot.y = state.q.y(); s.rot.z = state.q.z(); s.rot.w = state.q.w();
  for (int i = 0; i < 3; i++) { s.pos(i, 0) = state.p(i); s.bg(i, 0) = state.b.gyro(i); s.ba(i, 0) = state.b.accel(i); }
  s.grav = flimo_host::S2(0., 0., -(double)gravity_);
  flimo_host::Mat3 Rd;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rd(i, j) = (double)extr.lidar2baselink.R(i, j);
  s.offset_R_L_I = flimo_host::rot_to_quat(Rd);
  for (int i = 0; i < 3; i++) s.offset_T_L_I(i, 0) = extr.lidar2baselink.t(i);
  ikfom_->change_x(s);
  Esekf::Cov P = Esekf::Cov::identity();
  for (int i = 6; i < 12; i++) P(i, i) = 0.000001;
  for (int i = 15; i < 18; i++) P(i, i) = 0.00001;
  for (int i = 18; i < 21; i++) P(i, i) = 0.0001;
  P(21, 21) = P(22, 22) = 0.000001;
  ikfom_->change_P(P);
}

pcl::PointCloud<PointType>::Ptr Localizer::get_pointcloud() { return final_scan; }
pcl::PointCloud<PointType>::Ptr Localizer::get_finalraw_pointcloud() { return final_raw_scan; }
pcl::PointCloud<PointType>::ConstPtr Localizer::get_orig_pointcloud() { return original_scan; }
pcl::PointCloud<PointType>::ConstPtr Localizer::get_deskewed_pointcloud() { return deskewed_scan; }
pcl::PointCloud<PointType>::Ptr Localizer::get_pc2match_pointcloud() { return pc2match; }
Matches& Localizer::get_matches() { return matches; }
bool Localizer::is_calibrated() { return imu_calibrated_; }
void Localizer::set_sensor_type(uint8_t type) { sensor = type < 5 ? static_cast<SensorType>(type) : SensorType::UNKNOWN; }
SensorType Localizer::get_sensor_type() { return sensor; }
double Localizer::get_propagate_time() { return last_propagate_time_; }

State Localizer::getWorldState() {                                 // Localizer.cpp:176-190
  if (!is_calibrated()) return State();
  State out(ikfom_->get_x());
  out.w = last_imu.ang_vel;
  out.a = last_imu.lin_accel;
  out.time = imu_stamp;
  float R[9];
  quat_to_R(out.q, R);
  out.v = Eigen::Vector3f(s3(R[0] * out.v(0), R[3] * out.v(1), R[6] * out.v(2)), s3(R[1] * out.v(0), R[4] * out.v(1), R[7] * out.v(2)),
                          s3(R[2] * out.v(0), R[5] * out.v(1), R[8] * out.v(2)));
  return out;
}
State Localizer::getBodyState() {                                  // Localizer.cpp:158-174
  if (!is_calibrated()) return State();
  State out(ikfom_->get_x());
  out.w = last_imu.ang_vel;
  out.a = last_imu.lin_accel;
  out.time = imu_stamp;
  out.p = Eigen::Vector3f(out.p(0) + out.pLI(0), out.p(1) + out.pLI(1), out.p(2) + out.pLI(2));
  out.q = quat_mul(out.q, out.qLI);
  float R[9];
  quat_to_R(out.q, R);
  out.v = Eigen::Vector3f(s3(R[0] * out.v(0), R[3] * out.v(1), R[6] * out.v(2)), s3(R[1] * out.v(0), R[4] * out.v(1), R[7] * out.v(2)),
                          s3(R[2] * out.v(0), R[5] * out.v(1), R[8] * out.v(2)));
  return out;
}
std::vector<double> Localizer::getPoseCovariance() {               // Localizer.cpp:209-224 (column-major 6x6)
  std::vector<double> cov(36, 0.0);
  if (!is_calibrated()) return cov;
  const Esekf::Cov& P = ikfom_->get_P();
  double Pp[6][6];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    Pp[i][j] = P(3 + i, 3 + j); Pp[i][3 + j] = P(3 + i, j); Pp[3 + i][j] = P(i, 3 + j); Pp[3 + i][3 + j] = P(i, j);
  }
  for (int c = 0; c < 6; c++) for (int r = 0; r < 6; r++) cov[c * 6 + r] = Pp[r][c];
  return cov;
}
std::vector<double> Localizer::getTwistCovariance() {              // Localizer.cpp:226-239
  std::vector<double> cov(36, 0.0);
  if (!is_calibrated()) return cov;
  const Esekf::Cov& P = ikfom_->get_P();
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) cov[j * 6 + i] = P(6 + i, 6 + j);
  for (int i = 3; i < 6; i++) cov[i * 6 + i] = config.ikfom.cov_gyro;
  return cov;
}
void Localizer::get_cpu_stats(float& comput_time, float& max_comput_time, float& mean_comput_time, float& cpu_cores,
                              float& cpu_load, float& cpu_max_load, float& ram_usage) {
  comput_time = cpu_time; max_comput_time = cpu_max_time; mean_comput_time = cpu_mean_time;
  cpu_cores = 0.f; cpu_load = 0.f; cpu_max_load = 0.f; ram_usage = 0.f;   // debug board is out of scope
}

IMUmeas Localizer::imu2baselink(IMUmeas& imu) {                    // Localizer.cpp:696-731
  IMUmeas o;
  double dt = imu.stamp - prev_imu_stamp;
  if ((dt == 0.) || (dt > 0.1)) dt = 1.0 / 200.0;
  const Eigen::Vector3f ang = mat3_mul(extr.imu2baselink.R, imu.ang_vel);
  if (!have_prev_ang_) { ang_vel_cg_prev_ = ang; have_prev_ang_ = true; }
  Eigen::Vector3f acc = mat3_mul(extr.imu2baselink.R, imu.lin_accel);
  const Eigen::Vector3f nt(-extr.imu2baselink.t(0), -extr.imu2baselink.t(1), -extr.imu2baselink.t(2));
  const float fdt = (float)dt;
  const Eigen::Vector3f dw((ang(0) - ang_vel_cg_prev_(0)) / fdt, (ang(1) - ang_vel_cg_prev_(1)) / fdt, (ang(2) - ang_vel_cg_prev_(2)) / fdt);
  const Eigen::Vector3f c1 = cross(dw, nt), c2 = cross(ang, cross(ang, nt));
  acc = Eigen::Vector3f((acc(0) + c1(0)) + c2(0), (acc(1) + c1(1)) + c2(1), (acc(2) + c1(2)) + c2(2));
  ang_vel_cg_prev_ = ang;
  o.ang_vel = ang;
  o.lin_accel = acc;
  o.dt = dt;
  o.stamp = imu.stamp;
  Eigen::Quaternionf q = quat_from_R(extr.imu2baselink.R);
  const float qn = std::sqrt(q.x() * q.x() + q.y() * q.y() + q.z() * q.z() + q.w() * q.w());
  q = Eigen::Quaternionf(q.w() / qn, q.x() / qn, q.y() / qn, q.z() / qn);
  o.q = quat_mul(q, imu.q);
  prev_imu_stamp = imu.stamp;
  return o;
}

void Localizer::propagateImu(const IMUmeas& imu) {                 // Localizer.cpp:583-608
  flimo_host::InputIkfom in;
  for (int i = 0; i < 3; i++) { in.acc(i, 0) = (double)imu.lin_accel(i); in.gyro(i, 0) = (double)imu.ang_vel(i); }
  flimo_host::Mat<12, 12> Q = flimo_host::Mat<12, 12>::identity();
  for (int i = 0; i < 3; i++) {
    Q(i, i) = config.ikfom.cov_gyro; Q(3 + i, 3 + i) = config.ikfom.cov_acc;
    Q(6 + i, 6 + i) = config.ikfom.cov_bias_gyro; Q(9 + i, 9 + i) = config.ikfom.cov_bias_acc;
  }
  mtx_ikfom.lock();
  ikfom_->predict(imu.dt, Q, in);
  const StateIkfom xs = ikfom_->get_x();
  mtx_ikfom.unlock();
  mtx_prop.lock();
  propagated_buffer.push_front(State(xs, imu.stamp, imu.lin_accel, imu.ang_vel));
  if (propagated_buffer.size() > 2000) propagated_buffer.pop_back();
  mtx_prop.unlock();
  last_propagate_time_ = imu.stamp;
}

// Localizer.cpp:917-949: the IMU samples between two stamps, oldest first (the reference hands out reverse iterators)
bool Localizer::imuMeasFromTimeRange(double start_time, double end_time, std::vector<IMUmeas>& meas) {
  meas.clear();
  if (imu_buffer.empty() || imu_buffer.front().stamp < end_time) return false;
  size_t it = 0, last = 0;
  it++;
  const size_t n = imu_buffer.size();
  while (it != n && imu_buffer[it].stamp >= end_time) { last = it; it++; }
  while (it != n && imu_buffer[it].stamp >= start_time) it++;
  if (it == n) return false;
  it++;
  for (size_t k = it; k-- > last;) meas.push_back(imu_buffer[k]);         // reverse iteration it-1 ... last, as in the reference
  imu_range_end_stamp_ = last > 0 ? imu_buffer[last - 1].stamp : imu_buffer[last].stamp;   // `end_imu_it->stamp`
  return true;
}

void Localizer::propagateImu(double t1, double t2) {                 // Localizer.cpp:610-654
  flimo_host::Mat<12, 12> Q = flimo_host::Mat<12, 12>::identity();
  for (int i = 0; i < 3; i++) {
    Q(i, i) = config.ikfom.cov_gyro; Q(3 + i, 3 + i) = config.ikfom.cov_acc;
    Q(6 + i, 6 + i) = config.ikfom.cov_bias_gyro; Q(9 + i, 9 + i) = config.ikfom.cov_bias_acc;
  }
  std::vector<IMUmeas> meas;
  if (!imuMeasFromTimeRange(t1, t2, meas)) { std::cout << "FAST_LIMO::propagateImu(): not enough IMU measurements\n"; return; }
  mtx_ikfom.lock();
  mtx_prop.lock();
  for (const IMUmeas& imu : meas) {
    flimo_host::InputIkfom in;
    for (int i = 0; i < 3; i++) { in.acc(i, 0) = (double)imu.lin_accel(i); in.gyro(i, 0) = (double)imu.ang_vel(i); }
    ikfom_->predict(imu.dt, Q, in);
    propagated_buffer.push_front(State(ikfom_->get_x(), imu.stamp, imu.lin_accel, imu.ang_vel));
    if (propagated_buffer.size() > 2000) propagated_buffer.pop_back();
  }
  mtx_ikfom.unlock();
  mtx_prop.unlock();
  last_propagate_time_ = imu_range_end_stamp_;                       // Localizer.cpp:653
}

void Localizer::calculate_H(const state_ikfom& s, const Matches& ms, Eigen::MatrixXd& H, Eigen::VectorXd& h) {   // Localizer.hpp:176
  const size_t N = ((int)ms.size() > config.ikfom.mapping.MAX_NUM_MATCHES) ? (size_t)config.ikfom.mapping.MAX_NUM_MATCHES : ms.size();
  H = Eigen::MatrixXd::Zero((long)N, 12);
  h = Eigen::VectorXd::Zero((long)N);
  std::vector<double> Hf(N * 12, 0.0), hf(N, 0.0);
  std::vector<float> pg(N * 3), nn(N * 4), dd(N);
  for (size_t i = 0; i < N; i++) {
    const Eigen::Vector3f g = ms[i].get_global_point();
    const Eigen::Vector4f nv = ms[i].plane.get_normal();
    for (int a = 0; a < 3; a++) pg[3 * i + a] = g(a);
    for (int a = 0; a < 4; a++) nn[4 * i + a] = nv(a);
    dd[i] = ms[i].dist;
  }
  double x26[26];
  s.to_flat(x26);
  flimo_calculate_H_host(x26, pg.data(), nn.data(), dd.data(), N, config.ikfom.estimate_extrinsics ? 1 : 0, Hf.data(), hf.data());
  for (size_t i = 0; i < N; i++) {
    for (int j = 0; j < 12; j++) H((long)i, j) = Hf[i * 12 + j];
    h((long)i) = hf[i];
  }
  if (config.debug) matches = ms;
}

// IMU calibration while the robot stands still (Localizer.cpp:411-493): average gyro / accel for
// imu_calib_time_, then gravity-align the attitude (Quaternionf::FromTwoVectors), derive the biases and
// seed the filter state.
void Localizer::calibrateStandStill(const IMUmeas& imu) {
  if ((imu.stamp - first_imu_stamp) < imu_calib_time_) {
    calib_n_++;
    for (int i = 0; i < 3; i++) { calib_gyro_(i) += imu.ang_vel(i); calib_accel_(i) += imu.lin_accel(i); }
    return;
  }
  const float nf = (float)calib_n_;                      // `gyro_avg /= num_samples` (int -> float)
  Eigen::Vector3f gyro_avg(calib_gyro_(0) / nf, calib_gyro_(1) / nf, calib_gyro_(2) / nf);
  Eigen::Vector3f accel_avg(calib_accel_(0) / nf, calib_accel_(1) / nf, calib_accel_(2) / nf);
  Eigen::Vector3f grav_vec(0.f, 0.f, gravity_);
  state.q = imu.q;
  if (config.gravity_align) {
    // grav_vec = (accel_avg - b.accel).normalized() * |g|
    Eigen::Vector3f d(accel_avg(0) - state.b.accel(0), accel_avg(1) - state.b.accel(1), accel_avg(2) - state.b.accel(2));
    const float dn = std::sqrt(s3(d(0) * d(0), d(1) * d(1), d(2) * d(2)));
    const float ag = std::fabs(gravity_);
    grav_vec = Eigen::Vector3f(d(0) / dn * ag, d(1) / dn * ag, d(2) / dn * ag);
    // Eigen::Quaternionf::FromTwoVectors(grav_vec, (0,0,g))  (Eigen/src/Geometry/Quaternion.h setFromTwoVectors)
    const float gn = std::sqrt(s3(grav_vec(0) * grav_vec(0), grav_vec(1) * grav_vec(1), grav_vec(2) * grav_vec(2)));
    const Eigen::Vector3f v0(grav_vec(0) / gn, grav_vec(1) / gn, grav_vec(2) / gn);
    const float zn = std::sqrt(s3(0.f, 0.f, gravity_ * gravity_));
    const Eigen::Vector3f v1(0.f / zn, 0.f / zn, gravity_ / zn);
    const float c = s3(v1(0) * v0(0), v1(1) * v0(1), v1(2) * v0(2));
    if (c < -1.0f + 1e-5f) {
      // opposite vectors: Eigen falls back to an SVD to pick an axis; not reproduced (robot upside down)
      std::cout << "FAST_LIMO::WARNING: gravity points opposite to +z, attitude left unaligned\n";
    } else {
      const Eigen::Vector3f axis = cross(v0, v1);
      const float sq = std::sqrt((1.0f + c) * 2.0f);
      const float invs = 1.0f / sq;
      state.q = Eigen::Quaternionf(sq * 0.5f, axis(0) * invs, axis(1) * invs, axis(2) * invs);
    }
    state.g = grav_vec;
  }
  if (config.calibrate_accel)
    state.b.accel = Eigen::Vector3f(accel_avg(0) - grav_vec(0), accel_avg(1) - grav_vec(1), accel_avg(2) - grav_vec(2));
  if (config.calibrate_gyro) state.b.gyro = gyro_avg;
  {
    const float qn = std::sqrt(state.q.x() * state.q.x() + state.q.y() * state.q.y() + state.q.z() * state.q.z() + state.q.w() * state.q.w());
    state.q = Eigen::Quaternionf(state.q.w() / qn, state.q.x() / qn, state.q.y() / qn, state.q.z() / qn);
  }
  init_iKFoM_state();
  imu_calibrated_ = true;
}

void Localizer::updateIMU(IMUmeas& raw_imu) {                      // Localizer.cpp:401-531
  imu_stamp = raw_imu.stamp;
  IMUmeas imu = imu2baselink(raw_imu);
  if (first_imu_stamp == 0.0) first_imu_stamp = imu.stamp;
  if (!imu_calibrated_) {              // stand-still calibration (Localizer.cpp:411-509)
    calibrateStandStill(imu);
    return;
  }
  const Eigen::Vector3f sm = mat3_mul(imu_accel_sm_, imu.lin_accel);
  imu.lin_accel = Eigen::Vector3f(sm(0) - state.b.accel(0), sm(1) - state.b.accel(1), sm(2) - state.b.accel(2));
  imu.ang_vel = Eigen::Vector3f(imu.ang_vel(0) - state.b.gyro(0), imu.ang_vel(1) - state.b.gyro(1), imu.ang_vel(2) - state.b.gyro(2));
  last_imu = imu;
  imu_buffer.push_front(imu);
  if (imu_buffer.size() > 2000) imu_buffer.pop_back();
  propagateImu(imu);
  cv_prop_stamp.notify_one();
}

bool Localizer::isInRange(const PointType& p) {                    // Localizer.cpp:873-876
  if (!config.filters.fov_active) return true;
  return std::fabs(std::atan2(p.y, p.x)) < config.filters.fov_angle;
}

// Localizer::propagatedFromTimeRange + integrateImu (Localizer.cpp:855-915).  The reference waits on
// cv_prop_stamp without bound and so does this class by default (propagation_wait_s < 0); the C handle bounds the wait
// (flimo_loc_set_propagation_wait) because its callers usually feed IMU and sweeps from one thread.
bool Localizer::propagatedFromTimeRange(double start_time, double end_time, States& frames) {
  frames.clear();
  std::unique_lock<std::mutex> lock(mtx_prop);
  if (propagated_buffer.empty() || propagated_buffer.front().time < end_time) {
    auto reached = [this, &end_time] { return !propagated_buffer.empty() && propagated_buffer.front().time >= end_time; };
    if (propagation_wait_s < 0.0) cv_prop_stamp.wait(lock, reached);
    else cv_prop_stamp.wait_for(lock, std::chrono::duration<double>(propagation_wait_s), reached);
    if (propagated_buffer.empty() || propagated_buffer.front().time < end_time) return false;
  }
  const size_t n = propagated_buffer.size();
  size_t it = 0, last = 0;
  it++;
  while (it != n && propagated_buffer[it].time >= end_time) { last = it; it++; }
  while (it != n && propagated_buffer[it].time >= start_time) it++;
  if (it == n) return false;
  it++;
  for (size_t k = it; k-- > last;) frames.push_back(propagated_buffer[k]);   // oldest -> newest
  return true;
}


// ---------------------------------------------------------------------------------------------------------------------
// Time order of a sweep (reference Localizer.cpp:789-790: std::partial_sort_copy of the whole cloud with a "<" / ">"
// comparator on the stamp).  For equal-size ranges the library call is copy + make_heap + sort_heap, so WHICH of two
// points with equal stamps comes first is decided by the heap's moves and has to be reproduced move for move:
//   * no two equal stamps (and no NaN): the order is unique -- identity / reversal / std::sort give it;
//   * ties: heap_order() below restates libstdc++'s heap routines (bits/stl_heap.h: __make_heap, __adjust_heap,
//     __push_heap, __pop_heap, __sort_heap; bits/stl_algo.h: __partial_sort_copy) on packed (key, index) records with a
//     branch-free child choice -- the same comparisons in the same order, hence the same permutation (tests compare
//     it with the library call on the host's libstdc++), at less than half the time of the library on 16-byte records;
//   * NaN stamps: the comparator is no strict weak order; the library call itself runs (use_library).
// Keys are mapped to unsigned integers that order like the stamps (-0.0 == +0.0 kept equal); descending = complement.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct HeapRec64 { uint64_t key; uint32_t idx; uint32_t pad; };
struct LessPacked64 { bool operator()(uint64_t a, uint64_t b) const { return (a >> 32) < (b >> 32); } };           // key in the high half
struct LessPacked32 { int shift; bool operator()(uint32_t a, uint32_t b) const { return (a >> shift) < (b >> shift); } };   // key above the index bits
struct LessRec64 { bool operator()(const HeapRec64& a, const HeapRec64& b) const { return a.key < b.key; } };

// (no software prefetch: on the Zen 5 host of the MI355X boxes it costs 7 %; smaller records are what helps)
template <class R, class Less>
inline void heap_adjust(R* f, ptrdiff_t hole, ptrdiff_t len, R v, Less less) {           // __adjust_heap + __push_heap
  const ptrdiff_t top = hole;
  ptrdiff_t c = hole;
  const ptrdiff_t lim = (len - 1) / 2;
  while (c < lim) {
    c = 2 * (c + 1);
    c -= (ptrdiff_t)less(f[c], f[c - 1]);                                                // the larger child; right one on ties
    f[hole] = f[c];
    hole = c;
  }
  if ((len & 1) == 0 && c == (len - 2) / 2) {
    c = 2 * (c + 1);
    f[hole] = f[c - 1];
    hole = c - 1;
  }
  ptrdiff_t parent = (hole - 1) / 2;
  while (hole > top && less(f[parent], v)) {
    f[hole] = f[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  f[hole] = v;
}
template <class R, class Less>
void heap_order(R* f, ptrdiff_t len, Less less) {                                        // make_heap + sort_heap
  if (len < 2) return;
  for (ptrdiff_t parent = (len - 2) / 2;; parent--) {
    heap_adjust(f, parent, len, f[parent], less);
    if (parent == 0) break;
  }
  for (ptrdiff_t last = len - 1; last > 0; last--) {                                     // __pop_heap(first, last, last)
    const R v = f[last];
    f[last] = f[0];
    heap_adjust(f, (ptrdiff_t)0, last, v, less);
  }
}
inline uint32_t ord_u32(float x) {
  x += 0.0f;                                                                             // -0.0 -> +0.0
  uint32_t b;
  std::memcpy(&b, &x, 4);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
inline uint64_t ord_u64(double x) {
  x += 0.0;
  uint64_t b;
  std::memcpy(&b, &x, 8);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
template <class T>
void library_order(const T* k, size_t n, bool desc, std::vector<uint32_t>& order) {
  struct Rec { T key; uint32_t idx; };
  std::vector<Rec> in(n), out(n);
  for (size_t i = 0; i < n; i++) in[i] = Rec{k[i], (uint32_t)i};
  if (desc) std::partial_sort_copy(in.begin(), in.end(), out.begin(), out.end(), [](const Rec& a, const Rec& b) { return a.key > b.key; });
  else std::partial_sort_copy(in.begin(), in.end(), out.begin(), out.end(), [](const Rec& a, const Rec& b) { return a.key < b.key; });
  for (size_t i = 0; i < n; i++) order[i] = out[i].idx;
}
template <class T>
void time_order_t(const T* k, size_t n, bool desc, bool use_library, std::vector<uint32_t>& order) {
  order.resize(n);
  if (n == 0) return;
  // one sequential pass: NaN, adjacent ties (the usual way ties show up: the columns of a spinning sensor), monotonicity
  bool nan = false, tie = false, ascending = true, descending = true, nondecreasing = true, nonincreasing = true;
  size_t runs = 1;                                     // runs of equal adjacent keys
  for (size_t i = 0; i < n; i++) {
    nan = nan || (k[i] != k[i]);
    if (i > 0) {
      ascending = ascending && k[i - 1] < k[i];
      descending = descending && k[i - 1] > k[i];
      nondecreasing = nondecreasing && k[i - 1] <= k[i];
      nonincreasing = nonincreasing && k[i - 1] >= k[i];
      tie = tie || k[i - 1] == k[i];
      runs += (k[i - 1] != k[i]) ? 1 : 0;
    }
  }
  if (nan || use_library) { library_order(k, n, desc, order); return; }
  if (!tie && ((!desc && ascending) || (desc && descending))) {                          // strictly ordered already
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)i;
    return;
  }
  if (!tie && ((!desc && descending) || (desc && ascending))) {                          // strictly ordered, reversed
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)(n - 1 - i);
    return;
  }
  // A cloud that is already in time order up to its ties (the columns of a spinning sensor, in firing order) has its keys'
  // ranks for free -- the number of key changes so far -- and the heap only ever compares keys: 4-byte (rank, index) records
  // give the same moves on half the memory (7 % faster on 64k points, 17 % on 256k).
  {
    int idx_bits = 1;
    while (idx_bits < 31 && ((size_t)1 << idx_bits) < n) idx_bits++;
    if ((nondecreasing || nonincreasing) && idx_bits < 31 && runs <= ((size_t)1 << (32 - idx_bits))) {
      std::vector<uint32_t> r(n);
      uint32_t changes = 0;
      for (size_t i = 0; i < n; i++) {
        if (i > 0 && k[i - 1] != k[i]) changes++;
        uint32_t rank = nondecreasing ? changes : (uint32_t)(runs - 1) - changes;       // larger key <=> larger rank
        if (desc) rank = (uint32_t)(runs - 1) - rank;
        r[i] = (rank << idx_bits) | (uint32_t)i;
      }
      // The heap only ever compares ranks, so the permutation is a pure function of this (rank, index) array -- and a spinning
      // sensor that delivers every point of every column repeats it sweep after sweep (same columns, same rings): the last
      // array and its permutation are remembered (per thread: one sensor stream per thread), a repeat costs one comparison of
      // 4 n bytes instead of the heap sort (64k points: 10 us instead of 1.5 ms).
      thread_local std::vector<uint32_t> memo_in, memo_out;
      if (memo_in.size() == n && std::memcmp(memo_in.data(), r.data(), n * sizeof(uint32_t)) == 0) {
        std::memcpy(order.data(), memo_out.data(), n * sizeof(uint32_t));
        return;
      }
      memo_in = r;
      heap_order(r.data(), (ptrdiff_t)n, LessPacked32{idx_bits});
      const uint32_t mask = ((uint32_t)1 << idx_bits) - 1u;
      for (size_t i = 0; i < n; i++) order[i] = r[i] & mask;
      memo_out.assign(order.begin(), order.end());
      return;
    }
  }
  if (sizeof(T) == 4) {
    std::vector<uint64_t> r(n);
    for (size_t i = 0; i < n; i++) {
      uint32_t u;
      if (std::is_same<T, uint32_t>::value) { std::memcpy(&u, &k[i], 4); } else { float x; std::memcpy(&x, &k[i], 4); u = ord_u32(x); }
      if (desc) u = ~u;
      r[i] = ((uint64_t)u << 32) | (uint64_t)i;
    }
    heap_order(r.data(), (ptrdiff_t)n, LessPacked64());
    for (size_t i = 0; i < n; i++) order[i] = (uint32_t)(r[i] & 0xffffffffu);
  } else {
    std::vector<HeapRec64> r(n);
    for (size_t i = 0; i < n; i++) {
      double x;
      std::memcpy(&x, &k[i], 8);
      uint64_t u = ord_u64(x);
      if (desc) u = ~u;
      r[i] = HeapRec64{u, (uint32_t)i, 0};
    }
    heap_order(r.data(), (ptrdiff_t)n, LessRec64());
    for (size_t i = 0; i < n; i++) order[i] = r[i].idx;
  }
}
}  // namespace
// kind: 0 = uint32 (OUSTER t), 1 = float (VELODYNE time), 2 = double (HESAI / LIVOX timestamp)
void fast_limo::time_order(const void* keys, int kind, size_t n, bool desc, bool use_library, std::vector<uint32_t>& order) {
  if (kind == 0) time_order_t(static_cast<const uint32_t*>(keys), n, desc, use_library, order);
  else if (kind == 1) time_order_t(static_cast<const float*>(keys), n, desc, use_library, order);
  else time_order_t(static_cast<const double*>(keys), n, desc, use_library, order);
}

pcl::PointCloud<PointType>::Ptr Localizer::deskewPointCloud(pcl::PointCloud<PointType>::Ptr& pc, double& start_time) {
  const pcl::PointCloud<PointType>::Ptr none;      // the reference returns the deskewed cloud; an early return hands back an empty pointer here   // Localizer.cpp:733-853
  if (pc->points.size() < 1) return none;
  const double sweep_ref_time = start_time;
  const bool eos = config.end_of_sweep;
  std::function<bool(const PointType&, const PointType&)> cmp;
  std::function<double(const PointType&)> extract;
  if (sensor == SensorType::OUSTER) {
    cmp = [eos](const PointType& a, const PointType& b) { return eos ? a.t > b.t : a.t < b.t; };
    extract = [sweep_ref_time, eos](const PointType& p) { return eos ? sweep_ref_time - p.t * 1e-9f : sweep_ref_time + p.t * 1e-9f; };
  } else if (sensor == SensorType::VELODYNE) {
    cmp = [eos](const PointType& a, const PointType& b) { return eos ? a.time > b.time : a.time < b.time; };
    extract = [sweep_ref_time, eos](const PointType& p) { return eos ? sweep_ref_time - p.time : sweep_ref_time + p.time; };
  } else if (sensor == SensorType::HESAI) {
    cmp = [](const PointType& a, const PointType& b) { return a.timestamp < b.timestamp; };
    extract = [](const PointType& p) { return p.timestamp; };
  } else if (sensor == SensorType::LIVOX) {
    cmp = [](const PointType& a, const PointType& b) { return a.timestamp < b.timestamp; };
    extract = [](const PointType& p) { return p.timestamp * 1e-9f; };
  } else {
    std::cout << "FAST_LIMO::FATAL ERROR: LiDAR sensor type unknown or not specified!\n";
    return none;
  }
  // Time order with the same library call as the reference (:789-790) so that ties land identically:
  // std::partial_sort_copy is oblivious to the payload, so running it on 16-byte (key, index) records
  // with a comparator that sees exactly what the reference's comparator sees yields the same
  // permutation as sorting the 32-byte points through std::function, several times faster.
  static const bool prof = std::getenv("FLIMO_PROF_DESKEW") != nullptr;     // developer timing of the host stages
  const double tp0 = prof ? now_s() : 0.0;
  auto sorted = fast_limo::make_shared<pcl::PointCloud<PointType>>();
  const bool desc = eos && (sensor == SensorType::OUSTER || sensor == SensorType::VELODYNE);
  // WHO needs the exact permutation of that call?  The deskew itself works point by point (each point carries its stamp); the
  // order of pc2match is visible (a) through MAX_NUM_PC2MATCH / MAX_NUM_MATCHES ("the first N in pc2match order"), (b) through
  // the float32 centroid sums of the voxel grid, (c) in the clouds handed back to the caller.  When neither cap can bind and the
  // voxel grid is off, the GPU gets the sweep in ARRIVAL order (registration and map insert do not wait for a 1.5 ms sequential
  // heap sort of tied stamps); the permutation is then computed only if the caller wants the clouds, while the GPU works,
  // and applied to the host copies (lazy_order_).  Pose and stored map are the same with and without the clouds.
  lazy_order_.clear();
  arrival_keys_pending_ = false;
  {
    const size_t n0 = pc->points.size();
    const auto& mc = config.ikfom.mapping;
    const bool caps = (mc.MAX_NUM_PC2MATCH >= 0 && n0 > (size_t)mc.MAX_NUM_PC2MATCH) || (mc.MAX_NUM_MATCHES >= 0 && n0 > (size_t)mc.MAX_NUM_MATCHES);
    bool arrival = lazy_time_order && !caps && !config.filters.voxel_active;
    if (arrival) {
      // the stamp of the point the sort would put last: the largest key (the smallest when sorting descending) -- only the KEY of
      // that point is ever read (extract() below), so any point carrying it will do.  Two branch-free passes (the compiler
      // vectorises them): the extreme key with a NaN check, then one index holding it.  NaN stamps: the library path.
      const std::vector<PointType>& P = pc->points;
      size_t last = 0;
      bool nan = false;
      if (sensor == SensorType::OUSTER) {
        uint32_t best = P[0].t;
        if (desc) { for (size_t i = 1; i < n0; i++) best = P[i].t < best ? P[i].t : best; }
        else { for (size_t i = 1; i < n0; i++) best = P[i].t > best ? P[i].t : best; }
        for (size_t i = 0; i < n0; i++) if (P[i].t == best) { last = i; break; }
      } else if (sensor == SensorType::VELODYNE) {
        float best = P[0].time, acc = 0.f;
        if (desc) { for (size_t i = 0; i < n0; i++) { const float v = P[i].time; acc += v * 0.f; best = v < best ? v : best; } }
        else { for (size_t i = 0; i < n0; i++) { const float v = P[i].time; acc += v * 0.f; best = v > best ? v : best; } }
        nan = !(acc == 0.f) || best != best;                    // NaN or inf * 0 anywhere poisons the sum
        for (size_t i = 0; i < n0 && !nan; i++) if (P[i].time == best) { last = i; break; }
      } else {
        double best = P[0].timestamp, acc = 0.0;
        for (size_t i = 0; i < n0; i++) { const double v = P[i].timestamp; acc += v * 0.0; best = v > best ? v : best; }
        nan = !(acc == 0.0) || best != best;
        for (size_t i = 0; i < n0 && !nan; i++) if (P[i].timestamp == best) { last = i; break; }
      }
      if (nan) arrival = false;
      else {
        sorted = pc;                                   // arrival order, no copy
        arrival_last_ = last;
        arrival_keys_pending_ = download_clouds;       // the permutation is wanted for the host clouds only
      }
    }
    arrival_order_ = arrival;
  }
  if (!arrival_order_) {
    const size_t n = pc->points.size();
    const std::vector<PointType>& P = pc->points;
    std::vector<uint32_t> order;
    if (sensor == SensorType::OUSTER) {
      std::vector<uint32_t> k(n);
      for (size_t i = 0; i < n; i++) k[i] = P[i].t;
      time_order(k.data(), 0, n, desc, false, order);
    } else if (sensor == SensorType::VELODYNE) {
      std::vector<float> k(n);
      for (size_t i = 0; i < n; i++) k[i] = P[i].time;
      time_order(k.data(), 1, n, desc, false, order);
    } else {
      std::vector<double> k(n);
      for (size_t i = 0; i < n; i++) k[i] = P[i].timestamp;
      time_order(k.data(), 2, n, desc, false, order);
    }
    sorted->points.resize(n);
    for (size_t i = 0; i < n; i++) sorted->points[i] = P[order[i]];
  }
  (void)cmp;
  const double tp1 = prof ? now_s() : 0.0;
  double offset = 0.0;
  const PointType& last_pt = arrival_order_ ? sorted->points[arrival_last_] : sorted->points.back();
  if (config.time_offset) {
    offset = imu_stamp - extract(last_pt) - 1.e-4;
    if (offset > 0.0) offset = 0.0;
  }
  scan_stamp = extract(last_pt) + offset;
  States frames;
  if (!propagatedFromTimeRange(prev_scan_stamp, scan_stamp, frames) || frames.empty()) {
    std::cout << "FAST_LIMO::propagatedFromTimeRange(): not enough propagated states!\n";
    return none;
  }
  mtx_ikfom.lock();
  const StateIkfom xs = ikfom_->get_x();
  mtx_ikfom.unlock();
  last_state = State(xs);
  // hand over to the GPU: per-point absolute times + the IMU frames
  const size_t n = sorted->points.size();
  std::vector<double> t(n);
  {
    // the same expressions as `extract` above, without a std::function call per point
    const std::vector<PointType>& S = sorted->points;
    if (sensor == SensorType::OUSTER) {
      if (eos) for (size_t k = 0; k < n; k++) t[k] = (sweep_ref_time - S[k].t * 1e-9f) + offset;
      else for (size_t k = 0; k < n; k++) t[k] = (sweep_ref_time + S[k].t * 1e-9f) + offset;
    } else if (sensor == SensorType::VELODYNE) {
      if (eos) for (size_t k = 0; k < n; k++) t[k] = (sweep_ref_time - S[k].time) + offset;
      else for (size_t k = 0; k < n; k++) t[k] = (sweep_ref_time + S[k].time) + offset;
    } else if (sensor == SensorType::HESAI) {
      for (size_t k = 0; k < n; k++) t[k] = S[k].timestamp + offset;
    } else {
      for (size_t k = 0; k < n; k++) t[k] = S[k].timestamp * 1e-9f + offset;
    }
  }
  std::vector<flimo_frame> fr(frames.size());
  for (size_t i = 0; i < frames.size(); i++) {
    const State& F = frames[i];
    flimo_frame& o = fr[i];
    for (int a = 0; a < 3; a++) { o.p[a] = F.p(a); o.v[a] = F.v(a); o.g[a] = F.g(a); o.w[a] = F.w(a); o.a[a] = F.a(a); o.bg[a] = F.b.gyro(a); o.ba[a] = F.b.accel(a); }
    o.q[0] = F.q.x(); o.q[1] = F.q.y(); o.q[2] = F.q.z(); o.q[3] = F.q.w();
    o.time = F.time;
  }
  double x26[26];
  xs.to_flat(x26);
  flimo_ctx* c = map_->ctx();
  if (!c) return none;
  // raw scan + times become resident; frames are kept for registerResident()
  const double tp2 = prof ? now_s() : 0.0;
  int rc = flimo_raw_scan_set(c, &sorted->points[0].x, n, sizeof(PointType), t.data());
  const double tp3 = prof ? now_s() : 0.0;
  if (rc != FLIMO_OK) { std::cout << "FAST_LIMO::raw scan upload failed: " << flimo_last_error(c) << "\n"; return none; }
  rs_frames_.assign(fr.begin(), fr.end());
  compat::to_row_major(extr.lidar2baselink_T, rs_l2b_);
  rc = flimo_deskew_resident(c, rs_frames_.data(), rs_frames_.size(), rs_l2b_, x26);
  if (rc != FLIMO_OK) { std::cout << "FAST_LIMO::deskew failed: " << flimo_last_error(c) << "\n"; return none; }
  const double tp4 = prof ? now_s() : 0.0;
  if (prof)
    fprintf(stderr, "[flimo deskew] time sort + gather %.0f us, frames/times %.0f us, upload + Morton sort %.0f us, deskew call %.0f us (n = %zu)\n",
            (tp1 - tp0) * 1e6, (tp2 - tp1) * 1e6, (tp3 - tp2) * 1e6, (tp4 - tp3) * 1e6, n);
  if (download_clouds && arrival_order_) {
    // the GPU is busy with the deskew (and has the sweep): now the permutation of the reference's sort, for the host clouds
    const std::vector<PointType>& P = sorted->points;
    if (sensor == SensorType::OUSTER) {
      std::vector<uint32_t> k(n);
      for (size_t i = 0; i < n; i++) k[i] = P[i].t;
      time_order(k.data(), 0, n, desc, false, lazy_order_);
    } else if (sensor == SensorType::VELODYNE) {
      std::vector<float> k(n);
      for (size_t i = 0; i < n; i++) k[i] = P[i].time;
      time_order(k.data(), 1, n, desc, false, lazy_order_);
    } else {
      std::vector<double> k(n);
      for (size_t i = 0; i < n; i++) k[i] = P[i].timestamp;
      time_order(k.data(), 2, n, desc, false, lazy_order_);
    }
    std::vector<float> xyz(n * 3);
    size_t m = 0;
    flimo_scan_get(c, xyz.data(), n, &m);
    auto out = fast_limo::make_shared<pcl::PointCloud<PointType>>();
    out->points.resize(n);
    for (size_t k = 0; k < n; k++) {
      const size_t j = lazy_order_[k];                 // position k of pc2match = arrival index j
      PointType p = P[j];
      p.x = xyz[3 * j]; p.y = xyz[3 * j + 1]; p.z = xyz[3 * j + 2];
      out->points[k] = p;
    }
    pc2match = out;
  } else if (download_clouds) {
    std::vector<float> xyz(n * 3);
    size_t m = 0;
    flimo_scan_get(c, xyz.data(), n, &m);
    pc2match = sorted;     // keeps intensity / time of the time-sorted points
    for (size_t k = 0; k < n; k++) { pc2match->points[k].x = xyz[3 * k]; pc2match->points[k].y = xyz[3 * k + 1]; pc2match->points[k].z = xyz[3 * k + 2]; }
  } else {
    pc2match = sorted;
  }
  return pc2match;
}

// The whole pre-update pipeline of a sweep on the GPU: NaN removal, crop box, rate and min-distance
// filters, per-point stamps (flimo_raw_scan_filter_set), Morton order, deskew.  Taken when the sweep may reach the GPU in arrival
// order (see deskewPointCloud) and nobody asked for host copies of the clouds; the FoV filter (host libm's atan2) and NaN stamps
// stay on the host path.  Returns 0 = not applicable (take the host path), 1 = deskewed scan resident, -1 = sweep rejected
// (the same early returns as deskewPointCloud).
bool Localizer::deviceFrontEndEnabled() const {
  if (!gpu_filters || !lazy_time_order) return false;
  return sensor == SensorType::OUSTER || sensor == SensorType::VELODYNE || sensor == SensorType::HESAI || sensor == SensorType::LIVOX;
}

int Localizer::deskewOnDevice(const PointType* raw_points, size_t n, double start_time) {
  const auto& fl = config.filters;
  const auto& mc = config.ikfom.mapping;
  dev_front_end_ = false;
  dev_tied_ = false;
  order_ctx_ = nullptr;
  if (!deviceFrontEndEnabled()) return 0;
  // Upload, filters, stamps and the time order do not read the map: they run on the Mapper's second context while the main one
  // still carries the previous sweep's insert (Mapper::add runs behind publish() there); the main context is waited for only
  // when the sweep is handed over, right before the deskew.
  static const bool prof = std::getenv("FLIMO_PROF_FRONT") != nullptr;     // developer timing of this function's stages
  const double tf0 = prof ? now_s() : 0.0;
  flimo_ctx* const front = map_->front_ctx();
  flimo_ctx* c = front ? front : map_->ctx();
  if (!c) return 0;
  const double tf1 = prof ? now_s() : 0.0;
  // WHO needs the reference's time order on the device?  "The first N of pc2match" (MAX_NUM_PC2MATCH / MAX_NUM_MATCHES) and the
  // float sums of the voxel grid.  It is produced there when the stamps are pairwise different (a radix sort gives the unique
  // sorted order); with equal stamps only the host routine reproduces the library's order: host path.
  const bool caps = (mc.MAX_NUM_PC2MATCH >= 0 && n > (size_t)mc.MAX_NUM_PC2MATCH) || (mc.MAX_NUM_MATCHES >= 0 && n > (size_t)mc.MAX_NUM_MATCHES);
  const bool need_order = caps || fl.voxel_active;
  flimo_filter_cfg fc;
  std::memset(&fc, 0, sizeof(fc));
  fc.crop_active = fl.crop_active ? 1 : 0;
  for (int a = 0; a < 3; a++) { fc.crop_min[a] = fl.cropBoxMin[a]; fc.crop_max[a] = fl.cropBoxMax[a]; }
  fc.dist_active = fl.dist_active ? 1 : 0; fc.min_dist = (float)fl.min_dist;
  fc.rate_active = (fl.rate_active && fl.rate_value >= 1) ? 1 : 0; fc.rate_value = fl.rate_value;
  fc.time_kind = sensor == SensorType::OUSTER ? 0 : (sensor == SensorType::VELODYNE ? 1 : (sensor == SensorType::HESAI ? 2 : 3));
  fc.end_of_sweep = config.end_of_sweep ? 1 : 0;
  fc.sweep_ref_time = start_time;
  fc.fov_active = fl.fov_active ? 1 : 0; fc.fov_angle = fl.fov_angle;
  size_t kept = 0;
  double last_stamp = 0.0;
  int nan = 0, tied = 0;
  static_assert(sizeof(PointType) == 32, "PointType layout");
  const void* src = raw_points;
  bool rec16 = false;
  if (n >= 32768) {
    // pageable cloud -> the context's pinned upload buffer, shared with two helpers (chunks are taken from a common counter: a
    // helper that wakes late finds nothing left and nobody waits for it)
    // A sensor whose stamp is a 32-bit word (OUSTER, VELODYNE) is packed into 16-byte records on the way -- x, y, z and that word
    // are all the device reads of a point: half the bytes over PCIe.
    static const bool no_pack = std::getenv("FLIMO_NO_PACK16") != nullptr;       // developer A/B
    const bool pack = fc.time_kind <= 1 && !no_pack;
    void* stage = nullptr;
    if (flimo_upload_stage(c, n * (pack ? 16 : sizeof(PointType)), &stage) == FLIMO_OK && stage) {
      if (!helpers_) helpers_.reset(new flimo_host::Helpers(3));
      struct CopyJob {
        const char* src; char* dst; size_t n_pts, chunk_pts, nchunks; bool pack;
        std::atomic<size_t> next{0}, done{0};
        void work() {
          for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= nchunks) return;
            const size_t p0 = i * chunk_pts, p1 = std::min(n_pts, p0 + chunk_pts);
            if (!pack) {
              std::memcpy(dst + p0 * 32, src + p0 * 32, (p1 - p0) * 32);
            } else {
#if defined(__SSE2__)
              // (x, y, z, w) and the time word -> (x, y, z, time): two loads, two shuffles, one store per point
              for (size_t p = p0; p < p1; p++) {
                const __m128 a = _mm_loadu_ps(reinterpret_cast<const float*>(src + p * 32));
                const __m128 t = _mm_load_ss(reinterpret_cast<const float*>(src + p * 32 + 24));
                const __m128 zt = _mm_shuffle_ps(a, t, _MM_SHUFFLE(0, 0, 2, 2));                    // (z, z, t, t)
                _mm_storeu_ps(reinterpret_cast<float*>(dst + p * 16), _mm_shuffle_ps(a, zt, _MM_SHUFFLE(2, 0, 1, 0)));   // (x, y, z, t)
              }
#else
              for (size_t p = p0; p < p1; p++) {
                uint32_t rec[4];
                std::memcpy(rec, src + p * 32, 12);
                std::memcpy(rec + 3, src + p * 32 + 24, 4);
                std::memcpy(dst + p * 16, rec, 16);
              }
#endif
            }
            done.fetch_add(1, std::memory_order_release);
          }
        }
      };
      auto job = std::make_shared<CopyJob>();
      job->src = (const char*)src; job->dst = (char*)stage; job->n_pts = n; job->chunk_pts = 4096; job->pack = pack;
      job->nchunks = (n + job->chunk_pts - 1) / job->chunk_pts;
      for (int w = 1; w <= 2; w++) helpers_->run(w, [job] { job->work(); });
      job->work();
      while (job->done.load(std::memory_order_acquire) < job->nchunks) {}
      src = stage;
      if (pack) rec16 = true;
    }
  }
  const double tf2 = prof ? now_s() : 0.0;
  // (equal stamps -- every spinning sensor: all rings of a column share one -- keep their arrival order on the device unless the
  //  caller insists on the order the reference's partial_sort_copy leaves them in: `exact_tied_order`, the host routine)
  if (flimo_raw_scan_filter_order_set(c, src, n, &fc, (need_order ? 1 : 0) | (fl.voxel_active ? 2 : 0) | (rec16 ? 4 : 0) | (exact_tied_order ? 0 : 8),
                                      &kept, &last_stamp, &nan, &tied) != FLIMO_OK || nan || (tied && exact_tied_order))
    return 0;
  dev_tied_ = tied != 0;
  const double tf3 = prof ? now_s() : 0.0;
  double tf4 = tf3;
  order_ctx_ = c;
  if (front) {
    c = map_->ctx();                                                       // waits for the running insert
    if (!c) return 0;
    tf4 = prof ? now_s() : 0.0;
    if (flimo_scan_adopt(c, front) != FLIMO_OK) {
      std::cout << "FAST_LIMO::scan hand-over failed: " << flimo_last_error(c) << "\n";
      return -1;
    }
  }
  if (prof)
    std::fprintf(stderr, "front: since last insert hand-off %.1f us | ctx %.1f  stage copy %.1f  filter %.1f  wait insert %.1f  adopt %.1f | last insert %.1f us\n",
                 1e6 * (tf0 - map_->last_handoff_time()), 1e6 * (tf1 - tf0), 1e6 * (tf2 - tf1), 1e6 * (tf3 - tf2), 1e6 * (tf4 - tf3),
                 1e6 * (now_s() - tf4), 1e6 * map_->last_insert_seconds());
  lazy_order_.clear();
  arrival_order_ = !need_order;
  dev_front_end_ = true;
  dev_time_ordered_ = need_order;
  dev_voxel_ = false;
  releaseRawCloud();                                                       // the raw cloud is uploaded
  pc2match = fast_limo::make_shared<pcl::PointCloud<PointType>>();      // put together after the update when somebody wants it
  if (kept < 1) return -1;
  double offset = 0.0;
  if (config.time_offset) {
    offset = imu_stamp - last_stamp - 1.e-4;
    if (offset > 0.0) offset = 0.0;
  }
  scan_stamp = last_stamp + offset;
  States frames;
  if (!propagatedFromTimeRange(prev_scan_stamp, scan_stamp, frames) || frames.empty()) {
    std::cout << "FAST_LIMO::propagatedFromTimeRange(): not enough propagated states!\n";
    return -1;
  }
  mtx_ikfom.lock();
  const StateIkfom xs = ikfom_->get_x();
  mtx_ikfom.unlock();
  last_state = State(xs);
  std::vector<flimo_frame> fr(frames.size());
  for (size_t i = 0; i < frames.size(); i++) {
    const State& F = frames[i];
    flimo_frame& o = fr[i];
    for (int a = 0; a < 3; a++) { o.p[a] = F.p(a); o.v[a] = F.v(a); o.g[a] = F.g(a); o.w[a] = F.w(a); o.a[a] = F.a(a); o.bg[a] = F.b.gyro(a); o.ba[a] = F.b.accel(a); }
    o.q[0] = F.q.x(); o.q[1] = F.q.y(); o.q[2] = F.q.z(); o.q[3] = F.q.w();
    o.time = F.time;
  }
  double x26[26];
  xs.to_flat(x26);
  rs_frames_.assign(fr.begin(), fr.end());
  compat::to_row_major(extr.lidar2baselink_T, rs_l2b_);
  int rc = flimo_deskew_resident_offset(c, rs_frames_.data(), rs_frames_.size(), rs_l2b_, x26, offset);
  if (rc != FLIMO_OK) { std::cout << "FAST_LIMO::deskew failed: " << flimo_last_error(c) << "\n"; return -1; }
  if (fl.voxel_active) {                                          // VoxelGrid (:313-321) on the resident scan, in time order
    size_t nv = 0;
    rc = flimo_scan_voxel_filter(c, fl.leafSize[0], &nv);
    if (rc != FLIMO_OK) { std::cout << "FAST_LIMO::voxel filter failed: " << flimo_last_error(c) << "\n"; return -1; }
    dev_voxel_ = true;
  }
  return 1;
}

// Input filters of updatePointCloud in ONE pass over the raw cloud: removeNaNFromPointCloud (Localizer.cpp:263-265), negative
// CropBox (:268-271), distance / rate / FoV filters (:274-302).  As in the reference, *raw_pc itself ends up NaN-free and cropped
// (both filters write back into it), and the rate filter counts positions in that cropped cloud.
void Localizer::filterInput(pcl::PointCloud<PointType>::Ptr& raw_pc, pcl::PointCloud<PointType>::Ptr& input_pc, bool compact_raw) {
  const size_t k = filterInput(raw_pc->points.data(), raw_pc->points.size(), input_pc, compact_raw);
  if (!compact_raw) return;
  raw_pc->points.resize(k);
  raw_pc->is_dense = true;
}
// ... on plain memory; returns the number of points that pass NaN removal and the crop box (with compact_raw they are moved to the
// front of P, which is otherwise only read)
size_t Localizer::filterInput(PointType* P, size_t n, pcl::PointCloud<PointType>::Ptr& input_pc, bool compact_raw) {
  const bool crop = config.filters.crop_active, dist = config.filters.dist_active;
  const bool rate_on = config.filters.rate_active && config.filters.rate_value >= 1;   // (the reference divides by the value)
  const bool fov = config.filters.fov_active;
  const float mn0 = crop ? config.filters.cropBoxMin[0] : 0.f, mn1 = crop ? config.filters.cropBoxMin[1] : 0.f,
              mn2 = crop ? config.filters.cropBoxMin[2] : 0.f;
  const float mx0 = crop ? config.filters.cropBoxMax[0] : 0.f, mx1 = crop ? config.filters.cropBoxMax[1] : 0.f,
              mx2 = crop ? config.filters.cropBoxMax[2] : 0.f;
  const float min_dist = (float)config.filters.min_dist;
  const float fov_angle = config.filters.fov_angle;
  const long rate = config.filters.rate_value;
  std::vector<PointType>& Q = input_pc->points;
  Q.resize(n);                                   // upper bound; trimmed below (no per-point capacity checks)
  size_t k = 0, m = 0;
  long phase = 0;                                // k % rate without a division per point
  // Branches on the data itself (which side of the crop box, nearer than min_dist) mispredict on every other point of a
  // real sweep; the tests are evaluated without short-circuit and only their rarely-true combination is branched on, the
  // kept points are stored unconditionally and the output cursor advances by the verdict.
  for (size_t i = 0; i < n; i++) {
    const PointType p = P[i];
    const bool finite = std::isfinite(p.x) & std::isfinite(p.y) & std::isfinite(p.z);
    const bool outside = (p.x < mn0) | (p.y < mn1) | (p.z < mn2) | (p.x > mx0) | (p.y > mx1) | (p.z > mx2);   // not strictly inside the box
    if (!(finite & (!crop | outside))) continue;
    if (compact_raw && k != i) P[k] = p;
    const bool pick = !rate_on || phase == 0;
    if (rate_on && ++phase == rate) phase = 0;
    k++;
    if (!pick) continue;
    bool keep = true;
    if (fov) keep = std::fabs(std::atan2(p.y, p.x)) < fov_angle;                          // isInRange (:873-876)
    if (dist) keep = keep & (std::sqrt(s3(p.x * p.x, p.y * p.y, p.z * p.z)) > min_dist);
    Q[m] = p;
    m += keep ? 1 : 0;
  }
  Q.resize(m);
  return k;
}

// The part of filterInput that writes back into *raw_pc (NaN removal and crop box, Localizer.cpp:263-271), for a caller that ran
// filterInput(..., false) while somebody else was still reading the raw cloud.
void Localizer::compactRaw(pcl::PointCloud<PointType>::Ptr& raw_pc) {
  std::vector<PointType>& P = raw_pc->points;
  const bool crop = config.filters.crop_active;
  const float mn0 = crop ? config.filters.cropBoxMin[0] : 0.f, mn1 = crop ? config.filters.cropBoxMin[1] : 0.f,
              mn2 = crop ? config.filters.cropBoxMin[2] : 0.f;
  const float mx0 = crop ? config.filters.cropBoxMax[0] : 0.f, mx1 = crop ? config.filters.cropBoxMax[1] : 0.f,
              mx2 = crop ? config.filters.cropBoxMax[2] : 0.f;
  const size_t n = P.size();
  size_t k = 0;
  for (size_t i = 0; i < n; i++) {
    const PointType p = P[i];
    const bool finite = std::isfinite(p.x) & std::isfinite(p.y) & std::isfinite(p.z);
    const bool outside = (p.x < mn0) | (p.y < mn1) | (p.z < mn2) | (p.x > mx0) | (p.y > mx1) | (p.z > mx2);
    if (!(finite & (!crop | outside))) continue;
    if (k != i) P[k] = p;
    k++;
  }
  P.resize(k);
  raw_pc->is_dense = true;
}

// Device front end: the clouds the caller may ask for (get_pointcloud(), get_pc2match_pointcloud(), with `debug` get_orig_pointcloud())
// are put together AFTER the update -- the filter's mutex is free, the pose is out -- from the device's buffers and the host copy of
// the sweep: the same input filters on the host copy (which also leaves *raw_pc filtered in place, as the reference does), the
// order of the kept points from the device (its time order, or -- a sweep left in arrival order -- the host's time order routine).
void Localizer::startCloudPrep(pcl::PointCloud<PointType>::Ptr& raw_pc) {
  startCloudPrep(raw_pc->points.data(), raw_pc->points.size());
  prep_raw_ = &raw_pc;                                             // (the caller's pointer outlives the task: materializeClouds waits)
}
void Localizer::startCloudPrep(PointType* raw_points, size_t n_raw) {
  // host-only part, on helper 0 (always the same thread: time_order's memo is per thread) while the GPU runs the passes
  if (!helpers_) helpers_.reset(new flimo_host::Helpers(3));
  prep_input_ = fast_limo::make_shared<pcl::PointCloud<PointType>>();
  prep_started_ = true;
  if (n_raw >= 16384)                                              // the helpers that will share the assembly: awake and polling by then
    for (int w = 1; w < helpers_->size(); w++) helpers_->run(w, [] {});
  // which order will the device hold the sweep in?  (the same rule as deskewOnDevice; if it declines the sweep the host path does
  // its own sort and the order computed here is not used)
  const auto& mc = config.ikfom.mapping;
  const bool caps = (mc.MAX_NUM_PC2MATCH >= 0 && n_raw > (size_t)mc.MAX_NUM_PC2MATCH) || (mc.MAX_NUM_MATCHES >= 0 && n_raw > (size_t)mc.MAX_NUM_MATCHES);
  const bool host_order = !(caps || config.filters.voxel_active);
  helpers_->run(0, [this, raw_points, n_raw, host_order] {
    // the device front end may still be reading the raw cloud (upload): filter WITHOUT writing back
    filterInput(raw_points, n_raw, prep_input_, false);
    if (config.debug) original_scan = fast_limo::make_shared<pcl::PointCloud<PointType>>(*prep_input_);
    if (!host_order) return;                                       // the order comes from the device
    const std::vector<PointType>& P = prep_input_->points;
    const size_t m = P.size();
    std::vector<uint32_t>& order = prep_order_;
    const bool desc = config.end_of_sweep && (sensor == SensorType::OUSTER || sensor == SensorType::VELODYNE);
    if (sensor == SensorType::OUSTER) {
      std::vector<uint32_t> k(m);
      for (size_t i = 0; i < m; i++) k[i] = P[i].t;
      time_order(k.data(), 0, m, desc, false, order);
    } else if (sensor == SensorType::VELODYNE) {
      std::vector<float> k(m);
      for (size_t i = 0; i < m; i++) k[i] = P[i].time;
      time_order(k.data(), 1, m, desc, false, order);
    } else {
      std::vector<double> k(m);
      for (size_t i = 0; i < m; i++) k[i] = P[i].timestamp;
      time_order(k.data(), 2, m, desc, false, order);
    }
  });
  prep_raw_ = nullptr;
}

// The raw cloud is no longer read by anybody else (uploaded, or the device front end declined the sweep): it is left NaN-free and
// cropped, as the reference leaves it -- on a helper, beside the passes.
void Localizer::releaseRawCloud() {
  if (!prep_started_ || !prep_raw_) return;
  pcl::PointCloud<PointType>::Ptr* raw = prep_raw_;
  prep_raw_ = nullptr;
  helpers_->run(0, [this, raw] { compactRaw(*raw); });           // helper 0: behind its read-only filter pass over the same cloud
}

// The device's side of the clouds: deskewed points (body frame), the same in the world frame, and -- a sweep the device put into
// time order -- that order.  Called right after the last pass, before the map insert is handed to the Mapper's thread.
void Localizer::downloadClouds(const double x26[26]) {
  flimo_ctx* c = map_->ctx();
  if (!c) return;
  // resident pc2match (the deskewed points, or the voxel centroids) and its world-frame image: one round trip, no repacking
  size_t n_dev = 0, got = 0;
  if (flimo_scan_clouds(c, x26, &mat_body4_, &mat_world4_, &n_dev) != FLIMO_OK) return;
  mat_n_dev_ = n_dev;
  if (dev_time_ordered_) {
    size_t m = 0;
    flimo_ctx* oc = order_ctx_ ? order_ctx_ : c;                    // the context that ran the input stage keeps the time order
    flimo_raw_scan_order(oc, nullptr, 0, &m);
    prep_order_.resize(m);
    flimo_raw_scan_order(oc, prep_order_.data(), m, &got);
  }
  mat_downloaded_ = true;
}

void Localizer::materializeClouds(size_t n_raw) {
  static const bool prof = std::getenv("FLIMO_PROF_CLOUDS") != nullptr;     // developer timing of the stages
  const double tp0 = prof ? now_s() : 0.0;
  if (!prep_started_) return;
  helpers_->wait();
  prep_started_ = false;
  pcl::PointCloud<PointType>::Ptr input_pc = prep_input_;
  prep_input_.reset();
  const size_t m = input_pc->points.size();
  if (last_status_ != 0 || m == 0 || !mat_downloaded_) { pc2match = fast_limo::make_shared<pcl::PointCloud<PointType>>(); return; }
  mat_downloaded_ = false;
  const double tp1 = prof ? now_s() : 0.0;
  // pc2match position -> index in input_pc
  const std::vector<uint32_t>& order = prep_order_;
  if (!dev_voxel_ && order.size() != m) {
    std::cout << "FAST_LIMO::WARNING: device and host input filters disagree (" << order.size() << " vs " << m << " points)\n";
    return;
  }
  const size_t n_dev = mat_n_dev_;
  const float* body = mat_body4_;
  const float* world = mat_world4_;
  // Without the voxel grid the device's kept set must BE the host's: a different count (a filter decided differently at a boundary:
  // atan2f of the FoV filter on another libm, a NaN rule) would pair every later point with the wrong device point -- and in the
  // arrival-order layout index the pinned buffers beyond their n_dev valid records.  Refused, loudly, instead.
  if (!dev_voxel_ && n_dev != m) {
    std::cout << "FAST_LIMO::WARNING: device and host input filters kept different sets (" << n_dev << " vs " << m
              << " points): no clouds for this sweep\n";
    pc2match = fast_limo::make_shared<pcl::PointCloud<PointType>>();
    final_scan = fast_limo::make_shared<pcl::PointCloud<PointType>>();
    return;
  }
  // (the storage of the last sweep's clouds is taken over when the caller has let go of them: no fresh pages to fault in)
  if (pc2match == mat_pm_) pc2match.reset();
  if (final_scan == mat_fs_) final_scan.reset();
  pcl::PointCloud<PointType>::Ptr pm = (mat_pm_ && mat_pm_.use_count() == 1) ? mat_pm_ : fast_limo::make_shared<pcl::PointCloud<PointType>>();
  pcl::PointCloud<PointType>::Ptr fs = (mat_fs_ && mat_fs_.use_count() == 1) ? mat_fs_ : fast_limo::make_shared<pcl::PointCloud<PointType>>();
  pm->points.resize(n_dev);
  fs->points.resize(n_dev);
  PointType* pmp = pm->points.data();
  PointType* fsp = fs->points.data();
  const PointType* in = input_pc->points.data();
  const bool vox = dev_voxel_, ordered = dev_time_ordered_;
  const size_t n_out = vox ? n_dev : std::min(n_dev, m);
  auto assemble = [=, &order](size_t k0, size_t k1) {
    if (vox) {
      for (size_t k = k0; k < k1; k++) {
        PointType p{};
        p.x = body[4 * k]; p.y = body[4 * k + 1]; p.z = body[4 * k + 2];
        pmp[k] = p;
        p.x = world[4 * k]; p.y = world[4 * k + 1]; p.z = world[4 * k + 2];
        fsp[k] = p;
      }
      return;
    }
    // device order: its time order (position k = input_pc[order[k]]) or arrival order (position j = input_pc[j], shown at rank k)
    for (size_t k = k0; k < k1; k++) {
      const size_t src = order[k];
      const size_t dev = ordered ? k : src;
      if (src >= m || dev >= n_dev) continue;                       // (cannot happen after the count check above; never index past the buffers)
      if (k + 24 < k1) {                                           // the sweep's time order is a scattered walk over three arrays
        const size_t nsrc = order[k + 24];
        __builtin_prefetch(&in[nsrc]);
        if (!ordered) { __builtin_prefetch(&body[4 * nsrc]); __builtin_prefetch(&world[4 * nsrc]); }
      }
      PointType p = in[src];
      p.x = body[4 * dev]; p.y = body[4 * dev + 1]; p.z = body[4 * dev + 2];
      pmp[k] = p;
      p.x = world[4 * dev]; p.y = world[4 * dev + 1]; p.z = world[4 * dev + 2];
      fsp[k] = p;
    }
  };
  // four slices: three helpers and this thread (small clouds: not worth waking anybody)
  const int parts = n_out >= 16384 ? helpers_->size() + 1 : 1;
  const size_t per = (n_out + parts - 1) / parts;
  for (int w = 1; w < parts; w++) helpers_->run(w - 1, [=] { assemble(std::min(n_out, w * per), std::min(n_out, (w + 1) * per)); });
  assemble(0, std::min(n_out, per));
  if (parts > 1) helpers_->wait();
  mat_pm_.reset(); mat_fs_.reset();
  pc2match = pm;
  final_scan = fs;
  mat_pm_ = pm; mat_fs_ = fs;
  if (prof)
    fprintf(stderr, "[flimo clouds] wait for the host filters / order %.0f us, assembly %.0f us (%zu -> %zu points, %zu resident)\n",
            (tp1 - tp0) * 1e6, (now_s() - tp1) * 1e6, n_raw, m, n_dev);
}

// Benchmark entry (inputs resident in HBM): restores the given prior, then GPU deskew of the
// resident raw scan + the iterated update.  No host filters, no PCIe upload, no map insert.
int Localizer::registerResident(const double x26_prior[26], const double* P_prior) {
  flimo_ctx* c = map_->ctx();
  if (!c || rs_frames_.empty()) return -1;
  StateIkfom xs;
  xs.from_flat(x26_prior);
  Esekf::Cov P;
  std::memcpy(&P.a[0][0], P_prior, sizeof(double) * 23 * 23);
  mtx_ikfom.lock();
  ikfom_->change_x(xs);
  ikfom_->change_P(P);
  const double t0 = now_s();
  int rc = flimo_deskew_resident(c, rs_frames_.data(), rs_frames_.size(), rs_l2b_, x26_prior);
  const double t1 = now_s();
  if (rc == FLIMO_OK && flimo_scan_size(c) > 1) {
    ikfom_->update_iterated_dyn_share_modified(0.001, 5.0);
    if (ikfom_->failed) rc = FLIMO_ERR_HIP;                        // a pass failed: state and covariance are the prior's again
  }
  prof_[0] += t1 - t0;
  prof_[1] += now_s() - t1;
  mtx_ikfom.unlock();
  return rc;
}

// updatePointCloud for a caller that holds the sweep as plain memory (a language binding): when the sweep takes the device's input
// stage the points go from the caller's memory straight into the upload buffer (and, when host clouds are wanted, through the
// helper thread's filter pass) -- no cloud object is built.  false: not applicable, nothing was done (the caller builds the cloud and calls updatePointCloud).
bool Localizer::updatePointCloudView(const PointType* points, size_t n, double time_stamp) {
  if (!points || n < 1 || !imu_calibrated_ || imu_buffer.empty()) return false;
  if (!deviceFrontEndEnabled()) return false;
  last_status_ = 0;
  const double t0_dev = now_s();
  prep_started_ = false;
  // (the clouds' host-only part reads the caller's memory too, and leaves it as it is: there is no cloud object to filter in place)
  if (download_clouds || config.debug) startCloudPrep(const_cast<PointType*>(points), n);
  const int on_device = deskewOnDevice(points, n, time_stamp);
  if (on_device == 0) {                                               // (the updatePointCloud that follows does not ask again)
    if (prep_started_) { helpers_->wait(); prep_started_ = false; prep_input_.reset(); }
    device_declined_ = true;
    return false;
  }
  mat_downloaded_ = false;
  finishUpdate(on_device > 0, t0_dev, t0_dev, now_s());
  if (download_clouds || config.debug) {
    const double tm0 = now_s();
    materializeClouds(n);
    stage_t_[0] = now_s() - tm0;
  }
  return true;
}

void Localizer::updatePointCloud(pcl::PointCloud<PointType>::Ptr& raw_pc, double time_stamp) {   // Localizer.cpp:245-399
  const double t0 = now_s();
  last_status_ = 0;
  if (!raw_pc || raw_pc->points.size() < 1) { std::cout << "FAST_LIMO::Raw PointCloud is empty!\n"; last_status_ = -1; return; }
  if (!imu_calibrated_) { last_status_ = -2; return; }
  if (imu_buffer.empty()) { std::cout << "FAST_LIMO::IMU buffer is empty!\n"; last_status_ = -3; return; }
  struct Rearm { bool& f; ~Rearm() { f = false; } } rearm{device_declined_};
  const double t0_dev = now_s();
  // the clouds the caller may ask for: their host-only part (filters on the host copy, time order) starts now, on a helper thread
  const bool want_clouds = download_clouds || config.debug;
  prep_started_ = false;
  if (want_clouds && deviceFrontEndEnabled()) startCloudPrep(raw_pc);
  const int on_device = device_declined_ ? 0 : deskewOnDevice(&raw_pc->points[0], raw_pc->points.size(), time_stamp);   // filters + stamps (+ time order, voxel grid) + deskew on the GPU
  releaseRawCloud();                                                     // (deskewOnDevice does so itself as early as it can)
  if (on_device != 0) {
    const double t2d = now_s();
    mat_downloaded_ = false;
    finishUpdate(on_device > 0, t0_dev, t0_dev, t2d);
    if (download_clouds || config.debug) {
      const double tm0 = now_s();
      materializeClouds(raw_pc->points.size());
      stage_t_[0] = now_s() - tm0;                                       // (host work of this path: the clouds, after the pose)
    }
    return;
  }
  pcl::PointCloud<PointType>::Ptr input_pc;
  if (prep_started_) {                                           // the device front end declined the sweep: the helper has filtered it
    helpers_->wait();
    prep_started_ = false;
    input_pc = prep_input_;
    prep_input_.reset();
  } else {
    input_pc = fast_limo::make_shared<pcl::PointCloud<PointType>>();
    filterInput(raw_pc, input_pc);
    if (config.debug) original_scan = fast_limo::make_shared<pcl::PointCloud<PointType>>(*input_pc);
  }
  const double t1 = now_s();
  bool ok = (bool)deskewPointCloud(input_pc, time_stamp);       // sets pc2match (:307)
  if (!ok) pc2match = fast_limo::make_shared<pcl::PointCloud<PointType>>();
  if (ok && config.filters.voxel_active && map_->ctx()) {          // VoxelGrid (:313-321) on the GPU
    size_t nv = 0;
    const int rc = flimo_scan_voxel_filter(map_->ctx(), config.filters.leafSize[0], &nv);
    if (rc != FLIMO_OK) { std::cout << "FAST_LIMO::voxel filter failed: " << flimo_last_error(map_->ctx()) << "\n"; ok = false; }
    else if (download_clouds) {
      std::vector<float> xyz(nv * 3);
      size_t m = 0;
      flimo_scan_get(map_->ctx(), xyz.data(), nv, &m);
      auto vox = fast_limo::make_shared<pcl::PointCloud<PointType>>();
      vox->points.resize(nv);
      for (size_t k = 0; k < nv; k++) { vox->points[k].x = xyz[3 * k]; vox->points[k].y = xyz[3 * k + 1]; vox->points[k].z = xyz[3 * k + 2]; }
      pc2match = vox;
    }
  }
  finishUpdate(ok, t0, t1, now_s());
}

// Second half of updatePointCloud (Localizer.cpp:323-399): iterated update on the resident scan, state hand-over, path exit
// (transform + Mapper::add), bookkeeping.  t0 / t1 / t2: entry, end of the host preparation, end of the deskew stage.
void Localizer::finishUpdate(bool ok, double t0, double t1, double t2) {
  double t3 = t2, t4 = t2;
  flimo_ctx* c = map_->ctx();
  if (ok && c && flimo_scan_size(c) > 1) {
    mtx_ikfom.lock();
    ikfom_->update_iterated_dyn_share_modified(0.001 /*LiDAR noise*/, 5.0 /*degeneracy threshold*/);   // :333
    if (ikfom_->failed) {
      // A pass of the update failed on the GPU (the reference's Mapper::match cannot, Mapper.cpp:59-86).  The filter is back at the
      // propagated state and covariance; the scan is not inserted into the map at an unmeasured pose; the caller sees status -4 and
      // the reference's own line for a scan that produced no update (Localizer.cpp:379-380).  The next sweep registers normally.
      mtx_ikfom.unlock();
      std::cout << "-------------- FAST_LIMO::NULL ITERATION --------------\n";
      last_status_ = -4;
      stage_t_[0] = t1 - t0; stage_t_[1] = t2 - t1; stage_t_[2] = now_s() - t2; stage_t_[3] = 0.0;
      prev_scan_stamp = scan_stamp;
      return;
    }
    map_->matches.clear();
    State corrected(ikfom_->get_x());
    if (config.calibrate_gyro) corrected.b.gyro = state.b.gyro;
    if (config.calibrate_accel) corrected.b.accel = state.b.accel;
    if (config.gravity_align) corrected.g = state.g;
    state = corrected;
    state.w = last_imu.ang_vel;
    state.a = last_imu.lin_accel;
    double x26[26];
    ikfom_->get_x().to_flat(x26);
    mtx_ikfom.unlock();
    t3 = now_s();
    extr.lidar2baselink_T = state.get_extr_RT();                   // :356
    // transformPointCloud(pc2match -> final_scan) (:361-371) + Mapper::add (:377)
    if (download_clouds && !dev_front_end_) {                       // (device front end: materializeClouds, after this)
      const size_t n = flimo_scan_size(c);
      std::vector<float> w(n * 3);
      flimo_scan_to_world(c, x26, w.data(), n);
      final_scan = fast_limo::make_shared<pcl::PointCloud<PointType>>(*pc2match);
      const bool perm = lazy_order_.size() == n;                    // the device holds the sweep in arrival order
      for (size_t k = 0; k < n && k < final_scan->points.size(); k++) {
        const size_t j = perm ? lazy_order_[k] : k;
        final_scan->points[k].x = w[3 * j]; final_scan->points[k].y = w[3 * j + 1]; final_scan->points[k].z = w[3 * j + 2];
      }
    }
    if (dev_front_end_ && (download_clouds || config.debug)) downloadClouds(x26);   // before the insert takes the context
    if (add_to_map) map_->add_scan(x26, scan_stamp);               // returns at once; the insert overlaps the next scan's host work
    t4 = now_s();
  } else {
    std::cout << "-------------- FAST_LIMO::NULL ITERATION --------------\n";
    last_status_ = 1;
  }
  stage_t_[0] = t1 - t0; stage_t_[1] = t2 - t1; stage_t_[2] = t3 - t2; stage_t_[3] = t4 - t3;
  cpu_time = (float)(t4 - t0);
  if (cpu_time > cpu_max_time) cpu_max_time = cpu_time;
  scans_timed_++;
  cpu_mean_time += (cpu_time - cpu_mean_time) / (float)scans_timed_;
  prev_scan_stamp = scan_stamp;
}

