// fast_limo/Modules/Localizer.hpp -- the reference's Localizer API (Modules/Localizer.hpp:138-209)
// on top of the MI355X hot path.  updatePointCloud = filters -> time sort -> GPU deskew ->
// host IESKF with the GPU measurement seam -> GPU transform -> map insert.
#ifndef __FASTLIMO_LOCALIZER_HPP__
#define __FASTLIMO_LOCALIZER_HPP__
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include "fast_limo/Common.hpp"
#include "flimo_c.h"
#include "fast_limo/Modules/Mapper.hpp"
#include "fast_limo/Objects/Match.hpp"
#include "fast_limo/Objects/State.hpp"
#include "fast_limo/Utils/Config.hpp"

namespace flimo_host { class Esekf; struct StateIkfom; class Helpers; }
typedef flimo_host::StateIkfom state_ikfom;     // the reference's name of the filter state (IKFoM/use-ikfom.hpp:12-21)

class fast_limo::Localizer {
 public:
  pcl::PointCloud<PointType>::Ptr pc2match;   // body frame at scan end (Localizer.hpp:36)

  Localizer();
  ~Localizer();
  void init(Config& cfg);

  // callbacks
  void updateIMU(IMUmeas& raw_imu);
  void updatePointCloud(pcl::PointCloud<PointType>::Ptr& raw_pc, double time_stamp);
  bool updatePointCloudView(const PointType* points, size_t n, double time_stamp);   // bindings: see fast_limo.cpp

  // outputs
  pcl::PointCloud<PointType>::Ptr get_pointcloud();
  pcl::PointCloud<PointType>::Ptr get_finalraw_pointcloud();
  pcl::PointCloud<PointType>::ConstPtr get_orig_pointcloud();
  pcl::PointCloud<PointType>::ConstPtr get_deskewed_pointcloud();
  pcl::PointCloud<PointType>::Ptr get_pc2match_pointcloud();
  Matches& get_matches();
  State getWorldState();
  State getBodyState();
  std::vector<double> getPoseCovariance();
  std::vector<double> getTwistCovariance();
  double get_propagate_time();
  void get_cpu_stats(float& comput_time, float& max_comput_time, float& mean_comput_time, float& cpu_cores,
                     float& cpu_load, float& cpu_max_load, float& ram_usage);
  bool is_calibrated();
  void set_sensor_type(uint8_t type);
  fast_limo::SensorType get_sensor_type();
  void propagateImu(const IMUmeas& imu);
  void propagateImu(double t1, double t2);                 // Localizer.cpp:610-654 (unused by the reference's own callers)
  // iKFoM measurement model (Localizer.hpp:176, Localizer.cpp:537-577): H is N x 12, N = min(matches, MAX_NUM_MATCHES).  The
  // registration path builds the same rows on the GPU.
  void calculate_H(const state_ikfom&, const Matches&, Eigen::MatrixXd& H, Eigen::VectorXd& h);

  // --- MI355X additions -----------------------------------------------------------------------
  explicit Localizer(Mapper* map);      // non-singleton instances (one per GPU, SURVEY.md 8 e)
  Mapper& mapper() { return *map_; }
  flimo_host::Esekf& filter() { return *ikfom_; }
  int last_status() const { return last_status_; }    // 0 ok, 1 null iteration, <0 early return
  double last_scan_stamp() const { return scan_stamp; }
  // timings of the last updatePointCloud [s]: host prep (filters+sort), deskew, update, map insert
  void get_stage_times(double t[4]) const { for (int i = 0; i < 4; i++) t[i] = stage_t_[i]; }
  // benchmark entry: re-register the scan made resident by the last updatePointCloud from a given
  // prior (x26, P 23x23 row-major); GPU deskew + iterated update only
  int registerResident(const double x26_prior[26], const double* P_prior);
  // host-side profile accumulators [s]: deskew call, whole update, time inside flimo_match_reduce, passes
  double prof_[4] = {0, 0, 0, 0};
  bool add_to_map = true;               // benchmarks may freeze the map
  bool download_clouds = true;          // keep pc2match / final_scan host copies up to date
  bool gpu_filters = true;              // input filters + stamps (+ time order, when the stamps are pairwise different) on the GPU; the clouds
                                        // the caller gets are then put together AFTER the update (materializeClouds)
  double propagation_wait_s = -1.0;     // propagatedFromTimeRange: < 0 waits for the IMU stream without bound (the reference,
                                        // Localizer.cpp:859-863); >= 0 gives up after that many seconds (single-threaded drivers)
  bool lazy_time_order = true;          // the GPU gets the sweep in arrival order whenever the time order is not observable through
                                        // caps / voxel sums (deskewPointCloud); false: always the reference's permutation first

  bool exact_tied_order = false;        // equal stamps (a spinning sensor) in a sweep whose time order is observable (caps / voxel grid on):
                                        // false: the device's stable order, arrival order among the equal ones -- the sweep never leaves the
                                        // GPU; true: the order std::partial_sort_copy leaves them in (Localizer.cpp:789-790), reproduced move
                                        // for move by the host front end (bit-exact centroids and cap membership, 1.5 ms per 64k sweep)
  bool last_sweep_tied() const { return dev_tied_; }      // the last sweep of the device front end had equal stamps

  static Localizer& getInstance() {
    static Localizer* loc = new Localizer();
    return *loc;
  }

 private:
  void init_iKFoM();
  void init_iKFoM_state();
  IMUmeas imu2baselink(IMUmeas& imu);
  void calibrateStandStill(const IMUmeas& imu);
  pcl::PointCloud<PointType>::Ptr deskewPointCloud(pcl::PointCloud<PointType>::Ptr& pc, double& start_time);   // Localizer.hpp:191
  int deskewOnDevice(const PointType* raw_points, size_t n, double start_time);       // filters + stamps + deskew on the GPU (f-2)
  void finishUpdate(bool ok, double t0, double t1, double t2);
  void filterInput(pcl::PointCloud<PointType>::Ptr& raw_pc, pcl::PointCloud<PointType>::Ptr& input_pc, bool compact_raw = true);   // Localizer.cpp:262-302 in one pass
  size_t filterInput(PointType* P, size_t n, pcl::PointCloud<PointType>::Ptr& input_pc, bool compact_raw);
  void materializeClouds(size_t n_raw);     // device front end: host clouds after the update
  bool propagatedFromTimeRange(double start_time, double end_time, States& frames);
  bool imuMeasFromTimeRange(double start_time, double end_time, std::vector<IMUmeas>& meas);   // Localizer.cpp:917-949, oldest first
  bool isInRange(const PointType& p);

  Mapper* map_;
  bool own_map_;
  flimo_host::Esekf* ikfom_;
  std::unique_ptr<flimo_chain_io> chain_io_;          // arguments / results of flimo_update_chain (20 kB: kept, not on the stack)
  std::mutex mtx_ikfom, mtx_prop;
  std::condition_variable cv_prop_stamp;
  State state, last_state;
  Extrinsics extr;
  SensorType sensor;
  IMUmeas last_imu;
  std::deque<IMUmeas> imu_buffer;         // front = newest, capacity 2000
  double imu_range_end_stamp_ = 0.0;      // stamp of the sample right after the last imuMeasFromTimeRange range
  std::deque<State> propagated_buffer;
  Config config;
  Eigen::Matrix3f imu_accel_sm_;
  pcl::PointCloud<PointType>::ConstPtr original_scan, deskewed_scan;
  pcl::PointCloud<PointType>::Ptr final_raw_scan, final_scan;
  Matches matches;
  double scan_stamp, prev_scan_stamp, imu_stamp, prev_imu_stamp, first_imu_stamp, last_propagate_time_;
  double imu_calib_time_;
  float gravity_;
  bool imu_calibrated_;
  int num_threads_;
  bool have_prev_ang_;
  Eigen::Vector3f ang_vel_cg_prev_;
  std::vector<flimo_frame> rs_frames_;  // frames of the resident raw scan
  std::vector<uint32_t> lazy_order_;    // arrival-order sweeps: pc2match position -> arrival index (empty: device order = pc2match order)
  bool arrival_order_ = false, arrival_keys_pending_ = false;
  bool dev_front_end_ = false;          // the last sweep went through the device front end (clouds materialized afterwards)
  bool device_declined_ = false;        // updatePointCloudView asked the device front end for this sweep and was turned down
  flimo_ctx* order_ctx_ = nullptr;      // the context that ran this sweep's input stage (it keeps the time order)
  bool dev_tied_ = false;
  bool dev_time_ordered_ = false;       // ... and the device holds it in the reference's time order (stamps pairwise different)
  bool dev_voxel_ = false;
  const float* mat_body4_ = nullptr;          // the downloaded clouds (float4 records in the context's pinned memory)
  const float* mat_world4_ = nullptr;
  // Materialization in three steps: the host-only part (input filters on the host copy, debug copy, host time order) runs on a
  // helper thread beside the GPU's front end and passes; the downloads follow the last pass (before the map insert is handed to
  // the Mapper's thread, which then owns the context); the two clouds are assembled by the helpers and the caller's thread.
  std::unique_ptr<flimo_host::Helpers> helpers_;
  pcl::PointCloud<PointType>::Ptr prep_input_;
  std::vector<uint32_t> prep_order_;         // materialized pc2match position -> index among the host-filtered points
  size_t mat_n_dev_ = 0;
  bool mat_downloaded_ = false, prep_started_ = false;
  pcl::PointCloud<PointType>::Ptr* prep_raw_ = nullptr;   // the caller's cloud while its in-place filtering is still owed (releaseRawCloud)
  void releaseRawCloud();
  bool deviceFrontEndEnabled() const;
  void compactRaw(pcl::PointCloud<PointType>::Ptr& raw_pc);
  void startCloudPrep(pcl::PointCloud<PointType>::Ptr& raw_pc);
  void startCloudPrep(PointType* raw_points, size_t n_raw);
  void downloadClouds(const double x26[26]);
  pcl::PointCloud<PointType>::Ptr mat_pm_, mat_fs_;   // the clouds it handed out last (their storage is reused once the caller let go)
  size_t arrival_last_ = 0;             // index of the point the reference's sort would put last
  float rs_l2b_[16];
  int calib_n_ = 0;
  Eigen::Vector3f calib_gyro_, calib_accel_;
  int last_status_;
  double stage_t_[4];
  float cpu_time, cpu_max_time, cpu_mean_time;
  long scans_timed_;

  Localizer(const Localizer&) = delete;
  Localizer& operator=(const Localizer&) = delete;
};
#endif
