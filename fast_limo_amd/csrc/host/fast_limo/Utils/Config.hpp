// fast_limo/Utils/Config.hpp -- field-for-field mirror of reference Utils/Config.hpp:23-95.
#ifndef __FASTLIMO_CONFIG_HPP__
#define __FASTLIMO_CONFIG_HPP__
#include "fast_limo/Common.hpp"

struct fast_limo::Config {
  struct Topics { std::string lidar; std::string imu; } topics;
  struct Extrinsics {
    std::vector<float> imu2baselink_t, imu2baselink_R, lidar2baselink_t, lidar2baselink_R;
  } extrinsics;
  struct Intrinsics { std::vector<float> accel_bias, gyro_bias, imu_sm; } intrinsics;
  struct Filters {
    std::vector<float> cropBoxMin, cropBoxMax;
    bool crop_active;
    std::vector<float> leafSize;
    bool voxel_active;
    double min_dist;
    bool dist_active;
    int rate_value;
    bool rate_active;
    float fov_angle;
    bool fov_active;
  } filters;
  struct iKFoM {
    struct Mapping {
      int NUM_MATCH_POINTS;
      int MAX_NUM_MATCHES;
      int MAX_NUM_PC2MATCH;
      double MAX_DIST_PLANE;
      double PLANE_THRESHOLD;
      struct Octree { int bucket_size; float min_extent; bool downsampling; } octree;
    } mapping;
    int MAX_NUM_ITERS;
    std::vector<double> LIMITS;
    bool estimate_extrinsics;
    double cov_gyro, cov_acc, cov_bias_gyro, cov_bias_acc;
  } ikfom;
  bool gravity_align, calibrate_accel, calibrate_gyro, time_offset, end_of_sweep;
  bool debug, verbose;
  int sensor_type, num_threads;
  double imu_calib_time;
  // MI355X additions (ignored by the reference): GPU ordinal and grid cell edge [m]
  int gpu_device = 0;
  float gpu_cell_size = 0.f;

  // defaults of the reference's load_config (src/main.cpp:101-168)
  static Config defaults();
};
#endif
