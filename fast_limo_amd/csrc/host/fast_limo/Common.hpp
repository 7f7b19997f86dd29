// fast_limo/Common.hpp -- MI355X-native fast_limo: shared types of the public API.
// Mirrors reference include/fast_limo/Common.hpp:100-172 (Point, Extrinsics, IMUmeas, typedefs).
// The reference's API traffics in Eigen and PCL types; neither library exists in the build image,
// so unless FLIMO_USE_EIGEN_PCL is defined the handful of types on the API surface are provided
// as minimal stand-ins with the same names and member spellings (only what src/main.cpp and
// ROSutils.hpp touch).  With FLIMO_USE_EIGEN_PCL the real headers are used instead.
#ifndef __FASTLIMO_COMMON_HPP__
#define __FASTLIMO_COMMON_HPP__

#include <cstdint>
#include <cmath>
#include <memory>
#include <string>
#include <vector>

#define FAST_LIMO_v "2.1.0-mi355x"

#ifdef FLIMO_USE_EIGEN_PCL
#include <Eigen/Dense>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#else
namespace Eigen {
struct Vector3f {
  float d[3];
  Vector3f() : d{0.f, 0.f, 0.f} {}
  Vector3f(float x, float y, float z) : d{x, y, z} {}
  float& operator()(int i) { return d[i]; }
  float operator()(int i) const { return d[i]; }
  float& operator[](int i) { return d[i]; }
  float operator[](int i) const { return d[i]; }
  float x() const { return d[0]; }
  float y() const { return d[1]; }
  float z() const { return d[2]; }
  static Vector3f Zero() { return Vector3f(); }
};
struct Vector4f {
  float d[4];
  Vector4f() : d{0.f, 0.f, 0.f, 0.f} {}
  Vector4f(float x, float y, float z, float w) : d{x, y, z, w} {}
  float& operator()(int i) { return d[i]; }
  float operator()(int i) const { return d[i]; }
};
struct Quaternionf {
  float qx, qy, qz, qw;
  Quaternionf() : qx(0.f), qy(0.f), qz(0.f), qw(1.f) {}
  Quaternionf(float w, float x, float y, float z) : qx(x), qy(y), qz(z), qw(w) {}   // Eigen order (w,x,y,z)
  float x() const { return qx; }
  float y() const { return qy; }
  float z() const { return qz; }
  float w() const { return qw; }
  static Quaternionf Identity() { return Quaternionf(); }
};
struct Matrix3f {
  float m[9];   // row-major storage; access through (row, col)
  Matrix3f() { for (int i = 0; i < 9; i++) m[i] = 0.f; }
  float& operator()(int r, int c) { return m[r * 3 + c]; }
  float operator()(int r, int c) const { return m[r * 3 + c]; }
  static Matrix3f Identity() { Matrix3f o; o.m[0] = o.m[4] = o.m[8] = 1.f; return o; }
};
struct Matrix4f {
  float m[16];
  Matrix4f() { for (int i = 0; i < 16; i++) m[i] = 0.f; }
  float& operator()(int r, int c) { return m[r * 4 + c]; }
  float operator()(int r, int c) const { return m[r * 4 + c]; }
  static Matrix4f Identity() { Matrix4f o; o.m[0] = o.m[5] = o.m[10] = o.m[15] = 1.f; return o; }
};
}  // namespace Eigen
namespace pcl {
struct PointXYZ {
  float x, y, z, pad_;
  PointXYZ() : x(0.f), y(0.f), z(0.f), pad_(1.f) {}
  PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_), pad_(1.f) {}
};
template <typename PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  bool is_dense = true;
  std::uint32_t width = 0, height = 1;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
};
}  // namespace pcl
#endif  // FLIMO_USE_EIGEN_PCL

namespace fast_limo {
enum class SensorType { OUSTER, VELODYNE, HESAI, LIVOX, UNKNOWN };

class Localizer;
class Mapper;
class State;
class Plane;
class Match;
struct Config;

// PointType: 32 bytes, xyz1 + intensity + time union (reference Common.hpp:100-113)
struct alignas(16) Point {
  float x, y, z, data_w;
  float intensity;
  union {
    std::uint32_t t;   // (Ouster) ns since the beginning of the scan
    float time;        // (Velodyne) s since the beginning of the scan
    double timestamp;  // (Hesai) absolute s / (Livox) absolute ns
  };
  Point() : x(0.f), y(0.f), z(0.f), data_w(1.f), intensity(0.f), timestamp(0.0) {}
  Point(float x_, float y_, float z_) : x(x_), y(y_), z(z_), data_w(1.f), intensity(0.f), timestamp(0.0) {}
};
static_assert(sizeof(Point) == 32, "PointType must stay 32 bytes");

struct Extrinsics {
  struct SE3 {
    Eigen::Vector3f t;
    Eigen::Matrix3f R;
  };
  SE3 imu2baselink;
  SE3 lidar2baselink;
  Eigen::Matrix4f imu2baselink_T;
  Eigen::Matrix4f lidar2baselink_T;
};

struct IMUmeas {
  double stamp;
  double dt;
  Eigen::Vector3f ang_vel;
  Eigen::Vector3f lin_accel;
  Eigen::Quaternionf q;
};

template <typename T>
using shared_ptr = std::shared_ptr<T>;
template <typename T, typename... Args>
std::shared_ptr<T> make_shared(Args&&... args) { return std::make_shared<T>(std::forward<Args>(args)...); }
// Order of a sweep's points by time stamp exactly as the reference's std::partial_sort_copy leaves it (Localizer.cpp:789-790),
// ties included.  kind: 0 = uint32 (OUSTER t), 1 = float (VELODYNE time), 2 = double (HESAI / LIVOX timestamp).
void time_order(const void* keys, int kind, size_t n, bool descending, bool use_library, std::vector<uint32_t>& order);
}  // namespace fast_limo

typedef fast_limo::Point PointType;
typedef pcl::PointXYZ MapPoint;
typedef std::vector<pcl::PointXYZ> MapPoints;
typedef std::vector<fast_limo::Match> Matches;
typedef std::vector<fast_limo::State> States;

#endif
