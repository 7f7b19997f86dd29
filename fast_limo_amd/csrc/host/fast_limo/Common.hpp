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
#include <pcl/pcl_config.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#if PCL_VERSION_COMPARE(<, 1, 11, 0)
#include <boost/make_shared.hpp>
#include <boost/shared_ptr.hpp>
#endif
#else
// Stand-ins: storage is PRIVATE, the only way in is the part of Eigen's / PCL's own interface the reference's callers use
// (src/main.cpp, ROSutils.hpp, Localizer.hpp) -- so code that compiles against them compiles against the real headers as far
// as these types go: no member spelling of the stand-ins can leak into the library (Eigen's matrices are column-major and
// expose no such members).
namespace Eigen {
// (templates on the scalar only so that `v.cast<double>()` -- what ROSutils.hpp does with a State's members -- has a type to
//  return; Vector3f / Vector3d / Quaternionf / Quaterniond are the spellings the reference's callers use)
template <typename S>
class FlimoVector3 {
  S d_[3];
 public:
  FlimoVector3() : d_{S(0), S(0), S(0)} {}
  FlimoVector3(S x, S y, S z) : d_{x, y, z} {}
  S& operator()(int i) { return d_[i]; }
  S operator()(int i) const { return d_[i]; }
  S& operator[](int i) { return d_[i]; }
  S operator[](int i) const { return d_[i]; }
  S x() const { return d_[0]; }
  S y() const { return d_[1]; }
  S z() const { return d_[2]; }
  static FlimoVector3 Zero() { return FlimoVector3(); }
  template <typename T>
  FlimoVector3<T> cast() const { return FlimoVector3<T>((T)d_[0], (T)d_[1], (T)d_[2]); }
};
typedef FlimoVector3<float> Vector3f;
typedef FlimoVector3<double> Vector3d;
class Vector4f {
  float d_[4];
 public:
  Vector4f() : d_{0.f, 0.f, 0.f, 0.f} {}
  Vector4f(float x, float y, float z, float w) : d_{x, y, z, w} {}
  float& operator()(int i) { return d_[i]; }
  float operator()(int i) const { return d_[i]; }
};
template <typename S>
class FlimoQuaternion {
  S qx_, qy_, qz_, qw_;
 public:
  FlimoQuaternion() : qx_(S(0)), qy_(S(0)), qz_(S(0)), qw_(S(1)) {}
  FlimoQuaternion(S w, S x, S y, S z) : qx_(x), qy_(y), qz_(z), qw_(w) {}   // Eigen order (w,x,y,z)
  S x() const { return qx_; }
  S y() const { return qy_; }
  S z() const { return qz_; }
  S w() const { return qw_; }
  static FlimoQuaternion Identity() { return FlimoQuaternion(); }
  template <typename T>
  FlimoQuaternion<T> cast() const { return FlimoQuaternion<T>((T)qw_, (T)qx_, (T)qy_, (T)qz_); }
};
typedef FlimoQuaternion<float> Quaternionf;
typedef FlimoQuaternion<double> Quaterniond;
class Matrix3f {
  float m_[9];
 public:
  Matrix3f() { for (int i = 0; i < 9; i++) m_[i] = 0.f; }
  float& operator()(int r, int c) { return m_[r * 3 + c]; }
  float operator()(int r, int c) const { return m_[r * 3 + c]; }
  static Matrix3f Identity() { Matrix3f o; o(0, 0) = o(1, 1) = o(2, 2) = 1.f; return o; }
};
class Matrix4f {
  float m_[16];
 public:
  Matrix4f() { for (int i = 0; i < 16; i++) m_[i] = 0.f; }
  float& operator()(int r, int c) { return m_[r * 4 + c]; }
  float operator()(int r, int c) const { return m_[r * 4 + c]; }
  static Matrix4f Identity() { Matrix4f o; o(0, 0) = o(1, 1) = o(2, 2) = o(3, 3) = 1.f; return o; }
};
// dynamic double matrix / vector of the measurement seam (Localizer::calculate_H, Localizer.hpp:176)
class MatrixXd {
  std::vector<double> v_;
  long r_, c_;
 public:
  MatrixXd() : r_(0), c_(0) {}
  MatrixXd(long r, long c) : v_((size_t)(r * c), 0.0), r_(r), c_(c) {}
  static MatrixXd Zero(long r, long c) { return MatrixXd(r, c); }
  long rows() const { return r_; }
  long cols() const { return c_; }
  double& operator()(long r, long c) { return v_[(size_t)(r * c_ + c)]; }
  double operator()(long r, long c) const { return v_[(size_t)(r * c_ + c)]; }
};
class VectorXd {
  std::vector<double> v_;
 public:
  VectorXd() {}
  explicit VectorXd(long n) : v_((size_t)n, 0.0) {}
  static VectorXd Zero(long n) { return VectorXd(n); }
  long size() const { return (long)v_.size(); }
  double& operator()(long i) { return v_[(size_t)i]; }
  double operator()(long i) const { return v_[(size_t)i]; }
};
}  // namespace Eigen
namespace pcl {
struct PointXYZ {
  float x, y, z;
 private:
  float pad_;
 public:
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

// the smart pointer PCL's clouds use (reference Common.hpp:134-152): boost's before PCL 1.11, the standard one since
#ifdef FLIMO_USE_EIGEN_PCL
#if PCL_VERSION_COMPARE(<, 1, 11, 0)
#define FLIMO_PCL_BOOST_PTR 1
#endif
#endif
#ifdef FLIMO_PCL_BOOST_PTR
template <typename T>
using shared_ptr = boost::shared_ptr<T>;
template <typename T, typename... Args>
boost::shared_ptr<T> make_shared(Args&&... args) { return boost::make_shared<T>(std::forward<Args>(args)...); }
#else
template <typename T>
using shared_ptr = std::shared_ptr<T>;
template <typename T, typename... Args>
std::shared_ptr<T> make_shared(Args&&... args) { return std::make_shared<T>(std::forward<Args>(args)...); }
#endif

// The library's own accessors for the few places where a matrix crosses the C ABI as a flat array: written against
// operator()(row, col) only, so they read the same numbers from a stand-in and from a (column-major) Eigen matrix.
namespace compat {
inline void to_row_major(const Eigen::Matrix4f& M, float out[16]) {
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out[r * 4 + c] = M(r, c);
}
inline void to_row_major(const Eigen::Matrix3f& M, float out[9]) {
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out[r * 3 + c] = M(r, c);
}
}  // namespace compat
// Order of a sweep's points by time stamp exactly as the reference's std::partial_sort_copy leaves it (Localizer.cpp:789-790),
// ties included.  kind: 0 = uint32 (OUSTER t), 1 = float (VELODYNE time), 2 = double (HESAI / LIVOX timestamp).
void time_order(const void* keys, int kind, size_t n, bool descending, bool use_library, std::vector<uint32_t>& order);
}  // namespace fast_limo

#ifdef FLIMO_USE_EIGEN_PCL
POINT_CLOUD_REGISTER_POINT_STRUCT(fast_limo::Point,
                                 (float, x, x)(float, y, y)(float, z, z)(float, intensity, intensity)
                                 (std::uint32_t, t, t)(float, time, time)(double, timestamp, timestamp))
#endif
typedef fast_limo::Point PointType;
typedef pcl::PointXYZ MapPoint;
typedef std::vector<pcl::PointXYZ> MapPoints;
typedef std::vector<fast_limo::Match> Matches;
typedef std::vector<fast_limo::State> States;

#endif
