// fast_limo/Objects/Match.hpp -- one point-to-plane correspondence (reference Objects/Match.hpp:25-47,
// Objects/Match.cpp:23-53).  On the registration path matches stay on the GPU (reduced seam); Mapper::match
// materialises them on request with the plane the fit kernel produced.
#ifndef __FASTLIMO_MATCH_HPP__
#define __FASTLIMO_MATCH_HPP__
#include "fast_limo/Common.hpp"
#include "fast_limo/Objects/Plane.hpp"
#include "fast_limo/Objects/State.hpp"

class fast_limo::Match {
 public:
  fast_limo::Plane plane;
  float dist;                 // signed point-to-plane distance (Match.cpp:23-28)

  Match(const Eigen::Vector3f& p_global_, const Eigen::Vector3f& p_local_, const fast_limo::Plane& H)
      : plane(H), p_global(p_global_), p_local(p_local_) { dist = plane.dist2plane(p_global); }
  Match() : dist(0.f) {}

  bool lisanAlGaib() const { return plane.good_fit(); }
  void update_global(fast_limo::State& s) {                                                    // Match.cpp:34-37
    const Eigen::Matrix4f T = s.get_RT();
    const Eigen::Vector4f l = get_4Dlocal();
    for (int r = 0; r < 3; r++) p_global(r) = ((T(r, 0) * l(0) + T(r, 1) * l(1)) + T(r, 2) * l(2)) + T(r, 3) * l(3);
  }
  Eigen::Vector4f get_4Dglobal() const { return Eigen::Vector4f(p_global(0), p_global(1), p_global(2), 1.0f); }
  Eigen::Vector4f get_4Dlocal() const { return Eigen::Vector4f(p_local(0), p_local(1), p_local(2), 1.0f); }
  Eigen::Vector3f get_global_point() const { return p_global; }
  Eigen::Vector3f get_local_point() const { return p_local; }

 private:
  Eigen::Vector3f p_global, p_local;
};
#endif
