// fast_limo/Objects/Match.hpp -- one point-to-plane correspondence (reference Objects/Match.hpp:25-47,
// Objects/Plane.hpp:24-61 folded in: the plane lives on the GPU, only its result is exposed).
#ifndef __FASTLIMO_MATCH_HPP__
#define __FASTLIMO_MATCH_HPP__
#include "fast_limo/Common.hpp"

class fast_limo::Match {
 public:
  float dist;                 // signed point-to-plane distance (Match.cpp:23-28)
  Eigen::Vector4f n_ABCD;     // plane.get_normal()
  bool good_fit;              // plane.good_fit()
  Match() : dist(0.f), good_fit(false) {}
  bool lisanAlGaib() const { return good_fit; }
  Eigen::Vector4f get_4Dglobal() const { return Eigen::Vector4f(p_global(0), p_global(1), p_global(2), 1.0f); }
  Eigen::Vector3f get_global_point() const { return p_global; }
  Eigen::Vector3f get_local_point() const { return p_local; }
  Eigen::Vector3f p_global, p_local;
};
#endif
