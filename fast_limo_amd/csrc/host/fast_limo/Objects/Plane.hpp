// fast_limo/Objects/Plane.hpp -- the reference's Plane object (Objects/Plane.hpp:24-61, Objects/Plane.cpp:23-120) for
// callers that build planes themselves.  On the registration path the planes are fitted inside the GPU fit kernel;
// this class evaluates the SAME routines (flimo_math.h, compiled for the host into libflimo_hip.so) so that both give
// identical bits.  Only NUM_MATCH_POINTS == 5 is supported, like the GPU path.
#ifndef __FASTLIMO_PLANE_HPP__
#define __FASTLIMO_PLANE_HPP__
#include <cmath>
#include <vector>
#include "fast_limo/Common.hpp"
#include "fast_limo/Utils/Config.hpp"
#include "flimo_c.h"
#include "flimo_dev.h"      // flimo_plane_fit5_host / flimo_plane_eval5_host: the fit kernel's plane routines compiled for the host

class fast_limo::Plane {
 public:
  Plane(const MapPoints& p, const std::vector<float>& d, Config::iKFoM::Mapping* config_ptr)   // Plane.cpp:23-31
      : is_plane(false), cfg_ptr(config_ptr) {
    if (!enough_points(p)) return;
    if (!close_enough(d)) return;
    fit_plane(p);
  }
  Plane() : is_plane(false), cfg_ptr(nullptr) {}
  // MI355X addition: wrap a plane the GPU fit kernel already produced
  Plane(const Eigen::Vector4f& n, bool good, Config::iKFoM::Mapping* config_ptr) : n_ABCD(n), is_plane(good), cfg_ptr(config_ptr) {}

  Eigen::Vector4f get_normal() const { return n_ABCD; }
  bool good_fit() const { return is_plane; }
  float dist2plane(const Eigen::Vector3f& p) const { return n_ABCD(0) * p(0) + n_ABCD(1) * p(1) + n_ABCD(2) * p(2) + n_ABCD(3); }
  float dist2plane(const PointType& p) const { return n_ABCD(0) * p.x + n_ABCD(1) * p.y + n_ABCD(2) * p.z + n_ABCD(3); }
  bool on_plane(const Eigen::Vector3f& p) const { return is_plane && std::fabs(dist2plane(p)) < cfg_ptr->PLANE_THRESHOLD; }
  bool on_plane(const PointType& p) const { return is_plane && std::fabs(dist2plane(p)) < cfg_ptr->PLANE_THRESHOLD; }
  bool enough_points(const MapPoints& p) { return is_plane = (int)p.size() >= cfg_ptr->NUM_MATCH_POINTS; }      // :41-43
  bool close_enough(const std::vector<float>& d) {                                                              // :45-48
    if (d.size() < 1) return is_plane = false;
    return is_plane = d.back() < cfg_ptr->MAX_DIST_PLANE;          // squared distance against metres, as in the reference
  }

 private:
  Eigen::Vector3f centroid;
  Eigen::Vector4f n_ABCD;
  bool is_plane;
  Config::iKFoM::Mapping* cfg_ptr;

  void fit_plane(const MapPoints& p) {                                                                         // :80-91
    if (p.size() != 5) { is_plane = false; return; }
    float xyz[15], n[4];
    for (int j = 0; j < 5; j++) { xyz[3 * j] = p[j].x; xyz[3 * j + 1] = p[j].y; xyz[3 * j + 2] = p[j].z; }
    flimo_plane_fit5_host(xyz, n);
    n_ABCD = Eigen::Vector4f(n[0], n[1], n[2], n[3]);
    is_plane = flimo_plane_eval5_host(n, xyz, (float)cfg_ptr->PLANE_THRESHOLD) != 0;
    if (is_plane) {
      centroid = Eigen::Vector3f(0.f, 0.f, 0.f);
      for (int j = 0; j < 5; j++) { centroid(0) += p[j].x; centroid(1) += p[j].y; centroid(2) += p[j].z; }
      for (int a = 0; a < 3; a++) centroid(a) /= 5.f;
    }
  }
};
#endif
