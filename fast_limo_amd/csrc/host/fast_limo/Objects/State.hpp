// fast_limo/Objects/State.hpp -- float32 mirror of the filter state (reference Objects/State.hpp:22-66).
#ifndef __FASTLIMO_STATE_HPP__
#define __FASTLIMO_STATE_HPP__
#include "fast_limo/Common.hpp"

namespace flimo_host { struct StateIkfom; }

class fast_limo::State {
 public:
  struct IMUbias { Eigen::Vector3f gyro, accel; };
  Eigen::Vector3f p;
  Eigen::Quaternionf q;
  Eigen::Vector3f v, g, w, a;
  Eigen::Quaternionf qLI;
  Eigen::Vector3f pLI;
  double time;
  IMUbias b;

  State();
  explicit State(const flimo_host::StateIkfom& s);                       // State.cpp:38-55
  State(const flimo_host::StateIkfom& s, double t);
  State(const flimo_host::StateIkfom& s, double t, Eigen::Vector3f a, Eigen::Vector3f w);
  explicit State(Eigen::Matrix4f& T);                                      // State.cpp:68-74
  void operator+=(const State& s);                                         // State.cpp:121-134
  // State::update(t) (State.cpp:76-119): host form for callers; the deskew kernel runs the same steps per point on the GPU
  void update(double t);

  Eigen::Matrix4f get_RT() const;            // State.cpp:136-143
  Eigen::Matrix4f get_RT_inv() const;        // :145-153
  Eigen::Matrix4f get_extr_RT() const;       // :155-162
  Eigen::Matrix4f get_extr_RT_inv() const;   // :164-172
};
#endif
