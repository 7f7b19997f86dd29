// fast_limo_amd/csrc/hip/flimo_pose.h
// The float32 pose constants of a measurement pass from the filter's float64 state, exactly as the reference forms them per
// pass: State(state_ikfom) casts (Objects/State.cpp:38-55), State::get_RT / get_RT_inv / get_extr_RT_inv (:136-172) and the two
// conjugate rotations of calculate_H (Modules/Localizer.cpp:554-555).  Host AND device (the device filter computes the next
// pass's constants itself): compiled without FMA contraction, Eigen's evaluation order (3-term reductions c0 + (c1 + c2)).
#pragma once
#include <hip/hip_runtime.h>
#include "flimo_types.h"

#pragma clang fp contract(off)

namespace flimo {

__host__ __device__ inline float hsum3(float a, float b, float c) { return a + (b + c); }
__host__ __device__ inline void quat_to_rot_f(const float q[4] /*x y z w*/, float R[9]) {   // Eigen toRotationMatrix
  const float tx = 2.f * q[0], ty = 2.f * q[1], tz = 2.f * q[2];
  const float twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const float txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const float tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1.f - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.f - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.f - (txx + tyy);
}
__host__ __device__ inline void quat_to_rot_d(const double q[4], double R[9]) {
  const double tx = 2.0 * q[0], ty = 2.0 * q[1], tz = 2.0 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
__host__ __device__ inline void se3_from(const float q[4], const float p[3], float T[16]) {      // State::get_RT / get_extr_RT
  float R[9];
  quat_to_rot_f(q, R);
  T[0] = R[0]; T[1] = R[1]; T[2] = R[2];  T[3] = p[0];
  T[4] = R[3]; T[5] = R[4]; T[6] = R[5];  T[7] = p[1];
  T[8] = R[6]; T[9] = R[7]; T[10] = R[8]; T[11] = p[2];
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f;  T[15] = 1.f;
}
__host__ __device__ inline void se3_inv_from(const float q[4], const float p[3], float T[16]) {  // State::get_RT_inv / get_extr_RT_inv
  float R[9];
  quat_to_rot_f(q, R);
  // rot^T and -rot^T * p, coefficient products reduced as c0 + (c1 + c2)
  const float Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
  float t[3];
  for (int i = 0; i < 3; i++) t[i] = hsum3((-Rt[i * 3 + 0]) * p[0], (-Rt[i * 3 + 1]) * p[1], (-Rt[i * 3 + 2]) * p[2]);
  T[0] = Rt[0]; T[1] = Rt[1]; T[2] = Rt[2];  T[3] = t[0];
  T[4] = Rt[3]; T[5] = Rt[4]; T[6] = Rt[5];  T[7] = t[1];
  T[8] = Rt[6]; T[9] = Rt[7]; T[10] = Rt[8]; T[11] = t[2];
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f;     T[15] = 1.f;
}
__host__ __device__ inline void pose_from_x26(const double x[26], PoseMats& P) {
  // State(state_ikfom): casts (State.cpp:38-55)
  const float p[3] = {(float)x[0], (float)x[1], (float)x[2]};
  const float q[4] = {(float)x[3], (float)x[4], (float)x[5], (float)x[6]};
  const float qLI[4] = {(float)x[7], (float)x[8], (float)x[9], (float)x[10]};
  const float pLI[3] = {(float)x[11], (float)x[12], (float)x[13]};
  se3_from(q, p, P.RT);
  se3_inv_from(q, p, P.RT_inv);
  se3_inv_from(qLI, pLI, P.TLI_inv);
  // s.rot.conjugate().toRotationMatrix().cast<float>()  (Localizer.cpp:554-555)
  const double qc[4] = {-x[3], -x[4], -x[5], x[6]};
  const double lc[4] = {-x[7], -x[8], -x[9], x[10]};
  double Rd[9], Ld[9];
  quat_to_rot_d(qc, Rd);
  quat_to_rot_d(lc, Ld);
  for (int i = 0; i < 9; i++) { P.R_inv[i] = (float)Rd[i]; P.RLI_inv[i] = (float)Ld[i]; }
}

}  // namespace flimo
