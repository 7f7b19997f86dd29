// fast_limo_amd/csrc/hip/flimo_ieskf.h  -- gfx950 device code.
//
// Small-matrix / manifold helpers of the device filter (flimo_ieskf.hip): the SAME arithmetic as the host filter
// (csrc/host/flimo_ikfom.cpp, which states the reference's MTK operations: SO(3) exp / log mtk/types/SOn.hpp:284-297, A_matrix
// and cos_sinc_sqrt mtk/src/mtkmath.hpp:143-174,236-247, the S2<double,98090,10000,1> charts mtk/types/S2.hpp:129-281),
// element by element in the same order, so the two agree to the last bits of sin / cos / atan / atan2 (device libm vs glibc;
// sqrt and division are correctly rounded on both sides).
#pragma once
#include <hip/hip_runtime.h>
#include "flimo_types.h"
#include "flimo_pose.h"
#include "flimo_chain.h"

#pragma clang fp contract(off)

namespace flimo {

constexpr int IK_N = 23;                 // degrees of freedom
constexpr double IK_TOL = 1e-11;         // MTK::tolerance<double>()
constexpr double IK_S2L = 98090.0 / 10000.0;

// shared-memory layout (doubles)
constexpr int IKL_P = 0;                 // P_ 23 x 23
constexpr int IKL_L = IKL_P + 529;       // L 23 x 23 (last pass only)
constexpr int IKL_PR = IKL_L + 529;      // P_[:, 0:12] / R   23 x 12
constexpr int IKL_W = IKL_PR + 276;      // 23 x 12
constexpr int IKL_KX = IKL_W + 276;      // K_x[:, 0:12]      23 x 12
constexpr int IKL_HTH = IKL_KX + 276;    // 12 x 12
constexpr int IKL_T = IKL_HTH + 144;     // 12 x 12 (LU in place)
constexpr int IKL_X = IKL_T + 144;       // 12 x 12 inverse
constexpr int IKL_HTh = IKL_X + 144;     // 12
constexpr int IKL_DX = IKL_HTh + 12;     // dx       23
constexpr int IKL_DXN = IKL_DX + 23;     // dx_new   23
constexpr int IKL_KH = IKL_DXN + 23;     // K_h      23
constexpr int IKL_DXU = IKL_KH + 23;     // dx_      23
constexpr int IKL_J = IKL_DXU + 23;      // J blocks: [0..8] rot, [9..17] offset_R, [18..21] grav (2 x 2)
constexpr int IKL_XN = IKL_J + 22;       // x after boxplus 26
constexpr int IKL_MISC = IKL_XN + 26;    // [0] pivot row, [1] flags
constexpr int IESKF_LDS_DOUBLES = IKL_MISC + 8;

// ---- small dense helpers, written like the host's Mat<> operators (sum from 0.0, k ascending) ------------------------------
struct Q4 { double x, y, z, w; };
__device__ inline Q4 ik_qmul(const Q4& a, const Q4& b) {
  Q4 r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
__device__ inline void ik_q2r(const Q4& q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__device__ inline void ik_hat(const double v[3], double H[9]) {
  H[0] = 0.0; H[1] = -v[2]; H[2] = v[1];
  H[3] = v[2]; H[4] = 0.0; H[5] = -v[0];
  H[6] = -v[1]; H[7] = v[0]; H[8] = 0.0;
}
__device__ inline void ik_mm33(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}
__device__ inline void ik_mv3(const double A[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; i++) {
    double s = 0.0;
    for (int k = 0; k < 3; k++) s += A[i * 3 + k] * v[k];
    o[i] = s;
  }
}
// MTK A_matrix (mtkmath.hpp:236-247), transposed result: J = A(v)^T
__device__ inline void ik_A_T(const double v[3], double JT[9]) {
  const double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const double norm = sqrt(sq);
  double A[9];
  if (norm < IK_TOL) {
    for (int i = 0; i < 9; i++) A[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    double H[9], HH[9];
    ik_hat(v, H);
    ik_mm33(H, H, HH);
    const double c1 = (1 - cos(norm)) / sq, c2 = (1 - sin(norm) / norm) / sq;
    for (int i = 0; i < 9; i++) A[i] = (((i % 4 == 0) ? 1.0 : 0.0) + c1 * H[i]) + c2 * HH[i];
  }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) JT[i * 3 + j] = A[j * 3 + i];
}
// MTK::cos_sinc_sqrt (mtkmath.hpp:143-174)
__device__ inline void ik_cos_sinc_sqrt(double x2, double& c, double& s) {
  const double b0 = 2.220446049250313e-16;
  const double b2 = sqrt(b0);
  const double bn = sqrt(b2);
  if (x2 >= bn) {
    const double x = sqrt(x2);
    c = cos(x);
    s = sin(x) / x;
    return;
  }
  const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
  double cosi = 1., sinc = 1;
  double term = -1 / 2. * x2;
  for (int i = 0; i < 3; ++i) {
    cosi += term;
    term *= inv[2 * i];
    sinc += term;
    term *= -inv[2 * i + 1] * x2;
  }
  c = cosi;
  s = sinc;
}
__device__ inline Q4 ik_exp_quat(const double v[3], double scale) {
  const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  double c, s;
  ik_cos_sinc_sqrt(scale * scale * n2, c, s);
  const double mult = s * scale;
  Q4 q;
  q.w = c; q.x = mult * v[0]; q.y = mult * v[1]; q.z = mult * v[2];
  return q;
}
__device__ inline void ik_so3_log(const Q4& q, double o[3]) {
  double nv = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  if (nv < IK_TOL) nv = IK_TOL;
  const double s = 2.0 / nv * atan(nv / q.w);
  o[0] = s * q.x; o[1] = s * q.y; o[2] = s * q.z;
}
// S2<double,98090,10000,1> charts (mtk/types/S2.hpp:179-281)
__device__ inline void ik_s2_Bx(const double v[3], double B[6] /*3 x 2*/) {
  const double L = IK_S2L;
  for (int i = 0; i < 6; i++) B[i] = 0.0;
  if (v[0] + L > IK_TOL) {
    B[0] = -v[1];                       B[1] = -v[2];
    B[2] = L - v[1] * v[1] / (L + v[0]); B[3] = -v[2] * v[1] / (L + v[0]);
    B[4] = -v[2] * v[1] / (L + v[0]);    B[5] = L - v[2] * v[2] / (L + v[0]);
    for (int i = 0; i < 6; i++) B[i] /= L;
  } else {
    B[3] = -1; B[4] = 1;
  }
}
__device__ inline void ik_s2_boxminus(const double a[3] /*this*/, const double o[3] /*other*/, double out[2]) {
  double Ha[9], hv[3];
  ik_hat(a, Ha);
  ik_mv3(Ha, o, hv);
  const double v_sin = sqrt(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2]);
  const double v_cos = a[0] * o[0] + a[1] * o[1] + a[2] * o[2];
  const double theta = atan2(v_sin, v_cos);
  if (v_sin < IK_TOL) {
    if (fabs(theta) > IK_TOL) { out[0] = 3.1415926; out[1] = 0; }
    else { out[0] = 0; out[1] = 0; }
    return;
  }
  double B[6], Ho[9], t[3];
  ik_s2_Bx(o, B);
  ik_hat(o, Ho);
  ik_mv3(Ho, a, t);
  const double f = theta / v_sin;
  for (int j = 0; j < 2; j++) out[j] = f * (B[0 + j] * t[0] + B[2 + j] * t[1] + B[4 + j] * t[2]);
}
// Nx_yy = (1 / L / L) * (Bx^T * hat(vec))   2 x 3
__device__ inline void ik_s2_Nx(const double v[3], double N[6]) {
  double B[6], H[9];
  ik_s2_Bx(v, B);
  ik_hat(v, H);
  const double f = 1 / IK_S2L / IK_S2L;
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += B[k * 2 + i] * H[k * 3 + j];
      N[i * 3 + j] = f * s;
    }
}
// Mx(delta) (S2.hpp:259-281) with the reference's exp(Bu, scalar(1/2)) == identity rotation (integer division); 3 x 2
__device__ inline void ik_s2_Mx(const double v[3], const double delta[2], double M[6]) {
  double B[6], H[9];
  ik_s2_Bx(v, B);
  ik_hat(v, H);
  const double dn = sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
  double HA[9];
  if (dn < IK_TOL) {
    for (int i = 0; i < 9; i++) HA[i] = H[i];
  } else {
    double Bu[3], AT[9];
    for (int i = 0; i < 3; i++) Bu[i] = B[i * 2 + 0] * delta[0] + B[i * 2 + 1] * delta[1];
    ik_A_T(Bu, AT);                     // A_matrix(Bu).T()
    // E = quat_to_rot(exp(Bu, 0)) is the identity exactly: (E * hat) == hat bit for bit
    ik_mm33(H, AT, HA);
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 2; j++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += HA[i * 3 + k] * B[k * 2 + j];
      M[i * 2 + j] = -1.0 * s;
    }
}
// the 2 x 2 S2 block of the re-projection: Nx_yy(now) * Mx(prop, delta)
__device__ inline void ik_s2_J(const double now[3], const double prop[3], const double delta[2], double J[4]) {
  double N[6], M[6];
  ik_s2_Nx(now, N);
  ik_s2_Mx(prop, delta, M);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += N[i * 3 + k] * M[k * 2 + j];
      J[i * 2 + j] = s;
    }
}

// ---- 12 x 12 inverse on ONE wave, in registers: Gauss-Jordan with partial pivoting, rows marked instead of exchanged -- the steps of
//      the host's inverse_gj (csrc/host/flimo_ikfom.cpp), element for element.  Lane l holds row l / 4, columns 3 (l % 4) .. + 2 of
//      the matrix and of the accumulated right-hand side (six doubles); a step is: quad broadcast of the row's entry in column k, the
//      maximum over the rows (two DPP rotations inside a row of 16 lanes, three v_readlane across), the pivot row's six values by
//      ds_bpermute, one division, six multiply-subtracts.  T, X: row-major 12 x 12 in shared memory; all 64 lanes of the wave call it.
//      Returns false (wave-uniform) when a pivot is zero.
template <int CTRL>
__device__ __forceinline__ double ik_dpp(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int l2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  const int h2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(h2, l2);
}
__device__ __forceinline__ double ik_readlane(double x, int lane_uniform) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  return __hiloint2double(__builtin_amdgcn_readlane(hi, lane_uniform), __builtin_amdgcn_readlane(lo, lane_uniform));
}
__device__ __forceinline__ double ik_shfl(double x, int src_lane) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int l2 = __builtin_amdgcn_ds_bpermute(src_lane << 2, lo), h2 = __builtin_amdgcn_ds_bpermute(src_lane << 2, hi);
  return __hiloint2double(h2, l2);
}
__device__ inline bool ik_gj12_wave(const double* __restrict__ T, double* __restrict__ X, int lane) {
  const int r = lane >> 2, cg = lane & 3;
  const bool live_row = r < 12;
  double a[3], x[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    a[j] = live_row ? T[r * 12 + 3 * cg + j] : 0.0;
    x[j] = (live_row && r == 3 * cg + j) ? 1.0 : 0.0;
  }
  bool used = !live_row;
  int my_k = 0;                       // the column this lane's row became the pivot of
  double my_d = 1.0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 12; k++) {
    constexpr int dummy = 0; (void)dummy;
    const int kc = k / 3, kr = k % 3;
    const double ak = a[kr];
    // the row's entry in column k, in all four lanes of the row (a quad): quad_perm [kc, kc, kc, kc]
    double ark;
    switch (kc) {
      case 0: ark = ik_dpp<0x00>(ak); break;
      case 1: ark = ik_dpp<0x55>(ak); break;
      case 2: ark = ik_dpp<0xAA>(ak); break;
      default: ark = ik_dpp<0xFF>(ak); break;
    }
    const double mag = used ? -1.0 : fabs(ark);
    double m = fmax(mag, ik_dpp<0x124>(mag));          // row_ror:4
    m = fmax(m, ik_dpp<0x128>(m));                     // row_ror:8
    const double mx = fmax(fmax(ik_readlane(m, 0), ik_readlane(m, 16)), ik_readlane(m, 32));
    if (!(mx > 0.0)) { ok = false; break; }            // wave-uniform
    const unsigned long long cand = __ballot(cg == 0 && !used && mag == mx);
    const int p_lane = __ffsll((long long)cand) - 1;   // lowest row among equals
    const int p = p_lane >> 2;
    const double d = ik_readlane(ark, p_lane);
    const int src = (p << 2) | cg;
    double ap[3], xp[3];
#pragma unroll
    for (int j = 0; j < 3; j++) { ap[j] = ik_shfl(a[j], src); xp[j] = ik_shfl(x[j], src); }
    const double f = ark / d;
    if (r != p) {
#pragma unroll
      for (int j = 0; j < 3; j++) { a[j] = a[j] - f * ap[j]; x[j] = x[j] - f * xp[j]; }
    } else {
      used = true; my_k = k; my_d = d;
    }
  }
  if (!ok) return false;
  if (live_row) {
#pragma unroll
    for (int j = 0; j < 3; j++) X[my_k * 12 + 3 * cg + j] = x[j] / my_d;
  }
  return true;
}

}  // namespace flimo
