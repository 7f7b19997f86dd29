// fast_limo_amd/csrc/hip/flimo_ieskf.h  -- gfx950 device code (+ host evaluation of the same helpers).
//
// The 23-dof algebra of one outer iteration of esekf::update_iterated_dyn_share_modified (IKFoM_toolkit/esekfom/esekfom.hpp:
// 1620-1823) inside a pass's reducing launch (flimo_chain.h): workgroup-wide routines for
//   ik_extra_block   the measurement-independent half (:1652-1697) -- one extra workgroup of the launch, beside the pass;
//   ik_final_stage   from the pass's 91 sums to the next state (:1722-1764) -- the workgroup that completes the launch.
// It is the SAME arithmetic as the host filter (csrc/host/flimo_ikfom.cpp, which states the reference's algebra and MTK operations:
// SO(3) exp / log mtk/types/SOn.hpp:284-297, A_matrix and cos_sinc_sqrt mtk/src/mtkmath.hpp:143-174,236-247, the
// S2<double,98090,10000,1> charts mtk/types/S2.hpp:129-281; gain through the matrix-inversion-lemma form of :1722-1729 with a
// Gauss-Jordan 12 x 12 solve, step :1733 as PR (S (H^T h + H^T H dx_new)) - dx_new), element by element in the same order, so the
// two agree to the last bits of sin / cos / atan / atan2 (device libm vs glibc; sqrt and division are correctly rounded on both
// sides).  What the device does NOT do: the covariance update of the last iteration (:1764-1820), the M < 23 branch (dense H,
// :1701-1709), the eigen-decomposition of a degenerate H^T H (:1736-1744) and the repair of exactly tied distances: the chain
// stops there and the host filter goes on from the state -- and, where they are usable, the sums -- the device hands back.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "flimo_types.h"
#include "flimo_pose.h"
#include "flimo_chain.h"

#pragma clang fp contract(off)

namespace flimo {

constexpr int IK_N = 23;                 // degrees of freedom
constexpr double IK_TOL = 1e-11;         // MTK::tolerance<double>()
constexpr double IK_S2L = 98090.0 / 10000.0;
constexpr int IK_LIVE = 91, IK_LIVE_PAD = 96, IK_GROUPS = 8;     // = FIT_LIVE, FIT_LIVE_PAD, FIT_GROUPS (flimo_kernels.h)

// shared-memory layout (doubles)
constexpr int IKL_P = 0;                 // P_ 23 x 23 (extra workgroup only)
constexpr int IKL_PR = IKL_P + 529;      // P_[:, 0:12] / R   23 x 12 (extra workgroup only); rows 0..11 = A11
constexpr int IKL_AI = IKL_PR + 276;     // A11^-1            12 x 12
constexpr int IKL_G2 = IKL_AI + 144;     // PR[12:23] A11^-1  11 x 12
constexpr int IKL_HTH = IKL_G2 + 132;    // 12 x 12
constexpr int IKL_T = IKL_HTH + 144;     // 12 x 12
constexpr int IKL_HTh = IKL_T + 144;     // 12
constexpr int IKL_DX = IKL_HTh + 12;     // dx       23
constexpr int IKL_DXN = IKL_DX + 23;     // dx_new   23
constexpr int IKL_V = IKL_DXN + 23;      // v = H^T h + H^T H dx_new[0:12]   12
constexpr int IKL_U = IKL_V + 12;        // u = S v                           12
constexpr int IKL_DXU = IKL_U + 12;      // dx_      23
constexpr int IKL_J = IKL_DXU + 23;      // J blocks: [0..8] rot, [9..17] offset_R, [18..21] grav (2 x 2)
constexpr int IKL_XN = IKL_J + 22;       // x after boxplus 26
constexpr int IKL_XC = IKL_XN + 26;      // x       26
constexpr int IKL_XP = IKL_XC + 26;      // x_prop  26
constexpr int IKL_LIM = IKL_XP + 26;     // limit   23
constexpr int IKL_LIVE = IKL_LIM + 23;   // the pass's sums 96
constexpr int IKL_END = IKL_LIVE + IK_LIVE_PAD;
constexpr int IESKF_LDS_BYTES = IKL_END * 8 + 32 * 4;      // + int scratch s_i[32]

// sin and cos of one argument: one argument reduction on the device (ocml's sincos returns what its sin and cos return)
__host__ __device__ inline void ik_sincos(double x, double& sn, double& cs) {
#if defined(__HIP_DEVICE_COMPILE__)
  sincos(x, &sn, &cs);
#else
  sn = sin(x); cs = cos(x);
#endif
}
// ---- small dense helpers, written like the host's Mat<> operators (sum from 0.0, k ascending) ------------------------------
struct Q4 { double x, y, z, w; };
__host__ __device__ inline Q4 ik_qmul(const Q4& a, const Q4& b) {
  Q4 r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
__host__ __device__ inline void ik_q2r(const Q4& q, double R[9]) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
__host__ __device__ inline void ik_hat(const double v[3], double H[9]) {
  H[0] = 0.0; H[1] = -v[2]; H[2] = v[1];
  H[3] = v[2]; H[4] = 0.0; H[5] = -v[0];
  H[6] = -v[1]; H[7] = v[0]; H[8] = 0.0;
}
__host__ __device__ inline void ik_mm33(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0.0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}
__host__ __device__ inline void ik_mv3(const double A[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; i++) {
    double s = 0.0;
    for (int k = 0; k < 3; k++) s += A[i * 3 + k] * v[k];
    o[i] = s;
  }
}
// MTK A_matrix (mtkmath.hpp:236-247), transposed result: J = A(v)^T
__host__ __device__ inline void ik_A_T(const double v[3], double JT[9]) {
  const double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  const double norm = sqrt(sq);
  double A[9];
  if (norm < IK_TOL) {
    for (int i = 0; i < 9; i++) A[i] = (i % 4 == 0) ? 1.0 : 0.0;
  } else {
    double H[9], HH[9];
    ik_hat(v, H);
    ik_mm33(H, H, HH);
    double sn, cs;
    ik_sincos(norm, sn, cs);
    const double c1 = (1 - cs) / sq, c2 = (1 - sn / norm) / sq;
    for (int i = 0; i < 9; i++) A[i] = (((i % 4 == 0) ? 1.0 : 0.0) + c1 * H[i]) + c2 * HH[i];
  }
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) JT[i * 3 + j] = A[j * 3 + i];
}
// MTK::cos_sinc_sqrt (mtkmath.hpp:143-174)
__host__ __device__ inline void ik_cos_sinc_sqrt(double x2, double& c, double& s) {
  const double b0 = 2.220446049250313e-16;
  const double b2 = sqrt(b0);
  const double bn = sqrt(b2);
  if (x2 >= bn) {
    const double x = sqrt(x2);
    double sn, cs;
    ik_sincos(x, sn, cs);
    c = cs;
    s = sn / x;
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
__host__ __device__ inline Q4 ik_exp_quat(const double v[3], double scale) {
  const double n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  double c, s;
  ik_cos_sinc_sqrt(scale * scale * n2, c, s);
  const double mult = s * scale;
  Q4 q;
  q.w = c; q.x = mult * v[0]; q.y = mult * v[1]; q.z = mult * v[2];
  return q;
}
__host__ __device__ inline void ik_so3_log(const Q4& q, double o[3]) {
  double nv = sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  if (nv < IK_TOL) nv = IK_TOL;
  const double s = 2.0 / nv * atan(nv / q.w);
  o[0] = s * q.x; o[1] = s * q.y; o[2] = s * q.z;
}
// S2<double,98090,10000,1> charts (mtk/types/S2.hpp:179-281)
__host__ __device__ inline void ik_s2_Bx(const double v[3], double B[6] /*3 x 2*/) {
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
__host__ __device__ inline void ik_s2_boxminus(const double a[3] /*this*/, const double o[3] /*other*/, double out[2]) {
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
__host__ __device__ inline void ik_s2_Nx(const double v[3], double N[6]) {
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
__host__ __device__ inline void ik_s2_Mx(const double v[3], const double delta[2], double M[6]) {
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
__host__ __device__ inline void ik_s2_J(const double now[3], const double prop[3], const double delta[2], double J[4]) {
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
  // (full row / bank masks + bound_ctrl: every lane is written, no "old" value to keep -- no copy in front of the DPP move)
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int l2 = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  const int h2 = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(h2, l2);
}
__device__ __forceinline__ double ik_readlane(double x, int lane_uniform) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  return __hiloint2double(__builtin_amdgcn_readlane(hi, lane_uniform), __builtin_amdgcn_readlane(lo, lane_uniform));
}
__device__ __forceinline__ double ik_shfl(double x, int src_lane4) {      // src_lane4 = 4 * source lane
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const int l2 = __builtin_amdgcn_ds_bpermute(src_lane4, lo), h2 = __builtin_amdgcn_ds_bpermute(src_lane4, hi);
  return __hiloint2double(h2, l2);
}
__device__ __forceinline__ double ik_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }   // (no NaNs here: no canonicalisation)
// Branch-free: a zero pivot only clears the return value (the steps run on, on garbage nobody reads).
// T u = v is solved directly: the right-hand side is one more column (its entry of row r rides in all four lanes of the row), the
// inverse itself is never formed.  u[k] = b[row that became the pivot of column k] * (1 / pivot).
__device__ inline bool ik_gj12_solve_wave(const double* __restrict__ T, const double* __restrict__ v, double* __restrict__ u, int lane) {
  const int r = lane >> 2, cg = lane & 3;
  const bool live_row = r < 12;
  double a[3];
#pragma unroll
  for (int j = 0; j < 3; j++) a[j] = live_row ? T[r * 12 + 3 * cg + j] : 0.0;
  double b = live_row ? v[r] : 0.0;
  bool used = !live_row;
  int my_k = 0;                       // the column this lane's row became the pivot of
  double my_d = 1.0;                  // ... and the reciprocal of its pivot
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 12; k++) {
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
    // every row's reciprocal, beside the search for the pivot (the two chains are independent: the division's latency hides behind
    // the reduction's); the pivot row's is the one that is used.  (The empty asm keeps the compiler from moving the division
    // behind the v_readlane, onto the critical path.)
    double rown = 1.0 / ark;
    asm volatile("" : "+v"(rown));
    double m = ik_max(mag, ik_dpp<0x124>(mag));        // row_ror:4
    m = ik_max(m, ik_dpp<0x128>(m));                   // row_ror:8
    const double mx = ik_max(ik_max(ik_readlane(m, 0), ik_readlane(m, 16)), ik_readlane(m, 32));
    ok = ok && (mx > 0.0);                             // wave-uniform
    const unsigned long long cand = __ballot(cg == 0 && !used && mag == mx);
    const int p_lane = cand ? __ffsll((long long)cand) - 1 : 0;    // lowest row among equals
    const double rinv = ik_readlane(rown, p_lane);     // 1 / A[p][k]
    const double bp = ik_readlane(b, p_lane);          // the pivot row's right-hand side
    const int src4 = (p_lane + cg) << 2;               // byte address of lane 4 p + cg for ds_bpermute
    double ap[3];
#pragma unroll
    for (int j = 0; j < 3; j++) ap[j] = ik_shfl(a[j], src4);
    const bool is_p = (lane >> 2) == (p_lane >> 2);
    const double f = is_p ? 0.0 : ark * rinv;          // (the pivot row itself: row - 0 * row)
#pragma unroll
    for (int j = 0; j < 3; j++) a[j] = a[j] - f * ap[j];
    b = b - f * bp;
    if (is_p) { used = true; my_k = k; my_d = rinv; }
  }
  if (live_row && cg == 0) u[my_k] = b * my_d;
  return ok;
}

// The inverse itself (the measurement-independent half needs A11^-1): the same elimination on [T | I], the right-hand side's three
// columns of this lane's column group riding along.  X row-major 12 x 12.
__device__ inline bool ik_gj12_inverse_wave(const double* __restrict__ T, double* __restrict__ X, int lane) {
  const int r = lane >> 2, cg = lane & 3;
  const bool live_row = r < 12;
  double a[3], x[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    a[j] = live_row ? T[r * 12 + 3 * cg + j] : 0.0;
    x[j] = (live_row && r == 3 * cg + j) ? 1.0 : 0.0;
  }
  bool used = !live_row;
  int my_k = 0;
  double my_d = 1.0;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 12; k++) {
    const int kc = k / 3, kr = k % 3;
    const double ak = a[kr];
    double ark;
    switch (kc) {
      case 0: ark = ik_dpp<0x00>(ak); break;
      case 1: ark = ik_dpp<0x55>(ak); break;
      case 2: ark = ik_dpp<0xAA>(ak); break;
      default: ark = ik_dpp<0xFF>(ak); break;
    }
    const double mag = used ? -1.0 : fabs(ark);
    double rown = 1.0 / ark;
    asm volatile("" : "+v"(rown));
    double m = ik_max(mag, ik_dpp<0x124>(mag));
    m = ik_max(m, ik_dpp<0x128>(m));
    const double mx = ik_max(ik_max(ik_readlane(m, 0), ik_readlane(m, 16)), ik_readlane(m, 32));
    ok = ok && (mx > 0.0);
    const unsigned long long cand = __ballot(cg == 0 && !used && mag == mx);
    const int p_lane = cand ? __ffsll((long long)cand) - 1 : 0;
    const double rinv = ik_readlane(rown, p_lane);
    const int src4 = (p_lane + cg) << 2;
    double ap[3], xp[3];
#pragma unroll
    for (int j = 0; j < 3; j++) { ap[j] = ik_shfl(a[j], src4); xp[j] = ik_shfl(x[j], src4); }
    const bool is_p = (lane >> 2) == (p_lane >> 2);
    const double f = is_p ? 0.0 : ark * rinv;
#pragma unroll
    for (int j = 0; j < 3; j++) { a[j] = a[j] - f * ap[j]; x[j] = x[j] - f * xp[j]; }
    if (is_p) { used = true; my_k = k; my_d = rinv; }
  }
  if (live_row) {
#pragma unroll
    for (int j = 0; j < 3; j++) X[my_k * 12 + 3 * cg + j] = x[j] * my_d;
  }
  return ok;
}
// ... and serially for the host (flimo_update_chain forms iteration -1's half itself): the steps of csrc/host/flimo_ikfom.cpp: inverse_gj
__host__ inline bool ik_inverse_gj12_serial(const double* Ain, double* Ainv) {
  const int n = 12;
  double A[144], X[144], dk[12];
  int prow[12];
  bool used[12];
  for (int i = 0; i < 144; i++) { A[i] = Ain[i]; X[i] = 0.0; }
  for (int i = 0; i < n; i++) { X[i * n + i] = 1.0; used[i] = false; }
  for (int k = 0; k < n; k++) {
    int p = -1;
    double best = -1.0;
    for (int r = 0; r < n; r++) {
      if (used[r]) continue;
      const double m = fabs(A[r * n + k]);
      if (m > best) { best = m; p = r; }
    }
    if (p < 0 || !(best > 0.0)) return false;
    const double rinv = 1.0 / A[p * n + k];
    for (int r = 0; r < n; r++) {
      if (r == p) continue;
      const double f = A[r * n + k] * rinv;
      for (int j = k + 1; j < n; j++) A[r * n + j] = A[r * n + j] - f * A[p * n + j];
      for (int c = 0; c < n; c++) X[r * n + c] = X[r * n + c] - f * X[p * n + c];
    }
    used[p] = true; prow[k] = p; dk[k] = rinv;
  }
  for (int k = 0; k < n; k++)
    for (int c = 0; c < n; c++) Ainv[k * n + c] = X[prow[k] * n + c] * dk[k];
  return true;
}

// ---- data handed between workgroups of ONE launch: written through / read past the XCD's L2 (agent scope, relaxed) -------------
__device__ __forceinline__ void ik_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ik_sti(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ik_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ik_ldi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 16-byte store of {value, tag} written through to (host) memory.  (s_nop 1: a VMEM store of more than 64 bits must be followed by
// two wait states before a VALU instruction may overwrite its data registers on gfx940+ -- the compiler's hazard recognizer inserts
// them for its own stores and cannot see into inline assembly.  Without them the low dword of a stored double was, now and then,
// the NEXT value's: 1e-6 relative, run-to-run different.)
typedef double ik_v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ik_put(double2* base, int slot, double value, unsigned long long tag) {
  ik_v2d_t g;
  g.x = value;
  g.y = __longlong_as_double((long long)tag);
  double2* o = base + slot;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(o), "v"(g) : "memory");
}

// Developer-only phase stamps (tools/ieskf_bench.hip builds with -DIESKF_STAMPS): thread 0 stores the 100 MHz wall clock
#ifdef IESKF_STAMPS
__device__ unsigned long long g_ik_stamps[32];
#define IK_STAMP(k) do { if (threadIdx.x == 0) g_ik_stamps[k] = wall_clock64(); } while (0)
#else
#define IK_STAMP(k) do {} while (0)
#endif

// (a) rows idx .. idx+B of M(.., 0:ncols) <- J * rows;  (b) cols idx .. idx+B of M <- cols * J^T    (host left_block / right_block_T)
template <int B>
__device__ __forceinline__ void ik_left_block(double* M, int ld, int idx, const double* J, int ncols, int tid) {
  if (tid >= 0 && tid < ncols) {
    const int c = tid;
    double t[B];
#pragma unroll
    for (int r = 0; r < B; r++) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < B; k++) s += J[r * B + k] * M[(idx + k) * ld + c];
      t[r] = s;
    }
#pragma unroll
    for (int r = 0; r < B; r++) M[(idx + r) * ld + c] = t[r];
  }
}
template <int B>
__device__ __forceinline__ void ik_right_block_T(double* M, int ld, int idx, const double* J, int nrows, int tid) {
  if (tid >= 0 && tid < nrows) {
    const int r = tid;
    double t[B];
#pragma unroll
    for (int c = 0; c < B; c++) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < B; k++) s += M[r * ld + idx + k] * J[c * B + k];
      t[c] = s;
    }
#pragma unroll
    for (int c = 0; c < B; c++) M[r * ld + idx + c] = t[c];
  }
}

// ---- the measurement-independent half of an iteration (esekfom.hpp:1652-1697): x boxminus x_prop, the SO(3) / S2 blocks, P_prop
//      through them, PR = P_[:, 0:12] / R.  In: xc, xp, P_ (= P_prop) in shared memory.  Out (shared memory): dx, dxn, Jb, P_, PR.
//      Workgroup-wide (256 threads); ends with a barrier. ----
__device__ __forceinline__ void ik_pre_block(double* lds, double R, int tid) {
  const int n = IK_N;
  double* P_ = lds + IKL_P;
  double* PR = lds + IKL_PR;
  double* dx = lds + IKL_DX;
  double* dxn = lds + IKL_DXN;
  double* Jb = lds + IKL_J;
  const double* xc = lds + IKL_XC;
  const double* xp = lds + IKL_XP;
  // independent chains: rot and offset_R_L_I run the same code in two lanes of one wave; the S2 chain on another wave
  if (tid < 2) {
    const int o = (tid == 0) ? 3 : 7;                        // rot / offset_R_L_I in the flat state
    const int idx = (tid == 0) ? 3 : 6;
    const Q4 a{xc[o], xc[o + 1], xc[o + 2], xc[o + 3]};
    const Q4 b{-xp[o], -xp[o + 1], -xp[o + 2], xp[o + 3]};   // conj(other)
    double r[3];
    ik_so3_log(ik_qmul(b, a), r);
    for (int i = 0; i < 3; i++) dx[idx + i] = r[i];
    ik_A_T(r, Jb + 9 * tid);
  } else if (tid == 64) {
    double d[2];
    ik_s2_boxminus(xc + 23, xp + 23, d);
    dx[21] = d[0]; dx[22] = d[1];
    ik_s2_J(xc + 23, xp + 23, d, Jb + 18);
  } else if (tid >= 128 && tid < 128 + 15) {
    const int e = tid - 128, seg = e / 3, i = e % 3;         // pos, offset_T_L_I, vel, bg, ba
    const int xo = seg == 0 ? 0 : 8 + 3 * seg, dxo = seg == 0 ? 0 : 6 + 3 * seg;
    dx[dxo + i] = xc[xo + i] - xp[xo + i];
  }
  __syncthreads();
  // dx_new = dx with the three manifold segments through their blocks; P_ <- J P_ J^T block by block, in the reference's order
  if (tid >= 64 && tid < 64 + n) {
    const int i = tid - 64;
    double v = dx[i];
    if (i >= 3 && i < 9) {
      const int idx = i < 6 ? 3 : 6;
      const double* J = Jb + (i < 6 ? 0 : 9);
      const int r = i - idx;
      v = J[r * 3 + 0] * dx[idx] + J[r * 3 + 1] * dx[idx + 1] + J[r * 3 + 2] * dx[idx + 2];
    } else if (i >= 21) {
      const double* J = Jb + 18;
      const int r = i - 21;
      v = J[r * 2 + 0] * dx[21] + J[r * 2 + 1] * dx[22];
    }
    dxn[i] = v;
  }
  ik_left_block<3>(P_, n, 3, Jb, n, tid);
  __syncthreads();
  ik_right_block_T<3>(P_, n, 3, Jb, n, tid);
  __syncthreads();
  ik_left_block<3>(P_, n, 6, Jb + 9, n, tid);
  __syncthreads();
  ik_right_block_T<3>(P_, n, 6, Jb + 9, n, tid);
  __syncthreads();
  ik_left_block<2>(P_, n, 21, Jb + 18, n, tid);
  __syncthreads();
  ik_right_block_T<2>(P_, n, 21, Jb + 18, n, tid);
  __syncthreads();
  for (int e = tid; e < n * 12; e += 256) PR[e] = P_[(e / 12) * n + (e % 12)] / R;
  __syncthreads();
  // The gain of :1722-1729 is taken through the block-inverse identity (see ik_final_stage): A11^-1 and G2 = A21 A11^-1 of
  // A = P_ / R do not depend on the measurement
  double* AI = lds + IKL_AI;
  double* G2 = lds + IKL_G2;
  if (tid < 64) (void)ik_gj12_inverse_wave(PR, AI, tid);       // A11 = PR[0:12] (a zero pivot: the final stage's solve reports it)
  __syncthreads();
  if (tid < 132) {
    const int i = tid / 12, j = tid % 12;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 12; k++) acc += PR[(12 + i) * 12 + k] * AI[k * 12 + j];
    G2[tid] = acc;
  }
  __syncthreads();
}

// The same half, serially, for the host (flimo_update_chain computes iteration -1's with it: x == x_prop there, so no
// transcendental function is evaluated and host and device agree bit for bit).  dxn[23], PR[276] out.
__host__ inline void ik_pre_serial(const double xc[26], const double xp[26], const double* P_prop, double R, double* dxn, double* AI, double* G2) {
  const int n = IK_N;
  double dx[IK_N], Jb[22], P_[529], PR[276];
  for (int s = 0; s < 2; s++) {
    const int o = s == 0 ? 3 : 7, idx = s == 0 ? 3 : 6;
    const Q4 a{xc[o], xc[o + 1], xc[o + 2], xc[o + 3]};
    const Q4 b{-xp[o], -xp[o + 1], -xp[o + 2], xp[o + 3]};
    double r[3];
    ik_so3_log(ik_qmul(b, a), r);
    for (int i = 0; i < 3; i++) dx[idx + i] = r[i];
    ik_A_T(r, Jb + 9 * s);
  }
  {
    double d[2];
    ik_s2_boxminus(xc + 23, xp + 23, d);
    dx[21] = d[0]; dx[22] = d[1];
    ik_s2_J(xc + 23, xp + 23, d, Jb + 18);
  }
  for (int i = 0; i < 3; i++) {
    dx[i] = xc[i] - xp[i];
    dx[9 + i] = xc[11 + i] - xp[11 + i];
    dx[12 + i] = xc[14 + i] - xp[14 + i];
    dx[15 + i] = xc[17 + i] - xp[17 + i];
    dx[18 + i] = xc[20 + i] - xp[20 + i];
  }
  for (int i = 0; i < n; i++) dxn[i] = dx[i];
  for (int s = 0; s < 2; s++) {
    const int idx = s == 0 ? 3 : 6;
    const double* J = Jb + 9 * s;
    for (int r = 0; r < 3; r++) dxn[idx + r] = J[r * 3 + 0] * dx[idx] + J[r * 3 + 1] * dx[idx + 1] + J[r * 3 + 2] * dx[idx + 2];
  }
  for (int r = 0; r < 2; r++) dxn[21 + r] = Jb[18 + r * 2 + 0] * dx[21] + Jb[18 + r * 2 + 1] * dx[22];
  for (int i = 0; i < 529; i++) P_[i] = P_prop[i];
  auto left = [&](int B, int idx, const double* J) {
    for (int c = 0; c < n; c++) {
      double t[3];
      for (int r = 0; r < B; r++) { double s = 0; for (int k = 0; k < B; k++) s += J[r * B + k] * P_[(idx + k) * n + c]; t[r] = s; }
      for (int r = 0; r < B; r++) P_[(idx + r) * n + c] = t[r];
    }
  };
  auto right = [&](int B, int idx, const double* J) {
    for (int r = 0; r < n; r++) {
      double t[3];
      for (int c = 0; c < B; c++) { double s = 0; for (int k = 0; k < B; k++) s += P_[r * n + idx + k] * J[c * B + k]; t[c] = s; }
      for (int c = 0; c < B; c++) P_[r * n + idx + c] = t[c];
    }
  };
  left(3, 3, Jb); right(3, 3, Jb); left(3, 6, Jb + 9); right(3, 6, Jb + 9); left(2, 21, Jb + 18); right(2, 21, Jb + 18);
  for (int e = 0; e < n * 12; e++) PR[e] = P_[(e / 12) * n + (e % 12)] / R;
  if (!ik_inverse_gj12_serial(PR, AI)) for (int e = 0; e < 144; e++) AI[e] = 0.0;      // (singular A11: the solve's zero pivot hands the loop back)
  for (int i = 0; i < 11; i++)
    for (int j = 0; j < 12; j++) {
      double acc = 0.0;
      for (int k = 0; k < 12; k++) acc += PR[(12 + i) * 12 + k] * AI[k * 12 + j];
      G2[i * 12 + j] = acc;
    }
}

// ---- the extra workgroup of a pass's reducing launch: leaves the measurement-independent half of THIS iteration (and, in the
//      first pass, the prior itself) in the device filter, written through; the caller then lets it arrive at the ticket ----
__device__ __forceinline__ void ik_extra_block(const ChainCtl& ch, double* lds, int tid) {
  ChainState* S = ch.S;
  const int n = IK_N;
  if (ch.prior) {
    const ChainPrior* pr = ch.prior;
    for (int i = tid; i < 529; i += 256) ik_st(&S->P_prop[i], pr->P[i]);
    for (int e = tid; e < 276; e += 256) ik_st(&S->pre_AG[e], pr->AG[e]);
    if (tid < 26) { const double v = pr->x[tid]; ik_st(&S->x[tid], v); ik_st(&S->x_prop[tid], v); }
    else if (tid >= 32 && tid < 32 + n) ik_st(&S->limit[tid - 32], pr->limit[tid - 32]);
    else if (tid >= 64 && tid < 64 + n) ik_st(&S->pre_dxn[tid - 64], pr->dxn[tid - 64]);
    else if (tid == 128) {
      ik_st(&S->R, pr->R); ik_st(&S->D, pr->D);
      ik_sti(&S->max_iter, pr->max_iter); ik_sti(&S->it, -1); ik_sti(&S->t, 0); ik_sti(&S->passes, 0);
    }
  } else {
    double* P_ = lds + IKL_P;
    double* xc = lds + IKL_XC;
    double* xp = lds + IKL_XP;
    for (int i = tid; i < 529; i += 256) P_[i] = S->P_prop[i];                    // :1655
    if (tid < 26) { xc[tid] = ik_ld(&S->x[tid]); xp[tid] = ik_ld(&S->x_prop[tid]); }   // (x: stored by the algebra of the last iteration, maybe microseconds ago)
    const double R = S->R;
    __syncthreads();
    ik_pre_block(lds, R, tid);
    const double* AG = lds + IKL_AI;                             // A11^-1 (144) and G2 (132): contiguous
    const double* dxn = lds + IKL_DXN;
    for (int e = tid; e < 276; e += 256) ik_st(&S->pre_AG[e], AG[e]);
    if (tid >= 64 && tid < 64 + n) ik_st(&S->pre_dxn[tid - 64], dxn[tid - 64]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// ---- the end of the chain, wherever it happens: the loop goes back to the host filter with the state this iteration measured at,
//      the loop variables, what every pass counted and -- for the reasons whose pass the host need not repeat -- the pass's sums ----
__device__ __forceinline__ void ik_hand_back(const ChainCtl& ch, double* lds, int reason, int it, int t, int passes, int M, int n_strag,
                                             int n_ties, int tid) {
  ChainState* S = ch.S;
  const double* xc = lds + IKL_XC;
  const double* live = lds + IKL_LIVE;
  if (tid < 26) ik_put(ch.res, CH_X + tid, xc[tid], ch.tag);
  if (tid >= 32 && tid < 32 + 3 * CH_MAX_PASSES) {
    const int k = tid - 32, p = k / 3;
    double v = 0.0;
    if (p < passes) v = ik_ld(&S->info[k]);
    else if (p == passes) v = (k % 3 == 0) ? (double)M : (k % 3 == 1 ? (double)n_strag : (double)n_ties);
    ik_put(ch.res, CH_PASSINFO + k, v, ch.tag);
  }
  if (tid >= 96 && tid < 96 + IK_LIVE) ik_put(ch.res, CH_SUMS + tid - 96, live[tid - 96], ch.tag);
  if (tid == 224) {
    ik_put(ch.res, CH_BAIL, (double)reason, ch.tag); ik_put(ch.res, CH_PASSES, (double)passes, ch.tag);
    ik_put(ch.res, CH_IT, (double)it, ch.tag); ik_put(ch.res, CH_T, (double)t, ch.tag);
    ik_put(ch.res, CH_STATUS, 2.0, ch.tag);
    ik_sti(&S->head.status, 2);
  }
}

// The next pass's float32 constants from the new state (Objects/State.cpp:38-55,136-172, Localizer.cpp:554-555), three independent
// pieces on three waves, written straight to the device filter's head
__device__ __forceinline__ void ik_stf(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ik_store_pose(const double* xn, ChainHead* H, int tid) {
  if (tid == 0) {
    const float p[3] = {(float)xn[0], (float)xn[1], (float)xn[2]};
    const float q[4] = {(float)xn[3], (float)xn[4], (float)xn[5], (float)xn[6]};
    float T[16];
    se3_from(q, p, T);
#pragma unroll
    for (int i = 0; i < 16; i++) ik_stf(&H->pose.RT[i], T[i]);
  } else if (tid == 64 || tid == 65) {
    const int o = tid == 64 ? 3 : 7, po = tid == 64 ? 0 : 11;    // (rot, pos) / (offset_R_L_I, offset_T_L_I)
    const float p[3] = {(float)xn[po], (float)xn[po + 1], (float)xn[po + 2]};
    const float q[4] = {(float)xn[o], (float)xn[o + 1], (float)xn[o + 2], (float)xn[o + 3]};
    float T[16];
    se3_inv_from(q, p, T);
    float* dst = tid == 64 ? H->pose.RT_inv : H->pose.TLI_inv;
#pragma unroll
    for (int i = 0; i < 16; i++) ik_stf(&dst[i], T[i]);
  } else if (tid == 128 || tid == 129) {
    const int o = tid == 128 ? 3 : 7;
    const double qc[4] = {-xn[o], -xn[o + 1], -xn[o + 2], xn[o + 3]};
    double Rd[9];
    quat_to_rot_d(qc, Rd);
    float* dst = tid == 128 ? H->pose.R_inv : H->pose.RLI_inv;
#pragma unroll
    for (int i = 0; i < 9; i++) ik_stf(&dst[i], (float)Rd[i]);
  }
}

// ---- the workgroup that completes a pass: from the pass's sums to the next state (:1722-1764), or the hand-back.
//      gran / the device filter were written by other workgroups of this launch (or by earlier launches): read past the L2.
//      used_RT: body -> world matrix this pass ran with (kernel argument of the first pass, the filter's head afterwards).
//      CHECK_TAGS: the sums must carry `seq` (the algebra as a launch of its own: a pass that failed left older ones).
//      Workgroup-wide, 256 threads; lds: IKL_END doubles, s_i: 32 ints.
//      Returns true when the chain goes on (the next pass's constants are stored), false when the loop went back to the host. ----
template <bool CHECK_TAGS>
__device__ __forceinline__ bool ik_final_stage(const ChainCtl& ch, unsigned long long seq, const float* __restrict__ used_RT, double* lds,
                                               int* s_i, int tid) {
  ChainState* S = ch.S;
  const int n = IK_N;
  double* AI = lds + IKL_AI;
  double* G2 = lds + IKL_G2;
  double* HTH = lds + IKL_HTH;
  double* T = lds + IKL_T;
  double* HTh = lds + IKL_HTh;
  double* dxn = lds + IKL_DXN;
  double* vv = lds + IKL_V;
  double* uu = lds + IKL_U;
  double* dxu = lds + IKL_DXU;
  double* xn = lds + IKL_XN;
  double* xc = lds + IKL_XC;
  double* lim = lds + IKL_LIM;
  double* live = lds + IKL_LIVE;
  IK_STAMP(0);
  // ---- everything this iteration reads, in one round trip: the pass's sums (groups added in slot order, as the host adds them),
  //      the state, the limits, the measurement-independent half ----
  bool pass_ok = true;
  if (tid < IK_LIVE + 2) {
    // granule k of group g: {sum, pass number}; granules IK_LIVE / IK_LIVE + 1 of group 0: stragglers, ties
    const int groups = tid < IK_LIVE ? IK_GROUPS : 1;
    double g[IK_GROUPS];
    unsigned long long tg[IK_GROUPS];
#pragma unroll
    for (int q = 0; q < IK_GROUPS; q++) {
      const double* gp = reinterpret_cast<const double*>(ch.gran + (size_t)(q < groups ? q : 0) * IK_LIVE_PAD + tid);
      g[q] = ik_ld(gp);
      if (CHECK_TAGS) tg[q] = (unsigned long long)__double_as_longlong(ik_ld(gp + 1));
    }
    double r = g[0];
    if (CHECK_TAGS) pass_ok = tg[0] == seq;
#pragma unroll
    for (int q = 1; q < IK_GROUPS; q++)
      if (q < groups) { r += g[q]; if (CHECK_TAGS) pass_ok = pass_ok && tg[q] == seq; }
    live[tid] = r;
    // the sums in the form the algebra reads them: full H^T H (entry tid of the upper triangle -> (i, j) and (j, i)), H^T h
    if (tid < 78) {
      int i = 0, k = tid;
#pragma unroll
      for (int s = 0; s < 11; s++) if (k >= 12 - i) { k -= 12 - i; i++; }
      const int j = i + k;
      HTH[i * 12 + j] = r;
      HTH[j * 12 + i] = r;
    } else if (tid < 90) {
      HTh[tid - 78] = r;
    }
  } else if (tid >= 128 && tid < 128 + 26) {
    xc[tid - 128] = ik_ld(&S->x[tid - 128]);
  } else if (tid >= 160 && tid < 160 + n) {
    lim[tid - 160] = ik_ld(&S->limit[tid - 160]);
  } else if (tid >= 192 && tid < 192 + n) {
    dxn[tid - 192] = ik_ld(&S->pre_dxn[tid - 192]);
  } else if (tid == 224) {
    s_i[20] = ik_ldi(&S->it); s_i[21] = ik_ldi(&S->t); s_i[22] = ik_ldi(&S->passes); s_i[23] = ik_ldi(&S->max_iter);
  }
  for (int e = tid; e < 276; e += 256) AI[e] = ik_ld(&S->pre_AG[e]);            // A11^-1 and G2 (contiguous)
  const int all_ok = __syncthreads_and(pass_ok ? 1 : 0);
  IK_STAMP(1);
  const int M = (int)llrint(live[IK_LIVE - 1]);
  const int n_strag = (int)llrint(live[IK_LIVE]), n_ties = (int)llrint(live[IK_LIVE + 1]);
  const int it = s_i[20], t_in = s_i[21], passes = s_i[22], max_iter = s_i[23];
  // ---- branches the host filter takes over: a pass that did not publish, M < 23, exact distance ties ----
  if (!all_ok || M < n || n_ties > 0) {
    ik_hand_back(ch, lds, !all_ok ? CH_R_FAILED : (M < n ? CH_R_FEW : CH_R_TIES), it, t_in, passes, M, n_strag, n_ties, tid);
    return false;
  }
  // ---- gain (:1722-1729).  With A = P_ / R and B = H^T H:  P_inv = (A^-1 + E B E^T)^-1, and by the block-inverse identity
  //      P_inv E = [I; A21 A11^-1] (A11^-1 + B)^-1  -- the reference's formula without the two 23 x 23 inverses and without forming
  //      I + B A11 (whose 1 drowns in B A11 ~ 1e7: three digits worse on the measured states, tests/test_host_logic.py).
  //      N = A11^-1 + H^T H;  v = H^T h + H^T H dx_new[0:12];  N z = v;  dx_ = [z; G2 z] - dx_new ----
  if (tid < 144) {
    T[tid] = AI[tid] + HTH[tid];
  } else if (tid >= 192 && tid < 204) {
    const int i = tid - 192;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 12; k++) acc += HTH[i * 12 + k] * dxn[k];
    vv[i] = HTh[i] + acc;
  }
  __syncthreads();
  IK_STAMP(4);
  // Wave 0: N z = v by Gauss-Jordan in its registers, then dx_ = [z; G2 z] - dx_new (the step of :1733 in the order the host
  // filter also takes it; shared memory written and read by the same wave: in order, no workgroup barrier).  Beside it, one
  // lane of another wave: degeneracy (:1736-1744) -- when H^T H[0:6,0:6] - D I is positive definite every eigenvalue is >= D
  // and the projector is the identity (the host's shortcut); otherwise the host does the eigen-decomposition.
  if (tid < 64) {
    const bool ok = ik_gj12_solve_wave(T, vv, uu, tid);
    if (tid == 0) s_i[17] = ok ? 0 : 1;
    IK_STAMP(5);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (tid < 12) {
      dxu[tid] = uu[tid] - dxn[tid];
    } else if (tid < n) {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < 12; m++) acc += G2[(tid - 12) * 12 + m] * uu[m];
      dxu[tid] = acc - dxn[tid];
    }
  } else if (tid == 192) {
    const double D = ik_ld(&S->D);
    double Lc[21];                                            // lower triangle, row-major packed
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = 0; j <= i; j++) {
        double s = 0.5 * (HTH[i * 12 + j] + HTH[j * 12 + i]) - (i == j ? D : 0.0);
#pragma unroll
        for (int k = 0; k < j; k++) s -= Lc[i * (i + 1) / 2 + k] * Lc[j * (j + 1) / 2 + k];
        if (i == j) {
          if (!(s > 1e-9 * D)) ok = false;
          Lc[i * (i + 1) / 2 + i] = sqrt(ok ? s : 1.0);
        } else {
          Lc[i * (i + 1) / 2 + j] = s / Lc[j * (j + 1) / 2 + j];
        }
      }
    s_i[19] = ok ? 0 : 1;
  }
  __syncthreads();
  IK_STAMP(7);
  if (s_i[17] != 0 || s_i[19] != 0) {
    ik_hand_back(ch, lds, CH_R_DEGENERATE, it, t_in, passes, M, n_strag, n_ties, tid);
    return false;
  }
  // ---- x boxplus dx_ (:1747), convergence (:1757-1764) ----
  if (tid < 2) {
    const int o = (tid == 0) ? 3 : 7, idx = (tid == 0) ? 3 : 6;
    const Q4 a{xc[o], xc[o + 1], xc[o + 2], xc[o + 3]};
    const double v[3] = {dxu[idx], dxu[idx + 1], dxu[idx + 2]};
    const Q4 r = ik_qmul(a, ik_exp_quat(v, 0.5));              // SO3::boxplus: exp(v, scale / 2)
    xn[o] = r.x; xn[o + 1] = r.y; xn[o + 2] = r.z; xn[o + 3] = r.w;
  } else if (tid == 64) {
    double B[6], Bu[3], Rm[9], g[3];
    ik_s2_Bx(xc + 23, B);
    for (int i = 0; i < 3; i++) Bu[i] = B[i * 2 + 0] * dxu[21] + B[i * 2 + 1] * dxu[22];
    ik_q2r(ik_exp_quat(Bu, 0.5), Rm);
    ik_mv3(Rm, xc + 23, g);
    for (int i = 0; i < 3; i++) xn[23 + i] = g[i];
  } else if (tid >= 128 && tid < 128 + 15) {
    const int e = tid - 128, seg = e / 3, i = e % 3;         // pos, offset_T_L_I, vel, bg, ba
    const int xo = seg == 0 ? 0 : 8 + 3 * seg, dxo = seg == 0 ? 0 : 6 + 3 * seg;
    xn[xo + i] = xc[xo + i] + dxu[dxo + i];
  } else if (tid >= 192 && tid < 256) {
    const int l = tid - 192;
    const bool over = l < n && fabs(dxu[l]) > lim[l];
    const bool conv = __ballot(over) == 0ull;
    if (l == 0) {
      const int t = t_in + (conv ? 1 : 0);
      s_i[18] = (t > 1 || it == max_iter - 1) ? 1 : 0;        // this iteration ends the loop (covariance update due)?
      s_i[21] = t;
    }
  }
  __syncthreads();
  IK_STAMP(8);
  if (s_i[18] != 0) {
    // the iteration that ends the loop (:1764): the host filter runs it from these sums -- state, covariance, log -- without a pass
    ik_hand_back(ch, lds, CH_R_FINAL, it, t_in, passes, M, n_strag, n_ties, tid);
    IK_STAMP(11);
    return false;
  }
  const int t_out = s_i[21];
  // per-pass bookkeeping, the optional log, the next pass's constants and the bound's reference pose (the pose this pass ran with)
  // (everything another workgroup reads later is written through: the next pass may already be polling for it)
  if (tid == 200 && passes < CH_MAX_PASSES) { ik_st(&S->info[3 * passes], (double)M); ik_st(&S->info[3 * passes + 1], (double)n_strag); ik_st(&S->info[3 * passes + 2], (double)n_ties); }
  if (ch.log && passes < CH_MAX_PASSES) {
    double2* lg = ch.log + (size_t)passes * CH_LOGN;
    if (tid < 144) ik_put(lg, tid, HTH[tid], ch.tag);
    if (tid >= 144 && tid < 156) ik_put(lg, tid, HTh[tid - 144], ch.tag);
    if (tid >= 160 && tid < 160 + n) ik_put(lg, 156 + tid - 160, dxu[tid - 160], ch.tag);
    if (tid >= 192 && tid < 192 + 26) ik_put(lg, 179 + tid - 192, xn[tid - 192], ch.tag);
  }
  if (tid >= 224 && tid < 240) ik_stf(&S->head.prev_RT[tid - 224], used_RT[tid - 224]);
  __syncthreads();                                              // (used_RT may BE head.pose.RT: read above before it is overwritten)
  ik_store_pose(xn, &S->head, tid);
  if (tid == 32) { ik_sti(&S->head.status, 0); ik_sti(&S->it, it + 1); ik_sti(&S->t, t_out); ik_sti(&S->passes, passes + 1); }
  if (tid >= 96 && tid < 96 + 26) ik_st(&S->x[tid - 96], xn[tid - 96]);
  IK_STAMP(10);
  return true;
}

}  // namespace flimo
