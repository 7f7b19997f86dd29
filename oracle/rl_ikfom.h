// oracle/rl_ikfom.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement (float64) of the IKFoM pieces the fast_LIMO hot path uses:
//   state_ikfom manifold          reference include/IKFoM/use-ikfom.hpp:12-21
//   MTK::exp / log / cos_sinc_sqrt include/IKFoM/IKFoM_toolkit/mtk/src/mtkmath.hpp:143-174,250-290
//   MTK::A_matrix                 mtkmath.hpp:236-247
//   SO3 boxplus/boxminus/exp/log  mtk/types/SOn.hpp:233-240,284-297
//   S2  boxplus/boxminus/S2_Bx/S2_Nx_yy/S2_Mx  mtk/types/S2.hpp:136-167,179-232,259-281
//        with S2<double,98090,10000,1>: length 9.809, chart type 1 (use-ikfom.hpp:8)
//   compound boxplus/boxminus/oplus  mtk/build_manifold.hpp:192-200
//   process model get_f/df_dx/df_dw  include/IKFoM/use-ikfom.cpp:43-84
//   esekf::predict                esekfom/esekfom.hpp:279-384
//   esekf::update_iterated_dyn_share_modified   esekfom.hpp:1620-1823
// Quirks kept (SURVEY.md section 8 a-notes 4-7): `scalar_type(1/2)` is integer division == 0
// in predict (:312,:344) and S2_Mx (S2.hpp:277); HTH is defined as 0 when M < 23 (the
// reference leaves it uninitialised there); convergence is tested on dx_, not on the
// degeneracy-projected step; the loop runs maximum_iter+1 passes at most.
// Dense products are plain left-to-right loops (Eigen's own order is unverifiable here:
// PARITY UNPINNED at the 1e-16 relative level; pose tolerance is 1e-4).
#pragma once
#include <vector>
#include <functional>
#include <utility>
#include "rl_linalg.h"

namespace oracle {

static const int NDOF = 23;   // state::DOF
static const int NDIM = 24;   // state::DIM
static const double MTK_TOL = 1e-11;          // MTK::tolerance<double>() mtkmath.hpp:122
static const double S2_LEN = 98090.0 / 10000.0;

// ---- MTK scalar helpers -----------------------------------------------------------------
inline std::pair<double, double> cos_sinc_sqrt(double x2) {     // mtkmath.hpp:143-174
  static const double taylor_0_bound = 2.220446049250313e-16;   // boost epsilon<double>
  static const double taylor_2_bound = std::sqrt(taylor_0_bound);
  static const double taylor_n_bound = std::sqrt(taylor_2_bound);
  if (x2 >= taylor_n_bound) {
    double x = std::sqrt(x2);
    return std::make_pair(std::cos(x), std::sin(x) / x);
  }
  static const double inv[] = {1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.};
  double cosi = 1., sinc = 1;
  double term = -1 / 2. * x2;
  for (int i = 0; i < 3; ++i) {
    cosi += term;
    term *= inv[2 * i];
    sinc += term;
    term *= -inv[2 * i + 1] * x2;
  }
  return std::make_pair(cosi, sinc);
}

// MTK::exp<scalar,3>: writes the vector part, returns the scalar part.  mtkmath.hpp:250-256
inline double mtk_exp3(double out[3], const double v[3], double scale) {
  double norm2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  std::pair<double, double> cs = cos_sinc_sqrt(scale * scale * norm2);
  double mult = cs.second * scale;
  out[0] = mult * v[0]; out[1] = mult * v[1]; out[2] = mult * v[2];
  return cs.first;
}

inline Quatd so3_exp(const double v[3], double scale = 1.0) {   // SOn.hpp:284-288
  double vec[3];
  double w = mtk_exp3(vec, v, scale / 2);
  Quatd q;
  q.w = w; q.x = vec[0]; q.y = vec[1]; q.z = vec[2];
  return q;
}

inline void so3_log(const Quatd& q, double out[3]) {            // SOn.hpp:293-297 + mtkmath.hpp:268-287
  double nv = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  if (nv < MTK_TOL) nv = MTK_TOL;                               // plus_minus_periodicity == true
  double s = 2.0 / nv * std::atan(nv / q.w);
  out[0] = s * q.x; out[1] = s * q.y; out[2] = s * q.z;
}

inline void hat3(const double v[3], double H[3][3]) {           // mtkmath.hpp:176-183
  H[0][0] = 0;     H[0][1] = -v[2]; H[0][2] = v[1];
  H[1][0] = v[2];  H[1][1] = 0;     H[1][2] = -v[0];
  H[2][0] = -v[1]; H[2][1] = v[0];  H[2][2] = 0;
}

inline void mat3_mul(const double A[3][3], const double B[3][3], double C[3][3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += A[i][k] * B[k][j];
      C[i][j] = s;
    }
}

inline void A_matrix(const double v[3], double A[3][3]) {       // mtkmath.hpp:236-247
  double sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  double norm = std::sqrt(sq);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = (i == j) ? 1.0 : 0.0;
  if (norm < MTK_TOL) return;
  double H[3][3], HH[3][3];
  hat3(v, H);
  mat3_mul(H, H, HH);
  double c1 = (1 - std::cos(norm)) / sq;
  double c2 = (1 - std::sin(norm) / norm) / sq;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i][j] = A[i][j] + c1 * H[i][j] + c2 * HH[i][j];
}

// ---- S2 (chart type 1, length 9.809) ----------------------------------------------------
struct S2g {
  double vec[3];
  S2g() { vec[0] = S2_LEN; vec[1] = 0; vec[2] = 0; }            // S2.hpp:114-118 (S2_typ == 1)
  S2g(double x, double y, double z) {                           // :119-122 normalise then scale
    double n = std::sqrt(x * x + y * y + z * z);
    vec[0] = x / n * S2_LEN; vec[1] = y / n * S2_LEN; vec[2] = z / n * S2_LEN;
  }
  void Bx(double B[3][2]) const {                               // :179-232, S2_typ == 1 branch
    const double L = S2_LEN;
    if (vec[0] + L > MTK_TOL) {
      B[0][0] = -vec[1];                              B[0][1] = -vec[2];
      B[1][0] = L - vec[1] * vec[1] / (L + vec[0]);   B[1][1] = -vec[2] * vec[1] / (L + vec[0]);
      B[2][0] = -vec[2] * vec[1] / (L + vec[0]);      B[2][1] = L - vec[2] * vec[2] / (L + vec[0]);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 2; j++) B[i][j] /= L;
    } else {
      for (int i = 0; i < 3; i++) for (int j = 0; j < 2; j++) B[i][j] = 0;
      B[1][1] = -1; B[2][0] = 1;
    }
  }
  void rotate_by(const Quatd& q) {
    double R[3][3];
    quat_to_rot<double>(q, R);
    double o[3];
    for (int i = 0; i < 3; i++) o[i] = R[i][0] * vec[0] + R[i][1] * vec[1] + R[i][2] * vec[2];
    vec[0] = o[0]; vec[1] = o[1]; vec[2] = o[2];
  }
  void boxplus(const double d[2], double scale = 1.0) {         // :136-142
    double B[3][2];
    Bx(B);
    double Bu[3];
    for (int i = 0; i < 3; i++) Bu[i] = B[i][0] * d[0] + B[i][1] * d[1];
    Quatd r;
    double v3[3];
    r.w = mtk_exp3(v3, Bu, scale / 2);
    r.x = v3[0]; r.y = v3[1]; r.z = v3[2];
    rotate_by(r);
  }
  void oplus(const double d[3], double scale = 1.0) {           // :129-134
    Quatd r;
    double v3[3];
    r.w = mtk_exp3(v3, d, scale / 2);
    r.x = v3[0]; r.y = v3[1]; r.z = v3[2];
    rotate_by(r);
  }
  void boxminus(double res[2], const S2g& other) const {        // :144-167
    double H[3][3];
    hat3(vec, H);
    double hv[3];
    for (int i = 0; i < 3; i++) hv[i] = H[i][0] * other.vec[0] + H[i][1] * other.vec[1] + H[i][2] * other.vec[2];
    double v_sin = std::sqrt(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2]);
    double v_cos = vec[0] * other.vec[0] + vec[1] * other.vec[1] + vec[2] * other.vec[2];
    double theta = std::atan2(v_sin, v_cos);
    if (v_sin < MTK_TOL) {
      if (std::fabs(theta) > MTK_TOL) { res[0] = 3.1415926; res[1] = 0; }
      else { res[0] = 0; res[1] = 0; }
    } else {
      double B[3][2];
      other.Bx(B);
      double Ho[3][3];
      hat3(other.vec, Ho);
      double t[3];
      for (int i = 0; i < 3; i++) t[i] = Ho[i][0] * vec[0] + Ho[i][1] * vec[1] + Ho[i][2] * vec[2];
      double f = theta / v_sin;
      for (int j = 0; j < 2; j++) res[j] = f * (B[0][j] * t[0] + B[1][j] * t[1] + B[2][j] * t[2]);
    }
  }
  void Nx_yy(double N[2][3]) const {                            // :259-264
    double B[3][2], H[3][3];
    Bx(B);
    hat3(vec, H);
    const double f = 1 / S2_LEN / S2_LEN;
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) N[i][j] = f * (B[0][i] * H[0][j] + B[1][i] * H[1][j] + B[2][i] * H[2][j]);
  }
  void Mx(double M[3][2], const double delta[2]) const {        // :266-281
    double B[3][2], H[3][3];
    Bx(B);
    hat3(vec, H);
    double dn = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
    if (dn < MTK_TOL) {
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) M[i][j] = -(H[i][0] * B[0][j] + H[i][1] * B[1][j] + H[i][2] * B[2][j]);
    } else {
      double Bu[3];
      for (int i = 0; i < 3; i++) Bu[i] = B[i][0] * delta[0] + B[i][1] * delta[1];
      // exp_delta = exp(Bu, scalar(1/2)) with 1/2 == 0 (integer division): identity rotation.
      Quatd e;
      double v3[3];
      e.w = mtk_exp3(v3, Bu, double(1 / 2));
      e.x = v3[0]; e.y = v3[1]; e.z = v3[2];
      double E[3][3], A[3][3], EH[3][3], EHA[3][3];
      quat_to_rot<double>(e, E);
      A_matrix(Bu, A);
      double At[3][3];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[i][j] = A[j][i];
      mat3_mul(E, H, EH);
      mat3_mul(EH, At, EHA);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 2; j++) M[i][j] = -(EHA[i][0] * B[0][j] + EHA[i][1] * B[1][j] + EHA[i][2] * B[2][j]);
    }
  }
};

// ---- state_ikfom ------------------------------------------------------------------------
struct StateIkfom {                                             // use-ikfom.hpp:12-21
  double pos[3];
  Quatd rot;
  Quatd offset_R_L_I;
  double offset_T_L_I[3];
  double vel[3], bg[3], ba[3];
  S2g grav;
  StateIkfom() {
    for (int i = 0; i < 3; i++) pos[i] = offset_T_L_I[i] = vel[i] = bg[i] = ba[i] = 0.0;
  }
  // DOF layout: pos 0 | rot 3 | offR 6 | offT 9 | vel 12 | bg 15 | ba 18 | grav 21(2)
  void boxplus(const double d[NDOF]) {                          // build_manifold.hpp:192-194
    for (int i = 0; i < 3; i++) pos[i] += d[i];
    rot = qmul(rot, so3_exp(d + 3));
    offset_R_L_I = qmul(offset_R_L_I, so3_exp(d + 6));
    for (int i = 0; i < 3; i++) offset_T_L_I[i] += d[9 + i];
    for (int i = 0; i < 3; i++) vel[i] += d[12 + i];
    for (int i = 0; i < 3; i++) bg[i] += d[15 + i];
    for (int i = 0; i < 3; i++) ba[i] += d[18 + i];
    grav.boxplus(d + 21);
  }
  void oplus(const double f[NDIM], double dt) {                 // build_manifold.hpp:195-197
    for (int i = 0; i < 3; i++) pos[i] += dt * f[i];
    rot = qmul(rot, so3_exp(f + 3, dt));
    offset_R_L_I = qmul(offset_R_L_I, so3_exp(f + 6, dt));
    for (int i = 0; i < 3; i++) offset_T_L_I[i] += dt * f[9 + i];
    for (int i = 0; i < 3; i++) vel[i] += dt * f[12 + i];
    for (int i = 0; i < 3; i++) bg[i] += dt * f[15 + i];
    for (int i = 0; i < 3; i++) ba[i] += dt * f[18 + i];
    grav.oplus(f + 21, dt);
  }
  void boxminus(double r[NDOF], const StateIkfom& o) const {    // build_manifold.hpp:198-200
    for (int i = 0; i < 3; i++) r[i] = pos[i] - o.pos[i];
    so3_log(qmul(o.rot.conjugate(), rot), r + 3);
    so3_log(qmul(o.offset_R_L_I.conjugate(), offset_R_L_I), r + 6);
    for (int i = 0; i < 3; i++) r[9 + i] = offset_T_L_I[i] - o.offset_T_L_I[i];
    for (int i = 0; i < 3; i++) r[12 + i] = vel[i] - o.vel[i];
    for (int i = 0; i < 3; i++) r[15 + i] = bg[i] - o.bg[i];
    for (int i = 0; i < 3; i++) r[18 + i] = ba[i] - o.ba[i];
    grav.boxminus(r + 21, o.grav);
  }
  // flat export: pos3 rot(xyzw) offR(xyzw) offT3 vel3 bg3 ba3 grav3 = 26 doubles
  void to_flat(double* o) const {
    int k = 0;
    for (int i = 0; i < 3; i++) o[k++] = pos[i];
    o[k++] = rot.x; o[k++] = rot.y; o[k++] = rot.z; o[k++] = rot.w;
    o[k++] = offset_R_L_I.x; o[k++] = offset_R_L_I.y; o[k++] = offset_R_L_I.z; o[k++] = offset_R_L_I.w;
    for (int i = 0; i < 3; i++) o[k++] = offset_T_L_I[i];
    for (int i = 0; i < 3; i++) o[k++] = vel[i];
    for (int i = 0; i < 3; i++) o[k++] = bg[i];
    for (int i = 0; i < 3; i++) o[k++] = ba[i];
    for (int i = 0; i < 3; i++) o[k++] = grav.vec[i];
  }
  void from_flat(const double* o) {
    int k = 0;
    for (int i = 0; i < 3; i++) pos[i] = o[k++];
    rot.x = o[k++]; rot.y = o[k++]; rot.z = o[k++]; rot.w = o[k++];
    offset_R_L_I.x = o[k++]; offset_R_L_I.y = o[k++]; offset_R_L_I.z = o[k++]; offset_R_L_I.w = o[k++];
    for (int i = 0; i < 3; i++) offset_T_L_I[i] = o[k++];
    for (int i = 0; i < 3; i++) vel[i] = o[k++];
    for (int i = 0; i < 3; i++) bg[i] = o[k++];
    for (int i = 0; i < 3; i++) ba[i] = o[k++];
    for (int i = 0; i < 3; i++) grav.vec[i] = o[k++];
  }
};

struct InputIkfom { double acc[3], gyro[3]; };

inline void quat_rotate_d(const Quatd& q, const double v[3], double out[3]) {   // _transformVector
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  for (int i = 0; i < 3; i++) uv[i] += uv[i];
  double c[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
  for (int i = 0; i < 3; i++) out[i] = v[i] + q.w * uv[i] + c[i];
}

// ---- process model (use-ikfom.cpp:43-84) ------------------------------------------------
inline void ikfom_get_f(const StateIkfom& s, const InputIkfom& in, double f[NDIM]) {
  for (int i = 0; i < NDIM; i++) f[i] = 0;
  double a_b[3] = {in.acc[0] - s.ba[0], in.acc[1] - s.ba[1], in.acc[2] - s.ba[2]};
  double a_in[3];
  quat_rotate_d(s.rot, a_b, a_in);
  for (int i = 0; i < 3; i++) {
    f[i] = s.vel[i];
    f[i + 3] = in.gyro[i] - s.bg[i];
    f[i + 12] = a_in[i] + s.grav.vec[i];
  }
}
inline void ikfom_df_dx(const StateIkfom& s, const InputIkfom& in, double F[NDIM][NDOF]) {
  for (int i = 0; i < NDIM; i++) for (int j = 0; j < NDOF; j++) F[i][j] = 0;
  for (int i = 0; i < 3; i++) F[i][12 + i] = 1.0;
  double acc_[3] = {in.acc[0] - s.ba[0], in.acc[1] - s.ba[1], in.acc[2] - s.ba[2]};
  double R[3][3], H[3][3], RH[3][3];
  quat_to_rot<double>(s.rot, R);
  hat3(acc_, H);
  mat3_mul(R, H, RH);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { F[12 + i][3 + j] = -RH[i][j]; F[12 + i][18 + j] = -R[i][j]; }
  double zero2[2] = {0, 0};
  double gm[3][2];
  s.grav.Mx(gm, zero2);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 2; j++) F[12 + i][21 + j] = gm[i][j];
  for (int i = 0; i < 3; i++) F[3 + i][15 + i] = -1.0;
}
inline void ikfom_df_dw(const StateIkfom& s, const InputIkfom&, double G[NDIM][12]) {
  for (int i = 0; i < NDIM; i++) for (int j = 0; j < 12; j++) G[i][j] = 0;
  double R[3][3];
  quat_to_rot<double>(s.rot, R);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) G[12 + i][3 + j] = -R[i][j];
  for (int i = 0; i < 3; i++) { G[3 + i][i] = -1.0; G[15 + i][6 + i] = 1.0; G[18 + i][9 + i] = 1.0; }
}

// ---- the filter -------------------------------------------------------------------------
struct MeasOut {                 // what h_share_model hands back (use-ikfom.cpp:10-31)
  int M = 0;                     // rows of h_x
  std::vector<double> h_x;       // M x 12 row-major
  std::vector<double> h;         // M
};

struct IterLog {                 // per-pass trace used by the golden fixtures / parity tests
  int M;
  double HTH[144];
  double HTh[12];
  double dx[NDOF];
  double x_after[26];
};

struct Esekf {
  StateIkfom x_;
  double P_[NDOF][NDOF];
  int maximum_iter = 0;
  double limit[NDOF];
  std::function<void(const StateIkfom&, MeasOut&)> h_dyn_share;
  std::vector<IterLog> log;      // filled by update (cleared at entry)

  Esekf() {
    for (int i = 0; i < NDOF; i++) { for (int j = 0; j < NDOF; j++) P_[i][j] = (i == j) ? 1.0 : 0.0; limit[i] = 1e-3; }
  }

  // esekfom.hpp:279-384
  void predict(double dt, const double Q[12][12], const InputIkfom& in) {
    const int n = NDOF;
    double f_[NDIM];
    static thread_local double f_x_[NDIM][NDOF], f_w_[NDIM][12];
    ikfom_get_f(x_, in, f_);
    ikfom_df_dx(x_, in, f_x_);
    ikfom_df_dw(x_, in, f_w_);
    double f_x_final[NDOF][NDOF], f_w_final[NDOF][12], F_x1[NDOF][NDOF];
    StateIkfom x_before = x_;
    x_.oplus(f_, dt);
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) F_x1[i][j] = (i == j) ? 1.0 : 0.0;
    // vect_state: (idx, dim, dof)
    static const int vs[5][2] = {{0, 0}, {9, 9}, {12, 12}, {15, 15}, {18, 18}};
    for (int v = 0; v < 5; v++) {
      int idx = vs[v][0], dim = vs[v][1];
      for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) f_x_final[idx + j][i] = f_x_[dim + j][i];
      for (int i = 0; i < 12; i++) for (int j = 0; j < 3; j++) f_w_final[idx + j][i] = f_w_[dim + j][i];
    }
    static const int so3s[2][2] = {{3, 3}, {6, 6}};
    for (int s = 0; s < 2; s++) {
      int idx = so3s[s][0], dim = so3s[s][1];
      double seg[3];
      for (int i = 0; i < 3; i++) seg[i] = -1 * f_[dim + i] * dt;
      Quatd res;
      double v3[3];
      res.w = mtk_exp3(v3, seg, double(1 / 2));     // integer division quirk: identity
      res.x = v3[0]; res.y = v3[1]; res.z = v3[2];
      double Rr[3][3];
      quat_to_rot<double>(res, Rr);
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) F_x1[idx + i][idx + j] = Rr[i][j];
      double A[3][3];
      A_matrix(seg, A);
      for (int i = 0; i < n; i++)
        for (int r = 0; r < 3; r++)
          f_x_final[idx + r][i] = A[r][0] * f_x_[dim][i] + A[r][1] * f_x_[dim + 1][i] + A[r][2] * f_x_[dim + 2][i];
      for (int i = 0; i < 12; i++)
        for (int r = 0; r < 3; r++)
          f_w_final[idx + r][i] = A[r][0] * f_w_[dim][i] + A[r][1] * f_w_[dim + 1][i] + A[r][2] * f_w_[dim + 2][i];
    }
    {
      int idx = 21, dim = 21;
      double seg[3];
      for (int i = 0; i < 3; i++) seg[i] = f_[dim + i] * dt;
      double zero2[2] = {0, 0};
      Quatd res;
      double v3[3];
      res.w = mtk_exp3(v3, seg, double(1 / 2));
      res.x = v3[0]; res.y = v3[1]; res.z = v3[2];
      double Rr[3][3];
      quat_to_rot<double>(res, Rr);
      double Nx[2][3], Mx[3][2];
      x_.grav.Nx_yy(Nx);
      x_before.grav.Mx(Mx, zero2);
      double NR[2][3];
      for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) NR[i][j] = Nx[i][0] * Rr[0][j] + Nx[i][1] * Rr[1][j] + Nx[i][2] * Rr[2][j];
      for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++)
        F_x1[idx + i][idx + j] = NR[i][0] * Mx[0][j] + NR[i][1] * Mx[1][j] + NR[i][2] * Mx[2][j];
      double xh[3][3], A[3][3];
      hat3(x_before.grav.vec, xh);
      A_matrix(seg, A);
      double NRH[2][3], T[2][3];
      for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) NRH[i][j] = NR[i][0] * xh[0][j] + NR[i][1] * xh[1][j] + NR[i][2] * xh[2][j];
      for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++)
        T[i][j] = -(NRH[i][0] * A[j][0] + NRH[i][1] * A[j][1] + NRH[i][2] * A[j][2]);   // * A^T
      for (int i = 0; i < n; i++)
        for (int r = 0; r < 2; r++)
          f_x_final[idx + r][i] = T[r][0] * f_x_[dim][i] + T[r][1] * f_x_[dim + 1][i] + T[r][2] * f_x_[dim + 2][i];
      for (int i = 0; i < 12; i++)
        for (int r = 0; r < 2; r++)
          f_w_final[idx + r][i] = T[r][0] * f_w_[dim][i] + T[r][1] * f_w_[dim + 1][i] + T[r][2] * f_w_[dim + 2][i];
    }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) F_x1[i][j] += f_x_final[i][j] * dt;
    // P = F P F^T + (dt G) Q (dt G)^T
    double FP[NDOF][NDOF], Pn[NDOF][NDOF];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < n; k++) s += F_x1[i][k] * P_[k][j]; FP[i][j] = s; }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < n; k++) s += FP[i][k] * F_x1[j][k]; Pn[i][j] = s; }
    double GQ[NDOF][12];
    for (int i = 0; i < n; i++) for (int j = 0; j < 12; j++) { double s = 0; for (int k = 0; k < 12; k++) s += (dt * f_w_final[i][k]) * Q[k][j]; GQ[i][j] = s; }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { double s = 0; for (int k = 0; k < 12; k++) s += GQ[i][k] * (dt * f_w_final[j][k]); P_[i][j] = Pn[i][j] + s; }
  }

  // Left/right re-projection of P (and optionally dx / K_x rows) through a 3x3 or 2x2 block J
  // at DOF offset idx:  rows <- J*rows ; cols <- cols*J^T      (esekfom.hpp:1659-1697)
  template <int B>
  static void reproject_rows(double M[NDOF][NDOF], int idx, const double J[B][B], int ncols) {
    for (int i = 0; i < ncols; i++) {
      double t[B];
      for (int r = 0; r < B; r++) { double s = 0; for (int k = 0; k < B; k++) s += J[r][k] * M[idx + k][i]; t[r] = s; }
      for (int r = 0; r < B; r++) M[idx + r][i] = t[r];
    }
  }
  template <int B>
  static void reproject_cols(double M[NDOF][NDOF], int idx, const double J[B][B]) {
    for (int i = 0; i < NDOF; i++) {
      double t[B];
      for (int c = 0; c < B; c++) { double s = 0; for (int k = 0; k < B; k++) s += M[i][idx + k] * J[c][k]; t[c] = s; }
      for (int c = 0; c < B; c++) M[i][idx + c] = t[c];
    }
  }

  // esekfom.hpp:1620-1823.  R = measurement variance, D = degeneracy threshold.
  void update_iterated_dyn_share_modified(double R, double D) {
    const int n = NDOF;
    log.clear();
    int t = 0;
    StateIkfom x_propagated = x_;
    static thread_local double P_propagated[NDOF][NDOF];
    std::memcpy(P_propagated, P_, sizeof(P_));
    double K_h[NDOF];
    static thread_local double K_x[NDOF][NDOF];
    double dx_new[NDOF];
    for (int i = 0; i < n; i++) dx_new[i] = 0;
    MeasOut meas;

    for (int it = -1; it < maximum_iter; it++) {
      h_dyn_share(x_, meas);                                    // :1637
      const int M = meas.M;
      double dx[NDOF];
      x_.boxminus(dx, x_propagated);                            // :1652
      for (int i = 0; i < n; i++) dx_new[i] = dx[i];
      std::memcpy(P_, P_propagated, sizeof(P_));                // :1655

      static const int so3_idx[2] = {3, 6};
      for (int s = 0; s < 2; s++) {                             // :1659-1674
        int idx = so3_idx[s];
        double A[3][3], J[3][3];
        A_matrix(dx + idx, A);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[i][j] = A[j][i];
        double tv[3];
        for (int r = 0; r < 3; r++) tv[r] = J[r][0] * dx_new[idx] + J[r][1] * dx_new[idx + 1] + J[r][2] * dx_new[idx + 2];
        for (int r = 0; r < 3; r++) dx_new[idx + r] = tv[r];
        reproject_rows<3>(P_, idx, J, n);
        reproject_cols<3>(P_, idx, J);
      }
      {                                                         // :1676-1697
        int idx = 21;
        double Nx[2][3], Mx[3][2], J[2][2];
        x_.grav.Nx_yy(Nx);
        x_propagated.grav.Mx(Mx, dx + idx);
        for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) J[i][j] = Nx[i][0] * Mx[0][j] + Nx[i][1] * Mx[1][j] + Nx[i][2] * Mx[2][j];
        double tv[2];
        for (int r = 0; r < 2; r++) tv[r] = J[r][0] * dx_new[idx] + J[r][1] * dx_new[idx + 1];
        for (int r = 0; r < 2; r++) dx_new[idx + r] = tv[r];
        reproject_rows<2>(P_, idx, J, n);
        reproject_cols<2>(P_, idx, J);
      }

      double HTH[12][12], HTh[12];
      for (int i = 0; i < 12; i++) { HTh[i] = 0; for (int j = 0; j < 12; j++) HTH[i][j] = 0; }   // a-note 5

      if (n > M) {                                              // :1701-1709
        // K = P Hc^T (Hc P Hc^T / R + I)^-1 / R ,  Hc = [H 0]
        std::vector<double> PHt((size_t)n * M), S((size_t)M * M), Sinv((size_t)M * M), K((size_t)n * M);
        for (int i = 0; i < n; i++) for (int j = 0; j < M; j++) { double s = 0; for (int k = 0; k < 12; k++) s += P_[i][k] * meas.h_x[(size_t)j * 12 + k]; PHt[(size_t)i * M + j] = s; }
        for (int i = 0; i < M; i++) for (int j = 0; j < M; j++) { double s = 0; for (int k = 0; k < 12; k++) s += meas.h_x[(size_t)i * 12 + k] * PHt[(size_t)k * M + j]; S[(size_t)i * M + j] = s / R + (i == j ? 1.0 : 0.0); }
        if (M > 0) lu_inverse(M, S.data(), Sinv.data());
        for (int i = 0; i < n; i++) for (int j = 0; j < M; j++) { double s = 0; for (int k = 0; k < M; k++) s += PHt[(size_t)i * M + k] * Sinv[(size_t)k * M + j]; K[(size_t)i * M + j] = s / R; }
        for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < M; k++) s += K[(size_t)i * M + k] * meas.h[k]; K_h[i] = s; }
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
          double s = 0;
          if (j < 12) for (int k = 0; k < M; k++) s += K[(size_t)i * M + k] * meas.h_x[(size_t)k * 12 + j];
          K_x[i][j] = s;
        }
      } else {                                                  // :1722-1729
        for (int m = 0; m < M; m++) {
          const double* row = &meas.h_x[(size_t)m * 12];
          for (int i = 0; i < 12; i++) {
            for (int j = 0; j < 12; j++) HTH[i][j] += row[i] * row[j];
            HTh[i] += row[i] * meas.h[m];
          }
        }
        static thread_local double PR[NDOF * NDOF], P_temp[NDOF * NDOF], P_inv[NDOF * NDOF];
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) PR[i * n + j] = P_[i][j] / R;
        lu_inverse(n, PR, P_temp);
        for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) P_temp[i * n + j] += HTH[i][j];
        lu_inverse(n, P_temp, P_inv);
        for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < 12; k++) s += P_inv[i * n + k] * HTh[k]; K_h[i] = s; }
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
          double s = 0;
          if (j < 12) for (int k = 0; k < 12; k++) s += P_inv[i * n + k] * HTH[k][j];
          K_x[i][j] = s;
        }
      }

      double dx_[NDOF];                                         // :1733
      for (int i = 0; i < n; i++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += (K_x[i][k] - (i == k ? 1.0 : 0.0)) * dx_new[k];
        dx_[i] = K_h[i] + s;
      }

      // degeneracy :1736-1744: Eigen::EigenSolver of HTH[0:6,0:6] (restated in rl_linalg.h: the eigenpairs come in the order and
      // with the signs of Eigen's Hessenberg + Francis QR + back substitution), `.real()` of values and vectors, the row-zeroing
      // "projector" VEPs^-1 * selVEPs applied as Eigen evaluates the product: (VEPs^-1 * selVEPs) * dx_.head(6)
      double S6[36], w[6], wim[6], V[6][6];
      for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) S6[i * 6 + j] = HTH[i][j];
      eigen_solver6(S6, w, wim, V);
      double prod = w[0];
      for (int i = 1; i < 6; i++) prod *= w[i];
      if (prod < 1e-20) for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) V[i][j] = (i == j) ? 1.0 : 0.0;
      double sel[6][6];
      for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) sel[i][j] = V[i][j];
      for (int v = 0; v < 6; v++) if (w[v] < D) for (int j = 0; j < 6; j++) sel[v][j] *= 0;   // ROW v zeroed (quirk)
      double Vflat[36], Vinv[36];
      for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) Vflat[i * 6 + j] = V[i][j];
      lu_inverse(6, Vflat, Vinv);
      double dx_nd[NDOF];
      for (int i = 0; i < n; i++) dx_nd[i] = dx_[i];
      {
        double Pm[6][6];
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { double s = 0; for (int k = 0; k < 6; k++) s += Vinv[i * 6 + k] * sel[k][j]; Pm[i][j] = s; }
        for (int i = 0; i < 6; i++) { double s = 0; for (int k = 0; k < 6; k++) s += Pm[i][k] * dx_[k]; dx_nd[i] = s; }
      }

      x_.boxplus(dx_nd);                                        // :1747
      bool converge = true;
      for (int i = 0; i < n; i++) if (std::fabs(dx_[i]) > limit[i]) { converge = false; break; }
      if (converge) t++;

      IterLog lg;
      lg.M = M;
      for (int i = 0; i < 12; i++) { lg.HTh[i] = HTh[i]; for (int j = 0; j < 12; j++) lg.HTH[i * 12 + j] = HTH[i][j]; }
      for (int i = 0; i < n; i++) lg.dx[i] = dx_[i];
      x_.to_flat(lg.x_after);
      log.push_back(lg);

      if (t > 1 || it == maximum_iter - 1) {                    // :1764-1820
        static thread_local double L_[NDOF][NDOF];
        std::memcpy(L_, P_, sizeof(P_));
        for (int s = 0; s < 2; s++) {
          int idx = so3_idx[s];
          double A[3][3], J[3][3];
          A_matrix(dx_ + idx, A);
          for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[i][j] = A[j][i];
          for (int i = 0; i < n; i++) {       // L rows from P rows
            for (int r = 0; r < 3; r++) L_[idx + r][i] = J[r][0] * P_[idx][i] + J[r][1] * P_[idx + 1][i] + J[r][2] * P_[idx + 2][i];
          }
          reproject_rows<3>(K_x, idx, J, 12);
          reproject_cols<3>(L_, idx, J);
          reproject_cols<3>(P_, idx, J);
        }
        {
          int idx = 21;
          double Nx[2][3], Mx[3][2], J[2][2];
          x_.grav.Nx_yy(Nx);
          x_propagated.grav.Mx(Mx, dx_ + idx);
          for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) J[i][j] = Nx[i][0] * Mx[0][j] + Nx[i][1] * Mx[1][j] + Nx[i][2] * Mx[2][j];
          for (int i = 0; i < n; i++)
            for (int r = 0; r < 2; r++) L_[idx + r][i] = J[r][0] * P_[idx][i] + J[r][1] * P_[idx + 1][i];
          reproject_rows<2>(K_x, idx, J, 12);
          reproject_cols<2>(L_, idx, J);
          reproject_cols<2>(P_, idx, J);
        }
        static thread_local double Pn[NDOF][NDOF];
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
          double s = 0;
          for (int k = 0; k < 12; k++) s += K_x[i][k] * P_[k][j];
          Pn[i][j] = L_[i][j] - s;
        }
        std::memcpy(P_, Pn, sizeof(P_));
        return;
      }
    }
  }
};

}  // namespace oracle
