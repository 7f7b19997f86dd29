// fast_limo_amd/csrc/host/flimo_ikfom.hpp
// Host-side (float64) iterated error-state Kalman filter on the 23-dof fast_LIMO state manifold.
// This is the product's replacement for the vendored IKFoM toolkit on the hot path:
//   state_ikfom                         reference include/IKFoM/use-ikfom.hpp:12-21
//   boxplus / boxminus / oplus          IKFoM_toolkit/mtk/build_manifold.hpp:192-200,
//                                       mtk/types/SOn.hpp:233-240,284-297, mtk/types/S2.hpp:129-167
//   A_matrix, exp, log                  mtk/src/mtkmath.hpp:143-174,236-290
//   S2 charts (Bx, Nx, Mx)              mtk/types/S2.hpp:179-281 with S2<double,98090,10000,1>
//   process model f, df_dx, df_dw       include/IKFoM/use-ikfom.cpp:43-84
//   esekf::predict                      IKFoM_toolkit/esekfom/esekfom.hpp:279-384
//   esekf::update_iterated_dyn_share_modified   esekfom.hpp:1620-1823
// The measurement seam is REDUCED: instead of the M x 12 Jacobian the plug-in returns
// H^T H (12x12), H^T h (12) and M, which is all the M >= 23 branch uses (esekfom.hpp:1722-1729);
// the dense H is requested through a second callback only when M < 23 (:1701-1709).
// The reference's observable quirks are kept: `scalar(1/2)` == 0 in predict and S2_Mx, HTH := 0
// when M < 23, row-zeroing degeneracy projector over Eigen::EigenSolver's eigenpair order (restated), convergence tested on the
// un-projected step.
// The 23x23 algebra is ~25 kflop per pass.  Two layouts of the loop: this filter calls the measurement once per iteration (one
// launch -> result round trip each), or the device runs the iterations back to back with the same algebra between them
// (Esekf::device_chain -> flimo_update_chain, csrc/hip/flimo_ieskf.h) and this filter finishes the iteration whose covariance update
// is due from the sums handed back.  inverse_gj / solve_gj below are the elimination the device's one-wave solver runs: the two
// layouts take the same steps.
#pragma once
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

namespace flimo_host {

// ---------------------------------------------------------------------------------------------
// small dense matrices
// ---------------------------------------------------------------------------------------------
template <int R, int C>
struct Mat {
  double a[R][C];
  static Mat zero() { Mat m; std::memset(m.a, 0, sizeof(m.a)); return m; }
  static Mat identity() { Mat m = zero(); for (int i = 0; i < (R < C ? R : C); i++) m.a[i][i] = 1.0; return m; }
  double& operator()(int i, int j) { return a[i][j]; }
  double operator()(int i, int j) const { return a[i][j]; }
  Mat<C, R> T() const { Mat<C, R> t; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) t.a[j][i] = a[i][j]; return t; }
};
template <int R, int K, int C>
inline Mat<R, C> operator*(const Mat<R, K>& A, const Mat<K, C>& B) {
  Mat<R, C> o;
  for (int i = 0; i < R; i++)
    for (int j = 0; j < C; j++) {
      double s = 0.0;
      for (int k = 0; k < K; k++) s += A.a[i][k] * B.a[k][j];
      o.a[i][j] = s;
    }
  return o;
}
template <int R, int C>
inline Mat<R, C> operator+(const Mat<R, C>& A, const Mat<R, C>& B) { Mat<R, C> o; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) o.a[i][j] = A.a[i][j] + B.a[i][j]; return o; }
template <int R, int C>
inline Mat<R, C> operator-(const Mat<R, C>& A, const Mat<R, C>& B) { Mat<R, C> o; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) o.a[i][j] = A.a[i][j] - B.a[i][j]; return o; }
template <int R, int C>
inline Mat<R, C> operator*(double s, const Mat<R, C>& A) { Mat<R, C> o; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) o.a[i][j] = s * A.a[i][j]; return o; }
typedef Mat<3, 1> Vec3;
typedef Mat<3, 3> Mat3;

// LU with partial pivoting, runtime size (row-major n x n).  Returns false on a zero pivot.
bool inverse_lu(int n, const double* A, double* Ainv);
// Gauss-Jordan with partial pivoting, rows marked instead of exchanged (n <= 32): the order of operations of the device filter
bool inverse_gj(int n, const double* A, double* Ainv);
bool solve_gj(int n, const double* A, const double* v, double* u);   // A u = v by the same elimination
template <int N>
inline bool inverse(const Mat<N, N>& A, Mat<N, N>& out) { return inverse_lu(N, &A.a[0][0], &out.a[0][0]); }
// Eigen::EigenSolver<Matrix<double,6,6>> as esekfom.hpp:1736-1738 uses it: eigenvalues (real, imaginary part) in the solver's order
// -- the diagonal of the real Schur form from the top -- and the real parts of its normalised eigenvectors as columns of V
void eigen_solver6(const Mat<6, 6>& A, double wr[6], double wi[6], Mat<6, 6>& V);

// ---------------------------------------------------------------------------------------------
// manifold pieces
// ---------------------------------------------------------------------------------------------
struct Quat { double x = 0, y = 0, z = 0, w = 1; };
Quat quat_mul(const Quat& a, const Quat& b);
inline Quat quat_conj(const Quat& q) { Quat r; r.x = -q.x; r.y = -q.y; r.z = -q.z; r.w = q.w; return r; }
Mat3 quat_to_rot(const Quat& q);
Quat rot_to_quat(const Mat3& R);
Vec3 quat_rotate(const Quat& q, const Vec3& v);
Mat3 hat(const Vec3& v);
Mat3 A_matrix(const Vec3& v);                  // mtkmath.hpp:236-247
Quat so3_exp(const Vec3& v, double scale = 1.0);   // SOn.hpp:284-288
Vec3 so3_log(const Quat& q);                   // SOn.hpp:293-297

static const double kS2Length = 98090.0 / 10000.0;

struct S2 {                                    // S2<double,98090,10000,1>
  Vec3 vec;
  S2();
  S2(double x, double y, double z);
  Mat<3, 2> Bx() const;
  void boxplus(const double d[2], double scale = 1.0);
  void oplus(const Vec3& d, double scale = 1.0);
  void boxminus(double out[2], const S2& other) const;
  Mat<2, 3> Nx_yy() const;
  Mat<3, 2> Mx(const double delta[2]) const;
};

static const int kDof = 23, kDim = 24;

struct StateIkfom {
  Vec3 pos;
  Quat rot;
  Quat offset_R_L_I;
  Vec3 offset_T_L_I;
  Vec3 vel, bg, ba;
  S2 grav;
  StateIkfom();
  void boxplus(const double d[kDof]);
  void oplus(const double f[kDim], double dt);
  void boxminus(double out[kDof], const StateIkfom& o) const;
  void to_flat(double x26[26]) const;
  void from_flat(const double x26[26]);
};

struct InputIkfom { Vec3 acc, gyro; };

// ---------------------------------------------------------------------------------------------
// the filter
// ---------------------------------------------------------------------------------------------
struct DenseMeas {        // only for the M < 23 branch
  std::vector<double> H;  // M x 12 row-major
  std::vector<double> h;  // M
};
struct PassLog {
  int M;
  double HTH[144], HTh[12], dx[kDof], x_after[26];
};

// The whole update run by the measurement side in one go (the GPU library's flimo_update_chain): what comes back
struct ReducedMeas {      // what the measurement plug-in returns per pass
  int M = 0;
  double HTH[144];
  double HTh[12];
};
struct ChainResult {
  int status = 0;         // 0 declined (nothing ran), 2 handed back (resume the loop at it_next with t, from x)
  int passes = 0;         // outer iterations completed
  int it_next = -1, t = 0;
  double x[26];
  bool have_meas = false; // the pass of iteration it_next has run: `meas` holds its sums (the loop does not repeat it)
  ReducedMeas meas;
};

class Esekf {
 public:
  typedef Mat<kDof, kDof> Cov;
  // Optional: the iterations of the update on the measurement side (device).  Called once at the start of
  // update_iterated_dyn_share_modified with the propagated state; may decline, and hands the loop back at the iteration whose
  // covariance update is due (with that iteration's sums) or earlier (M < 23, distance ties, degenerate H^T H:
  // esekfom.hpp:1701-1709,1736-1744 stay on the host).  Completed iterations are appended to `log` by the callee when keep_log is set.
  std::function<void(const double x26[26], const Cov& P, const double* limits, double R, double D, int max_iter, ChainResult& out)> device_chain;
  // Set by a measurement callback whose pass failed (a GPU error, a timeout): the update is abandoned -- x_ and P_ are restored to
  // the propagated values and update_iterated_dyn_share_modified returns (the reference's plug-in cannot fail, Mapper.cpp:59-86)
  bool failed = false;
  std::function<void(const StateIkfom&, ReducedMeas&)> h_reduced;   // replaces h_dyn_share (esekfom.hpp:128)
  // the same with work of the filter to run while the measurement is in flight (called at most once; optional: when unset, or
  // when the plug-in did not get to call it, the filter runs the work itself after h_reduced)
  std::function<void(const StateIkfom&, ReducedMeas&, const std::function<void()>&)> h_reduced_overlap;
  std::function<void(DenseMeas&)> h_dense;                          // dense rows of the SAME pass
  // called once when update_iterated_dyn_share_modified leaves, whichever way: a plug-in that queued a pass ahead of the loop's
  // next iteration (pipelined host loop, flimo_c.h: flimo_pass_pipeline_end) lets it go
  std::function<void()> h_update_end;
  // called before the measurement of the loop's last possible iteration (i == maximum_iter - 1): a plug-in that queues passes ahead
  // (flimo_pass_pipeline_last) need not queue one behind it
  std::function<void()> h_last_iteration;
  std::vector<PassLog> log;
  bool keep_log = false;
  // true: literal two-inverse form of esekfom.hpp:1722-1729 and unconditional eigen-decomposition;
  // false (default): the same formula through the block-inverse identity -- one 12x12 system, as accurate as the literal form --
  // + Cholesky shortcut
  bool reference_solve = false;

  Esekf();
  void init(int maximum_iteration, const double* limits);            // init_dyn_share :237-254
  void predict(double dt, const Mat<12, 12>& Q, const InputIkfom& in);   // :279-384
  void update_iterated_dyn_share_modified(double R, double D);       // :1620-1823
  const StateIkfom& get_x() const { return x_; }
  const Cov& get_P() const { return P_; }
  void change_x(const StateIkfom& s) { x_ = s; }
  void change_P(const Cov& P) { P_ = P; }

 private:
  StateIkfom x_;
  Cov P_;
  int maximum_iter_ = 0;
  double limit_[kDof];
};

}  // namespace flimo_host
