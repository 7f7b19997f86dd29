// oracle/rl_linalg.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Small fixed-size linear algebra used by the CPU restatement ("oracle") of the
// fast_LIMO registration hot path.  The reference does all of this through
// Eigen3, which is NOT available in this image, so every Eigen routine that the
// hot path calls is restated here from its documented algorithm:
//
//   * float 3/4-vectors, 3x3 / 4x4 float matrices, float quaternion
//       (Eigen::Quaternionf::toRotationMatrix / operator* / _transformVector /
//        Quaternionf(Matrix3f); call sites: reference
//        include/fast_limo/Objects/State.cpp:106-111,139,147)
//   * column-pivoted Householder QR least squares for the 5x3 plane fit
//       (Eigen::ColPivHouseholderQR::compute/solve; call site
//        include/fast_limo/Objects/Plane.cpp:95)
//   * partial-pivot LU inverse in double
//       (Eigen::Matrix<double,23,23>::inverse(); call sites
//        include/IKFoM/IKFoM_toolkit/esekfom/esekfom.hpp:1706,1722,1726,1744)
//   * Eigen::EigenSolver<Matrix<double,6,6>> (esekfom.hpp:1736): Householder Hessenberg reduction, Francis double-shift QR,
//        eigenvectors by back substitution -- eigen_solver6 below
//
// PARITY UNPINNED: Eigen's expression templates fix an evaluation order that can
// only be confirmed by compiling against Eigen.  Where the order is known from the
// Eigen 3.3 sources it is followed (3-element reductions are  a0 + (a1 + a2);
// 4x4*4 products accumulate column by column left to right); elsewhere plain
// left-to-right loops are used.  All float code must be compiled with
// -ffp-contract=off (the reference is built without FMA: CMakeLists.txt:4-5,17-21).
#pragma once
#include <cmath>
#include <limits>
#include <algorithm>
#include <cstring>
#include <cstdint>
#include <algorithm>

namespace oracle {

// ----------------------------------------------------------------------------------------
// float32 geometry
// ----------------------------------------------------------------------------------------
struct V3f {
  float x, y, z;
  V3f() : x(0), y(0), z(0) {}
  V3f(float a, float b, float c) : x(a), y(b), z(c) {}
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3f operator+(const V3f& a, const V3f& b) { return V3f(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3f operator-(const V3f& a, const V3f& b) { return V3f(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3f operator*(float s, const V3f& a) { return V3f(s * a.x, s * a.y, s * a.z); }
inline V3f operator/(const V3f& a, float s) { return V3f(a.x / s, a.y / s, a.z / s); }

// Eigen redux over 3 coefficients: func(c0, func(c1, c2))  (Eigen/src/Core/Redux.h,
// redux_novec_unroller with HalfLength = 3/2 = 1).
inline float sum3(float a, float b, float c) { return a + (b + c); }
inline float dot3(const V3f& a, const V3f& b) { return sum3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline float sqnorm3(const V3f& a) { return sum3(a.x * a.x, a.y * a.y, a.z * a.z); }
inline float norm3(const V3f& a) { return std::sqrt(sqnorm3(a)); }
inline V3f cross3(const V3f& a, const V3f& b) {
  return V3f(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

struct M3f {
  float m[3][3];  // m[row][col]
  static M3f identity() {
    M3f r;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r.m[i][j] = (i == j) ? 1.f : 0.f;
    return r;
  }
  M3f transpose() const {
    M3f r;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r.m[i][j] = m[j][i];
    return r;
  }
};
// coefficient-based lazy product: coeff(i) = (row_i .* v).sum()
inline V3f mul(const M3f& A, const V3f& v) {
  return V3f(sum3(A.m[0][0] * v.x, A.m[0][1] * v.y, A.m[0][2] * v.z),
             sum3(A.m[1][0] * v.x, A.m[1][1] * v.y, A.m[1][2] * v.z),
             sum3(A.m[2][0] * v.x, A.m[2][1] * v.y, A.m[2][2] * v.z));
}
inline M3f mul(const M3f& A, const M3f& B) {
  M3f r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      r.m[i][j] = sum3(A.m[i][0] * B.m[0][j], A.m[i][1] * B.m[1][j], A.m[i][2] * B.m[2][j]);
  return r;
}

struct V4f {
  float v[4];
  V4f() { v[0] = v[1] = v[2] = v[3] = 0.f; }
  V4f(float a, float b, float c, float d) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
};
struct M4f {
  float m[4][4];  // m[row][col]
  static M4f identity() {
    M4f r;
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.f : 0.f;
    return r;
  }
};
// 4x4 * 4: packet product, res = ((col0*v0 + col1*v1) + col2*v2) + col3*v3
inline V4f mul(const M4f& A, const V4f& x) {
  V4f r;
  for (int i = 0; i < 4; i++) {
    float acc = A.m[i][0] * x.v[0];
    acc = acc + A.m[i][1] * x.v[1];
    acc = acc + A.m[i][2] * x.v[2];
    acc = acc + A.m[i][3] * x.v[3];
    r.v[i] = acc;
  }
  return r;
}
inline M4f mul(const M4f& A, const M4f& B) {
  M4f r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float acc = A.m[i][0] * B.m[0][j];
      acc = acc + A.m[i][1] * B.m[1][j];
      acc = acc + A.m[i][2] * B.m[2][j];
      acc = acc + A.m[i][3] * B.m[3][j];
      r.m[i][j] = acc;
    }
  return r;
}

template <typename T>
struct QuatT {
  T x, y, z, w;
  QuatT() : x(0), y(0), z(0), w(1) {}
  QuatT(T w_, T x_, T y_, T z_) : x(x_), y(y_), z(z_), w(w_) {}  // Eigen ctor order (w,x,y,z)
  QuatT conjugate() const { return QuatT(w, -x, -y, -z); }
};
typedef QuatT<float> Quatf;
typedef QuatT<double> Quatd;

// Eigen::QuaternionBase::operator* (generic scalar path)
template <typename T>
inline QuatT<T> qmul(const QuatT<T>& a, const QuatT<T>& b) {
  return QuatT<T>(a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
                  a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
                  a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
                  a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x);
}

// Eigen::QuaternionBase::toRotationMatrix
template <typename T>
inline void quat_to_rot(const QuatT<T>& q, T R[3][3]) {
  const T tx = T(2) * q.x, ty = T(2) * q.y, tz = T(2) * q.z;
  const T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0][0] = T(1) - (tyy + tzz);
  R[0][1] = txy - twz;
  R[0][2] = txz + twy;
  R[1][0] = txy + twz;
  R[1][1] = T(1) - (txx + tzz);
  R[1][2] = tyz - twx;
  R[2][0] = txz - twy;
  R[2][1] = tyz + twx;
  R[2][2] = T(1) - (txx + tyy);
}
inline M3f quat_to_M3f(const Quatf& q) {
  M3f r;
  quat_to_rot<float>(q, r.m);
  return r;
}

// Eigen quaternion-from-rotation-matrix (Eigen/src/Geometry/Quaternion.h,
// quaternionbase_assign_impl<Other,3,3>)
template <typename T>
inline QuatT<T> rot_to_quat(const T m[3][3]) {
  QuatT<T> q;
  T t = m[0][0] + (m[1][1] + m[2][2]);   // trace() = diagonal().sum(): 3-coefficient redux
  if (t > T(0)) {
    t = std::sqrt(t + T(1.0));
    q.w = T(0.5) * t;
    t = T(0.5) / t;
    q.x = (m[2][1] - m[1][2]) * t;
    q.y = (m[0][2] - m[2][0]) * t;
    q.z = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + T(1.0));
    T c[3];
    c[i] = T(0.5) * t;
    t = T(0.5) / t;
    q.w = (m[k][j] - m[j][k]) * t;
    c[j] = (m[j][i] + m[i][j]) * t;
    c[k] = (m[k][i] + m[i][k]) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}

// Eigen::QuaternionBase::_transformVector
inline V3f quat_rotate(const Quatf& q, const V3f& v) {
  V3f qv(q.x, q.y, q.z);
  V3f uv = cross3(qv, v);
  uv = uv + uv;
  V3f wuv = q.w * uv;
  return (v + wuv) + cross3(qv, uv);
}

// ----------------------------------------------------------------------------------------
// float32 5x3 column-pivoted Householder QR least squares (Plane.cpp:95).
// Follows Eigen 3.3 ColPivHouseholderQR::computeInPlace + _solve_impl step by step.
// A is row-major [rows][3]; b has `rows` entries.  Returns x (3 entries).
// ----------------------------------------------------------------------------------------
inline void colpiv_qr_solve_nx3(int rows, const float* A_in, const float* b_in, float x_out[3]) {
  const int cols = 3;
  const int size = rows < cols ? rows : cols;
  float qr[16][3];
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < cols; j++) qr[i][j] = A_in[i * 3 + j];

  float hCoeffs[3] = {0, 0, 0};
  int transp[3] = {0, 1, 2};
  float normsUpdated[3], normsDirect[3];
  for (int k = 0; k < cols; k++) {
    float s = 0.f;
    for (int i = 0; i < rows; i++) s = s + qr[i][k] * qr[i][k];
    normsDirect[k] = std::sqrt(s);
    normsUpdated[k] = normsDirect[k];
  }
  const float eps = 1.1920929e-07f;  // NumTraits<float>::epsilon()
  float maxn = normsUpdated[0];
  for (int k = 1; k < cols; k++) if (normsUpdated[k] > maxn) maxn = normsUpdated[k];
  float th = maxn * eps;
  const float threshold_helper = (th * th) / float(rows);
  const float norm_downdate_threshold = std::sqrt(eps);

  int nonzero_pivots = size;
  float maxpivot = 0.f;

  for (int k = 0; k < size; k++) {
    // column of biggest (updated) norm among k..cols-1, first one wins on ties (maxCoeff)
    int big = k;
    float bign = normsUpdated[k];
    for (int j = k + 1; j < cols; j++)
      if (normsUpdated[j] > bign) { bign = normsUpdated[j]; big = j; }
    float big_sq = bign * bign;
    if (nonzero_pivots == size && big_sq < threshold_helper * float(rows - k)) nonzero_pivots = k;

    transp[k] = big;
    if (k != big) {
      for (int i = 0; i < rows; i++) std::swap(qr[i][k], qr[i][big]);
      std::swap(normsUpdated[k], normsUpdated[big]);
      std::swap(normsDirect[k], normsDirect[big]);
    }

    // makeHouseholderInPlace on qr[k..rows-1][k]
    float tailSq = 0.f;
    for (int i = k + 1; i < rows; i++) tailSq = tailSq + qr[i][k] * qr[i][k];
    float c0 = qr[k][k];
    float tau, beta;
    const float tol = 1.17549435e-38f;  // numeric_limits<float>::min()
    if ((rows - k) == 1 || tailSq <= tol) {
      if ((rows - k) == 1) tailSq = 0.f;
      tau = 0.f;
      beta = c0;
      for (int i = k + 1; i < rows; i++) qr[i][k] = 0.f;
    } else {
      beta = std::sqrt(c0 * c0 + tailSq);
      if (c0 >= 0.f) beta = -beta;
      float denom = c0 - beta;
      for (int i = k + 1; i < rows; i++) qr[i][k] = qr[i][k] / denom;
      tau = (beta - c0) / beta;
    }
    hCoeffs[k] = tau;
    qr[k][k] = beta;
    if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);

    // applyHouseholderOnTheLeft to qr[k..rows-1][k+1..cols-1]
    int brows = rows - k;
    if (k + 1 < cols) {
      if (brows == 1) {
        for (int j = k + 1; j < cols; j++) qr[k][j] = qr[k][j] * (1.f - tau);
      } else if (tau != 0.f) {
        for (int j = k + 1; j < cols; j++) {
          float tmp = 0.f;
          for (int i = k + 1; i < rows; i++) tmp = tmp + qr[i][k] * qr[i][j];  // essential^T * bottom
          tmp = tmp + qr[k][j];
          qr[k][j] = qr[k][j] - tau * tmp;
          for (int i = k + 1; i < rows; i++) qr[i][j] = qr[i][j] - (tau * qr[i][k]) * tmp;
        }
      }
    }

    // LAPACK-style column-norm downdate (lawn176)
    for (int j = k + 1; j < cols; j++) {
      if (normsUpdated[j] != 0.f) {
        float temp = std::fabs(qr[k][j]) / normsUpdated[j];
        temp = (1.f + temp) * (1.f - temp);
        temp = temp < 0.f ? 0.f : temp;
        float ratio = normsUpdated[j] / normsDirect[j];
        float temp2 = temp * (ratio * ratio);
        if (temp2 <= norm_downdate_threshold) {
          float s = 0.f;
          for (int i = k + 1; i < rows; i++) s = s + qr[i][j] * qr[i][j];
          normsDirect[j] = std::sqrt(s);
          normsUpdated[j] = normsDirect[j];
        } else {
          normsUpdated[j] = normsUpdated[j] * std::sqrt(temp);
        }
      }
    }
  }

  // column permutation P = T_0 T_1 ... (colsPermutation.applyTranspositionOnTheRight)
  int perm[3] = {0, 1, 2};
  for (int k = 0; k < size; k++) std::swap(perm[k], perm[transp[k]]);

  // ---- solve ----
  if (nonzero_pivots == 0) {
    x_out[0] = x_out[1] = x_out[2] = 0.f;
    return;
  }
  float c[16];
  for (int i = 0; i < rows; i++) c[i] = b_in[i];
  // c = Q^T c : apply H_0, H_1, ... H_{nzp-1} in order
  for (int k = 0; k < nonzero_pivots; k++) {
    int brows = rows - k;
    float tau = hCoeffs[k];
    if (brows == 1) {
      c[k] = c[k] * (1.f - tau);
    } else if (tau != 0.f) {
      float tmp = 0.f;
      for (int i = k + 1; i < rows; i++) tmp = tmp + qr[i][k] * c[i];
      tmp = tmp + c[k];
      c[k] = c[k] - tau * tmp;
      for (int i = k + 1; i < rows; i++) c[i] = c[i] - (tau * qr[i][k]) * tmp;
    }
  }
  // back substitution on the upper-left nzp x nzp triangle
  for (int i = nonzero_pivots - 1; i >= 0; i--) {
    float s = c[i];
    for (int j = i + 1; j < nonzero_pivots; j++) s = s - qr[i][j] * c[j];
    c[i] = s / qr[i][i];
  }
  float xs[3] = {0.f, 0.f, 0.f};
  for (int i = 0; i < nonzero_pivots; i++) xs[perm[i]] = c[i];
  for (int i = nonzero_pivots; i < cols; i++) xs[perm[i]] = 0.f;
  x_out[0] = xs[0]; x_out[1] = xs[1]; x_out[2] = xs[2];
}

// ----------------------------------------------------------------------------------------
// float64 dense helpers (row-major, runtime n <= NMAX)
// ----------------------------------------------------------------------------------------
// Partial-pivot LU inverse (stands in for Eigen's PartialPivLU-based inverse()).
// A, Ainv are n x n row-major.  Returns false when a zero pivot is met.
inline bool lu_inverse(int n, const double* A, double* Ainv) {
  const int NMAX = 64;
  if (n > NMAX) return false;
  static thread_local double lu[NMAX * NMAX];
  int piv[NMAX];
  for (int i = 0; i < n * n; i++) lu[i] = A[i];
  for (int i = 0; i < n; i++) piv[i] = i;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(lu[k * n + k]);
    for (int i = k + 1; i < n; i++) {
      double v = std::fabs(lu[i * n + k]);
      if (v > best) { best = v; p = i; }
    }
    if (best == 0.0) return false;
    if (p != k) {
      for (int j = 0; j < n; j++) std::swap(lu[k * n + j], lu[p * n + j]);
      std::swap(piv[k], piv[p]);
    }
    double d = lu[k * n + k];
    for (int i = k + 1; i < n; i++) {
      double f = lu[i * n + k] / d;
      lu[i * n + k] = f;
      for (int j = k + 1; j < n; j++) lu[i * n + j] -= f * lu[k * n + j];
    }
  }
  // solve for each unit vector
  for (int col = 0; col < n; col++) {
    double y[NMAX];
    for (int i = 0; i < n; i++) {
      double s = (piv[i] == col) ? 1.0 : 0.0;
      for (int j = 0; j < i; j++) s -= lu[i * n + j] * y[j];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      for (int j = i + 1; j < n; j++) s -= lu[i * n + j] * y[j];
      y[i] = s / lu[i * n + i];
    }
    for (int i = 0; i < n; i++) Ainv[i * n + col] = y[i];
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Eigen::EigenSolver<Matrix<double,6,6>> restated (esekfom.hpp:1736-1738: `.eigenvalues().real()`, `.eigenvectors().real()`).
//
// Third-party arithmetic absent from /root/reference: Eigen3, version unpinned by the reference (find_package(Eigen3),
// CMakeLists.txt:14); the reference's Docker base (osrf/ros:noetic) ships Eigen 3.3.7.  The degeneracy projector zeroes ROW i of
// the eigenvector matrix when eigenVALUE i is below the threshold (esekfom.hpp:1741), so what reaches the state is decided by
// the ORDER in which the solver returns the eigenpairs and by the SIGN of every eigenvector -- both are properties of the
// algorithm, restated here from Eigen's published sources (Eigenvalues/HessenbergDecomposition.h, RealSchur.h, EigenSolver.h,
// Householder/Householder.h, Jacobi/Jacobi.h of 3.3.7), step for step:
//   1. scale by the largest |coefficient|, Householder reduction to Hessenberg form (makeHouseholder: beta = -sign(c0) |x|,
//      essential = tail / (c0 - beta), tau = (beta - c0) / beta), Q accumulated backwards from the identity;
//   2. Francis double-shift QR on the Hessenberg matrix (RealSchur::computeFromHessenberg): deflation test
//      |T(i,i-1)| <= max(eps (|T(i-1,i-1)| + |T(i,i)|), considerAsZero), one root / two roots (splitOffTwoRows with a Jacobi
//      rotation) / Francis step with the shift of the trailing 2x2 block and the exceptional shifts at iterations 10 and 30;
//      eigenvalues are read off the diagonal of T from top to bottom -- THIS fixes their order;
//   3. eigenvectors by back substitution on T (EigenSolver::doComputeEigenvectors), multiplied by Q, each column normalised.
// Floating-point sums inside a step (Eigen vectorises dot products and norms; the order depends on alignment and packet size) are
// taken left to right here, so the last bits may differ from a given Eigen build; order and signs do not depend on them.
// Row-major 6x6 in, wr / wi eigenvalues, V = real parts of the normalised eigenvectors as COLUMNS (V[r][c]).
// ---------------------------------------------------------------------------------------------------------------
namespace eig6 {
constexpr int N = 6;
struct Hh { double tau, beta; };
// Householder of x[0..m): x[0] becomes beta, x[1..m) the essential part (MatrixBase::makeHouseholderInPlace)
inline Hh make_householder(double* x, int m, int stride) {
  double tail2 = 0.0;
  for (int i = 1; i < m; i++) tail2 += x[i * stride] * x[i * stride];
  const double c0 = x[0];
  Hh h;
  if (tail2 <= std::numeric_limits<double>::min()) {
    h.tau = 0.0; h.beta = c0;
    for (int i = 1; i < m; i++) x[i * stride] = 0.0;
  } else {
    double beta = std::sqrt(c0 * c0 + tail2);
    if (c0 >= 0.0) beta = -beta;
    for (int i = 1; i < m; i++) x[i * stride] = x[i * stride] / (c0 - beta);
    h.tau = (beta - c0) / beta;
    h.beta = beta;
  }
  return h;
}
// M(r0.., c0..) of `rows` x `cols` <- (I - tau v v^T) M, v = [1; ess]      (applyHouseholderOnTheLeft)
inline void apply_left(double M[N][N], int r0, int c0, int rows, int cols, const double* ess, double tau) {
  if (rows == 1) { for (int j = 0; j < cols; j++) M[r0][c0 + j] *= (1.0 - tau); return; }
  if (tau == 0.0) return;
  for (int j = 0; j < cols; j++) {
    double tmp = 0.0;
    for (int i = 1; i < rows; i++) tmp += ess[i - 1] * M[r0 + i][c0 + j];
    tmp += M[r0][c0 + j];
    M[r0][c0 + j] -= tau * tmp;
    for (int i = 1; i < rows; i++) M[r0 + i][c0 + j] -= tau * ess[i - 1] * tmp;
  }
}
// M(r0.., c0..) <- M (I - tau v v^T)                                         (applyHouseholderOnTheRight)
inline void apply_right(double M[N][N], int r0, int c0, int rows, int cols, const double* ess, double tau) {
  if (cols == 1) { for (int i = 0; i < rows; i++) M[r0 + i][c0] *= (1.0 - tau); return; }
  if (tau == 0.0) return;
  for (int i = 0; i < rows; i++) {
    double tmp = 0.0;
    for (int j = 1; j < cols; j++) tmp += M[r0 + i][c0 + j] * ess[j - 1];
    tmp += M[r0 + i][c0];
    M[r0 + i][c0] -= tau * tmp;
    for (int j = 1; j < cols; j++) M[r0 + i][c0 + j] -= tau * tmp * ess[j - 1];
  }
}
struct Givens { double c, s; };
inline Givens make_givens(double p, double q) {                             // JacobiRotation::makeGivens (real)
  Givens g;
  if (q == 0.0) { g.c = p < 0 ? -1.0 : 1.0; g.s = 0.0; }
  else if (p == 0.0) { g.c = 0.0; g.s = q < 0 ? 1.0 : -1.0; }
  else if (std::fabs(p) > std::fabs(q)) {
    const double t = q / p;
    double u = std::sqrt(1.0 + t * t);
    if (p < 0) u = -u;
    g.c = 1.0 / u; g.s = -t * g.c;
  } else {
    const double t = p / q;
    double u = std::sqrt(1.0 + t * t);
    if (q < 0) u = -u;
    g.s = -1.0 / u; g.c = -t * g.s;
  }
  return g;
}
// x_i <- c x_i + s y_i, y_i <- -s x_i + c y_i                                (apply_rotation_in_the_plane)
inline void rot(double& x, double& y, double c, double s) { const double xi = x, yi = y; x = c * xi + s * yi; y = -s * xi + c * yi; }
}  // namespace eig6

inline void eigen_solver6(const double* A, double wr[6], double wi[6], double V[6][6]) {
  using namespace eig6;
  const double eps = std::numeric_limits<double>::epsilon();
  double T[N][N], U[N][N];
  // ---- RealSchur::compute ----
  double scale = 0.0;
  for (int i = 0; i < N * N; i++) scale = std::fabs(A[i]) > scale ? std::fabs(A[i]) : scale;
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) { T[i][j] = 0.0; U[i][j] = (i == j) ? 1.0 : 0.0; }
  bool converged = true;
  if (!(scale < std::numeric_limits<double>::min())) {
    double M[N][N];
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) M[i][j] = A[i * N + j] / scale;
    // HessenbergDecomposition::_compute
    double hco[N - 1];
    for (int i = 0; i < N - 1; i++) {
      const int rem = N - i - 1;
      const Hh h = make_householder(&M[i + 1][i], rem, N);
      M[i + 1][i] = h.beta;
      hco[i] = h.tau;
      double ess[N];
      for (int k = 0; k < rem - 1; k++) ess[k] = M[i + 2 + k][i];
      apply_left(M, i + 1, i + 1, rem, rem, ess, h.tau);          // A = H A H'
      apply_right(M, 0, i + 1, N, rem, ess, h.tau);
    }
    // matrixQ = H_0 H_1 ... H_{n-2}, accumulated from the last factor (HouseholderSequence::evalTo)
    for (int k = N - 2; k >= 0; k--) {
      const int corner = N - k - 1;
      double ess[N];
      for (int t = 0; t < corner - 1; t++) ess[t] = M[k + 2 + t][k];
      apply_left(U, N - corner, N - corner, corner, corner, ess, hco[k]);
    }
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) T[i][j] = (i <= j + 1) ? M[i][j] : 0.0;     // matrixH
    // ---- RealSchur::computeFromHessenberg ----
    const int max_iters = 40 * N;
    int iu = N - 1, iter = 0, total_iter = 0;
    double exshift = 0.0;
    double norm = 0.0;                                               // computeNormOfT
    for (int j = 0; j < N; j++) for (int i = 0; i < (j + 2 < N ? j + 2 : N); i++) norm += std::fabs(T[i][j]);
    const double tiny = std::max(norm * eps * eps, std::numeric_limits<double>::min());
    if (norm != 0.0) {
      while (iu >= 0) {
        int il = iu;                                                 // findSmallSubdiagEntry
        while (il > 0) {
          double sdiag = std::fabs(T[il - 1][il - 1]) + std::fabs(T[il][il]);
          sdiag = std::max(sdiag * eps, tiny);
          if (std::fabs(T[il][il - 1]) <= sdiag) break;
          il--;
        }
        if (il == iu) {                                              // one root found
          T[iu][iu] = T[iu][iu] + exshift;
          if (iu > 0) T[iu][iu - 1] = 0.0;
          iu--; iter = 0;
        } else if (il == iu - 1) {                                   // two roots found: splitOffTwoRows
          const double p = 0.5 * (T[iu - 1][iu - 1] - T[iu][iu]);
          const double q = p * p + T[iu][iu - 1] * T[iu - 1][iu];
          T[iu][iu] += exshift;
          T[iu - 1][iu - 1] += exshift;
          if (q >= 0.0) {                                            // two real eigenvalues
            const double z = std::sqrt(std::fabs(q));
            const Givens g = (p >= 0.0) ? make_givens(p + z, T[iu][iu - 1]) : make_givens(p - z, T[iu][iu - 1]);
            for (int j = iu - 1; j < N; j++) rot(T[iu - 1][j], T[iu][j], g.c, -g.s);      // applyOnTheLeft(rot.adjoint())
            for (int i = 0; i <= iu; i++) rot(T[i][iu - 1], T[i][iu], g.c, -g.s);          // applyOnTheRight(rot)
            T[iu][iu - 1] = 0.0;
            for (int i = 0; i < N; i++) rot(U[i][iu - 1], U[i][iu], g.c, -g.s);
          }
          if (iu > 1) T[iu - 1][iu - 2] = 0.0;
          iu -= 2; iter = 0;
        } else {                                                     // no convergence yet: one Francis QR step
          double sh[3];                                              // computeShift
          sh[0] = T[iu][iu]; sh[1] = T[iu - 1][iu - 1]; sh[2] = T[iu][iu - 1] * T[iu - 1][iu];
          if (iter == 10) {                                          // Wilkinson's original ad hoc shift
            exshift += sh[0];
            for (int i = 0; i <= iu; i++) T[i][i] -= sh[0];
            const double s = std::fabs(T[iu][iu - 1]) + std::fabs(T[iu - 1][iu - 2]);
            sh[0] = 0.75 * s; sh[1] = 0.75 * s; sh[2] = -0.4375 * s * s;
          }
          if (iter == 30) {                                          // MATLAB's new ad hoc shift
            double s = (sh[1] - sh[0]) / 2.0;
            s = s * s + sh[2];
            if (s > 0.0) {
              s = std::sqrt(s);
              if (sh[1] < sh[0]) s = -s;
              s = s + (sh[1] - sh[0]) / 2.0;
              s = sh[0] - sh[2] / s;
              exshift += s;
              for (int i = 0; i <= iu; i++) T[i][i] -= s;
              sh[0] = sh[1] = sh[2] = 0.964;
            }
          }
          iter++; total_iter++;
          if (total_iter > max_iters) { converged = false; break; }
          int im;                                                    // initFrancisQRStep
          double v[3] = {0, 0, 0};
          for (im = iu - 2; im >= il; --im) {
            const double Tmm = T[im][im];
            const double r = sh[0] - Tmm, s = sh[1] - Tmm;
            v[0] = (r * s - sh[2]) / T[im + 1][im] + T[im][im + 1];
            v[1] = T[im + 1][im + 1] - Tmm - r - s;
            v[2] = T[im + 2][im + 1];
            if (im == il) break;
            const double lhs = T[im][im - 1] * (std::fabs(v[1]) + std::fabs(v[2]));
            const double rhs = v[0] * (std::fabs(T[im - 1][im - 1]) + std::fabs(Tmm) + std::fabs(T[im + 1][im + 1]));
            if (std::fabs(lhs) < eps * rhs) break;
          }
          for (int k = im; k <= iu - 2; ++k) {                       // performFrancisQRStep
            const bool first = (k == im);
            double w[3];
            if (first) { w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; }
            else { w[0] = T[k][k - 1]; w[1] = T[k + 1][k - 1]; w[2] = T[k + 2][k - 1]; }
            const Hh h = make_householder(w, 3, 1);
            if (h.beta != 0.0) {
              if (first && k > il) T[k][k - 1] = -T[k][k - 1];
              else if (!first) T[k][k - 1] = h.beta;
              apply_left(T, k, k, 3, N - k, &w[1], h.tau);
              apply_right(T, 0, k, (iu < k + 3 ? iu : k + 3) + 1, 3, &w[1], h.tau);
              apply_right(U, 0, k, N, 3, &w[1], h.tau);
            }
          }
          {
            double w[2] = {T[iu - 1][iu - 2], T[iu][iu - 2]};
            const Hh h = make_householder(w, 2, 1);
            if (h.beta != 0.0) {
              T[iu - 1][iu - 2] = h.beta;
              apply_left(T, iu - 1, iu - 1, 2, N - iu + 1, &w[1], h.tau);
              apply_right(T, 0, iu - 1, iu + 1, 2, &w[1], h.tau);
              apply_right(U, 0, iu - 1, N, 2, &w[1], h.tau);
            }
          }
          for (int i = im + 2; i <= iu; ++i) {                       // clean up pollution due to round-off errors
            T[i][i - 2] = 0.0;
            if (i > im + 2) T[i][i - 3] = 0.0;
          }
        }
      }
    }
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) T[i][j] *= scale;
  }
  (void)converged;
  // ---- EigenSolver::compute: eigenvalues from the (quasi-)triangular T, top to bottom ----
  {
    int i = 0;
    while (i < N) {
      if (i == N - 1 || T[i + 1][i] == 0.0) { wr[i] = T[i][i]; wi[i] = 0.0; ++i; }
      else {
        const double p = 0.5 * (T[i][i] - T[i + 1][i + 1]);
        double t0 = T[i + 1][i], t1 = T[i][i + 1];
        const double maxval = std::max(std::fabs(p), std::max(std::fabs(t0), std::fabs(t1)));
        t0 /= maxval; t1 /= maxval;
        const double p0 = p / maxval;
        const double z = maxval * std::sqrt(std::fabs(p0 * p0 + t0 * t1));
        wr[i] = T[i + 1][i + 1] + p; wi[i] = z;
        wr[i + 1] = T[i + 1][i + 1] + p; wi[i + 1] = -z;
        i += 2;
      }
    }
  }
  // ---- EigenSolver::doComputeEigenvectors: back substitution on T, then back transformation with U ----
  double nrm = 0.0;
  for (int j = 0; j < N; j++) for (int k = (j - 1 > 0 ? j - 1 : 0); k < N; k++) nrm += std::fabs(T[j][k]);
  if (nrm != 0.0) {
    for (int n = N - 1; n >= 0; n--) {
      const double p = wr[n], q = wi[n];
      if (q == 0.0) {                                                // real eigenvalue: real vector
        double lastr = 0.0, lastw = 0.0;
        int l = n;
        T[n][n] = 1.0;
        for (int i = n - 1; i >= 0; i--) {
          const double w = T[i][i] - p;
          double r = 0.0;
          for (int k = l; k <= n; k++) r += T[i][k] * T[k][n];
          if (wi[i] < 0.0) { lastw = w; lastr = r; }
          else {
            l = i;
            if (wi[i] == 0.0) {
              if (w != 0.0) T[i][n] = -r / w;
              else T[i][n] = -r / (eps * nrm);
            } else {                                                 // solve real equations
              const double x = T[i][i + 1], y = T[i + 1][i];
              const double denom = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i];
              const double t = (x * lastr - lastw * r) / denom;
              T[i][n] = t;
              if (std::fabs(x) > std::fabs(lastw)) T[i + 1][n] = (-r - w * t) / x;
              else T[i + 1][n] = (-lastr - y * t) / lastw;
            }
            const double t = std::fabs(T[i][n]);                     // overflow control
            if ((eps * t) * t > 1.0) for (int k = i; k < N; k++) T[k][n] /= t;
          }
        }
      } else if (q < 0.0 && n > 0) {                                 // complex pair (n-1, n): complex vector
        double lastra = 0.0, lastsa = 0.0, lastw = 0.0;
        int l = n - 1;
        if (std::fabs(T[n][n - 1]) > std::fabs(T[n - 1][n])) {
          T[n - 1][n - 1] = q / T[n][n - 1];
          T[n - 1][n] = -(T[n][n] - p) / T[n][n - 1];
        } else {
          // cc = (0, -T(n-1,n)) / (T(n-1,n-1) - p, q)
          const double ar = 0.0, ai = -T[n - 1][n], br = T[n - 1][n - 1] - p, bi = q;
          const double d = br * br + bi * bi;
          T[n - 1][n - 1] = (ar * br + ai * bi) / d;
          T[n - 1][n] = (ai * br - ar * bi) / d;
        }
        T[n][n - 1] = 0.0;
        T[n][n] = 1.0;
        for (int i = n - 2; i >= 0; i--) {
          double ra = 0.0, sa = 0.0;
          for (int k = l; k <= n; k++) { ra += T[i][k] * T[k][n - 1]; sa += T[i][k] * T[k][n]; }
          const double w = T[i][i] - p;
          if (wi[i] < 0.0) { lastw = w; lastra = ra; lastsa = sa; }
          else {
            l = i;
            if (wi[i] == 0.0) {
              const double br = w, bi = q, d = br * br + bi * bi;    // cc = (-ra, -sa) / (w, q)
              T[i][n - 1] = (-ra * br + -sa * bi) / d;
              T[i][n] = (-sa * br - -ra * bi) / d;
            } else {
              const double x = T[i][i + 1], y = T[i + 1][i];
              double vr = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i] - q * q;
              const double vi = (wr[i] - p) * 2.0 * q;
              if (vr == 0.0 && vi == 0.0) vr = eps * nrm * (std::fabs(w) + std::fabs(q) + std::fabs(x) + std::fabs(y) + std::fabs(lastw));
              const double ar = x * lastra - lastw * ra + q * sa, ai = x * lastsa - lastw * sa - q * ra;
              const double d = vr * vr + vi * vi;
              T[i][n - 1] = (ar * vr + ai * vi) / d;
              T[i][n] = (ai * vr - ar * vi) / d;
              if (std::fabs(x) > (std::fabs(lastw) + std::fabs(q))) {
                T[i + 1][n - 1] = (-ra - w * T[i][n - 1] + q * T[i][n]) / x;
                T[i + 1][n] = (-sa - w * T[i][n] - q * T[i][n - 1]) / x;
              } else {
                const double cr = -lastra - y * T[i][n - 1], ci = -lastsa - y * T[i][n];       // / (lastw, q)
                const double d2 = lastw * lastw + q * q;
                T[i + 1][n - 1] = (cr * lastw + ci * q) / d2;
                T[i + 1][n] = (ci * lastw - cr * q) / d2;
              }
            }
            const double t = std::max(std::fabs(T[i][n - 1]), std::fabs(T[i][n]));             // overflow control
            if ((eps * t) * t > 1.0) for (int k = i; k < N; k++) { T[k][n - 1] /= t; T[k][n] /= t; }
          }
        }
        n--;                                                         // the pair's other eigenvalue is done too
      }
    }
    for (int j = N - 1; j >= 0; j--) {                               // back transformation
      double tmp[N];
      for (int i = 0; i < N; i++) { double s = 0.0; for (int k = 0; k <= j; k++) s += U[i][k] * T[k][j]; tmp[i] = s; }
      for (int i = 0; i < N; i++) U[i][j] = tmp[i];
    }
  }
  // ---- EigenSolver::eigenvectors(): normalised columns; `.real()` of a conjugate pair is the pair's real part ----
  for (int j = 0; j < N; j++) {
    const bool real_ev = (std::fabs(wi[j]) <= std::fabs(wr[j]) * 2.0 * eps) || j + 1 == N;      // isMuchSmallerThan(imag, real, 2 eps)
    if (real_ev) {
      double n2 = 0.0;
      for (int i = 0; i < N; i++) n2 += U[i][j] * U[i][j];
      const double nn = std::sqrt(n2);
      for (int i = 0; i < N; i++) V[i][j] = (n2 > 0.0) ? U[i][j] / nn : U[i][j];
    } else {
      double n2 = 0.0;
      for (int i = 0; i < N; i++) n2 += U[i][j] * U[i][j] + U[i][j + 1] * U[i][j + 1];
      const double nn = std::sqrt(n2);
      for (int i = 0; i < N; i++) { V[i][j] = (n2 > 0.0) ? U[i][j] / nn : U[i][j]; V[i][j + 1] = V[i][j]; }
      ++j;
    }
  }
}

}  // namespace oracle
