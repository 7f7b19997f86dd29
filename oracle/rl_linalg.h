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
//   * symmetric 6x6 eigen-decomposition (stands in for Eigen::EigenSolver,
//        esekfom.hpp:1736 -- see the note at jacobi_eig6)
//
// PARITY UNPINNED: Eigen's expression templates fix an evaluation order that can
// only be confirmed by compiling against Eigen.  Where the order is known from the
// Eigen 3.3 sources it is followed (3-element reductions are  a0 + (a1 + a2);
// 4x4*4 products accumulate column by column left to right); elsewhere plain
// left-to-right loops are used.  All float code must be compiled with
// -ffp-contract=off (the reference is built without FMA: CMakeLists.txt:4-5,17-21).
#pragma once
#include <cmath>
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

// Cyclic Jacobi eigen-decomposition of a symmetric 6x6 (row-major).  Stands in for
// Eigen::EigenSolver<Matrix<double,6,6>> at esekfom.hpp:1736.  HTH[0:6,0:6] is symmetric
// PSD so its eigenpairs are real; Eigen's solver returns them in an order fixed by its
// Hessenberg/QR iteration which is not reproduced here.  The order only matters when an
// eigenvalue falls below the degeneracy threshold D (SURVEY.md section 8 a-note 6):
// in the non-degenerate case VEPs^-1 * VEPs == I for any ordering.
// Output: w[6] eigenvalues, V[6][6] eigenvectors as COLUMNS (V[r][c]).
inline void jacobi_eig6(const double* S, double w[6], double V[6][6]) {
  const int n = 6;
  double a[6][6];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      a[i][j] = 0.5 * (S[i * n + j] + S[j * n + i]);
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0.0;
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) off += a[i][j] * a[i][j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        if (a[p][q] == 0.0) continue;
        double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) {
          double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; i++) w[i] = a[i][i];
}

}  // namespace oracle
