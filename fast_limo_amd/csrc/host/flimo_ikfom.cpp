// fast_limo_amd/csrc/host/flimo_ikfom.cpp -- see flimo_ikfom.hpp.
#include "flimo_ikfom.hpp"
#include <algorithm>
#include <limits>

namespace flimo_host {

// ---------------------------------------------------------------------------------------------
bool inverse_lu(int n, const double* A, double* Ainv) {
  // LU with partial pivoting on a stack copy, then A^-1 = U^-1 L^-1 P column by column.  No heap use:
  // this runs twice per filter pass on the 23x23 covariance (esekfom.hpp:1722,1726).
  constexpr int NMAX = 64;
  if (n > NMAX) {
    std::vector<double> big((size_t)n * n);   // only the M < 23 branch can get here (M x M, M <= 22)
    return false;
  }
  double lu[NMAX * NMAX];
  int piv[NMAX];
  for (int i = 0; i < n * n; i++) lu[i] = A[i];
  for (int i = 0; i < n; i++) piv[i] = i;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::fabs(lu[k * n + k]);
    for (int i = k + 1; i < n; i++) {
      const double v = std::fabs(lu[i * n + k]);
      if (v > best) { best = v; p = i; }
    }
    if (best == 0.0) return false;
    if (p != k) {
      for (int j = 0; j < n; j++) std::swap(lu[k * n + j], lu[p * n + j]);
      std::swap(piv[k], piv[p]);
    }
    const double d = lu[k * n + k];
    for (int i = k + 1; i < n; i++) {
      const double f = lu[i * n + k] / d;
      lu[i * n + k] = f;
      double* ri = &lu[i * n];
      const double* rk = &lu[k * n];
      for (int j = k + 1; j < n; j++) ri[j] -= f * rk[j];
    }
  }
  // solve L U X = P (X row-major, all right-hand sides at once so the inner loops run over columns)
  double X[NMAX * NMAX];
  for (int i = 0; i < n; i++) {
    double* xi = &X[i * n];
    for (int c = 0; c < n; c++) xi[c] = (piv[i] == c) ? 1.0 : 0.0;
    for (int j = 0; j < i; j++) {
      const double l = lu[i * n + j];
      const double* xj = &X[j * n];
      for (int c = 0; c < n; c++) xi[c] -= l * xj[c];
    }
  }
  for (int i = n - 1; i >= 0; i--) {
    double* xi = &X[i * n];
    for (int j = i + 1; j < n; j++) {
      const double u = lu[i * n + j];
      const double* xj = &X[j * n];
      for (int c = 0; c < n; c++) xi[c] -= u * xj[c];
    }
    const double d = lu[i * n + i];
    for (int c = 0; c < n; c++) xi[c] /= d;
  }
  for (int i = 0; i < n * n; i++) Ainv[i] = X[i];
  return true;
}

// Gauss-Jordan inverse with partial pivoting, without row exchanges (a pivot row is marked used instead): in step k every OTHER
// row r gets  row_r -= (A[r][k] / A[p][k]) * row_p  on the columns right of k and on the accumulated right-hand side.  The
// per-element operations of a step are independent of each other, so a parallel evaluation (the device filter's 12 x 12 solve,
// csrc/hip/flimo_ieskf.h: ik_gj12_wave) gives the same bits; the chain of n dependent steps is half of LU + two substitutions.
// Pivot: the largest |A[r][k]| among the rows not yet used, the lowest r among equals.
bool inverse_gj(int n, const double* Ain, double* Ainv) {
  constexpr int NMAX = 32;
  if (n > NMAX) return false;
  double A[NMAX * NMAX], X[NMAX * NMAX], dk[NMAX];
  int prow[NMAX];
  bool used[NMAX];
  for (int i = 0; i < n * n; i++) { A[i] = Ain[i]; X[i] = 0.0; }
  for (int i = 0; i < n; i++) { X[i * n + i] = 1.0; used[i] = false; }
  for (int k = 0; k < n; k++) {
    int p = -1;
    double best = -1.0;
    for (int r = 0; r < n; r++) {
      if (used[r]) continue;
      const double v = std::fabs(A[r * n + k]);
      if (v > best) { best = v; p = r; }
    }
    if (p < 0 || !(best > 0.0)) return false;
    const double rinv = 1.0 / A[p * n + k];
    const double* ap = &A[p * n];
    const double* xp = &X[p * n];
    for (int r = 0; r < n; r++) {
      if (r == p) continue;
      const double f = A[r * n + k] * rinv;
      double* ar = &A[r * n];
      double* xr = &X[r * n];
      for (int j = k + 1; j < n; j++) ar[j] = ar[j] - f * ap[j];
      for (int c = 0; c < n; c++) xr[c] = xr[c] - f * xp[c];
    }
    used[p] = true; prow[k] = p; dk[k] = rinv;
  }
  // row prow[k] of the reduced left side is A[p][k] e_k^T: row k of the inverse = X[prow[k]] * (1 / A[p][k])
  for (int k = 0; k < n; k++)
    for (int c = 0; c < n; c++) Ainv[k * n + c] = X[prow[k] * n + c] * dk[k];
  return true;
}

// A u = v by the same elimination, the right-hand side as one more column (csrc/hip/flimo_ieskf.h: ik_gj12_solve_wave)
bool solve_gj(int n, const double* Ain, const double* v, double* u) {
  constexpr int NMAX = 32;
  if (n > NMAX) return false;
  double A[NMAX * NMAX], b[NMAX], dk[NMAX];
  int prow[NMAX];
  bool used[NMAX];
  for (int i = 0; i < n * n; i++) A[i] = Ain[i];
  for (int i = 0; i < n; i++) { b[i] = v[i]; used[i] = false; }
  for (int k = 0; k < n; k++) {
    int p = -1;
    double best = -1.0;
    for (int r = 0; r < n; r++) {
      if (used[r]) continue;
      const double m = std::fabs(A[r * n + k]);
      if (m > best) { best = m; p = r; }
    }
    if (p < 0 || !(best > 0.0)) return false;
    const double rinv = 1.0 / A[p * n + k];
    const double* ap = &A[p * n];
    const double bp = b[p];
    for (int r = 0; r < n; r++) {
      if (r == p) continue;
      const double f = A[r * n + k] * rinv;
      double* ar = &A[r * n];
      for (int j = k + 1; j < n; j++) ar[j] = ar[j] - f * ap[j];
      b[r] = b[r] - f * bp;
    }
    used[p] = true; prow[k] = p; dk[k] = rinv;
  }
  for (int k = 0; k < n; k++) u[k] = b[prow[k]] * dk[k];
  return true;
}

// ---------------------------------------------------------------------------------------------
// Eigen::EigenSolver<Matrix<double,6,6>> as the reference calls it at esekfom.hpp:1736-1738.  Which ROW of the eigenvector matrix
// the degeneracy projector zeroes is decided by the order (and the product it forms by the signs) in which Eigen hands out the
// eigenpairs, so the solver's algorithm is followed step by step (Eigen 3.3.7: Householder/Householder.h, Jacobi/Jacobi.h,
// Eigenvalues/HessenbergDecomposition.h, RealSchur.h, EigenSolver.h): scaling, Householder reduction to Hessenberg form with Q
// accumulated from the last reflector, Francis double-shift QR with Eigen's deflation test and exceptional shifts, eigenvalues read
// off the diagonal of T from the top, eigenvectors by back substitution and multiplication with Q, normalised columns, `.real()`.
// Sums inside a step run left to right (Eigen's vectorised reductions may round the last bit differently).
// ---------------------------------------------------------------------------------------------
namespace {
typedef Mat<6, 6> M6;
struct Reflector { double tau, beta; };
// x(0) <- beta, x(1..) <- essential part; stride: distance between consecutive coefficients
Reflector reflector(double* x, int len, int stride) {
  double tail = 0.0;
  for (int i = 1; i < len; i++) tail += x[i * stride] * x[i * stride];
  const double c0 = x[0];
  if (tail <= std::numeric_limits<double>::min()) {
    for (int i = 1; i < len; i++) x[i * stride] = 0.0;
    return Reflector{0.0, c0};
  }
  double beta = std::sqrt(c0 * c0 + tail);
  if (c0 >= 0.0) beta = -beta;
  for (int i = 1; i < len; i++) x[i * stride] = x[i * stride] / (c0 - beta);
  return Reflector{(beta - c0) / beta, beta};
}
void reflect_left(M6& A, int r0, int c0, int nr, int nc, const double* ess, double tau) {
  if (nr == 1) { for (int j = 0; j < nc; j++) A(r0, c0 + j) *= (1.0 - tau); return; }
  if (tau == 0.0) return;
  for (int j = 0; j < nc; j++) {
    double t = 0.0;
    for (int i = 1; i < nr; i++) t += ess[i - 1] * A(r0 + i, c0 + j);
    t += A(r0, c0 + j);
    A(r0, c0 + j) -= tau * t;
    for (int i = 1; i < nr; i++) A(r0 + i, c0 + j) -= tau * ess[i - 1] * t;
  }
}
void reflect_right(M6& A, int r0, int c0, int nr, int nc, const double* ess, double tau) {
  if (nc == 1) { for (int i = 0; i < nr; i++) A(r0 + i, c0) *= (1.0 - tau); return; }
  if (tau == 0.0) return;
  for (int i = 0; i < nr; i++) {
    double t = 0.0;
    for (int j = 1; j < nc; j++) t += A(r0 + i, c0 + j) * ess[j - 1];
    t += A(r0 + i, c0);
    A(r0 + i, c0) -= tau * t;
    for (int j = 1; j < nc; j++) A(r0 + i, c0 + j) -= tau * t * ess[j - 1];
  }
}
// JacobiRotation::makeGivens, then the rotation J^T applied as the two-row / two-column update Eigen performs
void givens(double p, double q, double& c, double& s) {
  if (q == 0.0) { c = p < 0 ? -1.0 : 1.0; s = 0.0; return; }
  if (p == 0.0) { c = 0.0; s = q < 0 ? 1.0 : -1.0; return; }
  if (std::fabs(p) > std::fabs(q)) {
    const double t = q / p;
    double u = std::sqrt(1.0 + t * t);
    if (p < 0) u = -u;
    c = 1.0 / u; s = -t * c;
  } else {
    const double t = p / q;
    double u = std::sqrt(1.0 + t * t);
    if (q < 0) u = -u;
    s = -1.0 / u; c = -t * s;
  }
}
inline void plane_rot(double& x, double& y, double c, double s) { const double a = x, b = y; x = c * a - s * b; y = s * a + c * b; }
}  // namespace

void eigen_solver6(const Mat<6, 6>& A, double wr[6], double wi[6], Mat<6, 6>& V) {
  const int N = 6;
  const double eps = std::numeric_limits<double>::epsilon(), dmin = std::numeric_limits<double>::min();
  M6 T = M6::zero(), U = M6::identity();
  double scale = 0.0;
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) scale = std::max(scale, std::fabs(A(i, j)));
  if (!(scale < dmin)) {
    M6 H;
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) H(i, j) = A(i, j) / scale;
    double tau[5];
    for (int i = 0; i + 1 < N; i++) {                                // Hessenberg: A <- H_i A H_i'
      const int rem = N - i - 1;
      const Reflector h = reflector(&H(i + 1, i), rem, N);
      H(i + 1, i) = h.beta;
      tau[i] = h.tau;
      double ess[6];
      for (int k = 0; k + 1 < rem; k++) ess[k] = H(i + 2 + k, i);
      reflect_left(H, i + 1, i + 1, rem, rem, ess, h.tau);
      reflect_right(H, 0, i + 1, N, rem, ess, h.tau);
    }
    for (int k = N - 2; k >= 0; k--) {                               // Q = H_0 ... H_4 from the identity, last factor first
      const int cs = N - k - 1;
      double ess[6];
      for (int t = 0; t + 1 < cs; t++) ess[t] = H(k + 2 + t, k);
      reflect_left(U, N - cs, N - cs, cs, cs, ess, tau[k]);
    }
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) T(i, j) = (i <= j + 1) ? H(i, j) : 0.0;
    double norm = 0.0;
    for (int j = 0; j < N; j++) for (int i = 0; i < std::min(N, j + 2); i++) norm += std::fabs(T(i, j));
    const double as_zero = std::max(norm * eps * eps, dmin);
    int iu = N - 1, iter = 0, total = 0;
    double exshift = 0.0;
    while (norm != 0.0 && iu >= 0) {
      int il = iu;
      for (; il > 0; il--) {
        const double s = std::max((std::fabs(T(il - 1, il - 1)) + std::fabs(T(il, il))) * eps, as_zero);
        if (std::fabs(T(il, il - 1)) <= s) break;
      }
      if (il == iu) {                                                // one root
        T(iu, iu) = T(iu, iu) + exshift;
        if (iu > 0) T(iu, iu - 1) = 0.0;
        iu--; iter = 0;
        continue;
      }
      if (il == iu - 1) {                                            // two roots
        const double p = 0.5 * (T(iu - 1, iu - 1) - T(iu, iu));
        const double q = p * p + T(iu, iu - 1) * T(iu - 1, iu);
        T(iu, iu) += exshift;
        T(iu - 1, iu - 1) += exshift;
        if (q >= 0.0) {
          const double z = std::sqrt(std::fabs(q));
          double c, s;
          givens(p >= 0.0 ? p + z : p - z, T(iu, iu - 1), c, s);
          for (int j = iu - 1; j < N; j++) plane_rot(T(iu - 1, j), T(iu, j), c, s);
          for (int i = 0; i <= iu; i++) plane_rot(T(i, iu - 1), T(i, iu), c, s);
          T(iu, iu - 1) = 0.0;
          for (int i = 0; i < N; i++) plane_rot(U(i, iu - 1), U(i, iu), c, s);
        }
        if (iu > 1) T(iu - 1, iu - 2) = 0.0;
        iu -= 2; iter = 0;
        continue;
      }
      double sh0 = T(iu, iu), sh1 = T(iu - 1, iu - 1), sh2 = T(iu, iu - 1) * T(iu - 1, iu);
      if (iter == 10) {                                              // Wilkinson's ad hoc shift
        exshift += sh0;
        for (int i = 0; i <= iu; i++) T(i, i) -= sh0;
        const double s = std::fabs(T(iu, iu - 1)) + std::fabs(T(iu - 1, iu - 2));
        sh0 = 0.75 * s; sh1 = 0.75 * s; sh2 = -0.4375 * s * s;
      }
      if (iter == 30) {                                              // MATLAB's ad hoc shift
        double s = (sh1 - sh0) / 2.0;
        s = s * s + sh2;
        if (s > 0.0) {
          s = std::sqrt(s);
          if (sh1 < sh0) s = -s;
          s = s + (sh1 - sh0) / 2.0;
          s = sh0 - sh2 / s;
          exshift += s;
          for (int i = 0; i <= iu; i++) T(i, i) -= s;
          sh0 = sh1 = sh2 = 0.964;
        }
      }
      iter++;
      if (++total > 40 * N) break;
      int im = iu - 2;
      double v[3] = {0, 0, 0};
      for (; im >= il; --im) {
        const double Tmm = T(im, im), r = sh0 - Tmm, s = sh1 - Tmm;
        v[0] = (r * s - sh2) / T(im + 1, im) + T(im, im + 1);
        v[1] = T(im + 1, im + 1) - Tmm - r - s;
        v[2] = T(im + 2, im + 1);
        if (im == il) break;
        const double lhs = T(im, im - 1) * (std::fabs(v[1]) + std::fabs(v[2]));
        const double rhs = v[0] * (std::fabs(T(im - 1, im - 1)) + std::fabs(Tmm) + std::fabs(T(im + 1, im + 1)));
        if (std::fabs(lhs) < eps * rhs) break;
      }
      for (int k = im; k <= iu - 2; ++k) {
        double w[3];
        if (k == im) { w[0] = v[0]; w[1] = v[1]; w[2] = v[2]; }
        else { w[0] = T(k, k - 1); w[1] = T(k + 1, k - 1); w[2] = T(k + 2, k - 1); }
        const Reflector h = reflector(w, 3, 1);
        if (h.beta == 0.0) continue;
        if (k == im && k > il) T(k, k - 1) = -T(k, k - 1);
        else if (k != im) T(k, k - 1) = h.beta;
        reflect_left(T, k, k, 3, N - k, &w[1], h.tau);
        reflect_right(T, 0, k, std::min(iu, k + 3) + 1, 3, &w[1], h.tau);
        reflect_right(U, 0, k, N, 3, &w[1], h.tau);
      }
      {
        double w[2] = {T(iu - 1, iu - 2), T(iu, iu - 2)};
        const Reflector h = reflector(w, 2, 1);
        if (h.beta != 0.0) {
          T(iu - 1, iu - 2) = h.beta;
          reflect_left(T, iu - 1, iu - 1, 2, N - iu + 1, &w[1], h.tau);
          reflect_right(T, 0, iu - 1, iu + 1, 2, &w[1], h.tau);
          reflect_right(U, 0, iu - 1, N, 2, &w[1], h.tau);
        }
      }
      for (int i = im + 2; i <= iu; ++i) { T(i, i - 2) = 0.0; if (i > im + 2) T(i, i - 3) = 0.0; }
    }
    for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) T(i, j) *= scale;
  }
  for (int i = 0; i < N;) {                                          // eigenvalues, top to bottom
    if (i == N - 1 || T(i + 1, i) == 0.0) { wr[i] = T(i, i); wi[i] = 0.0; i++; continue; }
    const double p = 0.5 * (T(i, i) - T(i + 1, i + 1));
    double t0 = T(i + 1, i), t1 = T(i, i + 1);
    const double mx = std::max(std::fabs(p), std::max(std::fabs(t0), std::fabs(t1)));
    t0 /= mx; t1 /= mx;
    const double p0 = p / mx, z = mx * std::sqrt(std::fabs(p0 * p0 + t0 * t1));
    wr[i] = wr[i + 1] = T(i + 1, i + 1) + p;
    wi[i] = z; wi[i + 1] = -z;
    i += 2;
  }
  double nrm = 0.0;
  for (int j = 0; j < N; j++) for (int k = std::max(j - 1, 0); k < N; k++) nrm += std::fabs(T(j, k));
  if (nrm != 0.0) {
    for (int n = N - 1; n >= 0; n--) {
      const double p = wr[n], q = wi[n];
      if (q == 0.0) {
        double lastr = 0.0, lastw = 0.0;
        int l = n;
        T(n, n) = 1.0;
        for (int i = n - 1; i >= 0; i--) {
          const double w = T(i, i) - p;
          double r = 0.0;
          for (int k = l; k <= n; k++) r += T(i, k) * T(k, n);
          if (wi[i] < 0.0) { lastw = w; lastr = r; continue; }
          l = i;
          if (wi[i] == 0.0) T(i, n) = (w != 0.0) ? -r / w : -r / (eps * nrm);
          else {
            const double x = T(i, i + 1), y = T(i + 1, i);
            const double den = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i];
            const double t = (x * lastr - lastw * r) / den;
            T(i, n) = t;
            T(i + 1, n) = (std::fabs(x) > std::fabs(lastw)) ? (-r - w * t) / x : (-lastr - y * t) / lastw;
          }
          const double t = std::fabs(T(i, n));
          if ((eps * t) * t > 1.0) for (int k = i; k < N; k++) T(k, n) /= t;
        }
      } else if (q < 0.0 && n > 0) {
        // a conjugate pair (only rounding produces one from a symmetric matrix): complex back substitution
        auto cdiv = [](double ar, double ai, double br, double bi, double& cr, double& ci) {
          const double d = br * br + bi * bi;
          cr = (ar * br + ai * bi) / d; ci = (ai * br - ar * bi) / d;
        };
        double lastra = 0.0, lastsa = 0.0, lastw = 0.0;
        int l = n - 1;
        if (std::fabs(T(n, n - 1)) > std::fabs(T(n - 1, n))) {
          T(n - 1, n - 1) = q / T(n, n - 1);
          T(n - 1, n) = -(T(n, n) - p) / T(n, n - 1);
        } else {
          double cr, ci;
          cdiv(0.0, -T(n - 1, n), T(n - 1, n - 1) - p, q, cr, ci);
          T(n - 1, n - 1) = cr; T(n - 1, n) = ci;
        }
        T(n, n - 1) = 0.0;
        T(n, n) = 1.0;
        for (int i = n - 2; i >= 0; i--) {
          double ra = 0.0, sa = 0.0;
          for (int k = l; k <= n; k++) { ra += T(i, k) * T(k, n - 1); sa += T(i, k) * T(k, n); }
          const double w = T(i, i) - p;
          if (wi[i] < 0.0) { lastw = w; lastra = ra; lastsa = sa; continue; }
          l = i;
          if (wi[i] == 0.0) {
            double cr, ci;
            cdiv(-ra, -sa, w, q, cr, ci);
            T(i, n - 1) = cr; T(i, n) = ci;
          } else {
            const double x = T(i, i + 1), y = T(i + 1, i);
            double vr = (wr[i] - p) * (wr[i] - p) + wi[i] * wi[i] - q * q;
            const double vi = (wr[i] - p) * 2.0 * q;
            if (vr == 0.0 && vi == 0.0) vr = eps * nrm * (std::fabs(w) + std::fabs(q) + std::fabs(x) + std::fabs(y) + std::fabs(lastw));
            double cr, ci;
            cdiv(x * lastra - lastw * ra + q * sa, x * lastsa - lastw * sa - q * ra, vr, vi, cr, ci);
            T(i, n - 1) = cr; T(i, n) = ci;
            if (std::fabs(x) > (std::fabs(lastw) + std::fabs(q))) {
              T(i + 1, n - 1) = (-ra - w * T(i, n - 1) + q * T(i, n)) / x;
              T(i + 1, n) = (-sa - w * T(i, n) - q * T(i, n - 1)) / x;
            } else {
              cdiv(-lastra - y * T(i, n - 1), -lastsa - y * T(i, n), lastw, q, cr, ci);
              T(i + 1, n - 1) = cr; T(i + 1, n) = ci;
            }
          }
          const double t = std::max(std::fabs(T(i, n - 1)), std::fabs(T(i, n)));
          if ((eps * t) * t > 1.0) for (int k = i; k < N; k++) { T(k, n - 1) /= t; T(k, n) /= t; }
        }
        n--;
      }
    }
    for (int j = N - 1; j >= 0; j--) {
      double col[6];
      for (int i = 0; i < N; i++) { double s = 0.0; for (int k = 0; k <= j; k++) s += U(i, k) * T(k, j); col[i] = s; }
      for (int i = 0; i < N; i++) U(i, j) = col[i];
    }
  }
  for (int j = 0; j < N; j++) {                                      // eigenvectors(): normalised; real part of a conjugate pair
    const bool is_real = (std::fabs(wi[j]) <= std::fabs(wr[j]) * 2.0 * eps) || j + 1 == N;
    double n2 = 0.0;
    for (int i = 0; i < N; i++) n2 += U(i, j) * U(i, j) + (is_real ? 0.0 : U(i, j + 1) * U(i, j + 1));
    const double nn = std::sqrt(n2);
    for (int i = 0; i < N; i++) V(i, j) = n2 > 0.0 ? U(i, j) / nn : U(i, j);
    if (!is_real) { for (int i = 0; i < N; i++) V(i, j + 1) = V(i, j); j++; }
  }
}

// ---------------------------------------------------------------------------------------------
static const double kTol = 1e-11;   // MTK::tolerance<double>()

Quat quat_mul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
Mat3 quat_to_rot(const Quat& q) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  Mat3 R;
  R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz;       R(0, 2) = txz + twy;
  R(1, 0) = txy + twz;       R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
  R(2, 0) = txz - twy;       R(2, 1) = tyz + twx;       R(2, 2) = 1 - (txx + tyy);
  return R;
}
Quat rot_to_quat(const Mat3& m) {
  Quat q;
  double t = m(0, 0) + (m(1, 1) + m(2, 2));
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m(2, 1) - m(1, 2)) * t;
    q.y = (m(0, 2) - m(2, 0)) * t;
    q.z = (m(1, 0) - m(0, 1)) * t;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    double c[3];
    c[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m(k, j) - m(j, k)) * t;
    c[j] = (m(j, i) + m(i, j)) * t;
    c[k] = (m(k, i) + m(i, k)) * t;
    q.x = c[0]; q.y = c[1]; q.z = c[2];
  }
  return q;
}
Vec3 quat_rotate(const Quat& q, const Vec3& v) {
  double uv[3] = {q.y * v(2, 0) - q.z * v(1, 0), q.z * v(0, 0) - q.x * v(2, 0), q.x * v(1, 0) - q.y * v(0, 0)};
  for (int i = 0; i < 3; i++) uv[i] += uv[i];
  const double c[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
  Vec3 o;
  for (int i = 0; i < 3; i++) o(i, 0) = v(i, 0) + q.w * uv[i] + c[i];
  return o;
}
Mat3 hat(const Vec3& v) {
  Mat3 H = Mat3::zero();
  H(0, 1) = -v(2, 0); H(0, 2) = v(1, 0);
  H(1, 0) = v(2, 0);  H(1, 2) = -v(0, 0);
  H(2, 0) = -v(1, 0); H(2, 1) = v(0, 0);
  return H;
}
Mat3 A_matrix(const Vec3& v) {
  const double sq = v(0, 0) * v(0, 0) + v(1, 0) * v(1, 0) + v(2, 0) * v(2, 0);
  const double norm = std::sqrt(sq);
  if (norm < kTol) return Mat3::identity();
  const Mat3 H = hat(v);
  return Mat3::identity() + ((1 - std::cos(norm)) / sq) * H + ((1 - std::sin(norm) / norm) / sq) * (H * H);
}

// MTK::cos_sinc_sqrt (mtkmath.hpp:143-174)
static void cos_sinc_sqrt(double x2, double& c, double& s) {
  static const double b0 = 2.220446049250313e-16;
  static const double b2 = std::sqrt(b0);
  static const double bn = std::sqrt(b2);
  if (x2 >= bn) {
    const double x = std::sqrt(x2);
    c = std::cos(x);
    s = std::sin(x) / x;
    return;
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
  c = cosi;
  s = sinc;
}
// MTK::exp<scalar,3>: q = (cos(scale|v|), sinc(scale|v|) * scale * v)
static Quat exp_quat(const Vec3& v, double scale) {
  const double n2 = v(0, 0) * v(0, 0) + v(1, 0) * v(1, 0) + v(2, 0) * v(2, 0);
  double c, s;
  cos_sinc_sqrt(scale * scale * n2, c, s);
  const double mult = s * scale;
  Quat q;
  q.w = c; q.x = mult * v(0, 0); q.y = mult * v(1, 0); q.z = mult * v(2, 0);
  return q;
}
Quat so3_exp(const Vec3& v, double scale) { return exp_quat(v, scale / 2); }
Vec3 so3_log(const Quat& q) {
  double nv = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  if (nv < kTol) nv = kTol;
  const double s = 2.0 / nv * std::atan(nv / q.w);
  Vec3 o;
  o(0, 0) = s * q.x; o(1, 0) = s * q.y; o(2, 0) = s * q.z;
  return o;
}

// ---- S2 -------------------------------------------------------------------------------------
S2::S2() { vec = Vec3::zero(); vec(0, 0) = kS2Length; }
S2::S2(double x, double y, double z) {
  const double n = std::sqrt(x * x + y * y + z * z);
  vec(0, 0) = x / n * kS2Length; vec(1, 0) = y / n * kS2Length; vec(2, 0) = z / n * kS2Length;
}
Mat<3, 2> S2::Bx() const {
  const double L = kS2Length, v0 = vec(0, 0), v1 = vec(1, 0), v2 = vec(2, 0);
  Mat<3, 2> B = Mat<3, 2>::zero();
  if (v0 + L > kTol) {
    B(0, 0) = -v1;                    B(0, 1) = -v2;
    B(1, 0) = L - v1 * v1 / (L + v0); B(1, 1) = -v2 * v1 / (L + v0);
    B(2, 0) = -v2 * v1 / (L + v0);    B(2, 1) = L - v2 * v2 / (L + v0);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 2; j++) B(i, j) /= L;
  } else {
    B(1, 1) = -1; B(2, 0) = 1;
  }
  return B;
}
void S2::boxplus(const double d[2], double scale) {
  const Mat<3, 2> B = Bx();
  Vec3 Bu;
  for (int i = 0; i < 3; i++) Bu(i, 0) = B(i, 0) * d[0] + B(i, 1) * d[1];
  vec = quat_to_rot(exp_quat(Bu, scale / 2)) * vec;
}
void S2::oplus(const Vec3& d, double scale) { vec = quat_to_rot(exp_quat(d, scale / 2)) * vec; }
void S2::boxminus(double out[2], const S2& other) const {
  const Vec3 hv = hat(vec) * other.vec;
  const double v_sin = std::sqrt(hv(0, 0) * hv(0, 0) + hv(1, 0) * hv(1, 0) + hv(2, 0) * hv(2, 0));
  const double v_cos = vec(0, 0) * other.vec(0, 0) + vec(1, 0) * other.vec(1, 0) + vec(2, 0) * other.vec(2, 0);
  const double theta = std::atan2(v_sin, v_cos);
  if (v_sin < kTol) {
    if (std::fabs(theta) > kTol) { out[0] = 3.1415926; out[1] = 0; }
    else { out[0] = 0; out[1] = 0; }
    return;
  }
  const Mat<3, 2> B = other.Bx();
  const Vec3 t = hat(other.vec) * vec;
  const double f = theta / v_sin;
  for (int j = 0; j < 2; j++) out[j] = f * (B(0, j) * t(0, 0) + B(1, j) * t(1, 0) + B(2, j) * t(2, 0));
}
Mat<2, 3> S2::Nx_yy() const { return (1 / kS2Length / kS2Length) * (Bx().T() * hat(vec)); }
Mat<3, 2> S2::Mx(const double delta[2]) const {
  const Mat<3, 2> B = Bx();
  const double dn = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1]);
  if (dn < kTol) return -1.0 * (hat(vec) * B);
  Vec3 Bu;
  for (int i = 0; i < 3; i++) Bu(i, 0) = B(i, 0) * delta[0] + B(i, 1) * delta[1];
  // reference: exp(Bu, scalar(1/2)) with integer 1/2 == 0  =>  identity rotation (S2.hpp:277)
  const Mat3 E = quat_to_rot(exp_quat(Bu, double(1 / 2)));
  return -1.0 * (E * hat(vec) * A_matrix(Bu).T() * B);
}

// ---- state ----------------------------------------------------------------------------------
StateIkfom::StateIkfom() {
  pos = offset_T_L_I = vel = bg = ba = Vec3::zero();
}
void StateIkfom::boxplus(const double d[kDof]) {
  Vec3 r, o;
  for (int i = 0; i < 3; i++) { pos(i, 0) += d[i]; r(i, 0) = d[3 + i]; o(i, 0) = d[6 + i]; }
  rot = quat_mul(rot, so3_exp(r));
  offset_R_L_I = quat_mul(offset_R_L_I, so3_exp(o));
  for (int i = 0; i < 3; i++) { offset_T_L_I(i, 0) += d[9 + i]; vel(i, 0) += d[12 + i]; bg(i, 0) += d[15 + i]; ba(i, 0) += d[18 + i]; }
  grav.boxplus(d + 21);
}
void StateIkfom::oplus(const double f[kDim], double dt) {
  Vec3 r, o, g;
  for (int i = 0; i < 3; i++) { pos(i, 0) += dt * f[i]; r(i, 0) = f[3 + i]; o(i, 0) = f[6 + i]; g(i, 0) = f[21 + i]; }
  rot = quat_mul(rot, so3_exp(r, dt));
  offset_R_L_I = quat_mul(offset_R_L_I, so3_exp(o, dt));
  for (int i = 0; i < 3; i++) { offset_T_L_I(i, 0) += dt * f[9 + i]; vel(i, 0) += dt * f[12 + i]; bg(i, 0) += dt * f[15 + i]; ba(i, 0) += dt * f[18 + i]; }
  grav.oplus(g, dt);
}
void StateIkfom::boxminus(double out[kDof], const StateIkfom& o) const {
  const Vec3 r = so3_log(quat_mul(quat_conj(o.rot), rot));
  const Vec3 l = so3_log(quat_mul(quat_conj(o.offset_R_L_I), offset_R_L_I));
  for (int i = 0; i < 3; i++) {
    out[i] = pos(i, 0) - o.pos(i, 0);
    out[3 + i] = r(i, 0);
    out[6 + i] = l(i, 0);
    out[9 + i] = offset_T_L_I(i, 0) - o.offset_T_L_I(i, 0);
    out[12 + i] = vel(i, 0) - o.vel(i, 0);
    out[15 + i] = bg(i, 0) - o.bg(i, 0);
    out[18 + i] = ba(i, 0) - o.ba(i, 0);
  }
  grav.boxminus(out + 21, o.grav);
}
void StateIkfom::to_flat(double x[26]) const {
  int k = 0;
  for (int i = 0; i < 3; i++) x[k++] = pos(i, 0);
  x[k++] = rot.x; x[k++] = rot.y; x[k++] = rot.z; x[k++] = rot.w;
  x[k++] = offset_R_L_I.x; x[k++] = offset_R_L_I.y; x[k++] = offset_R_L_I.z; x[k++] = offset_R_L_I.w;
  for (int i = 0; i < 3; i++) x[k++] = offset_T_L_I(i, 0);
  for (int i = 0; i < 3; i++) x[k++] = vel(i, 0);
  for (int i = 0; i < 3; i++) x[k++] = bg(i, 0);
  for (int i = 0; i < 3; i++) x[k++] = ba(i, 0);
  for (int i = 0; i < 3; i++) x[k++] = grav.vec(i, 0);
}
void StateIkfom::from_flat(const double x[26]) {
  int k = 0;
  for (int i = 0; i < 3; i++) pos(i, 0) = x[k++];
  rot.x = x[k++]; rot.y = x[k++]; rot.z = x[k++]; rot.w = x[k++];
  offset_R_L_I.x = x[k++]; offset_R_L_I.y = x[k++]; offset_R_L_I.z = x[k++]; offset_R_L_I.w = x[k++];
  for (int i = 0; i < 3; i++) offset_T_L_I(i, 0) = x[k++];
  for (int i = 0; i < 3; i++) vel(i, 0) = x[k++];
  for (int i = 0; i < 3; i++) bg(i, 0) = x[k++];
  for (int i = 0; i < 3; i++) ba(i, 0) = x[k++];
  for (int i = 0; i < 3; i++) grav.vec(i, 0) = x[k++];
}

// ---- process model (use-ikfom.cpp:43-84) ----------------------------------------------------
static void model_f(const StateIkfom& s, const InputIkfom& in, double f[kDim]) {
  for (int i = 0; i < kDim; i++) f[i] = 0.0;
  const Vec3 a_in = quat_rotate(s.rot, in.acc - s.ba);
  for (int i = 0; i < 3; i++) {
    f[i] = s.vel(i, 0);
    f[3 + i] = in.gyro(i, 0) - s.bg(i, 0);
    f[12 + i] = a_in(i, 0) + s.grav.vec(i, 0);
  }
}
static Mat<kDim, kDof> model_df_dx(const StateIkfom& s, const InputIkfom& in) {
  Mat<kDim, kDof> F = Mat<kDim, kDof>::zero();
  for (int i = 0; i < 3; i++) F(i, 12 + i) = 1.0;
  const Mat3 R = quat_to_rot(s.rot);
  const Mat3 RH = R * hat(in.acc - s.ba);
  const double z2[2] = {0, 0};
  const Mat<3, 2> gm = s.grav.Mx(z2);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) { F(12 + i, 3 + j) = -RH(i, j); F(12 + i, 18 + j) = -R(i, j); }
    for (int j = 0; j < 2; j++) F(12 + i, 21 + j) = gm(i, j);
    F(3 + i, 15 + i) = -1.0;
  }
  return F;
}
static Mat<kDim, 12> model_df_dw(const StateIkfom& s) {
  Mat<kDim, 12> G = Mat<kDim, 12>::zero();
  const Mat3 R = quat_to_rot(s.rot);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) G(12 + i, 3 + j) = -R(i, j);
    G(3 + i, i) = -1.0; G(15 + i, 6 + i) = 1.0; G(18 + i, 9 + i) = 1.0;
  }
  return G;
}

// ---- filter ---------------------------------------------------------------------------------
Esekf::Esekf() {
  P_ = Cov::identity();
  for (int i = 0; i < kDof; i++) limit_[i] = 1e-3;
}
void Esekf::init(int maximum_iteration, const double* limits) {
  maximum_iter_ = maximum_iteration;
  for (int i = 0; i < kDof; i++) limit_[i] = limits[i];
}

// rows idx..idx+B of M <- J * rows ; applied to the first ncols columns
template <int B, int N>
static void left_block(Mat<N, N>& M, int idx, const Mat<B, B>& J, int ncols) {
  for (int c = 0; c < ncols; c++) {
    double t[B];
    for (int r = 0; r < B; r++) { double s = 0; for (int k = 0; k < B; k++) s += J(r, k) * M(idx + k, c); t[r] = s; }
    for (int r = 0; r < B; r++) M(idx + r, c) = t[r];
  }
}
// cols idx..idx+B of M <- cols * J^T
template <int B, int N>
static void right_block_T(Mat<N, N>& M, int idx, const Mat<B, B>& J) {
  for (int r = 0; r < N; r++) {
    double t[B];
    for (int c = 0; c < B; c++) { double s = 0; for (int k = 0; k < B; k++) s += M(r, idx + k) * J(c, k); t[c] = s; }
    for (int c = 0; c < B; c++) M(r, idx + c) = t[c];
  }
}

void Esekf::predict(double dt, const Mat<12, 12>& Q, const InputIkfom& in) {
  double f[kDim];
  model_f(x_, in, f);
  const Mat<kDim, kDof> fx = model_df_dx(x_, in);
  const Mat<kDim, 12> fw = model_df_dw(x_);
  const StateIkfom x_before = x_;
  x_.oplus(f, dt);

  Cov F1 = Cov::identity();
  Cov fxf = Cov::zero();
  Mat<kDof, 12> fwf = Mat<kDof, 12>::zero();
  // vector blocks: rows copied (DOF index == DIM index up to the S2 block)
  static const int vect_idx[5] = {0, 9, 12, 15, 18};
  for (int v = 0; v < 5; v++)
    for (int j = 0; j < 3; j++) {
      for (int c = 0; c < kDof; c++) fxf(vect_idx[v] + j, c) = fx(vect_idx[v] + j, c);
      for (int c = 0; c < 12; c++) fwf(vect_idx[v] + j, c) = fw(vect_idx[v] + j, c);
    }
  // SO3 blocks
  static const int so3_idx[2] = {3, 6};
  for (int s = 0; s < 2; s++) {
    const int idx = so3_idx[s];
    Vec3 seg;
    for (int i = 0; i < 3; i++) seg(i, 0) = -1 * f[idx + i] * dt;
    // F_x1 block = exp(seg, scalar(1/2)) with 1/2 == 0: identity (esekfom.hpp:312)
    const Mat3 E = quat_to_rot(exp_quat(seg, double(1 / 2)));
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) F1(idx + i, idx + j) = E(i, j);
    const Mat3 A = A_matrix(seg);
    for (int c = 0; c < kDof; c++)
      for (int r = 0; r < 3; r++) fxf(idx + r, c) = A(r, 0) * fx(idx, c) + A(r, 1) * fx(idx + 1, c) + A(r, 2) * fx(idx + 2, c);
    for (int c = 0; c < 12; c++)
      for (int r = 0; r < 3; r++) fwf(idx + r, c) = A(r, 0) * fw(idx, c) + A(r, 1) * fw(idx + 1, c) + A(r, 2) * fw(idx + 2, c);
  }
  // S2 block (DOF 21..22, DIM 21..23)
  {
    const int idx = 21;
    Vec3 seg;
    for (int i = 0; i < 3; i++) seg(i, 0) = f[idx + i] * dt;
    const double z2[2] = {0, 0};
    const Mat3 E = quat_to_rot(exp_quat(seg, double(1 / 2)));      // identity (esekfom.hpp:344)
    const Mat<2, 3> Nx = x_.grav.Nx_yy();
    const Mat<3, 2> Mx = x_before.grav.Mx(z2);
    const Mat<2, 3> NE = Nx * E;
    const Mat<2, 2> blk = NE * Mx;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) F1(idx + i, idx + j) = blk(i, j);
    const Mat<2, 3> T = -1.0 * (NE * hat(x_before.grav.vec) * A_matrix(seg).T());
    for (int c = 0; c < kDof; c++)
      for (int r = 0; r < 2; r++) fxf(idx + r, c) = T(r, 0) * fx(idx, c) + T(r, 1) * fx(idx + 1, c) + T(r, 2) * fx(idx + 2, c);
    for (int c = 0; c < 12; c++)
      for (int r = 0; r < 2; r++) fwf(idx + r, c) = T(r, 0) * fw(idx, c) + T(r, 1) * fw(idx + 1, c) + T(r, 2) * fw(idx + 2, c);
  }
  F1 = F1 + dt * fxf;
  const Mat<kDof, 12> G = dt * fwf;
  // P_ = F1 * P_ * F1.T() + G * Q * G.T(), the sums in the order the dense products take them (k ascending) with the terms whose
  // factor of F1 / G is exactly zero left out -- adding 0 * p changes no partial sum (F1 is the identity plus a few 3 x 3 blocks,
  // G has twelve rows): a sample's covariance propagation in 2 us instead of 9, twenty samples between two sweeps
  int nzF[kDof][kDof], nF[kDof], nzG[kDof][12], nG[kDof];
  for (int i = 0; i < kDof; i++) {
    nF[i] = 0; nG[i] = 0;
    for (int k = 0; k < kDof; k++) if (F1(i, k) != 0.0) nzF[i][nF[i]++] = k;
    for (int k = 0; k < 12; k++) if (G(i, k) != 0.0) nzG[i][nG[i]++] = k;
  }
  Cov T;                                                      // F1 * P_
  for (int i = 0; i < kDof; i++) {
    double row[kDof];
    for (int j = 0; j < kDof; j++) row[j] = 0.0;
    for (int q = 0; q < nF[i]; q++) {
      const int k = nzF[i][q];
      const double fik = F1(i, k);
      for (int j = 0; j < kDof; j++) row[j] += fik * P_(k, j);
    }
    for (int j = 0; j < kDof; j++) T(i, j) = row[j];
  }
  Mat<kDof, 12> GQ;                                           // G * Q
  for (int i = 0; i < kDof; i++)
    for (int j = 0; j < 12; j++) {
      double sum = 0.0;
      for (int q = 0; q < nG[i]; q++) sum += G(i, nzG[i][q]) * Q(nzG[i][q], j);
      GQ(i, j) = sum;
    }
  Cov Pn;
  for (int i = 0; i < kDof; i++)
    for (int j = 0; j < kDof; j++) {
      double a = 0.0, b = 0.0;
      for (int q = 0; q < nF[j]; q++) a += T(i, nzF[j][q]) * F1(j, nzF[j][q]);        // (T * F1^T)(i, j)
      for (int q = 0; q < nG[j]; q++) b += GQ(i, nzG[j][q]) * G(j, nzG[j][q]);        // (GQ * G^T)(i, j)
      Pn(i, j) = a + b;
    }
  P_ = Pn;
}

// (AVX2 clone chosen at load time where the CPU has it: the 12- and 23-wide inner loops run four doubles at a time; no FMA in either
//  clone -- the sums keep their order and rounding)
__attribute__((target_clones("avx2", "default")))
void Esekf::update_iterated_dyn_share_modified(double R, double D) {
  const int n = kDof;
  log.clear();
  int t = 0;
  const StateIkfom x_prop = x_;
  const Cov P_prop = P_;
  double K_h[kDof];
  Cov K_x = Cov::zero();
  double dx_new[kDof];
  ReducedMeas meas;
  static const int so3_idx[2] = {3, 6};
  failed = false;
  struct AtExit { std::function<void()>& f; ~AtExit() { if (f) f(); } } at_exit{h_update_end};

  // The whole loop on the measurement side, when it offers that: every iteration below (same arithmetic, flimo_ieskf.hip) enqueued
  // at once.  It may hand the loop back at any iteration: the rare branches stay here.
  int it0 = -1;
  bool inject = false;                                         // the first iteration of the loop below already has its pass's sums
  if (device_chain) {
    ChainResult r;
    double x26[26];
    x_.to_flat(x26);
    device_chain(x26, P_, limit_, R, D, maximum_iter_, r);
    if (failed) return;                                        // (x_, P_ untouched: still the propagated values)
    if (r.status == 2) {
      x_.from_flat(r.x);
      it0 = r.it_next;
      t = r.t;
      if (r.have_meas) { inject = true; meas = r.meas; }
    }
  }

  for (int it = it0; it < maximum_iter_; it++) {
    // The part of the iteration that does not depend on the measurement (:1652-1697: x boxminus x_propagated, P through the
    // manifold Jacobians) runs while the pass is in flight on the GPU when the plug-in offers that; the values are the same.
    double dx[kDof];
    bool pre_done = false;
    // what the default gain needs of A = P_ / R (measurement-independent: formed in pre(), beside the pass): PR = A[:, 0:12],
    // Ai = A11^-1, G2 = A21 A11^-1
    double PR[kDof][12], G2[kDof - 12][12];
    Mat<12, 12> Ai, Nm;
    auto pre = [&]() {
    pre_done = true;
    x_.boxminus(dx, x_prop);                                   // :1652
    for (int i = 0; i < n; i++) dx_new[i] = dx[i];
    P_ = P_prop;                                               // :1655

    for (int s = 0; s < 2; s++) {                              // :1659-1674
      const int idx = so3_idx[s];
      Vec3 seg;
      for (int i = 0; i < 3; i++) seg(i, 0) = dx[idx + i];
      const Mat3 J = A_matrix(seg).T();
      double tv[3];
      for (int r = 0; r < 3; r++) tv[r] = J(r, 0) * dx_new[idx] + J(r, 1) * dx_new[idx + 1] + J(r, 2) * dx_new[idx + 2];
      for (int r = 0; r < 3; r++) dx_new[idx + r] = tv[r];
      left_block<3, kDof>(P_, idx, J, n);
      right_block_T<3, kDof>(P_, idx, J);
    }
    {                                                          // :1676-1697
      const int idx = 21;
      const Mat<2, 2> J = x_.grav.Nx_yy() * x_prop.grav.Mx(dx + idx);
      double tv[2];
      for (int r = 0; r < 2; r++) tv[r] = J(r, 0) * dx_new[idx] + J(r, 1) * dx_new[idx + 1];
      for (int r = 0; r < 2; r++) dx_new[idx + r] = tv[r];
      left_block<2, kDof>(P_, idx, J, n);
      right_block_T<2, kDof>(P_, idx, J);
    }
    if (!reference_solve) {
      for (int i = 0; i < n; i++)
        for (int k = 0; k < 12; k++) PR[i][k] = P_(i, k) / R;
      Mat<12, 12> A11;
      for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) A11(i, j) = PR[i][j];
      if (!inverse_gj(12, &A11.a[0][0], &Ai.a[0][0])) inverse<12>(A11, Ai);
      for (int i = 0; i < n - 12; i++)
        for (int j = 0; j < 12; j++) { double a = 0; for (int k = 0; k < 12; k++) a += PR[12 + i][k] * Ai(k, j); G2[i][j] = a; }
    }
    };
    if (inject) inject = false;                                 // (the device's pass at this very state)
    else {
      if (it == maximum_iter_ - 1 && h_last_iteration) h_last_iteration();      // (no pass can follow this one)
      if (h_reduced_overlap) h_reduced_overlap(x_, meas, pre);    // esekfom.hpp:1637
      else h_reduced(x_, meas);
    }
    if (failed) { x_ = x_prop; P_ = P_prop; return; }          // the pass did not happen: nothing to update with
    if (!pre_done) pre();
    const int M = meas.M;

    bool lemma_step = false;
    double dx_lemma[kDof];
    Mat<12, 12> HTH = Mat<12, 12>::zero();                      // defined as 0 when M < 23 (a-note 5)
    double HTh[12];
    for (int i = 0; i < 12; i++) HTh[i] = 0.0;

    if (n > M) {                                               // :1701-1709
      DenseMeas dm;
      if (M > 0 && h_dense) h_dense(dm);
      const double* H = dm.H.data();
      std::vector<double> PHt((size_t)n * M), S((size_t)M * M), Sinv((size_t)M * M), K((size_t)n * M);
      for (int i = 0; i < n; i++) for (int j = 0; j < M; j++) { double s = 0; for (int k = 0; k < 12; k++) s += P_(i, k) * H[(size_t)j * 12 + k]; PHt[(size_t)i * M + j] = s; }
      for (int i = 0; i < M; i++) for (int j = 0; j < M; j++) { double s = 0; for (int k = 0; k < 12; k++) s += H[(size_t)i * 12 + k] * PHt[(size_t)k * M + j]; S[(size_t)i * M + j] = s / R + (i == j ? 1.0 : 0.0); }
      if (M > 0) inverse_lu(M, S.data(), Sinv.data());
      for (int i = 0; i < n; i++) for (int j = 0; j < M; j++) { double s = 0; for (int k = 0; k < M; k++) s += PHt[(size_t)i * M + k] * Sinv[(size_t)k * M + j]; K[(size_t)i * M + j] = s / R; }
      for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < M; k++) s += K[(size_t)i * M + k] * dm.h[k]; K_h[i] = s; }
      K_x = Cov::zero();
      for (int i = 0; i < n; i++) for (int j = 0; j < 12; j++) { double s = 0; for (int k = 0; k < M; k++) s += K[(size_t)i * M + k] * H[(size_t)k * 12 + j]; K_x(i, j) = s; }
    } else {                                                   // :1722-1729
      for (int i = 0; i < 12; i++) { HTh[i] = meas.HTh[i]; for (int j = 0; j < 12; j++) HTH(i, j) = meas.HTH[i * 12 + j]; }
      if (reference_solve) {
        // literal form of esekfom.hpp:1722-1729: two general 23x23 inverses
        Cov P_temp, P_inv, PR;
        for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) PR(i, j) = P_(i, j) / R;
        inverse<kDof>(PR, P_temp);
        for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) P_temp(i, j) += HTH(i, j);
        inverse<kDof>(P_temp, P_inv);
        for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < 12; k++) s += P_inv(i, k) * HTh[k]; K_h[i] = s; }
        K_x = Cov::zero();
        for (int i = 0; i < n; i++) for (int j = 0; j < 12; j++) { double s = 0; for (int k = 0; k < 12; k++) s += P_inv(i, k) * HTH(k, j); K_x(i, j) = s; }
      } else {
        // The reference's formula,  P_inv = ((P/R)^-1 + E B E^T)^-1  with A = P/R and B = H^T H (only its 12x12 block is non-zero),
        // through the block-inverse identity
        //     P_inv E = [ I ; A21 A11^-1 ] (A11^-1 + B)^-1 :
        // the Schur complement of A^-1's lower block IS A11^-1, so the 12 x 12 system N = A11^-1 + B is exactly what the two
        // 23 x 23 inverses of :1722,1726 solve -- as accurate on the measured states as the literal form (1e-15 of a 3e-2 m step)
        // where the round-3 form A[:, 0:12] (I + B A11)^-1 lost three digits to the 1 that drowns in B A11 ~ 1e7
        // (tests/test_host_logic.py: ..._against_80_bit_arithmetic).  A11^-1 and G2 = A21 A11^-1 do not depend on the measurement:
        // the device filter forms them beside the pass (csrc/hip/flimo_ieskf.h: ik_pre_block); N z = v is the only solve between the
        // pass's sums and the step.  Same operations in the same order on both sides.
        // (PR, Ai, G2: pre(), while the pass was in flight)
        for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) Nm(i, j) = Ai(i, j) + HTH(i, j);
        // The step of :1733, dx_ = K_h + (K_x - I) dx_new with K_h = W H^T h, K_x = W H^T H (W = P_inv E), taken as
        //   v = H^T h + H^T H dx_new[0:12];  N z = v;  dx_ = [z; G2 z] - dx_new
        // (one 12-vector solve between the sums and the step: the device filter's critical path)
        lemma_step = true;
        double v[12], z[12];
        for (int i = 0; i < 12; i++) { double a = 0; for (int k = 0; k < 12; k++) a += HTH(i, k) * dx_new[k]; v[i] = HTh[i] + a; }
        if (!solve_gj(12, &Nm.a[0][0], v, z)) {
          Mat<12, 12> Ninv;
          (void)(inverse_gj(12, &Nm.a[0][0], &Ninv.a[0][0]) || inverse<12>(Nm, Ninv));
          for (int m = 0; m < 12; m++) { double a = 0; for (int k = 0; k < 12; k++) a += Ninv(m, k) * v[k]; z[m] = a; }
        }
        for (int i = 0; i < 12; i++) dx_lemma[i] = z[i] - dx_new[i];
        for (int i = 12; i < n; i++) { double a = 0; for (int m = 0; m < 12; m++) a += G2[i - 12][m] * z[m]; dx_lemma[i] = a - dx_new[i]; }
        // (K_x is only read by the covariance update of the iteration that ends the loop: formed there)
      }
    }

    double dx_[kDof];                                          // :1733
    for (int i = 0; i < n; i++) {
      if (lemma_step) { dx_[i] = dx_lemma[i]; continue; }
      double s = 0;
      for (int k = 0; k < n; k++) s += (K_x(i, k) - (i == k ? 1.0 : 0.0)) * dx_new[k];
      dx_[i] = K_h[i] + s;
    }

    // degeneracy handling :1736-1744 (row-zeroing "projector" kept as in the reference).
    // When every eigenvalue of HTH[0:6,0:6] is >= D the projector VEPs^-1 * VEPs is the identity
    // (to rounding), so the eigen-decomposition is only run when the cheap test "HTH6 - D*I is
    // positive definite" (6x6 Cholesky) fails.
    double dx_nd[kDof];
    for (int i = 0; i < n; i++) dx_nd[i] = dx_[i];
    bool well_conditioned = !reference_solve;
    if (well_conditioned) {
      double Lc[6][6];
      for (int i = 0; i < 6 && well_conditioned; i++)
        for (int j = 0; j <= i; j++) {
          double s = 0.5 * (HTH(i, j) + HTH(j, i)) - (i == j ? D : 0.0);
          for (int k = 0; k < j; k++) s -= Lc[i][k] * Lc[j][k];
          if (i == j) {
            if (!(s > 1e-9 * D)) { well_conditioned = false; break; }
            Lc[i][i] = std::sqrt(s);
          } else {
            Lc[i][j] = s / Lc[j][j];
          }
        }
    }
    if (!well_conditioned) {
      Mat<6, 6> S6, V, Vinv, sel;
      for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) S6(i, j) = HTH(i, j);
      double w[6], wim[6];
      eigen_solver6(S6, w, wim, V);                              // eigenvalues().real(), eigenvectors().real() in Eigen's order
      double prod = w[0];
      for (int i = 1; i < 6; i++) prod *= w[i];
      if (prod < 1e-20) V = Mat<6, 6>::identity();
      sel = V;
      for (int v = 0; v < 6; v++) if (w[v] < D) for (int j = 0; j < 6; j++) sel(v, j) *= 0;
      inverse<6>(V, Vinv);
      const Mat<6, 6> Pm = Vinv * sel;
      for (int i = 0; i < 6; i++) { double s = 0; for (int k = 0; k < 6; k++) s += Pm(i, k) * dx_[k]; dx_nd[i] = s; }
    }

    x_.boxplus(dx_nd);                                         // :1747
    bool converge = true;
    for (int i = 0; i < n; i++) if (std::fabs(dx_[i]) > limit_[i]) { converge = false; break; }
    if (converge) t++;

    if (keep_log) {
      PassLog lg;
      lg.M = M;
      for (int i = 0; i < 12; i++) { lg.HTh[i] = HTh[i]; for (int j = 0; j < 12; j++) lg.HTH[i * 12 + j] = HTH(i, j); }
      for (int i = 0; i < n; i++) lg.dx[i] = dx_[i];
      x_.to_flat(lg.x_after);
      log.push_back(lg);
    }

    if (t > 1 || it == maximum_iter_ - 1) {                    // :1764-1820
      if (lemma_step) {
        // K_x = P_inv E B = [Ninv; G2 Ninv] B
        Mat<12, 12> Ninv;
        (void)(inverse_gj(12, &Nm.a[0][0], &Ninv.a[0][0]) || inverse<12>(Nm, Ninv));
        double W[kDof][12];
        for (int i = 0; i < 12; i++) for (int j = 0; j < 12; j++) W[i][j] = Ninv(i, j);
        for (int i = 12; i < n; i++)
          for (int j = 0; j < 12; j++) { double a = 0; for (int k = 0; k < 12; k++) a += G2[i - 12][k] * Ninv(k, j); W[i][j] = a; }
        K_x = Cov::zero();
        for (int i = 0; i < n; i++) {
          double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (int k = 0; k < 12; k++) { const double wk = W[i][k]; for (int j = 0; j < 12; j++) acc[j] += wk * HTH(k, j); }
          for (int j = 0; j < 12; j++) K_x(i, j) = acc[j];
        }
      }
      Cov L = P_;
      for (int s = 0; s < 2; s++) {
        const int idx = so3_idx[s];
        Vec3 seg;
        for (int i = 0; i < 3; i++) seg(i, 0) = dx_[idx + i];
        const Mat3 J = A_matrix(seg).T();
        for (int c = 0; c < n; c++)
          for (int r = 0; r < 3; r++) L(idx + r, c) = J(r, 0) * P_(idx, c) + J(r, 1) * P_(idx + 1, c) + J(r, 2) * P_(idx + 2, c);
        left_block<3, kDof>(K_x, idx, J, 12);
        right_block_T<3, kDof>(L, idx, J);
        right_block_T<3, kDof>(P_, idx, J);
      }
      {
        const int idx = 21;
        const Mat<2, 2> J = x_.grav.Nx_yy() * x_prop.grav.Mx(dx_ + idx);
        for (int c = 0; c < n; c++)
          for (int r = 0; r < 2; r++) L(idx + r, c) = J(r, 0) * P_(idx, c) + J(r, 1) * P_(idx + 1, c);
        left_block<2, kDof>(K_x, idx, J, 12);
        right_block_T<2, kDof>(L, idx, J);
        right_block_T<2, kDof>(P_, idx, J);
      }
      Cov Pn;
      for (int i = 0; i < n; i++) {
        double acc[kDof];
        for (int j = 0; j < n; j++) acc[j] = 0.0;
        for (int k = 0; k < 12; k++) { const double kk = K_x(i, k); for (int j = 0; j < n; j++) acc[j] += kk * P_(k, j); }
        for (int j = 0; j < n; j++) Pn(i, j) = L(i, j) - acc[j];
      }
      P_ = Pn;
      return;
    }
  }
}

}  // namespace flimo_host
