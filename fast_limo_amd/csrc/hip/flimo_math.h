// fast_limo_amd/csrc/hip/flimo_math.h
// Device-side float32 geometry of the registration hot path, written for gfx950.
//
// Every function here must produce results that are BIT-IDENTICAL to the reference's float32
// arithmetic (x86-64 SSE2, no FMA: CMakeLists.txt:4-5,17-21), so
//   * this header is compiled with -ffp-contract=off and the pragma below,
//   * sqrt and divide are the correctly rounded forms (fl_sqrt / fl_div below),
//   * evaluation order follows Eigen's expression evaluation as documented per function.
#pragma once
#include <hip/hip_runtime.h>
#include "flimo_types.h"

#pragma clang fp contract(off)

namespace flimo {

#define FLIMO_DEV __host__ __device__ __forceinline__

// Correctly rounded sqrt / divide.  NOTE: HIP's __fsqrt_rn maps to __ocml_native_sqrt_f32 (1 ulp,
// measured 16 % mismatches against IEEE on gfx950), so it must NOT be used here; __builtin_sqrtf
// and operator/ are lowered to the correctly rounded sequences (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt) -- verified bit-exact on the GPU by tools/devmath_check.
FLIMO_DEV float fl_sqrt(float x) { return __builtin_sqrtf(x); }
FLIMO_DEV float fl_div(float a, float b) { return a / b; }

// sinf / cosf as the host's libm computes them (the reference's State::update calls std::sin / std::cos on a float,
// State.cpp:88-90).  glibc >= 2.28 evaluates both in double: for |x| < pi/4 odd / even polynomials of degree 7 / 8, beyond that
// the reduction x - n * pi/2 with n = round(x * 2/pi) and the polynomial the quadrant selects; |x| < 2^-12 returns x / 1.
// Restated here with the published coefficients (ARM optimized-routines sincosf, the source of glibc's s_sincosf_data.c); checked
// against the host's libm on 10M arguments by tools/devmath_check (no mismatch for |x| < pi/4, 3e-7 of the arguments up to 100 rad,
// where libm's FMA build contracts the reduction).  |x| >= 120 (never a deskew angle) falls back to the device's own sinf / cosf.
FLIMO_DEV void libm_sincosf(float x, float& s_out, float& c_out) {
  const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
  const double C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
  const float ax = __builtin_fabsf(x);
  if (!(ax < 120.0f)) { s_out = __builtin_sinf(x); c_out = __builtin_cosf(x); return; }
  if (ax < 0x1p-12f) { s_out = x; c_out = 1.0f; return; }
  double xr = (double)x;
  int n = 0;
  if (!(ax < 0x1.921fb6p-1f)) {
    const double r = xr * 0x1.45F306DC9C883p+23;
    n = ((int)r + 0x800000) >> 24;
    xr = xr - (double)n * 0x1.921FB54442D18p0;
  }
  const double x2 = xr * xr, x4 = x2 * x2;
  const double x3 = xr * x2, x7 = x3 * x2, x6 = x4 * x2;
  const double ps = (xr + x3 * S1) + x7 * (S2 + x2 * S3);                    // sin(xr)
  const double pc = ((1.0 + x2 * C1) + x4 * C2) + x6 * (C3 + x2 * C4);      // cos(xr)
  const double sv = (n & 1) ? pc : ps, cv = (n & 1) ? ps : pc;
  s_out = (float)((n & 2) ? -sv : sv);
  c_out = (float)(((n + 1) & 2) ? -cv : cv);
}

// std::atan2 of two floats as the host's libm evaluates it (the reference's FoV filter: fabs(atan2(p.y, p.x)) < fov_angle,
// Localizer.cpp:873-876).  glibc's atan2f up to 2.40 is the fdlibm routine in float arithmetic (e_atan2f.c over s_atanf.c:
// argument reduction to [0, 7/16] by four breakpoints, odd/even split of an 11-term polynomial, hi/lo table of atan at the
// breakpoints); restated here operation for operation (plain IEEE float ops, no contraction), checked against the host's libm on
// 16M argument pairs by tools/devmath_check.
FLIMO_DEV float libm_atanf(float x) {
  const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
  const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
  const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                        6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
  const unsigned int hxu = __builtin_bit_cast(unsigned int, x);
  const int hx = (int)hxu;
  const int ix = hx & 0x7fffffff;
  int id;
  if (ix >= 0x4c000000) {                       // |x| >= 2^25
    if (ix > 0x7f800000) return x + x;          // NaN
    return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {                        // |x| < 0.4375
    if (ix < 0x31000000) return x;              // |x| < 2^-29
    id = -1;
  } else {
    x = __builtin_fabsf(x);
    if (ix < 0x3f980000) {                      // |x| < 1.1875
      if (ix < 0x3f300000) { id = 0; x = fl_div(2.0f * x - 1.0f, 2.0f + x); }       // 7/16 <= |x| < 11/16
      else { id = 1; x = fl_div(x - 1.0f, x + 1.0f); }                             // 11/16 <= |x| < 19/16
    } else {
      if (ix < 0x401c0000) { id = 2; x = fl_div(x - 1.5f, 1.0f + 1.5f * x); }       // |x| < 2.4375
      else { id = 3; x = fl_div(-1.0f, x); }
    }
  }
  const float z = x * x;
  const float w = z * z;
  const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  const float hi = id == 0 ? atanhi[0] : (id == 1 ? atanhi[1] : (id == 2 ? atanhi[2] : atanhi[3]));
  const float lo = id == 0 ? atanlo[0] : (id == 1 ? atanlo[1] : (id == 2 ? atanlo[2] : atanlo[3]));
  const float r = hi - ((x * (s1 + s2) - lo) - x);
  return hx < 0 ? -r : r;
}
FLIMO_DEV float libm_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  const int hx = (int)__builtin_bit_cast(unsigned int, x), hy = (int)__builtin_bit_cast(unsigned int, y);
  const int ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;              // NaN
  if (hx == 0x3f800000) return libm_atanf(y);                         // x = 1
  const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                 // 2 * sign(x) + sign(y)
  if (iy == 0) {
    if (m == 0 || m == 1) return y;
    return m == 2 ? pi + tiny : -pi - tiny;
  }
  if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000) {
    if (iy == 0x7f800000) {
      if (m == 0) return pi_o_4 + tiny;
      if (m == 1) return -pi_o_4 - tiny;
      if (m == 2) return 3.0f * pi_o_4 + tiny;
      return -3.0f * pi_o_4 - tiny;
    }
    if (m == 0) return 0.0f;
    if (m == 1) return -0.0f;
    if (m == 2) return pi + tiny;
    return -pi - tiny;
  }
  if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
  const int k = (iy - ix) >> 23;
  float z;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                              // |y/x| > 2^60
  else if (hx < 0 && k < -60) z = 0.0f;                               // |y|/x < -2^60
  else z = libm_atanf(__builtin_fabsf(fl_div(y, x)));
  if (m == 0) return z;
  if (m == 1) return -z;
  if (m == 2) return pi - (z - pi_lo);
  return (z - pi_lo) - pi;
}

// 3-coefficient Eigen reduction: c0 + (c1 + c2)
FLIMO_DEV float sum3(float a, float b, float c) { return a + (b + c); }

// (q - p).squaredNorm()  -- reference Objects/Octree.hpp:572
FLIMO_DEV float sqdist3(float qx, float qy, float qz, float px, float py, float pz) {
  float dx = qx - px, dy = qy - py, dz = qz - pz;
  return sum3(dx * dx, dy * dy, dz * dz);
}

// Matrix4f * Vector4f(x,y,z,1): ((c0*x + c1*y) + c2*z) + c3*1   (Mapper.cpp:71-72, Localizer.cpp:549-550)
FLIMO_DEV void xform4(const float* M, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = ((M[0] * x + M[1] * y) + M[2] * z) + M[3];
  oy = ((M[4] * x + M[5] * y) + M[6] * z) + M[7];
  oz = ((M[8] * x + M[9] * y) + M[10] * z) + M[11];
}

// Matrix3f * Vector3f, coefficient-based: row . v = r0*v0 + (r1*v1 + r2*v2)   (Localizer.cpp:564-565)
FLIMO_DEV void mul3(const float* R, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = sum3(R[0] * x, R[1] * y, R[2] * z);
  oy = sum3(R[3] * x, R[4] * y, R[5] * z);
  oz = sum3(R[6] * x, R[7] * y, R[8] * z);
}

FLIMO_DEV void cross3(float ax, float ay, float az, float bx, float by, float bz, float& ox, float& oy, float& oz) {
  ox = ay * bz - az * by;
  oy = az * bx - ax * bz;
  oz = ax * by - ay * bx;
}

// ------------------------------------------------------------------------------------------
// Plane::estimate_plane (Objects/Plane.cpp:80-105): least squares A x = -1 on the 5 neighbours by
// column-pivoted Householder QR (Eigen::ColPivHouseholderQR::compute + solve), then
// n = x/|x|, d = 1/|x|.  Fully unrolled 5x3, registers only.  px/py/pz: the 5 neighbours in
// ascending-distance order.  Returns n_ABCD in n[4].
// ------------------------------------------------------------------------------------------
FLIMO_DEV void plane_fit5(const float (&px)[5], const float (&py)[5], const float (&pz)[5], float (&n)[4]) {
  float a[3][5];  // a[col][row]
#pragma unroll
  for (int i = 0; i < 5; i++) { a[0][i] = px[i]; a[1][i] = py[i]; a[2][i] = pz[i]; }
  float nU[3], nD[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; i++) s = s + a[k][i] * a[k][i];
    nD[k] = fl_sqrt(s);
    nU[k] = nD[k];
  }
  const float eps = 1.1920929e-07f;
  float maxn = nU[0];
  if (nU[1] > maxn) maxn = nU[1];
  if (nU[2] > maxn) maxn = nU[2];
  const float th = maxn * eps;
  const float threshold_helper = fl_div(th * th, 5.0f);
  const float downdate_thr = fl_sqrt(eps);
  int nzp = 3;
  int perm0 = 0, perm1 = 1, perm2 = 2;
  float hC[3];

#pragma unroll
  for (int k = 0; k < 3; k++) {
    // pivot: first column of largest updated norm among k..2
    int big = k;
    float bign = nU[k];
#pragma unroll
    for (int j = k + 1; j < 3; j++)
      if (nU[j] > bign) { bign = nU[j]; big = j; }
    const float big_sq = bign * bign;
    if (nzp == 3 && big_sq < threshold_helper * float(5 - k)) nzp = k;
    // column swap k <-> big (register selects, no dynamic indexing)
#pragma unroll
    for (int j = k + 1; j < 3; j++) {
      const bool sw = (big == j);
#pragma unroll
      for (int i = 0; i < 5; i++) { float t0 = a[k][i], t1 = a[j][i]; a[k][i] = sw ? t1 : t0; a[j][i] = sw ? t0 : t1; }
      { float t0 = nU[k], t1 = nU[j]; nU[k] = sw ? t1 : t0; nU[j] = sw ? t0 : t1; }
      { float t0 = nD[k], t1 = nD[j]; nD[k] = sw ? t1 : t0; nD[j] = sw ? t0 : t1; }
      // perm: swap(perm[k], perm[big])
      if (k == 0) {
        if (j == 1) { int t0 = perm0, t1 = perm1; perm0 = sw ? t1 : t0; perm1 = sw ? t0 : t1; }
        if (j == 2) { int t0 = perm0, t1 = perm2; perm0 = sw ? t1 : t0; perm2 = sw ? t0 : t1; }
      } else if (k == 1) {
        if (j == 2) { int t0 = perm1, t1 = perm2; perm1 = sw ? t1 : t0; perm2 = sw ? t0 : t1; }
      }
    }
    // Householder vector of a[k][k..4]
    float tailSq = 0.f;
#pragma unroll
    for (int i = k + 1; i < 5; i++) tailSq = tailSq + a[k][i] * a[k][i];
    const float c0 = a[k][k];
    float tau, beta;
    const float tol = 1.17549435e-38f;
    if (tailSq <= tol) {
      tau = 0.f;
      beta = c0;
#pragma unroll
      for (int i = k + 1; i < 5; i++) a[k][i] = 0.f;
    } else {
      beta = fl_sqrt(c0 * c0 + tailSq);
      if (c0 >= 0.f) beta = -beta;
      const float denom = c0 - beta;
#pragma unroll
      for (int i = k + 1; i < 5; i++) a[k][i] = fl_div(a[k][i], denom);
      tau = fl_div(beta - c0, beta);
    }
    hC[k] = tau;
    a[k][k] = beta;
    // apply H_k to the remaining columns
    if (tau != 0.f) {
#pragma unroll
      for (int j = k + 1; j < 3; j++) {
        float tmp = 0.f;
#pragma unroll
        for (int i = k + 1; i < 5; i++) tmp = tmp + a[k][i] * a[j][i];
        tmp = tmp + a[j][k];
        a[j][k] = a[j][k] - tau * tmp;
#pragma unroll
        for (int i = k + 1; i < 5; i++) a[j][i] = a[j][i] - (tau * a[k][i]) * tmp;
      }
    }
    // column-norm downdate (LAPACK xGEQPF style)
#pragma unroll
    for (int j = k + 1; j < 3; j++) {
      if (nU[j] != 0.f) {
        float temp = fl_div(fabsf(a[j][k]), nU[j]);
        temp = (1.f + temp) * (1.f - temp);
        temp = temp < 0.f ? 0.f : temp;
        const float ratio = fl_div(nU[j], nD[j]);
        const float temp2 = temp * (ratio * ratio);
        if (temp2 <= downdate_thr) {
          float s = 0.f;
#pragma unroll
          for (int i = k + 1; i < 5; i++) s = s + a[j][i] * a[j][i];
          nD[j] = fl_sqrt(s);
          nU[j] = nD[j];
        } else {
          nU[j] = nU[j] * fl_sqrt(temp);
        }
      }
    }
  }

  // ---- solve  min |A x - b|, b = -1 ----
  float c[5] = {-1.f, -1.f, -1.f, -1.f, -1.f};
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float tau = hC[k];
    if (k < nzp && tau != 0.f) {
      float tmp = 0.f;
#pragma unroll
      for (int i = k + 1; i < 5; i++) tmp = tmp + a[k][i] * c[i];
      tmp = tmp + c[k];
      c[k] = c[k] - tau * tmp;
#pragma unroll
      for (int i = k + 1; i < 5; i++) c[i] = c[i] - (tau * a[k][i]) * tmp;
    }
  }
  // back substitution on the leading nzp x nzp upper triangle: R[i][j] = a[j][i]
#pragma unroll
  for (int i = 2; i >= 0; i--) {
    if (i < nzp) {
      float s = c[i];
#pragma unroll
      for (int j = i + 1; j < 3; j++)
        if (j < nzp) s = s - a[j][i] * c[j];
      c[i] = fl_div(s, a[i][i]);
    }
  }
  const float y0 = (0 < nzp) ? c[0] : 0.f;
  const float y1 = (1 < nzp) ? c[1] : 0.f;
  const float y2 = (2 < nzp) ? c[2] : 0.f;
  // x[perm[i]] = y[i]
  float x0 = 0.f, x1 = 0.f, x2 = 0.f;
  x0 = (perm0 == 0) ? y0 : x0; x1 = (perm0 == 1) ? y0 : x1; x2 = (perm0 == 2) ? y0 : x2;
  x0 = (perm1 == 0) ? y1 : x0; x1 = (perm1 == 1) ? y1 : x1; x2 = (perm1 == 2) ? y1 : x2;
  x0 = (perm2 == 0) ? y2 : x0; x1 = (perm2 == 1) ? y2 : x1; x2 = (perm2 == 2) ? y2 : x2;

  const float nn = fl_sqrt(sum3(x0 * x0, x1 * x1, x2 * x2));   // normvec.norm()
  n[0] = fl_div(x0, nn);
  n[1] = fl_div(x1, nn);
  n[2] = fl_div(x2, nn);
  n[3] = fl_div(1.0f, nn);   // (float)(1.0 / n): double rounding is innocuous for division
}

// The same for M neighbours (NUM_MATCH_POINTS 3..8, Mapper.cpp:106-109, Plane.cpp:41-43,80-105): Eigen's
// ColPivHouseholderQR::computeInPlace + solve on an M x 3 matrix, including the 1-row block at the last step of M = 3
// (makeHouseholder with an empty tail, applyHouseholderOnTheLeft on a single row).  The hot path keeps its unrolled 5 x 3 form.
template <int M>
FLIMO_DEV void plane_fit_m(const float (&px)[M], const float (&py)[M], const float (&pz)[M], float (&n)[4]) {
  static_assert(M >= 3 && M <= 8, "3..8 neighbours");
  float a[3][M];  // a[col][row]
#pragma unroll
  for (int i = 0; i < M; i++) { a[0][i] = px[i]; a[1][i] = py[i]; a[2][i] = pz[i]; }
  float nU[3], nD[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < M; i++) s = s + a[k][i] * a[k][i];
    nD[k] = fl_sqrt(s);
    nU[k] = nD[k];
  }
  const float eps = 1.1920929e-07f;
  float maxn = nU[0];
  if (nU[1] > maxn) maxn = nU[1];
  if (nU[2] > maxn) maxn = nU[2];
  const float th = maxn * eps;
  const float threshold_helper = fl_div(th * th, (float)M);
  const float downdate_thr = fl_sqrt(eps);
  int nzp = 3;
  int perm0 = 0, perm1 = 1, perm2 = 2;
  float hC[3];

#pragma unroll
  for (int k = 0; k < 3; k++) {
    int big = k;
    float bign = nU[k];
#pragma unroll
    for (int j = k + 1; j < 3; j++)
      if (nU[j] > bign) { bign = nU[j]; big = j; }
    const float big_sq = bign * bign;
    if (nzp == 3 && big_sq < threshold_helper * float(M - k)) nzp = k;
#pragma unroll
    for (int j = k + 1; j < 3; j++) {
      const bool sw = (big == j);
#pragma unroll
      for (int i = 0; i < M; i++) { float t0 = a[k][i], t1 = a[j][i]; a[k][i] = sw ? t1 : t0; a[j][i] = sw ? t0 : t1; }
      { float t0 = nU[k], t1 = nU[j]; nU[k] = sw ? t1 : t0; nU[j] = sw ? t0 : t1; }
      { float t0 = nD[k], t1 = nD[j]; nD[k] = sw ? t1 : t0; nD[j] = sw ? t0 : t1; }
      if (k == 0) {
        if (j == 1) { int t0 = perm0, t1 = perm1; perm0 = sw ? t1 : t0; perm1 = sw ? t0 : t1; }
        if (j == 2) { int t0 = perm0, t1 = perm2; perm0 = sw ? t1 : t0; perm2 = sw ? t0 : t1; }
      } else if (k == 1) {
        if (j == 2) { int t0 = perm1, t1 = perm2; perm1 = sw ? t1 : t0; perm2 = sw ? t0 : t1; }
      }
    }
    float tailSq = 0.f;
#pragma unroll
    for (int i = k + 1; i < M; i++) tailSq = tailSq + a[k][i] * a[k][i];
    const float c0 = a[k][k];
    float tau, beta;
    const float tol = 1.17549435e-38f;
    const bool one_row = (M - k) == 1;
    if (one_row || tailSq <= tol) {
      tau = 0.f;
      beta = c0;
#pragma unroll
      for (int i = k + 1; i < M; i++) a[k][i] = 0.f;
    } else {
      beta = fl_sqrt(c0 * c0 + tailSq);
      if (c0 >= 0.f) beta = -beta;
      const float denom = c0 - beta;
#pragma unroll
      for (int i = k + 1; i < M; i++) a[k][i] = fl_div(a[k][i], denom);
      tau = fl_div(beta - c0, beta);
    }
    hC[k] = tau;
    a[k][k] = beta;
    if (one_row) {
#pragma unroll
      for (int j = k + 1; j < 3; j++) a[j][k] = a[j][k] * (1.f - tau);
    } else if (tau != 0.f) {
#pragma unroll
      for (int j = k + 1; j < 3; j++) {
        float tmp = 0.f;
#pragma unroll
        for (int i = k + 1; i < M; i++) tmp = tmp + a[k][i] * a[j][i];
        tmp = tmp + a[j][k];
        a[j][k] = a[j][k] - tau * tmp;
#pragma unroll
        for (int i = k + 1; i < M; i++) a[j][i] = a[j][i] - (tau * a[k][i]) * tmp;
      }
    }
#pragma unroll
    for (int j = k + 1; j < 3; j++) {
      if (nU[j] != 0.f) {
        float temp = fl_div(fabsf(a[j][k]), nU[j]);
        temp = (1.f + temp) * (1.f - temp);
        temp = temp < 0.f ? 0.f : temp;
        const float ratio = fl_div(nU[j], nD[j]);
        const float temp2 = temp * (ratio * ratio);
        if (temp2 <= downdate_thr) {
          float s = 0.f;
#pragma unroll
          for (int i = k + 1; i < M; i++) s = s + a[j][i] * a[j][i];
          nD[j] = fl_sqrt(s);
          nU[j] = nD[j];
        } else {
          nU[j] = nU[j] * fl_sqrt(temp);
        }
      }
    }
  }

  float c[M];
#pragma unroll
  for (int i = 0; i < M; i++) c[i] = -1.f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float tau = hC[k];
    if (k < nzp) {
      if ((M - k) == 1) {
        c[k] = c[k] * (1.f - tau);
      } else if (tau != 0.f) {
        float tmp = 0.f;
#pragma unroll
        for (int i = k + 1; i < M; i++) tmp = tmp + a[k][i] * c[i];
        tmp = tmp + c[k];
        c[k] = c[k] - tau * tmp;
#pragma unroll
        for (int i = k + 1; i < M; i++) c[i] = c[i] - (tau * a[k][i]) * tmp;
      }
    }
  }
#pragma unroll
  for (int i = 2; i >= 0; i--) {
    if (i < nzp) {
      float s = c[i];
#pragma unroll
      for (int j = i + 1; j < 3; j++)
        if (j < nzp) s = s - a[j][i] * c[j];
      c[i] = fl_div(s, a[i][i]);
    }
  }
  const float y0 = (0 < nzp) ? c[0] : 0.f;
  const float y1 = (1 < nzp) ? c[1] : 0.f;
  const float y2 = (2 < nzp) ? c[2] : 0.f;
  float x0 = 0.f, x1 = 0.f, x2 = 0.f;
  x0 = (perm0 == 0) ? y0 : x0; x1 = (perm0 == 1) ? y0 : x1; x2 = (perm0 == 2) ? y0 : x2;
  x0 = (perm1 == 0) ? y1 : x0; x1 = (perm1 == 1) ? y1 : x1; x2 = (perm1 == 2) ? y1 : x2;
  x0 = (perm2 == 0) ? y2 : x0; x1 = (perm2 == 1) ? y2 : x1; x2 = (perm2 == 2) ? y2 : x2;
  const float nn = fl_sqrt(sum3(x0 * x0, x1 * x1, x2 * x2));
  n[0] = fl_div(x0, nn);
  n[1] = fl_div(x1, nn);
  n[2] = fl_div(x2, nn);
  n[3] = fl_div(1.0f, nn);
}

template <int M>
FLIMO_DEV bool plane_eval_m(const float (&n)[4], const float (&px)[M], const float (&py)[M], const float (&pz)[M], float thres) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < M; j++) {
    const float res = n[0] * px[j] + n[1] * py[j] + n[2] * pz[j] + n[3];
    ok = ok && !(fabsf(res) > thres);
  }
  return ok;
}

// Plane::plane_eval (Objects/Plane.cpp:107-114)
FLIMO_DEV bool plane_eval5(const float (&n)[4], const float (&px)[5], const float (&py)[5], const float (&pz)[5],
                           float thres) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const float res = n[0] * px[j] + n[1] * py[j] + n[2] * pz[j] + n[3];
    ok = ok && !(fabsf(res) > thres);
  }
  return ok;
}

// One row of Localizer::calculate_H (Localizer.cpp:546-572) for a match with world point g and plane n:
// v[0:3] = n, v[3:6] = p_imu x C, v[6:9] = p_lidar x (R_LI^-1 C), v[9:12] = C  (the last six only when the extrinsics
// are estimated, else 0), with C = R^-1 n.  Shared by the fit kernel and the host-side Localizer::calculate_H.
FLIMO_DEV void h_row(const PoseMats& P, float gx, float gy, float gz, const float (&n4)[4], int estimate_extrinsics,
                     float (&v)[12]) {
  float ix, iy, iz, lx, ly, lz;
  xform4(P.RT_inv, gx, gy, gz, ix, iy, iz);       // p_imu    (Localizer.cpp:549)
  xform4(P.TLI_inv, ix, iy, iz, lx, ly, lz);      // p_lidar  (Localizer.cpp:550)
  float Cx, Cy, Cz, Dx, Dy, Dz, Bx, By, Bz, Ax, Ay, Az;
  mul3(P.R_inv, n4[0], n4[1], n4[2], Cx, Cy, Cz); // C = R_inv * n
  mul3(P.RLI_inv, Cx, Cy, Cz, Dx, Dy, Dz);        // I_R_L_inv * C
  cross3(lx, ly, lz, Dx, Dy, Dz, Bx, By, Bz);     // B
  cross3(ix, iy, iz, Cx, Cy, Cz, Ax, Ay, Az);     // A
  v[0] = n4[0]; v[1] = n4[1]; v[2] = n4[2]; v[3] = Ax; v[4] = Ay; v[5] = Az;
  // element-wise selects, not a branch around six stores (the compiler turned that into a private-memory round trip on the device)
  const bool ee = estimate_extrinsics != 0;
  v[6] = ee ? Bx : 0.f; v[7] = ee ? By : 0.f; v[8] = ee ? Bz : 0.f;
  v[9] = ee ? Cx : 0.f; v[10] = ee ? Cy : 0.f; v[11] = ee ? Cz : 0.f;
}

}  // namespace flimo
