// so3.hpp -- device-side SO(3) / small fixed-size algebra for gfx950 (fp64: the small vector / matrix types are templates, every
// instantiation is double).
//
// Conventions follow the reference (quaternion storage x,y,z,w; right perturbation R*exp(d)):
//   exp/log          src/sophus_lib/so3.hpp:534-569 / 220-262
//   Jr / Jr^-1       src/utils/sophus_utils.hpp:166-199 / 210-242
//   q*q renormalise  src/sophus_lib/so3.hpp:338-355
// The same FUNCTIONS as the reference's, not the same branches: below |phi| = 0.5 rad (every knot-to-knot rotation of a usable
// spline) exp, Jr and Jr^-1 are evaluated by their Taylor series (truncation below 1e-17: the closed forms (1 - cos t) / t^2,
// (t - sin t) / t^3, 1 / t^2 - (1 + cos t) / (2 t sin t) lose ~1e-16 / t^2 to cancellation there and need sqrt / sin / cos), above it
// by the reference's closed forms; log keeps the reference's atan form and its 1e-10 thresholds.  tests/test_properties.py checks the
// identities and the continuity at the switch on this very header compiled for the host.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CTV_DI __host__ __device__ __forceinline__
#else  // plain g++ build of the same math, used ONLY by tests/host_math_check.cpp (not a product path)
#include <cmath>
#define CTV_DI inline
#endif

namespace ctv {

struct V3 { double x, y, z; };
struct Q4 { double x, y, z, w; };
struct M3 { double m[9]; };  // row-major

CTV_DI V3 mk(double x, double y, double z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
CTV_DI V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
CTV_DI V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
CTV_DI V3 operator*(double s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
CTV_DI V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
CTV_DI double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
CTV_DI V3 cross(V3 a, V3 b) {
  return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

CTV_DI M3 m3_id() { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = 0.0; r.m[0] = r.m[4] = r.m[8] = 1.0; return r; }
CTV_DI M3 m3_zero() { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = 0.0; return r; }
CTV_DI M3 mul(const M3 &A, const M3 &B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}
CTV_DI M3 mulT(const M3 &A, const M3 &B) {  // A * B^T
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[3 * j] + A.m[3 * i + 1] * B.m[3 * j + 1] + A.m[3 * i + 2] * B.m[3 * j + 2];
  return C;
}
CTV_DI V3 mul(const M3 &A, V3 v) {
  return mk(A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z,
               A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z);
}
CTV_DI M3 scale(const M3 &A, double s) { M3 C; for (int i = 0; i < 9; ++i) C.m[i] = s * A.m[i]; return C; }
CTV_DI M3 add(const M3 &A, const M3 &B) { M3 C; for (int i = 0; i < 9; ++i) C.m[i] = A.m[i] + B.m[i]; return C; }
CTV_DI M3 sub(const M3 &A, const M3 &B) { M3 C; for (int i = 0; i < 9; ++i) C.m[i] = A.m[i] - B.m[i]; return C; }
// hat: src/sophus_lib/so3.hpp:618-627
CTV_DI M3 hat(V3 w) {
  M3 H;
  H.m[0] = 0; H.m[1] = -w.z; H.m[2] = w.y; H.m[3] = w.z; H.m[4] = 0; H.m[5] = -w.x; H.m[6] = -w.y; H.m[7] = w.x; H.m[8] = 0;
  return H;
}
// A * hat(w)
CTV_DI M3 mul_hat(const M3 &A, V3 w) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double a0 = A.m[3 * i], a1 = A.m[3 * i + 1], a2 = A.m[3 * i + 2];
    C.m[3 * i] = a1 * w.z - a2 * w.y;
    C.m[3 * i + 1] = a2 * w.x - a0 * w.z;
    C.m[3 * i + 2] = a0 * w.y - a1 * w.x;
  }
  return C;
}

CTV_DI Q4 qmk(double x, double y, double z, double w) { Q4 q; q.x = x; q.y = y; q.z = z; q.w = w; return q; }
CTV_DI Q4 qconj(Q4 a) { return qmk(-a.x, -a.y, -a.z, a.w); }
CTV_DI Q4 qmul_raw(Q4 a, Q4 b) {
  return qmk(a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z);
}
CTV_DI Q4 qmul(Q4 a, Q4 b) {
  Q4 o = qmul_raw(a, b);
  const double n2 = o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w;
  const double s = 2.0 / (1.0 + n2);
  o.x *= s; o.y *= s; o.z *= s; o.w *= s;
  return o;
}
CTV_DI V3 qrot(Q4 q, V3 v) {
  const V3 qv = mk(q.x, q.y, q.z);
  V3 uv = cross(qv, v);
  uv = 2.0 * uv;
  return v + q.w * uv + cross(qv, uv);
}
CTV_DI M3 q2R(Q4 q) {
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3 R;
  R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz; R.m[2] = txz + twy;
  R.m[3] = txy + twz; R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
  R.m[6] = txz - twy; R.m[7] = tyz + twx; R.m[8] = 1 - (txx + tyy);
  return R;
}

// ---- exp: so3.hpp:534-569
CTV_DI Q4 so3_exp(V3 w) {
  const double th2 = dot(w, w);
  double im, re;
  if (th2 < 0.25) {
    // |theta| < 0.5 (lambda * d between neighbouring knots): Taylor series in h^2 = (theta/2)^2 up to h^14 / h^15,
    // truncation < 3e-21 relative -- the same function as the closed form to the last bit or two, without sin/cos/divide
    const double h2 = 0.25 * th2;
    im = 0.5 * (1.0 + h2 * (-1.0 / 6.0 + h2 * (1.0 / 120.0 + h2 * (-1.0 / 5040.0 + h2 * (1.0 / 362880.0 + h2 * (-1.0 / 39916800.0 +
         h2 * (1.0 / 6227020800.0 + h2 * (-1.0 / 1307674368000.0))))))));
    re = 1.0 + h2 * (-0.5 + h2 * (1.0 / 24.0 + h2 * (-1.0 / 720.0 + h2 * (1.0 / 40320.0 + h2 * (-1.0 / 3628800.0 +
         h2 * (1.0 / 479001600.0 + h2 * (-1.0 / 87178291200.0)))))));
  } else {
    const double th = sqrt(th2);
    im = sin(0.5 * th) / th;
    re = cos(0.5 * th);
  }
  return qmk(im * w.x, im * w.y, im * w.z, re);
}

// ---- log: so3.hpp:220-262 (atan form)
CTV_DI V3 so3_log(Q4 q) {
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z, n = sqrt(n2), w = q.w;
  double f;
  if (n < 1e-10) f = 2.0 / w - 2.0 * n2 / (w * w * w);
  else if (fabs(w) < 1e-10) f = (w > 0 ? 3.14159265358979323846 : -3.14159265358979323846) / n;
  else f = 2.0 * atan(n / w) / n;
  return mk(f * q.x, f * q.y, f * q.z);
}

// ---- Jr: I - a*hat + b*hat^2,  a = (1-cos t)/t^2, b = (t - sin t)/t^3   (sophus_utils.hpp:166-199)
CTV_DI void jr_coeffs(double n2, double &a, double &b) {
  if (n2 < 0.25) {
    // |t| < 0.5 (lambda * d between neighbouring knots): alternating Taylor series, truncation < 1e-17 relative -- the same
    // function as the reference's closed form (which itself loses ~1e-16 / t^2 to cancellation), without sqrt / sin / cos
    a = 0.5 + n2 * (-1.0 / 24.0 + n2 * (1.0 / 720.0 + n2 * (-1.0 / 40320.0 + n2 * (1.0 / 3628800.0 + n2 * (-1.0 / 479001600.0 +
        n2 * (1.0 / 87178291200.0 + n2 * (-1.0 / 20922789888000.0)))))));
    b = 1.0 / 6.0 + n2 * (-1.0 / 120.0 + n2 * (1.0 / 5040.0 + n2 * (-1.0 / 362880.0 + n2 * (1.0 / 39916800.0 + n2 * (-1.0 / 6227020800.0 +
        n2 * (1.0 / 1307674368000.0 + n2 * (-1.0 / 355687428096000.0)))))));
  } else { const double n = sqrt(n2); a = (1 - cos(n)) / n2; b = (n - sin(n)) / (n2 * n); }
}
// (hat^2 = phi phi^T - |phi|^2 I written out: 19 operations instead of a 3 x 3 product and 18 more)
CTV_DI M3 so3_Jr(V3 phi) {
  double a, b;
  const double n2 = dot(phi, phi);
  jr_coeffs(n2, a, b);
  const double bx = b * phi.x, by = b * phi.y, bz = b * phi.z, ax = a * phi.x, ay = a * phi.y, az = a * phi.z, d0 = 1.0 - b * n2;
  const double xy = bx * phi.y, xz = bx * phi.z, yz = by * phi.z;
  M3 J;
  J.m[0] = d0 + bx * phi.x; J.m[1] = xy + az;         J.m[2] = xz - ay;
  J.m[3] = xy - az;         J.m[4] = d0 + by * phi.y; J.m[5] = yz + ax;
  J.m[6] = xz + ay;         J.m[7] = yz - ax;         J.m[8] = d0 + bz * phi.z;
  return J;
}
// ---- the same functions for |phi| < 0.5 ONLY (the caller has checked the knot-pair logs of its spline segment: lambda in [0, 1] only
// shrinks them), without the closed-form branch: straight-line code for the evaluation kernels.
CTV_DI Q4 so3_exp_small(V3 w) {
  const double h2 = 0.25 * dot(w, w);
  const double im = 0.5 * (1.0 + h2 * (-1.0 / 6.0 + h2 * (1.0 / 120.0 + h2 * (-1.0 / 5040.0 + h2 * (1.0 / 362880.0 + h2 * (-1.0 / 39916800.0 +
                    h2 * (1.0 / 6227020800.0 + h2 * (-1.0 / 1307674368000.0))))))));
  const double re = 1.0 + h2 * (-0.5 + h2 * (1.0 / 24.0 + h2 * (-1.0 / 720.0 + h2 * (1.0 / 40320.0 + h2 * (-1.0 / 3628800.0 +
                    h2 * (1.0 / 479001600.0 + h2 * (-1.0 / 87178291200.0)))))));
  return qmk(im * w.x, im * w.y, im * w.z, re);
}
// Jr = I - a hat + b hat^2 with hat^2 = phi phi^T - |phi|^2 I written out: 19 operations instead of a 3 x 3 product
CTV_DI M3 so3_Jr_small(V3 phi) {
  const double n2 = dot(phi, phi);
  const double a = 0.5 + n2 * (-1.0 / 24.0 + n2 * (1.0 / 720.0 + n2 * (-1.0 / 40320.0 + n2 * (1.0 / 3628800.0 + n2 * (-1.0 / 479001600.0 +
                   n2 * (1.0 / 87178291200.0 + n2 * (-1.0 / 20922789888000.0)))))));
  const double b = 1.0 / 6.0 + n2 * (-1.0 / 120.0 + n2 * (1.0 / 5040.0 + n2 * (-1.0 / 362880.0 + n2 * (1.0 / 39916800.0 + n2 * (-1.0 / 6227020800.0 +
                   n2 * (1.0 / 1307674368000.0 + n2 * (-1.0 / 355687428096000.0)))))));
  const double bx = b * phi.x, by = b * phi.y, bz = b * phi.z, ax = a * phi.x, ay = a * phi.y, az = a * phi.z, d0 = 1.0 - b * n2;
  const double xy = bx * phi.y, xz = bx * phi.z, yz = by * phi.z;
  M3 J;
  J.m[0] = d0 + bx * phi.x; J.m[1] = xy + az;         J.m[2] = xz - ay;
  J.m[3] = xy - az;         J.m[4] = d0 + by * phi.y; J.m[5] = yz + ax;
  J.m[6] = xz + ay;         J.m[7] = yz - ax;         J.m[8] = d0 + bz * phi.z;
  return J;
}
// Product of two unit quaternions with Sophus' renormalisation 2 / (1 + n2) (so3.hpp:395-411) expanded around n2 = 1: the factors are
// unit to rounding, e = n2 - 1 is a few ulp, and 1 - e/2 + e^2/4 equals the quotient to e^3/8 -- no division.
CTV_DI Q4 qmul_unit(Q4 a, Q4 b) {
  Q4 o = qmul_raw(a, b);
  const double e = (o.x * o.x + o.y * o.y + o.z * o.z + o.w * o.w) - 1.0;
  const double s = 1.0 + e * (-0.5 + 0.25 * e);
  o.x *= s; o.y *= s; o.z *= s; o.w *= s;
  return o;
}

// compile-time choice between the two (SMALL: the caller has bounded |phi| < 0.5)
template <bool SMALL> CTV_DI Q4 so3_exp_sel(V3 w) {
  if constexpr (SMALL) return so3_exp_small(w);
  else return so3_exp(w);
}
template <bool SMALL> CTV_DI M3 so3_Jr_sel(V3 phi) {
  if constexpr (SMALL) return so3_Jr_small(phi);
  else return so3_Jr(phi);
}

// ---- Jr^-1: I + hat/2 + c*hat^2, c = 1/t^2 - (1+cos t)/(2 t sin t)   (sophus_utils.hpp:210-242)
CTV_DI double jrinv_coeff(double n2) {
  if (n2 < 0.25)   // (1 - (t/2) cot(t/2)) / t^2 = sum |B_2k| t^(2k-2) / (2k)!, truncation < 1e-17
    return 1.0 / 12.0 + n2 * (1.0 / 720.0 + n2 * (1.0 / 30240.0 + n2 * (1.0 / 1209600.0 + n2 * (1.0 / 47900160.0 +
           n2 * (691.0 / 1307674368000.0 + n2 * (1.0 / 74724249600.0 + n2 * (3617.0 / 10670622842880000.0)))))));
  const double n = sqrt(n2);
  return 1.0 / n2 - (1 + cos(n)) / (2 * n * sin(n));
}
CTV_DI M3 so3_Jr_inv(V3 phi) {
  const double c = jrinv_coeff(dot(phi, phi));
  const M3 H = hat(phi), H2 = mul(H, H);
  M3 J = m3_id();
#pragma unroll
  for (int i = 0; i < 9; ++i) J.m[i] += 0.5 * H.m[i] + c * H2.m[i];
  return J;
}

// ---- uniform cubic B-spline basis (reference src/spline/spline_common.h:76-153; evaluated as in
//      so3_spline_view.h:438-459, rd_spline_view.h:124-145).  c[i] = idt^D * sum_j M[i][j] * b_D(u)[j].
template <bool CUMULATIVE, int D> CTV_DI void basis(double u, double idt_pow, double c[4]) {
  // monomial derivative vector p[j] = base(D,j) * u^(j-D)
  double p[4] = {0.0, 0.0, 0.0, 0.0};
  if (D == 0) { p[0] = 1; p[1] = u; p[2] = u * u; p[3] = u * u * u; }
  else if (D == 1) { p[1] = 1; p[2] = 2 * u; p[3] = 3 * u * u; }
  else { p[2] = 2; p[3] = 6 * u; }
  const double s = idt_pow * (1.0 / 6.0);
  if (CUMULATIVE) {
    c[0] = s * (6 * p[0]);
    c[1] = s * (5 * p[0] + 3 * p[1] - 3 * p[2] + p[3]);
    c[2] = s * (p[0] + 3 * p[1] + 3 * p[2] - 2 * p[3]);
    c[3] = s * (p[3]);
  } else {
    c[0] = s * (p[0] - 3 * p[1] + 3 * p[2] - p[3]);
    c[1] = s * (4 * p[0] - 6 * p[2] + 3 * p[3]);
    c[2] = s * (p[0] + 3 * p[1] + 3 * p[2] - 3 * p[3]);
    c[3] = s * (p[3]);
  }
}

}  // namespace ctv
