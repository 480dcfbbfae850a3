/*
 * ctvo.c -- CPU fp64 ORACLE (test infrastructure, see ctvo.h; PARITY UNPINNED).
 *
 * Restates, in dependency-free C, the arithmetic of the Ctrl-VIO sliding-window solve.
 * Every function cites the reference file:line (relative to /root/reference) it follows.
 * Nothing here is copied: the reference is Eigen/Sophus/Ceres C++; this is scalar C over
 * flat arrays indexed by *global knot index* instead of Ceres parameter-pointer lists.
 */
#include "ctvo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CTVO_PI 3.14159265358979323846
#define SOPHUS_EPS 1e-10 /* src/sophus_lib/common.hpp:143-144 */
#define S_TO_NS 1e9      /* src/spline/spline_segment.h:35 */

/* ------------------------------------------------------------------ small algebra */
typedef double m3[9];

static void m3_mul(const double *A, const double *B, double *C) {
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(C, t, sizeof t);
}
static void m3_mulT(const double *A, const double *B, double *C) { /* A * B^T */
  double t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      t[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
  memcpy(C, t, sizeof t);
}
static void m3_vec(const double *A, const double *v, double *o) {
  double t[3];
  for (int i = 0; i < 3; ++i) t[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
  o[0] = t[0]; o[1] = t[1]; o[2] = t[2];
}
static void m3_scale(double *A, double s) { for (int i = 0; i < 9; ++i) A[i] *= s; }
static void m3_id(double *A) { memset(A, 0, 72); A[0] = A[4] = A[8] = 1.0; }
static void m3_sub(double *A, const double *B) { for (int i = 0; i < 9; ++i) A[i] -= B[i]; }
static void m3_add(double *A, const double *B) { for (int i = 0; i < 9; ++i) A[i] += B[i]; }
/* src/sophus_lib/so3.hpp:618-627 */
static void hat(const double *w, double *H) {
  H[0] = 0; H[1] = -w[2]; H[2] = w[1];
  H[3] = w[2]; H[4] = 0; H[5] = -w[0];
  H[6] = -w[1]; H[7] = w[0]; H[8] = 0;
}
static void cross(const double *a, const double *b, double *c) {
  double t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
  c[0] = t0; c[1] = t1; c[2] = t2;
}

/* quaternion storage (x,y,z,w) -- so3_spline_view.h:83 maps 4 doubles onto Sophus::SO3 */
static void q_mul_raw(const double *a, const double *b, double *o) {
  double t[4];
  t[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  t[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  t[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  t[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  memcpy(o, t, sizeof t);
}
/* SO3 * SO3 with the first-order renormalisation: src/sophus_lib/so3.hpp:338-355 */
static void q_mul(const double *a, const double *b, double *o) {
  q_mul_raw(a, b, o);
  double n2 = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
  if (n2 != 1.0) {
    double s = 2.0 / (1.0 + n2);
    o[0] *= s; o[1] *= s; o[2] *= s; o[3] *= s;
  }
}
/* src/sophus_lib/so3.hpp:202-204 */
static void q_inv(const double *a, double *o) { o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3]; }
/* point rotation, src/sophus_lib/so3.hpp:321-323 (Eigen _transformVector) */
static void q_rot(const double *q, const double *v, double *o) {
  double uv[3], t[3];
  cross(q, v, uv);
  uv[0] *= 2; uv[1] *= 2; uv[2] *= 2;
  cross(q, uv, t);
  double r0 = v[0] + q[3] * uv[0] + t[0], r1 = v[1] + q[3] * uv[1] + t[1], r2 = v[2] + q[3] * uv[2] + t[2];
  o[0] = r0; o[1] = r1; o[2] = r2;
}
/* matrix(): src/sophus_lib/so3.hpp:283-285 (Eigen toRotationMatrix) */
void ctvo_quat_to_R(const double q[4], double R[9]) {
  double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

/* src/sophus_lib/so3.hpp:534-569 */
void ctvo_so3_exp(const double w[3], double q[4]) {
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double th = sqrt(th2), half = 0.5 * th, im, re;
  if (th < SOPHUS_EPS) {
    double th4 = th2 * th2;
    im = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
    re = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
  } else {
    im = sin(half) / th;
    re = cos(half);
  }
  q[0] = im * w[0]; q[1] = im * w[1]; q[2] = im * w[2]; q[3] = re;
}

/* src/sophus_lib/so3.hpp:220-262 */
void ctvo_so3_log(const double q[4], double o[3]) {
  double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  double n = sqrt(n2), w = q[3], f;
  if (n < SOPHUS_EPS) {
    f = 2.0 / w - 2.0 * n2 / (w * w * w);
  } else if (fabs(w) < SOPHUS_EPS) {
    f = (w > 0 ? CTVO_PI : -CTVO_PI) / n;
  } else {
    f = 2.0 * atan(n / w) / n;
  }
  o[0] = f * q[0]; o[1] = f * q[1]; o[2] = f * q[2];
}

/* src/utils/sophus_utils.hpp:166-199 */
void ctvo_so3_Jr(const double phi[3], double J[9]) {
  double n2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  m3 H, H2;
  hat(phi, H);
  m3_mul(H, H, H2);
  m3_id(J);
  double a, b;
  if (n2 > SOPHUS_EPS) {
    double n = sqrt(n2);
    a = (1 - cos(n)) / n2;
    b = (n - sin(n)) / (n2 * n);
  } else {
    a = 0.5;
    b = 1.0 / 6.0;
  }
  for (int i = 0; i < 9; ++i) J[i] += -a * H[i] + b * H2[i];
}

/* src/utils/sophus_utils.hpp:210-242 */
void ctvo_so3_Jr_inv(const double phi[3], double J[9]) {
  double n2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  m3 H, H2;
  hat(phi, H);
  m3_mul(H, H, H2);
  m3_id(J);
  double b;
  if (n2 > SOPHUS_EPS) {
    double n = sqrt(n2);
    b = 1.0 / n2 - (1 + cos(n)) / (2 * n * sin(n));
  } else {
    b = 1.0 / 12.0;
  }
  for (int i = 0; i < 9; ++i) J[i] += 0.5 * H[i] + b * H2[i];
}

/* ------------------------------------------------------------------ spline basis
 * src/spline/spline_common.h:76-115 (blending, cumulative or not), :134-153 (base coeffs);
 * evaluated as in so3_spline_view.h:438-459 / rd_spline_view.h:124-145.               */
static const double M_BLEND[4][4] = {{1 / 6., -3 / 6., 3 / 6., -1 / 6.},
                                     {4 / 6., 0, -6 / 6., 3 / 6.},
                                     {1 / 6., 3 / 6., 3 / 6., -3 / 6.},
                                     {0, 0, 0, 1 / 6.}};
static const double M_CUMUL[4][4] = {{6 / 6., 0, 0, 0},
                                     {5 / 6., 3 / 6., -3 / 6., 1 / 6.},
                                     {1 / 6., 3 / 6., 3 / 6., -2 / 6.},
                                     {0, 0, 0, 1 / 6.}};
static const double BASE_C[4][4] = {{1, 1, 1, 1}, {0, 1, 2, 3}, {0, 0, 2, 6}, {0, 0, 0, 6}};

static void basis(int cumulative, int deriv, double u, double inv_dt_pow, double c[4]) {
  double p[4] = {0, 0, 0, 0};
  p[deriv] = BASE_C[deriv][deriv];
  double t = u;
  for (int j = deriv + 1; j < 4; ++j) { p[j] = BASE_C[deriv][j] * t; t *= u; }
  for (int i = 0; i < 4; ++i) {
    double s = 0;
    for (int j = 0; j < 4; ++j) s += (cumulative ? M_CUMUL[i][j] : M_BLEND[i][j]) * p[j];
    c[i] = inv_dt_pow * s;
  }
}

/* time -> (segment, u): src/spline/spline_segment.h:72-88, rd_spline.h:117-133 */
static void t_index(const ctvo_window *w, int64_t t_ns, int *s, double *u) {
  int64_t st = t_ns - w->t0_ns;
  *s = (int)(st / w->dt_ns);
  *u = (double)(st % w->dt_ns) / (double)w->dt_ns;
}
static double inv_dt(const ctvo_window *w) { return S_TO_NS / (double)w->dt_ns; } /* spline_segment.h:58 */

/* ------------------------------------------------------------------ spline views */
typedef struct { m3 d[4]; } jac4; /* d_val_d_knot, so3_spline_view.h:53-57 */

/* So3SplineView::EvaluateRp, so3_spline_view.h:136-198.  q4 = 4 consecutive knots. */
static void eval_Rp(const double *q4, double u, double *res_q, jac4 *J) {
  double coeff[4];
  basis(1, 0, u, 1.0, coeff);
  double acc[4] = {0, 0, 0, 1};
  m3 A_post_inv[4], Jr_inv_delta[3], Jr_kdelta[3];
  m3_id(A_post_inv[3]);
  for (int i = 2; i >= 0; --i) {
    double r0i[4], r01[4], delta[3], kd[3], nkd[3], e[4];
    q_inv(q4 + 4 * i, r0i);
    q_mul(r0i, q4 + 4 * (i + 1), r01);
    ctvo_so3_log(r01, delta);
    for (int k = 0; k < 3; ++k) { kd[k] = delta[k] * coeff[i + 1]; nkd[k] = -kd[k]; }
    ctvo_so3_exp(nkd, e);
    q_mul(acc, e, acc);
    if (J) {
      ctvo_so3_Jr_inv(delta, Jr_inv_delta[i]);
      ctvo_so3_Jr(kd, Jr_kdelta[i]);
      ctvo_quat_to_R(acc, A_post_inv[i]);
    }
  }
  double acc_inv[4];
  q_inv(acc, acc_inv);
  q_mul(q4, acc_inv, res_q);
  if (J) {
    memcpy(J->d[0], A_post_inv[0], 72);
    for (int i = 0; i < 3; ++i) {
      m3 Jh, t;
      m3_mul(A_post_inv[i + 1], Jr_kdelta[i], Jh);
      m3_scale(Jh, coeff[i + 1]);
      m3_mulT(Jh, Jr_inv_delta[i], t);
      m3_sub(J->d[i], t);
      m3_mul(Jh, Jr_inv_delta[i], J->d[i + 1]);
    }
  }
}

/* So3SplineView::EvaluateRTp, so3_spline_view.h:208-276.  Returns R(t)^T. */
static void eval_RTp(const double *q4, double u, double *res_q, jac4 *J) {
  double coeff[4];
  basis(1, 0, u, 1.0, coeff);
  double S[4][4];
  m3 Jr_inv_delta[3], Jr_kdelta[3], Ri_A_pre[4];
  memcpy(S[0], q4, 32);
  for (int i = 0; i < 3; ++i) {
    double r0i[4], r01[4], delta[3], kd[3], nkd[3], e[4];
    q_inv(q4 + 4 * i, r0i);
    q_mul(r0i, q4 + 4 * (i + 1), r01);
    ctvo_so3_log(r01, delta);
    for (int k = 0; k < 3; ++k) { kd[k] = delta[k] * coeff[i + 1]; nkd[k] = -kd[k]; }
    ctvo_so3_exp(kd, e);
    q_mul(S[i], e, S[i + 1]);
    if (J) {
      ctvo_so3_Jr_inv(delta, Jr_inv_delta[i]);
      ctvo_so3_Jr(nkd, Jr_kdelta[i]);
    }
  }
  q_inv(S[3], res_q);
  if (J) {
    for (int i = 0; i < 4; ++i) ctvo_quat_to_R(S[i], Ri_A_pre[i]);
    memcpy(J->d[0], Ri_A_pre[0], 72);
    for (int i = 0; i < 3; ++i) {
      m3 Jh, t;
      m3_mul(Ri_A_pre[i], Jr_kdelta[i], Jh);
      m3_scale(Jh, coeff[i + 1]);
      m3_mulT(Jh, Jr_inv_delta[i], t);
      m3_sub(J->d[i], t);
      m3_mul(Jh, Jr_inv_delta[i], J->d[i + 1]);
    }
  }
}

/* So3SplineView::VelocityBody (value), so3_spline_view.h:356-411 */
static void eval_omega(const double *q4, double u, double idt, double *omega) {
  double coeff[4], dcoeff[4];
  basis(1, 0, u, 1.0, coeff);
  basis(1, 1, u, idt, dcoeff);
  double delta[3][3], e[3][4];
  for (int i = 2; i >= 0; --i) {
    double r0i[4], r01[4], nkd[3];
    q_inv(q4 + 4 * i, r0i);
    q_mul(r0i, q4 + 4 * (i + 1), r01);
    ctvo_so3_log(r01, delta[i]);
    for (int k = 0; k < 3; ++k) nkd[k] = -coeff[i + 1] * delta[i][k];
    ctvo_so3_exp(nkd, e[i]);
  }
  double rv[3] = {delta[0][0] * dcoeff[1], delta[0][1] * dcoeff[1], delta[0][2] * dcoeff[1]};
  for (int i = 1; i < 3; ++i) {
    double t[3];
    q_rot(e[i], rv, t);
    for (int k = 0; k < 3; ++k) rv[k] = t[k] + delta[i][k] * dcoeff[i + 1];
  }
  omega[0] = rv[0]; omega[1] = rv[1]; omega[2] = rv[2];
}

/* RdSplineView::evaluate<D>, rd_spline_view.h:63-94.  p4 = 4 consecutive 3-vectors. */
static void eval_rd(const double *p4, int deriv, double u, double idt_pow, double *res, double *coeff_out) {
  double c[4];
  basis(0, deriv, u, idt_pow, c);
  res[0] = res[1] = res[2] = 0;
  for (int i = 0; i < 4; ++i) {
    res[0] += c[i] * p4[3 * i]; res[1] += c[i] * p4[3 * i + 1]; res[2] += c[i] * p4[3 * i + 2];
    if (coeff_out) coeff_out[i] = c[i];
  }
}

/* Sensitivity study only (tests/jacobian_noise_study.py): multiplies every Jacobian entry of the IMU / visual blocks
 * by (1 + rel * xi), xi uniform in [-1, 1] from a fixed LCG -- emulates Jacobians evaluated in lower precision while
 * residuals, accumulation and the solve stay fp64.  0 (default) = off. */
static double g_jn_imu = 0.0, g_jn_vis = 0.0;
static int g_round_products = 0;   /* 1: every block's J^T J / J^T r contribution is rounded to fp32 before it is added */
void ctvo_set_product_rounding(int on) { g_round_products = on; }
static unsigned long long g_jn_state = 88172645463325252ull;
void ctvo_set_jacobian_noise(double imu_rel, double vis_rel) { g_jn_imu = imu_rel; g_jn_vis = vis_rel; g_jn_state = 88172645463325252ull; }
static void jn_apply(double *J, int n, double rel) {
  if (rel == 0.0) return;
  for (int i = 0; i < n; ++i) {
    g_jn_state ^= g_jn_state << 13; g_jn_state ^= g_jn_state >> 7; g_jn_state ^= g_jn_state << 17;
    const double xi = (double)(g_jn_state >> 11) / 9007199254740992.0 * 2.0 - 1.0;
    J[i] *= 1.0 + rel * xi;
  }
}

/* ------------------------------------------------------------------ IMU block
 * SplitSpineView::Evaluate (split_spline_view.h:67-214) fused with
 * IMUFactor::Evaluate (trajectory_value_factor.h:141-248).
 * The reference's R_accum[DEG-1] stack overrun (split_spline_view.h:186-193) is NOT restated:
 * three entries are kept, which is the evident intent.                               */
void ctvo_imu_block(const ctvo_window *w, int m, double *r, double *J, int32_t *s_out) {
  int s; double u;
  t_index(w, w->imu_t[m], &s, &u);
  if (s_out) *s_out = s;
  const double *q4 = w->quat + 4 * s, *p4 = w->pos + 3 * s;
  const double idt = inv_dt(w);
  double lam_a[4], lam_R[4], lam_w[4];
  basis(0, 2, u, idt * idt, lam_a);
  basis(1, 0, u, 1.0, lam_R);
  basis(1, 1, u, idt, lam_w);

  double accel[3] = {0, 0, 0};
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 3; ++k) accel[k] += lam_a[i] * p4[3 * i + k];

  double d_vec[3][3], A_rot_inv[3][4], acc[4] = {0, 0, 0, 1};
  m3 A_post_inv[4], Jr_dvec_inv[3], Jr_kdelta[3];
  m3_id(A_post_inv[3]);
  for (int i = 2; i >= 0; --i) {
    double r0i[4], r01[4], nkd[3];
    q_inv(q4 + 4 * i, r0i);
    q_mul(r0i, q4 + 4 * (i + 1), r01);
    ctvo_so3_log(r01, d_vec[i]);
    for (int k = 0; k < 3; ++k) nkd[k] = -lam_R[i + 1] * d_vec[i][k];
    ctvo_so3_exp(nkd, A_rot_inv[i]);
    q_mul(acc, A_rot_inv[i], acc);
    if (J) {
      ctvo_quat_to_R(acc, A_post_inv[i]);
      ctvo_so3_Jr_inv(d_vec[i], Jr_dvec_inv[i]);
      ctvo_so3_Jr(nkd, Jr_kdelta[i]);
    }
  }
  double omega[4][3] = {{0, 0, 0}};
  for (int i = 0; i < 3; ++i) {
    double t[3];
    q_rot(A_rot_inv[i], omega[i], t);
    for (int k = 0; k < 3; ++k) omega[i + 1][k] = t[k] + lam_w[i + 1] * d_vec[i][k];
  }
  double Ri_inv[4], R_inv_q[4];
  q_inv(q4, Ri_inv);
  q_mul(acc, Ri_inv, R_inv_q);
  double ag[3] = {accel[0] + w->gravity[0], accel[1] + w->gravity[1], accel[2] + w->gravity[2]};
  double a_pred[3];
  q_rot(R_inv_q, ag, a_pred);

  const double *bg = w->bias + 6 * w->imu_bias[m], *ba = bg + 3;
  const double *iw = w->imu_w;
  for (int k = 0; k < 3; ++k) {
    r[k] = iw[k] * (omega[3][k] - (w->imu_gyro[3 * m + k] - bg[k]));
    r[3 + k] = iw[3 + k] * (a_pred[k] - (w->imu_acc[3 * m + k] - ba[k]));
  }
  if (!J) return;

  jac4 Jw, Ja;
  memset(&Jw, 0, sizeof Jw);
  memset(&Ja, 0, sizeof Ja);
  /* d(omega)/d(d_j): split_spline_view.h:157-181 */
  m3 dod[3];
  memcpy(dod[0], A_post_inv[1], 72);
  m3_scale(dod[0], lam_w[1]);
  for (int i = 1; i < 3; ++i) {
    m3 Hh, t;
    hat(omega[i], Hh);
    m3_mul(A_post_inv[i], Hh, t);
    m3_mul(t, Jr_kdelta[i], t);
    m3_scale(t, lam_R[i + 1]);
    memcpy(dod[i], A_post_inv[i + 1], 72);
    m3_scale(dod[i], lam_w[i + 1]);
    m3_add(dod[i], t);
  }
  for (int i = 0; i < 3; ++i) {
    m3 t;
    m3_mulT(dod[i], Jr_dvec_inv[i], t);
    m3_sub(Jw.d[i], t);
    m3_mul(dod[i], Jr_dvec_inv[i], t);
    m3_add(Jw.d[i + 1], t);
  }
  /* accel: split_spline_view.h:183-211 */
  m3 R_accum[3], R_inv, lhs, Hh;
  ctvo_quat_to_R(q4, R_accum[0]);
  for (int i = 1; i < 3; ++i) {
    m3 Ai;
    ctvo_quat_to_R(A_rot_inv[i - 1], Ai);
    m3_mulT(R_accum[i - 1], Ai, R_accum[i]);
  }
  ctvo_quat_to_R(R_inv_q, R_inv);
  hat(ag, Hh);
  m3_mul(R_inv, Hh, lhs);
  {
    m3 t;
    m3_mul(lhs, R_accum[0], t);
    m3_add(Ja.d[0], t);
  }
  for (int i = 0; i < 3; ++i) {
    m3 dad, t;
    m3_mul(lhs, R_accum[i], dad);
    m3_mul(dad, Jr_kdelta[i], dad);
    m3_scale(dad, lam_R[i + 1]);
    m3_mulT(dad, Jr_dvec_inv[i], t);
    m3_sub(Ja.d[i], t);
    m3_mul(dad, Jr_dvec_inv[i], t);
    m3_add(Ja.d[i + 1], t);
  }
  /* assemble 6x30: trajectory_value_factor.h:198-245 */
  memset(J, 0, sizeof(double) * 6 * 30);
  for (int k = 0; k < 4; ++k)
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        J[a * 30 + 3 * k + b] = iw[a] * Jw.d[k][3 * a + b];
        J[(3 + a) * 30 + 3 * k + b] = iw[3 + a] * Ja.d[k][3 * a + b];
        J[(3 + a) * 30 + 12 + 3 * k + b] = iw[3 + a] * lam_a[k] * R_inv[3 * a + b];
      }
  for (int a = 0; a < 3; ++a) {
    J[a * 30 + 24 + a] = iw[a];
    J[(3 + a) * 30 + 27 + a] = iw[3 + a];
  }
}

/* ------------------------------------------------------------------ visual block
 * ImageFeatureDelayFactor::Evaluate, image_feature_factor.h:63-269.                  */
void ctvo_visual_block(const ctvo_window *w, int v, double *r, double *J, int32_t *si_out, int32_t *sj_out) {
  const double d_inv = w->rho[w->v_lm[v]];
  const int64_t ld_ns = (int64_t)(w->ld * S_TO_NS); /* :72 truncation */
  const int64_t tau_i = w->v_ti[v] + (int64_t)w->v_rowi[v] * ld_ns;
  const int64_t tau_j = w->v_tj[v] + (int64_t)w->v_rowj[v] * ld_ns;
  int si, sj; double ui, uj;
  t_index(w, tau_i, &si, &ui);
  t_index(w, tau_j, &sj, &uj);
  if (si_out) *si_out = si;
  if (sj_out) *sj_out = sj;
  const double idt = inv_dt(w);
  const double p_i[3] = {w->v_pi[2 * v], w->v_pi[2 * v + 1], 1.0};
  const double p_j[2] = {w->v_pj[2 * v], w->v_pj[2 * v + 1]};

  double x_ci[3] = {p_i[0] / d_inv, p_i[1] / d_inv, p_i[2] / d_inv};
  double p_Ii[3];
  q_rot(w->q_CI, x_ci, p_Ii);
  for (int k = 0; k < 3; ++k) p_Ii[k] += w->p_CI[k];

  double S_IitoG[4], p_IiinG[3], S_GtoIj[4], p_IjinG[3];
  double Omega_i[3] = {0, 0, 0}, v_i[3] = {0, 0, 0}, Omega_j[3] = {0, 0, 0}, v_j[3] = {0, 0, 0};
  jac4 JR0, JR1;
  double cp0[4], cp1[4];
  if (J) {
    eval_omega(w->quat + 4 * si, ui, idt, Omega_i);
    eval_rd(w->pos + 3 * si, 1, ui, idt, v_i, NULL);
    eval_omega(w->quat + 4 * sj, uj, idt, Omega_j);
    eval_rd(w->pos + 3 * sj, 1, uj, idt, v_j, NULL);
  }
  eval_Rp(w->quat + 4 * si, ui, S_IitoG, J ? &JR0 : NULL);
  eval_rd(w->pos + 3 * si, 0, ui, 1.0, p_IiinG, cp0);
  double p_G[3];
  q_rot(S_IitoG, p_Ii, p_G);
  for (int k = 0; k < 3; ++k) p_G[k] += p_IiinG[k];
  eval_RTp(w->quat + 4 * sj, uj, S_GtoIj, J ? &JR1 : NULL);
  eval_rd(w->pos + 3 * sj, 0, uj, 1.0, p_IjinG, cp1);

  double S_ItoC[4], S_GtoCj[4];
  q_inv(w->q_CI, S_ItoC);
  q_mul(S_ItoC, S_GtoIj, S_GtoCj);
  double dpg[3] = {p_G[0] - p_IjinG[0], p_G[1] - p_IjinG[1], p_G[2] - p_IjinG[2]};
  double x_j[3], t3[3];
  q_rot(S_GtoCj, dpg, x_j);
  q_rot(S_ItoC, w->p_CI, t3);
  for (int k = 0; k < 3; ++k) x_j[k] -= t3[k];

  const double dji = 1.0 / x_j[2];
  const double sw = w->img_w;
  r[0] = sw * (x_j[0] * dji - p_j[0]);
  r[1] = sw * (x_j[1] * dji - p_j[1]);
  if (!J) return;

  /* J_v 2x3: :184-186 */
  double Jv[6] = {dji, 0, -dji * dji * x_j[0], 0, dji, -dji * dji * x_j[1]};
  m3 RGCj, RIiG, RGCjRi, Hh, t;
  ctvo_quat_to_R(S_GtoCj, RGCj);
  {
    double qq[4];
    q_mul(S_GtoCj, S_IitoG, qq);
    ctvo_quat_to_R(qq, RGCjRi);
  }
  ctvo_quat_to_R(S_IitoG, RIiG);
  /* 2x3 left factors :192-197 */
  double lhsR0[6], lhsP0[6], lhsR1[6], lhsP1[6];
  hat(p_Ii, Hh);
  m3_mul(RGCjRi, Hh, t);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 3; ++b) {
      double s0 = 0, s1 = 0;
      for (int k = 0; k < 3; ++k) { s0 += Jv[3 * a + k] * t[3 * k + b]; s1 += Jv[3 * a + k] * RGCj[3 * k + b]; }
      lhsR0[3 * a + b] = -s0;
      lhsP0[3 * a + b] = s1;
      lhsP1[3 * a + b] = -s1;
    }
  hat(dpg, Hh);
  m3_mul(RGCj, Hh, t);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 3; ++b) {
      double s0 = 0;
      for (int k = 0; k < 3; ++k) s0 += Jv[3 * a + k] * t[3 * k + b];
      lhsR1[3 * a + b] = s0;
    }
  memset(J, 0, sizeof(double) * 2 * 50);
  for (int k = 0; k < 4; ++k)
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 3; ++b) {
        double s0 = 0, s1 = 0;
        for (int c = 0; c < 3; ++c) {
          s0 += lhsR0[3 * a + c] * JR0.d[k][3 * c + b];
          s1 += lhsR1[3 * a + c] * JR1.d[k][3 * c + b];
        }
        J[a * 50 + 3 * k + b] = sw * s0;
        J[a * 50 + 24 + 3 * k + b] = sw * s1;
        J[a * 50 + 12 + 3 * k + b] = sw * cp0[k] * lhsP0[3 * a + b];
        J[a * 50 + 36 + 3 * k + b] = sw * cp1[k] * lhsP1[3 * a + b];
      }
  /* inverse depth :239-248 */
  {
    double qq[4], y[3];
    q_mul(S_GtoCj, S_IitoG, qq);
    q_mul(qq, w->q_CI, qq);
    q_rot(qq, x_ci, y);
    for (int a = 0; a < 2; ++a) {
      double s0 = 0;
      for (int k = 0; k < 3; ++k) s0 += Jv[3 * a + k] * (-y[k] / d_inv);
      J[a * 50 + 48] = sw * s0;
    }
  }
  /* line delay :251-264 */
  {
    const double ri = (double)w->v_rowi[v], rj = (double)w->v_rowj[v];
    double a1[3], a2[3], a3[3], tmp[3], Jx[3];
    for (int k = 0; k < 3; ++k) tmp[k] = ri * v_i[k] - rj * v_j[k];
    q_rot(S_GtoIj, tmp, a1);
    /* rowj * hat(Omega_j)^T * R_j^T (p_G - p_j) */
    m3 RGIj;
    ctvo_quat_to_R(S_GtoIj, RGIj);
    m3_vec(RGIj, dpg, tmp);
    cross(Omega_j, tmp, a2); /* hat(w)^T x = -w x x */
    for (int k = 0; k < 3; ++k) a2[k] = -rj * a2[k];
    /* rowi * R_j^T R_i hat(Omega_i) p_Ii */
    cross(Omega_i, p_Ii, tmp);
    m3_vec(RIiG, tmp, tmp);
    m3_vec(RGIj, tmp, a3);
    for (int k = 0; k < 3; ++k) tmp[k] = a1[k] + a2[k] + ri * a3[k];
    q_rot(S_ItoC, tmp, Jx);
    for (int a = 0; a < 2; ++a)
      J[a * 50 + 49] = sw * (Jv[3 * a] * Jx[0] + Jv[3 * a + 1] * Jx[1] + Jv[3 * a + 2] * Jx[2]);
  }
}

/* BiasFactor::Evaluate, trajectory_value_factor.h:45-99 (dt = 1 at every call site,
 * trajectory_manager.cpp:449-450, so sqrt_info/sqrt(dt) = sqrt_info). */
void ctvo_bias_block(const ctvo_window *w, int b, double *r, double *Jdiag) {
  const double *bi = w->bias + 6 * w->bc_i[b], *bj = w->bias + 6 * w->bc_j[b];
  for (int k = 0; k < 6; ++k) {
    r[k] = w->bc_w[6 * b + k] * (bj[k] - bi[k]);
    if (Jdiag) Jdiag[k] = w->bc_w[6 * b + k];
  }
}

/* MarginalizationFactor::Evaluate, marginalization_factor.cpp:326-355 */
static const double *prior_block_ptr(const ctvo_window *w, int kind, int idx) {
  switch (kind) {
    case CTVO_PK_ROT: return w->quat + 4 * idx;
    case CTVO_PK_POS: return w->pos + 3 * idx;
    case CTVO_PK_BG: return w->bias + 6 * idx;
    case CTVO_PK_BA: return w->bias + 6 * idx + 3;
    default: return &w->ld;
  }
}
static int prior_block_size(int kind) { return kind == CTVO_PK_LD ? 1 : 3; }
static int prior_block_unknown(const ctvo_window *w, int kind, int idx) {
  switch (kind) {
    case CTVO_PK_ROT: return 6 * idx;
    case CTVO_PK_POS: return 6 * idx + 3;
    case CTVO_PK_BG: return 6 * w->K + 6 * idx;
    case CTVO_PK_BA: return 6 * w->K + 6 * idx + 3;
    default: return 6 * w->K + 6 * w->F;
  }
}
void ctvo_prior_residual(const ctvo_window *w, double *r, double *dx) {
  const int n = w->pn;
  for (int b = 0; b < w->pnb; ++b) {
    const double *x = prior_block_ptr(w, w->p_kind[b], w->p_index[b]);
    const double *x0 = w->p_x0 + 4 * b;
    const int off = w->p_off[b];
    if (w->p_kind[b] == CTVO_PK_ROT) {
      double q0i[4], dq[4];
      q_inv(x0, q0i);
      q_mul_raw(q0i, x, dq);
      double sgn = (dq[3] >= 0) ? 2.0 : -2.0; /* :346-350 */
      dx[off] = sgn * dq[0]; dx[off + 1] = sgn * dq[1]; dx[off + 2] = sgn * dq[2];
    } else {
      for (int k = 0; k < prior_block_size(w->p_kind[b]); ++k) dx[off + k] = x[k] - x0[k];
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = w->pr0[i];
    for (int j = 0; j < n; ++j) s += w->pJ0[(size_t)j * n + i] * dx[j];
    r[i] = s;
  }
}

/* ------------------------------------------------------------------ robust loss
 * ceres::CauchyLoss(a) + Corrector, restated in-repo at marginalization_factor.cpp:39-67. */
static void cauchy(double a, double s, double rho[3]) {
  const double b = a * a, c = 1.0 / b, sum = 1.0 + s * c, inv = 1.0 / sum;
  rho[0] = b * log(sum);
  rho[1] = inv > 2.2250738585072014e-308 ? inv : 2.2250738585072014e-308;
  rho[2] = -c * inv * inv;
}
static double robustify(double a, int nres, int ncol, double *r, double *J) {
  double s = 0;
  for (int i = 0; i < nres; ++i) s += r[i] * r[i];
  if (a <= 0) return 0.5 * s;
  double rho[3];
  cauchy(a, s, rho);
  const double sq = sqrt(rho[1]);
  double rs, alpha_sq;
  if (s == 0.0 || rho[2] <= 0.0) { rs = sq; alpha_sq = 0.0; }
  else {
    const double D = 1.0 + 2.0 * s * rho[2] / rho[1];
    const double alpha = 1.0 - sqrt(D);
    rs = sq / (1 - alpha);
    alpha_sq = alpha / s;
  }
  if (J) {
    for (int c = 0; c < ncol; ++c) {
      double rj = 0;
      for (int i = 0; i < nres; ++i) rj += r[i] * J[i * ncol + c];
      for (int i = 0; i < nres; ++i) J[i * ncol + c] = sq * (J[i * ncol + c] - alpha_sq * r[i] * rj);
    }
  }
  for (int i = 0; i < nres; ++i) r[i] *= rs;
  return 0.5 * rho[0];
}

/* ------------------------------------------------------------------ cost / normal equations */
static int nP(const ctvo_window *w) { return 6 * w->K + 6 * w->F + 1; }
static int nN(const ctvo_window *w) { return nP(w) + w->L; }

double ctvo_cost(const ctvo_window *w) {
  double cost = 0, r[6];
  for (int m = 0; m < w->M; ++m) {
    ctvo_imu_block(w, m, r, NULL, NULL);
    for (int k = 0; k < 6; ++k) cost += 0.5 * r[k] * r[k];
  }
  for (int v = 0; v < w->V; ++v) {
    ctvo_visual_block(w, v, r, NULL, NULL, NULL);
    cost += robustify(w->v_cauchy ? w->v_cauchy[v] : w->cauchy_a, 2, 0, r, NULL);
  }
  for (int b = 0; b < w->NB; ++b) {
    ctvo_bias_block(w, b, r, NULL);
    for (int k = 0; k < 6; ++k) cost += 0.5 * r[k] * r[k];
  }
  if (w->pn > 0) {
    double *pr = (double *)malloc(sizeof(double) * 2 * w->pn);
    ctvo_prior_residual(w, pr, pr + w->pn);
    for (int i = 0; i < w->pn; ++i) cost += 0.5 * pr[i] * pr[i];
    free(pr);
  }
  return cost;
}

static void scatter(double *H, double *g, int N, int nres, int ncol, const double *r, const double *J, const int *idx) {
  for (int a = 0; a < ncol; ++a) {
    if (idx[a] < 0) continue;
    double ga = 0;
    for (int i = 0; i < nres; ++i) ga += J[i * ncol + a] * r[i];
    g[idx[a]] += g_round_products ? (double)(float)ga : ga;
    for (int b = 0; b < ncol; ++b) {
      if (idx[b] < 0) continue;
      double h = 0;
      for (int i = 0; i < nres; ++i) h += J[i * ncol + a] * J[i * ncol + b];
      H[(size_t)idx[a] * N + idx[b]] += g_round_products ? (double)(float)h : h;
    }
  }
}

double ctvo_build_normal(const ctvo_window *w, double *H, double *g) {
  const int N = nN(w), P = nP(w), K = w->K;
  memset(H, 0, sizeof(double) * (size_t)N * N);
  memset(g, 0, sizeof(double) * N);
  double cost = 0;
  double r[6], J[6 * 30];
  int idx[50];
  for (int m = 0; m < w->M; ++m) {
    int32_t s;
    ctvo_imu_block(w, m, r, J, &s);
    jn_apply(J, 6 * 30, g_jn_imu);
    for (int k = 0; k < 4; ++k)
      for (int c = 0; c < 3; ++c) { idx[3 * k + c] = 6 * (s + k) + c; idx[12 + 3 * k + c] = 6 * (s + k) + 3 + c; }
    for (int c = 0; c < 6; ++c) idx[24 + c] = 6 * K + 6 * w->imu_bias[m] + c;
    for (int k = 0; k < 6; ++k) cost += 0.5 * r[k] * r[k];
    scatter(H, g, N, 6, 30, r, J, idx);
  }
  double Jv[2 * 50];
  for (int v = 0; v < w->V; ++v) {
    int32_t si, sj;
    ctvo_visual_block(w, v, r, Jv, &si, &sj);
    jn_apply(Jv, 2 * 50, g_jn_vis);
    cost += robustify(w->v_cauchy ? w->v_cauchy[v] : w->cauchy_a, 2, 50, r, Jv);
    for (int k = 0; k < 4; ++k)
      for (int c = 0; c < 3; ++c) {
        idx[3 * k + c] = 6 * (si + k) + c; idx[12 + 3 * k + c] = 6 * (si + k) + 3 + c;
        idx[24 + 3 * k + c] = 6 * (sj + k) + c; idx[36 + 3 * k + c] = 6 * (sj + k) + 3 + c;
      }
    idx[48] = P + w->v_lm[v];
    idx[49] = P - 1;
    scatter(H, g, N, 2, 50, r, Jv, idx);
  }
  for (int b = 0; b < w->NB; ++b) {
    double d[6];
    ctvo_bias_block(w, b, r, d);
    for (int k = 0; k < 6; ++k) {
      const int ii = 6 * K + 6 * w->bc_i[b] + k, jj = 6 * K + 6 * w->bc_j[b] + k;
      cost += 0.5 * r[k] * r[k];
      g[ii] += -d[k] * r[k]; g[jj] += d[k] * r[k];
      H[(size_t)ii * N + ii] += d[k] * d[k]; H[(size_t)jj * N + jj] += d[k] * d[k];
      H[(size_t)ii * N + jj] -= d[k] * d[k]; H[(size_t)jj * N + ii] -= d[k] * d[k];
    }
  }
  if (w->pn > 0) { /* prior: J = J0[:, block] (marginalization_factor.cpp:356-371) */
    const int n = w->pn;
    double *pr = (double *)malloc(sizeof(double) * 2 * n);
    int *col = (int *)malloc(sizeof(int) * n);
    ctvo_prior_residual(w, pr, pr + n);
    for (int i = 0; i < n; ++i) { cost += 0.5 * pr[i] * pr[i]; col[i] = -1; }
    for (int b = 0; b < w->pnb; ++b) {
      const int u0 = prior_block_unknown(w, w->p_kind[b], w->p_index[b]);
      for (int k = 0; k < prior_block_size(w->p_kind[b]); ++k) col[w->p_off[b] + k] = u0 + k;
    }
    for (int a = 0; a < n; ++a) {
      if (col[a] < 0) continue;
      double ga = 0;
      for (int i = 0; i < n; ++i) ga += w->pJ0[(size_t)a * n + i] * pr[i];
      g[col[a]] += ga;
      for (int b = 0; b < n; ++b) {
        if (col[b] < 0) continue;
        double h = 0;
        for (int i = 0; i < n; ++i) h += w->pJ0[(size_t)a * n + i] * w->pJ0[(size_t)b * n + i];
        H[(size_t)col[a] * N + col[b]] += h;
      }
    }
    free(pr); free(col);
  }
  return cost;
}

/* Which unknowns are in Ceres' reduced program: referenced by some residual block and not
 * constant (trajectory_estimator.cpp:134-138, 236-245, 312-313).  Parameter blocks follow the
 * reference's spans: IMU 4 knots (:225), visual [t, t+0.039 s] per end (:299).          */
void ctvo_active_mask(const ctvo_window *w, uint8_t *active) {
  const int N = nN(w), P = nP(w), K = w->K;
  memset(active, 0, N);
  for (int m = 0; m < w->M; ++m) {
    int s; double u;
    t_index(w, w->imu_t[m], &s, &u);
    for (int c = 0; c < 24; ++c) active[6 * s + c] = 1;
    for (int c = 0; c < 6; ++c) active[6 * K + 6 * w->imu_bias[m] + c] = 1;
  }
  for (int v = 0; v < w->V; ++v) {
    const int64_t tt[2] = {w->v_ti[v], w->v_tj[v]};
    for (int e = 0; e < 2; ++e) {
      int s0, s1; double u;
      t_index(w, tt[e], &s0, &u);
      t_index(w, tt[e] + (int64_t)(0.039 * S_TO_NS), &s1, &u);
      for (int k = s0; k < s1 + 4 && k < K; ++k)
        for (int c = 0; c < 6; ++c) active[6 * k + c] = 1;
    }
    active[P + w->v_lm[v]] = 1;
    active[P - 1] = 1;
  }
  for (int b = 0; b < w->NB; ++b)
    for (int c = 0; c < 6; ++c) { active[6 * K + 6 * w->bc_i[b] + c] = 1; active[6 * K + 6 * w->bc_j[b] + c] = 1; }
  for (int b = 0; b < w->pnb; ++b) {
    const int u0 = prior_block_unknown(w, w->p_kind[b], w->p_index[b]);
    for (int k = 0; k < prior_block_size(w->p_kind[b]); ++k) active[u0 + k] = 1;
  }
  for (int k = 0; k <= w->fixed_upto && k < K; ++k)
    for (int c = 0; c < 6; ++c) active[6 * k + c] = 0;
  if (w->knot_const)   /* per-knot SetParameterBlockConstant (trajectory_estimator.cpp:134-138): need not be a prefix */
    for (int k = 0; k < K; ++k)
      if (w->knot_const[k]) for (int c = 0; c < 6; ++c) active[6 * k + c] = 0;
  for (int f = 0; f < w->F; ++f)
    for (int c = 0; c < 3; ++c) {
      if (w->lock_bg) active[6 * K + 6 * f + c] = 0;
      if (w->lock_ba) active[6 * K + 6 * f + 3 + c] = 0;
    }
  if (w->fix_ld) active[P - 1] = 0;
}

/* ------------------------------------------------------------------ retraction
 * LieAnalyticLocalParameterization::Plus (ceres_local_param.h:137-145): q <- q * exp(d);
 * additive elsewhere; line delay projected on its box (trajectory_estimator.cpp:316-317). */
void ctvo_plus(ctvo_window *w, const double *d) {
  const int K = w->K, P = nP(w);
  for (int k = 0; k < K; ++k) {
    double e[4];
    ctvo_so3_exp(d + 6 * k, e);
    q_mul(w->quat + 4 * k, e, w->quat + 4 * k);
    for (int c = 0; c < 3; ++c) w->pos[3 * k + c] += d[6 * k + 3 + c];
  }
  for (int i = 0; i < 6 * w->F; ++i) w->bias[i] += d[6 * K + i];
  w->ld += d[P - 1];
  if (!w->fix_ld) {
    if (w->ld < w->ld_lo) w->ld = w->ld_lo;
    if (w->ld > w->ld_hi) w->ld = w->ld_hi;
  }
  for (int l = 0; l < w->L; ++l) w->rho[l] += d[P + l];
}

/* ambient state vector helpers (reduced program = active blocks only) */
typedef struct { double *quat, *pos, *bias, *rho; double ld; } state_copy;
static void state_save(const ctvo_window *w, state_copy *s) {
  s->quat = (double *)malloc(sizeof(double) * (4 * w->K + 3 * w->K + 6 * w->F + w->L + 1));
  s->pos = s->quat + 4 * w->K;
  s->bias = s->pos + 3 * w->K;
  s->rho = s->bias + 6 * w->F;
  memcpy(s->quat, w->quat, sizeof(double) * 4 * w->K);
  memcpy(s->pos, w->pos, sizeof(double) * 3 * w->K);
  memcpy(s->bias, w->bias, sizeof(double) * 6 * w->F);
  memcpy(s->rho, w->rho, sizeof(double) * w->L);
  s->ld = w->ld;
}
static void state_restore(ctvo_window *w, const state_copy *s) {
  memcpy(w->quat, s->quat, sizeof(double) * 4 * w->K);
  memcpy(w->pos, s->pos, sizeof(double) * 3 * w->K);
  memcpy(w->bias, s->bias, sizeof(double) * 6 * w->F);
  memcpy(w->rho, s->rho, sizeof(double) * w->L);
  w->ld = s->ld;
}
static void state_free(state_copy *s) { free(s->quat); }
/* |x|^2 and |x - y|^2 over active ambient parameters */
static void state_norms(const ctvo_window *w, const state_copy *y, const uint8_t *act, double *x2, double *d2, double *dinf) {
  const int K = w->K, P = nP(w);
  double a = 0, b = 0, m = 0, t;
  for (int k = 0; k < K; ++k) {
    if (act[6 * k])
      for (int c = 0; c < 4; ++c) {
        a += w->quat[4 * k + c] * w->quat[4 * k + c];
        if (y) { t = w->quat[4 * k + c] - y->quat[4 * k + c]; b += t * t; if (fabs(t) > m) m = fabs(t); }
      }
    if (act[6 * k + 3])
      for (int c = 0; c < 3; ++c) {
        a += w->pos[3 * k + c] * w->pos[3 * k + c];
        if (y) { t = w->pos[3 * k + c] - y->pos[3 * k + c]; b += t * t; if (fabs(t) > m) m = fabs(t); }
      }
  }
  for (int i = 0; i < 6 * w->F; ++i)
    if (act[6 * K + i]) {
      a += w->bias[i] * w->bias[i];
      if (y) { t = w->bias[i] - y->bias[i]; b += t * t; if (fabs(t) > m) m = fabs(t); }
    }
  if (act[P - 1]) {
    a += w->ld * w->ld;
    if (y) { t = w->ld - y->ld; b += t * t; if (fabs(t) > m) m = fabs(t); }
  }
  for (int l = 0; l < w->L; ++l)
    if (act[P + l]) {
      a += w->rho[l] * w->rho[l];
      if (y) { t = w->rho[l] - y->rho[l]; b += t * t; if (fabs(t) > m) m = fabs(t); }
    }
  if (x2) *x2 = a;
  if (d2) *d2 = b;
  if (dinf) *dinf = m;
}

/* ------------------------------------------------------------------ linear algebra */
/* in-place lower Cholesky of row-major n*n (uses lower triangle); returns 0 on success */
static int chol(double *A, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0) || !isfinite(d)) return 1;
    d = sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[(size_t)i * n + j];
      const double *ai = A + (size_t)i * n, *aj = A + (size_t)j * n;
      for (int k = 0; k < j; ++k) s -= ai[k] * aj[k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  return 0;
}
static void chol_solve(const double *Lm, int n, double *b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= Lm[(size_t)i * n + k] * b[k];
    b[i] = s / Lm[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= Lm[(size_t)k * n + i] * b[k];
    b[i] = s / Lm[(size_t)i * n + i];
  }
}

/* Solve (Hs + D2) y = -gs for the *scaled* system.  Hs: N*N (destroyed/unchanged per path).
 * Inactive unknowns have identity rows.  Returns 0 on success.                         */
static int lin_solve(const double *Hs, const double *gs, const double *D2, int N, int P, int use_schur, double *y) {
  if (!use_schur) {
    double *A = (double *)malloc(sizeof(double) * (size_t)N * N);
    memcpy(A, Hs, sizeof(double) * (size_t)N * N);
    for (int i = 0; i < N; ++i) { A[(size_t)i * N + i] += D2[i]; y[i] = -gs[i]; }
    int rc = chol(A, N);
    if (!rc) chol_solve(A, N, y);
    free(A);
    return rc;
  }
  /* landmarks are 1-D blocks with no landmark-landmark coupling: S = App - W Dl^-1 W^T */
  const int L = N - P;
  double *S = (double *)malloc(sizeof(double) * (size_t)P * P);
  double *rhs = (double *)malloc(sizeof(double) * P);
  double *dl = (double *)malloc(sizeof(double) * (L > 0 ? L : 1));
  for (int l = 0; l < L; ++l) {
    double d = Hs[(size_t)(P + l) * N + P + l] + D2[P + l];
    if (!(d > 0)) { free(S); free(rhs); free(dl); return 1; }
    dl[l] = 1.0 / d;
  }
  for (int i = 0; i < P; ++i) {
    const double *hi = Hs + (size_t)i * N;
    double ri = -gs[i];
    for (int l = 0; l < L; ++l) ri += hi[P + l] * dl[l] * gs[P + l];
    rhs[i] = ri;
    for (int j = 0; j <= i; ++j) {
      const double *hj = Hs + (size_t)j * N;
      double s = hi[j];
      for (int l = 0; l < L; ++l) s -= hi[P + l] * dl[l] * hj[P + l];
      S[(size_t)i * P + j] = s;
    }
    S[(size_t)i * P + i] += D2[i];
  }
  int rc = chol(S, P);
  if (!rc) {
    chol_solve(S, P, rhs);
    memcpy(y, rhs, sizeof(double) * P);
    for (int l = 0; l < L; ++l) {
      double s = -gs[P + l];
      for (int i = 0; i < P; ++i) s -= Hs[(size_t)i * N + P + l] * y[i];
      y[P + l] = s * dl[l];
    }
  }
  free(S); free(rhs); free(dl);
  return rc;
}

/* mask + scale the normal equations in place: Hs = C H C, gs = C g, identity on inactive */
static void mask_scale(double *H, double *g, const uint8_t *act, const double *c, int N) {
  for (int i = 0; i < N; ++i) {
    if (!act[i]) {
      for (int j = 0; j < N; ++j) { H[(size_t)i * N + j] = 0; H[(size_t)j * N + i] = 0; }
      H[(size_t)i * N + i] = 1.0;
      g[i] = 0;
    }
  }
  for (int i = 0; i < N; ++i) {
    if (!act[i]) continue;
    g[i] *= c[i];
    for (int j = 0; j < N; ++j)
      if (act[j]) H[(size_t)i * N + j] *= c[i] * c[j];
  }
}

/* One trust-region step on the scaled system (Ceres LevenbergMarquardtStrategy::ComputeStep +
 * TrustRegionMinimizer::ComputeTrustRegionStep):
 *   D2 = clamp(diag(Hs), 1e-6, 1e32)/mu ; (Hs + D2) y = -gs ; model = -y'(gs + Hs y / 2).  */
static int tr_step(const double *Hs, const double *gs, const uint8_t *act, const double *c, double mu, int N, int P,
                   int use_schur, double *delta, double *model_change) {
  double *D2 = (double *)malloc(sizeof(double) * N), *y = (double *)malloc(sizeof(double) * N);
  for (int i = 0; i < N; ++i) {
    double d = Hs[(size_t)i * N + i];
    if (d < 1e-6) d = 1e-6;
    if (d > 1e32) d = 1e32;
    D2[i] = act[i] ? d / mu : 0.0;
  }
  int rc = lin_solve(Hs, gs, D2, N, P, use_schur, y);
  if (!rc) {
    double mc = 0;
    for (int i = 0; i < N; ++i) {
      if (!act[i]) { y[i] = 0; continue; }
      double hy = 0;
      for (int j = 0; j < N; ++j)
        if (act[j]) hy += Hs[(size_t)i * N + j] * y[j];
      mc -= y[i] * (gs[i] + 0.5 * hy);
    }
    *model_change = mc;
    for (int i = 0; i < N; ++i) {
      delta[i] = act[i] ? c[i] * y[i] : 0.0;
      if (!isfinite(delta[i])) rc = 1;
    }
  }
  free(D2); free(y);
  return rc;
}

static void jacobi_scale(const double *H, const uint8_t *act, int N, double *c) {
  for (int i = 0; i < N; ++i) c[i] = act[i] ? 1.0 / (1.0 + sqrt(H[(size_t)i * N + i])) : 1.0;
}

double ctvo_lm_step(const ctvo_window *w, double mu, int use_schur, double *delta) {
  const int N = nN(w), P = nP(w);
  double *H = (double *)malloc(sizeof(double) * (size_t)N * N), *g = (double *)malloc(sizeof(double) * N);
  double *c = (double *)malloc(sizeof(double) * N);
  uint8_t *act = (uint8_t *)malloc(N);
  ctvo_active_mask(w, act);
  ctvo_build_normal(w, H, g);
  jacobi_scale(H, act, N, c);
  mask_scale(H, g, act, c, N);
  double mc = 0;
  if (tr_step(H, g, act, c, mu, N, P, use_schur, delta, &mc)) mc = -1;
  free(H); free(g); free(c); free(act);
  return mc;
}

/* max-norm of x - Plus(x, -g) over ambient parameters (Ceres gradient_max_norm) */
static double gradient_max_norm(ctvo_window *w, const double *g_unscaled, const uint8_t *act) {
  const int N = nN(w);
  state_copy x0;
  state_save(w, &x0);
  double *d = (double *)malloc(sizeof(double) * N);
  for (int i = 0; i < N; ++i) d[i] = act[i] ? -g_unscaled[i] : 0.0;
  ctvo_plus(w, d);
  double dinf;
  state_norms(w, &x0, act, NULL, NULL, &dinf);
  state_restore(w, &x0);
  state_free(&x0);
  free(d);
  return dinf;
}

/* ------------------------------------------------------------------ Ceres' projected line search
 * When any parameter block of the reduced program has bounds (the line delay, trajectory_estimator.cpp:311-318)
 * Ceres 1.14 sets Minimizer::Options::is_constrained and TrustRegionMinimizer::Minimize runs DoLineSearch(x, gradient,
 * cost, &delta) between ComputeTrustRegionStep and ComputeCandidatePointAndEvaluateCost (external source:
 * internal/ceres/trust_region_minimizer.cc, line_search.cc, polynomial.cc; restated from their published algorithm):
 * an ARMIJO search along Plus(x, alpha * delta) (Plus projects onto the box) starting at alpha = 1 with
 * sufficient_decrease 1e-4, CUBIC interpolation (value AND gradient at every trial point), step contraction limited to
 * [1e-3, 0.6] x the current alpha, at most 20 iterations, min step 1e-9.  On success delta *= alpha; on failure delta is
 * left unchanged.  model_cost_change stays the one of the unscaled step.                                             */
typedef struct { double x, value, gradient; int value_is_valid, gradient_is_valid; } ls_sample;

static double poly_eval(const double *p, int n /* coefficients, highest first */, double x) {
  double v = 0;
  for (int i = 0; i < n; ++i) v = v * x + p[i];
  return v;
}
/* FindInterpolatingPolynomial: values and gradients of the samples -> coefficients (highest power first). */
static int poly_interpolate(const ls_sample *s, int ns, double *coef) {
  int nc = 0;
  for (int i = 0; i < ns; ++i) nc += (s[i].value_is_valid != 0) + (s[i].gradient_is_valid != 0);
  const int deg = nc - 1;
  double A[6][7];
  int row = 0;
  for (int i = 0; i < ns; ++i) {
    if (s[i].value_is_valid) {
      for (int j = 0; j <= deg; ++j) A[row][j] = pow(s[i].x, deg - j);
      A[row][nc] = s[i].value; ++row;
    }
    if (s[i].gradient_is_valid) {
      for (int j = 0; j < deg; ++j) A[row][j] = (deg - j) * pow(s[i].x, deg - j - 1);
      A[row][deg] = 0.0;
      A[row][nc] = s[i].gradient; ++row;
    }
  }
  for (int c = 0; c < nc; ++c) { /* Gauss-Jordan with partial pivoting (Ceres: Eigen fullPivLu, same solution) */
    int piv = c;
    for (int r = c + 1; r < nc; ++r) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
    if (A[piv][c] == 0.0) return 0;
    if (piv != c) for (int j = 0; j <= nc; ++j) { double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
    for (int r = 0; r < nc; ++r) {
      if (r == c) continue;
      const double f = A[r][c] / A[c][c];
      for (int j = c; j <= nc; ++j) A[r][j] -= f * A[c][j];
    }
  }
  for (int c = 0; c < nc; ++c) coef[c] = A[c][nc] / A[c][c];
  return nc;
}
/* real parts of all (complex) roots of a polynomial of degree <= 4 (Ceres FindPolynomialRoots: closed forms up to degree 2,
 * companion-matrix eigenvalues above; here Durand-Kerner, same roots to rounding).  Returns the number of roots. */
static int poly_root_real_parts(const double *p_in, int n, double *re) {
  while (n > 0 && p_in[0] == 0.0) { ++p_in; --n; }   /* RemoveLeadingZeros */
  const int deg = n - 1;
  if (deg <= 0) return 0;
  if (deg == 1) { re[0] = -p_in[1] / p_in[0]; return 1; }
  if (deg == 2) {
    const double a = p_in[0], b = p_in[1], c = p_in[2], D = b * b - 4 * a * c, sD = sqrt(fabs(D));
    if (D >= 0) {
      if (b >= 0) { re[0] = (-b - sD) / (2.0 * a); re[1] = (2.0 * c) / (-b - sD); }
      else { re[0] = (2.0 * c) / (-b + sD); re[1] = (-b + sD) / (2.0 * a); }
    } else { re[0] = re[1] = -b / (2.0 * a); }
    return 2;
  }
  double q[5], zr[4], zi[4];
  for (int i = 0; i <= deg; ++i) q[i] = p_in[i] / p_in[0];
  double rad = 0;
  for (int i = 1; i <= deg; ++i) if (fabs(q[i]) > rad) rad = fabs(q[i]);
  rad = 1.0 + rad;
  for (int k = 0; k < deg; ++k) { const double ang = 2.0 * 3.14159265358979323846 * k / deg + 0.4; zr[k] = 0.5 * rad * cos(ang); zi[k] = 0.5 * rad * sin(ang); }
  for (int it = 0; it < 500; ++it) {
    double change = 0;
    for (int k = 0; k < deg; ++k) {
      double pr = 1.0, pi = 0.0;   /* p(z_k) by Horner */
      for (int i = 1; i <= deg; ++i) { const double tr = pr * zr[k] - pi * zi[k] + q[i], ti = pr * zi[k] + pi * zr[k]; pr = tr; pi = ti; }
      double dr = 1.0, di = 0.0;   /* prod_{j != k} (z_k - z_j) */
      for (int j = 0; j < deg; ++j) {
        if (j == k) continue;
        const double ar = zr[k] - zr[j], ai = zi[k] - zi[j], tr = dr * ar - di * ai, ti = dr * ai + di * ar;
        dr = tr; di = ti;
      }
      const double den = dr * dr + di * di;
      if (den == 0.0) continue;
      const double cr = (pr * dr + pi * di) / den, ci = (pi * dr - pr * di) / den;
      zr[k] -= cr; zi[k] -= ci;
      change += fabs(cr) + fabs(ci);
    }
    if (change < 1e-15 * rad) break;
  }
  for (int k = 0; k < deg; ++k) re[k] = zr[k];
  return deg;
}
/* MinimizeInterpolatingPolynomial (polynomial.cc): minimum over [x_min, x_max] of the interpolating polynomial, candidates =
 * midpoint, both ends, the real part of every root of the derivative inside the interval, the sample points inside it. */
static double poly_minimize_interpolating(const ls_sample *s, int ns, double x_min, double x_max) {
  double coef[6], der[5], roots[4];
  const int nc = poly_interpolate(s, ns, coef);
  if (nc <= 0) return 0.5 * (x_min + x_max);
  double best_x = 0.5 * (x_min + x_max), best = poly_eval(coef, nc, best_x), v;
  v = poly_eval(coef, nc, x_min); if (v < best) { best = v; best_x = x_min; }
  v = poly_eval(coef, nc, x_max); if (v < best) { best = v; best_x = x_max; }
  if (nc > 2) {
    for (int i = 0; i < nc - 1; ++i) der[i] = (nc - 1 - i) * coef[i];
    const int nr = poly_root_real_parts(der, nc - 1, roots);
    for (int i = 0; i < nr; ++i) {
      if (roots[i] < x_min || roots[i] > x_max) continue;
      v = poly_eval(coef, nc, roots[i]);
      if (v < best) { best = v; best_x = roots[i]; }
    }
  }
  for (int i = 0; i < ns; ++i) {
    if (s[i].x < x_min || s[i].x > x_max) continue;
    v = poly_eval(coef, nc, s[i].x);
    if (v < best) { best = v; best_x = s[i].x; }
  }
  return best_x;
}
/* test hook: exercise the interpolation machinery from Python */
double ctvo_ls_interpolate(int ns, const double *x, const double *value, const double *gradient, double x_min, double x_max) {
  ls_sample s[3];
  for (int i = 0; i < ns && i < 3; ++i) { s[i].x = x[i]; s[i].value = value[i]; s[i].gradient = gradient[i]; s[i].value_is_valid = s[i].gradient_is_valid = 1; }
  return poly_minimize_interpolating(s, ns, x_min, x_max);
}

/* TrustRegionMinimizer::Minimize of Ceres 1.14 as configured at trajectory_estimator.cpp:371-398
 * (external source, restated from its documentation and published algorithm), including the
 * projected Armijo line search of bounds-constrained problems (above).                                        */
static double g_ftol = 1e-6, g_gtol = 1e-10, g_ptol = 1e-8; /* Ceres defaults */
void ctvo_set_tolerances(double ftol, double gtol, double ptol) { g_ftol = ftol; g_gtol = gtol; g_ptol = ptol; }
static int g_line_search = 1;
void ctvo_set_line_search(int on) { g_line_search = on; }

int ctvo_solve(ctvo_window *w, int max_iters, int use_schur, ctvo_summary *out) {
  const int N = nN(w), P = nP(w);
  const double ftol = g_ftol, gtol = g_gtol, ptol = g_ptol, min_rel_dec = 1e-3;
  const double max_radius = 1e16, min_radius = 1e-32;
  double *H = (double *)malloc(sizeof(double) * (size_t)N * N), *g = (double *)malloc(sizeof(double) * N);
  double *c = (double *)malloc(sizeof(double) * N), *delta = (double *)malloc(sizeof(double) * N);
  double *H2 = NULL, *g2 = (double *)malloc(sizeof(double) * N), *dtrial = (double *)malloc(sizeof(double) * N);
  uint8_t *act = (uint8_t *)malloc(N);
  ctvo_summary sm;
  memset(&sm, 0, sizeof sm);
  ctvo_active_mask(w, act);
  const int constrained = g_line_search && !w->fix_ld && act[P - 1];   /* Program::IsBoundsConstrained of the reduced program */
  /* IterationZero: project onto the feasible set, evaluate */
  if (!w->fix_ld) {
    if (w->ld < w->ld_lo) w->ld = w->ld_lo;
    if (w->ld > w->ld_hi) w->ld = w->ld_hi;
  }
  double cost = ctvo_build_normal(w, H, g);
  sm.initial_cost = cost;
  sm.cost_hist[0] = cost;
  double gmax = gradient_max_norm(w, g, act);
  jacobi_scale(H, act, N, c); /* computed once at iteration 0 */
  mask_scale(H, g, act, c, N);
  double x2;
  state_norms(w, NULL, act, &x2, NULL, NULL);
  double x_norm = sqrt(x2);
  double mu = 1e4, nu = 2.0;
  int iter = 0, invalid = 0, last_ok = 1, term = 0;
  for (;;) {
    if (iter >= max_iters) { term = 0; break; }
    if (last_ok && gmax <= gtol) { term = 1; break; }
    if (mu <= min_radius) { term = 4; break; }
    ++iter;
    double model_change = 0;
    int rc = tr_step(H, g, act, c, mu, N, P, use_schur, delta, &model_change);
    if (rc || !(model_change > 0.0)) {
      if (++invalid >= 5) { term = 5; if (iter < 64) sm.cost_hist[iter] = cost; break; }
      mu /= nu; nu *= 2; last_ok = 0; sm.num_unsuccessful++;
      if (iter < 64) sm.cost_hist[iter] = cost;
      continue;
    }
    invalid = 0;
    state_copy xs;
    state_save(w, &xs);
    double cand_cost;
    if (constrained) {
      /* DoLineSearch: g here is the Jacobi-scaled gradient gs = c .* g and delta = c .* y, so g^T delta = sum gs_i delta_i / c_i */
      double gd = 0, dmax = 0;
      for (int i = 0; i < N; ++i)
        if (act[i]) { gd += g[i] * delta[i] / c[i]; if (fabs(delta[i]) > dmax) dmax = fabs(delta[i]); }
      ls_sample initial = {0.0, cost, gd, 1, 1}, previous = {0, 0, 0, 0, 0}, current = {1.0, 0, 0, 0, 0};
      ctvo_plus(w, delta);
      if (!H2) H2 = (double *)malloc(sizeof(double) * (size_t)N * N);
      current.value = ctvo_build_normal(w, H2, g2);   /* CUBIC interpolation: value and gradient at every trial point */
      current.gradient = 0;
      for (int i = 0; i < N; ++i) if (act[i]) current.gradient += g2[i] * delta[i];
      current.value_is_valid = isfinite(current.value) && isfinite(current.gradient);
      current.gradient_is_valid = current.value_is_valid;
      int ls_iters = 0, ls_ok = 1;
      while (!current.value_is_valid || current.value > cost + 1e-4 * gd * current.x) {
        if (++ls_iters >= 20) { ls_ok = 0; break; }
        double step;
        if (!current.value_is_valid) {
          step = fmin(fmax(current.x * 0.5, 1e-3 * current.x), 0.6 * current.x);
        } else {
          ls_sample smp[3];
          int ns = 0;
          smp[ns++] = initial; smp[ns++] = current;
          if (previous.value_is_valid) smp[ns++] = previous;
          step = poly_minimize_interpolating(smp, ns, 1e-3 * current.x, 0.6 * current.x);
        }
        if (step * dmax < 1e-9) { ls_ok = 0; break; }
        previous = current;
        current.x = step;
        state_restore(w, &xs);
        for (int i = 0; i < N; ++i) dtrial[i] = step * delta[i];
        ctvo_plus(w, dtrial);
        current.value = ctvo_build_normal(w, H2, g2);
        current.gradient = 0;
        for (int i = 0; i < N; ++i) if (act[i]) current.gradient += g2[i] * delta[i];
        current.value_is_valid = isfinite(current.value) && isfinite(current.gradient);
        current.gradient_is_valid = current.value_is_valid;
      }
      sm.num_line_search_steps += ls_iters;
      const double alpha = ls_ok ? current.x : 1.0;
      if (alpha != 1.0) {
        sm.num_line_search_reduced++;
        for (int i = 0; i < N; ++i) delta[i] *= alpha;
      }
      state_restore(w, &xs);
      ctvo_plus(w, delta);
      cand_cost = ctvo_cost(w);
    } else {
      ctvo_plus(w, delta);
      cand_cost = ctvo_cost(w);
    }
    double d2;
    state_norms(w, &xs, act, NULL, &d2, NULL);
    const double step_norm = sqrt(d2);
    if (step_norm <= ptol * (x_norm + ptol)) { /* ParameterToleranceReached: candidate dropped */
      state_restore(w, &xs); state_free(&xs); term = 2;
      if (iter < 64) sm.cost_hist[iter] = cost;
      break;
    }
    const double cost_change = cost - cand_cost;
    if (fabs(cost_change) <= ftol * cost) { /* FunctionToleranceReached: candidate dropped */
      state_restore(w, &xs); state_free(&xs); term = 3;
      if (iter < 64) sm.cost_hist[iter] = cost;
      break;
    }
    const double rel = cost_change / model_change;
    if (rel > min_rel_dec) {
      state_free(&xs);
      cost = ctvo_build_normal(w, H, g);
      gmax = gradient_max_norm(w, g, act);
      mask_scale(H, g, act, c, N);
      state_norms(w, NULL, act, &x2, NULL, NULL);
      x_norm = sqrt(x2);
      double t = 2.0 * rel - 1.0, f = 1.0 - t * t * t;
      if (f < 1.0 / 3.0) f = 1.0 / 3.0;
      mu = mu / f;
      if (mu > max_radius) mu = max_radius;
      nu = 2.0; last_ok = 1; sm.num_successful++;
    } else {
      state_restore(w, &xs); state_free(&xs);
      mu /= nu; nu *= 2; last_ok = 0; sm.num_unsuccessful++;
    }
    if (iter < 64) sm.cost_hist[iter] = cost;
  }
  sm.iterations = iter;
  sm.termination = term;
  sm.final_cost = cost;
  sm.final_radius = mu;
  if (out) *out = sm;
  free(H); free(g); free(c); free(delta); free(act); free(g2); free(dtrial); free(H2);
  return term;
}

/* ------------------------------------------------------------------ trajectory query
 * Se3Spline::poseNs (se3_spline.h:391-399) = So3Spline::evaluate (so3_spline.h:240-289) +
 * RdSpline::evaluate; transVelWorld (:369-372), rotVelBody (:377-380, so3_spline.h:291-322),
 * transAccelWorld (:361-364).                                                          */
void ctvo_spline_eval(const ctvo_window *w, int n, const int64_t *t_ns, double *pose7, double *vel3, double *omega3,
                      double *acc3) {
  const double idt = inv_dt(w);
  for (int i = 0; i < n; ++i) {
    int s; double u;
    t_index(w, t_ns[i], &s, &u);
    const double *q4 = w->quat + 4 * s, *p4 = w->pos + 3 * s;
    if (pose7) {
      double coeff[4], res[4];
      basis(1, 0, u, 1.0, coeff);
      memcpy(res, q4, 32);
      for (int k = 0; k < 3; ++k) {
        double r0i[4], r01[4], delta[3], e[4];
        q_inv(q4 + 4 * k, r0i);
        q_mul(r0i, q4 + 4 * (k + 1), r01);
        ctvo_so3_log(r01, delta);
        for (int c = 0; c < 3; ++c) delta[c] *= coeff[k + 1];
        ctvo_so3_exp(delta, e);
        q_mul(res, e, res);
      }
      eval_rd(p4, 0, u, 1.0, pose7 + 7 * i, NULL);
      memcpy(pose7 + 7 * i + 3, res, 32);
    }
    if (vel3) eval_rd(p4, 1, u, idt, vel3 + 3 * i, NULL);
    if (acc3) eval_rd(p4, 2, u, idt * idt, acc3 + 3 * i, NULL);
    if (omega3) eval_omega(q4, u, idt, omega3 + 3 * i);
  }
}

/* ------------------------------------------------------------------------------------------------ gauge restore
 * reference trajectory_manager.cpp:485-516 (double2vector), utility.h:74-113 (R2ypr in degrees, ypr2R = Rz Ry Rx). */
static void gr_q2R(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
static void gr_R2ypr(const double R[9], double ypr[3]) {
  const double n0 = R[0], n1 = R[3], n2 = R[6], o0 = R[1], o1 = R[4], a0 = R[2], a1 = R[5];
  const double y = atan2(n1, n0);
  const double p = atan2(-n2, n0 * cos(y) + n1 * sin(y));
  const double r = atan2(a0 * sin(y) - a1 * cos(y), -o0 * sin(y) + o1 * cos(y));
  ypr[0] = y / CTVO_PI * 180.0; ypr[1] = p / CTVO_PI * 180.0; ypr[2] = r / CTVO_PI * 180.0;
}
void ctvo_gauge_restore(int K, double *quat, double *pos, int knot, const double q0[4], const double t0[3]) {
  double R0[9], R00[9], e0[3], e00[3], Rd[9], td[3];
  gr_q2R(q0, R0);
  gr_q2R(quat + 4 * knot, R00);
  gr_R2ypr(R0, e0);
  gr_R2ypr(R00, e00);
  if (fabs(fabs(e0[1]) - 90.0) < 1.0 || fabs(fabs(e00[1]) - 90.0) < 1.0) {   /* Euler singularity: R0 R00^T */
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += R0[3 * i + k] * R00[3 * j + k];
        Rd[3 * i + j] = s;
      }
  } else {
    const double y = (e0[0] - e00[0]) / 180.0 * CTVO_PI;
    Rd[0] = cos(y); Rd[1] = -sin(y); Rd[2] = 0; Rd[3] = sin(y); Rd[4] = cos(y); Rd[5] = 0; Rd[6] = 0; Rd[7] = 0; Rd[8] = 1;
  }
  const double *p00 = pos + 3 * knot;
  for (int i = 0; i < 3; ++i) td[i] = t0[i] - (Rd[3 * i] * p00[0] + Rd[3 * i + 1] * p00[1] + Rd[3 * i + 2] * p00[2]);
  /* rot_diff as a unit quaternion (Sophus SO3(matrix) -> Eigen::Quaterniond(R): trace / largest-diagonal branches) */
  double qd[4];
  {
    const double tr = Rd[0] + Rd[4] + Rd[8];
    if (tr > 0) { const double s = sqrt(tr + 1.0) * 2; qd[3] = 0.25 * s; qd[0] = (Rd[7] - Rd[5]) / s; qd[1] = (Rd[2] - Rd[6]) / s; qd[2] = (Rd[3] - Rd[1]) / s; }
    else if (Rd[0] > Rd[4] && Rd[0] > Rd[8]) { const double s = sqrt(1.0 + Rd[0] - Rd[4] - Rd[8]) * 2; qd[3] = (Rd[7] - Rd[5]) / s; qd[0] = 0.25 * s; qd[1] = (Rd[1] + Rd[3]) / s; qd[2] = (Rd[2] + Rd[6]) / s; }
    else if (Rd[4] > Rd[8]) { const double s = sqrt(1.0 + Rd[4] - Rd[0] - Rd[8]) * 2; qd[3] = (Rd[2] - Rd[6]) / s; qd[0] = (Rd[1] + Rd[3]) / s; qd[1] = 0.25 * s; qd[2] = (Rd[5] + Rd[7]) / s; }
    else { const double s = sqrt(1.0 + Rd[8] - Rd[0] - Rd[4]) * 2; qd[3] = (Rd[3] - Rd[1]) / s; qd[0] = (Rd[2] + Rd[6]) / s; qd[1] = (Rd[5] + Rd[7]) / s; qd[2] = 0.25 * s; }
  }
  for (int k = knot; k < K; ++k) {   /* knot <- SE3(Rd, td) * knot : q <- qd * q (Hamilton, renormalised), p <- Rd p + td */
    const double *qk = quat + 4 * k;
    double q[4];
    q[0] = qd[3] * qk[0] + qd[0] * qk[3] + qd[1] * qk[2] - qd[2] * qk[1];
    q[1] = qd[3] * qk[1] - qd[0] * qk[2] + qd[1] * qk[3] + qd[2] * qk[0];
    q[2] = qd[3] * qk[2] + qd[0] * qk[1] - qd[1] * qk[0] + qd[2] * qk[3];
    q[3] = qd[3] * qk[3] - qd[0] * qk[0] - qd[1] * qk[1] - qd[2] * qk[2];
    const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= nq;
    double *pk = pos + 3 * k, pn[3];
    for (int i = 0; i < 3; ++i) pn[i] = Rd[3 * i] * pk[0] + Rd[3 * i + 1] * pk[1] + Rd[3 * i + 2] * pk[2] + td[i];
    for (int i = 0; i < 4; ++i) quat[4 * k + i] = q[i];
    for (int i = 0; i < 3; ++i) pk[i] = pn[i];
  }
}

/* ------------------------------------------------------------------------------------------------ marginalisation
 * reference marginalization_factor.cpp:189-265 (MarginalizationInfo::marginalize); Eigen::SelfAdjointEigenSolver is
 * restated as a cyclic Jacobi iteration (eigenvalues to ~1e-15 relative to the largest). */
void ctvo_sym_eig(int n, double *A, double *ev, double *V) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, dia = 0;
    for (int i = 0; i < n; ++i) {
      dia += A[(size_t)i * n + i] * A[(size_t)i * n + i];
      for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j];
    }
    if (off <= 1e-60 || off <= 1e-32 * dia) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
        const double theta = (aqq - app) / (2.0 * apq);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < n; ++k) {   /* A <- A G (columns p, q) */
          const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = c * akp - s * akq;
          A[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {   /* A <- G^T A (rows p, q) */
          const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = c * apk - s * aqk;
          A[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {   /* V <- V G */
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c * vkp - s * vkq;
          V[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; ++i) ev[i] = A[(size_t)i * n + i];
  for (int i = 0; i < n - 1; ++i) {   /* ascending order (selection sort, columns of V follow) */
    int m = i;
    for (int j = i + 1; j < n; ++j) if (ev[j] < ev[m]) m = j;
    if (m != i) {
      const double te = ev[i]; ev[i] = ev[m]; ev[m] = te;
      for (int k = 0; k < n; ++k) { const double tv = V[(size_t)k * n + i]; V[(size_t)k * n + i] = V[(size_t)k * n + m]; V[(size_t)k * n + m] = tv; }
    }
  }
}

int ctvo_marginalize(const ctvo_window *w, const int8_t *role, double eps, int32_t *kept, double *J0, double *r0) {
  const int N = nN(w);
  int m = 0, n = 0;
  int *im = (int *)malloc(sizeof(int) * N), *ik = (int *)malloc(sizeof(int) * N);
  for (int i = 0; i < N; ++i) { if (role[i] == 1) im[m++] = i; else if (role[i] == 0) ik[n++] = i; }
  if (n <= 0) { free(im); free(ik); return 0; }
  double *H = (double *)malloc(sizeof(double) * (size_t)N * N), *g = (double *)malloc(sizeof(double) * N);
  ctvo_build_normal(w, H, g);
  /* Amm^+ (marginalization_factor.cpp:236-240) */
  double *Amm = (double *)malloc(sizeof(double) * (size_t)(m ? m : 1) * (m ? m : 1)), *em = (double *)malloc(sizeof(double) * (m ? m : 1));
  double *Vm = (double *)malloc(sizeof(double) * (size_t)(m ? m : 1) * (m ? m : 1));
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) Amm[(size_t)i * m + j] = 0.5 * (H[(size_t)im[i] * N + im[j]] + H[(size_t)im[j] * N + im[i]]);
  if (m) ctvo_sym_eig(m, Amm, em, Vm);
  /* T = Amm^+ [Amr | bm]  (m x (n + 1)) through the eigenbasis: V diag(1/e) V^T */
  double *X = (double *)malloc(sizeof(double) * (size_t)(m ? m : 1) * (n + 1)), *Y = (double *)malloc(sizeof(double) * (size_t)(m ? m : 1) * (n + 1));
  for (int a = 0; a < m; ++a)          /* Y = diag(1/e) V^T [Amr | bm] */
    for (int c = 0; c <= n; ++c) {
      double s = 0;
      for (int i = 0; i < m; ++i) s += Vm[(size_t)i * m + a] * (c < n ? H[(size_t)im[i] * N + ik[c]] : g[im[i]]);
      Y[(size_t)a * (n + 1) + c] = em[a] > eps ? s / em[a] : 0.0;
    }
  for (int i = 0; i < m; ++i)
    for (int c = 0; c <= n; ++c) {
      double s = 0;
      for (int a = 0; a < m; ++a) s += Vm[(size_t)i * m + a] * Y[(size_t)a * (n + 1) + c];
      X[(size_t)i * (n + 1) + c] = s;
    }
  /* A' = Arr - Arm X[:, :n],  b' = br - Arm X[:, n]  (:242-248) */
  double *Ar = (double *)malloc(sizeof(double) * (size_t)n * n), *br = (double *)malloc(sizeof(double) * n);
  for (int r = 0; r < n; ++r) {
    for (int c = 0; c < n; ++c) {
      double s = H[(size_t)ik[r] * N + ik[c]];
      for (int i = 0; i < m; ++i) s -= H[(size_t)ik[r] * N + im[i]] * X[(size_t)i * (n + 1) + c];
      Ar[(size_t)r * n + c] = s;
    }
    double s = g[ik[r]];
    for (int i = 0; i < m; ++i) s -= H[(size_t)ik[r] * N + im[i]] * X[(size_t)i * (n + 1) + n];
    br[r] = s;
  }
  /* A' = V S V^T; J0 = sqrt(S) V^T, r0 = S^-1/2 V^T b'  (:250-262).  (The reference does not symmetrise A' first; it is
   * symmetric up to rounding, and only its symmetric part enters an eigendecomposition.) */
  for (int r = 0; r < n; ++r)
    for (int c = r + 1; c < n; ++c) { const double s = 0.5 * (Ar[(size_t)r * n + c] + Ar[(size_t)c * n + r]); Ar[(size_t)r * n + c] = s; Ar[(size_t)c * n + r] = s; }
  double *er = (double *)malloc(sizeof(double) * n), *Vr = (double *)malloc(sizeof(double) * (size_t)n * n);
  ctvo_sym_eig(n, Ar, er, Vr);
  for (int a = 0; a < n; ++a) {
    const double S = er[a] > eps ? er[a] : 0.0, sq = sqrt(S), isq = S > 0 ? 1.0 / sqrt(S) : 0.0;
    double s = 0;
    for (int i = 0; i < n; ++i) { J0[(size_t)a * n + i] = sq * Vr[(size_t)i * n + a]; s += Vr[(size_t)i * n + a] * br[i]; }
    r0[a] = isq * s;
  }
  for (int i = 0; i < n; ++i) kept[i] = ik[i];
  free(im); free(ik); free(H); free(g); free(Amm); free(em); free(Vm); free(X); free(Y); free(Ar); free(br); free(er); free(Vr);
  return n;
}
