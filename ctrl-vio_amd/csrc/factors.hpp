// factors.hpp -- per-residual-block device evaluation (residual + analytic Jacobian), fp64.  Knots are addressed by global index:
// the 4 active control points of an evaluation at time t are knots s..s+3 with s = (t - t0)/dt, so the reference's SplineMeta /
// parameter-pointer bookkeeping (src/spline/spline_segment.h:131-191, trajectory_estimator.cpp:114-141) disappears.
//
//   imu_eval*                         SplitSpineView::Evaluate (split_spline_view.h:67-214) fused with IMUFactor::Evaluate
//                                     (trajectory_value_factor.h:141-248): general form and the product kernel's staged form
//   vis_anchor_eval / vis_block_eval  ImageFeatureDelayFactor::Evaluate (image_feature_factor.h:63-269) with So3SplineView::EvaluateRp /
//                                     EvaluateRTp / VelocityBody (so3_spline_view.h:136-276,356-411), RdSplineView::evaluate
//                                     (rd_spline_view.h:63-94) and ceres::CauchyLoss + Corrector (restated in the reference at
//                                     marginalization_factor.cpp:39-67), factored through the anchor end
#pragma once
#include "so3.hpp"

namespace ctv {

struct Knots4 {
  Q4 q[4];
  V3 p[4];  // positions relative to a per-block origin (residuals are translation invariant)
};

// Quantities of a 4-knot group that do not depend on the evaluation time u.
struct SegConst {
  V3 d[3];      // d_i = log(R_i^-1 R_{i+1})
  M3 JrI[3];    // Jr^-1(d_i)
  CTV_DI M3 jri(int i) const { return JrI[i]; }
};
// The same with Jr^-1(d_i) left in the per-window table (k_knot_prep) and fetched where it is used: the 27 values per knot
// group are not held in registers across the whole block evaluation (54 VGPRs per spline end).
struct SegConstLazy {
  V3 d[3];
  const double *tab;   // [3][9]
  CTV_DI M3 jri(int i) const {
    M3 J;
#pragma unroll
    for (int e = 0; e < 9; ++e) J.m[e] = (double)tab[9 * i + e];
    return J;
  }
};
// The same with everything held by value in wave-uniform registers (the caller has made the values scalar): Jr^-1 costs no vector
// registers and no loads inside the evaluation loop.
struct SegConstS {
  V3 d[3];
  M3 JrI[3];
  CTV_DI const M3 &jri(int i) const { return JrI[i]; }
};
CTV_DI void seg_const_lazy(const double *kd, const double *kjri, SegConstLazy &sc) {
#pragma unroll
  for (int i = 0; i < 3; ++i) sc.d[i] = mk((double)kd[3 * i], (double)kd[3 * i + 1], (double)kd[3 * i + 2]);
  sc.tab = kjri;
}
CTV_DI void seg_const(const Knots4 &k, SegConst &sc, bool want_jac) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sc.d[i] = so3_log(qmul(qconj(k.q[i]), k.q[i + 1]));
    if (want_jac) sc.JrI[i] = so3_Jr_inv(sc.d[i]);
  }
}

// The same from the per-window tables k_knot_prep fills once per state (d of every consecutive knot pair and Jr^-1(d)): the per-pair
// quantities do not depend on the residual block, so they are hoisted out of the per-block evaluation (the reference recomputes
// them inside every factor, so3_spline_view.h:160-166).  One table entry: d = log(q_a^-1 q_b) and Jr^-1(d).
CTV_DI void knot_pair_const(const double *qa, const double *qb, double *d3, double *jri9) {
  const V3 dd = so3_log(qmul(qconj(qmk(qa[0], qa[1], qa[2], qa[3])), qmk(qb[0], qb[1], qb[2], qb[3])));
  d3[0] = dd.x; d3[1] = dd.y; d3[2] = dd.z;
  if (jri9) {
    const M3 J = so3_Jr_inv(mk(dd.x, dd.y, dd.z));
#pragma unroll
    for (int e = 0; e < 9; ++e) jri9[e] = J.m[e];
  }
}
CTV_DI void seg_const_load(const double *kd, const double *kjri, SegConst &sc, bool want_jac) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sc.d[i] = mk((double)kd[3 * i], (double)kd[3 * i + 1], (double)kd[3 * i + 2]);
    if (want_jac) {
#pragma unroll
      for (int e = 0; e < 9; ++e) sc.JrI[i].m[e] = (double)kjri[9 * i + e];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// IMU block.  Local column order of J (6 x 30): rot k0..k3 (12) | pos k0..k3 (12) | bg (3) | ba (3).
// Sink::put_col(col, v[6]) receives every column of J (all 6 rows, structural zeros included);
// r[6] is returned whitened.
//
// Local frame (general form only): the caller passes the knots expressed relative to a reference knot,
// q'_k = q_ref^-1 q_k, p'_k = R_ref^T (p_k - p_ref), and gravity as R_ref^T g.  Residuals are invariant under this
// change of gauge and right-perturbation rotation Jacobians are unchanged; only the position Jacobians need
// J_p = J_p' R_ref^T (RrefT).  (The staged form of the product kernel works in the global frame.)
// Jacobian of one IMU block in factored form: the 6 x 30 matrix is w .* [Jw | 0 | I3 | 0 ; Ja | lamA (x) Rinv_g | 0 | I3]
// (trajectory_value_factor.h:198-245).  Consumers read it column by column (imu_emit_cols) or row by row (imu_row_*).
struct ImuJac {
  M3 Jw[4], Ja[4];   // d(gyro) / d(rot knot k), d(accel) / d(rot knot k)
  M3 Rinv_g;         // R(t)^T in the global frame: d(accel) / d(pos knot k) = lamA[k] * Rinv_g
  double lamA[4];
};

template <class SC> 
CTV_DI void imu_eval_core(const Knots4 &k, const SC &sc, double u, double idt, V3 gravity, const double bias[6],
                          const double gyro[3], const double acc[3], const double w[6], const M3 &RrefT, double r[6], bool want_jac, ImuJac &J) {
  double lamR[4], lamW[4];
  double (&lamA)[4] = J.lamA;
  basis<false, 2>(u, idt * idt, lamA);
  basis<true, 0>(u, 1.0, lamR);
  basis<true, 1>(u, idt, lamW);

  V3 accel = mk(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) accel = accel + lamA[i] * k.p[i];

  Q4 Ainv[3], accq = qmk(0, 0, 0, 1);
  M3 Apost[4], JrK[3];
  Apost[3] = m3_id();
#pragma unroll
  for (int i = 2; i >= 0; --i) {
    const V3 nkd = (-lamR[i + 1]) * sc.d[i];
    Ainv[i] = so3_exp(nkd);
    accq = qmul_unit(accq, Ainv[i]);
    if (want_jac) { Apost[i] = q2R(accq); JrK[i] = so3_Jr(nkd); }
  }
  V3 om[4];
  om[0] = mk(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) om[i + 1] = qrot(Ainv[i], om[i]) + lamW[i + 1] * sc.d[i];

  const Q4 Rinv_q = qmul_unit(accq, qconj(k.q[0]));
  const V3 ag = accel + gravity;
  const V3 a_pred = qrot(Rinv_q, ag);
  r[0] = w[0] * (om[3].x - (gyro[0] - bias[0]));
  r[1] = w[1] * (om[3].y - (gyro[1] - bias[1]));
  r[2] = w[2] * (om[3].z - (gyro[2] - bias[2]));
  r[3] = w[3] * (a_pred.x - (acc[0] - bias[3]));
  r[4] = w[4] * (a_pred.y - (acc[1] - bias[4]));
  r[5] = w[5] * (a_pred.z - (acc[2] - bias[5]));
  if (!want_jac) return;

  // gyro rows: d(omega)/d(d_j), split_spline_view.h:157-181
  M3 (&Jw)[4] = J.Jw;
  M3 (&Ja)[4] = J.Ja;
#pragma unroll
  for (int i = 0; i < 4; ++i) { Jw[i] = m3_zero(); Ja[i] = m3_zero(); }
  {
    M3 dod = scale(Apost[1], lamW[1]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i > 0) dod = add(scale(mul(mul_hat(Apost[i], om[i]), JrK[i]), lamR[i + 1]), scale(Apost[i + 1], lamW[i + 1]));
      const M3 JrIi = sc.jri(i);
      Jw[i] = sub(Jw[i], mulT(dod, JrIi));
      Jw[i + 1] = add(Jw[i + 1], mul(dod, JrIi));
    }
  }
  // accel rows, split_spline_view.h:183-211 (three R_accum entries: the reference's 2-entry array is a bug)
  const M3 Rinv = q2R(Rinv_q);
  J.Rinv_g = mul(Rinv, RrefT);  // R(t)^T in the global frame, for the position-knot columns
  {
    const M3 lhs = mul_hat(Rinv, ag);
    M3 Racc = q2R(k.q[0]);
    Ja[0] = add(Ja[0], mul(lhs, Racc));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i > 0) Racc = mulT(Racc, q2R(Ainv[i - 1]));
      const M3 dad = scale(mul(mul(lhs, Racc), JrK[i]), lamR[i + 1]);
      const M3 JrIi = sc.jri(i);
      Ja[i] = sub(Ja[i], mulT(dad, JrIi));
      Ja[i + 1] = add(Ja[i + 1], mul(dad, JrIi));
    }
  }
}

// Gyro row a on the product kernel's compact 16 columns [rot k0..k3 (12) | bg (3) | r]
CTV_DI void imu_row_gyro2(const M3 (&Jw)[4], const double w[6], const double r[6], int a, double out[16]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int b = 0; b < 3; ++b) out[3 * kk + b] = w[a] * Jw[kk].m[3 * a + b];
  out[12] = out[13] = out[14] = 0.0;
  out[12 + a] = w[a];
  out[15] = r[a];
}

// ---- Staged form (k_imu_linearize_f64's product path: values, then the gyro Jacobians, then the accelerometer Jacobians, each stage's
// rows consumed before the next starts, so the two 36-entry Jacobians are never live together), algebraically the same Jacobian with the rotation chain of the accelerometer
// rows rewritten: with A_i = exp(lamR[i+1] d_i), Apost_i = (A_i .. A_2)^T and R(t)^T = Apost_0 R_0^T,
//   R(t)^T hat(a + g) R_0 A_0 .. A_{i-1} = Apost_i hat(b_i),   b_0 = R_0^T (a + g),  b_{i+1} = A_i^T b_i
// (R hat(v) R^T = hat(R v) applied i + 1 times), so d(accel)/d(d_i) = lamR[i+1] Apost_i hat(b_i) Jr(-lamR[i+1] d_i): the very shape of the
// gyro rows (Apost_i hat(omega_i) Jr) -- no rotation matrices of q_0 / A_0 / A_1, no running product.  Small-angle series only
// (so3_exp_small / so3_Jr_small): the kernel checks |d_i| < 0.5 for the group and takes the general body otherwise.
struct ImuMid3 {
  M3 Apost[3];           // Apost[i] = (A_i .. A_2)^T
  M3 JrK[3];             // Jr(-lamR[i + 1] d_i)
  V3 om1, om2;           // omega recursion
  V3 b[3];               // b_i above
  double lamR[4], lamW[4], lamA[4];
};
// gc: the group's other constants, read where they are used (the kernel keeps them in LDS: one broadcast read each per pass instead of
// a register pair for the whole loop): [0..11] knot positions relative to knot 0, [12..20] R_0^T (row major), [21..23] gravity, [24..29] bias
template <class SC> 
CTV_DI void imu_eval_values3(const double *gc, const SC &sc, double u, double idt, const double gyro[3], const double acc[3], const double w[6], double r[6], ImuMid3 &md) {
  basis<false, 2>(u, idt * idt, md.lamA);
  basis<true, 0>(u, 1.0, md.lamR);
  basis<true, 1>(u, idt, md.lamW);
  V3 ag = mk(gc[21], gc[22], gc[23]);   // spline acceleration + gravity (world)
#pragma unroll
  for (int i = 1; i < 4; ++i) ag = ag + md.lamA[i] * mk(gc[3 * i], gc[3 * i + 1], gc[3 * i + 2]);   // (p_0 - p_0 = 0)
  Q4 Ainv[3], accq;
#pragma unroll
  for (int i = 2; i >= 0; --i) {
    const V3 nkd = (-md.lamR[i + 1]) * sc.d[i];
    Ainv[i] = so3_exp_small(nkd);
    accq = i == 2 ? Ainv[2] : qmul_unit(accq, Ainv[i]);
    md.Apost[i] = q2R(accq);
    md.JrK[i] = so3_Jr_small(nkd);
  }
  V3 om3;
  md.om1 = md.lamW[1] * sc.d[0];
  md.om2 = qrot(Ainv[1], md.om1) + md.lamW[2] * sc.d[1];
  om3 = qrot(Ainv[2], md.om2) + md.lamW[3] * sc.d[2];
  md.b[0] = mk(gc[12] * ag.x + gc[13] * ag.y + gc[14] * ag.z, gc[15] * ag.x + gc[16] * ag.y + gc[17] * ag.z,
                  gc[18] * ag.x + gc[19] * ag.y + gc[20] * ag.z);   // R_0^T (a + g)
  md.b[1] = qrot(Ainv[0], md.b[0]);
  md.b[2] = qrot(Ainv[1], md.b[1]);
  const V3 a_pred = qrot(Ainv[2], md.b[2]);   // = R(t)^T (a + g)
  r[0] = w[0] * (om3.x - (gyro[0] - gc[24]));
  r[1] = w[1] * (om3.y - (gyro[1] - gc[25]));
  r[2] = w[2] * (om3.z - (gyro[2] - gc[26]));
  r[3] = w[3] * (a_pred.x - (acc[0] - gc[27]));
  r[4] = w[4] * (a_pred.y - (acc[1] - gc[28]));
  r[5] = w[5] * (a_pred.z - (acc[2] - gc[29]));
}
// M (3 x 3) against the knot-pair table: J[i] -= M JrI_i^T, J[i + 1] += M JrI_i
template <class SC> CTV_DI void imu_apply_pair(const M3 &M, const SC &sc, int i, M3 &Ji, M3 &Ji1, bool first) {
  const M3 JrIi = sc.jri(i);
  const M3 A = mulT(M, JrIi), B = mul(M, JrIi);
#pragma unroll
  for (int e = 0; e < 9; ++e) { Ji.m[e] = first ? -A.m[e] : Ji.m[e] - A.m[e]; Ji1.m[e] = B.m[e]; }
}
template <class SC> CTV_DI void imu_jac_gyro3(const ImuMid3 &md, const SC &sc, M3 (&Jw)[4]) {
  imu_apply_pair(scale(md.Apost[1], md.lamW[1]), sc, 0, Jw[0], Jw[1], true);
  imu_apply_pair(add(scale(mul(mul_hat(md.Apost[1], md.om1), md.JrK[1]), md.lamR[2]), scale(md.Apost[2], md.lamW[2])), sc, 1, Jw[1], Jw[2], false);
  M3 dod = scale(mul(mul_hat(md.Apost[2], md.om2), md.JrK[2]), md.lamR[3]);
  dod.m[0] += md.lamW[3]; dod.m[4] += md.lamW[3]; dod.m[8] += md.lamW[3];
  imu_apply_pair(dod, sc, 2, Jw[2], Jw[3], false);
}
// R(t)^T = Apost_0 R_0^T (gc[12..20])
template <class SC> 
CTV_DI void imu_jac_accel3(const ImuMid3 &md, const SC &sc, const double *gc, M3 (&Ja)[4], M3 &Rinv_g) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      Rinv_g.m[3 * i + j] = md.Apost[0].m[3 * i] * gc[12 + j] + md.Apost[0].m[3 * i + 1] * gc[15 + j] + md.Apost[0].m[3 * i + 2] * gc[18 + j];
  Ja[0] = mul_hat(md.Apost[0], md.b[0]);
#pragma unroll
  for (int i = 0; i < 3; ++i)
    imu_apply_pair(scale(mul(mul_hat(md.Apost[i], md.b[i]), md.JrK[i]), md.lamR[i + 1]), sc, i, Ja[i], Ja[i + 1], false);
}

// Accelerometer row a in the product kernel's two-tile order: T0 = [rot k0..k3 (12) | ba (3) | r], T1 = [pos k0..k3 (12)] (columns 28..31 unused)
CTV_DI void imu_row_accel3(const M3 (&Ja)[4], const double lamA[4], const M3 &Rinv_g, const double w[6], const double r[6], int a, double out[28]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      out[3 * kk + b] = w[3 + a] * Ja[kk].m[3 * a + b];
      out[16 + 3 * kk + b] = w[3 + a] * lamA[kk] * Rinv_g.m[3 * a + b];
    }
  out[12] = out[13] = out[14] = 0.0;
  out[12 + a] = w[3 + a];
  out[15] = r[3 + a];
}
// trajectory_value_factor.h:198-245, column by column
template <class Sink> CTV_DI void imu_emit_cols(const ImuJac &J, const double w[6], Sink &sink) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      double c6[6], p6[6];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        c6[a] = w[a] * J.Jw[kk].m[3 * a + b];
        c6[3 + a] = w[3 + a] * J.Ja[kk].m[3 * a + b];
        p6[a] = 0.0;
        p6[3 + a] = w[3 + a] * J.lamA[kk] * J.Rinv_g.m[3 * a + b];
      }
      sink.put_col(3 * kk + b, c6);
      sink.put_col(12 + 3 * kk + b, p6);
    }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    double g6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, h6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    g6[a] = w[a];
    h6[3 + a] = w[3 + a];
    sink.put_col(24 + a, g6);
    sink.put_col(27 + a, h6);
  }
}
// The same entries row by row (same expressions, hence bit-identical values):
//   accelerometer row a: 32 columns [rot 0..11 | pos 12..23 | bg 24..26 = 0 | ba 27..29 | residual 30 | 0]
//   gyroscope row a    : its 16 non-zero columns [rot 0..11 | bg (columns 24..26) | residual (column 30)]
CTV_DI void imu_row_accel(const ImuJac &J, const double w[6], const double r[6], int a, double out[32]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      out[3 * kk + b] = w[3 + a] * J.Ja[kk].m[3 * a + b];
      out[12 + 3 * kk + b] = w[3 + a] * J.lamA[kk] * J.Rinv_g.m[3 * a + b];
    }
#pragma unroll
  for (int c = 24; c < 32; ++c) out[c] = 0.0;
  out[27 + a] = w[3 + a];
  out[30] = r[3 + a];
}
CTV_DI void imu_row_gyro(const ImuJac &J, const double w[6], const double r[6], int a, double out[16]) {
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int b = 0; b < 3; ++b) out[3 * kk + b] = w[a] * J.Jw[kk].m[3 * a + b];
  out[12] = out[13] = out[14] = 0.0;
  out[12 + a] = w[a];
  out[15] = r[a];
}

template <class Sink, class SC> 
CTV_DI void imu_eval(const Knots4 &k, const SC &sc, double u, double idt, V3 gravity, const double bias[6],
                     const double gyro[3], const double acc[3], const double w[6], const M3 &RrefT, double r[6], bool want_jac, Sink &sink) {
  ImuJac J;
  imu_eval_core<SC>(k, sc, u, idt, gravity, bias, gyro, acc, w, RrefT, r, want_jac, J);
  if (want_jac) imu_emit_cols(J, w, sink);
}

// ------------------------------------------------------------------------------------------------
// SO(3) spline views on 4 knots.  (Products of unit quaternions -- normalised knots, exp() -- through qmul_unit: Sophus' renormalising
// product without the division, so3.hpp.)
// EvaluateRp (so3_spline_view.h:136-198): returns R(t); J[k] = per-knot 3x3 "partial" Jacobians.
template <class SC, bool SMALL = false> CTV_DI Q4 eval_Rp(const Q4 q[4], const SC &sc, double u, M3 J[4], bool want_jac) {
  double c[4];
  basis<true, 0>(u, 1.0, c);
  Q4 accq = qmk(0, 0, 0, 1);
  M3 Apost[4], JrK[3];
  Apost[3] = m3_id();
#pragma unroll
  for (int i = 2; i >= 0; --i) {
    const V3 kd = c[i + 1] * sc.d[i];
    accq = qmul_unit(accq, so3_exp_sel<SMALL>(neg(kd)));
    if (want_jac) { JrK[i] = so3_Jr_sel<SMALL>(kd); Apost[i] = q2R(accq); }
  }
  const Q4 res = qmul_unit(q[0], qconj(accq));
  if (want_jac) {
    J[0] = Apost[0];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const M3 Jh = scale(mul(Apost[i + 1], JrK[i]), c[i + 1]);
      J[i] = sub(J[i], mulT(Jh, sc.jri(i)));
      J[i + 1] = mul(Jh, sc.jri(i));
    }
  }
  return res;
}
// EvaluateRTp (so3_spline_view.h:208-276): returns R(t)^T.
template <class SC, bool SMALL = false> CTV_DI Q4 eval_RTp(const Q4 q[4], const SC &sc, double u, M3 J[4], bool want_jac) {
  double c[4];
  basis<true, 0>(u, 1.0, c);
  Q4 S[4];
  M3 JrK[3];
  S[0] = q[0];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const V3 kd = c[i + 1] * sc.d[i];
    S[i + 1] = qmul_unit(S[i], so3_exp_sel<SMALL>(kd));
    if (want_jac) JrK[i] = so3_Jr_sel<SMALL>(neg(kd));
  }
  if (want_jac) {
    J[0] = q2R(S[0]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const M3 Jh = scale(mul(q2R(S[i]), JrK[i]), c[i + 1]);
      J[i] = sub(J[i], mulT(Jh, sc.jri(i)));
      J[i + 1] = mul(Jh, sc.jri(i));
    }
  }
  return qconj(S[3]);
}
// VelocityBody value (so3_spline_view.h:356-411)
template <class SC, bool SMALL = false> CTV_DI V3 eval_omega(const SC &sc, double u, double idt) {
  double c[4], dc[4];
  basis<true, 0>(u, 1.0, c);
  basis<true, 1>(u, idt, dc);
  V3 rv = dc[1] * sc.d[0];
#pragma unroll
  for (int i = 1; i < 3; ++i) rv = qrot(so3_exp_sel<SMALL>((-c[i + 1]) * sc.d[i]), rv) + dc[i + 1] * sc.d[i];
  return rv;
}
// R(t) only (So3Spline::evaluate, so3_spline.h:240-289)
template <class SC> CTV_DI Q4 eval_R(const Q4 q[4], const SC &sc, double u) {
  double c[4];
  basis<true, 0>(u, 1.0, c);
  Q4 res = q[0];
#pragma unroll
  for (int i = 0; i < 3; ++i) res = qmul_unit(res, so3_exp(c[i + 1] * sc.d[i]));
  return res;
}

// Streaming forms of the two views: the per-knot partial Jacobians are handed to `f(knot, J)` one at a time, as soon as they
// are final, instead of being returned as four 3 x 3 matrices -- the same operations in the same order (bit-identical
// values), but only ~5 matrices are live at any time (the array forms keep 11, i.e. ~200 fp64 registers for both ends).
//   EvaluateRp : J_3 = H_2 JrI_2 ; J_i = H_{i-1} JrI_{i-1} - H_i JrI_i^T ; J_0 = Apost_0 - H_0 JrI_0^T, H_i = c_{i+1} Apost_{i+1} Jr(c_{i+1} d_i):
//                Apost is built from the last knot backwards, so the knots come out 3, 2, 1, 0.
//   EvaluateRTp: H_i = c_{i+1} R(S_i) Jr(-c_{i+1} d_i), S built forwards: knots come out 0, 1, 2, 3.
template <class SC, bool SMALL = false, class F> CTV_DI void eval_Rp_jac_stream(const SC &sc, double u, F &&f) {
  double c[4];
  basis<true, 0>(u, 1.0, c);
  Q4 accq = qmk(0, 0, 0, 1);
  M3 Ap = m3_id(), pending = m3_zero();
#pragma unroll
  for (int i = 2; i >= 0; --i) {
    const V3 kd = c[i + 1] * sc.d[i];
    const M3 Jh = scale(mul(Ap, so3_Jr_sel<SMALL>(kd)), c[i + 1]);
    const M3 JrIi = sc.jri(i);
    const M3 Jn = mul(Jh, JrIi);
    f(i + 1, i == 2 ? Jn : add(Jn, pending));
    pending = scale(mulT(Jh, JrIi), -1.0);
    accq = qmul_unit(accq, so3_exp_sel<SMALL>(neg(kd)));
    Ap = q2R(accq);
  }
  f(0, add(Ap, pending));
}
template <class SC, bool SMALL = false, class F> CTV_DI void eval_RTp_jac_stream(const Q4 q[4], const SC &sc, double u, F &&f) {
  double c[4];
  basis<true, 0>(u, 1.0, c);
  Q4 S = q[0];
  M3 pending = q2R(S);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const V3 kd = c[i + 1] * sc.d[i];
    const M3 Jh = scale(mul(q2R(S), so3_Jr_sel<SMALL>(neg(kd))), c[i + 1]);
    const M3 JrIi = sc.jri(i);
    f(i, sub(pending, mulT(Jh, JrIi)));
    pending = mul(Jh, JrIi);
    S = qmul_unit(S, so3_exp_sel<SMALL>(kd));
  }
  f(3, pending);
}

// ------------------------------------------------------------------------------------------------
// Visual block (ImageFeatureDelayFactor::Evaluate, image_feature_factor.h:63-269), FACTORED THROUGH THE ANCHOR END.
//
// The reprojection residual depends on the pose unknowns only through the 3-vector x_j (the point in camera j):
//     x_j = R_CI^T (R_Ij^T (p_G - p_Ij) - p_CI),     p_G = R_Ii p_Ii + p_IiinG   (the point in the global frame),
// so every column of the block's 2 x 50 Jacobian is A (2 x 3) times a 3-vector, A = sw J_v R_CI^T R_Ij^T (:184-197), and the
// robust corrector (linear in J) turns A into A~ once.  Everything on the anchor side of p_G -- EvaluateRp and its per-knot
// Jacobians, VelocityBody, p(t_i), v(t_i) (:104-131, 199-216) -- depends on (t_i, row_i, p_i, rho) alone, and the reference gives
// all blocks of a feature the same (t_i, row_i, p_i) (trajectory_manager.cpp:367-383): it is evaluated ONCE per anchor
// (vis_anchor_eval -> a record of AREC doubles) instead of once per block (5-6 times per landmark), and a block evaluates only its
// own j end (vis_block_eval).  Column by column (local order rot_i 12 | pos_i 12 | rot_j 12 | pos_j 12 | rho | ld):
//     rot_i (k, b)  = A~ GR_k[:, b],   GR_k = -R_Ii hat(p_Ii) J^Rp_k              (record)
//     pos_i (k, b)  = cp0[k] A~[:, b]                                              (record: cp0)
//     rot_j (k, b)  = (A~ hat(p_G - p_Ij)) J^RTp_k[:, b]                           (block)
//     pos_j (k, b)  = -cp1[k] A~[:, b]                                             (block: cp1)
//     rho           = A~ y,   y = -(1 / rho) R_Ii R_CI x_ci                        (record)
//     ld            = B~ (R_Ij^T (h - row_j v_j) - row_j Om_j x (R_Ij^T (p_G - p_Ij))),  h = row_i (v_i + R_Ii (Om_i x p_Ii))  (record),
//                     B~ = corrector(sw J_v R_CI^T), A~ = B~ R_Ij^T
// The i-end columns are never materialised per block: the assembly rebuilds them from A~ and the record, the rows of W take
// sum_b (A~^T J~_rho) over the anchor's blocks times the record.
constexpr int AREC = 50;        // doubles per anchor record
constexpr int AR_PG = 0;        // [3]  p_G
constexpr int AR_GR = 3;        // [12][3] GR[c][m] = GR_k(m, b), c = 3 k + b: the three factors of column c are consecutive
constexpr int AR_CP0 = 39;      // [4]  blending coefficients of the position spline at t_i
constexpr int AR_Y = 43;        // [3]
constexpr int AR_H = 46;        // [3]  (entry 49 unused: records are 16-byte multiples)

// q0 / p[4]: rotation of the first active knot and the positions of the four active knots of the anchor end (global frame).
template <bool SMALL, class SC>
CTV_DI void vis_anchor_eval(const Q4 &q0, const V3 p[4], const SC &sc, double u, double idt, const Q4 &q_CI,
                            const V3 &p_CI, double pix, double piy, double rowi, double d_inv, bool want_jac, double *rec) {
  const double inv_d = 1.0 / d_inv;
  const V3 c_i = qrot(q_CI, mk(pix * inv_d, piy * inv_d, inv_d));   // R_CI x_ci
  const V3 p_Ii = c_i + p_CI;
  Q4 qk[4];
  qk[0] = q0;
  const Q4 S_IitoG = eval_Rp<SC, SMALL>(qk, sc, u, (M3 *)nullptr, false);
  const M3 RIiG = q2R(S_IitoG);
  double cp0[4];
  basis<false, 0>(u, 1.0, cp0);
  V3 pI = mk(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) pI = pI + cp0[i] * p[i];
  const V3 pG = mul(RIiG, p_Ii) + pI;
  rec[AR_PG] = pG.x; rec[AR_PG + 1] = pG.y; rec[AR_PG + 2] = pG.z;
  if (!want_jac) return;
  double dcp0[4];
  basis<false, 1>(u, idt, dcp0);
  V3 v_i = mk(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) v_i = v_i + dcp0[i] * p[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) rec[AR_CP0 + i] = cp0[i];
  const V3 y = (-inv_d) * mul(RIiG, c_i);
  rec[AR_Y] = y.x; rec[AR_Y + 1] = y.y; rec[AR_Y + 2] = y.z;
  const V3 Om_i = eval_omega<SC, SMALL>(sc, u, idt);
  const V3 h = rowi * (v_i + mul(RIiG, cross(Om_i, p_Ii)));
  rec[AR_H] = h.x; rec[AR_H + 1] = h.y; rec[AR_H + 2] = h.z;
  const M3 M0 = scale(mul_hat(RIiG, p_Ii), -1.0);     // -R_Ii hat(p_Ii)
  eval_Rp_jac_stream<SC, SMALL>(sc, u, [&](int kk, const M3 &Jk) {
    const M3 G = mul(M0, Jk);
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int mm = 0; mm < 3; ++mm) rec[AR_GR + 3 * (3 * kk + b) + mm] = G.m[3 * mm + b];
  });
}

// Entries of a block's record in Dev::Jt (VT_ROWS doubles, robust-corrected): the assembly's input.
constexpr int VB_JROT = 0;      // [12][2] rotation columns of the j end: entry 2 c + residual row, c = 3 k + b
constexpr int VB_RHO = 24;      // [2]  inverse-depth column
constexpr int VB_LD = 26;       // [2]  line-delay column
constexpr int VB_RES = 28;      // [2]  r~
constexpr int VB_AT = 30;       // [3][2] A~: entry 30 + 2 m + residual row
constexpr int VB_CP1 = 36;      // [4]  blending coefficients of the position spline at t_j

// q0 / p[4]: the j end's knots.  RCIT = R_CI^T (row major).  Emit receives the block record entry by entry -- put(entry, value) --
// with the inverse-depth column first (sinks that form J_rho^T J_c need it up front); r[2] = r~; returns the block's cost rho(s)/2.
// The j end's rotation spline is walked ONCE: A_i = exp(c_{i+1} d_i) serves EvaluateRTp's chain S_{i+1} = S_i A_i
// (so3_spline_view.h:208-276), VelocityBody's conjugates (:356-411) and the Jacobian recursion, and the per-knot partial Jacobians
// J_i = H_{i-1} JrI_{i-1} - H_i JrI_i^T, H_i = c_{i+1} R(S_i) Jr(-c_{i+1} d_i) are never formed: the rotation columns need only
// E J_i with E = A~ hat(p_G - p_Ij) (2 x 3), so the recursion runs on the 2 x 3 products E H_i (a third of the multiplications).
template <bool SMALL, class SC, class Emit>
CTV_DI double vis_block_eval(const double *rec, const Q4 &q0, const V3 p[4], const SC &sc, double u, double idt,
                             const M3 &RCIT, const V3 &p_CI, double sw, double cauchy_a, double pjx, double pjy, double rowj,
                             double r[2], bool want_jac, Emit &emit) {
  double c[4], cp1[4];
  basis<true, 0>(u, 1.0, c);
  basis<false, 0>(u, 1.0, cp1);
  Q4 A[3], S[4];
  S[0] = q0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    A[i] = so3_exp_sel<SMALL>(c[i + 1] * sc.d[i]);
    S[i + 1] = qmul_unit(S[i], A[i]);
  }
  const Q4 S_GtoIj = qconj(S[3]);
  V3 pIj = mk(0, 0, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) pIj = pIj + cp1[i] * p[i];
  const V3 dpg = mk(rec[AR_PG], rec[AR_PG + 1], rec[AR_PG + 2]) - pIj;
  const V3 bj = qrot(S_GtoIj, dpg);             // R_Ij^T (p_G - p_Ij)
  const V3 x_j = mul(RCIT, bj - p_CI);
  const double dji = 1.0 / x_j.z;
  const double r0 = sw * (x_j.x * dji - pjx), r1 = sw * (x_j.y * dji - pjy);
  // robust loss (Cauchy): rho(s) = b log(1 + s/b), rho' = 1/(1+s/b), rho'' = -rho'^2/b; corrector as marginalization_factor.cpp:39-67
  const double s = r0 * r0 + r1 * r1;
  double cost, sq = 1.0, rs = 1.0, alpha_sq = 0.0;
  if (cauchy_a > 0.0) {
    const double b = cauchy_a * cauchy_a, cc = 1.0 / b;
    const double inv = 1.0 / (1.0 + s * cc);
    cost = 0.5 * b * log1p(s * cc);
    const double rho1 = inv, rho2 = -cc * inv * inv;
    sq = sqrt(rho1);
    if (s == 0.0 || rho2 <= 0.0) { rs = sq; alpha_sq = 0.0; }
    else { const double D = 1.0 + 2.0 * s * rho2 / rho1; const double al = 1.0 - sqrt(D); rs = sq / (1.0 - al); alpha_sq = al / s; }
  } else {
    cost = 0.5 * s;
  }
  r[0] = rs * r0;
  r[1] = rs * r1;
  if (!want_jac) return cost;
  // B~ = corrector(sw J_v R_CI^T), J_v = [dji 0 -dji^2 x; 0 dji -dji^2 y] (image_feature_factor.h:184-186)
  double Bt[6], At[6];
  {
    const double fx = -dji * dji * x_j.x, fy = -dji * dji * x_j.y;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double j0 = sw * (dji * RCIT.m[k] + fx * RCIT.m[6 + k]), j1 = sw * (dji * RCIT.m[3 + k] + fy * RCIT.m[6 + k]);
      const double rj = r0 * j0 + r1 * j1;
      Bt[k] = sq * (j0 - alpha_sq * r0 * rj);
      Bt[3 + k] = sq * (j1 - alpha_sq * r1 * rj);
    }
  }
  const M3 RGIj = q2R(S_GtoIj);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int k = 0; k < 3; ++k) At[3 * a + k] = Bt[3 * a] * RGIj.m[k] + Bt[3 * a + 1] * RGIj.m[3 + k] + Bt[3 * a + 2] * RGIj.m[6 + k];
  // inverse depth (image_feature_factor.h:239-248)
  emit.put(VB_RHO, At[0] * rec[AR_Y] + At[1] * rec[AR_Y + 1] + At[2] * rec[AR_Y + 2]);
  emit.put(VB_RHO + 1, At[3] * rec[AR_Y] + At[4] * rec[AR_Y + 1] + At[5] * rec[AR_Y + 2]);
#pragma unroll
  for (int mm = 0; mm < 3; ++mm) { emit.put(VB_AT + 2 * mm, At[mm]); emit.put(VB_AT + 2 * mm + 1, At[3 + mm]); }
#pragma unroll
  for (int i = 0; i < 4; ++i) emit.put(VB_CP1 + i, cp1[i]);
  // line delay (image_feature_factor.h:251-264), in the frame of IMU j
  {
    double dc[4], dcp1[4];
    basis<true, 1>(u, idt, dc);
    basis<false, 1>(u, idt, dcp1);
    V3 v_j = mk(0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) v_j = v_j + dcp1[i] * p[i];
    V3 Om_j = dc[1] * sc.d[0];                  // VelocityBody: conj(A_i) = exp(-c_{i+1} d_i)
#pragma unroll
    for (int i = 1; i < 3; ++i) Om_j = qrot(qconj(A[i]), Om_j) + dc[i + 1] * sc.d[i];
    const V3 hv = mk(rec[AR_H], rec[AR_H + 1], rec[AR_H + 2]) - rowj * v_j;
    const V3 Jx = qrot(S_GtoIj, hv) - rowj * cross(Om_j, bj);
    emit.put(VB_LD, Bt[0] * Jx.x + Bt[1] * Jx.y + Bt[2] * Jx.z);
    emit.put(VB_LD + 1, Bt[3] * Jx.x + Bt[4] * Jx.y + Bt[5] * Jx.z);
  }
  // rotation columns of the j end (image_feature_factor.h:199-216): E = A~ hat(p_G - p_Ij), row by row A~[a] x dpg;
  // knots come out 0, 1, 2, 3
  {
    const V3 E0 = cross(mk(At[0], At[1], At[2]), dpg), E1 = cross(mk(At[3], At[4], At[5]), dpg);
    const double E[6] = {E0.x, E0.y, E0.z, E1.x, E1.y, E1.z};
    auto mul23 = [](const double X[6], const M3 &M, double Y[6]) {          // Y = X M
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) Y[3 * a + k] = X[3 * a] * M.m[k] + X[3 * a + 1] * M.m[3 + k] + X[3 * a + 2] * M.m[6 + k];
    };
    auto mul23T = [](const double X[6], const M3 &M, double Y[6]) {         // Y = X M^T
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 3; ++k) Y[3 * a + k] = X[3 * a] * M.m[3 * k] + X[3 * a + 1] * M.m[3 * k + 1] + X[3 * a + 2] * M.m[3 * k + 2];
    };
    double Ep[6];                                       // E times the pending term of the recursion
    mul23(E, q2R(S[0]), Ep);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double ER[6], EH[6], t6[6];
      if (i == 0) {
#pragma unroll
        for (int e = 0; e < 6; ++e) ER[e] = Ep[e];
      } else {
        mul23(E, q2R(S[i]), ER);
      }
      mul23(ER, so3_Jr_sel<SMALL>((-c[i + 1]) * sc.d[i]), EH);
#pragma unroll
      for (int e = 0; e < 6; ++e) EH[e] *= c[i + 1];
      const M3 JrIi = sc.jri(i);
      mul23T(EH, JrIi, t6);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        emit.put(VB_JROT + 2 * (3 * i + b), Ep[b] - t6[b]);
        emit.put(VB_JROT + 2 * (3 * i + b) + 1, Ep[3 + b] - t6[3 + b]);
      }
      mul23(EH, JrIi, Ep);
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) { emit.put(VB_JROT + 2 * (9 + b), Ep[b]); emit.put(VB_JROT + 2 * (9 + b) + 1, Ep[3 + b]); }
  }
  return cost;
}

}  // namespace ctv
