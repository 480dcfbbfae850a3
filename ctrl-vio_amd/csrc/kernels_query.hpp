// kernels_query.hpp -- After the solve: k_gauge_restore (double2vector), k_residual_summary, k_spline_eval (trajectory queries, one window or a batch).
// Part of kernels.hpp (included from there, in order; not a stand-alone header).
#pragma once

namespace ctv {

// ------------------------------------------------------------------------------------------------ update
// 4-DoF gauge restore after a solve (reference TrajectoryManager::double2vector, trajectory_manager.cpp:485-516): one rigid
// transform puts the yaw and the position of knot `knot[w]` back to their pre-solve values (q0, t0) and is applied to
// knots knot..K-1.  One workgroup per requested window; all fp64.  Utility::R2ypr / ypr2R: visual_odometry/utility.h:74-113.
__global__ void k_gauge_restore(Dev d, int n, const int32_t *ids, const int32_t *knot, const double *q0, const double *t0) {
  const int e = blockIdx.x;
  if (e >= n) return;
  const WinMeta &m = d.wins[ids[e]];
  const int K = m.K, k0 = knot[e], base = m.knot0;
  __shared__ double sh[16];   // Rd (9), td (3), qd (4)
  if (threadIdx.x == 0) {
    const double *qr = d.quat + 4 * (base + k0), *pr = d.pos + 3 * (base + k0);
    const M3 R0 = q2R(qmk(q0[4 * e], q0[4 * e + 1], q0[4 * e + 2], q0[4 * e + 3]));
    const M3 R00 = q2R(qmk(qr[0], qr[1], qr[2], qr[3]));
    auto ypr = [](const M3 &R, double &y, double &p) {   // degrees
      y = atan2(R.m[3], R.m[0]);
      p = atan2(-R.m[6], R.m[0] * cos(y) + R.m[3] * sin(y)) / 3.14159265358979323846 * 180.0;
      y = y / 3.14159265358979323846 * 180.0;
    };
    double y0, p0, y00, p00;
    ypr(R0, y0, p0);
    ypr(R00, y00, p00);
    M3 Rd;
    if (fabs(fabs(p0) - 90.0) < 1.0 || fabs(fabs(p00) - 90.0) < 1.0) {   // Euler singularity: R0 R00^T
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rd.m[3 * i + j] = R0.m[3 * i] * R00.m[3 * j] + R0.m[3 * i + 1] * R00.m[3 * j + 1] + R0.m[3 * i + 2] * R00.m[3 * j + 2];
    } else {
      const double y = (y0 - y00) / 180.0 * 3.14159265358979323846;
      Rd = m3_id();
      Rd.m[0] = cos(y); Rd.m[1] = -sin(y); Rd.m[3] = sin(y); Rd.m[4] = cos(y);
    }
    for (int i = 0; i < 9; ++i) sh[i] = Rd.m[i];
    for (int i = 0; i < 3; ++i) sh[9 + i] = t0[3 * e + i] - (Rd.m[3 * i] * pr[0] + Rd.m[3 * i + 1] * pr[1] + Rd.m[3 * i + 2] * pr[2]);
    // unit quaternion of Rd (Eigen::Quaterniond(R): trace / largest-diagonal branches)
    double qd[4];
    const double *r = Rd.m, tr = r[0] + r[4] + r[8];
    if (tr > 0) { const double s = sqrt(tr + 1.0) * 2; qd[3] = 0.25 * s; qd[0] = (r[7] - r[5]) / s; qd[1] = (r[2] - r[6]) / s; qd[2] = (r[3] - r[1]) / s; }
    else if (r[0] > r[4] && r[0] > r[8]) { const double s = sqrt(1.0 + r[0] - r[4] - r[8]) * 2; qd[3] = (r[7] - r[5]) / s; qd[0] = 0.25 * s; qd[1] = (r[1] + r[3]) / s; qd[2] = (r[2] + r[6]) / s; }
    else if (r[4] > r[8]) { const double s = sqrt(1.0 + r[4] - r[0] - r[8]) * 2; qd[3] = (r[2] - r[6]) / s; qd[0] = (r[1] + r[3]) / s; qd[1] = 0.25 * s; qd[2] = (r[5] + r[7]) / s; }
    else { const double s = sqrt(1.0 + r[8] - r[0] - r[4]) * 2; qd[3] = (r[3] - r[1]) / s; qd[0] = (r[2] + r[6]) / s; qd[1] = (r[5] + r[7]) / s; qd[2] = 0.25 * s; }
    for (int i = 0; i < 4; ++i) sh[12 + i] = qd[i];
  }
  __syncthreads();   // the reference knot is read before any knot is rewritten
  for (int k = k0 + threadIdx.x; k < K; k += blockDim.x) {
    double *qk = d.quat + 4 * (base + k), *pk = d.pos + 3 * (base + k);
    const double *qd = sh + 12;
    double q[4];
    q[0] = qd[3] * qk[0] + qd[0] * qk[3] + qd[1] * qk[2] - qd[2] * qk[1];
    q[1] = qd[3] * qk[1] - qd[0] * qk[2] + qd[1] * qk[3] + qd[2] * qk[0];
    q[2] = qd[3] * qk[2] + qd[0] * qk[1] - qd[1] * qk[0] + qd[2] * qk[3];
    q[3] = qd[3] * qk[3] - qd[0] * qk[0] - qd[1] * qk[1] - qd[2] * qk[2];
    const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double pn[3];
    for (int i = 0; i < 3; ++i) pn[i] = sh[3 * i] * pk[0] + sh[3 * i + 1] * pk[1] + sh[3 * i + 2] * pk[2] + sh[9 + i];
    for (int i = 0; i < 4; ++i) qk[i] = q[i] / nq;
    for (int i = 0; i < 3; ++i) pk[i] = pn[i];
  }
}

// ------------------------------------------------------------------------------------------------ residual summary
// ResidualSummary::AddResidualInfo (reference trajectory_estimator.cpp:36-67): per factor type, the sum of |r_i| of every
// residual component over all blocks (the cost functions' raw whitened residuals: no robust loss) and the block count.
// Diagnostic entry (fp64 evaluation, one workgroup per call): out = [imu 6 | bias 6 | image 2 | prior pn].
__global__ __launch_bounds__(256) void k_residual_summary(Dev d, int w, double *out) {
  const WinMeta &m = d.wins[w];
  extern __shared__ __attribute__((aligned(16))) double smr[];   // [14 + pn] sums, then [pn] dx
  const int tid = threadIdx.x, n = m.pn;
  double *sums = smr, *dx = smr + 14 + n;
  for (int i = tid; i < 14 + 2 * n; i += 256) smr[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < m.M; i += 256) {
    const int idx = m.imu0 + i;
    const ImuGroup grp = d.groups[d.imu_grp[idx]];
    Knots4 k;
    LocalFrame lf;
    lf.init(d.quat, d.pos, m.knot0 + grp.s);
    lf.load(d.quat, d.pos, m.knot0 + grp.s, k);
    SegConst sc;
    seg_const(k, sc, false);
    double b[6], wgt[6], gy[3], ac[3], r[6];
    const double *bp = d.bias + 6 * (m.bias0 + grp.bias);
    for (int c = 0; c < 6; ++c) { b[c] = bp[c]; wgt[c] = m.imu_w[c]; }
    for (int c = 0; c < 3; ++c) { gy[c] = (double)d.imu_meas[(size_t)c * d.Mtot + idx]; ac[c] = (double)d.imu_meas[(size_t)(3 + c) * d.Mtot + idx]; }
    ImuJac J;
    imu_eval_core(k, sc, (double)d.imu_u[idx], m.inv_dt, lf.rotate(m.gravity), b, gy, ac, wgt, lf.RrefT(), r, false, J);
    for (int c = 0; c < 6; ++c) atomicAdd(&sums[c], fabs(r[c]));
  }
  for (int e = tid; e < m.NB * 6; e += 256) {
    const int b = e / 6, k = e % 6;
    const int bi = d.bc_i[m.bc0 + b], bj = d.bc_j[m.bc0 + b];
    const double r = d.bc_w[(size_t)(m.bc0 + b) * 6 + k] * (d.bias[6 * (m.bias0 + bj) + k] - d.bias[6 * (m.bias0 + bi) + k]);
    atomicAdd(&sums[6 + k], fabs(r));
  }
  for (int i = tid; i < m.Vp; i += 256) {
    const int v = m.vis0 + i;
    if (d.v_win[v] < 0) continue;   // padding slot
    // raw residual at the current state: anchor value and block value evaluated here, pair constants straight from the knots
    // (independent of the tables and of the records the solver keeps)
    const int a = d.v_anc[v];
    int si, sj;
    double ui, uj;
    const double ld = d.ld[w];
    const int rowi = d.a_row[a], rowj = d.v_rowj[v];
    vis_times(m, d.a_t[a], rowi, ld, si, ui);
    vis_times(m, d.v_tj[v], rowj, ld, sj, uj);
    si = max(0, min(si, m.K - 4)); sj = max(0, min(sj, m.K - 4));
    Knots4 gi, gj;
    const double z3[3] = {0, 0, 0};
    load_knots(d.quat, d.pos, m.knot0 + si, z3, gi);
    load_knots(d.quat, d.pos, m.knot0 + sj, z3, gj);
    SegConst sci, scj;
    seg_const(gi, sci, false);
    seg_const(gj, scj, false);
    const Q4 q_CI = qmk(m.q_CI[0], m.q_CI[1], m.q_CI[2], m.q_CI[3]);
    const V3 p_CI = mk(m.p_CI[0], m.p_CI[1], m.p_CI[2]);
    const M3 R = q2R(q_CI);
    M3 RCIT;
    for (int aa = 0; aa < 3; ++aa) for (int bb = 0; bb < 3; ++bb) RCIT.m[3 * aa + bb] = R.m[3 * bb + aa];
    double rec[AREC], r[2];
    vis_anchor_eval<false>(gi.q[0], gi.p, sci, ui, m.inv_dt, q_CI, p_CI, d.a_obs[a], d.a_obs[(size_t)d.Atot + a], (double)rowi,
                           d.rho[m.lm0 + d.v_lm[v]], false, rec);
    VisNullSink sink;
    vis_block_eval<false>(rec, gj.q[0], gj.p, scj, uj, m.inv_dt, RCIT, p_CI, m.img_w, -1.0 /* raw residual */, (double)d.v_obs[v],
                          (double)d.v_obs[(size_t)d.Vtot + v], (double)rowj, r, false, sink);
    atomicAdd(&sums[12], fabs(r[0]));
    atomicAdd(&sums[13], fabs(r[1]));
  }
  if (n > 0) {   // prior r = r0 + J0 dx (MarginalizationFactor::Evaluate, marginalization_factor.cpp:326-353)
    for (int b = tid; b < m.pnb; b += 256) {
      const int kind = d.p_kind[m.pblk0 + b], idx = d.p_index[m.pblk0 + b], off = d.p_off[m.pblk0 + b];
      const double *x = prior_block_ptr(m, kind, idx, d.quat, d.pos, d.bias, d.ld, w);
      const double *x0 = d.p_x0 + 4 * (size_t)(m.pblk0 + b);
      if (kind == 0) {
        const Q4 dq = qmul_raw(qmk(-x0[0], -x0[1], -x0[2], x0[3]), qmk(x[0], x[1], x[2], x[3]));
        const double sg = (dq.w >= 0) ? 2.0 : -2.0;
        dx[off] = sg * dq.x; dx[off + 1] = sg * dq.y; dx[off + 2] = sg * dq.z;
      } else {
        const int sz = (kind == 4) ? 1 : 3;
        for (int k = 0; k < sz; ++k) dx[off + k] = x[k] - x0[k];
      }
    }
    __syncthreads();
    const double *pJ = d.pJ0 + m.pH0, *pr0 = d.pr0 + m.pv0;   // J0 (column-major, as uploaded) and r0
    for (int i = tid; i < n; i += 256) {
      double r = pr0[i];
      for (int j = 0; j < n; ++j) r += pJ[(size_t)j * n + i] * dx[j];
      sums[14 + i] = fabs(r);
    }
  }
  __syncthreads();
  for (int i = tid; i < 14 + n; i += 256) out[i] = sums[i];
}

// ------------------------------------------------------------------------------------------------ trajectory query
// Se3Spline::poseNs / transVelWorld / rotVelBody / transAccelWorld (se3_spline.h:361-399), fp64, one lane per query.
// win_ids == nullptr: every query belongs to window w; otherwise query i belongs to window win_ids[i] (one launch for a whole batch).
__global__ void k_spline_eval(Dev d, int w, const int32_t *win_ids, int n, const long long *t_rel, double *pose7, double *vel3, double *omega3,
                              double *acc3, int *err, SensorExt ext) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (win_ids) w = win_ids[i];
  const WinMeta &m = d.wins[w];
  const long long st = t_rel[i];
  const int s = (int)(st / m.dt_ns);
  if (st < 0 || s < 0 || s + 3 >= m.K) { atomicExch(err, 1); return; }
  const double u = (double)(st % m.dt_ns) / (double)m.dt_ns;
  const double zero3[3] = {0, 0, 0};
  Knots4 k;
  load_knots(d.quat, d.pos, m.knot0 + s, zero3, k);
  SegConst sc;
  seg_const(k, sc, false);
  const double idt = m.inv_dt;
  if (pose7) {
    double c[4];
    basis<false, 0>(u, 1.0, c);
    V3 p = mk(0, 0, 0);
    for (int j = 0; j < 4; ++j) p = p + c[j] * k.p[j];
    Q4 q = eval_R(k.q, sc, u);
    if (ext.on) {   // Trajectory::GetSensorPose (trajectory.cpp:39-56): pose_S_to_G = pose_I_to_G * T_StoI
      p = p + qrot(q, mk(ext.p[0], ext.p[1], ext.p[2]));
      q = qmul(q, qmk(ext.q[0], ext.q[1], ext.q[2], ext.q[3]));
    }
    double *o = pose7 + 7 * (size_t)i;
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
  }
  if (vel3) {
    double c[4];
    basis<false, 1>(u, idt, c);
    V3 p = mk(0, 0, 0);
    for (int j = 0; j < 4; ++j) p = p + c[j] * k.p[j];
    vel3[3 * (size_t)i] = p.x; vel3[3 * (size_t)i + 1] = p.y; vel3[3 * (size_t)i + 2] = p.z;
  }
  if (acc3) {
    double c[4];
    basis<false, 2>(u, idt * idt, c);
    V3 p = mk(0, 0, 0);
    for (int j = 0; j < 4; ++j) p = p + c[j] * k.p[j];
    acc3[3 * (size_t)i] = p.x; acc3[3 * (size_t)i + 1] = p.y; acc3[3 * (size_t)i + 2] = p.z;
  }
  if (omega3) {
    const V3 o = eval_omega(sc, u, idt);
    omega3[3 * (size_t)i] = o.x; omega3[3 * (size_t)i + 1] = o.y; omega3[3 * (size_t)i + 2] = o.z;
  }
}

}  // namespace ctv
