// host_math_check.cpp -- TEST-ONLY g++ build of the device math headers (ctrl-vio_amd/csrc/so3.hpp,
// factors.hpp) so that the per-block residual/Jacobian code of the HIP kernels can be checked against
// the oracle on a machine without a GPU (pytest -m "not gpu").  This library is never loaded by the
// product: libctvio.so has no CPU path.
#include "../ctrl-vio_amd/csrc/factors.hpp"

using namespace ctv;
constexpr int VT_ROWS_HOST = 40;   // = device_types.hpp: VT_ROWS (entries of a block record)

namespace {
struct ImuSink {
  double *J;
  void put_col(int col, const double v[6]) { for (int r = 0; r < 6; ++r) J[r * 30 + col] = v[r]; }
};
struct RecSink {
  double *e;
  void put(int entry, double v) { e[entry] = v; }
};

// local frame of the reference knot (q_ref, p_ref): the same preparation the kernels do (LocalFrame in kernels.hpp)
struct HostLocalFrame {
  Q4 qi; M3 RT; double o[3];
  HostLocalFrame(const double *q, const double *p) {
    qi = qmk(-q[0], -q[1], -q[2], q[3]);
    const M3 R = q2R(qmk(q[0], q[1], q[2], q[3]));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) RT.m[3 * i + j] = R.m[3 * j + i];
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
  }
  void load(const double *q, const double *p, Knots4 &k) const {
    for (int i = 0; i < 4; ++i) {
      const Q4 ql = qmul(qi, qmk(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]));
      const V3 pl = mul(RT, mk(p[3 * i] - o[0], p[3 * i + 1] - o[1], p[3 * i + 2] - o[2]));
      k.q[i] = qmk(ql.x, ql.y, ql.z, ql.w);
      k.p[i] = mk(pl.x, pl.y, pl.z);
    }
  }
  M3 RrefT() const { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = RT.m[i]; return r; }
  V3 rotate(const double *v) const { const V3 r = mul(RT, mk(v[0], v[1], v[2])); return mk(r.x, r.y, r.z); }
};


void imu_eval_t(const double *q, const double *p, double u, double idt, const double *g, const double *bias,
                const double *gyro, const double *acc, const double *w, double *r, double *J) {
  Knots4 k;
  HostLocalFrame lf(q, p);
  lf.load(q, p, k);
  SegConst sc;   // as on the device: pair constants from the fp64 table (k_knot_prep)
  {
    double dt[9]; double jt[27];
    for (int i = 0; i < 3; ++i) knot_pair_const(q + 4 * i, q + 4 * i + 4, dt + 3 * i, jt + 9 * i);
    seg_const_load(dt, jt, sc, true);
  }
  double b[6], gy[3], ac[3], ww[6], rr[6], JJ[180];
  for (int i = 0; i < 6; ++i) { b[i] = bias[i]; ww[i] = w[i]; }
  for (int i = 0; i < 3; ++i) { gy[i] = gyro[i]; ac[i] = acc[i]; }
  for (int i = 0; i < 180; ++i) JJ[i] = 0;
  ImuSink sink{JJ};
  imu_eval(k, sc, u, idt, lf.rotate(g), b, gy, ac, ww, lf.RrefT(), rr, true, sink);
  for (int i = 0; i < 6; ++i) r[i] = rr[i];
  for (int i = 0; i < 180; ++i) J[i] = JJ[i];
}

// The factored visual block (factors.hpp: vis_anchor_eval + vis_block_eval), composed back into the reference's 2 x 50 Jacobian
// (local column order rot_i 12 | pos_i 12 | rot_j 12 | pos_j 12 | rho | ld) exactly as the kernels consume it: i-end rotation
// columns = A~ GR, position columns = cp0 / -cp1 times A~.
template <bool SMALL>
double visual_eval_t(const double *qi, const double *pi, const double *qj, const double *pj, double ui, double uj,
                     double idt, const double *q_CI, const double *p_CI, double img_w, double cauchy_a, const double *obs,
                     double rowi, double rowj, double d_inv, double *r, double *J) {
  SegConst sci, scj;   // as on the device: pair constants from the fp64 table (k_knot_prep)
  {
    double dt[9], jt[27];
    for (int i = 0; i < 3; ++i) knot_pair_const(qi + 4 * i, qi + 4 * i + 4, dt + 3 * i, jt + 9 * i);
    seg_const_load(dt, jt, sci, true);
    for (int i = 0; i < 3; ++i) knot_pair_const(qj + 4 * i, qj + 4 * i + 4, dt + 3 * i, jt + 9 * i);
    seg_const_load(dt, jt, scj, true);
  }
  V3 Pi[4], Pj[4];
  for (int i = 0; i < 4; ++i) { Pi[i] = mk(pi[3 * i], pi[3 * i + 1], pi[3 * i + 2]); Pj[i] = mk(pj[3 * i], pj[3 * i + 1], pj[3 * i + 2]); }
  const Q4 qci = qmk(q_CI[0], q_CI[1], q_CI[2], q_CI[3]);
  const V3 pci = mk(p_CI[0], p_CI[1], p_CI[2]);
  double rec[AREC];
  vis_anchor_eval<SMALL>(qmk(qi[0], qi[1], qi[2], qi[3]), Pi, sci, ui, idt, qci, pci, obs[0], obs[1], rowi, d_inv, true, rec);
  const M3 R = q2R(qci);
  M3 RCIT;
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) RCIT.m[3 * a + b] = R.m[3 * b + a];
  double blk[VT_ROWS_HOST];
  for (int i = 0; i < VT_ROWS_HOST; ++i) blk[i] = 0;
  RecSink sink{blk};
  const double cost = vis_block_eval<SMALL>(rec, qmk(qj[0], qj[1], qj[2], qj[3]), Pj, scj, uj, idt, RCIT, pci, img_w, cauchy_a, obs[2], obs[3],
                                            rowj, r, true, sink);
  for (int rr = 0; rr < 2; ++rr) {
    double *Jr = J + 50 * rr;
    const double At[3] = {blk[VB_AT + rr], blk[VB_AT + 2 + rr], blk[VB_AT + 4 + rr]};
    for (int c = 0; c < 12; ++c) {
      Jr[c] = At[0] * rec[AR_GR + 3 * c] + At[1] * rec[AR_GR + 3 * c + 1] + At[2] * rec[AR_GR + 3 * c + 2];
      Jr[24 + c] = blk[VB_JROT + 2 * c + rr];
    }
    for (int k = 0; k < 4; ++k)
      for (int b = 0; b < 3; ++b) { Jr[12 + 3 * k + b] = rec[AR_CP0 + k] * At[b]; Jr[36 + 3 * k + b] = -blk[VB_CP1 + k] * At[b]; }
    Jr[48] = blk[VB_RHO + rr];
    Jr[49] = blk[VB_LD + rr];
  }
  return cost;
}
}  // namespace

extern "C" {
void hm_imu_eval(const double *q, const double *p, double u, double idt, const double *g, const double *bias,
                 const double *gyro, const double *acc, const double *w, double *r, double *J) {
  imu_eval_t(q, p, u, idt, g, bias, gyro, acc, w, r, J);
}
double hm_visual_eval(int small_angle, const double *qi, const double *pi, const double *qj, const double *pj, double ui, double uj,
                      double idt, const double *q_CI, const double *p_CI, double img_w, double cauchy_a, const double *obs,
                      double rowi, double rowj, double d_inv, double *r, double *J) {
  if (small_angle) return visual_eval_t<true>(qi, pi, qj, pj, ui, uj, idt, q_CI, p_CI, img_w, cauchy_a, obs, rowi, rowj, d_inv, r, J);
  return visual_eval_t<false>(qi, pi, qj, pj, ui, uj, idt, q_CI, p_CI, img_w, cauchy_a, obs, rowi, rowj, d_inv, r, J);
}
void hm_so3(const double *phi, double *exp_q, double *Jr, double *JrInv, double *log_of_exp) {
  V3 v = mk(phi[0], phi[1], phi[2]);
  Q4 q = so3_exp(v);
  exp_q[0] = q.x; exp_q[1] = q.y; exp_q[2] = q.z; exp_q[3] = q.w;
  M3 a = so3_Jr(v), b = so3_Jr_inv(v);
  for (int i = 0; i < 9; ++i) { Jr[i] = a.m[i]; JrInv[i] = b.m[i]; }
  V3 l = so3_log(q);
  log_of_exp[0] = l.x; log_of_exp[1] = l.y; log_of_exp[2] = l.z;
}
}
