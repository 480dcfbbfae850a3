// host_math_check.cpp -- TEST-ONLY g++ build of the device math headers (ctrl-vio_amd/csrc/so3.hpp,
// factors.hpp) so that the per-block residual/Jacobian code of the HIP kernels can be checked against
// the oracle on a machine without a GPU (pytest -m "not gpu").  This library is never loaded by the
// product: libctvio.so has no CPU path.
#include "../ctrl-vio_amd/csrc/factors.hpp"

using namespace ctv;

namespace {
template <class T> struct ImuSink {
  T *J;
  void put_col(int col, const T v[6]) { for (int r = 0; r < 6; ++r) J[r * 30 + col] = v[r]; }
};
template <class T> struct VisSink {
  T *J;
  void put(int col, T j0, T j1) { J[col] = j0; J[50 + col] = j1; }
  void put_pos(const T *, const T *, const T *) {}
};

// local frame of the reference knot (q_ref, p_ref): the same preparation the kernels do (LocalFrame in kernels.hpp)
template <class T> struct HostLocalFrame {
  Q4<double> qi; M3<double> RT; double o[3];
  HostLocalFrame(const double *q, const double *p) {
    qi = qmk<double>(-q[0], -q[1], -q[2], q[3]);
    const M3<double> R = q2R(qmk<double>(q[0], q[1], q[2], q[3]));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) RT.m[3 * i + j] = R.m[3 * j + i];
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
  }
  void load(const double *q, const double *p, Knots4<T> &k) const {
    for (int i = 0; i < 4; ++i) {
      const Q4<double> ql = qmul(qi, qmk<double>(q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]));
      const V3<double> pl = mul(RT, mk<double>(p[3 * i] - o[0], p[3 * i + 1] - o[1], p[3 * i + 2] - o[2]));
      k.q[i] = qmk<T>((T)ql.x, (T)ql.y, (T)ql.z, (T)ql.w);
      k.p[i] = mk<T>((T)pl.x, (T)pl.y, (T)pl.z);
    }
  }
  M3<T> RrefT() const { M3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = (T)RT.m[i]; return r; }
  V3<T> rotate(const double *v) const { const V3<double> r = mul(RT, mk<double>(v[0], v[1], v[2])); return mk<T>((T)r.x, (T)r.y, (T)r.z); }
};

template <class T>
void imu_eval_t(const double *q, const double *p, double u, double idt, const double *g, const double *bias,
                const double *gyro, const double *acc, const double *w, double *r, double *J) {
  Knots4<T> k;
  HostLocalFrame<T> lf(q, p);
  lf.load(q, p, k);
  SegConst<T> sc;   // as on the device: pair constants from the fp64 table (k_knot_prep)
  {
    double dt[9]; T jt[27];
    for (int i = 0; i < 3; ++i) knot_pair_const<T>(q + 4 * i, q + 4 * i + 4, dt + 3 * i, jt + 9 * i);
    seg_const_load(dt, jt, sc, true);
  }
  T b[6], gy[3], ac[3], ww[6], rr[6], JJ[180];
  for (int i = 0; i < 6; ++i) { b[i] = (T)bias[i]; ww[i] = (T)w[i]; }
  for (int i = 0; i < 3; ++i) { gy[i] = (T)gyro[i]; ac[i] = (T)acc[i]; }
  for (int i = 0; i < 180; ++i) JJ[i] = 0;
  ImuSink<T> sink{JJ};
  imu_eval<T>(k, sc, (T)u, (T)idt, lf.rotate(g), b, gy, ac, ww, lf.RrefT(), rr, true, sink);
  for (int i = 0; i < 6; ++i) r[i] = rr[i];
  for (int i = 0; i < 180; ++i) J[i] = JJ[i];
}

template <class T>
double visual_eval_t(const double *qi, const double *pi, const double *qj, const double *pj, double ui, double uj,
                     double idt, const double *q_CI, const double *p_CI, double img_w, double cauchy_a, const double *obs,
                     double rowi, double rowj, double d_inv, double *r, double *J) {
  Knots4<T> ki, kj;
  HostLocalFrame<T> lf(qi, pi);   // both ends relative to the first knot of the i-end
  lf.load(qi, pi, ki);
  lf.load(qj, pj, kj);
  Calib<T> cal;
  cal.q_CI = qmk<T>((T)q_CI[0], (T)q_CI[1], (T)q_CI[2], (T)q_CI[3]);
  cal.p_CI = mk<T>((T)p_CI[0], (T)p_CI[1], (T)p_CI[2]);
  cal.img_w = (T)img_w;
  cal.cauchy_a = (T)cauchy_a;
  T rr[2], JJ[100];
  for (int i = 0; i < 100; ++i) JJ[i] = 0;
  VisSink<T> sink{JJ};
  SegConst<T> sci, scj;   // as on the device: pair constants from the fp64 table (k_knot_prep)
  {
    double dt[9]; T jt[27];
    for (int i = 0; i < 3; ++i) knot_pair_const<T>(qi + 4 * i, qi + 4 * i + 4, dt + 3 * i, jt + 9 * i);
    seg_const_load(dt, jt, sci, true);
    for (int i = 0; i < 3; ++i) knot_pair_const<T>(qj + 4 * i, qj + 4 * i + 4, dt + 3 * i, jt + 9 * i);
    seg_const_load(dt, jt, scj, true);
  }
  T cost = visual_eval<T>(ki, kj, sci, scj, (T)ui, (T)uj, (T)idt, cal, lf.RrefT(), (T)obs[0], (T)obs[1], (T)obs[2], (T)obs[3], (T)rowi, (T)rowj,
                          (T)d_inv, rr, true, sink);
  r[0] = rr[0]; r[1] = rr[1];
  for (int i = 0; i < 100; ++i) J[i] = JJ[i];
  return (double)cost;
}
}  // namespace

extern "C" {
void hm_imu_eval(int fp32, const double *q, const double *p, double u, double idt, const double *g, const double *bias,
                 const double *gyro, const double *acc, const double *w, double *r, double *J) {
  if (fp32) imu_eval_t<float>(q, p, u, idt, g, bias, gyro, acc, w, r, J);
  else imu_eval_t<double>(q, p, u, idt, g, bias, gyro, acc, w, r, J);
}
double hm_visual_eval(int fp32, const double *qi, const double *pi, const double *qj, const double *pj, double ui, double uj,
                      double idt, const double *q_CI, const double *p_CI, double img_w, double cauchy_a, const double *obs,
                      double rowi, double rowj, double d_inv, double *r, double *J) {
  if (fp32) return visual_eval_t<float>(qi, pi, qj, pj, ui, uj, idt, q_CI, p_CI, img_w, cauchy_a, obs, rowi, rowj, d_inv, r, J);
  return visual_eval_t<double>(qi, pi, qj, pj, ui, uj, idt, q_CI, p_CI, img_w, cauchy_a, obs, rowi, rowj, d_inv, r, J);
}
void hm_so3(int fp32, const double *phi, double *exp_q, double *Jr, double *JrInv, double *log_of_exp) {
  if (fp32) {
    V3<float> v = mk<float>((float)phi[0], (float)phi[1], (float)phi[2]);
    Q4<float> q = so3_exp(v);
    exp_q[0] = q.x; exp_q[1] = q.y; exp_q[2] = q.z; exp_q[3] = q.w;
    M3<float> a = so3_Jr(v), b = so3_Jr_inv(v);
    for (int i = 0; i < 9; ++i) { Jr[i] = a.m[i]; JrInv[i] = b.m[i]; }
    V3<float> l = so3_log(q);
    log_of_exp[0] = l.x; log_of_exp[1] = l.y; log_of_exp[2] = l.z;
  } else {
    V3<double> v = mk<double>(phi[0], phi[1], phi[2]);
    Q4<double> q = so3_exp(v);
    exp_q[0] = q.x; exp_q[1] = q.y; exp_q[2] = q.z; exp_q[3] = q.w;
    M3<double> a = so3_Jr(v), b = so3_Jr_inv(v);
    for (int i = 0; i < 9; ++i) { Jr[i] = a.m[i]; JrInv[i] = b.m[i]; }
    V3<double> l = so3_log(q);
    log_of_exp[0] = l.x; log_of_exp[1] = l.y; log_of_exp[2] = l.z;
  }
}
}
