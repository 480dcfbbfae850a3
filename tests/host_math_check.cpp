// host_math_check.cpp -- TEST-ONLY g++ build of the device math headers (ctrl-vio_amd/csrc/so3.hpp,
// factors.hpp) so that the per-block residual/Jacobian code of the HIP kernels can be checked against
// the oracle on a machine without a GPU (pytest -m "not gpu").  This library is never loaded by the
// product: libctvio.so has no CPU path.
#include "../ctrl-vio_amd/csrc/factors.hpp"

using namespace ctv;

namespace {
template <class T> struct ImuSink {
  T *J;
  void put(int row, int col, T v) { J[row * 30 + col] = v; }
};
template <class T> struct VisSink {
  T *J;
  void put(int col, T j0, T j1) { J[col] = j0; J[50 + col] = j1; }
};

template <class T>
void imu_eval_t(const double *q, const double *p, double u, double idt, const double *g, const double *bias,
                const double *gyro, const double *acc, const double *w, double *r, double *J) {
  Knots4<T> k;
  for (int i = 0; i < 4; ++i) {
    k.q[i] = qmk<T>((T)q[4 * i], (T)q[4 * i + 1], (T)q[4 * i + 2], (T)q[4 * i + 3]);
    k.p[i] = mk<T>((T)(p[3 * i] - p[0]), (T)(p[3 * i + 1] - p[1]), (T)(p[3 * i + 2] - p[2]));
  }
  SegConst<T> sc;
  seg_const(k, sc, true);
  T b[6], gy[3], ac[3], ww[6], rr[6], JJ[180];
  for (int i = 0; i < 6; ++i) { b[i] = (T)bias[i]; ww[i] = (T)w[i]; }
  for (int i = 0; i < 3; ++i) { gy[i] = (T)gyro[i]; ac[i] = (T)acc[i]; }
  for (int i = 0; i < 180; ++i) JJ[i] = 0;
  ImuSink<T> sink{JJ};
  imu_eval<T>(k, sc, (T)u, (T)idt, mk<T>((T)g[0], (T)g[1], (T)g[2]), b, gy, ac, ww, rr, true, sink);
  for (int i = 0; i < 6; ++i) r[i] = rr[i];
  for (int i = 0; i < 180; ++i) J[i] = JJ[i];
}

template <class T>
double visual_eval_t(const double *qi, const double *pi, const double *qj, const double *pj, double ui, double uj,
                     double idt, const double *q_CI, const double *p_CI, double img_w, double cauchy_a, const double *obs,
                     double rowi, double rowj, double d_inv, double *r, double *J) {
  Knots4<T> ki, kj;
  for (int i = 0; i < 4; ++i) {
    ki.q[i] = qmk<T>((T)qi[4 * i], (T)qi[4 * i + 1], (T)qi[4 * i + 2], (T)qi[4 * i + 3]);
    kj.q[i] = qmk<T>((T)qj[4 * i], (T)qj[4 * i + 1], (T)qj[4 * i + 2], (T)qj[4 * i + 3]);
    // common origin = first knot of the i-end
    ki.p[i] = mk<T>((T)(pi[3 * i] - pi[0]), (T)(pi[3 * i + 1] - pi[1]), (T)(pi[3 * i + 2] - pi[2]));
    kj.p[i] = mk<T>((T)(pj[3 * i] - pi[0]), (T)(pj[3 * i + 1] - pi[1]), (T)(pj[3 * i + 2] - pi[2]));
  }
  Calib<T> cal;
  cal.q_CI = qmk<T>((T)q_CI[0], (T)q_CI[1], (T)q_CI[2], (T)q_CI[3]);
  cal.p_CI = mk<T>((T)p_CI[0], (T)p_CI[1], (T)p_CI[2]);
  cal.img_w = (T)img_w;
  cal.cauchy_a = (T)cauchy_a;
  T rr[2], JJ[100];
  for (int i = 0; i < 100; ++i) JJ[i] = 0;
  VisSink<T> sink{JJ};
  T cost = visual_eval<T>(ki, kj, (T)ui, (T)uj, (T)idt, cal, (T)obs[0], (T)obs[1], (T)obs[2], (T)obs[3], (T)rowi, (T)rowj,
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
