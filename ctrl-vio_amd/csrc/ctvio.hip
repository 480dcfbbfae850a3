// ctvio.hip -- host runtime + C ABI (include/ctvio.h) of the MI355X sliding-window solve.
//
// The host packs windows (the reference's TrajectoryManager::UpdateTrajectory factor set,
// src/estimator/trajectory_manager.cpp:331-451) into flat HBM arrays, then drives a fixed kernel
// sequence per LM iteration on one HIP stream.  All LM decisions (step validity, acceptance,
// radius update, termination: Ceres 1.14 TrustRegionMinimizer) are taken on
// the device; the host only polls a "windows still running" counter every few iterations.
// There is no CPU fallback: without a HIP device ctvio_create fails with CTVIO_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/ctvio.h"
#include "kernels.hpp"
#include "marginalize.hpp"

namespace ctv {

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return fail(CTVIO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

template <class U> struct DBuf {
  U *p = nullptr;
  size_t n = 0;
  ~DBuf() { release(); }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  hipError_t alloc(size_t count) {
    if (count <= n && p) return hipSuccess;
    release();
    n = std::max<size_t>(count, 1);
    return hipMalloc((void **)&p, n * sizeof(U));
  }
  hipError_t upload(const std::vector<U> &h, hipStream_t st) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpyAsync(p, h.data(), h.size() * sizeof(U), hipMemcpyHostToDevice, st);
  }
};

// Host copy of one window (caller buffers are only read inside ctvio_add_window).
struct HostWindow {
  ctvio_window w;  // scalars; pointers unused
  std::vector<double> quat, pos, bias, rho, imu_gyro, imu_acc, bc_w, v_pi, v_pj, pJ0, pr0, p_x0;
  std::vector<int64_t> imu_t, v_ti, v_tj;
  std::vector<int32_t> imu_bias, bc_i, bc_j, v_lm, v_rowi, v_rowj, p_kind, p_index, p_off;
};

template <class U> static void copy_in(std::vector<U> &dst, const U *src, size_t n) {
  dst.assign(src ? src : nullptr, src ? src + n : nullptr);
  if (!src) dst.assign(n, U(0));
}

struct SolverBase {
  virtual ~SolverBase() {}
  virtual int clear() = 0;
  virtual int add_window(const ctvio_window *w, int32_t *id) = 0;
  virtual int upload() = 0;
  virtual int num_windows() const = 0;
  virtual int solve(int max_iters, ctvio_summary *out) = 0;
  virtual int get_state(int id, double *quat, double *pos, double *bias, double *rho, double *ld) = 0;
  virtual int set_state(int id, const double *quat, const double *pos, const double *bias, const double *rho, double ld) = 0;
  virtual int linearize(int id, double *Hpp, double *W, double *Hll, double *g, double *cost) = 0;
  virtual int cost(int id, double *cost) = 0;
  virtual int lm_step(int id, double mu, double *delta, double *mc) = 0;
  virtual int spline_eval(int id, int n, const int64_t *t_ns, double *pose7, double *vel3, double *omega3, double *acc3) = 0;
  virtual int gauge_restore(int n, const int32_t *ids, const int32_t *knot, const double *q0, const double *t0) = 0;
  virtual int marginalize(int id, const int8_t *role, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0) = 0;
  virtual int bind() = 0;   // make the solver's device current on the calling thread (every ABI entry: callers use threads)
  virtual int snapshot(int restore) = 0;
  virtual int last_timing(double *ms8, int32_t *n8) = 0;
  virtual int set_profiling(int on) = 0;
  virtual void *stream() = 0;
};

static int prior_block_size(int kind) { return kind == CTVIO_PK_LD ? 1 : 3; }

template <class T> class SolverImpl : public SolverBase {
 public:
  static constexpr int VCH = 16;   // visual blocks per work item (k_assemble_vis): with the fp64 LDS accumulators 16 leaves room for 8 staging areas
  static constexpr size_t vis_stage_bytes() { return (size_t)8 * 102 * (VCH + 2) * sizeof(T) + (size_t)8 * 2 * VCH * sizeof(int); }
  explicit SolverImpl(const ctvio_options &o) : opt_(o), mixed_(sizeof(T) == 4 && o.fp64_residuals != 0) {}
  ~SolverImpl() override {
    if (stream_) (void)hipStreamDestroy(stream_);
    for (auto &e : ev_) if (e) (void)hipEventDestroy(e);
    for (auto &e : pev_) (void)hipEventDestroy(e);
  }
  int init() {
    HIPCHK(hipSetDevice(opt_.device));
    HIPCHK(hipStreamCreate(&stream_));
    for (auto &e : ev_) HIPCHK(hipEventCreate(&e));
    // kernels that need more than 64 KiB of dynamic LDS
    HIPCHK(hipFuncSetAttribute((const void *)k_cholesky_solve<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_assemble_vis<T, VCH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_assemble_vis<T, VCH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (sizeof(T) == 4) HIPCHK(hipFuncSetAttribute((const void *)k_assemble_vis_mfma<VCH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return CTVIO_OK;
  }
  int bind() override { HIPCHK(hipSetDevice(opt_.device)); return CTVIO_OK; }
  int clear() override { wins_.clear(); uploaded_ = false; return CTVIO_OK; }
  int num_windows() const override { return (int)wins_.size(); }
  void *stream() override { return (void *)stream_; }

  int add_window(const ctvio_window *w, int32_t *id) override {
    if (!w) return fail(CTVIO_ERR_INVALID, "null window");
    if (w->K < 4 || w->F < 1 || w->L < 0 || w->M < 0 || w->NB < 0 || w->V < 0 || w->dt_ns <= 0)
      return fail(CTVIO_ERR_INVALID, "bad sizes (need K >= 4, F >= 1, dt_ns > 0)");
    if (!w->quat || !w->pos || !w->bias || (w->L && !w->rho)) return fail(CTVIO_ERR_INVALID, "null state pointer");
    HostWindow h;
    h.w = *w;
    copy_in(h.quat, w->quat, (size_t)4 * w->K); copy_in(h.pos, w->pos, (size_t)3 * w->K);
    copy_in(h.bias, w->bias, (size_t)6 * w->F); copy_in(h.rho, w->rho, (size_t)w->L);
    copy_in(h.imu_t, w->imu_t, (size_t)w->M); copy_in(h.imu_gyro, w->imu_gyro, (size_t)3 * w->M);
    copy_in(h.imu_acc, w->imu_acc, (size_t)3 * w->M); copy_in(h.imu_bias, w->imu_bias, (size_t)w->M);
    copy_in(h.bc_i, w->bc_i, (size_t)w->NB); copy_in(h.bc_j, w->bc_j, (size_t)w->NB); copy_in(h.bc_w, w->bc_w, (size_t)6 * w->NB);
    copy_in(h.v_lm, w->v_lm, (size_t)w->V); copy_in(h.v_ti, w->v_ti, (size_t)w->V); copy_in(h.v_tj, w->v_tj, (size_t)w->V);
    copy_in(h.v_rowi, w->v_rowi, (size_t)w->V); copy_in(h.v_rowj, w->v_rowj, (size_t)w->V);
    copy_in(h.v_pi, w->v_pi, (size_t)2 * w->V); copy_in(h.v_pj, w->v_pj, (size_t)2 * w->V);
    copy_in(h.pJ0, w->pJ0, (size_t)w->pn * w->pn); copy_in(h.pr0, w->pr0, (size_t)w->pn);
    copy_in(h.p_kind, w->p_kind, (size_t)w->pnb); copy_in(h.p_index, w->p_index, (size_t)w->pnb);
    copy_in(h.p_off, w->p_off, (size_t)w->pnb); copy_in(h.p_x0, w->p_x0, (size_t)4 * w->pnb);
    // ---- validation (the reference asserts / prints: spline_segment.h:74-81)
    const int64_t tmax = w->t0_ns + (int64_t)(w->K - 3) * w->dt_ns;
    for (int m = 0; m < w->M; ++m) {
      if (h.imu_t[m] < w->t0_ns || h.imu_t[m] >= tmax) return fail(CTVIO_ERR_INVALID, "IMU time outside the spline");
      if (h.imu_bias[m] < 0 || h.imu_bias[m] >= w->F) return fail(CTVIO_ERR_INVALID, "IMU bias index out of range");
    }
    const int64_t ldmax_ns = (int64_t)((w->fix_ld ? w->ld : std::max(w->ld, w->ld_hi)) * 1e9);
    for (int v = 0; v < w->V; ++v) {
      if (h.v_lm[v] < 0 || h.v_lm[v] >= w->L) return fail(CTVIO_ERR_INVALID, "visual landmark index out of range");
      if (h.v_rowi[v] < 0 || h.v_rowj[v] < 0) return fail(CTVIO_ERR_INVALID, "negative image row");
      const int64_t a = h.v_ti[v], b = h.v_tj[v];
      if (a < w->t0_ns || b < w->t0_ns || a + h.v_rowi[v] * ldmax_ns >= tmax || b + h.v_rowj[v] * ldmax_ns >= tmax)
        return fail(CTVIO_ERR_INVALID, "visual time (+ row * line delay) outside the spline");
    }
    for (int b = 0; b < w->NB; ++b)
      if (h.bc_i[b] < 0 || h.bc_i[b] >= w->F || h.bc_j[b] < 0 || h.bc_j[b] >= w->F) return fail(CTVIO_ERR_INVALID, "bias chain index out of range");
    for (int b = 0; b < w->pnb; ++b) {
      const int kind = h.p_kind[b], idx = h.p_index[b];
      const int lim = (kind <= CTVIO_PK_POS) ? w->K : (kind <= CTVIO_PK_BA ? w->F : 1);
      if (kind < 0 || kind > CTVIO_PK_LD || idx < 0 || idx >= lim || h.p_off[b] < 0 || h.p_off[b] + prior_block_size(kind) > w->pn)
        return fail(CTVIO_ERR_INVALID, "prior block out of range");
    }
    if (!w->fix_ld) h.w.ld = std::min(std::max(w->ld, w->ld_lo), w->ld_hi);  // Ceres IterationZero: project on the feasible set
    wins_.push_back(std::move(h));
    if (id) *id = (int32_t)wins_.size() - 1;
    uploaded_ = false;
    return CTVIO_OK;
  }

  // ---------------------------------------------------------------------------------------- pack + upload
  int upload() override {
    const int nw = (int)wins_.size();
    if (nw == 0) return fail(CTVIO_ERR_STATE, "no windows");
    meta_.assign(nw, WinMeta());
    std::vector<double> quat, pos, bias, rho, ld, bc_w, pH, pb0, pc0(nw, 0.0), p_x0;
    std::vector<int32_t> knot_win, bias_win, lm_win, imu_grp, v_win, v_lm, v_rowi, v_rowj, bc_win, bc_i, bc_j, pcol, p_kind, p_index, p_off;
    std::vector<int64_t> v_ti, v_tj;
    std::vector<ImuGroup> groups;
    std::vector<VisItem> vitems;
    std::vector<int32_t> lm_blk_off, lm_blk;
    int maxL = 0, maxLdw = 0;
    size_t vis_lds_bytes = vis_stage_bytes(), vis_glb_bytes = vis_stage_bytes();
    std::vector<T> imu_u;
    std::vector<double> imu_ud;
    std::vector<uint8_t> active;
    int64_t H0 = 0, W0 = 0, pH0 = 0;
    int K0 = 0, F0 = 0, L0 = 0, M0 = 0, V0 = 0, B0 = 0, U0 = 0, Pp0 = 0, pv0 = 0, pb = 0;
    int maxN = 0, maxP = 0, maxPn = 0;
    Mtot_ = 0; Vtot_ = 0;
    for (const auto &h : wins_) { Mtot_ += h.w.M; Vtot_ += h.w.V; }
    std::vector<T> imu_meas((size_t)6 * std::max(Mtot_, 1)), v_obs((size_t)4 * std::max(Vtot_, 1));
    std::vector<double> imu_meas_d((size_t)6 * std::max(Mtot_, 1)), v_obs_d((size_t)4 * std::max(Vtot_, 1));
    for (int wi = 0; wi < nw; ++wi) {
      const HostWindow &h = wins_[wi];
      const ctvio_window &w = h.w;
      WinMeta &m = meta_[wi];
      m.K = w.K; m.F = w.F; m.L = w.L; m.M = w.M; m.NB = w.NB; m.V = w.V;
      m.P = 6 * w.K + 6 * w.F + 1; m.N = m.P + w.L; m.pn = w.pn; m.pnb = w.pnb;
      m.knot0 = K0; m.bias0 = F0; m.lm0 = L0; m.imu0 = M0; m.vis0 = V0; m.bc0 = B0; m.u0 = U0; m.p0 = Pp0;
      m.ldw = (m.P + 1 + 31) / 32 * 32; m.Lpad = std::max(2, (w.L + 1) / 2 * 2);
      m.pv0 = pv0; m.pblk0 = pb; m.fix_ld = w.fix_ld; m.lock_bg = w.lock_bg; m.lock_ba = w.lock_ba; m.fixed_upto = w.fixed_upto;
      m.H0 = H0; m.W0 = W0; m.pH0 = pH0; m.ldh = (m.P + 15) / 16 * 16; m.dt_ns = w.dt_ns; m.inv_dt = 1e9 / (double)w.dt_ns;
      for (int i = 0; i < 4; ++i) m.q_CI[i] = w.q_CI[i];
      for (int i = 0; i < 3; ++i) { m.p_CI[i] = w.p_CI[i]; m.gravity[i] = w.gravity[i]; }
      for (int i = 0; i < 6; ++i) m.imu_w[i] = w.imu_w[i];
      m.img_w = w.img_w; m.cauchy_a = w.cauchy_a; m.ld_lo = w.ld_lo; m.ld_hi = w.ld_hi;
      // state
      quat.insert(quat.end(), h.quat.begin(), h.quat.end()); pos.insert(pos.end(), h.pos.begin(), h.pos.end());
      bias.insert(bias.end(), h.bias.begin(), h.bias.end()); rho.insert(rho.end(), h.rho.begin(), h.rho.end());
      ld.push_back(w.ld);
      knot_win.insert(knot_win.end(), w.K, wi); bias_win.insert(bias_win.end(), w.F, wi); lm_win.insert(lm_win.end(), w.L, wi);
      // IMU: sort by (segment, bias) and cut into groups
      std::vector<int> order(w.M), seg(w.M);
      std::vector<double> uu(w.M);
      for (int i = 0; i < w.M; ++i) {
        const int64_t st = h.imu_t[i] - w.t0_ns;
        seg[i] = (int)(st / w.dt_ns);
        uu[i] = (double)(st % w.dt_ns) / (double)w.dt_ns;
        order[i] = i;
      }
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (seg[a] != seg[b]) return seg[a] < seg[b];
        return h.imu_bias[a] < h.imu_bias[b];
      });
      m.grp0 = (int)groups.size();
      for (int i = 0; i < w.M; ++i) {
        const int src = order[i];
        if (i == 0 || seg[src] != seg[order[i - 1]] || h.imu_bias[src] != h.imu_bias[order[i - 1]])
          groups.push_back(ImuGroup{wi, seg[src], h.imu_bias[src], i, 0});
        groups.back().count++;
        imu_grp.push_back((int)groups.size() - 1);
        imu_u.push_back((T)uu[src]);
        imu_ud.push_back(uu[src]);
        for (int c = 0; c < 3; ++c) {
          imu_meas[(size_t)c * Mtot_ + M0 + i] = (T)h.imu_gyro[3 * src + c];
          imu_meas[(size_t)(3 + c) * Mtot_ + M0 + i] = (T)h.imu_acc[3 * src + c];
          imu_meas_d[(size_t)c * Mtot_ + M0 + i] = h.imu_gyro[3 * src + c];
          imu_meas_d[(size_t)(3 + c) * Mtot_ + M0 + i] = h.imu_acc[3 * src + c];
        }
      }
      m.ngrp = (int)groups.size() - m.grp0;
      // visual: sort by frame pair (then rows) so that consecutive blocks hit the same knot quadruples, cut into items
      std::vector<int> vord(w.V);
      std::iota(vord.begin(), vord.end(), 0);
      std::stable_sort(vord.begin(), vord.end(), [&](int a, int b) {
        if (h.v_ti[a] != h.v_ti[b]) return h.v_ti[a] < h.v_ti[b];
        if (h.v_tj[a] != h.v_tj[b]) return h.v_tj[a] < h.v_tj[b];
        if (h.v_rowi[a] != h.v_rowi[b]) return h.v_rowi[a] < h.v_rowi[b];
        return h.v_rowj[a] < h.v_rowj[b];
      });
      m.vitem0 = (int)vitems.size();
      for (int i = 0; i < w.V; ++i) {
        const int v = vord[i];
        const bool fresh = (i == 0) || h.v_ti[v] != h.v_ti[vord[i - 1]] || h.v_tj[v] != h.v_tj[vord[i - 1]] || vitems.back().count >= VCH;
        if (fresh) vitems.push_back(VisItem{V0 + i, 0});
        vitems.back().count++;
        v_win.push_back(wi); v_lm.push_back(h.v_lm[v]);
        v_ti.push_back(h.v_ti[v] - w.t0_ns); v_tj.push_back(h.v_tj[v] - w.t0_ns);
        v_rowi.push_back(h.v_rowi[v]); v_rowj.push_back(h.v_rowj[v]);
        v_obs[(size_t)0 * Vtot_ + V0 + i] = (T)h.v_pi[2 * v]; v_obs[(size_t)1 * Vtot_ + V0 + i] = (T)h.v_pi[2 * v + 1];
        v_obs[(size_t)2 * Vtot_ + V0 + i] = (T)h.v_pj[2 * v]; v_obs[(size_t)3 * Vtot_ + V0 + i] = (T)h.v_pj[2 * v + 1];
        v_obs_d[(size_t)0 * Vtot_ + V0 + i] = h.v_pi[2 * v]; v_obs_d[(size_t)1 * Vtot_ + V0 + i] = h.v_pi[2 * v + 1];
        v_obs_d[(size_t)2 * Vtot_ + V0 + i] = h.v_pj[2 * v]; v_obs_d[(size_t)3 * Vtot_ + V0 + i] = h.v_pj[2 * v + 1];
      }
      m.nvitem = (int)vitems.size() - m.vitem0;
      {  // CSR landmark -> blocks (positions in the sorted order)
        std::vector<std::vector<int>> per(w.L);
        for (int i = 0; i < w.V; ++i) per[h.v_lm[vord[i]]].push_back(V0 + i);
        for (int l = 0; l < w.L; ++l) {
          lm_blk_off.push_back((int)lm_blk.size());
          lm_blk.insert(lm_blk.end(), per[l].begin(), per[l].end());
        }
      }
      {
        const size_t K6 = 6 * (size_t)w.K, nG = K6 + 1, nH = K6 * (K6 + 1) / 2 + K6 + 1 + nG;
        const size_t need = ((nH + 3) & ~(size_t)3) * sizeof(double) + 32 + vis_stage_bytes();   // fp64 accumulators in LDS
        const size_t need_glb = ((nG + 3) & ~(size_t)3) * sizeof(double) + 16 + vis_stage_bytes();
        m.vis_lds = need <= 160 * 1024 ? 1 : 0;
        vis_lds_bytes = std::max(vis_lds_bytes, m.vis_lds ? need : need_glb);
        vis_glb_bytes = std::max(vis_glb_bytes, need_glb);
      }
      // bias chain
      for (int b = 0; b < w.NB; ++b) { bc_win.push_back(wi); bc_i.push_back(h.bc_i[b]); bc_j.push_back(h.bc_j[b]); }
      bc_w.insert(bc_w.end(), h.bc_w.begin(), h.bc_w.end());
      // prior: J0^T J0 (row-major n*n), J0^T r0, r0^T r0 in fp64; J0 is column-major (Eigen)
      const int n = w.pn;
      if (n > 0) {
        std::vector<int32_t> col(n, -1);
        for (int b = 0; b < w.pnb; ++b) {
          const int kind = h.p_kind[b], idx = h.p_index[b];
          int u0 = 0;
          switch (kind) {
            case CTVIO_PK_ROT: u0 = 6 * idx; break;
            case CTVIO_PK_POS: u0 = 6 * idx + 3; break;
            case CTVIO_PK_BG: u0 = 6 * w.K + 6 * idx; break;
            case CTVIO_PK_BA: u0 = 6 * w.K + 6 * idx + 3; break;
            default: u0 = m.P - 1;
          }
          for (int k = 0; k < prior_block_size(kind); ++k) col[h.p_off[b] + k] = u0 + k;
        }
        pcol.insert(pcol.end(), col.begin(), col.end());
        for (int i = 0; i < n; ++i) {
          double bi = 0;
          for (int r = 0; r < n; ++r) bi += h.pJ0[(size_t)i * n + r] * h.pr0[r];
          pb0.push_back(bi);
          for (int j = 0; j < n; ++j) {
            double s = 0;
            for (int r = 0; r < n; ++r) s += h.pJ0[(size_t)i * n + r] * h.pJ0[(size_t)j * n + r];
            pH.push_back(s);
          }
        }
        double c0 = 0;
        for (int r = 0; r < n; ++r) c0 += h.pr0[r] * h.pr0[r];
        pc0[wi] = c0;
      }
      p_kind.insert(p_kind.end(), h.p_kind.begin(), h.p_kind.end()); p_index.insert(p_index.end(), h.p_index.begin(), h.p_index.end());
      p_off.insert(p_off.end(), h.p_off.begin(), h.p_off.end()); p_x0.insert(p_x0.end(), h.p_x0.begin(), h.p_x0.end());
      // reduced program: referenced and not constant (trajectory_estimator.cpp:114-141, 236-245, 311-318)
      std::vector<uint8_t> act(m.N, 0);
      for (int i = 0; i < w.M; ++i) {
        for (int c = 0; c < 24; ++c) act[6 * seg[i] + c] = 1;
        for (int c = 0; c < 6; ++c) act[6 * w.K + 6 * h.imu_bias[i] + c] = 1;
      }
      const int64_t pad_ns = (int64_t)(0.039 * 1e9);  // AddImageFeatureDelayAnalytic spans [t, t + 0.039 s] (trajectory_estimator.cpp:299)
      for (int v = 0; v < w.V; ++v) {
        const int64_t tt[2] = {h.v_ti[v], h.v_tj[v]};
        for (int e = 0; e < 2; ++e) {
          const int s0 = (int)((tt[e] - w.t0_ns) / w.dt_ns), s1 = (int)((tt[e] + pad_ns - w.t0_ns) / w.dt_ns);
          for (int k = s0; k < s1 + 4 && k < w.K; ++k)
            for (int c = 0; c < 6; ++c) act[6 * k + c] = 1;
        }
        act[m.P + h.v_lm[v]] = 1;
        act[m.P - 1] = 1;
      }
      for (int b = 0; b < w.NB; ++b)
        for (int c = 0; c < 6; ++c) { act[6 * w.K + 6 * h.bc_i[b] + c] = 1; act[6 * w.K + 6 * h.bc_j[b] + c] = 1; }
      for (int i = 0; i < n; ++i) act[pcol[pv0 + i]] = 1;
      for (int k = 0; k <= w.fixed_upto && k < w.K; ++k)
        for (int c = 0; c < 6; ++c) act[6 * k + c] = 0;
      for (int f = 0; f < w.F; ++f)
        for (int c = 0; c < 3; ++c) {
          if (w.lock_bg) act[6 * w.K + 6 * f + c] = 0;
          if (w.lock_ba) act[6 * w.K + 6 * f + 3 + c] = 0;
        }
      if (w.fix_ld) act[m.P - 1] = 0;
      active.insert(active.end(), act.begin(), act.end());
      // advance offsets
      K0 += w.K; F0 += w.F; L0 += w.L; M0 += w.M; V0 += w.V; B0 += w.NB; U0 += m.N; Pp0 += m.P; pv0 += n; pb += w.pnb;
      H0 += (int64_t)m.P * m.ldh; W0 += (int64_t)m.Lpad * m.ldw; pH0 += (int64_t)n * n;
      maxN = std::max(maxN, m.N); maxP = std::max(maxP, m.P); maxPn = std::max(maxPn, n);
      maxL = std::max(maxL, m.L); maxLdw = std::max(maxLdw, m.ldw);
    }
    const size_t chol_lds = (size_t)(2 * 32 * 34 + 32 + 34 + (size_t)((std::max(maxP - 32, 0) + 1 + 15) / 16 * 16) * 32) * sizeof(double);
    if (chol_lds > 160 * 1024) return fail(CTVIO_ERR_INVALID, "window too large for the single-workgroup Cholesky (P > ~600)");
    // ---- device buffers
    Dev<T> &d = dev_;
    std::memset(&d, 0, sizeof d);
    d.nwin = nw; d.Ktot = K0; d.Ftot = F0; d.Ltot = L0; d.Mtot = Mtot_; d.Gtot = (int)groups.size(); d.Vtot = Vtot_;
    d.NBtot = B0; d.Utot = U0; d.maxN = maxN; d.maxP = maxP; d.maxPn = maxPn;
    HIPCHK(b_meta_.upload(meta_, stream_)); d.wins = b_meta_.p;
    HIPCHK(b_quat_.upload(quat, stream_)); HIPCHK(b_pos_.upload(pos, stream_)); HIPCHK(b_bias_.upload(bias, stream_));
    HIPCHK(b_rho_.upload(rho, stream_)); HIPCHK(b_ld_.upload(ld, stream_));
    HIPCHK(b_cquat_.alloc(quat.size())); HIPCHK(b_cpos_.alloc(pos.size())); HIPCHK(b_cbias_.alloc(bias.size()));
    HIPCHK(b_crho_.alloc(rho.size())); HIPCHK(b_cld_.alloc(ld.size()));
    d.quat = b_quat_.p; d.pos = b_pos_.p; d.bias = b_bias_.p; d.rho = b_rho_.p; d.ld = b_ld_.p;
    HIPCHK(b_kd_.alloc(3 * quat.size() / 4)); HIPCHK(b_ckd_.alloc(3 * quat.size() / 4)); HIPCHK(b_kjri_.alloc(9 * quat.size() / 4));
    d.kd = b_kd_.p; d.ckd = b_ckd_.p; d.kjri = b_kjri_.p;
    d.cquat = b_cquat_.p; d.cpos = b_cpos_.p; d.cbias = b_cbias_.p; d.crho = b_crho_.p; d.cld = b_cld_.p;
    HIPCHK(b_knot_win_.upload(knot_win, stream_)); HIPCHK(b_bias_win_.upload(bias_win, stream_)); HIPCHK(b_lm_win_.upload(lm_win, stream_));
    d.knot_win = b_knot_win_.p; d.bias_win = b_bias_win_.p; d.lm_win = b_lm_win_.p;
    HIPCHK(b_groups_.upload(groups, stream_)); HIPCHK(b_imu_grp_.upload(imu_grp, stream_)); HIPCHK(b_imu_u_.upload(imu_u, stream_));
    HIPCHK(b_imu_meas_.upload(imu_meas, stream_)); HIPCHK(b_tiles_.alloc(groups.size() * 1024));
    d.groups = b_groups_.p; d.imu_grp = b_imu_grp_.p; d.imu_u = b_imu_u_.p; d.imu_meas = b_imu_meas_.p; d.imu_tiles = b_tiles_.p;
    HIPCHK(b_imu_ud_.upload(imu_ud, stream_)); HIPCHK(b_imu_meas_d_.upload(imu_meas_d, stream_)); HIPCHK(b_v_obs_d_.upload(v_obs_d, stream_));
    d.imu_ud = b_imu_ud_.p; d.imu_meas_d = b_imu_meas_d_.p; d.v_obs_d = b_v_obs_d_.p;
    HIPCHK(b_v_win_.upload(v_win, stream_)); HIPCHK(b_v_lm_.upload(v_lm, stream_)); HIPCHK(b_v_ti_.upload(v_ti, stream_));
    HIPCHK(b_v_tj_.upload(v_tj, stream_)); HIPCHK(b_v_rowi_.upload(v_rowi, stream_)); HIPCHK(b_v_rowj_.upload(v_rowj, stream_));
    HIPCHK(b_v_obs_.upload(v_obs, stream_));
    if (mixed_) { HIPCHK(b_imu_rc_.alloc((size_t)6 * std::max(Mtot_, 1))); HIPCHK(b_vis_rc_.alloc((size_t)3 * std::max(Vtot_, 1))); d.imu_rc = b_imu_rc_.p; d.vis_rc = b_vis_rc_.p; }
    HIPCHK(b_Jv_.alloc((size_t)100 * std::max(Vtot_, 1))); HIPCHK(b_rv_.alloc((size_t)2 * std::max(Vtot_, 1))); HIPCHK(b_vs_.alloc((size_t)2 * std::max(Vtot_, 1)));
    d.v_win = b_v_win_.p; d.v_lm = b_v_lm_.p; d.v_ti = b_v_ti_.p; d.v_tj = b_v_tj_.p; d.v_rowi = b_v_rowi_.p; d.v_rowj = b_v_rowj_.p;
    d.v_obs = b_v_obs_.p; d.Jv = b_Jv_.p; d.rv = b_rv_.p; d.vs = b_vs_.p;
    HIPCHK(b_Wc_.alloc((size_t)WC_STRIDE * std::max(Vtot_, 1))); d.Wc = b_Wc_.p;
    {  // row of every block in landmark order = its position in the CSR list
      std::vector<int32_t> v_slot(std::max(Vtot_, 1), 0);
      for (size_t p = 0; p < lm_blk.size(); ++p) v_slot[lm_blk[p]] = (int32_t)p;
      HIPCHK(b_v_slot_.upload(v_slot, stream_)); d.v_slot = b_v_slot_.p;
    }
    HIPCHK(b_vitems_.upload(vitems, stream_)); d.vitems = b_vitems_.p;
    lm_blk_off.push_back((int)lm_blk.size());
    HIPCHK(b_lm_blk_off_.upload(lm_blk_off, stream_)); HIPCHK(b_lm_blk_.upload(lm_blk, stream_));
    d.lm_blk_off = b_lm_blk_off_.p; d.lm_blk = b_lm_blk_.p; d.maxL = maxL; d.maxLdw = maxLdw;
    HIPCHK(b_bc_win_.upload(bc_win, stream_)); HIPCHK(b_bc_i_.upload(bc_i, stream_)); HIPCHK(b_bc_j_.upload(bc_j, stream_)); HIPCHK(b_bc_w_.upload(bc_w, stream_));
    d.bc_win = b_bc_win_.p; d.bc_i = b_bc_i_.p; d.bc_j = b_bc_j_.p; d.bc_w = b_bc_w_.p;
    HIPCHK(b_pH_.upload(pH, stream_)); HIPCHK(b_pb0_.upload(pb0, stream_)); HIPCHK(b_pc0_.upload(pc0, stream_)); HIPCHK(b_pcol_.upload(pcol, stream_));
    HIPCHK(b_p_kind_.upload(p_kind, stream_)); HIPCHK(b_p_index_.upload(p_index, stream_)); HIPCHK(b_p_off_.upload(p_off, stream_)); HIPCHK(b_p_x0_.upload(p_x0, stream_));
    d.pH = b_pH_.p; d.pb0 = b_pb0_.p; d.pc0 = b_pc0_.p; d.pcol = b_pcol_.p; d.p_kind = b_p_kind_.p; d.p_index = b_p_index_.p; d.p_off = b_p_off_.p; d.p_x0 = b_p_x0_.p;
    HIPCHK(b_Hpp_.alloc((size_t)H0)); HIPCHK(b_S_.alloc((size_t)H0)); HIPCHK(b_W_.alloc((size_t)W0)); HIPCHK(b_Hll_.alloc((size_t)L0));
    HIPCHK(b_g_.alloc((size_t)U0)); HIPCHK(b_rhs_.alloc((size_t)Pp0)); HIPCHK(b_dd_.alloc((size_t)U0)); HIPCHK(b_dinv_.alloc((size_t)L0));
    HIPCHK(b_cscale_.alloc((size_t)U0)); HIPCHK(b_delta_.alloc((size_t)U0)); HIPCHK(b_active_.upload(active, stream_));
    HIPCHK(b_lm_.alloc((size_t)nw)); HIPCHK(b_nact_.alloc(1)); HIPCHK(b_dbg_.alloc(64));
    d.dbg = std::getenv("CTVIO_DEBUG_STAMPS") ? b_dbg_.p : nullptr;
    d.chol_nblk = (maxP + 31) / 32; HIPCHK(b_chol_inv_.alloc((size_t)nw * d.chol_nblk * 1024)); d.chol_inv = b_chol_inv_.p;
    d.Hpp = b_Hpp_.p; d.S = b_S_.p; d.W = b_W_.p; d.Hll = b_Hll_.p; d.g = b_g_.p; d.rhs = b_rhs_.p; d.dd = b_dd_.p; d.dinv = b_dinv_.p;
    d.cscale = b_cscale_.p; d.delta = b_delta_.p; d.active = b_active_.p; d.lm = b_lm_.p; d.n_active = b_nact_.p;
    HIPCHK(hipMemsetAsync(b_lm_.p, 0, sizeof(Lm) * nw, stream_));
    HIPCHK(hipMemsetAsync(b_W_.p, 0, sizeof(T) * std::max<size_t>((size_t)W0, 1), stream_));
    HIPCHK(hipMemsetAsync(b_Hll_.p, 0, sizeof(double) * std::max(L0, 1), stream_));
    HIPCHK(hipMemsetAsync(b_g_.p, 0, sizeof(double) * std::max(U0, 1), stream_));
    HIPCHK(hipMemsetAsync(b_delta_.p, 0, sizeof(double) * std::max(U0, 1), stream_));
    HIPCHK(hipMemsetAsync(b_cscale_.p, 0, sizeof(double) * std::max(U0, 1), stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    chol_lds_ = chol_lds;
    snap_valid_ = false;
    any_vis_lds_ = any_vis_glb_ = false;
    for (const auto &mm : meta_) { if (mm.vis_lds) any_vis_lds_ = true; else if (mm.V > 0) any_vis_glb_ = true; }
    vis_lds_ = vis_lds_bytes;
    vis_glb_ = vis_glb_bytes;
    uploaded_ = true;
    return CTVIO_OK;
  }

  // ---------------------------------------------------------------------------------------- launches
  static int nblk(long long n, int b) { return (int)std::max<long long>((n + b - 1) / b, 1); }
  int vis_parts() const { return std::min(8, std::max(1, 256 / std::max(dev_.nwin, 1))); }
  void set_params(int max_iters) {
    LmParams &p = dev_.prm;
    p.ftol = opt_.function_tolerance; p.gtol = opt_.gradient_tolerance; p.ptol = opt_.parameter_tolerance;
    p.max_radius = opt_.max_radius; p.min_radius = opt_.min_radius; p.min_rel_dec = opt_.min_relative_decrease;
    p.min_diag = opt_.min_lm_diagonal; p.max_diag = opt_.max_lm_diagonal; p.max_invalid = opt_.max_consecutive_invalid_steps;
    p.max_iters = max_iters;
  }
  // Per-phase timing with HIP events on the solver's stream (only when profiling is switched on).
  enum { PH_IMU_LIN = 0, PH_VIS_LIN, PH_ASM_VIS, PH_ASM_REST, PH_SCHUR, PH_CHOL, PH_REST, PH_COUNT };
  void ph_begin(int ph) {
    if (!profiling_) return;
    if (pev_used_ + 2 > pev_.size()) {
      for (int i = 0; i < 64; ++i) { hipEvent_t e; (void)hipEventCreate(&e); pev_.push_back(e); }
    }
    (void)hipEventRecord(pev_[pev_used_], stream_);
    pev_phase_.push_back(ph);
    pev_used_ += 2;
  }
  void ph_end() {
    if (!profiling_) return;
    (void)hipEventRecord(pev_[pev_used_ - 1], stream_);
  }
  void ph_collect() {
    std::fill(ph_ms_, ph_ms_ + 8, 0.0);
    std::fill(ph_n_, ph_n_ + 8, 0);
    for (size_t i = 0; i < pev_phase_.size(); ++i) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, pev_[2 * i], pev_[2 * i + 1]) == hipSuccess) { ph_ms_[pev_phase_[i]] += ms; ph_n_[pev_phase_[i]] += 1; }
    }
    pev_phase_.clear();
    pev_used_ = 0;
  }
  void launch_linearize() {
    const Dev<T> &d = dev_;
    const int nw = d.nwin;
    constexpr int CH = sizeof(T) == 4 ? 64 : 32;
    ph_begin(PH_ASM_REST);
    hipLaunchKernelGGL((k_zero_normal<T>), dim3(64, nw), dim3(256), 0, stream_, d, vis_parts() == 1 ? 1 : 0);
    ph_end();
    hipLaunchKernelGGL((k_knot_prep<T>), dim3(nblk(d.Ktot, 256)), dim3(256), 0, stream_, d, d.quat, d.kd, d.kjri);
    ph_begin(PH_IMU_LIN);
    const size_t imu_lds = (sizeof(T) == 4 ? (size_t)3 * CH * 33 * sizeof(T) : (size_t)32 * (6 * CH + 4) * sizeof(T)) + 0;
    if (d.Gtot) {
      if (mixed_) hipLaunchKernelGGL((k_imu_linearize<T, CH, double>), dim3(d.Gtot), dim3(64), imu_lds, stream_, d);
      else hipLaunchKernelGGL((k_imu_linearize<T, CH, T>), dim3(d.Gtot), dim3(64), imu_lds, stream_, d);
    }
    ph_end();
    ph_begin(PH_VIS_LIN);
    if (d.Vtot) {
      if (mixed_) hipLaunchKernelGGL((k_vis_eval<T, true, double>), dim3(nblk(d.Vtot, 64)), dim3(64), 0, stream_, d, d.quat, d.pos, d.rho, d.ld, d.kd, 0);
      else hipLaunchKernelGGL((k_vis_eval<T, true, T>), dim3(nblk(d.Vtot, 64)), dim3(64), 0, stream_, d, d.quat, d.pos, d.rho, d.ld, d.kd, 0);
    }
    ph_end();
  }
  void launch_assemble() {
    const Dev<T> &d = dev_;
    const int nw = d.nwin;
    ph_begin(PH_ASM_VIS);
    {  // few windows: split each window's items over several workgroups to fill the chip
      const int parts = vis_parts();
      if (any_vis_lds_) launch_assemble_vis_lds(parts);
      if (any_vis_glb_) hipLaunchKernelGGL((k_assemble_vis<T, VCH, false>), dim3(nw, parts), dim3(512), vis_glb_, stream_, d);
    }
    ph_end();
    ph_begin(PH_ASM_REST);
    if (d.maxL) hipLaunchKernelGGL((k_build_W<T>), dim3(d.maxL, nw), dim3(64), (size_t)d.maxLdw * sizeof(double), stream_, d);
    if (d.Gtot) hipLaunchKernelGGL((k_assemble_imu<T>), dim3(d.Gtot), dim3(256), 0, stream_, d);
    hipLaunchKernelGGL((k_misc<T, true>), dim3(nw), dim3(256), std::max(d.maxPn, 1) * sizeof(double), stream_, d, d.quat, d.pos, d.bias, d.ld, 0);
    hipLaunchKernelGGL((k_post_linearize<T>), dim3(nblk(d.maxN, 256), nw), dim3(256), 0, stream_, d);
    ph_end();
  }
  void launch_step() {
    const Dev<T> &d = dev_;
    const int nw = d.nwin;
    ph_begin(PH_REST);
    hipLaunchKernelGGL((k_damping<T>), dim3(nblk(d.maxN, 256), nw), dim3(256), 0, stream_, d);
    ph_end();
    ph_begin(PH_SCHUR);
    launch_schur();
    ph_end();
    if (!schur_makes_rhs()) {
      ph_begin(PH_REST);
      hipLaunchKernelGGL((k_rhs<T>), dim3(nblk(d.maxP, 64), nw), dim3(256), 0, stream_, d);
      ph_end();
    }
    ph_begin(PH_CHOL);
    hipLaunchKernelGGL((k_cholesky_solve<T>), dim3(nw), dim3(256), chol_lds_, stream_, d);
    ph_end();
    ph_begin(PH_REST);
    hipLaunchKernelGGL((k_backsub<T>), dim3(nw), dim3(256), (size_t)d.maxP * sizeof(double), stream_, d);
    ph_end();
  }
  void launch_schur();
  void launch_assemble_vis_lds(int parts);
  bool schur_makes_rhs() const { return sizeof(T) == 4 && opt_.use_mfma != 0; }
  void launch_cost(bool candidate, int force) {
    const Dev<T> &d = dev_;
    const double *q = candidate ? d.cquat : d.quat, *p = candidate ? d.cpos : d.pos, *b = candidate ? d.cbias : d.bias;
    const double *r = candidate ? d.crho : d.rho, *l = candidate ? d.cld : d.ld;
    double *kd = candidate ? d.ckd : d.kd;
    hipLaunchKernelGGL((k_knot_prep<T>), dim3(nblk(d.Ktot, 256)), dim3(256), 0, stream_, d, q, kd, (T *)nullptr);
    if (d.Mtot) {
      if (mixed_) hipLaunchKernelGGL((k_imu_cost<T, double>), dim3(nblk(d.Mtot, 256)), dim3(256), 0, stream_, d, q, p, b, kd, force);
      else hipLaunchKernelGGL((k_imu_cost<T, T>), dim3(nblk(d.Mtot, 256)), dim3(256), 0, stream_, d, q, p, b, kd, force);
    }
    if (d.Vtot) {
      if (mixed_) hipLaunchKernelGGL((k_vis_eval<T, false, double>), dim3(nblk(d.Vtot, 64)), dim3(64), 0, stream_, d, q, p, r, l, kd, force);
      else hipLaunchKernelGGL((k_vis_eval<T, false, T>), dim3(nblk(d.Vtot, 64)), dim3(64), 0, stream_, d, q, p, r, l, kd, force);
    }
    hipLaunchKernelGGL((k_misc<T, false>), dim3(d.nwin), dim3(256), std::max(d.maxPn, 1) * sizeof(double), stream_, d, q, p, b, l, force);
  }
  int n_state() const { return dev_.Ktot + dev_.Ftot + dev_.Ltot + dev_.nwin; }

  int solve(int max_iters, ctvio_summary *out) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (max_iters < 0) return fail(CTVIO_ERR_INVALID, "max_iterations < 0");
    Dev<T> &d = dev_;
    const int nw = d.nwin, wb = nblk(nw, 64);
    set_params(max_iters);
    profiling_ = profiling_requested_;
    pev_phase_.clear(); pev_used_ = 0;
    HIPCHK(hipEventRecord(ev_[8], stream_));
    hipLaunchKernelGGL((k_lm_init<T>), dim3(wb), dim3(64), 0, stream_, d, opt_.initial_radius, 0);
    launch_cost(false, 1);
    hipLaunchKernelGGL((k_set_initial_cost<T>), dim3(wb), dim3(64), 0, stream_, d);
    int it = 0;
    const int check = std::max(1, opt_.check_every);
    for (; it <= max_iters; ++it) {
      launch_linearize();
      launch_assemble();
      HIPCHK(hipMemsetAsync(d.n_active, 0, sizeof(int32_t), stream_));
      hipLaunchKernelGGL((k_begin_iter<T>), dim3(wb), dim3(64), 0, stream_, d);
      if (it == max_iters) break;  // the last pass only finalises (max-iterations termination)
      launch_step();
      ph_begin(PH_REST);
      hipLaunchKernelGGL((k_update<T, false>), dim3(nblk(n_state(), 256)), dim3(256), 0, stream_, d);
      launch_cost(true, 0);
      hipLaunchKernelGGL((k_lm_control<T>), dim3(wb), dim3(64), 0, stream_, d);
      hipLaunchKernelGGL((k_update<T, true>), dim3(nblk(n_state(), 256)), dim3(256), 0, stream_, d);
      ph_end();
      if ((it + 1) % check == 0 && it + 1 < max_iters) {
        int32_t na = 0;
        HIPCHK(hipMemcpyAsync(&na, d.n_active, sizeof na, hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        if (na == 0) { ++it; break; }
      }
    }
    HIPCHK(hipEventRecord(ev_[9], stream_));
    std::vector<Lm> lm(nw);
    HIPCHK(hipMemcpyAsync(lm.data(), d.lm, sizeof(Lm) * nw, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev_[8], ev_[9]));
    if (profiling_) ph_collect(); else { std::fill(ph_ms_, ph_ms_ + 8, 0.0); std::fill(ph_n_, ph_n_ + 8, 0); }
    profiling_ = false;
    std::copy(ph_ms_, ph_ms_ + 7, timing_);
    timing_[7] = ms;
    last_iters_ = it;
    if (d.dbg) {
      long long st[64];
      HIPCHK(hipMemcpy(st, d.dbg, sizeof st, hipMemcpyDeviceToHost));
      std::fprintf(stderr, "[ctvio] cholesky clock64 deltas:");
      for (int i = 1; i < 24; ++i) std::fprintf(stderr, " %lld", st[i] - st[i - 1]);
      std::fprintf(stderr, "\n[ctvio] imu_linearize clock64 deltas:");
      for (int i = 33; i < 48; ++i) std::fprintf(stderr, " %lld", st[i] - st[i - 1]);
      std::fprintf(stderr, "\n[ctvio] assemble_vis clock64 deltas (zero | rounds | imu tiles | H flush | g flush):");
      for (int i = 49; i < 63; ++i) std::fprintf(stderr, " %lld", st[i] - st[i - 1]);
      std::fprintf(stderr, "\n");
    }
    if (out)
      for (int w = 0; w < nw; ++w) {
        out[w].iterations = lm[w].iter; out[w].num_successful = lm[w].nsucc; out[w].num_unsuccessful = lm[w].nunsucc;
        out[w].termination = lm[w].status > 0 ? lm[w].status - 1 : 0;
        out[w].initial_cost = lm[w].initial_cost; out[w].final_cost = lm[w].cost; out[w].final_radius = lm[w].mu;
      }
    return CTVIO_OK;
  }

  int get_state(int id, double *quat, double *pos, double *bias, double *rho, double *ld) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
    const WinMeta &m = meta_[id];
    if (quat) HIPCHK(hipMemcpyAsync(quat, dev_.quat + 4 * (size_t)m.knot0, sizeof(double) * 4 * m.K, hipMemcpyDeviceToHost, stream_));
    if (pos) HIPCHK(hipMemcpyAsync(pos, dev_.pos + 3 * (size_t)m.knot0, sizeof(double) * 3 * m.K, hipMemcpyDeviceToHost, stream_));
    if (bias) HIPCHK(hipMemcpyAsync(bias, dev_.bias + 6 * (size_t)m.bias0, sizeof(double) * 6 * m.F, hipMemcpyDeviceToHost, stream_));
    if (rho && m.L) HIPCHK(hipMemcpyAsync(rho, dev_.rho + m.lm0, sizeof(double) * m.L, hipMemcpyDeviceToHost, stream_));
    if (ld) HIPCHK(hipMemcpyAsync(ld, dev_.ld + id, sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    return CTVIO_OK;
  }
  int set_state(int id, const double *quat, const double *pos, const double *bias, const double *rho, double ld) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
    const WinMeta &m = meta_[id];
    if (!m.fix_ld) ld = std::min(std::max(ld, m.ld_lo), m.ld_hi);
    if (quat) HIPCHK(hipMemcpyAsync(dev_.quat + 4 * (size_t)m.knot0, quat, sizeof(double) * 4 * m.K, hipMemcpyHostToDevice, stream_));
    if (pos) HIPCHK(hipMemcpyAsync(dev_.pos + 3 * (size_t)m.knot0, pos, sizeof(double) * 3 * m.K, hipMemcpyHostToDevice, stream_));
    if (bias) HIPCHK(hipMemcpyAsync(dev_.bias + 6 * (size_t)m.bias0, bias, sizeof(double) * 6 * m.F, hipMemcpyHostToDevice, stream_));
    if (rho && m.L) HIPCHK(hipMemcpyAsync(dev_.rho + m.lm0, rho, sizeof(double) * m.L, hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemcpyAsync(dev_.ld + id, &ld, sizeof(double), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    return CTVIO_OK;
  }

  // device-side copy of the whole batch state (restore != 0: copy back)
  int snapshot(int restore) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    Dev<T> &d = dev_;
    HIPCHK(b_snap_.alloc((size_t)4 * d.Ktot + 3 * d.Ktot + 6 * d.Ftot + d.Ltot + d.nwin));
    double *p = b_snap_.p;
    double *parts[5] = {d.quat, d.pos, d.bias, d.rho, d.ld};
    const size_t sz[5] = {(size_t)4 * d.Ktot, (size_t)3 * d.Ktot, (size_t)6 * d.Ftot, (size_t)d.Ltot, (size_t)d.nwin};
    if (restore && !snap_valid_) return fail(CTVIO_ERR_STATE, "no snapshot taken");
    for (int i = 0; i < 5; ++i) {
      if (sz[i]) HIPCHK(hipMemcpyAsync(restore ? parts[i] : p, restore ? p : parts[i], sz[i] * sizeof(double), hipMemcpyDeviceToDevice, stream_));
      p += sz[i];
    }
    HIPCHK(hipStreamSynchronize(stream_));
    snap_valid_ = true;
    return CTVIO_OK;
  }

  int linearize(int id, double *Hpp, double *W, double *Hll, double *g, double *cost) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
    Dev<T> &d = dev_;
    const int nw = d.nwin, wb = nblk(nw, 64);
    set_params(1);
    hipLaunchKernelGGL((k_lm_init<T>), dim3(wb), dim3(64), 0, stream_, d, opt_.initial_radius, 0);
    launch_cost(false, 1);
    hipLaunchKernelGGL((k_set_initial_cost<T>), dim3(wb), dim3(64), 0, stream_, d);
    launch_linearize();
    launch_assemble();
    const WinMeta &m = meta_[id];
    const int P = m.P;
    if (Hpp) {
      HIPCHK(hipMemcpy2DAsync(Hpp, sizeof(double) * (size_t)P, d.Hpp + m.H0, sizeof(double) * (size_t)m.ldh, sizeof(double) * (size_t)P, (size_t)P,
                              hipMemcpyDeviceToHost, stream_));
    }
    std::vector<T> Wh;
    if (W && m.L) {
      Wh.resize((size_t)m.Lpad * m.ldw);
      HIPCHK(hipMemcpyAsync(Wh.data(), d.W + m.W0, sizeof(T) * Wh.size(), hipMemcpyDeviceToHost, stream_));
    }
    if (Hll && m.L) HIPCHK(hipMemcpyAsync(Hll, d.Hll + m.lm0, sizeof(double) * m.L, hipMemcpyDeviceToHost, stream_));
    if (g) HIPCHK(hipMemcpyAsync(g, d.g + m.u0, sizeof(double) * m.N, hipMemcpyDeviceToHost, stream_));
    Lm lm;
    HIPCHK(hipMemcpyAsync(&lm, d.lm + id, sizeof(Lm), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (Hpp)
      for (int i = 0; i < P; ++i)
        for (int j = i + 1; j < P; ++j) Hpp[(size_t)i * P + j] = Hpp[(size_t)j * P + i];
    if (W && m.L)
      for (int i = 0; i < P; ++i)
        for (int l = 0; l < m.L; ++l) W[(size_t)i * m.L + l] = (double)Wh[(size_t)l * m.ldw + i];
    if (cost) *cost = lm.cost;
    return CTVIO_OK;
  }
  int cost(int id, double *cost) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
    Dev<T> &d = dev_;
    const int wb = nblk(d.nwin, 64);
    set_params(1);
    hipLaunchKernelGGL((k_lm_init<T>), dim3(wb), dim3(64), 0, stream_, d, opt_.initial_radius, 1);
    launch_cost(false, 1);
    Lm lm;
    HIPCHK(hipMemcpyAsync(&lm, d.lm + id, sizeof(Lm), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (cost) *cost = lm.cand_cost;
    return CTVIO_OK;
  }
  int lm_step(int id, double mu, double *delta, double *mc) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
    Dev<T> &d = dev_;
    const int wb = nblk(d.nwin, 64);
    set_params(1);
    hipLaunchKernelGGL((k_lm_init<T>), dim3(wb), dim3(64), 0, stream_, d, mu, 0);
    launch_cost(false, 1);   // mixed mode: the linearisation reuses the residuals of a cost pass at the same state
    launch_linearize();
    launch_assemble();
    HIPCHK(hipMemsetAsync(d.n_active, 0, sizeof(int32_t), stream_));
    hipLaunchKernelGGL((k_begin_iter<T>), dim3(wb), dim3(64), 0, stream_, d);
    launch_step();
    const WinMeta &m = meta_[id];
    Lm lm;
    if (delta) HIPCHK(hipMemcpyAsync(delta, d.delta + m.u0, sizeof(double) * m.N, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipMemcpyAsync(&lm, d.lm + id, sizeof(Lm), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (mc) *mc = lm.step_valid ? lm.model_change : -1.0;
    return CTVIO_OK;
  }
  // Prior construction: A, b of the window's factors on the device (the linearise kernels), elimination of
  // the marginalised unknowns and the factorisation into (J0, r0) on the host (csrc/marginalize.hpp).
  int marginalize(int id, const int8_t *role, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
    if (!role || !n_keep || !kept || !J0 || !r0 || !(eps >= 0)) return fail(CTVIO_ERR_INVALID, "bad arguments");
    const WinMeta &m = meta_[id];
    const int N = m.N, P = m.P, L = m.L;
    for (int i = 0; i < N; ++i) if (role[i] < -1 || role[i] > 1) return fail(CTVIO_ERR_INVALID, "role must be -1, 0 or 1");
    std::vector<double> Hpp((size_t)P * P), W((size_t)P * std::max(L, 1)), Hll(std::max(L, 1)), g(N);
    const int rc = linearize(id, Hpp.data(), L ? W.data() : nullptr, L ? Hll.data() : nullptr, g.data(), nullptr);
    if (rc != CTVIO_OK) return rc;
    std::vector<double> A((size_t)N * N, 0.0);
    for (int i = 0; i < P; ++i) {
      for (int j = 0; j < P; ++j) A[(size_t)i * N + j] = Hpp[(size_t)i * P + j];
      for (int l = 0; l < L; ++l) { A[(size_t)i * N + P + l] = W[(size_t)i * L + l]; A[(size_t)(P + l) * N + i] = W[(size_t)i * L + l]; }
    }
    for (int l = 0; l < L; ++l) A[(size_t)(P + l) * N + P + l] = Hll[l];
    std::vector<int32_t> kv;
    std::vector<double> Jv, rv;
    const int n = marginalize_dense(N, A.data(), g.data(), role, eps, kv, Jv, rv);
    *n_keep = n;
    std::copy(kv.begin(), kv.end(), kept);
    std::copy(Jv.begin(), Jv.end(), J0);
    std::copy(rv.begin(), rv.end(), r0);
    return CTVIO_OK;
  }
  int gauge_restore(int n, const int32_t *ids, const int32_t *knot, const double *q0, const double *t0) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (n < 0 || (n && (!ids || !knot || !q0 || !t0))) return fail(CTVIO_ERR_INVALID, "bad arguments");
    for (int i = 0; i < n; ++i) {
      if (ids[i] < 0 || ids[i] >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
      if (knot[i] < 0 || knot[i] >= meta_[ids[i]].K) return fail(CTVIO_ERR_INVALID, "knot index out of range");
      for (int j = 0; j < i; ++j) if (ids[j] == ids[i]) return fail(CTVIO_ERR_INVALID, "window listed twice");
    }
    if (n == 0) return CTVIO_OK;
    DBuf<int32_t> di, dk; DBuf<double> dq, dt;
    HIPCHK(di.upload(std::vector<int32_t>(ids, ids + n), stream_)); HIPCHK(dk.upload(std::vector<int32_t>(knot, knot + n), stream_));
    HIPCHK(dq.upload(std::vector<double>(q0, q0 + 4 * (size_t)n), stream_)); HIPCHK(dt.upload(std::vector<double>(t0, t0 + 3 * (size_t)n), stream_));
    hipLaunchKernelGGL((k_gauge_restore<T>), dim3(n), dim3(64), 0, stream_, dev_, n, di.p, dk.p, dq.p, dt.p);
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    return CTVIO_OK;
  }
  int spline_eval(int id, int n, const int64_t *t_ns, double *pose7, double *vel3, double *omega3, double *acc3) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin || n < 0 || (n && !t_ns)) return fail(CTVIO_ERR_INVALID, "bad arguments");
    if (n == 0) return CTVIO_OK;
    std::vector<long long> rel(n);
    for (int i = 0; i < n; ++i) rel[i] = (long long)(t_ns[i] - wins_[id].w.t0_ns);
    DBuf<long long> dt; DBuf<double> dp, dv, dw, da; DBuf<int> derr;
    HIPCHK(dt.alloc(n)); HIPCHK(derr.alloc(1));
    HIPCHK(hipMemcpyAsync(dt.p, rel.data(), sizeof(long long) * n, hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemsetAsync(derr.p, 0, sizeof(int), stream_));
    if (pose7) HIPCHK(dp.alloc((size_t)7 * n));
    if (vel3) HIPCHK(dv.alloc((size_t)3 * n));
    if (omega3) HIPCHK(dw.alloc((size_t)3 * n));
    if (acc3) HIPCHK(da.alloc((size_t)3 * n));
    hipLaunchKernelGGL((k_spline_eval<T>), dim3(nblk(n, 256)), dim3(256), 0, stream_, dev_, id, n, dt.p, pose7 ? dp.p : nullptr,
                       vel3 ? dv.p : nullptr, omega3 ? dw.p : nullptr, acc3 ? da.p : nullptr, derr.p);
    int err = 0;
    if (pose7) HIPCHK(hipMemcpyAsync(pose7, dp.p, sizeof(double) * 7 * n, hipMemcpyDeviceToHost, stream_));
    if (vel3) HIPCHK(hipMemcpyAsync(vel3, dv.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
    if (omega3) HIPCHK(hipMemcpyAsync(omega3, dw.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
    if (acc3) HIPCHK(hipMemcpyAsync(acc3, da.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipMemcpyAsync(&err, derr.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (err) return fail(CTVIO_ERR_INVALID, "query time outside the spline");
    return CTVIO_OK;
  }
  int last_timing(double *ms8, int32_t *n8) override {
    if (ms8) std::copy(timing_, timing_ + 8, ms8);
    if (n8) { std::copy(ph_n_, ph_n_ + 7, n8); n8[7] = last_iters_; }
    return CTVIO_OK;
  }
  int set_profiling(int on) override { profiling_requested_ = on != 0; return CTVIO_OK; }

 private:
  ctvio_options opt_;
  hipStream_t stream_ = nullptr;
  hipEvent_t ev_[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool uploaded_ = false, profiling_ = false, profiling_requested_ = false, mixed_ = false;
  double timing_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_ms_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int32_t ph_n_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_iters_ = 0;
  std::vector<hipEvent_t> pev_;
  std::vector<int> pev_phase_;
  size_t pev_used_ = 0;
  std::vector<HostWindow> wins_;
  std::vector<WinMeta> meta_;
  Dev<T> dev_;
  int Mtot_ = 0, Vtot_ = 0;
  size_t chol_lds_ = 0, vis_lds_ = 0, vis_glb_ = 0;
  DBuf<VisItem> b_vitems_;
  DBuf<int32_t> b_lm_blk_off_, b_lm_blk_, b_v_slot_;
  DBuf<WinMeta> b_meta_;
  DBuf<double> b_quat_, b_pos_, b_bias_, b_rho_, b_ld_, b_cquat_, b_cpos_, b_cbias_, b_crho_, b_cld_, b_bc_w_, b_pH_, b_pb0_, b_pc0_, b_p_x0_;
  DBuf<double> b_chol_inv_, b_Hpp_, b_S_, b_Hll_, b_g_, b_rhs_, b_dd_, b_dinv_, b_cscale_, b_delta_;
  DBuf<int32_t> b_knot_win_, b_bias_win_, b_lm_win_, b_imu_grp_, b_v_win_, b_v_lm_, b_v_rowi_, b_v_rowj_, b_bc_win_, b_bc_i_, b_bc_j_, b_pcol_,
      b_p_kind_, b_p_index_, b_p_off_, b_vs_, b_nact_;
  DBuf<int64_t> b_v_ti_, b_v_tj_;
  DBuf<ImuGroup> b_groups_;
  DBuf<T> b_imu_rc_, b_vis_rc_, b_kjri_, b_imu_u_, b_imu_meas_, b_tiles_, b_v_obs_, b_Jv_, b_rv_, b_W_, b_Wc_;
  DBuf<uint8_t> b_active_;
  DBuf<Lm> b_lm_;
  DBuf<long long> b_dbg_;
  DBuf<double> b_snap_, b_imu_ud_, b_imu_meas_d_, b_v_obs_d_, b_kd_, b_ckd_;
  bool snap_valid_ = false, any_vis_lds_ = false, any_vis_glb_ = false;
};

template <> void SolverImpl<float>::launch_schur() {
  const Dev<float> &d = dev_;
  const int nt = (d.maxP + 1 + 31) / 32;
  if (opt_.use_mfma) {
    const size_t lds = ((size_t)2 * 16 * d.maxLdw + d.maxLdw + 32) * sizeof(float);
    // few windows: one workgroup per window leaves the chip idle and serialises 13 chunk round trips -- the per-tile
    // kernel (28 independent waves per window, W re-read per tile) has the shorter latency there
    const bool small = d.nwin < 192 || std::getenv("CTVIO_SCHUR_TILES");   // measured crossover ~256 windows per launch
    if (small) hipLaunchKernelGGL(k_schur_mfma, dim3(nt * (nt + 1) / 2 * 8 * ((d.nwin + 7) / 8)), dim3(64), 0, stream_, d, nt * (nt + 1) / 2);
    else if (d.maxLdw <= 224 && nt * (nt + 1) / 2 <= 32) hipLaunchKernelGGL((k_schur_window<7>), dim3(d.nwin), dim3(512), lds, stream_, d);
    else if (d.maxLdw <= 448 && nt * (nt + 1) / 2 <= 32) hipLaunchKernelGGL((k_schur_window<14>), dim3(d.nwin), dim3(512), lds, stream_, d);
    else hipLaunchKernelGGL(k_schur_mfma, dim3(nt * (nt + 1) / 2 * 8 * ((d.nwin + 7) / 8)), dim3(64), 0, stream_, d, nt * (nt + 1) / 2);
  }
  else hipLaunchKernelGGL((k_schur_generic<float>), dim3(nblk((long long)d.maxP * d.maxP, 256), d.nwin), dim3(256), 0, stream_, d);
}
template <> void SolverImpl<float>::launch_assemble_vis_lds(int parts) {
  const Dev<float> &d = dev_;
  if (opt_.use_mfma) hipLaunchKernelGGL((k_assemble_vis_mfma<VCH>), dim3(d.nwin, parts), dim3(512), vis_lds_, stream_, d);
  else hipLaunchKernelGGL((k_assemble_vis<float, VCH, true>), dim3(d.nwin, parts), dim3(512), vis_lds_, stream_, d);
}
template <> void SolverImpl<double>::launch_assemble_vis_lds(int parts) {
  const Dev<double> &d = dev_;
  hipLaunchKernelGGL((k_assemble_vis<double, VCH, true>), dim3(d.nwin, parts), dim3(512), vis_lds_, stream_, d);
}
template <> void SolverImpl<double>::launch_schur() {
  const Dev<double> &d = dev_;
  if (opt_.use_mfma) {   // fp64 matrix cores, one wave per 16 x 16 tile; k_rhs forms the reduced right-hand side
    const int nt = (d.maxP + 15) / 16, ntile = nt * (nt + 1) / 2;
    hipLaunchKernelGGL(k_schur_tile_f64, dim3(ntile * 8 * ((d.nwin + 7) / 8)), dim3(64), 0, stream_, d, ntile);
  } else {
    hipLaunchKernelGGL((k_schur_generic<double>), dim3(nblk((long long)d.maxP * d.maxP, 256), d.nwin), dim3(256), 0, stream_, d);
  }
}

}  // namespace ctv

// ================================================================================================ C ABI
struct ctvio_solver { std::unique_ptr<ctv::SolverBase> impl; };

extern "C" {

void ctvio_default_options(ctvio_options *o) {
  if (!o) return;
  std::memset(o, 0, sizeof *o);
  o->device = 0; o->precision = CTVIO_FP32; o->use_mfma = 1; o->check_every = 4;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
  o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32; o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32; o->max_consecutive_invalid_steps = 5;
  o->fp64_residuals = 1;
}
const char *ctvio_status_string(int32_t s) {
  switch (s) {
    case CTVIO_OK: return "ok";
    case CTVIO_ERR_INVALID: return "invalid argument";
    case CTVIO_ERR_NO_DEVICE: return "no HIP device (the product path has no CPU fallback)";
    case CTVIO_ERR_HIP: return "HIP runtime error";
    case CTVIO_ERR_STATE: return "call order violated";
    default: return "unknown status";
  }
}
const char *ctvio_last_error(void) { return ctv::g_err.c_str(); }
int32_t ctvio_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int32_t ctvio_create(const ctvio_options *opt, ctvio_solver **out) {
  if (!out) return ctv::fail(CTVIO_ERR_INVALID, "null out");
  *out = nullptr;
  ctvio_options o;
  if (opt) o = *opt; else ctvio_default_options(&o);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return ctv::fail(CTVIO_ERR_NO_DEVICE, "hipGetDeviceCount found no device");
  if (o.device < 0 || o.device >= n) return ctv::fail(CTVIO_ERR_INVALID, "device ordinal out of range");
  std::unique_ptr<ctvio_solver> s(new ctvio_solver);
  int rc;
  if (o.precision == CTVIO_FP64) { auto *p = new ctv::SolverImpl<double>(o); s->impl.reset(p); rc = p->init(); }
  else { auto *p = new ctv::SolverImpl<float>(o); s->impl.reset(p); rc = p->init(); }
  if (rc != CTVIO_OK) return rc;
  *out = s.release();
  return CTVIO_OK;
}
void ctvio_destroy(ctvio_solver *s) { delete s; }
// every entry point: null check, then the solver's device becomes current on this thread -- HIP's current device is per thread
// (default 0), and a multi-GPU rank that drives several solver handles from worker threads would otherwise launch on device 0
#define CHK_S if (!s) return ctv::fail(CTVIO_ERR_INVALID, "null solver"); if (int rc_bind_ = s->impl->bind()) return rc_bind_
int32_t ctvio_clear(ctvio_solver *s) { CHK_S; return s->impl->clear(); }
int32_t ctvio_add_window(ctvio_solver *s, const ctvio_window *w, int32_t *id) { CHK_S; return s->impl->add_window(w, id); }
int32_t ctvio_upload(ctvio_solver *s) { CHK_S; return s->impl->upload(); }
int32_t ctvio_num_windows(const ctvio_solver *s) { return s ? s->impl->num_windows() : 0; }
int32_t ctvio_solve(ctvio_solver *s, int32_t max_iterations, ctvio_summary *out) { CHK_S; return s->impl->solve(max_iterations, out); }
int32_t ctvio_get_state(ctvio_solver *s, int32_t id, double *quat, double *pos, double *bias, double *rho, double *ld) {
  CHK_S; return s->impl->get_state(id, quat, pos, bias, rho, ld);
}
int32_t ctvio_set_state(ctvio_solver *s, int32_t id, const double *quat, const double *pos, const double *bias, const double *rho, double ld) {
  CHK_S; return s->impl->set_state(id, quat, pos, bias, rho, ld);
}
int32_t ctvio_linearize(ctvio_solver *s, int32_t id, double *Hpp, double *W, double *Hll, double *g, double *cost) {
  CHK_S; return s->impl->linearize(id, Hpp, W, Hll, g, cost);
}
int32_t ctvio_cost(ctvio_solver *s, int32_t id, double *cost) { CHK_S; return s->impl->cost(id, cost); }
int32_t ctvio_lm_step(ctvio_solver *s, int32_t id, double mu, double *delta, double *model_cost_change) {
  CHK_S; return s->impl->lm_step(id, mu, delta, model_cost_change);
}
int32_t ctvio_marginalize(ctvio_solver *s, int32_t id, const int8_t *role, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0) {
  CHK_S; return s->impl->marginalize(id, role, eps, n_keep, kept, J0, r0);
}
int32_t ctvio_gauge_restore(ctvio_solver *s, int32_t n, const int32_t *ids, const int32_t *knot, const double *q0, const double *t0) {
  CHK_S; return s->impl->gauge_restore(n, ids, knot, q0, t0);
}
int32_t ctvio_spline_eval(ctvio_solver *s, int32_t id, int32_t n, const int64_t *t_ns, double *pose7, double *vel3, double *omega3, double *acc3) {
  CHK_S; return s->impl->spline_eval(id, n, t_ns, pose7, vel3, omega3, acc3);
}
int32_t ctvio_last_timing(ctvio_solver *s, double *ms8, int32_t *launches8) { CHK_S; return s->impl->last_timing(ms8, launches8); }
int32_t ctvio_snapshot_state(ctvio_solver *s) { CHK_S; return s->impl->snapshot(0); }
int32_t ctvio_restore_state(ctvio_solver *s) { CHK_S; return s->impl->snapshot(1); }
int32_t ctvio_set_profiling(ctvio_solver *s, int32_t on) { CHK_S; return s->impl->set_profiling(on); }
void *ctvio_stream(ctvio_solver *s) { return s ? s->impl->stream() : nullptr; }

}  // extern "C"
