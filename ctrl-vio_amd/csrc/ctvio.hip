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
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/ctvio.h"
#include "kernels.hpp"
#include "marginalize.hpp"
#include "marg_device.hpp"
#include "host_pack.hpp"

namespace ctv {

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return fail(CTVIO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Diagnostic / A-B switches (include/ctvio.h, "Diagnostic switches"): read from the environment ONCE per handle, in ctvio_create -- a later
// change of the environment cannot make upload and solve disagree about a kernel choice.  None of them is needed in production.
struct DebugSwitches {
  int stamps = 0;            // CTVIO_DEBUG_STAMPS=1     clock64 stamps of a few kernels, printed by ctvio_solve (disables the hipGraph)
  int store_path = -1;       // CTVIO_STORE_PATH=0/1     force the store-semantics assembly tail off / on
  int split_linearize = 0;   // CTVIO_SPLIT_LINEARIZE=1  IMU and visual evaluation as separate launches also for small batches (rocprofv3 runs)
  int merge_linearize = -1;  // CTVIO_MERGE_LINEARIZE=0/1 force the merged launch off / on
  int no_imu_band = 0;       // CTVIO_NO_IMU_BAND=1      k_misc adds the IMU knot blocks to Hpp without the LDS band
  int zero_kernel = 0;       // CTVIO_ZERO_KERNEL=1      k_zero_normal instead of the IMU kernel's zeroing share
  int imu_waves = 2048;      // CTVIO_IMU_WAVES=n        walking waves of k_imu_linearize_f64
  int imu_general = 0;       // CTVIO_IMU_GENERAL=1      every IMU group through the general body (same as use_mfma = 2)
  int schur_tiles = 0;       // CTVIO_SCHUR_TILES=1      tile Schur kernels also for large batches of small windows
  int schur_copy_plain = 0;  // CTVIO_SCHUR_COPY_PLAIN=1 the per-window Schur kernel copies product-free tiles to S
  int chol_tiles = -1;       // CTVIO_CHOL_TILES=0/1/3   P <= 223: panel kernel / k_cholesky_tiles (round 5: barriers) / k_cholesky_flow (default)
  int dense = 0;             // CTVIO_DENSE=1            the sparsity plan degenerates to the dense one
  int schur_tile2 = -1;      // CTVIO_SCHUR_TILE2=0/1    one wave per tile / per 2 x 2 tiles
  int marg_debug = 0;        // CTVIO_MARG_DEBUG=1       sweep trace of the device eigen-solver on stderr
  int marg_host = 0;         // CTVIO_MARG_HOST=1        ctvio_marginalize on the host factorisation
  int shard_oversubscribe = 0;   // CTVIO_SHARD_OVERSUBSCRIBE=1  TEST ONLY: more shards than devices (ctvio_shards_used)
};
static DebugSwitches read_debug_switches() {
  DebugSwitches g;
  struct { const char *name; int *dst; } const tab[] = {
      {"CTVIO_DEBUG_STAMPS", &g.stamps}, {"CTVIO_STORE_PATH", &g.store_path}, {"CTVIO_SPLIT_LINEARIZE", &g.split_linearize},
      {"CTVIO_MERGE_LINEARIZE", &g.merge_linearize}, {"CTVIO_NO_IMU_BAND", &g.no_imu_band}, {"CTVIO_ZERO_KERNEL", &g.zero_kernel},
      {"CTVIO_IMU_WAVES", &g.imu_waves}, {"CTVIO_IMU_GENERAL", &g.imu_general}, {"CTVIO_SCHUR_TILES", &g.schur_tiles},
      {"CTVIO_SCHUR_COPY_PLAIN", &g.schur_copy_plain}, {"CTVIO_CHOL_TILES", &g.chol_tiles}, {"CTVIO_DENSE", &g.dense},
      {"CTVIO_SCHUR_TILE2", &g.schur_tile2}, {"CTVIO_MARG_DEBUG", &g.marg_debug}, {"CTVIO_MARG_HOST", &g.marg_host},
      {"CTVIO_SHARD_OVERSUBSCRIBE", &g.shard_oversubscribe}};
  for (const auto &t : tab)
    if (const char *e = std::getenv(t.name)) *t.dst = (e[0] == '\0') ? 1 : std::atoi(e);   // (set but empty counts as 1)
  g.imu_waves = std::max(1, g.imu_waves);
  return g;
}

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

// Owning host copy of one window (ctvio_add_window: the caller's buffers are only read inside that call).
struct HostWindow {
  ctvio_window w;  // scalars + pointers into the vectors below
  std::vector<double> quat, pos, bias, rho, imu_gyro, imu_acc, bc_w, v_pi, v_pj, pJ0, pr0, p_x0, v_cauchy;
  std::vector<uint8_t> knot_const;
  std::vector<int64_t> imu_t, v_ti, v_tj;
  std::vector<int32_t> imu_bias, bc_i, bc_j, v_lm, v_rowi, v_rowj, p_kind, p_index, p_off;
  template <class U> static const U *own(std::vector<U> &dst, const U *src, size_t n) {
    if (src) dst.assign(src, src + n); else dst.assign(n, U(0));
    return dst.data();
  }
  explicit HostWindow(const ctvio_window &c) : w(c) {
    w.quat = own(quat, c.quat, (size_t)4 * c.K); w.pos = own(pos, c.pos, (size_t)3 * c.K);
    w.bias = own(bias, c.bias, (size_t)6 * c.F); w.rho = own(rho, c.rho, (size_t)c.L);
    w.imu_t = own(imu_t, c.imu_t, (size_t)c.M); w.imu_gyro = own(imu_gyro, c.imu_gyro, (size_t)3 * c.M);
    w.imu_acc = own(imu_acc, c.imu_acc, (size_t)3 * c.M); w.imu_bias = own(imu_bias, c.imu_bias, (size_t)c.M);
    w.bc_i = own(bc_i, c.bc_i, (size_t)c.NB); w.bc_j = own(bc_j, c.bc_j, (size_t)c.NB); w.bc_w = own(bc_w, c.bc_w, (size_t)6 * c.NB);
    w.v_lm = own(v_lm, c.v_lm, (size_t)c.V); w.v_ti = own(v_ti, c.v_ti, (size_t)c.V); w.v_tj = own(v_tj, c.v_tj, (size_t)c.V);
    w.v_rowi = own(v_rowi, c.v_rowi, (size_t)c.V); w.v_rowj = own(v_rowj, c.v_rowj, (size_t)c.V);
    w.v_pi = own(v_pi, c.v_pi, (size_t)2 * c.V); w.v_pj = own(v_pj, c.v_pj, (size_t)2 * c.V);
    w.pJ0 = own(pJ0, c.pJ0, (size_t)c.pn * c.pn); w.pr0 = own(pr0, c.pr0, (size_t)c.pn);
    w.p_kind = own(p_kind, c.p_kind, (size_t)c.pnb); w.p_index = own(p_index, c.p_index, (size_t)c.pnb);
    w.p_off = own(p_off, c.p_off, (size_t)c.pnb); w.p_x0 = own(p_x0, c.p_x0, (size_t)4 * c.pnb);
    if (c.v_cauchy) w.v_cauchy = own(v_cauchy, c.v_cauchy, (size_t)c.V);
    if (c.knot_const) w.knot_const = own(knot_const, c.knot_const, (size_t)c.K);
  }
  HostWindow(const HostWindow &) = delete;
  HostWindow &operator=(const HostWindow &) = delete;
};

struct SolverBase {
  virtual ~SolverBase() {}
  virtual int clear() = 0;
  virtual int add_window(const ctvio_window *w, int32_t *id) = 0;
  virtual int upload() = 0;
  virtual int set_batch(int n, const ctvio_window *wins) = 0;
  virtual int num_windows() const = 0;
  virtual int solve(int max_iters, ctvio_summary *out) = 0;
  virtual int get_state(int id, double *quat, double *pos, double *bias, double *rho, double *ld) = 0;
  virtual int get_batch_state(double *quat, double *pos, double *bias, double *rho, double *ld) = 0;
  virtual int set_state(int id, const double *quat, const double *pos, const double *bias, const double *rho, double ld) = 0;
  virtual int linearize(int id, double *Hpp, double *W, double *Hll, double *g, double *cost) = 0;
  virtual int cost(int id, double *cost) = 0;
  virtual int lm_step(int id, double mu, double *delta, double *mc) = 0;
  virtual int spline_eval(int id, int n, const int64_t *t_ns, double *pose7, double *vel3, double *omega3, double *acc3,
                          const double *q_SI = nullptr, const double *p_SI = nullptr) = 0;
  virtual int spline_eval_batch(int64_t n, const int32_t *win, const int64_t *t_ns, double *pose7, double *vel3, double *omega3, double *acc3,
                                double *kernel_ms) = 0;
  virtual int gauge_restore(int n, const int32_t *ids, const int32_t *knot, const double *q0, const double *t0) = 0;
  virtual int marginalize(int id, const int8_t *role, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0) = 0;
  virtual int marginalize_batch(const int8_t *role, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0) = 0;
  virtual int residual_summary(int id, double *sums, int32_t *counts4) = 0;
  virtual int bind() = 0;   // make the solver's device current on the calling thread (every ABI entry: callers use threads)
  virtual int snapshot(int restore) = 0;
  virtual int last_timing(double *ms8, int32_t *n8) = 0;
  virtual int set_profiling(int on) = 0;
  virtual void *stream() = 0;
  virtual int graph_captures() const = 0;
  virtual int marg_ran_on_host() const = 0;
};

class SolverImpl : public SolverBase {
 public:
  // visual blocks per work item (k_assemble_vis_mfma): eight per-wave staging areas must fit beside the fp64 LDS Hessian -- sized by the
  // constexpr the kernel itself lays its LDS out with (kernels_assemble.hpp: vis_stage_bytes)
  static constexpr int VCH = 8;
  static constexpr size_t vis_stage_bytes() { return ctv::vis_stage_bytes(8, VCH); }
  explicit SolverImpl(const ctvio_options &o) : opt_(o), dbg_(read_debug_switches()) {}
  ~SolverImpl() override {
    if (stream_) (void)hipStreamDestroy(stream_);
    for (auto &e : ev_) if (e) (void)hipEventDestroy(e);
    for (auto &e : pev_) (void)hipEventDestroy(e);
    if (lm_host_) (void)hipHostFree(lm_host_);
    if (state_host_) (void)hipHostFree(state_host_);
    if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
    if (call_host_) (void)hipHostFree(call_host_);
  }
  int init() {
    HIPCHK(hipSetDevice(opt_.device));
    HIPCHK(hipStreamCreate(&stream_));
    for (auto &e : ev_) HIPCHK(hipEventCreate(&e));
    // kernels that need more than 64 KiB of dynamic LDS
    HIPCHK(hipFuncSetAttribute((const void *)k_cholesky_solve<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_cholesky_solve<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_cholesky_tiles<16, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_cholesky_flow, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_assemble_vis_mfma<VCH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_assemble_vis_mfma<VCH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_assemble_vis_mfma<VCH, true, 8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_assemble_vis_mfma<VCH, true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_schur_window_f64<7, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_schur_window_f64<5, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_schur_window_f64<5, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void *)k_misc, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));   // (+ 2 KB of static LDS)
    return CTVIO_OK;
  }
  int bind() override { HIPCHK(hipSetDevice(opt_.device)); return CTVIO_OK; }
  int clear() override { own_.clear(); uploaded_ = false; return CTVIO_OK; }
  int num_windows() const override { return uploaded_ ? (int)meta_.size() : (int)own_.size(); }
  void *stream() override { return (void *)stream_; }
  int graph_captures() const override { return graph_captures_; }
  int marg_ran_on_host() const override { return marg_ran_on_host_; }

  int add_window(const ctvio_window *w, int32_t *id) override {
    std::string err;
    if (!validate_window(w, err)) return fail(CTVIO_ERR_INVALID, err);
    own_.emplace_back(new HostWindow(*w));
    if (id) *id = (int32_t)own_.size() - 1;
    uploaded_ = false;
    return CTVIO_OK;
  }
  int upload() override {
    if (own_.empty()) return fail(CTVIO_ERR_STATE, "no windows");
    std::vector<const ctvio_window *> ptr(own_.size());
    for (size_t i = 0; i < own_.size(); ++i) ptr[i] = &own_[i]->w;
    return pack_and_upload(ptr, false);
  }
  // ctvio_set_batch: the windows are read straight from the caller's buffers (no intermediate copy)
  int set_batch(int n, const ctvio_window *wins) override {
    if (n <= 0 || !wins) return fail(CTVIO_ERR_INVALID, "empty batch");
    own_.clear();
    uploaded_ = false;
    std::vector<const ctvio_window *> ptr((size_t)n);
    for (int i = 0; i < n; ++i) ptr[i] = wins + i;
    return pack_and_upload(ptr, true);
  }

  // ---------------------------------------------------------------------------------------- pack + upload
  // Two passes over the windows, both spread over host threads: (1) validate, sort, count; (2) fill the pinned staging
  // arena, which mirrors the device input arena byte for byte -- one hipMemcpyAsync carries the batch to HBM.  Work
  // buffers live in a second, device-only arena.  Both arenas only ever grow, so a stream of equally sized batches
  // allocates nothing after the first one.
  int pack_and_upload(const std::vector<const ctvio_window *> &wins, bool validate) {
    const int nw = (int)wins.size();
    const int nth = host_threads(opt_.host_threads);
    std::vector<PackTmp> tmp((size_t)nw);
    std::atomic<int> first_bad{nw};
    // The batch's factorisation: P <= 223 for every window -> the register-resident tile Cholesky, which keeps the whole triangle (dense
    // envelope); otherwise the panel kernel, which works inside every window's envelope.  (Sizes alone decide: known before planning.)
    int maxP_pre = 0;
    for (int wi = 0; wi < nw; ++wi) if (wins[wi]) maxP_pre = std::max(maxP_pre, 6 * wins[wi]->K + 6 * wins[wi]->F + 1);
    chol_tiles_ = chol_tiles_for(maxP_pre);   // (decided here, once per batch: launch_step and the Schur launch use the member)
    const bool dense_env = chol_tiles_ != 0 || sparsity_off();
    pool_.run(nw, nth, [&](int wi) {
      if (validate && !validate_window(wins[wi], tmp[wi].err)) {
        int cur = first_bad.load();
        while (wi < cur && !first_bad.compare_exchange_weak(cur, wi)) {}
        return;
      }
      plan_window(wins[wi], VCH, tmp[wi]);
      if (tmp[wi].err.empty()) plan_sparsity(wins[wi], dense_env, sparsity_off(), tmp[wi]);
      if (!tmp[wi].err.empty()) {
        int cur = first_bad.load();
        while (wi < cur && !first_bad.compare_exchange_weak(cur, wi)) {}
      }
    });
    if (first_bad.load() < nw) return fail(CTVIO_ERR_INVALID, "window " + std::to_string(first_bad.load()) + ": " + tmp[first_bad.load()].err);
    // ---- offsets (serial prefix sums)
    meta_.assign(nw, WinMeta());
    t0_.resize(nw);
    int64_t H0 = 0, W0 = 0, pH0 = 0;
    int K0 = 0, F0 = 0, L0 = 0, M0 = 0, V0 = 0, B0 = 0, U0 = 0, Pp0 = 0, pv0 = 0, pb = 0, G0 = 0, I0 = 0, A0 = 0, TR0 = 0, maxSpan = 1;
    int maxN = 0, maxP = 0, maxPn = 0, maxL = 0, maxLdw = 0, maxK = 0, maxSchurTiles = 0;
    size_t vis_lds_bytes = vis_stage_bytes(), vis_glb_bytes = vis_stage_bytes();
    for (int wi = 0; wi < nw; ++wi) {
      const ctvio_window &w = *wins[wi];
      WinMeta &m = meta_[wi];
      t0_[wi] = w.t0_ns;
      m.K = w.K; m.F = w.F; m.L = w.L; m.M = w.M; m.NB = w.NB; m.V = w.V;
      m.P = 6 * w.K + 6 * w.F + 1; m.N = m.P + w.L; m.pn = w.pn; m.pnb = w.pnb;
      m.knot0 = K0; m.bias0 = F0; m.lm0 = L0; m.imu0 = M0; m.vis0 = V0; m.bc0 = B0; m.u0 = U0; m.p0 = Pp0;
      m.grp0 = G0; m.ngrp = tmp[wi].ngrp; m.vitem0 = I0; m.nvitem = tmp[wi].nvitem; m.Vp = tmp[wi].Vp;
      m.anc0 = A0; m.A = tmp[wi].A;
      m.tr0 = TR0; m.ntr = tmp[wi].ntr; m.Lobs = tmp[wi].Lobs; TR0 += tmp[wi].ntr; maxSpan = std::max(maxSpan, tmp[wi].max_span);
      m.ldw = (m.P + 1 + 31) / 32 * 32; m.Lpad = std::max(2, (w.L + 1) / 2 * 2);
      m.pv0 = pv0; m.pblk0 = pb; m.fix_ld = w.fix_ld; m.lock_bg = w.lock_bg; m.lock_ba = w.lock_ba; m.fixed_upto = w.fixed_upto;
      m.H0 = H0; m.W0 = W0; m.pH0 = pH0; m.ldh = (m.P + 15) / 16 * 16; m.dt_ns = w.dt_ns; m.inv_dt = 1e9 / (double)w.dt_ns;
      for (int i = 0; i < 4; ++i) m.q_CI[i] = w.q_CI[i];
      for (int i = 0; i < 3; ++i) { m.p_CI[i] = w.p_CI[i]; m.gravity[i] = w.gravity[i]; }
      for (int i = 0; i < 6; ++i) m.imu_w[i] = w.imu_w[i];
      m.img_w = w.img_w; m.cauchy_a = w.cauchy_a; m.ld_lo = w.ld_lo; m.ld_hi = w.ld_hi;
      {
        const size_t K6 = 6 * (size_t)w.K, nG = K6 + 1, nH = K6 * (K6 + 1) / 2 + K6 + 1 + nG;
        const size_t need = ((nH + 3) & ~(size_t)3) * sizeof(double) + 32 + vis_stage_bytes();   // fp64 accumulators in LDS
        const size_t need_glb = ((nG + 3) & ~(size_t)3) * sizeof(double) + 16 + vis_stage_bytes();
        m.vis_lds = need <= 160 * 1024 ? 1 : 0;
        vis_lds_bytes = std::max(vis_lds_bytes, m.vis_lds ? need : need_glb);
        vis_glb_bytes = std::max(vis_glb_bytes, need_glb);
      }
      K0 += w.K; F0 += w.F; L0 += w.L; M0 += w.M; V0 += m.Vp; B0 += w.NB; U0 += m.N; Pp0 += m.P; pv0 += w.pn; pb += w.pnb;
      G0 += m.ngrp; I0 += m.nvitem; A0 += m.A;
      H0 += (int64_t)m.P * m.ldh; W0 += (int64_t)m.Lpad * m.ldw; pH0 += (int64_t)w.pn * w.pn;
      maxN = std::max(maxN, m.N); maxP = std::max(maxP, m.P); maxPn = std::max(maxPn, w.pn);
      maxL = std::max(maxL, m.L); maxLdw = std::max(maxLdw, m.ldw); maxK = std::max(maxK, m.K);
      {   // 16 x 16 tiles of the reduced system that receive Schur products (k_schur_window_f64): knot columns, line delay, rhs row
        const int ntl = m.ldw / 16, K6 = 6 * m.K;
        int cnt = 0;
        for (int ti = 0; ti < ntl; ++ti)
          for (int tj = 0; tj <= ti; ++tj) {
            const bool nzr = (16 * ti < K6) || (m.P >= 16 * ti && m.P - 1 < 16 * ti + 16);
            const bool nzc = (16 * tj < K6) || (m.P - 1 >= 16 * tj && m.P - 1 < 16 * tj + 16);
            cnt += (nzr && nzc) ? 1 : 0;
          }
        maxSchurTiles = std::max(maxSchurTiles, cnt);
      }
    }
    const size_t chol_lds = (size_t)(2 * 32 * 34 + 32 + 34 + 32 + (size_t)((std::max(maxP - 32, 0) + 1 + 15) / 16 * 16) * 32) * sizeof(double);
    if (chol_lds > 160 * 1024) return fail(CTVIO_ERR_INVALID, "window too large for the single-workgroup Cholesky (P > ~600)");
    const size_t Mt = (size_t)std::max(M0, 1), Vt = (size_t)std::max(V0, 1), At = (size_t)std::max(A0, 1);
    Mtot_ = M0; Vtot_ = V0;
    // ---- input arena layout (host mirror + device)
    size_t off = 0;
    auto seg = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_meta = seg(sizeof(WinMeta) * nw);
    const size_t o_state = seg(sizeof(double) * ((size_t)7 * K0 + 6 * F0 + L0 + nw));   // quat | pos | bias | rho | ld, contiguous
    const size_t o_knot_win = seg(4 * (size_t)K0), o_bias_win = seg(4 * (size_t)F0), o_lm_win = seg(4 * (size_t)L0);
    const size_t o_groups = seg(sizeof(ImuGroup) * (size_t)G0), o_imu_grp = seg(4 * Mt);
    const size_t o_imu_u = seg(sizeof(double) * Mt), o_imu_meas = seg(sizeof(double) * 6 * Mt);
    const size_t o_v_win = seg(4 * Vt), o_v_lm = seg(4 * Vt), o_v_anc = seg(4 * Vt), o_v_rowj = seg(4 * Vt);
    const size_t o_v_tj = seg(8 * Vt), o_v_obs = seg(sizeof(double) * 2 * Vt);
    const size_t o_v_cauchy = seg(8 * Vt), o_vb_win = seg(4 * (Vt / 64 + 1));
    const size_t o_a_win = seg(4 * At), o_a_lm = seg(4 * At), o_a_row = seg(4 * At), o_a_t = seg(8 * At), o_a_obs = seg(8 * 2 * At);
    const size_t o_vitems = seg(sizeof(VisItem) * (size_t)std::max(I0, 1)), o_vblk = seg(4 * Vt), o_vblk_anc = seg(4 * Vt);
    const size_t o_bc_win = seg(4 * (size_t)B0), o_bc_i = seg(4 * (size_t)B0), o_bc_j = seg(4 * (size_t)B0), o_bc_w = seg(8 * 6 * (size_t)B0);
    const size_t o_pJ0 = seg(8 * (size_t)pH0), o_pr0 = seg(8 * (size_t)pv0);
    const size_t o_pH = seg(8 * (size_t)pH0), o_pb0 = seg(8 * (size_t)pv0), o_pc0 = seg(8 * (size_t)nw), o_p_x0 = seg(8 * 4 * (size_t)pb);
    const size_t o_pcol = seg(4 * (size_t)pv0), o_p_kind = seg(4 * (size_t)pb), o_p_index = seg(4 * (size_t)pb), o_p_off = seg(4 * (size_t)pb);
    const size_t o_pinv = seg(4 * (size_t)Pp0), o_bgl_off = seg(4 * ((size_t)F0 + nw)), o_bgl = seg(4 * (size_t)std::max(G0, 1));
    const size_t o_active = seg((size_t)U0);
    const size_t o_lm_pos = seg(4 * (size_t)L0), o_lm_at = seg(4 * (size_t)L0), o_lm_klo = seg(4 * (size_t)L0), o_lm_khi = seg(4 * (size_t)L0);
    const size_t o_tl_beg = seg(4 * (size_t)TR0), o_tl_end = seg(4 * (size_t)TR0), o_env_first = seg(4 * (size_t)TR0), o_env_tile = seg(4 * (size_t)TR0);
    const size_t in_bytes = off;
    bool grew = false;
    HIPCHK(hipStreamSynchronize(stream_));   // the previous batch may still be reading the staging arena (H2D in flight)
    HIPCHK(in_.reserve(in_bytes, true, &grew));
    char *hb = in_.host, *db = in_.dev;
#define CTV_H(type, o) reinterpret_cast<type *>(hb + (o))
#define CTV_D(type, o) reinterpret_cast<type *>(db + (o))
    std::memcpy(CTV_H(WinMeta, o_meta), meta_.data(), sizeof(WinMeta) * nw);
    double *h_quat = CTV_H(double, o_state), *h_pos = h_quat + (size_t)4 * K0, *h_bias = h_pos + (size_t)3 * K0, *h_rho = h_bias + (size_t)6 * F0,
           *h_ld = h_rho + L0;
    int32_t *h_knot_win = CTV_H(int32_t, o_knot_win), *h_bias_win = CTV_H(int32_t, o_bias_win), *h_lm_win = CTV_H(int32_t, o_lm_win);
    ImuGroup *h_groups = CTV_H(ImuGroup, o_groups);
    int32_t *h_imu_grp = CTV_H(int32_t, o_imu_grp);
    double *h_imu_u = CTV_H(double, o_imu_u), *h_imu_meas = CTV_H(double, o_imu_meas), *h_v_obs = CTV_H(double, o_v_obs);
    double *h_v_cauchy = CTV_H(double, o_v_cauchy);
    int32_t *h_vb_win = CTV_H(int32_t, o_vb_win);
    int32_t *h_v_win = CTV_H(int32_t, o_v_win), *h_v_lm = CTV_H(int32_t, o_v_lm), *h_v_anc = CTV_H(int32_t, o_v_anc), *h_v_rowj = CTV_H(int32_t, o_v_rowj);
    int64_t *h_v_tj = CTV_H(int64_t, o_v_tj);
    int32_t *h_a_win = CTV_H(int32_t, o_a_win), *h_a_lm = CTV_H(int32_t, o_a_lm), *h_a_row = CTV_H(int32_t, o_a_row);
    int64_t *h_a_t = CTV_H(int64_t, o_a_t);
    double *h_a_obs = CTV_H(double, o_a_obs);
    VisItem *h_vitems = CTV_H(VisItem, o_vitems);
    int32_t *h_vblk = CTV_H(int32_t, o_vblk), *h_vblk_anc = CTV_H(int32_t, o_vblk_anc);
    int32_t *h_bc_win = CTV_H(int32_t, o_bc_win), *h_bc_i = CTV_H(int32_t, o_bc_i), *h_bc_j = CTV_H(int32_t, o_bc_j);
    double *h_pJ0 = CTV_H(double, o_pJ0), *h_pr0 = CTV_H(double, o_pr0);
    double *h_bc_w = CTV_H(double, o_bc_w), *h_pH = CTV_H(double, o_pH), *h_pb0 = CTV_H(double, o_pb0), *h_pc0 = CTV_H(double, o_pc0),
           *h_p_x0 = CTV_H(double, o_p_x0);
    int32_t *h_pcol = CTV_H(int32_t, o_pcol), *h_p_kind = CTV_H(int32_t, o_p_kind), *h_p_index = CTV_H(int32_t, o_p_index), *h_p_off = CTV_H(int32_t, o_p_off);
    uint8_t *h_active = CTV_H(uint8_t, o_active);
    int32_t *h_pinv = CTV_H(int32_t, o_pinv), *h_bgl_off = CTV_H(int32_t, o_bgl_off), *h_bgl = CTV_H(int32_t, o_bgl);
    int32_t *h_lm_pos = CTV_H(int32_t, o_lm_pos), *h_lm_at = CTV_H(int32_t, o_lm_at), *h_lm_klo = CTV_H(int32_t, o_lm_klo), *h_lm_khi = CTV_H(int32_t, o_lm_khi);
    int32_t *h_tl_beg = CTV_H(int32_t, o_tl_beg), *h_tl_end = CTV_H(int32_t, o_tl_end), *h_env_first = CTV_H(int32_t, o_env_first), *h_env_tile = CTV_H(int32_t, o_env_tile);
    h_lm_pos_ = h_lm_pos; h_ld_ = h_ld;
    // ---- second pass: every window fills its own slices
    pool_.run(nw, nth, [&](int wi) {
      const ctvio_window &w = *wins[wi];
      const WinMeta &m = meta_[wi];
      const PackTmp &t = tmp[wi];
      std::memcpy(h_quat + (size_t)4 * m.knot0, w.quat, sizeof(double) * 4 * w.K);
      std::memcpy(h_pos + (size_t)3 * m.knot0, w.pos, sizeof(double) * 3 * w.K);
      std::memcpy(h_bias + (size_t)6 * m.bias0, w.bias, sizeof(double) * 6 * w.F);
      if (w.L) std::memcpy(h_rho + m.lm0, w.rho, sizeof(double) * w.L);
      h_ld[wi] = w.fix_ld ? w.ld : std::min(std::max(w.ld, w.ld_lo), w.ld_hi);   // Ceres IterationZero: project on the feasible set
      std::fill(h_knot_win + m.knot0, h_knot_win + m.knot0 + w.K, wi);
      std::fill(h_bias_win + m.bias0, h_bias_win + m.bias0 + w.F, wi);
      std::fill(h_lm_win + m.lm0, h_lm_win + m.lm0 + w.L, wi);
      if (w.L) {   // sparsity plan: rows of W in sorted landmark order and their knot spans
        std::memcpy(h_lm_pos + m.lm0, t.lm_pos.data(), 4 * (size_t)w.L); std::memcpy(h_lm_at + m.lm0, t.lm_at.data(), 4 * (size_t)w.L);
        std::memcpy(h_lm_klo + m.lm0, t.row_klo.data(), 4 * (size_t)w.L); std::memcpy(h_lm_khi + m.lm0, t.row_khi.data(), 4 * (size_t)w.L);
      }
      std::memcpy(h_tl_beg + m.tr0, t.tl_beg.data(), 4 * (size_t)m.ntr); std::memcpy(h_tl_end + m.tr0, t.tl_end.data(), 4 * (size_t)m.ntr);
      std::memcpy(h_env_first + m.tr0, t.env_first.data(), 4 * (size_t)m.ntr); std::memcpy(h_env_tile + m.tr0, t.env_tile.data(), 4 * (size_t)m.ntr);
      // IMU samples in (segment, bias) order; groups = runs of equal (segment, bias)
      int g = m.grp0 - 1;
      for (int i = 0; i < w.M; ++i) {
        const int src = t.iorder[i];
        if (i == 0 || t.iseg[src] != t.iseg[t.iorder[i - 1]] || w.imu_bias[src] != w.imu_bias[t.iorder[i - 1]])
          h_groups[++g] = ImuGroup{wi, t.iseg[src], w.imu_bias[src], i, 0, m.knot0 + t.iseg[src], m.bias0 + w.imu_bias[src], m.imu0 + i};
        h_groups[g].count++;
        const size_t e = (size_t)m.imu0 + i;
        h_imu_grp[e] = g;
        const int64_t st = w.imu_t[src] - w.t0_ns;
        const double uu = (double)(st % w.dt_ns) / (double)w.dt_ns;
        h_imu_u[e] = (double)uu;
        for (int c = 0; c < 3; ++c) {
          h_imu_meas[(size_t)c * Mt + e] = (double)w.imu_gyro[3 * src + c];
          h_imu_meas[(size_t)(3 + c) * Mt + e] = (double)w.imu_acc[3 * src + c];
        }
      }
      // anchors (the i ends, landmark-major) and visual blocks: evaluation slots in landmark-major order (padding slots: window -1,
      // harmless values)
      for (int a = 0; a < m.A; ++a) {
        const int v = t.anc_rep[a];
        const size_t e = (size_t)m.anc0 + a;
        h_a_win[e] = wi; h_a_lm[e] = w.v_lm[v]; h_a_row[e] = w.v_rowi[v]; h_a_t[e] = w.v_ti[v] - w.t0_ns;
        h_a_obs[e] = w.v_pi[2 * v]; h_a_obs[At + e] = w.v_pi[2 * v + 1];
      }
      std::fill(h_vb_win + m.vis0 / 64, h_vb_win + (m.vis0 + m.Vp) / 64, wi);
      for (int i = 0; i < m.Vp; ++i) {
        const int v = t.lord[i];
        const size_t e = (size_t)m.vis0 + i;
        if (v < 0) {
          h_v_win[e] = -1; h_v_lm[e] = 0; h_v_anc[e] = m.anc0; h_v_tj[e] = 0; h_v_rowj[e] = 0; h_v_cauchy[e] = 0.0;
          for (int c = 0; c < 2; ++c) h_v_obs[(size_t)c * Vt + e] = 0.0;
          continue;
        }
        h_v_win[e] = wi; h_v_lm[e] = w.v_lm[v]; h_v_anc[e] = m.anc0 + t.anc_of[v];
        h_v_tj[e] = w.v_tj[v] - w.t0_ns;
        h_v_rowj[e] = w.v_rowj[v];
        h_v_obs[e] = (double)w.v_pj[2 * v]; h_v_obs[Vt + e] = (double)w.v_pj[2 * v + 1];
        h_v_cauchy[e] = w.v_cauchy ? w.v_cauchy[v] : w.cauchy_a;
      }
      // the assembly's items: <= VCH blocks of one frame pair, frame-pair order, as lists of slots (vblk)
      int it = m.vitem0 - 1;
      for (int i = 0; i < w.V; ++i) {
        const int v = t.vord[i];
        const bool fresh = (i == 0) || w.v_ti[v] != w.v_ti[t.vord[i - 1]] || w.v_tj[v] != w.v_tj[t.vord[i - 1]] || h_vitems[it].count >= VCH;
        if (fresh) h_vitems[++it] = VisItem{m.vis0 + i, 0};
        h_vitems[it].count++;
        h_vblk[(size_t)m.vis0 + i] = m.vis0 + t.vpos[v];
        h_vblk_anc[(size_t)m.vis0 + i] = m.anc0 + t.anc_of[v];
      }
      for (int i = w.V; i < m.Vp; ++i) { h_vblk[(size_t)m.vis0 + i] = m.vis0; h_vblk_anc[(size_t)m.vis0 + i] = m.anc0; }   // (unused tail of the window's list)
      for (int b = 0; b < w.NB; ++b) { h_bc_win[m.bc0 + b] = wi; h_bc_i[m.bc0 + b] = w.bc_i[b]; h_bc_j[m.bc0 + b] = w.bc_j[b]; }
      if (w.NB) std::memcpy(h_bc_w + (size_t)6 * m.bc0, w.bc_w, sizeof(double) * 6 * w.NB);
      // prior: J0^T J0 (row-major n*n), J0^T r0, r0^T r0 in fp64; J0 is column-major (Eigen)
      const int n = w.pn;
      h_pc0[wi] = 0.0;
      int32_t *col = h_pcol + m.pv0;
      if (n > 0) {
        for (int b = 0; b < w.pnb; ++b) {
          const int kind = w.p_kind[b], idx = w.p_index[b];
          int u0 = 0;
          switch (kind) {
            case CTVIO_PK_ROT: u0 = 6 * idx; break;
            case CTVIO_PK_POS: u0 = 6 * idx + 3; break;
            case CTVIO_PK_BG: u0 = 6 * w.K + 6 * idx; break;
            case CTVIO_PK_BA: u0 = 6 * w.K + 6 * idx + 3; break;
            default: u0 = m.P - 1;
          }
          for (int k = 0; k < prior_block_size(kind); ++k) col[w.p_off[b] + k] = u0 + k;
        }
        double *pH = h_pH + m.pH0, *pb0 = h_pb0 + m.pv0;
        std::memcpy(h_pJ0 + m.pH0, w.pJ0, sizeof(double) * (size_t)n * n);
        std::memcpy(h_pr0 + m.pv0, w.pr0, sizeof(double) * (size_t)n);
        for (int i = 0; i < n; ++i) {
          const double *Ji = w.pJ0 + (size_t)i * n;
          double bi = 0;
          for (int r = 0; r < n; ++r) bi += Ji[r] * w.pr0[r];
          pb0[i] = bi;
          for (int j = 0; j <= i; ++j) {
            const double *Jj = w.pJ0 + (size_t)j * n;
            double s = 0;
            for (int r = 0; r < n; ++r) s += Ji[r] * Jj[r];
            pH[(size_t)i * n + j] = s; pH[(size_t)j * n + i] = s;
          }
        }
        double c0 = 0;
        for (int r = 0; r < n; ++r) c0 += w.pr0[r] * w.pr0[r];
        h_pc0[wi] = c0;
        std::memcpy(h_p_kind + m.pblk0, w.p_kind, 4 * (size_t)w.pnb); std::memcpy(h_p_index + m.pblk0, w.p_index, 4 * (size_t)w.pnb);
        std::memcpy(h_p_off + m.pblk0, w.p_off, 4 * (size_t)w.pnb); std::memcpy(h_p_x0 + (size_t)4 * m.pblk0, w.p_x0, 8 * 4 * (size_t)w.pnb);
      }
      active_mask(&w, t, m.P, col, h_active + m.u0);
      // inverse column map of the prior, and the IMU groups of every bias state (group order) -- read by the store-semantics assembly
      std::fill(h_pinv + m.p0, h_pinv + m.p0 + m.P, -1);
      for (int i = 0; i < n; ++i) h_pinv[m.p0 + col[i]] = i;
      {
        int32_t *off = h_bgl_off + m.bias0 + wi;
        std::fill(off, off + w.F + 1, 0);
        for (int gi = 0; gi < m.ngrp; ++gi) off[h_groups[m.grp0 + gi].bias + 1]++;
        for (int f = 0; f < w.F; ++f) off[f + 1] += off[f];
        std::vector<int32_t> fill(off, off + w.F);
        for (int gi = 0; gi < m.ngrp; ++gi) h_bgl[m.grp0 + fill[h_groups[m.grp0 + gi].bias]++] = m.grp0 + gi;
        for (int f = 0; f <= w.F; ++f) off[f] += m.grp0;   // absolute positions in bgl
      }
    });
    // ---- device pointers of the input arena
    Dev &d = dev_;
    std::memset(&d, 0, sizeof d);
    d.nwin = nw; d.Ktot = K0; d.Ftot = F0; d.Ltot = L0; d.Mtot = M0; d.Gtot = G0; d.Vtot = V0; d.Atot = A0;
    d.NBtot = B0; d.Utot = U0; d.maxN = maxN; d.maxP = maxP; d.maxPn = maxPn; d.maxL = maxL; d.maxLdw = maxLdw; maxK_ = maxK; max_schur_tiles_ = maxSchurTiles;
    d.wins = CTV_D(WinMeta, o_meta);
    d.quat = CTV_D(double, o_state); d.pos = d.quat + (size_t)4 * K0; d.bias = d.pos + (size_t)3 * K0; d.rho = d.bias + (size_t)6 * F0; d.ld = d.rho + L0;
    d.knot_win = CTV_D(int32_t, o_knot_win); d.bias_win = CTV_D(int32_t, o_bias_win); d.lm_win = CTV_D(int32_t, o_lm_win);
    d.groups = CTV_D(ImuGroup, o_groups); d.imu_grp = CTV_D(int32_t, o_imu_grp); d.imu_u = CTV_D(double, o_imu_u); d.imu_meas = CTV_D(double, o_imu_meas);
    d.v_cauchy = CTV_D(double, o_v_cauchy); d.vb_win = CTV_D(int32_t, o_vb_win);
    d.v_win = CTV_D(int32_t, o_v_win); d.v_lm = CTV_D(int32_t, o_v_lm); d.v_anc = CTV_D(int32_t, o_v_anc); d.v_rowj = CTV_D(int32_t, o_v_rowj);
    d.v_tj = CTV_D(int64_t, o_v_tj); d.v_obs = CTV_D(double, o_v_obs);
    d.a_win = CTV_D(int32_t, o_a_win); d.a_lm = CTV_D(int32_t, o_a_lm); d.a_row = CTV_D(int32_t, o_a_row); d.a_t = CTV_D(int64_t, o_a_t);
    d.a_obs = CTV_D(double, o_a_obs);
    d.vitems = CTV_D(VisItem, o_vitems); d.vblk = CTV_D(int32_t, o_vblk); d.vblk_anc = CTV_D(int32_t, o_vblk_anc);
    d.bc_win = CTV_D(int32_t, o_bc_win); d.bc_i = CTV_D(int32_t, o_bc_i); d.bc_j = CTV_D(int32_t, o_bc_j); d.bc_w = CTV_D(double, o_bc_w);
    d.pJ0 = CTV_D(double, o_pJ0); d.pr0 = CTV_D(double, o_pr0);
    d.pH = CTV_D(double, o_pH); d.pb0 = CTV_D(double, o_pb0); d.pc0 = CTV_D(double, o_pc0); d.p_x0 = CTV_D(double, o_p_x0);
    d.pcol = CTV_D(int32_t, o_pcol); d.p_kind = CTV_D(int32_t, o_p_kind); d.p_index = CTV_D(int32_t, o_p_index); d.p_off = CTV_D(int32_t, o_p_off);
    d.active = CTV_D(uint8_t, o_active);
    d.pinv = CTV_D(int32_t, o_pinv); d.bgl_off = CTV_D(int32_t, o_bgl_off); d.bgl = CTV_D(int32_t, o_bgl);
    d.lm_pos = CTV_D(int32_t, o_lm_pos); d.lm_at = CTV_D(int32_t, o_lm_at); d.lm_klo = CTV_D(int32_t, o_lm_klo); d.lm_khi = CTV_D(int32_t, o_lm_khi);
    d.tl_beg = CTV_D(int32_t, o_tl_beg); d.tl_end = CTV_D(int32_t, o_tl_end); d.env_first = CTV_D(int32_t, o_env_first); d.env_tile = CTV_D(int32_t, o_env_tile);
    d.max_span6 = 6 * maxSpan;
#undef CTV_H
#undef CTV_D
    HIPCHK(hipMemcpyAsync(in_.dev, in_.host, in_bytes, hipMemcpyHostToDevice, stream_));
    in_bytes_ = in_bytes;
    any_vis_lds_ = any_vis_glb_ = false;
    // (a window whose packed Hessian does not fit in LDS counts as "global" even without visual blocks -- e.g. an IMU-only predict of a
    // long spline: the store-semantics tail only finishes LDS-resident windows, so such a batch must take the accumulate path)
    for (const auto &mm : meta_) { if (mm.vis_lds) any_vis_lds_ = true; else any_vis_glb_ = true; }
    all_windows_have_imu_ = !meta_.empty();
    for (const auto &mm : meta_) if (mm.ngrp == 0) all_windows_have_imu_ = false;
    deterministic_ = opt_.deterministic > 0 || (opt_.deterministic < 0 && nw <= 64);
    // (the member goes into the launch signature and selects kernels: it must say what RUNS -- the default falls back to the accumulate path
    //  for batches the order-fixed assembly cannot cover, and then it is off)
    if (!any_vis_lds_ || any_vis_glb_) { if (opt_.deterministic <= 0) deterministic_ = false; }
    // The order-fixed accumulation exists for batches whose every window keeps its packed Hessian in LDS, on the matrix-core kernels.
    // An explicit request that cannot be honoured is an error; the default (-1) falls back to the accumulate path for such batches.
    if (opt_.deterministic > 0 && (!any_vis_lds_ || any_vis_glb_))
      return fail(CTVIO_ERR_INVALID, "deterministic = 1 needs every window's packed Hessian in LDS (K <= 25): this batch would "
                                     "fall back to floating-point atomics");
    maxK_ = maxK;
    // ---- work arena (device only)
    state_doubles_ = (size_t)7 * K0 + 6 * F0 + L0 + nw;
    off = 0;
    const size_t o_cstate = seg(8 * state_doubles_), o_snap = seg(8 * state_doubles_);
    const size_t o_lkd = seg(8 * 3 * (size_t)K0), o_kjri = seg(sizeof(double) * 9 * (size_t)K0);
    const size_t o_tiles = seg(sizeof(double) * 1024 * (size_t)G0);
    const size_t o_imu_cost = seg(8 * (size_t)std::max(G0, 1)), o_vis_cost = seg(8 * ((Vt + 63) / 64)), o_misc_cost = seg(8 * (size_t)nw);
    // packed partial Hessians of the multi-part store-semantics assembly (knot triangle + line-delay row + gradient per part)
    const size_t part_stride = ((size_t)6 * maxK * (6 * maxK + 1) / 2 + 2 * (6 * (size_t)maxK + 1) + 7) & ~(size_t)7;
    const int nparts_alloc = store_path() ? vis_parts() : 1;
    const size_t o_pgrad = seg(8 * (size_t)std::max(pv0, 1)), o_Hpart = seg(nparts_alloc > 1 ? 8 * part_stride * nparts_alloc * (size_t)nw : 8);
    const size_t o_Jt = seg(sizeof(double) * VT_ROWS * 64 * ((Vt + 63) / 64)), o_vsj = seg(4 * Vt);
    const size_t o_arec = seg(8 * (size_t)AREC * At), o_a_s = seg(4 * At);
    // two normal-equation sets (current linearisation / speculative linearisation at the candidate, Lm::cur)
    const size_t o_Hpp = seg(8 * (size_t)H0), o_Hpp1 = seg(8 * (size_t)H0), o_S = seg(8 * (size_t)H0);
    const size_t o_zero0 = off;   // ---- zeroed at every upload from here ...
    const size_t o_W = seg(sizeof(double) * (size_t)W0), o_W1 = seg(sizeof(double) * (size_t)W0), o_Hll = seg(8 * (size_t)L0), o_Hll1 = seg(8 * (size_t)L0),
                 o_g = seg(8 * (size_t)U0), o_g1 = seg(8 * (size_t)U0), o_delta = seg(8 * (size_t)U0),
                 o_cscale = seg(8 * (size_t)U0), o_lm = seg(sizeof(Lm) * (size_t)nw), o_nact = seg(16), o_dbg = seg(8 * 128);
    const size_t o_zero1 = off;   // ---- ... to here
    const size_t o_rhs = seg(8 * (size_t)Pp0), o_dd = seg(8 * (size_t)U0), o_dinv = seg(8 * (size_t)L0), o_grs = seg(8 * (size_t)L0);
    d.chol_nblk = (maxP + 31) / 32;
    d.line_search = opt_.line_search ? 1 : 0;
    const size_t o_chol_inv = seg(8 * (size_t)nw * d.chol_nblk * 1024);
    HIPCHK(work_.reserve(off, false, &grew));
    if (grew) HIPCHK(hipMemsetAsync(work_.dev, 0, work_.cap, stream_));   // fresh memory may hold NaN patterns (0 * NaN in masked products)
    char *wb = work_.dev;
#define CTV_W(type, o) reinterpret_cast<type *>(wb + (o))
    d.cquat = CTV_W(double, o_cstate); d.cpos = d.cquat + (size_t)4 * K0; d.cbias = d.cpos + (size_t)3 * K0; d.crho = d.cbias + (size_t)6 * F0; d.cld = d.crho + L0;
    snap_ = CTV_W(double, o_snap);
    d.lkd = CTV_W(double, o_lkd); d.kjri = CTV_W(double, o_kjri); d.imu_tiles = CTV_W(double, o_tiles);
    d.imu_cost = CTV_W(double, o_imu_cost); d.vis_cost = CTV_W(double, o_vis_cost); d.misc_cost = CTV_W(double, o_misc_cost);
    d.pgrad = CTV_W(double, o_pgrad); d.Hpart = CTV_W(double, o_Hpart); d.npart_stride = (int32_t)part_stride;
    d.Jt = CTV_W(double, o_Jt); d.vsj = CTV_W(int32_t, o_vsj); d.arec = CTV_W(double, o_arec); d.a_s = CTV_W(int32_t, o_a_s);
    d.HppS[0] = CTV_W(double, o_Hpp); d.HppS[1] = CTV_W(double, o_Hpp1); d.S = CTV_W(double, o_S);
    d.WS[0] = CTV_W(double, o_W); d.WS[1] = CTV_W(double, o_W1); d.HllS[0] = CTV_W(double, o_Hll); d.HllS[1] = CTV_W(double, o_Hll1);
    d.gS[0] = CTV_W(double, o_g); d.gS[1] = CTV_W(double, o_g1);
    d.delta = CTV_W(double, o_delta); d.cscale = CTV_W(double, o_cscale); d.lm = CTV_W(Lm, o_lm); d.n_active = CTV_W(int32_t, o_nact); d.span_viol = d.n_active + 1;
    d.dbg = dbg_.stamps ? CTV_W(long long, o_dbg) : nullptr;
    d.rhs = CTV_W(double, o_rhs); d.dd = CTV_W(double, o_dd); d.dinv = CTV_W(double, o_dinv); d.grs = CTV_W(double, o_grs); d.chol_inv = CTV_W(double, o_chol_inv);
#undef CTV_W
    HIPCHK(hipMemsetAsync(wb + o_zero0, 0, o_zero1 - o_zero0, stream_));
    // pinned landing areas of the results
    if ((size_t)nw > lm_host_cap_) {
      if (lm_host_) (void)hipHostFree(lm_host_);
      lm_host_cap_ = (size_t)nw + nw / 8 + 16;
      HIPCHK(hipHostMalloc((void **)&lm_host_, sizeof(Lm) * lm_host_cap_, hipHostMallocDefault));
    }
    chol_lds_ = chol_lds;
    snap_valid_ = false;
    vis_lds_ = vis_lds_bytes;
    vis_glb_ = vis_glb_bytes;
    d.schur_plain_in_H = schur_plain_in_H_for_batch();
    uploaded_ = true;
    return CTVIO_OK;
  }

  // ---------------------------------------------------------------------------------------- launches
  static int nblk(long long n, int b) { return (int)std::max<long long>((n + b - 1) / b, 1); }
  // Workgroups per window of the visual assembly: batches smaller than the chip split a window's items over several parts.  In the
  // deterministic mode a part is ONE wave (its LDS additions happen in program order) and there are more of them.
  int vis_parts() const {
    const int nw = std::max((int)meta_.size(), 1);
    if (store_path()) return std::min(32, std::max(1, 512 / nw));
    return std::min(8, std::max(1, 256 / nw));
  }
  // Store-semantics assembly tail (kernels.hpp: bias_rows_store; every entry written once, no atomics): the deterministic mode, when
  // every window's packed Hessian is LDS resident.  (CTVIO_STORE_PATH=1 forces it for the throughput mode too: measured slower there,
  // the bias-row gather costs more than the zeroing + atomic passes it replaces -- 14.5 vs 13.4 ms per 2048-window solve.)
  bool store_path() const {
    if (!any_vis_lds_ || any_vis_glb_) return false;
    if (dbg_.store_path >= 0) return dbg_.store_path == 1;
    return deterministic_;
  }
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
  // Linearisation of every window the mode selects (kernels.hpp: LIN_AT_X / LIN_SPEC / COST_AT_X) into its normal-equation set;
  // the cost partials of the evaluated state come out on the way.
  void launch_linearize(int mode) {
    const Dev &d = dev_;
    const int nw = d.nwin;
    const bool merged = merge_linearize();
    ph_begin(PH_ASM_REST);
    if (!store_path()) { if (!imu_zero_mode()) hipLaunchKernelGGL(k_zero_normal, dim3(64, nw), dim3(256), 0, stream_, d, vis_parts() == 1 ? 1 : 0, mode); }
    else if (!merged) hipLaunchKernelGGL(k_misc, dim3(nw), dim3(256), std::max(d.maxPn, 1) * sizeof(double), stream_, d, mode, 1, 0);   // prior gradient + cost share
    ph_end();
    if (merged) {
      // Small batches: TWO launches for the whole linearisation.  k_pre_linearize: the anchors' records, the IMU groups the specialised body
      // leaves out and (store-semantics path) the prior gradient + cost share -- three launches of 5 - 7 us each until round 5; then
      // k_linearize_f64: both evaluations (independent work: their latencies overlap on batches smaller than the chip).  A profiled
      // solve (and CTVIO_SPLIT_LINEARIZE=1, for rocprofv3 runs) keeps everything apart so that each kernel gets its own timing.
      const int nab = nblk(d.Atot, 64), with_misc = store_path() ? 1 : 0;
      hipLaunchKernelGGL(k_pre_linearize, dim3(nab + nw + (with_misc ? nw : 0)), dim3(64), 0, stream_, d, mode, imu_general_only(), imu_zero_mode(), nab, with_misc);
      hipLaunchKernelGGL(k_linearize_f64, dim3(d.Gtot + nblk(d.Vtot, 64)), dim3(64), 0, stream_, d, mode, imu_general_only(), imu_zero_mode());
      return;
    }
    ph_begin(PH_IMU_LIN);
    if (d.Gtot) launch_imu_linearize(mode);
    ph_end();
    ph_begin(PH_VIS_LIN);   // (one timed group: the anchors' records, then the blocks)
    if (d.Atot) hipLaunchKernelGGL(k_vis_anchor, dim3(nblk(d.Atot, 64)), dim3(64), 0, stream_, d, mode);   // the i ends, once per anchor
    if (d.Vtot) hipLaunchKernelGGL(k_vis_eval, dim3(nblk(d.Vtot, 64)), dim3(64), 0, stream_, d, mode);
    ph_end();
  }
  // The merged launch runs the visual body with the IMU body's register allocation (one wave per SIMD): only for batches smaller than
  // the chip, where the single-wave latencies of the two evaluations overlap instead of adding up.  CTVIO_MERGE_LINEARIZE = 0 / 1 forces
  // the choice (A/B measurements).
  bool merge_linearize() const {
    const Dev &d = dev_;
    if (!(d.Gtot && d.Vtot && !profiling_ && !dbg_.split_linearize)) return false;
    if (dbg_.merge_linearize >= 0) return dbg_.merge_linearize == 1;
    return d.nwin <= 128;
  }
  void launch_assemble(int mode) {
    const Dev &d = dev_;
    const int nw = d.nwin;
    const int parts = vis_parts();   // few windows: split each window's items over several workgroups to fill the chip
    if (store_path()) {
      // every entry of Hpp / g is written once, completely, with a plain store: no zeroing pass, no k_assemble_imu, no atomics
      ph_begin(PH_ASM_VIS);
      launch_assemble_vis_store(parts, mode);
      if (parts > 1) hipLaunchKernelGGL(k_reduce_finalize, dim3(deterministic_ ? 48 : 24, nw), dim3(256), 0, stream_, d, mode, parts);
      else hipLaunchKernelGGL(k_bias_rows, dim3(8, nw), dim3(256), 0, stream_, d, mode);
      ph_end();
      if (mode != LIN_SPEC) {   // (the candidate's gradient norm: k_pass_end)
        ph_begin(PH_ASM_REST);
        hipLaunchKernelGGL(k_post_linearize, dim3(nblk(d.maxN, 256), nw), dim3(256), 0, stream_, d, mode);
        ph_end();
      }
      return;
    }
    ph_begin(PH_ASM_VIS);
    if (any_vis_lds_) launch_assemble_vis_lds(parts, mode);
    if (any_vis_glb_) launch_assemble_vis_glb(parts, mode);
    ph_end();
    ph_begin(PH_ASM_REST);
    // (the IMU tiles' bias rows, the bias chain and the prior in ONE launch: k_misc with assemble_imu_window in front)
    {
      // (windows without the LDS-resident Hessian: the IMU knot blocks are summed in an LDS band before they go to Hpp -- when it fits)
      const size_t dxb = (size_t)((std::max(d.maxPn, 1) + 1) & ~1) * sizeof(double), bandb = (size_t)144 * maxK_ * sizeof(double);
      const bool band = any_vis_glb_ && d.Gtot && dxb + bandb <= 150 * 1024 && !dbg_.no_imu_band;
      hipLaunchKernelGGL(k_misc, dim3(nw), dim3(256), band ? dxb + bandb : dxb, stream_, d, mode, 0, d.Gtot ? (band ? 2 : 1) : 0);
    }
    if (mode != LIN_SPEC) hipLaunchKernelGGL(k_post_linearize, dim3(nblk(d.maxN, 256), nw), dim3(256), 0, stream_, d, mode);
    ph_end();
  }
  // Trust-region step of every window that starts a new iteration (damping, Schur complement, Cholesky, back-substitution), then the
  // candidate x (+) alpha delta of every window with a valid step (also those inside the line search: new alpha, same delta).
  void launch_step() {
    const Dev &d = dev_;
    const int nw = d.nwin;
    ph_begin(PH_SCHUR);
    launch_schur();
    ph_end();
    ph_begin(PH_CHOL);
    // P <= 223: the register-resident tile kernel (S read once, nothing written back; 16 waves per window) for batches smaller than
    // the chip, where latency counts; large batches: the panel kernel with 4 waves, two windows per CU (throughput); windows beyond
    // 223 unknowns: the panel kernel, with 8 waves when there are fewer windows than CUs
    if (chol_tiles()) {
      const int ntr = d.maxP / 16 + 1;
      const size_t lds = (size_t)(272 + 2 * ntr * 272 + 32 * ntr + 4 + 768) * sizeof(double);   // identity + panel + inverses + vectors + parked tiles
      const size_t lds_flow = (size_t)(272 + 5 * ntr * 272 + 32 * ntr + 48) * sizeof(double);   // identity + inverses + sub-diagonal tiles + three panels + vectors + flags
      if (chol_tiles() == 1) hipLaunchKernelGGL((k_cholesky_tiles<16, 7>), dim3(nw), dim3(1024), lds, stream_, d);   // (A/B: round 5's kernel)
      else hipLaunchKernelGGL(k_cholesky_flow, dim3(nw), dim3(1024), lds_flow, stream_, d);
    }
    // (8 waves also when the panel's LDS footprint allows one workgroup per CU anyway -- P = 571: 157 KB -- where 4 waves left three quarters
    //  of the CU's wave slots empty)
    else if (nw <= 192 || chol_lds_ > 80 * 1024) hipLaunchKernelGGL((k_cholesky_solve<8>), dim3(nw), dim3(512), chol_lds_, stream_, d);
    else hipLaunchKernelGGL((k_cholesky_solve<4>), dim3(nw), dim3(256), chol_lds_, stream_, d);
    ph_end();
    ph_begin(PH_REST);
    // (fewer windows than CUs: 16 waves per window shorten the landmark back-substitution from 7 trips to 2)
    // (fewer windows than CUs: 8 waves per window shorten the landmark back-substitution; 16 waves -- a 128-register cap -- spilled 18
    //  registers to scratch and were measured slower: 3.15 vs 3.09 ms per single-window solve)
    if (nw <= 192) hipLaunchKernelGGL((k_step_finish<8>), dim3(nw), dim3(512), (size_t)d.maxP * sizeof(double), stream_, d);
    else hipLaunchKernelGGL((k_step_finish<4>), dim3(nw), dim3(256), (size_t)d.maxP * sizeof(double), stream_, d);
    ph_end();
  }
  void launch_schur();
  void launch_imu_linearize(int mode);
  // use_mfma = 2 (or CTVIO_IMU_GENERAL=1): every IMU group through the general body (k_imu_linearize_rest) -- the cross-check of the
  // specialised one and the tests' way into the path that large knot-to-knot rotations / anisotropic accelerometer weights take
  // The IMU linearisation kernels clear the accumulated parts of the normal equations on the side (kernels.hpp: imu_zero_share) when every
  // window has IMU groups: 1 = bias rows only (one visual-assembly part stores the knot x knot block), 2 = everything; 0 = k_zero_normal.
  int imu_zero_mode() const {
    if (store_path() || !all_windows_have_imu_ || dbg_.zero_kernel) return 0;
    return vis_parts() == 1 ? 1 : 2;
  }
  int imu_walk_waves() const { return dbg_.imu_waves; }
  int imu_general_only() const { return (dbg_.imu_general || opt_.use_mfma == 2) ? 1 : 0; }
  void launch_assemble_vis_lds(int parts, int mode);
  void launch_assemble_vis_glb(int parts, int mode);
  void launch_assemble_vis_store(int parts, int mode) {
    const Dev &d = dev_;
    if (deterministic_) hipLaunchKernelGGL((k_assemble_vis_mfma<VCH, true, 1, true>), dim3(d.nwin, parts), dim3(64), vis_lds_, stream_, d, mode);
    else hipLaunchKernelGGL((k_assemble_vis_mfma<VCH, true, 8, true>), dim3(d.nwin, parts), dim3(512), vis_lds_, stream_, d, mode);
  }
  // Large batches of small windows take the per-window Schur kernel (W staged through LDS once); everything else the tile kernels.
  bool schur_window_path() const {
    const Dev &d = dev_;
    const int nt = (d.maxLdw + 15) / 16, ntile = nt * (nt + 1) / 2;
    const size_t lds = schur_window_lds();
    const bool small = d.nwin < 192 || dbg_.schur_tiles;
    const int nc = 6 * maxK_ + 2;   // compact columns of W per landmark: knots, line delay, g_rho
    return !small && d.maxLdw <= 224 && ntile <= 112 && lds <= 160 * 1024 && nc <= 224;
  }
  size_t schur_window_lds() const { return ((size_t)2 * 16 * (dev_.maxLdw + 16) + 3 * dev_.maxLdw + 32 + 64) * sizeof(double); }   // + column vectors + the list of tiles with products
  // Dev::schur_plain_in_H is part of the Dev struct the captured graph is keyed on: decided once per upload, never inside a launch
  // (launch_schur used to set it, so every upload -- which clears Dev -- invalidated the cached hipGraph of the headline configuration).
  int schur_plain_in_H_for_batch() const { return (schur_window_path() && chol_tiles() != 0 && !dbg_.schur_copy_plain) ? 1 : 0; }
  // CTVIO_CHOL_TILES = 0 / 1 / 3 forces the choice (A/B measurements: panel kernel / k_cholesky_tiles / k_cholesky_flow)
  int chol_tiles_for(int maxP) const {
    if (maxP > 223) return 0;
    if (dbg_.chol_tiles >= 0) return dbg_.chol_tiles;
    return 3;   // (register-resident tiles as a data-flow of waves: k_cholesky_flow; 1 = round 5's k_cholesky_tiles, the barrier-per-panel form)
  }
  int chol_tiles() const { return chol_tiles_; }   // the batch's choice, taken in pack_and_upload
  // CTVIO_DENSE=1: the sparsity plan degenerates to the dense one (every row range = all landmarks, envelope = the whole triangle) -- the
  // A/B switch of the sparsity-aware kernels and the cross-check of tests/test_gpu_sparsity.py
  bool sparsity_off() const { return dbg_.dense == 1; }
  int n_state() const { return dev_.Ktot + dev_.Ftot + dev_.Ltot + dev_.nwin; }

  // One PASS of the device-resident LM: every running window advances by one phase -- a new trust-region iteration (damp, Schur,
  // Cholesky, back-substitute, candidate) or, inside Ceres' projected line search, one trial step (candidate at the new alpha) --
  // and the candidate is evaluated ONCE: speculative linearisation into the window's other normal-equation set, cost as a
  // by-product; k_pass_end accepts / rejects / continues the search, swaps the sets on acceptance and starts the next iteration
  // (continuation tests, LM diagonal).  The launch list is fixed: kernels skip windows that are not in the matching phase.
  void launch_pass() {
    Dev &d = dev_;
    const int nw = d.nwin, wb = nblk(nw, 64);
    launch_step();
    launch_linearize(LIN_SPEC);
    launch_assemble(LIN_SPEC);
    ph_begin(PH_REST);
    // (windows that start another pass count themselves in k_pass_end; the counter was cleared by k_step_finish)
    hipLaunchKernelGGL(k_pass_end, dim3(nw), dim3(256), 0, stream_, d);
    ph_end();
  }
  // The first linearisation of a solve (and of the diagnostic entries): knot-pair constants, normal equations and cost of the
  // current state in set 0, Jacobi scaling.
  void launch_initial(double mu, int keep_scale) {
    Dev &d = dev_;
    const int nw = d.nwin, wb = nblk(nw, 64);
    hipLaunchKernelGGL(k_lm_init, dim3(wb), dim3(64), 0, stream_, d, mu, keep_scale);
    hipLaunchKernelGGL(k_knot_prep, dim3(nblk(d.Ktot, 256)), dim3(256), 0, stream_, d);
    launch_linearize(LIN_AT_X);
    launch_assemble(LIN_AT_X);
    hipLaunchKernelGGL(k_initial_cost, dim3(nw), dim3(64), 0, stream_, d, 0);
  }
  // The pass as a hipGraph (captured once per batch shape: the kernel arguments are the Dev struct, so equal shapes in the
  // grow-only arenas give identical graphs), replayed instead of ~25 launches.
  // Everything the launch list depends on besides the Dev struct: dynamic LDS sizes, kernel choices (template arguments, which
  // assembly variants run).  Two batches with identical totals can differ in these (e.g. the same sum K split differently).
  std::vector<long long> launch_signature() const {
    return {(long long)vis_lds_, (long long)vis_glb_, (long long)any_vis_lds_, (long long)any_vis_glb_, (long long)maxK_, (long long)max_schur_tiles_,
            (long long)chol_lds_, (long long)opt_.use_mfma, (long long)vis_parts(), (long long)deterministic_, (long long)chol_tiles(), (long long)imu_zero_mode(), (long long)merge_linearize(), (long long)dev_.nwin,
            (long long)dbg_.schur_tile2};
  }
  int ensure_graph() {
    const std::vector<long long> sig = launch_signature();
    if (graph_exec_ && std::memcmp(&graph_dev_, &dev_, sizeof dev_) == 0 && sig == graph_sig_) return CTVIO_OK;
    if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
    launch_pass();
    HIPCHK(hipStreamEndCapture(stream_, &g));
    hipError_t e = hipGraphInstantiate(&graph_exec_, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { graph_exec_ = nullptr; return fail(CTVIO_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
    graph_dev_ = dev_;
    graph_sig_ = sig;
    ++graph_captures_;
    return CTVIO_OK;
  }

  // k_vis_eval counts evaluations that fall outside the knot span the packer planned for their landmark (host_pack.hpp: plan_sparsity): such a
  // row of W was written into a neighbour's columns.  Every entry point that launches the kernel ends with this check; the counter is cleared
  // again so that a later call on the same batch (after ctvio_set_state / ctvio_restore_state) starts clean.  The stream must be idle.
  int check_span_violation() {
    int32_t *viol = reinterpret_cast<int32_t *>(lm_host_ + lm_host_cap_ - 1) + 2;   // (pinned scratch record, beside the "still running" word)
    HIPCHK(hipMemcpyAsync(viol, dev_.span_viol, sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    if (*viol == 0) return CTVIO_OK;
    const int n = *viol;
    HIPCHK(hipMemsetAsync(dev_.span_viol, 0, sizeof(int32_t), stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    return fail(CTVIO_ERR_INTERNAL, std::to_string(n) + " evaluation(s) fell outside the planned knot span of their landmark (host_pack.hpp: plan_sparsity): "
                                    "the normal equations of this call are not to be trusted");
  }

  int solve(int max_iters, ctvio_summary *out) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (max_iters < 0) return fail(CTVIO_ERR_INVALID, "max_iterations < 0");
    Dev &d = dev_;
    const int nw = d.nwin, wb = nblk(nw, 64);
    set_params(max_iters);
    profiling_ = profiling_requested_;
    pev_phase_.clear(); pev_used_ = 0;
    const bool graph = opt_.use_graph && !profiling_ && !d.dbg;
    if (graph) { const int rc = ensure_graph(); if (rc != CTVIO_OK) return rc; }
    HIPCHK(hipEventRecord(ev_[8], stream_));
    launch_initial(opt_.initial_radius, 0);
    HIPCHK(hipMemsetAsync(d.n_active, 0, sizeof(int32_t), stream_));
    hipLaunchKernelGGL(k_begin_iter, dim3(nw), dim3(256), 0, stream_, d);   // the first iteration; later ones start in k_pass_end
    // max_iters passes finish every window that never enters the line search; the host looks at the "windows that start another
    // pass" counter every check_every passes and keeps launching while any is left
    const int check = std::max(1, opt_.check_every);
    const int pass_cap = (max_iters + 1) * 22 + 4;   // every LM iteration may take up to 20 trial steps + 1 re-evaluation
    int it = 0;
    for (;;) {
      if (graph) HIPCHK(hipGraphLaunch(graph_exec_, stream_)); else launch_pass();
      ++it;
      if (it >= pass_cap) break;
      if (it >= max_iters || it % check == 0) {
        int32_t *na = reinterpret_cast<int32_t *>(lm_host_ + lm_host_cap_ - 1);   // pinned scratch word
        HIPCHK(hipMemcpyAsync(na, d.n_active, sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        if (*na == 0) break;
      }
    }
    HIPCHK(hipEventRecord(ev_[9], stream_));
    Lm *lm = lm_host_;   // pinned: the copy does not stage through a runtime bounce buffer
    HIPCHK(hipMemcpyAsync(lm, d.lm, sizeof(Lm) * nw, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (const int rc = check_span_violation()) return rc;
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev_[8], ev_[9]));
    if (profiling_) ph_collect(); else { std::fill(ph_ms_, ph_ms_ + 8, 0.0); std::fill(ph_n_, ph_n_ + 8, 0); }
    profiling_ = false;
    std::copy(ph_ms_, ph_ms_ + 7, timing_);
    timing_[7] = ms;
    last_iters_ = it;
    if (d.dbg) {
      long long st[128];
      HIPCHK(hipMemcpy(st, d.dbg, sizeof st, hipMemcpyDeviceToHost));
      std::fprintf(stderr, "[ctvio] imu fast body, group 5000, clock64 deltas (prologue | per pass: loads+values, gyro jac, gyro rows+MFMA, accel jac, accel rows+MFMA | .. | epilogue):");
      for (int i = 65; i < 64 + 16 && st[i] != 0; ++i) std::fprintf(stderr, " %lld", st[i] - st[i - 1]);
      std::fprintf(stderr, "\n");
      std::fprintf(stderr, "[ctvio] cholesky clock64 deltas:");
      for (int i = 1; i < 24; ++i) std::fprintf(stderr, " %lld", st[i] - st[i - 1]);
      std::fprintf(stderr, "\n[ctvio] vis_eval<LIN> wave 1000, clock64 deltas (evaluation | contributions + segmented sums | record copy-out | per sweep: scatter, rows out):");
      for (int i = 32; i < 40; ++i) std::fprintf(stderr, " %lld", st[i] - st[i - 1]);
      std::fprintf(stderr, " | landmarks %lld, rows per sweep %lld", st[42] / 1000000, st[43] / 1000);
      std::fprintf(stderr, "\n[ctvio] schur_window clock64 deltas (prologue + first chunk staged | the chunk loop | product tiles out | other tiles out):");
      for (int i = 97; i < 128 && st[i] != 0; ++i) std::fprintf(stderr, " %lld", st[i] - st[i - 1]);
      std::fprintf(stderr, "\n[ctvio] assemble_vis clock64 deltas (zero | per item of rounds 0, 1: staged, run products.., scatter | rounds | imu tiles | H flush | g flush):");
      for (int i = 49; i < 63; ++i) std::fprintf(stderr, " %lld", st[i] - st[i - 1]);
      std::fprintf(stderr, "\n");
    }
    if (out)
      for (int w = 0; w < nw; ++w) {
        out[w].iterations = lm[w].iter; out[w].num_successful = lm[w].nsucc; out[w].num_unsuccessful = lm[w].nunsucc;
        out[w].termination = lm[w].status > 0 ? lm[w].status - 1 : 0;
        out[w].initial_cost = lm[w].initial_cost; out[w].final_cost = lm[w].cost; out[w].final_radius = lm[w].mu;
        out[w].num_line_search_steps = lm[w].nls_steps; out[w].num_line_search_reduced = lm[w].nls_reduced;
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
    else if (ld != h_ld_[id]) return fail(CTVIO_ERR_INVALID, "the line delay of a fix_ld window cannot change after the upload: the knot spans of its landmarks were planned for it");
    if (quat) HIPCHK(hipMemcpyAsync(dev_.quat + 4 * (size_t)m.knot0, quat, sizeof(double) * 4 * m.K, hipMemcpyHostToDevice, stream_));
    if (pos) HIPCHK(hipMemcpyAsync(dev_.pos + 3 * (size_t)m.knot0, pos, sizeof(double) * 3 * m.K, hipMemcpyHostToDevice, stream_));
    if (bias) HIPCHK(hipMemcpyAsync(dev_.bias + 6 * (size_t)m.bias0, bias, sizeof(double) * 6 * m.F, hipMemcpyHostToDevice, stream_));
    if (rho && m.L) HIPCHK(hipMemcpyAsync(dev_.rho + m.lm0, rho, sizeof(double) * m.L, hipMemcpyHostToDevice, stream_));
    HIPCHK(hipMemcpyAsync(dev_.ld + id, &ld, sizeof(double), hipMemcpyHostToDevice, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    return CTVIO_OK;
  }

  // device-side copy of the whole batch state (restore != 0: copy back); the state is one contiguous block
  int snapshot(int restore) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (restore && !snap_valid_) return fail(CTVIO_ERR_STATE, "no snapshot taken");
    HIPCHK(hipMemcpyAsync(restore ? dev_.quat : snap_, restore ? snap_ : dev_.quat, state_doubles_ * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    if (!restore) { HIPCHK(hipStreamSynchronize(stream_)); snap_valid_ = true; }
    return CTVIO_OK;
  }
  // every window's state in one device-to-host copy (concatenated in window order, like the device arrays)
  int get_batch_state(double *quat, double *pos, double *bias, double *rho, double *ld) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    const Dev &d = dev_;
    if (state_doubles_ > state_host_cap_) {
      if (state_host_) (void)hipHostFree(state_host_);
      state_host_cap_ = state_doubles_ + state_doubles_ / 8;
      HIPCHK(hipHostMalloc((void **)&state_host_, sizeof(double) * state_host_cap_, hipHostMallocDefault));
    }
    HIPCHK(hipMemcpyAsync(state_host_, d.quat, state_doubles_ * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    const double *p = state_host_;
    if (quat) std::memcpy(quat, p, sizeof(double) * 4 * d.Ktot);
    p += (size_t)4 * d.Ktot;
    if (pos) std::memcpy(pos, p, sizeof(double) * 3 * d.Ktot);
    p += (size_t)3 * d.Ktot;
    if (bias) std::memcpy(bias, p, sizeof(double) * 6 * d.Ftot);
    p += (size_t)6 * d.Ftot;
    if (rho && d.Ltot) std::memcpy(rho, p, sizeof(double) * d.Ltot);
    p += d.Ltot;
    if (ld) std::memcpy(ld, p, sizeof(double) * d.nwin);
    return CTVIO_OK;
  }

  int linearize(int id, double *Hpp, double *W, double *Hll, double *g, double *cost) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
    Dev &d = dev_;
    const int nw = d.nwin, wb = nblk(nw, 64);
    set_params(1);
    launch_initial(opt_.initial_radius, 0);
    const WinMeta &m = meta_[id];
    const int P = m.P;
    if (Hpp) {
      HIPCHK(hipMemcpy2DAsync(Hpp, sizeof(double) * (size_t)P, d.HppS[0] + m.H0, sizeof(double) * (size_t)m.ldh, sizeof(double) * (size_t)P, (size_t)P,
                              hipMemcpyDeviceToHost, stream_));
    }
    std::vector<double> Wh;
    if (W && m.L) {
      Wh.resize((size_t)m.Lpad * m.ldw);
      HIPCHK(hipMemcpyAsync(Wh.data(), d.WS[0] + m.W0, sizeof(double) * Wh.size(), hipMemcpyDeviceToHost, stream_));
    }
    if (Hll && m.L) HIPCHK(hipMemcpyAsync(Hll, d.HllS[0] + m.lm0, sizeof(double) * m.L, hipMemcpyDeviceToHost, stream_));
    if (g) HIPCHK(hipMemcpyAsync(g, d.gS[0] + m.u0, sizeof(double) * m.N, hipMemcpyDeviceToHost, stream_));
    Lm lm;
    HIPCHK(hipMemcpyAsync(&lm, d.lm + id, sizeof(Lm), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (const int rc = check_span_violation()) return rc;
    if (Hpp)
      for (int i = 0; i < P; ++i)
        for (int j = i + 1; j < P; ++j) Hpp[(size_t)i * P + j] = Hpp[(size_t)j * P + i];
    if (W && m.L)
      for (int i = 0; i < P; ++i)
        for (int l = 0; l < m.L; ++l) W[(size_t)i * m.L + l] = (double)Wh[(size_t)h_lm_pos_[m.lm0 + l] * m.ldw + i];   // (rows of W: sorted landmark order)
    if (cost) *cost = lm.cost;
    return CTVIO_OK;
  }
  int cost(int id, double *cost) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
    Dev &d = dev_;
    const int wb = nblk(d.nwin, 64);
    set_params(1);
    hipLaunchKernelGGL(k_lm_init, dim3(wb), dim3(64), 0, stream_, d, opt_.initial_radius, 1);
    hipLaunchKernelGGL(k_knot_prep, dim3(nblk(d.Ktot, 256)), dim3(256), 0, stream_, d);
    if (d.Gtot) launch_imu_linearize(COST_AT_X);
    if (d.Atot) hipLaunchKernelGGL(k_vis_anchor, dim3(nblk(d.Atot, 64)), dim3(64), 0, stream_, d, (int)COST_AT_X);
    if (d.Vtot) hipLaunchKernelGGL(k_vis_eval, dim3(nblk(d.Vtot, 64)), dim3(64), 0, stream_, d, (int)COST_AT_X);
    hipLaunchKernelGGL(k_misc, dim3(d.nwin), dim3(256), std::max(d.maxPn, 1) * sizeof(double), stream_, d, (int)COST_AT_X, 0, 0);
    hipLaunchKernelGGL(k_initial_cost, dim3(d.nwin), dim3(64), 0, stream_, d, 1);
    Lm lm;
    HIPCHK(hipMemcpyAsync(&lm, d.lm + id, sizeof(Lm), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (const int rc = check_span_violation()) return rc;
    if (cost) *cost = lm.cand_cost;
    return CTVIO_OK;
  }
  int lm_step(int id, double mu, double *delta, double *mc) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
    Dev &d = dev_;
    const int wb = nblk(d.nwin, 64);
    set_params(1);
    launch_initial(mu, 0);
    HIPCHK(hipMemsetAsync(d.n_active, 0, sizeof(int32_t), stream_));
    hipLaunchKernelGGL(k_begin_iter, dim3(d.nwin), dim3(256), 0, stream_, d);
    launch_step();
    const WinMeta &m = meta_[id];
    Lm lm;
    if (delta) HIPCHK(hipMemcpyAsync(delta, d.delta + m.u0, sizeof(double) * m.N, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipMemcpyAsync(&lm, d.lm + id, sizeof(Lm), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (const int rc = check_span_violation()) return rc;
    if (mc) *mc = lm.step_valid ? lm.model_change : -1.0;
    return CTVIO_OK;
  }
  // Prior construction, all on the device: A, b of every window's factors by the linearise kernels, then one
  // workgroup per window eliminates the marginalised unknowns and factors the rest (csrc/marg_device.hpp: parallel Jacobi
  // in LDS).  role: concatenated per window (sum N entries, window i at its unknown offset); `only` >= 0 restricts the work
  // to that window.  Outputs: n_keep[nwin]; kept at the window's unknown offset; J0 / r0 packed tightly in window order.
  int marg_device(const int8_t *role_all, int only, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0, bool *too_large,
                  int *stalled = nullptr) {
    if (stalled) *stalled = -1;
    Dev &d = dev_;
    const int nw = d.nwin, wb = nblk(nw, 64);
    *too_large = false;
    std::vector<MargMeta> metas((size_t)nw);
    std::vector<int32_t> iscr;
    size_t scr = 0, outd = 0;
    for (int w = 0; w < nw; ++w) {
      const WinMeta &m = meta_[w];
      MargMeta &mm = metas[w];
      std::memset(&mm, 0, sizeof mm);
      mm.N = m.N; mm.P = m.P;
      if (only >= 0 && w != only) { n_keep[w] = 0; continue; }
      const int8_t *role = role_all + m.u0;
      mm.idx0 = (int32_t)iscr.size();
      for (int i = 0; i < m.N; ++i) if (role[i] == 1) { iscr.push_back(i); mm.m++; }
      for (int i = 0; i < m.N; ++i) if (role[i] == 0) { iscr.push_back(i); kept[m.u0 + mm.n] = i; mm.n++; }
      n_keep[w] = mm.n;
      if (mm.m > MARG_MAXD || mm.n > MARG_MAXD) { *too_large = true; return CTVIO_OK; }
      const int np = std::max(mm.m, mm.n) + (std::max(mm.m, mm.n) & 1);
      mm.A0 = (int64_t)scr; scr += (size_t)m.N * m.N;
      mm.V0 = (int64_t)scr; scr += (size_t)mm.m * mm.m;
      mm.X0 = (int64_t)scr; scr += (size_t)mm.m * (mm.n + 1);
      mm.Y0 = (int64_t)scr; scr += (size_t)mm.m * (mm.n + 1);
      mm.rot0 = (int64_t)scr; scr += (size_t)MARG_MAX_SWEEPS * std::max(np - 1, 1) * (np / 2) * 2;
      mm.b0 = (int64_t)scr; scr += (size_t)mm.n;
      mm.J0 = (int64_t)outd; outd += (size_t)mm.n * mm.n;
      mm.r0 = (int64_t)outd; outd += (size_t)mm.n;
    }
    // normal equations of every window at its current state
    set_params(1);
    launch_initial(opt_.initial_radius, 0);
    HIPCHK(mg_meta_.upload(metas, stream_));
    if (iscr.empty()) iscr.push_back(0);
    HIPCHK(mg_idx_.upload(iscr, stream_));
    HIPCHK(mg_scr_.alloc(scr));
    HIPCHK(mg_out_.alloc(outd));
    constexpr size_t lds = ((size_t)MARG_MAXD * (MARG_MAXD + 1) / 2 + 4 * MARG_MAXD + 512) * sizeof(double) + 2 * MARG_MAXD * sizeof(int);
    if (!marg_attr_set_) { HIPCHK(hipFuncSetAttribute((const void *)k_marginalize, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); marg_attr_set_ = true; }
    hipLaunchKernelGGL(k_marginalize, dim3(nw), dim3(256), lds, stream_, d, mg_meta_.p, mg_idx_.p, mg_scr_.p, mg_out_.p, eps);
    std::vector<double> outh(std::max<size_t>(outd, 1));
    HIPCHK(hipMemcpyAsync(outh.data(), mg_out_.p, sizeof(double) * std::max<size_t>(outd, 1), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipMemcpyAsync(metas.data(), mg_meta_.p, sizeof(MargMeta) * nw, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (const int rc = check_span_violation()) return rc;
    size_t oj = 0, orr = 0;
    for (int w = 0; w < nw; ++w) {
      const MargMeta &mm = metas[w];
      if (mm.n <= 0) continue;
      if (dbg_.marg_debug && w == (only >= 0 ? only : 0)) {
        std::fprintf(stderr, "[ctvio] marg window %d: m %d n %d sweeps %d / %d; off/dia per sweep (A'):", w, mm.m, mm.n, mm.sweeps_m, mm.sweeps_n);
        for (int i = 0; i < 26 && i <= std::max(mm.sweeps_n, 0) + 1; ++i) std::fprintf(stderr, " %.2e", mm.trace[26 + i]);
        std::fprintf(stderr, "\n");
      }
      if (mm.status) { if (stalled) *stalled = w; return fail(CTVIO_ERR_HIP, "device eigen-solver did not converge (window " + std::to_string(w) + ")"); }
      std::memcpy(J0 + oj, outh.data() + mm.J0, sizeof(double) * (size_t)mm.n * mm.n);
      std::memcpy(r0 + orr, outh.data() + mm.r0, sizeof(double) * (size_t)mm.n);
      oj += (size_t)mm.n * mm.n; orr += (size_t)mm.n;
    }
    return CTVIO_OK;
  }
  int marginalize_batch(const int8_t *role, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (!role || !n_keep || !kept || !J0 || !r0 || !(eps >= 0)) return fail(CTVIO_ERR_INVALID, "bad arguments");
    for (int i = 0; i < dev_.Utot; ++i) if (role[i] < -1 || role[i] > 1) return fail(CTVIO_ERR_INVALID, "role must be -1, 0 or 1");
    bool too_large = false;
    int stalled = -1;
    marg_ran_on_host_ = 0;
    const int rc = marg_device(role, -1, eps, n_keep, kept, J0, r0, &too_large, &stalled);
    if (rc != CTVIO_OK && stalled >= 0)
      return fail(CTVIO_ERR_HIP, "device eigen-solver did not converge for window " + std::to_string(stalled) + ": call ctvio_marginalize for it (host factorisation)");
    if (rc != CTVIO_OK) return rc;
    if (too_large) return fail(CTVIO_ERR_INVALID, "a window has more than " + std::to_string(MARG_MAXD) + " marginalised or kept unknowns: use ctvio_marginalize");
    return CTVIO_OK;
  }
  // one window; windows beyond the device eigen-solver's size (m or n > MARG_MAXD) take the host path (csrc/marginalize.hpp)
  int marginalize(int id, const int8_t *role, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "window id out of range");
    if (!role || !n_keep || !kept || !J0 || !r0 || !(eps >= 0)) return fail(CTVIO_ERR_INVALID, "bad arguments");
    const WinMeta &m = meta_[id];
    const int N = m.N, P = m.P, L = m.L;
    for (int i = 0; i < N; ++i) if (role[i] < -1 || role[i] > 1) return fail(CTVIO_ERR_INVALID, "role must be -1, 0 or 1");
    marg_ran_on_host_ = 0;
    if (!dbg_.marg_host) {
      std::vector<int8_t> role_all((size_t)dev_.Utot, (int8_t)-1);
      std::copy(role, role + N, role_all.begin() + m.u0);
      std::vector<int32_t> nk((size_t)dev_.nwin), kv((size_t)dev_.Utot);
      bool too_large = false;
      int stalled = -1;
      const int rc = marg_device(role_all.data(), id, eps, nk.data(), kv.data(), J0, r0, &too_large, &stalled);
      if (rc != CTVIO_OK && stalled < 0) return rc;
      // the in-LDS Jacobi sweep stalled above its (tight) off-diagonal bound: the host Householder / QL path below takes over
      if (rc == CTVIO_OK && !too_large) {
        *n_keep = nk[id];
        std::copy(kv.begin() + m.u0, kv.begin() + m.u0 + nk[id], kept);
        return CTVIO_OK;
      }
    }
    // Host leg (csrc/marginalize.hpp: Householder tridiagonalisation + QL on the host cores; the normal equations still come from the device
    // kernels): taken when the window is beyond the device eigen-solver's size, when its Jacobi sweeps stalled, or when forced.  The caller
    // can tell: ctvio_marginalize_ran_on_host.
    marg_ran_on_host_ = 1;
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
  // ResidualSummary (reference trajectory_estimator.h:37-59): per-type sums of |r_i| at the current state
  int residual_summary(int id, double *sums, int32_t *counts4) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin || !sums) return fail(CTVIO_ERR_INVALID, "bad arguments");
    const WinMeta &m = meta_[id];
    const int n = 14 + m.pn;
    DBuf<double> out;
    HIPCHK(out.alloc(n));
    hipLaunchKernelGGL(k_residual_summary, dim3(1), dim3(256), (size_t)(14 + 2 * m.pn) * sizeof(double), stream_, dev_, id, out.p);
    HIPCHK(hipMemcpyAsync(sums, out.p, sizeof(double) * n, hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (counts4) { counts4[0] = m.M; counts4[1] = m.NB; counts4[2] = m.V; counts4[3] = m.pn > 0 ? 1 : 0; }
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
    // one grow-only scratch buffer and one staged copy: [ids | knot] as int32, then [q0 | t0] as doubles
    const size_t nbytes = (((size_t)2 * n * sizeof(int32_t) + 15) & ~(size_t)15) + (size_t)7 * n * sizeof(double);
    if (const int rc = call_scratch(nbytes)) return rc;
    char *hs = call_host_, *ds = call_dev_.p;
    const size_t ioff = ((size_t)2 * n * sizeof(int32_t) + 15) & ~(size_t)15;
    std::memcpy(hs, ids, sizeof(int32_t) * n); std::memcpy(hs + sizeof(int32_t) * n, knot, sizeof(int32_t) * n);
    std::memcpy(hs + ioff, q0, sizeof(double) * 4 * n); std::memcpy(hs + ioff + sizeof(double) * 4 * n, t0, sizeof(double) * 3 * n);
    HIPCHK(hipMemcpyAsync(ds, hs, nbytes, hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(k_gauge_restore, dim3(n), dim3(64), 0, stream_, dev_, n, reinterpret_cast<const int32_t *>(ds),
                       reinterpret_cast<const int32_t *>(ds) + n, reinterpret_cast<const double *>(ds + ioff), reinterpret_cast<const double *>(ds + ioff) + 4 * (size_t)n);
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    return CTVIO_OK;
  }
  int spline_eval(int id, int n, const int64_t *t_ns, double *pose7, double *vel3, double *omega3, double *acc3, const double *q_SI,
                  const double *p_SI) override {
    SensorExt ext{};
    if (q_SI && p_SI) {
      const double nq = std::sqrt(q_SI[0] * q_SI[0] + q_SI[1] * q_SI[1] + q_SI[2] * q_SI[2] + q_SI[3] * q_SI[3]);
      if (!(nq > 0.0) || !std::isfinite(nq)) return fail(CTVIO_ERR_INVALID, "sensor extrinsic: quaternion must be non-zero and finite");
      for (int i = 0; i < 4; ++i) ext.q[i] = q_SI[i] / nq;
      for (int i = 0; i < 3; ++i) ext.p[i] = p_SI[i];
      ext.on = 1;
    }
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (id < 0 || id >= dev_.nwin || n < 0 || (n && !t_ns)) return fail(CTVIO_ERR_INVALID, "bad arguments");
    if (n == 0) return CTVIO_OK;
    // grow-only scratch (pinned host mirror): [t_rel n x i64 | err | pose 7n | vel 3n | omega 3n | acc 3n]
    const size_t o_err = sizeof(long long) * (size_t)n, o_out = o_err + 16;
    const size_t nd = (size_t)n * ((pose7 ? 7 : 0) + (vel3 ? 3 : 0) + (omega3 ? 3 : 0) + (acc3 ? 3 : 0));
    if (const int rc = call_scratch(o_out + nd * sizeof(double))) return rc;
    char *hs = call_host_, *ds = call_dev_.p;
    long long *rel = reinterpret_cast<long long *>(hs);
    for (int i = 0; i < n; ++i) rel[i] = (long long)(t_ns[i] - t0_[id]);
    *reinterpret_cast<int *>(hs + o_err) = 0;
    HIPCHK(hipMemcpyAsync(ds, hs, o_out, hipMemcpyHostToDevice, stream_));
    double *dp = reinterpret_cast<double *>(ds + o_out), *dv = dp + (pose7 ? (size_t)7 * n : 0), *dw = dv + (vel3 ? (size_t)3 * n : 0),
           *da = dw + (omega3 ? (size_t)3 * n : 0);
    hipLaunchKernelGGL(k_spline_eval, dim3(nblk(n, 256)), dim3(256), 0, stream_, dev_, id, (const int32_t *)nullptr, n, reinterpret_cast<const long long *>(ds),
                       pose7 ? dp : nullptr, vel3 ? dv : nullptr, omega3 ? dw : nullptr, acc3 ? da : nullptr, reinterpret_cast<int *>(ds + o_err), ext);
    HIPCHK(hipMemcpyAsync(hs + o_err, ds + o_err, 16 + nd * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    const int err = *reinterpret_cast<int *>(hs + o_err);
    const double *ho = reinterpret_cast<const double *>(hs + o_out);
    if (pose7) { std::memcpy(pose7, ho, sizeof(double) * 7 * n); ho += (size_t)7 * n; }
    if (vel3) { std::memcpy(vel3, ho, sizeof(double) * 3 * n); ho += (size_t)3 * n; }
    if (omega3) { std::memcpy(omega3, ho, sizeof(double) * 3 * n); ho += (size_t)3 * n; }
    if (acc3) std::memcpy(acc3, ho, sizeof(double) * 3 * n);
    if (err) return fail(CTVIO_ERR_INVALID, "query time outside the spline");
    return CTVIO_OK;
  }
  // Queries of any windows of the batch in ONE launch (query i: window win[i], absolute time t_ns[i]).
  int spline_eval_batch(int64_t n64, const int32_t *win, const int64_t *t_ns, double *pose7, double *vel3, double *omega3, double *acc3,
                        double *kernel_ms) override {
    if (!uploaded_) return fail(CTVIO_ERR_STATE, "ctvio_upload not called");
    if (n64 < 0 || n64 > (int64_t)1 << 30 || (n64 && (!t_ns || !win))) return fail(CTVIO_ERR_INVALID, "bad arguments");
    const int n = (int)n64;
    if (kernel_ms) *kernel_ms = 0.0;
    if (n == 0) return CTVIO_OK;
    // grow-only scratch (pinned host mirror): [t_rel n x i64 | window n x i32 | err | pose 7n | vel 3n | omega 3n | acc 3n]
    const size_t o_win = sizeof(long long) * (size_t)n, o_err = (o_win + sizeof(int32_t) * (size_t)n + 15) & ~(size_t)15, o_out = o_err + 16;
    const size_t nd = (size_t)n * ((pose7 ? 7 : 0) + (vel3 ? 3 : 0) + (omega3 ? 3 : 0) + (acc3 ? 3 : 0));
    if (const int rc = call_scratch(o_out + nd * sizeof(double))) return rc;
    char *hs = call_host_, *ds = call_dev_.p;
    long long *rel = reinterpret_cast<long long *>(hs);
    int32_t *hw = reinterpret_cast<int32_t *>(hs + o_win);
    for (int i = 0; i < n; ++i) {
      if (win[i] < 0 || win[i] >= dev_.nwin) return fail(CTVIO_ERR_INVALID, "query " + std::to_string(i) + ": window id out of range");
      hw[i] = win[i];
      rel[i] = (long long)(t_ns[i] - t0_[win[i]]);
    }
    *reinterpret_cast<int *>(hs + o_err) = 0;
    HIPCHK(hipMemcpyAsync(ds, hs, o_out, hipMemcpyHostToDevice, stream_));
    double *dp = reinterpret_cast<double *>(ds + o_out), *dv = dp + (pose7 ? (size_t)7 * n : 0), *dw = dv + (vel3 ? (size_t)3 * n : 0),
           *da = dw + (omega3 ? (size_t)3 * n : 0);
    if (kernel_ms) HIPCHK(hipEventRecord(ev_[10], stream_));
    hipLaunchKernelGGL(k_spline_eval, dim3(nblk(n, 256)), dim3(256), 0, stream_, dev_, 0, reinterpret_cast<const int32_t *>(ds + o_win), n,
                       reinterpret_cast<const long long *>(ds), pose7 ? dp : nullptr, vel3 ? dv : nullptr, omega3 ? dw : nullptr, acc3 ? da : nullptr,
                       reinterpret_cast<int *>(ds + o_err), SensorExt{});
    if (kernel_ms) HIPCHK(hipEventRecord(ev_[11], stream_));
    HIPCHK(hipMemcpyAsync(hs + o_err, ds + o_err, 16 + nd * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipStreamSynchronize(stream_));
    HIPCHK(hipGetLastError());
    if (kernel_ms) { float ms = 0; HIPCHK(hipEventElapsedTime(&ms, ev_[10], ev_[11])); *kernel_ms = ms; }
    const int err = *reinterpret_cast<int *>(hs + o_err);
    const double *ho = reinterpret_cast<const double *>(hs + o_out);
    if (pose7) { std::memcpy(pose7, ho, sizeof(double) * 7 * n); ho += (size_t)7 * n; }
    if (vel3) { std::memcpy(vel3, ho, sizeof(double) * 3 * n); ho += (size_t)3 * n; }
    if (omega3) { std::memcpy(omega3, ho, sizeof(double) * 3 * n); ho += (size_t)3 * n; }
    if (acc3) std::memcpy(acc3, ho, sizeof(double) * 3 * n);
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
  const DebugSwitches dbg_;   // environment switches as they were when the handle was created
  int chol_tiles_ = 3;        // the uploaded batch's factorisation kernel (0 panel kernel, 1 / 3 register tiles): pack_and_upload
  int marg_ran_on_host_ = 0;  // the last ctvio_marginalize(_batch) call: 1 if the factorisation ran on the host
  hipStream_t stream_ = nullptr;
  hipEvent_t ev_[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool uploaded_ = false, profiling_ = false, profiling_requested_ = false;
  double timing_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_ms_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int32_t ph_n_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_iters_ = 0;
  std::vector<hipEvent_t> pev_;
  std::vector<int> pev_phase_;
  size_t pev_used_ = 0;
  std::vector<std::unique_ptr<HostWindow>> own_;   // windows recorded by ctvio_add_window (owning copies)
  std::vector<WinMeta> meta_;
  std::vector<int64_t> t0_;
  Dev dev_;
  int Mtot_ = 0, Vtot_ = 0;
  size_t chol_lds_ = 0, vis_lds_ = 0, vis_glb_ = 0, in_bytes_ = 0, state_doubles_ = 0;
  Arena in_, work_;          // uploaded inputs (pinned mirror) / device-only work buffers
  WorkerPool pool_;          // the handle's packing threads (created on first use, kept)
  // scratch of the small per-call entries (spline query, gauge restore): grow-only device buffer + pinned host mirror
  DBuf<char> call_dev_;
  char *call_host_ = nullptr; size_t call_host_cap_ = 0;
  int call_scratch(size_t bytes) {
    HIPCHK(call_dev_.alloc(bytes));
    if (bytes > call_host_cap_) {
      if (call_host_) (void)hipHostFree(call_host_);
      call_host_cap_ = bytes + bytes / 4 + 4096;
      HIPCHK(hipHostMalloc((void **)&call_host_, call_host_cap_, hipHostMallocDefault));
    }
    return CTVIO_OK;
  }
  DBuf<MargMeta> mg_meta_;   // device marginalisation: descriptors, index lists, scratch, outputs
  DBuf<int32_t> mg_idx_;
  DBuf<double> mg_scr_, mg_out_;
  bool marg_attr_set_ = false;
  double *snap_ = nullptr;   // state snapshot (inside work_)
  Lm *lm_host_ = nullptr; size_t lm_host_cap_ = 0;
  int graph_captures_ = 0;                // how many times the pass was captured (ctvio_graph_captures: a stream of equal batches captures once)
  hipGraphExec_t graph_exec_ = nullptr;   // one LM pass (launch_pass) as a graph, valid while dev_ == graph_dev_
  Dev graph_dev_;
  std::vector<long long> graph_sig_;
  bool deterministic_ = false;   // order-fixed accumulation for this batch (ctvio_options.deterministic)
  double *state_host_ = nullptr; size_t state_host_cap_ = 0;
  bool snap_valid_ = false, any_vis_lds_ = false, any_vis_glb_ = false, all_windows_have_imu_ = false;
  int maxK_ = 0, max_schur_tiles_ = 0;
  const double *h_ld_ = nullptr;        // the line delays as uploaded (same arena)
  const int32_t *h_lm_pos_ = nullptr;   // host mirror of Dev::lm_pos (inside in_.host: valid while the batch is uploaded)
};

void SolverImpl::launch_imu_linearize(int mode) {
  const Dev &d = dev_;
  // (at most 2048 waves -- two rounds of one wave per SIMD -- each walking its share of the groups with the next group's data in flight)
  hipLaunchKernelGGL(k_imu_linearize_f64, dim3(std::min(d.Gtot, imu_walk_waves())), dim3(64), (size_t)(72 * 33 + 64) * sizeof(double), stream_, d, mode, imu_general_only(), imu_zero_mode());
  hipLaunchKernelGGL(k_imu_linearize_rest, dim3(d.nwin), dim3(64), (size_t)64 * 33 * sizeof(double), stream_, d, mode, imu_general_only(), imu_zero_mode());
}
void SolverImpl::launch_assemble_vis_lds(int parts, int mode) {
  const Dev &d = dev_;
  hipLaunchKernelGGL((k_assemble_vis_mfma<VCH, true>), dim3(d.nwin, parts), dim3(512), vis_lds_, stream_, d, mode);
}
// windows whose packed Hessian does not fit in LDS (K > 25): run products on the MFMA units, added to Hpp with global atomics
void SolverImpl::launch_assemble_vis_glb(int parts, int mode) {
  const Dev &d = dev_;
  hipLaunchKernelGGL((k_assemble_vis_mfma<VCH, false>), dim3(d.nwin, parts), dim3(512), vis_glb_, stream_, d, mode);
}
void SolverImpl::launch_schur() {
  const Dev &d = dev_;
  // large batches: one workgroup per window, W staged through LDS once; small batches: one wave per 16 x 16 tile (shorter latency, W re-read
  // per tile).  Every variant also produces the reduced right-hand side (g_rho rides as column P).
  const int nc = 6 * maxK_ + 2;   // compact columns of W per landmark: knots, line delay, g_rho
  if (schur_window_path()) {
    const size_t lds = schur_window_lds();
    if (16 * nc <= 5 * 512 && max_schur_tiles_ <= 56) hipLaunchKernelGGL((k_schur_window_f64<5, 7>), dim3(d.nwin), dim3(512), lds, stream_, d);
    else if (16 * nc <= 5 * 512) hipLaunchKernelGGL((k_schur_window_f64<5, 14>), dim3(d.nwin), dim3(512), lds, stream_, d);
    else hipLaunchKernelGGL((k_schur_window_f64<7, 14>), dim3(d.nwin), dim3(512), lds, stream_, d);
    return;
  }
  const int nt2 = d.maxP / 16 + 1, ntile2 = nt2 * (nt2 + 1) / 2;   // tile rows up to index P (the rhs row)
  const int nb2 = (nt2 + 1) / 2, nblk2 = nb2 * (nb2 + 1) / 2;      // 32 x 32 blocks of the lower triangle
  // enough tiles to fill the chip several times over (config 5: 666 per window): one wave per 2 x 2 tiles, half the operand loads
  // per product; otherwise one wave per tile (more waves in flight).  DebugSwitches::schur_tile2 = 0 / 1 forces the choice (A/B).
  const int force2 = dbg_.schur_tile2;
  const bool tile2 = force2 >= 0 ? force2 != 0 : (long long)d.nwin * ntile2 >= 16384;
  if (tile2) hipLaunchKernelGGL(k_schur_tile2_f64, dim3(nblk2 * 8 * ((d.nwin + 7) / 8)), dim3(64), 0, stream_, d, nblk2);
  else hipLaunchKernelGGL(k_schur_tile_f64, dim3(ntile2 * 8 * ((d.nwin + 7) / 8)), dim3(64), 0, stream_, d, ntile2);
}

}  // namespace ctv

// ================================================================================================ C ABI
struct ctvio_solver { std::unique_ptr<ctv::SolverBase> impl; };

extern "C" {

void ctvio_default_options(ctvio_options *o) {
  if (!o) return;
  std::memset(o, 0, sizeof *o);
  o->device = 0; o->precision = CTVIO_FP64; o->use_mfma = 1; o->check_every = 4;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
  o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32; o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32; o->max_consecutive_invalid_steps = 5;
  o->deterministic = -1; o->host_threads = 0; o->use_graph = 1; o->line_search = 1;
}
const char *ctvio_status_string(int32_t s) {
  switch (s) {
    case CTVIO_OK: return "ok";
    case CTVIO_ERR_INVALID: return "invalid argument";
    case CTVIO_ERR_NO_DEVICE: return "no HIP device (the product path has no CPU fallback)";
    case CTVIO_ERR_HIP: return "HIP runtime error";
    case CTVIO_ERR_STATE: return "call order violated";
    case CTVIO_ERR_INTERNAL: return "internal consistency check failed";
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
  if (o.precision != CTVIO_FP64) return ctv::fail(CTVIO_ERR_INVALID, "precision: only CTVIO_FP64 exists (the mixed fp32 mode was removed: it missed the 1e-4 contract)");
  if (o.use_mfma != 1 && o.use_mfma != 2) return ctv::fail(CTVIO_ERR_INVALID, "use_mfma: 1 or 2 (the vector-ALU cross-check kernels of use_mfma = 0 were removed: the oracle is the cross-check)");
  { auto *p = new ctv::SolverImpl(o); s->impl.reset(p); rc = p->init(); }
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
int32_t ctvio_set_batch(ctvio_solver *s, int32_t n, const ctvio_window *wins) { CHK_S; return s->impl->set_batch(n, wins); }
int32_t ctvio_num_windows(const ctvio_solver *s) { return s ? s->impl->num_windows() : 0; }
int32_t ctvio_solve(ctvio_solver *s, int32_t max_iterations, ctvio_summary *out) { CHK_S; return s->impl->solve(max_iterations, out); }
int32_t ctvio_get_state(ctvio_solver *s, int32_t id, double *quat, double *pos, double *bias, double *rho, double *ld) {
  CHK_S; return s->impl->get_state(id, quat, pos, bias, rho, ld);
}
int32_t ctvio_get_batch_state(ctvio_solver *s, double *quat, double *pos, double *bias, double *rho, double *ld) {
  CHK_S; return s->impl->get_batch_state(quat, pos, bias, rho, ld);
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
int32_t ctvio_residual_summary(ctvio_solver *s, int32_t id, double *sums, int32_t *counts4) { CHK_S; return s->impl->residual_summary(id, sums, counts4); }
int32_t ctvio_marginalize_batch(ctvio_solver *s, const int8_t *role, double eps, int32_t *n_keep, int32_t *kept, double *J0, double *r0) {
  CHK_S; return s->impl->marginalize_batch(role, eps, n_keep, kept, J0, r0);
}
int32_t ctvio_gauge_restore(ctvio_solver *s, int32_t n, const int32_t *ids, const int32_t *knot, const double *q0, const double *t0) {
  CHK_S; return s->impl->gauge_restore(n, ids, knot, q0, t0);
}
int32_t ctvio_spline_eval(ctvio_solver *s, int32_t id, int32_t n, const int64_t *t_ns, double *pose7, double *vel3, double *omega3, double *acc3) {
  CHK_S; return s->impl->spline_eval(id, n, t_ns, pose7, vel3, omega3, acc3);
}
int32_t ctvio_spline_eval_batch(ctvio_solver *s, int64_t n, const int32_t *win, const int64_t *t_ns, double *pose7, double *vel3, double *omega3,
                                double *acc3, double *kernel_ms) {
  CHK_S; return s->impl->spline_eval_batch(n, win, t_ns, pose7, vel3, omega3, acc3, kernel_ms);
}
int32_t ctvio_sensor_pose(ctvio_solver *s, int32_t id, int32_t n, const int64_t *t_ns, const double *q_SI, const double *p_SI, double *pose7) {
  CHK_S;
  if (!q_SI || !p_SI || (n && !pose7)) return ctv::fail(CTVIO_ERR_INVALID, "ctvio_sensor_pose: null argument");
  return s->impl->spline_eval(id, n, t_ns, pose7, nullptr, nullptr, nullptr, q_SI, p_SI);
}
// ---- multi-device host entry: w mod G, one host thread + solver handle per device
int32_t ctvio_shard_of(int32_t window_id, int32_t n_devices) { return n_devices > 0 ? window_id % n_devices : 0; }
int32_t ctvio_shard_count(int32_t n, int32_t device, int32_t n_devices) {
  if (n_devices <= 0 || device < 0 || device >= n_devices || n <= 0) return 0;
  return n / n_devices + (device < n % n_devices ? 1 : 0);
}
namespace {
// (memcmp over the struct would compare its tail padding: indeterminate bytes of a caller's stack object)
bool same_options(const ctvio_options &a, const ctvio_options &b) {
  return a.device == b.device && a.precision == b.precision && a.use_mfma == b.use_mfma && a.check_every == b.check_every &&
         a.function_tolerance == b.function_tolerance && a.gradient_tolerance == b.gradient_tolerance && a.parameter_tolerance == b.parameter_tolerance &&
         a.initial_radius == b.initial_radius && a.max_radius == b.max_radius && a.min_radius == b.min_radius &&
         a.min_relative_decrease == b.min_relative_decrease && a.min_lm_diagonal == b.min_lm_diagonal && a.max_lm_diagonal == b.max_lm_diagonal &&
         a.max_consecutive_invalid_steps == b.max_consecutive_invalid_steps && a.deterministic == b.deterministic && a.host_threads == b.host_threads &&
         a.use_graph == b.use_graph && a.line_search == b.line_search;
}
std::mutex g_shard_mu;
std::vector<ctvio_solver *> g_shard_solvers;   // one per shard, created on first use (and again when the options change)
std::vector<ctvio_options> g_shard_opts;       // the options each handle was created with
// One persistent host thread per shard (shard 0 runs on the caller's thread): a call posts its per-shard job and waits -- no thread is
// created or joined per call.  The threads live until ctvio_sharded_release.
struct ShardWorker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, done = true, quit = false;
  ~ShardWorker() { stop(); }   // (a process that never calls ctvio_sharded_release: the idle thread is told to quit and joined at exit)
  void loop() {
    for (;;) {
      std::function<void()> f;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return has_job || quit; });
        if (quit) return;
        f = std::move(job);
        has_job = false;
      }
      f();
      { std::lock_guard<std::mutex> lk(mu); done = true; }
      cv.notify_all();
    }
  }
  void post(std::function<void()> f) {
    { std::lock_guard<std::mutex> lk(mu); job = std::move(f); has_job = true; done = false; }
    cv.notify_all();
  }
  void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
  void stop() {
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};
std::vector<std::unique_ptr<ShardWorker>> g_shard_workers;   // [g - 1] for shard g >= 1
}
// The number of shards ctvio_solve_sharded uses: min(requested or all devices, devices present, windows).  With the TEST-ONLY
// environment switch CTVIO_SHARD_OVERSUBSCRIBE=1 the device count does not clamp it (shard g runs on device g mod #devices), so that
// the multi-shard path can be exercised on a box with one GPU.
int32_t ctvio_shards_used(int32_t n_devices, int32_t n) {
  const int ndev = ctvio_device_count();
  if (ndev <= 0 || n <= 0) return 0;
  if (ctv::read_debug_switches().shard_oversubscribe && n_devices > 0) return std::min(n_devices, n);
  return std::min(n_devices > 0 ? std::min(n_devices, ndev) : ndev, n);
}
void ctvio_sharded_release(void) {
  std::lock_guard<std::mutex> lk(g_shard_mu);
  for (auto &wk : g_shard_workers) wk->stop();
  g_shard_workers.clear();
  for (auto *sv : g_shard_solvers) if (sv) ctvio_destroy(sv);
  g_shard_solvers.clear();
  g_shard_opts.clear();
}
int32_t ctvio_solve_sharded(const ctvio_options *opt, int32_t n_devices, int32_t n, const ctvio_window *wins, int32_t max_iterations,
                            ctvio_summary *out, double *quat, double *pos, double *bias, double *rho, double *ld) {
  if (n <= 0 || !wins) return ctv::fail(CTVIO_ERR_INVALID, "empty batch");
  const int ndev = ctvio_device_count();
  if (ndev <= 0) return ctv::fail(CTVIO_ERR_NO_DEVICE, "hipGetDeviceCount found no device");
  const int G = ctvio_shards_used(n_devices, n);
  std::lock_guard<std::mutex> lk(g_shard_mu);   // one sharded solve at a time per process (the handles are shared)
  if ((int)g_shard_solvers.size() < G) { g_shard_solvers.resize((size_t)G, nullptr); g_shard_opts.resize((size_t)G); }
  // offsets of every window in the caller's concatenated state arrays
  std::vector<size_t> k0((size_t)n + 1, 0), f0((size_t)n + 1, 0), l0((size_t)n + 1, 0);
  for (int i = 0; i < n; ++i) { k0[i + 1] = k0[i] + (size_t)std::max(wins[i].K, 0); f0[i + 1] = f0[i] + (size_t)std::max(wins[i].F, 0); l0[i + 1] = l0[i] + (size_t)std::max(wins[i].L, 0); }
  std::vector<int> rcs((size_t)G, CTVIO_OK);
  std::vector<std::string> errs((size_t)G);
  auto work_body = [&](int g) {
    auto failed = [&](int rc) { rcs[g] = rc; errs[g] = "shard " + std::to_string(g) + ": " + ctvio_last_error(); };   // (thread-local error text)
    ctvio_options o;
    if (opt) o = *opt; else ctvio_default_options(&o);
    o.device = g % ndev;
    if (g_shard_solvers[g] && !same_options(o, g_shard_opts[g])) {   // the caller changed the options: a fresh handle
      ctvio_destroy(g_shard_solvers[g]);
      g_shard_solvers[g] = nullptr;
    }
    if (!g_shard_solvers[g]) {
      if (const int rc = ctvio_create(&o, &g_shard_solvers[g])) return failed(rc);
      g_shard_opts[g] = o;
    }
    ctvio_solver *sv = g_shard_solvers[g];
    std::vector<ctvio_window> mine;
    std::vector<int> ids;
    for (int i = g; i < n; i += G) { mine.push_back(wins[i]); ids.push_back(i); }
    const int nm = (int)mine.size();
    if (const int rc = ctvio_set_batch(sv, nm, mine.data())) return failed(rc);
    std::vector<ctvio_summary> sm((size_t)nm);
    if (const int rc = ctvio_solve(sv, max_iterations, sm.data())) return failed(rc);
    size_t K = 0, F = 0, L = 0;
    for (const auto &w : mine) { K += w.K; F += w.F; L += w.L; }
    std::vector<double> q(4 * K), p(3 * K), b(6 * F), r(std::max<size_t>(L, 1)), l((size_t)nm);
    if (const int rc = ctvio_get_batch_state(sv, q.data(), p.data(), b.data(), r.data(), l.data())) return failed(rc);
    size_t ka = 0, fa = 0, la = 0;
    for (int j = 0; j < nm; ++j) {
      const int i = ids[j];
      const ctvio_window &w = mine[j];
      if (out) out[i] = sm[j];
      if (quat) std::memcpy(quat + 4 * k0[i], q.data() + 4 * ka, sizeof(double) * 4 * w.K);
      if (pos) std::memcpy(pos + 3 * k0[i], p.data() + 3 * ka, sizeof(double) * 3 * w.K);
      if (bias) std::memcpy(bias + 6 * f0[i], b.data() + 6 * fa, sizeof(double) * 6 * w.F);
      if (rho && w.L) std::memcpy(rho + l0[i], r.data() + la, sizeof(double) * w.L);
      if (ld) ld[i] = l[j];
      ka += w.K; fa += w.F; la += w.L;
    }
  };
  auto work = [&](int g) {   // (an exception escaping a std::thread would terminate the process)
    try { work_body(g); }
    catch (const std::exception &e) { rcs[g] = CTVIO_ERR_HIP; errs[g] = "shard " + std::to_string(g) + ": " + e.what(); }
    catch (...) { rcs[g] = CTVIO_ERR_HIP; errs[g] = "shard " + std::to_string(g) + ": unknown exception"; }
  };
  while ((int)g_shard_workers.size() < G - 1) {
    g_shard_workers.emplace_back(new ShardWorker());
    ShardWorker *wk = g_shard_workers.back().get();
    wk->th = std::thread([wk] { wk->loop(); });
  }
  // (jobs hold references to this frame: whatever was posted is waited for before the function unwinds, also when a later post throws)
  int posted = 0;
  try {
    for (int g = 1; g < G; ++g) { g_shard_workers[g - 1]->post([&work, g] { work(g); }); posted = g; }
    work(0);
  } catch (...) {
    for (int g = 1; g <= posted; ++g) g_shard_workers[g - 1]->wait();
    return ctv::fail(CTVIO_ERR_HIP, "ctvio_solve_sharded: could not hand a shard to its worker thread");
  }
  for (int g = 1; g < G; ++g) g_shard_workers[g - 1]->wait();
  for (int g = 0; g < G; ++g) if (rcs[g] != CTVIO_OK) return ctv::fail(rcs[g], errs[g]);
  return CTVIO_OK;
}

int32_t ctvio_last_timing(ctvio_solver *s, double *ms8, int32_t *launches8) { CHK_S; return s->impl->last_timing(ms8, launches8); }
int32_t ctvio_snapshot_state(ctvio_solver *s) { CHK_S; return s->impl->snapshot(0); }
int32_t ctvio_restore_state(ctvio_solver *s) { CHK_S; return s->impl->snapshot(1); }
int32_t ctvio_set_profiling(ctvio_solver *s, int32_t on) { CHK_S; return s->impl->set_profiling(on); }
void *ctvio_stream(ctvio_solver *s) { return s ? s->impl->stream() : nullptr; }
int32_t ctvio_graph_captures(const ctvio_solver *s) { return s ? s->impl->graph_captures() : 0; }
int32_t ctvio_marginalize_ran_on_host(const ctvio_solver *s) { return s ? s->impl->marg_ran_on_host() : 0; }

}  // extern "C"
