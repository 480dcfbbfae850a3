// ctvio_window_io.hpp -- header-only reader / writer of batches of ctvio_window in a flat binary file, so that C / C++ callers can run the SAME
// synthetic windows bench.py and the tests use (the generator lives in ctrl-vio_amd/synth.py -- NumPy's Philox streams
// are not reproducible from C++ -- and `python tools/export_windows.py config2 1000 64 out.ctvw` writes them in this format;
// ctrl-vio_amd/window_io.py is the Python twin).  A window here is what TrajectoryManager::UpdateTrajectory hands a fresh TrajectoryEstimator
// (reference src/estimator/trajectory_manager.cpp:331-451), addressed by index: include/ctvio.h, struct ctvio_window.
//
// File: "CTVW0001" | int32 n | n records.  Record: int32 K F L M NB V pn pnb | int64 t0_ns dt_ns | double ld ld_lo ld_hi |
// int32 fix_ld lock_bg lock_ba fixed_upto | double q_CI[4] p_CI[3] gravity[3] imu_w[6] img_w cauchy_a | int32 has_v_cauchy has_knot_const |
// the arrays in the order of the struct: quat pos bias rho | imu_t imu_gyro imu_acc imu_bias | bc_i bc_j bc_w | v_lm v_ti v_tj v_rowi v_rowj v_pi
// v_pj | pJ0 (column-major) pr0 p_kind p_index p_off p_x0 | [v_cauchy] [knot_const].  Little-endian, no padding.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "ctvio.h"

namespace ctvio {

// A window that owns its arrays (the ctvio_window inside points into them).
struct OwnedWindow {
  ctvio_window w{};
  std::vector<double> quat, pos, bias, rho, imu_gyro, imu_acc, bc_w, v_pi, v_pj, pJ0, pr0, p_x0, v_cauchy;
  std::vector<int64_t> imu_t, v_ti, v_tj;
  std::vector<int32_t> imu_bias, bc_i, bc_j, v_lm, v_rowi, v_rowj, p_kind, p_index, p_off;
  std::vector<uint8_t> knot_const;
  void bind() {
    w.quat = quat.data(); w.pos = pos.data(); w.bias = bias.data(); w.rho = rho.data();
    w.imu_t = imu_t.data(); w.imu_gyro = imu_gyro.data(); w.imu_acc = imu_acc.data(); w.imu_bias = imu_bias.data();
    w.bc_i = bc_i.data(); w.bc_j = bc_j.data(); w.bc_w = bc_w.data();
    w.v_lm = v_lm.data(); w.v_ti = v_ti.data(); w.v_tj = v_tj.data(); w.v_rowi = v_rowi.data(); w.v_rowj = v_rowj.data();
    w.v_pi = v_pi.data(); w.v_pj = v_pj.data();
    w.pJ0 = pJ0.data(); w.pr0 = pr0.data(); w.p_kind = p_kind.data(); w.p_index = p_index.data(); w.p_off = p_off.data(); w.p_x0 = p_x0.data();
    w.v_cauchy = v_cauchy.empty() ? nullptr : v_cauchy.data();
    w.knot_const = knot_const.empty() ? nullptr : knot_const.data();
  }
};

namespace detail {
template <class U> inline bool rd(std::FILE *f, U *p, size_t n) { return n == 0 || std::fread(p, sizeof(U), n, f) == n; }
template <class U> inline bool rdv(std::FILE *f, std::vector<U> &v, size_t n) { v.resize(n); return rd(f, v.data(), n); }
template <class U> inline bool wr(std::FILE *f, const U *p, size_t n) { return n == 0 || std::fwrite(p, sizeof(U), n, f) == n; }
}  // namespace detail

// Reads every window of `path`.  Returns an empty string on success, a message otherwise.
inline std::string load_windows(const char *path, std::vector<std::unique_ptr<OwnedWindow>> &out) {
  using namespace detail;
  std::FILE *f = std::fopen(path, "rb");
  if (!f) return std::string("cannot open ") + path;
  struct Closer { std::FILE *f; ~Closer() { std::fclose(f); } } closer{f};
  char magic[8];
  int32_t n = 0;
  if (!rd(f, magic, 8) || std::memcmp(magic, "CTVW0001", 8) != 0 || !rd(f, &n, 1) || n < 0) return "not a CTVW0001 file";
  for (int i = 0; i < n; ++i) {
    std::unique_ptr<OwnedWindow> o(new OwnedWindow);
    ctvio_window &w = o->w;
    int32_t sz[8], fl[4], has[2];
    int64_t tt[2];
    double ld3[3], cal[18];
    if (!rd(f, sz, 8) || !rd(f, tt, 2) || !rd(f, ld3, 3) || !rd(f, fl, 4) || !rd(f, cal, 18) || !rd(f, has, 2)) return "truncated header";
    w.K = sz[0]; w.F = sz[1]; w.L = sz[2]; w.M = sz[3]; w.NB = sz[4]; w.V = sz[5]; w.pn = sz[6]; w.pnb = sz[7];
    for (int k = 0; k < 8; ++k) if (sz[k] < 0 || sz[k] > (1 << 24)) return "implausible sizes";
    w.t0_ns = tt[0]; w.dt_ns = tt[1]; w.ld = ld3[0]; w.ld_lo = ld3[1]; w.ld_hi = ld3[2];
    w.fix_ld = fl[0]; w.lock_bg = fl[1]; w.lock_ba = fl[2]; w.fixed_upto = fl[3];
    std::memcpy(w.q_CI, cal, 4 * 8); std::memcpy(w.p_CI, cal + 4, 3 * 8); std::memcpy(w.gravity, cal + 7, 3 * 8); std::memcpy(w.imu_w, cal + 10, 6 * 8);
    w.img_w = cal[16]; w.cauchy_a = cal[17];
    const size_t K = w.K, F = w.F, L = w.L, M = w.M, NB = w.NB, V = w.V, pn = w.pn, pnb = w.pnb;
    bool ok = rdv(f, o->quat, 4 * K) && rdv(f, o->pos, 3 * K) && rdv(f, o->bias, 6 * F) && rdv(f, o->rho, L) &&
              rdv(f, o->imu_t, M) && rdv(f, o->imu_gyro, 3 * M) && rdv(f, o->imu_acc, 3 * M) && rdv(f, o->imu_bias, M) &&
              rdv(f, o->bc_i, NB) && rdv(f, o->bc_j, NB) && rdv(f, o->bc_w, 6 * NB) &&
              rdv(f, o->v_lm, V) && rdv(f, o->v_ti, V) && rdv(f, o->v_tj, V) && rdv(f, o->v_rowi, V) && rdv(f, o->v_rowj, V) &&
              rdv(f, o->v_pi, 2 * V) && rdv(f, o->v_pj, 2 * V) &&
              rdv(f, o->pJ0, pn * pn) && rdv(f, o->pr0, pn) && rdv(f, o->p_kind, pnb) && rdv(f, o->p_index, pnb) && rdv(f, o->p_off, pnb) && rdv(f, o->p_x0, 4 * pnb);
    if (ok && has[0]) ok = rdv(f, o->v_cauchy, V);
    if (ok && has[1]) ok = rdv(f, o->knot_const, K);
    if (!ok) return "truncated window " + std::to_string(i);
    o->bind();
    out.push_back(std::move(o));
  }
  return std::string();
}

inline std::string save_windows(const char *path, const ctvio_window *wins, int32_t n) {
  using namespace detail;
  std::FILE *f = std::fopen(path, "wb");
  if (!f) return std::string("cannot create ") + path;
  struct Closer { std::FILE *f; ~Closer() { std::fclose(f); } } closer{f};
  bool ok = wr(f, "CTVW0001", 8) && wr(f, &n, 1);
  for (int i = 0; ok && i < n; ++i) {
    const ctvio_window &w = wins[i];
    const int32_t sz[8] = {w.K, w.F, w.L, w.M, w.NB, w.V, w.pn, w.pnb}, fl[4] = {w.fix_ld, w.lock_bg, w.lock_ba, w.fixed_upto};
    const int32_t has[2] = {w.v_cauchy ? 1 : 0, w.knot_const ? 1 : 0};
    const int64_t tt[2] = {w.t0_ns, w.dt_ns};
    const double ld3[3] = {w.ld, w.ld_lo, w.ld_hi};
    double cal[18];
    std::memcpy(cal, w.q_CI, 4 * 8); std::memcpy(cal + 4, w.p_CI, 3 * 8); std::memcpy(cal + 7, w.gravity, 3 * 8); std::memcpy(cal + 10, w.imu_w, 6 * 8);
    cal[16] = w.img_w; cal[17] = w.cauchy_a;
    const size_t K = w.K, F = w.F, L = w.L, M = w.M, NB = w.NB, V = w.V, pn = w.pn, pnb = w.pnb;
    ok = wr(f, sz, 8) && wr(f, tt, 2) && wr(f, ld3, 3) && wr(f, fl, 4) && wr(f, cal, 18) && wr(f, has, 2) &&
         wr(f, w.quat, 4 * K) && wr(f, w.pos, 3 * K) && wr(f, w.bias, 6 * F) && wr(f, w.rho, L) &&
         wr(f, w.imu_t, M) && wr(f, w.imu_gyro, 3 * M) && wr(f, w.imu_acc, 3 * M) && wr(f, w.imu_bias, M) &&
         wr(f, w.bc_i, NB) && wr(f, w.bc_j, NB) && wr(f, w.bc_w, 6 * NB) &&
         wr(f, w.v_lm, V) && wr(f, w.v_ti, V) && wr(f, w.v_tj, V) && wr(f, w.v_rowi, V) && wr(f, w.v_rowj, V) && wr(f, w.v_pi, 2 * V) && wr(f, w.v_pj, 2 * V) &&
         wr(f, w.pJ0, pn * pn) && wr(f, w.pr0, pn) && wr(f, w.p_kind, pnb) && wr(f, w.p_index, pnb) && wr(f, w.p_off, pnb) && wr(f, w.p_x0, 4 * pnb) &&
         (!has[0] || wr(f, w.v_cauchy, V)) && (!has[1] || wr(f, w.knot_const, K));
  }
  return ok ? std::string() : std::string("write failed: ") + path;
}

}  // namespace ctvio
