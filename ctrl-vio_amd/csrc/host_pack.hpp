// host_pack.hpp -- host side of a batch upload: validation and packing of the caller's windows (the factor set of the
// reference's TrajectoryManager::UpdateTrajectory, src/estimator/trajectory_manager.cpp:331-451) into ONE pinned staging
// arena that mirrors the device input arena byte for byte, so that a whole batch reaches HBM with a single
// hipMemcpyAsync.  Windows are independent: both passes (validate + count, then fill) run over the windows with a pool
// of host threads.  No device code in this file.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ctvio.h"
#include "device_types.hpp"

namespace ctv {

inline int host_threads(int requested) {
  if (requested > 0) return requested;
  const unsigned hw = std::thread::hardware_concurrency();
  return (int)std::max(1u, std::min(hw ? hw : 1u, 16u));
}

// f(i) for i in [0, n), dynamic distribution over `nthreads` threads (the calling thread included).  The threads belong to the solver handle and
// live as long as it does: a batch is packed in two passes, and a pass that creates and joins its threads pays for them every time -- for a batch of 8
// windows (BASELINE configs[3] as written: 8 windows per GPU) that was more than the packing itself.
class WorkerPool {
 public:
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  template <class F> void run(int n, int nthreads, F &&f) {
    if (nthreads <= 1 || n <= 1) { for (int i = 0; i < n; ++i) f(i); return; }
    const int extra = std::min(nthreads, n) - 1;
    while ((int)th_.size() < extra) th_.emplace_back([this] { loop(); });
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = [&f](int i) { f(i); };
      n_ = n; next_.store(0, std::memory_order_relaxed); wanted_ = extra; started_ = 0; running_ = 0; ++epoch_;
    }
    cv_.notify_all();
    work();                                  // the calling thread takes items too
    std::unique_lock<std::mutex> lk(mu_);    // every helper that picked the job up has finished its last item (helpers that never woke take none)
    wanted_ = 0;
    done_.wait(lk, [&] { return running_ == 0; });
    job_ = nullptr;
  }

 private:
  void work() { for (;;) { const int i = next_.fetch_add(1, std::memory_order_relaxed); if (i >= n_) break; job_(i); } }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return quit_ || (epoch_ != seen && started_ < wanted_); });
        if (quit_) return;
        seen = epoch_; ++started_; ++running_;
      }
      work();
      { std::lock_guard<std::mutex> lk(mu_); --running_; }
      done_.notify_all();
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::function<void(int)> job_;
  std::atomic<int> next_{0};
  int n_ = 0, wanted_ = 0, started_ = 0, running_ = 0;
  unsigned long long epoch_ = 0;
  bool quit_ = false;
};

// Grow-only device buffer, optionally mirrored by pinned host memory of the same size.
struct Arena {
  char *dev = nullptr, *host = nullptr;
  size_t cap = 0;
  bool mirrored = false;
  ~Arena() { release(); }
  void release() {
    if (dev) (void)hipFree(dev);
    if (host) (void)hipHostFree(host);
    dev = host = nullptr; cap = 0;
  }
  // returns hipSuccess; *grew tells the caller that the contents (and every pointer into the arena) are gone
  hipError_t reserve(size_t bytes, bool mirror, bool *grew) {
    if (grew) *grew = false;
    if (bytes <= cap && dev && (!mirror || host)) return hipSuccess;
    release();
    const size_t want = std::max<size_t>(bytes + bytes / 8, 4096);   // headroom: a slightly larger next batch does not reallocate
    hipError_t e = hipMalloc((void **)&dev, want);
    if (e != hipSuccess) { dev = nullptr; return e; }
    if (mirror) {
      e = hipHostMalloc((void **)&host, want, hipHostMallocDefault);
      if (e != hipSuccess) { host = nullptr; release(); return e; }
    }
    cap = want; mirrored = mirror;
    if (grew) *grew = true;
    return hipSuccess;
  }
};

static inline int prior_block_size(int kind) { return kind == CTVIO_PK_LD ? 1 : 3; }

// Per-window results of the first pass.
struct PackTmp {
  std::vector<int32_t> iorder, iseg;    // IMU: sample order by (segment, bias state); segment of every sample (input order)
  std::vector<int32_t> vord;            // visual blocks: order by (ti, tj, rowi, rowj)  (frame-pair order: the assembly's items)
  std::vector<int32_t> lord, vpos;      // landmark-major slots: lord[slot] = block or -1 (padding), vpos[block] = slot
  std::vector<int32_t> anc_of, anc_rep; // anchors (distinct i ends), numbered landmark-major: anchor of every block / a block that carries it
  int32_t ngrp = 0, nvitem = 0, Vp = 0; // Vp: slots incl. padding, a multiple of 64
  int32_t A = 0;                        // number of anchors
  // sparsity plan (plan_sparsity): rows of W in sorted landmark order, their knot spans, per-tile row ranges, envelope of the reduced system
  std::vector<int32_t> lm_pos, lm_at, row_klo, row_khi, tl_beg, tl_end, env_first, env_tile;
  int32_t Lobs = 0, max_span = 0, ntr = 0;
  std::string err;
};

// reference asserts / prints when a time falls outside the spline (spline_segment.h:74-81); here every input is checked
inline bool validate_window(const ctvio_window *w, std::string &err) {
  auto bad = [&](const char *m) { err = m; return false; };
  if (!w) return bad("null window");
  if (w->K < 4 || w->F < 1 || w->L < 0 || w->M < 0 || w->NB < 0 || w->V < 0 || w->dt_ns <= 0 || w->pn < 0 || w->pnb < 0)
    return bad("bad sizes (need K >= 4, F >= 1, dt_ns > 0)");
  if (!w->quat || !w->pos || !w->bias || (w->L && !w->rho)) return bad("null state pointer");
  if (!std::isfinite(w->ld)) return bad("non-finite line delay");
  if (!w->fix_ld && !(w->ld_lo <= w->ld_hi && std::isfinite(w->ld_lo) && std::isfinite(w->ld_hi))) return bad("line-delay bounds: need finite ld_lo <= ld_hi");
  if (w->M && (!w->imu_t || !w->imu_gyro || !w->imu_acc || !w->imu_bias)) return bad("null IMU pointer");
  if (w->NB && (!w->bc_i || !w->bc_j || !w->bc_w)) return bad("null bias-chain pointer");
  if (w->V && (!w->v_lm || !w->v_ti || !w->v_tj || !w->v_rowi || !w->v_rowj || !w->v_pi || !w->v_pj)) return bad("null visual pointer");
  const int64_t tmax = w->t0_ns + (int64_t)(w->K - 3) * w->dt_ns;
  for (int m = 0; m < w->M; ++m) {
    if (w->imu_t[m] < w->t0_ns || w->imu_t[m] >= tmax) return bad("IMU time outside the spline");
    if (w->imu_bias[m] < 0 || w->imu_bias[m] >= w->F) return bad("IMU bias index out of range");
  }
  const int64_t ldmax_ns = (int64_t)((w->fix_ld ? w->ld : std::max(w->ld, w->ld_hi)) * 1e9);
  const int64_t ldmin_ns = (int64_t)((w->fix_ld ? w->ld : std::min(w->ld, w->ld_lo)) * 1e9);
  for (int v = 0; v < w->V; ++v) {
    if (w->v_lm[v] < 0 || w->v_lm[v] >= w->L) return bad("visual landmark index out of range");
    if (w->v_rowi[v] < 0 || w->v_rowj[v] < 0) return bad("negative image row");
    const int64_t a = w->v_ti[v], b = w->v_tj[v];
    if (a < w->t0_ns || b < w->t0_ns || a + w->v_rowi[v] * ldmax_ns >= tmax || b + w->v_rowj[v] * ldmax_ns >= tmax ||
        a + w->v_rowi[v] * ldmin_ns < w->t0_ns || b + w->v_rowj[v] * ldmin_ns < w->t0_ns)     // (a negative lower bound of the line delay)
      return bad("visual time (+ row * line delay) outside the spline");
    if (!std::isfinite(w->v_pi[2 * v]) || !std::isfinite(w->v_pi[2 * v + 1]) || !std::isfinite(w->v_pj[2 * v]) || !std::isfinite(w->v_pj[2 * v + 1]))
      return bad("non-finite visual observation");
  }
  for (int b = 0; b < w->NB; ++b)
    if (w->bc_i[b] < 0 || w->bc_i[b] >= w->F || w->bc_j[b] < 0 || w->bc_j[b] >= w->F) return bad("bias chain index out of range");
  if (w->pn > 0) {
    if (!w->pJ0 || !w->pr0 || !w->p_x0 || !w->p_kind || !w->p_index || !w->p_off) return bad("null prior pointer");
    // the kept blocks must tile the prior's columns exactly once (a hole would leave an unmapped column)
    std::vector<uint8_t> cover((size_t)w->pn, 0);
    for (int b = 0; b < w->pnb; ++b) {
      const int kind = w->p_kind[b], idx = w->p_index[b];
      const int lim = (kind <= CTVIO_PK_POS) ? w->K : (kind <= CTVIO_PK_BA ? w->F : 1);
      if (kind < 0 || kind > CTVIO_PK_LD || idx < 0 || idx >= lim || w->p_off[b] < 0 || w->p_off[b] + prior_block_size(kind) > w->pn)
        return bad("prior block out of range");
      for (int k = 0; k < prior_block_size(kind); ++k)
        if (cover[w->p_off[b] + k]++) return bad("prior blocks overlap");
    }
    for (int i = 0; i < w->pn; ++i)
      if (!cover[i]) return bad("prior column not covered by any kept block");
  } else if (w->pnb != 0) {
    return bad("prior blocks without a prior");
  }
  return true;
}

// first pass: IMU samples sorted by (segment, bias state) and cut into groups, visual blocks sorted by frame pair (then rows)
// and cut into items of <= vch blocks -- only the orders and the counts are kept
inline void plan_window(const ctvio_window *w, int vch, PackTmp &t) {
  const int M = w->M, V = w->V;
  t.iseg.resize(M); t.iorder.resize(M);
  bool sorted = true;
  for (int i = 0; i < M; ++i) {
    t.iseg[i] = (int32_t)((w->imu_t[i] - w->t0_ns) / w->dt_ns);
    t.iorder[i] = i;
    if (i && (t.iseg[i] < t.iseg[i - 1] || (t.iseg[i] == t.iseg[i - 1] && w->imu_bias[i] < w->imu_bias[i - 1]))) sorted = false;
  }
  if (!sorted)
    std::stable_sort(t.iorder.begin(), t.iorder.end(), [&](int a, int b) {
      if (t.iseg[a] != t.iseg[b]) return t.iseg[a] < t.iseg[b];
      return w->imu_bias[a] < w->imu_bias[b];
    });
  t.ngrp = 0;
  for (int i = 0; i < M; ++i) {
    const int s = t.iorder[i];
    if (i == 0 || t.iseg[s] != t.iseg[t.iorder[i - 1]] || w->imu_bias[s] != w->imu_bias[t.iorder[i - 1]]) t.ngrp++;
  }
  t.vord.resize(V);
  std::iota(t.vord.begin(), t.vord.end(), 0);
  auto vless = [&](int a, int b) {
    if (w->v_ti[a] != w->v_ti[b]) return w->v_ti[a] < w->v_ti[b];
    if (w->v_tj[a] != w->v_tj[b]) return w->v_tj[a] < w->v_tj[b];
    if (w->v_rowi[a] != w->v_rowi[b]) return w->v_rowi[a] < w->v_rowi[b];
    return w->v_rowj[a] < w->v_rowj[b];
  };
  if (!std::is_sorted(t.vord.begin(), t.vord.end(), vless)) std::stable_sort(t.vord.begin(), t.vord.end(), vless);
  t.nvitem = 0;
  int cnt = 0;
  for (int i = 0; i < V; ++i) {
    const int v = t.vord[i];
    const bool fresh = (i == 0) || w->v_ti[v] != w->v_ti[t.vord[i - 1]] || w->v_tj[v] != w->v_tj[t.vord[i - 1]] || cnt >= vch;
    if (fresh) { t.nvitem++; cnt = 0; }
    cnt++;
  }
  // Anchors: the distinct i ends (landmark, t_i, row_i, p_i).  The reference adds every observation of a feature against the
  // feature's first one (trajectory_manager.cpp:367-383), i.e. one anchor per landmark; the C ABI takes arbitrary blocks, so a
  // landmark may own several (found by a linear search over the landmark's short list).  Numbered landmark-major.
  const int L = w->L;
  std::vector<int32_t> first((size_t)L, -1), next, rep, tmp_anc((size_t)V), acount;
  auto same_anchor = [&](int a, int b) {
    return w->v_ti[a] == w->v_ti[b] && w->v_rowi[a] == w->v_rowi[b] && w->v_pi[2 * a] == w->v_pi[2 * b] && w->v_pi[2 * a + 1] == w->v_pi[2 * b + 1];
  };
  std::vector<int32_t> lcount((size_t)L + 1, 0);
  for (int v = 0; v < V; ++v)   // (before the anchor search below: its per-landmark lists are linear, a malformed window must not make it quadratic)
    if (++lcount[w->v_lm[v]] > 64) { t.err = "more than 64 observations of one landmark"; return; }
  for (int i = 0; i < V; ++i) {
    const int v = t.vord[i], l = w->v_lm[v];
    int a = first[l], prev = -1;
    while (a >= 0 && !same_anchor(rep[a], v)) { prev = a; a = next[a]; }
    if (a < 0) {
      a = (int)rep.size();
      rep.push_back(v); next.push_back(-1); acount.push_back(0);
      if (prev < 0) first[l] = a; else next[prev] = a;
    }
    tmp_anc[v] = a;
    acount[a]++;
  }
  // Evaluation order: landmark-major (inside a landmark anchor by anchor, frame-pair order inside an anchor), so that the wave of
  // k_vis_eval that evaluates a landmark's blocks also forms its row of W.  A landmark never straddles a group of 64 slots (padding
  // slots in front of it); the window's slot count is a multiple of 64 (windows start on a wave boundary).
  t.A = (int)rep.size();
  t.anc_rep.resize((size_t)t.A);
  std::vector<int32_t> newid((size_t)t.A), astart((size_t)t.A);
  int pos = 0, na = 0;
  for (int l = 0; l < L; ++l) {
    const int c = lcount[l];
    if ((pos & 63) + c > 64) pos = (pos + 63) & ~63;
    for (int a = first[l]; a >= 0; a = next[a]) {
      newid[a] = na; t.anc_rep[na] = rep[a]; astart[na] = pos;
      pos += acount[a];
      ++na;
    }
  }
  t.Vp = (pos + 63) & ~63;
  t.lord.assign((size_t)t.Vp, -1);
  t.vpos.resize(V);
  t.anc_of.resize(V);
  for (int i = 0; i < V; ++i) {     // frame-pair order inside an anchor
    const int v = t.vord[i], a = newid[tmp_anc[v]];
    const int slot = astart[a]++;
    t.lord[slot] = v;
    t.vpos[v] = slot;
    t.anc_of[v] = a;
  }
}

// The segment (first active knot) of an observation at line delay `ld`, exactly as the device computes it (kernels_visual.hpp: vis_times + clamp).
inline int vis_segment(const ctvio_window *w, int64_t t, int row, double ld) {
  const long long ld_ns = (long long)(ld * 1e9);
  const long long tau = (t - w->t0_ns) + (long long)row * ld_ns;
  const int s = (int)(tau / w->dt_ns);
  return std::max(0, std::min(s, w->K - 4));
}

// Sparsity plan of one window: what the reference leaves to SPARSE_NORMAL_CHOLESKY (trajectory_estimator.cpp:371-384) is decided here, once per
// upload, from the factor structure alone.
//   * Every landmark's KNOT SPAN [klo, khi]: the knots its residual blocks can touch (4 per spline end, image_feature_factor.h:79-101), over the
//     whole box of the line delay -- the row time t + row * ld moves with ld (image_feature_factor.h:72), and both extremes of a monotone
//     function bound it.  The rows of W are ordered by (klo, khi), landmarks without observations last.
//   * For every 16-column tile of the pose unknowns: the range of sorted rows that can be non-zero there (the Schur kernels multiply only those).
//   * The ENVELOPE of the reduced system S = Hpp - W^T Hll^-1 W: first[u] = the smallest unknown coupled to u by an IMU group (4 knots + its bias
//     state), a bias-chain link, the prior (all its columns mutually), or a landmark (its span's knots mutually, and each with the line delay);
//     Cholesky fill stays inside the row envelope, so tiles (r, c < env_first[r]) are never formed, stored or multiplied.  dense = true (batches
//     whose factorisation keeps the whole triangle in registers: P <= 223) sets env_first = 0 -- every tile is formed and loaded -- and leaves
//     the raw tile envelope in env_tile, by which k_cholesky_tiles skips the PRODUCTS of empty tiles; full_ranges widens every non-empty row
//     range to all observed rows and drops the envelope (the dense cross-check).
inline void plan_sparsity(const ctvio_window *w, bool dense, bool full_ranges, PackTmp &t) {
  const int K = w->K, F = w->F, L = w->L, V = w->V, K6 = 6 * K, P = K6 + 6 * F + 1;
  std::vector<int32_t> klo((size_t)L, K), khi((size_t)L, -1);
  const double ld_a = w->fix_ld ? w->ld : w->ld_lo, ld_b = w->fix_ld ? w->ld : w->ld_hi;
  for (int v = 0; v < V; ++v) {
    const int l = w->v_lm[v];
    const int s[4] = {vis_segment(w, w->v_ti[v], w->v_rowi[v], ld_a), vis_segment(w, w->v_ti[v], w->v_rowi[v], ld_b),
                      vis_segment(w, w->v_tj[v], w->v_rowj[v], ld_a), vis_segment(w, w->v_tj[v], w->v_rowj[v], ld_b)};
    const int lo = std::min(std::min(s[0], s[1]), std::min(s[2], s[3])), hi = std::max(std::max(s[0], s[1]), std::max(s[2], s[3])) + 3;
    klo[l] = std::min(klo[l], lo); khi[l] = std::max(khi[l], hi);
  }
  t.lm_at.resize((size_t)L); t.lm_pos.resize((size_t)L); t.row_klo.resize((size_t)L); t.row_khi.resize((size_t)L);
  std::iota(t.lm_at.begin(), t.lm_at.end(), 0);
  std::stable_sort(t.lm_at.begin(), t.lm_at.end(), [&](int a, int b) { return klo[a] != klo[b] ? klo[a] < klo[b] : khi[a] < khi[b]; });
  t.Lobs = 0; t.max_span = 0;
  for (int r = 0; r < L; ++r) {
    const int l = t.lm_at[r];
    t.lm_pos[l] = r; t.row_klo[r] = klo[l]; t.row_khi[r] = khi[l];
    if (khi[l] >= 0) { t.Lobs = r + 1; t.max_span = std::max(t.max_span, khi[l] - klo[l] + 1); }
  }
  const int ntr = P / 16 + 1;
  t.ntr = ntr;
  t.tl_beg.assign((size_t)ntr, 0); t.tl_end.assign((size_t)ntr, 0); t.env_first.assign((size_t)ntr, 0); t.env_tile.assign((size_t)ntr, 0);
  for (int c = 0; c < ntr; ++c) {
    int beg = L, end = 0;
    if (16 * c < K6) {
      const int kf = 16 * c / 6, kl = std::min(16 * c + 15, K6 - 1) / 6;
      for (int r = 0; r < t.Lobs; ++r)
        if (t.row_klo[r] <= kl && t.row_khi[r] >= kf) { beg = std::min(beg, r); end = r + 1; }
    }
    if (16 * c <= P - 1 && P - 1 < 16 * c + 16 && t.Lobs > 0) { beg = 0; end = t.Lobs; }   // the line-delay column: every observed landmark
    if (end <= beg) beg = end = 0;
    if (full_ranges && end > 0) { beg = 0; end = t.Lobs; }     // (CTVIO_DENSE: the A/B switch -- every tile with products multiplies every row)
    t.tl_beg[c] = beg; t.tl_end[c] = end;
  }
  if (full_ranges) return;   // (CTVIO_DENSE: no envelope at all)
  // envelope, in columns: per knot block, per bias block, line delay
  std::vector<int32_t> fk((size_t)K), fb((size_t)F);
  for (int k = 0; k < K; ++k) fk[k] = 6 * k;
  for (int f = 0; f < F; ++f) fb[f] = K6 + 6 * f;
  int fld = P - 1;
  for (int i = 0; i < w->M; ++i) {
    const int s = t.iseg[i], b = w->imu_bias[i];
    for (int j = 1; j < 4; ++j) fk[s + j] = std::min(fk[s + j], 6 * s);
    fb[b] = std::min(fb[b], 6 * s);
  }
  for (int b = 0; b < w->NB; ++b) {
    const int i = std::min(w->bc_i[b], w->bc_j[b]), j = std::max(w->bc_i[b], w->bc_j[b]);
    fb[j] = std::min(fb[j], K6 + 6 * i);
  }
  for (int l = 0; l < L; ++l) {
    if (khi[l] < 0) continue;
    for (int k = klo[l] + 1; k <= khi[l]; ++k) fk[k] = std::min(fk[k], 6 * klo[l]);
    fld = std::min(fld, 6 * klo[l]);
  }
  if (w->pn > 0) {
    int m = P;
    auto col0 = [&](int b) {
      const int kind = w->p_kind[b], idx = w->p_index[b];
      return kind == CTVIO_PK_ROT ? 6 * idx : kind == CTVIO_PK_POS ? 6 * idx + 3 : kind == CTVIO_PK_BG ? K6 + 6 * idx : kind == CTVIO_PK_BA ? K6 + 6 * idx + 3 : P - 1;
    };
    for (int b = 0; b < w->pnb; ++b) m = std::min(m, col0(b));
    for (int b = 0; b < w->pnb; ++b) {
      const int kind = w->p_kind[b], idx = w->p_index[b];
      if (kind <= CTVIO_PK_POS) fk[idx] = std::min(fk[idx], m);
      else if (kind <= CTVIO_PK_BA) fb[idx] = std::min(fb[idx], m);
      else fld = std::min(fld, m);
    }
  }
  for (int r = 0; r < ntr; ++r) {
    int f = 16 * r;                                            // (a row's own diagonal entry)
    for (int u = 16 * r; u < std::min(16 * r + 16, P); ++u) f = std::min(f, u < K6 ? fk[u / 6] : (u < P - 1 ? fb[(u - K6) / 6] : fld));
    if (16 * r <= P && P < 16 * r + 16) f = 0;                 // the rhs row rides along as row P: dense
    // The panel Cholesky (k_cholesky_solve) works in 32-column panels: the envelope starts on a panel boundary (even tile column), and every
    // tile row reaches at least the panel before its own 32-row block (so that the next diagonal block always takes part in a panel's
    // trailing update: its look-ahead relies on that).  The tiles this adds hold zeros.
    int ft = f / 16;
    t.env_tile[r] = ft;                                        // the envelope as it is (k_cholesky_tiles skips the products of empty tiles)
    if (r >= 2) ft = std::min(ft, 2 * (r / 2) - 2);
    t.env_first[r] = (dense || r < 2) ? 0 : (ft & ~1);
  }
}

// Unknowns of Ceres' reduced program: referenced by some residual block and not constant
// (trajectory_estimator.cpp:114-141, 236-245, 311-318).
inline void active_mask(const ctvio_window *w, const PackTmp &t, int P, const int32_t *pcol, uint8_t *act) {
  const int N = P + w->L, K = w->K;
  std::memset(act, 0, (size_t)N);
  for (int i = 0; i < w->M; ++i) {
    std::memset(act + 6 * t.iseg[i], 1, 24);
    std::memset(act + 6 * K + 6 * w->imu_bias[i], 1, 6);
  }
  const int64_t pad_ns = (int64_t)(0.039 * 1e9);  // AddImageFeatureDelayAnalytic spans [t, t + 0.039 s] (trajectory_estimator.cpp:299)
  for (int v = 0; v < w->V; ++v) {
    const int64_t tt[2] = {w->v_ti[v], w->v_tj[v]};
    for (int e = 0; e < 2; ++e) {
      const int s0 = (int)((tt[e] - w->t0_ns) / w->dt_ns), s1 = (int)((tt[e] + pad_ns - w->t0_ns) / w->dt_ns);
      for (int k = s0; k < s1 + 4 && k < K; ++k) std::memset(act + 6 * k, 1, 6);
    }
    act[P + w->v_lm[v]] = 1;
    act[P - 1] = 1;
  }
  for (int b = 0; b < w->NB; ++b) { std::memset(act + 6 * K + 6 * w->bc_i[b], 1, 6); std::memset(act + 6 * K + 6 * w->bc_j[b], 1, 6); }
  for (int i = 0; i < w->pn; ++i) act[pcol[i]] = 1;
  for (int k = 0; k <= w->fixed_upto && k < K; ++k) std::memset(act + 6 * k, 0, 6);
  if (w->knot_const)
    for (int k = 0; k < K; ++k) if (w->knot_const[k]) std::memset(act + 6 * k, 0, 6);
  for (int f = 0; f < w->F; ++f) {
    if (w->lock_bg) std::memset(act + 6 * K + 6 * f, 0, 3);
    if (w->lock_ba) std::memset(act + 6 * K + 6 * f + 3, 0, 3);
  }
  if (w->fix_ld) act[P - 1] = 0;
}

}  // namespace ctv
