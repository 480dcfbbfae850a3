// kernels.hpp -- gfx950 kernels of the sliding-window solve (all fp64).  One launch covers a whole batch of windows.
//
//   per state : k_knot_prep (d = log(R_k^-1 R_k+1) and Jr^-1(d) of every knot pair, shared by all blocks; the candidate's table is made by
//               k_step_finish)
//   linearise : k_linearize_f64 = k_imu_linearize_f64 (one wave per IMU group: rows through LDS, per-group A^T A on the fp64 matrix
//               cores; it also clears the accumulated parts of the normal equations) + k_vis_eval (landmark-major: J~ block-major via
//               LDS, the rows of W, Hll, g_rho formed in the same kernel) in one launch; k_imu_linearize_rest (groups the specialised
//               IMU body leaves out); k_imu_linearize<T, CHUNK> = vector-ALU cross-check (use_mfma = 0); k_zero_normal only for batches
//               with an IMU-less window
//   assemble  : k_assemble_vis_mfma (MFMA + fp64 LDS Hessian, or global atomics for K > 25; STORE = the order-fixed tail of the
//               deterministic mode with k_reduce_finalize / k_bias_rows; k_assemble_vis = register-tile cross-check), k_assemble_imu,
//               k_misc (bias chain + prior), k_post_linearize (first linearisation only)
//   step      : k_schur_window_f64 (large batches) / k_schur_tile_f64 (small; both also produce the reduced rhs), k_schur_generic + k_rhs
//               (vector fallback), k_cholesky_tiles (register-resident 16 x 16 tiles; k_cholesky_solve = panel kernel for P > 223),
//               k_step_finish (back-substitution, candidate x (+) alpha delta, the candidate's knot-pair table)
//   control   : k_lm_init, k_initial_cost, k_begin_iter, k_pass_end (gradient norm, cost, Ceres 1.14 accept / reject / terminate / Armijo,
//               set swap, next iteration's damping)
//   after     : k_gauge_restore (double2vector), k_spline_eval (trajectory query), k_residual_summary
#pragma once
#include <utility>

#include "device_types.hpp"
#include "factors.hpp"

namespace ctv {

// SPECULATIVE LINEARISATION.  Every pass evaluates the candidate x (+) alpha delta exactly once -- residuals, Jacobians and the
// normal equations together, into the normal-equation set that is NOT the current one (Lm::cur).  The cost at the candidate is
// a by-product (per-group / per-wave partial sums, added up in a fixed order by k_pass_end); on acceptance the sets swap and the
// next iteration starts from a finished linearisation; on rejection the current set is still intact.  The trial points of
// Ceres' projected line search need value and gradient anyway.  Only the last allowed iteration (nothing can follow it) is
// costed without Jacobians.  Modes of the linearisation kernels:
enum { LIN_AT_X = 0,      // the current state into the current set (the first linearisation of a solve, diagnostics)
       LIN_SPEC = 1,      // the candidate into the other set (every pass)
       COST_AT_X = 2 };   // residuals of the current state only (ctvio_cost)
__device__ __forceinline__ bool lin_run(const Lm &lm, int mode) { return lm.status == 0 && (mode != LIN_SPEC || lm.step_valid != 0); }
__device__ __forceinline__ bool lin_cost_only(const Lm &lm, int mode, const LmParams &p) {
  return mode == COST_AT_X || (mode == LIN_SPEC && lm.iter >= p.max_iters && lm.ls_active == 0);
}
__device__ __forceinline__ int lin_target(const Lm &lm, int mode) { return mode == LIN_SPEC ? 1 - lm.cur : lm.cur; }

// a lane's double as a wave-uniform value (two v_readlane_b32)
__device__ __forceinline__ double readlane_d(double x, int lane) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// local column -> unknown index maps
__device__ __forceinline__ int imu_col(int c, int s, int K, int bias) {
  if (c < 12) return 6 * (s + c / 3) + c % 3;
  if (c < 24) return 6 * (s + (c - 12) / 3) + 3 + (c - 12) % 3;
  return 6 * K + 6 * bias + (c - 24);
}
__device__ __forceinline__ int vis_col(int c, int si, int sj, int P) {
  if (c < 12) return 6 * (si + c / 3) + c % 3;
  if (c < 24) return 6 * (si + (c - 12) / 3) + 3 + (c - 12) % 3;
  if (c < 36) return 6 * (sj + (c - 24) / 3) + (c - 24) % 3;
  if (c < 48) return 6 * (sj + (c - 36) / 3) + 3 + (c - 36) % 3;
  if (c == 48) return -1;  // inverse depth: landmark block
  return P - 1;            // line delay
}

template <class T> __device__ __forceinline__ void load_knots(const double *quat, const double *pos, int k0, const double *origin,
                                                            Knots4<T> &k) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double *q = quat + 4 * (k0 + i), *p = pos + 3 * (k0 + i);
    k.q[i] = qmk<T>((T)q[0], (T)q[1], (T)q[2], (T)q[3]);
    k.p[i] = mk<T>((T)(p[0] - origin[0]), (T)(p[1] - origin[1]), (T)(p[2] - origin[2]));
  }
}

// Knots k0..k0+3 in the local frame of the reference knot `kref` (fp64 arithmetic, then cast):
//   q'_k = q_ref^-1 q_k (near identity), p'_k = R_ref^T (p_k - p_ref).
template <class T> struct LocalFrame {
  Q4<double> qref_inv;
  M3<double> RT;      // R_ref^T
  double o[3];
  __device__ __forceinline__ void init(const double *quat, const double *pos, int kref) {
    const double *q = quat + 4 * kref, *p = pos + 3 * kref;
    qref_inv = qmk<double>(-q[0], -q[1], -q[2], q[3]);
    const M3<double> R = q2R(qmk<double>(q[0], q[1], q[2], q[3]));
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) RT.m[3 * i + j] = R.m[3 * j + i];
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
  }
  __device__ __forceinline__ void load(const double *quat, const double *pos, int k0, Knots4<T> &k) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const double *q = quat + 4 * (k0 + i), *p = pos + 3 * (k0 + i);
      const Q4<double> ql = qmul_raw(qref_inv, qmk<double>(q[0], q[1], q[2], q[3]));   // unit x unit: no renormalisation in fp64
      const V3<double> pl = mul(RT, mk<double>(p[0] - o[0], p[1] - o[1], p[2] - o[2]));
      k.q[i] = qmk<T>((T)ql.x, (T)ql.y, (T)ql.z, (T)ql.w);
      k.p[i] = mk<T>((T)pl.x, (T)pl.y, (T)pl.z);
    }
  }
  __device__ __forceinline__ V3<T> rotate(const double *v) const {  // R_ref^T v
    const V3<double> r = mul(RT, mk<double>(v[0], v[1], v[2]));
    return mk<T>((T)r.x, (T)r.y, (T)r.z);
  }
  __device__ __forceinline__ M3<T> RrefT() const {
    M3<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.m[i] = (T)RT.m[i];
    return r;
  }
};

// ------------------------------------------------------------------------------------------------ control
template <class T> __global__ void k_lm_init(Dev<T> d, double mu, int keep_scale) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= d.nwin) return;
  Lm &lm = d.lm[w];
  lm.cost = lm.cand_cost = lm.initial_cost = 0;
  lm.mu = mu; lm.nu = 2.0; lm.model_change = 0;
  lm.step2 = lm.xnorm2 = lm.cand_xnorm2 = 0;
  lm.gmax_bits = 0ull; lm.cand_gmax_bits = 0ull; lm.cand_gd = 0;
  lm.iter = 0; lm.invalid = 0; lm.status = 0; lm.cur = 0;
  lm.scaled = keep_scale ? lm.scaled : 0; lm.last_ok = 1; lm.step_valid = 0; lm.chol_fail = 0; lm.accept = 0;
  lm.nsucc = lm.nunsucc = 0; lm.have_grad = 0;
  const WinMeta &m = d.wins[w];
  lm.ls_on = (d.line_search && !m.fix_ld && d.active[m.u0 + m.P - 1]) ? 1 : 0;   // Program::IsBoundsConstrained of the reduced program
  lm.ls_active = 0; lm.ls_iters = 0; lm.ls_prev_valid = lm.ls_cur_valid = 0; lm.nls_steps = lm.nls_reduced = 0;
  lm.alpha = 1.0; lm.ls_gd0 = 0; lm.ls_dmax = 0;
  lm.ls_cur_x = lm.ls_cur_v = lm.ls_cur_g = lm.ls_prev_x = lm.ls_prev_v = lm.ls_prev_g = 0;
}

// ---- interpolation of the next trial step (Ceres polynomial.cc: FindInterpolatingPolynomial / MinimizePolynomial)
struct LsSample { double x, v, g; };
__device__ inline double ls_poly_eval(const double *p, int n, double x) {
  double v = 0;
  for (int i = 0; i < n; ++i) v = v * x + p[i];
  return v;
}
__device__ inline double ls_ipow(double x, int e) { double r = 1.0; for (int i = 0; i < e; ++i) r *= x; return r; }
// real parts of all (complex) roots of a polynomial of degree <= 4, coefficients highest power first
__device__ inline int ls_root_real_parts(const double *p_in, int n, double *re) {
  while (n > 0 && p_in[0] == 0.0) { ++p_in; --n; }
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
  for (int i = 1; i <= deg; ++i) rad = fmax(rad, fabs(q[i]));
  rad = 1.0 + rad;
  for (int k = 0; k < deg; ++k) { const double ang = 2.0 * 3.14159265358979323846 * k / deg + 0.4; zr[k] = 0.5 * rad * cos(ang); zi[k] = 0.5 * rad * sin(ang); }
  for (int it = 0; it < 500; ++it) {   // Durand-Kerner
    double change = 0;
    for (int k = 0; k < deg; ++k) {
      double pr = 1.0, pi = 0.0;
      for (int i = 1; i <= deg; ++i) { const double tr = pr * zr[k] - pi * zi[k] + q[i], ti = pr * zi[k] + pi * zr[k]; pr = tr; pi = ti; }
      double dr = 1.0, di = 0.0;
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
__device__ inline double ls_minimize_interpolating(const LsSample *s, int ns, double x_min, double x_max) {
  const int nc = 2 * ns, deg = nc - 1;
  double A[6][7], coef[6], der[5], roots[4];
  for (int i = 0; i < ns; ++i) {
    for (int j = 0; j <= deg; ++j) A[2 * i][j] = ls_ipow(s[i].x, deg - j);
    A[2 * i][nc] = s[i].v;
    for (int j = 0; j < deg; ++j) A[2 * i + 1][j] = (deg - j) * ls_ipow(s[i].x, deg - j - 1);
    A[2 * i + 1][deg] = 0.0;
    A[2 * i + 1][nc] = s[i].g;
  }
  for (int c = 0; c < nc; ++c) {
    int piv = c;
    for (int r = c + 1; r < nc; ++r) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
    if (A[piv][c] == 0.0) return 0.5 * (x_min + x_max);
    if (piv != c) for (int j = 0; j <= nc; ++j) { const double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; }
    for (int r = 0; r < nc; ++r) {
      if (r == c) continue;
      const double f = A[r][c] / A[c][c];
      for (int j = c; j <= nc; ++j) A[r][j] -= f * A[c][j];
    }
  }
  for (int c = 0; c < nc; ++c) coef[c] = A[c][nc] / A[c][c];
  double best_x = 0.5 * (x_min + x_max), best = ls_poly_eval(coef, nc, best_x), v;
  v = ls_poly_eval(coef, nc, x_min); if (v < best) { best = v; best_x = x_min; }
  v = ls_poly_eval(coef, nc, x_max); if (v < best) { best = v; best_x = x_max; }
  for (int i = 0; i < nc - 1; ++i) der[i] = (nc - 1 - i) * coef[i];
  const int nr = ls_root_real_parts(der, nc - 1, roots);
  for (int i = 0; i < nr; ++i) {
    if (roots[i] < x_min || roots[i] > x_max) continue;
    v = ls_poly_eval(coef, nc, roots[i]);
    if (v < best) { best = v; best_x = roots[i]; }
  }
  for (int i = 0; i < ns; ++i) {
    if (s[i].x < x_min || s[i].x > x_max) continue;
    v = ls_poly_eval(coef, nc, s[i].x);
    if (v < best) { best = v; best_x = s[i].x; }
  }
  return best_x;
}

// Sum of the cost partials of window w over the 64 lanes of one wave, in a fixed order (lane-strided partial sums, then a butterfly):
// every lane returns the same total.  IMU groups, visual waves (a window's block slots start on a wave boundary), bias chain + prior.
template <class T> __device__ __forceinline__ double window_cost_sum(const Dev<T> &d, const WinMeta &m, int w, int lane) {
  double c = 0.0;
  for (int g = lane; g < m.ngrp; g += 64) c += d.imu_cost[m.grp0 + g];
  const int vw0 = m.vis0 >> 6, nvw = m.Vp >> 6;
  for (int i = lane; i < nvw; i += 64) c += d.vis_cost[vw0 + i];
  if (lane == 0) c += d.misc_cost[w];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  return c;
}

// After the first linearisation of a solve (LIN_AT_X): cost of the initial state, Jacobi scaling is in place.
template <class T> __global__ __launch_bounds__(64) void k_initial_cost(Dev<T> d, int as_candidate) {
  const int w = blockIdx.x;
  Lm &lm = d.lm[w];
  if (lm.status) return;
  const WinMeta &m = d.wins[w];
  const int lane = threadIdx.x;
  const double c = window_cost_sum(d, m, w, lane);
  // |x|^2 over the ambient coordinates of the reduced program's parameter blocks (Ceres x_norm), lane-strided, fixed order
  double x2 = 0.0;
  const uint8_t *act = d.active + m.u0;
  for (int k = lane; k < m.K; k += 64) {
    const double *q = d.quat + 4 * (m.knot0 + k), *p = d.pos + 3 * (m.knot0 + k);
    if (act[6 * k]) x2 += q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    for (int cc = 0; cc < 3; ++cc) if (act[6 * k + 3 + cc]) x2 += p[cc] * p[cc];
  }
  for (int j = lane; j < 6 * m.F; j += 64) if (act[6 * m.K + j]) { const double b = d.bias[6 * m.bias0 + j]; x2 += b * b; }
  for (int l = lane; l < m.L; l += 64) if (act[m.P + l]) { const double r = d.rho[m.lm0 + l]; x2 += r * r; }
  if (lane == 0 && act[m.P - 1]) x2 += d.ld[w] * d.ld[w];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x2 += __shfl_xor(x2, off);
  if (lane != 0) return;
  if (as_candidate) { lm.cand_cost = c; return; }   // ctvio_cost
  lm.cost = lm.initial_cost = c;
  lm.cand_cost = 0;
  lm.xnorm2 = x2;
  lm.scaled = 1;
}

// Decision of one window after its candidate has been evaluated (lane 0 of k_lm_control): returns 1 when the candidate is accepted.
template <class T> __device__ inline int lm_decide(const Dev<T> &d, Lm &lm, double cand_cost, double gd, bool have_grad) {
  lm.cand_cost = cand_cost;
  lm.cand_gd = gd;
  lm.have_grad = have_grad ? 1 : 0;
  lm.accept = 0;
  if (lm.ls_on && lm.ls_active != 2) {
    const bool valid = isfinite(cand_cost) && (!have_grad || isfinite(gd));
    const bool ok = valid && !(cand_cost > lm.cost + 1e-4 * lm.ls_gd0 * lm.alpha);
    if (!ok) {
      if (!have_grad) { lm.ls_active = 3; return 0; }   // (last iteration, costed only) the same trial again, linearised
      if (lm.ls_active != 1) { lm.ls_active = 1; lm.ls_iters = 0; lm.ls_prev_valid = 0; lm.ls_cur_x = 1.0; }
      lm.ls_cur_v = cand_cost; lm.ls_cur_g = gd; lm.ls_cur_valid = valid ? 1 : 0;
      if (++lm.ls_iters >= 20) { lm.ls_active = 2; lm.alpha = 1.0; lm.nls_steps += lm.ls_iters; return 0; }   // max_num_line_search_step_size_iterations: the full step is kept
      const double lo = 1e-3 * lm.ls_cur_x, hi = 0.6 * lm.ls_cur_x;   // max_step_contraction, min_step_contraction
      double step;
      if (!valid) {
        step = fmin(fmax(lm.ls_cur_x * 0.5, lo), hi);
      } else {
        LsSample sp[3];
        int ns = 0;
        sp[ns++] = LsSample{0.0, lm.cost, lm.ls_gd0};
        sp[ns++] = LsSample{lm.ls_cur_x, lm.ls_cur_v, lm.ls_cur_g};
        if (lm.ls_prev_valid) sp[ns++] = LsSample{lm.ls_prev_x, lm.ls_prev_v, lm.ls_prev_g};
        step = ls_minimize_interpolating(sp, ns, lo, hi);
      }
      if (step * lm.ls_dmax < 1e-9) { lm.ls_active = 2; lm.alpha = 1.0; lm.nls_steps += lm.ls_iters; return 0; }   // min_line_search_step_size
      lm.ls_prev_x = lm.ls_cur_x; lm.ls_prev_v = lm.ls_cur_v; lm.ls_prev_g = lm.ls_cur_g; lm.ls_prev_valid = valid ? 1 : 0;
      lm.ls_cur_x = step;
      lm.alpha = step;
      return 0;   // next pass: candidate at the new alpha
    }
    if (lm.ls_active == 1) { lm.nls_steps += lm.ls_iters; lm.nls_reduced += 1; }
  }
  lm.ls_active = 0;
  const double step_norm = sqrt(lm.step2), x_norm = sqrt(lm.xnorm2);
  if (step_norm <= d.prm.ptol * (x_norm + d.prm.ptol)) { lm.status = 1 + 2; return 0; }
  const double cost_change = lm.cost - cand_cost;
  if (fabs(cost_change) <= d.prm.ftol * lm.cost) { lm.status = 1 + 3; return 0; }
  const double rel = cost_change / lm.model_change;
  if (rel > d.prm.min_rel_dec && isfinite(cand_cost)) {
    lm.accept = 1;
    lm.cost = cand_cost;
    lm.xnorm2 = lm.cand_xnorm2;
    const double t = 2.0 * rel - 1.0;
    double f = 1.0 - t * t * t;
    if (f < 1.0 / 3.0) f = 1.0 / 3.0;
    lm.mu = fmin(lm.mu / f, d.prm.max_radius);
    lm.nu = 2.0; lm.last_ok = 1; lm.nsucc += 1;
    if (have_grad) { lm.cur ^= 1; lm.gmax_bits = lm.cand_gmax_bits; }   // the speculative linearisation is the current one now
    return 1;
  }
  lm.mu /= lm.nu; lm.nu *= 2.0; lm.last_ok = 0; lm.nunsucc += 1;
  return 0;
}


// End of a pass, one workgroup per window.  For a window whose candidate has been evaluated: cost of the candidate (fixed-order sum
// of the partials), gradient max-norm at the candidate and its directional derivative g(candidate) . delta, then
//   * ArmijoLineSearch::DoSearch + LineSearch::InterpolatingPolynomialMinimizingStepSize (Ceres line_search.cc) for windows whose
//     reduced program is bounds-constrained: the trial is kept when f(alpha) <= f(0) + 1e-4 alpha f'(0); a sample whose value or
//     gradient is not finite is invalid and fails; otherwise the next trial step comes from the cubic / quintic interpolation,
//     contracted into [1e-3, 0.6] x alpha, and the window stays in the search (ls_active = 1);
//   * ParameterToleranceReached / FunctionToleranceReached / IsStepSuccessful / LM radius update; on acceptance the speculative
//     normal equations become the current ones (cur ^= 1).
template <class T> __global__ __launch_bounds__(256) void k_pass_end(Dev<T> d) {
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Lm &lm = d.lm[w];
  __shared__ double s_red[4];
  __shared__ unsigned long long s_gmax[4];
  __shared__ int s_acc, s_go;
  const WinMeta &m = d.wins[w];
  if (!lm.status && lm.step_valid) {   // (uniform: the window evaluated a candidate this pass)
    const bool have_grad = !lin_cost_only(lm, LIN_SPEC, d.prm);
    const int tg = 1 - lm.cur;
    const double *gc = d.gS[tg] + m.u0, *dl = d.delta + m.u0;
    // gradient max-norm at the candidate and g(candidate) . delta: block reductions in a fixed order (max is exact in any order)
    double gd = 0.0, gm = 0.0;
    if (have_grad)
      for (int j = tid; j < m.N; j += 256)
        if (d.active[m.u0 + j]) { gd += gc[j] * dl[j]; gm = fmax(gm, grad_norm_entry(d, m, w, j, gc, true)); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { gd += __shfl_xor(gd, off); gm = fmax(gm, __shfl_xor(gm, off)); }
    if (lane == 0) { s_red[wave] = gd; s_gmax[wave] = (unsigned long long)__double_as_longlong(gm); }
    __syncthreads();
    if (wave == 0) {
      const double cand_cost = window_cost_sum(d, m, w, lane);
      if (lane == 0) {
        const double gdt = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        lm.cand_gmax_bits = max(max(s_gmax[0], s_gmax[1]), max(s_gmax[2], s_gmax[3]));   // (non-negative doubles order like their bit patterns)
        s_acc = lm_decide(d, lm, cand_cost, lm.ls_on ? gdt : 0.0, have_grad);
      }
    }
    __syncthreads();
    if (s_acc) {   // the accepted candidate becomes the current state (the reference: Ceres writes through the parameter pointers)
      for (int t = tid; t < 4 * m.K; t += 256) d.quat[4 * m.knot0 + t] = d.cquat[4 * m.knot0 + t];
      for (int t = tid; t < 3 * m.K; t += 256) d.pos[3 * m.knot0 + t] = d.cpos[3 * m.knot0 + t];
      for (int t = tid; t < 6 * m.F; t += 256) d.bias[6 * m.bias0 + t] = d.cbias[6 * m.bias0 + t];
      for (int t = tid; t < m.L; t += 256) d.rho[m.lm0 + t] = d.crho[m.lm0 + t];
      if (tid == 0) d.ld[w] = d.cld[w];
    }
    __syncthreads();
  }
  // ---- the next iteration starts here: continuation tests, LM diagonal of the (possibly swapped) current normal equations
  begin_iteration(d, w, &s_go);
}

// ------------------------------------------------------------------------------------------------ zero
template <class T> __global__ void k_zero_normal(Dev<T> d, int single_part, int mode) {
  const int w = blockIdx.y;
  const Lm &lm = d.lm[w];
  if (!lin_run(lm, mode) || lin_cost_only(lm, mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  const int tg = lin_target(lm, mode);
  double *Hpp = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  const long long nH = (long long)m.P * m.ldh;
  const long long stride = (long long)gridDim.x * blockDim.x;
  // with a single k_assemble_vis part the LDS path overwrites the whole knot x knot block and the line-delay row
  // (plain stores, issued after this kernel), so only the bias rows and the line-delay row need zeroing
  const long long first = (m.vis_lds && single_part) ? (long long)6 * m.K * m.ldh : 0;
  for (long long i = first + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nH; i += stride) Hpp[i] = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m.P; i += stride) g[i] = 0.0;
  // W, Hll and g[P..N) are written (not accumulated) by k_vis_eval
  if (blockIdx.x == 0 && threadIdx.x == 0) { if (mode == LIN_SPEC) d.lm[w].cand_gmax_bits = 0ull; else d.lm[w].gmax_bits = 0ull; }
}

// Knot-pair constants (Dev::lkd, kjri) of every window at its CURRENT state, before the first linearisation of a solve (the
// candidates' are formed by k_step_finish): one thread per knot.
template <class T> __global__ void k_knot_prep(Dev<T> d) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= d.Ktot) return;
  const int w = d.knot_win[g];
  const WinMeta &m = d.wins[w];
  if (g - m.knot0 >= m.K - 1) return;   // the last knot of a window starts no pair
  knot_pair_const<T>(d.quat + 4 * g, d.quat + 4 * g + 4, d.lkd + 3 * g, d.kjri + 9 * g);
}

// ------------------------------------------------------------------------------------------------ IMU
template <class T, int N> struct alignas(N * sizeof(T)) VecN { T v[N]; };

// VALU cross-check of k_imu_linearize_f64 (use_mfma = 0): the rows of A = [J | r] (row k = 6 * lane + r, 32 columns) staged
// column-major in LDS, A^T[32][KS], 4 consecutive k per ds_read; 4 x 4 register tile per lane (rows {ti+8a}, cols {tj+8b}).
template <class T> struct ImuLdsSink {
  T *A;
  int lane, stride;
  __device__ __forceinline__ void put_col(int col, const T v[6]) {
#pragma unroll
    for (int r = 0; r < 6; ++r) A[col * stride + 6 * lane + r] = v[r];
  }
};
template <class T> struct NullSink {
  __device__ __forceinline__ void put_col(int, const T *) {}
};

typedef double f64x4 __attribute__((ext_vector_type(4)));
// One workgroup (one wave) per IMU group.  The 4 active knots of the group are loaded once; every lane evaluates one sample and
// its 6 Jacobian rows + residual; then the wave forms the group's 31 x 31 block A^T A = [J^T J, J^T r; r^T J, r^T r].
// The tile is stored, not accumulated -- no atomics, deterministic.
template <class T, int CHUNK> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_imu_linearize(Dev<T> d, int mode) {
  constexpr int KCH = 6 * CHUNK, KS = KCH + 4;
  extern __shared__ __attribute__((aligned(32))) unsigned char smraw[];
  T *A = reinterpret_cast<T *>(smraw);
  const ImuGroup grp = d.groups[blockIdx.x];
  const int w = grp.win;
  if (!lin_run(d.lm[w], mode)) return;
  const bool jac = !lin_cost_only(d.lm[w], mode, d.prm);
  const WinMeta &m = d.wins[w];
  const int lane = threadIdx.x;
  Knots4<T> k;
  LocalFrame<T> lf;
  const bool at_cand = mode == LIN_SPEC;
  double csum = 0.0;
  const double *s_quat = at_cand ? d.cquat : d.quat, *s_pos = at_cand ? d.cpos : d.pos, *s_bias = at_cand ? d.cbias : d.bias;
  lf.init(s_quat, s_pos, m.knot0 + grp.s);
  lf.load(s_quat, s_pos, m.knot0 + grp.s, k);
  const M3<T> RrefT = lf.RrefT();
  SegConst<T> sc;
  seg_const_load(d.lkd + 3 * (m.knot0 + grp.s), d.kjri + 9 * (m.knot0 + grp.s), sc, true);
  T bias[6], wgt[6];
  const double *bp = s_bias + 6 * (m.bias0 + grp.bias);
#pragma unroll
  for (int i = 0; i < 6; ++i) { bias[i] = (T)bp[i]; wgt[i] = (T)m.imu_w[i]; }
  const V3<T> grav = lf.rotate(m.gravity);
  const T idt = (T)m.inv_dt;
  const int ti = lane >> 3, tj = lane & 7;
  T acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = T(0);
  const T zero6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
  for (int c0 = 0; c0 < grp.count; c0 += CHUNK) {
    const int nval = min(CHUNK, grp.count - c0);
    const int idx = m.imu0 + grp.start + c0 + lane;
    T gy[3], ac[3], r[6];
    if (lane < nval) {
#pragma unroll
      for (int i = 0; i < 3; ++i) { gy[i] = d.imu_meas[(size_t)i * d.Mtot + idx]; ac[i] = d.imu_meas[(size_t)(3 + i) * d.Mtot + idx]; }
    }
    const int kmax = (6 * nval + 3) & ~3;
    ImuLdsSink<T> sink{A, lane, KS};
    if (lane < nval) {
      imu_eval<T>(k, sc, d.imu_u[idx], idt, grav, bias, gy, ac, wgt, RrefT, r, jac, sink);
#pragma unroll
      for (int i = 0; i < 6; ++i) csum += 0.5 * (double)(r[i] * r[i]);
      sink.put_col(30, r);
      sink.put_col(31, zero6);
    } else if (6 * lane < kmax) {  // at most one partial lane: rows up to the multiple of 4 must read as zero
#pragma unroll
      for (int c = 0; c < 32; ++c) sink.put_col(c, zero6);
    }
    __syncthreads();
    for (int k0 = 0; k0 < kmax; k0 += 4) {
      VecN<T, 4> av[4], bv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        av[a] = *reinterpret_cast<const VecN<T, 4> *>(A + (ti + 8 * a) * KS + k0);
        bv[a] = *reinterpret_cast<const VecN<T, 4> *>(A + (tj + 8 * a) * KS + k0);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc[a][b] += av[a].v[kk] * bv[b].v[kk];
    }
    __syncthreads();
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off);
  if (lane == 0) d.imu_cost[blockIdx.x] = csum;
  if (!jac) return;
  T *tile = d.imu_tiles + (size_t)blockIdx.x * 1024;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) tile[(ti + 8 * a) * 32 + (tj + 8 * b)] = acc[a][b];
}

// The zeroing of the normal equations' accumulated parts, done by the IMU groups of the window instead of a pass of its own (k_zero_normal
// is HBM-bound, ~100 us per 2048 windows; the stores cost this compute-bound kernel nothing): group gi of the window clears its
// share of Hpp -- only the bias rows and the line-delay row when the single visual-assembly part overwrites the knot x knot block with plain
// stores (zero_mode 1), everything otherwise (zero_mode 2) -- and the first group the gradient and the max-norm cell.  Every accumulating
// kernel (k_assemble_vis*, k_assemble_imu, k_misc) is launched after the linearisation kernels.  zero_mode 0: k_zero_normal did it.
__device__ __forceinline__ void imu_zero_share(const Dev<double> &d, int mode, const ImuGroup &grp, int gidx, int zero_mode) {
  if (!zero_mode) return;
  const int w = grp.win, lane = threadIdx.x;
  const WinMeta &m = d.wins[w];
  const int tg = lin_target(d.lm[w], mode);
  double *Hpp = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  const int nH = m.P * m.ldh, first = (m.vis_lds && zero_mode == 1) ? 6 * m.K * m.ldh : 0;   // (ldh is a multiple of 16: both even)
  const int gi = gidx - m.grp0, per = (((nH - first) / 2 + m.ngrp - 1) / m.ngrp) * 2;
  const int lo = first + gi * per, hi = min(lo + per, nH);
  for (int i = lo + 2 * lane; i < hi; i += 128) *reinterpret_cast<double2 *>(Hpp + i) = double2{0.0, 0.0};
  if (gi == 0) {
    for (int i = lane; i < m.P; i += 64) g[i] = 0.0;
    if (lane == 0) { if (mode == LIN_SPEC) d.lm[w].cand_gmax_bits = 0ull; else d.lm[w].gmax_bits = 0ull; }
  }
}

// All-fp64 product path: one wave per IMU group, 64 samples per pass (one per lane), A^T A on the fp64 matrix cores.
// The 6 x 30 Jacobian of a sample stays in REGISTERS in factored form (ImuJac); its six rows are streamed through LDS one
// row index at a time -- phase a: row a of all 64 samples ([64][33] doubles = 16.9 KB, so 8 waves fit a CU and every lane
// evaluates a sample), then 16 K-steps of v_mfma_f64_16x16x4_f64 per output tile.  Accelerometer rows feed the three lower
// 16 x 16 tiles of the 32-column space, gyro rows (non-zero in rotation, gyro-bias and residual columns only) one 16 x 16
// tile on compacted columns.  MFMA operand layout (measured, tools/mfma_f64_layout.hip): A lane l = X[k = l/16][i = l%16],
// B lane l = Y[k = l/16][j = l%16], D register r of lane l = D[(l/16) + 4r][l%16].
__device__ __forceinline__ void imu_linearize_f64_body(const Dev<double> &d, int mode, double *A /* LDS [64][33] */, int gidx, int zero_mode) {
  const ImuGroup grp = d.groups[gidx];
  const int w = grp.win;
  if (!lin_run(d.lm[w], mode)) return;
  const bool jac = !lin_cost_only(d.lm[w], mode, d.prm);   // (uniform) the last allowed iteration only costs its candidate
  if (jac) imu_zero_share(d, mode, grp, gidx, zero_mode);
  const WinMeta &m = d.wins[w];
  const int lane = threadIdx.x, q4 = lane >> 4, l15 = lane & 15;
  const bool at_cand = mode == LIN_SPEC;
  double csum = 0.0;
  const double *s_quat = at_cand ? d.cquat : d.quat, *s_pos = at_cand ? d.cpos : d.pos, *s_bias = at_cand ? d.cbias : d.bias;
  Knots4<double> k;
  LocalFrame<double> lf;
  lf.init(s_quat, s_pos, m.knot0 + grp.s);
  lf.load(s_quat, s_pos, m.knot0 + grp.s, k);
  const M3<double> RrefT = lf.RrefT();
  SegConstLazy<double> sc;   // Jr^-1 of the three knot pairs: fetched from the table where it is used
  seg_const_lazy(d.lkd + 3 * (m.knot0 + grp.s), d.kjri + 9 * (m.knot0 + grp.s), sc);
  double bias[6], wgt[6];
  const double *bp = s_bias + 6 * (m.bias0 + grp.bias);
#pragma unroll
  for (int i = 0; i < 6; ++i) { bias[i] = bp[i]; wgt[i] = m.imu_w[i]; }
  const V3<double> grav = lf.rotate(m.gravity);
  const double idt = m.inv_dt;
  f64x4 acc00 = {0.0, 0.0, 0.0, 0.0}, acc10 = {0.0, 0.0, 0.0, 0.0}, acc11 = {0.0, 0.0, 0.0, 0.0}, gacc = {0.0, 0.0, 0.0, 0.0};
  const size_t Mt = (size_t)d.Mtot;
  if (!jac) {   // residuals only (a separate, small code path: the full one below keeps its compile-time `want_jac = true`)
    for (int c0 = 0; c0 < grp.count; c0 += 64) {
      const bool live = c0 + lane < grp.count;
      const int idx = m.imu0 + grp.start + min(c0 + lane, grp.count - 1);
      double gy[3], ac[3], r[6], wl[6];
#pragma unroll
      for (int i = 0; i < 3; ++i) { gy[i] = d.imu_meas[(size_t)i * Mt + idx]; ac[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
#pragma unroll
      for (int i = 0; i < 6; ++i) wl[i] = live ? wgt[i] : 0.0;
      ImuJac<double> J;
      imu_eval_core<double>(k, sc, d.imu_u[idx], idt, grav, bias, gy, ac, wl, RrefT, r, false, J);
#pragma unroll
      for (int i = 0; i < 6; ++i) csum += 0.5 * r[i] * r[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off);
    if (lane == 0) d.imu_cost[gidx] = csum;
    return;
  }
  for (int c0 = 0; c0 < grp.count; c0 += 64) {
    const int nval = min(64, grp.count - c0);
    const bool live = lane < nval;
    const int idx = m.imu0 + grp.start + min(c0 + lane, grp.count - 1);   // clamped: every lane evaluates (uniform control flow around the MFMAs)
    double gy[3], ac[3], r[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) { gy[i] = d.imu_meas[(size_t)i * Mt + idx]; ac[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
    // lanes past the end of the group evaluate a clamped sample with ZERO weights: every row of w .* [J | r] is then exactly
    // zero (one select per weight instead of one per stored entry: 288 v_cndmask per pass)
    double wl[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wl[i] = live ? wgt[i] : 0.0;
    const int kmax = (nval + 3) & ~3;
    ImuJac<double> J;
    imu_eval_core<double>(k, sc, d.imu_u[idx], idt, grav, bias, gy, ac, wl, RrefT, r, true, J);
#pragma unroll
    for (int i = 0; i < 6; ++i) csum += 0.5 * r[i] * r[i];   // (dead lanes: zero weights, zero residual)
    // ---- accelerometer rows: 32 columns, tiles (0,0), (1,0), (1,1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {   // unrolled: the row index must be static (a dynamic index would push ImuJac to scratch)
      double row[32];
      imu_row_accel<double>(J, wl, r, a, row);
      __builtin_amdgcn_wave_barrier();   // the previous phase's operand reads are complete (consumed by its MFMAs)
#pragma unroll
      for (int c = 0; c < 32; ++c) A[lane * 33 + c] = row[c];
      __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the rows are in LDS
      __builtin_amdgcn_wave_barrier();
      for (int k0 = 0; k0 < kmax; k0 += 4) {
        const double lo = A[(k0 + q4) * 33 + l15], hi = A[(k0 + q4) * 33 + 16 + l15];
        acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(lo, lo, acc00, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, lo, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, hi, acc11, 0, 0, 0);
      }
    }
    // ---- gyro rows: 16 compacted columns, one tile
#pragma unroll
    for (int a = 0; a < 3; ++a) {   // unrolled: the row index must be static (a dynamic index would push ImuJac to scratch)
      double row[16];
      imu_row_gyro<double>(J, wl, r, a, row);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 16; ++c) A[lane * 17 + c] = row[c];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      for (int k0 = 0; k0 < kmax; k0 += 4) {
        const double v = A[(k0 + q4) * 17 + l15];
        gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, gacc, 0, 0, 0);
      }
    }
  }
  // ---- the group's share of the cost: fixed-order sum over the lanes (butterfly), one store
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off);
  if (lane == 0) d.imu_cost[gidx] = csum;
  // ---- combine in LDS into the full symmetric 32 x 32 tile, then one coalesced store
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = q4 + 4 * r, col = l15;
    A[row * 32 + col] = acc00[r];
    A[(16 + row) * 32 + 16 + col] = acc11[r];
    A[(16 + row) * 32 + col] = acc10[r];
    A[col * 32 + 16 + row] = acc10[r];     // mirror of the off-diagonal tile
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  {
    const int tc = l15 < 12 ? l15 : (l15 < 15 ? l15 + 12 : 30);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int grow = q4 + 4 * r;
      const int tr = grow < 12 ? grow : (grow < 15 ? grow + 12 : 30);
      A[tr * 32 + tc] += gacc[r];
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  double *tile = d.imu_tiles + (size_t)gidx * 1024;
#pragma unroll
  for (int i = 0; i < 16; ++i) tile[i * 64 + lane] = A[i * 64 + lane];
}

// The MFMA chains of one row phase, two K-steps per trip: the operands of step k + 1 are requested before the MFMAs of step k are issued
// (clock stamps of one group, 3 gyro + 3 accelerometer row phases of a full pass: 6980 + 11372 cycles with the read of step k issued right
// before its MFMA, 6364 + 10384 like this; three steps ahead -- four steps per trip, K rounded to 16 rows -- 6332 + 10828 and the partial
// passes lose to the rounding: not kept; a second gyro accumulator changes nothing either: the chain is not waiting for its own results).
// K is rounded up to a multiple of 8 rows: the rows past the last sample are zero rows (dead lanes write zeros), the buffer has 8 spare rows
// for the last prefetch.
// (The operand fetches and their waits are inline assembly: left to itself the compiler re-loads the carried operand at the top of the
//  next trip -- one ds_read2, one wait, two MFMAs, the very serialisation this removes; volatile loads become flat loads with a full wait
//  each.  The waits carry the operand as an in/out so that the MFMA that consumes it stays behind them; a final lgkmcnt(0) leaves nothing
//  in flight that the compiler's own wait counting does not know about.)
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void imu_chain_gyro(const double *A, int q4, int l15, int kmax, f64x4 &gacc) {
  unsigned addr = (unsigned)(size_t)(A + q4 * 17 + l15);
  double v0, v1;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v0) : "v"(addr));
  for (int k0 = 0; k0 < kmax; k0 += 8) {
    asm volatile("ds_read_b64 %0, %1 offset:544" : "=v"(v1) : "v"(addr));          // step k0 + 4 (4 rows of 17 doubles on)
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(v0));
    gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, gacc, 0, 0, 0);
    addr += 8 * 17 * 8;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v0) : "v"(addr));                     // step k0 + 8
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(v1));
    gacc = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, gacc, 0, 0, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0));
}
__device__ __forceinline__ void imu_chain_accel(const double *A, int q4, int l15, int kmax, f64x4 &acc00, f64x4 &acc10) {
  unsigned addr = (unsigned)(size_t)(A + q4 * 33 + l15);
  f64x2 o0, o1;   // (lo, hi) = columns l15 and 16 + l15 of four rows
  asm volatile("ds_read2_b64 %0, %1 offset1:16" : "=v"(o0) : "v"(addr));
  for (int k0 = 0; k0 < kmax; k0 += 8) {
    asm volatile("ds_read2_b64 %0, %1 offset0:132 offset1:148" : "=v"(o1) : "v"(addr));   // step k0 + 4 (4 rows of 33 doubles on)
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(o0));
    acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(o0[0], o0[0], acc00, 0, 0, 0);
    acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(o0[1], o0[0], acc10, 0, 0, 0);
    addr += 8 * 33 * 8;
    asm volatile("ds_read2_b64 %0, %1 offset1:16" : "=v"(o0) : "v"(addr));                   // step k0 + 8
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(o1));
    acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(o1[0], o1[0], acc00, 0, 0, 0);
    acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(o1[1], o1[0], acc10, 0, 0, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o0));
}

// The fast body's groups: every knot-pair log below 0.5 rad, isotropic accelerometer weights.  Asked in two places (the fast body about its
// own group, k_imu_linearize_rest about every group of its window) that must agree to the bit: the operations are spelled out (no
// contraction choices left to the compiler).
__device__ __forceinline__ bool imu_fast_pred(const double kd[9], const double *imu_w) {
  double mx = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) mx = fmax(mx, __fma_rn(kd[3 * i + 2], kd[3 * i + 2], __fma_rn(kd[3 * i + 1], kd[3 * i + 1], __dmul_rn(kd[3 * i], kd[3 * i]))));
  return mx < 0.25 && imu_w[3] == imu_w[4] && imu_w[3] == imu_w[5];
}
// ---- The product path's body for the usual group (imu_group_fast: knot-pair logs below 0.5 rad, isotropic accelerometer weights).
// Same wave-per-group scheme and row streaming as the general body above, with
//   * the evaluation in stages (factors.hpp, staged form): values, gyro Jacobians -> three row phases, accelerometer Jacobians -> three row
//     phases, so the two 36-entry Jacobians are never live together; small-angle series, no branch in the loop;
//   * global frame (the local frame of the general body is an fp32 device), Jr^-1 of the three knot pairs and their logs in SGPRs;
//   * the accelerometer rows in two 16-column tiles T0 = [rot 12 | ba 3 | r], T1 = [pos 12]: T0^T T0 and T1^T T0 on the matrix cores,
//     T1^T T1 = w^2 sum_s lamA_k lamA_k' I3 from ten per-lane sums (R(t)^T W^2 R(t) = w^2 I): 2 MFMAs per K-step instead of 3;
//   * the next pass's measurements requested before the current pass is evaluated.
// (fp64 MFMA and fp64 VALU instructions share one datapath on gfx950 -- tools/mfma_valu_overlap.hip: one wave's MFMAs and FMAs add up,
//  two waves on a SIMD do not overlap them either -- so the kernel's time is the SUM of its vector and matrix work: both are cut here.)
// A wave WALKS its groups g0, g0 + stride, ... (k_imu_linearize_f64: 2048 waves for the whole batch) and everything the NEXT group's
// record locates -- pair logs and Jr^-1, first knot's rotation, knot positions, bias, gravity, weights, 1 / dt, the window's LM flags, one
// element per lane -- is requested while the CURRENT group is evaluated, and the record after that is on its way as well; the next group's
// first 64 samples are requested by the current group's last pass.  A group's own prologue (three dependent round trips group -> window ->
// data at one wave per SIMD: ~10 k of a group's 55 k cycles, measured) shrinks to a few dozen v_readlane.
struct ImuPre { double pc, kq; int fl; };
__device__ __forceinline__ void imu_prefetch(const Dev<double> &d, int mode, const ImuGroup &g, int lane, ImuPre &o) {
  const bool at_cand = mode == LIN_SPEC;
  const double *s_quat = at_cand ? d.cquat : d.quat, *s_pos = at_cand ? d.cpos : d.pos, *s_bias = at_cand ? d.cbias : d.bias;
  const WinMeta &m = d.wins[g.win];
  const Lm &lm = d.lm[g.win];
  const double *kd = d.lkd + 3 * g.kabs, *kj = d.kjri + 9 * g.kabs;
  const double *q = s_quat + 4 * g.kabs, *pp = s_pos + 3 * g.kabs, *bp = s_bias + 6 * g.babs;
  // pc: lanes 0..8 the pair logs, 9..35 Jr^-1 (row major per pair)
  o.pc = *(lane < 9 ? kd + lane : kj + (min(lane, 35) - 9));
  // kq: 0..3 q_0 | 4..15 the four knot positions | 21..23 gravity | 24..29 bias | 30..35 weights | 36 1 / dt   (unconditional loads on valid addresses)
  const double *src = lane < 4 ? q + lane : lane < 16 ? pp + (lane - 4) : lane < 21 ? q : lane < 24 ? m.gravity + (lane - 21)
                      : lane < 30 ? bp + (lane - 24) : lane < 36 ? m.imu_w + (lane - 30) : &m.inv_dt;
  o.kq = *src;
  // fl: lanes 0..3 the window's LM flags (not written by any linearisation kernel)
  const int32_t *fp = lane == 0 ? &lm.status : lane == 1 ? &lm.step_valid : lane == 2 ? &lm.iter : &lm.ls_active;
  o.fl = *fp;
}
__device__ __forceinline__ void imu_linearize_f64_fast(const Dev<double> &d, int mode, double *A /* LDS [72][33] + 64 */, int g0, int stride, int zero_mode) {
  const int lane = threadIdx.x, q4 = lane >> 4, l15 = lane & 15;
  const size_t Mt = (size_t)d.Mtot;
  int gidx = g0;
  ImuGroup grp = d.groups[gidx];
  ImuPre cur;
  imu_prefetch(d, mode, grp, lane, cur);                 // (the walk's first group: its round trips are exposed once)
  double gyn[3], acn[3], un;   // the next pass's measurements, in flight while the current pass is evaluated
  {
    const int idx = grp.iabs + min(lane, grp.count - 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) { gyn[i] = d.imu_meas[(size_t)i * Mt + idx]; acn[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
    un = d.imu_u[idx];
  }
  bool has_next = gidx + stride < d.Gtot;
  ImuGroup grpn = d.groups[has_next ? gidx + stride : gidx];
  for (;;) {
  // ---- the group after the next one's record and the next one's constants: on their way during this group
  const bool has_next2 = has_next && gidx + 2 * stride < d.Gtot;
  const ImuGroup grpn2 = d.groups[has_next2 ? gidx + 2 * stride : gidx];
  ImuPre nxt;
  imu_prefetch(d, mode, grpn, lane, nxt);
  bool nmeas = false;          // the next group's first pass has been requested (by this group's last pass)
  do {
  const int w = grp.win;
  const int base = grp.iabs;
  // the group's constants: knot-pair logs and Jr^-1 (used a dozen times per pass) in scalar registers, the rest (used once or twice per
  // pass) in LDS behind the row buffer -- [0..11] knot positions relative to knot 0, [12..20] R_0^T, [21..23] gravity, [24..29] bias,
  // [30..35] weights.  (All of them in scalar registers overflow the SGPR file: 250 v_readlane per pass to fetch them back.)
  SegConstS<double> sc;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    sc.d[i] = mk<double>(readlane_d(cur.pc, 3 * i), readlane_d(cur.pc, 3 * i + 1), readlane_d(cur.pc, 3 * i + 2));
#pragma unroll
    for (int e = 0; e < 9; ++e) sc.JrI[i].m[e] = readlane_d(cur.pc, 9 + 9 * i + e);
  }
  // ---- the window: LM state
  const int f_status = __builtin_amdgcn_readlane(cur.fl, 0), f_valid = __builtin_amdgcn_readlane(cur.fl, 1), f_iter = __builtin_amdgcn_readlane(cur.fl, 2),
            f_ls = __builtin_amdgcn_readlane(cur.fl, 3);
  if (!(f_status == 0 && (mode != LIN_SPEC || f_valid != 0))) break;                                   // lin_run
  const bool jac = !(mode == COST_AT_X || (mode == LIN_SPEC && f_iter >= d.prm.max_iters && f_ls == 0));   // lin_cost_only: (uniform) the last allowed iteration only costs its candidate
  // is this group the fast body's?  (imu_group_fast, decided HERE from the pair logs and the weights already in registers)
  {
    const double kd9[9] = {sc.d[0].x, sc.d[0].y, sc.d[0].z, sc.d[1].x, sc.d[1].y, sc.d[1].z, sc.d[2].x, sc.d[2].y, sc.d[2].z};
    const double w6[6] = {0.0, 0.0, 0.0, readlane_d(cur.kq, 33), readlane_d(cur.kq, 34), readlane_d(cur.kq, 35)};
    if (!imu_fast_pred(kd9, w6)) break;   // (uniform) left to k_imu_linearize_rest
  }
  long long *dbg = (d.dbg && gidx == 5000 && jac) ? d.dbg + 64 : nullptr;   // CTVIO_DEBUG_STAMPS: clock64 of lane 0 at the phase boundaries
  int dbi = 0;
#define CTV_ISTAMP(x) do { if (dbg && lane == 0 && dbi < 16) dbg[dbi++] = clock64() + (long long)((x) * 0.0); } while (0)
  CTV_ISTAMP(0.0);
  double *gc = A + 72 * 33;   // (8 spare rows behind the 64: the chains' last prefetch)
  {
    double gcv = cur.kq;      // lanes 21..35: gravity, bias, weights as requested
    const M3<double> R0 = q2R(qmk<double>(readlane_d(cur.kq, 0), readlane_d(cur.kq, 1), readlane_d(cur.kq, 2), readlane_d(cur.kq, 3)));
    const double pk = __shfl(cur.kq, 4 + min(lane, 11)), p0 = __shfl(cur.kq, 4 + min(lane, 11) % 3);
    if (lane < 12) gcv = pk - p0;
    else if (lane < 21) {   // R_0^T, row major (a select chain: no dynamically indexed register array)
      const int e = lane - 12, src = 3 * (e % 3) + e / 3;
#pragma unroll
      for (int i = 0; i < 9; ++i) gcv = src == i ? R0.m[i] : gcv;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 36) gc[lane] = gcv;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  const double idt = readlane_d(cur.kq, 36);
  double csum = 0.0;
  if (!jac) {   // residuals only
    for (int c0 = 0; c0 < grp.count; c0 += 64) {
      const bool live = c0 + lane < grp.count;
      const int idx = base + min(c0 + lane, grp.count - 1);
      double gy[3], ac[3], r[6], wl[6];
#pragma unroll
      for (int i = 0; i < 3; ++i) { gy[i] = d.imu_meas[(size_t)i * Mt + idx]; ac[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
#pragma unroll
      for (int i = 0; i < 6; ++i) wl[i] = live ? gc[30 + i] : 0.0;
      ImuMid3<double> md;
      imu_eval_values3<double>(gc, sc, d.imu_u[idx], idt, gy, ac, wl, r, md);
#pragma unroll
      for (int i = 0; i < 6; ++i) csum += 0.5 * r[i] * r[i];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) csum += __shfl_xor(csum, off);
    if (lane == 0) d.imu_cost[gidx] = csum;
    break;
  }
  f64x4 acc00 = {0.0, 0.0, 0.0, 0.0}, acc10 = {0.0, 0.0, 0.0, 0.0}, gacc = {0.0, 0.0, 0.0, 0.0};
  double spp[10] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int c0 = 0; c0 < grp.count; c0 += 64) {
    CTV_ISTAMP(csum);
    const int nval = min(64, grp.count - c0);
    const bool live = lane < nval;
    double gy[3], ac[3], r[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) { gy[i] = gyn[i]; ac[i] = acn[i]; }
    const double u = un;
    {
      // the next pass's samples -- after the group's last pass the NEXT GROUP's first ones (without one, a valid sample that is dropped)
      const bool last = c0 + 64 >= grp.count;
      const int idx = (last && has_next) ? grpn.iabs + min(lane, grpn.count - 1) : base + min(c0 + 64 + lane, grp.count - 1);
      nmeas = last;
#pragma unroll
      for (int i = 0; i < 3; ++i) { gyn[i] = d.imu_meas[(size_t)i * Mt + idx]; acn[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
      un = d.imu_u[idx];
    }
    // lanes past the end of the group evaluate a clamped sample with ZERO weights: every row of w .* [J | r] is then exactly zero
    double wl[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wl[i] = live ? gc[30 + i] : 0.0;
    const int kmax = (nval + 3) & ~3;
    ImuMid3<double> md;
    imu_eval_values3<double>(gc, sc, u, idt, gy, ac, wl, r, md);
#pragma unroll
    for (int i = 0; i < 6; ++i) csum += 0.5 * r[i] * r[i];   // (dead lanes: zero weights, zero residual)
    CTV_ISTAMP(csum);
    {
      M3<double> Jw[4];
      imu_jac_gyro3<double>(md, sc, Jw);
      CTV_ISTAMP(Jw[3].m[8]);
#pragma unroll
      for (int a = 0; a < 3; ++a) {   // unrolled: the row index must be static
        double row[16];
        imu_row_gyro2<double>(Jw, wl, r, a, row);
        __builtin_amdgcn_wave_barrier();   // the previous phase's operand reads are complete (consumed by its MFMAs)
#pragma unroll
        for (int c = 0; c < 16; ++c) A[lane * 17 + c] = row[c];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the rows are in LDS
        __builtin_amdgcn_wave_barrier();
        imu_chain_gyro(A, q4, l15, kmax, gacc);
      }
    }
    CTV_ISTAMP(gacc[0]);
    {
      M3<double> Ja[4], Rinv_g;
      imu_jac_accel3<double>(md, sc, gc, Ja, Rinv_g);
      CTV_ISTAMP(Ja[3].m[8] + Rinv_g.m[8]);
      {
        double la[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) la[kk] = wl[3] * md.lamA[kk];
        int e = 0;
#pragma unroll
        for (int ka = 0; ka < 4; ++ka)
#pragma unroll
          for (int kb = 0; kb <= ka; ++kb) { spp[e] += la[ka] * la[kb]; ++e; }
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        double row[28];
        imu_row_accel3<double>(Ja, md.lamA, Rinv_g, wl, r, a, row);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 28; ++c) A[lane * 33 + c] = row[c];   // (columns 28..31 feed accumulator rows nobody reads)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        imu_chain_accel(A, q4, l15, kmax, acc00, acc10);
      }
    }
  }
  CTV_ISTAMP(acc00[0] + acc10[0]);
  // ---- combine in LDS into the full symmetric 32 x 32 tile in the local column order [rot 12 | pos 12 | bg 3 | ba 3 | r | -]
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 16; ++i) A[i * 64 + lane] = 0.0;
  {
    // eleven sums over the 64 lanes in a fixed order -- the ten of the pos x pos block and the group's share of the cost -- through the
    // free half of the buffer: lane (e, part) adds 16 lanes' values, two butterfly steps join the four parts (one LDS round trip for all
    // of them instead of a six-step butterfly per value)
    double *S = A + 1024;
#pragma unroll
    for (int e = 0; e < 10; ++e) S[e * 64 + lane] = spp[e];
    S[10 * 64 + lane] = csum;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    const int e = min(lane >> 2, 10), part = lane & 3;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += S[e * 64 + part * 16 + i];
    t += __shfl_xor(t, 1);
    t += __shfl_xor(t, 2);
    __builtin_amdgcn_wave_barrier();
    if (lane < 40 && part == 0) S[704 + e] = t;
    if (lane == 40) d.imu_cost[gidx] = t;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  {
    // T0 (accelerometer rows: rot 12 | ba 3 | r) and the gyro tile (rot 12 | bg 3 | r) share the accumulator layout: where neither index
    // is a bias one the two land on the same entry and are added in registers; a bias index sends them to the ba / bg columns
    const int c0 = l15 < 12 ? l15 : (l15 < 15 ? l15 + 15 : 30);   // T0 index -> local column (ba at 27..29)
    const int cg = l15 < 12 ? l15 : (l15 < 15 ? l15 + 12 : 30);   // gyro tile index -> local column (bg at 24..26)
    const bool cb = l15 >= 12 && l15 < 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = q4 + 4 * r;
      const int r0 = t < 12 ? t : (t < 15 ? t + 15 : 30), rg = t < 12 ? t : (t < 15 ? t + 12 : 30);
      const bool shared = !cb && !(t >= 12 && t < 15);
      A[r0 * 32 + c0] = shared ? acc00[r] + gacc[r] : acc00[r];
      if (!shared) A[rg * 32 + cg] = gacc[r];
      if (t < 12) { A[(12 + t) * 32 + c0] = acc10[r]; A[c0 * 32 + 12 + t] = acc10[r]; }
    }
    // lane (ka, kb, b) < 48 places one entry of the pos x pos block
    const int ka = lane / 12, kb = (lane / 3) & 3, b = lane % 3;
    const int hi = max(ka, kb), lo = min(ka, kb);
    if (lane < 48) A[(12 + 3 * ka + b) * 32 + 12 + 3 * kb + b] = A[1024 + 704 + hi * (hi + 1) / 2 + lo];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  double *tile = d.imu_tiles + (size_t)gidx * 1024;
#pragma unroll
  for (int i = 0; i < 8; ++i)   // 16 bytes per lane: 8 stores of 1 KiB (under load a store costs ~100 cycles whatever its width)
    *reinterpret_cast<double2 *>(tile + i * 128 + 2 * lane) = *reinterpret_cast<const double2 *>(A + i * 128 + 2 * lane);
  // (last: the memory counter is in-order, a load issued after these stores would wait for their acknowledgement)
  imu_zero_share(d, mode, grp, gidx, zero_mode);
  CTV_ISTAMP(0.0);
#undef CTV_ISTAMP
  } while (false);
  // ---- on to the wave's next group
  if (!has_next) break;
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();          // (the tile copy-out has read the LDS buffer before the next group writes its constants)
  if (!nmeas) {                             // this group left early: the next one's first pass has not been asked for yet
    const int idx = grpn.iabs + min(lane, grpn.count - 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) { gyn[i] = d.imu_meas[(size_t)i * Mt + idx]; acn[i] = d.imu_meas[(size_t)(3 + i) * Mt + idx]; }
    un = d.imu_u[idx];
  }
  gidx += stride;
  grp = grpn; grpn = grpn2; cur = nxt;
  has_next = has_next2;
  }
}

// One wave per SIMD: the evaluation needs ~430 fp64-pair registers; with a 512-register budget the overflow lives in AGPRs.
// (Two waves per SIMD with the overflow spilled to scratch was measured 3x slower: 1690 vs 540 us per 1024 windows.)
// The fast body evaluates the small-angle series only: a group whose knot-pair logs reach 0.5 rad (28.6 degrees between two knots 50 ms
// apart) takes the general body.  It also takes the pos x pos block from R(t)^T W^2 R(t) = w^2 I: isotropic accelerometer weights (the
// reference's: one scalar per sensor) -- any other weighting takes the general body as well.
__device__ __forceinline__ bool imu_group_fast(const Dev<double> &d, int gidx) {
  const ImuGroup grp = d.groups[gidx];
  const double *kd = d.lkd + 3 * grp.kabs;
  const double kd9[9] = {kd[0], kd[1], kd[2], kd[3], kd[4], kd[5], kd[6], kd[7], kd[8]};
  return imu_fast_pred(kd9, d.wins[grp.win].imu_w);
}
// The groups the fast body leaves out are picked up by k_imu_linearize_rest (one wave per WINDOW: its lanes look at the window's groups,
// the wave then takes the flagged ones in turn -- 12 us per launch when there is nothing to do, which is the rule): the two bodies in
// one kernel cost the fast one registers.
// (general_only: every group through the general body -- ctvio_options.use_mfma = 2 / CTVIO_IMU_GENERAL=1, the tests' way into it)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_imu_linearize_f64(Dev<double> d, int mode, int general_only, int zero_mode) {
  extern __shared__ __attribute__((aligned(32))) unsigned char smraw[];
  if (!general_only) imu_linearize_f64_fast(d, mode, reinterpret_cast<double *>(smraw), blockIdx.x, gridDim.x, zero_mode);   // (skips the groups that are not its own)
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_imu_linearize_rest(Dev<double> d, int mode, int general_only, int zero_mode) {
  extern __shared__ __attribute__((aligned(32))) unsigned char smraw[];
  const WinMeta &m = d.wins[blockIdx.x];
  for (int g0 = 0; g0 < m.ngrp; g0 += 64) {
    const int gl = g0 + (int)threadIdx.x;
    const bool need = gl < m.ngrp && (general_only || !imu_group_fast(d, m.grp0 + gl));
    unsigned long long todo = __ballot(need);
    while (todo) {
      const int b = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      imu_linearize_f64_body(d, mode, reinterpret_cast<double *>(smraw), m.grp0 + g0 + b, zero_mode);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Scatter the group tiles into Hpp (lower triangle, fp64) and g.
// One workgroup per WINDOW walking its groups (one per group was 43 k workgroups of 195 useful threads: dispatch-bound).
template <class T> __global__ __launch_bounds__(256) void k_assemble_imu(Dev<T> d, int mode) {
  const int w = blockIdx.x;
  if (!lin_run(d.lm[w], mode) || lin_cost_only(d.lm[w], mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  const int K = m.K, ldh = m.ldh, tg = lin_target(d.lm[w], mode);
  double *Hpp = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  if (m.vis_lds) {
    // the knot x knot part is accumulated in LDS by k_assemble_vis; what is left is the bias rows (6 x 24 against the
    // knots, the 6 x 6 lower triangle) and the gradient: 195 entries per group -- one load, one atomic each
    for (int i = threadIdx.x; i < 195 * m.ngrp; i += 256) {
      const int gi = i / 195, t = i - 195 * gi;
      const ImuGroup grp = d.groups[m.grp0 + gi];
      const T *tile = d.imu_tiles + (size_t)(m.grp0 + gi) * 1024;
      int a, b;
      if (t < 144) { a = 24 + t / 24; b = t % 24; }
      else if (t < 165) {
        const int q = t - 144;                     // lower triangle of the bias block, row-major
        const int r = q < 1 ? 0 : q < 3 ? 1 : q < 6 ? 2 : q < 10 ? 3 : q < 15 ? 4 : 5;
        a = 24 + r; b = 24 + q - r * (r + 1) / 2;
      } else { a = t - 165; b = 30; }
      // (the tile is symmetric: the gradient column is read as row 30, next to the bias rows -- 7 consecutive rows of the tile instead of a
      //  cache line of every row)
      const double v = (double)(b == 30 ? tile[30 * 32 + a] : tile[a * 32 + b]);
      const int ga = imu_col(a, grp.s, K, grp.bias);
      if (b == 30) { atomicAdd(&g[ga], v); continue; }
      const int gb = imu_col(b, grp.s, K, grp.bias);
      atomicAdd(&Hpp[(long long)max(ga, gb) * ldh + min(ga, gb)], v);
    }
    return;
  }
  for (int i = threadIdx.x; i < 31 * 30 * m.ngrp; i += 256) {
    const int gi = i / 930, e = i - 930 * gi;
    const ImuGroup grp = d.groups[m.grp0 + gi];
    const T *tile = d.imu_tiles + (size_t)(m.grp0 + gi) * 1024;
    const int b = e / 30, a = e % 30;  // a < 30 : unknown row; b <= 30
    const double v = (double)tile[a * 32 + b];
    const int ga = imu_col(a, grp.s, K, grp.bias);
    if (b == 30) { atomicAdd(&g[ga], v); continue; }
    const int gb = imu_col(b, grp.s, K, grp.bias);
    if (ga >= gb) atomicAdd(&Hpp[(long long)ga * ldh + gb], v);
  }
}

// ------------------------------------------------------------------------------------------------ visual
// Entry (row = 2 * local column + residual row, < 100; 100 / 101 = the residual) of the robust-corrected 2 x 50 Jacobian of the block
// in slot v with anchor `anc`, rebuilt from the block record and the anchor record (factors.hpp): the cross-check assembly's input.
__device__ __forceinline__ double vis_J_entry(const Dev<double> &d, int row, unsigned v, unsigned anc) {
  const double *J = d.Jt + (size_t)v * VT_ROWS;
  if (row >= 100) return J[VB_RES + row - 100];
  const int col = row >> 1, rr = row & 1;
  if (col >= 48) return J[(col == 48 ? VB_RHO : VB_LD) + rr];
  const double *rec = d.arec + (size_t)anc * AREC;
  if (col < 12) return J[VB_AT + rr] * rec[AR_GR + 3 * col] + J[VB_AT + 2 + rr] * rec[AR_GR + 3 * col + 1] + J[VB_AT + 4 + rr] * rec[AR_GR + 3 * col + 2];
  if (col < 24) { const int c = col - 12; return rec[AR_CP0 + c / 3] * J[VB_AT + 2 * (c % 3) + rr]; }
  if (col < 36) return J[VB_JROT + 2 * (col - 24) + rr];
  const int c = col - 36;
  return -(J[VB_CP1 + c / 3] * J[VB_AT + 2 * (c % 3) + rr]);
}

// time -> (first active knot, u) in integer ns (reference spline_segment.h:83-85); the line delay is
// truncated to integer ns exactly as image_feature_factor.h:72.
__device__ __forceinline__ void vis_times(const WinMeta &m, long long t_rel, int row, double ld, int &s, double &u) {
  const long long ld_ns = (long long)(ld * 1e9);
  const long long tau = t_rel + (long long)row * ld_ns;
  s = (int)(tau / m.dt_ns);
  u = (double)(tau % m.dt_ns) / (double)m.dt_ns;
}

// One lane per ANCHOR (the i end shared by a feature's blocks: factors.hpp): the record of the state being linearised.  The usual wave --
// every knot-pair log of its anchors below 0.5 rad (a ballot) -- takes the series-only evaluation (no branch, no closed-form code on the
// path), the others the general one; both in the global frame.
constexpr int AREC_LD = AREC + 1;   // odd LDS stride of the staged records
__global__ __launch_bounds__(64) void k_vis_anchor(Dev<double> d, int mode) {
  // the records of the wave's 64 anchors are staged in LDS and written as ONE contiguous region with 16-byte stores (a lane writing
  // its own 400-byte record entry by entry costs 50 scattered partial-line stores: 229 MB of write traffic for 164 MB of records)
  __shared__ __attribute__((aligned(16))) double srec[64 * AREC_LD];
  const int a = blockIdx.x * 64 + threadIdx.x;
  bool run = false;
  if (a < d.Atot) run = lin_run(d.lm[d.a_win[a]], mode);
  const unsigned long long run_mask = __ballot(run);
  if (run_mask == 0) return;                   // (wave-uniform)
  if (run) {
  const int w = d.a_win[a];
  const Lm &lm = d.lm[w];
  const WinMeta &m = d.wins[w];
  const bool jac = !lin_cost_only(lm, mode, d.prm);
  const bool at_cand = mode == LIN_SPEC;
  const double *quat = at_cand ? d.cquat : d.quat, *pos = at_cand ? d.cpos : d.pos, *rho = at_cand ? d.crho : d.rho, *ldp = at_cand ? d.cld : d.ld;
  int si;
  double ui;
  const int rowi = d.a_row[a];
  vis_times(m, d.a_t[a], rowi, ldp[w], si, ui);
  si = max(0, min(si, m.K - 4));   // host validated the worst case; clamp keeps loads in range regardless
  SegConstLazy<double> sc;   // Jr^-1 of the knot pairs stays in the table until the streamed Jacobians need it
  seg_const_lazy(d.lkd + 3 * (m.knot0 + si), d.kjri + 9 * (m.knot0 + si), sc);
  double dmax = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) dmax = fmax(dmax, dot(sc.d[i], sc.d[i]));
  const bool small = __ballot(dmax >= 0.25) == 0ull;
  const double *qi = quat + 4 * (m.knot0 + si), *pi = pos + 3 * (m.knot0 + si);
  const Q4<double> q0 = qmk<double>(qi[0], qi[1], qi[2], qi[3]);
  V3<double> p[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] = mk<double>(pi[3 * i], pi[3 * i + 1], pi[3 * i + 2]);
  const Q4<double> q_CI = qmk<double>(m.q_CI[0], m.q_CI[1], m.q_CI[2], m.q_CI[3]);
  const V3<double> p_CI = mk<double>(m.p_CI[0], m.p_CI[1], m.p_CI[2]);
  double *rec = srec + AREC_LD * threadIdx.x;
  if (!jac)                                    // (a costed record carries p_G alone; the rest goes out as zeros, not as stale LDS)
    for (int e = 0; e < AREC; ++e) rec[e] = 0.0;
  const double pix = d.a_obs[a], piy = d.a_obs[(size_t)d.Atot + a], d_inv = rho[m.lm0 + d.a_lm[a]];
  if (small) vis_anchor_eval<true>(q0, p, sc, ui, m.inv_dt, q_CI, p_CI, pix, piy, (double)rowi, d_inv, jac, rec);
  else vis_anchor_eval<false>(q0, p, sc, ui, m.inv_dt, q_CI, p_CI, pix, piy, (double)rowi, d_inv, jac, rec);
  d.a_s[a] = si;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);          // (one wave per workgroup: the wave's own LDS writes have completed)
  __builtin_amdgcn_wave_barrier();
  {
    const int lane = threadIdx.x;
    double *dst = d.arec + (size_t)(blockIdx.x * 64) * AREC;
    constexpr int HP = AREC / 2;                 // pairs per record
#pragma unroll 5
    for (int k = 0; k < HP; ++k) {               // 64 * HP pairs, 64 per store (cost-only records carry p_G alone: the rest is never read)
      const int i = k * 64 + lane, bl = i / HP, r = 2 * (i - bl * HP);
      VecN<double, 2> pr;
      pr.v[0] = srec[AREC_LD * bl + r];
      pr.v[1] = srec[AREC_LD * bl + r + 1];
      if ((run_mask >> bl) & 1ull) *reinterpret_cast<VecN<double, 2> *>(dst + (size_t)bl * AREC + r) = pr;
    }
  }
}

// k_vis_eval<LIN> stages the records of its 64 blocks in LDS, block-major like the copy in HBM ([64][VT_LD]: this lane's block starts
// at J[0]; the odd stride spreads the lanes over the banks) and forms the landmark rows from it after the evaluation.
constexpr int VT_LD = VT_ROWS + 1;
struct VisRecSink {
  double *J;
  __device__ __forceinline__ void put(int e, double v) { J[e] = v; }
};
struct VisNullSink {
  __device__ __forceinline__ void put(int, double) {}
};

// One lane per visual block, landmark-major: evaluate the block's own (j) end against its anchor's record -- r~ and the record of J~
// (robust-corrected), materialised block-major -- and form the rows of W, Hll, g_rho of the wave's landmarks into the normal-equation
// set the mode selects.  The wave's share of the cost goes to Dev::vis_cost (a window's block slots start on a wave boundary: one window
// per wave).  A window on its last allowed iteration is only costed (residuals, no Jacobians, nothing else written).
constexpr int VIS_LDS_BYTES = 64 * VT_LD * 8;   // the records of a wave's 64 blocks, afterwards the fp64 rows of W of the wave's landmarks
__device__ __forceinline__ void vis_eval_body(const Dev<double> &d, int mode, unsigned char *smt, long long *rowoff, int *rlm, int vblock) {
  const int v = vblock * 64 + threadIdx.x;
  const long long t_entry = d.dbg ? clock64() : 0ll;
  constexpr int LDS_BYTES = VIS_LDS_BYTES;
  double *wcs = reinterpret_cast<double *>(smt);
  const bool at_cand = mode == LIN_SPEC;
  const double *quat = at_cand ? d.cquat : d.quat, *pos = at_cand ? d.cpos : d.pos, *ldp = at_cand ? d.cld : d.ld;
  double c = 0.0;
  int w = -1, ksj = 0, my_lm = -1, my_anc = -1, tg = 0;
  bool on = false, costed = false;
  if (v < d.Vtot) {
    w = d.v_win[v];
    const Lm &lm = d.lm[max(w, 0)];
    const bool run = w >= 0 && lin_run(lm, mode);
    if (run) {
      const WinMeta &m = d.wins[w];
      const bool jac = !lin_cost_only(lm, mode, d.prm);
      tg = lin_target(lm, mode);
      int sj;
      double uj;
      const int rowj = d.v_rowj[v];
      vis_times(m, d.v_tj[v], rowj, ldp[w], sj, uj);
      sj = max(0, min(sj, m.K - 4));  // host validated the worst case; clamp keeps loads in range regardless
      SegConstLazy<double> scj;   // Jr^-1 of the knot pairs stays in the table until the streamed Jacobians need it
      seg_const_lazy(d.lkd + 3 * (m.knot0 + sj), d.kjri + 9 * (m.knot0 + sj), scj);
      // The usual wave: every knot-pair log of its blocks below 0.5 rad -> series-only evaluation (uniform choice: a ballot over the
      // running lanes); otherwise the general form.  Global frame, absolute positions (fp64).
      double dmax = 0.0;
#pragma unroll
      for (int i = 0; i < 3; ++i) dmax = fmax(dmax, dot(scj.d[i], scj.d[i]));
      const bool small = __ballot(dmax >= 0.25) == 0ull;
      const double *qj = quat + 4 * (m.knot0 + sj), *pj = pos + 3 * (m.knot0 + sj);
      const Q4<double> q0 = qmk<double>(qj[0], qj[1], qj[2], qj[3]);
      V3<double> p[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = mk<double>(pj[3 * i], pj[3 * i + 1], pj[3 * i + 2]);
      const int anc = d.v_anc[v];
      const double *rec = reinterpret_cast<const double *>(__builtin_assume_aligned(d.arec + (size_t)anc * AREC, 16));
      M3<double> RCIT;
      {
        const M3<double> R = q2R(qmk<double>(m.q_CI[0], m.q_CI[1], m.q_CI[2], m.q_CI[3]));
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) RCIT.m[3 * a + b] = R.m[3 * b + a];
      }
      const V3<double> p_CI = mk<double>(m.p_CI[0], m.p_CI[1], m.p_CI[2]);
      const double pjx = d.v_obs[v], pjy = d.v_obs[(size_t)d.Vtot + v], ca = d.v_cauchy[v];
      double r[2];
      if (jac) {
        VisRecSink sink{wcs + VT_LD * threadIdx.x};
        on = true;
        my_lm = d.v_lm[v];
        my_anc = anc;
        if (small) c = vis_block_eval<true>(rec, q0, p, scj, uj, m.inv_dt, RCIT, p_CI, m.img_w, ca, pjx, pjy, (double)rowj, r, true, sink);
        else c = vis_block_eval<false>(rec, q0, p, scj, uj, m.inv_dt, RCIT, p_CI, m.img_w, ca, pjx, pjy, (double)rowj, r, true, sink);
        sink.J[VB_RES] = r[0]; sink.J[VB_RES + 1] = r[1];
        d.vsj[v] = sj;
        ksj = sj;
      } else {
        VisNullSink nsink;
        costed = true;
        c = vis_block_eval<false>(rec, q0, p, scj, uj, m.inv_dt, RCIT, p_CI, m.img_w, ca, pjx, pjy, (double)rowj, r, false, nsink);
      }
    } else {
      w = -1;
    }
  }
  {
    const int lane = threadIdx.x;
    const unsigned long long on_mask = __ballot(on);
    if (on_mask != 0 || __any(costed)) {       // the wave's share of the cost: fixed-order sum over the lanes, one store
      double cs = c;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) cs += __shfl_xor(cs, off);
      if (lane == 0) d.vis_cost[vblock] = cs;
    }
    if (on_mask == 0) return;                  // (wave-uniform)
    // the wave's window (a window's slots start on a wave boundary) and its normal-equation set
    const int first_on = __ffsll((long long)on_mask) - 1;
    const int tgw = __builtin_amdgcn_readfirstlane(__shfl(tg, first_on)), wu = __builtin_amdgcn_readfirstlane(__shfl(w, first_on));
    const WinMeta &mu = d.wins[wu];
    const int P = mu.P, K6 = 6 * mu.K, ldw = mu.ldw, lm0 = mu.lm0, u0 = mu.u0;
    const long long W0 = mu.W0;
    double *Wset = d.WS[tgw];
    double *Hllset = d.HllS[tgw], *gset = d.gS[tgw];
    // One wave per workgroup: LDS hand-overs only need the wave's own LDS operations to have completed.  (__syncthreads() also
    // waits for vmcnt(0), i.e. for the J~ and W stores in flight to be acknowledged -- ~5 us per barrier here, measured.)
#define LDS_SYNC() do { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); } while (0)
    long long *dbg = (d.dbg && vblock == 1000) ? d.dbg + 32 : nullptr;   // (profiling aid: clock stamps of one wave)
    if (dbg && lane == 0) { dbg[-1] = t_entry; dbg[0] = clock64() + (long long)(c * 0); }
    LDS_SYNC();
    // ---- this lane's contributions to its landmark's row of W.  With jr = J~_rho (2) and n3 = A~^T jr (3): the columns of the block's own
    //      (j) end are jr^T J~_rot and -cp1[k] n3; the anchor end's are (sum over the anchor's blocks of n3)^T [GR | cp0 (x) I] -- formed
    //      once per anchor from the record; line delay, Hll, g_rho ride with that sum.
    double wj[24], s6[6];
    {
      const double *Jl = wcs + VT_LD * lane;
      const double jr0 = Jl[VB_RHO], jr1 = Jl[VB_RHO + 1];
#pragma unroll
      for (int cc = 0; cc < 12; ++cc) wj[cc] = jr0 * Jl[VB_JROT + 2 * cc] + jr1 * Jl[VB_JROT + 2 * cc + 1];
#pragma unroll
      for (int b = 0; b < 3; ++b) s6[b] = jr0 * Jl[VB_AT + 2 * b] + jr1 * Jl[VB_AT + 2 * b + 1];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double cp1 = Jl[VB_CP1 + k];
#pragma unroll
        for (int b = 0; b < 3; ++b) wj[12 + 3 * k + b] = -(cp1 * s6[b]);
      }
      s6[3] = jr0 * Jl[VB_LD] + jr1 * Jl[VB_LD + 1];
      s6[4] = jr0 * jr0 + jr1 * jr1;
      s6[5] = jr0 * Jl[VB_RES] + jr1 * Jl[VB_RES + 1];
    }
    // the anchor record's GR and cp0 again (every lane asks for its own anchor's: same lines as during the evaluation; only the head
    // lane of an anchor uses them) -- requested here, consumed after the copy-out below, which hides the round trip
    double hg[40];
    int ksi;
    {
      const double *rec = reinterpret_cast<const double *>(__builtin_assume_aligned(d.arec + (size_t)max(my_anc, 0) * AREC, 16));
#pragma unroll
      for (int e = 0; e < 40; ++e) hg[e] = rec[AR_GR + e];     // GR[12][3], cp0[4]: entries 3 .. 42
      ksi = d.a_s[max(my_anc, 0)];
    }
    // ---- rows of W.  A landmark's blocks are consecutive lanes (the host keeps a landmark inside one wave), anchor by anchor.
    const int prev_lm = __shfl_up(my_lm, 1), prev_anc = __shfl_up(my_anc, 1);
    const bool head = on && (lane == 0 || prev_lm != my_lm), head_a = on && (lane == 0 || prev_anc != my_anc);
    const unsigned long long heads = __ballot(head), heads_a = __ballot(head_a);
    const int ord = __popcll(heads & ((2ull << lane) - 1ull)) - 1;     // ordinal of this lane's landmark in the wave
    const int nlm = __popcll(heads);
    const int ha = 63 - __clzll((long long)(heads_a & ((2ull << lane) - 1ull)));       // head lane of this lane's anchor
    int maxlen = on ? lane - ha + 1 : 0;                                               // longest anchor of the wave (uniform)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    for (int off = 1; off < maxlen; off <<= 1) {     // segmented sums over the lanes of an anchor (an LDS atomic of several lanes on ONE
      const int oh = __shfl_down(on ? ha : -1, off);  // address costs ~64 cycles per lane)
      const bool take = on && (lane + off < 64) && oh == ha;
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) { const double o = __shfl_down(s6[cc], off); s6[cc] += take ? o : 0.0; }
    }
    if (dbg && lane == 0) dbg[1] = clock64() + (long long)(s6[0] * 0);
    // ---- the records go out block-major: the 64 x VT_ROWS entries of the wave's blocks are one contiguous region, written as pairs
    //      of entries (16 bytes per lane, 1 KiB per store: under load a store costs ~100 cycles whatever its width, measured)
    {
      double *dst = d.Jt + (size_t)(vblock * 64) * VT_ROWS;
      constexpr int HP = VT_ROWS / 2;            // pairs per block
#pragma unroll 4
      for (int k = 0; k < HP; ++k) {             // 64 * HP pairs, 64 per store
        const int i = k * 64 + lane, bl = i / HP, r = 2 * (i - bl * HP);
        VecN<double, 2> pr;
        pr.v[0] = wcs[VT_LD * bl + r];
        pr.v[1] = wcs[VT_LD * bl + r + 1];
        if ((on_mask >> bl) & 1ull) *reinterpret_cast<VecN<double, 2> *>(dst + (size_t)bl * VT_ROWS + r) = pr;
      }
    }
    if (dbg && lane == 0) dbg[2] = clock64();
    // the anchor end's 24 columns from the segmented sum (used by the head lane of the anchor)
    double wi[24];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int cc = 3 * k + b;
        wi[cc] = s6[0] * hg[3 * cc] + s6[1] * hg[3 * cc + 1] + s6[2] * hg[3 * cc + 2];
        wi[12 + cc] = hg[36 + k] * s6[b];
      }
    // The buffer becomes NR fp64 rows ([0, K6) knot columns, K6 line delay, K6 + 1 Hll, K6 + 2 g_rho); every lane adds the 24 values of
    // its own end into the row of its landmark (LDS atomics: the ends of different blocks may share knots), the head lane of every
    // anchor the anchor end's 24 + 3; NR landmarks per sweep; then the knot and line-delay columns of every row, Hll and g_rho are
    // written: W is complete when this kernel ends.
    double *rows = reinterpret_cast<double *>(smt);
    const int RS = K6 + 3;                                             // odd row stride (K6 is even)
    const int NR = max(1, min(nlm, (int)(LDS_BYTES / 8) / RS));
    if (head) { rowoff[ord] = W0 + (long long)my_lm * ldw; rlm[ord] = my_lm; }
    LDS_SYNC();   // every lane has read its record, the copy-out has read them all
    for (int c0 = 0; c0 < nlm; c0 += NR) {
      const int nr = min(NR, nlm - c0);
      for (int i = 2 * lane; i < nr * RS; i += 128) *reinterpret_cast<VecN<double, 2> *>(rows + i) = VecN<double, 2>{{0.0, 0.0}};   // (NR RS + 1 doubles fit)
      LDS_SYNC();
      if (on && ord >= c0 && ord < c0 + nr) {
        double *row = rows + (size_t)(ord - c0) * RS;
        if (head_a) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
              atomicAdd(&row[6 * (ksi + k) + b], wi[3 * k + b]);
              atomicAdd(&row[6 * (ksi + k) + 3 + b], wi[12 + 3 * k + b]);
            }
          atomicAdd(&row[K6], s6[3]);
          atomicAdd(&row[K6 + 1], s6[4]);
          atomicAdd(&row[K6 + 2], s6[5]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            atomicAdd(&row[6 * (ksj + k) + b], wj[3 * k + b]);
            atomicAdd(&row[6 * (ksj + k) + 3 + b], wj[12 + 3 * k + b]);
          }
      }
      LDS_SYNC();
      if (dbg && lane == 0) dbg[3 + 2 * (c0 / NR)] = clock64();
      // write-out: the nr rows' knot columns as ONE flat list of 16-byte column pairs (K6 is even, a row starts on a 256-byte boundary),
      // 64 pairs per store instruction -- row by row it took three mostly empty stores per row, and under load a store costs ~100
      // cycles whatever its width.  The rows of a wave belong to one window: same K6.
      {
        const int npair = K6 >> 1;                             // column pairs per row
        const int total = nr * npair;
        int q = lane / npair, cp = lane - q * npair;           // (one division per lane; afterwards incremental)
        for (int it = lane; it < total; it += 64) {
          double *Wr = Wset + rowoff[c0 + q];
          const double *row = rows + (size_t)q * RS;
          VecN<double, 2> rv;
          rv.v[0] = row[2 * cp]; rv.v[1] = row[2 * cp + 1];
          *reinterpret_cast<VecN<double, 2> *>(Wr + 2 * cp) = rv;
          cp += 64;
          while (cp >= npair) { cp -= npair; ++q; }
        }
        if (lane < nr) {                                       // the line-delay column of row `lane`, its Hll and g_rho
          const double *row = rows + (size_t)lane * RS;
          const int l = rlm[c0 + lane];
          Wset[rowoff[c0 + lane] + P - 1] = row[K6];
          Hllset[lm0 + l] = row[K6 + 1];
          gset[u0 + P + l] = row[K6 + 2];
        }
      }
      LDS_SYNC();
      if (dbg && lane == 0) { dbg[4 + 2 * (c0 / NR)] = clock64(); dbg[10] = nlm * 1000000ll; dbg[11] = NR * 1000; }
    }
#undef LDS_SYNC
  }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) void k_vis_eval(Dev<double> d, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned char smt[VIS_LDS_BYTES];
  __shared__ long long rowoff[64];
  __shared__ int rlm[64];
  vis_eval_body(d, mode, smt, rowoff, rlm, blockIdx.x);
}

// Both evaluations in ONE launch: workgroups [0, Gtot) take an IMU group each, the others a wave of 64 visual block slots.  The two are
// independent; for a batch smaller than the chip their single-wave latencies (23 us each for one window) overlap instead of adding
// up, and large batches lose nothing.  The IMU rows use the head of the visual kernel's LDS buffer.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_linearize_f64(Dev<double> d, int mode, int general_only, int zero_mode) {
  static_assert(VIS_LDS_BYTES >= (72 * 33 + 64) * 8, "the IMU rows use the head of the visual body's LDS buffer");
  __shared__ __attribute__((aligned(32))) unsigned char smt[VIS_LDS_BYTES];
  __shared__ long long rowoff[64];
  __shared__ int rlm[64];
  if ((int)blockIdx.x < d.Gtot) {
    if (!general_only) imu_linearize_f64_fast(d, mode, reinterpret_cast<double *>(smt), blockIdx.x, d.Gtot, zero_mode);   // (one group per wave here)
  } else vis_eval_body(d, mode, smt, rowoff, rlm, blockIdx.x - d.Gtot);
}

// Visual assembly: gridDim.y workgroups (8 waves each) per window.  The host sorted the visual blocks by
// frame pair and cut them into items of <= CH blocks; blocks of an item that evaluate on the same knot
// quadruples (si, sj) form a run.  A wave stages its item's J~ (100 x n) and r~ in LDS (all loads of a pass
// in flight together), then forms the run's 50 x 50 product [J~_pose | r~]^T [J~_pose | r~] with a 7 x 7
// register tile per lane (rows {ti+8a}, cols {tj+8b}; K = 2 * run length) and adds it into an LDS-resident
// copy of the window's visual Hessian (packed lower triangle over the 6K knot unknowns + the line-delay
// row): ~1.2k ds_add per RUN instead of ~1.3k global atomics per BLOCK.  The two ends of a block may share
// knots (reference image_feature_factor.h:165-180,215,233): every ordered column pair whose unknowns satisfy
// g(a) >= g(b) is added, so shared knots sum correctly.  Landmark terms (W row, Hll, g_rho) stay per block.
// Windows whose packed Hessian does not fit in LDS (vis_lds = 0) add straight into Hpp.
template <class T, int CH, bool LDSH> __global__ __launch_bounds__(512) void k_assemble_vis(Dev<T> d, int mode) {
  constexpr int CHP = CH + 2, NW = 8, RPP = 64 / CH, NPASS = (102 + RPP - 1) / RPP;   // even row stride: 8-byte aligned pairs
  const long long t_begin = d.dbg ? clock64() : 0;
  const int w = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
  if (!lin_run(d.lm[w], mode) || lin_cost_only(d.lm[w], mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  const int tgset = lin_target(d.lm[w], mode);
  // fields used after LDS/global atomics are copied to registers: the compiler must otherwise re-read them from
  // memory every time (a store could alias), one L2 round trip each
  const int P = m.P, K = m.K, nvitem = m.nvitem, vitem0 = m.vitem0, ngrp = m.ngrp, grp0 = m.grp0, u0 = m.u0, ldh = m.ldh;
  if ((m.vis_lds != 0) != LDSH) return;   // the host launches both variants; each window is handled by one of them
  if (m.V == 0 && !LDSH) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smv[];
  const int K6 = 6 * K, tri = K6 * (K6 + 1) / 2;
  const int nHh = LDSH ? tri + K6 + 1 : 0;     // packed Hessian entries
  double *gs = reinterpret_cast<double *>(smv);                       // [K6 + 1] pose gradient, fp64 (ds_add_f32 is ~20x slower)
  T *Hs = reinterpret_cast<T *>(gs + ((K6 + 2) & ~1));                // [nHh]
  T *stage = Hs + ((nHh + 3) & ~3);                                   // [NW][102][CHP]
  int *keys = reinterpret_cast<int *>(stage + NW * 102 * CHP);        // [NW][2][CH]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < nHh; i += 512) Hs[i] = T(0);
  for (int i = tid; i < K6 + 1; i += 512) gs[i] = 0.0;
  __syncthreads();
  T *Js = stage + wave * 102 * CHP;
  int *ks = keys + wave * 2 * CH;
  const size_t V = (size_t)d.Vtot;
  const int per_round = NW * nparts;
  const int rounds = (nvitem + per_round - 1) / per_round;
  double *Hg = d.HppS[tgset] + m.H0;
  const int ti = lane >> 3, tj = lane & 7;
  // local column c (0..47 knot columns, 48 line delay, 49 residual) -> first of its two staging rows
  int rowa[7], rowb[7];
#pragma unroll
  for (int a = 0; a < 7; ++a) {
    const int ca = ti + 8 * a, cb = tj + 8 * a;
    rowa[a] = ca < 48 ? 2 * ca : (ca == 48 ? 98 : (ca == 49 ? 100 : -1));
    rowb[a] = cb < 48 ? 2 * cb : (cb == 48 ? 98 : (cb == 49 ? 100 : -1));
  }
  long long *dbg = (d.dbg && w == 0 && part == 0) ? d.dbg + 48 : nullptr;
  int dbi = 0;
#define CTV_STAMP() do { if (dbg && tid == 0 && dbi < 15) dbg[dbi++] = clock64(); } while (0)
  if (dbg && tid == 0) dbg[dbi++] = t_begin;
  CTV_STAMP();
  for (int r = 0; r < rounds; ++r) {
    const int it = (r * nparts + part) * NW + wave;
    int n = 0, v0 = 0;
    if (it < nvitem) { const VisItem I = d.vitems[vitem0 + it]; n = I.count; v0 = I.start; }
    {
      const int c = lane % CH, rr = lane / CH;
      const unsigned blk = (unsigned)d.vblk[v0 + (c < n ? c : 0)];   // slot of the item's block c (landmark-major evaluation order)
      const unsigned anc = (unsigned)d.vblk_anc[v0 + (c < n ? c : 0)];
      T tmp[NPASS];
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int row = i * RPP + rr;
        tmp[i] = T(0);
        if (row < 102 && c < n) tmp[i] = vis_J_entry(d, row, blk, anc);
      }
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int row = i * RPP + rr;
        if (row < 102) Js[row * CHP + c] = tmp[i];
      }
    }
    if (lane < n) { ks[lane] = d.a_s[d.vblk_anc[v0 + lane]]; ks[CH + lane] = d.vsj[d.vblk[v0 + lane]]; }
    __syncthreads();
    
    int start = 0;
    while (start < n) {
      const int si = ks[start], sj = ks[CH + start];
      const bool diff = (lane > start && lane < n) && (ks[lane] != si || ks[CH + lane] != sj);
      const unsigned long long mask = __ballot(diff);
      const int end = mask ? (__ffsll((long long)mask) - 1) : n;
      // two blocks per step (one 8-byte LDS read per operand); blocks outside [start, end) are masked to zero.
      // (v_pk_fma_f32 on the block pair was measured slower than scalar FMAs here: 354k vs 308k cycles per window.)
      T acc[7][7];
#pragma unroll
      for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = 0; b < 7; ++b) acc[a][b] = T(0);
      for (int v2 = start & ~1; v2 < end; v2 += 2) {
        const T m0 = (v2 >= start) ? T(1) : T(0), m1 = (v2 + 1 < end) ? T(1) : T(0);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          VecN<T, 2> av[7], bv[7];
#pragma unroll
          for (int a = 0; a < 7; ++a) {
            av[a].v[0] = av[a].v[1] = bv[a].v[0] = bv[a].v[1] = T(0);
            if (rowa[a] >= 0) av[a] = *reinterpret_cast<const VecN<T, 2> *>(Js + (rowa[a] + rr) * CHP + v2);
            if (rowb[a] >= 0) bv[a] = *reinterpret_cast<const VecN<T, 2> *>(Js + (rowb[a] + rr) * CHP + v2);
            av[a].v[0] *= m0; av[a].v[1] *= m1;
          }
#pragma unroll
          for (int a = 0; a < 7; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) acc[a][b] += av[a].v[0] * bv[b].v[0] + av[a].v[1] * bv[b].v[1];   // symmetric: blocks a >= b
        }
      }
      int ga[7], gb[7];
#pragma unroll
      for (int a = 0; a < 7; ++a) {
        const int ca = ti + 8 * a, cb = tj + 8 * a;
        ga[a] = ca < 48 ? vis_col(ca, si, sj, P) : (ca == 48 ? P - 1 : (ca == 49 ? -2 : -1));
        gb[a] = cb < 48 ? vis_col(cb, si, sj, P) : (cb == 48 ? P - 1 : (cb == 49 ? -2 : -1));
      }
      // Each unordered column pair {ca, cb} is held exactly once: blocks a > b by this lane, and for a == b by the lane
      // with ti >= tj.  It goes to H[max(g)][min(g)]; two different local columns that map to the same unknown (ends
      // sharing a knot) contribute twice to the diagonal entry.
#pragma unroll
      for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          if (a == b && ti < tj) continue;
          int gA = ga[a], gB = gb[b];
          T hv = acc[a][b];
          if (gA == -2 || gB == -2) {         // column 49 = residual: J~^T r~ (r~^T r~ itself is not needed)
            const int gX = gA == -2 ? gB : gA;
            if (gX >= 0) atomicAdd(&gs[gX == P - 1 ? K6 : gX], (double)hv);
          } else if (gA >= 0 && gB >= 0) {
            if (gA == gB && !(a == b && ti == tj)) hv *= T(2);
            if (gA < gB) { const int t = gA; gA = gB; gB = t; }
            if (LDSH) atomicAdd(&Hs[(gA == P - 1) ? tri + (gB == P - 1 ? K6 : gB) : gA * (gA + 1) / 2 + gB], hv);
            else atomicAdd(&Hg[(long long)gA * ldh + gB], (double)hv);
          }
        }
      start = end;
      
    }
    __syncthreads();
    
  }
  CTV_STAMP();
  if (LDSH) {
    // IMU group tiles: the knot x knot part (24 x 24 per group, overlapping between consecutive segments)
    for (int gi = part * NW + wave; gi < ngrp; gi += per_round) {
      const ImuGroup grp = d.groups[grp0 + gi];
      const T *tile = d.imu_tiles + (size_t)(grp0 + gi) * 1024;
      T tv[9];   // 24 x 24 = 9 x 64 entries: all loads in flight together
#pragma unroll
      for (int u = 0; u < 9; ++u) { const int e = lane + 64 * u; tv[u] = tile[(e / 24) * 32 + e % 24]; }
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        const int e = lane + 64 * u, a = e / 24, b = e % 24;
        const int ga = imu_col(a, grp.s, K, grp.bias), gb = imu_col(b, grp.s, K, grp.bias);
        if (ga >= gb) atomicAdd(&Hs[ga * (ga + 1) / 2 + gb], tv[u]);
      }
    }
    __syncthreads();
    CTV_STAMP();
    for (int i = tid; i < nHh; i += 512) {
      const T hv = Hs[i];
      if (nparts > 1 && hv == T(0)) continue;
      int ga, gb;
      if (i < tri) {
        ga = (int)((sqrtf(8.0f * (float)i + 1.0f) - 1.0f) * 0.5f);
        while ((ga + 1) * (ga + 2) / 2 <= i) ++ga;
        while (ga * (ga + 1) / 2 > i) --ga;
        gb = i - ga * (ga + 1) / 2;
      } else {
        ga = P - 1;
        gb = (i - tri) < K6 ? (i - tri) : P - 1;
      }
      if (nparts > 1) atomicAdd(&Hg[(long long)ga * ldh + gb], (double)hv);
      else Hg[(long long)ga * ldh + gb] = (double)hv;  // first writer after k_zero_normal; later kernels add atomically
    }
  }
  CTV_STAMP();
  if (!LDSH) __syncthreads();
  for (int i = tid; i < K6 + 1; i += 512) {
    const double gv = gs[i];
    if (gv != 0.0) atomicAdd(&d.gS[tgset][u0 + (i < K6 ? i : P - 1)], gv);
  }
  CTV_STAMP();
#undef CTV_STAMP
}

// ---- store-semantics tail of the assembly (product path, windows whose packed Hessian is LDS resident).  Every entry of Hpp / g is
// formed completely by ONE thread and written with a plain store -- the knot x knot block and the line-delay row from the packed
// (LDS or summed-partials) Hessian, the bias rows by a gather over the IMU group tiles of that bias state (fixed order), the bias
// chain and the prior (J0^T J0 looked up through the inverse column map): no pre-zeroing pass, no floating-point atomics, and the
// value does not depend on any execution order.  Entries no factor reaches are zeroed once at upload and never written.
template <class T> __device__ __forceinline__ double prior_H(const Dev<T> &d, const WinMeta &m, int ga, int gb) {
  if (m.pn <= 0) return 0.0;
  const int pi = d.pinv[m.p0 + ga], pj = d.pinv[m.p0 + gb];
  return (pi >= 0 && pj >= 0) ? d.pH[m.pH0 + (size_t)pi * m.pn + pj] : 0.0;
}
template <class T> __device__ __forceinline__ double prior_g(const Dev<T> &d, const WinMeta &m, int u) {
  if (m.pn <= 0) return 0.0;
  const int pi = d.pinv[m.p0 + u];
  return pi >= 0 ? d.pgrad[m.pv0 + pi] : 0.0;
}
// packed index i of the knot block / line-delay row -> (row, column) unknowns
__device__ __forceinline__ void packed_decode(int i, int tri, int K6, int P, int &ga, int &gb) {
  if (i < tri) {
    ga = (int)((sqrtf(8.0f * (float)i + 1.0f) - 1.0f) * 0.5f);
    ga += ((ga + 1) * (ga + 2) / 2 <= i) ? 1 : 0;      // the float estimate is off by at most one either way
    ga -= (ga * (ga + 1) / 2 > i) ? 1 : 0;
    gb = i - ga * (ga + 1) / 2;
  } else {
    ga = P - 1;
    gb = (i - tri) < K6 ? (i - tri) : P - 1;
  }
}
// Bias rows (and the bias columns of the line-delay row) of Hpp and the bias entries of g of window w: item e of [0, nitems)
// handled by thread e of a grid-stride loop.  `bias` = the linearisation state (candidate or current).
template <class T>
__device__ __forceinline__ void bias_rows_store(const Dev<T> &d, const WinMeta &m, int tg, const double *bias, int first, int stride) {
  const int K = m.K, F = m.F, P = m.P, K6 = 6 * K, ldh = m.ldh, nbr = 6 * F;
  double *Hg = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  const int32_t *boff = d.bgl_off + m.bias0 + (int)(&m - d.wins);   // F + 1 offsets of this window's per-bias group lists
  // items: rows r = K6 .. P - 2 with all columns c <= r (triangle over the bias rows, rectangle over the knot columns), then the
  // line-delay row's bias columns, then the bias entries of g
  const int n_rect = nbr * K6, n_tri = nbr * (nbr + 1) / 2, n_ld = nbr, n_g = nbr;
  for (int e = first; e < n_rect + n_tri + n_ld + n_g; e += stride) {
    if (e < n_rect + n_tri) {
      int rb, c;   // rb: bias row index (0 .. 6F), c: column unknown
      if (e < n_rect) { rb = e / K6; c = e - rb * K6; }
      else {
        const int t = e - n_rect;
        int i2 = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        i2 += ((i2 + 1) * (i2 + 2) / 2 <= t) ? 1 : 0;
        i2 -= (i2 * (i2 + 1) / 2 > t) ? 1 : 0;
        rb = i2; c = K6 + t - i2 * (i2 + 1) / 2;
      }
      const int f = rb / 6, a = rb - 6 * f, r = K6 + rb;
      double v = 0.0;
      const int g0 = boff[f], g1 = boff[f + 1];
      if (c < K6) {
        const int k = c / 6, cc = c - 6 * k;
        for (int q = g0; q < g1; ++q) {
          const int gi = d.bgl[q];
          const int sg = d.groups[gi].s;
          if (k >= sg && k <= sg + 3) v += d.imu_tiles[(size_t)gi * 1024 + (24 + a) * 32 + (cc < 3 ? 3 * (k - sg) + cc : 12 + 3 * (k - sg) + cc - 3)];
        }
      } else {
        const int cb = c - K6, f2 = cb / 6, a2 = cb - 6 * f2;
        if (f2 == f)
          for (int q = g0; q < g1; ++q) v += d.imu_tiles[(size_t)d.bgl[q] * 1024 + (24 + a) * 32 + 24 + a2];
        if (a2 == a)
          for (int b = 0; b < m.NB; ++b) {
            const int bi = d.bc_i[m.bc0 + b], bj = d.bc_j[m.bc0 + b];
            const double wv = d.bc_w[(size_t)(m.bc0 + b) * 6 + a];
            if (f2 == f) { if (bi == f) v += wv * wv; if (bj == f) v += wv * wv; }
            else if ((bi == f2 && bj == f) || (bi == f && bj == f2)) v -= wv * wv;
          }
      }
      Hg[(long long)r * ldh + c] = v + prior_H(d, m, r, c);
    } else if (e < n_rect + n_tri + n_ld) {
      const int c = K6 + e - n_rect - n_tri;
      Hg[(long long)(P - 1) * ldh + c] = prior_H(d, m, P - 1, c);
    } else {
      const int rb = e - n_rect - n_tri - n_ld, f = rb / 6, a = rb - 6 * f, r = K6 + rb;
      double v = 0.0;
      for (int q = boff[f]; q < boff[f + 1]; ++q) v += d.imu_tiles[(size_t)d.bgl[q] * 1024 + (24 + a) * 32 + 30];
      for (int b = 0; b < m.NB; ++b) {
        const int bi = d.bc_i[m.bc0 + b], bj = d.bc_j[m.bc0 + b];
        if (bi != f && bj != f) continue;
        const double wv = d.bc_w[(size_t)(m.bc0 + b) * 6 + a];
        const double rr = wv * (bias[6 * (m.bias0 + bj) + a] - bias[6 * (m.bias0 + bi) + a]);
        if (bi == f) v -= wv * rr;
        if (bj == f) v += wv * rr;
      }
      g[r] = v + prior_g(d, m, r);
    }
  }
}

// fp32 visual assembly on the matrix cores (windows whose packed Hessian is LDS resident).  Same decomposition as
// k_assemble_vis (items of <= CH blocks per frame pair, runs of equal knot quadruples inside an item), but
//   * the run's [J~_pose | r~]^T [J~_pose | r~] (50 x 50, padded to 64 = 4 x 4 tiles of 16; the lower 10 tiles) is formed
//     with v_mfma_f32_16x16x4_f32: K = 4 is two blocks x two residual rows, the operands are plain LDS reads of the staged
//     item ([102][CH + 2], row = 2 * column + residual row), the A and B operand of a tile pair are the same registers;
//   * the staging area of a wave is private, so there is no workgroup barrier inside the item loop, and the next item's
//     J~ (51 values per lane) is requested before the current item is processed: its latency hides under the products.
// C/D layout of the 16x16x4 fp32 MFMA: register r of lane l = D[4 (l / 16) + r][l % 16].
// T = double (product path): the same kernel on v_mfma_f64_16x16x4_f64 (D register r of lane l = D[(l / 16) + 4 r][l % 16]), items of
// <= 8 blocks so that eight fp64 staging areas fit beside the packed Hessian.
template <class T> struct MfmaAcc;
template <> struct MfmaAcc<double> { typedef f64x4 type; };
__device__ __forceinline__ f64x4 mfma16(double a, double b, f64x4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
// NW = waves per workgroup: 8, or 1 in the deterministic mode (all LDS additions of a partial Hessian then come from one wave, in program
// order).  STORE (LDS-resident windows of the product path): store-semantics tail -- with one part the workgroup finishes the window
// itself (prior added on the way out, bias rows gathered); with several parts every part writes its packed partial Hessian to
// Dev::Hpart and k_reduce_finalize sums them in part order.
template <class T, int CH, bool LDSH, int NW = 8, bool STORE = false> __global__ __launch_bounds__(64 * NW) void k_assemble_vis_mfma(Dev<T> d, int mode) {
  constexpr bool F64 = sizeof(T) == 8;
  typedef typename MfmaAcc<T>::type acc_t;
  constexpr int CHP = CH + 2, RPP = 64 / CH, NT = 64 * NW;
  static_assert(!STORE || LDSH, "the store-semantics tail needs the LDS-resident Hessian");
  // staged rows per item: 0..95 pose columns (row = 2 * column + residual row), 98/99 line delay, 100/101 residual, 102..107 A~,
  // 108..111 cp0, 112..115 cp1.  38 of them come from the block records in HBM (rows 48..71 = the j end's rotation columns, 98..107,
  // 112..115) and 4 from the anchor records (cp0); the anchor end's rotation rows 0..23 are A~ times the anchor's GR (9 record entries
  // per lane, held in registers), the 48 position rows (24..47, 72..95) are rebuilt in LDS from rows 102..115; the inverse-depth
  // column is not needed here.
  static_assert(CH == 8 && F64, "the staging pattern is written for items of 8 blocks");
  constexpr int SROWS = 116, NEXP = 48 / RPP;
  const long long t_begin = d.dbg ? clock64() : 0;
  const int w = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
  if (!lin_run(d.lm[w], mode) || lin_cost_only(d.lm[w], mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  const int tgset = lin_target(d.lm[w], mode);
  const int P = m.P, K = m.K, nvitem = m.nvitem, vitem0 = m.vitem0, ngrp = m.ngrp, grp0 = m.grp0, u0 = m.u0, ldh = m.ldh;
  if ((m.vis_lds != 0) != LDSH) return;   // the host launches both variants; each window is handled by one of them
  if (!LDSH && m.V == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smv[];
  // fp64 accumulators: ds_add_f64 sustains ~8 cycles per wave instruction on gfx950, ds_add_f32 ~190 (measured,
  // tools/lds_atomic_bench.hip) -- and the fp64 sums do not depend on the order of the additions to ~1e-16
  double *Hs = reinterpret_cast<double *>(smv);
  const int K6 = 6 * K, tri = K6 * (K6 + 1) / 2;
  const int nHh = LDSH ? tri + K6 + 1 : 0;     // packed Hessian entries: knot x knot lower triangle, line-delay row
  const int nH = nHh + K6 + 1;                 // + gradient of the pose columns (knots, line delay)
  double *gs = Hs + nHh;
  T *stage = reinterpret_cast<T *>(Hs + ((nH + 3) & ~3));     // [NW][SROWS][CHP]
  int *keys = reinterpret_cast<int *>(stage + NW * SROWS * CHP);        // [NW][2][CH]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int per_round = NW * nparts;
  // IMU group tiles (knot x knot part, 24 x 24 per group, overlapping between consecutive segments): the loads of this wave's
  // first NGI groups are issued before the LDS Hessian is zeroed and added right after -- at the end of the kernel they
  // were three exposed memory round trips (24 k of 243 k cycles, measured)
  constexpr int NGI = 3;
  T tv[NGI][9], tgv[NGI];   // tgv: the group's gradient entries of the knot rows (tile column 30), store-semantics tail only
  int gs_[NGI], gb_[NGI];
  if (LDSH) {
#pragma unroll
    for (int u = 0; u < NGI; ++u) {
      const int gi = min(part * NW + wave + u * per_round, max(ngrp - 1, 0));
      const ImuGroup grp = d.groups[grp0 + gi];
      gs_[u] = grp.s; gb_[u] = grp.bias;
      const T *tile = d.imu_tiles + (size_t)(grp0 + gi) * 1024;
#pragma unroll
      for (int q = 0; q < 9; ++q) { const int e = lane + 64 * q; tv[u][q] = ngrp > 0 ? tile[(e / 24) * 32 + e % 24] : T(0); }
      tgv[u] = (STORE && ngrp > 0) ? tile[min(lane, 23) * 32 + 30] : T(0);
    }
  }
  for (int i = 2 * tid; i < ((nH + 3) & ~3); i += 2 * NT) *reinterpret_cast<double2 *>(Hs + i) = double2{0.0, 0.0};
  __syncthreads();
  if (LDSH) {
#pragma unroll
    for (int u = 0; u < NGI; ++u) {
      if (part * NW + wave + u * per_round >= ngrp) continue;
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const int e = lane + 64 * q, a = e / 24, b = e % 24;
        const int ga = imu_col(a, gs_[u], K, gb_[u]), gb = imu_col(b, gs_[u], K, gb_[u]);
        if (ga >= gb) atomicAdd(&Hs[ga * (ga + 1) / 2 + gb], (double)tv[u][q]);
      }
      if (STORE && lane < 24) atomicAdd(&Hs[nHh + imu_col(lane, gs_[u], K, gb_[u])], (double)tgv[u]);   // (gs = Hs + nHh)
    }
  }
  T *Js = stage + wave * SROWS * CHP;
  int *ks = keys + wave * 2 * CH;
  const size_t V = (size_t)d.Vtot;
  // every wave owns a contiguous range of items: a run that continues into the wave's next item keeps its accumulators
  // and is scattered once (the scatter costs as much as the products of an item: ~5 k cycles, measured)
  const int it0 = (int)((long long)nvitem * (part * NW + wave) / per_round);
  const int rounds = (int)((long long)nvitem * (part * NW + wave + 1) / per_round) - it0;
  double *Hg = d.HppS[tgset] + m.H0;
  const int q4 = lane >> 4, l15 = lane & 15, bsel = q4 >> 1, rr = q4 & 1;   // MFMA k index = 2 * (block of the pair) + residual row
  const int sc = lane % CH, srr = lane / CH;                               // staging: column (block) and row parity of this lane
  long long *dbg = (d.dbg && w == 0 && part == 0) ? d.dbg + 48 : nullptr;
  int dbi = 0;
#define CTV_STAMP() do { if (dbg && tid == 0 && dbi < 15) dbg[dbi++] = clock64(); } while (0)
  if (dbg && tid == 0) dbg[dbi++] = t_begin;
  CTV_STAMP();
  // MFMA operand row offsets of this lane: tile row I -> knot column 16 I + l15, at k = q4 (block bsel of the pair, residual
  // row rr).  The line-delay column (staged rows 98, 99) and the residual (rows 100, 101) ride on the same operand
  // values with plain FMAs: every lane multiplies its three J entries by J_ld[k] and r[k] of its own k; the four k
  // groups (lanes l15 + 16 q4) are summed with two shuffles per value at the end of the run.
  int orow[3];
#pragma unroll
  for (int I = 0; I < 3; ++I) orow[I] = (2 * (16 * I + l15) + rr) * CHP;
  const int ldrow = (98 + rr) * CHP, rrow = (100 + rr) * CHP;
  T tj[5], tg9[9], tc0 = T(0);
  int n = 0, v0 = 0, key_i = 0, key_j = 0;
  // the (start, count) of this wave's items: lane r holds item r, read once -- a per-item load of the descriptor would put a
  // full memory round trip in front of every item's J~ request
  int my_start, my_count;
  {
    const int it = it0 + lane;
    const VisItem I = d.vitems[vitem0 + min(it, max(nvitem - 1, 0))];   // clamped: always a valid descriptor
    my_start = I.start;
    my_count = (lane < rounds && it < nvitem) ? I.count : 0;
  }
  auto item_desc = [&](int r, int &istart, int &icount) {
    if (r < 64) { istart = __shfl(my_start, r); icount = __shfl(my_count, r); }
    else {
      const int it = it0 + r;
      const VisItem I = d.vitems[vitem0 + min(it, nvitem - 1)];
      istart = I.start; icount = (r < rounds && it < nvitem) ? I.count : 0;
    }
    if (r >= rounds) icount = 0;
  };
  // The blocks of an item are slots of the landmark-major evaluation order, listed in Dev::vblk: the slot of this lane's block
  // (c = sc) and of the block whose keys it reads (lane) are requested one item ahead, so that the J~ loads of an item do not
  // wait for its slot list.
  int idn = 0, idkn = 0, ian = 0, iakn = 0;
  auto load_ids = [&](int r) {
    int istart, icount;
    item_desc(r, istart, icount);
    idn = d.vblk[istart + (sc < icount ? sc : 0)];
    ian = d.vblk_anc[istart + (sc < icount ? sc : 0)];
    idkn = d.vblk[istart + (lane < icount ? lane : 0)];
    iakn = d.vblk_anc[istart + (lane < icount ? lane : 0)];
  };
  auto fetch = [&](int r) {   // request item r of this wave: unconditional loads on clamped addresses, masked when staged
    int istart, icount;
    item_desc(r, istart, icount);
    n = icount;
    v0 = istart;
    // The block records are block-major ([slot][VT_ROWS]); lane (sc, srr) takes entries srr + 8 i of its block: 0..23 the j end's
    // rotation columns, 26..33 line delay / residual / A~[0..3], 34..39 A~[4, 5] and cp1 (the inverse-depth entries 24, 25 are
    // skipped) -- the RPP lanes of a block read RPP consecutive entries.  From the anchor record: the three factors GR[c][0..2] of the
    // lane's three anchor-end rotation entries e = srr + 8 i (column c = e / 2) and one of the four cp0.
    const unsigned jb = (unsigned)idn * (unsigned)VT_ROWS + (unsigned)srr;
#pragma unroll
    for (int i = 0; i < 3; ++i) tj[i] = d.Jt[jb + (unsigned)(8 * i)];
    tj[3] = d.Jt[jb + 26u];
    tj[4] = d.Jt[(unsigned)idn * (unsigned)VT_ROWS + (unsigned)min(34 + srr, VT_ROWS - 1)];
    const double *rec = d.arec + (size_t)ian * AREC;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int mm = 0; mm < 3; ++mm) tg9[3 * i + mm] = rec[AR_GR + 3 * ((srr + 8 * i) >> 1) + mm];
    tc0 = rec[AR_CP0 + (srr & 3)];
    key_i = d.a_s[iakn];
    key_j = d.vsj[idkn];
    load_ids(r + 1);
  };
  if (nvitem > 0) load_ids(0);
  if (nvitem > 0) fetch(0);
  // accumulators of the open run (asi, asj): 3 x 3 lower tiles of the 48 x 48 pose block, line-delay column, residual
  acc_t acc[6];
  T pl[3], pr[3], pll, prl;
  int asi = -1, asj = -1;
  auto reset_acc = [&]() {
#pragma unroll
    for (int q = 0; q < 6; ++q) acc[q] = acc_t{T(0), T(0), T(0), T(0)};
#pragma unroll
    for (int I = 0; I < 3; ++I) { pl[I] = T(0); pr[I] = T(0); }
    pll = T(0); prl = T(0);
  };
  // ---- scatter: local column -> unknown, each unordered local pair once; pairs of different local columns that map
  //      to the same unknown (ends sharing a knot) count twice on the diagonal
  auto scatter = [&](int si, int sj) {
#pragma unroll
    for (int I = 0; I < 3; ++I) {
      pl[I] += __shfl_xor(pl[I], 16); pl[I] += __shfl_xor(pl[I], 32);
      pr[I] += __shfl_xor(pr[I], 16); pr[I] += __shfl_xor(pr[I], 32);
    }
    pll += __shfl_xor(pll, 16); pll += __shfl_xor(pll, 32);
    prl += __shfl_xor(prl, 16); prl += __shfl_xor(prl, 32);
    // unknown index and triangular row offset g (g + 1) / 2 of this lane's 3 tile columns and 12 tile rows, once per run
    int gcol[3], tcol[3], grow[3][4], trow[3][4];
#pragma unroll
    for (int J = 0; J < 3; ++J) {
      gcol[J] = vis_col(16 * J + l15, si, sj, P);
      tcol[J] = gcol[J] * (gcol[J] + 1) / 2;
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        grow[J][rg] = vis_col(16 * J + (F64 ? q4 + 4 * rg : 4 * q4 + rg), si, sj, P);
        trow[J][rg] = grow[J][rg] * (grow[J][rg] + 1) / 2;
      }
    }
    int q = 0;
#pragma unroll
    for (int I = 0; I < 3; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J, ++q) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int ca = 16 * I + (F64 ? q4 + 4 * rg : 4 * q4 + rg), cb = 16 * J + l15;
          if (I == J && ca < cb) continue;
          const int gA = grow[I][rg], gB = gcol[J];
          T hv = acc[q][rg];
          if (gA == gB && ca != cb) hv *= T(2);
          const bool ge = gA >= gB;
          if (LDSH) atomicAdd(&Hs[(ge ? trow[I][rg] : tcol[J]) + (ge ? gB : gA)], (double)hv);
          else atomicAdd(&Hg[(long long)(ge ? gA : gB) * ldh + (ge ? gB : gA)], (double)hv);
        }
      }
    // line-delay row of the Hessian and the pose gradient (every k group holds the totals; group q4 = 0 adds them)
    if (q4 == 0) {
#pragma unroll
      for (int J = 0; J < 3; ++J) {
        if (LDSH) atomicAdd(&Hs[tri + gcol[J]], (double)pl[J]);
        else atomicAdd(&Hg[(long long)(P - 1) * ldh + gcol[J]], (double)pl[J]);
        atomicAdd(&gs[gcol[J]], (double)pr[J]);
      }
      if (l15 == 0) {   // (ld, ld) and r . J_ld
        if (LDSH) atomicAdd(&Hs[tri + K6], (double)pll);
        else atomicAdd(&Hg[(long long)(P - 1) * ldh + (P - 1)], (double)pll);
        atomicAdd(&gs[K6], (double)prl);
      }
    }
  };
  for (int r = 0; r < rounds && nvitem > 0; ++r) {
    // ---- stage the fetched item (LDS operations of one wave are ordered: no barrier), then request the next one
    const int ncur = n;
    {
      const bool in = sc < ncur;
#pragma unroll
      for (int i = 0; i < 3; ++i) Js[(48 + srr + 8 * i) * CHP + sc] = in ? tj[i] : T(0);
      Js[(98 + srr) * CHP + sc] = in ? tj[3] : T(0);                     // rows 98..105: line delay, residual, A~[0..3]
      if (srr < 2) Js[(106 + srr) * CHP + sc] = in ? tj[4] : T(0);       // A~[4, 5]
      else if (srr < 6) Js[(110 + srr) * CHP + sc] = in ? tj[4] : T(0);  // cp1 (entries 36..39 -> rows 112..115)
      if (srr < 4) Js[(108 + srr) * CHP + sc] = in ? tc0 : T(0);         // cp0 (anchor record)
    }
    if (lane < CH) { ks[lane] = key_i; ks[CH + lane] = key_j; }
    __builtin_amdgcn_wave_barrier();
    // anchor end's rotation rows: entry e = srr + 8 i (= 2 * column + residual row) = A~[rr][0..2] . GR[column][0..2]
    {
      const int rr2 = srr & 1;
      const T a0 = Js[(102 + rr2) * CHP + sc], a1 = Js[(104 + rr2) * CHP + sc], a2 = Js[(106 + rr2) * CHP + sc];   // (zero for sc >= ncur)
#pragma unroll
      for (int i = 0; i < 3; ++i) Js[(srr + 8 * i) * CHP + sc] = a0 * tg9[3 * i] + a1 * tg9[3 * i + 1] + a2 * tg9[3 * i + 2];
    }
    // position rows: column 12 + 3 k + b (i end) = cp0[k] P~[b], column 36 + 3 k + b (j end) = -cp1[k] P~[b]
#pragma unroll
    for (int e = 0; e < NEXP; ++e) {
      constexpr int HALF = 24 / RPP;
      const int side = e / HALF, rem = (e % HALF) * RPP + srr;    // rem = 2 * (3 k + b) + residual row
      const int pc = rem >> 1, kk = pc / 3, b = pc - 3 * kk;
      const T pv = Js[(102 + 2 * b + (rem & 1)) * CHP + sc], cv = Js[(108 + 4 * side + kk) * CHP + sc];
      Js[(24 + 48 * side + rem) * CHP + sc] = side ? -(cv * pv) : cv * pv;
    }
    __builtin_amdgcn_wave_barrier();
    if (r < 2) CTV_STAMP();
    fetch(r + 1);
    int start = 0;
    while (start < ncur) {
      const int si = ks[start], sj = ks[CH + start];
      const bool diff = (lane > start && lane < ncur) && (ks[lane] != si || ks[CH + lane] != sj);
      const unsigned long long mask = __ballot(diff);
      const int end = mask ? (__ffsll((long long)mask) - 1) : ncur;
      if (asi != si || asj != sj) {     // a run that continues from the previous item keeps accumulating
        if (asi >= 0) scatter(asi, asj);
        reset_acc();
        asi = si; asj = sj;
      }
      // 4 K-steps (8 blocks) per trip: the operand reads first, then the products -- one LDS latency per trip
      for (int v8 = start; v8 < end; v8 += 8) {
        T a[4][3], ldv[4], rv[4];
        // (the usual item is one run that ends with the item: the columns past it were staged as zeros, nothing to mask)
        const bool nomask = end == ncur && v8 + 8 <= CH;     // uniform
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int blk = v8 + 2 * s + bsel;
          const int bc = min(blk, CH + 1);   // a valid LDS address even when past the run (value discarded)
#pragma unroll
          for (int I = 0; I < 3; ++I) a[s][I] = Js[orow[I] + bc];
          ldv[s] = Js[ldrow + bc];
          rv[s] = Js[rrow + bc];
        }
        if (!nomask) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const bool in = v8 + 2 * s + bsel < end;
#pragma unroll
            for (int I = 0; I < 3; ++I) a[s][I] = in ? a[s][I] : T(0);
            ldv[s] = in ? ldv[s] : T(0);
            rv[s] = in ? rv[s] : T(0);
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (v8 + 2 * s >= end) break;     // uniform
          int q = 0;
#pragma unroll
          for (int I = 0; I < 3; ++I)
#pragma unroll
            for (int J = 0; J <= I; ++J, ++q) acc[q] = mfma16(a[s][I], a[s][J], acc[q]);
#pragma unroll
          for (int I = 0; I < 3; ++I) { pl[I] += a[s][I] * ldv[s]; pr[I] += a[s][I] * rv[s]; }
          pll += ldv[s] * ldv[s];
          prl += rv[s] * ldv[s];
        }
      }
      if (r < 2) CTV_STAMP();
      start = end;
    }
    if (r < 2) CTV_STAMP();
  }
  if (asi >= 0) scatter(asi, asj);
  __syncthreads();
  CTV_STAMP();
  // IMU group tiles: the knot x knot part (24 x 24 per group, overlapping between consecutive segments); without the LDS
  // Hessian k_assemble_imu adds them
  for (int gi = part * NW + wave + NGI * per_round; LDSH && gi < ngrp; gi += per_round) {   // groups beyond the prefetched ones
    const ImuGroup grp = d.groups[grp0 + gi];
    const T *tile = d.imu_tiles + (size_t)(grp0 + gi) * 1024;
    T tv[9];   // 24 x 24 = 9 x 64 entries: all loads in flight together
#pragma unroll
    for (int u = 0; u < 9; ++u) { const int e = lane + 64 * u; tv[u] = tile[(e / 24) * 32 + e % 24]; }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int e = lane + 64 * u, a = e / 24, b = e % 24;
      const int ga = imu_col(a, grp.s, K, grp.bias), gb = imu_col(b, grp.s, K, grp.bias);
      if (ga >= gb) atomicAdd(&Hs[ga * (ga + 1) / 2 + gb], (double)tv[u]);
    }
    if (STORE && lane < 24) atomicAdd(&gs[imu_col(lane, grp.s, K, grp.bias)], (double)tile[lane * 32 + 30]);
  }
  __syncthreads();
  CTV_STAMP();
  if constexpr (STORE) {
    if (nparts > 1) {   // this part's packed Hessian + gradient: summed with the others, in part order, by k_reduce_finalize
      double *dst = d.Hpart + ((size_t)w * nparts + part) * d.npart_stride;
      for (int i = tid; i < nH; i += NT) dst[i] = Hs[i];
      return;
    }
    double *gq = d.gS[tgset] + u0;
    for (int i0 = tid; i0 < nHh; i0 += 4 * NT) {     // 4 entries per trip: the LDS reads first, then decode + prior + store
      double hv4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) hv4[u] = Hs[min(i0 + NT * u, nHh - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + NT * u;
        if (i >= nHh) continue;
        int ga, gb;
        packed_decode(i, tri, K6, P, ga, gb);
        Hg[(long long)ga * ldh + gb] = hv4[u] + prior_H(d, m, ga, gb);
      }
    }
    for (int i = tid; i < K6 + 1; i += NT) { const int uu = i < K6 ? i : P - 1; gq[uu] = gs[i] + prior_g(d, m, uu); }
    return;   // (the bias rows: k_bias_rows -- a gather with dependent loads wants more waves per CU than this kernel's LDS allows)
  }
  for (int i0 = tid; LDSH && i0 < nHh; i0 += 4 * NT) {     // 4 entries per trip: the LDS reads first, then decode + store
    double hv4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) hv4[u] = Hs[min(i0 + NT * u, nHh - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + NT * u;
      const double hv = hv4[u];
      if (i >= nHh || (nparts > 1 && hv == 0.0)) continue;
      int ga, gb;
      packed_decode(i, tri, K6, P, ga, gb);
      if (nparts > 1) atomicAdd(&Hg[(long long)ga * ldh + gb], hv);
      else Hg[(long long)ga * ldh + gb] = hv;  // first writer after k_zero_normal; later kernels add atomically
    }
  }
  CTV_STAMP();
  for (int i = tid; i < K6 + 1; i += NT) {
    const double gv = gs[i];
    if (gv != 0.0) atomicAdd(&d.gS[tgset][u0 + (i < K6 ? i : P - 1)], gv);
  }
  CTV_STAMP();
#undef CTV_STAMP
}

// Bias rows of the single-part store-semantics assembly: grid (blocks, windows).
template <class T> __global__ __launch_bounds__(256) void k_bias_rows(Dev<T> d, int mode) {
  const int w = blockIdx.y;
  if (!lin_run(d.lm[w], mode) || lin_cost_only(d.lm[w], mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  if (!m.vis_lds) return;
  bias_rows_store(d, m, lin_target(d.lm[w], mode), mode == LIN_SPEC ? d.cbias : d.bias, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// Several parts per window (batches smaller than the chip; the deterministic mode): sum of the parts' packed Hessians in part
// order, prior added, plain stores; the bias rows by gather.  Grid (blocks, windows).
template <class T> __global__ __launch_bounds__(256) void k_reduce_finalize(Dev<T> d, int mode, int nparts) {
  const int w = blockIdx.y;
  if (!lin_run(d.lm[w], mode) || lin_cost_only(d.lm[w], mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  if (!m.vis_lds) return;
  const int tg = lin_target(d.lm[w], mode);
  const int P = m.P, K6 = 6 * m.K, tri = K6 * (K6 + 1) / 2, nHh = tri + K6 + 1, nH = nHh + K6 + 1, ldh = m.ldh;
  double *Hg = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  const double *src = d.Hpart + (size_t)w * nparts * d.npart_stride;
  const int first = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (int i = first; i < nH; i += stride) {
    double v = 0.0;
    int p = 0;
    for (; p + 8 <= nparts; p += 8) {   // eight loads in flight, added in part order
      double t8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t8[q] = src[(size_t)(p + q) * d.npart_stride + i];
#pragma unroll
      for (int q = 0; q < 8; ++q) v += t8[q];
    }
    for (; p < nparts; ++p) v += src[(size_t)p * d.npart_stride + i];
    if (i < nHh) {
      int ga, gb;
      packed_decode(i, tri, K6, P, ga, gb);
      Hg[(long long)ga * ldh + gb] = v + prior_H(d, m, ga, gb);
    } else {
      const int uu = (i - nHh) < K6 ? (i - nHh) : P - 1;
      g[uu] = v + prior_g(d, m, uu);
    }
  }
  bias_rows_store(d, m, tg, mode == LIN_SPEC ? d.cbias : d.bias, first, stride);
}

// ------------------------------------------------------------------------------------------------ bias chain + prior
__device__ __forceinline__ const double *prior_block_ptr(const WinMeta &m, int kind, int idx, const double *quat, const double *pos,
                                                         const double *bias, const double *ldp, int w) {
  switch (kind) {
    case 0: return quat + 4 * (m.knot0 + idx);
    case 1: return pos + 3 * (m.knot0 + idx);
    case 2: return bias + 6 * (m.bias0 + idx);
    case 3: return bias + 6 * (m.bias0 + idx) + 3;
    default: return ldp + w;
  }
}

// BiasFactor (trajectory_value_factor.h:45-99) and MarginalizationFactor (marginalization_factor.cpp:326-373), fp64.
// With the prior written r = r0 + J0 dx:  J^T r = J0^T r0 + (J0^T J0) dx,  |r|^2 = r0^T r0 + 2 b0.dx + dx^T (J0^T J0) dx.
// Adds to Hpp / g of the set the mode selects (not on a cost-only pass) and stores the window's cost share (Dev::misc_cost).
// store != 0 (store-semantics assembly tail): nothing is added here -- the prior's gradient J0^T r0 + (J0^T J0) dx goes to Dev::pgrad,
// and the assembly looks the prior and the chain up when it writes each entry.
template <class T>
__global__ __launch_bounds__(256) void k_misc(Dev<T> d, int mode, int store) {
  const int w = blockIdx.x;
  const Lm &lm = d.lm[w];
  if (!lin_run(lm, mode)) return;
  const bool LIN = !lin_cost_only(lm, mode, d.prm) && !store;
  const WinMeta &m = d.wins[w];
  const bool at_cand = mode == LIN_SPEC;
  const double *quat = at_cand ? d.cquat : d.quat, *pos = at_cand ? d.cpos : d.pos, *bias = at_cand ? d.cbias : d.bias, *ldp = at_cand ? d.cld : d.ld;
  const int tg = lin_target(lm, mode);
  double *Hpp = d.HppS[tg] + m.H0, *g = d.gS[tg] + m.u0;
  extern __shared__ __attribute__((aligned(16))) double smd[];
  double *dx = smd;                 // [pn]
  __shared__ double red[256];
  const int tid = threadIdx.x;
  double cost = 0.0;
  for (int e = tid; e < m.NB * 6; e += 256) {
    const int b = e / 6, k = e % 6;
    const int bi = d.bc_i[m.bc0 + b], bj = d.bc_j[m.bc0 + b];
    const double wv = d.bc_w[(size_t)(m.bc0 + b) * 6 + k];
    const double r = wv * (bias[6 * (m.bias0 + bj) + k] - bias[6 * (m.bias0 + bi) + k]);
    cost += 0.5 * r * r;
    if (LIN) {
      const int ii = 6 * m.K + 6 * bi + k, jj = 6 * m.K + 6 * bj + k;
      atomicAdd(&g[ii], -wv * r);
      atomicAdd(&g[jj], wv * r);
      atomicAdd(&Hpp[(long long)ii * m.ldh + ii], wv * wv);
      atomicAdd(&Hpp[(long long)jj * m.ldh + jj], wv * wv);
      const int hi = max(ii, jj), lo = min(ii, jj);
      atomicAdd(&Hpp[(long long)hi * m.ldh + lo], -wv * wv);
    }
  }
  const int n = m.pn;
  if (n > 0) {
    for (int i = tid; i < n; i += 256) dx[i] = 0.0;
    __syncthreads();
    for (int b = tid; b < m.pnb; b += 256) {
      const int kind = d.p_kind[m.pblk0 + b], idx = d.p_index[m.pblk0 + b], off = d.p_off[m.pblk0 + b];
      const double *x = prior_block_ptr(m, kind, idx, quat, pos, bias, ldp, w);
      const double *x0 = d.p_x0 + 4 * (size_t)(m.pblk0 + b);
      if (kind == 0) {  // dx = 2 vec(q0^-1 q), sign-fixed (marginalization_factor.cpp:344-350)
        const Q4<double> dq = qmul_raw(qmk<double>(-x0[0], -x0[1], -x0[2], x0[3]), qmk<double>(x[0], x[1], x[2], x[3]));
        const double sg = (dq.w >= 0) ? 2.0 : -2.0;
        dx[off] = sg * dq.x; dx[off + 1] = sg * dq.y; dx[off + 2] = sg * dq.z;
      } else {
        const int sz = (kind == 4) ? 1 : 3;
        for (int k = 0; k < sz; ++k) dx[off + k] = x[k] - x0[k];
      }
    }
    __syncthreads();
    const double *pH = d.pH + m.pH0, *b0 = d.pb0 + m.pv0;
    const int *pcol = d.pcol + m.pv0;
    for (int i = tid; i < n; i += 256) {
      double hd = 0.0;
      for (int j = 0; j < n; ++j) hd += pH[(size_t)j * n + i] * dx[j];   // (J0^T J0 is symmetric: column i, coalesced over the threads)
      cost += dx[i] * (b0[i] + 0.5 * hd);
      if (store) d.pgrad[m.pv0 + i] = b0[i] + hd;
      if (LIN && pcol[i] >= 0) atomicAdd(&g[pcol[i]], b0[i] + hd);
    }
    if (tid == 0) cost += 0.5 * d.pc0[w];
    if (LIN) {
      for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e % n;
        const int ci = pcol[i], cj = pcol[j];
        if (ci >= 0 && cj >= 0 && ci >= cj) atomicAdd(&Hpp[(long long)ci * m.ldh + cj], pH[e]);
      }
    }
  }
  red[tid] = cost;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
  if (tid == 0) {
    d.misc_cost[w] = red[0];
    if (store && !lin_cost_only(lm, mode, d.prm)) {   // (the generic path resets these in k_zero_normal)
      if (mode == LIN_SPEC) d.lm[w].cand_gmax_bits = 0ull; else d.lm[w].gmax_bits = 0ull;
    }
  }
}

// Jacobi scaling (computed once, at iteration 0: Ceres jacobi_scaling), gradient max-norm of
// x - Plus(x, -g) (Ceres gradient_max_norm) and |x|^2 of the reduced program.
// |x - Plus(x, -g)| of unknown j (Ceres gradient_max_norm: ambient difference for a rotation block, the box of the line delay)
template <class T> __device__ __forceinline__ double grad_norm_entry(const Dev<T> &d, const WinMeta &m, int w, int j, const double *g, bool at_cand) {
  const double *squat = at_cand ? d.cquat : d.quat, *sld = at_cand ? d.cld : d.ld;
  const int K6 = 6 * m.K;
  if (j < K6) {
    const int k = j / 6, c = j % 6;
    if (c == 0) {  // rotation block: ambient difference q - q*exp(-g)
      const double *q = squat + 4 * (m.knot0 + k);
      const Q4<double> q0 = qmk<double>(q[0], q[1], q[2], q[3]);
      const Q4<double> q1 = qmul(q0, so3_exp(mk<double>(-g[j], -g[j + 1], -g[j + 2])));
      return fmax(fmax(fabs(q0.x - q1.x), fabs(q0.y - q1.y)), fmax(fabs(q0.z - q1.z), fabs(q0.w - q1.w)));
    }
    return c >= 3 ? fabs(g[j]) : 0.0;
  }
  if (j == m.P - 1) {
    const double ld = sld[w];
    double nl = ld - g[j];
    if (!m.fix_ld) nl = fmin(fmax(nl, m.ld_lo), m.ld_hi);
    return fabs(ld - nl);
  }
  return fabs(g[j]);
}

// After the first linearisation of a solve (LIN_AT_X): Jacobi scaling (computed once, at iteration 0: Ceres jacobi_scaling) and the
// gradient max-norm of the initial state.  (Every later pass: k_pass_end.)
template <class T> __global__ void k_post_linearize(Dev<T> d, int mode) {
  const int w = blockIdx.y;
  Lm &lm = d.lm[w];
  if (!lin_run(lm, mode) || lin_cost_only(lm, mode, d.prm)) return;
  const WinMeta &m = d.wins[w];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m.N) return;
  const bool at_cand = mode == LIN_SPEC;
  const int tg = lin_target(lm, mode);
  const bool act = d.active[m.u0 + j] != 0;
  if (!at_cand && !lm.scaled) {
    const double h = (j < m.P) ? d.HppS[tg][m.H0 + (long long)j * m.ldh + j] : d.HllS[tg][m.lm0 + j - m.P];
    d.cscale[m.u0 + j] = act ? 1.0 / (1.0 + sqrt(fmax(h, 0.0))) : 1.0;
  }
  if (!act) return;
  const double gm = grad_norm_entry(d, m, w, j, d.gS[tg] + m.u0, at_cand);
  if (gm > 0.0) atomicMax(at_cand ? &lm.cand_gmax_bits : &lm.gmax_bits, (unsigned long long)__double_as_longlong(gm));
}

// ------------------------------------------------------------------------------------------------ Schur + solve
// Start of an iteration, by the threads of one workgroup.  Thread 0: FinalizeIterationAndCheckIfMinimizerCanContinue (windows inside
// the line search only report that they are still running).  Then, for the windows that start an iteration: the LM diagonal
// D^2 = clamp(diag(J^T J), min, max) / mu on the Jacobi-scaled system (Ceres LevenbergMarquardtStrategy::ComputeStep), expressed for
// the unscaled system: dd_j = clamp(c_j^2 H_jj) / (mu c_j^2), and 1 / (Hll + dd) of the landmarks.
template <class T> __device__ __forceinline__ void begin_iteration(const Dev<T> &d, int w, int *s_go) {
  Lm &lm = d.lm[w];
  if (threadIdx.x == 0) {
    int go = 0;
    if (!lm.status) {
      if (lm.ls_active) atomicAdd(d.n_active, 1);   // inside the line search: no new LM iteration
      else if (lm.iter >= d.prm.max_iters) lm.status = 1 + 0;
      else if (lm.last_ok && __longlong_as_double((long long)lm.gmax_bits) <= d.prm.gtol) lm.status = 1 + 1;
      else if (lm.mu <= d.prm.min_radius) lm.status = 1 + 4;
      else {
        lm.iter += 1;
        lm.accept = 0; lm.step_valid = 0; lm.chol_fail = 0; lm.alpha = 1.0;
        atomicAdd(d.n_active, 1);
        go = 1;
      }
    }
    *s_go = go;
  }
  __syncthreads();
  if (!*s_go) return;
  const WinMeta &m = d.wins[w];
  const double mu = lm.mu;
  const double *Hd = d.HppS[lm.cur] + m.H0, *Hl = d.HllS[lm.cur] + m.lm0;
  for (int j = threadIdx.x; j < m.N; j += blockDim.x) {
    const bool act = d.active[m.u0 + j] != 0;
    const double c = d.cscale[m.u0 + j];
    const double h = (j < m.P) ? Hd[(long long)j * m.ldh + j] : Hl[j - m.P];
    const double sc = fmin(fmax(c * c * h, d.prm.min_diag), d.prm.max_diag);
    const double dd = act ? sc / (mu * c * c) : 0.0;
    d.dd[m.u0 + j] = dd;
    if (j >= m.P) d.dinv[m.lm0 + j - m.P] = (act && (h + dd) > 0.0) ? 1.0 / (h + dd) : 0.0;
  }
}
// The first iteration of a solve (every later one starts at the end of the previous pass: k_pass_end).
template <class T> __global__ __launch_bounds__(256) void k_begin_iter(Dev<T> d) {
  __shared__ int s_go;
  begin_iteration(d, blockIdx.x, &s_go);
}

__device__ __forceinline__ void tile_decode(int t, int &bi, int &bj) {  // t -> (bi >= bj), row-major over the lower triangle
  bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  bj = t - bi * (bi + 1) / 2;
}

// fp64 product path, large batches: the window kernel on the fp64 matrix cores.  One workgroup (8 waves) per window; W is read
// from HBM once, staged through LDS in double-buffered chunks of 16 landmarks (masked by the active flags, g_rho appended as
// column P so that the tile row holding index P also produces the reduced right-hand side: no k_rhs pass).  Output tiles are
// 16 x 16 (v_mfma_f64_16x16x4_f64, K = 4 landmarks per instruction); tile t of the lower triangle belongs to wave t % 8, which
// keeps its <= NTQ accumulators in registers over the whole landmark loop; tiles over bias-only columns have no products.
// NPRE = compact chunk elements per thread (16 (6K + 2) / 512 rounded up); NTQ = ceil(tiles with products / 8).
template <int NPRE, int NTQ> __global__ __launch_bounds__(512, NTQ <= 7 ? 4 : 2) void k_schur_window_f64(Dev<double> d) {
  const int w = blockIdx.x;
  if (d.lm[w].status || d.lm[w].ls_active) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, L = m.L, ldw = m.ldw, u0 = m.u0, K6 = 6 * m.K, ldh = m.ldh;
  const int nt = ldw >> 4, ntile = nt * (nt + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) double smd64[];
  double *Wb = smd64;                    // [2][16][ldw]
  double *acts = Wb + 2 * 16 * ldw;      // [ldw] 1 / 0 (0 beyond P)
  double *dch = acts + ldw;              // [2][16] 1 / (Hll + D) of the chunk's landmarks (0 beyond L)
  int *tlist = reinterpret_cast<int *>(dch + 32);   // [8 NTQ] tiles with products (bi << 8 | bj), any order
  int &tcount = tlist[8 * NTQ];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, q4 = lane >> 4, l15 = lane & 15;
  const double *Wp = d.WS[d.lm[w].cur] + m.W0;
  const double *dinv = d.dinv + m.lm0, *gl = d.gS[d.lm[w].cur] + u0 + P;
  for (int c = tid; c < ldw; c += 512) acts[c] = (c < P && d.active[u0 + min(c, P - 1)]) ? 1.0 : 0.0;
  if (tid == 0) tcount = 0;
  // W is non-zero only in the knot columns [0, 6K) and the line-delay column P - 1 (plus the rhs row P): a tile has products
  // when its row tile and its column tile both hold such a column.  Those tiles (55 of 105 at K = 24) are listed and dealt to
  // the waves; the others only need the epilogue (S = Hpp + D).
  auto nz_row = [&](int b) { return (16 * b < K6) || (P >= 16 * b && P - 1 < 16 * b + 16); };
  auto nz_col = [&](int b) { return (16 * b < K6) || (P - 1 >= 16 * b && P - 1 < 16 * b + 16); };
  __syncthreads();   // tcount
  for (int t = tid; t < ntile; t += 512) {
    int ti, tj;
    tile_decode(t, ti, tj);
    if (nz_row(ti) && nz_col(tj)) { const int pos = atomicAdd(&tcount, 1); if (pos < 8 * NTQ) tlist[pos] = (ti << 8) | tj; }
  }
  // Only the knot columns [0, 6K), the line-delay column P - 1 and the appended g_rho column P are fetched and staged (NC
  // compact columns per landmark); every other column of the two LDS buffers is zeroed once and stays zero.
  const int nchunk = (L + 15) >> 4, nel = 16 * ldw, NC = K6 + 2, nelc = 16 * NC;
  for (int e = tid; e < 2 * nel; e += 512) Wb[e] = 0.0;
  double pre[NPRE];
  double pre_d = 0.0;
  int pre_lc[NPRE];     // chunk row << 16 | window column of this thread's elements (the same for every chunk)
#pragma unroll
  for (int k = 0; k < NPRE; ++k) {
    const int e = min(tid + 512 * k, nelc - 1), cc = e % NC;
    pre_lc[k] = ((e / NC) << 16) | (cc < K6 ? cc : P - 1 + (cc - K6));
  }
  auto fetch = [&](int ch) {     // unconditional loads on clamped rows; masked when stored
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int l = min(16 * ch + (pre_lc[k] >> 16), L - 1), c = pre_lc[k] & 0xffff;
      pre[k] = (c == P) ? gl[l] : Wp[(long long)l * ldw + c];
    }
    if (tid < 16) pre_d = dinv[min(16 * ch + tid, L - 1)];
  };
  auto stash = [&](int ch, int buf) {
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int lr = pre_lc[k] >> 16, c = pre_lc[k] & 0xffff;
      const bool lv = 16 * ch + lr < L;
      if (tid + 512 * k < nelc) Wb[buf * nel + lr * ldw + c] = lv ? ((c == P) ? pre[k] : pre[k] * acts[c]) : 0.0;
    }
    if (tid < 16) dch[16 * buf + tid] = (16 * ch + tid < L) ? pre_d : 0.0;
  };
  __syncthreads();   // acts, zeroed buffers, tile list
  const int nact = min(tcount, 8 * NTQ);
  // this wave's tiles: slot q holds list entry wave + 8 q; slots past the end repeat the wave's first tile (products computed,
  // result dropped) so that the tile loop below has no branches and the operand reads of a tile overlap the previous products
  int tij[NTQ];
#pragma unroll
  for (int q = 0; q < NTQ; ++q) tij[q] = __builtin_amdgcn_readfirstlane(tlist[(wave + 8 * q < nact) ? wave + 8 * q : min(wave, max(nact - 1, 0))]);   // SGPRs
  f64x4 acc[NTQ];
#pragma unroll
  for (int q = 0; q < NTQ; ++q) acc[q] = f64x4{0.0, 0.0, 0.0, 0.0};
  if (nchunk > 0) { fetch(0); stash(0, 0); }
  __syncthreads();
  for (int ch = 0; ch < nchunk && nact > 0; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nchunk) fetch(ch + 1);
    const double *B = Wb + buf * nel + q4 * ldw + l15;
    double dl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) dl[s] = dch[16 * buf + 4 * s + q4];
#pragma unroll
    for (int q = 0; q < NTQ; ++q) {
      double a[4], b[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        a[s] = B[4 * s * ldw + 16 * (tij[q] >> 8)];
        b[s] = B[4 * s * ldw + 16 * (tij[q] & 255)];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s] * dl[s], acc[q], 0, 0, 0);
    }
    if (ch + 1 < nchunk) stash(ch + 1, buf ^ 1);
    __syncthreads();
  }
  // epilogue: S = Hpp - W^T Hll^-1 W + D on the active lower triangle, identity rows for fixed unknowns; rhs row.
  double *S = d.S + m.H0, *rhs = d.rhs + m.p0;
  const double *H = d.HppS[d.lm[w].cur] + m.H0;
  auto write_tile = [&](int ti, int tj, const f64x4 &av) {
    const int jj = 16 * tj + l15, jc = min(jj, P - 1);
    const bool act_j = d.active[u0 + jc] != 0;
    const double dd_j = d.dd[u0 + jc], g_j = d.gS[d.lm[w].cur][u0 + jc];
    double hv[4];
    unsigned char act_i[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ic = min(16 * ti + q4 + 4 * r, P - 1);
      act_i[r] = d.active[u0 + ic];
      hv[r] = H[(long long)ic * ldh + min(jc, ic)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ii = 16 * ti + q4 + 4 * r;
      if (ii < P && jj <= ii) {
        const bool on = act_i[r] && act_j;
        S[(long long)ii * ldh + jj] = on ? hv[r] - av[r] + (ii == jj ? dd_j : 0.0) : (ii == jj ? 1.0 : 0.0);
      } else if (ii == P && jj < P) {
        rhs[jj] = act_j ? av[r] - g_j : 0.0;
      }
    }
  };
#pragma unroll
  for (int q = 0; q < NTQ; ++q) {
    if (wave + 8 * q >= nact) continue;
    write_tile(tij[q] >> 8, tij[q] & 255, acc[q]);
  }
  const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
  for (int t = wave; t < ntile; t += 8) {     // tiles without products
    int ti, tj;
    tile_decode(t, ti, tj);
    if (nz_row(ti) && nz_col(tj)) continue;
    write_tile(ti, tj, zero4);
  }
}

// fp64 path: the same SYRK on the fp64 matrix cores, one wave per 16 x 16 tile of the lower triangle
// (v_mfma_f64_16x16x4_f64: A operand lane l = X[k = l/16][i = l%16], B operand lane l = Y[k = l/16][j = l%16],
// D register r of lane l = D[(l/16) + 4r][l%16]; measured with tools/mfma_f64_layout.hip).  Operands straight from W,
// 16 landmarks (4 products) per trip with all loads of a trip in flight; the reduced rhs is left to k_rhs.
__global__ __launch_bounds__(64) void k_schur_tile_f64(Dev<double> d, int ntile_max) {
  // XCD-aware tile -> workgroup map: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the tiles of one
  // window get ids that are congruent mod 8: they all run on one XCD and the window's W (re-read by every tile) comes out of
  // that L2 instead of being fetched 8 times over the fabric.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int w = (slot / ntile_max) * 8 + xcd, tile = slot % ntile_max;
  if (w >= d.nwin) return;
  if (d.lm[w].status || d.lm[w].ls_active) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, L = m.L, ldw = m.ldw, u0 = m.u0, K6 = 6 * m.K, ldh = m.ldh;
  const int nt = P / 16 + 1;   // tile rows up to index P: the rhs rides along as row P (g_rho on the A side), so the tile row that
  if (tile >= nt * (nt + 1) / 2) return;   // holds it also produces W^T diag(dinv) g_rho -- no separate k_rhs pass
  int bi, bj;
  tile_decode(tile, bi, bj);
  const int lane = threadIdx.x, q4 = lane >> 4, l15 = lane & 15;
  const int i = min(16 * bi + l15, ldw - 1), j = min(16 * bj + l15, ldw - 1);
  const bool rhs_lane = 16 * bi + l15 == P;
  const double ai = (16 * bi + l15 < P && d.active[u0 + min(i, P - 1)]) ? 1.0 : 0.0;
  const double aj = (16 * bj + l15 < P && d.active[u0 + min(j, P - 1)]) ? 1.0 : 0.0;
  const double *Wp = d.WS[d.lm[w].cur] + m.W0;
  const double *dinv = d.dinv + m.lm0, *gl = d.gS[d.lm[w].cur] + u0 + P;
  // W is non-zero only in the knot columns [0, 6K) and the line-delay column P-1: tiles over bias columns skip the loop
  const bool nz_i = (16 * bi < K6) || (P >= 16 * bi && P - 1 < 16 * bi + 16);
  const bool nz_j = (16 * bj < K6) || (P - 1 >= 16 * bj && P - 1 < 16 * bj + 16);
  const int lend = (nz_i && nz_j) ? L : 0;   // L == 0: the loop (and its clamped row L - 1) is skipped
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int l0 = 0; l0 < lend; l0 += 16) {
    double wa[4], wb[4], dv[4], gv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {   // unconditional loads on clamped rows, masked below
      const int lc = min(l0 + 4 * s + q4, L - 1);
      wa[s] = Wp[(long long)lc * ldw + i];
      wb[s] = Wp[(long long)lc * ldw + j];
      dv[s] = dinv[lc];
      gv[s] = gl[lc];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) wa[s] = rhs_lane ? gv[s] : wa[s] * ai;   // (unconditional loads, selected afterwards)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bool lv = l0 + 4 * s + q4 < L;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[s], lv ? wb[s] * aj * dv[s] : 0.0, acc, 0, 0, 0);
    }
  }
  double *S = d.S + m.H0, *rhs = d.rhs + m.p0;
  const double *H = d.HppS[d.lm[w].cur] + m.H0;
  const int jj = 16 * bj + l15, jc = min(jj, P - 1);
  const bool act_j = d.active[u0 + jc] != 0;
  const double dd_j = d.dd[u0 + jc], g_j = d.gS[d.lm[w].cur][u0 + jc];
  double hv[4];
  unsigned char act_i[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ic = min(16 * bi + q4 + 4 * r, P - 1);
    act_i[r] = d.active[u0 + ic];
    hv[r] = H[(long long)ic * ldh + min(jc, ic)];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ii = 16 * bi + q4 + 4 * r;
    if (ii < P && jj <= ii) {
      const bool on = act_i[r] && act_j;
      S[(long long)ii * ldh + jj] = on ? hv[r] - acc[r] + (ii == jj ? dd_j : 0.0) : (ii == jj ? 1.0 : 0.0);
    } else if (ii == P && jj < P) {
      rhs[jj] = act_j ? acc[r] - g_j : 0.0;   // reduced right-hand side: -g_p + W^T diag(dinv) g_rho
    }
  }
}

template <class T> __global__ void k_schur_generic(Dev<T> d) {
  const int w = blockIdx.y;
  if (d.lm[w].status || d.lm[w].ls_active) return;
  const WinMeta &m = d.wins[w];
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)m.P * m.P) return;
  const int ii = (int)(e / m.P), jj = (int)(e % m.P);
  if (jj > ii) return;
  const bool on = d.active[m.u0 + ii] && d.active[m.u0 + jj];
  double val;
  if (on) {
    const T *Wp = d.WS[d.lm[w].cur] + m.W0;
    double acc = 0.0;
    for (int l = 0; l < m.L; ++l) acc += (double)Wp[(long long)l * m.ldw + ii] * (double)Wp[(long long)l * m.ldw + jj] * d.dinv[m.lm0 + l];
    val = d.HppS[d.lm[w].cur][m.H0 + (long long)ii * m.ldh + jj] - acc + (ii == jj ? d.dd[m.u0 + ii] : 0.0);
  } else {
    val = (ii == jj) ? 1.0 : 0.0;
  }
  d.S[m.H0 + (long long)ii * m.ldh + jj] = val;
}

// rhs_p = -g_p + W^T diag(dinv) g_l.  256 threads = 64 unknowns x 4 landmark slices (coalesced over the unknowns).
template <class T> __global__ __launch_bounds__(256) void k_rhs(Dev<T> d) {
  const int w = blockIdx.y;
  if (d.lm[w].status || d.lm[w].ls_active) return;
  const WinMeta &m = d.wins[w];
  __shared__ double part[4][64];
  const int li = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + li;
  double v = 0.0;
  if (i < m.P && d.active[m.u0 + i]) {
    const T *Wp = d.WS[d.lm[w].cur] + m.W0 + i;
    const double *dinv = d.dinv + m.lm0, *gl = d.gS[d.lm[w].cur] + m.u0 + m.P;
    for (int l = sl; l < m.L; l += 4) v += (double)Wp[(long long)l * m.ldw] * (dinv[l] * gl[l]);
  }
  part[sl][li] = v;
  __syncthreads();
  if (sl == 0 && i < m.P) {
    const double s = part[0][li] + part[1][li] + part[2][li] + part[3][li];
    d.rhs[m.p0 + i] = d.active[m.u0 + i] ? s - d.gS[d.lm[w].cur][m.u0 + i] : 0.0;
  }
}

// Dense fp64 Cholesky of the P x P reduced system + solve, one workgroup (4 waves) per window, right-looking with
// 32-column panels, the matrix products on the fp64 matrix cores (v_mfma_f64_16x16x4_f64):
//   1. wave 0 factors the 32 x 32 diagonal block, one row per lane in registers, with v_readlane broadcasts (no LDS,
//      no barriers inside the 32 pivot steps) and forms L11^-1 in the same sweep (lane = column of the inverse).
//      Meanwhile waves 1-3 stage the panel rows A21 (and the rhs row) into LDS, k-major.
//   2. L21 = A21 L11^-T as an MFMA product, in place in the LDS panel (a 16-row tile is owned by one wave);
//   3. trailing update A22 -= L21 L21^T: one 16 x 16 tile per wave at a time, 8 MFMAs, read-modify-write of S.
// The right-hand side rides along as an extra matrix row (Cholesky of [S b; b^T .]), so y = L^-1 b needs no
// separate forward substitution; only the block back-substitution L^T x = y remains.  Result in delta[0..P).
// MFMA register layout (measured, tools/mfma_f64_layout.hip): A operand lane l = A[l%16][l/16], B operand lane l =
// B[l/16][l%16], D register r of lane l = D[(l/16) + 4r][l%16].

// Diagonal block of k_cholesky_solve: factorisation fused with the inversion, on ONE register array.  Lanes 0-31 hold the rows
// of the block (v[c] = A[lane][c]), lanes 32-63 the columns of X = L11^-1 in the making (v[c] = X[c][lane - 32], identity at the
// start).  The rank-1 update of pivot J, a_c -= a_J s with s = L[C][J] = v[J] of lane C, is also the substitution step
// x_c -= x_J s of the inverse: one v_readlane pair and ONE v_fma per (J, C) serve both halves of the wave.
// One update as an asm block so that the broadcast value lives for exactly these instructions (left to the compiler, every
// broadcast was spilled and reloaded).
template <int C> __device__ __forceinline__ void chol_bcast_update(double &vc, double vj, int vj_lo, int vj_hi) {
  asm("v_readlane_b32 s96, %2, %4\n\tv_readlane_b32 s97, %3, %4\n\ts_nop 1\n\t"
      "v_fma_f64 %0, -%1, s[96:97], %0"
      : "+v"(vc)
      : "v"(vj), "v"(vj_lo), "v"(vj_hi), "n"(C)
      : "s96", "s97");
}
// The same for the first column after the pivot, whose result feeds the next pivot's v_readlane straight away: gfx950 needs a
// wait state between a VALU write of a VGPR and a v_readlane of it (and between the compiler's scaling of v[J] and the first
// v_readlane here); the hazard recogniser cannot see into an asm block, so the s_nops are spelled out.
template <int C> __device__ __forceinline__ void chol_bcast_update_first(double &vc, double vj, int vj_lo, int vj_hi) {
  asm("s_nop 1\n\tv_readlane_b32 s96, %2, %4\n\tv_readlane_b32 s97, %3, %4\n\ts_nop 1\n\t"
      "v_fma_f64 %0, -%1, s[96:97], %0\n\ts_nop 1"
      : "+v"(vc)
      : "v"(vj), "v"(vj_lo), "v"(vj_hi), "n"(C)
      : "s96", "s97");
}
// 1 / sqrt(p) of the pivot: hardware estimate + two Newton steps (short dependent chain instead of sqrt + divide)
__device__ __forceinline__ double chol_pivot_rsqrt(double pj, int &bad) {
  const bool ok = (pj > 0.0) && isfinite(pj);
  if (!ok) bad = 1;
  const double ps = ok ? pj : 1.0;
  double di = __builtin_amdgcn_rsq(ps);
  const double hp = 0.5 * ps;
  di = di * (1.5 - hp * di * di);
  di = di * (1.5 - hp * di * di);
  return di;
}
// Four columns at once, each broadcast in its own SGPR pair: with a single pair every update waited for the previous FMA to
// release it (~42 cycles per update, measured: 25 k cycles per 32 x 32 block); here the eight v_readlane run ahead of the
// four FMAs, which also puts the two wait states gfx950 wants between a VALU write of an SGPR and its VALU read in between.
template <int C> __device__ __forceinline__ void chol_bcast_update4(double &v0, double &v1, double &v2, double &v3, double vj, int vj_lo, int vj_hi) {
  asm("v_readlane_b32 s92, %5, %7\n\tv_readlane_b32 s93, %6, %7\n\t"
      "v_readlane_b32 s94, %5, %8\n\tv_readlane_b32 s95, %6, %8\n\t"
      "v_readlane_b32 s96, %5, %9\n\tv_readlane_b32 s97, %6, %9\n\t"
      "v_readlane_b32 s98, %5, %10\n\tv_readlane_b32 s99, %6, %10\n\t"
      "v_fma_f64 %0, -%4, s[92:93], %0\n\tv_fma_f64 %1, -%4, s[94:95], %1\n\t"
      "v_fma_f64 %2, -%4, s[96:97], %2\n\tv_fma_f64 %3, -%4, s[98:99], %3"
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)
      : "v"(vj), "v"(vj_lo), "v"(vj_hi), "n"(C), "n"(C + 1), "n"(C + 2), "n"(C + 3)
      : "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99");
}
// columns C .. 31 of pivot J
template <int J, int C> __device__ __forceinline__ void chol_row_updates(double (&v)[32], int lo, int hi) {
  if constexpr (C + 3 <= 31) {
    chol_bcast_update4<C>(v[C], v[C + 1], v[C + 2], v[C + 3], v[J], lo, hi);
    chol_row_updates<J, C + 4>(v, lo, hi);
  } else if constexpr (C <= 31) {
    chol_bcast_update<C>(v[C], v[J], lo, hi);
    chol_row_updates<J, C + 1>(v, lo, hi);
  }
}
// Pivot J with its 1 / sqrt already known (di): scale column J, update column J + 1 first, start the NEXT pivot's reciprocal
// square root from it (its dependent chain of ~10 fp64 operations then overlaps the remaining updates), update the rest.
template <int J> __device__ __forceinline__ double chol_diag_step(double (&v)[32], double di, int &bad) {
  v[J] *= di;   // lanes < 32: lane J sqrt(p_J), lanes > J L[i][J]; lanes >= 32: X[J][.], final (every k < J has been eliminated)
  const int lo = __double2loint(v[J]), hi = __double2hiint(v[J]);
  double di_next = 0.0;
  if constexpr (J < 31) {
    chol_bcast_update_first<J + 1>(v[J + 1], v[J], lo, hi);
    di_next = chol_pivot_rsqrt(readlane_d(v[J + 1], J + 1), bad);
    if constexpr (J < 30) chol_row_updates<J, J + 2>(v, lo, hi);
  }
  return di_next;
}
template <int J> __device__ __forceinline__ void chol_diag_from(double (&v)[32], double di, int &bad) {
  const double dn = chol_diag_step<J>(v, di, bad);
  if constexpr (J < 31) chol_diag_from<J + 1>(v, dn, bad);
}
__device__ __forceinline__ void chol_diag_all(double (&v)[32], int &bad) {
  chol_diag_from<0>(v, chol_pivot_rsqrt(readlane_d(v[0], 0), bad), bad);
}
// NW waves per window: 4 for large batches (two windows share a CU), 8 when there are fewer windows than CUs (the parallel
// phases -- L21, trailing update, staging -- go twice as fast; the diagonal blocks hide behind the trailing updates).
template <class T, int NW> __global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_cholesky_solve(Dev<T> d) {
  constexpr int NT = 64 * NW;
  const int w = blockIdx.x;
  Lm &lm = d.lm[w];
  if (lm.status || lm.ls_active) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, ldh = m.ldh, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  extern __shared__ __attribute__((aligned(16))) double smc[];
  double *Lb = smc;                 // [32][34] NEXT diagonal block (row-major), deposited by the trailing update of the current panel
  double *LiT = Lb + 32 * 34;       // [32][34] L11^-1 transposed: LiT[k][j] = Linv[j][k]
  double *dinvs = LiT + 32 * 34;    // [32] 1 / L_jj
  double *yb = dinvs + 32;          // [32]
  int &s_fail = *reinterpret_cast<int *>(yb + 32);
  int &s_trip = reinterpret_cast<int *>(yb + 32)[1];   // next unclaimed tile of the trailing update
  double *LpT = yb + 34;            // [32][RS] panel (+ rhs row) k-major: LpT[k][r]
  double *S = d.S + m.H0;
  double *y = d.rhs + m.p0;         // augmented row; becomes L^-1 rhs
  double *x = d.delta + m.u0;
  if (tid == 0) s_fail = 0;
  for (int e = tid; e < 32 * 32; e += NT) {   // first diagonal block -> LDS (rows / columns clamped; masked when read)
    const int r = e >> 5, c = e & 31;
    Lb[r * 34 + c] = S[(long long)min(r, P - 1) * ldh + min(c, P - 1)];
  }
  __syncthreads();
  long long *dbg = (d.dbg && w == 0) ? d.dbg : nullptr;
  int dbi = 0;
#define CTV_STAMP() do { if (dbg && tid == 0 && dbi < 30) dbg[dbi++] = clock64(); } while (0)
  CTV_STAMP();
  const int q4 = lane >> 4, l15 = lane & 15;
  // ---- diagonal block at column jb (one wave): lanes 0-31 the rows (lanes >= nb of the last, partial block carry identity
  //      rows), lanes 32-63 the columns of the inverse (identity).  The block is in LDS (Lb): the first one staged above, the
  //      later ones left there by the trailing update.  Result: LiT (LDS) and chol_inv (HBM, for the back-substitution); L11
  //      itself is not written back, nothing reads it.
  auto diag_block = [&](int jb) {
    const int nb = min(32, P - jb);
    double v[32];
#pragma unroll
    for (int c = 0; c < 32; c += 2) {
      const VecN<double, 2> v2 = *reinterpret_cast<const VecN<double, 2> *>(Lb + (lane & 31) * 34 + c);
      v[c] = v2.v[0]; v[c + 1] = v2.v[1];
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const bool in = lane < nb && c < nb && c <= lane;
      v[c] = in ? v[c] : ((c == (lane & 31)) ? 1.0 : 0.0);
    }
    int bad = 0;
    chol_diag_all(v, bad);
    if (lane >= 32) {
      const int col = lane - 32;
      double *gi = d.chol_inv + ((size_t)w * d.chol_nblk + (jb >> 5)) * 1024;
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        LiT[col * 34 + i] = v[i];   // LiT[k = col][j = i] = Linv[i][col]
        gi[i * 32 + col] = v[i];    // row-major Linv[i][col]
      }
    }
    if (lane == 0 && bad) s_fail = 1;
  };
  for (int jb = 0; jb < P; jb += 32) {
    const int nb = min(32, P - jb), r0 = jb + nb, nt = P - r0, ntr = nt + 1;  // ntr: trailing rows incl. the rhs row
    const int RS = (ntr + 15) & ~15, ntile = RS >> 4;
    // ---- panel rows (and the rhs row) into the LDS panel, LpT[k][r].  First panel: wave 0 factors the diagonal block
    //      meanwhile; the later diagonal blocks were factored during the previous trailing update (look-ahead, below).
    {
      const int first = jb == 0 ? 64 : 0, nthr = NT - first;
      if (jb == 0 && wave == 0) diag_block(0);
      for (int r = tid - first; r >= 0 && r < RS; r += nthr) {
        const double *src = (r < nt) ? S + (long long)(r0 + r) * ldh + jb : y + jb;
        double tmp[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) tmp[k] = src[min(k, nb - 1)];   // unconditional: 32 loads in flight
        const bool live = r < ntr;
#pragma unroll
        for (int k = 0; k < 32; ++k) LpT[k * RS + r] = (live && k < nb) ? tmp[k] : 0.0;
      }
    }
    if (tid == 0) s_trip = 4;   // wave 0 starts with tiles 0-3 (they hold the next diagonal block)
    // LDS-only barrier: what the next phase reads (LiT, LpT) is in LDS; wave 0's global stores of the block inverse may
    // stay in flight (a full __syncthreads would wait for them; they are read after later full barriers only)
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    CTV_STAMP();
    // ---- L21 = A21 L11^-T, in place: Linv is lower triangular, so output columns 0..15 need k < 16 only
    for (int tr = wave; tr < ntile; tr += NW) {
      f64x4 c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
      const double *pa = LpT + 16 * tr + l15;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int k = 4 * kk + q4;
        const double av = pa[k * RS];
        if (kk < 4) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, LiT[k * 34 + l15], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, LiT[k * 34 + 16 + l15], c1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tr + q4 + 4 * r;
        LpT[l15 * RS + row] = c0[r];
        LpT[(16 + l15) * RS + row] = c1[r];
        if (row < ntr) {
          double *dst = (row < nt) ? S + (long long)(r0 + row) * ldh + jb : y + jb;
          if (l15 < nb) dst[l15] = c0[r];
          if (16 + l15 < nb) dst[16 + l15] = c1[r];
        }
      }
    }
    __syncthreads();
    CTV_STAMP();
    // ---- trailing update A22 -= L21 L21^T on the lower triangle (16 x 16 tiles) and the rhs row.  Trips of 4 consecutive
    //      tiles are claimed from an LDS counter.  Wave 0 takes tiles 0-3 first -- (0,0), (1,0), (1,1) are the next diagonal
    //      block, left in Lb -- then factors that block (LOOK-AHEAD: 21 k cycles on one wave that used to sit between the
    //      panels with three waves idle) while the other waves work through the rest, then joins them.
    const int ntt = nt > 0 ? ntile * (ntile + 1) / 2 : 0;   // last panel: nothing left to update
    // (requesting the next trip's S values before the current products was tried: no gain, it is bandwidth not latency)
    bool first_trip = wave == 0;
    while (true) {   // 4 tiles per trip: 16 loads in flight, 32 MFMAs, 16 stores
      int tb;
      if (first_trip) tb = 0;
      else tb = __builtin_amdgcn_readfirstlane(lane == 0 ? atomicAdd(&s_trip, 4) : 0);
      if (tb >= ntt) break;
      double sv[4][4];
      int ti4[4], tj4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        tile_decode(min(tb + u, ntt - 1), ti4[u], tj4[u]);
        const int col = 16 * tj4[u] + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti4[u] + q4 + 4 * r;
          const double *src = (row < nt) ? S + (long long)(r0 + row) * ldh + r0 : y + r0;
          sv[u][r] = src[min(col, nt - 1)];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f64x4 c = {0.0, 0.0, 0.0, 0.0};
        const double *pa = LpT + 16 * ti4[u] + l15, *pb = LpT + 16 * tj4[u] + l15;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int k = 4 * kk + q4;
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k * RS], pb[k * RS], c, 0, 0, 0);
        }
        const int col = 16 * tj4[u] + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ti4[u] + q4 + 4 * r;
          if (tb + u < ntt && col < nt && ((row < nt && col <= row) || row == nt)) {
            double *dst = (row < nt) ? S + (long long)(r0 + row) * ldh + r0 : y + r0;
            const double nv = sv[u][r] - c[r];
            dst[col] = nv;
            if (row < 32 && row < nt) Lb[row * 34 + col] = nv;   // tiles (0,0), (1,0), (1,1): the next diagonal block
          }
        }
      }
      if (first_trip) {
        first_trip = false;
        __builtin_amdgcn_s_waitcnt(0xc07f);   // this wave's Lb writes
        diag_block(r0);
      }
    }
    __syncthreads();
    CTV_STAMP();
  }
  // ---- block back-substitution L^T x = y with the stored block inverses: x_b = Linv_b^T t_b, then t_j -= L[b][j]^T x_b
  //      for the rows above.  x lives in LDS; per block the loads of Linv_b (wave 0) and of the panel rows (everyone) do
  //      not depend on x and are issued together, before the block solve.
  double *xs = LpT;   // the panel is no longer needed
  for (int i = tid; i < P; i += NT) xs[i] = y[i];
  __syncthreads();
  const int nblk = (P + 31) / 32;
  for (int b = nblk - 1; b >= 0; --b) {
    const int jb = 32 * b, nb = min(32, P - jb);
    double lv[32];   // column j of the panel rows of this block: L[jb + ii][j]
    const bool upd = tid < jb;
    {
      const int j = min(tid, max(jb - 1, 0));
#pragma unroll
      for (int ii = 0; ii < 32; ++ii) lv[ii] = S[(long long)(jb + min(ii, nb - 1)) * ldh + j];
    }
    if (wave == 0) {
      const double *gi = d.chol_inv + ((size_t)w * d.chol_nblk + b) * 1024;
      double li[32];
      const int l31 = lane & 31;
#pragma unroll
      for (int i = 0; i < 32; ++i) li[i] = gi[i * 32 + l31];
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < 32; ++i) acc += li[i] * ((i < nb) ? xs[jb + i] : 0.0);   // Linv is lower triangular: rows i >= lane
      __builtin_amdgcn_wave_barrier();
      if (lane < nb) { xs[jb + lane] = acc; yb[lane] = acc; }
    }
    __syncthreads();
    if (upd) {
      double sacc = 0.0;
#pragma unroll
      for (int ii = 0; ii < 32; ++ii) sacc += lv[ii] * ((ii < nb) ? yb[ii] : 0.0);
      xs[tid] -= sacc;
    }
    for (int j = tid + NT; j < jb; j += NT) {   // P > NT + 32: remaining rows
      double sacc = 0.0;
      for (int ii = 0; ii < nb; ++ii) sacc += S[(long long)(jb + ii) * ldh + j] * yb[ii];
      xs[j] -= sacc;
    }
    __syncthreads();
  }
  for (int i = tid; i < P; i += NT) x[i] = xs[i];
  CTV_STAMP();
  if (tid == 0) lm.chol_fail = s_fail;
#undef CTV_STAMP
}

// ---- Register-resident tile Cholesky (windows with P <= 223): the whole lower triangle of the reduced system lives in the
// VGPRs of ONE workgroup as 16 x 16 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64 (tile t = i (i + 1) / 2 + j, i >= j,
// belongs to wave t % NW, slot t / NW: 105 tiles at P = 211 -> 7 slots x 4 registers per lane on 16 waves).  S is read from HBM
// exactly once and never written back; the right-hand side rides along as row P (so y = L^-1 b falls out of the factorisation).
// Per 16-column panel k:
//   A. the owner of the diagonal tile moves it through LDS into row-per-lane form and factors it with 16 v_readlane pivots
//      (lanes 0-15 the rows, lanes 16-31 the columns of the inverse: the fused scheme of k_cholesky_solve), leaves L_kk^-1 in LDS;
//   C. the owners of the tiles below it form L_ik = A_ik L_kk^-T (4 MFMAs; the accumulator -> operand transposition goes through
//      the tile's slice of the LDS panel) and publish L_ik there;
//   E. every owner of a trailing tile (i, j), j > k, subtracts L_ik L_jk^T (4 MFMAs, operands from the LDS panel).
// Back-substitution L^T x = y runs over the tiles still in registers: x_b = L_bb^-T t_b, then t_j -= L_bj^T x_b by the single
// owner of tile (b, j) -- no atomics anywhere, the summation order is fixed (bitwise reproducible).
// Pivots with index >= P (the rhs row, padding rows) are forced to 1 and never flagged.
template <int J, int C> __device__ __forceinline__ void chol16_row_updates(double (&v)[16], int lo, int hi) {
  if constexpr (C + 3 <= 15) {
    chol_bcast_update4<C>(v[C], v[C + 1], v[C + 2], v[C + 3], v[J], lo, hi);
    chol16_row_updates<J, C + 4>(v, lo, hi);
  } else if constexpr (C <= 15) {
    chol_bcast_update<C>(v[C], v[J], lo, hi);
    chol16_row_updates<J, C + 1>(v, lo, hi);
  }
}
template <int J> __device__ __forceinline__ void chol16_from(double (&v)[16], double di, int nreal, int &bad) {
  v[J] *= di;
  const int lo = __double2loint(v[J]), hi = __double2hiint(v[J]);
  if constexpr (J < 15) {
    chol_bcast_update_first<J + 1>(v[J + 1], v[J], lo, hi);
    double di_next = 1.0;
    if (J + 1 < nreal) di_next = chol_pivot_rsqrt(readlane_d(v[J + 1], J + 1), bad);   // (uniform branch; without it -- the 16 pivots as one
    // basic block, so that the scheduler may put the row updates of pivot J into the bubbles of pivot J + 1's rsq / Newton chain -- the
    // diagonal tile took 8.2 k cycles instead of 7.6 k: measured, not kept)
    if constexpr (J < 14) chol16_row_updates<J, J + 2>(v, lo, hi);
    chol16_from<J + 1>(v, di_next, nreal, bad);
  }
}
__device__ __forceinline__ double f64x4_get(const f64x4 &a, int r) { return r == 0 ? a[0] : (r == 1 ? a[1] : (r == 2 ? a[2] : a[3])); }

// (A BLOCKED diagonal tile -- the 16 pivots in four blocks of four, at most three broadcast-and-FMA per pivot inside a block and the block's
//  rank-4 update of the later columns, for the tile and for the inverse in the making, as two v_mfma_f64_16x16x4 -- was built and measured in
//  round 4: 8.7 k cycles per tile against 8.3 k, the factorisation unchanged at 84 us.  The tile's time is not its row updates but the
//  sixteen sequential pivots: readlane -> class test -> v_rsq_f64 -> two Newton steps -> scale -> readlane is ~300 dependent cycles each,
//  211 of them per factorisation, whatever happens between them.  Not kept.)
template <class T, int NW, int NS> __global__ __launch_bounds__(64 * NW) void k_cholesky_tiles(Dev<T> d) {
  constexpr int NT = 64 * NW, TS = 16 * 17;    // a 16 x 16 block in LDS: row stride 17
  const int w = blockIdx.x;
  Lm &lm = d.lm[w];
  if (lm.status || lm.ls_active) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, ldh = m.ldh, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int q4 = lane >> 4, l15 = lane & 15;
  const int NTR = P / 16 + 1, ntiles = NTR * (NTR + 1) / 2, ip = P / 16, rp = P % 16;   // the rhs row P sits in tile row ip, local row rp
  extern __shared__ __attribute__((aligned(16))) double smt[];
  double *Li = smt;                    // [NTR][TS] inverses of the diagonal blocks, Li[b][j * 17 + k] = Linv_b[j][k]
  double *Pn = Li + NTR * TS;          // [NTR][TS] panel: Pn[i][m * 17 + c] = L_ik[m][c] of the current panel
  double *tv = Pn + NTR * TS;          // [16 NTR] y, then the running right-hand side of the back-substitution
  double *xs = tv + 16 * NTR;          // [16 NTR] solution
  int &s_fail = *reinterpret_cast<int *>(xs + 16 * NTR);
  const double *S = d.S + m.H0, *y = d.rhs + m.p0;
  if (tid == 0) s_fail = 0;
  for (int i = tid; i < 16 * NTR; i += NT) tv[i] = 0.0;
  // ---- this wave's tiles (SGPRs) and their contents
  int ti[NS], tj[NS];
  f64x4 acc[NS];
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    const int t = wave + NW * q;
    int a, b;
    tile_decode(min(t, ntiles - 1), a, b);
    ti[q] = __builtin_amdgcn_readfirstlane(t < ntiles ? a : -1);
    tj[q] = __builtin_amdgcn_readfirstlane(t < ntiles ? b : 1 << 20);   // (never equal to a panel, never a trailing tile: ti < tj)
    // unconditional loads on clamped addresses straight into the tile registers; fixed up below
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rc = min(16 * a + q4 + 4 * r, P - 1);
      acc[q][r] = S[(long long)rc * ldh + min(16 * b + l15, rc)];
    }
  }
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    if (ti[q] < 0) continue;
    const int col = 16 * tj[q] + l15;
    if (ti[q] == tj[q]) {             // diagonal tile: the upper half is not stored in S
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[q][r] = (col <= 16 * ti[q] + q4 + 4 * r) ? acc[q][r] : 0.0;
    }
    if (ti[q] == ip) {                // tile row of the rhs row P; identity beyond it
      const double yv = y[min(col, P - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ip + q4 + 4 * r;
        acc[q][r] = row < P ? acc[q][r] : (row == P ? (col < P ? yv : 0.0) : (row == col ? 1.0 : 0.0));
      }
    }
  }
  __syncthreads();
  long long *dbg = (d.dbg && w == 0) ? d.dbg : nullptr;   // CTVIO_DEBUG_STAMPS: clock64 of thread 0 at the phase boundaries
  int dbi = 0;
#define CTV_STAMP() do { if (dbg && tid == 0 && dbi < 30) dbg[dbi++] = clock64(); } while (0)
  CTV_STAMP();
  for (int k = 0; k < NTR; ++k) {
    // ---- A. diagonal tile (k, k)
    const int td = k * (k + 1) / 2 + k, od = td % NW, sd = td / NW;
    if (wave == od) {
      double *Dg = Li + k * TS;
#pragma unroll
      for (int q = 0; q < NS; ++q)
        if (q == sd) {
#pragma unroll
          for (int r = 0; r < 4; ++r) Dg[(q4 + 4 * r) * 17 + l15] = acc[q][r];
        }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      double v[16];
      int opaque0;   // a zero the compiler cannot see through: without it the 16 identity columns below are hoisted out of the panel
      asm volatile("s_mov_b32 %0, 0" : "=s"(opaque0));   // loop as loop invariants and, for lack of registers, kept in scratch
      const int lz = l15 + opaque0;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double a = Dg[l15 * 17 + c];
        v[c] = lane < 16 ? (c <= lz ? a : 0.0) : (c == lz ? 1.0 : 0.0);
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();      // every lane has read its row before the block is overwritten with the inverse
      const int nreal = P - 16 * k;         // pivots below this are real; the rhs row and the padding rows are not factored
      int bad = 0;
      double di0 = 1.0;
      if (nreal > 0) di0 = chol_pivot_rsqrt(readlane_d(v[0], 0), bad);
      chol16_from<0>(v, di0, nreal, bad);
      if (lane >= 16 && lane < 32) {
#pragma unroll
        for (int i = 0; i < 16; ++i) Dg[i * 17 + l15] = v[i];   // Linv[i][column l15]
      }
      if (k == ip && lane == rp) {          // the part of y inside the last diagonal tile: L[P][16 ip + c], c < rp
#pragma unroll
        for (int c = 0; c < 16; ++c) if (c < rp) tv[16 * ip + c] = v[c];
      }
      if (lane == 0 && bad) s_fail = 1;
    }
    if (k < 4) CTV_STAMP();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    if (k < 4) CTV_STAMP();
    // ---- C. L_ik = A_ik L_kk^-T for the tiles below the diagonal one
    const double *Lk = Li + k * TS;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (tj[q] != k || ti[q] <= k) continue;   // (uniform)
      double *blk = Pn + ti[q] * TS;
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[(q4 + 4 * r) * 17 + l15] = acc[q][r];
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      double a[4], b[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) { a[s4] = blk[l15 * 17 + 4 * s4 + q4]; b[s4] = Lk[l15 * 17 + 4 * s4 + q4]; }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();          // operands are in registers before the slice is overwritten
      f64x4 c = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], c, 0, 0, 0);
      acc[q] = c;
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[(q4 + 4 * r) * 17 + l15] = c[r];
      if (ti[q] == ip && q4 == (rp & 3)) tv[16 * k + l15] = f64x4_get(c, rp >> 2);   // y: row P of L
    }
    if (k < 4) CTV_STAMP();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    if (k < 4) CTV_STAMP();
    // ---- E. trailing tiles (i, j), j > k: A_ij -= L_ik L_jk^T
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (ti[q] < 0 || tj[q] <= k || tj[q] >= (1 << 20)) continue;   // (uniform)
      const double *pa = Pn + ti[q] * TS + l15 * 17 + q4, *pb = Pn + tj[q] * TS + l15 * 17 + q4;
      double a[4], b[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) { a[s4] = -pa[4 * s4]; b[s4] = pb[4 * s4]; }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], acc[q], 0, 0, 0);
    }
    // (K-step outermost, so that consecutive MFMAs go to different tiles, was measured slower: 5.7 k cycles for the first panel's
    //  updates either way, and the diagonal tiles waited longer.  LOOK-AHEAD -- the owner of tile (k + 1, k + 1) updates that tile
    //  first and factors it while the other waves do their updates, the LDS panel double buffered, its own remaining updates
    //  deferred to the next panel -- was built and measured slower too: 1.63 vs 1.30 ms per 16 single-window factorisations,
    //  13.5 vs 10.6 ms per 2048-window solve.  The pivot chain takes ~12 k cycles instead of ~7 k when the other 15 waves are
    //  busy on the same SIMDs / LDS, s_setprio 3 does not change that, and step C grows by the pending updates.)
    // (the next panel's step C overwrites the LDS panel only after the barrier that follows its step A)
    if (k < 4) CTV_STAMP();
  }
  CTV_STAMP();
  __syncthreads();
  CTV_STAMP();
  // ---- back-substitution L^T x = y over the tiles in registers
  for (int b = NTR - 1; b >= 0; --b) {
    if (wave == (b % NW)) {   // x_b[j] = sum_k Linv[k][j] t[k]: lane (q4, j = l15) sums k = 4 q4 .. 4 q4 + 3, two shuffles add the quarters
      const double *Lb = Li + b * TS;
      double xa = 0.0;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) xa += Lb[(4 * q4 + kk) * 17 + l15] * tv[16 * b + 4 * q4 + kk];
      xa += __shfl_xor(xa, 16);
      xa += __shfl_xor(xa, 32);
      if (q4 == 0) xs[16 * b + l15] = xa;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      if (ti[q] != b || tj[q] >= b) continue;   // tiles (b, j), j < b: t_j -= L_bj^T x_b
      double part = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) part += acc[q][r] * xs[16 * b + q4 + 4 * r];
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      if (q4 == 0) tv[16 * tj[q] + l15] -= part;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
  }
  CTV_STAMP();
  double *x = d.delta + m.u0;
  for (int i = tid; i < P; i += NT) x[i] = xs[i];
  if (tid == 0) lm.chol_fail = s_fail;
#undef CTV_STAMP
}

// (Fusing this kernel into k_cholesky_tiles -- same workgroup, the pose step straight from LDS -- was built and measured: no gain for
//  one window (3.09 vs 3.05 ms per solve) and slower for 2048 (15.4 vs 13.6 ms for the two phases): the fused kernel spills, and
//  the landmark back-substitution wants more workgroups per CU than the tile kernel's registers allow.  Kept apart.)
// delta_l = dinv_l (-g_l - W_l . delta_p), one wave per landmark (coalesced over the row of W);
// model_cost_change = 1/2 delta^T (D^2 delta - g)  (equals Ceres' -(J y)^T (r + J y / 2) when
// (H + D^2) delta = -g);  then ComputeTrustRegionStep validity / HandleInvalidStep.
// Then, in the same workgroup (one per window): the candidate x (+) alpha delta of this pass (Plus: q <- q exp(d),
// ceres_local_param.h:137-145; additive elsewhere; the line delay projected on its box, trajectory_estimator.cpp:316-317), |step|^2 and
// |x|^2 of the reduced program (fixed-order block reductions, no atomics) and the knot-pair constants of the candidate for the
// linearisation that follows.  Windows inside the line search skip the solve part: their step is the same, only alpha changed.
template <class T, int NWV> __global__ __launch_bounds__(64 * NWV) void k_step_finish(Dev<T> d) {
  constexpr int NT = 64 * NWV;
  const int w = blockIdx.x;
  // the pass's "windows that start another pass" counter (k_pass_end adds to it, several launches later): cleared here instead of by a
  // memset node of its own
  if (w == 0 && threadIdx.x == 0) *d.n_active = 0;
  Lm &lm = d.lm[w];
  if (lm.status) return;
  const WinMeta &m = d.wins[w];
  const int P = m.P, L = m.L, N = m.N, u0 = m.u0, lm0 = m.lm0, ldw = m.ldw;
  extern __shared__ __attribute__((aligned(16))) double xs[];   // [P] pose step
  __shared__ double red[NWV], red_gd[NWV], red_dm[NWV];
  __shared__ int bad, s_go;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (lm.ls_active) {
    if (tid == 0) s_go = lm.step_valid;
  } else {
  double *x = d.delta + u0;
  const double *g = d.gS[d.lm[w].cur] + u0, *dd = d.dd + u0;
  const T *Wp = d.WS[d.lm[w].cur] + m.W0;
  for (int i = tid; i < P; i += NT) xs[i] = x[i];
  if (tid == 0) bad = 0;
  __syncthreads();
  // delta_rho_l = -(g_l + W_l . delta_p) / (Hll_l + D_l): a wave takes 8 rows of W per pass, 32 loads per lane in flight
  for (int l0 = 8 * wave; l0 < L; l0 += 8 * NWV) {
    double acc8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc8[u] = 0.0;
    // the row this lane will finish (see the reduction below): its g, 1/(Hll + D) and active flag travel with the W loads
    const int lrow = min(l0 + (lane >> 3), L - 1);
    const double g_l = g[P + lrow], dinv_l = d.dinv[lm0 + lrow];
    const bool act_l = d.active[u0 + P + lrow] != 0;
    // W is non-zero in the knot columns [0, 6K) and the line-delay column P - 1 only: NCB compact columns
    const int K6 = 6 * m.K, NCB = K6 + 1;
    for (int i0 = 0; i0 < NCB; i0 += 256) {
      T wv[8][4];
      int col[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int cc = min(i0 + lane + 64 * k, NCB - 1); col[k] = cc < K6 ? cc : P - 1; }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // clamped, unconditional loads (a predicated load compiles to branch + load + s_waitcnt: one round trip EACH);
          // out-of-range columns are masked through xi below, out-of-range rows are never written
          const int l = min(l0 + u, L - 1);
          wv[u][k] = Wp[(long long)l * ldw + col[k]];
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double xi = (i0 + lane + 64 * k < NCB) ? xs[col[k]] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc8[u] += (double)wv[u][k] * xi;
      }
    }
    // 8 row sums over 64 lanes with 10 shuffles: each butterfly step halves the rows a lane carries (bit 5 of the lane
    // picks rows 0-3 / 4-7, bit 4 the pair, bit 3 the row), then three plain steps; row u = lane >> 3 ends up in lane 8u
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    double v4[4], v2[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) v4[q] = (b5 ? acc8[4 + q] : acc8[q]) + __shfl_xor(b5 ? acc8[q] : acc8[4 + q], 32);
#pragma unroll
    for (int q = 0; q < 2; ++q) v2[q] = (b4 ? v4[2 + q] : v4[q]) + __shfl_xor(b4 ? v4[q] : v4[2 + q], 16);
    double v1 = (b3 ? v2[1] : v2[0]) + __shfl_xor(b3 ? v2[0] : v2[1], 8);
    v1 += __shfl_xor(v1, 4);
    v1 += __shfl_xor(v1, 2);
    v1 += __shfl_xor(v1, 1);
    if ((lane & 7) == 0 && l0 + (lane >> 3) < L) x[P + l0 + (lane >> 3)] = act_l ? (-g_l - v1) * dinv_l : 0.0;
  }
  __syncthreads();
  double mc = 0.0, gd = 0.0, dm = 0.0;   // model change; g . delta and |delta|_inf for the projected line search
  for (int j = tid; j < N; j += NT) {
    const double dj = x[j];
    if (!isfinite(dj)) bad = 1;
    if (d.active[u0 + j]) { mc += 0.5 * dj * (dd[j] * dj - g[j]); gd += g[j] * dj; dm = fmax(dm, fabs(dj)); }
  }
  for (int off = 32; off > 0; off >>= 1) { mc += __shfl_down(mc, off); gd += __shfl_down(gd, off); dm = fmax(dm, __shfl_down(dm, off)); }
  if (lane == 0) { red[wave] = mc; red_gd[wave] = gd; red_dm[wave] = dm; }
  __syncthreads();
  if (tid == 0) {
    double mc_t = 0.0, gd_t = 0.0, dm_t = 0.0;
    for (int q = 0; q < NWV; ++q) { mc_t += red[q]; gd_t += red_gd[q]; dm_t = fmax(dm_t, red_dm[q]); }   // fixed order
    lm.model_change = mc_t;
    lm.ls_gd0 = gd_t;
    lm.ls_dmax = dm_t;
    const bool valid = !lm.chol_fail && !bad && (mc_t > 0.0);
    if (valid) { lm.step_valid = 1; lm.invalid = 0; }
    else {
      lm.step_valid = 0;
      if (++lm.invalid >= d.prm.max_invalid) lm.status = 1 + 5;
      else { lm.mu /= lm.nu; lm.nu *= 2.0; lm.last_ok = 0; lm.nunsucc += 1; }
    }
    s_go = valid ? 1 : 0;
  }
  }   // (solve part)
  __syncthreads();
  if (!s_go) return;
  // ---- candidate = Plus(x, alpha delta)
  {
    const double al = lm.alpha;   // 1, or the trial step size of the projected line search
    const double *dl = d.delta + u0;
    const uint8_t *act = d.active + u0;
    double step2 = 0.0, x2 = 0.0;
    const int nst = m.K + m.F + L + 1;
    for (int t = tid; t < nst; t += NT) {
      if (t < m.K) {
        const int gk = m.knot0 + t;
        const bool ar = act[6 * t] != 0, ap = act[6 * t + 3] != 0;
        const Q4<double> q0 = qmk<double>(d.quat[4 * gk], d.quat[4 * gk + 1], d.quat[4 * gk + 2], d.quat[4 * gk + 3]);
        Q4<double> q1 = q0;
        if (ar) q1 = qmul(q0, so3_exp(mk<double>(al * dl[6 * t], al * dl[6 * t + 1], al * dl[6 * t + 2])));
        d.cquat[4 * gk] = q1.x; d.cquat[4 * gk + 1] = q1.y; d.cquat[4 * gk + 2] = q1.z; d.cquat[4 * gk + 3] = q1.w;
        if (ar) {
          step2 += (q1.x - q0.x) * (q1.x - q0.x) + (q1.y - q0.y) * (q1.y - q0.y) + (q1.z - q0.z) * (q1.z - q0.z) + (q1.w - q0.w) * (q1.w - q0.w);
          x2 += q1.x * q1.x + q1.y * q1.y + q1.z * q1.z + q1.w * q1.w;
        }
        for (int c = 0; c < 3; ++c) {
          const double p0 = d.pos[3 * gk + c], p1 = ap ? p0 + al * dl[6 * t + 3 + c] : p0;
          d.cpos[3 * gk + c] = p1;
          if (ap) { step2 += (p1 - p0) * (p1 - p0); x2 += p1 * p1; }
        }
      } else if (t < m.K + m.F) {
        const int f = t - m.K, gf = m.bias0 + f, u = 6 * m.K + 6 * f;
        for (int c = 0; c < 6; ++c) {
          const bool a = act[u + c] != 0;
          const double b0 = d.bias[6 * gf + c], b1 = a ? b0 + al * dl[u + c] : b0;
          d.cbias[6 * gf + c] = b1;
          if (a) { step2 += (b1 - b0) * (b1 - b0); x2 += b1 * b1; }
        }
      } else if (t < m.K + m.F + L) {
        const int l = t - m.K - m.F;
        const bool a = act[P + l] != 0;
        const double r0 = d.rho[lm0 + l], r1 = a ? r0 + al * dl[P + l] : r0;
        d.crho[lm0 + l] = r1;
        if (a) { step2 += (r1 - r0) * (r1 - r0); x2 += r1 * r1; }
      } else {
        const bool a = act[P - 1] != 0;
        const double l0 = d.ld[w];
        double l1 = a ? l0 + al * dl[P - 1] : l0;
        if (a && !m.fix_ld) l1 = fmin(fmax(l1, m.ld_lo), m.ld_hi);
        d.cld[w] = l1;
        if (a) { step2 += (l1 - l0) * (l1 - l0); x2 += l1 * l1; }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { step2 += __shfl_xor(step2, off); x2 += __shfl_xor(x2, off); }
    __syncthreads();   // (red / red_gd of the solve part have been consumed)
    if (lane == 0) { red[wave] = step2; red_gd[wave] = x2; }
    __syncthreads();   // also: the candidate knots are visible to the whole workgroup
    if (tid == 0) {
      double s2 = 0.0, x2t = 0.0;
      for (int q = 0; q < NWV; ++q) { s2 += red[q]; x2t += red_gd[q]; }   // fixed order
      lm.step2 = s2;
      lm.cand_xnorm2 = x2t;
    }
  }
  // ---- knot-pair constants of the candidate (shared by all residual blocks of the linearisation that follows)
  for (int t = tid; t < m.K - 1; t += NT) {
    const int gk = m.knot0 + t;
    knot_pair_const<T>(d.cquat + 4 * gk, d.cquat + 4 * gk + 4, d.lkd + 3 * gk, d.kjri + 9 * gk);
  }
}

// ------------------------------------------------------------------------------------------------ update
// 4-DoF gauge restore after a solve (reference TrajectoryManager::double2vector, trajectory_manager.cpp:485-516): one rigid
// transform puts the yaw and the position of knot `knot[w]` back to their pre-solve values (q0, t0) and is applied to
// knots knot..K-1.  One workgroup per requested window; all fp64.  Utility::R2ypr / ypr2R: visual_odometry/utility.h:74-113.
template <class T> __global__ void k_gauge_restore(Dev<T> d, int n, const int32_t *ids, const int32_t *knot, const double *q0, const double *t0) {
  const int e = blockIdx.x;
  if (e >= n) return;
  const WinMeta &m = d.wins[ids[e]];
  const int K = m.K, k0 = knot[e], base = m.knot0;
  __shared__ double sh[16];   // Rd (9), td (3), qd (4)
  if (threadIdx.x == 0) {
    const double *qr = d.quat + 4 * (base + k0), *pr = d.pos + 3 * (base + k0);
    const M3<double> R0 = q2R(qmk<double>(q0[4 * e], q0[4 * e + 1], q0[4 * e + 2], q0[4 * e + 3]));
    const M3<double> R00 = q2R(qmk<double>(qr[0], qr[1], qr[2], qr[3]));
    auto ypr = [](const M3<double> &R, double &y, double &p) {   // degrees
      y = atan2(R.m[3], R.m[0]);
      p = atan2(-R.m[6], R.m[0] * cos(y) + R.m[3] * sin(y)) / 3.14159265358979323846 * 180.0;
      y = y / 3.14159265358979323846 * 180.0;
    };
    double y0, p0, y00, p00;
    ypr(R0, y0, p0);
    ypr(R00, y00, p00);
    M3<double> Rd;
    if (fabs(fabs(p0) - 90.0) < 1.0 || fabs(fabs(p00) - 90.0) < 1.0) {   // Euler singularity: R0 R00^T
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rd.m[3 * i + j] = R0.m[3 * i] * R00.m[3 * j] + R0.m[3 * i + 1] * R00.m[3 * j + 1] + R0.m[3 * i + 2] * R00.m[3 * j + 2];
    } else {
      const double y = (y0 - y00) / 180.0 * 3.14159265358979323846;
      Rd = m3_id<double>();
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
template <class T> __global__ __launch_bounds__(256) void k_residual_summary(Dev<T> d, int w, double *out) {
  const WinMeta &m = d.wins[w];
  extern __shared__ __attribute__((aligned(16))) double smr[];   // [14 + pn] sums, then [pn] dx
  const int tid = threadIdx.x, n = m.pn;
  double *sums = smr, *dx = smr + 14 + n;
  for (int i = tid; i < 14 + 2 * n; i += 256) smr[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < m.M; i += 256) {
    const int idx = m.imu0 + i;
    const ImuGroup grp = d.groups[d.imu_grp[idx]];
    Knots4<double> k;
    LocalFrame<double> lf;
    lf.init(d.quat, d.pos, m.knot0 + grp.s);
    lf.load(d.quat, d.pos, m.knot0 + grp.s, k);
    SegConst<double> sc;
    seg_const(k, sc, false);
    double b[6], wgt[6], gy[3], ac[3], r[6];
    const double *bp = d.bias + 6 * (m.bias0 + grp.bias);
    for (int c = 0; c < 6; ++c) { b[c] = bp[c]; wgt[c] = m.imu_w[c]; }
    for (int c = 0; c < 3; ++c) { gy[c] = (double)d.imu_meas[(size_t)c * d.Mtot + idx]; ac[c] = (double)d.imu_meas[(size_t)(3 + c) * d.Mtot + idx]; }
    ImuJac<double> J;
    imu_eval_core<double>(k, sc, (double)d.imu_u[idx], m.inv_dt, lf.rotate(m.gravity), b, gy, ac, wgt, lf.RrefT(), r, false, J);
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
    Knots4<double> gi, gj;
    const double z3[3] = {0, 0, 0};
    load_knots<double>(d.quat, d.pos, m.knot0 + si, z3, gi);
    load_knots<double>(d.quat, d.pos, m.knot0 + sj, z3, gj);
    SegConst<double> sci, scj;
    seg_const(gi, sci, false);
    seg_const(gj, scj, false);
    const Q4<double> q_CI = qmk<double>(m.q_CI[0], m.q_CI[1], m.q_CI[2], m.q_CI[3]);
    const V3<double> p_CI = mk<double>(m.p_CI[0], m.p_CI[1], m.p_CI[2]);
    const M3<double> R = q2R(q_CI);
    M3<double> RCIT;
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
        const Q4<double> dq = qmul_raw(qmk<double>(-x0[0], -x0[1], -x0[2], x0[3]), qmk<double>(x[0], x[1], x[2], x[3]));
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
template <class T>
__global__ void k_spline_eval(Dev<T> d, int w, const int32_t *win_ids, int n, const long long *t_rel, double *pose7, double *vel3, double *omega3,
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
  Knots4<double> k;
  load_knots<double>(d.quat, d.pos, m.knot0 + s, zero3, k);
  SegConst<double> sc;
  seg_const(k, sc, false);
  const double idt = m.inv_dt;
  if (pose7) {
    double c[4];
    basis<double, false, 0>(u, 1.0, c);
    V3<double> p = mk<double>(0, 0, 0);
    for (int j = 0; j < 4; ++j) p = p + c[j] * k.p[j];
    Q4<double> q = eval_R(k.q, sc, u);
    if (ext.on) {   // Trajectory::GetSensorPose (trajectory.cpp:39-56): pose_S_to_G = pose_I_to_G * T_StoI
      p = p + qrot(q, mk<double>(ext.p[0], ext.p[1], ext.p[2]));
      q = qmul(q, qmk<double>(ext.q[0], ext.q[1], ext.q[2], ext.q[3]));
    }
    double *o = pose7 + 7 * (size_t)i;
    o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
  }
  if (vel3) {
    double c[4];
    basis<double, false, 1>(u, idt, c);
    V3<double> p = mk<double>(0, 0, 0);
    for (int j = 0; j < 4; ++j) p = p + c[j] * k.p[j];
    vel3[3 * (size_t)i] = p.x; vel3[3 * (size_t)i + 1] = p.y; vel3[3 * (size_t)i + 2] = p.z;
  }
  if (acc3) {
    double c[4];
    basis<double, false, 2>(u, idt * idt, c);
    V3<double> p = mk<double>(0, 0, 0);
    for (int j = 0; j < 4; ++j) p = p + c[j] * k.p[j];
    acc3[3 * (size_t)i] = p.x; acc3[3 * (size_t)i + 1] = p.y; acc3[3 * (size_t)i + 2] = p.z;
  }
  if (omega3) {
    const V3<double> o = eval_omega(sc, u, idt);
    omega3[3 * (size_t)i] = o.x; omega3[3 * (size_t)i + 1] = o.y; omega3[3 * (size_t)i + 2] = o.z;
  }
}

}  // namespace ctv
