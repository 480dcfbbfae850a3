// kernels_control.hpp -- LM control on the device: k_lm_init, Ceres' polynomial interpolation for the line search, the cost sum, lm_decide / k_pass_end
// (accept / reject / terminate / Armijo, set swap, next damping), k_zero_normal, k_knot_prep.
// Part of kernels.hpp (included from there, in order; not a stand-alone header).
#pragma once

namespace ctv {

// ------------------------------------------------------------------------------------------------ control
__global__ void k_lm_init(Dev d, double mu, int keep_scale) {
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
__device__ __forceinline__ double window_cost_sum(const Dev &d, const WinMeta &m, int w, int lane) {
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
__global__ __launch_bounds__(64) void k_initial_cost(Dev d, int as_candidate) {
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
__device__ inline int lm_decide(const Dev &d, Lm &lm, double cand_cost, double gd, bool have_grad) {
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
__global__ __launch_bounds__(256) void k_pass_end(Dev d) {
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
__global__ void k_zero_normal(Dev d, int single_part, int mode) {
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
__global__ void k_knot_prep(Dev d) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= d.Ktot) return;
  const int w = d.knot_win[g];
  const WinMeta &m = d.wins[w];
  if (g - m.knot0 >= m.K - 1) return;   // the last knot of a window starts no pair
  knot_pair_const(d.quat + 4 * g, d.quat + 4 * g + 4, d.lkd + 3 * g, d.kjri + 9 * g);
}

}  // namespace ctv
