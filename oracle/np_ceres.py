"""Second, independent restatement of the SOLVER LOOP the reference configures (Ceres 1.14: TRUST_REGION + LEVENBERG_MARQUARDT
with an exact dense linear solve, jacobi scaling, monotonic steps, and -- because a free line delay carries bounds
(reference src/estimator/trajectory_estimator.cpp:311-318, options at :371-398) -- the projected ARMIJO line search with CUBIC
interpolation), written in NumPy from Ceres' published algorithm (docs "Solving Non-linear Least Squares": TrustRegionMinimizer,
LevenbergMarquardtStrategy, Line Search Methods; defaults of Solver::Options).

TEST INFRASTRUCTURE (PARITY UNPINNED, see oracle/ctvo.h).  oracle/ctvo.c restates the same loop in C with analytic Jacobians, a
hand-written Cholesky / Schur elimination and a Durand-Kerner root finder; this file shares NOTHING with it: residuals come from
oracle/np_oracle.py (scipy rotations), Jacobians from central finite differences, the damped normal equations are solved with
numpy.linalg.solve on the FULL system (no Schur complement), interpolating polynomials with numpy.linalg.solve / numpy.roots.
tests/golden/make_golden.py records its per-iteration history as fixtures; tests assert that ctvo.c and the HIP path reproduce the
accept / reject sequence, the line-search step counts, the termination and (to finite-difference accuracy) costs and radii.
"""
from __future__ import annotations

import numpy as np

import np_oracle as npo

# Solver::Options defaults of Ceres 1.14 (the reference overrides only max_num_iterations, the linear solver and threads)
OPT = dict(initial_trust_region_radius=1e4, max_trust_region_radius=1e16, min_trust_region_radius=1e-32, min_relative_decrease=1e-3,
           min_lm_diagonal=1e-6, max_lm_diagonal=1e32, function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8,
           max_num_consecutive_invalid_steps=5,
           line_search_sufficient_function_decrease=1e-4, max_line_search_step_contraction=1e-3, min_line_search_step_contraction=0.6,
           max_num_line_search_step_size_iterations=20, min_line_search_step_size=1e-9)


class Problem:
    """The reduced program of one window: active unknowns (tangent space), the line-delay box, evaluation by np_oracle."""

    def __init__(self, w, active):
        self.w0 = w
        self.act = np.flatnonzero(active)
        self.P = w.P
        self.ld_free = bool(active[w.P - 1]) and not w.fix_ld
        self.k_ld = int(np.searchsorted(self.act, w.P - 1)) if self.ld_free else -1

    def plus(self, w, delta_act):
        """Plus(x, delta) of the reduced program followed by the projection on the feasible set (line delay box)."""
        xi = np.zeros(w.N)
        xi[self.act] = delta_act
        w2 = npo.retract(w, xi)
        if self.ld_free:
            w2.ld = float(min(max(w2.ld, w.ld_lo), w.ld_hi))
        return w2

    def cost(self, w):
        return npo.cost(w)

    def evaluate(self, w):
        """cost, corrected residuals r~ and corrected Jacobian J~ (Triggs corrector of the Cauchy loss per visual block) wrt the active
        unknowns, by central differences of the raw residuals."""
        r = npo.residuals(w)
        J = npo.fd_jacobian(w, list(self.act))
        rs = npo.stacked(w).copy()
        n_imu = r["imu"].size
        cost = 0.5 * np.sum(r["imu"] ** 2) + 0.5 * np.sum(r["bias"] ** 2) + 0.5 * np.sum(r["prior"] ** 2)
        vc = getattr(w, "v_cauchy", None)
        for v in range(w.V):
            a = float(vc[v]) if vc is not None else float(w.cauchy_a)
            sl = slice(n_imu + 2 * v, n_imu + 2 * v + 2)
            s = float(np.sum(rs[sl] ** 2))
            if a <= 0:
                cost += 0.5 * s
                continue
            b2 = a * a
            rho1 = 1.0 / (1.0 + s / b2)
            rho2 = -(1.0 / b2) * rho1 * rho1
            cost += 0.5 * b2 * np.log1p(s / b2)
            sq = np.sqrt(rho1)
            if s == 0.0 or rho2 <= 0.0:
                J[sl] *= sq
                rs[sl] *= sq
            else:
                D = 1 + 2 * s * rho2 / rho1
                al = 1 - np.sqrt(D)
                J[sl] = sq * (J[sl] - (al / s) * np.outer(rs[sl], rs[sl] @ J[sl]))
                rs[sl] *= sq / (1 - al)
        return float(cost), rs, J

    def x_ambient(self, w):
        """ambient coordinates of the active parameter blocks (quaternion 4, everything else as is) -- what Ceres' x_norm / step_norm see"""
        K, F, P = w.K, w.F, w.P
        out = []
        a = np.zeros(w.N, bool)
        a[self.act] = True
        for k in range(K):
            if a[6 * k]:
                out.append(w.quat[k])
            if a[6 * k + 3]:
                out.append(w.pos[k])
        for f in range(F):
            if a[6 * K + 6 * f]:
                out.append(w.bias[f, :3])
            if a[6 * K + 6 * f + 3]:
                out.append(w.bias[f, 3:])
        if a[P - 1]:
            out.append([w.ld])
        out.append(w.rho[a[P:]])
        return np.concatenate([np.ravel(o) for o in out])


def _interpolating_polynomial(samples):
    """Ceres FindInterpolatingPolynomial: lowest-degree polynomial through the given values / gradients (highest power first)."""
    n = sum(1 + (1 if s[2] is not None else 0) for s in samples)
    A, b = [], []
    for x, v, g in samples:
        A.append([x ** (n - 1 - j) for j in range(n)])
        b.append(v)
        if g is not None:
            A.append([(n - 1 - j) * x ** (n - 2 - j) if n - 1 - j > 0 else 0.0 for j in range(n)])
            b.append(g)
    return np.linalg.solve(np.array(A, float), np.array(b, float))


def _minimize_polynomial(p, x_min, x_max):
    """Ceres MinimizePolynomial: the minimum over [x_min, x_max] among the mid point, the end points and the real parts of the roots of p'."""
    best_x = 0.5 * (x_min + x_max)
    best = np.polyval(p, best_x)
    for x in (x_min, x_max):
        v = np.polyval(p, x)
        if v < best:
            best, best_x = v, x
    dp = np.polyder(p)
    dp = np.trim_zeros(dp, "f")
    if dp.size > 1:
        for r in np.roots(dp):
            x = float(np.real(r))
            if x < x_min or x > x_max:
                continue
            v = np.polyval(p, x)
            if v < best:
                best, best_x = v, x
    return best_x


def armijo_search(prob, w, delta, cost0, g_dot_delta, log=None):
    """ArmijoLineSearch::DoSearch on phi(alpha) = cost(Project(x (+) alpha delta)), CUBIC interpolation (value and gradient at every
    trial point).  Returns (success, alpha, iterations)."""
    o = OPT
    dmax = float(np.max(np.abs(delta))) if delta.size else 0.0

    def sample(alpha):
        wc = prob.plus(w, alpha * delta)
        c, r, J = prob.evaluate(wc)
        g = float((J.T @ r) @ delta)
        ok = bool(np.isfinite(c) and np.isfinite(g))
        return dict(x=alpha, v=c, g=g, ok=ok)

    prev = None
    cur = sample(1.0)
    it = 0
    while (not cur["ok"]) or cur["v"] > cost0 + o["line_search_sufficient_function_decrease"] * g_dot_delta * cur["x"]:
        it += 1
        if it >= o["max_num_line_search_step_size_iterations"]:
            return False, 1.0, it
        lo, hi = o["max_line_search_step_contraction"] * cur["x"], o["min_line_search_step_contraction"] * cur["x"]
        if not cur["ok"]:
            step = min(max(cur["x"] * 0.5, lo), hi)
        else:
            smp = [(0.0, cost0, g_dot_delta), (cur["x"], cur["v"], cur["g"])]
            if prev is not None and prev["ok"]:
                smp.append((prev["x"], prev["v"], prev["g"]))
            step = _minimize_polynomial(_interpolating_polynomial(smp), lo, hi)
        if step * dmax < o["min_line_search_step_size"]:
            return False, 1.0, it
        prev = cur
        cur = sample(step)
        if log is not None:
            log.append(step)
    return True, cur["x"], it


def solve(w0, active, max_iters=15, verbose=False):
    """TrustRegionMinimizer::Minimize.  Returns (final window, history dict)."""
    o = OPT
    prob = Problem(w0, active)
    w = w0.copy()
    if prob.ld_free:   # IterationZero: project on the feasible set
        w.ld = float(min(max(w.ld, w.ld_lo), w.ld_hi))
    cost, r, J = prob.evaluate(w)
    scale = 1.0 / (1.0 + np.sqrt(np.sum(J * J, 0)))          # jacobi_scaling, once
    g = J.T @ r

    def gmax(w, g):
        wn = prob.plus(w, -g)
        return float(np.max(np.abs(prob.x_ambient(w) - prob.x_ambient(wn))))
    gm = gmax(w, g)
    x_norm = float(np.linalg.norm(prob.x_ambient(w)))
    radius, dec = o["initial_trust_region_radius"], 2.0
    hist = dict(cost=[cost], accepted=[], radius=[radius], alpha=[], ls_iters=[], invalid=[])
    it, invalid, last_ok, term = 0, 0, True, "NO_CONVERGENCE"
    nsucc = nuns = nls = nred = 0
    while True:
        if it >= max_iters:
            term = "NO_CONVERGENCE"
            break
        if last_ok and gm <= o["gradient_tolerance"]:
            term = "CONVERGENCE_GRADIENT"
            break
        if radius <= o["min_trust_region_radius"]:
            term = "CONVERGENCE_RADIUS"
            break
        it += 1
        Js = J * scale[None, :]
        H = Js.T @ Js
        D = np.clip(np.diag(H), o["min_lm_diagonal"], o["max_lm_diagonal"]) / radius
        gs = Js.T @ r
        try:
            y = np.linalg.solve(H + np.diag(D), -gs)
            Jy = Js @ y
            model_change = float(-Jy @ (r + 0.5 * Jy))
            valid = bool(np.all(np.isfinite(y)) and model_change > 0.0)
        except np.linalg.LinAlgError:
            valid, model_change = False, 0.0
        if not valid:
            invalid += 1
            hist["invalid"].append(it)
            if invalid >= o["max_num_consecutive_invalid_steps"]:
                term = "FAILURE"
                hist["cost"].append(cost)
                break
            radius /= dec
            dec *= 2
            last_ok = False
            nuns += 1
            hist["cost"].append(cost); hist["accepted"].append(False); hist["radius"].append(radius); hist["alpha"].append(0.0); hist["ls_iters"].append(0)
            continue
        invalid = 0
        delta = y * scale
        alpha, ls_it = 1.0, 0
        if prob.ld_free:
            ok, a, ls_it = armijo_search(prob, w, delta, cost, float(g @ delta))
            nls += ls_it
            if ok:
                alpha = a
            if alpha != 1.0:
                nred += 1
                delta = delta * alpha
        wc = prob.plus(w, delta)
        cand = prob.cost(wc)
        step_norm = float(np.linalg.norm(prob.x_ambient(wc) - prob.x_ambient(w)))
        if step_norm <= o["parameter_tolerance"] * (x_norm + o["parameter_tolerance"]):
            term = "CONVERGENCE_PARAMETER"
            hist["cost"].append(cost)
            break
        if abs(cost - cand) <= o["function_tolerance"] * cost:
            term = "CONVERGENCE_FUNCTION"
            hist["cost"].append(cost)
            break
        quality = (cost - cand) / model_change
        if quality > o["min_relative_decrease"]:
            w = wc
            cost, r, J = prob.evaluate(w)
            g = J.T @ r
            gm = gmax(w, g)
            x_norm = float(np.linalg.norm(prob.x_ambient(w)))
            radius = min(o["max_trust_region_radius"], radius / max(1.0 / 3.0, 1.0 - (2.0 * quality - 1.0) ** 3))
            dec = 2.0
            last_ok = True
            nsucc += 1
            hist["accepted"].append(True)
        else:
            radius /= dec
            dec *= 2
            last_ok = False
            nuns += 1
            hist["accepted"].append(False)
        hist["cost"].append(cost); hist["radius"].append(radius); hist["alpha"].append(alpha); hist["ls_iters"].append(ls_it)
        if verbose:
            print(f"  it {it:2d} cost {cost:.9g} acc {hist['accepted'][-1]} radius {radius:.4g} alpha {alpha:.6g} ls {ls_it}")
    hist.update(iterations=it, termination=term, num_successful=nsucc, num_unsuccessful=nuns, num_line_search_steps=nls,
                num_line_search_reduced=nred, final_cost=cost, final_radius=radius)
    return w, hist
