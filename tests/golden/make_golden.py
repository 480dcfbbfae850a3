"""Generates the committed golden fixtures in tests/golden/*.npz.

Run ONLY in the build container:  python tests/golden/make_golden.py   (all the small fixtures), and
    python tests/golden/make_golden.py bench_fd | bench_lm                (the fixtures at the benchmarked sizes: minutes each)
Source of truth = oracle/np_oracle.py (independent NumPy/SciPy restatement: residuals by a
different code path, Jacobians by central finite differences, minimiser = scipy trf).  The
fixtures pin oracle/ctvo.c (tests/test_oracle_golden.py) and, through it, the HIP path.
The reference itself has no tests/golden vectors and cannot be run here (PARITY UNPINNED).
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
cv = importlib.import_module("ctrl-vio_amd")
import np_oracle as npo  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def corrected_normal(w):
    """Dense H, g, cost from FD Jacobians with the Ceres corrector applied per visual block."""
    N = w.N
    r = npo.residuals(w)
    J = npo.fd_jacobian(w, list(range(N)))           # raw stacked J (rows: imu, vis, bias, prior)
    n_imu, n_vis = r["imu"].size, r["vis"].size
    rs = npo.stacked(w).copy()
    cost = 0.5 * np.sum(r["imu"] ** 2) + 0.5 * np.sum(r["bias"] ** 2) + 0.5 * np.sum(r["prior"] ** 2)
    b2 = w.cauchy_a ** 2
    for v in range(w.V):
        sl = slice(n_imu + 2 * v, n_imu + 2 * v + 2)
        s = float(np.sum(rs[sl] ** 2))
        rho1 = 1.0 / (1.0 + s / b2)
        rho2 = -(1.0 / b2) * rho1 * rho1
        cost += 0.5 * b2 * np.log1p(s / b2)
        sq = np.sqrt(rho1)
        if s == 0.0 or rho2 <= 0.0:
            J[sl] *= sq; rs[sl] *= sq
        else:  # never taken for Cauchy; kept to mirror the corrector
            D = 1 + 2 * s * rho2 / rho1; al = 1 - np.sqrt(D)
            J[sl] = sq * (J[sl] - (al / s) * np.outer(rs[sl], rs[sl] @ J[sl])); rs[sl] *= sq / (1 - al)
    return J.T @ J, J.T @ rs, float(cost), J, n_imu, n_vis


LM_CASES = (("lm_tiny_seed7", "tiny", 7, {}),                                   # no step is shortened
            ("lm_tiny_rs_seed3020", "tiny", 3020, dict(img_h=640, ld_true=3.0e-5)),   # rolling-shutter stress: 4 steps shortened by the line search
            ("lm_tiny_rs_seed3028", "tiny", 3028, dict(img_h=640, ld_true=3.0e-5)),   # 12 shortened steps and one unsuccessful step
            ("lm_config1_seed1001", "config1", 1001, {}))


def lm_fixtures():
    """(d) per-iteration history of the INDEPENDENT NumPy restatement of Ceres 1.14's trust-region loop + projected Armijo line search
    (oracle/np_ceres.py: FD Jacobians, numpy.linalg / numpy.roots): accept / reject sequence, costs, radii, step sizes."""
    import np_ceres
    import pyctvo
    for name, cfg, seed, kw in LM_CASES:
        w0 = cv.synth.make_window(cfg, seed=seed, **kw)
        active = pyctvo.OracleWindow(w0.copy()).active_mask()   # structure only (which unknowns are referenced and not constant)
        wf, h = np_ceres.solve(w0, active, 15)
        d = w0.to_dict("w_")
        d.update(wf.to_dict("f_"))
        d.update(hist_cost=np.array(h["cost"]), hist_accepted=np.array(h["accepted"], np.int8), hist_radius=np.array(h["radius"]),
                 hist_alpha=np.array(h["alpha"]), hist_ls_iters=np.array(h["ls_iters"], np.int32), iterations=h["iterations"],
                 termination=h["termination"], num_successful=h["num_successful"], num_unsuccessful=h["num_unsuccessful"],
                 num_line_search_steps=h["num_line_search_steps"], num_line_search_reduced=h["num_line_search_reduced"],
                 final_cost=h["final_cost"], final_radius=h["final_radius"])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, {k: h[k] for k in ("iterations", "termination", "num_successful", "num_unsuccessful", "num_line_search_steps",
                                       "num_line_search_reduced", "final_cost", "final_radius")})


# ---- fixtures at the sizes that are BENCHMARKED (round-3 verdict: the independent fixtures stopped at tiny / config-1 size)
BENCH_FD = (("fd_config2_seed1000", "config2", 1000), ("fd_config3_seed1001", "config3", 1001))
BENCH_LM = (("lm_config2_seed1002", "config2", 1002, {}), ("lm_config3_seed1003", "config3", 1003, {}),
            ("lm_config3_seed1006", "config3", 1006, {}),     # seed 1006: six steps shortened by the line search and one unsuccessful step
            # round 5: a LONG window (16 frames, K = 34, P = 301) -- the shape the envelope panel Cholesky factors (its reduced system is far
            # from dense); the independent loop solves the full un-eliminated system with numpy.linalg.solve: no sparsity anywhere
            ("lm_long_k34_seed1400", "config1", 1400, dict(F=16, L=60, M=750)))


def bench_size_fd():
    """(e) dense normal equations at x0 of one config-2 (10 KF / 200 landmarks / 2000 IMU) and one config-3 (300 landmarks, rolling-shutter
    stress) window from finite-difference Jacobians of the independent NumPy restatement.  The landmark block of H is diagonal, so the
    fixture keeps Hpp (P x P), W (P x L), diag(Hll), g and the cost instead of the dense N x N matrix."""
    for name, cfg, seed in BENCH_FD:
        w = cv.synth.make_window(cfg, seed=seed)
        w.ld = 13000.5e-9          # half-ns margin: int64(ld*1e9) is unambiguous
        H, g, cost, J, n_imu, n_vis = corrected_normal(w)
        P = w.P
        off = H[P:, P:] - np.diag(np.diag(H[P:, P:]))
        assert np.abs(off).max() <= 1e-9 * np.abs(np.diag(H[P:, P:])).max(), "landmarks must not couple"
        d = w.to_dict("w_")
        d.update(Hpp=H[:P, :P], W=H[:P, P:], Hll=np.diag(H)[P:].copy(), g=g, cost=cost)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "N", w.N, "V", w.V, "cost", cost)


def bench_size_lm():
    """(f) LM histories of the independent loop restatement (np_ceres) at config-2 and config-3 size."""
    import np_ceres
    import pyctvo
    for name, cfg, seed, kw in BENCH_LM:
        if len(sys.argv) > 2 and name not in sys.argv[2:]:
            continue
        w0 = cv.synth.make_window(cfg, seed=seed, **kw)
        active = pyctvo.OracleWindow(w0.copy()).active_mask()
        wf, h = np_ceres.solve(w0, active, 15)
        d = w0.to_dict("w_")
        d.update(wf.to_dict("f_"))
        d.update(hist_cost=np.array(h["cost"]), hist_accepted=np.array(h["accepted"], np.int8), hist_radius=np.array(h["radius"]),
                 hist_alpha=np.array(h["alpha"]), hist_ls_iters=np.array(h["ls_iters"], np.int32), iterations=h["iterations"],
                 termination=h["termination"], num_successful=h["num_successful"], num_unsuccessful=h["num_unsuccessful"],
                 num_line_search_steps=h["num_line_search_steps"], num_line_search_reduced=h["num_line_search_reduced"],
                 final_cost=h["final_cost"], final_radius=h["final_radius"])
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, {k: h[k] for k in ("iterations", "termination", "num_successful", "num_unsuccessful", "num_line_search_steps",
                                       "num_line_search_reduced", "final_cost", "final_radius")})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "lm":
        return lm_fixtures()
    if len(sys.argv) > 1 and sys.argv[1] == "bench_fd":
        return bench_size_fd()
    if len(sys.argv) > 1 and sys.argv[1] == "bench_lm":
        return bench_size_lm()
    # ---- (a) tiny window: per-block residuals, FD Jacobians, dense H/g/cost at x0
    w = cv.synth.make_window("tiny", seed=7)
    w.ld = 13000.5e-9          # half-ns margin: int64(ld*1e9) is unambiguous
    H, g, cost, J, n_imu, n_vis = corrected_normal(w)
    r = npo.residuals(w)
    d = w.to_dict("w_")
    d.update(r_imu=r["imu"], r_vis=r["vis"], r_bias=r["bias"], r_prior=r["prior"], H=H, g=g, cost=cost, J_corrected=J)
    np.savez_compressed(os.path.join(OUT, "tiny_seed7.npz"), **d)
    print("tiny: N", w.N, "cost", cost)

    # ---- (b) edge variants of the tiny window: ld at both bounds, extreme rows
    for name, mod in (("ld_lo", dict(ld=0.0)), ("ld_hi", dict(ld=3.5e-5)), ("rows", dict(ld=20000.5e-9))):
        w2 = w.copy()
        w2.ld = mod["ld"]
        if name == "rows":
            w2.v_rowi[::3] = 0; w2.v_rowj[1::3] = 1023; w2.v_rowj[2::3] = 0
        r2 = npo.residuals(w2)
        cols = list(range(w2.N))
        J2 = npo.fd_jacobian(w2, cols)
        d2 = w2.to_dict("w_")
        d2.update(r_imu=r2["imu"], r_vis=r2["vis"], J_raw_vis=J2[r2["imu"].size:r2["imu"].size + r2["vis"].size],
                  cost=npo.cost(w2))
        np.savez_compressed(os.path.join(OUT, f"tiny_{name}.npz"), **d2)
        print(name, "cost", d2["cost"])

    # ---- (c) converged states from scipy (tiny + config1)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyctvo
    for cfg, seed in (("tiny", 7), ("config1", 1000)):
        w0 = cv.synth.make_window(cfg, seed=seed)
        active = pyctvo.OracleWindow(w0.copy()).active_mask()   # which unknowns exist (structure only)
        wf, res = solve_with_fd(w0, active)
        d3 = w0.to_dict("w_")
        d3.update(wf.to_dict("f_"))
        d3.update(final_cost=npo.cost(wf), nfev=res.nfev, status=res.status)
        np.savez_compressed(os.path.join(OUT, f"{cfg}_seed{seed}_converged.npz"), **d3)
        print(cfg, "scipy cost", d3["final_cost"], "status", res.status, res.message, "nfev", res.nfev)
    lm_fixtures()


def solve_with_fd(w0, active):
    """scipy trf with a central-difference Jacobian in the tangent space, re-anchored a few times."""
    from scipy.optimize import least_squares
    act = np.flatnonzero(active)
    P = w0.P
    w = w0
    for outer in range(4):
        scale = np.ones(w.N); scale[P - 1] = 1e-5

        def fun(z, w=w, scale=scale):
            xi = np.zeros(w.N); xi[act] = z * scale[act]
            return npo.stacked(npo.retract(w, xi), robust_sqrt=True)

        def jac(z, w=w, scale=scale):
            xi = np.zeros(w.N); xi[act] = z * scale[act]
            wz = npo.retract(w, xi)
            # d/dz of retract(w, xi(z)) ~ d/d(eta) retract(wz, eta) at eta=0 (exact for additive parts,
            # first order for rotations: steps within one outer round are small after re-anchoring)
            return npo.fd_jacobian(wz, list(act), robust_sqrt=True) * scale[act][None, :]

        lo = np.full(len(act), -np.inf); hi = np.full(len(act), np.inf)
        if active[P - 1] and not w.fix_ld:
            k = int(np.searchsorted(act, P - 1))
            lo[k], hi[k] = (w.ld_lo - w.ld) / 1e-5 - 1e-9, (w.ld_hi - w.ld) / 1e-5 + 1e-9
        res = least_squares(fun, np.zeros(len(act)), jac=jac, bounds=(lo, hi), method="trf", xtol=1e-15, ftol=1e-15,
                            gtol=1e-15, max_nfev=60)
        xi = np.zeros(w.N); xi[act] = res.x * scale[act]
        w = npo.retract(w, xi)
        print("   outer", outer, "cost", npo.cost(w), "nfev", res.nfev, "|step|", np.linalg.norm(res.x))
    return w, res


if __name__ == "__main__":
    main()
