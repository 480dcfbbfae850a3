"""GPU (-m gpu): the HIP path, called through the C ABI, against the fp64 oracle on the same seeded windows,
the committed golden fixtures, and size-independent properties at BASELINE.json's full sizes.

Tolerances.  The product is all-fp64, like the reference: every kernel in double -- it must reproduce the oracle's iterates (final
state <= 1e-6 asserted, ~1e-9 measured; the BASELINE contract is 1e-4).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scaled(H):
    sc = np.sqrt(np.maximum(np.diag(H), 1e-30))
    return sc


@pytest.fixture(scope="module")
def win_cfg1(cv):
    w = cv.synth.make_window("config1", seed=1000)
    w.ld = 1.1e-5
    return w


@pytest.mark.parametrize("prec,tol", [("fp64", 1e-10)])
def test_linearize_matches_oracle(cv, oracle, win_cfg1, prec, tol):
    w = win_cfg1.copy()
    H, g, cost = oracle.OracleWindow(w.copy()).build_normal()
    P = w.P
    sc = _scaled(H)
    with cv.Solver(precision=prec) as s:
        s.set_windows([w])
        Hg, Wg, Hllg, gg, costg = s.linearize(0)
    assert costg == pytest.approx(cost, rel=1e-12 if prec == "fp64" else 1e-6)
    assert np.abs((Hg - H[:P, :P]) / np.outer(sc[:P], sc[:P])).max() < tol
    assert np.abs((Wg - H[:P, P:]) / np.outer(sc[:P], sc[P:])).max() < tol
    assert np.abs(Hllg / np.diag(H)[P:] - 1).max() < tol
    assert np.abs((gg - g) / sc).max() < tol * np.abs(g / sc).max()


def test_several_anchors_per_landmark_and_large_rotation_visual(cv, oracle, win_cfg1):
    """The C ABI takes arbitrary blocks: a landmark whose blocks do NOT share the i end (different t_i / row_i / p_i) owns several
    anchors (host_pack.hpp finds the distinct ones, k_vis_anchor writes a record for each, the rows of W sum over them), and a window
    with knot-to-knot rotations above 0.5 rad takes the general (closed-form) bodies of both visual kernels.  Dense normal equations
    and a full solve against the oracle, which evaluates every block on its own."""
    rng = np.random.default_rng(17)
    w = win_cfg1.copy()
    # every third block gets its own i end: another row, a shifted observation, and for some an i time one frame later (kept before t_j)
    odd = np.arange(w.V) % 3 == 1
    w.v_rowi = np.where(odd, rng.integers(0, 1024, w.V), w.v_rowi).astype(w.v_rowi.dtype)
    w.v_pi = w.v_pi + odd[:, None] * rng.normal(0.0, 0.01, (w.V, 2))
    later = odd & (w.v_tj - w.v_ti >= 200_000_000) & (rng.random(w.V) < 0.5)
    w.v_ti = np.where(later, w.v_ti + 100_000_000, w.v_ti).astype(w.v_ti.dtype)
    w.normalize()
    big = w.copy()                                    # the same blocks on a spline with 0.7 rad between knots 5 / 6 and 11 / 12
    for k in (6, 12):
        c, sn = np.cos(0.35), np.sin(0.35)
        dq = np.array([sn * 0.6, sn * 0.0, sn * 0.8, c])
        for j in range(k, big.K):
            x1, y1, z1, w1 = big.quat[j]; x2, y2, z2, w2 = dq
            big.quat[j] = [w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
                           w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2]
    big.normalize()
    for win, tol in ((w, 1e-10), (big, 1e-9)):
        H, g, cost = oracle.OracleWindow(win.copy()).build_normal()
        P = win.P
        sc = _scaled(H)
        with cv.Solver() as s:
            s.set_windows([win.copy()])
            Hg, Wg, Hllg, gg, costg = s.linearize(0)
        assert costg == pytest.approx(cost, rel=1e-12)
        assert np.abs((Hg - H[:P, :P]) / np.outer(sc[:P], sc[:P])).max() < tol
        assert np.abs((Wg - H[:P, P:]) / np.outer(sc[:P], sc[P:])).max() < tol
        assert np.abs(Hllg / np.diag(H)[P:] - 1).max() < tol
        assert np.abs((gg - g) / sc).max() < tol * np.abs(g / sc).max()
    ref = w.copy()
    so = oracle.OracleWindow(ref).solve(15)
    with cv.Solver() as s:
        wg = w.copy()
        s.set_windows([wg])
        sm = s.solve(15)[0]
    assert sm["iterations"] == so.iterations and sm["final_cost"] == pytest.approx(so.final_cost, rel=1e-9)
    assert cv.rel_state_error(wg, ref)["state"] < 1e-6


@pytest.mark.parametrize("case", ["general_body", "anisotropic_accel", "large_rotation"])
def test_imu_linearize_general_body(cv, oracle, win_cfg1, case):
    """k_imu_linearize_f64 specialises the usual IMU group (knot-pair rotations < 0.5 rad, isotropic accelerometer weights); the others go
    through the general body (k_imu_linearize_rest).  All three ways into it against the oracle: forced (use_mfma = 2), accelerometer weights
    that differ per axis, and a window whose knots turn by ~0.7 rad from one to the next (beyond the small-angle series)."""
    w = win_cfg1.copy()
    mfma = 1
    if case == "general_body":
        mfma = 2
    elif case == "anisotropic_accel":
        w.imu_w = np.array([250.0, 250.0, 250.0, 12.5, 10.0, 15.0])
    else:
        from scipy.spatial.transform import Rotation as R
        for k in range(w.quat.shape[0]):   # q_k <- q_k * exp(k * 0.7 e_z): consecutive knots 0.7 rad further apart
            w.quat[k] = (R.from_quat(w.quat[k]) * R.from_rotvec([0.0, 0.0, 0.7 * k])).as_quat()
    H, g, cost = oracle.OracleWindow(w.copy()).build_normal()
    P = w.P
    sc = _scaled(H)
    with cv.Solver(use_mfma=mfma) as s:
        s.set_windows([w])
        Hg, Wg, Hllg, gg, costg = s.linearize(0)
    assert costg == pytest.approx(cost, rel=1e-12)
    assert np.abs((Hg - H[:P, :P]) / np.outer(sc[:P], sc[:P])).max() < 1e-10
    assert np.abs((Wg - H[:P, P:]) / np.outer(sc[:P], sc[P:])).max() < 1e-10      # (large_rotation: the visual blocks' general form too)
    assert np.abs(Hllg / np.diag(H)[P:] - 1).max() < 1e-10
    assert np.abs((gg - g) / sc).max() < 1e-10 * np.abs(g / sc).max()


@pytest.mark.parametrize("prec,tol", [("fp64", 1e-8)])
@pytest.mark.parametrize("mfma", [1, 2])
def test_lm_step_matches_oracle(cv, oracle, win_cfg1, prec, tol, mfma):
    """Schur complement + fp64 Cholesky + back-substitution == the oracle's dense solve (use_mfma = 2: the IMU groups through the general
    body)."""
    w = win_cfg1.copy()
    d_o, mc_o = oracle.OracleWindow(w.copy()).lm_step(1e4, use_schur=False)
    with cv.Solver(precision=prec, use_mfma=mfma) as s:
        s.set_windows([w])
        d_g, mc_g = s.lm_step(0, 1e4)
    assert np.abs(d_g - d_o).max() <= tol * np.abs(d_o).max()
    assert mc_g == pytest.approx(mc_o, rel=max(tol * 1e-2, 1e-9))


def test_removed_modes_are_refused(cv):
    """ctvio_create refuses what no longer exists instead of silently running something else: the vector-ALU cross-check kernels (use_mfma = 0,
    removed in round 6) and the mixed fp32 precision (removed in round 3)."""
    with pytest.raises(cv.capi.CtvioError, match="use_mfma"):
        cv.Solver(use_mfma=0)
    with pytest.raises(ValueError):
        cv.Solver(precision="fp32")


def test_cost_kernels(cv, oracle, win_cfg1):
    w = win_cfg1.copy()
    c = oracle.OracleWindow(w.copy()).cost()
    for prec, rel in (("fp64", 1e-12),):
        with cv.Solver(precision=prec) as s:
            s.set_windows([w.copy()])
            assert s.cost(0) == pytest.approx(c, rel=rel)


def test_solve_fp64_reproduces_oracle_iterates(cv, oracle):
    """Same LM decisions, same iteration count, same final state (Ceres semantics restated twice)."""
    for cfg, seed in (("tiny", 7), ("config1", 1001)):
        w0 = cv.synth.make_window(cfg, seed=seed)
        wo = w0.copy()
        sm_o = oracle.OracleWindow(wo).solve(15)
        with cv.Solver(precision="fp64") as s:
            wg = w0.copy()
            s.set_windows([wg])
            sm = s.solve(15)[0]
        assert sm["iterations"] == sm_o.iterations and sm["num_successful"] == sm_o.num_successful
        assert sm["final_cost"] == pytest.approx(sm_o.final_cost, rel=1e-9)
        assert cv.rel_state_error(wg, wo)["state"] < 1e-6


N_SEEDS = 32


@pytest.mark.parametrize("cfg", ["config1", "config2", "config3", "tumrs"])
def test_product_parity_every_window(cv, oracle_solved, cfg):
    """BASELINE target: final state within 1e-4 (relative) of the fp64 reference solve with identical Ceres settings (15
    iterations, function tolerance 1e-6, projected line search) on EVERY window -- 32 seeds per config, solved as one batch
    by the product path (all-fp64 HIP); "tumrs" is the reference's native operating point (200 Hz IMU: 10 samples per group; <= 150
    features per frame).  The same restated solver runs on both sides, so the device must reproduce the
    reference's decisions: iteration count, successful / unsuccessful steps and line-search steps are compared exactly; the
    contract bound is 1e-4, the engineering bound asserted on top of it is 1e-6 (measured ~1e-9)."""
    ws = [cv.synth.make_window(cfg, seed=1000 + i) for i in range(N_SEEDS)]
    refs, sms_o = zip(*[oracle_solved(cfg, 1000 + i) for i in range(N_SEEDS)])
    with cv.Solver() as s:
        batch = [w.copy() for w in ws]
        s.set_windows(batch)
        sms = s.solve(15)
    worst = 0.0
    for i, (sm, so) in enumerate(zip(sms, sms_o)):
        assert (sm["iterations"], sm["num_successful"], sm["num_unsuccessful"]) == (so.iterations, so.num_successful, so.num_unsuccessful), (i, sm)
        assert (sm["num_line_search_steps"], sm["num_line_search_reduced"]) == (so.num_line_search_steps, so.num_line_search_reduced), (i, sm)
        assert sm["final_cost"] == pytest.approx(so.final_cost, rel=1e-9), i
        err = cv.rel_state_error(batch[i], refs[i])["state"]
        worst = max(worst, err)
        assert err < 1e-4, (i, err)          # the contract (BASELINE.json north_star)
    assert worst < 1e-6, worst               # what the all-fp64 path actually delivers
    if cfg == "config3":                     # the rolling-shutter stress windows exercise the projected line search
        assert sum(so.num_line_search_reduced for so in sms_o) > 0


def test_config2_batch_of_64_equals_64_singles(cv):
    """BASELINE configs[3] shape: 64 independent config-2 windows (seeds 1000..1063) in one batch give, window by window, what
    64 single-window solves give (same kernels, different launch geometry / tile maps / atomics order)."""
    ws = [cv.synth.make_window("config2", seed=1000 + i) for i in range(64)]
    with cv.Solver() as s:
        batch = [w.copy() for w in ws]
        s.set_windows(batch)
        sms = s.solve(15)
    with cv.Solver() as s1:
        for i, w in enumerate(ws):
            w1 = w.copy()
            s1.set_windows([w1])
            sm1 = s1.solve(15)[0]
            assert sms[i]["iterations"] == sm1["iterations"] and sms[i]["num_successful"] == sm1["num_successful"]
            assert sms[i]["final_cost"] == pytest.approx(sm1["final_cost"], rel=1e-10)
            assert cv.rel_state_error(batch[i], w1)["state"] < 1e-7, i


@pytest.mark.parametrize("prec", ["fp64"])
def test_large_batch_kernels_match_small_batch_kernels(cv, prec):
    """From 192 windows per launch on, the per-window kernels take over (k_schur_window_f64 / k_schur_window, single-part
    visual assembly): 208 windows (8 distinct config-1 windows, 26 copies each) against the same 8 solved in a small batch
    (per-tile kernels).  Both are the same arithmetic in a different order."""
    base = [cv.synth.make_window("config1", seed=1100 + i) for i in range(8)]
    with cv.Solver(precision=prec) as s:
        small = [w.copy() for w in base]
        s.set_windows(small)
        sm_small = s.solve(15)
        big = [base[i % 8].copy() for i in range(208)]
        s.set_windows(big)
        sm_big = s.solve(15)
    tol = 1e-7 if prec == "fp64" else 1e-3
    for i in range(208):
        assert sm_big[i]["iterations"] == sm_small[i % 8]["iterations"]
        assert sm_big[i]["final_cost"] == pytest.approx(sm_small[i % 8]["final_cost"], rel=1e-10 if prec == "fp64" else 1e-5)
        assert cv.rel_state_error(big[i], small[i % 8])["state"] < tol, i


def test_large_batch_fixed_unknowns_and_tiles_taken_from_hpp(cv, oracle, monkeypatch):
    """Large batches: k_schur_window_f64 leaves the 16 x 16 tiles W never reaches unwritten and the tile Cholesky (k_cholesky_flow / k_cholesky_tiles) forms them from Hpp + D itself,
    fixed unknowns included (activity as ballot masks).  208 windows (4 distinct: constant knots in and out of the prefix, a fixed line
    delay, locked gyro biases) against the same 4 in a small batch (tile Schur kernel: S written whole) and the oracle -- and against the same batch with
    the tiles copied by the Schur kernel as before (CTVIO_SCHUR_COPY_PLAIN: identical arithmetic, S in HBM instead of straight from Hpp)."""
    base = [cv.synth.make_window("config1", seed=1300 + i) for i in range(4)]
    base[0].knot_const = np.zeros(base[0].K, np.uint8); base[0].knot_const[[0, 1, 2]] = 1
    base[1].knot_const = np.zeros(base[1].K, np.uint8); base[1].knot_const[[0, 4, 11, base[1].K - 1]] = 1
    base[2].fix_ld = True
    base[3].lock_bg = True      # fixed unknowns INSIDE the tiles that come from Hpp (bias rows and columns)
    def run(n):
        with cv.Solver() as s:
            ws = [base[i % 4].copy() for i in range(n)]
            s.set_windows(ws)
            return ws, s.solve(15)
    small, sm_small = run(4)
    big, sm_big = run(208)
    monkeypatch.setenv("CTVIO_SCHUR_COPY_PLAIN", "1")
    big_copy, sm_copy = run(208)
    monkeypatch.delenv("CTVIO_SCHUR_COPY_PLAIN")
    for i in range(208):
        assert sm_big[i]["iterations"] == sm_small[i % 4]["iterations"]
        assert sm_big[i]["final_cost"] == pytest.approx(sm_small[i % 4]["final_cost"], rel=1e-10)
        assert cv.rel_state_error(big[i], small[i % 4])["state"] < 1e-7, i
        # (throughput mode accumulates with atomics: two runs of the SAME path differ as well -- seed 1300 with its three constant knots by
        # 5e-11 in the state as a rule and 4e-9 / 1.4e-12 in the cost once in a while, tools/studies/r4_margins.py -- so: the bounds of the
        # comparison with the small batch)
        assert sm_big[i]["final_cost"] == pytest.approx(sm_copy[i]["final_cost"], rel=1e-10) and sm_big[i]["iterations"] == sm_copy[i]["iterations"]
        assert cv.rel_state_error(big[i], big_copy[i])["state"] < 1e-7, i
    np.testing.assert_array_equal(big[0].quat[[0, 1, 2]], base[0].quat[[0, 1, 2]])
    np.testing.assert_array_equal(big[1].pos[[0, 4, 11, base[1].K - 1]], base[1].pos[[0, 4, 11, base[1].K - 1]])
    for i in range(4):
        wo = base[i].copy()
        so = oracle.OracleWindow(wo).solve(15)
        assert sm_big[i]["iterations"] == so.iterations
        assert sm_big[i]["final_cost"] == pytest.approx(so.final_cost, rel=1e-8)
        assert cv.rel_state_error(big[i], wo)["state"] < 1e-6


@pytest.mark.parametrize("dt_ms,K", [(42, 26), (40, 27)])
def test_large_batch_schur_variants_k26_k27(cv, oracle, dt_ms, K):
    """The other instantiations of the per-window fp64 Schur kernel: K = 26 (66 tiles with products: 14 accumulators per wave,
    k_schur_window_f64<5,14>) and K = 27 (164 compact W columns: <7,14>); both also take the global-atomic MFMA visual assembly
    (K > 25).  10 frames so that P <= 224.  207 windows (3 distinct, 69 copies each) against the same 3 in a small batch (tile
    Schur kernel) and against the oracle.  Seed 1201 is part of the set: with this knot spacing its solution is determined to ~1e-4
    only (the 15th iterate moves by that much under ANY change of the summation order: big-batch kernels vs small-batch kernels vs
    the oracle), so it is compared on cost, iterations and decisions only, while the well-determined seeds keep 1e-6 on the state."""
    seeds = (1200, 1202, 1203, 1201)
    base = [cv.synth.make_window("config1", seed=sd, F=10, dt_ns=dt_ms * 1_000_000) for sd in seeds]
    assert base[0].K == K and base[0].P <= 224
    with cv.Solver() as s:
        small = [w.copy() for w in base]
        s.set_windows(small)
        sm_small = s.solve(15)
        big = [base[i % 4].copy() for i in range(208)]
        s.set_windows(big)
        sm_big = s.solve(15)
    # seed 1201: COST parity only (its state is outside the 1e-4 contract: BASELINE.md section 6 says so) -- the kernels must still agree on
    # the cost, the iteration count and every decision; the three well-determined seeds keep the 1e-6 state bound
    ill = lambda i: seeds[i % 4] == 1201
    for i in range(208):
        assert sm_big[i]["iterations"] == sm_small[i % 4]["iterations"]
        assert sm_big[i]["final_cost"] == pytest.approx(sm_small[i % 4]["final_cost"], rel=1e-6 if ill(i) else 1e-8)
        # (the ill-determined seed keeps a loose but non-trivial bound -- its 15th iterate moves by ~1e-4 under a change of the
        #  summation order, 1e-3 would be a different solution)
        assert cv.rel_state_error(big[i], small[i % 4])["state"] < (1e-3 if ill(i) else 1e-6), i
    for i in range(4):
        wo = base[i].copy()
        so = oracle.OracleWindow(wo).solve(15)
        assert sm_big[i]["final_cost"] == pytest.approx(so.final_cost, rel=1e-6 if ill(i) else 1e-8)
        assert cv.rel_state_error(big[i], wo)["state"] < (1e-3 if ill(i) else 1e-6)
        if not ill(i):
            assert sm_big[i]["iterations"] == so.iterations


def test_golden_converged_state(cv, oracle, golden_dir):
    """Committed scipy fixture (tests/golden/config1_seed1000_converged.npz): independent minimiser."""
    d = np.load(os.path.join(golden_dir, "config1_seed1000_converged.npz"))
    w = cv.Window.from_dict(d, "w_")
    wf = cv.Window.from_dict(d, "f_")
    wo = w.copy()
    so = oracle.OracleWindow(wo).solve(50)
    with cv.Solver() as s:
        s.set_windows([w])
        sm = s.solve(50)[0]
    assert sm["final_cost"] == pytest.approx(float(d["final_cost"]), rel=2e-6)
    # With Ceres' default tolerances the solve stops ~2.5e-4 short of scipy's tightly converged minimiser on this window (the function
    # tolerance fires) -- so at these settings the device is held to the ORACLE's stopping point, iterate for iterate ...
    assert sm["iterations"] == so.iterations and sm["final_cost"] == pytest.approx(so.final_cost, rel=1e-9)
    assert cv.rel_state_error(w, wo)["state"] < 1e-6
    # ... and with the tolerances tightened (as tests/test_oracle_golden.py does for the oracle) it lands on scipy's minimiser: cost to
    # 5e-9, state to the 2e-5 the two independent optimisers agree on (Jacobi-scaled condition number ~1e10 in the near-gauge directions)
    w2 = cv.Window.from_dict(d, "w_")
    with cv.Solver(function_tolerance=1e-15, gradient_tolerance=1e-15, parameter_tolerance=1e-14) as s:
        s.set_windows([w2])
        sm2 = s.solve(200)[0]
    assert sm2["final_cost"] == pytest.approx(float(d["final_cost"]), rel=5e-9)
    assert cv.rel_state_error(w2, wf)["state"] < 2e-5


def test_spline_eval(cv, oracle, win_cfg1):
    w = win_cfg1.copy()
    t = np.linspace(w.t0_ns, w.max_time_ns() - 1, 257).astype(np.int64)
    ref = oracle.OracleWindow(w.copy()).spline_eval(t)
    with cv.Solver() as s:
        s.set_windows([w])
        got = s.spline_eval(0, t)
        for a, b in zip(got, ref):
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-11)
        with pytest.raises(cv.capi.CtvioError):
            s.spline_eval(0, np.array([w.max_time_ns()], np.int64))


def test_spline_eval_batch_all_windows_one_launch(cv, oracle):
    """ctvio_spline_eval_batch: queries of ANY windows of the batch in one launch (per-row poses of
    rolling-shutter frames for a whole batch) -- a ragged batch (different K, dt, t0), interleaved window ids, against the oracle's
    spline evaluation window by window and against the one-window entry; an out-of-range time or window id is an error."""
    ws = [cv.synth.make_window("tiny", seed=21), cv.synth.make_window("config1", seed=1004), cv.synth.make_window("config3", seed=1005)]
    ws[1].t0_ns = 1_000_000_007                       # (absolute times: every window has its own t0)
    ws[1].imu_t = ws[1].imu_t + 1_000_000_007; ws[1].v_ti = ws[1].v_ti + 1_000_000_007; ws[1].v_tj = ws[1].v_tj + 1_000_000_007
    rng = np.random.default_rng(4)
    win = rng.integers(0, len(ws), 4000).astype(np.int32)
    t = np.array([rng.integers(ws[i].t0_ns, ws[i].max_time_ns()) for i in win], np.int64)
    with cv.Solver() as s:
        s.set_windows([w.copy() for w in ws])
        out, ms = s.spline_eval_batch(win, t, want=("pose", "vel", "omega", "acc"))
        assert ms > 0.0
        for i, w in enumerate(ws):
            sel = np.flatnonzero(win == i)
            ref = oracle.OracleWindow(w.copy()).spline_eval(t[sel])        # pose, vel, omega, acc
            one = s.spline_eval(i, t[sel])
            for k, key in enumerate(("pose", "vel", "omega", "acc")):
                np.testing.assert_allclose(out[key][sel], ref[k], rtol=1e-12, atol=1e-11)
                np.testing.assert_array_equal(out[key][sel], one[k])          # the same kernel body: bitwise
        only, _ = s.spline_eval_batch(win[:10], t[:10], want=("omega",))
        assert set(only) == {"omega"}
        np.testing.assert_array_equal(only["omega"], out["omega"][:10])
        with pytest.raises(cv.capi.CtvioError):
            s.spline_eval_batch(np.array([0], np.int32), np.array([ws[0].max_time_ns()], np.int64))
        with pytest.raises(cv.capi.CtvioError):
            s.spline_eval_batch(np.array([3], np.int32), np.array([ws[0].t0_ns], np.int64))


def test_sensor_pose(cv, oracle, win_cfg1):
    """ctvio_sensor_pose = Trajectory::GetSensorPose (trajectory.cpp:39-56): pose_I_to_G(t) * T_StoI, against the oracle's
    poseNs composed with the reference's camera extrinsic on the host (numpy, fp64)."""
    w = win_cfg1.copy()
    t = np.linspace(w.t0_ns, w.max_time_ns() - 1, 129).astype(np.int64)
    pose = oracle.OracleWindow(w.copy()).spline_eval(t)[0]
    q_SI = np.asarray(w.q_CI, float); p_SI = np.asarray(w.p_CI, float)

    def qmul(a, b):
        ax, ay, az, aw = a.T; bx, by, bz, bw = b.T
        return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                         aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], -1)

    def qrot(q, v):
        qv = np.concatenate([np.broadcast_to(v, (q.shape[0], 3)), np.zeros((q.shape[0], 1))], 1)
        qc = q * np.array([-1.0, -1.0, -1.0, 1.0])
        return qmul(qmul(q, qv), qc)[:, :3]

    q = pose[:, 3:7]
    ref = np.concatenate([pose[:, :3] + qrot(q, p_SI), qmul(q, np.broadcast_to(q_SI / np.linalg.norm(q_SI), q.shape))], 1)
    with cv.Solver() as s:
        s.set_windows([w])
        got = s.sensor_pose(0, t, q_SI, p_SI)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-11)
        ident = s.sensor_pose(0, t, [0, 0, 0, 1], [0, 0, 0])
        np.testing.assert_allclose(ident, pose, rtol=1e-12, atol=1e-11)
        with pytest.raises(cv.capi.CtvioError):
            s.sensor_pose(0, np.array([w.max_time_ns()], np.int64), q_SI, p_SI)
        with pytest.raises(cv.capi.CtvioError):
            s.sensor_pose(0, t, [0, 0, 0, 0], p_SI)


def test_config3_rolling_shutter_stress(cv, oracle):
    """BASELINE configs[3]: 300 landmarks, 640-row images, 30 us line delay (every block's two ends evaluate at their own
    per-row times), line delay estimated from 0.  These windows are NOT converged after Ceres' 15 iterations, so the 15th
    iterate is only determined up to the solver's own stopping slop (oracle at 15 iterations vs oracle run to 1e-13):
      * the product (all-fp64) path must reproduce the oracle's iterate itself (same decisions, state to 1e-6).
    Then the spline is evaluated at every row time of every frame (11 x 640 = 7040 timestamps) against the oracle."""
    w0 = cv.synth.make_window("config3", seed=1003)
    wo = w0.copy()
    sm_o = oracle.OracleWindow(wo).solve(15)
    oracle.set_tolerances(1e-13, 1e-14, 1e-13)
    try:
        wt = w0.copy()
        oracle.OracleWindow(wt).solve(300)
    finally:
        oracle.set_tolerances()
    slop = cv.rel_state_error(wo, wt)["state"]
    with cv.Solver(precision="fp64") as s:
        w64 = w0.copy()
        s.set_windows([w64])
        sm64 = s.solve(15)[0]
    assert sm64["iterations"] == sm_o.iterations and sm64["num_successful"] == sm_o.num_successful
    assert sm64["final_cost"] == pytest.approx(sm_o.final_cost, rel=1e-9)
    assert cv.rel_state_error(w64, wo)["state"] < 1e-6
    with cv.Solver() as s:
        wg = w0.copy()
        s.set_windows([wg])
        sm = s.solve(15)[0]
        assert sm["iterations"] == sm_o.iterations
        assert cv.rel_state_error(wg, wo)["state"] < 1e-6
        assert abs(wg.ld - wo.ld) < 1e-6 * abs(wo.ld) + 1e-12
        frames = np.unique(np.concatenate([wg.v_ti, wg.v_tj]))
        ld_ns = int(wg.ld * 1e9)
        t = (frames[:, None] + np.arange(640, dtype=np.int64)[None, :] * ld_ns).reshape(-1)
        t = t[(t >= wg.t0_ns) & (t < wg.max_time_ns())]
        assert t.size >= 6000
        got = s.spline_eval(0, t)
        ref = oracle.OracleWindow(wg.copy()).spline_eval(t)
        for a, b in zip(got, ref):
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-10)


def test_marginalize_prior_construction(cv, oracle):
    """ctvio_marginalize: A, b assembled on the device, eliminated and factored on the host, against the
    oracle's restatement of MarginalizationInfo::marginalize.  J0 is unique only up to the eigenvector basis: what the next
    window uses is compared -- J0^T J0, J0^T r0, |r0|^2 (measured: fp64 path 1e-11 .. 6e-9, default mixed path 7e-10 on
    J0^T J0 and 2e-7 .. 5e-7 on J0^T r0, configs 1-2)."""
    w = cv.synth.make_window("config1", seed=1000)
    w.cauchy_a = 1.0                                   # the reference marginalises with CauchyLoss(1.0)
    role = np.zeros(w.N, np.int8)
    role[:12] = 1                                      # two oldest knots
    role[6 * w.K:6 * w.K + 6] = 1                      # oldest bias state
    role[w.P:w.P + w.L // 2] = 1                       # landmarks anchored in the dropped frame
    ko, Jo, ro = oracle.OracleWindow(w.copy()).marginalize(role, 1e-8)
    Ho, go, co = Jo.T @ Jo, Jo.T @ ro, ro @ ro
    for prec, tol in (("fp64", 1e-7),):
        with cv.Solver(precision=prec) as s:
            s.set_windows([w.copy()])
            kept, J0, r0 = s.marginalize(0, role, 1e-8)
            # (every other unknown of the window is KEPT here: n = N - 43 = 218 > 180, beyond the in-LDS eigen-solver -- this case takes
            #  ctvio_marginalize's host leg, and the handle says so; the reference's own drop set (26 / 91) runs on the device: test_gpu_slide.py)
            assert s.marginalize_ran_on_host() == (w.N - int(role.sum()) > 180)
        assert np.array_equal(kept, ko)
        assert np.abs(J0.T @ J0 - Ho).max() <= tol * np.abs(Ho).max(), prec
        assert np.abs(J0.T @ r0 - go).max() <= tol * np.abs(go).max(), prec
        assert abs(r0 @ r0 - co) <= tol * co, prec
    with pytest.raises(cv.capi.CtvioError):
        bad = role.copy(); bad[3] = 2
        s2 = cv.Solver(); s2.set_windows([w.copy()]); s2.marginalize(0, bad)


def test_prior_chain_on_device(cv):
    """Windows can be chained on the device: ctvio_marginalize on the factors of a dropped landmark set, its (J0, r0) fed
    to the window of the remaining factors as ctvio_window.pJ0 / pr0 -- the Gauss-Newton step of that window equals the
    step of the full window on the shared unknowns (fp64 path; gauge fixed in both solves, see the CPU twin in
    tests/test_marginalize_host.py)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from chain_helpers import chain_case, prior_arrays
    w, wD, wR, mapR = chain_case("config1", 1000)
    with cv.Solver(precision="fp64") as s:
        s.set_windows([wD.copy()])
        Hpp = s.linearize(0)[0]
        role = np.where(np.arange(wD.N) >= wD.P, 1, np.where(np.concatenate([np.diag(Hpp), np.ones(wD.L)]) > 0, 0, -1)).astype(np.int8)
        kept, J0, r0 = s.marginalize(0, role, 1e-8)
    wR.pJ0, wR.pr0, wR.p_kind, wR.p_index, wR.p_off, wR.p_x0 = prior_arrays(wR, kept, J0, r0)
    wR.normalize()
    w.fixed_upto = 3
    wR.fixed_upto = 3
    with cv.Solver(precision="fp64") as s:
        s.set_windows([w.copy(), wR.copy()])
        d_full, _ = s.lm_step(0, 1e16)
        d_red, _ = s.lm_step(1, 1e16)
    P = w.P
    assert np.abs(d_red[:P] - d_full[:P]).max() < 1e-6 * np.abs(d_full[:P]).max()
    assert np.abs(d_red[P:] - d_full[P + mapR]).max() < 1e-6 * np.abs(d_full[P:]).max()


def test_gauge_restore(cv, oracle):
    """ctvio_gauge_restore (reference double2vector, the step right after Solve) on two windows of a batch at once, against
    the oracle: regular case (yaw only) and a reference pose pitched to the Euler singularity (full rotation)."""
    from scipy.spatial.transform import Rotation as R
    ws = [cv.synth.make_window("config1", seed=1000 + i) for i in range(3)]
    with cv.Solver() as s:
        s.set_windows(ws)
        q0 = np.stack([ws[0].quat[2], (R.from_euler("y", 89.7, degrees=True) * R.from_quat(ws[2].quat[5])).as_quat()])
        t0 = np.stack([ws[0].pos[2], ws[2].pos[5] + 0.3])
        s.solve(15)                                   # moves the knots away from their initial gauge
        solved = [w.copy() for w in ws]
        s.gauge_restore([0, 2], [2, 5], q0, t0)
        out = [s.get_state(i) for i in range(3)]
    for e, (wi, k) in enumerate(((0, 2), (2, 5))):
        qo, po = oracle.gauge_restore(solved[wi].quat.copy(), solved[wi].pos.copy(), k, q0[e], t0[e])
        np.testing.assert_allclose(out[wi].quat, qo, rtol=0, atol=1e-13)
        np.testing.assert_allclose(out[wi].pos, po, rtol=0, atol=1e-12)
        np.testing.assert_allclose(out[wi].pos[k], t0[e], atol=1e-12)
    np.testing.assert_array_equal(out[1].quat, solved[1].quat)            # window not listed: untouched
    np.testing.assert_array_equal(out[0].quat[:2], solved[0].quat[:2])    # knots before the reference knot: untouched
    with pytest.raises(cv.capi.CtvioError):
        s2 = cv.Solver(); s2.set_windows([ws[0]]); s2.gauge_restore([0], [99], q0[:1], t0[:1])


def test_ragged_batch_equals_single(cv):
    """Windows of different sizes in one batch; each must match its own single-window solve."""
    ws = [cv.synth.make_window("config1", seed=1000 + i) for i in range(3)] + [cv.synth.make_window("tiny", seed=5),
                                                                                  cv.synth.make_window("config2", seed=1003)]
    with cv.Solver(precision="fp64") as s:
        batch = [w.copy() for w in ws]
        s.set_windows(batch)
        sms = s.solve(15)
    for i, w in enumerate(ws):
        with cv.Solver(precision="fp64") as s1:
            w1 = w.copy()
            s1.set_windows([w1])
            sm1 = s1.solve(15)[0]
        assert sms[i]["iterations"] == sm1["iterations"]
        assert cv.rel_state_error(batch[i], w1)["state"] < 1e-7


def test_large_ragged_batch_equals_small_batch(cv):
    """The same in a LARGE batch (per-window kernels: each workgroup takes its tile grid, its staged columns and the tiles it reads from Hpp
    from its own window's sizes): 210 windows of five shapes -- config 1 / 2, the tiny window, 8 and 9 frames (K = 21 / 24) -- against the
    five solved in a small batch."""
    base = [cv.synth.make_window("config1", seed=1320), cv.synth.make_window("tiny", seed=6), cv.synth.make_window("config2", seed=1321),
            cv.synth.make_window("config1", seed=1322, F=8), cv.synth.make_window("config1", seed=1323, F=9)]
    assert len({w.P for w in base}) >= 4
    def run(n):
        with cv.Solver() as s:
            ws = [base[i % 5].copy() for i in range(n)]
            s.set_windows(ws)
            return ws, s.solve(15)
    small, sm_small = run(5)
    big, sm_big = run(210)
    for i in range(210):
        assert sm_big[i]["iterations"] == sm_small[i % 5]["iterations"]
        assert sm_big[i]["final_cost"] == pytest.approx(sm_small[i % 5]["final_cost"], rel=1e-9)
        assert cv.rel_state_error(big[i], small[i % 5])["state"] < 1e-7, i


def test_random_factor_structures_match_oracle(cv, oracle):
    """Randomised structure, one ragged batch: random subsets of the visual blocks (landmarks left with one block or none, frame pairs
    thinned out), blocks with their own i end (several anchors per landmark), blocks reordered, free / fixed line delay at random
    values, locked biases, constant knots in the middle, no prior -- every window's dense normal equations and cost from the device
    against the oracle's, which knows nothing of anchors, slots or items."""
    rng = np.random.default_rng(2024)
    ws = []
    for i in range(14):
        w = cv.synth.make_window("tiny" if i % 3 else "config1", seed=500 + i, with_prior=bool(i % 2))
        keep = rng.random(w.V) < rng.uniform(0.3, 1.0)
        if i == 5:
            keep[:] = False                                    # no visual blocks at all (landmarks stay, unobserved)
        own = rng.random(w.V) < (0.25 if i % 4 == 1 else 0.0)  # blocks with an i end of their own
        w.v_rowi = np.where(own, rng.integers(0, 1024, w.V), w.v_rowi).astype(w.v_rowi.dtype)
        w.v_pi = w.v_pi + own[:, None] * rng.normal(0.0, 0.01, (w.V, 2))
        order = rng.permutation(np.flatnonzero(keep))          # the caller's order is arbitrary
        for name in ("v_lm", "v_ti", "v_tj", "v_rowi", "v_rowj", "v_pi", "v_pj"):
            setattr(w, name, np.ascontiguousarray(getattr(w, name)[order]))
        w.ld = float(rng.uniform(0.0, 3.5e-5))
        w.fix_ld = bool(i % 5 == 0)
        w.lock_bg = bool(i % 7 == 3); w.lock_ba = bool(i % 7 == 4)
        if i % 6 == 2:
            kc = np.zeros(w.K, np.uint8); kc[rng.integers(2, w.K - 2, 2)] = 1
            w.knot_const = kc
        w.normalize()
        ws.append(w)
    with cv.Solver() as s:
        s.set_windows([w.copy() for w in ws])
        for i, w in enumerate(ws):
            H, g, cost = oracle.OracleWindow(w.copy()).build_normal()
            P = w.P
            sc = np.sqrt(np.maximum(np.diag(H), 1e-30))
            Hg, Wg, Hllg, gg, costg = s.linearize(i)
            assert costg == pytest.approx(cost, rel=1e-12), i
            assert np.abs((Hg - H[:P, :P]) / np.outer(sc[:P], sc[:P])).max() < 1e-10, i
            if w.L:
                assert np.abs((Wg - H[:P, P:]) / np.outer(sc[:P], sc[P:])).max() < 1e-10, i
                obs = np.diag(H)[P:] > 0
                assert np.abs(Hllg[obs] / np.diag(H)[P:][obs] - 1).max() < 1e-10 if obs.any() else True, i
            assert np.abs((gg - g) / np.maximum(sc, 1e-12)).max() < 1e-10 * max(np.abs(g / np.maximum(sc, 1e-12)).max(), 1.0), i


def test_edge_cases(cv, oracle):
    """IMU-only predict with fixed knots and locked biases (reference InitTrajectory, trajectory_manager.cpp:288-315),
    no prior, fixed line delay, a landmark without observations."""
    w0 = cv.synth.make_window("tiny", seed=9, with_prior=False)
    # (a) IMU-only, biases locked, knots <= 5 fixed, 8 iterations
    wa = w0.copy()
    wa.v_lm = wa.v_lm[:0]; wa.v_ti = wa.v_ti[:0]; wa.v_tj = wa.v_tj[:0]; wa.v_rowi = wa.v_rowi[:0]; wa.v_rowj = wa.v_rowj[:0]
    wa.v_pi = wa.v_pi[:0]; wa.v_pj = wa.v_pj[:0]
    wa.bc_i = wa.bc_i[:0]; wa.bc_j = wa.bc_j[:0]; wa.bc_w = wa.bc_w[:0]
    wa.lock_bg = wa.lock_ba = True; wa.fixed_upto = 5
    wa.normalize()
    # (b) fixed line delay + one unobserved landmark appended
    wb = w0.copy()
    wb.fix_ld = True; wb.ld = 2.0e-5
    wb.rho = np.concatenate([wb.rho, [0.3]])
    wb.normalize()
    for w, iters in ((wa, 8), (wb, 15)):
        wo = w.copy()
        sm_o = oracle.OracleWindow(wo).solve(iters)
        with cv.Solver(precision="fp64") as s:
            wg = w.copy()
            s.set_windows([wg])
            sm = s.solve(iters)[0]
        assert sm["iterations"] == sm_o.iterations
        assert sm["final_cost"] == pytest.approx(sm_o.final_cost, rel=1e-8)
        assert cv.rel_state_error(wg, wo)["state"] < 1e-6
    assert wb.rho[-1] == 0.3  # unobserved landmark untouched (not in the reduced program)


def test_invalid_inputs(cv):
    w = cv.synth.make_window("tiny", seed=2)
    with cv.Solver() as s:
        bad = w.copy(); bad.imu_t = bad.imu_t.copy(); bad.imu_t[0] = w.max_time_ns() + 5
        with pytest.raises(cv.capi.CtvioError):
            s.add_window(bad)
        bad = w.copy(); bad.v_lm = bad.v_lm.copy(); bad.v_lm[0] = w.L + 3
        with pytest.raises(cv.capi.CtvioError):
            s.add_window(bad)
        with pytest.raises(cv.capi.CtvioError):
            s.upload()            # no windows
        s.add_window(w)
        with pytest.raises(cv.capi.CtvioError):
            s.solve(5)            # not uploaded
    # more than 64 observations of one landmark: the landmark's blocks must fit in one wave of k_vis_eval (rejected at upload)
    big = w.copy()
    rep = 70
    for name in ("v_lm", "v_ti", "v_tj", "v_rowi", "v_rowj"):
        a = np.asarray(getattr(big, name)); setattr(big, name, np.concatenate([a, np.repeat(a[:1], rep)]))
    for name in ("v_pi", "v_pj"):
        a = np.asarray(getattr(big, name)).reshape(-1, 2); setattr(big, name, np.concatenate([a, np.repeat(a[:1], rep, axis=0)]))
    with cv.Solver() as s:
        with pytest.raises(cv.capi.CtvioError, match="64 observations"):
            s.set_windows([big])
        # what the sparsity plan needs from its inputs: finite observations, an ordered line-delay box ...
        bad = w.copy(); bad.v_pj = bad.v_pj.copy(); bad.v_pj[3, 1] = np.nan
        with pytest.raises(cv.capi.CtvioError, match="non-finite"):
            s.set_windows([bad])
        bad = w.copy(); bad.ld_lo, bad.ld_hi = 3.0e-5, 1.0e-5
        with pytest.raises(cv.capi.CtvioError, match="ld_lo <= ld_hi"):
            s.set_windows([bad])
        # ... and a FIXED line delay stays what it was at upload (the landmarks' knot spans were planned for it); a free one is projected
        fx = w.copy(); fx.fix_ld = True; fx.ld = 2.0e-5
        s.set_windows([fx])
        moved = fx.copy(); moved.ld = 2.5e-5
        with pytest.raises(cv.capi.CtvioError, match="fix_ld"):
            s.set_state(0, moved)
        s.set_state(0, fx)                       # the same value: fine
        fr = w.copy(); fr.ld = 1.0e-5
        s.set_windows([fr])
        out = fr.copy(); out.ld = 9.0e-5         # outside the box [0, 3.5e-5]: projected, as Ceres projects a bounded parameter
        s.set_state(0, out)
        assert s.get_state(0).ld == pytest.approx(fr.ld_hi)


def test_full_size_properties(cv):
    """config2 / config5 sizes: size-independent properties -- cost decreases monotonically over accepted steps,
    a solved window is a fixed point (re-solving it terminates at once), and the batch solve is invariant to
    the order of the windows."""
    w2 = cv.synth.make_window("config2", seed=1010)
    w5 = cv.synth.make_window("config5", seed=1011)
    with cv.Solver() as s:
        a, b = w2.copy(), w5.copy()
        s.set_windows([a, b])
        sm = s.solve(30)
        assert all(m["final_cost"] < m["initial_cost"] for m in sm)
        c, d = a.copy(), b.copy()
        s.set_windows([d, c])                      # swapped order, start from the solution
        sm2 = s.solve(30)
        assert all(m["iterations"] <= 2 for m in sm2), sm2
        assert cv.rel_state_error(c, a)["state"] < 1e-5 and cv.rel_state_error(d, b)["state"] < 1e-5


@pytest.mark.parametrize("prec", ["fp64"])
def test_imu_only_window_without_landmarks(cv, oracle, prec):
    """L = 0, V = 0 (IMU-only window, e.g. the predict solve of the reference's InitTrajectory with no features yet): the Schur
    kernels must not touch a landmark row (k_schur_mfma / k_schur_tile_f64 used to clamp to row L - 1 = -1)."""
    w = cv.synth.make_window("config1", seed=1004)
    z = lambda a: a[:0]
    w.v_lm, w.v_ti, w.v_tj, w.v_rowi, w.v_rowj, w.v_pi, w.v_pj = z(w.v_lm), z(w.v_ti), z(w.v_tj), z(w.v_rowi), z(w.v_rowj), z(w.v_pi), z(w.v_pj)
    w.rho = w.rho[:0]
    w.fix_ld = True
    w.normalize()
    assert w.L == 0 and w.V == 0
    wo = w.copy()
    sm_o = oracle.OracleWindow(wo).solve(15)
    with cv.Solver(precision=prec) as s:
        wg = w.copy()
        s.set_windows([wg])
        sm = s.solve(15)[0]
    assert np.isfinite(sm["final_cost"]) and sm["termination"] != "failure"
    # (the mixed mode has no 1e-4 contract; an IMU-only window has unobservable directions in which its 15th iterate drifts)
    assert sm["final_cost"] == pytest.approx(sm_o.final_cost, rel=1e-9 if prec == "fp64" else 1e-3)
    assert cv.rel_state_error(wg, wo)["state"] < (1e-6 if prec == "fp64" else 5e-2)


def test_large_batch_with_imu_only_windows(cv, oracle):
    """A large batch (per-window Schur kernel, tiles without products read from Hpp by the tile Cholesky) in which every other window has no
    landmarks at all: no chunk of W to stage, accumulators that only ever hold -Hpp, the rhs from the gradient alone.  Against the same
    windows in a small batch and the oracle's cost."""
    def imu_only(seed):
        w = cv.synth.make_window("config1", seed=seed)
        z = lambda a: a[:0]
        w.v_lm, w.v_ti, w.v_tj, w.v_rowi, w.v_rowj, w.v_pi, w.v_pj = z(w.v_lm), z(w.v_ti), z(w.v_tj), z(w.v_rowi), z(w.v_rowj), z(w.v_pi), z(w.v_pj)
        w.rho = w.rho[:0]
        w.fix_ld = True
        w.normalize()
        return w
    base = [cv.synth.make_window("config1", seed=1310), imu_only(1311), cv.synth.make_window("config1", seed=1312), imu_only(1313)]
    assert base[1].L == 0 and base[1].V == 0
    def run(n):
        with cv.Solver() as s:
            ws = [base[i % 4].copy() for i in range(n)]
            s.set_windows(ws)
            return ws, s.solve(15)
    small, sm_small = run(4)
    big, sm_big = run(208)
    for i in range(208):
        assert np.isfinite(sm_big[i]["final_cost"]) and sm_big[i]["termination"] != "failure"
        assert sm_big[i]["iterations"] == sm_small[i % 4]["iterations"]
        assert sm_big[i]["final_cost"] == pytest.approx(sm_small[i % 4]["final_cost"], rel=1e-9)
        # (an IMU-only window has unobservable directions in which its 15th iterate drifts: cost parity only)
        if base[i % 4].L > 0:
            assert cv.rel_state_error(big[i], small[i % 4])["state"] < 1e-7, i
    for i in range(4):
        so = oracle.OracleWindow(base[i].copy()).solve(15)
        assert sm_big[i]["final_cost"] == pytest.approx(so.final_cost, rel=1e-8)


def test_mixed_batch_tiny_window_and_imu_only_long_spline_deterministic(cv, oracle):
    """A window whose packed Hessian does not fit in LDS (K >= 25) and that has NO visual blocks -- the IMU-only predict of a long
    spline -- batched with an LDS-resident window, deterministic mode on: the store-semantics tail only finishes LDS-resident windows,
    so the batch must take the accumulate path; both windows against the oracle (round-3 advisor finding: the IMU-only window's
    normal equations were left unwritten)."""
    tiny = cv.synth.make_window("tiny", seed=11)
    big = cv.synth.make_window("config1", seed=1200, F=10, dt_ns=40_000_000, with_prior=False)    # K = 27
    assert big.K >= 25
    pred = cv.Solver.predict_window(big, fixed_upto=-1)
    assert pred.V == 0
    ref = []
    for w0 in (tiny, pred):
        wo = w0.copy()
        ref.append((wo, oracle.OracleWindow(wo).solve(8)))
    with cv.Solver() as s:               # (default: the deterministic mode for batches of <= 64 windows where it applies)
        ws = [tiny.copy(), pred.copy()]
        s.set_windows(ws)
        sms = s.solve(8)
    with cv.Solver(deterministic=1) as s:   # an explicit request that cannot be honoured is refused, not silently dropped
        with pytest.raises(cv.capi.CtvioError):
            s.set_windows([tiny.copy(), pred.copy()])
    for wg, sm, (wo, sm_o) in zip(ws, sms, ref):
        assert sm["iterations"] == sm_o.iterations
        assert sm["final_cost"] == pytest.approx(sm_o.final_cost, rel=1e-8)
        assert cv.rel_state_error(wg, wo)["state"] < 1e-6


def test_imu_only_predict_named_entry(cv, oracle):
    """Solver.predict = the reference's InitTrajectory (trajectory_manager.cpp:288-315): IMU factors only, biases locked, knots
    up to the fixed index constant, Solve(8) -- on a config-2-sized window, product precision and the mixed mode."""
    w = cv.synth.make_window("config2", seed=1005, with_prior=False)
    fixed = w.K - 5                                       # only the newly added control points are optimised
    wo = cv.Solver.predict_window(w, fixed_upto=fixed)
    sm_o = oracle.OracleWindow(wo).solve(8)
    for prec, tol in (("fp64", 1e-6),):
        with cv.Solver(precision=prec) as s:
            wg = w.copy()
            sm = s.predict([wg], fixed_upto=[fixed])[0]
        assert sm["iterations"] == sm_o.iterations
        assert sm["final_cost"] == pytest.approx(sm_o.final_cost, rel=1e-8 if prec == "fp64" else 1e-4)
        np.testing.assert_array_equal(wg.quat[:fixed + 1], w.quat[:fixed + 1])   # constant blocks untouched
        np.testing.assert_array_equal(wg.bias, w.bias)
        np.testing.assert_array_equal(wg.rho, w.rho)
        assert np.abs(wg.quat - wo.quat).max() < tol and np.abs(wg.pos - wo.pos).max() < tol


@pytest.mark.parametrize("name", ["tiny_ld_lo.npz", "tiny_ld_hi.npz", "tiny_rows.npz", "tiny_seed7.npz"])
@pytest.mark.parametrize("prec,tol", [("fp64", 1e-9)])
def test_golden_edge_fixtures_through_the_hip_path(cv, oracle, golden_dir, name, prec, tol):
    """The committed edge fixtures (line delay at both bounds, rows 0 / 1023; made by the independent NumPy restatement) pushed
    through k_vis_eval / k_imu_linearize and the assembly: cost against the fixture's own value, dense H / g against the oracle
    (which tests/test_oracle_golden.py pins block by block to the same fixtures) and, where the fixture carries them, against the fixture's
    own finite-difference H / g."""
    d = np.load(os.path.join(golden_dir, name))
    w = cv.Window.from_dict(d, "w_")
    H, g, cost = oracle.OracleWindow(w.copy()).build_normal()
    P = w.P
    sc = _scaled(H)
    with cv.Solver(precision=prec) as s:
        s.set_windows([w.copy()])
        Hg, Wg, Hllg, gg, costg = s.linearize(0)
    assert costg == pytest.approx(float(d["cost"]), rel=1e-10 if prec == "fp64" else 1e-6)
    assert np.abs((Hg - H[:P, :P]) / np.outer(sc[:P], sc[:P])).max() < tol
    assert np.abs((Wg - H[:P, P:]) / np.outer(sc[:P], sc[P:])).max() < tol
    assert np.abs(Hllg / np.diag(H)[P:] - 1).max() < tol
    assert np.abs((gg - g) / sc).max() < tol * max(np.abs(g / sc).max(), 1.0)
    if "H" in d.files:
        # the fixture's own dense normal equations (finite-difference Jacobians of the independent NumPy restatement), straight against
        # the device's -- not via the oracle; tolerances = the finite-difference accuracy of the fixture (tests/test_oracle_golden.py)
        Hd = np.zeros_like(d["H"])
        Hd[:P, :P] = Hg; Hd[:P, P:] = Wg; Hd[P:, :P] = Wg.T; Hd[P:, P:] = np.diag(Hllg)
        scf = np.sqrt(np.maximum(np.diag(d["H"]), 1e-30))
        dn = np.abs(Hd / np.outer(scf, scf) - d["H"] / np.outer(scf, scf))
        ld = P - 1
        mask = np.ones(w.N, bool); mask[ld] = False
        assert dn[np.ix_(mask, mask)].max() < 2e-5 and dn[ld].max() < 1e-4
        gs = np.abs(d["g"]).max()
        assert np.abs(gg - d["g"])[mask].max() / gs < 1e-5 and abs(gg[ld] - d["g"][ld]) / abs(d["g"][ld]) < 1e-4


@pytest.mark.parametrize("name", ["fd_config2_seed1000.npz", "fd_config3_seed1001.npz"])
def test_bench_size_fd_fixtures_through_the_hip_path(cv, golden_dir, name):
    """The device's normal equations at the BENCHMARKED sizes (config 2 = the headline shape, config 3 = rolling-shutter stress) straight
    against the finite-difference fixtures of the independent NumPy restatement (make_golden.py bench_fd) -- not via the oracle;
    tolerances = the finite-difference accuracy of the fixtures (tests/test_oracle_golden.py uses the same)."""
    d = np.load(os.path.join(golden_dir, name))
    w = cv.Window.from_dict(d, "w_")
    P = w.P
    with cv.Solver() as s:
        s.set_windows([w.copy()])
        Hg, Wg, Hllg, gg, costg = s.linearize(0)
    assert costg == pytest.approx(float(d["cost"]), rel=1e-10)
    scp = np.sqrt(np.maximum(np.diag(d["Hpp"]), 1e-30)); scl = np.sqrt(np.maximum(d["Hll"], 1e-30))
    ld = P - 1
    mask = np.ones(P, bool); mask[ld] = False
    Hs = np.tril(Hg) + np.tril(Hg, -1).T                                  # (the device fills the lower triangle)
    dpp = np.abs(Hs - d["Hpp"]) / np.outer(scp, scp)
    dw = np.abs(Wg - d["W"]) / np.outer(scp, scl)
    assert dpp[np.ix_(mask, mask)].max() < 2e-5 and dpp[ld].max() < 1e-4
    assert dw[mask].max() < 2e-5 and dw[ld].max() < 1e-4
    assert np.abs(Hllg / d["Hll"] - 1).max() < 2e-5
    gs = np.abs(d["g"]).max()
    gm = np.ones(w.N, bool); gm[ld] = False
    assert np.abs(gg - d["g"])[gm].max() / gs < 1e-5 and abs(gg[ld] - d["g"][ld]) / abs(d["g"][ld]) < 1e-4


def test_config5_large_window_vs_oracle(cv, oracle_solved):
    """BASELINE configs[4]: 30 KF / 1000 landmarks / 6000 IMU (K = 64, P = 571, dense N = 1571): the product path against the
    oracle's solve, iterate for iterate, on 8 seeds solved as one batch."""
    ws = [cv.synth.make_window("config5", seed=1011 + i) for i in range(8)]
    refs, sms_o = zip(*[oracle_solved("config5", 1011 + i) for i in range(8)])
    with cv.Solver() as s:
        batch = [w.copy() for w in ws]
        s.set_windows(batch)
        sms = s.solve(15)
    for i, (sm, so) in enumerate(zip(sms, sms_o)):
        assert (sm["iterations"], sm["num_successful"], sm["num_unsuccessful"]) == (so.iterations, so.num_successful, so.num_unsuccessful), i
        assert sm["final_cost"] == pytest.approx(so.final_cost, rel=1e-9), i
        assert cv.rel_state_error(batch[i], refs[i])["state"] < 1e-6, i


def test_blocked_tile_schur_kernel_matches_the_one_tile_kernel(cv, oracle):
    """k_schur_tile2_f64 (one wave per 2 x 2 tiles: batches whose tile count fills the chip, e.g. config 5 x 128) against k_schur_tile_f64
    (one wave per tile) on the same windows -- forced through the A/B switch, since 3 windows would not select it -- and against the
    oracle: a config-5 window (P = 571: odd number of tile rows, rhs row in the last block) and two config-1 windows with K = 27."""
    ws = [cv.synth.make_window("config5", seed=1011)] + [cv.synth.make_window("config1", seed=1200 + i, F=10, dt_ns=40_000_000) for i in (0, 2)]
    res = {}
    for force in ("0", "1"):
        os.environ["CTVIO_SCHUR_TILE2"] = force
        try:
            with cv.Solver() as s:
                batch = [w.copy() for w in ws]
                s.set_windows(batch)
                res[force] = (batch, s.solve(15))
        finally:
            del os.environ["CTVIO_SCHUR_TILE2"]
    for i, w in enumerate(ws):
        a, b = res["0"][1][i], res["1"][1][i]
        assert (a["iterations"], a["num_successful"], a["num_unsuccessful"]) == (b["iterations"], b["num_successful"], b["num_unsuccessful"])
        assert a["final_cost"] == pytest.approx(b["final_cost"], rel=1e-10)
        assert cv.rel_state_error(res["1"][0][i], res["0"][0][i])["state"] < 1e-7
        ref = w.copy()
        so = oracle.OracleWindow(ref).solve(15)
        assert b["iterations"] == so.iterations and b["final_cost"] == pytest.approx(so.final_cost, rel=1e-9)
        assert cv.rel_state_error(res["1"][0][i], ref)["state"] < 1e-6


def test_deterministic_mode_is_bitwise_reproducible(cv):
    """ctvio_options.deterministic (default: on for batches of <= 64 windows): order-fixed accumulation everywhere -- cost and step
    reductions in fixed trees, the visual assembly in one-wave parts whose packed partial Hessians are summed in part order, bias rows /
    chain / prior by gather, no floating-point atomics -- so two solves of the same 64-window batch agree BITWISE: summaries and every
    state entry, and so do two separate solver handles.  (deterministic=0 on the same batch agrees to rounding only.)"""
    ws = [cv.synth.make_window("config2" if i % 2 else "config3", seed=1300 + i) for i in range(64)]
    runs = []
    for rep in range(3):
        with cv.Solver(deterministic=1) as s:
            batch = [w.copy() for w in ws]
            s.set_windows(batch)
            sms = s.solve(15)
            if rep == 0:                       # same handle, second solve from the same initial state
                again = [w.copy() for w in ws]
                s.set_windows(again)
                sms2 = s.solve(15)
                runs.append((sms2, again))
        runs.append((sms, batch))
    sm0, b0 = runs[0]
    for sm, b in runs[1:]:
        assert sm == sm0
        for x, y in zip(b, b0):
            assert np.array_equal(x.quat, y.quat) and np.array_equal(x.pos, y.pos) and np.array_equal(x.bias, y.bias)
            assert np.array_equal(x.rho, y.rho) and x.ld == y.ld
    with cv.Solver(deterministic=0) as s:      # the throughput mode: same answer to rounding
        batch = [w.copy() for w in ws]
        s.set_windows(batch)
        s.solve(15)
    assert max(cv.rel_state_error(x, y)["state"] for x, y in zip(batch, b0)) < 1e-6


def test_rccl_gather_world_size_1(cv):
    """sharding.gather_records on GPU tensors over the nccl backend (= RCCL on ROCm), world size 1, in a fresh interpreter
    (tests/rccl_gather_check.py): every window id comes back exactly once, values intact.  (N > 1 is covered with gloo on
    CPU: tests/test_sharding_gloo.py; bench.py --gpus N runs the same gather on N GPUs.)"""
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_gather_check.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_GATHER_OK 0 1" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_per_block_cauchy_and_non_prefix_constant_knots(cv, oracle):
    """ctvio_window.v_cauchy (one CauchyLoss width per residual block: the reference picks 1 | 2 per AddImageFeatureDelayAnalytic
    call, trajectory_estimator.cpp:320-323) and knot_const (SetParameterBlockConstant per AddControlPoints call, :134-138: any set
    of knots, not only a prefix) through the HIP path against the oracle: normal equations and the 15-iteration solve."""
    w0 = cv.synth.make_window("config1", seed=1021)
    w0.v_cauchy = np.where(np.arange(w0.V) % 3 == 0, 1.0, 2.0)
    w0.knot_const = np.zeros(w0.K, np.uint8); w0.knot_const[[0, 1, 5, 9]] = 1
    H, g, cost = oracle.OracleWindow(w0.copy()).build_normal()
    P = w0.P
    sc = _scaled(H)
    wo = w0.copy()
    sm_o = oracle.OracleWindow(wo).solve(15)
    with cv.Solver() as s:
        wg = w0.copy()
        s.set_windows([wg])
        Hg, Wg, Hllg, gg, costg = s.linearize(0)
        sm = s.solve(15)[0]
    assert costg == pytest.approx(cost, rel=1e-12)
    assert np.abs((Hg - H[:P, :P]) / np.outer(sc[:P], sc[:P])).max() < 1e-10
    assert np.abs((gg - g) / sc).max() < 1e-10 * np.abs(g / sc).max()
    assert (sm["iterations"], sm["num_successful"]) == (sm_o.iterations, sm_o.num_successful)
    assert sm["final_cost"] == pytest.approx(sm_o.final_cost, rel=1e-9)
    assert cv.rel_state_error(wg, wo)["state"] < 1e-6
    np.testing.assert_array_equal(wg.quat[[0, 1, 5, 9]], w0.quat[[0, 1, 5, 9]])
    np.testing.assert_array_equal(wg.pos[[0, 1, 5, 9]], w0.pos[[0, 1, 5, 9]])


@pytest.mark.parametrize("name", ["lm_tiny_seed7", "lm_tiny_rs_seed3020", "lm_tiny_rs_seed3028", "lm_config1_seed1001",
                                  "lm_config2_seed1002", "lm_config3_seed1003", "lm_config3_seed1006",
                                  "lm_long_k34_seed1400"])       # (P = 301: tile Schur kernels inside the envelope + envelope panel Cholesky)
def test_hip_path_reproduces_the_independent_lm_history(cv, golden_dir, name):
    """The device-resident LM (trust region + projected Armijo line search, speculative linearisation) against the committed
    per-iteration fixtures of the independent NumPy restatement of Ceres 1.14's loop (oracle/np_ceres.py, FD Jacobians): the same
    iteration count, accept / reject counts, line-search step counts and termination; cost, radius and state to the fixture's
    finite-difference accuracy."""
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    w = cv.Window.from_dict(d, "w_")
    wf = cv.Window.from_dict(d, "f_")
    term = {"NO_CONVERGENCE": "max-iterations", "CONVERGENCE_GRADIENT": "gradient-tolerance", "CONVERGENCE_PARAMETER": "parameter-tolerance",
            "CONVERGENCE_FUNCTION": "function-tolerance", "CONVERGENCE_RADIUS": "min-radius", "FAILURE": "failure"}[str(d["termination"])]
    with cv.Solver() as s:
        s.set_windows([w])
        sm = s.solve(15)[0]
    assert sm["iterations"] == int(d["iterations"]) and sm["termination"] == term
    assert (sm["num_successful"], sm["num_unsuccessful"]) == (int(d["num_successful"]), int(d["num_unsuccessful"]))
    assert (sm["num_line_search_steps"], sm["num_line_search_reduced"]) == (int(d["num_line_search_steps"]), int(d["num_line_search_reduced"]))
    assert sm["final_cost"] == pytest.approx(float(d["final_cost"]), rel=1e-5)
    assert sm["final_radius"] == pytest.approx(float(d["final_radius"]), rel=1e-2)
    assert cv.rel_state_error(w, wf)["state"] < 2e-5


def test_cxx_sharded_entry_on_the_visible_devices(cv):
    """ctvio_solve_sharded (one host thread + solver handle per device, window w -> device w mod G): a ragged batch through it equals
    the same batch through one handle, window by window, in the caller's order (one device on the test box: G = 1; asking for more
    devices than exist is clamped)."""
    import ctypes as C
    lib = cv.capi.load_library()
    ws = [cv.synth.make_window("config1", seed=1000 + i) for i in range(3)] + [cv.synth.make_window("tiny", seed=5), cv.synth.make_window("config2", seed=1003)]
    with cv.Solver() as s:
        ref = [w.copy() for w in ws]
        s.set_windows(ref)
        sms = s.solve(15)
    keep = []
    arr = (cv.capi.CWindow * len(ws))()
    for i, w in enumerate(ws):
        arr[i] = cv.capi.to_cwindow(w, keep)
    K = sum(w.K for w in ws); F = sum(w.F for w in ws); L = sum(w.L for w in ws)
    q = np.zeros((K, 4)); p = np.zeros((K, 3)); b = np.zeros((F, 6)); r = np.zeros(L); ld = np.zeros(len(ws))
    sm = (cv.capi.Summary * len(ws))()
    for ndev in (0, 1, 8):
        cv.capi.check(lib.ctvio_solve_sharded(None, ndev, len(ws), C.cast(arr, C.c_void_p), 15, C.cast(sm, C.c_void_p),
                                              cv.capi._p(q), cv.capi._p(p), cv.capi._p(b), cv.capi._p(r), cv.capi._p(ld)))
        k = f = l = 0
        for i, w in enumerate(ref):
            assert sm[i].iterations == sms[i]["iterations"] and sm[i].final_cost == pytest.approx(sms[i]["final_cost"], rel=1e-10)
            got = ws[i].copy()
            got.quat, got.pos, got.bias, got.rho, got.ld = q[k:k + w.K], p[k:k + w.K], b[f:f + w.F], r[l:l + w.L], float(ld[i])
            assert cv.rel_state_error(got, w)["state"] < 1e-7, i
            k += w.K; f += w.F; l += w.L
    lib.ctvio_sharded_release()
