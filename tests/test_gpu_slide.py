"""GPU: three consecutive sliding windows -- solve -> 4-DoF gauge restore -> marginalise the oldest frame -> next solve with
the new prior (tests/slide_helpers.py; reference trajectory_manager.cpp:122-286, 317-516) -- through the C ABI and through the
reference-shaped C++ adaptor (PrepareMarginalizationInfo / SaveMarginalizationInfo / AddMarginalizationFactor), against the
CPU oracle running the same protocol."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)


@pytest.fixture(scope="module")
def slide_reference(oracle):
    import slide_helpers as sh
    world = sh.make_world()
    st, rec = sh.run_slide(world, sh.OracleBackend())
    return sh, world, st, rec


def test_three_windows_through_the_c_abi(cv, slide_reference):
    sh, world, st_o, rec_o = slide_reference
    st_g, rec_g = sh.run_slide(world, sh.DeviceBackend("fp64"))
    for a, b in zip(rec_g, rec_o):
        assert a["iterations"] == b["iterations"] and a.get("n_keep") == b.get("n_keep"), (a, b)
        # (the prior's constant r0^T r0 depends on which noise-level eigenvalues of the rank-deficient A' fall on which side of
        #  eps = 1e-8 -- one of them differs between the two eigen-solvers here, as it would against Eigen's; the quadratic
        #  form, hence the minimiser, is unaffected: the states below agree to 1e-6)
        assert a["final_cost"] == pytest.approx(b["final_cost"], rel=1e-3)
    err = sh.state_error(st_g, st_o)
    assert max(err.values()) < 1e-6, err          # asked for: 1e-6 (fp64) -- and the product path IS the fp64 path (contract 1e-4)


def test_three_windows_through_the_cpp_adaptor(cv, slide_reference, tmp_path):
    sh, world, st_o, rec_o = slide_reference
    from test_gpu_adaptor import _dump
    exe = str(tmp_path / "slide_demo")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "slide_demo.cpp"),
                           "-L", os.path.join(ROOT, "ctrl-vio_amd"), "-lctvio", "-Wl,-rpath," + os.path.join(ROOT, "ctrl-vio_amd"), "-o", exe])
    _dump(world, str(tmp_path / "world.txt"))
    out = subprocess.check_output([exe, str(tmp_path / "world.txt"), str(tmp_path / "out.txt")], text=True)
    assert out.count("ctvio: iterations") == 3 and "ResidualSummary" in out and "- Image: num =" in out, out
    for k, r in enumerate(rec_o[:-1]):
        assert f"prior {k}: n = {r['n_keep']}," in out, out
    arr = np.array(open(tmp_path / "out.txt").read().split(), float)
    K, F, L = world.K, world.F, world.L
    kn = arr[:7 * K].reshape(K, 7)
    st = sh.State(world)
    st.quat, st.pos = kn[:, :4].copy(), kn[:, 4:].copy()
    st.bias = arr[7 * K:7 * K + 6 * F].reshape(F, 6).copy()
    st.rho = arr[7 * K + 6 * F:7 * K + 6 * F + L].copy()
    st.ld = float(arr[7 * K + 6 * F + L])
    err = sh.state_error(st, st_o)
    assert max(err.values()) < 1e-6, err
    # the Trajectory query surface of the adaptor (GetCameraPose / GetIMUState / getLastKnot / Get-SetDataStartTime), evaluated on the
    # device, against the oracle's spline evaluation of the same knots composed with the camera extrinsic in numpy
    tail = arr[7 * K + 6 * F + L + 1:]
    tq, t_start = int(tail[0]), int(tail[1])
    cam, imu_st, last = tail[2:9], tail[9:19], tail[19:26]
    assert t_start == 12345
    import pyctvo
    wq = world.copy(); wq.quat, wq.pos = st.quat.copy(), st.pos.copy()
    pose, vel, _, _ = pyctvo.OracleWindow(wq).spline_eval(np.array([tq], np.int64))
    np.testing.assert_allclose(imu_st[:7], pose[0], rtol=0, atol=1e-11)
    np.testing.assert_allclose(imu_st[7:], vel[0], rtol=0, atol=1e-10)
    from scipy.spatial.transform import Rotation as R
    Rg = R.from_quat(pose[0, 3:7])
    qc = (Rg * R.from_quat(world.q_CI / np.linalg.norm(world.q_CI))).as_quat()
    pc = pose[0, :3] + Rg.apply(world.p_CI)
    np.testing.assert_allclose(cam[:3], pc, rtol=0, atol=1e-11)
    assert min(np.abs(cam[3:] - qc).max(), np.abs(cam[3:] + qc).max()) < 1e-11
    np.testing.assert_allclose(last, np.concatenate([st.pos[-1], st.quat[-1]]), rtol=0, atol=0)


def test_residual_summary_matches_oracle(cv, oracle):
    """ctvio_residual_summary (reference ResidualSummary, trajectory_estimator.cpp:36-67): sums of |r_i| per factor type."""
    w = cv.synth.make_window("config1", seed=1003)
    o = oracle.OracleWindow(w.copy())
    r_imu = np.abs(np.array([o.imu_block(m, jac=False)[0] for m in range(w.M)])).sum(0)
    r_vis = np.abs(np.array([o.visual_block(v, jac=False)[0] for v in range(w.V)])).sum(0)
    r_bias = np.abs(np.array([o.bias_block(b)[0] for b in range(w.NB)])).sum(0)
    r_prior = np.abs(o.prior_residual()[0])
    for prec in ("fp64",):
        with cv.Solver(precision=prec) as s:
            s.set_windows([w.copy()])
            rs = s.residual_summary(0)
        assert rs["imu"][1] == w.M and rs["bias"][1] == w.NB and rs["image"][1] == w.V and rs["prior"][1] == 1
        np.testing.assert_allclose(rs["imu"][0], r_imu, rtol=1e-9)
        np.testing.assert_allclose(rs["bias"][0], r_bias, rtol=1e-9)
        np.testing.assert_allclose(rs["image"][0], r_vis, rtol=1e-9)
        np.testing.assert_allclose(rs["prior"][0], r_prior, rtol=1e-9, atol=1e-9)


def test_marginalize_batch_on_device(cv, slide_reference):
    """ctvio_marginalize_batch: 256 marginalisation windows (the MARGIN_OLD factor set of the first slide step, m = 26 dropped /
    n = 91 kept unknowns) in one launch -- Schur elimination and both eigendecompositions on the device (parallel Jacobi in
    LDS, csrc/marg_device.hpp) -- against the oracle, and under 2 ms per window."""
    import time
    sh, world, st_o, rec_o = slide_reference
    st = sh.State(world)
    w, info = sh.window_of(world, st, 0, sh.initial_prior(world))
    m, role = sh.marg_window_of(world, st, 0, w, info)
    import pyctvo
    ko, Jo, ro = pyctvo.OracleWindow(m.copy()).marginalize(role, 1e-8)
    Ho, go = Jo.T @ Jo, Jo.T @ ro
    nb = 256
    with cv.Solver() as s:
        s.set_windows([m.copy() for _ in range(nb)])
        s.marginalize_batch([role] * nb)                     # warm-up (allocations, kernel load)
        t0 = time.perf_counter()
        res = s.marginalize_batch([role] * nb)
        dt = time.perf_counter() - t0
        one = s.marginalize(3, role)                         # the single-window entry takes the same device path
    assert dt / nb < 2e-3, dt / nb
    for kept, J0, r0 in (res[0], res[nb // 2], res[-1], one):
        assert np.array_equal(kept, ko)
        assert np.abs(J0.T @ J0 - Ho).max() <= 1e-7 * np.abs(Ho).max()
        assert np.abs(J0.T @ r0 - go).max() <= 1e-7 * np.abs(go).max()
    print(f"marginalize_batch: {1e3 * dt / nb:.3f} ms per window ({nb} windows, {1e3 * dt:.1f} ms)")


def test_marginalize_host_leg_is_reported(cv, slide_reference, monkeypatch):
    """ctvio_marginalize has a host leg (csrc/marginalize.hpp: windows beyond the device eigen-solver's size or whose Jacobi sweeps stalled;
    reference counterpart marginalization_factor.cpp:178-265).  It must not be silent: the reference's drop set (m = 26 / n = 91) factors on
    the device and the handle says so; forced through the diagnostic switch (read once, in ctvio_create) the handle reports the host leg, the
    batch entry (which has none) clears the flag, and both legs give the same quadratic form."""
    sh, world, st_o, rec_o = slide_reference
    st = sh.State(world)
    w, info = sh.window_of(world, st, 0, sh.initial_prior(world))
    m, role = sh.marg_window_of(world, st, 0, w, info)
    with cv.Solver() as s:
        s.set_windows([m.copy()])
        kd, Jd, rd = s.marginalize(0, role)
        assert not s.marginalize_ran_on_host()
    monkeypatch.setenv("CTVIO_MARG_HOST", "1")
    with cv.Solver() as s:
        s.set_windows([m.copy()])
        kh, Jh, rh = s.marginalize(0, role)
        assert s.marginalize_ran_on_host()
        s.marginalize_batch([role])
        assert not s.marginalize_ran_on_host()
    assert np.array_equal(kd, kh)
    # (measured, tests/studies/r6_marg_legs.py: device vs oracle 9e-13 / 4e-12; host vs oracle 1.6e-6 / 7e-9 -- the marginalised block is rank
    #  deficient, and the Householder / QL eigenvalues of its noise directions land on the other side of eps = 1e-8 for three of them: rank 54
    #  instead of 51.  The reference's own SelfAdjointEigenSolver has the same freedom, marginalization_factor.cpp:206-214.)
    Hh = Jh.T @ Jh
    assert np.abs(Jd.T @ Jd - Hh).max() <= 1e-5 * np.abs(Hh).max()
    assert np.abs(Jd.T @ rd - Jh.T @ rh).max() <= 1e-6 * np.abs(Jh.T @ rh).max()
