"""CPU tests of the prior construction: the oracle's marginalisation against an independent NumPy
restatement of the reference's algebra, and the library's host algebra (csrc/marginalize.hpp, built with g++) against both."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
cv = importlib.import_module("ctrl-vio_amd")
import pyctvo  # noqa: E402


def numpy_marginalize(H, g, role, eps):
    """MarginalizationInfo::marginalize (marginalization_factor.cpp:189-265) with numpy.linalg.eigh."""
    im = np.where(role == 1)[0]; ik = np.where(role == 0)[0]
    Amm = 0.5 * (H[np.ix_(im, im)] + H[np.ix_(im, im)].T)
    e, V = np.linalg.eigh(Amm)
    inv = V @ np.diag(np.where(e > eps, 1.0 / np.where(e > eps, e, 1.0), 0.0)) @ V.T
    A = H[np.ix_(ik, ik)] - H[np.ix_(ik, im)] @ inv @ H[np.ix_(im, ik)]
    b = g[ik] - H[np.ix_(ik, im)] @ inv @ g[im]
    e2, V2 = np.linalg.eigh(0.5 * (A + A.T))
    S = np.where(e2 > eps, e2, 0.0)
    J0 = np.diag(np.sqrt(S)) @ V2.T
    r0 = np.diag(np.where(S > 0, 1.0 / np.sqrt(np.where(S > 0, S, 1.0)), 0.0)) @ V2.T @ b
    return ik, J0, r0


def roles(w, rng=None):
    """Drop the two oldest knots, the oldest bias state and the landmarks anchored in the oldest frame (here: the first
    half); knots / landmarks no dropped factor touches would be -1 in a real marginalisation window."""
    role = np.zeros(w.N, np.int8)
    role[:12] = 1
    role[6 * w.K:6 * w.K + 6] = 1
    role[w.P:w.P + w.L // 2] = 1
    return role


@pytest.fixture(scope="module")
def hostlib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hm") / "libhostmarg.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "host_marginalize_check.cpp")])
    lib = C.CDLL(so)
    lib.hm_marginalize_dense.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hm_sym_eig.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def invariants_close(Ja, ra, Jb, rb, tol):
    """J0 is unique only up to the sign / rotation of (near-)degenerate eigenvectors: compare what the next window uses,
    J0^T J0 (the prior Hessian), J0^T r0 (its gradient) and |r0|^2 (its cost)."""
    Ha, Hb = Ja.T @ Ja, Jb.T @ Jb
    assert np.abs(Ha - Hb).max() <= tol * np.abs(Hb).max()
    ga, gb = Ja.T @ ra, Jb.T @ rb
    assert np.abs(ga - gb).max() <= tol * max(np.abs(gb).max(), 1e-30)
    assert abs(ra @ ra - rb @ rb) <= tol * max(rb @ rb, 1e-30)


@pytest.mark.parametrize("cfg,seed", [("tiny", 7), ("config1", 1000)])
def test_oracle_marginalize_against_numpy(cfg, seed):
    w = cv.synth.make_window(cfg, seed=seed)
    ow = pyctvo.OracleWindow(w.copy())
    H, g, _ = ow.build_normal()
    role = roles(w)
    kept, J0, r0 = ow.marginalize(role, 1e-8)
    ik, Jn, rn = numpy_marginalize(H, g, role, 1e-8)
    assert np.array_equal(kept, ik)
    # (the numerical rank is not compared: the reference's eps = 1e-8 lies far below the rounding noise of the eigenvalues,
    #  ~1e-16 * |A| ~ 1e-3, so which of the near-null gauge directions survive is arbitrary -- and immaterial below)
    invariants_close(J0, r0, Jn, rn, 1e-7)


def test_library_host_algebra(hostlib):
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 40, 133):     # the eigen-solver alone, incl. a rank-deficient matrix
        B = rng.normal(size=(n, max(n - 3, 1)))
        A = B @ B.T
        d = np.zeros(n); V = np.zeros((n, n))
        hostlib.hm_sym_eig(n, A.ctypes.data, d.ctypes.data, V.ctypes.data)
        np.testing.assert_allclose(d, np.linalg.eigvalsh(A), atol=1e-11 * max(np.abs(A).max(), 1.0))
        np.testing.assert_allclose(V @ np.diag(d) @ V.T, A, atol=1e-11 * max(np.abs(A).max(), 1.0))
        np.testing.assert_allclose(V.T @ V, np.eye(n), atol=1e-12)
    w = cv.synth.make_window("config1", seed=1001)
    ow = pyctvo.OracleWindow(w.copy())
    H, g, _ = ow.build_normal()
    role = roles(w)
    N = w.N
    kept = np.zeros(N, np.int32); J0 = np.zeros(N * N); r0 = np.zeros(N)
    Hc = np.ascontiguousarray(H)
    n = hostlib.hm_marginalize_dense(N, Hc.ctypes.data, g.ctypes.data, role.ctypes.data, 1e-8, kept.ctypes.data, J0.ctypes.data, r0.ctypes.data)
    ko, Jo, ro = ow.marginalize(role, 1e-8)
    assert n == len(ko) and np.array_equal(kept[:n], ko)
    # 1e-7: which near-null directions (eigenvalues at the rounding-noise level, see above) pass the eps test differs between
    # eigen-solvers; their contribution to J0^T r0 is of that order
    invariants_close(J0[:n * n].reshape(n, n), r0[:n], Jo, ro, 1e-7)
    ik, Jn, rn = numpy_marginalize(H, g, role, 1e-8)
    invariants_close(J0[:n * n].reshape(n, n), r0[:n], Jn, rn, 1e-7)


def test_prior_chain_consistency_oracle():
    """End-to-end meaning of the prior (oracle only, CPU): marginalise the landmarks of a dropped set D out of their own
    factors, hand (J0, r0) to the window of the remaining factors R as its prior -- the Gauss-Newton step of R + prior must
    equal the step of the full window on every unknown they share (exact for the linearised system: Schur complement).
    The gauge is fixed in both solves (first four knots constant, as InitTrajectory does); the marginalisation window is free."""
    from chain_helpers import chain_case, prior_arrays
    w, wD, wR, mapR = chain_case("config1", 1000)
    oD = pyctvo.OracleWindow(wD.copy())
    H, g, _ = oD.build_normal()
    role = np.where(np.arange(wD.N) >= wD.P, 1, np.where(np.diag(H) > 0, 0, -1)).astype(np.int8)
    kept, J0, r0 = oD.marginalize(role, 1e-8)
    wR.pJ0, wR.pr0, wR.p_kind, wR.p_index, wR.p_off, wR.p_x0 = prior_arrays(wR, kept, J0, r0)
    wR.normalize()
    w.fixed_upto = 3
    wR.fixed_upto = 3
    d_full, _ = pyctvo.OracleWindow(w.copy()).lm_step(1e16)
    d_red, _ = pyctvo.OracleWindow(wR.copy()).lm_step(1e16)
    P = w.P
    assert np.abs(d_red[:P] - d_full[:P]).max() < 1e-8 * np.abs(d_full[:P]).max()
    assert np.abs(d_red[P:] - d_full[P + mapR]).max() < 1e-8 * np.abs(d_full[P:]).max()


def test_marginalize_edge_cases(hostlib):
    """m = 0 (nothing to drop: the prior is the factorisation of A itself), uninvolved unknowns (-1) ignored, an exactly
    singular Amm (a marginalised unknown no factor touches) handled by the eps cut -- oracle and library alike."""
    rng = np.random.default_rng(9)
    N = 14
    B = rng.normal(size=(N, N + 3))
    H = B @ B.T
    g = rng.normal(size=N)

    def lib_marg(Hm, gv, role, eps=1e-8):
        kept = np.zeros(N, np.int32); J0 = np.zeros(N * N); r0 = np.zeros(N)
        Hc = np.ascontiguousarray(Hm); role = np.ascontiguousarray(role, np.int8)
        n = hostlib.hm_marginalize_dense(N, Hc.ctypes.data, gv.ctypes.data, role.ctypes.data, eps, kept.ctypes.data, J0.ctypes.data, r0.ctypes.data)
        return kept[:n], J0[:n * n].reshape(n, n), r0[:n]

    role = np.zeros(N, np.int8)                                   # m = 0
    k, J0, r0 = lib_marg(H, g, role)
    assert k.tolist() == list(range(N))
    np.testing.assert_allclose(J0.T @ J0, H, atol=1e-10 * np.abs(H).max())
    np.testing.assert_allclose(J0.T @ r0, g, atol=1e-9)
    role = np.array([1, 1, 0, 0, -1, 0, -1, 1, 0, 0, 0, -1, 0, 1], np.int8)   # mixed roles
    k, J0, r0 = lib_marg(H, g, role)
    ik, Jn, rn = numpy_marginalize(H, g, role, 1e-8)
    assert np.array_equal(k, ik)
    invariants_close(J0, r0, Jn, rn, 1e-9)
    H2 = H.copy(); H2[0, :] = 0; H2[:, 0] = 0                       # marginalised unknown 0 untouched by any factor: Amm singular
    g2 = g.copy(); g2[0] = 0
    k, J0, r0 = lib_marg(H2, g2, role)
    ik, Jn, rn = numpy_marginalize(H2, g2, role, 1e-8)
    invariants_close(J0, r0, Jn, rn, 1e-9)
    assert np.all(np.isfinite(J0)) and np.all(np.isfinite(r0))
