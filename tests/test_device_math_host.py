"""CPU: the device math headers (ctrl-vio_amd/csrc/so3.hpp, factors.hpp) compiled with g++ for the test
only (tests/host_math_check.cpp) and compared block by block with the fp64 oracle (fp64 like the kernels: agreement to rounding)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hm():
    out = os.path.join(HERE, "_build", "libhostmath.so")
    src = os.path.join(HERE, "host_math_check.cpp")
    hdrs = [os.path.join(HERE, "..", "ctrl-vio_amd", "csrc", f) for f in ("so3.hpp", "factors.hpp")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or any(os.path.getmtime(f) > os.path.getmtime(out) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, src])
    lib = C.CDLL(out)
    lib.hm_visual_eval.restype = C.c_double
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _d(x):
    return C.c_double(float(x))


def _imu(hm, w, m):
    st = int(w.imu_t[m]) - w.t0_ns
    s, u = st // w.dt_ns, (st % w.dt_ns) / w.dt_ns
    q = np.ascontiguousarray(w.quat[s:s + 4]); p = np.ascontiguousarray(w.pos[s:s + 4])
    r = np.zeros(6); J = np.zeros((6, 30))
    b = np.ascontiguousarray(w.bias[w.imu_bias[m]])
    hm.hm_imu_eval(_p(q), _p(p), _d(u), _d(1e9 / w.dt_ns), _p(w.gravity), _p(b),
                   _p(np.ascontiguousarray(w.imu_gyro[m])), _p(np.ascontiguousarray(w.imu_acc[m])), _p(w.imu_w), _p(r), _p(J))
    return r, J


def _vis(hm, w, v, small):
    ld_ns = int(w.ld * 1e9)
    ti = int(w.v_ti[v]) + int(w.v_rowi[v]) * ld_ns - w.t0_ns
    tj = int(w.v_tj[v]) + int(w.v_rowj[v]) * ld_ns - w.t0_ns
    si, ui = ti // w.dt_ns, (ti % w.dt_ns) / w.dt_ns
    sj, uj = tj // w.dt_ns, (tj % w.dt_ns) / w.dt_ns
    qi = np.ascontiguousarray(w.quat[si:si + 4]); pi = np.ascontiguousarray(w.pos[si:si + 4])
    qj = np.ascontiguousarray(w.quat[sj:sj + 4]); pj = np.ascontiguousarray(w.pos[sj:sj + 4])
    obs = np.array([w.v_pi[v, 0], w.v_pi[v, 1], w.v_pj[v, 0], w.v_pj[v, 1]])
    r = np.zeros(2); J = np.zeros((2, 50))
    cost = hm.hm_visual_eval(int(small), _p(qi), _p(pi), _p(qj), _p(pj), _d(ui), _d(uj), _d(1e9 / w.dt_ns), _p(w.q_CI), _p(w.p_CI),
                             _d(w.img_w), _d(w.cauchy_a), _p(obs), _d(w.v_rowi[v]), _d(w.v_rowj[v]), _d(w.rho[w.v_lm[v]]), _p(r), _p(J))
    return r, J, cost


def _corrected(w, r, J):
    s = float(r @ r); b2 = w.cauchy_a ** 2
    rho1 = 1.0 / (1.0 + s / b2)
    return np.sqrt(rho1) * r, np.sqrt(rho1) * J, 0.5 * b2 * np.log1p(s / b2)


def test_imu_block_matches_oracle(cv, oracle, hm):
    rtol = 1e-11
    w = cv.synth.make_window("config1", seed=1003)
    o = oracle.OracleWindow(w)
    for m in range(0, w.M, 37):
        r0, J0, _ = o.imu_block(m)
        r, J = _imu(hm, w, m)
        # whitened residuals are O(1e2) at the initial guess; compare on the scale of the block
        assert np.abs(r - r0).max() <= rtol * max(np.abs(r0).max(), 1.0)
        assert np.abs(J - J0).max() <= rtol * np.abs(J0).max()


@pytest.mark.parametrize("small", [0, 1])
def test_visual_block_matches_oracle(cv, oracle, hm, small):
    """The factored visual block (anchor record + j-end block, factors.hpp) composed back into the 2 x 50 Jacobian: the general
    form and the series-only form (|knot-pair log| < 0.5 rad) against the oracle's image_feature_factor.h restatement."""
    rtol = 1e-10
    w = cv.synth.make_window("config1", seed=1003)
    w.ld = 1.7e-5
    o = oracle.OracleWindow(w)
    for v in range(0, w.V, 7):
        r0, J0, si, sj = o.visual_block(v)
        rc, Jc, cost0 = _corrected(w, r0, J0)
        r, J, cost = _vis(hm, w, v, small)
        assert np.abs(r - rc).max() <= rtol * max(np.abs(rc).max(), 1.0)
        colscale = np.maximum(np.abs(Jc).max(0), 1e-3 * np.abs(Jc).max())
        assert (np.abs(J - Jc) / colscale).max() <= rtol
        assert cost == pytest.approx(cost0, rel=1e-10, abs=1e-12)


def test_lie_primitives(oracle, hm):
    tol = 1e-13
    rng = np.random.default_rng(5)
    for scale in (1e-7, 1e-4, 1e-2, 0.3, 0.99, 1.01, 2.5):
        phi = rng.normal(size=3); phi *= scale / np.linalg.norm(phi)
        q = np.zeros(4); Jr = np.zeros((3, 3)); Ji = np.zeros((3, 3)); lg = np.zeros(3)
        hm.hm_so3(_p(phi), _p(q), _p(Jr), _p(Ji), _p(lg))
        np.testing.assert_allclose(q, oracle.so3_exp(phi), atol=tol)
        np.testing.assert_allclose(Jr, oracle.so3_Jr(phi), atol=tol * 2)
        np.testing.assert_allclose(Ji, oracle.so3_Jr_inv(phi), atol=tol * 4)
        np.testing.assert_allclose(lg, phi, atol=tol * max(1.0, scale) * 4)
        np.testing.assert_allclose(Jr @ Ji, np.eye(3), atol=tol * 8)
