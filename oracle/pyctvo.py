"""ctypes binding of the CPU fp64 oracle (oracle/ctvo.c).

TEST INFRASTRUCTURE ONLY (PARITY UNPINNED, see ctvo.h): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product package.
Takes any object with the attributes of ctrl-vio_amd/window.py:Window (duck-typed).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libctvo_oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("ctvo.c", "ctvo.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _LIB


class _CWindow(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("F", C.c_int32), ("L", C.c_int32), ("M", C.c_int32), ("NB", C.c_int32), ("V", C.c_int32),
        ("pn", C.c_int32), ("pnb", C.c_int32),
        ("t0_ns", C.c_int64), ("dt_ns", C.c_int64),
        ("quat", C.c_void_p), ("pos", C.c_void_p), ("bias", C.c_void_p), ("rho", C.c_void_p),
        ("ld", C.c_double), ("ld_lo", C.c_double), ("ld_hi", C.c_double),
        ("fix_ld", C.c_int32), ("lock_bg", C.c_int32), ("lock_ba", C.c_int32), ("fixed_upto", C.c_int32),
        ("q_CI", C.c_double * 4), ("p_CI", C.c_double * 3), ("gravity", C.c_double * 3), ("imu_w", C.c_double * 6),
        ("img_w", C.c_double), ("cauchy_a", C.c_double),
        ("imu_t", C.c_void_p), ("imu_gyro", C.c_void_p), ("imu_acc", C.c_void_p), ("imu_bias", C.c_void_p),
        ("bc_i", C.c_void_p), ("bc_j", C.c_void_p), ("bc_w", C.c_void_p),
        ("v_lm", C.c_void_p), ("v_ti", C.c_void_p), ("v_tj", C.c_void_p), ("v_rowi", C.c_void_p), ("v_rowj", C.c_void_p),
        ("v_pi", C.c_void_p), ("v_pj", C.c_void_p),
        ("pJ0", C.c_void_p), ("pr0", C.c_void_p), ("p_kind", C.c_void_p), ("p_index", C.c_void_p), ("p_off", C.c_void_p),
        ("p_x0", C.c_void_p), ("v_cauchy", C.c_void_p), ("knot_const", C.c_void_p),
    ]


class Summary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("num_successful", C.c_int32), ("num_unsuccessful", C.c_int32),
                ("termination", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_radius", C.c_double), ("cost_hist", C.c_double * 64),
                ("num_line_search_steps", C.c_int32), ("num_line_search_reduced", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        # CTVO_ORACLE_LIB: another build of the same source (tests/test_oracle_asan.py: the -fsanitize=address,undefined target of the Makefile)
        _lib = C.CDLL(os.environ.get("CTVO_ORACLE_LIB") or build())
        _lib.ctvo_cost.restype = C.c_double
        _lib.ctvo_build_normal.restype = C.c_double
        _lib.ctvo_lm_step.restype = C.c_double
        _lib.ctvo_lm_step.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        _lib.ctvo_set_tolerances.argtypes = [C.c_double, C.c_double, C.c_double]
        _lib.ctvo_set_line_search.argtypes = [C.c_int]
        _lib.ctvo_ls_interpolate.restype = C.c_double
        _lib.ctvo_ls_interpolate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a.size else None


class OracleWindow:
    """Holds a C view of a Window; state arrays are shared with (and updated in) the Window."""

    def __init__(self, w):
        w.normalize()
        self.w = w
        self._pJ0_cm = np.asfortranarray(w.pJ0)        # column-major for the C side
        c = _CWindow()
        c.K, c.F, c.L, c.M, c.NB, c.V = w.K, w.F, w.L, w.M, w.NB, w.V
        c.pn, c.pnb = w.pn, int(w.p_kind.shape[0])
        c.t0_ns, c.dt_ns = w.t0_ns, w.dt_ns
        c.quat, c.pos, c.bias, c.rho = _p(w.quat), _p(w.pos), _p(w.bias), _p(w.rho)
        c.ld, c.ld_lo, c.ld_hi = w.ld, w.ld_lo, w.ld_hi
        c.fix_ld, c.lock_bg, c.lock_ba, c.fixed_upto = int(w.fix_ld), int(w.lock_bg), int(w.lock_ba), int(w.fixed_upto)
        c.q_CI[:] = w.q_CI.tolist(); c.p_CI[:] = w.p_CI.tolist(); c.gravity[:] = w.gravity.tolist()
        c.imu_w[:] = w.imu_w.tolist()
        c.img_w, c.cauchy_a = w.img_w, w.cauchy_a
        c.imu_t, c.imu_gyro, c.imu_acc, c.imu_bias = _p(w.imu_t), _p(w.imu_gyro), _p(w.imu_acc), _p(w.imu_bias)
        c.bc_i, c.bc_j, c.bc_w = _p(w.bc_i), _p(w.bc_j), _p(w.bc_w)
        c.v_lm, c.v_ti, c.v_tj = _p(w.v_lm), _p(w.v_ti), _p(w.v_tj)
        c.v_rowi, c.v_rowj, c.v_pi, c.v_pj = _p(w.v_rowi), _p(w.v_rowj), _p(w.v_pi), _p(w.v_pj)
        c.pJ0 = self._pJ0_cm.ctypes.data_as(C.c_void_p) if w.pn else None
        c.pr0, c.p_kind, c.p_index, c.p_off, c.p_x0 = _p(w.pr0), _p(w.p_kind), _p(w.p_index), _p(w.p_off), _p(w.p_x0)
        vc, kc = getattr(w, "v_cauchy", None), getattr(w, "knot_const", None)
        c.v_cauchy = _p(vc) if vc is not None else None
        c.knot_const = _p(kc) if kc is not None else None
        self.c = c

    def _sync_in(self):
        self.c.ld = self.w.ld

    def _sync_out(self):
        self.w.ld = float(self.c.ld)

    def imu_block(self, m, jac=True):
        self._sync_in()
        r = np.zeros(6); J = np.zeros((6, 30)); s = C.c_int32()
        lib().ctvo_imu_block(C.byref(self.c), int(m), _p(r), _p(J) if jac else None, C.byref(s))
        return r, (J if jac else None), s.value

    def visual_block(self, v, jac=True):
        self._sync_in()
        r = np.zeros(2); J = np.zeros((2, 50)); si, sj = C.c_int32(), C.c_int32()
        lib().ctvo_visual_block(C.byref(self.c), int(v), _p(r), _p(J) if jac else None, C.byref(si), C.byref(sj))
        return r, (J if jac else None), si.value, sj.value

    def bias_block(self, b):
        r = np.zeros(6); d = np.zeros(6)
        lib().ctvo_bias_block(C.byref(self.c), int(b), _p(r), _p(d))
        return r, d

    def prior_residual(self):
        self._sync_in()
        n = self.w.pn
        r = np.zeros(n); dx = np.zeros(n)
        if n:
            lib().ctvo_prior_residual(C.byref(self.c), _p(r), _p(dx))
        return r, dx

    def cost(self):
        self._sync_in()
        return float(lib().ctvo_cost(C.byref(self.c)))

    def build_normal(self):
        self._sync_in()
        N = self.w.N
        H = np.zeros((N, N)); g = np.zeros(N)
        cost = float(lib().ctvo_build_normal(C.byref(self.c), _p(H), _p(g)))
        return H, g, cost

    def active_mask(self):
        a = np.zeros(self.w.N, np.uint8)
        lib().ctvo_active_mask(C.byref(self.c), _p(a))
        return a.astype(bool)

    def lm_step(self, mu=1e4, use_schur=True):
        self._sync_in()
        d = np.zeros(self.w.N)
        mc = float(lib().ctvo_lm_step(C.byref(self.c), float(mu), int(use_schur), _p(d)))
        return d, mc

    def plus(self, delta):
        self._sync_in()
        d = np.ascontiguousarray(delta, np.float64)
        lib().ctvo_plus(C.byref(self.c), _p(d))
        self._sync_out()

    def solve(self, max_iters=15, use_schur=True):
        self._sync_in()
        sm = Summary()
        lib().ctvo_solve(C.byref(self.c), int(max_iters), int(use_schur), C.byref(sm))
        self._sync_out()
        return sm

    def marginalize(self, role, eps=1e-8):
        """role[N]: 1 marginalise, 0 keep, -1 not involved.  Returns kept indices, J0 (n x n), r0 (n)."""
        self._sync_in()
        role = np.ascontiguousarray(role, np.int8)
        N = self.w.N
        kept = np.zeros(N, np.int32); J0 = np.zeros(N * N); r0 = np.zeros(N)
        lib().ctvo_marginalize.restype = C.c_int
        lib().ctvo_marginalize.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        n = lib().ctvo_marginalize(C.byref(self.c), _p(role), float(eps), _p(kept), _p(J0), _p(r0))
        return kept[:n].copy(), J0[: n * n].reshape(n, n).copy(), r0[:n].copy()

    def spline_eval(self, t_ns):
        t = np.ascontiguousarray(t_ns, np.int64)
        n = t.shape[0]
        pose = np.zeros((n, 7)); vel = np.zeros((n, 3)); om = np.zeros((n, 3)); acc = np.zeros((n, 3))
        lib().ctvo_spline_eval(C.byref(self.c), n, _p(t), _p(pose), _p(vel), _p(om), _p(acc))
        return pose, vel, om, acc


def set_jacobian_noise(imu_rel=0.0, vis_rel=0.0):
    lib().ctvo_set_jacobian_noise.argtypes = [C.c_double, C.c_double]
    lib().ctvo_set_jacobian_noise(float(imu_rel), float(vis_rel))


def set_product_rounding(on):
    lib().ctvo_set_product_rounding(int(bool(on)))


def gauge_restore(quat, pos, knot, q0, t0):
    """In place on (K,4) / (K,3) fp64 arrays: reference double2vector (trajectory_manager.cpp:485-516)."""
    quat = np.ascontiguousarray(quat, np.float64); pos = np.ascontiguousarray(pos, np.float64)
    q0 = np.ascontiguousarray(q0, np.float64); t0 = np.ascontiguousarray(t0, np.float64)
    lib().ctvo_gauge_restore(int(quat.shape[0]), _p(quat), _p(pos), int(knot), _p(q0), _p(t0))
    return quat, pos


def so3_exp(w):
    q = np.zeros(4); w = np.ascontiguousarray(w, np.float64)
    lib().ctvo_so3_exp(_p(w), _p(q)); return q


def so3_log(q):
    w = np.zeros(3); q = np.ascontiguousarray(q, np.float64)
    lib().ctvo_so3_log(_p(q), _p(w)); return w


def so3_Jr(phi):
    J = np.zeros((3, 3)); phi = np.ascontiguousarray(phi, np.float64)
    lib().ctvo_so3_Jr(_p(phi), _p(J)); return J


def so3_Jr_inv(phi):
    J = np.zeros((3, 3)); phi = np.ascontiguousarray(phi, np.float64)
    lib().ctvo_so3_Jr_inv(_p(phi), _p(J)); return J


def set_tolerances(ftol=1e-6, gtol=1e-10, ptol=1e-8):
    """Test hook; defaults are Ceres' (function, gradient, parameter) tolerances."""
    lib().ctvo_set_tolerances(float(ftol), float(gtol), float(ptol))


def set_line_search(on=True):
    """Test hook: Ceres' projected Armijo line search of bounded problems (default on)."""
    lib().ctvo_set_line_search(int(bool(on)))


def ls_interpolate(x, value, gradient, x_min, x_max):
    """Minimiser over [x_min, x_max] of the polynomial interpolating (x, value, gradient) samples (2 -> cubic, 3 -> quintic)."""
    x = np.ascontiguousarray(x, np.float64); v = np.ascontiguousarray(value, np.float64); g = np.ascontiguousarray(gradient, np.float64)
    return float(lib().ctvo_ls_interpolate(int(x.shape[0]), _p(x), _p(v), _p(g), float(x_min), float(x_max)))
