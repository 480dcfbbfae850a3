"""Independent NumPy/SciPy restatement of the Ctrl-VIO residual blocks (RESIDUALS ONLY).

TEST INFRASTRUCTURE (PARITY UNPINNED, see oracle/ctvo.h).  Purpose: pin oracle/ctvo.c with a
second, differently-written implementation -- rotations go through scipy.spatial.transform
(rotation vectors / matrices), Jacobians come from central finite differences, and the
minimiser is scipy.optimize.least_squares instead of a hand-written LM.  Run only in the build
container by tests/golden/make_golden.py, which commits its outputs as fixtures.

Formulas restated from the reference (paths relative to /root/reference):
  IMU      src/estimator/factor/analytic_diff/trajectory_value_factor.h:165-171
           + split_spline_view.h:85-155 (omega recursion, R^T (p'' + g))
  visual   src/estimator/factor/analytic_diff/image_feature_factor.h:70-163,267
  bias     trajectory_value_factor.h:55-60
  prior    marginalization_factor.cpp:333-355
  Cauchy   ceres::CauchyLoss (rho(s) = a^2 log(1 + s/a^2)), trajectory_estimator.cpp:320-322
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation as Rot

_MB = np.array([[1, -3, 3, -1], [4, 0, -6, 3], [1, 3, 3, -3], [0, 0, 0, 1]], float) / 6
_MC = np.array([[6, 0, 0, 0], [5, 3, -3, 1], [1, 3, 3, -2], [0, 0, 0, 1]], float) / 6


def _powers(u, d):
    z, o = np.zeros_like(u), np.ones_like(u)
    return [np.stack([o, u, u ** 2, u ** 3], -1), np.stack([z, o, 2 * u, 3 * u ** 2], -1),
            np.stack([z, z, 2 * o, 6 * u], -1)][d]


def _seg(t, t0, dt):
    t = np.asarray(t)
    if t.dtype.kind == "f":                      # continuous time (only for differentiating wrt line delay)
        x = (t - t0) / float(dt)
        s = np.floor(x).astype(np.int64)
        return s, x - s
    st = t.astype(np.int64) - t0
    return st // dt, (st % dt) / float(dt)


def _rot_eval(quat, s, u, idt, want_omega=False):
    """R(t) as Rotation (and body rate) from the cumulative form R_s * prod exp(lam_i d_i)."""
    lam = _powers(u, 0) @ _MC.T
    dlam = (_powers(u, 1) @ _MC.T) * idt
    R = Rot.from_quat(quat[s])
    om = np.zeros((len(s), 3))
    for i in range(3):
        d = (Rot.from_quat(quat[s + i]).inv() * Rot.from_quat(quat[s + i + 1])).as_rotvec()
        A = Rot.from_rotvec(d * lam[:, i + 1:i + 2])
        R = R * A
        om = A.inv().apply(om) + d * dlam[:, i + 1:i + 2]
    return (R, om) if want_omega else R


def _pos_eval(pos, s, u, idt, d):
    c = (_powers(u, d) @ _MB.T) * idt ** d
    return np.einsum("ni,nij->nj", c, pos[s[:, None] + np.arange(4)[None]])


def residuals(w, robust_sqrt=False, trunc_ld=True):
    """All residual blocks of Window w -> dict of arrays.  With robust_sqrt, visual residuals are
    rescaled so that 1/2 |r|^2 == 1/2 rho(|r|^2) (same cost, plain least squares)."""
    idt = 1e9 / w.dt_ns
    out = {}
    # IMU
    s, u = _seg(w.imu_t, w.t0_ns, w.dt_ns)
    R, om = _rot_eval(w.quat, s, u, idt, True)
    acc = R.inv().apply(_pos_eval(w.pos, s, u, idt, 2) + w.gravity)
    b = w.bias[w.imu_bias]
    out["imu"] = np.concatenate([om - (w.imu_gyro - b[:, :3]), acc - (w.imu_acc - b[:, 3:])], 1) * w.imu_w
    # visual
    if w.V:
        if trunc_ld:                             # reference image_feature_factor.h:72: int64_t(l_delay * 1e9)
            ld_ns = np.int64(w.ld * 1e9)
            ti = w.v_ti + w.v_rowi.astype(np.int64) * ld_ns
            tj = w.v_tj + w.v_rowj.astype(np.int64) * ld_ns
        else:
            ti = w.v_ti + w.v_rowi * (w.ld * 1e9)
            tj = w.v_tj + w.v_rowj * (w.ld * 1e9)
        si, ui = _seg(ti, w.t0_ns, w.dt_ns)
        sj, uj = _seg(tj, w.t0_ns, w.dt_ns)
        Ri, Rj = _rot_eval(w.quat, si, ui, idt), _rot_eval(w.quat, sj, uj, idt)
        pi, pj = _pos_eval(w.pos, si, ui, idt, 0), _pos_eval(w.pos, sj, uj, idt, 0)
        RCI = Rot.from_quat(w.q_CI)
        x_ci = np.concatenate([w.v_pi, np.ones((w.V, 1))], 1) / w.rho[w.v_lm][:, None]
        p_G = Ri.apply(RCI.apply(x_ci) + w.p_CI) + pi
        x_j = RCI.inv().apply(Rj.inv().apply(p_G - pj) - w.p_CI)
        rv = w.img_w * (x_j[:, :2] / x_j[:, 2:3] - w.v_pj)
        if robust_sqrt and w.cauchy_a > 0:
            sq = np.sum(rv * rv, 1)
            b2 = w.cauchy_a ** 2
            rho = b2 * np.log1p(sq / b2)
            rv = rv * np.sqrt(np.where(sq > 0, rho / np.where(sq > 0, sq, 1), 1.0))[:, None]
        out["vis"] = rv
    else:
        out["vis"] = np.zeros((0, 2))
    # bias chain
    out["bias"] = (w.bias[w.bc_j] - w.bias[w.bc_i]) * w.bc_w if w.NB else np.zeros((0, 6))
    # prior
    if w.pn:
        dx = np.zeros(w.pn)
        for k, (kind, idx, off) in enumerate(zip(w.p_kind, w.p_index, w.p_off)):
            x0 = w.p_x0[k]
            if kind == 0:
                dq = (Rot.from_quat(x0).inv() * Rot.from_quat(w.quat[idx])).as_quat()
                if dq[3] < 0:
                    dq = -dq
                dx[off:off + 3] = 2 * dq[:3]
            elif kind == 1:
                dx[off:off + 3] = w.pos[idx] - x0[:3]
            elif kind == 2:
                dx[off:off + 3] = w.bias[idx, :3] - x0[:3]
            elif kind == 3:
                dx[off:off + 3] = w.bias[idx, 3:] - x0[:3]
            else:
                dx[off] = w.ld - x0[0]
        out["prior"] = w.pr0 + w.pJ0 @ dx
    else:
        out["prior"] = np.zeros(0)
    return out


def cost(w):
    r = residuals(w)
    c = 0.5 * np.sum(r["imu"] ** 2) + 0.5 * np.sum(r["bias"] ** 2) + 0.5 * np.sum(r["prior"] ** 2)
    sq = np.sum(r["vis"] ** 2, 1)
    if w.cauchy_a > 0:
        b2 = w.cauchy_a ** 2
        c += 0.5 * np.sum(b2 * np.log1p(sq / b2))
    else:
        c += 0.5 * np.sum(sq)
    return float(c)


def retract(w, xi):
    """x (+) xi in the package's unknown ordering (right-multiplicative on rotations)."""
    w2 = w.copy()
    K, P = w.K, w.P
    d = xi[:6 * K].reshape(K, 6)
    q = (Rot.from_quat(w.quat) * Rot.from_rotvec(d[:, :3])).as_quat()
    w2.quat = q * np.sign(np.sum(q * w.quat, 1, keepdims=True) + 1e-300)
    w2.pos = w.pos + d[:, 3:]
    w2.bias = w.bias + xi[6 * K:6 * K + 6 * w.F].reshape(w.F, 6)
    w2.ld = float(w.ld + xi[P - 1])
    w2.rho = w.rho + xi[P:]
    return w2


def stacked(w, robust_sqrt=False, trunc_ld=True):
    r = residuals(w, robust_sqrt, trunc_ld)
    return np.concatenate([r["imu"].ravel(), r["vis"].ravel(), r["bias"].ravel(), r["prior"].ravel()])


def fd_jacobian(w, cols, robust_sqrt=False, eps_rot=1e-6, eps_ld=2e-8):
    """Central-difference Jacobian of the stacked residual vector wrt the listed unknowns.
    The line-delay column is differentiated on the un-truncated time model (the reference truncates
    the delay to integer ns, which makes the residual piecewise constant below 1 ns)."""
    P = w.P
    J = []
    for u in cols:
        isld = (u == P - 1)
        e = eps_ld if isld else eps_rot
        xi = np.zeros(w.N); xi[u] = e
        J.append((stacked(retract(w, xi), robust_sqrt, not isld) - stacked(retract(w, -xi), robust_sqrt, not isld)) / (2 * e))
    return np.array(J).T
