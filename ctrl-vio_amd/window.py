"""Flat description of one sliding window (state + factors): the on-the-wire format of the C ABI.

Mirrors what `TrajectoryManager::UpdateTrajectory` hands to `TrajectoryEstimator`
(reference src/estimator/trajectory_manager.cpp:331-451) but by *index* instead of by
pointer: knots/biases/landmarks are addressed by their position in the window's arrays.

Unknown ordering used by every dense quantity in this package (same as include/ctvio.h):
    knot k : rot 6k..6k+2, pos 6k+3..6k+5 | bias f : bg 6K+6f.., ba 6K+6f+3.. | ld 6K+6F
    P = 6K+6F+1 ; inverse depth l : P+l ; N = P+L
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field

import numpy as np

PK_ROT, PK_POS, PK_BG, PK_BA, PK_LD = 0, 1, 2, 3, 4


def _f64(a, shape):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(shape)


@dataclass
class Window:
    # spline (reference src/spline/se3_spline.h:108-111; knots: so3_spline.h:410, rd_spline.h:317)
    t0_ns: int
    dt_ns: int
    quat: np.ndarray            # (K,4) x,y,z,w
    pos: np.ndarray             # (K,3)
    bias: np.ndarray            # (F,6) bg, ba
    rho: np.ndarray             # (L,) inverse depths
    ld: float = 0.0
    ld_lo: float = 0.0
    ld_hi: float = 3.5e-5
    fix_ld: bool = False
    lock_bg: bool = False
    lock_ba: bool = False
    fixed_upto: int = -1
    # calibration / weights
    q_CI: np.ndarray = field(default_factory=lambda: np.array([0, 0, 0, 1.0]))
    p_CI: np.ndarray = field(default_factory=lambda: np.zeros(3))
    gravity: np.ndarray = field(default_factory=lambda: np.array([0, 0, 9.80766]))
    imu_w: np.ndarray = field(default_factory=lambda: np.array([250.0] * 3 + [12.5] * 3))
    img_w: float = 800.0
    cauchy_a: float = 2.0
    # IMU factors
    imu_t: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    imu_gyro: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    imu_acc: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    imu_bias: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    # bias chain
    bc_i: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    bc_j: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    bc_w: np.ndarray = field(default_factory=lambda: np.zeros((0, 6)))
    # visual factors
    v_lm: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    v_ti: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    v_tj: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    v_rowi: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    v_rowj: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    v_pi: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    v_pj: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    # prior r = r0 + J0 dx (reference marginalization_factor.cpp:326-373)
    pJ0: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))   # (n,n) as a matrix J0[i,j]
    pr0: np.ndarray = field(default_factory=lambda: np.zeros(0))
    p_kind: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    p_index: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    p_off: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    p_x0: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))
    # optional per-block Cauchy width (V,) -- the reference picks CauchyLoss(1 | 2) per residual block
    # (trajectory_estimator.cpp:320-323); None: cauchy_a for every block
    v_cauchy: np.ndarray | None = None
    # optional per-knot constancy flags (K,) uint8 -- SetParameterBlockConstant per AddControlPoints call
    # (trajectory_estimator.cpp:134-138), on top of fixed_upto; None: none
    knot_const: np.ndarray | None = None

    def normalize(self) -> "Window":
        """Coerce dtypes/shapes in place (contiguous, fp64 / int32 / int64)."""
        self.t0_ns, self.dt_ns = int(self.t0_ns), int(self.dt_ns)
        self.quat = _f64(self.quat, (-1, 4)); self.pos = _f64(self.pos, (-1, 3))
        self.bias = _f64(self.bias, (-1, 6)); self.rho = _f64(self.rho, (-1,))
        self.q_CI = _f64(self.q_CI, (4,)); self.p_CI = _f64(self.p_CI, (3,))
        self.gravity = _f64(self.gravity, (3,)); self.imu_w = _f64(self.imu_w, (6,))
        self.imu_t = np.ascontiguousarray(self.imu_t, np.int64)
        self.imu_gyro = _f64(self.imu_gyro, (-1, 3)); self.imu_acc = _f64(self.imu_acc, (-1, 3))
        self.imu_bias = np.ascontiguousarray(self.imu_bias, np.int32)
        self.bc_i = np.ascontiguousarray(self.bc_i, np.int32); self.bc_j = np.ascontiguousarray(self.bc_j, np.int32)
        self.bc_w = _f64(self.bc_w, (-1, 6))
        self.v_lm = np.ascontiguousarray(self.v_lm, np.int32)
        self.v_ti = np.ascontiguousarray(self.v_ti, np.int64); self.v_tj = np.ascontiguousarray(self.v_tj, np.int64)
        self.v_rowi = np.ascontiguousarray(self.v_rowi, np.int32); self.v_rowj = np.ascontiguousarray(self.v_rowj, np.int32)
        self.v_pi = _f64(self.v_pi, (-1, 2)); self.v_pj = _f64(self.v_pj, (-1, 2))
        n = int(np.asarray(self.pr0).size)
        self.pJ0 = _f64(self.pJ0, (n, n)); self.pr0 = _f64(self.pr0, (n,))
        self.p_kind = np.ascontiguousarray(self.p_kind, np.int32); self.p_index = np.ascontiguousarray(self.p_index, np.int32)
        self.p_off = np.ascontiguousarray(self.p_off, np.int32); self.p_x0 = _f64(self.p_x0, (-1, 4))
        if self.v_cauchy is not None:
            self.v_cauchy = _f64(self.v_cauchy, (self.v_lm.shape[0],))
        if self.knot_const is not None:
            self.knot_const = np.ascontiguousarray(self.knot_const, np.uint8).reshape(self.quat.shape[0])
        return self

    # sizes
    @property
    def K(self): return self.quat.shape[0]
    @property
    def F(self): return self.bias.shape[0]
    @property
    def L(self): return self.rho.shape[0]
    @property
    def M(self): return self.imu_t.shape[0]
    @property
    def NB(self): return self.bc_i.shape[0]
    @property
    def V(self): return self.v_lm.shape[0]
    @property
    def P(self): return 6 * self.K + 6 * self.F + 1
    @property
    def N(self): return self.P + self.L
    @property
    def pn(self): return self.pr0.shape[0]

    def copy(self) -> "Window":
        return copy.deepcopy(self)

    _SCALARS = ("t0_ns", "dt_ns", "ld", "ld_lo", "ld_hi", "fix_ld", "lock_bg", "lock_ba", "fixed_upto", "img_w", "cauchy_a")
    _ARRAYS = ("quat", "pos", "bias", "rho", "q_CI", "p_CI", "gravity", "imu_w", "imu_t", "imu_gyro", "imu_acc",
               "imu_bias", "bc_i", "bc_j", "bc_w", "v_lm", "v_ti", "v_tj", "v_rowi", "v_rowj", "v_pi", "v_pj",
               "pJ0", "pr0", "p_kind", "p_index", "p_off", "p_x0")
    _OPTIONAL = ("v_cauchy", "knot_const")

    def to_dict(self, prefix: str = "") -> dict:
        """Flat dict of numpy values (np.savez-able); inverse of from_dict."""
        self.normalize()
        d = {prefix + k: np.asarray(getattr(self, k)) for k in self._ARRAYS}
        d.update({prefix + k: np.asarray(getattr(self, k)) for k in self._SCALARS})
        for k in self._OPTIONAL:
            if getattr(self, k) is not None:
                d[prefix + k] = np.asarray(getattr(self, k))
        return d

    @classmethod
    def from_dict(cls, d, prefix: str = "") -> "Window":
        kw = {k: np.array(d[prefix + k]) for k in cls._ARRAYS}
        for k in cls._SCALARS:
            v = d[prefix + k]
            kw[k] = v.item() if hasattr(v, "item") else v
        for k in ("fix_ld", "lock_bg", "lock_ba"):
            kw[k] = bool(kw[k])
        for k in cls._OPTIONAL:
            if prefix + k in d:
                kw[k] = np.array(d[prefix + k])
        return cls(**kw).normalize()

    def state_vector(self) -> np.ndarray:
        """Ambient state (quat, pos, bias, ld, rho) flattened; used for relative-error metrics."""
        return np.concatenate([self.quat.ravel(), self.pos.ravel(), self.bias.ravel(), [self.ld], self.rho.ravel()])

    def max_time_ns(self) -> int:
        """Se3Spline::maxTimeNs (reference src/spline/rd_spline.h maxTimeNs: t0 + (K-3) dt)."""
        return self.t0_ns + (self.K - 3) * self.dt_ns


def rel_state_error(a: Window, b: Window) -> dict:
    """Relative error of a vs b per state group (norm of difference / norm of b-group)."""
    def rel(x, y):
        d = np.linalg.norm(np.ravel(x) - np.ravel(y))
        return float(d / max(np.linalg.norm(np.ravel(y)), 1e-30))
    # quaternion sign ambiguity: align signs first
    qa = a.quat * np.sign(np.sum(a.quat * b.quat, axis=1, keepdims=True) + 1e-300)
    out = {
        "quat": rel(qa, b.quat), "pos": rel(a.pos, b.pos), "bias": rel(a.bias, b.bias),
        "rho": rel(a.rho, b.rho), "ld": rel([a.ld], [b.ld]) if b.ld != 0 else abs(a.ld - b.ld),
    }
    xa = np.concatenate([qa.ravel(), a.pos.ravel(), a.bias.ravel(), [a.ld], a.rho.ravel()])
    out["state"] = rel(xa, b.state_vector())
    return out
