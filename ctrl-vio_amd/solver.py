"""Python host side above the C ABI: a batch of sliding windows solved on one MI355X.

Mirrors the call pattern of the reference's TrajectoryManager::UpdateTrajectory
(src/estimator/trajectory_manager.cpp:350-463): build an estimator, add the window's factors,
Solve(max_iterations), copy the state back -- except that many independent windows are solved per call.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .window import Window


class Solver:
    """One handle = one HIP stream + the HBM buffers of a batch of windows."""

    def __init__(self, device: int = 0, precision: str = "fp64", use_mfma: bool = True, check_every: int = 4,
                 deterministic: int = -1, host_threads: int = 0, use_graph: bool = True, line_search: bool = True,
                 **tolerances):
        """All-fp64, like the reference (`precision` is kept for call compatibility: only "fp64" exists).  deterministic: 1 = order-fixed
        accumulation (bitwise reproducible), 0 = off, -1 = on for batches of <= 64 windows."""
        if precision not in ("fp64", capi.FP64):
            raise ValueError("only precision='fp64' exists (the mixed fp32 mode was removed: it missed the 1e-4 contract)")
        self._lib = capi.load_library()
        if self._lib.ctvio_device_count() <= 0:
            raise capi.CtvioError("no HIP device: ctrl-vio_amd has no CPU fallback")
        opt = capi.Options()
        self._lib.ctvio_default_options(C.byref(opt))
        opt.device = device
        opt.precision = capi.FP64
        opt.use_mfma = int(use_mfma)   # 0 / 1 / 2 (include/ctvio.h)
        opt.check_every = int(check_every)
        opt.deterministic = int(deterministic)
        opt.host_threads = int(host_threads)
        opt.use_graph = int(bool(use_graph))
        opt.line_search = int(bool(line_search))
        for k, v in tolerances.items():
            if not hasattr(opt, k):
                raise TypeError(f"unknown option {k}")
            setattr(opt, k, v)
        self._h = C.c_void_p()
        capi.check(self._lib.ctvio_create(C.byref(opt), C.byref(self._h)))
        self.windows: list[Window] = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ctvio_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- batch construction
    def clear(self):
        capi.check(self._lib.ctvio_clear(self._h))
        self.windows = []

    def add_window(self, w: Window) -> int:
        keep = []
        cw = capi.to_cwindow(w, keep)
        wid = C.c_int32(-1)
        capi.check(self._lib.ctvio_add_window(self._h, C.byref(cw), C.byref(wid)))
        self.windows.append(w)
        return wid.value

    def upload(self):
        capi.check(self._lib.ctvio_upload(self._h))

    def set_windows(self, windows):
        """ctvio_set_batch: validate + pack (C++ host threads) + one H2D copy of the whole batch."""
        windows = list(windows)
        keep = []
        arr = (capi.CWindow * len(windows))()
        for i, w in enumerate(windows):
            arr[i] = capi.to_cwindow(w, keep)
        capi.check(self._lib.ctvio_set_batch(self._h, len(windows), C.cast(arr, C.c_void_p)))
        self.windows = windows

    def set_cbatch(self, arr, n, windows=None):
        """Same from a prebuilt (CWindow * n) array (bench.py keeps the ctypes marshalling out of the timed region)."""
        capi.check(self._lib.ctvio_set_batch(self._h, int(n), C.cast(arr, C.c_void_p)))
        self.windows = list(windows) if windows is not None else []

    def get_batch_state(self):
        """Every window's state with one D2H copy: (quat (sumK,4), pos (sumK,3), bias (sumF,6), rho (sumL,), ld (n,))."""
        K = sum(w.K for w in self.windows); F = sum(w.F for w in self.windows); L = sum(w.L for w in self.windows); n = len(self.windows)
        q = np.zeros((K, 4)); p = np.zeros((K, 3)); b = np.zeros((F, 6)); r = np.zeros(max(L, 1)); ld = np.zeros(n)
        capi.check(self._lib.ctvio_get_batch_state(self._h, capi._p(q), capi._p(p), capi._p(b), capi._p(r), capi._p(ld)))
        return q, p, b, r[:L], ld

    def writeback_all(self):
        """get_batch_state scattered into the Window objects of the batch (in place, like Ceres updates its double*)."""
        q, p, b, r, ld = self.get_batch_state()
        k = f = l = 0
        for i, w in enumerate(self.windows):
            w.quat[:] = q[k:k + w.K]; w.pos[:] = p[k:k + w.K]; w.bias[:] = b[f:f + w.F]; w.rho[:] = r[l:l + w.L]; w.ld = float(ld[i])
            k += w.K; f += w.F; l += w.L

    @property
    def n(self) -> int:
        return int(self._lib.ctvio_num_windows(self._h))

    # ---- solve
    def solve(self, max_iterations: int = 15, writeback: bool = True):
        """Solve every window of the batch; returns list of summary dicts.  With writeback the
        Window objects passed to add_window are updated in place (Ceres updates double* in place)."""
        n = self.n
        sm = (capi.Summary * n)()
        capi.check(self._lib.ctvio_solve(self._h, int(max_iterations), C.cast(sm, C.c_void_p)))
        if writeback:
            self.writeback_all()
        return [s.as_dict() for s in sm]

    # ---- IMU-only predict (reference TrajectoryManager::InitTrajectory, src/estimator/trajectory_manager.cpp:288-315)
    @staticmethod
    def predict_window(w: Window, fixed_upto: int = -1) -> Window:
        """The factor set of InitTrajectory for window w: IMU blocks only (no visual blocks, bias chain or prior), both biases
        locked (option.lock_ab / lock_wb), knots 0..fixed_upto constant.  NB: the reference calls SetFixedIndex(max_bef_idx)
        AFTER its AddIMUMeasurementAnalytic loop, and constancy is decided when a knot is added
        (trajectory_estimator.cpp:134-138), so in the reference as written no knot is constant: pass fixed_upto = -1 for
        that behaviour, max_bef_idx for what the call order suggests was intended.  The per-block Cauchy widths go with the visual
        blocks; per-knot constancy (knot_const) is a property of the caller's window and is KEPT: a knot the caller holds constant
        stays constant in the predict as well."""
        p = w.copy()
        p.v_cauchy = None
        z = lambda a: a[:0]
        p.v_lm, p.v_ti, p.v_tj, p.v_rowi, p.v_rowj, p.v_pi, p.v_pj = z(p.v_lm), z(p.v_ti), z(p.v_tj), z(p.v_rowi), z(p.v_rowj), z(p.v_pi), z(p.v_pj)
        p.bc_i, p.bc_j, p.bc_w = z(p.bc_i), z(p.bc_j), z(p.bc_w)
        p.pJ0 = np.zeros((0, 0)); p.pr0 = np.zeros(0)
        p.p_kind, p.p_index, p.p_off, p.p_x0 = z(p.p_kind), z(p.p_index), z(p.p_off), z(p.p_x0)
        p.lock_bg = p.lock_ba = True
        p.fixed_upto = int(fixed_upto)
        return p.normalize()

    def predict(self, windows, fixed_upto=None, max_iterations: int = 8):
        """InitTrajectory for a batch: Solve(8) of the IMU-only problems; the knots of `windows` are updated in place."""
        fixed_upto = [-1] * len(windows) if fixed_upto is None else list(fixed_upto)
        pw = [self.predict_window(w, f) for w, f in zip(windows, fixed_upto)]
        self.set_windows(pw)
        sms = self.solve(max_iterations)
        for w, p in zip(windows, pw):
            w.quat[:] = p.quat; w.pos[:] = p.pos
        return sms

    def solve_raw(self, max_iterations: int = 15):
        """Solve without any host-side copy besides the summaries (used by bench.py)."""
        capi.check(self._lib.ctvio_solve(self._h, int(max_iterations), None))

    def get_state(self, wid: int, into: Window | None = None) -> Window:
        w = into if into is not None else self.windows[wid].copy()
        ld = C.c_double()
        capi.check(self._lib.ctvio_get_state(self._h, wid, capi._p(w.quat), capi._p(w.pos), capi._p(w.bias), capi._p(w.rho),
                                             C.cast(C.byref(ld), C.c_void_p)))
        w.ld = float(ld.value)
        return w

    def set_state(self, wid: int, w: Window):
        w.normalize()
        capi.check(self._lib.ctvio_set_state(self._h, wid, capi._p(w.quat), capi._p(w.pos), capi._p(w.bias), capi._p(w.rho), float(w.ld)))

    def snapshot_state(self):
        capi.check(self._lib.ctvio_snapshot_state(self._h))

    def restore_state(self):
        capi.check(self._lib.ctvio_restore_state(self._h))

    # ---- diagnostics
    def linearize(self, wid: int):
        w = self.windows[wid]
        P, L, N = w.P, w.L, w.N
        H = np.zeros((P, P)); W = np.zeros((P, max(L, 1))); Hll = np.zeros(max(L, 1)); g = np.zeros(N); cost = C.c_double()
        capi.check(self._lib.ctvio_linearize(self._h, wid, capi._p(H), capi._p(W), capi._p(Hll), capi._p(g),
                                             C.cast(C.byref(cost), C.c_void_p)))
        return H, W[:, :L], Hll[:L], g, float(cost.value)

    def cost(self, wid: int) -> float:
        c = C.c_double()
        capi.check(self._lib.ctvio_cost(self._h, wid, C.cast(C.byref(c), C.c_void_p)))
        return float(c.value)

    def lm_step(self, wid: int, mu: float = 1e4):
        w = self.windows[wid]
        d = np.zeros(w.N); mc = C.c_double()
        capi.check(self._lib.ctvio_lm_step(self._h, wid, float(mu), capi._p(d), C.cast(C.byref(mc), C.c_void_p)))
        return d, float(mc.value)

    def residual_summary(self, wid: int):
        """ResidualSummary of window wid at its current state: dict type -> (sum |r_i| per component, block count)."""
        w = self.windows[wid]
        sums = np.zeros(14 + w.pn); cnt = np.zeros(4, np.int32)
        capi.check(self._lib.ctvio_residual_summary(self._h, wid, capi._p(sums), capi._p(cnt)))
        return {"imu": (sums[:6].copy(), int(cnt[0])), "bias": (sums[6:12].copy(), int(cnt[1])), "image": (sums[12:14].copy(), int(cnt[2])),
                "prior": (sums[14:].copy(), int(cnt[3]))}

    def marginalize(self, wid: int, role, eps: float = 1e-8):
        """Prior construction from window `wid` (ctvio_marginalize): role[N] 1 = marginalise, 0 = keep, -1 = not involved.
        Returns (kept indices, J0 (n, n), r0 (n))."""
        N = self.windows[wid].N
        role = np.ascontiguousarray(role, np.int8)
        assert role.shape == (N,)
        kept = np.zeros(N, np.int32); J0 = np.zeros(N * N); r0 = np.zeros(N); n = C.c_int32(0)
        capi.check(self._lib.ctvio_marginalize(self._h, wid, capi._p(role), float(eps), C.byref(n), capi._p(kept), capi._p(J0), capi._p(r0)))
        n = n.value
        return kept[:n].copy(), J0[: n * n].reshape(n, n).copy(), r0[:n].copy()

    def marginalize_ran_on_host(self) -> bool:
        """Whether the last marginalize / marginalize_batch call factored on the host cores (ctvio_marginalize_ran_on_host)."""
        return bool(self._lib.ctvio_marginalize_ran_on_host(self._h))

    def marginalize_batch(self, roles, eps: float = 1e-8):
        """ctvio_marginalize_batch: roles = list of per-window role arrays.  Returns a list of (kept, J0, r0) per window."""
        Ns = [w.N for w in self.windows]
        role = np.ascontiguousarray(np.concatenate([np.asarray(r, np.int8) for r in roles]), np.int8)
        assert role.shape[0] == sum(Ns)
        nk = np.zeros(len(Ns), np.int32); kept = np.zeros(sum(Ns), np.int32)
        nkeep = [int((np.asarray(r) == 0).sum()) for r in roles]
        J0 = np.zeros(max(sum(k * k for k in nkeep), 1)); r0 = np.zeros(max(sum(nkeep), 1))
        capi.check(self._lib.ctvio_marginalize_batch(self._h, capi._p(role), float(eps), capi._p(nk), capi._p(kept), capi._p(J0), capi._p(r0)))
        out, u, oj, orr = [], 0, 0, 0
        for N, n in zip(Ns, nk):
            n = int(n)
            out.append((kept[u:u + n].copy(), J0[oj:oj + n * n].reshape(n, n).copy(), r0[orr:orr + n].copy()))
            u += N; oj += n * n; orr += n
        return out

    def gauge_restore(self, wids, knots, q0, t0):
        """4-DoF gauge restore (reference double2vector): windows `wids`, reference knot index per window, its pre-solve
        quaternion (n,4) (x,y,z,w) and position (n,3).  Acts on the device state; read it back with get_state."""
        ids = np.ascontiguousarray(wids, np.int32); kn = np.ascontiguousarray(knots, np.int32)
        q = np.ascontiguousarray(q0, np.float64).reshape(-1, 4); t = np.ascontiguousarray(t0, np.float64).reshape(-1, 3)
        assert ids.shape[0] == kn.shape[0] == q.shape[0] == t.shape[0]
        capi.check(self._lib.ctvio_gauge_restore(self._h, int(ids.shape[0]), capi._p(ids), capi._p(kn), capi._p(q), capi._p(t)))

    def spline_eval(self, wid: int, t_ns):
        t = np.ascontiguousarray(t_ns, np.int64)
        n = t.shape[0]
        pose = np.zeros((n, 7)); vel = np.zeros((n, 3)); om = np.zeros((n, 3)); acc = np.zeros((n, 3))
        capi.check(self._lib.ctvio_spline_eval(self._h, wid, n, capi._p(t), capi._p(pose), capi._p(vel), capi._p(om), capi._p(acc)))
        return pose, vel, om, acc

    def spline_eval_batch(self, win, t_ns, want=("pose", "vel", "omega")):
        """Queries of any windows of the batch in one launch (ctvio_spline_eval_batch): query i = (window win[i], absolute time
        t_ns[i]).  Returns (dict of the requested outputs, device milliseconds of the evaluation kernel alone)."""
        t = np.ascontiguousarray(t_ns, np.int64); wi = np.ascontiguousarray(win, np.int32)
        n = int(t.shape[0])
        out = {"pose": np.zeros((n, 7)) if "pose" in want else None, "vel": np.zeros((n, 3)) if "vel" in want else None,
               "omega": np.zeros((n, 3)) if "omega" in want else None, "acc": np.zeros((n, 3)) if "acc" in want else None}
        ms = C.c_double()
        ptr = lambda a: capi._p(a) if a is not None else None
        capi.check(self._lib.ctvio_spline_eval_batch(self._h, C.c_int64(n), capi._p(wi), capi._p(t), ptr(out["pose"]), ptr(out["vel"]),
                                                     ptr(out["omega"]), ptr(out["acc"]), C.cast(C.byref(ms), C.c_void_p)))
        return {k: v for k, v in out.items() if v is not None}, float(ms.value)

    def sensor_pose(self, wid: int, t_ns, q_SI, p_SI):
        """Trajectory::GetSensorPose (reference src/spline/trajectory.cpp:39-56): poseNs(t) * T_StoI, evaluated on the device.
        q_SI = (x,y,z,w).  Returns (n,7) = (p, q)."""
        t = np.ascontiguousarray(t_ns, np.int64)
        q = np.ascontiguousarray(q_SI, np.float64).reshape(4); p = np.ascontiguousarray(p_SI, np.float64).reshape(3)
        pose = np.zeros((t.shape[0], 7))
        capi.check(self._lib.ctvio_sensor_pose(self._h, wid, int(t.shape[0]), capi._p(t), capi._p(q), capi._p(p), capi._p(pose)))
        return pose

    PHASES = ("k_imu_linearize", "k_vis_eval", "k_assemble_vis", "assemble_rest", "k_schur_mfma", "k_cholesky_solve", "rest", "solve")

    def set_profiling(self, on: bool):
        capi.check(self._lib.ctvio_set_profiling(self._h, int(bool(on))))

    def last_timing(self):
        """(ms[8], launches[8]) of the last solve; see include/ctvio.h ctvio_last_timing."""
        ms = np.zeros(8); n = np.zeros(8, np.int32)
        capi.check(self._lib.ctvio_last_timing(self._h, capi._p(ms), capi._p(n)))
        return ms, n

    @property
    def graph_captures(self) -> int:
        """hipGraph captures of the LM pass so far (a stream of equally shaped batches captures once)."""
        return int(self._lib.ctvio_graph_captures(self._h))

    @property
    def stream(self) -> int:
        return int(self._lib.ctvio_stream(self._h) or 0)
