"""Host-side factor packing rules of `TrajectoryManager::UpdateTrajectory`
(reference src/estimator/trajectory_manager.cpp:317-451): which bias state an IMU sample is tied
to, and the bias random-walk sqrt-information between consecutive keyframes.
"""
from __future__ import annotations

import numpy as np


def imu_bias_index(imu_t: np.ndarray, frame_t: np.ndarray) -> np.ndarray:
    """Bias index per IMU sample (reference trajectory_manager.cpp:395-414).

    t < t_0 -> 0 ; t >= t_last -> last ; else the i-1 with t_{i-1} <= t < t_i.
    """
    imu_t = np.asarray(imu_t, np.int64)
    frame_t = np.asarray(frame_t, np.int64)
    idx = np.searchsorted(frame_t, imu_t, side="right") - 1
    return np.clip(idx, 0, len(frame_t) - 1).astype(np.int32)


def bias_chain_sqrt_info(imu_t: np.ndarray, frame_t: np.ndarray, sigma_bg: float, sigma_ba: float) -> np.ndarray:
    """(F-1, 6) sqrt-information of the bias random walk between frames i and i+1.

    Restates reference trajectory_manager.cpp:420-447: covariance propagated with F = I and
    G = I*dt over the IMU intervals [imu[k-1], imu[k]) with imu[k-1] >= t_i and imu[k] < t_{i+1};
    sqrt_info = diag of chol(cov^-1)^T = 1/sqrt(cov_kk) (cov is diagonal).
    """
    imu_t = np.asarray(imu_t, np.int64)
    frame_t = np.asarray(frame_t, np.int64)
    out = np.zeros((len(frame_t) - 1, 6))
    dts = np.diff(imu_t).astype(np.float64) * 1e-9
    for i in range(len(frame_t) - 1):
        left, right = frame_t[i], frame_t[i + 1]
        sel = (imu_t[:-1] >= left) & (imu_t[1:] < right)
        s2 = float(np.sum(dts[sel] ** 2))
        cov = np.array([sigma_bg ** 2 * s2] * 3 + [sigma_ba ** 2 * s2] * 3)
        out[i] = 1.0 / np.sqrt(cov)
    return out


# ------------------------------------------------------------------------------------------------ visual factors, depths
# (the rest of TrajectoryManager::UpdateTrajectory's packing around the solve)

def is_landmark_candidate(n_obs: int, start_frame: int, window_size: int) -> bool:
    """FeatureManager::isLandmarkCandidate (reference src/visual_odometry/feature_manager.h:58-65): a feature track
    enters the window as a landmark iff it has >= 2 observations and starts before frame WINDOW_SIZE - 2."""
    return n_obs >= 2 and start_frame < window_size - 2


def pack_visual(tracks, timestamps, window_size: int):
    """Visual blocks of a window, in the reference's order (trajectory_manager.cpp:358-383).

    tracks: iterable of dicts {"start_frame": int, "points": (n,3) normalised-plane points (x, y, 1),
    "uv": (n,2) pixel coordinates, "depth": estimated depth of the anchor observation}; observation k of a track belongs to
    frame start_frame + k.  timestamps: (window_size + 1,) frame times [ns].
    The first observation is the anchor (ti, rowi = round(v), pts_i); every later observation adds one block against it.
    Returns a dict with v_lm, v_ti, v_tj, v_rowi, v_rowj, v_pi, v_pj, rho (inverse depths, getDepthVector:
    feature_manager.cpp:110-123) and `track_of_landmark` (index into `tracks` of every landmark, for copy-back)."""
    timestamps = np.asarray(timestamps, np.int64)
    v_lm, v_ti, v_tj, v_rowi, v_rowj, v_pi, v_pj, rho, owner = [], [], [], [], [], [], [], [], []
    for ti_idx, tr in enumerate(tracks):
        pts = np.asarray(tr["points"], np.float64)
        uv = np.asarray(tr["uv"], np.float64)
        if not is_landmark_candidate(len(pts), int(tr["start_frame"]), window_size):
            continue
        lm = len(rho)
        rho.append(1.0 / float(tr["depth"]))
        owner.append(ti_idx)
        i = int(tr["start_frame"])
        rowi = int(np.floor(uv[0, 1] + 0.5))           # std::round: half away from zero (np.round rounds half to even); rows >= 0
        for k in range(1, len(pts)):
            j = i + k
            v_lm.append(lm); v_ti.append(int(timestamps[i])); v_tj.append(int(timestamps[j]))
            v_rowi.append(rowi); v_rowj.append(int(np.floor(uv[k, 1] + 0.5)))
            v_pi.append(pts[0, :2] / pts[0, 2]); v_pj.append(pts[k, :2] / pts[k, 2])
    return dict(v_lm=np.array(v_lm, np.int32), v_ti=np.array(v_ti, np.int64), v_tj=np.array(v_tj, np.int64),
                v_rowi=np.array(v_rowi, np.int32), v_rowj=np.array(v_rowj, np.int32),
                v_pi=np.array(v_pi, np.float64).reshape(-1, 2), v_pj=np.array(v_pj, np.float64).reshape(-1, 2),
                rho=np.array(rho, np.float64), track_of_landmark=np.array(owner, np.int32))


def imu_in_window(imu_t: np.ndarray, opt_min_time: int, opt_max_time: int) -> np.ndarray:
    """Boolean mask of the IMU samples that get a factor (trajectory_manager.cpp:386-394): opt_min <= t < opt_max, where
    opt_min = (index of the first knot active at timestamps[0]) * dt and opt_max = the spline's maxTimeNs (:322-325)."""
    imu_t = np.asarray(imu_t, np.int64)
    return (imu_t >= opt_min_time) & (imu_t < opt_max_time)


def opt_min_time(t_first_frame: int, t0_ns: int, dt_ns: int) -> int:
    """computeTIndexNs(timestamps[0]).second * getDtNs() (trajectory_manager.cpp:322; the reference's spline starts at 0):
    the time of the first knot that is active at the first frame, relative to the same origin as t_first_frame."""
    return t0_ns + ((int(t_first_frame) - int(t0_ns)) // int(dt_ns)) * int(dt_ns)


def depths_from_solution(rho: np.ndarray):
    """FeatureManager::setDepth after the solve (feature_manager.cpp:125-143): estimated_depth = 1 / rho per landmark and
    the solve flag (True = SovelSucc, False = SolveFail when the depth came out negative)."""
    rho = np.asarray(rho, np.float64)
    with np.errstate(divide="ignore"):
        depth = 1.0 / rho
    return depth, ~(depth < 0)


# ------------------------------------------------------------------------------------------------ sparsity of the reduced system
# Python mirror of csrc/host_pack.hpp: plan_sparsity (the device library plans for itself; this copy serves bench.py's flop / byte counts and
# the tests, which check it against the C++ planner entry for entry).

def landmark_spans(w):
    """(klo, khi) per landmark: first / last knot its residual blocks can touch over the whole box of the line delay (4 knots per spline
    end at t + row * ld, image_feature_factor.h:72-101; both ends of the box bound the monotone row time).  khi = -1, klo = K: no observation."""
    K, L = w.K, w.L
    ld_a, ld_b = (w.ld, w.ld) if w.fix_ld else (w.ld_lo, w.ld_hi)
    klo = np.full(L, K, np.int64); khi = np.full(L, -1, np.int64)

    def seg(t, row, ld):
        ld_ns = int(ld * 1e9)                        # C++ (long long)(ld * 1e9): truncation toward zero
        tau = (np.asarray(t, np.int64) - w.t0_ns) + np.asarray(row, np.int64) * ld_ns
        s = np.where(tau >= 0, tau // w.dt_ns, -((-tau) // w.dt_ns))       # C++ integer division truncates toward zero
        return np.clip(s, 0, K - 4)
    if w.V:
        ss = np.stack([seg(w.v_ti, w.v_rowi, ld_a), seg(w.v_ti, w.v_rowi, ld_b), seg(w.v_tj, w.v_rowj, ld_a), seg(w.v_tj, w.v_rowj, ld_b)])
        np.minimum.at(klo, w.v_lm, ss.min(axis=0))
        np.maximum.at(khi, w.v_lm, ss.max(axis=0) + 3)
    return klo, khi


def schur_nonzero_flops(w):
    """Flops of the Schur complement's NON-ZERO products: landmark l's row of W has nnz_l = 6 (khi - klo + 1) + 1 entries (its span's knot
    columns and the line delay), and contributes the lower triangle of their outer product: sum_l nnz_l (nnz_l + 1)."""
    klo, khi = landmark_spans(w)
    nnz = np.where(khi >= 0, 6 * (khi - klo + 1) + 1, 0)
    return int(np.sum(nnz * (nnz + 1)))


def reduced_system_envelope(w, align_panels=True):
    """env_first per 16-row tile of the reduced system S (P // 16 + 1 entries: the rhs rides along as row P): the first tile column that can
    be non-zero in S -- and in its Cholesky factor, whose fill stays inside the row envelope.  Couplings: an IMU group's 4 knots + bias state,
    bias-chain links, the prior's columns mutually, a landmark's span knots mutually and each with the line delay.  align_panels: the
    device's rounding to 32-column panels (host_pack.hpp)."""
    K, F, P = w.K, w.F, w.P
    K6 = 6 * K
    fk = 6 * np.arange(K); fb = K6 + 6 * np.arange(F); fld = P - 1
    seg = (w.imu_t - w.t0_ns) // w.dt_ns
    for s, b in {(int(a), int(c)) for a, c in zip(seg, w.imu_bias)}:
        fk[s + 1:s + 4] = np.minimum(fk[s + 1:s + 4], 6 * s)
        fb[b] = min(fb[b], 6 * s)
    for i, j in zip(w.bc_i, w.bc_j):
        fb[max(i, j)] = min(fb[max(i, j)], K6 + 6 * min(i, j))
    klo, khi = landmark_spans(w)
    for a, b in zip(klo, khi):
        if b >= 0:
            fk[a + 1:b + 1] = np.minimum(fk[a + 1:b + 1], 6 * a)
            fld = min(fld, 6 * a)
    if w.pn > 0:
        col0 = [{0: 6 * i, 1: 6 * i + 3, 2: K6 + 6 * i, 3: K6 + 6 * i + 3, 4: P - 1}[int(k)] for k, i in zip(w.p_kind, w.p_index)]
        m = min(col0)
        for k, i in zip(w.p_kind, w.p_index):
            if k <= 1:
                fk[i] = min(fk[i], m)
            elif k <= 3:
                fb[i] = min(fb[i], m)
            else:
                fld = min(fld, m)
    first = np.concatenate([np.repeat(fk, 6), np.repeat(fb, 6), [fld]])
    ntr = P // 16 + 1
    env = np.zeros(ntr, np.int64)
    for r in range(ntr):
        f = min(16 * r, int(first[16 * r:min(16 * r + 16, P)].min()) if 16 * r < P else 16 * r)
        if 16 * r <= P < 16 * r + 16:
            f = 0
        ft = f // 16
        if align_panels:
            ft = 0 if r < 2 else (min(ft, 2 * (r // 2) - 2) & ~1)
        env[r] = ft
    return env


def envelope_entries(w, dense=False):
    """Entries of the lower triangle of S inside the envelope (what a factorisation must read at least once); dense: P (P + 1) / 2."""
    P = w.P
    if dense:
        return P * (P + 1) // 2
    env = reduced_system_envelope(w)
    return int(sum(i - 16 * int(env[i // 16]) + 1 for i in range(P)))
