"""Deterministic synthetic sliding windows (SURVEY.md section 8d recipe; BASELINE.json configs).

The ground-truth trajectory is itself a cubic B-spline, so the estimator's model is exact.
Noise levels, weights, extrinsics and line-delay bounds are the TUM-RSVI values of the
reference's config (config/ct_odometry_tumrs.yaml:13-34, config/tumrs/cam_tumrs.yaml:9-10,
config/tumrs/imu_tumrs.yaml).  RNG: numpy Philox keyed by `seed` (counter-based, stable).
"""
from __future__ import annotations

import numpy as np

from . import splines as sp
from .packer import bias_chain_sqrt_info, imu_bias_index
from .window import PK_POS, PK_ROT, Window

# reference config/ct_odometry_tumrs.yaml:23-28 (camera -> IMU)
_R_CI = np.array([[-0.00276873, -0.999936, -0.0110011],
                  [-0.999987, 0.00281495, -0.00418819],
                  [0.00421888, 0.0109894, -0.999931]])
_P_CI = np.array([0.00699407, -0.0570823, -0.0422772])

CONFIGS = {
    # BASELINE.json configs[0..4] (configs[3] = 64 x config2 with seeds 1000..1063)
    "config1": dict(F=11, L=50, M=500),
    "config2": dict(F=11, L=200, M=2000),
    "config3": dict(F=11, L=300, M=2000, img_h=640, ld_true=3.0e-5),
    "config5": dict(F=31, L=1000, M=6000),
    # SURVEY's recipe anchors every landmark in frames 0..7 (feature_manager.h:58-65 is written for WINDOW_SIZE = 10), which leaves frames
    # 16..30 of a 30-KF window without a single visual factor; this variant anchors landmark l in frame l mod (F - 3): tracks all along the window
    "config5_spread": dict(F=31, L=1000, M=6000, anchor_frames="spread"),
    # the reference's native TUM-RSVI operating point: 11 frames at 10 Hz (config/tumrs/cam_tumrs.yaml:25), IMU at 200 Hz
    # (~10 samples per (segment, bias) group at the 0.05 s knot spacing of config/ct_odometry_tumrs.yaml:13), at most 150 tracked
    # features per frame (cam_tumrs.yaml:23 max_cnt) with track lengths of 2..11 frames (cut at the window end), WINDOW_SIZE = 10 (parameters.h:8)
    "tumrs": dict(F=11, L=190, M=200, track=(2, 7, 10)),
    # tiny case for unit tests
    "tiny": dict(F=5, L=12, M=80),
}


def _extrinsic_quat():
    U, _, Vt = np.linalg.svd(_R_CI)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R = -R
    return sp.R_to_quat(R), R


def make_window(config: str = "config2", seed: int = 1000, *, with_prior: bool = True, return_truth: bool = False,
                **overrides):
    """Build one synthetic window.  Returns Window (initial guess) [, truth Window]."""
    cfg = dict(F=11, L=200, M=2000, img_w=1280, img_h=1024, focal=740.0, ld_true=2.94737e-5,
               dt_ns=50_000_000, frame_dt_ns=100_000_000, pix_sigma=0.5, track=(3, 1, 6), anchor_frames="first8")
    cfg.update(CONFIGS[config])
    cfg.update(overrides)
    F, L, M = cfg["F"], cfg["L"], cfg["M"]
    dt_ns, fdt = cfg["dt_ns"], cfg["frame_dt_ns"]
    rng = np.random.Generator(np.random.Philox(key=int(seed)))

    frame_t = np.arange(F, dtype=np.int64) * fdt
    t_last = int(frame_t[-1])
    K = int(-(-(t_last + 40_000_000) // dt_ns)) + 3          # extendKnotsTo(t_img + 0.04 s), odometry_manager.cpp:253
    # --- ground-truth knots
    xi = rng.normal(0.0, 0.05, (K, 3))
    vk = rng.normal(0.0, 0.03, (K, 3)) + np.array([0.02, 0.01, 0.0])
    quat = np.zeros((K, 4)); quat[0, 3] = 1.0
    pos = np.zeros((K, 3))
    for k in range(K - 1):
        quat[k + 1] = sp.qmul(quat[k], sp.qexp(xi[k]))
        quat[k + 1] /= np.linalg.norm(quat[k + 1])
        pos[k + 1] = pos[k] + vk[k]
    bias_true = np.tile(rng.normal(0.0, 0.01, 6), (F, 1))       # constant true bias
    gravity = np.array([0.0, 0.0, 9.80766])
    q_CI, R_CI = _extrinsic_quat()
    ld_true = float(cfg["ld_true"])

    # --- IMU (uniform rate over [0, t_last))
    imu_t = (np.arange(M, dtype=np.int64) * t_last) // M
    ev = sp.eval_spline(quat, pos, 0, dt_ns, imu_t, want=("q", "a", "w"))
    R_wt = sp.quat_to_R(ev["q"])
    gyro = ev["w"] + bias_true[0, :3] + rng.normal(0.0, 4e-3, (M, 3))
    acc = np.einsum("nji,nj->ni", R_wt, ev["a"] + gravity) + bias_true[0, 3:] + rng.normal(0.0, 8e-2, (M, 3))
    imu_bias = imu_bias_index(imu_t, frame_t)
    bc_w = bias_chain_sqrt_info(imu_t, frame_t, 2.0e-5, 4.0e-4)

    # --- landmarks / rolling-shutter observations
    W_img, H_img, f = cfg["img_w"], cfg["img_h"], cfg["focal"]
    cx, cy = W_img / 2.0, H_img / 2.0

    def cam_pose(tau_ns):
        """spline pose at (n,) times: rotations (n,3,3), positions (n,3)."""
        e = sp.eval_spline(quat, pos, 0, dt_ns, tau_ns, want=("q", "p"))
        return sp.quat_to_R(e["q"]), e["p"]

    def project(Xw, t_frame):
        """Rolling-shutter projection of (n,3) world points into the frames starting at t_frame (n,): fixed point on the row
        time (6 rounds; an element stops once its row moved by < 1e-7).  Returns u, v, valid."""
        n = Xw.shape[0]
        v = np.full(n, cy); u = np.zeros(n)
        live = np.ones(n, bool); ok = np.ones(n, bool)
        for _ in range(6):
            if not live.any():
                break
            tau = t_frame[live] + np.round(v[live] * ld_true * 1e9).astype(np.int64)
            R, p = cam_pose(tau)
            Xc = np.einsum("ji,nj->ni", R_CI, np.einsum("nji,nj->ni", R, Xw[live] - p) - _P_CI)
            z = np.where(Xc[:, 2] < 0.3, 1.0, Xc[:, 2])
            un, vn = f * Xc[:, 0] / z + cx, f * Xc[:, 1] / z + cy
            good = (Xc[:, 2] >= 0.3) & (vn >= 2) & (vn < H_img - 2) & (un >= 2) & (un < W_img - 2)
            idx = np.flatnonzero(live)
            done = good & (np.abs(vn - v[idx]) < 1e-7)
            u[idx] = un; v[idx] = np.where(good, vn, v[idx])
            ok[idx[~good]] = False
            live[idx[~good | done]] = False
        return u, v, ok

    # landmarks are placed by rejection, all still-unplaced ones per round (vectorised; deterministic for a given seed)
    n_anchor = max(F - 3, 1) if cfg["anchor_frames"] == "spread" else min(8, max(F - 2, 1))
    anchors = np.arange(L) % n_anchor                            # anchor frame < WINDOW_SIZE-2 (feature_manager.h:58-65)
    t_base, t_mul, t_mod = cfg["track"]                          # track length of landmark l: base + (mul l) mod `mod`, cut at the window
    n_obs = np.minimum(F - anchors, t_base + (t_mul * np.arange(L)) % t_mod)
    max_obs = int(n_obs.max()) if L else 0
    rho_true = np.zeros(L)
    anchor_uv = np.zeros((L, 2)); obs_uv = np.zeros((L, max(max_obs, 1), 2))
    todo = np.arange(L)
    for _attempt in range(200):
        if todo.size == 0:
            break
        n = todo.size
        dr = rng.uniform(size=(n, 3))
        u0, v0, depth = 40 + (W_img - 80) * dr[:, 0], 40 + (H_img - 80) * dr[:, 1], 2.0 + 6.0 * dr[:, 2]
        tau = frame_t[anchors[todo]] + np.round(v0 * ld_true * 1e9).astype(np.int64)
        R, p = cam_pose(tau)
        Xc = depth[:, None] * np.stack([(u0 - cx) / f, (v0 - cy) / f, np.ones(n)], 1)
        Xw = np.einsum("nij,nj->ni", R, Xc @ R_CI.T + _P_CI) + p
        good = np.ones(n, bool)
        uvs = np.zeros((n, max(max_obs, 1), 2))
        for k in range(1, max_obs):
            sel = np.flatnonzero((k < n_obs[todo]) & good)
            if sel.size == 0:
                continue
            uu, vv, ok = project(Xw[sel], frame_t[anchors[todo[sel]] + k])
            uvs[sel, k, 0] = uu; uvs[sel, k, 1] = vv
            good[sel[~ok]] = False
        placed = todo[good]
        rho_true[placed] = 1.0 / depth[good]
        anchor_uv[placed, 0] = u0[good]; anchor_uv[placed, 1] = v0[good]
        obs_uv[placed] = uvs[good]
        todo = todo[~good]
    if todo.size:
        raise RuntimeError("could not place landmark")
    sig = cfg["pix_sigma"]
    noise = rng.normal(0.0, sig, (L, max(max_obs, 1), 2))
    v_lm, v_ti, v_tj, v_rowi, v_rowj, v_pi, v_pj = [], [], [], [], [], [], []
    for l in range(L):
        a = int(anchors[l])
        ua, va = anchor_uv[l, 0] + noise[l, 0, 0], anchor_uv[l, 1] + noise[l, 0, 1]
        for k in range(1, int(n_obs[l])):
            uo, vo = obs_uv[l, k, 0] + noise[l, k, 0], obs_uv[l, k, 1] + noise[l, k, 1]
            v_lm.append(l)
            v_ti.append(frame_t[a]); v_tj.append(frame_t[a + k])
            v_rowi.append(int(round(va))); v_rowj.append(int(round(vo)))
            v_pi.append([(ua - cx) / f, (va - cy) / f]); v_pj.append([(uo - cx) / f, (vo - cy) / f])

    # --- initial guess
    dq = rng.normal(0.0, 0.01, (K, 3))
    dp = rng.normal(0.0, 0.02, (K, 3))
    quat0 = sp.qmul(quat, sp.qexp(dq))
    quat0 /= np.linalg.norm(quat0, axis=1, keepdims=True)
    pos0 = pos + dp
    rho0 = rho_true * (1.0 + 0.1 * np.clip(rng.normal(0.0, 1.0, L), -2.5, 2.5))

    def build(q, p, b, r, ld):
        w = Window(t0_ns=0, dt_ns=dt_ns, quat=q.copy(), pos=p.copy(), bias=b.copy(), rho=r.copy(), ld=ld,
                   ld_lo=0.0, ld_hi=3.5e-5, q_CI=q_CI, p_CI=_P_CI.copy(), gravity=gravity,
                   imu_w=np.array([250.0] * 3 + [12.5] * 3), img_w=800.0, cauchy_a=2.0,
                   imu_t=imu_t, imu_gyro=gyro, imu_acc=acc, imu_bias=imu_bias,
                   bc_i=np.arange(F - 1), bc_j=np.arange(1, F), bc_w=bc_w,
                   v_lm=v_lm, v_ti=v_ti, v_tj=v_tj, v_rowi=v_rowi, v_rowj=v_rowj, v_pi=v_pi, v_pj=v_pj)
        if with_prior:
            # gauge anchor: J0 = 1e3*I on rot+pos of knots 0..3, linearised at the truth
            n = 24
            w.pJ0 = 1e3 * np.eye(n); w.pr0 = np.zeros(n)
            kinds, idxs, offs, x0 = [], [], [], []
            for k in range(4):
                kinds += [PK_ROT, PK_POS]; idxs += [k, k]; offs += [6 * k, 6 * k + 3]
                x0 += [quat[k].tolist(), pos[k].tolist() + [0.0]]
            w.p_kind, w.p_index, w.p_off, w.p_x0 = kinds, idxs, offs, np.array(x0)
        return w.normalize()

    w0 = build(quat0, pos0, np.zeros((F, 6)), rho0, 0.0)
    if return_truth:
        return w0, build(quat, pos, bias_true, rho_true, ld_true)
    return w0
