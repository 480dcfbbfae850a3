"""Host-side (NumPy, fp64) uniform cubic SO(3)xR^3 B-spline evaluation.

Used by the synthetic-window generator and by host utilities that need the trajectory at a few
timestamps without a device round trip.  Same conventions as the reference's `Se3Spline<4>`
(src/spline/se3_spline.h:391-399, so3_spline.h:240-322, rd_spline.h:229-259): quaternions are
(x,y,z,w), cumulative basis for rotation, plain basis for translation, times are int64 ns.
The device-side equivalent is `ctvio_spline_eval` (include/ctvio.h).
"""
from __future__ import annotations

import numpy as np

M_BLEND = np.array([[1, -3, 3, -1], [4, 0, -6, 3], [1, 3, 3, -3], [0, 0, 0, 1]], np.float64) / 6.0
M_CUMUL = np.array([[6, 0, 0, 0], [5, 3, -3, 1], [1, 3, 3, -2], [0, 0, 0, 1]], np.float64) / 6.0


def basis(u: np.ndarray, deriv: int, cumulative: bool, inv_dt: float) -> np.ndarray:
    """(n,4) blending coefficients of the deriv-th time derivative at normalised time u."""
    u = np.asarray(u, np.float64)
    if deriv == 0:
        pw = np.stack([np.ones_like(u), u, u * u, u * u * u], -1)
    elif deriv == 1:
        pw = np.stack([np.zeros_like(u), np.ones_like(u), 2 * u, 3 * u * u], -1)
    elif deriv == 2:
        pw = np.stack([np.zeros_like(u), np.zeros_like(u), 2 * np.ones_like(u), 6 * u], -1)
    else:
        raise ValueError(deriv)
    M = M_CUMUL if cumulative else M_BLEND
    return (pw @ M.T) * (inv_dt ** deriv)


def qmul(a, b):
    ax, ay, az, aw = np.moveaxis(np.asarray(a), -1, 0)
    bx, by, bz, bw = np.moveaxis(np.asarray(b), -1, 0)
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz], -1)


def qconj(a):
    a = np.asarray(a)
    return a * np.array([-1.0, -1.0, -1.0, 1.0])


def qrot(q, v):
    qv = np.asarray(q)[..., :3]
    w = np.asarray(q)[..., 3:4]
    uv = 2.0 * np.cross(qv, v)
    return v + w * uv + np.cross(qv, uv)


def qexp(w):
    w = np.asarray(w, np.float64)
    th = np.linalg.norm(w, axis=-1, keepdims=True)
    small = th < 1e-10
    ths = np.where(small, 1.0, th)
    im = np.where(small, 0.5 - th * th / 48.0, np.sin(0.5 * ths) / ths)
    re = np.where(small, 1.0 - th * th / 8.0, np.cos(0.5 * th))
    return np.concatenate([im * w, re], -1)


def qlog(q):
    q = np.asarray(q, np.float64)
    n = np.linalg.norm(q[..., :3], axis=-1, keepdims=True)
    w = q[..., 3:4]
    small = n < 1e-10
    ns = np.where(small, 1.0, n)
    f = np.where(small, 2.0 / w, 2.0 * np.arctan(ns / w) / ns)
    return f * q[..., :3]


def quat_to_R(q):
    q = np.asarray(q, np.float64)
    x, y, z, w = np.moveaxis(q, -1, 0)
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def R_to_quat(R):
    """Single 3x3 rotation -> (x,y,z,w)."""
    R = np.asarray(R, np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


def t_index(t_ns, t0_ns: int, dt_ns: int):
    """time -> (segment, u) in integer ns arithmetic (reference spline_segment.h:83-85)."""
    st = np.asarray(t_ns, np.int64) - np.int64(t0_ns)
    s = st // np.int64(dt_ns)
    u = (st % np.int64(dt_ns)).astype(np.float64) / float(dt_ns)
    return s.astype(np.int64), u


def eval_spline(quat, pos, t0_ns, dt_ns, t_ns, want=("q", "p", "v", "a", "w")):
    """Evaluate the spline at times t_ns.  Returns dict with q (n,4), p, v, a (world), w (body)."""
    t_ns = np.atleast_1d(np.asarray(t_ns, np.int64))
    s, u = t_index(t_ns, t0_ns, dt_ns)
    K = quat.shape[0]
    if np.any(s < 0) or np.any(s + 3 >= K):
        raise ValueError("spline time out of range")
    inv_dt = 1e9 / float(dt_ns)
    idx = s[:, None] + np.arange(4)[None, :]
    out = {}
    P4 = pos[idx]                                   # (n,4,3)
    if "p" in want:
        out["p"] = np.einsum("ni,nij->nj", basis(u, 0, False, inv_dt), P4)
    if "v" in want:
        out["v"] = np.einsum("ni,nij->nj", basis(u, 1, False, inv_dt), P4)
    if "a" in want:
        out["a"] = np.einsum("ni,nij->nj", basis(u, 2, False, inv_dt), P4)
    if "q" in want or "w" in want:
        Q4 = quat[idx]                              # (n,4,4)
        lam = basis(u, 0, True, inv_dt)
        dlam = basis(u, 1, True, inv_dt)
        q = Q4[:, 0]
        w = np.zeros((t_ns.shape[0], 3))
        for i in range(3):
            d = qlog(qmul(qconj(Q4[:, i]), Q4[:, i + 1]))
            e = qexp(d * lam[:, i + 1:i + 2])
            q = qmul(q, e)
            w = qrot(qconj(e), w) + d * dlam[:, i + 1:i + 2]
        out["q"] = q / np.linalg.norm(q, axis=-1, keepdims=True)
        out["w"] = w
    return out
