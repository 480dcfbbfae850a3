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
