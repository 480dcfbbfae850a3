"""Host packing rules around the solve, against hand-worked cases of the reference's loops."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv = importlib.import_module("ctrl-vio_amd")
pk = cv.packer


def test_landmark_candidate_rule():
    W = 10
    assert pk.is_landmark_candidate(2, 0, W) and pk.is_landmark_candidate(5, W - 3, W)
    assert not pk.is_landmark_candidate(1, 0, W)          # single observation
    assert not pk.is_landmark_candidate(4, W - 2, W)      # starts too late (start_frame < WINDOW_SIZE - 2)


def test_pack_visual_order_and_rounding():
    W = 10
    ts = np.arange(W + 1, dtype=np.int64) * 100_000_000
    tracks = [
        dict(start_frame=0, points=[[0.1, 0.2, 1.0], [0.11, 0.21, 1.0], [0.12, 0.22, 1.0]], uv=[[320.0, 100.5], [321.0, 101.49], [322.0, 99.5]], depth=4.0),
        dict(start_frame=3, points=[[0.0, 0.0, 1.0]], uv=[[1.0, 2.0]], depth=2.0),                                     # 1 observation: skipped
        dict(start_frame=8, points=[[0.3, 0.3, 1.0], [0.31, 0.3, 1.0]], uv=[[5.0, 6.0], [7.0, 8.0]], depth=3.0),       # starts too late: skipped
        dict(start_frame=2, points=[[0.2, -0.1, 2.0], [0.4, -0.2, 2.0]], uv=[[10.0, 479.6], [11.0, 0.4]], depth=-1.0),
    ]
    p = pk.pack_visual(tracks, ts, W)
    assert p["track_of_landmark"].tolist() == [0, 3]
    np.testing.assert_allclose(p["rho"], [0.25, -1.0])
    assert p["v_lm"].tolist() == [0, 0, 1]
    assert p["v_ti"].tolist() == [0, 0, 200_000_000] and p["v_tj"].tolist() == [100_000_000, 200_000_000, 300_000_000]
    assert p["v_rowi"].tolist() == [101, 101, 480]        # std::round(100.5) = 101, round(479.6) = 480
    assert p["v_rowj"].tolist() == [101, 100, 0]          # 101.49 -> 101, 99.5 -> 100, 0.4 -> 0
    np.testing.assert_allclose(p["v_pi"], [[0.1, 0.2], [0.1, 0.2], [0.1, -0.05]])
    np.testing.assert_allclose(p["v_pj"], [[0.11, 0.21], [0.12, 0.22], [0.2, -0.1]])


def test_imu_window_and_depth_copy_back():
    t = np.array([0, 49_999_999, 50_000_000, 120_000_000, 1_000_000_000], np.int64)
    lo = pk.opt_min_time(120_000_000, 0, 50_000_000)
    assert lo == 100_000_000
    assert pk.imu_in_window(t, lo, 1_000_000_000).tolist() == [False, False, False, True, False]
    depth, ok = pk.depths_from_solution([0.5, -0.25, 2.0])
    np.testing.assert_allclose(depth, [2.0, -4.0, 0.5])
    assert ok.tolist() == [True, False, True]
