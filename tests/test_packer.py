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


def test_cpp_packer_matches_python(tmp_path):
    """include/ctvio_packer.hpp (what a C++ caller of the C ABI uses) against ctrl-vio_amd/packer.py on random inputs."""
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "libhostpacker.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-shared", "-fPIC", "-o", so, os.path.join(root, "tests", "host_packer_check.cpp")])
    lib = C.CDLL(so)
    lib.hp_opt_min_time.restype = C.c_int64
    lib.hp_opt_min_time.argtypes = [C.c_int64] * 3
    rng = np.random.default_rng(11)
    W = 10
    frame_t = np.cumsum(rng.integers(80_000_000, 120_000_000, W + 1)).astype(np.int64)
    imu_t = np.sort(rng.integers(frame_t[0] - 50_000_000, frame_t[-1] + 50_000_000, 500)).astype(np.int64)
    idx = np.zeros(imu_t.size, np.int32)
    lib.hp_bias_index(C.c_int(imu_t.size), imu_t.ctypes.data_as(C.c_void_p), C.c_int(frame_t.size), frame_t.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p))
    assert np.array_equal(idx, pk.imu_bias_index(imu_t, frame_t))
    chain = np.zeros((W, 6))
    lib.hp_bias_chain.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    lib.hp_bias_chain(imu_t.size, imu_t.ctypes.data, frame_t.size, frame_t.ctypes.data, 1e-4, 2e-3, chain.ctypes.data)
    np.testing.assert_allclose(chain, pk.bias_chain_sqrt_info(imu_t, frame_t, 1e-4, 2e-3), rtol=1e-12)
    assert lib.hp_opt_min_time(123_456_789, 3_000_000, 50_000_000) == pk.opt_min_time(123_456_789, 3_000_000, 50_000_000)
    tracks = []
    for _ in range(60):
        n = int(rng.integers(1, 7)); s = int(rng.integers(0, W)); n = min(n, W + 1 - s)
        pts = np.concatenate([rng.normal(0, 0.3, (n, 2)), np.ones((n, 1))], axis=1) * rng.uniform(0.5, 2.0)
        uvp = np.stack([rng.uniform(0, 640, n), rng.integers(0, 960, n) * 0.5], axis=1)   # half-pixel rows exercise the rounding rule
        tracks.append(dict(start_frame=s, points=pts, uv=uvp, depth=float(rng.uniform(-1, 8) or 1.0)))
    ref = pk.pack_visual(tracks, frame_t, W)
    n_obs = np.array([len(t["points"]) for t in tracks], np.int32)
    start = np.array([t["start_frame"] for t in tracks], np.int32)
    depth = np.array([t["depth"] for t in tracks])
    points = np.concatenate([t["points"] for t in tracks]).astype(np.float64)
    uvs = np.concatenate([t["uv"] for t in tracks]).astype(np.float64)
    cap = int(n_obs.sum())
    v_lm = np.zeros(cap, np.int32); v_ti = np.zeros(cap, np.int64); v_tj = np.zeros(cap, np.int64); v_ri = np.zeros(cap, np.int32); v_rj = np.zeros(cap, np.int32)
    v_pi = np.zeros((cap, 2)); v_pj = np.zeros((cap, 2)); rho = np.zeros(len(tracks)); owner = np.zeros(len(tracks), np.int32); n_lm = C.c_int32(0)
    lib.hp_pack_visual.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 11
    V = lib.hp_pack_visual(len(tracks), n_obs.ctypes.data, start.ctypes.data, depth.ctypes.data, points.ctypes.data, uvs.ctypes.data, W, frame_t.ctypes.data,
                           v_lm.ctypes.data, v_ti.ctypes.data, v_tj.ctypes.data, v_ri.ctypes.data, v_rj.ctypes.data, v_pi.ctypes.data, v_pj.ctypes.data,
                           rho.ctypes.data, owner.ctypes.data, C.addressof(n_lm))
    assert V == len(ref["v_lm"]) and n_lm.value == len(ref["rho"]) and V > 20
    for got, key in ((v_lm, "v_lm"), (v_ti, "v_ti"), (v_tj, "v_tj"), (v_ri, "v_rowi"), (v_rj, "v_rowj")):
        assert np.array_equal(got[:V], ref[key]), key
    np.testing.assert_allclose(v_pi[:V], ref["v_pi"], rtol=1e-15); np.testing.assert_allclose(v_pj[:V], ref["v_pj"], rtol=1e-15)
    np.testing.assert_allclose(rho[:n_lm.value], ref["rho"], rtol=1e-15)
    assert np.array_equal(owner[:n_lm.value], ref["track_of_landmark"])


def test_native_tumrs_window_shape(cv):
    """synth "tumrs" = the reference's own operating point (config/tumrs/cam_tumrs.yaml:23-25, config/ct_odometry_tumrs.yaml:13,
    parameters.h:8): 11 frames at 10 Hz, 200 Hz IMU (10 samples per (segment, bias) group), at most 150 features in any frame."""
    w = cv.synth.make_window("tumrs", seed=1000)
    assert (w.F, w.M, w.K) == (11, 200, 24)
    assert np.all(np.diff(w.imu_t) == 5_000_000)
    groups = {}
    for s_, b in zip(w.imu_t // w.dt_ns, w.imu_bias):
        groups[(int(s_), int(b))] = groups.get((int(s_), int(b)), 0) + 1
    assert len(groups) == 20 and set(groups.values()) == {10}
    per = np.zeros(w.F, int)
    for l in range(w.L):
        sel = w.v_lm == l
        for f in set((w.v_tj[sel] // 100_000_000).tolist()) | set((w.v_ti[sel] // 100_000_000).tolist()):
            per[f] += 1
    assert per.max() <= 150 and per.max() >= 120
    n_obs = np.bincount(w.v_lm, minlength=w.L) + 1
    assert n_obs.min() >= 2 and n_obs.max() >= 10


def test_predict_window_handles_the_optional_fields(cv):
    """Solver.predict_window (InitTrajectory's factor set) on a window that carries per-block Cauchy widths and per-knot constancy:
    the widths go with the visual blocks, the constancy mask stays (no GPU needed: this is host-side packing)."""
    w = cv.synth.make_window("tiny", seed=5)
    w.v_cauchy = np.full(w.V, 1.0)
    kc = np.zeros(w.K, np.uint8); kc[3] = 1
    w.knot_const = kc
    w.normalize()
    p = cv.Solver.predict_window(w, fixed_upto=1)
    assert p.V == 0 and p.v_cauchy is None and p.NB == 0 and p.pn == 0
    assert p.lock_bg and p.lock_ba and p.fixed_upto == 1
    assert p.knot_const is not None and p.knot_const.tolist() == kc.tolist()
    assert p.M == w.M and p.K == w.K
