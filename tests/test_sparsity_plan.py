"""CPU: the sparsity plan of the reduced system (csrc/host_pack.hpp: plan_sparsity; the reference leaves this to Ceres'
SPARSE_NORMAL_CHOLESKY, trajectory_estimator.cpp:371-384) against the DENSE normal equations of the oracle: every non-zero of W lies
inside its landmark's planned knot span (for the line delay at both ends of its box and in between), inside the per-tile row ranges,
and every non-zero of the Schur complement S = Hpp - W Hll^-1 W^T lies inside the planned envelope -- for the benchmarked shapes, the
rolling-shutter stress shape, random factor structures and windows with unobserved landmarks."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROCM_INC = "/opt/rocm/include"


@pytest.fixture(scope="module")
def hp():
    if not os.path.isdir(ROCM_INC):
        pytest.skip("HIP headers not found")
    out = os.path.join(HERE, "_build", "libhostplan.so")
    src = os.path.join(HERE, "host_plan_check.cpp")
    hdrs = [os.path.join(HERE, "..", "ctrl-vio_amd", "csrc", f) for f in ("host_pack.hpp", "device_types.hpp")] + [os.path.join(HERE, "..", "include", "ctvio.h")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or any(os.path.getmtime(f) > os.path.getmtime(out) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I", ROCM_INC, "-o", out, src, "-L/opt/rocm/lib", "-lamdhip64",
                               "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    return C.CDLL(out)


def plan(hp, cv, w, dense=False, full=False):
    keep = []
    cw = cv.capi.to_cwindow(w, keep)
    L, ntr = w.L, w.P // 16 + 1
    a = {k: np.zeros(max(n, 1), np.int32) for k, n in (("lm_pos", L), ("lm_at", L), ("klo", L), ("khi", L), ("tl_beg", ntr), ("tl_end", ntr), ("env", ntr), ("env_tile", ntr))}
    Lobs = C.c_int32(); ms = C.c_int32(); nt = C.c_int32()
    err = C.create_string_buffer(256)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    rc = hp.hp_sparsity(C.byref(cw), int(dense), int(full), p(a["lm_pos"]), p(a["lm_at"]), p(a["klo"]), p(a["khi"]), p(a["tl_beg"]), p(a["tl_end"]), p(a["env"]), p(a["env_tile"]),
                        C.byref(Lobs), C.byref(ms), C.byref(nt), err, 256)
    assert rc == 0, err.value.decode()
    assert nt.value == ntr
    out = {k: v[: (L if k in ("lm_pos", "lm_at", "klo", "khi") else ntr)] for k, v in a.items()}
    out.update(Lobs=Lobs.value, max_span=ms.value)
    return out


def check_against_dense(cv, oracle, w, pl, lds):
    P, L, K = w.P, w.L, w.K
    K6 = 6 * K
    pos, at = pl["lm_pos"], pl["lm_at"]
    assert sorted(at.tolist()) == list(range(L)) and all(pos[at[r]] == r for r in range(L))
    key = [(int(pl["klo"][r]), int(pl["khi"][r])) for r in range(L)]
    assert key == sorted(key)                                   # rows sorted by (first, last) knot; unobserved (K, -1) last
    observed = np.zeros(L, bool); observed[w.v_lm] = True
    assert pl["Lobs"] == int(observed.sum()) and all(observed[at[r]] for r in range(pl["Lobs"]))
    assert all(pl["khi"][r] == -1 and pl["klo"][r] == K for r in range(pl["Lobs"], L))
    nzS = np.zeros((P, P), bool)
    for ld in lds:
        ww = w.copy(); ww.ld = ld
        H, g, cost = oracle.OracleWindow(ww).build_normal()
        Hpp, W, Hll = H[:P, :P], H[:P, P:], np.diag(H)[P:]
        nzW = W != 0.0
        for l in range(L):
            r = pos[l]
            cols = np.flatnonzero(nzW[:, l])
            if cols.size == 0:
                continue
            assert pl["khi"][r] >= 0
            knot = cols[cols < K6]
            assert knot.size == 0 or (knot.min() >= 6 * pl["klo"][r] and knot.max() < 6 * pl["khi"][r] + 6), (l, knot.min(), knot.max(), pl["klo"][r], pl["khi"][r])
            assert np.all((cols < K6) | (cols == P - 1))        # W has knot columns and the line-delay column only
            for c in np.unique(cols // 16):                     # per-tile row ranges
                assert pl["tl_beg"][c] <= r < pl["tl_end"][c], (l, r, c, pl["tl_beg"][c], pl["tl_end"][c])
        dinv = np.where(Hll > 0, 1.0 / np.where(Hll > 0, Hll, 1.0), 0.0)
        nzS |= (Hpp != 0.0) | ((np.abs(W) * dinv) @ np.abs(W).T > 0.0)      # structural: no cancellation
    ntr = P // 16 + 1
    env = pl["env"]
    for r in range(ntr):
        assert env[r] % 2 == 0 and (r < 2 and env[r] == 0 or r >= 2 and env[r] <= 2 * (r // 2) - 2)     # 32-column panels, next diagonal block takes part
        rows = nzS[16 * r:min(16 * r + 16, P), :]
        cols = np.flatnonzero(rows.any(axis=0))
        cols = cols[cols <= 16 * r + 15]
        if cols.size:
            assert cols.min() // 16 >= env[r], (r, cols.min(), env[r])
    assert env[P // 16] == 0                                    # the rhs row rides along as row P: dense
    et = pl["env_tile"]                                         # the raw tile envelope: tight where the aligned one is padded
    for r in range(ntr):
        cols = np.flatnonzero(nzS[16 * r:min(16 * r + 16, P), :16 * r + 16].any(axis=0))
        assert et[r] <= r and (cols.size == 0 or cols.min() // 16 >= et[r]) and env[r] <= et[r]
    return nzS, env


@pytest.mark.parametrize("cfg,seed", [("tiny", 3), ("config1", 1001), ("config2", 1000), ("config3", 1003), ("tumrs", 1002)])
def test_plan_covers_the_dense_structure(hp, cv, oracle, cfg, seed):
    w = cv.synth.make_window(cfg, seed=seed)
    pl = plan(hp, cv, w)
    check_against_dense(cv, oracle, w, pl, [w.ld_lo, 0.5 * (w.ld_lo + w.ld_hi), w.ld_hi])
    assert pl["max_span"] == int((pl["khi"] - pl["klo"] + 1)[:pl["Lobs"]].max())
    wf = w.copy(); wf.fix_ld = True; wf.ld = 2.0e-5              # a fixed line delay: the spans are exact for it
    plf = plan(hp, cv, wf)
    check_against_dense(cv, oracle, wf, plf, [wf.ld])
    assert np.all(plf["khi"] - plf["klo"] <= pl["khi"] - pl["klo"])


def test_plan_of_a_long_window_is_sparse(hp, cv, oracle):
    """K = 34 (16 frames), landmarks anchored all along the window: the envelope of the late knots starts late, short spans."""
    w = cv.synth.make_window("config1", seed=1400, F=16, L=60, M=750)
    w.v_ti = w.v_ti.copy()
    pl = plan(hp, cv, w)
    nzS, env = check_against_dense(cv, oracle, w, pl, [w.ld_lo, w.ld_hi])
    assert env[: 6 * w.K // 16].max() >= 2                      # not dense
    tiles = sum(r - env[r] + 1 for r in range(len(env)))
    assert tiles < len(env) * (len(env) + 1) // 2
    pd = plan(hp, cv, w, dense=True)
    assert not pd["env"].any() and np.array_equal(pd["lm_pos"], pl["lm_pos"]) and np.array_equal(pd["env_tile"], pl["env_tile"])
    pf = plan(hp, cv, w, dense=True, full=True)                 # CTVIO_DENSE: every tile with products multiplies every observed row
    assert not pf["env"].any() and not pf["env_tile"].any()
    assert all((b, e) in ((0, 0), (0, pf["Lobs"])) for b, e in zip(pf["tl_beg"], pf["tl_end"]))


def test_plan_random_structures_and_unobserved_landmarks(hp, cv, oracle):
    rng = np.random.default_rng(5)
    for trial in range(4):
        w = cv.synth.make_window("config1", seed=1500 + trial)
        keep = rng.random(w.V) < 0.6                             # drop blocks at random: some landmarks lose every observation
        keep[w.v_lm == 7] = False
        for a in ("v_lm", "v_ti", "v_tj", "v_rowi", "v_rowj", "v_pi", "v_pj"):
            setattr(w, a, getattr(w, a)[keep])
        w.normalize()
        if trial % 2:
            w.pJ0 = np.zeros((0, 0)); w.pr0 = np.zeros(0)
            w.p_kind = w.p_kind[:0]; w.p_index = w.p_index[:0]; w.p_off = w.p_off[:0]; w.p_x0 = w.p_x0[:0]
            w.normalize()
        pl = plan(hp, cv, w)
        assert pl["Lobs"] < w.L
        check_against_dense(cv, oracle, w, pl, [w.ld_lo, w.ld_hi])


@pytest.mark.parametrize("cfg,seed,kw", [("config2", 1000, {}), ("config5", 1011, {}), ("config5_spread", 1011, {}), ("config3", 1003, {}), ("tumrs", 1002, {}),
                                         ("config1", 1400, dict(F=16, L=60, M=750)), ("tiny", 3, {})])
def test_python_mirror_equals_the_cxx_planner(hp, cv, cfg, seed, kw):
    """ctrl-vio_amd/packer.py: landmark_spans / reduced_system_envelope (bench.py's non-zero flop and envelope byte counts) entry for entry
    against csrc/host_pack.hpp: plan_sparsity."""
    w = cv.synth.make_window(cfg, seed=seed, **kw)
    pl = plan(hp, cv, w)
    klo, khi = cv.packer.landmark_spans(w)
    assert np.array_equal(klo[pl["lm_at"]], pl["klo"]) and np.array_equal(khi[pl["lm_at"]], pl["khi"])
    assert np.array_equal(cv.packer.reduced_system_envelope(w), pl["env"])
    nnz = np.where(khi >= 0, 6 * (khi - klo + 1) + 1, 0)
    assert cv.packer.schur_nonzero_flops(w) == int((nnz * (nnz + 1)).sum()) <= (6 * w.K + 1) * (6 * w.K + 2) * w.L
    assert cv.packer.envelope_entries(w) <= cv.packer.envelope_entries(w, dense=True) == w.P * (w.P + 1) // 2
