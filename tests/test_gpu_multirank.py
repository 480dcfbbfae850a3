"""GPU: the N > 1 paths, exercised on ONE device so that the first run on an 8-GPU node is not the first run at all.

* `bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank): both ranks map to device 0
  (rank -> local % device_count; the collectives go over gloo when ranks share a device, over RCCL otherwise), windows shard by id,
  barrier + max-over-ranks timing + all-gather of the per-window records run, and rank 0 prints the contract's JSON line.
* `ctvio_solve_sharded` with two shards on one device (TEST-ONLY switch CTVIO_SHARD_OVERSUBSCRIBE=1): two host threads, two solver
  handles, results back in the caller's window order and equal to the single-handle solve.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_line(args, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # ONE JSON line, from rank 0 only
    assert p.stdout.strip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096     # the LAST stdout line, inside the driver's 8 KB tail
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in line, k
    with open(os.path.join(ROOT, line["details_file"])) as f:      # everything else (full precision) sits in the side file
        d = json.load(f)
    for k in ("value", "ms_per_step"):
        assert line[k] == pytest.approx(d[k], rel=1e-5)
    assert (line["n_gpus"], line["steps"], line["warmup"], line["scaling"], line["unit"]) == (d["n_gpus"], d["steps"], d["warmup"], d["scaling"], d["unit"])
    return d


def test_bench_two_ranks_on_one_device():
    d = _bench_line(["--gpus", "2", "--quick", "--windows", "128", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["unit"] == "solves/s" and d["higher_is_better"] is True
    assert d["ms_per_step"] > 0 and abs(d["value"] - 2 * 128 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]   # whole-job aggregate
    ss = d["solve_summary"]
    assert ss["windows"] == 256 and ss["window_ids_gathered_once"] is True          # every window id of both ranks, exactly once
    assert "all_gather" in ss["gathered_with"]
    assert d["config"]["windows_per_gpu_per_step"] == 128 and "mod 2" in d["config"]["sharding"]


def test_bench_eight_ranks_on_one_device_config4_shape():
    """BASELINE configs[3] as literally written: 64 windows over 8 ranks = 8 per rank, launched as the driver launches an 8-GPU run (all eight
    ranks share device 0 here; collectives over gloo).  The contract line must say n_gpus 8, carry every window id exactly once, one rate per
    rank (the first real SCALE run shows imbalance directly) and the per-step latency of the 8-per-rank shape."""
    d = _bench_line(["--gpus", "8", "--quick", "--windows", "8", "--streams", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=1500)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["steps"] == 2
    assert abs(d["value"] - 8 * 8 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]          # whole-job aggregate over the 8 ranks
    ss = d["solve_summary"]
    assert ss["windows"] == 64 and ss["window_ids_gathered_once"] is True
    assert len(d["per_rank_solves_per_s"]) == 8 and all(r > 0 for r in d["per_rank_solves_per_s"])
    assert max(8 * 2 / r for r in d["per_rank_solves_per_s"]) == pytest.approx(2 * d["ms_per_step"] / 1e3, rel=1e-9)   # the slowest rank sets the step time
    lat = d["small_batch_latency"]
    assert lat["windows_per_rank"] == 8 and len(lat["end_to_end_ms_per_step_by_rank"]) == 8 and max(lat["end_to_end_ms_per_step_by_rank"]) == pytest.approx(d["ms_per_step"], rel=1e-9)
    assert "mod 8" in d["config"]["sharding"]


def test_sharded_entry_two_shards_on_one_device(cv):
    lib = cv.capi.load_library()
    n = 7
    ws = [cv.synth.make_window("tiny", seed=300 + i) for i in range(n)]
    keep = []
    arr = (cv.capi.CWindow * n)()
    for i, w in enumerate(ws):
        arr[i] = cv.capi.to_cwindow(w, keep)
    K = sum(w.K for w in ws); F = sum(w.F for w in ws); L = sum(w.L for w in ws)
    opt = cv.capi.Options()
    lib.ctvio_default_options(C.byref(opt))
    os.environ["CTVIO_SHARD_OVERSUBSCRIBE"] = "1"
    try:
        assert lib.ctvio_shards_used(2, n) == 2 and lib.ctvio_shards_used(2, 1) == 1
        res = {}
        for tol in (1e4, 1e1):                         # the second call changes an option (the initial trust-region radius): the kept
            opt.initial_radius = tol                   # handles must be re-created, or the second result would repeat the first
            sm = (cv.capi.Summary * n)()
            q = np.zeros((K, 4)); p = np.zeros((K, 3)); b = np.zeros((F, 6)); r = np.zeros(L); ld = np.zeros(n)
            cv.capi.check(lib.ctvio_solve_sharded(C.byref(opt), 2, n, C.cast(arr, C.c_void_p), 15, C.cast(sm, C.c_void_p),
                                                  cv.capi._p(q), cv.capi._p(p), cv.capi._p(b), cv.capi._p(r), cv.capi._p(ld)))
            res[tol] = (q, p, b, r, ld, [s.as_dict() for s in sm])
    finally:
        del os.environ["CTVIO_SHARD_OVERSUBSCRIBE"]
        lib.ctvio_sharded_release()
    for tol in res:
        ref = [w.copy() for w in ws]
        with cv.Solver(initial_radius=tol) as s:
            s.set_windows(ref)
            sms = s.solve(15)
        q, p, b, r, ld, sm = res[tol]
        k0 = f0 = l0 = 0
        for i, w in enumerate(ref):                    # the caller's window order, whatever shard solved the window
            assert sm[i]["iterations"] == sms[i]["iterations"] and sm[i]["final_cost"] == pytest.approx(sms[i]["final_cost"], rel=1e-12)
            np.testing.assert_allclose(q[k0:k0 + w.K], w.quat, atol=1e-12)
            np.testing.assert_allclose(p[k0:k0 + w.K], w.pos, atol=1e-12)
            np.testing.assert_allclose(r[l0:l0 + w.L], w.rho, atol=1e-12)
            assert ld[i] == pytest.approx(w.ld, abs=1e-15)
            k0 += w.K; f0 += w.F; l0 += w.L
    # a different initial radius gives different iterates: the option change really took effect
    assert any(a["final_cost"] != c["final_cost"] or a["iterations"] != c["iterations"] for a, c in zip(res[1e4][5], res[1e1][5]))


def test_process_exit_without_sharded_release(cv):
    """The sharded entry keeps one idle host thread per shard alive between calls; a process that exits WITHOUT ctvio_sharded_release must
    still exit cleanly (the workers are told to quit and joined by their destructors)."""
    code = r'''
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.environ["CTV_ROOT"])
import numpy as np
cv = importlib.import_module("ctrl-vio_amd")
lib = cv.capi.load_library()
ws = [cv.synth.make_window("tiny", seed=310 + i) for i in range(4)]
keep = []; arr = (cv.capi.CWindow * 4)()
for i, w in enumerate(ws): arr[i] = cv.capi.to_cwindow(w, keep)
opt = cv.capi.Options(); lib.ctvio_default_options(C.byref(opt))
sm = (cv.capi.Summary * 4)()
cv.capi.check(lib.ctvio_solve_sharded(C.byref(opt), 2, 4, C.cast(arr, C.c_void_p), 5, C.cast(sm, C.c_void_p), None, None, None, None, None))
assert all(s.as_dict()["iterations"] > 0 for s in sm)
print("SHARDED_EXIT_OK")
'''
    env = dict(os.environ, CTV_ROOT=ROOT, CTVIO_SHARD_OVERSUBSCRIBE="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0 and "SHARDED_EXIT_OK" in p.stdout, (p.returncode, p.stderr[-2000:])
