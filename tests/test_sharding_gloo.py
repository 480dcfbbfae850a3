"""CPU, world_size 2 over gloo: windows shard by id, every window is solved exactly once, and the all-gathered
records do not depend on the number of ranks.  (The per-rank solve is stood in for by the oracle here -- this
test is about the partition + collective, which is all the multi-GPU path adds.)"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r'''
import importlib, os, sys
sys.path.insert(0, os.environ["CTV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["CTV_ROOT"], "oracle"))
import numpy as np, torch.distributed as dist
cv = importlib.import_module("ctrl-vio_amd")
import pyctvo
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n = int(os.environ["CTV_N"])
ids = cv.sharding.shard(n, rank, world)
sms = []
for wid in ids:
    w = cv.synth.make_window("tiny", seed=2000 + wid)
    sm = pyctvo.OracleWindow(w).solve(6)
    sms.append(dict(iterations=sm.iterations, termination=cv.capi.TERMINATION[sm.termination], initial_cost=sm.initial_cost,
                    final_cost=sm.final_cost, final_radius=sm.final_radius))
rec = cv.sharding.gather_records(cv.sharding.make_records(ids, sms), n)
if rank == 0:
    np.save(os.environ["CTV_OUT"], rec)
dist.destroy_process_group()
'''


def _run(world, n, out):
    env = dict(os.environ, CTV_ROOT=os.path.dirname(HERE), CTV_N=str(n), CTV_OUT=out)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + world), "-c", WORKER] if False else None
    script = os.path.join(os.path.dirname(out), "worker.py")
    open(script, "w").write(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29510 + world), script]
    subprocess.check_call(cmd, env=env, timeout=600)
    return np.load(out)


def test_partition_covers_every_window_once(cv):
    for n in (1, 5, 64):
        for world in (1, 2, 3, 8):
            ids = sorted(sum((cv.sharding.shard(n, r, world) for r in range(world)), []))
            assert ids == list(range(n))


def test_gather_independent_of_world_size(tmp_path):
    n = 5
    r1 = _run(1, n, str(tmp_path / "r1.npy"))
    r2 = _run(2, n, str(tmp_path / "r2.npy"))
    assert r1.shape == (n, 6) and not np.isnan(r1).any()
    np.testing.assert_array_equal(r1[:, 0], np.arange(n))
    np.testing.assert_allclose(r2, r1, rtol=0, atol=0)
