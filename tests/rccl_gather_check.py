"""Run by tests/test_gpu_parity.py::test_rccl_gather_world_size_1 in a fresh interpreter (torch initialises its HIP runtime
first, as in bench.py): a sharded batch solved on cuda:0, per-window records all-gathered over the nccl backend (= RCCL on
ROCm) on GPU tensors, every window id must come back exactly once with its values intact."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    cv = importlib.import_module("ctrl-vio_amd")
    n = 5
    ws = [cv.synth.make_window("tiny", seed=40 + i) for i in range(n)]
    ids = cv.sharding.shard(n, rank, world)
    with cv.Solver(device=local) as s:
        s.set_windows([ws[i] for i in ids])
        sms = s.solve(15)
    rec = cv.sharding.gather_records(cv.sharding.make_records(ids, sms), n, device=torch.device("cuda", local))
    assert rec.shape == (n, cv.sharding.RECORD) and not np.isnan(rec).any(), rec
    assert sorted(rec[:, 0].astype(int).tolist()) == list(range(n))
    for i, sm in zip(ids, sms):
        assert rec[i, 4] == sm["final_cost"] and int(rec[i, 1]) == sm["iterations"]
    dist.destroy_process_group()
    print("RCCL_GATHER_OK", rank, world)


if __name__ == "__main__":
    main()
