#!/usr/bin/env python
"""bench.py -- sliding-window solves/sec on MI355X (BASELINE.json metric).

A "step" = one batched solve of `--windows` independent config-2 windows (10 KF / 200 landmarks / 2000 IMU,
15 LM iterations max, Ceres tolerances) per GPU, inputs and initial state already resident in HBM
(the state is reset on the device between steps; pack + H2D are outside the timed region).
N > 1: one process per GPU (torch.distributed over RCCL), windows sharded by seed, no data-path collective;
the barrier + max-over-ranks timing uses torch.distributed.  value = windows solved by all ranks / time.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=64, help="independent windows per GPU per step")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic windows generated per GPU (replicated to --windows)")
    ap.add_argument("--config", default="config2")
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    cv = importlib.import_module("ctrl-vio_amd")

    # synthetic windows: seeds 1000 + rank*windows + i, a few unique ones replicated to fill the batch
    uniq = [cv.synth.make_window(args.config, seed=1000 + rank * args.windows + i) for i in range(min(args.unique, args.windows))]
    wins = [uniq[i % len(uniq)] for i in range(args.windows)]
    init = [w.copy() for w in uniq]
    solver = cv.Solver(device=local, precision=args.precision)
    solver.set_windows([w.copy() for w in wins])

    def reset():
        for i in range(args.windows):
            solver.set_state(i, init[i % len(init)])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        reset(); solver.solve_raw(args.iters)
    times = []
    kernel_ms = []
    barrier()
    t_total = 0.0
    for _ in range(args.steps):
        reset()
        barrier()
        t0 = time.perf_counter()
        solver.solve_raw(args.iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        times.append(dt)
        kernel_ms.append(solver.last_timing()[6])
    t_total = sum(times)
    if dist is not None:
        tt = torch.tensor([t_total], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_total = float(tt.item())
    n_solved = args.windows * world * args.steps
    value = n_solved / t_total

    out = {
        "metric": "sliding-window solves/sec (10 KF, 200 lm, 2000 IMU)", "value": value, "unit": "solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_total / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: 10 KF / 200 landmarks / 2000 IMU window, <=%d LM iterations" % args.iters,
                   "windows_per_gpu": args.windows, "sharding": f"independent windows x{world}"},
        "device_ms_per_step": float(np.mean(kernel_ms)),
    }
    if rank == 0:
        # quality of the timed solves
        sm = solver.solve(args.iters, writeback=False) if False else None
        out["roofline"] = None
        out["cpu_baseline"] = None
        if not args.no_cpu_baseline:
            import pyctvo
            w = init[0].copy()
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 10.0:
                ww = w.copy(); pyctvo.OracleWindow(ww).solve(args.iters); n += 1
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n / dt, "unit": "solves/s", "cores": 1, "kind": "port",
                                   "sample": f"{n} solves of one {args.config} window (seed 1000), fp64 C oracle, 1 thread"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
