#!/usr/bin/env python
"""bench.py -- sliding-window solves/sec on MI355X (BASELINE.json metric, configs[1] workload).

A "step" = one batched solve of `--windows` independent config-2 windows (10 KF / 200 landmarks / 2000 IMU,
<= 15 LM iterations, Ceres tolerances) per GPU.  Factors and the initial state are resident in HBM before the
timed region (ctvio_restore_state resets the state on the device between steps; packing + H2D are outside).
value = windows solved by all ranks / wall-clock of the K timed steps (max over ranks).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), windows sharded by seed, no data-path
collective (independent windows); RCCL only carries the barrier and the max-over-ranks time.

Extra objects on the JSON line:
  roofline      dominant kernel (largest share of a profiled solve, HIP events on the solver's stream):
                achieved = algorithmic bytes / average launch duration, against 8 TB/s HBM;
                roofline_mfma: the Schur SYRK against the 157.3 TFLOP/s fp32 MFMA peak.
  cpu_baseline  the fp64 C oracle (a port of the reference's Ceres path: oracle/ctvo.c) on 1 host core, ~10 s sample.
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

# phase of the solver's event profile -> kernel name in the rocprofv3 tables (profiles/)
KERNEL_OF_PHASE = {"k_imu_linearize": "k_imu_linearize", "k_vis_eval": "k_vis_eval<float, true, double>",
                   "k_assemble_vis": "k_assemble_vis_mfma", "k_schur_mfma": "k_schur_window", "k_cholesky_solve": "k_cholesky_solve"}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32


def algorithmic_bytes(w, phase, fp_bytes=4):
    """Algorithmic HBM bytes of ONE window for one launch of a kernel group."""
    K, F, L, M, V, P = w.K, w.F, w.L, w.M, w.V, w.P
    G = len({(int((t - w.t0_ns) // w.dt_ns), int(b)) for t, b in zip(w.imu_t, w.imu_bias)})
    if phase == "k_imu_linearize":   # per sample u + 6 measurements; per group 4 knots (fp64 state) + bias + 32x32 tile out
        return M * 7 * fp_bytes + G * (4 * 7 * 8 + 6 * 8 + 1024 * fp_bytes)
    if phase == "k_vis_eval":        # 284 B in, 408 B out per block (J materialised once)
        return V * (284 + 408 + 8)
    if phase == "k_assemble_vis":    # J read once + W/Hll/g rows + packed visual Hessian flushed once (fp64)
        K6 = 6 * K
        return V * (408 + 8 + 51 * fp_bytes) + (K6 * (K6 + 1) // 2 + K6 + 1) * 8
    if phase == "k_cholesky_solve":  # lower triangle read + written once, rhs in, solution out
        return (P * (P + 1) // 2) * 8 * 2 + 2 * P * 8
    if phase == "k_schur_mfma":      # W read, Hpp lower read, S lower written
        return L * P * fp_bytes + (P * (P + 1) // 2) * 16
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=1024, help="independent windows per GPU per step")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic windows generated per GPU (replicated to --windows)")
    ap.add_argument("--config", default="config2")
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-inclusive", action="store_true", help="also time pack + H2D + solve + D2H of one batch (DESIGN.md section 6; never the headline value)")
    ap.add_argument("--streams", type=int, default=4, help="solver handles (HIP streams) per GPU; the windows are split evenly among them")
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

    # synthetic windows, seeds 1000 + rank*unique + i; a few distinct ones replicated to fill the batch
    uniq = [cv.synth.make_window(args.config, seed=1000 + rank * args.unique + i) for i in range(min(args.unique, args.windows))]
    import threading
    nstream = max(1, args.streams)
    per = [args.windows // nstream + (1 if i < args.windows % nstream else 0) for i in range(nstream)]
    solvers = []
    for si_, cnt in enumerate(per):
        sv = cv.Solver(device=local, precision=args.precision)
        sv.set_windows([uniq[(i + si_) % len(uniq)].copy() for i in range(cnt)])
        sv.snapshot_state()
        solvers.append(sv)
    solver = solvers[0]

    def solve_all():
        if nstream == 1:
            solver.solve_raw(args.iters)
            return
        th = [threading.Thread(target=sv.solve_raw, args=(args.iters,)) for sv in solvers]
        for t in th: t.start()
        for t in th: t.join()

    def restore_all():
        for sv in solvers: sv.restore_state()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        restore_all()
        solve_all()
    t_total = 0.0
    dev_ms = []
    for _ in range(args.steps):
        restore_all()
        barrier()
        t0 = time.perf_counter()
        solve_all()                           # returns after every stream is drained (summaries copied back)
        torch.cuda.synchronize()
        t_total += time.perf_counter() - t0
        dev_ms.append(solver.last_timing()[0][7])
    barrier()
    if dist is not None:
        tt = torch.tensor([t_total], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_total = float(tt.item())
    n_solved = args.windows * world * args.steps
    out = {
        "metric": "sliding-window solves/sec (10 KF, 200 lm, 2000 IMU)", "value": n_solved / t_total, "unit": "solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_total / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "f64", "data": "synthetic",
        "config": {"workload": f"{args.config}: 10 KF / 200 landmarks / 2000 IMU sliding window, <= {args.iters} LM iterations",
                   "windows_per_gpu": args.windows, "streams_per_gpu": nstream, "sharding": f"independent windows, {world} rank(s), no data-path collective"},
        "device_ms_per_step": float(np.mean(dev_ms)),
    }
    if rank == 0:
        # ---- quality of what was timed: the solved batch against the fp64 oracle on the distinct windows
        solver.restore_state()
        sms = solver.solve(args.iters, writeback=False)
        out["solve_summary"] = {"iterations_mean": float(np.mean([m["iterations"] for m in sms])),
                                "terminations": sorted({m["termination"] for m in sms})}
        # ---- roofline: one more step of the same workload (all streams running, as in the timed region) with HIP events
        #      around every launch group of stream 0 -- the launches of that stream carry per[0] windows each
        restore_all()
        solver.set_profiling(True)
        solve_all()
        solver.set_profiling(False)
        torch.cuda.synchronize()
        ms, n = solver.last_timing()
        names = cv.Solver.PHASES
        shares = {names[i]: float(ms[i]) for i in range(7)}
        dom = max(range(6), key=lambda i: ms[i])            # named kernels only (0..5)
        w_ref = uniq[0]
        nbytes = sum(algorithmic_bytes(uniq[i % len(uniq)], names[dom]) for i in range(per[0]))
        avg_s = 1e-3 * ms[dom] / max(int(n[dom]), 1)
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):
            try:
                key = KERNEL_OF_PHASE.get(names[dom], names[dom])
                traffic = json.load(open(tfile)).get(key, {}).get(str(per[0]), {}).get("traffic_bytes")
            except Exception:
                traffic = None
        ach = nbytes / avg_s / 1e9
        out["roofline"] = {"kernel": KERNEL_OF_PHASE.get(names[dom], names[dom]).split("<")[0], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                           "avg_launch_us": 1e6 * avg_s, "launches": int(n[dom]), "windows_per_launch": per[0],
                           "algorithmic_bytes_per_launch": nbytes,
                           "share_of_profiled_solve": float(ms[dom] / max(sum(ms[:7]), 1e-12))}
        P, L = w_ref.P, w_ref.L
        fl = P * (P + 1) * L * per[0]                        # SYRK count
        avg_schur = 1e-3 * ms[4] / max(int(n[4]), 1)
        out["roofline_mfma"] = {"kernel": "k_schur_window", "bound": "mfma", "achieved": fl / avg_schur / 1e12,
                                "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / avg_schur / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                "avg_launch_us": 1e6 * avg_schur, "flops_per_launch": fl}
        out["phase_ms_profiled_solve"] = shares               # stream 0 only
        if args.host_inclusive:
            # the C ABI takes host buffers: pack (host, 1 thread) + H2D + solve + D2H of the states, one solver, one batch
            wl = [uniq[i % len(uniq)].copy() for i in range(args.windows)]
            with cv.Solver(device=local, precision=args.precision) as hs:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                hs.set_windows(wl)                          # ctvio_add_window x N + ctvio_upload
                t1 = time.perf_counter()
                hs.solve(args.iters, writeback=True)        # ctvio_solve + ctvio_get_state x N
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            out["host_inclusive"] = {"windows": args.windows, "pack_upload_s": t1 - t0, "solve_readback_s": t2 - t1,
                                     "solves_per_s": args.windows / (t2 - t0)}
        # ---- CPU baseline: the oracle (a port, not the reference binary: Ceres/Eigen are not installable here)
        out["cpu_baseline"] = None
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only: the scaling runs must not wait on a CPU loop
            import pyctvo
            t0 = time.perf_counter(); k = 0
            while time.perf_counter() - t0 < args.cpu_seconds:
                ww = uniq[k % len(uniq)].copy()
                pyctvo.OracleWindow(ww).solve(args.iters)
                k += 1
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": k / dt, "unit": "solves/s", "cores": 1, "kind": "port",
                                   "sample": f"{k} solves of {args.config} windows (seeds 1000..), fp64 C oracle (oracle/ctvo.c, gcc -O2), "
                                             f"1 thread, {dt:.1f} s; host has {os.cpu_count()} cores"}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
