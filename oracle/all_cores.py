"""TEST / BENCH INFRASTRUCTURE (not product code): the fp64 C oracle on every host core, one window per process -- SURVEY.md 8d's second
CPU baseline (configs[3]: independent windows, one per thread).  Run by bench.py's cpu_baseline leg in a FRESH interpreter (no HIP
runtime in the forked workers):   python oracle/all_cores.py <config> <max_iterations> <first_seed> <n_windows> [processes]
Prints one JSON line {"solves", "seconds", "processes", "host_cores", "usable_cpus"}; the windows are generated before the clock starts.
The default process count is the number of CPUs the container is ENTITLED to (affinity mask capped by the cgroup quota), not the
number it can see."""
import importlib
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

def effective_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container that sees 256 logical CPUs
    may be entitled to 16 of them: /sys/fs/cgroup/cpu.max = "1600000 100000")."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def main():
    config, iters, seed0, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    ncore = os.cpu_count() or 1
    ncpu = effective_cpus()
    nproc = int(sys.argv[5]) if len(sys.argv) > 5 else min(ncpu, n)
    import pyctvo
    pyctvo.build()
    seeds = list(range(seed0, seed0 + n))
    # every worker generates and keeps its own windows (static assignment: window i -> process i mod nproc), then all solve at once
    with mp.get_context("fork").Pool(nproc) as pool:
        chunks = [[s for s in seeds[i::nproc]] for i in range(nproc)]
        ws = {}
        for part in pool.map(_prepare, [(config, c) for c in chunks], chunksize=1):   # (windows come back to the parent: a chunk may be
            ws.update(part)                                                             # solved by another worker than the one that made it)
        t0 = time.perf_counter()
        its = pool.map(_run, [([ws[s] for s in c], iters) for c in chunks], chunksize=1)
        dt = time.perf_counter() - t0
    print(json.dumps({"solves": n, "seconds": dt, "processes": nproc, "host_cores": ncore, "usable_cpus": ncpu, "iterations_mean": sum(sum(i) for i in its) / n}))


def _prepare(args):
    config, seeds = args
    cv = importlib.import_module("ctrl-vio_amd")
    return {s: cv.synth.make_window(config, seed=s) for s in seeds}


def _run(args):
    wins, iters = args
    import pyctvo
    return [pyctvo.OracleWindow(w).solve(iters).iterations for w in wins]


if __name__ == "__main__":
    main()
