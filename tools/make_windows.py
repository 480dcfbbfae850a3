"""Synthetic windows of one config, seeds seed0 .. seed0 + n - 1, generated on a process pool in a FRESH interpreter (no HIP runtime in the
forked workers) and written as one .npz (Window.to_dict with a per-window prefix):  python tools/make_windows.py <config> <seed0> <n> <out.npz>
bench.py's `mixed_batch` leg uses it: 256 distinct windows take 2 s on 16 cores instead of 34 s in the bench process."""
import importlib
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(args):
    cv = importlib.import_module("ctrl-vio_amd")
    cfg, seed = args
    return cv.synth.make_window(cfg, seed=seed).to_dict()


def main():
    cfg, seed0, n, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    nproc = max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    with mp.get_context("fork").Pool(nproc) as pool:
        ds = pool.map(one, [(cfg, seed0 + i) for i in range(n)], chunksize=1)
    flat = {}
    for i, d in enumerate(ds):
        for k, v in d.items():
            flat[f"w{i}_{k}"] = v
    np.savez(out, **flat)


if __name__ == "__main__":
    main()
