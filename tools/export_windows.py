#!/usr/bin/env python
"""Write synthetic windows (ctrl-vio_amd/synth.py) to a flat binary file for C / C++ callers (include/ctvio_window_io.hpp).
usage: python tools/export_windows.py <config> <first seed> <count> <out.ctvw>      e.g.  config2 1000 64 bench_windows.ctvw"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cv = importlib.import_module("ctrl-vio_amd")
cfg, seed0, n, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cv.window_io.save_windows(out, [cv.synth.make_window(cfg, seed=seed0 + i) for i in range(n)])
print(f"{n} {cfg} windows (seeds {seed0}..{seed0 + n - 1}) -> {out}")
