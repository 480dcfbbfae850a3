"""Diagnostics (GPU box): big-batch vs small-batch vs oracle for K = 26 / 27 windows; CTVIO_SCHUR_TILES=1 forces the tile Schur."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
cv = importlib.import_module("ctrl-vio_amd")
import pyctvo
for dt_ms in (42, 40):
    base = [cv.synth.make_window("config1", seed=1200 + i, F=10, dt_ns=dt_ms * 1_000_000) for i in range(4)]
    with cv.Solver() as s:
        small = [w.copy() for w in base]; s.set_windows(small); sm_small = s.solve(15)
        big = [base[i % 4].copy() for i in range(208)]; s.set_windows(big); sm_big = s.solve(15)
    for i in range(4):
        wo = base[i].copy(); so = pyctvo.OracleWindow(wo).solve(15)
        print(dt_ms, i, "cost small/big/oracle", sm_small[i]["final_cost"], sm_big[i]["final_cost"], so.final_cost,
              "iters", sm_small[i]["iterations"], sm_big[i]["iterations"], so.iterations,
              "err small-oracle %.2e big-oracle %.2e big-small %.2e" % (cv.rel_state_error(small[i], wo)["state"], cv.rel_state_error(big[i], wo)["state"],
                                                                    cv.rel_state_error(big[i], small[i])["state"]))
