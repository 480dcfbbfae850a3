"""Repeat the parity-critical solves many times in one process, interleaved with the other API paths (different kernels,
different allocations), to expose races and reads of stale memory: every solve must reproduce the oracle.
Run on the GPU box: python tests/gpu_stress_repeat.py [repeats]"""
import importlib, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
cv = importlib.import_module('ctrl-vio_amd')
import pyctvo

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
cases = []
for cfg, seed in (("tiny", 7), ("config1", 1001), ("config2", 1000)):
    w0 = cv.synth.make_window(cfg, seed=seed)
    wo = w0.copy()
    cases.append((cfg, w0, wo, pyctvo.OracleWindow(wo).solve(15)))
for rep in range(reps):
    for cfg, w0, wo, smo in cases:
        # other API paths first: they leave different contents in freed device memory
        for mf in (True, False):
            with cv.Solver(precision="fp32", use_mfma=mf) as s:
                s.set_windows([w0.copy()])
                s.lm_step(0, 1e4)
                s.linearize(0)
        for prec in ("fp64", "fp32"):
            with cv.Solver(precision=prec) as s:
                wg = w0.copy()
                s.set_windows([wg, w0.copy(), w0.copy()][: 1 + rep % 3])
                sm = s.solve(15)[0]
            err = cv.rel_state_error(wg, wo)["state"]
            if prec == "fp64":
                ok = sm["iterations"] == smo.iterations and sm["num_successful"] == smo.num_successful and err < 1e-6 \
                    and abs(sm["final_cost"] - smo.final_cost) <= 1e-9 * smo.final_cost
            else:
                ok = abs(sm["iterations"] - smo.iterations) <= 1 and err < 2e-4
            if not ok:
                bad += 1
                print("BAD", cfg, prec, rep, sm["iterations"], smo.iterations, sm["num_successful"], smo.num_successful, err,
                      sm["final_cost"], smo.final_cost, sm["termination"])
print("bad", bad, "of", reps * len(cases) * 2)
