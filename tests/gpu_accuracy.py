"""Manual GPU study (not a test): how far apart are fp32-HIP, fp64-HIP and the oracle at Ceres' tolerances,
and how much of that is the optimiser's own stopping slop (same solver, tight tolerances)?"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
cv = importlib.import_module("ctrl-vio_amd")
import pyctvo

def run_gpu(w0, prec, iters, **tol):
    with cv.Solver(precision=prec, **tol) as s:
        w = w0.copy(); s.set_windows([w]); sm = s.solve(iters)[0]
    return w, sm

for cfg in sys.argv[1:] or ["config1", "config2"]:
    for seed in (1000, 1001, 1002):
        w0 = cv.synth.make_window(cfg, seed=seed)
        wo = w0.copy(); so = pyctvo.OracleWindow(wo).solve(15)
        pyctvo.set_tolerances(1e-13, 1e-14, 1e-13)
        wt = w0.copy(); st = pyctvo.OracleWindow(wt).solve(200)
        pyctvo.set_tolerances()
        w32, s32 = run_gpu(w0, "fp32", 15)
        w64, s64 = run_gpu(w0, "fp64", 15)
        w32t, s32t = run_gpu(w0, "fp32", 200, function_tolerance=1e-13, parameter_tolerance=1e-13)
        e = lambda a, b: cv.rel_state_error(a, b)
        fmt = lambda d: " ".join(f"{k}={v:.1e}" for k, v in d.items())
        print(f"[{cfg} seed {seed}] oracle15: it={so.iterations} cost={so.final_cost:.6f} | tight: it={st.iterations} cost={st.final_cost:.6f}")
        print(f"   oracle15 vs oracle-tight (stopping slop): {fmt(e(wo, wt))}")
        print(f"   fp32(15) vs oracle15 : it={s32['iterations']} {s32['termination']} cost={s32['final_cost']:.6f} {fmt(e(w32, wo))}")
        print(f"   fp32(15) vs oracle-tight               : {fmt(e(w32, wt))}")
        print(f"   fp64(15) vs oracle15 : it={s64['iterations']} {fmt(e(w64, wo))}")
        print(f"   fp32-tight vs oracle-tight: it={s32t['iterations']} {s32t['termination']} cost={s32t['final_cost']:.6f} {fmt(e(w32t, wt))}")
