"""Manual GPU diagnostic (not a test): prints per-stage errors of the HIP path vs the oracle."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
cv = importlib.import_module("ctrl-vio_amd")
import pyctvo

def nrm(A, B, sc=None):
    d = np.abs(A - B)
    return float(d.max() / max(np.abs(B).max(), 1e-300))

def stage(cfg, seed, prec, **kw):
    w0 = cv.synth.make_window(cfg, seed=seed)
    w0.ld = 1.1e-5
    o = pyctvo.OracleWindow(w0.copy())
    H, g, cost = o.build_normal()
    P = w0.P
    with cv.Solver(precision=prec, **kw) as s:
        s.set_windows([w0.copy()])
        t = time.time(); Hg, Wg, Hllg, gg, costg = s.linearize(0); dt = time.time() - t
        sc = np.sqrt(np.maximum(np.diag(H), 1e-30))
        print(f"[{cfg} {prec}] linearize {dt*1e3:.1f} ms  cost gpu {costg:.6f} oracle {cost:.6f} rel {abs(costg-cost)/cost:.2e}")
        print("   Hpp  scaled max err", np.abs(Hg / np.outer(sc[:P], sc[:P]) - H[:P, :P] / np.outer(sc[:P], sc[:P])).max())
        print("   W    scaled max err", np.abs(Wg / np.outer(sc[:P], sc[P:]) - H[:P, P:] / np.outer(sc[:P], sc[P:])).max())
        print("   Hll  rel err", nrm(Hllg, np.diag(H)[P:]), " g scaled err", np.abs((gg - g) / sc).max(), "of", np.abs(g / sc).max())
        d_o, mc_o = o.lm_step(1e4)
        d_g, mc_g = s.lm_step(0, 1e4)
        print("   lm_step: delta rel err", nrm(d_g, d_o), " model change", mc_g, mc_o)
        t_q = np.linspace(w0.t0_ns, w0.max_time_ns() - 1, 50).astype(np.int64)
        pg = s.spline_eval(0, t_q); po = o.spline_eval(t_q)
        print("   spline_eval err", [float(np.abs(a - b).max()) for a, b in zip(pg, po)])
        for iters in (15, 50):
            wo = w0.copy(); sm_o = pyctvo.OracleWindow(wo).solve(iters)
            wg = w0.copy(); s.set_state(0, wg); s.windows[0] = wg
            t = time.time(); sm = s.solve(iters)[0]; dt = time.time() - t
            print(f"   solve({iters}) {dt*1e3:.1f} ms gpu {sm}")
            print(f"      oracle iters {sm_o.iterations} succ {sm_o.num_successful} term {sm_o.termination} cost {sm_o.final_cost:.6f}")
            print("      rel err vs oracle", {k: f"{v:.2e}" for k, v in cv.rel_state_error(wg, wo).items()})

if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "config1", "config2"]
    for cfg in which:
        for prec in ("fp64", "fp32"):
            try:
                stage(cfg, 1000, prec)
            except Exception as e:
                import traceback; traceback.print_exc()
    # batch check
    ws = [cv.synth.make_window("config1", seed=1000 + i) for i in range(4)] + [cv.synth.make_window("tiny", seed=5)]
    with cv.Solver(precision="fp32") as s:
        batch = [w.copy() for w in ws]
        s.set_windows(batch)
        t = time.time(); sms = s.solve(15); dt = time.time() - t
        print("batch of", len(ws), f"{dt*1e3:.1f} ms", [m["final_cost"] for m in sms])
        for i, w in enumerate(ws):
            with cv.Solver(precision="fp32") as s1:
                w1 = w.copy(); s1.set_windows([w1]); s1.solve(15)
            print("   window", i, "batch vs single", f"{cv.rel_state_error(batch[i], w1)['state']:.2e}")
