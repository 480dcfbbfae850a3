"""Parity evidence for profiles/: the product path (mixed fp32/fp64) and the all-fp64 path against the fp64 oracle at Ceres
settings (15 iterations, function tolerance 1e-6), over many synthetic windows and repeated runs.  Windows are split by
how the ORACLE stopped: by a tolerance (converged) or by the iteration cap (not converged: the state is then only
determined up to the solver's own stopping slop, printed as the distance between the oracle at 15 iterations and the
oracle run to 1e-13), and by whether the device LM took the same accept / reject decisions as the oracle (same numbers of
successful and unsuccessful steps): a borderline step accepted by one and rejected by the other changes the trust-region
sequence, after which the two solvers follow different -- equally valid -- paths.  Run on the GPU box: python tests/gpu_parity_study.py [n_seeds] [repeats]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
cv = importlib.import_module("ctrl-vio_amd")
import pyctvo

nseed = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = []
for cfg in ("config1", "config2", "config3"):
    ws = [cv.synth.make_window(cfg, seed=2000 + i) for i in range(nseed)]
    ref = []
    for w in ws:
        wo = w.copy()
        smo = pyctvo.OracleWindow(wo).solve(15)
        pyctvo.set_tolerances(1e-13, 1e-14, 1e-13)
        wt = w.copy(); pyctvo.OracleWindow(wt).solve(300)
        pyctvo.set_tolerances()
        ref.append((wo, smo, cv.rel_state_error(wo, wt)["state"]))
    for prec in ("fp32", "fp64"):
        for rep in range(reps if prec == "fp32" else 1):
            with cv.Solver(precision=prec) as s:
                wg = [w.copy() for w in ws]
                s.set_windows(wg)
                sms = s.solve(15)
            for i, (w, sm) in enumerate(zip(wg, sms)):
                wo, smo, slop = ref[i]
                rows.append(dict(cfg=cfg, seed=2000 + i, prec=prec, rep=rep, capped=smo.termination == 0, slop=slop,
                                 same=(sm["num_successful"] == smo.num_successful and sm["num_unsuccessful"] == smo.num_unsuccessful),
                                 dit=sm["iterations"] - smo.iterations, cost=abs(sm["final_cost"] - smo.final_cost) / smo.final_cost,
                                 state=cv.rel_state_error(w, wo)["state"]))


def line(tag, r):
    if not r:
        return
    st = np.array([x["state"] for x in r]); co = np.array([x["cost"] for x in r]); di = np.array([x["dit"] for x in r])
    sl = np.array([x["slop"] for x in r])
    print(f"  {tag:34s} {len(r):3d} solves  state err: median {np.median(st):.1e} p90 {np.quantile(st, 0.9):.1e} max {st.max():.1e}"
          f" | > 1e-4: {(st > 1e-4).sum():2d} | cost rel max {co.max():.1e} | iter diff [{di.min()}, {di.max()}] | oracle slop median {np.median(sl):.1e} max {sl.max():.1e}")


for cfg in ("config1", "config2", "config3"):
    print(cfg)
    for prec in ("fp32", "fp64"):
        r = [x for x in rows if x["cfg"] == cfg and x["prec"] == prec]
        line(f"{prec} oracle converged (tolerance)", [x for x in r if not x["capped"]])
        line(f"{prec} oracle hit the iteration cap", [x for x in r if x["capped"]])
        if prec == "fp32":
            line("fp32 same accept/reject sequence", [x for x in r if x["same"]])
            line("fp32 one or more decisions differ", [x for x in r if not x["same"]])
for x in sorted([x for x in rows if x["prec"] == "fp32"], key=lambda x: -x["state"])[:6]:
    print("worst fp32:", x["cfg"], "seed", x["seed"], "run", x["rep"], "capped" if x["capped"] else "converged", "iter diff", x["dit"],
          "cost rel %.1e" % x["cost"], "state %.2e" % x["state"], "oracle slop %.1e" % x["slop"])
