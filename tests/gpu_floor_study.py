import importlib, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
cv = importlib.import_module("ctrl-vio_amd"); import pyctvo
for cfg, seed in (("config1", 1000), ("config2", 1000), ("config2", 1002)):
    w0 = cv.synth.make_window(cfg, seed=seed)
    pyctvo.set_tolerances(1e-13, 1e-14, 1e-13); wt = w0.copy(); so = pyctvo.OracleWindow(wt).solve(200); pyctvo.set_tolerances()
    wo = w0.copy(); so15 = pyctvo.OracleWindow(wo).solve(15)
    for rep in range(4):
        with cv.Solver(precision="fp32", function_tolerance=1e-13, parameter_tolerance=1e-13) as s:
            wg = w0.copy(); s.set_windows([wg]); sm = s.solve(200)[0]
        with cv.Solver(precision="fp32") as s:
            w15 = w0.copy(); s.set_windows([w15]); sm15 = s.solve(15)[0]
        print(cfg, seed, rep, "tight: it", sm["iterations"], sm["termination"], "cost rel %.1e" % (abs(sm["final_cost"]-so.final_cost)/so.final_cost),
              "state %.1e" % cv.rel_state_error(wg, wt)["state"], "| ceres15: it", sm15["iterations"], "cost rel %.1e" % (abs(sm15["final_cost"]-so15.final_cost)/so15.final_cost), "state %.1e" % cv.rel_state_error(w15, wo)["state"])
