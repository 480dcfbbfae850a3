import importlib, sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
cv=importlib.import_module('ctrl-vio_amd'); import pyctvo
bad=0
for cfg,seed in (("tiny",7),("config1",1001),("config2",1000)):
    w0=cv.synth.make_window(cfg,seed=seed); wo=w0.copy(); smo=pyctvo.OracleWindow(wo).solve(15)
    for rep in range(40):
        for prec in ("fp64","fp32"):
            with cv.Solver(precision=prec) as s:
                wg=w0.copy(); s.set_windows([wg, w0.copy(), w0.copy()]); sm=s.solve(15)[0]
            err=cv.rel_state_error(wg,wo)["state"]
            ok = (sm["iterations"]==smo.iterations and err<1e-6) if prec=="fp64" else (abs(sm["iterations"]-smo.iterations)<=1 and err<2e-4)
            if not ok: bad+=1; print("BAD",cfg,prec,rep,sm["iterations"],smo.iterations,err,sm["final_cost"],smo.final_cost, sm["termination"])
print("bad",bad)
