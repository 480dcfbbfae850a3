"""Quick parity distribution on the GPU box: default path vs the fp64 oracle, 16 fresh windows per config at Ceres settings."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
cv = importlib.import_module("ctrl-vio_amd"); import pyctvo
for cfg in sys.argv[1:] or ["config2"]:
    ws = [cv.synth.make_window(cfg, seed=2000 + i) for i in range(16)]
    ref = []
    for w in ws:
        wo = w.copy(); pyctvo.OracleWindow(wo).solve(15); ref.append(wo)
    with cv.Solver(precision="fp32") as s:
        wg = [w.copy() for w in ws]; s.set_windows(wg); s.solve(15)
    st = np.array([cv.rel_state_error(a, b)["state"] for a, b in zip(wg, ref)])
    print(cfg, "median %.1e p90 %.1e max %.1e  >1e-4: %d/16" % (np.median(st), np.quantile(st, .9), st.max(), (st > 1e-4).sum()), np.array2string(st, precision=1))
