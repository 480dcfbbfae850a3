"""CPU-only sensitivity study behind DESIGN.md section 3: how far does the 15-iteration Ceres solve move when the Jacobians
of one factor family carry a relative error (as fp32 evaluation does), everything else exact fp64?
python tests/jacobian_noise_study.py [n_seeds]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
cv = importlib.import_module("ctrl-vio_amd")
import pyctvo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ws = [cv.synth.make_window("config2", seed=2000 + i) for i in range(n)]
ref = []
for w in ws:
    wo = w.copy(); pyctvo.OracleWindow(wo).solve(15); ref.append(wo)
for tag, a, b in (("imu 1e-7", 1e-7, 0), ("vis 1e-7", 0, 1e-7), ("imu 1e-6", 1e-6, 0), ("vis 1e-6", 0, 1e-6), ("both 1e-6", 1e-6, 1e-6), ("both 1e-8", 1e-8, 1e-8)):
    pyctvo.set_jacobian_noise(a, b)
    st = []
    for w, wo in zip(ws, ref):
        wn = w.copy(); pyctvo.OracleWindow(wn).solve(15); st.append(cv.rel_state_error(wn, wo)["state"])
    st = np.array(st)
    print(f"{tag:10s} state err vs clean oracle: median {np.median(st):.1e}  p90 {np.quantile(st, .9):.1e}  max {st.max():.1e}  > 1e-4: {(st > 1e-4).sum()}/{n}")
pyctvo.set_jacobian_noise(0, 0)
pyctvo.set_product_rounding(True)
st = []
for w, wo in zip(ws, ref):
    wn = w.copy(); pyctvo.OracleWindow(wn).solve(15); st.append(cv.rel_state_error(wn, wo)["state"])
st = np.array(st)
print(f"per-block J^T J, J^T r rounded to fp32 before accumulation: median {np.median(st):.1e}  p90 {np.quantile(st, .9):.1e}  max {st.max():.1e}  > 1e-4: {(st > 1e-4).sum()}/{n}")
pyctvo.set_product_rounding(False)
