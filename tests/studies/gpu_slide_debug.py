import sys, os
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import numpy as np
import slide_helpers as sh
cv = sh.cv
world = sh.make_world()
ob, db = sh.OracleBackend(), sh.DeviceBackend("fp64")
st = sh.State(world); prior = sh.initial_prior(world)
w, info = sh.window_of(world, st, 0, prior)
wo = w.copy(); wd = w.copy()
print(ob.solve_and_restore(wo, 0)); print(db.solve_and_restore(wd, 0))
print("state err win0", cv.rel_state_error(wd, wo))
st.quat[info["kmin"]:info["kmin"] + w.K], st.pos[info["kmin"]:info["kmin"] + w.K] = wo.quat, wo.pos
st.bias[0:sh.WIN], st.rho[info["lms"]], st.ld = wo.bias, wo.rho, float(wo.ld)
m, role = sh.marg_window_of(world, st, 0, wo, info)
ko, Jo, ro = ob.marginalize(m.copy(), role)
kd, Jd, rd = db.marginalize(m.copy(), role)
print("kept equal", np.array_equal(ko, kd), len(ko))
Ho, Hd = Jo.T @ Jo, Jd.T @ Jd
print("H diff", np.abs(Ho - Hd).max() / np.abs(Ho).max(), "g diff", np.abs(Jo.T @ ro - Jd.T @ rd).max() / np.abs(Jo.T @ ro).max(), "c", ro @ ro, rd @ rd)
print("eig J0^T J0 oracle smallest", np.sort(np.linalg.eigvalsh(Ho))[:8])
print("nonzero rows", (np.abs(Jo).sum(1) > 0).sum(), (np.abs(Jd).sum(1) > 0).sum())
# same prior into both solvers for window 1
for name, (kk, JJ, rr) in (("oracle prior", (ko, Jo, ro)), ("device prior", (kd, Jd, rd))):
    pr = sh.prior_from(m, info, 0, kk, JJ, rr)
    w1, info1 = sh.window_of(world, st, 1, pr)
    co = ob.o.OracleWindow(w1.copy()).cost()
    with cv.Solver() as s:
        s.set_windows([w1.copy()]); cd = s.cost(0)
    a = w1.copy(); b = w1.copy()
    print(name, "cost oracle", co, "device", cd, ob.solve_and_restore(a, 0), db.solve_and_restore(b, 0))
