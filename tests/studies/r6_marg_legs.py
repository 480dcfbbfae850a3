import importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
cv = importlib.import_module("ctrl-vio_amd")
import pyctvo, slide_helpers as sh
world = sh.make_world()
st = sh.State(world)
w, info = sh.window_of(world, st, 0, sh.initial_prior(world))
m, role = sh.marg_window_of(world, st, 0, w, info)
ko, Jo, ro = pyctvo.OracleWindow(m.copy()).marginalize(role, 1e-8)
with cv.Solver() as s:
    s.set_windows([m.copy()]); kd, Jd, rd = s.marginalize(0, role); print("device leg host?", s.marginalize_ran_on_host())
os.environ["CTVIO_MARG_HOST"] = "1"
with cv.Solver() as s:
    s.set_windows([m.copy()]); kh, Jh, rh = s.marginalize(0, role); print("host leg host?", s.marginalize_ran_on_host())
def cmp(a, b, name):
    Ha, Hb = a[0].T @ a[0], b[0].T @ b[0]; ga, gb = a[0].T @ a[1], b[0].T @ b[1]
    print(name, "H rel", np.abs(Ha - Hb).max() / np.abs(Hb).max(), "g rel", np.abs(ga - gb).max() / np.abs(gb).max(), "ranks", np.linalg.matrix_rank(a[0]), np.linalg.matrix_rank(b[0]),
          "shapes", a[0].shape, b[0].shape)
cmp((Jd, rd), (Jo, ro), "device vs oracle"); cmp((Jh, rh), (Jo, ro), "host vs oracle"); cmp((Jd, rd), (Jh, rh), "device vs host")
print("kept equal", np.array_equal(kd, kh), np.array_equal(kd, ko))
