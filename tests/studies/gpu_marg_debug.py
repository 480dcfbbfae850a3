import sys, os
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import numpy as np
import slide_helpers as sh, pyctvo
cv = sh.cv
world = sh.make_world(); st = sh.State(world)
w, info = sh.window_of(world, st, 0, sh.initial_prior(world))
m, role = sh.marg_window_of(world, st, 0, w, info)
Ho, go, _ = pyctvo.OracleWindow(m.copy()).build_normal()
im = np.where(role == 1)[0]; ik = np.where(role == 0)[0]
def schur(H):
    Amm = H[np.ix_(im, im)]; em, Vm = np.linalg.eigh(Amm)
    X = Vm @ ((Vm.T @ H[np.ix_(im, ik)]) / em[:, None])
    Ap = H[np.ix_(ik, ik)] - H[np.ix_(ik, im)] @ X
    return 0.5 * (Ap + Ap.T)
with cv.Solver() as s:
    s.set_windows([m.copy()])
    Hpp, W, Hll, g, c = s.linearize(0)
    P = m.P
    Hd = np.zeros((m.N, m.N)); Hd[:P, :P] = Hpp; Hd[:P, P:] = W; Hd[P:, :P] = W.T; Hd[P:, P:] = np.diag(Hll)
    print("H dev vs oracle rel", np.abs(Hd - Ho).max() / np.abs(Ho).max())
    for name, H in (("oracle H", Ho), ("device H", Hd)):
        ev = np.sort(np.abs(np.linalg.eigvalsh(schur(H))))
        print(name, "A' |eig| smallest 24:", np.array2string(ev[:24], precision=1))
    kd, Jd, rd = s.marginalize(0, role)
    print("device marg nonzero rows", (np.abs(Jd).sum(1) > 0).sum(), "row norms^2 smallest", np.array2string(np.sort((Jd ** 2).sum(1))[:24], precision=1))
    os.environ["CTVIO_MARG_HOST"] = "1"
    kh, Jh, rh = s.marginalize(0, role)
    print("host   marg nonzero rows", (np.abs(Jh).sum(1) > 0).sum(), "row norms^2 smallest", np.array2string(np.sort((Jh ** 2).sum(1))[:24], precision=1))
ko, Jo, ro = pyctvo.OracleWindow(m.copy()).marginalize(role, 1e-8)
print("oracle marg nonzero rows", (np.abs(Jo).sum(1) > 0).sum(), "row norms^2 smallest", np.array2string(np.sort((Jo ** 2).sum(1))[:24], precision=1))
