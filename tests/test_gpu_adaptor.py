"""GPU: the reference-shaped C++ adaptor (include/ctvio_estimator.hpp) -- pointers in, results written back in
place -- must give the same solve as the index-based C ABI used by the Python host."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _dump(w, path):
    with open(path, "w") as f:
        p = lambda *a: f.write(" ".join(repr(float(x)) if isinstance(x, (float, np.floating)) else str(int(x)) for x in a) + "\n")
        p(w.K, w.F, w.L, w.M, w.NB, w.V, w.pn, len(w.p_kind), w.t0_ns, w.dt_ns)
        for k in range(w.K):
            p(*w.quat[k].tolist(), *w.pos[k].tolist())
        for fr in range(w.F):
            p(*w.bias[fr].tolist())
        for l in range(w.L):
            p(float(w.rho[l]))
        p(float(w.ld), float(w.ld_lo), float(w.ld_hi), int(w.fix_ld))
        p(*w.q_CI.tolist()); p(*w.p_CI.tolist()); p(*w.gravity.tolist()); p(*w.imu_w.tolist()); p(float(w.img_w))
        if w.pn:
            p(*np.asfortranarray(w.pJ0).ravel(order="F").tolist())
            p(*w.pr0.tolist())
            for b in range(len(w.p_kind)):
                f.write(f"{int(w.p_kind[b])} {int(w.p_index[b])} {int(w.p_off[b])} " + " ".join(repr(float(x)) for x in w.p_x0[b]) + "\n")
        for m in range(w.M):
            f.write(f"{int(w.imu_t[m])} " + " ".join(repr(float(x)) for x in (*w.imu_gyro[m], *w.imu_acc[m])) + f" {int(w.imu_bias[m])}\n")
        for b in range(w.NB):
            f.write(f"{int(w.bc_i[b])} {int(w.bc_j[b])} " + " ".join(repr(float(x)) for x in w.bc_w[b]) + "\n")
        for v in range(w.V):
            f.write(f"{int(w.v_lm[v])} {int(w.v_ti[v])} {int(w.v_tj[v])} {int(w.v_rowi[v])} {int(w.v_rowj[v])} "
                    + " ".join(repr(float(x)) for x in (*w.v_pi[v], *w.v_pj[v])) + "\n")


def test_adaptor_matches_c_abi(cv, tmp_path):
    exe = str(tmp_path / "estimator_demo")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "estimator_demo.cpp"),
                           "-L", os.path.join(ROOT, "ctrl-vio_amd"), "-lctvio", "-Wl,-rpath," + os.path.join(ROOT, "ctrl-vio_amd"), "-o", exe])
    w0 = cv.synth.make_window("config1", seed=1005)
    _dump(w0, str(tmp_path / "in.txt"))
    out = subprocess.check_output([exe, str(tmp_path / "in.txt"), str(tmp_path / "out.txt"), "15", "fp64"], text=True)
    assert "ctvio: iterations" in out
    vals = open(tmp_path / "out.txt").read().split()
    arr = np.array(vals, float)
    K, F, L = w0.K, w0.F, w0.L
    kn = arr[:7 * K].reshape(K, 7)
    wa = w0.copy()
    wa.quat, wa.pos = kn[:, :4].copy(), kn[:, 4:].copy()
    wa.bias = arr[7 * K:7 * K + 6 * F].reshape(F, 6).copy()
    wa.rho = arr[7 * K + 6 * F:7 * K + 6 * F + L].copy()
    wa.ld = float(arr[7 * K + 6 * F + L])
    with cv.Solver(precision="fp64") as s:
        wg = w0.copy()
        s.set_windows([wg])
        sm = s.solve(15)[0]
    assert int(arr[-2]) == sm["iterations"]
    assert cv.rel_state_error(wa, wg)["state"] < 1e-9
