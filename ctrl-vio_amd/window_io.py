"""Flat binary file of a batch of windows (include/ctvio_window_io.hpp is the C++ twin): lets C / C++ callers run the synthetic windows of
synth.py -- the generator's NumPy Philox streams are not reproducible from C++ -- e.g. `python tools/export_windows.py config2 1000 64 w.ctvw`."""
from __future__ import annotations

import struct

import numpy as np

from .window import Window

MAGIC = b"CTVW0001"


def save_windows(path: str, windows) -> None:
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<i", len(windows)))
        for w in windows:
            w.normalize()
            f.write(struct.pack("<8i", w.K, w.F, w.L, w.M, w.NB, w.V, w.pn, int(w.p_kind.shape[0])))
            f.write(struct.pack("<2q", w.t0_ns, w.dt_ns))
            f.write(struct.pack("<3d", w.ld, w.ld_lo, w.ld_hi))
            f.write(struct.pack("<4i", int(w.fix_ld), int(w.lock_bg), int(w.lock_ba), int(w.fixed_upto)))
            f.write(np.concatenate([w.q_CI, w.p_CI, w.gravity, w.imu_w, [w.img_w, w.cauchy_a]]).astype("<f8").tobytes())
            f.write(struct.pack("<2i", int(w.v_cauchy is not None), int(w.knot_const is not None)))
            for a, t in ((w.quat, "<f8"), (w.pos, "<f8"), (w.bias, "<f8"), (w.rho, "<f8"), (w.imu_t, "<i8"), (w.imu_gyro, "<f8"), (w.imu_acc, "<f8"),
                         (w.imu_bias, "<i4"), (w.bc_i, "<i4"), (w.bc_j, "<i4"), (w.bc_w, "<f8"), (w.v_lm, "<i4"), (w.v_ti, "<i8"), (w.v_tj, "<i8"),
                         (w.v_rowi, "<i4"), (w.v_rowj, "<i4"), (w.v_pi, "<f8"), (w.v_pj, "<f8"), (np.asfortranarray(w.pJ0).ravel(order="F"), "<f8"),
                         (w.pr0, "<f8"), (w.p_kind, "<i4"), (w.p_index, "<i4"), (w.p_off, "<i4"), (w.p_x0, "<f8")):
                f.write(np.ascontiguousarray(a).astype(t).tobytes())
            if w.v_cauchy is not None:
                f.write(np.ascontiguousarray(w.v_cauchy).astype("<f8").tobytes())
            if w.knot_const is not None:
                f.write(np.ascontiguousarray(w.knot_const).astype("u1").tobytes())


def load_windows(path: str):
    out = []
    with open(path, "rb") as f:
        if f.read(8) != MAGIC:
            raise ValueError("not a CTVW0001 file")
        n, = struct.unpack("<i", f.read(4))
        rd = lambda t, cnt: np.frombuffer(f.read(np.dtype(t).itemsize * cnt), dtype=t).copy()
        for _ in range(n):
            K, F, L, M, NB, V, pn, pnb = struct.unpack("<8i", f.read(32))
            t0, dt = struct.unpack("<2q", f.read(16))
            ld, lo, hi = struct.unpack("<3d", f.read(24))
            fix_ld, lbg, lba, fup = struct.unpack("<4i", f.read(16))
            cal = rd("<f8", 18)
            has_c, has_k = struct.unpack("<2i", f.read(8))
            w = Window(t0_ns=t0, dt_ns=dt, quat=rd("<f8", 4 * K).reshape(K, 4), pos=rd("<f8", 3 * K).reshape(K, 3), bias=rd("<f8", 6 * F).reshape(F, 6),
                       rho=rd("<f8", L), ld=ld, ld_lo=lo, ld_hi=hi, fix_ld=bool(fix_ld), lock_bg=bool(lbg), lock_ba=bool(lba), fixed_upto=fup,
                       q_CI=cal[:4], p_CI=cal[4:7], gravity=cal[7:10], imu_w=cal[10:16], img_w=float(cal[16]), cauchy_a=float(cal[17]))
            w.imu_t = rd("<i8", M); w.imu_gyro = rd("<f8", 3 * M).reshape(M, 3); w.imu_acc = rd("<f8", 3 * M).reshape(M, 3); w.imu_bias = rd("<i4", M)
            w.bc_i = rd("<i4", NB); w.bc_j = rd("<i4", NB); w.bc_w = rd("<f8", 6 * NB).reshape(NB, 6)
            w.v_lm = rd("<i4", V); w.v_ti = rd("<i8", V); w.v_tj = rd("<i8", V); w.v_rowi = rd("<i4", V); w.v_rowj = rd("<i4", V)
            w.v_pi = rd("<f8", 2 * V).reshape(V, 2); w.v_pj = rd("<f8", 2 * V).reshape(V, 2)
            w.pJ0 = rd("<f8", pn * pn).reshape(pn, pn, order="F"); w.pr0 = rd("<f8", pn)
            w.p_kind = rd("<i4", pnb); w.p_index = rd("<i4", pnb); w.p_off = rd("<i4", pnb); w.p_x0 = rd("<f8", 4 * pnb).reshape(pnb, 4)
            if has_c:
                w.v_cauchy = rd("<f8", V)
            if has_k:
                w.knot_const = rd("u1", K)
            out.append(w.normalize())
    return out
