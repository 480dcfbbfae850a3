"""CPU: the oracle under AddressSanitizer + UBSan (oracle/Makefile `asan` target, SURVEY section 5's race / memory checking plan for the CPU
side): the C restatement every GPU parity test trusts is run -- cost, normal equations, LM step with and without Schur elimination, a full
solve with the projected line search, spline queries, marginalisation -- on a tiny window, a rolling-shutter window whose line search
shortens steps and a window with constant knots / per-block losses, in a fresh interpreter with libasan preloaded.  Any out-of-bounds
access, use of uninitialised stack arrays through UB, signed overflow or misaligned access aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ["CTV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["CTV_ROOT"], "oracle"))
cv = importlib.import_module("ctrl-vio_amd")
import pyctvo
ws = [cv.synth.make_window("tiny", seed=7), cv.synth.make_window("tiny", seed=3020, img_h=640, ld_true=3.0e-5), cv.synth.make_window("config1", seed=1001)]
ws[2].knot_const = np.zeros(ws[2].K, np.uint8); ws[2].knot_const[[0, 5]] = 1
ws[2].v_cauchy = np.where(np.arange(ws[2].V) % 3 == 0, 1.0, 2.0)
for w in ws:
    o = pyctvo.OracleWindow(w.copy())
    c = o.cost(); H, g, c2 = o.build_normal()
    assert np.isfinite(c) and abs(c - c2) <= 1e-9 * abs(c)
    d1, m1 = o.lm_step(1e4, use_schur=True); d2, m2 = o.lm_step(1e4, use_schur=False)
    assert np.abs(d1 - d2).max() <= 1e-7 * np.abs(d2).max()
    ref = w.copy(); so = pyctvo.OracleWindow(ref).solve(15)
    assert so.final_cost < so.initial_cost
    t = w.t0_ns + np.arange(5, dtype=np.int64) * (w.dt_ns // 3)
    pyctvo.OracleWindow(w.copy()).spline_eval(t)
    role = np.full(w.N, -1, np.int8); role[:12] = 1; role[12:36] = 0
    pyctvo.OracleWindow(w.copy()).marginalize(role)
print("ASAN_ORACLE_OK")
"""


def test_oracle_under_address_and_ub_sanitizers():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    lib = os.path.join(ROOT, "oracle", "_build", "libctvo_oracle_asan.so")
    env = dict(os.environ, CTV_ROOT=ROOT, CTVO_ORACLE_LIB=lib, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0 and "ASAN_ORACLE_OK" in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])
