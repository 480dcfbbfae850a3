"""CPU: the flat window file (include/ctvio_window_io.hpp / ctrl-vio_amd/window_io.py): what Python writes a C++ program reads and writes back
byte for byte, and Python reads its own file back to the window it wrote -- the way C / C++ callers get the synthetic windows of bench.py."""
import filecmp
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def test_python_writes_cxx_reads_and_writes_back(cv, tmp_path):
    ws = [cv.synth.make_window("tiny", seed=5), cv.synth.make_window("config1", seed=1001), cv.synth.make_window("tiny", seed=6, with_prior=False)]
    ws[1].knot_const = np.zeros(ws[1].K, np.uint8); ws[1].knot_const[[0, 3]] = 1
    ws[1].v_cauchy = np.where(np.arange(ws[1].V) % 2 == 0, 1.0, 2.0)
    ws[0].fix_ld = True; ws[0].ld = 1.5e-5
    a, b = str(tmp_path / "a.ctvw"), str(tmp_path / "b.ctvw")
    cv.window_io.save_windows(a, ws)
    exe = os.path.join(HERE, "_build", "window_io_check")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(HERE, "window_io_check.cpp")])
    out = subprocess.run([exe, a, b], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("3 windows") and f"P0 = {ws[0].P}" in out.stdout
    assert filecmp.cmp(a, b, shallow=False)
    back = cv.window_io.load_windows(b)
    assert len(back) == 3
    for w, r in zip(ws, back):
        dw, dr = w.to_dict(), r.to_dict()
        assert dw.keys() == dr.keys()
        for k in dw:
            np.testing.assert_array_equal(dw[k], dr[k], err_msg=k)
