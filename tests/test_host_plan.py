"""CPU: invariants of the host-side planning of the visual blocks (csrc/host_pack.hpp, plan_window), compiled with g++ for the
test only (tests/host_plan_check.cpp): landmark-major slots, a landmark never straddles a group of 64 slots, the slot count is
a multiple of 64, every block has exactly one slot, the frame-pair order lists every block once, and more than 64 blocks of
one landmark are rejected."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROCM_INC = "/opt/rocm/include"


@pytest.fixture(scope="module")
def hp():
    if not os.path.isdir(ROCM_INC):
        pytest.skip("HIP headers not found")
    out = os.path.join(HERE, "_build", "libhostplan.so")
    src = os.path.join(HERE, "host_plan_check.cpp")
    hdrs = [os.path.join(HERE, "..", "ctrl-vio_amd", "csrc", f) for f in ("host_pack.hpp", "device_types.hpp")] + [os.path.join(HERE, "..", "include", "ctvio.h")]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or any(os.path.getmtime(f) > os.path.getmtime(out) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-I", ROCM_INC, "-o", out, src, "-L/opt/rocm/lib", "-lamdhip64",
                               "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    return C.CDLL(out)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def plan(hp, v_lm, v_ti, v_tj, v_rowi, v_rowj, L, vch=8):
    V = len(v_lm)
    a = [np.ascontiguousarray(x, t) for x, t in ((v_lm, np.int32), (v_ti, np.int64), (v_tj, np.int64), (v_rowi, np.int32), (v_rowj, np.int32))]
    cap = 64 * (V // 1 + 2) if V else 64
    Vp = C.c_int32(); nit = C.c_int32()
    lord = np.zeros(cap, np.int32); vpos = np.zeros(max(V, 1), np.int32); vord = np.zeros(max(V, 1), np.int32)
    err = C.create_string_buffer(256)
    rc = hp.hp_plan(V, L, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), vch, cap, C.byref(Vp), _p(lord), _p(vpos), _p(vord), C.byref(nit), err, 256)
    return rc, Vp.value, lord[:Vp.value], vpos[:V], vord[:V], nit.value, err.value.decode()


@pytest.mark.parametrize("seed", range(6))
def test_slot_layout_invariants(hp, seed):
    rng = np.random.default_rng(seed)
    L = int(rng.integers(1, 120))
    counts = rng.integers(0, 12, L)
    if seed == 3:
        counts[rng.integers(0, L)] = 64          # the largest landmark allowed
    v_lm = np.repeat(np.arange(L), counts)
    V = len(v_lm)
    perm = rng.permutation(V)                    # the caller's order is arbitrary
    v_lm = v_lm[perm]
    fi = rng.integers(0, 8, V); fj = fi + rng.integers(1, 6, V)
    v_ti = fi * 100; v_tj = fj * 100
    v_rowi = rng.integers(0, 1024, V); v_rowj = rng.integers(0, 1024, V)
    rc, Vp, lord, vpos, vord, nit, err = plan(hp, v_lm, v_ti, v_tj, v_rowi, v_rowj, L)
    assert rc == 0, err
    assert Vp % 64 == 0 and Vp >= V
    used = lord[lord >= 0]
    assert sorted(used.tolist()) == list(range(V))                       # every block exactly once
    assert all(lord[vpos[v]] == v for v in range(V))
    slots_of = {}
    for s, v in enumerate(lord):
        if v >= 0:
            slots_of.setdefault(int(v_lm[v]), []).append(s)
    prev_end = -1
    for l in sorted(slots_of):
        sl = slots_of[l]
        assert sl == list(range(sl[0], sl[0] + len(sl)))                  # consecutive slots
        assert sl[0] // 64 == sl[-1] // 64                                # inside one group of 64
        assert sl[0] > prev_end                                           # landmark-major, ascending
        prev_end = sl[-1]
        keys = [(int(v_ti[lord[s]]), int(v_tj[lord[s]]), int(v_rowi[lord[s]]), int(v_rowj[lord[s]])) for s in sl]
        assert keys == sorted(keys)                                       # frame-pair order inside a landmark
    assert sorted(vord.tolist()) == list(range(V))
    k = [(int(v_ti[v]), int(v_tj[v]), int(v_rowi[v]), int(v_rowj[v])) for v in vord]
    assert k == sorted(k)                                                 # the assembly's order
    # items: <= 8 blocks, never across a frame pair
    n_items, cnt, last = 0, 0, None
    for v in vord:
        fp = (int(v_ti[v]), int(v_tj[v]))
        if last != fp or cnt >= 8:
            n_items += 1; cnt = 0; last = fp
        cnt += 1
    assert nit == n_items


def test_more_than_64_blocks_of_a_landmark_rejected(hp):
    v_lm = np.zeros(65, np.int32)
    z = np.arange(65)
    rc, *_, err = plan(hp, v_lm, z * 0, z * 0 + 100, z, z, 1)
    assert rc == 1 and "64 observations" in err
