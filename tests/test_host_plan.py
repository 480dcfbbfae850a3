"""CPU: invariants of the host-side planning of the visual blocks (csrc/host_pack.hpp, plan_window), compiled with g++ for the
test only (tests/host_plan_check.cpp): landmark-major slots, a landmark never straddles a group of 64 slots, the slot count is
a multiple of 64, every block has exactly one slot, the frame-pair order lists every block once, anchors = the distinct
(landmark, t_i, row_i, p_i), numbered landmark-major with an anchor's blocks consecutive, and more than 64 blocks of one
landmark are rejected."""
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


def plan(hp, v_lm, v_ti, v_tj, v_rowi, v_rowj, L, vch=8, v_pi=None, anchors=False):
    V = len(v_lm)
    a = [np.ascontiguousarray(x, t) for x, t in ((v_lm, np.int32), (v_ti, np.int64), (v_tj, np.int64), (v_rowi, np.int32), (v_rowj, np.int32))]
    pi = np.ascontiguousarray(np.zeros((max(V, 1), 2)) if v_pi is None else v_pi, np.float64)
    cap = 64 * (V // 1 + 2) if V else 64
    Vp = C.c_int32(); nit = C.c_int32(); A = C.c_int32()
    lord = np.zeros(cap, np.int32); vpos = np.zeros(max(V, 1), np.int32); vord = np.zeros(max(V, 1), np.int32)
    anc_of = np.zeros(max(V, 1), np.int32); anc_rep = np.zeros(max(V, 1), np.int32)
    err = C.create_string_buffer(256)
    rc = hp.hp_plan(V, L, _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(a[4]), _p(pi), vch, cap, C.byref(Vp), _p(lord), _p(vpos), _p(vord), C.byref(nit),
                    C.byref(A), _p(anc_of), _p(anc_rep), err, 256)
    out = (rc, Vp.value, lord[:Vp.value], vpos[:V], vord[:V], nit.value, err.value.decode())
    return out + (anc_of[:V], anc_rep[:A.value]) if anchors else out


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
    v_rowj = rng.integers(0, 1024, V)
    # the reference's shape: every block of a landmark against the landmark's first observation (one anchor per landmark) -- except,
    # on odd seeds, a sprinkle of blocks with their own i end (the C ABI allows arbitrary blocks)
    lm_row = rng.integers(0, 1024, L); lm_pi = rng.normal(size=(L, 2)); lm_fi = rng.integers(0, 8, L)
    v_rowi = lm_row[v_lm]; v_pi = lm_pi[v_lm].copy(); fi = lm_fi[v_lm]; fj = fi + rng.integers(1, 6, V)
    if seed % 2:
        odd = rng.random(V) < 0.15
        v_rowi = np.where(odd, rng.integers(0, 1024, V), v_rowi)
        v_pi[odd, 0] += 0.5
        fi = np.where(rng.random(V) < 0.1, (fi + 1) % 8, fi); fj = np.maximum(fj, fi + 1)
    v_ti = fi * 100; v_tj = fj * 100
    rc, Vp, lord, vpos, vord, nit, err, anc_of, anc_rep = plan(hp, v_lm, v_ti, v_tj, v_rowi, v_rowj, L, v_pi=v_pi, anchors=True)
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
        ancs = [int(anc_of[lord[s]]) for s in sl]
        assert ancs == sorted(ancs)                                       # anchor by anchor inside a landmark
        for a in set(ancs):
            keys = [(int(v_ti[lord[s]]), int(v_tj[lord[s]]), int(v_rowi[lord[s]]), int(v_rowj[lord[s]])) for s in sl if anc_of[lord[s]] == a]
            assert keys == sorted(keys)                                   # frame-pair order inside an anchor
    # anchors: exactly the distinct (landmark, t_i, row_i, p_i); numbered landmark-major; the representative carries the anchor's key
    akey = lambda v: (int(v_lm[v]), int(v_ti[v]), int(v_rowi[v]), float(v_pi[v, 0]), float(v_pi[v, 1]))
    assert len(anc_rep) == len({akey(v) for v in range(V)})
    assert all(akey(anc_rep[anc_of[v]]) == akey(v) for v in range(V))
    assert [int(v_lm[v]) for v in anc_rep] == sorted(int(v_lm[v]) for v in anc_rep)
    if seed % 2 == 0:
        assert len(anc_rep) == len(set(v_lm.tolist()))                    # the reference's shape: one anchor per observed landmark
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
