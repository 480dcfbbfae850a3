"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/ctvio.h declares, and
refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest


def test_header_symbols_exported(cv):
    cv.capi.build_library()
    lib = cv.capi.load_library()
    hdr = open(cv.capi.HDR).read()
    declared = set(re.findall(r"\b(ctvio_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(cv.capi.SYMBOLS), declared ^ set(cv.capi.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s


def test_default_options_and_strings(cv):
    lib = cv.capi.load_library()
    o = cv.capi.Options()
    lib.ctvio_default_options(C.byref(o))
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e-10, 1e-8)
    assert o.initial_radius == 1e4 and o.max_consecutive_invalid_steps == 5 and o.precision == cv.capi.FP64 and o.use_graph == 1 and o.host_threads == 0
    assert lib.ctvio_status_string(0) == b"ok"
    assert b"no CPU fallback" in lib.ctvio_status_string(2)


def test_no_gpu_fails_loudly(cv):
    lib = cv.capi.load_library()
    if lib.ctvio_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.ctvio_create(None, C.byref(h))
    assert rc == 2 and not h.value
    with pytest.raises(cv.capi.CtvioError):
        cv.Solver()


def test_struct_layout_matches_header(cv):
    """ctvio_window / ctvio_options field order in the ctypes mirror follows the header."""
    hdr = open(cv.capi.HDR).read()
    body = hdr[hdr.index("typedef struct ctvio_window {"):hdr.index("} ctvio_window;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct ctvio_window {", "").strip()
        if not decl:
            continue
        decl = re.sub(r"^(const\s+)?(int32_t|int64_t|uint8_t|double)\s*", "", decl)
        for part in decl.split(","):
            part = part.strip().lstrip("*").strip()
            part = re.sub(r"\[\d+\]", "", part)
            if part:
                names.append(part)
    assert names == [f[0] for f in cv.capi.CWindow._fields_]


def test_adaptor_header_compiles_standalone():
    """include/ctvio_estimator.hpp (the reference-shaped C++ adaptor) and the demo that uses it must compile with the host
    compiler alone -- no HIP, no torch headers in the drop-in boundary."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for demo in ("estimator_demo.cpp", "slide_demo.cpp"):
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-fsyntax-only", "-I" + os.path.join(root, "include"),
                               os.path.join(root, "tests", demo)])
    src = open(os.path.join(root, "include", "ctvio.h")).read()
    includes = re.findall(r"#\s*include\s*[<\"]([^>\"]+)[>\"]", src)
    assert all(not inc.startswith(("hip", "torch", "ATen", "c10")) for inc in includes), includes


def test_cxx_shard_partition_matches_the_python_rule(cv):
    """ctvio_shard_of / ctvio_shard_count (the partition behind ctvio_solve_sharded: window w -> device w mod G) against
    ctrl-vio_amd/sharding.shard -- no device, no process group needed."""
    lib = cv.capi.load_library()
    for n in (1, 7, 64, 65):
        for G in (1, 2, 3, 8):
            owners = [lib.ctvio_shard_of(w, G) for w in range(n)]
            for g in range(G):
                mine = [w for w in range(n) if owners[w] == g]
                assert mine == list(cv.sharding.shard(n, g, G))
                assert lib.ctvio_shard_count(n, g, G) == len(mine)
            assert sum(lib.ctvio_shard_count(n, g, G) for g in range(G)) == n
    if lib.ctvio_device_count() <= 0:     # without a GPU the sharded entry fails loudly like every other one
        w = cv.synth.make_window("tiny", seed=1)
        keep = []
        arr = (cv.capi.CWindow * 1)()
        arr[0] = cv.capi.to_cwindow(w, keep)
        assert lib.ctvio_solve_sharded(None, 0, 1, C.cast(arr, C.c_void_p), 5, None, None, None, None, None, None) == 2
