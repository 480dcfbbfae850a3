"""ctypes binding of libctvio.so (include/ctvio.h) -- plumbing only: every computation happens in the
HIP library.  There is NO CPU fallback: if the library is missing or no GPU is present, construction
raises (the product path must fail loudly, never silently compute elsewhere).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTVIO_LIB_PATH") or os.path.join(_HERE, "libctvio.so")   # (CTVIO_LIB_PATH: another build of the same source -- compiler-flag A/B runs)
import glob
SRC = [os.path.join(_HERE, "csrc", "ctvio.hip")] + sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hpp")))   # every header is a dependency
HDR = os.path.join(os.path.dirname(_HERE), "include", "ctvio.h")

FP64 = 1   # the only precision (ctvio.h: the mixed fp32 mode was removed)
TERMINATION = {0: "max-iterations", 1: "gradient-tolerance", 2: "parameter-tolerance", 3: "function-tolerance", 4: "min-radius", 5: "failure"}


def build_library(force: bool = False, verbose: bool = False) -> str:
    """hipcc cross-compiles for gfx950 without a GPU; the .so is kept in-tree so it ships with the repo snapshot."""
    deps = SRC + [HDR]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(f) for f in deps):
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize",
           "-o", LIB_PATH, SRC[0]]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("precision", C.c_int32), ("use_mfma", C.c_int32), ("check_every", C.c_int32),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
                ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
                ("max_consecutive_invalid_steps", C.c_int32), ("deterministic", C.c_int32), ("host_threads", C.c_int32),
                ("use_graph", C.c_int32), ("line_search", C.c_int32)]


class CWindow(C.Structure):
    _fields_ = [
        ("K", C.c_int32), ("F", C.c_int32), ("L", C.c_int32), ("M", C.c_int32), ("NB", C.c_int32), ("V", C.c_int32),
        ("pn", C.c_int32), ("pnb", C.c_int32),
        ("t0_ns", C.c_int64), ("dt_ns", C.c_int64),
        ("quat", C.c_void_p), ("pos", C.c_void_p), ("bias", C.c_void_p), ("rho", C.c_void_p),
        ("ld", C.c_double), ("ld_lo", C.c_double), ("ld_hi", C.c_double),
        ("fix_ld", C.c_int32), ("lock_bg", C.c_int32), ("lock_ba", C.c_int32), ("fixed_upto", C.c_int32),
        ("q_CI", C.c_double * 4), ("p_CI", C.c_double * 3), ("gravity", C.c_double * 3), ("imu_w", C.c_double * 6),
        ("img_w", C.c_double), ("cauchy_a", C.c_double),
        ("imu_t", C.c_void_p), ("imu_gyro", C.c_void_p), ("imu_acc", C.c_void_p), ("imu_bias", C.c_void_p),
        ("bc_i", C.c_void_p), ("bc_j", C.c_void_p), ("bc_w", C.c_void_p),
        ("v_lm", C.c_void_p), ("v_ti", C.c_void_p), ("v_tj", C.c_void_p), ("v_rowi", C.c_void_p), ("v_rowj", C.c_void_p),
        ("v_pi", C.c_void_p), ("v_pj", C.c_void_p),
        ("pJ0", C.c_void_p), ("pr0", C.c_void_p), ("p_kind", C.c_void_p), ("p_index", C.c_void_p), ("p_off", C.c_void_p),
        ("p_x0", C.c_void_p), ("v_cauchy", C.c_void_p), ("knot_const", C.c_void_p),
    ]


class Summary(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("num_successful", C.c_int32), ("num_unsuccessful", C.c_int32),
                ("termination", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("final_radius", C.c_double), ("num_line_search_steps", C.c_int32), ("num_line_search_reduced", C.c_int32)]

    def as_dict(self):
        return dict(iterations=self.iterations, num_successful=self.num_successful, num_unsuccessful=self.num_unsuccessful,
                    termination=TERMINATION.get(self.termination, "?"), initial_cost=self.initial_cost,
                    final_cost=self.final_cost, final_radius=self.final_radius,
                    num_line_search_steps=self.num_line_search_steps, num_line_search_reduced=self.num_line_search_reduced)


# every symbol include/ctvio.h declares (tests check the .so exports all of them)
SYMBOLS = ["ctvio_default_options", "ctvio_status_string", "ctvio_last_error", "ctvio_device_count", "ctvio_create",
           "ctvio_destroy", "ctvio_clear", "ctvio_add_window", "ctvio_upload", "ctvio_set_batch", "ctvio_num_windows", "ctvio_solve",
           "ctvio_get_state", "ctvio_get_batch_state", "ctvio_set_state", "ctvio_snapshot_state", "ctvio_restore_state", "ctvio_linearize", "ctvio_cost", "ctvio_lm_step", "ctvio_spline_eval", "ctvio_sensor_pose", "ctvio_gauge_restore", "ctvio_marginalize", "ctvio_marginalize_batch", "ctvio_residual_summary",
           "ctvio_last_timing", "ctvio_set_profiling", "ctvio_stream", "ctvio_solve_sharded", "ctvio_sharded_release", "ctvio_shard_of",
           "ctvio_shard_count", "ctvio_shards_used", "ctvio_spline_eval_batch", "ctvio_graph_captures", "ctvio_marginalize_ran_on_host"]

_lib = None


def load_library():
    """dlopen libctvio.so (raises if it has not been built -- no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        lib = C.CDLL(LIB_PATH)
        lib.ctvio_status_string.restype = C.c_char_p
        lib.ctvio_last_error.restype = C.c_char_p
        lib.ctvio_stream.restype = C.c_void_p
        lib.ctvio_create.argtypes = [C.POINTER(Options), C.POINTER(C.c_void_p)]
        lib.ctvio_destroy.argtypes = [C.c_void_p]
        for name in ("ctvio_clear", "ctvio_upload", "ctvio_num_windows", "ctvio_snapshot_state", "ctvio_restore_state"):
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.ctvio_add_window.argtypes = [C.c_void_p, C.POINTER(CWindow), C.POINTER(C.c_int32)]
        lib.ctvio_set_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.ctvio_get_batch_state.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        lib.ctvio_solve.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.ctvio_get_state.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 5
        lib.ctvio_set_state.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 4 + [C.c_double]
        lib.ctvio_linearize.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 5
        lib.ctvio_cost.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
        lib.ctvio_lm_step.argtypes = [C.c_void_p, C.c_int32, C.c_double, C.c_void_p, C.c_void_p]
        lib.ctvio_spline_eval.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 5
        lib.ctvio_sensor_pose.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 4
        lib.ctvio_marginalize.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_double] + [C.c_void_p] * 4
        lib.ctvio_marginalize_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_double] + [C.c_void_p] * 4
        lib.ctvio_residual_summary.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        lib.ctvio_gauge_restore.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 4
        lib.ctvio_last_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ctvio_set_profiling.argtypes = [C.c_void_p, C.c_int32]
        lib.ctvio_stream.argtypes = [C.c_void_p]
        lib.ctvio_graph_captures.argtypes = [C.c_void_p]
        lib.ctvio_marginalize_ran_on_host.argtypes = [C.c_void_p]
        lib.ctvio_solve_sharded.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32] + [C.c_void_p] * 6
        lib.ctvio_shard_of.argtypes = [C.c_int32, C.c_int32]
        lib.ctvio_shard_count.argtypes = [C.c_int32, C.c_int32, C.c_int32]
        lib.ctvio_shards_used.argtypes = [C.c_int32, C.c_int32]
        lib.ctvio_spline_eval_batch.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 7
        _lib = lib
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def to_cwindow(w, keep):
    """Window (ctrl-vio_amd/window.py) -> ctvio_window.  `keep` collects arrays that must outlive the call."""
    w.normalize()
    pJ0_cm = np.asfortranarray(w.pJ0)
    keep.append(pJ0_cm)
    c = CWindow()
    c.K, c.F, c.L, c.M, c.NB, c.V = w.K, w.F, w.L, w.M, w.NB, w.V
    c.pn, c.pnb = w.pn, int(w.p_kind.shape[0])
    c.t0_ns, c.dt_ns = w.t0_ns, w.dt_ns
    c.quat, c.pos, c.bias, c.rho = _p(w.quat), _p(w.pos), _p(w.bias), _p(w.rho)
    c.ld, c.ld_lo, c.ld_hi = w.ld, w.ld_lo, w.ld_hi
    c.fix_ld, c.lock_bg, c.lock_ba, c.fixed_upto = int(w.fix_ld), int(w.lock_bg), int(w.lock_ba), int(w.fixed_upto)
    c.q_CI[:] = w.q_CI.tolist(); c.p_CI[:] = w.p_CI.tolist(); c.gravity[:] = w.gravity.tolist(); c.imu_w[:] = w.imu_w.tolist()
    c.img_w, c.cauchy_a = w.img_w, w.cauchy_a
    c.imu_t, c.imu_gyro, c.imu_acc, c.imu_bias = _p(w.imu_t), _p(w.imu_gyro), _p(w.imu_acc), _p(w.imu_bias)
    c.bc_i, c.bc_j, c.bc_w = _p(w.bc_i), _p(w.bc_j), _p(w.bc_w)
    c.v_lm, c.v_ti, c.v_tj = _p(w.v_lm), _p(w.v_ti), _p(w.v_tj)
    c.v_rowi, c.v_rowj, c.v_pi, c.v_pj = _p(w.v_rowi), _p(w.v_rowj), _p(w.v_pi), _p(w.v_pj)
    c.pJ0 = pJ0_cm.ctypes.data_as(C.c_void_p) if w.pn else None
    c.pr0, c.p_kind, c.p_index, c.p_off, c.p_x0 = _p(w.pr0), _p(w.p_kind), _p(w.p_index), _p(w.p_off), _p(w.p_x0)
    c.v_cauchy = _p(w.v_cauchy) if w.v_cauchy is not None else None
    c.knot_const = _p(w.knot_const) if w.knot_const is not None else None
    return c


class CtvioError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        lib = load_library()
        raise CtvioError(f"{lib.ctvio_status_string(rc).decode()} ({rc}): {lib.ctvio_last_error().decode()}")
