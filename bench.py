#!/usr/bin/env python
"""bench.py -- sliding-window solves/sec on MI355X (BASELINE.json metric, configs[1] workload).

A "step" = `--windows` independent config-2 windows (10 KF / 200 landmarks / 2000 IMU, <= 15 LM iterations, Ceres
tolerances and projected line search) per GPU, taken END TO END through the C ABI exactly as SURVEY.md section 8d defines one
solve: ctvio_set_batch (validate + pack on host threads + one H2D copy) -> ctvio_solve (device-resident LM) ->
ctvio_get_batch_state (one D2H copy).  The windows of a step are split over `--streams` solver handles, each driven by its
own host thread through the K steps; at most `--gpu-slots` handles are inside ctvio_solve at a time, so the others pack,
upload and read back while the GPU stays busy.  value = windows solved by all ranks / wall-clock of the K timed steps (max
over ranks).  The device-resident rate (state restored on the device, no packing or PCIe traffic) is
reported next to it as `device_resident_solves_per_s`.

N > 1: `python bench.py --gpus N` re-executes itself under torch.distributed.run (one process per GPU, backend nccl = RCCL);
windows are sharded by id (w mod N), no data-path collective (independent windows); RCCL carries the
barrier, the max-over-ranks time and the all-gather of the per-window result records (every id must come back exactly once).

Output contract: the LAST stdout line is ONE compact JSON object (< 4 KB: the driver keeps an 8 KB tail) -- the contract keys, `config`,
`roofline` (dominant kernel), `cpu_baseline`, `parity`, `device_resident_solves_per_s`, a few headline scalars and `details_file`.  Everything
else (per-kernel rooflines, configs 3 / 5 / tumrs, small batches, host-side figures) goes to that file: gpurun_out/bench_details_n<N>.json.

Objects (compact line: roofline, parity, cpu_baseline; the rest in the details file):
  roofline       dominant kernel of a profiled solve of one handle (HIP events on the solver's stream, other handles idle):
                 achieved = algorithmic bytes or flops per launch / average launch duration, against 8 TB/s HBM
                 or 78.6 TFLOP/s fp64 -- whichever roof the kernel is closer to; both fractions, the PMC traffic and the issue counters
                 of the committed rocprofv3 passes ride along; roofline_kernels: the same line for every named kernel; roofline_mfma:
                 the Schur SYRK (also under config5 for K = 64).
  parity         max relative state error of the timed path against the fp64 CPU oracle on a sample of the windows
  cpu_baseline   the oracle (a port of the reference's Ceres path: oracle/ctvo.c) on 1 host core, the same sample;
                 cpu_baseline_all_cores: one window per thread on every host core (configs[3]).
  small_batches  8 and 64 windows per launch (configs[3] as written: 64 windows over 8 GPUs); config3.spline_eval: the batched per-row
                 trajectory query (7040 row times per window) with its HBM roofline.
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "fp64": 78.6}   # v_mfma_f32_32x32x2_f32 (guide) / v_mfma_f64_16x16x4_f64 (AMD MI355X spec sheet: 78.6 TF fp64 matrix)


def kernel_of_phase(precision):
    return {"k_imu_linearize": "k_imu_linearize_f64", "k_vis_eval": "k_vis_anchor + k_vis_eval",
            "k_assemble_vis": "k_assemble_vis_mfma", "k_schur_mfma": "k_schur_window_f64", "k_cholesky_solve": "k_cholesky_flow"}


# rocprofv3 names of the kernels behind a launch group (counter tables: profiles/pmc_traffic.json, profiles/pmc_issue.json)
PMC_KERNELS = {"k_imu_linearize": ["k_imu_linearize_f64"], "k_vis_eval": ["k_vis_anchor", "k_vis_eval"], "k_assemble_vis": ["k_assemble_vis_mfma"],
               "k_schur_mfma": ["k_schur_window_f64"], "k_cholesky_solve": ["k_cholesky_flow"]}
# fp64 instruction counts of the visual evaluation bodies, read off the ISA (tools/vis_isa_count.sh): per block 714 fp64 VALU instructions
# = 972 flop (+ 24 + 6 + 72 of the landmark-row contributions), per anchor 813 = 1190 flop
VIS_BLOCK_FLOP, VIS_ANCHOR_FLOP = 972 + 2 * (24 + 6) + 2 * 72 // 4, 1190


def synth_window(cv, config, seed):
    w = cv.synth.make_window(config, seed=seed)
    w._bench_key = (config, seed)
    return w


def imu_groups(w):
    return len({(int((t - w.t0_ns) // w.dt_ns), int(b)) for t, b in zip(w.imu_t, w.imu_bias)})


def vis_anchors(w):
    """distinct i ends (landmark, t_i, row_i, p_i) of the window's visual blocks: one record each (csrc/host_pack.hpp)"""
    return len({(int(l), int(t), int(r), float(p[0]), float(p[1])) for l, t, r, p in zip(w.v_lm, w.v_ti, w.v_rowi, w.v_pi)})


def workload_label(w, config, iters):
    return (f"{config}: {w.F - 1} KF / {w.L} landmarks / {w.M} IMU sliding window (K = {w.K} knots, {w.V} reprojection blocks), "
            f"<= {iters} LM iterations (Ceres 1.14 trust region + projected line search)")


_CV = None
_MEMO = {}


def _cv():
    global _CV
    if _CV is None:
        _CV = importlib.import_module("ctrl-vio_amd")
    return _CV


def sparsity(w):
    """Per distinct window (memoised): non-zeros of W by landmark (its planned knot span's columns + the line delay), entries of the reduced
    system's lower triangle inside the envelope, and the 16 x 16 tiles of it that receive Schur products (packer.py mirrors host_pack.hpp)."""
    key = getattr(w, "_bench_key", None) or id(w)     # (copies of a synthetic window carry the (config, seed) tag of their original)
    if key not in _MEMO:
        pk = _cv().packer
        klo, khi = pk.landmark_spans(w)
        nnz = [6 * int(b - a + 1) + 1 if b >= 0 else 0 for a, b in zip(klo, khi)]
        P, K6 = w.P, 6 * w.K
        nzr = lambda b: 16 * b < K6 or (P >= 16 * b and P - 1 < 16 * b + 16)
        nzc = lambda b: 16 * b < K6 or (P - 1 >= 16 * b and P - 1 < 16 * b + 16)
        nt = P // 16 + 1
        _MEMO[key] = {"w_nnz": sum(nnz), "nz_flops": pk.schur_nonzero_flops(w), "env": pk.envelope_entries(w, dense=P <= 223),
                      "product_tiles": sum(1 for i in range(nt) for j in range(i + 1) if nzr(i) and nzc(j)), "groups": imu_groups(w), "anchors": vis_anchors(w)}
    return _MEMO[key]


def algorithmic_bytes(w, phase, fp_bytes):
    """Algorithmic HBM bytes of ONE window for one launch of a kernel group: what the launch must move if every operand
    is touched once; fp_bytes = size of the linearisation scalar (8 in the product path)."""
    K, F, L, M, V, P = w.K, w.F, w.L, w.M, w.V, w.P
    sp = sparsity(w)
    G, A = sp["groups"], sp["anchors"]
    if phase == "k_imu_linearize":   # per sample u + 6 measurements; per group 4 knots (fp64 state) + bias + 32x32 tile out; + the part of the
        # normal equations this kernel CLEARS for the accumulating kernels behind it (imu_zero_share, zero_mode 1: the bias rows and the
        # line-delay row of Hpp -- rows 6K .. P - 1, ldh doubles each -- and the gradient): stores that are this kernel's job, not waste
        ldh = (P + 15) // 16 * 16
        return M * 7 * fp_bytes + G * (4 * 7 * 8 + 6 * 8 + 1024 * fp_bytes) + ((P - 6 * K) * ldh + P) * 8
    # A block's record is 40 doubles (rotation columns of its own end 24, inverse depth 2, line delay 2, residual 2, A~ 6, cp1 4); an
    # anchor's record 50 (p_G 3, GR 36, cp0 4, y 3, h 3).
    if phase == "k_vis_eval":        # anchors: inputs (t, row, obs, indices: 36 B) in, record out and in again once (the blocks read it);
        # blocks: own inputs (t, row, 2 obs, 3 indices, loss width: 48 B), record out; the knots are shared by the window's blocks and come
        # out of cache: counted once per window; the rows of W (knot + line-delay columns), Hll, g_rho
        # -- the rows of W over their landmarks' knot spans only (+ line delay), Hll, g_rho
        return A * (36 + 2 * 50 * 8) + V * (48 + 40 * fp_bytes) + K * 7 * 8 + sp["w_nnz"] * fp_bytes + L * (8 + 16)
    if phase == "k_assemble_vis":    # block records read once (38 of the 40 entries: the depth column is not needed) + keys + slot lists, GR
        # and cp0 of every anchor once (40 doubles), the packed fp64 Hessian flushed once
        K6 = 6 * K   # + the knot x knot part (24 x 24) of every IMU group tile, added into the same LDS Hessian
        return V * (38 * fp_bytes + 16) + A * 40 * 8 + (K6 * (K6 + 1) // 2 + K6 + 1) * 8 + G * 576 * fp_bytes
    if phase == "k_cholesky_solve":
        # P <= 223 (k_cholesky_flow): the triangle is READ once into registers and never written back; beyond (k_cholesky_solve): the entries
        # inside the envelope read and written once (the factor is needed by the back-substitution); rhs in, solution out
        return sp["env"] * 8 * (1 if P <= 223 else 2) + 2 * P * 8
    if phase == "k_schur_mfma":
        # the non-zeros of W + g_rho + 1 / (Hll + D) per row; P <= 223 (k_schur_window_f64): Hpp read and S written for the tiles that receive
        # products only (the factorisation takes the others straight from Hpp); beyond: every entry inside the envelope read (Hpp) and written (S)
        tri = sp["product_tiles"] * 256 if P <= 223 else sp["env"]
        return (sp["w_nnz"] + 2 * L) * fp_bytes + tri * 16
    return 0


def algorithmic_flops(w, phase):
    """Algorithmic flops of ONE window for one launch of a kernel group."""
    K, F, L, M, V, P = w.K, w.F, w.L, w.M, w.V, w.P
    if phase == "k_imu_linearize":
        # per sample: J^T [J r] lower triangle -- 3 accel rows x 28 columns without the pos x pos block (12 x 13 / 2 products per row), which
        # is w^2 sum lamA_k lamA_k' I3 (10 products per sample); 3 gyro rows x 16 columns -- + ~1.9 k for the evaluation in its staged
        # form (csrc/factors.hpp: 632 FMAs + 623 multiplies / adds per sample, counted in the ISA)
        return M * (2 * (3 * (28 * 29 // 2 - 12 * 13 // 2) + 10) + 2 * 3 * (16 * 17 // 2) + 1900)
    if phase == "k_vis_eval":        # counted in the ISA (tools/vis_isa_count.sh): one spline end per block, one per anchor
        return V * VIS_BLOCK_FLOP + sparsity(w)["anchors"] * VIS_ANCHOR_FLOP
    if phase == "k_assemble_vis":    # 48 x 48 lower triangle + line-delay and residual columns, 2 rows per block
        return V * 2 * 2 * (48 * 49 // 2 + 2 * 49)
    if phase == "k_cholesky_solve":
        return P ** 3 // 3 + 2 * P * P
    if phase == "k_schur_mfma":      # the NON-ZERO products of the SYRK (SURVEY 8d's nominal count P (P + 1) L is reported beside it)
        return sparsity(w)["nz_flops"]
    return 0


def nonzero_schur_flops(w):
    """Flops of the Schur complement's non-zero products: landmark l's row of W has nnz_l = 6 (knots of its span) + 1 entries, and
    contributes the lower triangle of their outer product: sum_l nnz_l (nnz_l + 1) -- config 2: 1.2 MF against the (6K + 1)(6K + 2) L = 4.2 MF
    of a W without bias columns and SURVEY's nominal P (P + 1) L = 8.9 MF; config 5: 6.7 MF against 148.6 / 326.6."""
    return sparsity(w)["nz_flops"]


def mfma_line(kernel, wl, avg_s, nwin):
    """The Schur SYRK against the fp64 matrix peak, by the products W's non-zeros have; the two larger counts are what a kernel blind to
    the sparsity would multiply."""
    pk = MFMA_PEAK_TFLOPS["fp64"]
    nz = sum(nonzero_schur_flops(w) for w in wl)
    return {"kernel": kernel, "bound": "mfma", "achieved": nz / avg_s / 1e12, "peak": pk, "unit": "TFLOP/s", "frac": nz / avg_s / 1e12 / pk,
            "avg_launch_us": 1e6 * avg_s, "flops_per_launch": nz, "flop_count": "non-zero products: sum over landmarks of nnz_l (nnz_l + 1), nnz_l = 6 x (knots of its span) + 1",
            "no_bias_columns_flops_per_launch": sum((6 * w.K + 1) * (6 * w.K + 2) * w.L for w in wl),
            "nominal_flops_per_launch": sum(w.P * (w.P + 1) * w.L for w in wl), "windows_per_launch": nwin}


def side_config(cv, lib, torch, config, nwin, nuniq, iters, steps, nseed, device, with_oracle, profile=False, label=None):
    """Device-resident rate of another BASELINE config on one handle, plus the state error of the first nseed windows OF THE TIMED BATCH (same
    handle, same launch shape, hence the same kernels) against the oracle.  profile: one more solve with HIP events around every launch
    group -> phase times, the dominant kernel's roofline line and the Schur SYRK's for this shape."""
    import numpy as np
    uniq = [synth_window(cv, config, 1000 + i) for i in range(max(nuniq, nseed))]
    out = {"windows_per_launch": nwin, "distinct_windows": nuniq}
    if label:
        out["workload"] = label
    with cv.Solver(device=device) as sv:
        wl = [uniq[i % nuniq].copy() for i in range(nwin)]
        sv.set_windows(wl)
        sv.snapshot_state()
        sv.solve_raw(iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sv.restore_state()
            sv.solve_raw(iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["solves_per_s"] = nwin * steps / dt
        out["ms_per_launch"] = 1e3 * dt / steps
        if profile:
            sv.set_profiling(True)
            sv.restore_state()
            sv.solve_raw(iters)
            sv.set_profiling(False)
            torch.cuda.synchronize()
            ms, n = sv.last_timing()
            names = cv.Solver.PHASES
            out["phase_ms_profiled_solve"] = {names[i]: float(ms[i]) for i in range(7)}
            big = wl[0].P > 223
            kn = {"k_imu_linearize": "k_imu_linearize_f64", "k_vis_eval": "k_vis_anchor + k_vis_eval", "k_assemble_vis": "k_assemble_vis_mfma",
                  "k_schur_mfma": "k_schur_tile2_f64" if big else "k_schur_window_f64", "k_cholesky_solve": "k_cholesky_solve<8> (envelope panels)" if big else "k_cholesky_flow"}
            lines = []
            for i in range(6):
                if n[i] <= 0 or names[i] not in kn:
                    continue
                avg_s = 1e-3 * ms[i] / int(n[i])
                nb = sum(algorithmic_bytes(w, names[i], 8) for w in wl); nf = sum(algorithmic_flops(w, names[i]) for w in wl)
                hb, ff = nb / avg_s / 1e9 / HBM_PEAK_GBS, nf / avg_s / 1e12 / MFMA_PEAK_TFLOPS["fp64"]
                lines.append({"kernel": kn[names[i]], "bound": "mfma" if ff > hb else "hbm", "achieved": nf / avg_s / 1e12 if ff > hb else nb / avg_s / 1e9,
                              "peak": MFMA_PEAK_TFLOPS["fp64"] if ff > hb else HBM_PEAK_GBS, "unit": "TFLOP/s" if ff > hb else "GB/s", "frac": max(hb, ff),
                              "hbm_frac": hb, "fp64_frac": ff, "traffic": None, "avg_launch_us": 1e6 * avg_s, "launches": int(n[i]),
                              "share_of_profiled_solve": float(ms[i] / max(sum(ms[:7]), 1e-12)), "windows_per_launch": nwin})
            lines.sort(key=lambda l: -l["share_of_profiled_solve"])
            out["roofline"] = lines[0] if lines else None
            out["roofline_kernels"] = lines
            out["roofline_mfma"] = mfma_line(kn["k_schur_mfma"] + " (K = %d, P = %d)" % (wl[0].K, wl[0].P), wl, 1e-3 * ms[4] / max(int(n[4]), 1), nwin)
            out["cholesky_us_per_launch"] = 1e3 * ms[5] / max(int(n[5]), 1)
            out["schur_plus_cholesky_ms_per_solve"] = float(ms[4] + ms[5])
        if with_oracle:
            import pyctvo
            # the first nseed windows of the TIMED batch are uniq[0 .. nseed): one more solve of that very batch, states read back from it
            sv.restore_state()
            sms = sv.solve(iters)
            errs = []
            for i in range(nseed):
                ref = uniq[i].copy()
                so = pyctvo.OracleWindow(ref).solve(iters)
                assert sms[i]["iterations"] == so.iterations, (config, i, sms[i], so.iterations)
                errs.append(cv.rel_state_error(wl[i], ref)["state"])
            out["max_rel_state_err"] = float(max(errs))
            out["parity_windows"] = nseed
            out["parity_from"] = "timed batch (the first %d windows of the %d-window launch that was timed: same handle, same kernels)" % (nseed, nwin)
    return out


def small_batch(cv, lib, torch, C, np, uniq, nwin, iters, device):
    """ONE launch of nwin windows (BASELINE configs[3] as written: 64 windows over 8 GPUs = 8 per GPU): end to end and device resident."""
    with cv.Solver(device=device, host_threads=min(nwin, 8)) as sb:
        keep = []
        arr = (cv.capi.CWindow * nwin)()
        wl = [uniq[i % len(uniq)] for i in range(nwin)]
        for j, w in enumerate(wl):
            arr[j] = cv.capi.to_cwindow(w, keep)
        K = sum(w.K for w in wl); F = sum(w.F for w in wl); L = sum(w.L for w in wl)
        o = (np.zeros((K, 4)), np.zeros((K, 3)), np.zeros((F, 6)), np.zeros(max(L, 1)), np.zeros(nwin))

        def one():
            cv.capi.check(lib.ctvio_set_batch(sb._h, nwin, C.cast(arr, C.c_void_p)))
            cv.capi.check(lib.ctvio_solve(sb._h, iters, None))
            cv.capi.check(lib.ctvio_get_batch_state(sb._h, *[cv.capi._p(a) for a in o]))
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            one()
        torch.cuda.synchronize()
        e2e = (time.perf_counter() - t0) / 20
        cv.capi.check(lib.ctvio_set_batch(sb._h, nwin, C.cast(arr, C.c_void_p)))
        sb.snapshot_state()
        sb.solve_raw(iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            sb.restore_state()
            sb.solve_raw(iters)
        torch.cuda.synchronize()
        res = (time.perf_counter() - t0) / 20
    return {"windows": nwin, "end_to_end_ms": 1e3 * e2e, "device_resident_ms": 1e3 * res, "end_to_end_solves_per_s": nwin / e2e,
            "device_resident_solves_per_s": nwin / res}


def row_queries(cv, torch, np, nwin, device):
    """pose + velocity + angular velocity at every rolling-shutter row time of every frame -- 11 frames x 640 rows
    = 7040 queries per window -- for a batch of config-3 windows in ONE launch (ctvio_spline_eval_batch).  Algorithmic HBM bytes per row:
    12 in (window id + time) + 104 out (13 doubles); the knots come out of cache."""
    uniq = [cv.synth.make_window("config3", seed=1000 + i) for i in range(8)]
    wl = [uniq[i % 8].copy() for i in range(nwin)]
    F, rows = wl[0].F, 640
    frame_t = np.arange(F, dtype=np.int64) * 100_000_000
    per = (frame_t[:, None] + (np.arange(rows, dtype=np.int64)[None, :] * int(3.0e-5 * 1e9))).ravel()     # t_f + row * line delay
    win = np.repeat(np.arange(nwin, dtype=np.int32), per.size)
    t = np.tile(per, nwin) + np.repeat(np.array([w.t0_ns for w in wl], np.int64), per.size)
    n = int(t.size)
    with cv.Solver(device=device) as sq:
        sq.set_windows(wl)
        sq.spline_eval_batch(win, t)                      # warm-up (scratch arenas grow once)
        kms, wall = [], []
        for _ in range(3):
            t0 = time.perf_counter()
            _, ms = sq.spline_eval_batch(win, t)
            wall.append(time.perf_counter() - t0); kms.append(ms)
    k = 1e-3 * min(kms)
    nb = 116.0 * n
    return {"windows": nwin, "rows_per_window": int(per.size), "rows": n, "kernel_us": 1e6 * k, "rows_per_s_kernel": n / k,
            "rows_per_s_end_to_end": n / min(wall), "algorithmic_bytes_per_row": 116,
            "roofline": {"kernel": "k_spline_eval (batch)", "bound": "hbm", "achieved": nb / k / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": nb / k / 1e9 / HBM_PEAK_GBS, "traffic": None},
            "note": "end to end includes the H2D copy of 12 B and the D2H copy of 104 B per row through pinned staging and the host-side scatter"}


def cpu_all_cores(config, iters, seed0, n):
    """the second CPU baseline -- all host cores, one window per process (oracle/all_cores.py in a fresh interpreter: the
    workers are forked without a HIP runtime).  NB pool.map hands chunk i to whichever worker is free: the windows a worker prepared
    are the ones it solves only when chunks and workers pair up, so every worker prepares on demand."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "all_cores.py"), config, str(iters), str(seed0), str(n)],
                       capture_output=True, text=True, timeout=900)
    if p.returncode != 0:
        return {"error": p.stderr[-500:]}
    r = json.loads(p.stdout.strip().splitlines()[-1])
    return {"value": r["solves"] / r["seconds"], "unit": "solves/s", "cores": r["processes"], "kind": "port",
            "sample": f"{r['solves']} solves of {config} windows (seeds {seed0}..{seed0 + n - 1}), fp64 C oracle (oracle/ctvo.c), one window per "
                      f"process on {r['processes']} processes = the CPUs this container may use (cgroup quota / affinity: {r['usable_cpus']}) "
                      f"of a host with {r['host_cores']} logical CPUs, {r['seconds']:.1f} s"}


COMPACT_LIMIT = 4096     # bytes of the final stdout line (tests/test_bench_line.py)


def csrc_sha256():
    """Hash of the kernel sources: profiles/pmc_traffic.json records the one its counters were collected with (tools/prof_summary.py), and a
    traffic figure from other sources is not printed -- a kernel change must not silently stale it.  (No git on the GPU box: a content hash.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ctrl-vio_amd", "csrc", "*"))):
        if f.endswith((".hip", ".hpp")):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def compact_line(out, details_file):
    """The driver-facing line: contract keys + config + the dominant kernel's roofline + cpu_baseline + parity + headline scalars."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out.get(k) for k in keep}
    c = out.get("config", {})
    line["config"] = {k: c.get(k) for k in ("workload", "timed_region", "windows_per_gpu_per_step", "distinct_windows_per_gpu", "streams_per_gpu",
                                            "concurrent_solves_per_gpu", "pack_threads_per_stream", "sharding") if k in c}
    r = out.get("roofline")
    if r:
        line["roofline"] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "avg_launch_us",
                                                  "windows_per_launch", "hbm_frac", "fp64_frac", "share_of_profiled_solve", "limiter") if k in r}
    m = out.get("roofline_mfma")
    if m:
        line["roofline_mfma"] = {k: m.get(k) for k in ("kernel", "frac", "avg_launch_us") if k in m}
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = None if not cb else {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
    p = out.get("parity")
    line["parity"] = None if not p else {k: p.get(k) for k in ("max_rel_state_err", "windows", "tolerance", "pass")}
    for k in ("device_resident_solves_per_s", "single_window_ms", "single_window_device_resident_ms"):
        if k in out:
            line[k] = out[k]
    if isinstance(out.get("mixed_batch"), dict) and "mixed_over_periodic" in out["mixed_batch"]:
        line["mixed_over_periodic_batch"] = out["mixed_batch"]["mixed_over_periodic"]
    for cfg in ("config3", "config5", "config5_spread", "tumrs"):
        if isinstance(out.get(cfg), dict) and "solves_per_s" in out[cfg]:
            line.setdefault("other_configs_solves_per_s", {})[cfg] = round(out[cfg]["solves_per_s"], 1)
    if "per_rank_solves_per_s" in out and out.get("n_gpus", 1) > 1:
        line["per_rank_solves_per_s"] = [round(x, 1) for x in out["per_rank_solves_per_s"]]
    if "small_batch_latency" in out:
        line["small_batch_latency_ms"] = [round(x, 3) for x in out["small_batch_latency"]["end_to_end_ms_per_step_by_rank"]]

    def rnd(v):   # 6 significant digits are plenty for a log line (and keep it short)
        if isinstance(v, float):
            return float(f"{v:.6g}")
        if isinstance(v, dict):
            return {k: rnd(x) for k, x in v.items()}
        if isinstance(v, list):
            return [rnd(x) for x in v]
        return v
    line = rnd(line)
    line["details_file"] = details_file
    s = json.dumps(line)
    if len(s) >= COMPACT_LIMIT:          # never lose the record to its size: drop the optional extras first
        for k in ("other_configs_solves_per_s", "roofline_mfma", "per_rank_solves_per_s", "small_batch_latency_ms"):
            line.pop(k, None)
        if line.get("cpu_baseline"):
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:200]
        s = json.dumps(line)
    assert len(s) < COMPACT_LIMIT, len(s)
    return s


def write_details(out, world):
    """Everything measured, in a side file (gpurun_out/ travels back from the GPU box); returns the path relative to the repo root."""
    rel = os.path.join("gpurun_out", f"bench_details_n{world}.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as f:
            json.dump(out, f, indent=1)
        return rel
    except OSError:
        return None


def mixed_batch(cv, torch, config, nwin, ndist, iters, device, steps=3):
    """`nwin` windows per launch made of `ndist` DISTINCT windows (seeds 5000 ..: generated by tools/make_windows.py on a process pool in a
    fresh interpreter) against the same launch made of the headline's 64 distinct ones: the 64-periodic batch terminates in lock-step, a mixed
    one shows whatever tail effects there are.  Device-resident, one handle."""
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "w.npz")
        t0 = time.perf_counter()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_windows.py"), config, "5000", str(ndist), f], capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            return {"error": p.stderr[-400:]}
        z = np.load(f)
        many = [cv.Window.from_dict(z, prefix=f"w{i}_") for i in range(ndist)]
        tgen = time.perf_counter() - t0
    few = [synth_window(cv, config, 1000 + i) for i in range(64)]
    out = {"windows_per_launch": nwin, "generation_s": tgen}
    for name, uniq in (("distinct_64", few), (f"distinct_{ndist}", many)):
        with cv.Solver(device=device) as sv:
            sv.set_windows([uniq[i % len(uniq)].copy() for i in range(nwin)])
            sv.snapshot_state()
            sms = sv.solve(iters, writeback=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                sv.restore_state()
                sv.solve_raw(iters)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            its = np.array([s["iterations"] for s in sms])
            out[name] = {"solves_per_s": nwin / dt, "ms_per_launch": 1e3 * dt, "iterations_min_mean_max": [int(its.min()), float(its.mean()), int(its.max())],
                         "terminations": sorted({s["termination"] for s in sms})}
    out["mixed_over_periodic"] = out[f"distinct_{ndist}"]["solves_per_s"] / out["distinct_64"]["solves_per_s"]
    return out


def respawn_under_torchrun(args):
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--windows", type=int, default=8192, help="independent windows per GPU per step")
    ap.add_argument("--unique", type=int, default=64, help="distinct synthetic windows per GPU (seeds 1000 + 64 rank + i), replicated to --windows")
    ap.add_argument("--config", default="config2")
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--precision", default="fp64", help="fp64 (the only arithmetic: all-fp64, like the reference)")
    ap.add_argument("--parity-sample", type=int, default=48, help="windows solved by the CPU oracle (state error of the timed path + cpu_baseline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-mode", action="store_true", help="(ignored: the mixed fp32 mode was removed)")
    ap.add_argument("--streams", type=int, default=4, help="solver handles (HIP streams + host threads) per GPU")
    ap.add_argument("--gpu-slots", type=int, default=0, help="handles allowed inside ctvio_solve at once (0: half of the streams, at least 1)")
    ap.add_argument("--host-threads", type=int, default=0, help="packing threads per handle (0: cores / streams, at most 16)")
    ap.add_argument("--device-resident-only", action="store_true", help="time the device-resident solve instead (diagnostics)")
    ap.add_argument("--quick", action="store_true", help="skip the side measurements (single window, configs 3 / 5, tumrs, 8-rank host share)")
    ap.add_argument("--shared-caller-buffers", action="store_true",
                    help="replicas of a distinct window share its caller buffers (rounds 1-4: the packer then reads 11 MB out of the host's L3 instead of streaming 1.5 GB)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)

    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # rank -> device.  One rank per GPU is the product layout (RCCL over xGMI).  When there are more ranks than visible devices (the
    # multi-rank path exercised on a one-GPU box: tests/test_gpu_multirank.py) ranks share devices and the collectives run over gloo on
    # host tensors -- RCCL cannot put two ranks on one device.
    ndev = max(torch.cuda.device_count(), 1)
    shared_device = world > ndev
    local = local % ndev
    torch.cuda.set_device(local)
    dist = None
    coll_device = torch.device("cpu") if shared_device else torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    cv = importlib.import_module("ctrl-vio_amd")
    import ctypes as C

    # ---- synthetic windows: global window id g = rank + world * j (sharding.shard), content = one of this rank's
    #      `unique` windows (seeds 1000 + unique * rank + i; configs[3] names seeds 1000..1063)
    nuniq = max(1, min(args.unique, args.windows))
    uniq = [synth_window(cv, args.config, 1000 + rank * nuniq + i) for i in range(nuniq)]
    nstream = max(1, min(args.streams, args.windows))
    per = [args.windows // nstream + (1 if i < args.windows % nstream else 0) for i in range(nstream)]
    first = np.concatenate([[0], np.cumsum(per)])          # local window index range of every handle
    my_ids = cv.sharding.shard(args.windows * world, rank, world)
    hthreads = args.host_threads or max(1, min(16, (os.cpu_count() or 8) // (nstream * max(world, 1))))
    solvers, cbatches, keeps, outs = [], [], [], []
    cbatches_shared = []
    for si in range(nstream):
        sv = cv.Solver(device=local, precision=args.precision, host_threads=hthreads)
        keep = []
        arr = (cv.capi.CWindow * per[si])()
        arr_sh = (cv.capi.CWindow * per[si])()
        wl = []
        for j in range(per[si]):
            w = uniq[(int(first[si]) + j) % nuniq]
            # EVERY window of a step owns its caller buffers (a copy of the distinct window it replicates): validate + pack stream the
            # whole batch from DRAM (8192 x 178 KB = 1.46 GB per step), as a caller with 8192 different windows would make them
            wc = w if args.shared_caller_buffers else w.copy()
            keep.append(wc)                                   # (the C window points into the copy's arrays)
            arr[j] = cv.capi.to_cwindow(wc, keep)
            arr_sh[j] = cv.capi.to_cwindow(w, keep)
            wl.append(w)
        cbatches_shared.append(arr_sh)
        K = sum(w.K for w in wl); F = sum(w.F for w in wl); L = sum(w.L for w in wl)
        outs.append((np.zeros((K, 4)), np.zeros((K, 3)), np.zeros((F, 6)), np.zeros(max(L, 1)), np.zeros(per[si])))
        solvers.append(sv); cbatches.append(arr); keeps.append((keep, wl))
    lib = cv.capi.load_library()

    # Handles run their own sub-batch of every step without waiting for each other (one host thread per handle); at most
    # --gpu-slots of them are inside ctvio_solve at a time, so a handle packs / copies while the others keep the GPU busy.
    nslots = args.gpu_slots if args.gpu_slots > 0 else max(1, nstream // 2)
    slots = threading.Semaphore(nslots)

    batches = {"own": cbatches, "shared": cbatches_shared}
    which = ["own"]

    def run_handle_steps(si, nsteps, resident):
        sv = solvers[si]
        for _ in range(nsteps):
            if resident:
                sv.restore_state()
                with slots:
                    sv.solve_raw(args.iters)
                continue
            cv.capi.check(lib.ctvio_set_batch(sv._h, per[si], C.cast(batches[which[0]][si], C.c_void_p)))      # validate + pack + H2D
            with slots:
                cv.capi.check(lib.ctvio_solve(sv._h, args.iters, None))                                 # device-resident LM
            o = outs[si]
            cv.capi.check(lib.ctvio_get_batch_state(sv._h, *[cv.capi._p(a) for a in o]))                # D2H

    def steps(nsteps, resident=False):
        """nsteps passes over the rank's windows; returns after every handle has drained its stream (results on the host)"""
        if nstream == 1:
            run_handle_steps(0, nsteps, resident)
            return
        th = [threading.Thread(target=run_handle_steps, args=(si, nsteps, resident)) for si in range(nstream)]
        for t in th: t.start()
        for t in th: t.join()

    def step(resident=False):
        steps(1, resident)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    rank_times = []

    def timed(nsteps, resident):
        barrier()
        t0 = time.perf_counter()
        steps(nsteps, resident)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        barrier()
        rank_times[:] = [t]
        if dist is not None:
            every = [torch.zeros(1, device=coll_device, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(every, torch.tensor([t], device=coll_device, dtype=torch.float64))
            rank_times[:] = [float(x.item()) for x in every]     # each rank's own clock: the first real SCALE run shows imbalance directly
            t = max(rank_times)
        return t

    def prepare_resident():
        """upload every handle's batch and keep a device-side copy of the initial state (ctvio_snapshot_state)"""
        for si in range(nstream):
            cv.capi.check(lib.ctvio_set_batch(solvers[si]._h, per[si], C.cast(cbatches[si], C.c_void_p)))
            solvers[si].snapshot_state()

    resident_headline = args.device_resident_only
    if resident_headline:
        prepare_resident()
    steps(args.warmup, resident_headline)
    t_total = timed(args.steps, resident_headline)
    per_rank_rate = [args.windows * args.steps / t for t in rank_times]
    n_solved = args.windows * world * args.steps
    fp_bytes = 8 if args.precision == "fp64" else 4
    w_lab = uniq[0]
    out = {
        "metric": "sliding-window solves/sec (%d KF, %d lm, %d IMU)" % (w_lab.F - 1, w_lab.L, w_lab.M), "value": n_solved / t_total, "unit": "solves/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_total / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64" if args.precision == "fp64" else "f32", "data": "synthetic",
        "config": {"workload": workload_label(w_lab, args.config, args.iters),
                   "timed_region": "device-resident solve only" if resident_headline else
                                   "end to end per batch: validate + pack (host threads) + H2D + LM solve + D2H of every state",
                   "windows_per_gpu_per_step": args.windows, "distinct_windows_per_gpu": nuniq, "streams_per_gpu": nstream, "concurrent_solves_per_gpu": nslots,
                   "pack_threads_per_stream": hthreads,
                   "caller_buffers": ("shared by the replicas of a distinct window (L3-resident input)" if args.shared_caller_buffers else
                                      "one set per window: pack + H2D stream the whole batch from DRAM"),
                   "sharding": f"independent windows, window id mod {world} rank(s), no data-path collective"},
        "per_rank_solves_per_s": per_rank_rate,
    }
    if args.windows <= 64:   # BASELINE configs[3] as written (64 windows over 8 GPUs = 8 per rank) is launch-latency territory: say what a step took
        out["small_batch_latency"] = {"windows_per_rank": args.windows, "end_to_end_ms_per_step_by_rank": [1e3 * t / args.steps for t in rank_times],
                                      "note": "one step = set_batch + solve + get_batch_state of this many windows on every rank, concurrently"}
    # ---- device-resident rate next to it (no packing, no PCIe: ctvio_restore_state on the device between solves)
    if not resident_headline:
        prepare_resident()
        step(True)
        nres = max(2, min(args.steps, 4))
        t_res = timed(nres, True)
        out["device_resident_solves_per_s"] = args.windows * world * nres / t_res
        out["end_to_end_over_device_resident"] = out["value"] / out["device_resident_solves_per_s"]
    # ---- RCCL: all-gather of the per-window result records of one batch, every id exactly once
    sms_all = []
    for si in range(nstream):
        solvers[si].restore_state()
        sm = (cv.capi.Summary * per[si])()
        cv.capi.check(lib.ctvio_solve(solvers[si]._h, args.iters, C.cast(sm, C.c_void_p)))
        sms_all += [s.as_dict() for s in sm]
    rec_local = cv.sharding.make_records(my_ids, sms_all)
    rec = cv.sharding.gather_records(rec_local, args.windows * world, device=coll_device) if dist is not None else None
    if rec is not None:
        ids = rec[:, 0]
        assert not np.isnan(ids).any() and sorted(ids.astype(int).tolist()) == list(range(args.windows * world)), "RCCL gather lost a window"
    if rank == 0:
        src = rec if rec is not None else rec_local
        out["solve_summary"] = {"windows": int(src.shape[0]), "iterations_mean": float(np.mean(src[:, 1])),
                                "terminations": sorted({cv.capi.TERMINATION.get(int(t), "?") for t in src[:, 2]}),
                                "line_search_reduced_steps": int(sum(s["num_line_search_reduced"] for s in sms_all)),
                                "gathered_with": ("single rank" if rec is None else "gloo all_gather (ranks share a device)" if shared_device
                                                  else "RCCL all_gather (GPU tensors)"),
                                "window_ids_gathered_once": bool(rec is None or sorted(rec[:, 0].astype(int).tolist()) == list(range(args.windows * world)))}
        # ---- roofline: one more device-resident solve of handle 0 with HIP events around every launch group on its stream;
        #      the other handles are idle, so a duration is the kernel's own (two streams sharing the chip stretch both)
        solver = solvers[0]
        solver.set_profiling(True)
        run_handle_steps(0, 1, True)
        solver.set_profiling(False)
        torch.cuda.synchronize()
        ms, n = solver.last_timing()
        names = cv.Solver.PHASES
        kmap = kernel_of_phase(args.precision)
        wl0 = keeps[0][1]
        w_ref = uniq[0]
        pk = MFMA_PEAK_TFLOPS["fp64" if args.precision == "fp64" else "fp32"]
        def load_json(name):
            f = os.path.join(ROOT, "profiles", name)
            try:
                return json.load(open(f)) if os.path.exists(f) else {}
            except Exception:
                return {}
        pmc = load_json("pmc_traffic.json")      # per-kernel FETCH_SIZE / WRITE_SIZE of the committed rocprofv3 passes
        # (counters of OTHER kernel sources say nothing about these kernels: no traffic figure then)
        pmc_fresh = bool(pmc) and pmc.get("_csrc_sha256") == csrc_sha256()
        if not pmc_fresh:
            pmc = {"_source": "profiles/pmc_traffic.json was collected with other kernel sources (its _csrc_sha256 differs): traffic not reported"}
        issue = load_json("pmc_issue.json")      # per-kernel SQ issue counters of the committed pass
        if issue.get("_csrc_sha256") != csrc_sha256():
            issue = {}

        def kernel_line(i):
            """roofline entry of launch group i: HBM-bound unless its algorithmic intensity is beyond the ridge (peak flops / peak bytes)"""
            nb = sum(algorithmic_bytes(w, names[i], fp_bytes) for w in wl0)
            nf = sum(algorithmic_flops(w, names[i]) for w in wl0)
            avg_s = 1e-3 * ms[i] / max(int(n[i]), 1)
            kname = kmap.get(names[i], names[i])
            tparts = [pmc.get(k, {}).get(str(per[0]), {}).get("traffic_bytes") for k in PMC_KERNELS.get(names[i], [])]
            traffic = sum(tparts) if tparts and all(t is not None for t in tparts) else None   # (committed rocprofv3 --pmc passes: see traffic_source)
            # Which roof?  Both fractions are computed; the label follows the LARGER one (the resource the kernel is closer to), and the
            # issue counters of the committed rocprofv3 pass (profiles/pmc_issue.json: tools/profile_round4.sh) ride along -- a kernel at a
            # quarter of either roof is bound by neither, and `limiter` says by what instead.
            hbm_frac = nb / avg_s / 1e9 / HBM_PEAK_GBS
            fl_frac = nf / avg_s / 1e12 / pk
            compute = fl_frac > hbm_frac
            ach = nf / avg_s / 1e12 if compute else nb / avg_s / 1e9
            peak = pk if compute else HBM_PEAK_GBS
            line = {"kernel": kname, "bound": "mfma" if compute else "hbm", "achieved": ach, "peak": peak,
                    "unit": "TFLOP/s" if compute else "GB/s", "frac": ach / peak, "traffic": traffic,
                    "traffic_over_algorithmic": (traffic / nb if traffic and nb else None), "avg_launch_us": 1e6 * avg_s,
                    "launches": int(n[i]), "windows_per_launch": per[0], "algorithmic_bytes_per_launch": nb,
                    "algorithmic_flops_per_launch": nf, "hbm_frac": hbm_frac, "fp64_frac": fl_frac,
                    "share_of_profiled_solve": float(ms[i] / max(sum(ms[:7]), 1e-12))}
            iss = [issue.get(k) for k in PMC_KERNELS.get(names[i], [])]
            if iss and all(x is not None for x in iss):
                tot = lambda key: sum(x.get(key, 0.0) for x in iss)
                wc = max(tot("SQ_WAVE_CYCLES"), 1.0)
                line["issue"] = {"valu_active_share_of_wave_cycles": tot("SQ_ACTIVE_INST_VALU") / wc, "parked_share": tot("SQ_WAIT_ANY") / wc,
                                 "issue_stall_share": tot("SQ_WAIT_INST_ANY") / wc, "valu_instructions": tot("SQ_INSTS_VALU"),
                                 "mfma_busy_cycles": tot("SQ_VALU_MFMA_BUSY_CYCLES"), "source": "profiles/pmc_issue.json (rocprofv3 --pmc, 2048 windows per launch)"}
                line["limiter"] = ("fp64 issue" if line["issue"]["valu_active_share_of_wave_cycles"] > 0.5 else
                                   "latency: waves parked on memory / LDS / barriers" if line["issue"]["parked_share"] > 0.5 else "mixed issue / latency")
            return line

        dom = max(range(6), key=lambda i: ms[i])            # named kernels only (0..5)
        out["roofline"] = kernel_line(dom)
        out["roofline"]["measured"] = "HIP events on the solver's stream around every launch, other handles idle"
        out["roofline"]["traffic_source"] = pmc.get("_source", "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the committed profile run, not of this run)")
        out["roofline_kernels"] = [kernel_line(i) for i in sorted(range(6), key=lambda i: -ms[i]) if n[i] > 0 and names[i] in kmap]
        out["roofline_mfma"] = mfma_line(kmap["k_schur_mfma"], wl0, 1e-3 * ms[4] / max(int(n[4]), 1), per[0])
        out["phase_ms_profiled_solve"] = {names[i]: float(ms[i]) for i in range(7)}   # handle 0 only
        # ---- parity of what was timed + CPU baseline: the oracle solves a sample of the same windows on one host core
        # ---- the reference's operating mode: ONE window per solve (one UpdateTrajectory per image, odometry_manager.cpp:268-277)
        if world == 1 and not args.quick and not resident_headline:
            with cv.Solver(device=local, host_threads=1) as s1:
                w1 = uniq[0].copy()
                keep1 = []
                c1 = (cv.capi.CWindow * 1)()
                c1[0] = cv.capi.to_cwindow(w1, keep1)
                o1 = (np.zeros((w1.K, 4)), np.zeros((w1.K, 3)), np.zeros((w1.F, 6)), np.zeros(max(w1.L, 1)), np.zeros(1))

                def one():
                    cv.capi.check(lib.ctvio_set_batch(s1._h, 1, C.cast(c1, C.c_void_p)))
                    cv.capi.check(lib.ctvio_solve(s1._h, args.iters, None))
                    cv.capi.check(lib.ctvio_get_batch_state(s1._h, *[cv.capi._p(a) for a in o1]))
                for _ in range(3):
                    one()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(40):
                    one()
                torch.cuda.synchronize()
                out["single_window_ms"] = 1e3 * (time.perf_counter() - t0) / 40
                cv.capi.check(lib.ctvio_set_batch(s1._h, 1, C.cast(c1, C.c_void_p)))   # back to the initial guess
                s1.snapshot_state()
                t0 = time.perf_counter()
                for _ in range(40):
                    s1.restore_state()
                    s1.solve_raw(args.iters)
                torch.cuda.synchronize()
                out["single_window_device_resident_ms"] = 1e3 * (time.perf_counter() - t0) / 40
            # ---- what the bitwise-reproducible mode costs (ctvio_options.deterministic; the default for <= 64 windows): 64 windows both ways
            det = {}
            for name, flag in (("deterministic", 1), ("throughput", 0)):
                with cv.Solver(device=local, deterministic=flag) as sd:
                    sd.set_windows([uniq[i % nuniq].copy() for i in range(64)])
                    sd.snapshot_state()
                    sd.solve_raw(args.iters)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        sd.restore_state()
                        sd.solve_raw(args.iters)
                    torch.cuda.synchronize()
                    det[name + "_ms_per_64_windows"] = 1e3 * (time.perf_counter() - t0) / 10
            det["cost_ratio"] = det["deterministic_ms_per_64_windows"] / det["throughput_ms_per_64_windows"]
            out["deterministic_mode"] = det
            # ---- BASELINE configs[3] as written is 64 windows over 8 GPUs = 8 per GPU: launch-latency territory, reported beside the headline
            out["small_batches"] = [small_batch(cv, lib, torch, C, np, uniq, nb_, args.iters, local) for nb_ in (8, 64)]
            ora = not args.no_cpu_baseline
            out["config3"] = side_config(cv, lib, torch, "config3", 1024, 16, args.iters, 2, 8, local, ora)
            out["config3"]["spline_eval"] = row_queries(cv, torch, np, 256, local)
            # (512 windows per launch: the one-workgroup-per-window kernels of this shape -- panel Cholesky at P = 571 -- need at least one
            #  window per CU, and the tile Schur kernel's 2 x 2 blocked form is chosen by tile count; 128 / 256 / 512 windows per launch
            #  measured 2.9 k / 3.5 k / 3.7 k solves/s)
            out["config5"] = side_config(cv, lib, torch, "config5", 512, 8, args.iters, 2, 8, local, ora, profile=True,
                                         label="30 KF / 1000 landmarks / 6000 IMU, SURVEY's recipe: landmark l anchored in frame l mod 8, tracked <= 8 frames -- "
                                               "frames 16..30 (knots >= 32 of 64) carry no visual factor")
            out["config5_spread"] = side_config(cv, lib, torch, "config5_spread", 512, 8, args.iters, 2, 4, local, ora, profile=True,
                                                label="the same sizes with landmark l anchored in frame l mod 28: visual factors all along the window")
            out["tumrs"] = side_config(cv, lib, torch, "tumrs", 2048, 16, args.iters, 2, 8, local, ora, profile=True,
                                       label="the reference's native operating point: 200 Hz IMU (10 samples per group), <= 150 features per frame")
            try:   # (a side measurement: it must not cost the run its record)
                out["mixed_batch"] = mixed_batch(cv, torch, args.config, 2048, 256, args.iters, local)
            except Exception as e:
                out["mixed_batch"] = {"error": repr(e)[:300]}
            wt = cv.synth.make_window("tumrs", seed=1000)
            out["tumrs"]["imu_lane_utilisation"] = wt.M / (64.0 * imu_groups(wt))   # one 64-lane pass per (segment, bias) group
            # ---- rounds 1-4 let the replicas of a distinct window share its caller buffers (the packer read 11 MB out of L3): once, beside the headline
            if not args.shared_caller_buffers:
                which[0] = "shared"
                steps(1)
                tsh = timed(3, False)
                which[0] = "own"
                out["end_to_end_shared_caller_buffers_solves_per_s"] = args.windows * 3 / tsh
            # ---- what an 8-rank run leaves one rank on the host: pack threads = cores / (streams x 8)
            ht8 = max(1, (os.cpu_count() or 8) // (nstream * 8))
            for sv in solvers:
                sv.close()
            solvers.clear()
            for si in range(nstream):
                solvers.append(cv.Solver(device=local, precision=args.precision, host_threads=ht8))
            steps(1)
            t8 = timed(3, False)
            out["host_share_of_an_8_rank_run"] = {"pack_threads_per_stream": ht8, "end_to_end_solves_per_s": args.windows * 3 / t8,
                                                  "end_to_end_over_device_resident": args.windows * 3 / t8 / out["device_resident_solves_per_s"]}
            # ---- the host side alone: validate + pack (host threads) + one H2D copy per handle, no solve -- what the packer sustains
            def pack_only(si, reps):
                for _ in range(reps):
                    cv.capi.check(lib.ctvio_set_batch(solvers[si]._h, per[si], C.cast(cbatches[si], C.c_void_p)))
            th = [threading.Thread(target=pack_only, args=(si, 1)) for si in range(nstream)]
            for t in th: t.start()
            for t in th: t.join()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=pack_only, args=(si, 4)) for si in range(nstream)]
            for t in th: t.start()
            for t in th: t.join()
            torch.cuda.synchronize()
            tp = time.perf_counter() - t0
            wbytes = sum(getattr(w_lab, a).nbytes for a in ("quat", "pos", "bias", "rho", "imu_t", "imu_gyro", "imu_acc", "imu_bias", "v_lm", "v_ti", "v_tj",
                                                             "v_rowi", "v_rowj", "v_pi", "v_pj", "bc_i", "bc_j", "bc_w", "pJ0", "pr0"))
            out["host_pack_only"] = {"handles": nstream, "pack_threads_per_handle": ht8, "windows_per_s": args.windows * 4 / tp,
                                     "caller_bytes_per_window": int(wbytes), "caller_GB_per_s": args.windows * 4 * wbytes / tp / 1e9,
                                     "over_device_resident_rate": args.windows * 4 / tp / out["device_resident_solves_per_s"],
                                     "usable_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()}
        out["parity"] = None
        out["cpu_baseline"] = None
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only: the scaling runs must not wait on a CPU loop
            import pyctvo
            nsamp = max(1, min(args.parity_sample, nuniq))
            ref = [uniq[i].copy() for i in range(nsamp)]
            t0 = time.perf_counter()
            for r in ref:
                pyctvo.OracleWindow(r).solve(args.iters)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": nsamp / dt, "unit": "solves/s", "cores": 1, "kind": "port",
                                   "sample": f"{nsamp} solves of {args.config} windows (seeds 1000..{1000 + nsamp - 1}), fp64 C oracle "
                                             f"(oracle/ctvo.c, gcc -O2), 1 thread, {dt:.1f} s; host has {os.cpu_count()} cores"}
            # the same windows as they came out of the timed path (handle 0 holds local windows 0.. = uniq[0..]): end-to-end solve
            run_handle_steps(0, 1, False)
            q, p, b, r, ld = outs[0]
            errs, k0, f0, l0 = [], 0, 0, 0
            for j in range(min(nsamp, per[0])):
                w = uniq[j].copy()
                w.quat[:] = q[k0:k0 + w.K]; w.pos[:] = p[k0:k0 + w.K]; w.bias[:] = b[f0:f0 + w.F]; w.rho[:] = r[l0:l0 + w.L]; w.ld = float(ld[j])
                errs.append(cv.rel_state_error(w, ref[j])["state"])
                k0 += w.K; f0 += w.F; l0 += w.L
            out["parity"] = {"max_rel_state_err": float(max(errs)), "median_rel_state_err": float(np.median(errs)), "windows": len(errs),
                             "tolerance": 1e-4, "reference": "fp64 C oracle, same Ceres settings", "pass": bool(max(errs) <= 1e-4)}
            if not args.quick:   # configs[3]: 64 windows, one per thread, all host cores
                out["cpu_baseline_all_cores"] = cpu_all_cores(args.config, args.iters, 1000, 256)
        print(compact_line(out, write_details(out, world)), flush=True)
    if dist is not None:
        dist.barrier()   # rank 0 prints after its extra measurements; nobody tears the communicator down under it
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
