"""CPU: bench.py's final stdout line stays under 4 KB (the driver keeps an 8 KB tail of stdout + stderr; round 5's 20 KB line left
BENCH_r05.json.parsed null) and carries every key of the driver's contract -- built from a recorded full run
(profiles/r05_bench_n1.json: everything bench.py measured on the GPU box in round 5)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")


def _recorded():
    with open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_compact_line_is_under_4k_and_complete():
    import bench
    full = _recorded()
    assert len(json.dumps(full)) > 8192                      # the record that did not survive the driver's tail
    s = bench.compact_line(full, "gpurun_out/bench_details_n1.json")
    assert len(s) < bench.COMPACT_LIMIT == 4096 and "\n" not in s
    line = json.loads(s)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == float(f"{full['value']:.6g}") and line["config"]["workload"].startswith("config2")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - line["roofline"]["achieved"] / line["roofline"]["peak"]) < 1e-5
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["parity"]["pass"] is True and line["details_file"].endswith(".json")
    assert "device_resident_solves_per_s" in line


def test_compact_line_of_a_multi_rank_run_and_of_an_oversized_record():
    import bench
    full = _recorded()
    full["n_gpus"] = 8
    full["per_rank_solves_per_s"] = [46000.123456789 + i for i in range(8)]
    full["small_batch_latency"] = {"end_to_end_ms_per_step_by_rank": [3.123456789] * 8}
    full["cpu_baseline"] = None; full["parity"] = None       # rank 0 of an N > 1 run measures neither
    full["config"]["workload"] = full["config"]["workload"] + " " + "x" * 1500     # something grew: the optional extras go first
    s = bench.compact_line(full, None)
    assert len(s) < 4096
    line = json.loads(s)
    assert line["n_gpus"] == 8 and line["cpu_baseline"] is None and line["parity"] is None and "roofline" in line


def test_traffic_is_tied_to_the_kernel_sources():
    """profiles/pmc_traffic.json names the kernel sources its counters were collected with; bench.py prints `roofline.traffic` only when that hash
    is the one of ctrl-vio_amd/csrc today."""
    import bench
    h = bench.csrc_sha256()
    assert len(h) == 64 and h == bench.csrc_sha256()
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert "_source" in pmc      # (whether the hashes agree depends on the last profile run: the bench prints null when they do not)
