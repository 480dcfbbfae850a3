# default bench line as the driver runs it (N = 1), timed
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O; cd $R
[ -n "$PYTEST_K" ] && timeout 900 python -m pytest tests -m gpu -q -x -k "$PYTEST_K" 2>&1 | tail -6
T0=$(date +%s.%N); timeout 900 python bench.py $BENCH_ARGS > $O/bench.json 2> $O/bench.err; T1=$(date +%s.%N); echo "bench wall $(echo "$T1 - $T0" | bc) s"; tail -c 600 $O/bench.err | tail -5
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
keep={k:d.get(k) for k in ("value","ms_per_step","device_resident_solves_per_s","end_to_end_over_device_resident","single_window_ms","single_window_device_resident_ms","config3","config5","tumrs","host_share_of_an_8_rank_run","parity","cpu_baseline","roofline","roofline_mfma")}
print(json.dumps(keep, indent=1)[:6000])
PY
