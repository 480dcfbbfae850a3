# generic A/B of two environments on one box: 2048 windows, single stream, device resident (phase times from the profiled solve)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ab2; mkdir -p $O; cd $R
[ -n "$PYTEST_K" ] && env $B_ENV timeout 900 python -m pytest tests -m gpu -q -x -k "$PYTEST_K" 2>&1 | tail -4
for v in "$A_ENV" "$B_ENV" "$A_ENV" "$B_ENV"; do
  env $v timeout 300 python bench.py --windows 2048 --streams 1 --device-resident-only --steps 3 --warmup 1 --no-cpu-baseline > $O/x.json 2> $O/x.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/x.json").read().strip().splitlines()[-1])
    print("$v", "solves/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(x,3) for k,x in d["phase_ms_profiled_solve"].items()})
except Exception as e:
    print("$v", "FAILED", e, open("$O/x.err").read()[-800:])
PY
done
