# A/B on one box: store-semantics assembly tail on / off (CTVIO_NO_STORE_PATH=1), 1 window and 2048 windows
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ab; mkdir -p $O; export TMPDIR=/tmp; cd $R
[ -n "$PYTEST_K" ] && timeout 900 python -m pytest tests -m gpu -q -x -k "$PYTEST_K" 2>&1 | tail -6
run() {
  env $2 timeout 300 python bench.py $3 --no-cpu-baseline > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    print("$1", "solves/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(x,3) for k,x in d["phase_ms_profiled_solve"].items()})
except Exception as e:
    print("$1", "FAILED", e, open("$O/$1.err").read()[-800:])
PY
}
for v in "X=1" "${AB_ENV:-CTVIO_STORE_PATH=0}"; do
  run w1_$v "$v" "--windows 1 --streams 1 --device-resident-only --steps 50"
  run w2048_$v "$v" "--windows 2048 --streams 1 --device-resident-only --steps 3 --warmup 1"
done
