# quick GPU check: parity subset + device-resident timings (1 window, 2048 windows)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "${PYTEST_K:-product_parity or lm_step or linearize or edge or ragged or config5 or large_batch}" 2>&1 | tail -8
run() {
  env $2 timeout 300 python bench.py $3 --no-cpu-baseline > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    print("$1", "solves/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), "dev-res", d.get("device_resident_solves_per_s"), {k: round(x,3) for k,x in d["phase_ms_profiled_solve"].items()})
except Exception as e:
    print("$1", "FAILED", e, open("$O/$1.err").read()[-800:])
PY
}
run w1 "X=1" "--windows 1 --streams 1 --device-resident-only --steps 50"
run w2048 "X=1" "--windows 2048 --streams 1 --device-resident-only --steps 3 --warmup 1"
[ -n "$E2E" ] && run e2e "X=1" "--steps 6 --warmup 1"
true
