# Round-3 check (GPU box): GPU tests, then single-window / 2048-window device-resident timings and a short end-to-end bench.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c2; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $O/pytest.txt; cat $O/pytest.txt
run() {  # name, env, args
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
run w2048_t0 "CTVIO_CHOL_TILES=0" "--windows 2048 --streams 1 --device-resident-only --steps 3 --warmup 1"
run w2048_t2 "CTVIO_CHOL_TILES=2" "--windows 2048 --streams 1 --device-resident-only --steps 3 --warmup 1"
run e2e_t0 "CTVIO_CHOL_TILES=0" "--steps 6 --warmup 1"
run e2e_t2 "CTVIO_CHOL_TILES=2" "--steps 6 --warmup 1"
