# Round-3 check 1 (GPU box): full GPU test suite, then A/B of the Cholesky variants (panel kernel / register tiles 16x7 / 8x14)
# for one window alone and for 2048 windows per launch.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c1; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest.txt; cat $O/pytest.txt
for v in 0 1 2; do
  CTVIO_CHOL_TILES=$v timeout 300 python bench.py --windows 1 --streams 1 --device-resident-only --steps 50 --no-cpu-baseline > $O/w1_t$v.json 2> $O/w1_t$v.err
  CTVIO_CHOL_TILES=$v timeout 300 python bench.py --windows 2048 --streams 1 --device-resident-only --steps 3 --warmup 1 --no-cpu-baseline > $O/w2048_t$v.json 2> $O/w2048_t$v.err
  python - <<PY
import json
for n in ("w1","w2048"):
    try:
        d=json.loads(open("$O/%s_t$v.json"%n).read().strip().splitlines()[-1])
        print("tiles=$v", n, "solves/s %.1f ms/step %.3f" % (d["value"], d["ms_per_step"]), {k: round(x,3) for k,x in d["phase_ms_profiled_solve"].items()})
    except Exception as e:
        print("tiles=$v", n, "FAILED", e, open("$O/%s_t$v.err"%n).read()[-600:])
PY
done
