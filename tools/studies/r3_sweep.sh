# end-to-end rate against the number of solver handles / concurrent solves (one GPU)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3sw; mkdir -p $O; cd $R
for cfg in "4 2" "4 3" "4 4" "6 3" "8 4" "8 8"; do
  set -- $cfg
  timeout 300 python bench.py --quick --no-cpu-baseline --steps 6 --warmup 1 --streams $1 --gpu-slots $2 --windows ${WINDOWS:-8192} > $O/s$1_g$2.json 2> $O/s$1_g$2.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/s$1_g$2.json").read().strip().splitlines()[-1])
    print("streams $1 slots $2: e2e %.0f solves/s, device-resident %.0f" % (d["value"], d["device_resident_solves_per_s"]))
except Exception as e:
    print("streams $1 slots $2 FAILED", e, open("$O/s$1_g$2.err").read()[-400:])
PY
done
