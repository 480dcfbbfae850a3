# end-to-end rate with per-window caller buffers: handles x GPU slots x pack threads
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5s; mkdir -p $O; cd $R
for cfg in "4 2 16" "4 2 8" "4 3 16" "6 2 16" "6 3 8" "8 4 8" "8 3 4"; do
  set -- $cfg
  python bench.py --quick --no-cpu-baseline --steps 6 --warmup 2 --streams $1 --gpu-slots $2 --host-threads $3 > $O/b_$1_$2_$3.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('$O/b_$1_$2_$3.json').read().strip().splitlines()[-1])
print('streams $1 slots $2 threads $3: e2e %.0f resident %.0f ratio %.3f' % (d['value'], d['device_resident_solves_per_s'], d['end_to_end_over_device_resident']))
PY
done
