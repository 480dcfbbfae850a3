# round 4: the whole GPU suite on the cleaned-up sources, then the small-batch A/B of k_step_finish (16 waves with spills vs 8 waves)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4verify; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
for v in 16 8; do
  for n in 1 8 64; do
    CTVIO_STEP_WAVES=$v python bench.py --no-cpu-baseline --quick --streams 1 --windows $n --steps 40 --warmup 4 --device-resident-only > $O/w${n}_$v.json 2>/dev/null
    python -c "
import json; d=json.loads(open('$O/w${n}_$v.json').read().strip().splitlines()[-1]); print('step waves $v, $n windows: ms per solve', round(d['ms_per_step'], 4))"
  done
done
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --quick > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['device_resident_solves_per_s'], d['phase_ms_profiled_solve'])"
