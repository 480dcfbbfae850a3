# Round 6: the small-batch path (<= 128 windows): tests that run it, then the single-window / 8 / 64-window timings.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "not headline and not config5 and not large_batch and not multirank and not eight_ranks and not two_ranks" 2>&1 | tail -6
timeout 300 python bench.py --no-cpu-baseline --quick --steps 60 --warmup 5 --device-resident-only --streams 1 --windows 1 > $O/small1.json 2> $O/small.err
timeout 300 python bench.py --no-cpu-baseline --quick --steps 40 --warmup 5 --device-resident-only --streams 1 --windows 8 > $O/small8.json 2>> $O/small.err
timeout 300 python bench.py --no-cpu-baseline --quick --steps 20 --warmup 3 --device-resident-only --streams 1 --windows 64 > $O/small64.json 2>> $O/small.err
python - <<PY
import json
for f in ('small1', 'small8', 'small64'):
    l = json.loads(open('$O/' + f + '.json').read().strip().splitlines()[-1])
    print(f, 'ms per solve (device resident)', l['ms_per_step'])
PY
