# round 4 quick loop: the whole GPU suite, a single-stream 2048-window kernel trace, the default bench line (quick)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
cd /tmp
B1="env CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --steps 1 --warmup 1 --device-resident-only --streams 1 --windows 2048"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt -- $B1 > $O/bench_1s.json 2> $O/kt.err
cd $R; python tools/prof_summary.py stats $(find $O/kt1 -name "*.db") > $O/kernel_stats_1x2048.txt; head -14 $O/kernel_stats_1x2048.txt
find $O -name "*.db" -delete
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --quick > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['device_resident_solves_per_s'])"
