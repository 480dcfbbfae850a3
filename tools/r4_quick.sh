# round 4 quick loop: the parity subset, a single-stream 2048-window kernel trace, clock stamps, the default bench line
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "linearize or lm_step or product_parity or ragged or edge or golden or large_batch or mixed_batch or spline or cauchy or deterministic" 2>&1 | tail -5
cd /tmp
B1="env CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --steps 1 --warmup 1 --device-resident-only --streams 1 --windows 2048"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt -- $B1 > $O/bench_1s.json 2> $O/kt.err
cd $R; python tools/prof_summary.py stats $(find $O/kt1 -name "*.db") > $O/kernel_stats_1x2048.txt; head -14 $O/kernel_stats_1x2048.txt
find $O -name "*.db" -delete
CTVIO_DEBUG_STAMPS=1 python bench.py --no-cpu-baseline --streams 1 --windows 2048 --steps 1 --warmup 1 --device-resident-only 2>&1 >/dev/null | grep "ctvio\]" | tail -2 > $O/stamps.txt; cat $O/stamps.txt
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['device_resident_solves_per_s'], d.get('single_window_ms'))"
