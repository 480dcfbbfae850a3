set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -m pytest tests -m gpu -q -x -k "linearize or lm_step or product_parity or ragged or edge or golden or large_batch" 2>&1 | tail -5
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --streams 1 --windows 1024 --steps 2 --warmup 1 --device-resident-only > $O/bench_1s.json 2> $O/kt.err
python $R/tools/prof_summary.py stats $(find $O/kt -name "*.db") > $O/kstats.txt; find $O/kt -name "*.db" -delete
head -16 $O/kstats.txt
python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['device_resident_solves_per_s'], d['phase_ms_profiled_solve'])"
