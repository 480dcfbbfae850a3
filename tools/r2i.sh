set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -m pytest tests -m gpu -q -x -k "config5 or product_parity or golden or ragged or large_batch" 2>&1 | tail -5
cd /tmp
for C in config5 config2; do
W=128; [ $C = config2 ] && W=1024
rocprofv3 --kernel-trace --stats -d $O/kt_$C -o kt -- python $R/bench.py --no-cpu-baseline --config $C --streams 1 --windows $W --unique 8 --steps 2 --warmup 1 --device-resident-only > $O/bench_$C.json 2> $O/kt_$C.err
python $R/tests/prof_summary.py stats $(find $O/kt_$C -name "*.db") > $O/kstats_$C.txt; find $O/kt_$C -name "*.db" -delete
head -10 $O/kstats_$C.txt
python -c "
import json; d=json.loads(open('$O/bench_$C.json').read().strip().splitlines()[-1]); print('$C', d['value'], d['ms_per_step'])"
done
