R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2h; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for C in config5 config3; do
rocprofv3 --kernel-trace --stats -d $O/kt_$C -o kt -- python $R/bench.py --no-cpu-baseline --config $C --streams 1 --windows 128 --unique 8 --steps 2 --warmup 1 --device-resident-only > $O/bench_$C.json 2> $O/kt_$C.err
python $R/tests/prof_summary.py stats $(find $O/kt_$C -name "*.db") > $O/kstats_$C.txt; find $O/kt_$C -name "*.db" -delete
head -12 $O/kstats_$C.txt
python -c "
import json; d=json.loads(open('$O/bench_$C.json').read().strip().splitlines()[-1]); print('$C', d['value'], d['ms_per_step'])"
done
