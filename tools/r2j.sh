R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2j; mkdir -p $O; cd $R
for S in 1 2 3 4 6; do
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams $S > $O/b_$S.json 2> $O/b_$S.err
python -c "
import json; d=json.loads(open('$O/b_$S.json').read().strip().splitlines()[-1]); print('streams', $S, 'e2e', round(d['value']), 'resident', round(d['device_resident_solves_per_s']), d['config']['pack_threads_per_stream'])"
done
nproc; lscpu | grep -i "model name\|^CPU(s)"
