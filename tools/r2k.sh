R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2k; mkdir -p $O; cd $R
for CFG in "4 2 4096" "4 3 4096" "6 2 4096" "6 3 4096" "8 2 4096" "4 2 8192" "8 2 8192"; do
set -- $CFG
python bench.py --steps 6 --warmup 1 --no-cpu-baseline --streams $1 --gpu-slots $2 --windows $3 > $O/b_$1_$2.json 2> $O/b_$1_$2.err
python -c "
import json; d=json.loads(open('$O/b_$1_$2.json').read().strip().splitlines()[-1]); print('streams', $1, 'slots', $2, 'win', $3, 'e2e', round(d['value']), 'resident', round(d['device_resident_solves_per_s']))"
done
