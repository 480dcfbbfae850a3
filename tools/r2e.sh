R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2e; mkdir -p $O; cd $R
for S in 1 2 4 8; do for W in 2048 4096; do
python bench.py --no-cpu-baseline --steps 6 --warmup 1 --streams $S --windows $W > $O/b_${S}_${W}.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/b_${S}_${W}.json').read().strip().splitlines()[-1]); r=d['roofline']; print('streams $S windows $W: e2e %.0f resident %.0f ratio %.3f dom %s %.0fus frac %.3f'%(d['value'], d['device_resident_solves_per_s'], d['end_to_end_over_device_resident'], r['kernel'], r['avg_launch_us'], r['frac']))"
done; done
