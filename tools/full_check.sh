R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/full; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py --windows 1 --streams 1 --device-resident-only --steps 50 --no-cpu-baseline > $O/bench_w1.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/bench_w1.json').read().strip().splitlines()[-1]); print('single window', d['value'], d['ms_per_step'])"
python bench.py --config config5 --windows 128 --unique 8 --streams 1 --device-resident-only --steps 3 --no-cpu-baseline > $O/bench_c5.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/bench_c5.json').read().strip().splitlines()[-1]); print('config5 x128', d['value'], d['ms_per_step'])"
python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['device_resident_solves_per_s'], d['ms_per_step'], d['fast_mode_fp32'], d['parity'], d['cpu_baseline']['value'])"
