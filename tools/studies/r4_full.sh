# round 4: the whole GPU suite + the default bench line (all side measurements)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4full; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $O/pytest.txt; cat $O/pytest.txt
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; tail -5 $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['device_resident_solves_per_s'], d['ms_per_step'])
print('single', d.get('single_window_ms'), d.get('single_window_device_resident_ms'))
print('small', d.get('small_batches'))
print('c3', {k: v for k, v in d['config3'].items() if k != 'spline_eval'}); print('rows', d['config3'].get('spline_eval'))
print('c5', d['config5'])
print('tumrs', d['tumrs'])
print('roofline', d['roofline'])
print('cpu', d['cpu_baseline'], d.get('cpu_baseline_all_cores'))
print('parity', d['parity'])
print('host8', d.get('host_share_of_an_8_rank_run'))
"
