# Round-6 inner loop on the GPU box: [pytest selection] then the bench (compact line + the details file's highlights).
#   usage: bash tools/r6_check.sh "<pytest -k expression, empty for the whole GPU suite, or 'none'>" [skipbench|quick|full]
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp; cd $R
if [ "$1" = "none" ]; then :
elif [ -n "$1" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$1" 2>&1 | tail -25 | tee $O/pytest.txt
else timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee $O/pytest.txt; fi
if [ "$2" = "skipbench" ]; then exit 0; fi
if [ "$2" = "full" ]; then timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err
elif [ "$2" = "quick" ]; then timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --quick > $O/bench.json 2> $O/bench.err
else timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; fi
tail -3 $O/bench.err
python - <<PY
import json
l=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('LINE bytes', len(json.dumps(l))); print(json.dumps(l))
d=json.load(open(l['details_file']))
print('value', d['value'], 'resident', d.get('device_resident_solves_per_s'))
print('phases', d['phase_ms_profiled_solve'])
for k in d.get('roofline_kernels', []): print('  ', k['kernel'], round(k['avg_launch_us'],1), 'us frac', round(k['frac'],3), k['bound'])
print('single', d.get('single_window_ms'), d.get('single_window_device_resident_ms'), 'small', [(b['windows'], round(b['device_resident_ms'],2)) for b in d.get('small_batches',[])])
for k in ('config3','config5','config5_spread','tumrs'):
    c=d.get(k,{}); print(k, c.get('solves_per_s'), c.get('max_rel_state_err'), c.get('phase_ms_profiled_solve'))
print('mfma', d.get('roofline_mfma'))
print('parity', d.get('parity'), 'cpu', d.get('cpu_baseline'))
PY
