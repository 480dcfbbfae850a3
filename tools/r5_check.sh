# Round-5 inner loop on the GPU box: [pytest selection] then the quick bench line (side configs without the oracle).
#   usage: bash tools/r5_check.sh "<pytest -k expression or empty for the whole GPU suite>" [skipbench]
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5; mkdir -p $O; export TMPDIR=/tmp; cd $R
if [ -n "$1" ]; then timeout 1500 python -m pytest tests -m gpu -q -x -k "$1" 2>&1 | tail -25 | tee $O/pytest.txt
else timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee $O/pytest.txt; fi
if [ -z "$2" ]; then
timeout 900 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'resident', d.get('device_resident_solves_per_s'), 'shared', d.get('end_to_end_shared_caller_buffers_solves_per_s'))
print('phases', d['phase_ms_profiled_solve'])
print('single', d.get('single_window_ms'), d.get('single_window_device_resident_ms'), 'small', [(b['windows'], round(b['device_resident_ms'],2)) for b in d.get('small_batches',[])])
for k in ('config3','config5','config5_spread','tumrs'):
    c=d.get(k,{}); print(k, c.get('solves_per_s'), c.get('phase_ms_profiled_solve'), c.get('schur_plus_cholesky_ms_per_solve'))
print('mfma', d['roofline_mfma'])
PY
fi
