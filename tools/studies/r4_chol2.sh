# round 4: the shortened pivot chain of the Cholesky kernels (reciprocal on the chain, columns unscaled while in use)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4chol2; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "not multirank and not slide and not adaptor" 2>&1 | tail -5
python bench.py --no-cpu-baseline --quick --streams 1 --windows 2048 --steps 3 --warmup 1 --device-resident-only > $O/b.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('2048 windows', round(d['value']), round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['phase_ms_profiled_solve'].items()})"
python bench.py --no-cpu-baseline --quick --streams 1 --windows 1 --steps 40 --warmup 4 --device-resident-only > $O/w1.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/w1.json').read().strip().splitlines()[-1]); print('single window ms', d['ms_per_step'])"
python bench.py --config config5 --windows 512 --unique 8 --no-cpu-baseline --quick --streams 1 --steps 2 --warmup 1 --device-resident-only > $O/c5.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/c5.json').read().strip().splitlines()[-1]); print('config5 x512', round(d['value'],1), {k: round(x,2) for k,x in d['phase_ms_profiled_solve'].items()})"
CTVIO_DEBUG_STAMPS=1 python bench.py --no-cpu-baseline --streams 1 --windows 2048 --steps 1 --warmup 1 --device-resident-only 2>&1 >/dev/null | grep "cholesky" | tail -1
