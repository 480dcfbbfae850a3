# round 4: config 5 (K = 64, P = 571) -- the 2 x 2 blocked tile Schur kernel against the one-tile kernel, and the batch size swept
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c5; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "blocked_tile or config5 or large_batch or lm_step" 2>&1 | tail -4
for t2 in 0 1; do for n in 128 256 512; do
  CTVIO_SCHUR_TILE2=$t2 python bench.py --config config5 --windows $n --unique 8 --no-cpu-baseline --quick --streams 1 --steps 2 --warmup 1 --device-resident-only > $O/c5_${t2}_$n.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/c5_${t2}_$n.json').read().strip().splitlines()[-1]); print('config5 tile2=$t2 windows $n:', round(d['value'], 1), 'solves/s', {k: round(x, 2) for k, x in d['phase_ms_profiled_solve'].items()})"
done; done
