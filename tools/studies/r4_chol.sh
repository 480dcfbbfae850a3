# round 4: blocked diagonal tile of k_cholesky_tiles -- parity subset, then A/B against the unblocked form (CTVIO_CHOL_TILES=3)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4chol; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "lm_step or product_parity or ragged or edge or golden or large_batch or mixed_batch or deterministic or iterates or slide" 2>&1 | tail -5
for v in 1 3; do
  CTVIO_CHOL_TILES=$v python bench.py --no-cpu-baseline --quick --streams 1 --windows 2048 --steps 3 --warmup 1 --device-resident-only > $O/b_$v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/b_$v.json').read().strip().splitlines()[-1]); print('chol variant $v: 2048 windows', d['value'], d['ms_per_step'], d['phase_ms_profiled_solve'])"
  CTVIO_CHOL_TILES=$v python bench.py --no-cpu-baseline --quick --streams 1 --windows 1 --steps 30 --warmup 3 --device-resident-only > $O/w1_$v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/w1_$v.json').read().strip().splitlines()[-1]); print('chol variant $v: single window ms', d['ms_per_step'])"
done
CTVIO_DEBUG_STAMPS=1 python bench.py --no-cpu-baseline --streams 1 --windows 2048 --steps 1 --warmup 1 --device-resident-only 2>&1 >/dev/null | grep "cholesky" | tail -1
