# Round 6: the chain Cholesky against round 5's (A/B): SIMD placement probe, the Cholesky-sensitive tests, stamps, device-resident rates both ways.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "lm_step or schur_step_equals or flow_cholesky" 2>&1 | tail -8
timeout 1200 python -m pytest tests -m gpu -q -x -k "product_parity or large_batch or headline or deterministic or golden or slide or fixed_unknowns or marginalize" 2>&1 | tail -8
for ct in 1 3; do
  CTVIO_CHOL_TILES=$ct timeout 300 python bench.py --no-cpu-baseline --quick --steps 4 --warmup 1 --device-resident-only --streams 1 --windows 2048 > $O/ab_$ct.json 2> $O/ab_$ct.err
  CTVIO_CHOL_TILES=$ct timeout 300 python bench.py --no-cpu-baseline --quick --steps 40 --warmup 3 --device-resident-only --streams 1 --windows 1 > $O/ab1_$ct.json 2>> $O/ab_$ct.err
  python - <<PY
import json
for f in ('$O/ab_$ct.json', '$O/ab1_$ct.json'):
    l = json.loads(open(f).read().strip().splitlines()[-1]); d = json.load(open(l['details_file']))
    print('CHOL_TILES=$ct', l['config']['windows_per_gpu_per_step'], 'windows: value', l['value'], 'ms/step', l['ms_per_step'], 'chol ms', d['phase_ms_profiled_solve']['k_cholesky_solve'])
PY
done
CTVIO_DEBUG_STAMPS=1 timeout 120 python bench.py --no-cpu-baseline --quick --steps 1 --warmup 0 --device-resident-only --streams 1 --windows 1 2>&1 | grep -a "cholesky clock64" | tail -2
