set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -m pytest tests -m gpu -q -k "rccl or graft" 2>&1 | tail -3
python bench.py --steps 4 --warmup 1 > $O/bench_e2e.json 2> $O/bench_e2e.err; tail -c 600 $O/bench_e2e.err; cat $O/bench_e2e.json
python bench.py --steps 4 --warmup 1 --precision fp32 --no-cpu-baseline > $O/bench_e2e_mixed.json 2> $O/bench_e2e_mixed.err; tail -c 300 $O/bench_e2e_mixed.err; cat $O/bench_e2e_mixed.json
