cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt1 -o kt -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 2 --warmup 1 > /dev/null 2>&1
cd $R; python tests/prof_summary.py stats $(find gpurun_out/kt1 -name "*.db") | head -24; find gpurun_out/kt1 -name "*.db" -delete
