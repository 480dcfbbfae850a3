R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ab2; mkdir -p $O; export TMPDIR=/tmp
for v in "" "CTVIO_CHOL_TILES=0"; do
  cd /tmp; env $v rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- env $v CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --streams 1 --device-resident-only --steps 1 --warmup 1 --windows 2048 > $O/bench.json 2> $O/err.txt
  cd $R; echo "== [$v]"; python tools/prof_summary.py stats $(find $O/kt -name "*.db") | head -9; find $O/kt -name "*.db" -delete
done
