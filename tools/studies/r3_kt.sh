# per-kernel times of the 1 x 2048 device-resident configuration (rocprofv3 --kernel-trace --stats), kernels kept apart
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3kt; rm -rf $O; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
env CTVIO_SPLIT_LINEARIZE=1 $EXTRA_ENV timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --windows 2048 --streams 1 --device-resident-only --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
cd $R
DB=$(find $O/kt -name "*.db" | head -1)
python tools/prof_summary.py stats $DB | tee $O/kernel_stats.txt | head -${LINES_OUT:-24}
