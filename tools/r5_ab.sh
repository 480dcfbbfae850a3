# A/B of whole-library builds (compiler flags): kernel trace of one 2048-window solve per build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ab; mkdir -p $O; export TMPDIR=/tmp
for lib in "$@"; do
  cd /tmp; CTVIO_LIB_PATH=$R/ctrl-vio_amd/$lib rocprofv3 --kernel-trace --stats -d $O/kt_$lib -o kt -- env CTVIO_LIB_PATH=$R/ctrl-vio_amd/$lib CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --streams 1 --device-resident-only --steps 1 --warmup 1 --windows 2048 > $O/bench_$lib.json 2> $O/err_$lib.txt
  cd $R; echo "== $lib"; python tools/prof_summary.py stats $(find $O/kt_$lib -name "*.db") | head -12; find $O/kt_$lib -name "*.db" -delete
done
