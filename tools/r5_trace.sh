# kernel traces of the shapes that matter: config 2 (1 x 2048, single stream), config 5 x 512, tumrs x 2048 -> gpurun_out/r5t/*.txt
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5t; mkdir -p $O; export TMPDIR=/tmp
trace() {  # name, bench args...
  name=$1; shift
  cd /tmp; rocprofv3 --kernel-trace --stats -d $O/kt_$name -o kt -- env CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --streams 1 --device-resident-only "$@" > $O/bench_$name.json 2> $O/err_$name.txt
  cd $R; python tools/prof_summary.py stats $(find $O/kt_$name -name "*.db") > $O/kernel_stats_$name.txt; find $O/kt_$name -name "*.db" -delete; head -22 $O/kernel_stats_$name.txt
}
trace 1x2048 --steps 1 --warmup 1 --windows 2048
trace config5_x512 --config config5 --windows 512 --unique 8 --steps 2 --warmup 1
[ -n "$1" ] && trace tumrs_x2048 --config tumrs --windows 2048 --unique 16 --steps 1 --warmup 1
[ -n "$1" ] && trace config5_spread_x512 --config config5_spread --windows 512 --unique 8 --steps 2 --warmup 1
true
