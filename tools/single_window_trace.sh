# Kernel trace of the PRODUCTION single-window path (merged linearisation: k_pre_linearize + k_linearize_f64; the profile script's single-window
# trace sets CTVIO_SPLIT_LINEARIZE=1 so that the IMU and visual evaluations get their own lines).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof6; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt_single_merged -o kt -- python $R/bench.py --no-cpu-baseline --quick --streams 1 --device-resident-only --windows 1 --steps 20 --warmup 2 > $O/bench_single_merged.json 2> $O/err_single_merged.txt
cd $R; python tools/prof_summary.py stats $(find $O/kt_single_merged -name "*.db") > $O/kernel_stats_single_window_merged.txt; find $O/kt_single_merged -name "*.db" -delete; head -16 $O/kernel_stats_single_window_merged.txt
