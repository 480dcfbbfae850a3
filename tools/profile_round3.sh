# Round-3 profile on the GPU box (outputs under gpurun_out/prof3; the summaries are copied to profiles/ by hand):
#   kernel-trace stats of a single-stream 2048-window solve and of the default bench command, PMC HBM traffic (separate FETCH_SIZE /
#   WRITE_SIZE passes, as MI355X_MICROARCH.md prescribes), MFMA busy cycles of the fp64 matrix-core kernels, single-window trace.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof3; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
B1="env CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --steps 1 --warmup 1 --device-resident-only --streams 1 --windows 2048"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B1 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma -o m -- $B1 > /dev/null 2> $O/pmc_mfma.err
cd $R
rm -f $O/pmc_traffic.json
python tools/prof_summary.py pmc 2048 $O/pmc_traffic.json $(find $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv") > $O/pmc_table_2048.txt; head -32 $O/pmc_table_2048.txt
python tools/prof_summary.py counters $O/pmc_mfma.json $(find $O/pmc_mfma -name "*counter_collection.csv") > $O/pmc_mfma_table.txt; head -30 $O/pmc_mfma_table.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt -- $B1 > /dev/null 2>&1
cd $R; python tools/prof_summary.py stats $(find $O/kt1 -name "*.db") > $O/kernel_stats_1x2048.txt; head -24 $O/kernel_stats_1x2048.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ktw1 -o kt -- python $R/bench.py --no-cpu-baseline --quick --streams 1 --windows 1 --steps 20 --warmup 2 --device-resident-only > /dev/null 2>&1
cd $R; python tools/prof_summary.py stats $(find $O/ktw1 -name "*.db") > $O/kernel_stats_single_window.txt; head -20 $O/kernel_stats_single_window.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --quick --steps 3 --warmup 1 > $O/kt_bench.json 2> $O/kt.err
cd $R; python tools/prof_summary.py stats $(find $O/kt -name "*.db") > $O/kernel_stats_default_4x2048.txt
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.err
find $O -name "*.db" -delete; find $O -name "*.csv" -size +4M -delete
