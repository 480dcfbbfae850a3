# Round-4 profile on the GPU box (outputs under gpurun_out/prof4; the summaries are copied to profiles/ by hand):
#   kernel-trace stats of a single-stream 2048-window solve and of the default bench command, PMC HBM traffic (separate FETCH_SIZE /
#   WRITE_SIZE passes, as MI355X_MICROARCH.md prescribes), SQ issue counters (wave cycles, parked / stalled / active shares, VALU
#   instructions, MFMA busy cycles) in a pass of their own, single-window and config-5 traces.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof4; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
B1="env CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --steps 1 --warmup 1 --device-resident-only --streams 1 --windows 2048"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B1 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_issue -o m -- $B1 > /dev/null 2> $O/pmc_issue.err
cd $R
rm -f $O/pmc_traffic.json $O/pmc_issue.json
python tools/prof_summary.py pmc 2048 $O/pmc_traffic.json $(find $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv") > $O/pmc_traffic_table_2048.txt; head -32 $O/pmc_traffic_table_2048.txt
python tools/prof_summary.py counters $O/pmc_issue.json $(find $O/pmc_issue -name "*counter_collection.csv") > $O/pmc_issue_table_2048.txt; head -30 $O/pmc_issue_table_2048.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt -- $B1 > /dev/null 2>&1
cd $R; python tools/prof_summary.py stats $(find $O/kt1 -name "*.db") > $O/kernel_stats_1x2048.txt; head -24 $O/kernel_stats_1x2048.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ktw1 -o kt -- python $R/bench.py --no-cpu-baseline --quick --streams 1 --windows 1 --steps 20 --warmup 2 --device-resident-only > /dev/null 2>&1
cd $R; python tools/prof_summary.py stats $(find $O/ktw1 -name "*.db") > $O/kernel_stats_single_window.txt; head -20 $O/kernel_stats_single_window.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/ktc5 -o kt -- env CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --config config5 --windows 512 --unique 8 --no-cpu-baseline --quick --streams 1 --steps 2 --warmup 1 --device-resident-only > $O/bench_config5_x512.json 2>/dev/null
cd $R; python tools/prof_summary.py stats $(find $O/ktc5 -name "*.db") > $O/kernel_stats_config5_x512.txt; head -20 $O/kernel_stats_config5_x512.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --quick --steps 3 --warmup 1 > $O/kt_bench.json 2> $O/kt.err
cd $R; python tools/prof_summary.py stats $(find $O/kt -name "*.db") > $O/kernel_stats_default_4x2048.txt
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.err
find $O -name "*.db" -delete; find $O -name "*.csv" -size +4M -delete
