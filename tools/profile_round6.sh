# Round-6 profile on the GPU box (outputs under gpurun_out/prof6; the summaries are copied to profiles/ by hand):
#   PMC HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, as MI355X_MICROARCH.md prescribes), SQ issue counters in a pass of their own,
#   kernel-trace stats of a single-stream 2048-window solve (config 2), of the default bench command, of one window, of config 5 / config 5
#   spread x 512 and of the native TUM-RSVI shape x 2048, and the full bench line.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof6; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
B1="env CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --steps 1 --warmup 1 --device-resident-only --streams 1 --windows 2048"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B1 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_issue -o m -- $B1 > /dev/null 2> $O/pmc_issue.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/pmc_lds -o l -- $B1 > /dev/null 2> $O/pmc_lds.err
cd $R
rm -f $O/pmc_traffic.json $O/pmc_issue.json $O/pmc_lds.json
python tools/prof_summary.py pmc 2048 $O/pmc_traffic.json $(find $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv") > $O/pmc_traffic_table_2048.txt; head -32 $O/pmc_traffic_table_2048.txt
python tools/prof_summary.py counters $O/pmc_issue.json $(find $O/pmc_issue -name "*counter_collection.csv") > $O/pmc_issue_table_2048.txt; head -30 $O/pmc_issue_table_2048.txt
python tools/prof_summary.py counters $O/pmc_lds.json $(find $O/pmc_lds -name "*counter_collection.csv") > $O/pmc_lds_table_2048.txt; head -30 $O/pmc_lds_table_2048.txt
trace() {  # name, bench args...
  name=$1; shift
  cd /tmp; rocprofv3 --kernel-trace --stats -d $O/kt_$name -o kt -- env CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --streams 1 --device-resident-only "$@" > $O/bench_$name.json 2> $O/err_$name.txt
  cd $R; python tools/prof_summary.py stats $(find $O/kt_$name -name "*.db") > $O/kernel_stats_$name.txt; find $O/kt_$name -name "*.db" -delete; head -14 $O/kernel_stats_$name.txt
}
trace 1x2048 --steps 1 --warmup 1 --windows 2048
trace single_window --windows 1 --steps 20 --warmup 2
trace config5_x512 --config config5 --windows 512 --unique 8 --steps 2 --warmup 1
trace config5_spread_x512 --config config5_spread --windows 512 --unique 8 --steps 2 --warmup 1
trace tumrs_x2048 --config tumrs --windows 2048 --unique 16 --steps 1 --warmup 1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --quick --steps 3 --warmup 1 > $O/kt_bench.json 2> $O/kt.err
cd $R; python tools/prof_summary.py stats $(find $O/kt -name "*.db") > $O/kernel_stats_default_4x2048.txt
cp $O/pmc_traffic.json $O/pmc_issue.json $O/pmc_lds.json $R/profiles/   # (the bench record below prints roofline.traffic from THIS run's counters)
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.err; cp gpurun_out/bench_details_n1.json $O/bench_details_n1.json
python tools/imu_isa_count.py > $O/imu_isa_count.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +4M -delete
