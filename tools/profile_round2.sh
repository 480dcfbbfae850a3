# Round-2 profile on the GPU box (outputs under gpurun_out/prof2; the summaries are copied to profiles/ by hand):
#   kernel-trace stats of the default bench command, PMC HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, as
#   MI355X_MICROARCH.md prescribes), MFMA busy cycles / ops of the fp64 matrix-core kernels.
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof2; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|FETCH_SIZE|WRITE_SIZE" | head -40 > $O/counters_avail.txt
B1="python $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 --device-resident-only --streams 1 --windows 2048"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B1 > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc_mfma -o m -- $B1 > /dev/null 2> $O/pmc_mfma.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mops -o o -- $B1 > /dev/null 2> $O/pmc_mops.err
cd $R
rm -f $O/pmc_traffic.json
python tools/prof_summary.py pmc 2048 $O/pmc_traffic.json $(find $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv") > $O/pmc_table_2048.txt; head -30 $O/pmc_table_2048.txt
python tools/prof_summary.py counters $O/pmc_mfma.json $(find $O/pmc_mfma $O/pmc_mops -name "*counter_collection.csv") > $O/pmc_mfma_table.txt; head -40 $O/pmc_mfma_table.txt
tail -3 $O/pmc_mops.err
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; python bench.py --windows 1 --streams 1 --device-resident-only --steps 50 --no-cpu-baseline > $O/bench_single_window.json 2>/dev/null; python bench.py --config config5 --windows 128 --unique 8 --streams 1 --device-resident-only --steps 3 --no-cpu-baseline > $O/bench_config5_x128.json 2>/dev/null; tail -c 300 $O/bench_n1.err; cat $O/bench_n1.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/kt_bench.json 2> $O/kt.err
cd $R
python tools/prof_summary.py stats $(find $O/kt -name "*.db") > $O/kernel_stats_default.txt; head -24 $O/kernel_stats_default.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt1 -o kt -- $B1 > /dev/null 2>&1
cd $R; python tools/prof_summary.py stats $(find $O/kt1 -name "*.db") > $O/kernel_stats_1x2048.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +4M -delete
