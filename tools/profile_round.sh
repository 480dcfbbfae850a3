# Round profile on the GPU box: PMC traffic passes (separate FETCH_SIZE / WRITE_SIZE runs), the default bench line, the
# rocprofv3 kernel-trace stats of the same command.  Outputs under gpurun_out/prof (copy the summaries to profiles/).
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -o f -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/prof/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -o w -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/prof/pmc_write.err
cd $R
rm -f gpurun_out/prof/pmc_traffic.json
python tools/prof_summary.py pmc 256 gpurun_out/prof/pmc_traffic.json $(find gpurun_out/prof/pmc_fetch gpurun_out/prof/pmc_write -name "*counter_collection.csv") | tee gpurun_out/prof/pmc_table.txt
cp gpurun_out/prof/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --host-inclusive > gpurun_out/prof/bench_n1.json 2> gpurun_out/prof/bench_n1.err; tail -c 300 gpurun_out/prof/bench_n1.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/kt -o kt -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof/kt_bench.json 2> $R/gpurun_out/prof/kt.err
cd $R
python tools/prof_summary.py stats $(find gpurun_out/prof/kt -name "*.db") > gpurun_out/prof/kernel_stats.txt; head -24 gpurun_out/prof/kernel_stats.txt
find gpurun_out/prof -name "*.db" -delete; find gpurun_out/prof -name "*.csv" -size +8M -delete
