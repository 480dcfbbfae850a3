set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
python bench.py > gpurun_out/prof/bench_n1.json 2> gpurun_out/prof/bench_n1.err; tail -c 600 gpurun_out/prof/bench_n1.err
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/kt -o kt -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof/kt_bench.json 2> $R/gpurun_out/prof/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_fetch -o f -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/prof/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof/pmc_write -o w -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /dev/null 2> $R/gpurun_out/prof/pmc_write.err
cd $R
find gpurun_out/prof -type f | head -30
python tests/prof_summary.py stats $(find gpurun_out/prof/kt -name "*.db") > gpurun_out/prof/kernel_stats.txt; head -30 gpurun_out/prof/kernel_stats.txt
python tests/prof_summary.py pmc 256 gpurun_out/prof/pmc_traffic.json $(find gpurun_out/prof/pmc_fetch gpurun_out/prof/pmc_write -name "*counter_collection.csv") | tee gpurun_out/prof/pmc_table.txt
find gpurun_out/prof -name "*.db" -size +20M -delete; find gpurun_out/prof -name "*.csv" -size +20M -delete
