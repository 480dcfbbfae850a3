# round 4: clock stamps of the visual kernels under a full batch + PMC traffic of the new visual path + the new mixed-batch test
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c2; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "mixed_batch or edge or golden_edge" 2>&1 | tail -5
CTVIO_DEBUG_STAMPS=1 python bench.py --no-cpu-baseline --streams 1 --windows 2048 --steps 1 --warmup 1 --device-resident-only 2>&1 >/dev/null | grep "ctvio\]" | tail -4 > $O/stamps.txt; cat $O/stamps.txt
cd /tmp
B1="env CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --steps 1 --warmup 1 --device-resident-only --streams 1 --windows 2048"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- $B1 > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- $B1 > /dev/null 2> $O/pmc_write.err
cd $R
python tools/prof_summary.py pmc 2048 $O/pmc_traffic.json $(find $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv") > $O/pmc_table_2048.txt; cat $O/pmc_table_2048.txt
find $O -name "*.db" -delete; find $O -name "*.csv" -size +4M -delete
