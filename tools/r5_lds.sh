R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lds; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
B1="env CTVIO_SPLIT_LINEARIZE=1 python $R/bench.py --no-cpu-baseline --quick --steps 1 --warmup 1 --device-resident-only --streams 1 --windows 2048"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_lds -o l -- $B1 > /dev/null 2> $O/pmc_lds.err
cd $R; python tools/prof_summary.py counters $O/pmc_lds.json $(find $O/pmc_lds -name "*counter_collection.csv") > $O/pmc_lds_table_2048.txt; cat $O/pmc_lds_table_2048.txt; tail -3 $O/pmc_lds.err
