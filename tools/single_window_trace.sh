R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/w1; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --streams 1 --windows 1 --steps 20 --warmup 2 --device-resident-only > $O/b.json 2> $O/kt.err
python $R/tools/prof_summary.py stats $(find $O/kt -name "*.db") > $O/kstats.txt; find $O/kt -name "*.db" -delete
head -24 $O/kstats.txt
