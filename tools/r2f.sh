R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2f; mkdir -p $O; export TMPDIR=/tmp; cd $R
python -m pytest tests -m gpu -q -x -k "large_batch or lm_step" 2>&1 | tail -2
cd /tmp
for V in base occ2; do
  if [ $V = occ2 ]; then export CTVIO_SCHUR_OCC2=1; fi
  rocprofv3 --kernel-trace --stats -d $O/kt_$V -o kt -- python $R/bench.py --no-cpu-baseline --streams 1 --windows 1024 --steps 2 --warmup 1 --device-resident-only > $O/bench_$V.json 2> $O/kt_$V.err
  python $R/tests/prof_summary.py stats $(find $O/kt_$V -name "*.db") > $O/kstats_$V.txt; find $O/kt_$V -name "*.db" -delete
  grep -E "schur" $O/kstats_$V.txt
done
