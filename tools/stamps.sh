R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stamps; mkdir -p $O; cd $R
CTVIO_DEBUG_STAMPS=1 python bench.py --no-cpu-baseline --streams 1 --windows 1024 --steps 1 --warmup 1 --device-resident-only > $O/b.json 2> $O/b.err
grep "ctvio" $O/b.err | tail -6
