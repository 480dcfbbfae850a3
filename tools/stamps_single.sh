R=$GRAFT_REPO_ROOT; cd $R
CTVIO_DEBUG_STAMPS=1 python bench.py --no-cpu-baseline --streams 1 --windows 1 --steps 1 --warmup 1 --device-resident-only 2>&1 >/dev/null | grep "cholesky" | tail -1
