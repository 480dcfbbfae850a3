cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cp ctrl-vio_amd/libctvio.so /tmp/base.so
for v in NOEPI NOLOOP; do
  cp ctrl-vio_amd/libctvio_$v.so ctrl-vio_amd/libctvio.so
  echo "== $v"; cd /tmp; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kx_$v -o kt -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 1 --warmup 0 --iters 1 > /dev/null 2>&1
  cd $R; python tests/prof_summary.py stats $(find gpurun_out/kx_$v -name "*.db") | grep -E "schur|cholesky"; find gpurun_out/kx_$v -name "*.db" -delete
done
cp /tmp/base.so ctrl-vio_amd/libctvio.so
