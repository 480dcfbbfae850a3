# round-2 first look: fp64 vs mixed per-kernel profile (1 stream, 1024 windows), default bench lines
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2a
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
(find / \( -name ceres.h -o -name Dense -o -name "glog*.h" \) -not -path "/proc/*" 2>/dev/null | head -5) > $O/probe_ceres.txt
nproc >> $O/probe_ceres.txt
for P in fp64 fp32; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$P -o kt -- python $R/bench.py --no-cpu-baseline --streams 1 --steps 2 --warmup 1 --precision $P > $O/bench_1s_$P.json 2> $O/kt_$P.err
  python $R/tests/prof_summary.py stats $(find $O/kt_$P -name "*.db") > $O/kstats_$P.txt
  find $O/kt_$P -name "*.db" -delete
  python $R/bench.py --no-cpu-baseline --precision $P --steps 3 --warmup 1 > $O/bench_4s_$P.json 2> $O/bench_4s_$P.err
done
head -40 $O/kstats_fp64.txt
cat $O/bench_4s_fp64.json
