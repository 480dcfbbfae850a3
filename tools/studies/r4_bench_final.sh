# round 4: the default bench line of the final code (profiles/r04_bench_n1.json) + the whole GPU suite
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4final; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
( time python bench.py > $O/bench_n1.json 2> $O/bench_n1.err ) 2>&1 | tail -4; tail -c 300 $O/bench_n1.err
python -c "
import json; d=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1])
print(d['value'], d['device_resident_solves_per_s'], d['cpu_baseline']['value'], d['cpu_baseline_all_cores'], d['host_pack_only'], d['host_share_of_an_8_rank_run'])"
