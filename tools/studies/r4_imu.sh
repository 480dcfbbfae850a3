# round 4: the IMU kernel as a walk over groups with the next group's data in flight -- parity subset, then the wave count swept
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4imu; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "linearize or lm_step or product_parity or ragged or edge or golden or large_batch or mixed_batch or deterministic or iterates or imu" 2>&1 | tail -5
for v in 1000000 1024 2048 4096 8192; do
  CTVIO_IMU_WAVES=$v python bench.py --no-cpu-baseline --quick --streams 1 --windows 2048 --steps 3 --warmup 1 --device-resident-only > $O/b_$v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$O/b_$v.json').read().strip().splitlines()[-1]); print('imu waves $v: 2048 windows', round(d['value']), round(d['ms_per_step'],2), {k: round(x,2) for k,x in d['phase_ms_profiled_solve'].items()})"
done
CTVIO_DEBUG_STAMPS=1 python bench.py --no-cpu-baseline --streams 1 --windows 2048 --steps 1 --warmup 1 --device-resident-only 2>&1 >/dev/null | grep "imu fast" | tail -1
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --quick > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['device_resident_solves_per_s'])"
