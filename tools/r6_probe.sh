set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp; cd $R
hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_4x4_probe.hip -o /tmp/mfma4 2>/dev/null && timeout 120 /tmp/mfma4 | tee $O/mfma_f64_4x4_probe.txt | head -60
timeout 900 python -m pytest tests -m gpu -q -x -k "marginalize or slide or lm_step" 2>&1 | tail -5
CTVIO_DEBUG_STAMPS=1 timeout 120 python bench.py --no-cpu-baseline --quick --steps 1 --warmup 0 --device-resident-only --streams 1 --windows 1 2>&1 | grep -a "clock64" | tail -6
