# clock64 stamps of one wave of k_vis_eval / one workgroup of k_assemble_vis_mfma / k_cholesky_flow under a full batch (2048 windows)
R=$GRAFT_REPO_ROOT; cd $R
CTVIO_DEBUG_STAMPS=1 python bench.py --no-cpu-baseline --streams 1 --windows 2048 --steps 1 --warmup 1 --device-resident-only 2>&1 >/dev/null | grep "ctvio\]" | tail -4
