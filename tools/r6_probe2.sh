R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 /tmp/mfma4_layout.hip -o /tmp/mfma4_layout 2>/dev/null || hipcc --offload-arch=gfx950 -O3 $R/tools/mfma4_layout.hip -o /tmp/mfma4_layout 2>/dev/null
timeout 60 /tmp/mfma4_layout > $O/mfma4_layout.txt 2> $O/mfma4_layout.err; cat $O/mfma4_layout.err; wc -l $O/mfma4_layout.txt
