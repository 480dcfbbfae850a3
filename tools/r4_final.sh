# round 4, final: the whole GPU suite, then the round-4 profile (PMC passes, traces, full bench line)
set -x
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof4
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > gpurun_out/prof4/pytest.txt; cat gpurun_out/prof4/pytest.txt
bash tools/profile_round4.sh
