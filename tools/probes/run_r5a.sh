# round 5 (a): cache carry-over across launch boundaries
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 tools/probes/_build/l2_carry 2>&1 | tee gpurun_out/r5a_l2_carry.txt
