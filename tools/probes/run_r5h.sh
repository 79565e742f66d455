cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r5h_gpu_tests.txt
