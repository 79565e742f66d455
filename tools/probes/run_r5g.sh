cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "decode or gemv or skinny" 2>&1 | tail -5
timeout 900 python3 -m pytest tests/test_generate_gpu.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2; do python3 tools/ar_bench.py graph 2>/dev/null | tail -1; done | tee gpurun_out/r5g_ar.txt
