cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_generate_gpu.py tests/test_gen_head_gpu.py -x -q -k "decode or generat or ar_" 2>&1 | tail -15
rm -rf gpurun_out/prof_ar
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ar -- python3 tools/ar_bench.py graph > gpurun_out/r4c_ar_line.json 2>/dev/null
f=$(find gpurun_out/prof_ar -name "*kernel_stats.csv" | head -1); python3 tools/stat_of.py $f attn_decode gemv_ ar_sample finish_resid
rm -rf gpurun_out/prof_ar
python3 tools/ar_bench.py graph
