python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -k "attention or attn or L771 or L1603" > gpurun_out/r3l_tests.log 2>&1; tail -5 gpurun_out/r3l_tests.log
for i in 1 2; do
echo "== 16-row dQ"; UNIGEN_ATTN_DQ32=0 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
echo "== 32-row dQ"; UNIGEN_ATTN_DQ32=1 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
done
