python -m pytest tests/test_kernels_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -k "attention or attn or L771 or L1603" > gpurun_out/r3k_tests.log 2>&1; tail -12 gpurun_out/r3k_tests.log
for i in 1 2; do
echo "== 16-row forward"; UNIGEN_ATTN_FWD32=0 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
echo "== 32-row forward"; UNIGEN_ATTN_FWD32=1 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
done
