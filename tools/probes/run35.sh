python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" 2>&1 | tail -3 > gpurun_out/r2_t35.log
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ar --no-extra 2>&1 | cut -c1-1200 >> gpurun_out/r2_t35.log
