python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -k "wgrad_group or tiny or golden or L771" 2>&1 | tail -5 > gpurun_out/r2_t53.log
python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-ar --no-extra 2>&1 | cut -c1-1300 >> gpurun_out/r2_t53.log
