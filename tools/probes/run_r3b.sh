python -m pytest tests/test_full_depth_gpu.py -m gpu -x -q -s > gpurun_out/r3b_full_depth.log 2>&1
python -m pytest tests/test_ddp_gpu.py tests/test_model_gpu.py tests/test_bench_shapes_gpu.py tests/test_gen_head_gpu.py -m gpu -q -s > gpurun_out/r3b_tests.log 2>&1
tail -5 gpurun_out/r3b_full_depth.log gpurun_out/r3b_tests.log
