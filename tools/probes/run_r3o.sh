python -m pytest tests -m gpu -q > gpurun_out/r3o_gpu_tests.log 2>&1
tail -5 gpurun_out/r3o_gpu_tests.log
