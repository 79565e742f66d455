python -m pytest tests -m gpu -q -x > gpurun_out/r3f_gpu_tests.log 2>&1
tail -15 gpurun_out/r3f_gpu_tests.log
