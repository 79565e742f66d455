python -m pytest tests/test_siglip_gpu.py tests/test_sft_gpu.py -m gpu -q -x -s > gpurun_out/r3j_siglip.log 2>&1; tail -30 gpurun_out/r3j_siglip.log
