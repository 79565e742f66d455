python -m pytest tests -m gpu -q > gpurun_out/r3g_gpu_tests.log 2>&1
tail -8 gpurun_out/r3g_gpu_tests.log
python tools/ddp_contention.py > gpurun_out/r3g_contention.log 2>&1; cat gpurun_out/r3g_contention.log | tail -20
