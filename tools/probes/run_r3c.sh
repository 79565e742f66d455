python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_bf16_layouts and register_blocked" > gpurun_out/r3c_r4_tests.log 2>&1
tail -3 gpurun_out/r3c_r4_tests.log
TILES=-1,12 ROUNDS=3 python tools/gemm_sweep.py > gpurun_out/r3c_sweep.log 2>&1
cat gpurun_out/r3c_sweep.log
python -m pytest tests/test_ddp_gpu.py -m gpu -x -q -k rccl > gpurun_out/r3c_rccl.log 2>&1; tail -3 gpurun_out/r3c_rccl.log
python -m pytest tests/test_full_depth_gpu.py -m gpu -x -q -s > gpurun_out/r3c_full_depth.log 2>&1; tail -3 gpurun_out/r3c_full_depth.log
