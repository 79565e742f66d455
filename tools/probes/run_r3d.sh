for v in full nodma noreads nobar mfmaonly; do python tools/probes/gemm_r4_ablate.py $v 2>&1 | grep "^\[" ; done > gpurun_out/r3d_ablate.log 2>&1
cat gpurun_out/r3d_ablate.log
python -m pytest tests/test_gen_head_gpu.py tests/test_full_depth_gpu.py -m gpu -q -s 2>&1 | grep -E "passed|failed|AR on|distance to the fp32 grad" > gpurun_out/r3d_tests.log; cat gpurun_out/r3d_tests.log
