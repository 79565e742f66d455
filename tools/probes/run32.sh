python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm_bf16_layouts" 2>&1 | tail -4 > gpurun_out/r2_t32.log
TILES=-1,102 ONLY=sq8192,qkv_f,gu_f,down_f,gu_d,down_d,gu_w,down_w,qkv_w,head_f,head_d,head_w python tools/gemm_sweep.py >> gpurun_out/r2_t32.log 2>&1
