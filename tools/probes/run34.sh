TILES=-1,102 ONLY=sq8192,qkv_f,gu_f,down_f,gu_d,down_d,gu_w,down_w,head_f python tools/gemm_sweep.py > gpurun_out/r2_t34.log 2>&1
