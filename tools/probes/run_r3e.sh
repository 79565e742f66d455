python -m pytest tests/test_gen_head_gpu.py tests/test_ddp_gpu.py tests/test_generate_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "gen_head or rccl or maskgit_accepts or gemm_bf16_layouts or decode_fused or kernels_match_host" > gpurun_out/r3e_tests.log 2>&1
tail -40 gpurun_out/r3e_tests.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err; tail -5 gpurun_out/r3e_bench.err; cat gpurun_out/r3e_bench.json | cut -c1-6000
