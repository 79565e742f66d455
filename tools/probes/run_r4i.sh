cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -3
python3 tools/attn_bench.py 2>&1 | grep -v Warn
python -m pytest tests/test_ddp_gpu.py -x -q -k "rccl_world_one" 2>&1 | tail -3
