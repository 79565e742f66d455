cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -4
for v in 0 1; do echo "FWD32P=$v"; UNIGEN_ATTN_FWD_STAG=$v python3 tools/attn_bench.py 2>&1 | grep -v Warn; done
