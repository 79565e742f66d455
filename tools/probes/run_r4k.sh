cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -3
for v in 2 3 6; do echo "DKV_HEADS=$v"; UNIGEN_ATTN_DKV_HEADS=$v ROPE=2 python3 tools/attn_bench.py 2>&1 | grep -v Warn; UNIGEN_ATTN_DKV_HEADS=$v python3 tools/attn_bench.py 2>&1 | grep -v Warn;  done
