cd $GRAFT_REPO_ROOT
for v in 1 3 5 7; do echo "ORDER=$v: $(UNIGEN_ATTN_ORDER=$v ROPE=2 python3 tools/attn_bench.py 2>&1 | grep -E 'causal|full' | tr '\n' ' ') $(UNIGEN_ATTN_ORDER=$v python3 tools/attn_bench.py 2>&1 | grep causal)"; done
