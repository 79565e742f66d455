# round 5 (b): AR decode prefetch plans (nblocks -1 = in-wave touches)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 tools/ar_prefetch_sweep.py off "qkv>o:0-1:-1" "down>qkv:0-1:-1" "attn>gu:0-1:-1" "attn>gu:0-2:-1" "o>gu:0-1:-1" \
  "gu>down:0-1:-1" "gu>down:0-2:-1" "qkv>o:0-1:-1,attn>gu:0-2:-1,gu>down:0-2:-1,down>qkv:0-1:-1" \
  "qkv>o:0-1:-1,attn>gu:0-1:-1,o>gu:1-2:-1,gu>down:0-2:-1,down>qkv:0-1:-1" "qkv>o:0-1:-1,attn>gu:0-1:-1,down>qkv:0-1:-1" off 2>&1 | grep -v Warning | tee gpurun_out/r5b_prefetch_sweep_inwave.txt
