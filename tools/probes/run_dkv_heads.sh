# query heads per workgroup of the split-head dK / dV kernel: 2 (shipped), 3, 6 -- attention family ms and forward + backward ms
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for h in 2 3 6; do
  UNIGEN_ATTN_DKV_HEADS=$h UNIGEN_ATTN_DKV_MIN_WGS=256 python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 6 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d['roofline']['by_family']
print('heads=$h', d['ms_per_step'], d['roofline']['fwd_bwd_1p5b']['ms'], f['attention']['ms_per_step'], f['elementwise']['ms_per_step'])"
done; done
