# the q/k/v projection with RoPE in its epilogue (UNIGEN_FUSED_ROPE=1, default) against projection + ug_rope: step, forward + backward, families
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for v in 0 1; do
  UNIGEN_FUSED_ROPE=$v python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 8 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d['roofline']['by_family']
print('fused=$v', d['ms_per_step'], d['roofline']['fwd_bwd_1p5b']['ms'], f['gemm']['ms_per_step'], f['elementwise']['ms_per_step'])"
done; done
