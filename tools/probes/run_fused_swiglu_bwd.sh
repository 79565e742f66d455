# down dgrad with the SwiGLU backward in its epilogue (UNIGEN_FUSED_SWIGLU_BWD=1, default) against dgrad + ug_swiglu_bwd
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_kernels_gpu.py -x -q -k "swiglu" 2>&1 | tail -3
for rep in 1 2 3; do for v in 0 1; do
  UNIGEN_FUSED_SWIGLU_BWD=$v python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 8 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d['roofline']['by_family']
print('fused_bwd=$v', d['ms_per_step'], d['roofline']['fwd_bwd_1p5b']['ms'], f['gemm']['ms_per_step'], f['elementwise']['ms_per_step'])"
done; done
