cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in ship mo2; do
  lib=""; [ $v != ship ] && lib=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_$v.so
  UNIGEN_HIP_LIB=$lib python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 8 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d['roofline']['by_family']
print('$v', d['ms_per_step'], d['roofline']['fwd_bwd_1p5b']['ms'], f['gemm']['ms_per_step'])"
done; done
