# non-temporal epilogue stores of the 128...320-row kernel (-DUG_EPI_NT) against the shipped plain stores: fused launches and the step
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in ship ent; do
  lib=""; [ $v != ship ] && lib=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_$v.so
  echo "== $v $(UNIGEN_HIP_LIB=$lib python3 tools/swiglu_gemm_bench.py 2>&1 | grep 'M=' | tail -1) | $(UNIGEN_HIP_LIB=$lib python3 tools/swiglu_bwd_gemm_bench.py 2>&1 | grep 'M=' | tail -1)"
done; done
for rep in 1 2 3; do for v in ship ent; do
  lib=""; [ $v != ship ] && lib=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_$v.so
  UNIGEN_HIP_LIB=$lib python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 8 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d['roofline']['by_family']
print('$v', d['ms_per_step'], d['roofline']['fwd_bwd_1p5b']['ms'], f['gemm']['ms_per_step'])"
done; done
