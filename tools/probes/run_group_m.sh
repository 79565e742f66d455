# row panels per column sweep of the 128...320-row kernel (UG_P10_GROUP_M; shipped 4): multi-round launches, TF/s
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in ship gm2 gm3 gm6; do
  lib=""; [ $v != ship ] && lib=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_$v.so
  echo "== $v $(UNIGEN_HIP_LIB=$lib python3 tools/swiglu_gemm_bench.py 2>&1 | grep 'M=' | tail -1) | $(UNIGEN_HIP_LIB=$lib python3 tools/swiglu_bwd_gemm_bench.py 2>&1 | grep 'M=' | tail -1)"
done; done
