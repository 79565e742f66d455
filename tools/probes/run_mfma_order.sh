# MFMA issue order inside a k-tile (probe builds -DUG_MFMA_ORDER=n) against the shipped library (order 2): TF/s at the power cap
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in ship ${VARIANTS:-mo4}; do
  lib=""; [ $v != ship ] && lib=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_$v.so
  echo "== $v"; UNIGEN_HIP_LIB=$lib REPS=30 python3 tools/gemm_bench.py 2>&1 | grep -E "^(gu|down|o ) "
done; done
