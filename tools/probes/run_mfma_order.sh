# MFMA issue order inside a k-tile (probe builds -DUG_MFMA_ORDER=1/2/3) against the shipped library: TF/s at the power cap
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in ship mo3 mo1 mo2; do
  lib=""; [ $v != ship ] && lib=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_$v.so
  echo "== $v"; UNIGEN_HIP_LIB=$lib REPS=30 python3 tools/gemm_bench.py 2>&1 | grep -E "^(gu|down|o ) "
done; done
