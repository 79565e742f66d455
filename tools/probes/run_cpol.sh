# cache policy of the GEMM operand streams (nt on the A stream of the 128...320-row kernel / on the 256-wide stager / both): TF/s
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in ship cpa cpb cpab; do
  lib=""; [ $v != ship ] && lib=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_$v.so
  echo "== $v"; UNIGEN_HIP_LIB=$lib REPS=30 python3 tools/gemm_bench.py 2>&1 | grep -E "^(gu|down|o ) "
done; done
