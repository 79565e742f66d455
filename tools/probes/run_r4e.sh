cd $GRAFT_REPO_ROOT
for v in 0 2 8 0 2; do
  if [ $v = 0 ]; then unset UNIGEN_HIP_LIB; else export UNIGEN_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_adf$v.so; fi
  echo "ablate $v: $(python3 tools/ar_bench.py graph 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")')"
done | tee gpurun_out/r4e_adf_wall.txt
