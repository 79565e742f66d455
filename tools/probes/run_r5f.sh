cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in unset 0 1 unset 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  echo "HIP_FORCE_DEV_KERNARG=$v: $(python3 tools/ar_bench.py graph 2>/dev/null | tail -1)"
done | tee gpurun_out/r5f_kernarg.txt
