cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1 2 3 4 7 8; do
  rm -rf gpurun_out/prof_adf
  if [ $v = 0 ]; then unset UNIGEN_HIP_LIB; else export UNIGEN_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_adf$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_adf -- python3 tools/ar_bench.py graph > /dev/null 2>&1
  f=$(find gpurun_out/prof_adf -name "*kernel_stats.csv" | head -1)
  echo "== ablate $v"; python3 tools/stat_of.py $f attn_decode_fused gemv_ring
done > gpurun_out/r4b_adf_ablate.txt 2>&1
rm -rf gpurun_out/prof_adf
cat gpurun_out/r4b_adf_ablate.txt
