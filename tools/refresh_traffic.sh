# Re-measure profiles/gemm_traffic_current.json for the current gemm_bf16.hip (the four PMC passes of tools/prof_round.sh only):
#   gpurun -- 'bash tools/refresh_traffic.sh'   then copy gpurun_out/gemm_traffic_current.json to profiles/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_layers_* gpurun_out/pmc_head_*
for g in layers head; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${g}_$c -- python3 tools/gemm_step_mix.py $g > /dev/null 2>&1; done; done
python3 tools/gemm_traffic.py gpurun_out
find gpurun_out/pmc_* -name "*.csv" -size +2M -delete
