# RMSNorm backward: rows per workgroup (UNIGEN_RN_RPB) inside the step -- kernel-trace average of rmsnorm_bwd_kernel per setting
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in ${RPBS:-16 4 8 12 24 48}; do
  export UNIGEN_RN_RPB=$r
  rm -rf gpurun_out/prof_rn
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rn -- python3 bench.py --no-cpu-baseline --no-ar --no-extra --no-roofline > gpurun_out/q.json 2>/dev/null
  f=$(find gpurun_out/prof_rn -name "*kernel_stats.csv" | head -1)
  echo "rpb=$r step $(python3 -c "import json;print(json.load(open('gpurun_out/q.json'))['ms_per_step'])") $(grep rmsnorm_bwd $f | awk -F'","|",|,' '{print "rmsnorm_bwd avg ns", $(NF-4)}')"
done
