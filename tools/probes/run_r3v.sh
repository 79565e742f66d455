# RMSNorm forward: row kept in registers (UNIGEN_RN_FWD_REG = rows per wave, 0 = the two-pass form) inside the step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in ${REGS:-0 1 2 4 0 1}; do
  export UNIGEN_RN_FWD_REG=$r
  rm -rf gpurun_out/prof_rn
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rn -- python3 bench.py --no-cpu-baseline --no-ar --no-extra --no-roofline > gpurun_out/q.json 2>/dev/null
  f=$(find gpurun_out/prof_rn -name "*kernel_stats.csv" | head -1)
  echo "reg=$r step $(python3 -c "import json;print(json.load(open('gpurun_out/q.json'))['ms_per_step'])") $(grep rmsnorm_fwd $f | awk -F'","|",|,' '{print "rmsnorm_fwd avg ns", $(NF-4)}')"
done
