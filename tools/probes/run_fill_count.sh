# how many torch fill / copy kernels does ONE training step launch?  kernel-trace stats of bench.py at 5 and at 25 timed steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 5 25; do
  rm -rf gpurun_out/prof_fc$n
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fc$n -- python3 bench.py --no-cpu-baseline --no-ar --no-extra --no-roofline --steps $n > /dev/null 2>&1
  f=$(find gpurun_out/prof_fc$n -name "*kernel_stats.csv" | head -1)
  echo "== steps $n"; python3 tools/stat_of.py $f Fill copyBuffer fillBuffer elementwise distribution 2>/dev/null | cut -c1-150
  find gpurun_out/prof_fc$n -name "*.csv" -size +2M -delete
done
