# kernel-trace average of one kernel family inside the step: bash tools/probes/run_r3x.sh <pattern>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_x
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -- python3 bench.py --no-cpu-baseline --no-ar --no-extra --no-roofline > gpurun_out/q.json 2>/dev/null
f=$(find gpurun_out/prof_x -name "*kernel_stats.csv" | head -1)
echo "step $(python3 -c "import json;d=json.load(open('gpurun_out/q.json'));print(d['ms_per_step'], d['loss_first_last'])")"
grep -E "$1" $f | awk -F'","|",|,' '{print substr($1,1,60), "calls", $(NF-6), "avg ns", $(NF-4)}'
