# optimizer / tokenizer overlap timeline of the bench step (tools/overlap_timeline.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ov
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ov -- python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 4 > gpurun_out/ov_line.json 2>/dev/null
python3 tools/overlap_timeline.py gpurun_out/prof_ov > gpurun_out/ov_timeline.md
find gpurun_out/prof_ov -name "*.csv" -size +2M -delete
cat gpurun_out/ov_timeline.md
