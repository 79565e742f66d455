cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ar
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ar -- python3 tools/ar_bench.py graph > gpurun_out/r4a_ar_line.json 2>gpurun_out/r4a_ar.err
f=$(find gpurun_out/prof_ar -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04_ar_kernel_stats.csv
python3 tools/ar_timeline.py gpurun_out/prof_ar > gpurun_out/r04_ar_timeline.md 2>gpurun_out/r4a_tl.err
find gpurun_out/prof_ar -name "*.csv" -size +1M -delete
python3 tools/ar_bench.py > gpurun_out/r4a_ar_unprofiled.json 2>&1
tail -5 gpurun_out/r04_ar_timeline.md
