cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tok
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tok -- python3 tools/tokenizer_bench.py > gpurun_out/r4j_tok.txt 2>&1
f=$(find gpurun_out/prof_tok -name "*kernel_stats.csv" | head -1); python3 tools/stats_top.py $f 30 2>/dev/null || python3 tools/prof_summary.py $f 30
tail -5 gpurun_out/r4j_tok.txt
rm -rf gpurun_out/prof_tok
