# round 5 (c): per-launch timeline of the captured AR step under a few prefetch plans
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
i=0
for plan in off "qkv>o:0-1:64" "qkv>o:0-1:16" "attn>gu:0-1:64" "gu>down:0-1:32"; do
  i=$((i+1))
  rm -rf gpurun_out/prof_ar
  export UNIGEN_DECODE_PREFETCH="$plan"
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ar -- python3 tools/ar_bench.py graph > gpurun_out/r5c_line_$i.json 2>gpurun_out/r5c_$i.err
  echo "## plan: $plan" >> gpurun_out/r5c_timelines.md
  python3 tools/ar_timeline.py gpurun_out/prof_ar 2>>gpurun_out/r5c_$i.err | head -16 >> gpurun_out/r5c_timelines.md
  rm -rf gpurun_out/prof_ar
done
cat gpurun_out/r5c_timelines.md
