# round 5 (d): FETCH_SIZE of the consumer launches with and without a prefetch carried by the previous launch, + timelines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r5d_fetch.txt gpurun_out/r5d_timelines.md
i=0
for plan in off "attn>gu:0-1:-1" "qkv>o:0-1:-1" "gu>down:0-2:-1"; do
  i=$((i+1))
  export UNIGEN_DECODE_PREFETCH="$plan"
  rm -rf gpurun_out/pmc_ar
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_ar -- python3 tools/ar_bench.py graph > /dev/null 2>gpurun_out/r5d_$i.err
  f=$(find gpurun_out/pmc_ar -name "*counter_collection.csv" | head -1)
  echo "## plan: $plan" >> gpurun_out/r5d_fetch.txt
  python3 tools/pmc_summary.py $f gemv_ring attn_decode >> gpurun_out/r5d_fetch.txt
  rm -rf gpurun_out/pmc_ar gpurun_out/prof_ar
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ar -- python3 tools/ar_bench.py graph > gpurun_out/r5d_line_$i.json 2>>gpurun_out/r5d_$i.err
  echo "## plan: $plan" >> gpurun_out/r5d_timelines.md
  python3 tools/ar_timeline.py gpurun_out/prof_ar 2>>gpurun_out/r5d_$i.err | head -16 >> gpurun_out/r5d_timelines.md
  rm -rf gpurun_out/prof_ar
done
cat gpurun_out/r5d_fetch.txt
grep -v "^$" gpurun_out/r5d_timelines.md | grep "plan\|gemv\|attn\|span"
