cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_bench_shapes_gpu.py -m gpu -q -s -k L771 2>&1 | grep "pre-rounding\|passed\|failed" > gpurun_out/r2_t17.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02b -- python3 bench.py --no-cpu-baseline --no-ar --no-extra > gpurun_out/r02b_bench_line_profiled.json 2>/dev/null
f=$(find gpurun_out/prof_r02b -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r02b_bench_kernel_stats.csv; python3 tools/prof_summary.py $f 40 > gpurun_out/r02b_table.md
python3 bench.py > gpurun_out/r02b_bench_line.json 2>/dev/null
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc2_gemm_$c -- python3 tools/gemm_step_mix.py > /dev/null 2>&1
  B=4 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc2_conv_$c -- python3 tools/conv_bench.py > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc2_attn_$c -- python3 tools/attn_bench.py > /dev/null 2>&1
done
for d in gpurun_out/pmc2_*; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d" >> gpurun_out/r02b_pmc_raw.txt; python3 tools/pmc_summary.py $f gemm_kernel conv attn tail_finish >> gpurun_out/r02b_pmc_raw.txt; done
find gpurun_out/pmc2_* gpurun_out/prof_r02b -name "*.csv" -size +2M -delete
