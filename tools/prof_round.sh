# One round's evidence, run on the GPU box (gpurun -- 'bash tools/prof_round.sh r02c'): kernel stats of the bench command, the
# bench line itself, MFMA-busy / LDS-conflict counters of the three MFMA kernels, and the HBM traffic of the GEMM launches
# (separate --pmc passes as MI355X_MICROARCH.md prescribes).  Everything lands in gpurun_out/<tag>_*; copy what is to be judged
# into profiles/.
tag=${1:-r02c}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_* gpurun_out/pmc2_*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline --no-ar --no-extra > gpurun_out/${tag}_bench_line_profiled.json 2>/dev/null
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${tag}_bench_kernel_stats.csv; python3 tools/prof_summary.py $f 40 > gpurun_out/${tag}_table.md
for g in layers head; do for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${g}_$c -- python3 tools/gemm_step_mix.py $g > /dev/null 2>&1; done; done
python3 tools/gemm_traffic.py gpurun_out > gpurun_out/${tag}_traffic.log 2>&1
cp gpurun_out/gemm_traffic_current.json profiles/gemm_traffic_current.json     # (on the box: lets the bench line below carry it)
python3 bench.py > gpurun_out/${tag}_bench_line.json 2>/dev/null
rm -f gpurun_out/${tag}_pmc_raw.txt
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc2_gemm_$c -- python3 tools/gemm_step_mix.py > /dev/null 2>&1
  B=4 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc2_conv_$c -- python3 tools/conv_bench.py > /dev/null 2>&1
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc2_attn_$c -- python3 tools/attn_bench.py > /dev/null 2>&1
done
for d in gpurun_out/pmc2_*; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d" >> gpurun_out/${tag}_pmc_raw.txt; python3 tools/pmc_summary.py $f gemm_kernel conv attn tail_finish >> gpurun_out/${tag}_pmc_raw.txt; done
find gpurun_out/pmc_* gpurun_out/pmc2_* gpurun_out/prof_$tag -name "*.csv" -size +2M -delete
