# Issue / stall breakdown of the three MFMA kernel families (SQ counters, one pass each: eight SQ slots per pass):
#   gpurun -- 'bash tools/prof_stalls.sh r06'   ->  gpurun_out/<tag>_stalls_raw.txt
# WAIT_ANY (wave parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: MFMA RAW / pipe busy; WAIT_INST_LDS is its LDS part)
# + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots).
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES"
rm -rf gpurun_out/pmc3_*; rm -f gpurun_out/${tag}_stalls_raw.txt
rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc3_gemm -- python3 tools/gemm_step_mix.py > /dev/null 2>&1
B=4 rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc3_conv -- python3 tools/conv_bench.py > /dev/null 2>&1
rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc3_attn -- python3 tools/attn_bench.py > /dev/null 2>&1
for d in gpurun_out/pmc3_*; do f=$(find $d -name "*counter_collection.csv" | head -1); echo "== $d" >> gpurun_out/${tag}_stalls_raw.txt; python3 tools/pmc_summary.py $f gemm_kernel conv3x3 attn_ >> gpurun_out/${tag}_stalls_raw.txt; done
find gpurun_out/pmc3_* -name "*.csv" -size +2M -delete
