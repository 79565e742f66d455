cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/r3i_counters.txt
rm -rf gpurun_out/pmc3_*
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/pmc3_a -- python3 tools/attn_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc3_b -- python3 tools/attn_bench.py > /dev/null 2>&1
for d in gpurun_out/pmc3_a gpurun_out/pmc3_b; do f=$(find $d -name "*counter_collection.csv" | head -1); python3 tools/pmc_summary.py $f attn_fwd attn_bwd; done > gpurun_out/r3i_attn_pmc.txt 2>&1
cat gpurun_out/r3i_attn_pmc.txt
find gpurun_out/pmc3_* -name "*.csv" -size +2M -delete
