cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ar
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ar -- python3 tools/ar_bench.py graph > /dev/null 2>&1
python3 tools/probes/ar_attn_vs_len.py gpurun_out/prof_ar attn_decode "gemv_ring4_kernel<1, 1, 1, 4>" "gemv_ring_kernel<1, 1>"
rm -rf gpurun_out/prof_ar
