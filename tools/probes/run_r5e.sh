cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
UNIGEN_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_dtrace.so timeout 600 python3 tools/probes/decode_trace.py 2>&1 | grep -v Warning | tee gpurun_out/r5e_decode_trace.txt
