# in-step probes: the gate_up GEMM with the SwiGLU epilogue (UNIGEN_FUSED_SWIGLU=1), RMSNorm backward without its dw atomics
# (upper bound of what a two-stage dw reduction could save; UNIGEN_EW_NT=3432 is a timing probe with WRONG norm gradients)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-ar --no-extra > gpurun_out/q.json 2>/dev/null; echo "$1: $(python3 -c "import json;d=json.load(open('gpurun_out/q.json'));f=d['roofline']['by_family'];print(d['ms_per_step'], 'fwd_bwd', d['roofline']['fwd_bwd_1p5b']['ms'], 'ew', f['elementwise']['ms_per_step'], 'gemm', f['gemm']['ms_per_step'], 'attn', f['attention']['ms_per_step'])")"; }
run base
UNIGEN_FUSED_SWIGLU=1 run fused_swiglu
UNIGEN_EW_NT=3432 run rmsnorm_bwd_no_dw_atomics
run base
UNIGEN_FUSED_SWIGLU=1 run fused_swiglu
