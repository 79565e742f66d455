# in-step A/B of two builds of the library: UNIGEN_HIP_LIB points the binding at tools/probes/_build/lib_head.so (a copy of the
# build to compare against), the default is the tree's own; prints step / forward+backward / attention / GEMM ms per run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-ar --no-extra > gpurun_out/q.json 2>/dev/null; echo "$1: $(python3 -c "import json;d=json.load(open('gpurun_out/q.json'));f=d['roofline']['by_family'];print(d['ms_per_step'], 'fwd_bwd', d['roofline']['fwd_bwd_1p5b']['ms'], 'attn', f['attention']['ms_per_step'], 'gemm', f['gemm']['ms_per_step'])")"; }
for i in 1 2; do UNIGEN_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/_build/lib_head.so run head; run end_aligned; done
