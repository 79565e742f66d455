# the overlapped AdamW update in its lean (two elements per lane, <= 40 registers) form beside the tokenizer's convolutions:
# UNIGEN_ADAMW_LEAN = 0 (four-element kernel, 256 workgroups) | 1 (lean, 256 workgroups) | N > 1 (lean, N workgroups)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 300 python3 bench.py --no-cpu-baseline --no-ar --no-extra > gpurun_out/q.json 2>/dev/null; echo "$1: $(python3 -c "import json;d=json.load(open('gpurun_out/q.json'));f=d['roofline']['by_family'];print(d['ms_per_step'], 'fwd_bwd', d['roofline']['fwd_bwd_1p5b']['ms'], 'tok', f['tokenizer_and_towers']['ms_per_step'], 'adamw', f['adamw']['ms_per_step'], 'gemm', f['gemm']['ms_per_step'], d['loss_first_last'])")"; }
for l in ${LEANS:-0 1 512 1024 2048 0 1}; do UNIGEN_ADAMW_LEAN=$l run lean=$l; done
