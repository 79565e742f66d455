# in-step effect of non-temporal accesses in the streaming element-wise kernels (UNIGEN_EW_NT hex digits: adamw | rmsnorm_bwd |
# swiglu_bwd | swiglu_fwd; bit 1 = loads, bit 0 = stores): step ms, element-wise family ms, GEMM ms, tokenizer ms per setting
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nt in ${NTS:-0000 0032 0232 0332 3032 3232 0000 0232}; do
  export UNIGEN_EW_NT=$nt
  timeout 300 python3 bench.py --no-cpu-baseline --no-ar --no-extra > gpurun_out/ew_line_$nt.json 2>/dev/null
  echo "NT=$nt: $(python3 -c "import json;d=json.load(open('gpurun_out/ew_line_$nt.json'));f=d['roofline']['by_family'];print(d['ms_per_step'], 'ew', f['elementwise']['ms_per_step'], 'gemm', f['gemm']['ms_per_step'], 'tok', f['tokenizer_and_towers']['ms_per_step'], 'attn', f['attention']['ms_per_step'], 'adamw', f['adamw']['ms_per_step'])")"
done
