# the overlapped optimizer update beside the tokenizer: workgroup count of the lean kernel (UNIGEN_ADAMW_LEAN = 1: 256; n > 1: n; 0: the
# four-element kernel), step ms / fwd_bwd ms / tokenizer family ms / adamw family ms
cd $GRAFT_REPO_ROOT
for v in 1 64 128 512 1024 0 1; do
  UNIGEN_ADAMW_LEAN=$v python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 8 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d['roofline']['by_family']
print('lean=$v', d['ms_per_step'], d['roofline']['fwd_bwd_1p5b']['ms'], f['tokenizer_and_towers']['ms_per_step'], f['adamw']['ms_per_step'])"
done
