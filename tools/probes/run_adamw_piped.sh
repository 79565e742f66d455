# the software-pipelined lean update (UNIGEN_ADAMW_PIPED=1, default) against the plain one: step ms, then the overlap timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "adamw" 2>&1 | tail -2
for v in 0 1 0 1; do
  UNIGEN_ADAMW_PIPED=$v python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 8 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d['roofline']['by_family']
print('piped=$v', d['ms_per_step'], d['roofline']['fwd_bwd_1p5b']['ms'], f['tokenizer_and_towers']['ms_per_step'], f['adamw']['ms_per_step'])"
done
rm -rf gpurun_out/prof_ov
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ov -- python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 4 > /dev/null 2>&1
python3 tools/overlap_timeline.py gpurun_out/prof_ov | tail -7
rm -rf gpurun_out/prof_ov
