# what the overlapped optimizer update takes from the tokenizer beside it: the shipped kernel, its memory streams alone
# (-DUG_ADAMW_ABLATE=1), its arithmetic alone (=2), and no overlap at all (UNIGEN_ADAMW_OVERLAP=0); step ms and the timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ship adw1 adw2 ship; do
  lib=""; [ $v != ship ] && lib=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_$v.so
  UNIGEN_HIP_LIB=$lib python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 8 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); f = d['roofline']['by_family']
print('$v', d['ms_per_step'], d['roofline']['fwd_bwd_1p5b']['ms'], f['tokenizer_and_towers']['ms_per_step'], f['adamw']['ms_per_step'])"
done
for v in adw1 adw2; do
  rm -rf gpurun_out/prof_ov
  UNIGEN_HIP_LIB=$GRAFT_REPO_ROOT/tools/probes/_build/libunigen_hip_$v.so rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ov -- python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 4 > /dev/null 2>&1
  echo "== $v"; python3 tools/overlap_timeline.py gpurun_out/prof_ov | tail -7
done
rm -rf gpurun_out/prof_ov
