#!/bin/bash
# One recipe for every A/B this directory used to keep as a run_r3?.sh / run_r4?.sh of its own (round 5 housekeeping).
#
#   ab.sh env  VAR    "v1 v2 ..."          [-r REPS] [-o NAME] -- command ...     the command once per value with VAR=value
#   ab.sh lib  "ship name1 name2 ..."     [-r REPS] [-o NAME] -- command ...     ... with UNIGEN_HIP_LIB=_build/libunigen_hip_<name>.so
#                                                                                 ("ship" = the tree's own library); build the variants
#                                                                                 first: python tools/probes/build_variant.py <name> <file.hip> -D...
#   ab.sh prof NAME -- command ...                                                rocprofv3 --kernel-trace --stats of the command; the
#                                                                                 kernel stats land in gpurun_out/NAME_kernel_stats.csv
#
# Every line of the command's stdout is prefixed with the value / variant; -o NAME tees everything to gpurun_out/NAME.txt.  Run on the
# GPU box from the repo root (gpurun -- 'bash tools/probes/ab.sh ...').  Examples (the experiments of rounds 3-5 were these):
#   ab.sh env UNIGEN_FUSED_ROPE "0 1" -r 3 -- python3 bench.py --no-cpu-baseline --no-ar --no-extra --steps 8
#   ab.sh env UNIGEN_ATTN_DKV_HEADS "2 3 6" -- python3 tools/attn_bench.py
#   ab.sh lib "ship mo1 mo4" -r 3 -- env REPS=30 python3 tools/gemm_bench.py          (after build_variant.py mo1 gemm_bf16.hip -DUG_MFMA_ORDER=1 ...)
#   ab.sh lib "ship adf1 adf2 adf8" -- python3 tools/ar_bench.py graph                 (attention-decode ablations, -DUG_ADF_ABLATE=n)
#   ab.sh prof r05_ar -- python3 tools/ar_bench.py graph && python3 tools/ar_timeline.py gpurun_out/prof_r05_ar
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
mkdir -p gpurun_out
mode=$1; shift
reps=1; outname=""
case $mode in
  env) var=$1; values=$2; shift 2 ;;
  lib) values=$1; shift ;;
  prof) outname=$1; shift ;;
  *) echo "usage: see the header of $0" >&2; exit 2 ;;
esac
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case $1 in -r) reps=$2; shift 2 ;; -o) outname=$2; shift 2 ;; *) echo "unknown option $1" >&2; exit 2 ;; esac
done
shift
sink=/dev/null; [ -n "$outname" ] && [ $mode != prof ] && sink=gpurun_out/$outname.txt && : > "$sink"
if [ $mode = prof ]; then
  cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
  rm -rf gpurun_out/prof_$outname
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$outname -- "$@" > gpurun_out/${outname}_stdout.txt 2> gpurun_out/${outname}_stderr.txt
  f=$(find gpurun_out/prof_$outname -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/${outname}_kernel_stats.csv && python3 tools/stats_top.py "$f" 2>/dev/null | head -30
  exit 0
fi
for rep in $(seq $reps); do
  for v in $values; do
    if [ $mode = env ]; then
      env "$var=$v" "$@" 2>/dev/null | grep -v "amdgpu.ids" | sed "s|^|$var=$v: |" | tee -a "$sink"
    else
      lib=""; [ "$v" != ship ] && lib="$PWD/tools/probes/_build/libunigen_hip_$v.so"
      env UNIGEN_HIP_LIB="$lib" "$@" 2>/dev/null | grep -v "amdgpu.ids" | sed "s|^|$v: |" | tee -a "$sink"
    fi
  done
done
