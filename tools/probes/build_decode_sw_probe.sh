#!/bin/bash
# Builds tools/probes/_build/decode_sw_probe against the in-tree library (run from the repo root; needs `make -C ml-unigen_amd/csrc` first).
# Extra arguments go to hipcc (-DUG_HAVE_OGU: the fused o + gate/up launch of decode_ogu_r6.patch, applied to the tree's csrc + include first).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p "$ROOT/tools/probes/_build"
/opt/rocm/bin/hipcc -O2 -std=c++17 -x hip --offload-arch=gfx950 -I"$ROOT/include" "$@" "$ROOT/tools/probes/decode_sw_probe.cpp" \
  -L"$ROOT/ml-unigen_amd/csrc" -lunigen_hip -Wl,-rpath,'$ORIGIN/../../../ml-unigen_amd/csrc' -o "$ROOT/tools/probes/_build/decode_sw_probe"
