"""Probe build of the library with extra -D switches on ONE source file:

    python tools/probes/build_variant.py <name> <file.hip> -DSWITCH[=v] ...   ->  tools/probes/_build/libunigen_hip_<name>.so

The probe switches (ablations, traces, alternative cache policies / issue orders -- several compile to WRONG values by design) are not
part of the product source: tools/probes/probe_switches.patch re-inserts them into a scratch copy of ml-unigen_amd/csrc, and the
target file is compiled from that copy.  Run anything against the result with UNIGEN_HIP_LIB=<that path> (unigen_hip/lib.py honours it).
The objects of the other sources are the shipped ones (run `make -C ml-unigen_amd/csrc` first)."""
import os
import shutil
import subprocess
import sys
import tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "ml-unigen_amd", "csrc")
out = os.path.join(ROOT, "tools", "probes", "_build")
name, target = sys.argv[1], sys.argv[2]
flags = sys.argv[3:]
if target in ("decode.hip", "decode_sw.hip"):                 # as the Makefile builds them (kernarg preload)
    flags = ["-mllvm", "-amdgpu-kernarg-preload-count=16"] + flags
os.makedirs(out, exist_ok=True)
with tempfile.TemporaryDirectory() as td:
    scratch = os.path.join(td, "ml-unigen_amd", "csrc")
    os.makedirs(scratch)
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")):
            shutil.copy(os.path.join(src, f), scratch)
    os.makedirs(os.path.join(td, "tools"))
    os.symlink(os.path.join(ROOT, "tools", "probes"), os.path.join(td, "tools", "probes"))       # (UG_GEMM_R4 includes ../../tools/probes/*.inc)
    subprocess.check_call(["patch", "-p1", "-s", "-i", os.path.join(ROOT, "tools", "probes", "probe_switches.patch")], cwd=td)
    objs = []
    for f in sorted(os.listdir(src)):
        if not f.endswith(".hip"):
            continue
        o = os.path.join(src, f.replace(".hip", ".o"))
        if f == target:
            o = os.path.join(out, f"{name}_{f.replace('.hip', '.o')}")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", *flags,
                                   "-I" + scratch, "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(scratch, f), "-o", o])
        objs.append(o)
    so = os.path.join(out, f"libunigen_hip_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", so])
    os.remove(os.path.join(out, f"{name}_{target.replace('.hip', '.o')}"))
print(so)
