"""Probe build of the library with extra -D switches on ONE source file:

    python tools/probes/build_variant.py <name> <file.hip> -DSWITCH[=v] ...   ->  tools/probes/_build/libunigen_hip_<name>.so

Run anything against it with UNIGEN_HIP_LIB=<that path> (unigen_hip/lib.py honours it).  Probe switches are compiled out of the
shipped library; the objects of the other sources are the shipped ones (run `make -C ml-unigen_amd/csrc` first)."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(ROOT, "ml-unigen_amd", "csrc")
out = os.path.join(ROOT, "tools", "probes", "_build")
name, target = sys.argv[1], sys.argv[2]
flags = sys.argv[3:]
os.makedirs(out, exist_ok=True)
objs = []
for f in sorted(os.listdir(src)):
    if not f.endswith(".hip"):
        continue
    o = os.path.join(src, f.replace(".hip", ".o"))
    if f == target:
        o = os.path.join(out, f"{name}_{f.replace('.hip', '.o')}")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", *flags,
                               "-I" + src, "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(src, f), "-o", o])
    objs.append(o)
so = os.path.join(out, f"libunigen_hip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-ldl", "-o", so])
os.remove(os.path.join(out, f"{name}_{target.replace('.hip', '.o')}"))
print(so)
