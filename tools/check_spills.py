"""Register-spill check of the built gfx950 code objects (the fused-epilogue GEMM instantiations carry hand-counted
`s_waitcnt vmcnt` around inline-asm loads -- a scratch spill ahead of such a wait would silently corrupt values).

    python tools/check_spills.py [--all]      exit 1 if a kernel that must not spill does

Reads `.vgpr_spill_count` / `.sgpr_spill_count` / `.vgpr_count` from the code-object notes of every object under
ml-unigen_amd/csrc (the objects `make` left there).  Must-not-spill: every `gemm_kernel_*`, every attention kernel and every
decode kernel (the kernels with manual wait counts); `--all` extends the rule to every kernel.  Called by __graft_entry__.build()."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ml-unigen_amd", "csrc")


def _llvm_dir():
    """The LLVM tools that belong to the hipcc the Makefile uses (HIPCC / ROCM_PATH overrides honoured), not a fixed path."""
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
    cands = []
    try:
        out = subprocess.run([hipcc, "--print-prog-name=llvm-readelf"], capture_output=True, text=True, timeout=60).stdout.strip()
        if out and os.path.isabs(out):
            cands.append(os.path.dirname(out))
    except (OSError, subprocess.SubprocessError):
        pass
    rocm = os.environ.get("ROCM_PATH") or os.path.dirname(os.path.dirname(os.path.realpath(hipcc)))
    cands += [os.path.join(rocm, "lib", "llvm", "bin"), os.path.join(rocm, "llvm", "bin"), "/opt/rocm/lib/llvm/bin"]
    for c in cands:
        if os.path.exists(os.path.join(c, "llvm-readelf")) and os.path.exists(os.path.join(c, "llvm-objdump")):
            return c
    raise SystemExit(f"check_spills: llvm-readelf / llvm-objdump not found (looked in {cands}); set ROCM_PATH or HIPCC")


LLVM = _llvm_dir()
STRICT = re.compile(r"gemm_kernel|attn_|gemv_|conv3x3|conv_igemm")
# Audited spills (round 5, ISA read: `scratch_` lines of the device assembly): name fragment -> (max spilled dwords, what spills).
# None of them is an inline-asm load destination or sits between hand-counted loads and their `s_waitcnt vmcnt`: a scratch
# reload only makes the compiler's own wait stricter.  A count above the audited one fails the build check again.
AUDITED = {
    "gemm_kernelILi0ELb0ELb0ELb0EEE": (5, "128x128 kernel at its 128-VGPR bound: five loop-invariant 32-bit addresses, stored once, reloaded per k-tile"),
    "gemm_kernelILi1ELb0ELb0ELb0EEE": (5, "same"),
    "gemm_kernelILi2ELb0ELb0ELb0EEE": (5, "same"),
    "gemm_kernel_p10ILi0ELb1ELi10ELi10EEE": (4, "320-row dgrad tile (160 accumulators): one 64-bit operand pointer per main-loop form, reloaded at the loop head"),
    "gemm_kernel_p10ILi0ELb1ELi10ELi9EEE": (2, "304-row dgrad tile: one 64-bit operand pointer, reloaded at the loop head"),
    "gemm_kernel_p10ILi5ELb1ELi10ELi10EEE": (6, "320-row SwiGLU-backward tile: three 64-bit pointers (operand, gu row base); reloads precede the epilogue's asm loads"),
}


def kernels_of(obj):
    """[(name, vgprs, vgpr_spills, sgpr_spills)] of one host object with an embedded gfx950 bundle"""
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        cos = [os.path.join(td, f) for f in os.listdir(td) if "amdgcn" in f]
        out = []
        for co in cos:
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            # the notes carry the code object's metadata as a YAML document: amdhsa.kernels is the list of kernel records
            m = re.search(r"^\s*---\s*$(.*?)^\s*\.\.\.\s*$", notes, re.S | re.M)
            if not m:
                raise SystemExit(f"check_spills: no metadata document in the notes of {os.path.basename(obj)}")
            for k in (yaml.safe_load(m.group(1)) or {}).get("amdhsa.kernels", []):
                missing = [f for f in (".name", ".vgpr_count", ".vgpr_spill_count", ".sgpr_spill_count") if f not in k]
                if missing:
                    raise SystemExit(f"check_spills: kernel record in {os.path.basename(obj)} lacks {missing}: {k.get('.name')}")
                out.append((k[".name"], int(k[".vgpr_count"]), int(k[".vgpr_spill_count"]), int(k[".sgpr_spill_count"])))
        return out


def main():
    strict_all = "--all" in sys.argv
    bad, total = [], 0
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".o"):
            continue
        for name, vg, vs, ss in kernels_of(os.path.join(CSRC, f)):
            total += 1
            audited = next((v for k, v in AUDITED.items() if k in name), None)
            if vs and audited and vs <= audited[0] and not strict_all:
                print(f"audited: {f}: {name}: {vs} spilled dwords ({audited[1]})")
            elif vs and (strict_all or STRICT.search(name)):
                bad.append((f, name, vg, vs, ss))
            elif vs:
                print(f"note: {f}: {name}: {vs} VGPR spills ({vg} VGPRs) -- not in the must-not-spill set")
    for f, name, vg, vs, ss in bad:
        print(f"SPILL: {f}: {name}: vgpr_spill_count {vs} (vgprs {vg}, sgpr spills {ss})")
    print(f"check_spills: {total} kernels, {len(bad)} spilling in the must-not-spill set")
    if total == 0:
        print("check_spills: no kernel found under ml-unigen_amd/csrc (objects not built, or the metadata could not be read): that is a failure, not a pass")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
