"""Host side of the C ABI under AddressSanitizer, no GPU needed (aux: "one -fsanitize=address CPU run of the
argument-check paths").  `make -C ml-unigen_amd/csrc asan` builds the library with the HOST code instrumented (device code is
unaffected: GPU ASan needs xnack+ code objects); a child interpreter with the ASan runtime preloaded then

  1. calls EVERY entry point declared in include/unigen_hip.h with all-zero arguments (null pointers, zero sizes): each must come
     back with a non-zero status and an error string -- none may touch memory before its checks;
  2. calls a set of entry points with well-formed shapes and host buffers standing in for device memory: validation passes, the
     host-side launch preparation runs (argument structs, host tables of the grouped weight-gradient launch, grid selection) and
     the launch itself fails cleanly on the missing device.

Any invalid host access in those paths aborts the child with an ASan report and fails the test."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ml-unigen_amd", "csrc")
ASAN_LIB = os.path.join(CSRC, "_asan", "libunigen_hip_asan.so")

CHILD = r'''
import ctypes, importlib.util, sys
import numpy as np
root, so = sys.argv[1], sys.argv[2]
spec = importlib.util.spec_from_file_location("ug_lib_sigs", root + "/ml-unigen_amd/unigen_hip/lib.py")   # the table only: no torch here
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
lib = ctypes.CDLL(so)
lib.ug_last_error.restype = ctypes.c_char_p
assert lib.ug_abi_version() == m.ABI_VERSION
zero = {m.P: None, m.I64: 0, m.I32: 0, m.F32: 0.0}
# entry points for which "nothing" is a legal argument (documented no-ops) or that need a live runtime / communicator
NOOP_OK = {"ug_abi_version", "ug_destroy", "ug_comm_destroy", "ug_gemm_set_fused_tile_height",     # (height 0 = automatic)
           "ug_decode_sw_supported"}                                                                    # (a predicate: 0 = "not built for these sizes")
SKIP = {"ug_create", "ug_comm_unique_id", "ug_comm_init"}
n_checked = 0
for name, argtypes in m.SIGNATURES.items():
    if name in SKIP:
        continue
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = ctypes.c_int64 if name == "ug_comm_bytes_on_wire" else ctypes.c_int
    rc = fn(*[zero[t] for t in argtypes])
    if name in NOOP_OK or name == "ug_comm_bytes_on_wire":
        continue
    msg = lib.ug_last_error().decode("utf-8", "replace")
    assert rc != 0 and msg, (name, rc, msg)
    n_checked += 1
print("zero-argument calls rejected:", n_checked)

# ---- well-formed calls: host buffers stand in for device memory (nothing dereferences them on the host; no device -> the launch fails)
def buf(nbytes):
    a = np.zeros(nbytes + 64, dtype=np.uint8)
    p = (a.ctypes.data + 15) & ~15
    return a, ctypes.c_void_p(p)
keep = []
def P(nbytes):
    a, p = buf(nbytes); keep.append(a); return p
def call(name, *args):
    fn = getattr(lib, name)
    rc = fn(*args)
    msg = lib.ug_last_error().decode("utf-8", "replace")
    assert rc == -2, (name, rc, msg)             # UG_ERR_LAUNCH: the argument checks passed, the launch found no device
    return rc
M, H = 64, 1536
call("ug_rmsnorm_fwd", P(M * H * 4), P(H * 4), P(M * H * 2), P(M * 4), M, H, 1e-6, 0, None)
call("ug_rmsnorm_bwd", P(M * H * 2), P(M * H * 4), P(M * 4), P(H * 4), P(M * H * 4), P(H * 4), P(M * H * 2), M, H, None)
call("ug_swiglu_fwd", P(M * 512 * 2), P(M * 256 * 2), M, 256, None)
call("ug_adamw_flat", P(4096 * 4), P(4096 * 4), P(4096 * 4), P(4096 * 4), P(4096 * 2), 4096, 1e-4, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, 256, None)
call("ug_gemm_bf16", None, P(M * H * 2), H, 0, P(256 * H * 2), H, 0, P(M * 256 * 2), 256, M, 256, H, 0, None, None, 0, 0, None, -1, None)
# grouped weight gradients: HOST arrays of per-problem pointers and sizes
n = 2
PA, LA, IA = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
dy, x, dw = PA(P(M * 256 * 2), P(M * 512 * 2)), PA(P(M * H * 2), P(M * H * 2)), PA(P(256 * H * 4), P(512 * H * 4))
call("ug_gemm_bf16_wgrad_group", n, dy, LA(256, 512), x, LA(H, H), dw, LA(H, H), LA(256, 512), LA(H, H), IA(0, 1), LA(M, M), None)
B, L, Hq, Hk = 2, 128, 4, 2
ldq = (Hq + 2 * Hk) * 128
qkv = P(B * L * ldq * 2)
call("ug_attn_fwd", qkv, qkv, qkv, ldq, P(B * L * Hq * 128 * 2), Hq * 128, P(B * Hq * L * 4), P(B * L * 2 * 8), P(B * 2 * 2), B, L, 128, Hq, Hk, 128, 0.088, None)
call("ug_conv3x3_split", P(2 * 16 * 16 * 32 * 4), P(4), P(9 * 1 * 1 * 8192 * 2 + 16), P(128 * 4), None, P(2 * 16 * 16 * 128 * 4), 2, 16, 16, 32, 128, 128,
     None, None, None, 0, 0, P((2 * 32 * 2 + 1) * 8), 32, None)
call("ug_groupnorm_finalize", P(2 * 32 * 2 * 8), P(2 * 32 * 2 * 4), 2, 256, 128, 32, 1e-6, None)
call("ug_zero_ranges_f32", P(4096 * 4), P(4 * 8), 2, 1024, None)
print("well-formed calls failed cleanly on the missing device")
'''


def _asan_runtime():
    for clang in ("/opt/rocm/lib/llvm/bin/clang", "clang"):
        try:
            out = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, timeout=30).stdout.strip()
        except (OSError, subprocess.TimeoutExpired):
            continue
        if out and os.path.isabs(out) and os.path.exists(out):
            return out
    return None


def test_every_entry_point_checks_its_arguments_under_host_asan():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("clang's ASan runtime not found")
    res = subprocess.run(["make", "-C", CSRC, "asan", "-j8"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:exitcode=23:abort_on_error=0", HIP_VISIBLE_DEVICES="")
    child = subprocess.run([sys.executable, "-c", CHILD, ROOT, ASAN_LIB], capture_output=True, text=True, env=env, timeout=600)
    print(child.stdout[-2000:])
    assert child.returncode == 0, (child.returncode, child.stdout[-2000:], child.stderr[-6000:])
    assert "zero-argument calls rejected" in child.stdout and "failed cleanly" in child.stdout
