"""ctypes binding of libunigen_hip.so (the C ABI declared in include/unigen_hip.h).

The product path has no CPU fallback: if the shared library is missing the import of any op fails
loudly with the build command.  Nothing under oracle/ is ever imported from here.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.normpath(os.path.join(_HERE, "..", "csrc"))
LIB_PATH = os.environ.get("UNIGEN_HIP_LIB") or os.path.join(CSRC_DIR, "libunigen_hip.so")     # (probe builds: tools/probes/_build/*.so)

ABI_VERSION = 7
P = ctypes.c_void_p
I64 = ctypes.c_int64
I32 = ctypes.c_int
F32 = ctypes.c_float

# name -> argtypes (must mirror include/unigen_hip.h exactly; tests/test_abi.py cross-checks the
# exported symbol list against the header)
SIGNATURES = {
    "ug_abi_version": [],
    "ug_create": [P],
    "ug_destroy": [P],
    "ug_gemm_bf16": [P, P, I64, I32, P, I64, I32, P, I64, I64, I64, I64, I32, P, P, I64, I32, P, I32, P],
    "ug_gemm_bf16_qkv_rope": [P, P, I64, P, I64, P, P, I64, I64, I64, I64, P, P, I64, I64, I32, P],
    "ug_gemm_bf16_swiglu_bwd": [P, P, I64, P, I64, P, I64, P, I64, I64, I64, I64, P],
    "ug_gemm_bf16_swiglu": [P, P, I64, P, I64, P, I64, P, I64, I64, I64, I64, P],
    "ug_cast_f32_bf16": [P, P, I64, P],
    "ug_rmsnorm_fwd": [P, P, P, P, I64, I64, F32, I32, P],
    "ug_rmsnorm_bwd": [P, P, P, P, P, P, P, I64, I64, P],
    "ug_rope": [P, P, P, I64, I64, I64, I32, I32, I32, P],
    "ug_swiglu_fwd": [P, P, I64, I64, P],
    "ug_swiglu_bwd": [P, P, P, I64, I64, P],
    "ug_gelu": [P, P, P, I64, P],
    "ug_embed_fwd": [P, P, P, I64, I64, I64, P, P],
    "ug_embed_bwd": [P, P, P, I64, I64, I64, P],
    "ug_embed_bwd_sorted": [P, P, P, P, I64, I64, I64, F32, P],
    "ug_gather_rows_bf16": [P, I64, P, P, I64, I64, I64, I32, P],
    "ug_colsum_bf16": [P, I64, P, I64, I64, P],
    "ug_attn_mask_compress": [P, I32, I64, I64, P, P, I64, I64, P, P],
    "ug_attn_mask_causal": [P, P, P, I64, I64, P],
    "ug_attn_fwd": [P, P, P, I64, P, I64, P, P, P, I64, I64, I64, I32, I32, I32, F32, P],
    "ug_attn_bwd": [P, P, P, I64, P, P, I64, P, P, P, P, P, I64, P, P, I64, I64, I64, I32, I32, I32, F32, P, P, P, P, P],
    "ug_kv_store": [P, I64, I64, I64, P, P, I64, I64, I32, I32, I64, P, I32, P],
    "ug_rope_at": [P, P, P, I64, I64, I32, I32, P, I64, P],
    "ug_attn_decode": [P, I64, P, P, P, P, I64, I64, I32, I32, I32, I64, P, F32, P],
    "ug_gemv_bf16": [P, I64, I64, P, I64, P, I64, I64, I64, I64, P],
    "ug_decode_gemv": [P, I64, I64, P, I64, P, I64, I64, I64, P, I64, P, I64, P, P],
    "ug_decode_gemv_resid_norm": [P, P, I64, P, P, P, I64, P, I64, P, I64, I64, I64, P, I64, P, I64, P, P],
    "ug_decode_gemv_swiglu": [P, I64, P, F32, I64, I64, P, I64, P, I64, I64, I64, P, I64, P, I64, P, P],
    "ug_gemm_set_fused_tile_height": [I32],
    "ug_attn_decode_fused": [P, I64, P, F32, I64, P, P, P, P, P, P, P, P, I64, I64, I32, I32, I32, I64, I64, F32, P],
    "ug_decode_finish_resid_norm": [P, I64, P, P, P, I64, I64, F32, P, P, P],
    "ug_decode_sw_supported": [I64, I64, I64, I32],
    "ug_decode_sw_kblock": [P, I64, I64, P, I64, P, I64, I64, I64, P, I64, P, I64, P, P],
    "ug_decode_sw_resid": [P, I64, I64, P, I64, I64, I64, P, P],
    "ug_decode_sw_gate_up": [P, P, I64, P, P, F32, I64, I64, P, I64, I64, P, I64, P],
    "ug_decode_sw_head": [P, P, I64, P, P, F32, I64, I64, P, I64, I64, P, I64, P, P, P],
    "ug_t2i_assemble": [P, P, P, I64, P, I64, P, P, I64, I64, I64, I64, I64, I64, I64, P, P, P, P],
    "ug_attn_mask_from_ids": [P, I64, I64, I64, I64, I64, I32, P, P, P, P, P],
    "ug_maskgit_train_mask": [P, P, P, I64, I64, I64, I64, P, P, P],
    "ug_ar_sample": [P, I64, I64, I64, F32, F32, I32, P, P, I64, I64, P, I64, I64, I64, P, P, P, P],
    "ug_maskgit_step": [P, I64, I64, I64, I64, I32, F32, P, P, P, I64, I64, I64, F32, P, P, P, P, P, P],
    "ug_skinny_finish": [P, P, P, P, I64, I64, I32, P],
    "ug_ce_fwd": [P, I64, I64, I64, P, I64, P, P, P, P, P],
    "ug_ce_bwd": [P, I64, I64, I64, P, I64, P, P, P, P, P],
    "ug_adamw_flat": [P, P, P, P, P, I64, F32, F32, F32, F32, F32, I64, F32, I32, P],
    "ug_gemm_bf16_wgrad_group": [I32, P, P, P, P, P, P, P, P, P, P, P],
    "ug_grad_pack_bf16": [P, P, I64, F32, P],
    "ug_comm_unique_id": [P],
    "ug_comm_init": [P, I32, I32, P, I64],
    "ug_comm_allreduce_bucket": [P, P, I64, I32, P],
    "ug_comm_allgather": [P, P, P, I64, P],
    "ug_comm_wait": [P, P],
    "ug_comm_destroy": [P],
    "ug_comm_bytes_on_wire": [P],
    "ug_zero_ranges_f32": [P, P, I64, I64, P],
    "ug_grad_unpack_bf16": [P, P, I64, P],
    "ug_grad_sum_shards_bf16": [P, I32, I64, P, I64, F32, P],
    "ug_conv2d_f32": [P, P, P, P, P, I64, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, P],
    "ug_conv_split_weights": [P, P, I32, I32, I32, P],
    "ug_amax_f32": [P, I64, I64, I64, P, P],
    "ug_amax_f32_into_zeroed": [P, I64, I64, I64, P, P],
    "ug_conv2d_split": [P, P, P, P, P, P, I64, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, P, I32, P],
    "ug_conv3x3_split": [P, P, P, P, P, P, I64, I32, I32, I32, I32, I32, P, P, P, I32, I32, P, I32, P],
    "ug_groupnorm_finalize": [P, P, I64, I64, I32, I32, F32, P],
    "ug_groupnorm_stats": [P, P, P, I64, I64, I32, I32, F32, P],
    "ug_siglip_attn_f32": [P, I64, P, P, I64, I64, I64, I32, I32, F32, P],
    "ug_linear_split": [P, I64, P, P, P, P, I64, P, I64, I64, I64, I64, I32, I32, P],
    "ug_gemm_f32_nested": [P, I64, I64, I64, P, I64, I64, I64, I32, P, I64, I64, I64, I64, I64, I64, I64, I64, F32, P],
    "ug_gemm_f32": [P, I64, I64, P, I64, I64, I32, P, I64, I64, I64, I64, I64, I64, F32, P],
    "ug_groupnorm_swish": [P, P, P, P, P, I64, I64, I32, I32, F32, I32, P],
    "ug_softmax_rows_f32": [P, I64, I64, I64, F32, P],
    "ug_linear_f32": [P, I64, P, I64, P, P, I64, P, I64, I64, I64, I64, I32, P],
    "ug_layernorm_f32": [P, P, P, P, I64, I64, F32, P],
    "ug_layernorm_bwd_f32": [P, P, P, P, P, P, P, I64, I64, F32, P],
    "ug_gelu_tanh_f32": [P, P, P, I64, P],
    "ug_softmax_bwd_rows_f32": [P, P, I64, I64, I64, F32, P],
    "ug_colsum_f32": [P, I64, P, I64, I64, P],
    "ug_transpose_f32": [P, I64, I64, P, I64, I64, I64, I64, I64, P],
    "ug_nchw_to_nhwc": [P, P, I64, I32, I64, I32, P],
    "ug_nhwc_to_nchw": [P, P, I64, I32, I64, I32, P],
    "ug_lfq_pack": [P, I64, P, I64, I32, P],
    "ug_lfq_unpack": [P, P, I64, I32, P, P],
    "ug_probe_layouts": [P, I64, P],
}

_lib = None

# Launch-time breakdown by kernel family (bench.py's `roofline_by_family`): when PROFILE is a dict, every entry point that
# launches on the caller's stream is bracketed by a pair of HIP events recorded on torch's current stream -- the stream every
# op passes down (or the explicit stream of an overlapped launch, e.g. the optimizer update) -- and filed under its family.  None (the default) costs one global read per call.
PROFILE = None


def family_of(name):
    if name.startswith("ug_gemm_bf16"):
        return "gemm"
    if name in ("ug_attn_fwd", "ug_attn_bwd"):
        return "attention"
    if name.startswith(("ug_conv", "ug_groupnorm", "ug_amax", "ug_lfq", "ug_nchw", "ug_nhwc", "ug_gemm_f32", "ug_softmax_rows", "ug_linear_",
                        "ug_layernorm", "ug_siglip")):
        return "tokenizer_and_towers"
    if name == "ug_adamw_flat":
        return "adamw"
    return "elementwise"


def _profiled(name, fn):
    fam = family_of(name)

    def call(*args):
        prof = PROFILE
        if prof is None:
            return fn(*args)
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sp = args[-1]                                  # every launching entry point takes its stream last
        sp = getattr(sp, "value", sp) or 0
        st = torch.cuda.current_stream() if sp == torch.cuda.current_stream().cuda_stream else torch.cuda.ExternalStream(sp)
        e0.record(st)
        rc = fn(*args)
        e1.record(st)
        tag = name
        if name == "ug_gemm_bf16":                     # (handle, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, epilogue, ...)
            tag = "ug_gemm_bf16[M=%d N=%d K=%d ak=%d bk=%d epi=%d]" % (args[9], args[10], args[11], args[3], args[6], args[12])
        elif name == "ug_conv3x3_split":               # (x, x_amax, w_split, bias, residual, y, B, H, W, Cin, Cout, ...)
            tag = "ug_conv3x3_split[B=%d %dx%d %d->%d gn=%d]" % (args[6], args[7], args[8], args[9], args[10], 1 if args[12] else 0)
        elif name == "ug_gemm_bf16_wgrad_group":       # (n, dy, ld_dy, x, ld_x, dw, ld_dw, rows, cols, beta, K, stream)
            tag = "ug_gemm_bf16_wgrad_group[%s]" % " ".join("%dx%dx%d" % (args[7][i], args[8][i], args[10][i]) for i in range(args[0]))
        prof.setdefault(fam, []).append((e0, e1, tag))
        return rc
    call.__name__ = name
    return call


class UniGenHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise UniGenHipError("building libunigen_hip.so failed (see output above)")
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UniGenHipError(
            f"{LIB_PATH} is missing: the HIP extension is the only implementation of this path "
            f"(no CPU fallback). Build it with `make -C {CSRC_DIR}` or `python -c 'import __graft_entry__ as g; g.build()'`.")
    # torch ships its own libamdhip64.so.7; whichever copy of that soname is mapped first serves the whole process.
    # Loading ours first maps /opt/rocm's runtime, and a later `import torch` then runs on a HIP/HSA mix that finds no
    # device -- so torch's runtime always goes first (the host side of this library is PyTorch anyway).
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    lib.ug_last_error.restype = ctypes.c_char_p
    lib.ug_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch: fail loudly
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64 if name == "ug_comm_bytes_on_wire" else ctypes.c_int
        if name.startswith("ug_comm_") or name in ("ug_abi_version", "ug_create", "ug_destroy"):
            continue
        setattr(lib, name, _profiled(name, fn))
    if lib.ug_abi_version() != ABI_VERSION:
        raise UniGenHipError(f"ABI version mismatch: library reports {lib.ug_abi_version()}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, name="ug call"):
    if rc != 0:
        msg = load().ug_last_error().decode("utf-8", "replace")
        raise UniGenHipError(f"{name} failed (rc={rc}): {msg}")
