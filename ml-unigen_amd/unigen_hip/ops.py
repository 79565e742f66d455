"""Tensor-level wrappers over the C ABI: torch supplies device memory and the stream, the kernels do
the work.  Every wrapper launches on torch's current HIP stream and never synchronises."""
import ctypes
import math
import torch

from . import lib as _l

UG_EPI_BF16, UG_EPI_F32, UG_EPI_RESID = 0, 1, 2
_MASK_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.int64: 2, torch.bool: 3}


# bench.py sets this to a list to time every GEMM launch with HIP events on the launch stream
GEMM_PROFILE = None
GEMM_PROFILE_FUSED = None     # with GEMM_PROFILE: indices of the launches whose epilogue carries element-wise work (RoPE, SwiGLU forward / backward)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------ handles
# One ug_handle (the library's only allocation: scratch for k-sliced GEMM partials) per (device, stream) that launches a
# GEMM, created on first use outside stream capture.  Under capture no handle is created -- the GEMM then simply does not
# pick its k-sliced forms -- so nothing on the op path ever allocates or synchronises.
_HANDLES = {}
GEMM_POLICY = -1          # -1 auto; tests / tools pin a kernel with set_gemm_tile_policy (passed per call, no library state)
UG_GEMM_NARROW_EPILOGUE, UG_GEMM_POLICY_AUTO_BITS = 0x100, 0xff


def _handle():
    s = torch.cuda.current_stream()
    key = (s.device_index, s.cuda_stream)
    h = _HANDLES.get(key)
    if h is None:
        if torch.cuda.is_current_stream_capturing():
            return 0
        out = ctypes.c_void_p()
        _l.check(_l.load().ug_create(ctypes.byref(out)), "ug_create")
        h = _HANDLES[key] = out.value
    return h


def release_handles():
    """ug_destroy every handle (tests; the caller must have synchronised the streams that used them)."""
    lib = _l.load()
    for h in _HANDLES.values():
        lib.ug_destroy(h)
    _HANDLES.clear()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _l.UniGenHipError("unigen_hip ops need tensors in GPU memory (no CPU fallback exists)")


def round_up(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------ GEMM
def gemm(a, b, out=None, *, M=None, N=None, K=None, a_kmajor=False, b_kmajor=False, epilogue=UG_EPI_BF16,
         bias=None, resid=None, beta=0, alpha_dev=None, out_dtype=None):
    """out[M,N] = opA @ opB^T on the bf16 matrix cores.  Row-major operands are [rows, K] (K contiguous),
    k-major operands are [K, rows]; .stride(0) is the leading dimension either way."""
    _need_cuda(a, b)
    if M is None:
        M = a.shape[1] if a_kmajor else a.shape[0]
    if N is None:
        N = b.shape[1] if b_kmajor else b.shape[0]
    if K is None:
        K = a.shape[0] if a_kmajor else a.shape[1]
    if out is None:
        dt = out_dtype or (torch.bfloat16 if epilogue == UG_EPI_BF16 else torch.float32)
        out = torch.empty((M, N), dtype=dt, device=a.device)
    lib = _l.load()
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib.ug_gemm_bf16(_handle(), _p(a), a.stride(0), int(a_kmajor), _p(b), b.stride(0), int(b_kmajor), _p(out), out.stride(0),
                          M, N, K, epilogue, _p(bias), _p(resid), resid.stride(0) if resid is not None else 0, beta,
                          _p(alpha_dev), GEMM_POLICY, _stream())
    _l.check(rc, "ug_gemm_bf16")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K))
    return out


# On since round 4 (UNIGEN_FUSED_SWIGLU=0: two launches): the 128 ... 320-row kernel keeps gate and up of a hidden unit in one lane
# (permuted weight rows), so the activation is register arithmetic in the epilogue with no exchange between waves: per layer 611 ->
# 561 us for projection + activation, step -0.95 ms, forward + backward -0.8 ... -1.2 ms (two A/B pairs; values bit-identical to the
# two-launch form).  Round 3's form on the 256x256 kernel (partner strips + workgroup barriers) had cost +1.8 ms per step.
FUSED_SWIGLU = __import__("os").environ.get("UNIGEN_FUSED_SWIGLU", "1") == "1"


def gemm_swiglu(x, w_gate_up):
    """x bf16 [M, K], fused weight bf16 [2I, K] (gate rows | up rows) -> (gu bf16 [M, 2I], act bf16 [M, I]): the gate_up
    projection with the SwiGLU activation written by its epilogue (include/unigen_hip.h: ug_gemm_bf16_swiglu)."""
    _need_cuda(x, w_gate_up)
    if not FUSED_SWIGLU or GEMM_POLICY != -1:  # (a pinned tile policy applies to the two-launch form)
        gu = gemm(x, w_gate_up)
        return gu, swiglu_fwd(gu)
    M, K = x.shape
    I = w_gate_up.shape[0] // 2
    gu = torch.empty((M, 2 * I), dtype=torch.bfloat16, device=x.device)
    act = torch.empty((M, I), dtype=torch.bfloat16, device=x.device)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _l.check(_l.load().ug_gemm_bf16_swiglu(_handle(), _p(x), x.stride(0), _p(w_gate_up), w_gate_up.stride(0), _p(gu), gu.stride(0),
                                           _p(act), act.stride(0), M, I, K, _stream()), "ug_gemm_bf16_swiglu")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * 2 * I * K))
        if GEMM_PROFILE_FUSED is not None:
            GEMM_PROFILE_FUSED.append(len(prof) - 1)
    return gu, act


WGRAD_GROUP_MIN_TILES = 256      # below one round of 256x256 tiles the per-problem forms (k-slices) fill the chip better


def gemm_wgrad_group(problems):
    """problems: [(dy bf16 [K_i, rows], x bf16 [K_i, cols], dw fp32 [rows, cols], beta), ...]:
    dw (= / +=) dy^T x for all of them in one launch (include/unigen_hip.h: ug_gemm_bf16_wgrad_group) when together they are at
    least a round of 256x256 tiles, one by one otherwise.  The contraction length K_i may differ between problems."""
    tiles = sum(((dy.shape[1] + 255) // 256) * ((x.shape[1] + 255) // 256) for dy, x, _, _ in problems)
    if tiles < WGRAD_GROUP_MIN_TILES or len(problems) > 8 or GEMM_POLICY != -1:
        for dy, x, dw, beta in problems:
            gemm(dy, x, out=dw, a_kmajor=True, b_kmajor=True, epilogue=UG_EPI_F32, beta=beta)
        return
    n = len(problems)
    PA, LA, IA = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    for dy, x, dw, _ in problems:
        _need_cuda(dy, x, dw)
        assert dy.shape[0] == x.shape[0] and dw.shape == (dy.shape[1], x.shape[1])
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = _l.load().ug_gemm_bf16_wgrad_group(
        n, PA(*[_p(dy) for dy, _, _, _ in problems]), LA(*[dy.stride(0) for dy, _, _, _ in problems]),
        PA(*[_p(x) for _, x, _, _ in problems]), LA(*[x.stride(0) for _, x, _, _ in problems]),
        PA(*[_p(dw) for _, _, dw, _ in problems]), LA(*[dw.stride(0) for _, _, dw, _ in problems]),
        LA(*[dy.shape[1] for dy, _, _, _ in problems]), LA(*[x.shape[1] for _, x, _, _ in problems]),
        IA(*[int(b) for _, _, _, b in problems]), LA(*[dy.shape[0] for dy, _, _, _ in problems]), _stream())
    _l.check(rc, "ug_gemm_bf16_wgrad_group")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, sum(2.0 * dy.shape[1] * x.shape[1] * dy.shape[0] for dy, x, _, _ in problems)))


def set_gemm_tile_policy(policy):
    """Kernel selection passed with every following GEMM call (include/unigen_hip.h: ug_gemm_bf16 `policy`): -1 auto, 0 / 2 =
    128x128 tiles with two / one LDS stages, 3 = staggered 256x256, 10 = 320x256 tiles (where eligible), 32 + h / 16 = the same kernel h = 192 ... 320 rows high, 6 / 8 = k-sliced forms forced; 100 / 101 switch the 256x256
    kernel's LDS-transposed wide epilogue off / on while leaving the selection automatic; 102 / 104 force its one- / two-barrier
    main loop with automatic selection, 103 / 105 with the 256x256 kernel forced (A/B benchmarking and tests)."""
    global GEMM_POLICY
    policy = int(policy)
    if policy == 100:
        GEMM_POLICY = UG_GEMM_POLICY_AUTO_BITS | UG_GEMM_NARROW_EPILOGUE
    elif policy == 101:
        GEMM_POLICY = -1
    elif policy in (102, 104):             # automatic selection, main loop forced to one / two barriers per k-tile
        GEMM_POLICY = UG_GEMM_POLICY_AUTO_BITS | (0x200 if policy == 102 else 0x400)
    elif policy in (103, 105):             # staggered 256x256 kernel forced, one / two barriers
        GEMM_POLICY = 3 | (0x200 if policy == 103 else 0x400)
    else:
        GEMM_POLICY = policy


def set_fused_tile_height(rows):
    """Tile height of the fused-epilogue GEMM launches (gemm_swiglu / gemm_qkv_rope / gemm_swiglu_bwd): 0 = automatic; tests force
    every instantiated height (include/unigen_hip.h: ug_gemm_set_fused_tile_height)."""
    _l.check(_l.load().ug_gemm_set_fused_tile_height(int(rows)), "ug_gemm_set_fused_tile_height")


def gemm_nt(a, b, out=None, **kw):
    """Both operands row-major: out = a[M,K] @ b[N,K]^T."""
    return gemm(a, b, out, **kw)


def cast_bf16(x, out=None):
    _need_cuda(x)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _l.check(_l.load().ug_cast_f32_bf16(_p(x), _p(out), x.numel(), _stream()), "ug_cast_f32_bf16")
    return out


def zero_ranges_(buf, ranges, max_len):
    """Clear the spans {first, count} (int64 device table [n, 2]) of the fp32 buffer `buf` with one launch."""
    _l.check(_l.load().ug_zero_ranges_f32(_p(buf), _p(ranges), ranges.shape[0], int(max_len), _stream()), "ug_zero_ranges_f32")
    return buf


def grad_pack_bf16(g, out, scale):
    """out (bf16) = g (fp32) * scale: staging for the data-parallel bf16 all-reduce (unigen_hip/ddp.py)."""
    _need_cuda(g, out)
    _l.check(_l.load().ug_grad_pack_bf16(_p(g), _p(out), g.numel(), float(scale), _stream()), "ug_grad_pack_bf16")


def grad_unpack_bf16(src, g):
    _need_cuda(src, g)
    _l.check(_l.load().ug_grad_unpack_bf16(_p(src), _p(g), g.numel(), _stream()), "ug_grad_unpack_bf16")


def grad_sum_shards_bf16(shards, world, stride, out, scale):
    """out (bf16 [n]) = bf16(scale * sum over the `world` bf16 shards of n elements, `stride` apart), fp32 sum in rank order."""
    _need_cuda(shards, out)
    _l.check(_l.load().ug_grad_sum_shards_bf16(_p(shards), int(world), int(stride), _p(out), out.numel(), float(scale), _stream()),
             "ug_grad_sum_shards_bf16")


# ------------------------------------------------------------------------------------ row ops
def rmsnorm_fwd(x, w, eps, out_f32=False, want_rstd=True, out=None):
    _need_cuda(x, w)
    rows, cols = x.shape
    y = out if out is not None else torch.empty((rows, cols), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    rstd = torch.empty((rows,), dtype=torch.float32, device=x.device) if want_rstd else None
    _l.check(_l.load().ug_rmsnorm_fwd(_p(x), _p(w), _p(y), _p(rstd), rows, cols, eps, int(out_f32), _stream()),
             "ug_rmsnorm_fwd")
    return y, rstd


def rmsnorm_bwd(dy, x, rstd, w, dres, dw, want_bf16=False):
    """dres (fp32 [rows,cols]) += dx ; dw (fp32 [cols]) += sum dy*xhat.  want_bf16: also return bf16(dres) (the operand of
    the GEMMs that consume the updated gradient), written by the same kernel."""
    rows, cols = x.shape
    out = torch.empty((rows, cols), dtype=torch.bfloat16, device=x.device) if want_bf16 else None
    _l.check(_l.load().ug_rmsnorm_bwd(_p(dy), _p(x), _p(rstd), _p(w), _p(dres), _p(dw), _p(out), rows, cols, _stream()),
             "ug_rmsnorm_bwd")
    return out


def rope_inv_freq(head_dim, theta, scaling=None, seq_len=None):
    """inv_freq of transformers' Qwen2RotaryEmbedding (modeling_rope_utils.py ROPE_INIT_FUNCTIONS): "default";
    "linear" (positions divided by `factor`, i.e. inv_freq / factor); "dynamic" (NTK: the base grows once seq_len exceeds
    max_position_embeddings).  scaling = {"type" | "rope_type", "factor", "max_position_embeddings"} or None -- what the
    reference puts into config.rope_scaling (models/unigen.py:61-64)."""
    base = float(theta)
    kind = (scaling or {}).get("rope_type", (scaling or {}).get("type", "default"))
    factor = float((scaling or {}).get("factor", 1.0))
    if kind == "dynamic":
        max_pos = int((scaling or {}).get("max_position_embeddings", 32768))
        sl = torch.tensor(max(int(seq_len or 0), max_pos))       # (an int64 tensor in transformers: the new base is fp32 arithmetic)
        base = base * ((factor * sl / max_pos) - (factor - 1)) ** (head_dim / (head_dim - 2))
    elif kind not in ("default", "linear"):
        raise _l.UniGenHipError(f"rope_type {kind!r} is not implemented (default, linear and dynamic are)")
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    if kind == "linear":
        inv_freq = inv_freq / factor
    return inv_freq


def rope_tables(L, head_dim, theta, device, scaling=None):
    """cos/sin [L, head_dim/2] fp32, built with the same fp32 ops as transformers'
    Qwen2RotaryEmbedding (modeling_qwen2.py:80-102): inv_freq = 1/theta^(2i/d) (see rope_inv_freq); freqs = inv_freq*pos."""
    inv_freq = rope_inv_freq(head_dim, theta, scaling, L)
    pos = torch.arange(L, dtype=torch.float)
    freqs = (inv_freq[None, :, None] @ pos[None, None, :]).transpose(1, 2)[0]     # [L, d/2]
    return freqs.cos().contiguous().to(device), freqs.sin().contiguous().to(device)


def rope_(qkv, cos, sin, L, nheads, head_dim, backward=False):
    tokens = qkv.shape[0]
    _l.check(_l.load().ug_rope(_p(qkv), _p(cos), _p(sin), tokens, L, qkv.stride(0), nheads, head_dim,
                               int(backward), _stream()), "ug_rope")
    return qkv


def gemm_qkv_rope(x, w, bias, cos, sin, L, nheads, head_dim):
    """x bf16 [M, K], fused q/k/v weight bf16 [N, K] (+ bias bf16 [N]) -> qkv bf16 [M, N] with rotate-half RoPE applied to the first
    nheads * head_dim columns (q and k heads; row m at position m % L): the projection with the rotation in its epilogue
    (include/unigen_hip.h: ug_gemm_bf16_qkv_rope) -- the values of gemm_nt + rope_, bit for bit.  With a pinned tile policy (tests,
    A/B runs) the two launches are used so that the policy applies."""
    _need_cuda(x, w)
    M, K = x.shape
    N = w.shape[0]
    if GEMM_POLICY != -1:
        qkv = gemm(x, w, bias=bias)
        return rope_(qkv, cos, sin, L, nheads, head_dim)
    qkv = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _l.check(_l.load().ug_gemm_bf16_qkv_rope(_handle(), _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(qkv), qkv.stride(0), M, N, K,
                                              _p(cos), _p(sin), L, nheads * head_dim, head_dim, _stream()), "ug_gemm_bf16_qkv_rope")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * N * K))
        if GEMM_PROFILE_FUSED is not None:
            GEMM_PROFILE_FUSED.append(len(prof) - 1)
    return qkv


def swiglu_fwd(gu):
    tokens, two_i = gu.shape
    act = torch.empty((tokens, two_i // 2), dtype=torch.bfloat16, device=gu.device)
    _l.check(_l.load().ug_swiglu_fwd(_p(gu), _p(act), tokens, two_i // 2, _stream()), "ug_swiglu_fwd")
    return act


FUSED_SWIGLU_BWD = __import__("os").environ.get("UNIGEN_FUSED_SWIGLU_BWD", "1") == "1"


def gemm_swiglu_bwd(dy, w_down, gu):
    """dy bf16 [M, K] (gradient of the down projection's output), w_down bf16 [K, I] (the down weight as stored: the dgrad reads it
    k-major), gu bf16 [M, 2I] (the forward's gate | up) -> dgu bf16 [M, 2I]: the down projection's dgrad with the SwiGLU backward in
    its epilogue (include/unigen_hip.h: ug_gemm_bf16_swiglu_bwd) -- the values of gemm(dy, w_down, b_kmajor=True) + swiglu_bwd, bit
    for bit; shapes the fused kernel does not cover, a pinned tile policy and UNIGEN_FUSED_SWIGLU_BWD=0 take the two launches."""
    _need_cuda(dy, w_down, gu)
    M, K = dy.shape
    I = gu.shape[1] // 2
    ok = (FUSED_SWIGLU_BWD and GEMM_POLICY == -1 and I % 256 == 0 and K % 32 == 0 and w_down.shape == (K, I)
          and dy.stride(0) % 8 == 0 and w_down.stride(0) % 8 == 0 and gu.stride(0) % 8 == 0)
    if not ok:
        return swiglu_bwd(gu, gemm(dy, w_down, b_kmajor=True))
    dgu = torch.empty_like(gu)
    prof = GEMM_PROFILE
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _l.check(_l.load().ug_gemm_bf16_swiglu_bwd(_handle(), _p(dy), dy.stride(0), _p(w_down), w_down.stride(0), _p(gu), gu.stride(0),
                                                _p(dgu), dgu.stride(0), M, I, K, _stream()), "ug_gemm_bf16_swiglu_bwd")
    if prof is not None:
        e1.record()
        prof.append((e0, e1, 2.0 * M * I * K))
        if GEMM_PROFILE_FUSED is not None:
            GEMM_PROFILE_FUSED.append(len(prof) - 1)
    return dgu


def swiglu_bwd(gu, dact):
    dgu = torch.empty_like(gu)
    _l.check(_l.load().ug_swiglu_bwd(_p(gu), _p(dact), _p(dgu), gu.shape[0], gu.shape[1] // 2, _stream()),
             "ug_swiglu_bwd")
    return dgu


def embed_fwd(ids, W, err_flag=None):
    tokens = ids.numel()
    V, H = W.shape
    out = torch.empty((tokens, H), dtype=torch.float32, device=W.device)
    _l.check(_l.load().ug_embed_fwd(_p(ids), _p(W), _p(out), tokens, H, V, _p(err_flag), _stream()), "ug_embed_fwd")
    return out


def embed_bwd(ids, dout, dW):
    V, H = dW.shape
    _l.check(_l.load().ug_embed_bwd(_p(ids), _p(dout), _p(dW), ids.numel(), H, V, _stream()), "ug_embed_bwd")


def embed_bwd_sorted(ids_sorted, order, rows, dW, scale):
    """dW[id] += scale * sum of the rows of each run of equal ids, in sorted order (deterministic; ddp.py's lookup exchange)"""
    V, H = dW.shape
    _l.check(_l.load().ug_embed_bwd_sorted(_p(ids_sorted), _p(order), _p(rows), _p(dW), ids_sorted.numel(), H, V, float(scale), _stream()),
             "ug_embed_bwd_sorted")


def gather_rows(x, idx):
    n, C = idx.numel(), x.shape[1]
    out = torch.empty((n, C), dtype=torch.bfloat16, device=x.device)
    _l.check(_l.load().ug_gather_rows_bf16(_p(x), x.stride(0), _p(idx), _p(out), C, n, C, 0, _stream()), "ug_gather_rows_bf16")
    return out


def scatter_rows_(src, idx, out):
    n, C = idx.numel(), src.shape[1]
    _l.check(_l.load().ug_gather_rows_bf16(_p(src), src.stride(0), _p(idx), _p(out), out.stride(0), n, C, 1, _stream()),
             "ug_gather_rows_bf16")
    return out


def colsum_(x, out, R=None, C=None):
    R = x.shape[0] if R is None else R
    C = x.shape[1] if C is None else C
    _l.check(_l.load().ug_colsum_bf16(_p(x), x.stride(0), _p(out), R, C, _stream()), "ug_colsum_bf16")


# ------------------------------------------------------------------------------------ attention
class MaskBits:
    """Compressed attention mask: bits [B, L, nW] uint64 words + tileany [B, nW, nW]."""

    def __init__(self, bits, tileany, B, L):
        self.bits, self.tileany, self.B, self.L = bits, tileany, B, L
        self.nW = (L + 63) // 64
        self.Lp = self.nW * 64


def _alloc_mask(B, L, device):
    nW = (L + 63) // 64
    bits = torch.empty((B, L, nW), dtype=torch.int64, device=device)
    tileany = torch.empty((B, nW, nW), dtype=torch.uint8, device=device)
    return bits, tileany


def mask_compress(mask4d, err_flag=None):
    """mask4d: [B,1,L,L] (or [B,L,L]) additive fp32/bf16/int64 (0 = attend) or bool (True = attend)."""
    _need_cuda(mask4d)
    if mask4d.dim() == 4:
        if mask4d.shape[1] != 1:
            raise _l.UniGenHipError("per-head attention masks are not part of the reference path")
        m = mask4d[:, 0]
    else:
        m = mask4d
    B, L, L2 = m.shape
    if L != L2 or m.stride(2) != 1:
        raise _l.UniGenHipError(f"attention mask must be [B,1,L,L] with contiguous rows, got {tuple(mask4d.shape)}")
    if m.dtype not in _MASK_DTYPES:
        raise _l.UniGenHipError(f"unsupported attention-mask dtype {m.dtype}")
    bits, tileany = _alloc_mask(B, L, m.device)
    _l.check(_l.load().ug_attn_mask_compress(_p(m), _MASK_DTYPES[m.dtype], m.stride(0), m.stride(1), _p(bits),
                                             _p(tileany), B, L, _p(err_flag), _stream()), "ug_attn_mask_compress")
    return MaskBits(bits, tileany, B, L)


def mask_causal(B, L, device, key_valid=None):
    bits, tileany = _alloc_mask(B, L, device)
    kv = None if key_valid is None else key_valid.to(torch.uint8).contiguous()
    _l.check(_l.load().ug_attn_mask_causal(_p(kv), _p(bits), _p(tileany), B, L, _stream()), "ug_attn_mask_causal")
    return MaskBits(bits, tileany, B, L)


MASK_T2I, MASK_LM, MASK_MMU = 0, 1, 2


def mask_from_ids(ids, pad_id, soi_id, eoi_id, mode=MASK_T2I):
    """Compressed attention mask straight from token ids [B, L] (no dense [B,1,L,L] tensor): the reference's
    create_attention_mask_predict_next (mode MASK_T2I = rm_pad_in_image, MASK_LM) / create_attention_mask_for_mmu."""
    _need_cuda(ids)
    B, L = ids.shape
    ids = ids.to(torch.int64).contiguous()
    bits, tileany = _alloc_mask(B, L, ids.device)
    meta = torch.empty((B, 4), dtype=torch.int32, device=ids.device)
    flags = torch.empty((B, L), dtype=torch.uint8, device=ids.device)
    _l.check(_l.load().ug_attn_mask_from_ids(_p(ids), B, L, int(pad_id), int(soi_id), int(eoi_id), int(mode), _p(meta), _p(flags),
                                             _p(bits), _p(tileany), _stream()), "ug_attn_mask_from_ids")
    return MaskBits(bits, tileany, B, L)


def t2i_assemble(text_ids, image_in, image_labels, max_seq_len, pad_id, soi_id, eoi_id, conv_start, conv_end, ignore_id=-100):
    """t2i training rows on the device: text_ids = list of per-sample id lists (or (flat int64 tensor, offsets [B+1]));
    image_in / image_labels int64 [B, n] on the GPU.  -> (input_ids [B, L], attention01 [B, L] uint8, labels [B, L])."""
    dev = image_in.device
    if isinstance(text_ids, tuple):
        flat, offs = text_ids
    else:
        offs = [0]
        for t in text_ids:
            offs.append(offs[-1] + len(t))
        flat = torch.tensor([v for t in text_ids for v in t] or [0], dtype=torch.int64)
        offs = torch.tensor(offs, dtype=torch.int64)
    flat, offs = flat.to(dev), offs.to(dev)
    as_dev = lambda v: v.to(device=dev, dtype=torch.int64) if torch.is_tensor(v) else torch.as_tensor(list(v), dtype=torch.int64).to(dev)
    cs, ce = as_dev(conv_start), as_dev(conv_end)
    B, n = image_in.shape
    ids = torch.empty((B, max_seq_len), dtype=torch.int64, device=dev)
    labels = torch.empty_like(ids)
    attn = torch.empty((B, max_seq_len), dtype=torch.uint8, device=dev)
    _l.check(_l.load().ug_t2i_assemble(_p(flat), _p(offs), _p(cs), cs.numel(), _p(ce), ce.numel(), _p(image_in.to(torch.int64).contiguous()),
                                       _p(image_labels.to(torch.int64).contiguous()), B, n, max_seq_len, int(pad_id), int(soi_id),
                                       int(eoi_id), int(ignore_id), _p(ids), _p(labels), _p(attn), _stream()), "ug_t2i_assemble")
    return ids, attn, labels


def attn_fwd(qkv, mb, H, HKV, hd, scale=None):
    """qkv bf16 [B*L, (H+2*HKV)*hd] (q heads | k heads | v heads, RoPE already applied).  -> (o, lse)"""
    B, L, Lp = mb.B, mb.L, mb.Lp
    scale = 1.0 / math.sqrt(hd) if scale is None else scale
    q = qkv[:, : H * hd]
    k = qkv[:, H * hd: (H + HKV) * hd]
    v = qkv[:, (H + HKV) * hd:]
    o = torch.empty((B * L, H * hd), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    _l.check(_l.load().ug_attn_fwd(_p(q), _p(k), _p(v), qkv.stride(0), _p(o), o.stride(0), _p(lse), _p(mb.bits),
                                   _p(mb.tileany), B, L, Lp, H, HKV, hd, scale, _stream()), "ug_attn_fwd")
    return o, lse


_DKV_WS = {}


def _dkv_workspace(tokens, width, device):
    """zeroed fp32 [tokens, width] scratch for the split-head dK/dV accumulation (the kernels leave it zeroed)"""
    key = (device, tokens, width)
    if key not in _DKV_WS:
        _DKV_WS.clear()
        _DKV_WS[key] = torch.zeros((tokens, width), dtype=torch.float32, device=device)
    return _DKV_WS[key]


def attn_bwd(qkv, o, lse, dout, mb, H, HKV, hd, scale=None, split_heads=True, rope=None, dbias=None):
    """-> dqkv bf16 [B*L, (H+2*HKV)*hd]: the gradient w.r.t. the post-RoPE q / k and v, or -- rope = (cos, sin) tables [L, hd/2] --
    w.r.t. the projection's own output (RoPE transposed where dq / dk are stored).  dbias (fp32 [(H+2*HKV)*hd]): += the column
    sums of dqkv, the bias gradient of the fused projection."""
    B, L, Lp = mb.B, mb.L, mb.Lp
    scale = 1.0 / math.sqrt(hd) if scale is None else scale
    q = qkv[:, : H * hd]
    k = qkv[:, H * hd: (H + HKV) * hd]
    v = qkv[:, (H + HKV) * hd:]
    dqkv = torch.empty_like(qkv)
    dq = dqkv[:, : H * hd]
    dk = dqkv[:, H * hd: (H + HKV) * hd]
    dv = dqkv[:, (H + HKV) * hd:]
    delta = torch.empty((B, H, L), dtype=torch.float32, device=qkv.device)
    ws = _dkv_workspace(B * L, 2 * HKV * hd, qkv.device) if split_heads and H > HKV else None
    _l.check(_l.load().ug_attn_bwd(_p(q), _p(k), _p(v), qkv.stride(0), _p(o), _p(dout), o.stride(0),
                                   _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), dqkv.stride(0), _p(mb.bits),
                                   _p(mb.tileany), B, L, Lp, H, HKV, hd, scale, _p(ws), _p(rope[0] if rope else None),
                                   _p(rope[1] if rope else None), _p(dbias), _stream()), "ug_attn_bwd")
    return dqkv


# ------------------------------------------------------------------------------------ decode
def kv_store(qkv, cache_k, cache_v, rows, L, H, HKV, hd, Tmax, pos_dev=None, pos_host=0):
    _l.check(_l.load().ug_kv_store(_p(qkv), qkv.stride(0), H * hd, (H + HKV) * hd, _p(cache_k), _p(cache_v), rows, L, HKV, hd,
                                   Tmax, _p(pos_dev), pos_host, _stream()), "ug_kv_store")


def rope_at_(qkv, cos, sin, nheads, hd, pos_dev):
    _l.check(_l.load().ug_rope_at(_p(qkv), _p(cos), _p(sin), qkv.shape[0], qkv.stride(0), nheads, hd, _p(pos_dev),
                                  cos.shape[0], _stream()), "ug_rope_at")
    return qkv


def attn_decode(qkv, cache_k, cache_v, key_valid, H, HKV, hd, Tmax, len_dev, scale=None, out=None):
    rows = qkv.shape[0]
    scale = 1.0 / math.sqrt(hd) if scale is None else scale
    o = out if out is not None else torch.empty((rows, H * hd), dtype=torch.bfloat16, device=qkv.device)
    _l.check(_l.load().ug_attn_decode(_p(qkv), qkv.stride(0), _p(cache_k), _p(cache_v), _p(key_valid), _p(o), o.stride(0),
                                      rows, H, HKV, hd, Tmax, _p(len_dev), scale, _stream()), "ug_attn_decode")
    return o


def skinny_linear(x, w, bias=None, resid=None):
    """Decode-time Linear for a handful of rows: split-K fp32 accumulation (fills the chip while the weights
    stream once) + a finishing pass.  resid given -> in-place residual update, else bf16 output."""
    M, N = x.shape[0], w.shape[0]
    acc = torch.zeros((M, N), dtype=torch.float32, device=x.device)
    K = w.shape[1]
    if M <= 32 and K % 32 == 0:
        _l.check(_l.load().ug_gemv_bf16(_p(x), x.stride(0), M, _p(w), w.stride(0), _p(acc), N, 1, N, K, _stream()), "ug_gemv_bf16")
    else:
        gemm(x, w, out=acc, epilogue=UG_EPI_F32, beta=1)
    out = None if resid is not None else torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    _l.check(_l.load().ug_skinny_finish(_p(acc), _p(bias), _p(out), _p(resid), M, N, 0 if resid is None else 1, _stream()),
             "ug_skinny_finish")
    return resid if resid is not None else out


def gemv_acc_(x, w, acc):
    """acc[r, n] += sum_k x[r, k] w[n, k]; acc fp32 [rows, N] row-major (the decode accumulators)."""
    M, N, K = x.shape[0], w.shape[0], w.shape[1]
    _l.check(_l.load().ug_gemv_bf16(_p(x), x.stride(0), M, _p(w), w.stride(0), _p(acc), acc.stride(0), 1, N, K, _stream()),
             "ug_gemv_bf16")
    return acc


def decode_finish_resid_norm_(acc, x, w, xn, eps, advance=None):
    """advance = (pos, len) device int32 scalars of the decode state: incremented by this launch (the step's last reader is done)"""
    pos, ln = advance if advance is not None else (None, None)
    _l.check(_l.load().ug_decode_finish_resid_norm(_p(acc), acc.stride(0), _p(x), _p(w), _p(xn), x.shape[0], x.shape[1], eps,
                                                   _p(pos), _p(ln), _stream()), "ug_decode_finish_resid_norm")
    return xn


def _clr(t):
    return (_p(t), t.numel()) if t is not None else (None, 0)


def decode_gemv_(x, w, acc, zero0=None, zero1=None, ss_zero=None):
    """acc[r, n] += sum_k x[r, k] w[n, k] (bf16 operand) + the clears this launch carries."""
    _l.check(_l.load().ug_decode_gemv(_p(x), x.stride(0), x.shape[0], _p(w), w.stride(0), _p(acc), acc.stride(0), w.shape[0],
                                      w.shape[1], *_clr(zero0), *_clr(zero1), _p(ss_zero), _stream()), "ug_decode_gemv")
    return acc


def decode_gemv_resid_norm_(x_in, pending, norm_w, x_out, ss_out, w, acc, zero0=None, zero1=None, ss_zero=None):
    """Projection whose operand is bf16(norm_w * (x_in + bf16round(pending))); x_out / ss_out receive the updated
    residual stream and its row sums of squares."""
    _l.check(_l.load().ug_decode_gemv_resid_norm(_p(x_in), _p(pending), pending.stride(0), _p(norm_w), _p(x_out), _p(ss_out),
                                                 x_in.shape[0], _p(w), w.stride(0), _p(acc), acc.stride(0), w.shape[0],
                                                 w.shape[1], *_clr(zero0), *_clr(zero1), _p(ss_zero), _stream()),
             "ug_decode_gemv_resid_norm")
    return acc


def decode_gemv_swiglu_(gu_acc, ss_in, eps, norm_cols, w, acc, zero0=None, zero1=None, ss_zero=None):
    """Projection whose operand is SwiGLU of the raw gate/up accumulator scaled by the row's RMSNorm factor."""
    _l.check(_l.load().ug_decode_gemv_swiglu(_p(gu_acc), gu_acc.stride(0), _p(ss_in), eps, norm_cols, gu_acc.shape[0], _p(w),
                                             w.stride(0), _p(acc), acc.stride(0), w.shape[0], w.shape[1], *_clr(zero0),
                                             *_clr(zero1), _p(ss_zero), _stream()), "ug_decode_gemv_swiglu")
    return acc


def decode_sw_supported(hidden, inter, q_dim, head_dim):
    """whether the single-writer decode projections (include/unigen_hip.h: ug_decode_sw_*) are built for these sizes"""
    return bool(_l.load().ug_decode_sw_supported(int(hidden), int(inter), int(q_dim), int(head_dim)))


def decode_sw_resid_(x, w, h):
    """h[r, n] += float(bf16(sum_k x[r, k] w[n, k])) in place, one writer per element (o projection of a decode step)"""
    _l.check(_l.load().ug_decode_sw_resid(_p(x), x.stride(0), x.shape[0], _p(w), w.stride(0), w.shape[0], w.shape[1], _p(h), _stream()),
             "ug_decode_sw_resid")
    return h


def decode_sw_kblock_(x, w, acc, zero0=None, zero1=None, ss_zero=None):
    """acc[r, n] += sum_k x[r, k] w[n, k] in k-blocks of 1 792 (one atomic per element and k-block) + the clears this launch carries"""
    _l.check(_l.load().ug_decode_sw_kblock(_p(x), x.stride(0), x.shape[0], _p(w), w.stride(0), _p(acc), acc.stride(0), w.shape[0],
                                           w.shape[1], *_clr(zero0), *_clr(zero1), _p(ss_zero), _stream()), "ug_decode_sw_kblock")
    return acc


def decode_sw_gate_up_(h, norm_w, eps, w, act, pend=None, x_out=None):
    """act = bf16(bf16(silu(gate)) * up) with gate | up = Linear(RMSNorm(h [+ bf16round(pend)])); w = [2 I, H], gate rows first"""
    _l.check(_l.load().ug_decode_sw_gate_up(_p(h), _p(pend), pend.stride(0) if pend is not None else 0, _p(x_out), _p(norm_w), eps,
                                            h.shape[0], h.shape[1], _p(w), w.stride(0), w.shape[0] // 2, _p(act), act.stride(0), _stream()),
             "ug_decode_sw_gate_up")
    return act


def decode_sw_head_(h, norm_w, eps, w, logits, pend=None, x_out=None, advance=None):
    """logits fp32 [R, N] = Linear(RMSNorm(h [+ bf16round(pend)])) for the N rows `w` of the tied embedding; advance = (pos, len)"""
    pos, ln = advance if advance is not None else (None, None)
    _l.check(_l.load().ug_decode_sw_head(_p(h), _p(pend), pend.stride(0) if pend is not None else 0, _p(x_out), _p(norm_w), eps,
                                         h.shape[0], h.shape[1], _p(w), w.stride(0), w.shape[0], _p(logits), logits.stride(0),
                                         _p(pos), _p(ln), _stream()), "ug_decode_sw_head")
    return logits


def attn_decode_fused(acc_qkv, ss_in, eps, norm_cols, bias, cos, sin, pos_dev, cache_k, cache_v, key_valid, out, H, HKV, hd, Tmax,
                      scale=None):
    scale = 1.0 / math.sqrt(hd) if scale is None else scale
    _l.check(_l.load().ug_attn_decode_fused(_p(acc_qkv), acc_qkv.stride(0), _p(ss_in), eps, norm_cols, _p(bias), _p(cos), _p(sin),
                                            _p(pos_dev), _p(cache_k), _p(cache_v), _p(key_valid), _p(out), out.stride(0),
                                            acc_qkv.shape[0], H, HKV, hd, Tmax, cos.shape[0], scale, _stream()),
             "ug_attn_decode_fused")
    return out


# ------------------------------------------------------------------------------------ MaskGIT sampler
def maskgit_step(logits, N, n, cfg, guidance_scale, u_sample, u_conf, cur_ids, mask_id, id_offset, mask_len_sched, temperature,
                 want_masking=False):
    """One parallel-decoding round on bf16 code-book logits [(2 if cfg else 1)*N*n, V(+pad)] -> (sampled, next_cur,
    next_ids[, masking]) int64 [N, n]; see include/unigen_hip.h."""
    V = logits.shape[-1]
    logits = logits.reshape(-1, V)
    if logits.dtype != torch.bfloat16 or logits.stride(1) != 1:
        raise _l.UniGenHipError("maskgit_step: logits must be bf16 with contiguous rows")
    dev = logits.device
    sampled = torch.empty((N, n), dtype=torch.int64, device=dev)
    next_cur, next_ids = torch.empty_like(sampled), torch.empty_like(sampled)
    sel = torch.empty((N, n), dtype=torch.float32, device=dev)
    masking = torch.empty((N, n), dtype=torch.uint8, device=dev) if want_masking else None
    cur_ids = cur_ids.to(torch.int64).contiguous()
    _l.check(_l.load().ug_maskgit_step(_p(logits), logits.stride(0), V, N, n, int(bool(cfg)), float(guidance_scale),
                                       _p(u_sample.contiguous()), _p(u_conf.contiguous()), _p(cur_ids), int(mask_id),
                                       int(id_offset), int(mask_len_sched), float(temperature), _p(sampled), _p(sel), _p(next_cur),
                                       _p(next_ids), _p(masking), _stream()), "ug_maskgit_step")
    return (sampled, next_cur, next_ids, masking.bool()) if want_masking else (sampled, next_cur, next_ids)


def maskgit_train_mask(tokens, scores, num_masked, mask_id, ignore_id=-100):
    """tokens int64 [B, n], scores fp32 [B, n], num_masked fp32 [B] -> (input_ids, labels) int64 [B, n]; see
    include/unigen_hip.h: ug_maskgit_train_mask."""
    _need_cuda(tokens, scores, num_masked)
    B, n = tokens.shape
    tokens = tokens.to(torch.int64).contiguous()
    ids, labels = torch.empty_like(tokens), torch.empty_like(tokens)
    _l.check(_l.load().ug_maskgit_train_mask(_p(tokens), _p(scores.float().contiguous()), _p(num_masked.float().contiguous()), B, n,
                                             int(mask_id), int(ignore_id), _p(ids), _p(labels), _stream()), "ug_maskgit_train_mask")
    return ids, labels


def ar_sample_(acc, bsz, V, guidance_scale, temperature, greedy, uniforms, pos_dev, pos0, nsteps, embed_master, id_offset, tok,
               out_tokens, x):
    """Fused AR sampling step on the raw lm-head accumulator (see include/unigen_hip.h: ug_ar_sample)."""
    _l.check(_l.load().ug_ar_sample(_p(acc), acc.stride(0), bsz, V, float(guidance_scale), float(temperature), int(bool(greedy)),
                                    _p(uniforms), _p(pos_dev), int(pos0), int(nsteps), _p(embed_master), embed_master.stride(0),
                                    embed_master.shape[1], int(id_offset), _p(tok), _p(out_tokens), _p(x), _stream()),
             "ug_ar_sample")


# ------------------------------------------------------------------------------------ loss
def ce_fwd(logits, V, labels, ignore_index=-100, want_logp=False):
    """logits bf16 [R, ld>=V]; -> (loss_and_count [2], lse [R], loss_row [R], logp|None)"""
    R, ld = logits.shape[0], logits.stride(0)
    dev = logits.device
    lse = torch.empty((R,), dtype=torch.float32, device=dev)
    loss_row = torch.empty((R,), dtype=torch.float32, device=dev)
    logp = torch.empty((R,), dtype=torch.float32, device=dev) if want_logp else None
    lc = torch.empty((2,), dtype=torch.float32, device=dev)
    _l.check(_l.load().ug_ce_fwd(_p(logits), ld, R, V, _p(labels), ignore_index, _p(lse), _p(loss_row), _p(logp),
                                 _p(lc), _stream()), "ug_ce_fwd")
    return lc, lse, loss_row, logp


def ce_bwd_(logits, V, labels, lse, lc, gscale=None, ignore_index=-100, row_scale=None):
    R, ld = logits.shape[0], logits.stride(0)
    _l.check(_l.load().ug_ce_bwd(_p(logits), ld, R, V, _p(labels), ignore_index, _p(lse), _p(lc), _p(gscale),
                                 _p(row_scale), _stream()), "ug_ce_bwd")
    return logits


# flat fp32 master buffers that keep a bf16 compute mirror (FlatParams registers itself); FusedAdamW writes the
# mirror in the same pass so no separate cast pass follows an optimizer step
BF16_MIRRORS = []


def register_bf16_mirror(owner):
    """owner: object with .master (fp32 flat), .bf16 (same layout) and ._seen_version"""
    import weakref
    BF16_MIRRORS.append(weakref.ref(owner))


def find_bf16_mirror(ptr, numel):
    """-> (owner, bf16 address of the element at fp32 address `ptr`) if [ptr, ptr+4*numel) lies in a registered master"""
    for ref in list(BF16_MIRRORS):
        o = ref()
        if o is None:
            BF16_MIRRORS.remove(ref)
            continue
        base = o.master.data_ptr()
        if base <= ptr and ptr + 4 * numel <= base + 4 * o.master.numel():
            return o, o.bf16.data_ptr() + (ptr - base) // 2
    return None, 0


# ------------------------------------------------------------------------------------ optimizer
def adamw_flat_(p, g, m, v, p_bf16, lr, beta1, beta2, eps, wd, step, grad_scale=1.0, max_blocks=0):
    """max_blocks > 0: the small-grid, register-lean form used beside MFMA-bound kernels (include/unigen_hip.h: ug_adamw_flat)."""
    _l.check(_l.load().ug_adamw_flat(_p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), lr, beta1, beta2, eps, wd,
                                     step, grad_scale, max_blocks, _stream()), "ug_adamw_flat")


# ------------------------------------------------------------------------------------ tokenizer (fp32 NHWC)
def pack_conv_weight(w):
    """torch Conv2d weight [Cout, Cin, k, k] -> ([k*k, Cin, cout_pad] fp32, cout_pad). Done once at load."""
    cout, cin, kh, kw = w.shape
    cout_pad = round_up(cout, 128) if cout > 32 else 32
    wp = torch.zeros((kh * kw, cin, cout_pad), dtype=torch.float32, device=w.device)
    wp[:, :, :cout] = w.detach().float().permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
    return wp.contiguous(), cout_pad


def conv_split_eligible(cin, cout, cout_pad):
    """Shapes the split-f16 convolution (`ug_conv2d_split`) is used for (the kernel itself only needs Cin % 4 == 0)."""
    return cin % 32 == 0 and cout % 4 == 0 and cout_pad % 128 == 0 and cout >= 64


def split_conv_weight(wp):
    """Packed fp32 weights [taps, Cin, cout_pad] -> the two-plane scaled fp16 tile image `ug_conv2d_split` reads (+ the
    tensor's max|w| in the 8 trailing elements)."""
    taps, cin, cout_pad = wp.shape
    ws = torch.empty(2 * taps * round_up(cin, 32) * cout_pad + 8, dtype=torch.float16, device=wp.device)
    _l.check(_l.load().ug_conv_split_weights(_p(wp), _p(ws), taps, cin, cout_pad, _stream()), "ug_conv_split_weights")
    return ws


_AMAX_CONST = {}


def amax_const(value, device):
    """A cached device scalar holding a fixed upper bound of max|x| (only its binade matters to the split convolutions)."""
    key = (float(value), str(device))
    if key not in _AMAX_CONST:
        _AMAX_CONST[key] = torch.full((1,), float(value), dtype=torch.float32, device=device)
    return _AMAX_CONST[key]


def amax(x2d):
    """max|x| of a 2-D (or any contiguous) fp32 tensor as a device scalar: the scale bound of `conv2d_nhwc(w_split=)` /
    `linear_split` for inputs whose range is not known by construction."""
    _need_cuda(x2d)
    if x2d.dim() == 2 and x2d.stride(1) == 1:
        rows, cols, ld = x2d.shape[0], x2d.shape[1], x2d.stride(0)
    else:
        x2d = x2d.contiguous()
        rows, cols, ld = 1, x2d.numel(), x2d.numel()
    if torch.cuda.is_current_stream_capturing():
        # a pool slot is zeroed ONCE, when its pool is created: replays of a captured graph would keep taking the maximum into a
        # slot nobody clears (the bound would only ever grow).  Under capture the bound gets its own scalar and the launch that
        # carries a memset node (ug_amax_f32), so every replay starts from zero.
        out = torch.empty(1, dtype=torch.float32, device=x2d.device)
        _l.check(_l.load().ug_amax_f32(_p(x2d), rows, cols, ld, _p(out), _stream()), "ug_amax_f32")
        return out
    out = _amax_slot(x2d.device)
    _l.check(_l.load().ug_amax_f32_into_zeroed(_p(x2d), rows, cols, ld, _p(out), _stream()), "ug_amax_f32")
    return out


_AMAX_POOL = {}


def _amax_slot(device):
    """A zeroed fp32 scalar from a pool cleared 1024 slots at a time (one fill per 1024 bounds instead of a memset per bound); a slot is
    handed out once -- the returned view keeps its pool alive for as long as the bound is in use."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    pool = _AMAX_POOL.get(key)
    if pool is None or pool[1] >= pool[0].numel():
        pool = [torch.zeros(1024, dtype=torch.float32, device=device), 0]      # (on the current stream: ordered before its users)
        _AMAX_POOL[key] = pool
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1]


def conv2d_nhwc(x, wp, cout_pad, bias, cout, ksize, *, stride=1, pad=None, residual=None, upsample=False,
                asym_pad=False, w_split=None, x_amax=None, out_stats=None, out_groups=32):
    """x [B,H,W,Cin] fp32 NHWC -> [B,Ho,Wo,Cout].  asym_pad: the reference Downsample's pad (0,1,0,1) +
    stride-2 valid conv (common_modules.py:86-93).  With `w_split` (from `split_conv_weight`) the contraction runs as
    three f16 MFMA terms of the scaled two-way split operands instead of on the fp32 MFMA; `x_amax` (device scalar, an
    upper bound of max|x|) sets the activation scale, measured here when not given.  out_stats (split path, output
    pixels per image % 128 == 0): as `conv3x3_nhwc`."""
    B, H, W, Cin = x.shape
    He, We = (2 * H, 2 * W) if upsample else (H, W)
    if asym_pad:
        pt = pl = 0
        Ho, Wo = (He + 1 - ksize) // stride + 1, (We + 1 - ksize) // stride + 1
    else:
        pt = pl = (ksize // 2) if pad is None else pad
        Ho, Wo = (He + 2 * pt - ksize) // stride + 1, (We + 2 * pl - ksize) // stride + 1
    y = torch.empty((B, Ho, Wo, cout), dtype=torch.float32, device=x.device)
    if w_split is not None:
        if x_amax is None:
            x_amax = amax(x.view(-1, Cin))
        if out_stats is not None and (out_stats.dtype != torch.float64 or out_stats.numel() != B * out_groups * 2 + 1):
            raise _l.UniGenHipError("conv2d_nhwc: out_stats must be a zeroed fp64 buffer of B * groups * 2 + 1 elements (gn_stats_slots)")
        _l.check(_l.load().ug_conv2d_split(_p(x), _p(x_amax), _p(w_split), _p(bias), _p(residual), _p(y), B, H, W, Cin, cout,
                                           cout_pad, ksize, stride, pt, pl, Ho, Wo, int(upsample), _p(out_stats),
                                           int(out_groups) if out_stats is not None else 0, _stream()), "ug_conv2d_split")
        return y
    if out_stats is not None:
        raise _l.UniGenHipError("conv2d_nhwc: out_stats needs the split path (w_split)")
    _l.check(_l.load().ug_conv2d_f32(_p(x), _p(wp), _p(bias), _p(residual), _p(y), B, H, W, Cin, cout, cout_pad, ksize,
                                     stride, pt, pl, Ho, Wo, int(upsample), _stream()), "ug_conv2d_f32")
    return y


def gemm_f32(a, b, *, b_is_nk, M, N, K, batch=1, lda=None, ldb=None, stride_a=0, stride_b=0, alpha=1.0, out=None,
             ldc=None, stride_c=None):
    """Batched fp32 GEMM on the f32 matrix cores: C_z[M,N] = alpha * A_z[M,K] @ (B_z as [N,K] if b_is_nk else [K,N]);
    operand z starts stride_* elements after operand z-1 (heads of a fused projection are z * head_dim apart)."""
    if out is None:
        out = torch.empty((batch, M, N), dtype=torch.float32, device=a.device)
    ldc = N if ldc is None else ldc
    stride_c = M * N if stride_c is None else stride_c
    _l.check(_l.load().ug_gemm_f32(_p(a), lda, stride_a, _p(b), ldb, stride_b, int(b_is_nk), _p(out), ldc, stride_c, M, N,
                                   K, batch, alpha, _stream()), "ug_gemm_f32")
    return out


def gemm_f32_nested(a, b, out, *, b_is_nk, M, N, K, batch_in, batch_out, lda, ldb, ldc, sa, sb, sc, alpha=1.0):
    """`gemm_f32` over a two-level batch (inner, outer); sa / sb / sc = (inner stride, outer stride) in elements."""
    _l.check(_l.load().ug_gemm_f32_nested(_p(a), lda, sa[0], sa[1], _p(b), ldb, sb[0], sb[1], int(b_is_nk), _p(out), ldc,
                                          sc[0], sc[1], M, N, K, batch_in, batch_out, alpha, _stream()), "ug_gemm_f32_nested")
    return out


def groupnorm_swish(x, gamma, beta, *, groups=32, eps=1e-6, swish=True):
    """x NHWC [B,H,W,C] fp32."""
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    ws = torch.empty((B, groups, 2), dtype=torch.float64, device=x.device)
    _l.check(_l.load().ug_groupnorm_swish(_p(x), _p(gamma), _p(beta), _p(y), _p(ws), B, H * W, C, groups, eps,
                                          int(swish), _stream()), "ug_groupnorm_swish")
    return y


def groupnorm_stats(x, *, groups=32, eps=1e-6):
    """(mean, rstd) of GroupNorm(groups) over NHWC fp32 x, [B, groups, 2] fp32 -- what `conv3x3_nhwc(..., gn=)` applies
    on its load path."""
    B, H, W, C = x.shape
    ws = torch.empty((B, groups, 2), dtype=torch.float64, device=x.device)
    mr = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    _l.check(_l.load().ug_groupnorm_stats(_p(x), _p(ws), _p(mr), B, H * W, C, groups, eps, _stream()),
             "ug_groupnorm_stats")
    return mr


def gn_out_bound(gamma_absmax, beta_absmax, group_elems):
    """Upper bound of |gamma * xhat + beta| for a GroupNorm output, from the layer's ACTUAL affine parameters: a normalised
    group of n elements cannot exceed sqrt(n - 1) in magnitude (one outlier carrying all of the variance), swish only shrinks
    magnitudes.  Rounded up to a power of two (only the binade matters to the split convolutions, and few distinct device
    constants are cached).  The caller computes it once per weight version (models/multimodal_encoder/magvitv2.py)."""
    b = float(gamma_absmax) * math.sqrt(max(float(group_elems) - 1.0, 1.0)) + float(beta_absmax)
    return 2.0 ** math.ceil(math.log2(max(b, 1e-30)))


def gn_stats_slots(n, B, device, groups=32):
    """n zeroed buffers for `conv3x3_nhwc / conv2d_nhwc(..., out_stats=)` out of ONE allocation (one clear for a whole encoder
    pass): flat fp64 [B * groups * 2 + 1] each -- the sums per (image, group) and the trailing max|y| slot."""
    per = B * groups * 2 + 1
    pool = torch.zeros(n * per, dtype=torch.float64, device=device)
    return [pool[i * per:(i + 1) * per] for i in range(n)]


def stats_amax(stats):
    """the fp32 max|y| a convolution left in the trailing slot of its `out_stats` buffer (a 1-element device tensor)"""
    return stats[-1:].view(torch.float32)[:1]


def groupnorm_finalize(stats, B, HW, C, *, groups=32, eps=1e-6):
    """(mean, rstd) [B, groups, 2] fp32 from the fp64 sums a convolution gathered while storing its output
    (`conv3x3_nhwc(..., out_stats=)`): `groupnorm_stats` without the pass over the tensor."""
    if stats.numel() != B * groups * 2 + 1:
        raise _l.UniGenHipError("groupnorm_finalize: stats buffer does not match B x groups")
    mr = torch.empty((B, groups, 2), dtype=torch.float32, device=stats.device)
    _l.check(_l.load().ug_groupnorm_finalize(_p(stats), _p(mr), B, HW, C, groups, eps, _stream()), "ug_groupnorm_finalize")
    return mr


def conv3x3_nhwc(x, w_split, cout_pad, bias, cout, *, residual=None, gn=None, x_amax=None, gn_bound=None, out_stats=None,
                 out_groups=32):
    """3x3 / stride 1 / pad 1 convolution of NHWC fp32 x with split weights (`split_conv_weight`), input patch resident
    in LDS.  out_stats (a ZEROED buffer from `gn_stats_slots`): the epilogue also gathers the fp64 (sum, sum of squares) of the
    output per (image, group) and max|y| into it -- input of `groupnorm_finalize` for the GroupNorm that reads y, and of
    `stats_amax` for a split convolution that reads y directly.
    gn = (mu_rstd, gamma, beta, groups, swish): apply swish?(GroupNorm(x)) on the load path; the scale bound of the
    normalised tensor, `gn_bound`, is then known without a pass over it: `gn_out_bound` of the layer's gamma / beta (required
    with gn -- a fixed constant would silently saturate activations of a checkpoint with larger affine parameters).  Without
    gn the bound is measured unless `x_amax` is given."""
    B, H, W, Cin = x.shape
    y = torch.empty((B, H, W, cout), dtype=torch.float32, device=x.device)
    mr, ga, be, groups, swish = gn if gn is not None else (None, None, None, 0, 0)
    if x_amax is None:
        if gn is not None and gn_bound is None:
            raise _l.UniGenHipError("conv3x3_nhwc: GroupNorm on the load path needs gn_bound (ops.gn_out_bound of the layer's gamma / beta)")
        x_amax = amax_const(gn_bound, x.device) if gn is not None else amax(x.view(-1, Cin))
    if out_stats is not None and (out_stats.dtype != torch.float64 or out_stats.numel() != B * out_groups * 2 + 1):
        raise _l.UniGenHipError("conv3x3_nhwc: out_stats must be a zeroed fp64 buffer of B * groups * 2 + 1 elements (gn_stats_slots)")
    _l.check(_l.load().ug_conv3x3_split(_p(x), _p(x_amax), _p(w_split), _p(bias), _p(residual), _p(y), B, H, W, Cin, cout,
                                        cout_pad, _p(mr), _p(ga), _p(be), groups, int(swish), _p(out_stats),
                                        int(out_groups) if out_stats is not None else 0, _stream()), "ug_conv3x3_split")
    return y


def softmax_rows_(x2d, scale, cols=None):
    """in-place row softmax of scale*x over the first `cols` columns of a 2-D fp32 tensor (row stride = .stride(0))."""
    cols = x2d.shape[1] if cols is None else cols
    _l.check(_l.load().ug_softmax_rows_f32(_p(x2d), x2d.shape[0], cols, x2d.stride(0), scale, _stream()),
             "ug_softmax_rows_f32")
    return x2d


def linear_f32(x, W, bias=None, residual=None, act=0, out=None, M=None):
    """fp32 y = act(x @ W^T + bias) + residual on the f32 matrix cores; x [M,K] (row stride .stride(0)), W [N,K]."""
    M = x.shape[0] if M is None else M
    N, K = W.shape
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _l.check(_l.load().ug_linear_f32(_p(x), x.stride(0), _p(W), W.stride(0), _p(bias), _p(residual),
                                     residual.stride(0) if residual is not None else 0, _p(out), out.stride(0), M, N, K, act,
                                     _stream()), "ug_linear_f32")
    return out


def split_linear_weight(W):
    """nn.Linear weight [N, K] fp32 -> (split tile image of W^T as a 1x1 conv, n_pad) for `linear_split`."""
    N, K = W.shape
    n_pad = round_up(N, 128)
    wp = torch.zeros((1, K, n_pad), dtype=torch.float32, device=W.device)
    wp[0, :, :N] = W.detach().float().t()
    return split_conv_weight(wp), n_pad


def linear_split(x, w_split, n_pad, N, bias=None, residual=None, act=0, out=None, M=None, x_amax=None):
    """`linear_f32` with the scaled two-way f16 split contraction (fp32-accurate, f16 matrix cores); x [M,K] fp32.
    x_amax: device scalar bounding max|x| (measured here when not given)."""
    M = x.shape[0] if M is None else M
    K = x.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    if x_amax is None:
        x_amax = amax(x[:M])
    _l.check(_l.load().ug_linear_split(_p(x), x.stride(0), _p(x_amax), _p(w_split), _p(bias), _p(residual),
                                       residual.stride(0) if residual is not None else 0, _p(out), out.stride(0), M, N, K,
                                       n_pad, act, _stream()), "ug_linear_split")
    return out


def siglip_attn(qkv, out, B, T, H, hd, scale, qkv_amax=None):
    """Fused fp32-accurate attention of the SigLIP tower: qkv fp32 [>= B*T, 3*H*hd] (q | k | v), out fp32 [B*T, H*hd];
    see include/unigen_hip.h: ug_siglip_attn_f32.  qkv_amax: device scalar bounding max|qkv| (measured here if not given)."""
    _need_cuda(qkv, out)
    if qkv_amax is None:
        qkv_amax = amax(qkv[:B * T])
    _l.check(_l.load().ug_siglip_attn_f32(_p(qkv), qkv.stride(0), _p(qkv_amax), _p(out), out.stride(0), B, T, H, hd, float(scale),
                                          _stream()), "ug_siglip_attn_f32")
    return out


def layernorm_f32(x2d, gamma, beta, eps):
    y = torch.empty_like(x2d)
    _l.check(_l.load().ug_layernorm_f32(_p(x2d), _p(gamma), _p(beta), _p(y), x2d.shape[0], x2d.shape[1], eps, _stream()),
             "ug_layernorm_f32")
    return y


def layernorm_bwd_f32(dy, x, gamma, eps, dgamma, dbeta, dres_in=None):
    """dx = [dres_in +] LayerNorm'(dy); dgamma / dbeta (fp32 [cols]) accumulate."""
    dx = torch.empty_like(x)
    _l.check(_l.load().ug_layernorm_bwd_f32(_p(dy), _p(x), _p(gamma), _p(dres_in), _p(dx), _p(dgamma), _p(dbeta), x.shape[0], x.shape[1],
                                            eps, _stream()), "ug_layernorm_bwd_f32")
    return dx


def gelu_tanh_f32(pre, dy=None):
    """gelu_pytorch_tanh(pre), or dy * gelu'(pre) when dy is given; fp32."""
    out = torch.empty_like(pre)
    _l.check(_l.load().ug_gelu_tanh_f32(_p(pre), _p(dy), _p(out), pre.numel(), _stream()), "ug_gelu_tanh_f32")
    return out


def softmax_bwd_rows_(P2d, dP2d, scale, cols):
    """in place on dP2d: dS = scale * P * (dP - rowsum(dP * P)) over the first `cols` columns, padding columns zeroed"""
    _l.check(_l.load().ug_softmax_bwd_rows_f32(_p(P2d), _p(dP2d), P2d.shape[0], cols, P2d.stride(0), scale, _stream()),
             "ug_softmax_bwd_rows_f32")
    return dP2d


def colsum_f32_(x2d, out):
    """out[c] += sum_r x2d[r, c] (fp32)"""
    _l.check(_l.load().ug_colsum_f32(_p(x2d), x2d.stride(0), _p(out), x2d.shape[0], x2d.shape[1], _stream()), "ug_colsum_f32")
    return out


def transpose_f32(x, rows=None, cols=None, *, batch=1, ld_in=None, stride_in=0, pad_to=4):
    """out[z][c][r] = x[z][r][c]; output rows are round_up(rows, pad_to) long with a zero tail.  x: 2-D [rows, cols] (batch 1) or
    any buffer described by (ld_in, stride_in).  -> [batch, cols, ld_out]."""
    rows = x.shape[-2] if rows is None else rows
    cols = x.shape[-1] if cols is None else cols
    ld_in = x.stride(-2) if ld_in is None else ld_in
    ld_out = round_up(rows, pad_to)
    out = torch.empty((batch, cols, ld_out), dtype=torch.float32, device=x.device)
    _l.check(_l.load().ug_transpose_f32(_p(x), ld_in, stride_in, _p(out), ld_out, cols * ld_out, rows, cols, batch, _stream()),
             "ug_transpose_f32")
    return out


def nchw_to_nhwc(x, c_pad):
    B, C, H, W = x.shape
    out = torch.empty((B, H, W, c_pad), dtype=torch.float32, device=x.device)
    _l.check(_l.load().ug_nchw_to_nhwc(_p(x), _p(out), B, C, H * W, c_pad, _stream()), "ug_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x, C):
    B, H, W, Cp = x.shape
    out = torch.empty((B, C, H, W), dtype=torch.float32, device=x.device)
    _l.check(_l.load().ug_nhwc_to_nchw(_p(x), _p(out), B, C, H * W, Cp, _stream()), "ug_nhwc_to_nchw")
    return out


def lfq_pack(z2d, nbits):
    n = z2d.shape[0]
    idx = torch.empty((n,), dtype=torch.int64, device=z2d.device)
    _l.check(_l.load().ug_lfq_pack(_p(z2d), z2d.stride(0), _p(idx), n, nbits, _stream()), "ug_lfq_pack")
    return idx


def lfq_unpack(idx, nbits, err_flag=None):
    n = idx.numel()
    z = torch.empty((n, nbits), dtype=torch.float32, device=idx.device)
    _l.check(_l.load().ug_lfq_unpack(_p(idx), _p(z), n, nbits, _p(err_flag), _stream()), "ug_lfq_unpack")
    return z


def probe_layouts(device):
    out = torch.zeros((512,), dtype=torch.float32, device=device)
    _l.check(_l.load().ug_probe_layouts(_p(out), 512, _stream()), "ug_probe_layouts")
    return out


def gelu(x, dy=None):
    """bf16 GELU(erf): forward if dy is None, else the gradient dgelu(x) * dy."""
    out = torch.empty_like(x)
    _l.check(_l.load().ug_gelu(_p(x), _p(dy), _p(out), x.numel(), _stream()), "ug_gelu")
    return out
