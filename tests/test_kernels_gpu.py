"""Kernel-level parity: every HIP entry point against the CPU oracle (oracle/ops_ref.py) on seeded
inputs.  All tests call through the C ABI (ctypes) -- there is no other implementation."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from unigen_hip import ops
    return ops


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _maxabs(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


# ------------------------------------------------------------------ probes (recorded, not asserted beyond basics)
def test_probe_layouts(dev):
    ops = _ops()
    out = ops.probe_layouts(dev).cpu()
    c = out[256:512].view(64, 4)
    # D[i][j] = i*(j+1) ; lane holds row (lane>>4)*4 + r, col lane&15
    for lane in (0, 5, 17, 63):
        for r in range(4):
            i, j = (lane >> 4) * 4 + r, lane & 15
            assert c[lane, r].item() == i * (j + 1)
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/tr16_b64_layout.txt", "w") as f:
        for lane in range(64):
            f.write(f"{lane}: {[int(v) for v in out[lane*4:lane*4+4]]}\n")


# ------------------------------------------------------------------ GEMM
@pytest.fixture(params=[0, 2, 103, 105], ids=["two_lds_stages", "one_lds_stage", "staggered_256x256_one_barrier", "staggered_256x256_two_barriers"])
def tile_policy(request):
    ops = _ops()
    ops.set_gemm_tile_policy(request.param)
    yield request.param
    ops.set_gemm_tile_policy(-1)


@pytest.mark.parametrize("height", [128, 144, 160, 176, 192, 208, 224, 240, 272, 288, 304, 320])
@pytest.mark.parametrize("mode", ["nt", "dgrad"])
def test_gemm_bf16_tile_heights(dev, height, mode):
    """Round 4: the 128 ... 320 x 256 tile family (`gemm_kernel_p10<EPI, BKM, F0, F1>`; policy 32 + height / 16 forces one): bf16 (+ bias)
    and fp32-residual epilogues, B row-major and k-major, M ragged against every height (row tiles that end inside group 0,
    inside group 1, on an odd 16-row block).  And the automatic choice: the smallest height that keeps a one-round launch in one
    round gives the same values as the forced one."""
    ops = _ops()
    M, N, K = 2 * height + 16 * 7 + 5, 512, 256
    bk = mode == "dgrad"
    g = torch.Generator().manual_seed(height + (7 if bk else 0))
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    b = torch.randn(N, K, generator=g).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(torch.bfloat16)
    ref = a.float() @ b.float().t()
    A = a.to(dev)
    B = (b.t().contiguous() if bk else b).to(dev)
    try:
        ops.set_gemm_tile_policy(32 + height // 16)
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(A, B, out=out, M=M, N=N, K=K, b_kmajor=bk, bias=bias.to(dev))
        assert _rel(out, ref + bias.float()) < 4e-3
        if not bk:
            res = torch.randn(M, N, generator=g)
            r32 = torch.empty(M, N, dtype=torch.float32, device=dev)
            ops.gemm(A, B, out=r32, M=M, N=N, K=K, epilogue=ops.UG_EPI_RESID, resid=res.to(dev))
            want = res + ref.to(torch.bfloat16).float()
            assert ((r32.cpu() - want).abs() <= ref.abs() * 2.0 ** -7 + 1e-3).all()
    finally:
        ops.set_gemm_tile_policy(-1)
    # automatic: 12 336 x 1 536 is a one-round launch at 304 rows; same bits as the forced height (same tiles, same k order)
    if height == 304 and not bk:
        M2, N2, K2 = 12336, 1536, 256
        a2 = torch.randn(M2, K2, generator=g).to(torch.bfloat16).to(dev)
        b2 = torch.randn(N2, K2, generator=g).to(torch.bfloat16).to(dev)
        auto = ops.gemm(a2, b2)
        ops.set_gemm_tile_policy(32 + 304 // 16)
        try:
            forced = ops.gemm(a2, b2)
        finally:
            ops.set_gemm_tile_policy(-1)
        assert torch.equal(auto, forced) and _rel(auto, a2.float() @ b2.float().t()) < 4e-3


@pytest.mark.parametrize("mode", ["nt", "dgrad", "wgrad"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 200, 192), (771, 1536, 1536), (64, 336, 256),
                                   (130, 72, 100), (1536, 512, 771)])
def test_gemm_bf16_layouts(dev, tile_policy, mode, M, N, K):
    """forward (both row-major), dgrad (B k-major), wgrad (both k-major); ragged M/N/K edges."""
    ops = _ops()
    ak, bk = {"nt": (False, False), "dgrad": (False, True), "wgrad": (True, True)}[mode]
    if mode == "nt" and K % 8:
        K = K // 8 * 8
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    b = torch.randn(N, K, generator=g).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(torch.bfloat16)
    ref = a.float() @ b.float().t()

    def store(x, kmajor):       # [rows, K] logical -> device tensor in the requested storage order, padded ld
        rows, kk = x.shape
        if kmajor:
            buf = torch.full((kk, (rows + 7) // 8 * 8 + 8), 3.0, dtype=torch.bfloat16)   # finite garbage in the padding
            buf[:, :rows] = x.t()
        else:
            buf = torch.zeros((rows, (kk + 7) // 8 * 8), dtype=torch.bfloat16)             # K tail must be zero (contract)
            buf[:, :kk] = x
        return buf.to(dev)
    A, B = store(a, ak), store(b, bk)
    ldc = (N + 7) // 8 * 8
    cbuf = torch.zeros(M, ldc, dtype=torch.bfloat16, device=dev)
    ops.gemm(A, B, out=cbuf, M=M, N=N, K=K, a_kmajor=ak, b_kmajor=bk, bias=bias.to(dev))
    assert _rel(cbuf[:, :N], ref + bias.float()) < 4e-3
    if ldc > N:
        assert cbuf[:, N:].abs().max().item() == 0      # guarded stores never touch padding
    c32 = torch.ones(M, ldc, dtype=torch.float32, device=dev)
    ops.gemm(A, B, out=c32, M=M, N=N, K=K, a_kmajor=ak, b_kmajor=bk, epilogue=ops.UG_EPI_F32, beta=1)
    assert _rel(c32[:, :N], ref + 1.0) < 1e-5 * math.sqrt(K) + 1e-6
    if mode == "nt":
        res = torch.randn(M, ldc, generator=g)
        r32 = torch.empty(M, ldc, dtype=torch.float32, device=dev)
        ops.gemm(A, B, out=r32, M=M, N=N, K=K, epilogue=ops.UG_EPI_RESID, resid=res.to(dev))
        want = res[:, :N] + ref.to(torch.bfloat16).float()
        # one bf16 ulp of slack on the rounded projection
        assert ((r32[:, :N].cpu() - want).abs() <= ref.abs() * 2.0 ** -7 + 1e-3).all()


@pytest.mark.parametrize("M,L,N,rope_heads,K", [(3 * 257, 257, 2048, 14, 1536), (2 * 771, 771, 2048, 14, 256), (5 * 40, 40, 768, 4, 64),
                                                 (4 * 333, 333, 512, 2, 96), (2 * 100, 100, 2048, 14, 1536), (771 * 16, 771, 2048, 14, 128),
                                                 (64, 64, 384, 2, 64), (120, 60, 2048, 14, 72)])
def test_gemm_qkv_rope_is_gemm_then_rope(dev, M, L, N, rope_heads, K):
    """The fused q/k/v projection (`ug_gemm_bf16_qkv_rope`: RoPE in the GEMM's epilogue, round 4) against `ug_gemm_bf16` followed by
    `ug_rope`: the same bits -- every tile height the launcher can pick (M from 64 to 12 336 rows), ragged last row tiles, positions
    that wrap inside a tile (L = 40), roped and un-roped column tiles, and shapes that take the two-launch fallback (N % 256 != 0,
    K % 32 != 0)."""
    ops = _ops()
    hd = 128
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.2).to(torch.bfloat16).to(dev)
    b = torch.randn(N, generator=g).to(torch.bfloat16).to(dev)
    cos, sin = ops.rope_tables(L, hd, 1e6, dev)
    want = ops.rope_(ops.gemm(x, w, bias=b), cos, sin, L, rope_heads, hd)
    got = ops.gemm_qkv_rope(x, w, b, cos, sin, L, rope_heads, hd)
    assert torch.equal(got, want)
    ref = x.float().cpu() @ w.float().cpu().t() + b.float().cpu()                   # and the un-roped columns against the host
    assert _rel(got[:, rope_heads * hd:].float().cpu(), ref[:, rope_heads * hd:]) < 4e-3 or rope_heads * hd == N
    nob = ops.gemm_qkv_rope(x, w, None, cos, sin, L, rope_heads, hd)                 # no bias
    assert torch.equal(nob, ops.rope_(ops.gemm(x, w), cos, sin, L, rope_heads, hd))


@pytest.mark.parametrize("M,I,K", [(12336, 8960, 1536), (9288, 8960, 64), (771, 512, 256), (1542, 1024, 96), (200, 2048, 32), (4000, 1000, 128),
                                   (3084, 768, 160), (64, 256, 40)])
def test_gemm_swiglu_bwd_fused_epilogue_is_bit_identical(dev, M, I, K):
    """`ug_gemm_bf16_swiglu_bwd` (the down projection's dgrad whose epilogue turns d(act) into d(gate | up), round 4) against
    gemm(dy, W_down k-major) + `ug_swiglu_bwd`: bit-identical d(gate | up) -- fused shapes over several tile heights with ragged
    last row tiles, and the shapes that take the two launches (I % 256 != 0, K % 32 != 0); and against fp32 autograd on the host."""
    ops = _ops()
    g = torch.Generator(device=dev).manual_seed(M + I + K)
    dy = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    wd = (torch.randn(K, I, device=dev, generator=g) * (K ** -0.5)).to(torch.bfloat16)
    gu = (torch.randn(M, 2 * I, device=dev, generator=g) * 2).to(torch.bfloat16)
    gu[0, :8] = torch.tensor([-200.0, -90.0, 90.0, 200.0, 0.0, -0.0, 1e-30, -1e-30], device=dev).to(torch.bfloat16)     # exp overflow / zeros
    ops.set_gemm_tile_policy(3)                      # the reference path on the 256 x 256 kernel
    try:
        want = ops.swiglu_bwd(gu, ops.gemm(dy, wd, b_kmajor=True))
    finally:
        ops.set_gemm_tile_policy(-1)
    got = ops.gemm_swiglu_bwd(dy, wd, gu)
    assert got.shape == (M, 2 * I) and torch.equal(got, want) and torch.isfinite(got.float()).all()
    if M <= 1542:                                                                    # autograd of the bf16-op chain in fp32 on the host
        dact = (dy.float().cpu() @ wd.float().cpu()).to(torch.bfloat16).float()
        g32 = gu.float().cpu().clone().requires_grad_(True)
        (F.silu(g32[:, :I]) * g32[:, I:]).backward(dact)
        assert _rel(got[1:].float().cpu(), g32.grad[1:]) < 1.2e-2


FUSED_HEIGHTS = {"swiglu": (128, 160, 192, 208, 224, 256, 288, 320), "rope": (128, 160, 192, 208, 224, 256, 288, 320),
                 "swiglu_bwd": (128, 160, 192, 208, 224, 256, 272, 288, 320)}


@pytest.mark.parametrize("kind,height", [(k, h) for k, hs in FUSED_HEIGHTS.items() for h in hs])
def test_fused_epilogue_every_instantiated_tile_height(dev, kind, height):
    """the fused entry points pick their tile height from a cost model, so the bit-identity tests above reach only the
    heights their shapes happen to select.  `ug_gemm_set_fused_tile_height` forces each instantiated height in turn (ragged last
    row tile, several column tiles, the hand-counted `s_waitcnt vmcnt` of the SwiGLU-backward epilogue's prefetch included) and
    the result must equal GEMM + element-wise kernel bit for bit; a height an entry point does not instantiate is refused."""
    ops = _ops()
    from unigen_hip.lib import UniGenHipError
    g = torch.Generator(device=dev).manual_seed(height)
    M, K = 3 * 333, 160                                  # 999 rows: every height leaves a ragged last row tile
    try:
        if kind == "swiglu":
            I = 2048                                     # 4 x 16 = 64 tiles of 256 x 256: the smallest problem the fused form takes
            x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
            w = (torch.randn(2 * I, K, device=dev, generator=g) * (K ** -0.5)).to(torch.bfloat16)
            ops.set_gemm_tile_policy(3)
            gu_ref = ops.gemm(x, w)
            ops.set_gemm_tile_policy(-1)
            act_ref = ops.swiglu_fwd(gu_ref)
            ops.set_fused_tile_height(height)
            was, ops.FUSED_SWIGLU = ops.FUSED_SWIGLU, True
            try:
                gu, act = ops.gemm_swiglu(x, w)
            finally:
                ops.FUSED_SWIGLU = was
            assert torch.equal(gu, gu_ref) and torch.equal(act, act_ref)
        elif kind == "rope":
            L, N, heads, hd = 333, 1024, 6, 128
            x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev, generator=g) * 0.2).to(torch.bfloat16)
            b = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
            cos, sin = ops.rope_tables(L, hd, 1e6, dev)
            want = ops.rope_(ops.gemm(x, w, bias=b), cos, sin, L, heads, hd)
            ops.set_fused_tile_height(height)
            assert torch.equal(ops.gemm_qkv_rope(x, w, b, cos, sin, L, heads, hd), want)
        else:
            I = 1024
            dy = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
            wd = (torch.randn(K, I, device=dev, generator=g) * (K ** -0.5)).to(torch.bfloat16)
            gu = (torch.randn(M, 2 * I, device=dev, generator=g) * 2).to(torch.bfloat16)
            ops.set_gemm_tile_policy(3)
            want = ops.swiglu_bwd(gu, ops.gemm(dy, wd, b_kmajor=True))
            ops.set_gemm_tile_policy(-1)
            ops.set_fused_tile_height(height)
            assert torch.equal(ops.gemm_swiglu_bwd(dy, wd, gu), want)
        if kind != "swiglu_bwd":                         # 272 exists only for the SwiGLU backward
            ops.set_fused_tile_height(272)
            with pytest.raises(UniGenHipError):
                if kind == "swiglu":
                    was, ops.FUSED_SWIGLU = ops.FUSED_SWIGLU, True
                    try:
                        ops.gemm_swiglu(x, w)
                    finally:
                        ops.FUSED_SWIGLU = was
                else:
                    ops.gemm_qkv_rope(x, w, b, cos, sin, L, heads, hd)
    finally:
        ops.set_gemm_tile_policy(-1)
        ops.set_fused_tile_height(0)


@pytest.mark.parametrize("M,N,K", [(2900, 2816, 65528), (2000, 2816, 65536)])
def test_gemm_bf16_long_contraction_slices_up_to_199_tiles(dev, M, N, K):
    """132 output tiles of 256 x 256 with a 65 528-long contraction (the pt1 mixed batch's lm-head dgrad has 174 with K = 159 867):
    the automatic selection cuts every tile along K into private fp32 partials (round 4: up to 199 tiles, was 128) -- against
    sampled rows in fp64 on the host and against the 128 x 128 kernel (policy 0) on the whole output.  Second case (round 5, ADVICE
    r4): K % 32 == 0 makes the shape eligible for the one-round 128 ... 320-row kernel, which would run its 2 048 k-tiles on 176
    workgroups; the selection leaves such shapes (>= 2 048 k-tiles, < 200 workgroups) to the k-sliced form."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    a = (torch.randn(M, K, generator=g) * 0.25).to(torch.bfloat16)
    b = (torch.randn(K, N, generator=g) * 0.25).to(torch.bfloat16)          # k-major B: the dgrad layout
    A, B = a.to(dev), b.to(dev)
    out = ops.gemm(A, B, b_kmajor=True)
    rows = torch.tensor([0, 1, 255, 256, 1337, M - 341, M - 340, M - 1])
    want = a[rows].double() @ b.double()
    assert _rel(out[rows.to(dev)].double().cpu(), want) < 4e-3
    ops.set_gemm_tile_policy(0)
    try:
        ref = ops.gemm(A, B, b_kmajor=True)
    finally:
        ops.set_gemm_tile_policy(-1)
    assert _rel(out.float().cpu(), ref.float().cpu()) < 4e-3
    again = ops.gemm(A, B, b_kmajor=True)                                    # the scratch is left clean: same bits again
    assert torch.equal(out, again)


@pytest.mark.parametrize("mode", ["nt", "dgrad", "wgrad"])
def test_gemm_bf16_partial_last_round(dev, mode):
    """288 tiles of 256x256 = one full round + 32 leftover tiles: the leftovers are cut along K into fp32-scratch slices
    and finished by a second kernel (auto policy); all three epilogues, ragged M / N / K, scratch left zeroed (re-run)."""
    ops = _ops()
    ops.set_gemm_tile_policy(6)          # staggered kernel with the k-sliced tail forced on (auto needs K >= ~10k)
    ak, bk = {"nt": (False, False), "dgrad": (False, True), "wgrad": (True, True)}[mode]
    M, N, K = 4400, 4000, 3104
    g = torch.Generator().manual_seed(99)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(torch.bfloat16)
    ref = a.float() @ b.float().t()
    A = (a.t().contiguous() if ak else a).to(dev)
    B = (b.t().contiguous() if bk else b).to(dev)
    for rep in range(2):
        cbuf = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(A, B, out=cbuf, M=M, N=N, K=K, a_kmajor=ak, b_kmajor=bk, bias=bias.to(dev))
        assert _rel(cbuf, ref + bias.float()) < 4e-3, rep
    c32 = torch.ones(M, N, dtype=torch.float32, device=dev)
    ops.gemm(A, B, out=c32, M=M, N=N, K=K, a_kmajor=ak, b_kmajor=bk, epilogue=ops.UG_EPI_F32, beta=1)
    assert _rel(c32, ref + 1.0) < 1e-5 * math.sqrt(K) + 1e-6
    if mode == "nt":
        res = torch.randn(M, N, generator=g)
        r32 = torch.empty(M, N, dtype=torch.float32, device=dev)
        ops.gemm(A, B, out=r32, M=M, N=N, K=K, epilogue=ops.UG_EPI_RESID, resid=res.to(dev))
        want = res + ref.to(torch.bfloat16).float()
        assert ((r32.cpu() - want).abs() <= ref.abs() * 2.0 ** -7 + 1e-3).all()
    ops.set_gemm_tile_policy(-1)


def test_gemm_bf16_small_weight_gradient_private_partials(dev):
    """Attention-projection-sized weight gradient (48 tiles of 256x256): every tile is cut along K, each slice stores a
    private fp32 partial, a finishing pass sums them into the accumulating output (auto policy); ragged K, twice in a row."""
    ops = _ops()
    ops.set_gemm_tile_policy(-1)
    M, N, K = 2048, 1536, 5000
    g = torch.Generator().manual_seed(321)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16)
    ref = a.float() @ b.float().t()
    c32 = torch.full((M, N), 1.5, dtype=torch.float32, device=dev)
    A, B = a.t().contiguous().to(dev), b.t().contiguous().to(dev)
    for rep in (1, 2):
        ops.gemm(A, B, out=c32, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1)
        assert _rel(c32, rep * ref + 1.5) < 1e-5 * math.sqrt(K) + 1e-6, rep


@pytest.mark.parametrize("M,N,K", [(12336, 1536, 1536), (12336, 1536, 2048), (700, 512, 96), (321, 256, 32), (1000, 768, 160)])
def test_gemm_bf16_320_row_tiles(dev, M, N, K):
    """320 x 256 tiles (one round for the token-count x 1536 outputs of the attention block): forward with bias, forward into
    the fp32 residual, dgrad (B k-major); ragged M, 1 .. 64 k-tiles.  The two large shapes take the path by themselves
    (auto policy), the small ones force it."""
    ops = _ops()
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    b = (torch.randn(N, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
    res = torch.randn(M, N, device=dev, generator=g)
    ref = a.float() @ b.float().t()
    ops.set_gemm_tile_policy(-1 if M > 10000 else 10)
    try:
        out = ops.gemm(a, b, bias=bias)
        assert _rel(out, ref + bias.float()) < 4e-3
        r32 = torch.empty(M, N, dtype=torch.float32, device=dev)
        ops.gemm(a, b, out=r32, epilogue=ops.UG_EPI_RESID, resid=res)
        want = res + ref.to(torch.bfloat16).float()
        assert ((r32 - want).abs() <= ref.abs() * 2.0 ** -7 + 1e-3).all()
        out_d = ops.gemm(a, b.t().contiguous(), b_kmajor=True)
        assert _rel(out_d, ref) < 4e-3
        if M > 10000:                  # the same launches on the other kernels (policy 3 = 256 x 256 forced): same values up to
            ops.set_gemm_tile_policy(3)     # the summation order inside a k-tile, which is identical -- bit-equal
            assert torch.equal(ops.gemm(a, b, bias=bias), out)
    finally:
        ops.set_gemm_tile_policy(-1)


@pytest.mark.parametrize("K,shapes", [(12336, [(17920, 1536), (1536, 8960), (2048, 1536), (1536, 1536)]),
                                      (1000, [(300, 520), (64, 64), (777, 256), (256, 40), (130, 1000)])])
def test_gemm_wgrad_group_matches_single_launches(dev, K, shapes):
    """The grouped weight-gradient launch (a decoder layer's four dW = dY^T X in one grid) against the same problems launched
    one by one: first-write (beta 0) and accumulating (beta 1) outputs, ragged rows / columns / K, padded leading dimensions.
    Same tiles, same k order: the 256x256 kernel is bit-equal; problems the single launch k-slices agree to fp32 rounding."""
    ops = _ops()
    g = torch.Generator(device=dev).manual_seed(K)
    probs, refs = [], []
    for n, (rows, cols) in enumerate(shapes):
        Kn = K if n != 1 else max(40, K // 3 // 8 * 8)        # (round 4: the contraction length is per problem)
        dy = (torch.randn(Kn, ops.round_up(rows, 8) + 8, device=dev, generator=g) * 0.1).to(torch.bfloat16)[:, :rows]
        x = (torch.randn(Kn, ops.round_up(cols, 8), device=dev, generator=g) * 0.5).to(torch.bfloat16)[:, :cols]
        beta = n % 2
        dw = torch.full((rows, ops.round_up(cols, 4) + 4), 0.25, dtype=torch.float32, device=dev)[:, :cols]
        probs.append((dy, x, dw, beta))
        refs.append(dy.float().t() @ x.float() + (0.25 if beta else 0.0))
    old = ops.WGRAD_GROUP_MIN_TILES
    ops.WGRAD_GROUP_MIN_TILES = 0
    try:
        ops.gemm_wgrad_group(probs)
    finally:
        ops.WGRAD_GROUP_MIN_TILES = old
    for (dy, x, dw, beta), ref in zip(probs, refs):
        assert _rel(dw, ref) < 1e-5 * math.sqrt(K) + 1e-6, (tuple(dw.shape), beta)
        single = torch.full_like(dw, 0.25)
        ops.set_gemm_tile_policy(3)                    # whole 256x256 tiles, no k-slices: the same arithmetic
        ops.gemm(dy, x, out=single, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=beta)
        ops.set_gemm_tile_policy(-1)
        assert torch.equal(single, dw), tuple(dw.shape)


def test_gemm_bf16_long_contraction_few_tiles(dev):
    """lm-head dgrad shape class: 48 output tiles, K = 70 000 -- every tile is cut along K over several rounds with private
    partials and a summing pass that applies the bf16 epilogue (auto policy).  Reference: fp32 matmul on the same device
    (210 GFLOP is too much for the CPU oracle; the k-sliced path is also covered against the CPU on smaller shapes above)."""
    ops = _ops()
    ops.set_gemm_tile_policy(-1)
    M, N, K = 2000, 1500, 70000
    g = torch.Generator(device=dev).manual_seed(5)
    a = (torch.randn(M, K, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    bt = (torch.randn(K, N + 4, device=dev, generator=g) * 0.1).to(torch.bfloat16)        # B k-major, ld = N + 4
    ref = a.float() @ bt[:, :N].float()
    out = torch.zeros(M, N + 4, dtype=torch.bfloat16, device=dev)
    ops.gemm(a, bt, out=out, M=M, N=N, K=K, b_kmajor=True)
    assert _rel(out[:, :N], ref) < 4e-3 and float(out[:, N:].abs().max()) == 0.0


def test_gemm_rejects_bad_args(dev):
    ops = _ops()
    from unigen_hip.lib import UniGenHipError
    a = torch.zeros(64, 72, dtype=torch.bfloat16, device=dev)
    with pytest.raises(UniGenHipError):
        ops.gemm(a, a, K=68)                                # both row-major: K must be a multiple of 8
    with pytest.raises(UniGenHipError):
        ops.gemm(a, a, epilogue=ops.UG_EPI_RESID)           # residual epilogue without a residual


# ------------------------------------------------------------------ row ops
def test_rmsnorm_fwd_bwd(dev):
    ops = _ops()
    from oracle.ops_ref import rmsnorm_ref, rmsnorm_bwd_ref
    torch.manual_seed(0)
    for rows, cols in [(37, 1536), (130, 256)]:
        x = torch.randn(rows, cols) * 3
        w = torch.randn(cols) * 0.1 + 1
        y, rstd = ops.rmsnorm_fwd(x.to(dev), w.to(dev), 1e-6)
        ref = rmsnorm_ref(x, w, 1e-6)
        assert _maxabs(y, ref.to(torch.bfloat16)) <= 0.04 and _rel(y, ref) < 4e-3
        y32, _ = ops.rmsnorm_fwd(x.to(dev), w.to(dev), 1e-6, out_f32=True)
        assert _rel(y32, ref) < 1e-6
        dy = torch.randn(rows, cols).to(torch.bfloat16)
        dres0 = torch.randn(rows, cols)
        dres = dres0.clone().to(dev)
        dw = torch.zeros(cols, device=dev)
        d16 = ops.rmsnorm_bwd(dy.to(dev), x.to(dev), rstd, w.to(dev), dres, dw, want_bf16=True)
        dx_ref, dw_ref = rmsnorm_bwd_ref(dy.float(), x, w, 1e-6)
        assert _rel(dres, dres0 + dx_ref) < 1e-5
        assert _rel(dw, dw_ref) < 1e-5
        assert torch.equal(d16, dres.to(torch.bfloat16))          # the fused bf16 copy == a cast of the updated gradient


def test_rope_fwd_bwd(dev):
    ops = _ops()
    from oracle.ops_ref import rope_tables_ref, rope_ref
    torch.manual_seed(1)
    B, L, H, HKV, hd = 2, 45, 3, 1, 128
    qkv = torch.randn(B * L, (H + 2 * HKV) * hd).to(torch.bfloat16)
    cos, sin = ops.rope_tables(L, hd, 1e6, dev)
    cr, sr = rope_tables_ref(L, hd, 1e6)
    assert torch.equal(cos.cpu(), cr[:, : hd // 2]) and torch.equal(sin.cpu(), sr[:, : hd // 2])
    got = ops.rope_(qkv.clone().to(dev), cos, sin, L, H + HKV, hd).cpu()
    x = qkv[:, : (H + HKV) * hd].view(B, L, H + HKV, hd).permute(0, 2, 1, 3)       # [B, heads, L, d] bf16
    ref = rope_ref(x, cr, sr).to(torch.bfloat16)                                   # fp32 math, one bf16 round
    assert torch.equal(got[:, : (H + HKV) * hd].view(B, L, H + HKV, hd).permute(0, 2, 1, 3), ref)
    assert torch.equal(got[:, (H + HKV) * hd:], qkv[:, (H + HKV) * hd:])            # v untouched
    # backward = transpose of the rotation: <R x, y> == <x, R^T y>
    y = torch.randn_like(qkv)
    fx = ops.rope_(qkv.clone().to(dev), cos, sin, L, H + HKV, hd).float().cpu()[:, : (H + HKV) * hd]
    bty = ops.rope_(y.clone().to(dev), cos, sin, L, H + HKV, hd, backward=True).float().cpu()[:, : (H + HKV) * hd]
    lhs = (fx * y.float()[:, : (H + HKV) * hd]).sum()
    rhs = (qkv.float()[:, : (H + HKV) * hd] * bty).sum()
    # both sides carry bf16 output rounding (~2^-9 relative per term, random sign): scale by sqrt(#terms)
    assert abs(lhs - rhs) < 4e-3 * math.sqrt(fx.numel())


def test_swiglu_fwd_bwd(dev):
    ops = _ops()
    from oracle.ops_ref import swiglu_ref
    torch.manual_seed(2)
    gu = (torch.randn(77, 2 * 512) * 2).to(torch.bfloat16)
    act = ops.swiglu_fwd(gu.to(dev))
    ref = swiglu_ref(gu)                       # bf16 ops on CPU == the reference's autocast arithmetic
    assert _maxabs(act, ref) <= 0.07 and _rel(act, ref) < 6e-3
    dact = torch.randn(77, 512).to(torch.bfloat16)
    dgu = ops.swiglu_bwd(gu.to(dev), dact.to(dev))
    g32 = gu.float().clone().requires_grad_(True)
    i = 512
    (F.silu(g32[:, :i]) * g32[:, i:]).backward(dact.float())
    assert _rel(dgu, g32.grad) < 8e-3


def test_embed_fwd_bwd(dev):
    ops = _ops()
    torch.manual_seed(3)
    V, H = 333, 256
    W = torch.randn(V, H)
    ids = torch.randint(0, V, (97,))
    ids[::5] = 7
    out = ops.embed_fwd(ids.to(dev), W.to(dev))
    assert torch.equal(out.cpu(), W[ids])
    dout = torch.randn(97, H)
    dW = torch.zeros(V, H, device=dev)
    ops.embed_bwd(ids.to(dev), dout.to(dev), dW)
    ref = torch.zeros(V, H).index_add_(0, ids, dout)
    assert _rel(dW, ref) < 1e-6
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    bad = ids.clone(); bad[3] = V + 5
    ops.embed_fwd(bad.to(dev), W.to(dev), err)
    assert err.item() == 1


def test_embed_bwd_sorted_is_deterministic_and_matches_index_add(dev):
    """`ug_embed_bwd_sorted` (the data-parallel exchange's lookup scatter): pairs of W ranks padded with id -1, sorted by a stable
    sort; result = table + scale * index_add in that very order -- compared with a sequential fp32 host loop bit for bit, and
    two runs must agree bit for bit (the atomics form does not)."""
    ops = _ops()
    torch.manual_seed(5)
    V, H, W, cap = 333, 256, 4, 61
    ids = torch.randint(0, V, (W * cap,))
    ids[::3] = 11                                        # a long run (padding tokens repeat)
    ids[cap - 5:cap] = -1                                # rank 0's padding
    ids[-9:] = -1
    rows = torch.randn(W * cap, H)
    base = torch.randn(V, H)
    srt, order = torch.sort(ids.to(dev), stable=True)
    outs = []
    for _ in range(2):
        t = base.to(dev).clone()
        ops.embed_bwd_sorted(srt, order, rows.to(dev), t, 1.0 / W)
        outs.append(t.cpu())
    assert torch.equal(outs[0], outs[1])
    ref = base.clone()
    acc = {}
    for j in order.cpu().tolist():                       # sequential fp32 sums in sorted order, one scaled add per table row
        i = int(ids[j])
        if i >= 0:
            acc[i] = rows[j].clone() if i not in acc else acc[i] + rows[j]
    for i, a in acc.items():
        ref[i] = ref[i] + (1.0 / W) * a
    assert _rel(outs[0], ref) < 1e-6
    assert (outs[0] - ref).abs().max().item() < 1e-5


def test_colsum(dev):
    ops = _ops()
    x = torch.randn(1000, 200).to(torch.bfloat16)
    out = torch.ones(200, device=dev)
    ops.colsum_(x.to(dev), out)
    assert _rel(out, x.float().sum(0) + 1) < 1e-5


def test_zero_ranges_clears_exactly_the_listed_spans(dev):
    """ug_zero_ranges_f32: one launch clears the small accumulating gradients that lie between the big matrices of the flat
    gradient buffer (ragged span lengths, a span longer than one workgroup pass); everything else is untouched."""
    ops = _ops()
    n = 1 << 20
    buf = torch.arange(1, n + 1, dtype=torch.float32, device=dev)
    spans = [(0, 64), (4096, 1536), (70000, 4), (300000, 200000), (n - 128, 128)]
    table = torch.tensor([[lo, ln] for lo, ln in spans], dtype=torch.int64, device=dev)
    ops.zero_ranges_(buf, table, max(ln for _, ln in spans))
    want = torch.arange(1, n + 1, dtype=torch.float32)
    for lo, ln in spans:
        want[lo:lo + ln] = 0
    assert torch.equal(buf.cpu(), want)


def test_adamw_matches_torch(dev):
    ops = _ops()
    torch.manual_seed(4)
    n = 4099
    p0 = torch.randn(n); p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p = p0.clone().to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    pb = torch.empty(n, dtype=torch.bfloat16, device=dev)
    for step in range(1, 4):
        g = torch.randn(n)
        p_ref.grad = g.clone(); opt.step()
        ops.adamw_flat_(p, g.to(dev), m, v, pb, 1e-3, 0.9, 0.999, 1e-8, 0.01, step)
        assert _maxabs(p, p_ref.detach()) < 2e-6
    assert torch.equal(pb.cpu(), p.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("n", [1, 2, 3, 515, 4099, 3 * 2 * 512 * 5 + 7, 1 << 20])
def test_adamw_small_grid_form_is_the_same_update(dev, n):
    """The update as it runs beside the tokenizer (max_blocks > 0: `adamw_lean2_kernel`, a three-stage software pipeline with
    hand-counted waits, grid-stride over few workgroups) against the full-grid kernel: bit-identical p / m / v / bf16 copy over
    three steps, with trip counts 0, 1, 2, a multiple of the unroll and not (max_blocks = 2: 512 lanes)."""
    ops = _ops()
    g0 = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g0)
    state = []
    for mb in (0, 2, 256):
        p = p0.clone().to(dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
        pb = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        gg = torch.Generator().manual_seed(n + 1)
        for step in range(1, 4):
            g = torch.randn(n, generator=gg).to(dev)
            ops.adamw_flat_(p, g, m, v, pb, 1e-3, 0.9, 0.999, 1e-8, 0.01, step, grad_scale=0.5, max_blocks=mb)
        state.append((p.cpu(), m.cpu(), v.cpu(), pb.cpu()))
    for other in state[1:]:
        for a, b in zip(state[0], other):
            assert torch.equal(a, b)


def test_adamw_whole_chip_kernel_beyond_32_bit_indices(dev):
    """Flat runs of 2^29 elements and more keep the whole-chip four-element kernel (`adamw_kernel`, 64-bit indices; everything smaller
    runs the pipelined small-grid kernel since round 4): one update over 2^29 + 5 elements against the pipelined kernel on the two
    halves of a copy -- the same element function, the same bits."""
    ops = _ops()
    n = (1 << 29) + 5
    gen = torch.Generator(device=dev).manual_seed(11)
    p0 = torch.randn(n, device=dev, generator=gen)
    g = torch.randn(n, device=dev, generator=gen)
    outs = []
    for split in (False, True):
        p = p0.clone(); m = torch.full((n,), 0.01, device=dev); v = torch.full((n,), 0.02, device=dev)
        pb = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        if not split:
            ops.adamw_flat_(p, g, m, v, pb, 1e-3, 0.9, 0.999, 1e-8, 0.01, 2)
        else:
            h = (n // 2) & ~7
            for lo, hi in ((0, h), (h, n)):
                ops.adamw_flat_(p[lo:hi], g[lo:hi], m[lo:hi], v[lo:hi], pb[lo:hi], 1e-3, 0.9, 0.999, 1e-8, 0.01, 2)
        outs.append((p, m, v, pb))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    del outs


# ------------------------------------------------------------------ cross entropy
def test_ce_fwd_bwd(dev):
    ops = _ops()
    torch.manual_seed(5)
    R, V = 19, 1003
    ld = (V + 63) // 64 * 64
    logits = (torch.randn(R, V) * 3).to(torch.bfloat16)
    labels = torch.randint(0, V, (R,)); labels[::4] = -100
    buf = torch.full((R, ld), 7.0, dtype=torch.bfloat16, device=dev)
    buf[:, :V] = logits.to(dev)
    lc, lse, loss_row, logp = ops.ce_fwd(buf, V, labels.to(dev), want_logp=True)
    ref = F.cross_entropy(logits.float(), labels, ignore_index=-100)
    assert abs(lc[0].item() - ref.item()) / ref.item() < 1e-5
    assert lc[1].item() == (labels != -100).sum().item()
    lp_ref = torch.log_softmax(logits.float(), -1).gather(1, labels.clamp(min=0)[:, None])[:, 0]
    assert _maxabs(logp[labels != -100], lp_ref[labels != -100]) < 1e-4
    gscale = torch.tensor([0.5], device=dev)
    ops.ce_bwd_(buf, V, labels.to(dev), lse, lc, gscale)
    lg = logits.float().clone().requires_grad_(True)
    (0.5 * F.cross_entropy(lg, labels, ignore_index=-100)).backward()
    assert _rel(buf[:, :V], lg.grad) < 6e-3
    assert buf[:, V:].abs().max().item() == 0


# ------------------------------------------------------------------ attention
def _ref_masks(B, L, kind, gen):
    """additive masks with the structure the reference builders emit (prompting_utils.py:975-1074)"""
    neg = float(torch.iinfo(torch.int64).min)
    allow = torch.zeros(B, L, L, dtype=torch.bool)
    tril = torch.tril(torch.ones(L, L, dtype=torch.bool))
    for b in range(B):
        if kind == "causal":
            allow[b] = tril
        elif kind == "t2i":       # left pads | causal text | bidirectional image tail
            npad = int(torch.randint(0, L // 4, (1,), generator=gen))
            nimg = L // 3
            a = tril.clone()
            a[npad:, :npad] = False
            a[L - nimg:, npad:] = True
            allow[b] = a
        elif kind == "mmu":       # prefix fully visible, causal after
            pre = L // 2
            a = tril.clone(); a[:, :pre] = True
            allow[b] = a
        elif kind == "random":
            a = torch.rand(L, L, generator=gen) < 0.3
            a[torch.arange(L), torch.arange(L)] = True
            allow[b] = a
    add = torch.where(allow, torch.zeros(()), torch.full((), neg))
    return add[:, None].contiguous(), allow


@pytest.mark.parametrize("kind", ["causal", "t2i", "mmu", "random"])
@pytest.mark.parametrize("B,L,H,HKV", [(2, 70, 2, 1), (1, 200, 12, 2), (2, 129, 6, 2), (8, 333, 24, 4), (6, 400, 24, 4),
                                     (2, 128, 6, 2), (1, 256, 12, 2)])      # (whole 64-key tiles only: no clamped last tile)
def test_attention_fwd_bwd(dev, kind, B, L, H, HKV):
    """The last two shapes have >= 512 128-row query tiles, i.e. they run the eight-wave forward kernel (ragged last
    tile at 333; at 400 the last 128-row tile has no second half)."""
    ops = _ops()
    from oracle.ops_ref import attention_ref
    hd = 128
    gen = torch.Generator().manual_seed(B * 1000 + L + H)
    qkv = (torch.randn(B * L, (H + 2 * HKV) * hd, generator=gen)).to(torch.bfloat16)
    mask_add, allow = _ref_masks(B, L, kind, gen)
    for mdt in (torch.float32, torch.bfloat16):
        err = torch.zeros(1, dtype=torch.int32, device=dev)
        mb = ops.mask_compress(mask_add.to(mdt).to(dev), err)
        assert err.item() == 0
        bits = mb.bits.cpu()
        for b in range(B):
            for w in range(mb.nW):
                cols = allow[b, :, w * 64:(w + 1) * 64]
                want = (cols.long() << torch.arange(cols.shape[1])).sum(1)   # as signed int64 words
                assert torch.equal(bits[b, :, w], want)
    o, lse = ops.attn_fwd(qkv.to(dev), mb, H, HKV, hd)
    q = qkv[:, : H * hd].view(B, L, H, hd).permute(0, 2, 1, 3)
    k = qkv[:, H * hd:(H + HKV) * hd].view(B, L, HKV, hd).permute(0, 2, 1, 3)
    v = qkv[:, (H + HKV) * hd:].view(B, L, HKV, hd).permute(0, 2, 1, 3)
    qf, kf, vf = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ref = attention_ref(qf, kf, vf, mask_add, 1.0 / math.sqrt(hd))            # [B,H,L,d]
    got = o.view(B, L, H, hd).permute(0, 2, 1, 3)
    assert _rel(got, ref) < 8e-3, f"fwd rel {_rel(got, ref)}"
    dout = torch.randn(B * L, H * hd, generator=gen).to(torch.bfloat16)
    ref.backward(dout.float().view(B, L, H, hd).permute(0, 2, 1, 3))
    dqkv_det = ops.attn_bwd(qkv.to(dev), o, lse, dout.to(dev), mb, H, HKV, hd, split_heads=False).float().cpu()
    dqkv = ops.attn_bwd(qkv.to(dev), o, lse, dout.to(dev), mb, H, HKV, hd).float().cpu()      # split-head dK/dV (atomics)
    assert _rel(dqkv, dqkv_det) < 4e-3
    assert float(ops._dkv_workspace(B * L, 2 * HKV * hd, dev).abs().max()) == 0.0
    dq = dqkv[:, : H * hd].view(B, L, H, hd).permute(0, 2, 1, 3)
    dk = dqkv[:, H * hd:(H + HKV) * hd].view(B, L, HKV, hd).permute(0, 2, 1, 3)
    dv = dqkv[:, (H + HKV) * hd:].view(B, L, HKV, hd).permute(0, 2, 1, 3)
    assert _rel(dq, qf.grad) < 2e-2, f"dq rel {_rel(dq, qf.grad)}"
    assert _rel(dk, kf.grad) < 2e-2, f"dk rel {_rel(dk, kf.grad)}"
    assert _rel(dv, vf.grad) < 2e-2, f"dv rel {_rel(dv, vf.grad)}"


@pytest.mark.parametrize("B,L,H,HKV,split", [(8, 333, 24, 4, True), (6, 400, 12, 2, True), (2, 129, 6, 2, True), (1, 200, 12, 2, False),
                                             (16, 771, 12, 2, True)])
def test_attention_bwd_fused_rope_and_bias_sums(dev, B, L, H, HKV, split):
    """RoPE transposed and the q | k | v bias gradient applied where the attention backward stores dq / dk / dv (32-row dQ kernel,
    dK / dV finishing pass) against the stand-alone passes over the stored tensor (ug_rope backward, ug_colsum_bf16): the
    gradients bit for bit (same arithmetic on the same bf16 values; dK / dV carry the fp32 atomics' run-to-run order, so those are
    compared through the deterministic kernel when the shape has one), the bias gradient to fp32 summation order.  Small / odd
    shapes take the library's fall-back (the stand-alone passes, issued by the library itself)."""
    ops = _ops()
    hd = 128
    gen = torch.Generator().manual_seed(L * 7 + H)
    qkv = torch.randn(B * L, (H + 2 * HKV) * hd, generator=gen).to(torch.bfloat16).to(dev)
    mask_add, _ = _ref_masks(B, L, "t2i", gen)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    mb = ops.mask_compress(mask_add.to(dev), err)
    o, lse = ops.attn_fwd(qkv, mb, H, HKV, hd)
    dout = torch.randn(B * L, H * hd, generator=gen).to(torch.bfloat16).to(dev)
    inv = 1.0 / (1e6 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    ang = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
    cos, sin = ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev)
    plain = ops.attn_bwd(qkv, o, lse, dout, mb, H, HKV, hd, split_heads=split)
    want = plain.clone()
    ops.rope_(want, cos, sin, L, H + HKV, hd, backward=True)
    want_b = torch.zeros((H + 2 * HKV) * hd, device=dev)
    ops.colsum_(want, want_b)
    got_b = torch.full(((H + 2 * HKV) * hd,), 0.25, device=dev)                 # accumulates into what is there
    got = ops.attn_bwd(qkv, o, lse, dout, mb, H, HKV, hd, split_heads=split, rope=(cos, sin), dbias=got_b)
    nq = H * hd
    assert torch.equal(got[:, :nq], want[:, :nq])                                # dq: deterministic kernels
    if split:
        assert _rel(got[:, nq:].float(), want[:, nq:].float()) < 4e-3           # fp32 atomics order before the bf16 rounding
        ref_b = torch.zeros_like(want_b)
        ops.colsum_(got, ref_b)                                                  # the sums of what THIS run stored
    else:
        assert torch.equal(got[:, nq:], want[:, nq:])
        ref_b = want_b
    eb = (got_b - 0.25 - ref_b).abs().max().item()
    print(f"    fused RoPE / bias sums B={B} L={L} H={H}/{HKV} split={split}: bias-gradient max diff {eb:.2e} at max {ref_b.abs().max().item():.1f}")
    assert eb < 2e-4 * max(1.0, ref_b.abs().max().item())
    assert float(ops._dkv_workspace(B * L, 2 * HKV * hd, dev).abs().max()) == 0.0


def test_mask_rejects_soft_values(dev):
    ops = _ops()
    m = torch.zeros(1, 1, 64, 64, device=dev); m[0, 0, 3, 5] = -1.5
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.mask_compress(m, err)
    assert err.item() & 2


def test_mask_causal_builder(dev):
    ops = _ops()
    mb = ops.mask_causal(2, 100, dev)
    bits = mb.bits.cpu()
    tril = torch.tril(torch.ones(100, 100, dtype=torch.bool))
    for w in range(2):
        cols = tril[:, w * 64:(w + 1) * 64]
        assert torch.equal(bits[0, :, w], (cols.long() << torch.arange(cols.shape[1])).sum(1))


# ------------------------------------------------------------------ tokenizer kernels (exact fp32)
@pytest.mark.parametrize("cin,cout,k,stride,asym,ups,H", [
    (128, 128, 3, 1, False, False, 16), (4, 128, 3, 1, False, False, 16), (128, 256, 1, 1, False, False, 8),
    (128, 128, 3, 2, True, False, 16), (256, 256, 3, 1, False, True, 8), (512, 13, 3, 1, False, False, 16),
    (13, 13, 1, 1, False, False, 16), (13, 512, 3, 1, False, False, 16), (128, 3, 3, 1, False, False, 16)])
def test_conv2d_f32(dev, cin, cout, k, stride, asym, ups, H):
    ops = _ops()
    gen = torch.Generator().manual_seed(cin * 31 + cout)
    B = 2
    x = torch.randn(B, cin, H, H, generator=gen)
    w = torch.randn(cout, cin, k, k, generator=gen) / math.sqrt(cin * k * k)
    bias = torch.randn(cout, generator=gen)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    if asym:
        ref = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w, bias, stride=stride)
    else:
        ref = F.conv2d(xin, w, bias, stride=stride, padding=k // 2)
    res = torch.randn_like(ref)
    wp, cpad = ops.pack_conv_weight(w.to(dev))
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    y = ops.conv2d_nhwc(x_nhwc, wp, cpad, bias.to(dev), cout, k, stride=stride, asym_pad=asym, upsample=ups,
                        residual=res.permute(0, 2, 3, 1).contiguous().to(dev))
    got = y.permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert _maxabs(got, ref + res) < 2e-5, _maxabs(got, ref + res)


@pytest.mark.parametrize("cin,cout,k,stride,asym,ups,H", [
    (128, 128, 3, 1, False, False, 16), (128, 256, 1, 1, False, False, 8), (128, 128, 3, 2, True, False, 16),
    (256, 256, 3, 1, False, True, 8), (32, 192, 3, 1, False, False, 12), (512, 512, 3, 1, False, False, 6),
    (64, 68, 3, 1, False, False, 9)])
def test_conv2d_split_matches_fp64(dev, cin, cout, k, stride, asym, ups, H):
    """Scaled two-way f16 split convolution (three partial products): fp32-accurate -- within 3x of the exact fp32 MFMA
    chain's own distance to the fp64 result (measured ~1-2x; the reference's GPU path runs these convs in TF32, ~1000x)."""
    ops = _ops()
    gen = torch.Generator().manual_seed(cin * 17 + cout)
    B = 2
    x = torch.randn(B, cin, H, H, generator=gen) * 3.0
    w = torch.randn(cout, cin, k, k, generator=gen) / math.sqrt(cin * k * k)
    bias = torch.randn(cout, generator=gen)
    xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
    if asym:
        ref = F.conv2d(F.pad(xin.double(), (0, 1, 0, 1)), w.double(), bias.double(), stride=stride)
    else:
        ref = F.conv2d(xin.double(), w.double(), bias.double(), stride=stride, padding=k // 2)
    res = torch.randn(ref.shape, generator=gen)
    ref = ref + res.double()
    wp, cpad = ops.pack_conv_weight(w.to(dev))
    assert ops.conv_split_eligible(cin, cout, cpad)
    ws = ops.split_conv_weight(wp)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    kw = dict(stride=stride, asym_pad=asym, upsample=ups, residual=res.permute(0, 2, 3, 1).contiguous().to(dev))
    exact = ops.conv2d_nhwc(x_nhwc, wp, cpad, bias.to(dev), cout, k, **kw).permute(0, 3, 1, 2).cpu().double()
    split = ops.conv2d_nhwc(x_nhwc, wp, cpad, bias.to(dev), cout, k, w_split=ws, **kw).permute(0, 3, 1, 2).cpu().double()
    assert split.shape == ref.shape
    e_exact, e_split = (exact - ref).abs().max().item(), (split - ref).abs().max().item()
    print(f"    split conv cin={cin} cout={cout} k={k}: max err vs fp64 {e_split:.2e} (exact fp32 chain {e_exact:.2e}, ratio {e_split / max(e_exact, 1e-12):.2f})")
    assert e_split < 1e-4, (e_split, e_exact)
    assert e_split <= 3.0 * e_exact + 4e-6, (e_split, e_exact)


@pytest.mark.parametrize("cin,cout,H,W,gn", [(128, 128, 16, 16, False), (128, 128, 16, 32, True), (256, 192, 11, 21, True),
                                             (32, 64, 8, 16, False), (512, 512, 6, 6, True), (64, 68, 9, 40, False),
                                             (128, 64, 250, 263, True), (64, 128, 256, 128, False)])
def test_conv3x3_patch_matches_fp64(dev, cin, cout, H, W, gn):
    """LDS-resident-patch 3x3 conv (optionally with GroupNorm+swish on the load path) against fp64, and against the
    unfused path of the same library (GroupNorm kernel + im2col split conv)."""
    ops = _ops()
    gen = torch.Generator().manual_seed(cin * 13 + cout + W)
    B = 3 if H * W < 4096 else 2          # the two large cases reach the 16-row / eight-wave variant (>= 512 tiles)
    x = torch.randn(B, cin, H, W, generator=gen) * 2.0 + 0.3
    w = torch.randn(cout, cin, 3, 3, generator=gen) / math.sqrt(cin * 9)
    bias = torch.randn(cout, generator=gen)
    g = torch.randn(cin, generator=gen); be = torch.randn(cin, generator=gen)
    xin = x.double()
    if gn:
        xin = F.group_norm(xin, 32, g.double(), be.double(), eps=1e-6)
        xin = xin * torch.sigmoid(xin)
    ref = F.conv2d(xin, w.double(), bias.double(), padding=1)
    res = torch.randn(ref.shape, generator=gen)
    ref = ref + res.double()
    wp, cpad = ops.pack_conv_weight(w.to(dev))
    ws = ops.split_conv_weight(wp)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    res_d = res.permute(0, 2, 3, 1).contiguous().to(dev)
    fused_gn = (ops.groupnorm_stats(x_nhwc), g.to(dev), be.to(dev), 32, True) if gn else None
    # scale bound of the normalised tensor from the layer's actual gamma / beta (here N(0,1) draws: |gamma| up to ~3.5)
    bound = ops.gn_out_bound(g.abs().max().item(), be.abs().max().item(), (cin // 32) * H * W) if gn else None
    got = ops.conv3x3_nhwc(x_nhwc, ws, cpad, bias.to(dev), cout, residual=res_d, gn=fused_gn, gn_bound=bound)
    xn = ops.groupnorm_swish(x_nhwc, g.to(dev), be.to(dev), swish=True) if gn else x_nhwc
    unfused = ops.conv2d_nhwc(xn, wp, cpad, bias.to(dev), cout, 3, residual=res_d, w_split=ws)
    got64 = got.permute(0, 3, 1, 2).cpu().double()
    assert got64.shape == ref.shape
    assert (got64 - ref).abs().max().item() < 3e-5
    assert _maxabs(got, unfused) < 3e-5


@pytest.mark.parametrize("B,H,W", [(2, 37, 50), (3, 256, 256)])
def test_conv3x3_patch_four_channel_input(dev, B, H, W):
    """conv_in (3 -> 128 on RGB padded to 16-byte pixels, magvitv2.py:152-156) on the LDS-resident-patch kernel: x has 4 channels,
    the weights are packed for one 32-channel slab and the kernel reads the other 28 channels as zero.  Against fp64 and the exact
    fp32 im2col kernel, with the epilogue statistics."""
    ops = _ops()
    gen = torch.Generator().manual_seed(H + W)
    x = torch.rand(B, 3, H, W, generator=gen) * 2 - 1
    w = torch.randn(128, 3, 3, 3, generator=gen) / math.sqrt(27)
    bias = torch.randn(128, generator=gen)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    xd = ops.nchw_to_nhwc(x.to(dev), 4)
    w4 = torch.cat([w, w.new_zeros(128, 1, 3, 3)], 1).to(dev)
    w32 = torch.cat([w, w.new_zeros(128, 29, 3, 3)], 1).to(dev)
    wp4, cpad = ops.pack_conv_weight(w4)
    exact = ops.conv2d_nhwc(xd, wp4, cpad, bias.to(dev), 128, 3)
    wp32, cpad32 = ops.pack_conv_weight(w32)
    st = ops.gn_stats_slots(1, B, dev)[0]
    got = ops.conv3x3_nhwc(xd, ops.split_conv_weight(wp32), cpad32, bias.to(dev), 128, out_stats=st)
    e_split = (got.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item()
    e_exact = (exact.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item()
    print(f"    conv_in {B}x{H}x{W}: split patch kernel max err vs fp64 {e_split:.2e}, exact fp32 kernel {e_exact:.2e}")
    assert e_split < 3e-6 and e_split <= 3.0 * e_exact + 1e-6
    mr = ops.groupnorm_finalize(st, B, H * W, 128)
    assert (mr - ops.groupnorm_stats(got)).abs().max().item() == 0.0


@pytest.mark.parametrize("cin,cout,H,W", [(128, 128, 16, 32), (64, 256, 11, 21), (512, 512, 6, 6), (128, 128, 250, 263), (64, 256, 128, 128)])
def test_conv3x3_epilogue_groupnorm_sums(dev, cin, cout, H, W):
    """The sums of the consuming GroupNorm gathered in the convolution's epilogue (three kernel variants: 64- / 128-channel
    8-row workgroups, 16-row eight-wave workgroups; ragged tiles; a residual) give the (mean, rstd) of the separate pass over
    the stored output, and the fp64 statistics of that output."""
    ops = _ops()
    gen = torch.Generator().manual_seed(cin + 3 * cout + W)
    B = 3 if H * W < 4096 else 2
    x = (torch.randn(B, cin, H, W, generator=gen) * 2.0 + 0.3).permute(0, 2, 3, 1).contiguous().to(dev)
    w = torch.randn(cout, cin, 3, 3, generator=gen) / math.sqrt(cin * 9)
    bias = (torch.randn(cout, generator=gen) * 3.0).to(dev)
    res = (torch.randn(B, H, W, cout, generator=gen) + 1.5).to(dev)
    wp, cpad = ops.pack_conv_weight(w.to(dev))
    ws = ops.split_conv_weight(wp)
    plain = ops.conv3x3_nhwc(x, ws, cpad, bias, cout, residual=res)
    st, st2 = ops.gn_stats_slots(2, B, dev)
    y = ops.conv3x3_nhwc(x, ws, cpad, bias, cout, residual=res, out_stats=st)
    assert torch.equal(y, plain)
    assert torch.equal(ops.stats_amax(st), ops.amax(y.view(-1, cout)).view(1))
    if (H * W) % 128 == 0:                                   # the im2col form of the same convolution gathers the same sums
        y2 = ops.conv2d_nhwc(x, wp, cpad, bias, cout, 3, residual=res, w_split=ws, out_stats=st2)
        mr2 = ops.groupnorm_finalize(st2, B, H * W, cout, groups=32, eps=1e-6)
        assert (mr2 - ops.groupnorm_stats(y2, groups=32, eps=1e-6)).abs().max().item() == 0.0
        assert torch.equal(ops.stats_amax(st2), ops.amax(y2.view(-1, cout)).view(1))
    mr = ops.groupnorm_finalize(st, B, H * W, cout, groups=32, eps=1e-6)
    mr_pass = ops.groupnorm_stats(y, groups=32, eps=1e-6)
    y64 = y.double().view(B, H * W, 32, cout // 32)
    mean = y64.mean(dim=(1, 3))
    rstd = 1.0 / torch.sqrt(y64.var(dim=(1, 3), unbiased=False) + 1e-6)
    e_mean = (mr[..., 0].double() - mean).abs().max().item()
    e_rstd = ((mr[..., 1].double() - rstd).abs() / rstd).max().item()
    d = (mr - mr_pass).abs().max().item()
    print(f"    epilogue GroupNorm sums {cin}->{cout} {H}x{W}: mean err {e_mean:.1e}, rstd rel err {e_rstd:.1e}, vs the separate pass {d:.1e}")
    assert e_mean < 2e-7 * max(1.0, mean.abs().max().item()) and e_rstd < 2e-7
    assert d <= 2.4e-7 * max(1.0, mr_pass.abs().max().item())          # both are fp64 sums rounded to fp32 once


def test_conv3x3_groupnorm_on_load_with_large_affine_parameters(dev):
    """the GroupNorm-on-load convolution took a FIXED bound (|gamma| <= 1, |beta| <= 12) for the scale of the
    normalised tensor, so a checkpoint with larger affine parameters would have saturated silently.  The bound now comes
    from the layer's own gamma / beta (ops.gn_out_bound): gamma ~ 60 N(0,1), beta ~ 30 N(0,1) stays fp32-accurate."""
    ops = _ops()
    gen = torch.Generator().manual_seed(9)
    cin, cout, H, W, B = 128, 128, 32, 32, 2
    x = torch.randn(B, cin, H, W, generator=gen) * 2.0 + 0.3
    x[0, 5, 3, 4] = 400.0                                  # an outlier: its normalised value is far beyond the bulk
    w = torch.randn(cout, cin, 3, 3, generator=gen) / math.sqrt(cin * 9)
    bias = torch.randn(cout, generator=gen)
    g, be = 60.0 * torch.randn(cin, generator=gen), 30.0 * torch.randn(cin, generator=gen)
    xin = F.group_norm(x.double(), 32, g.double(), be.double(), eps=1e-6)
    xin = xin * torch.sigmoid(xin)
    ref = F.conv2d(xin, w.double(), bias.double(), padding=1)
    wp, cpad = ops.pack_conv_weight(w.to(dev))
    ws = ops.split_conv_weight(wp)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().to(dev)
    bound = ops.gn_out_bound(g.abs().max().item(), be.abs().max().item(), (cin // 32) * H * W)
    assert bound >= xin.abs().max().item()
    got = ops.conv3x3_nhwc(x_nhwc, ws, cpad, bias.to(dev), cout, gn=(ops.groupnorm_stats(x_nhwc), g.to(dev), be.to(dev), 32, True),
                           gn_bound=bound)
    err = (got.permute(0, 3, 1, 2).cpu().double() - ref).abs().max().item()
    print(f"    large-affine GroupNorm on load: max err vs fp64 {err:.2e} at max|ref| {ref.abs().max().item():.1f}, bound {bound:.0f}")
    assert err < 3e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(Exception):
        ops.conv3x3_nhwc(x_nhwc, ws, cpad, bias.to(dev), cout, gn=(ops.groupnorm_stats(x_nhwc), g.to(dev), be.to(dev), 32, True))


@pytest.mark.parametrize("M,N,K,act", [(300, 1152, 1152, 0), (257, 4304, 1152, 1), (129, 1152, 4304, 0), (64, 68, 36, 1)])
def test_linear_split_matches_fp64(dev, M, N, K, act):
    """Split-f16 linear (ragged K slabs, GELU, bias, residual, strided output) against fp64 and the fp32 MFMA path."""
    ops = _ops()
    gen = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=gen) * 1.5
    W = torch.randn(N, K, generator=gen) / math.sqrt(K)
    bias = torch.randn(N, generator=gen)
    res = torch.randn(M, N, generator=gen)
    pre = x.double() @ W.double().t() + bias.double()
    if act:
        pre = F.gelu(pre, approximate="tanh")
    ref = pre + res.double()
    ws, n_pad = ops.split_linear_weight(W.to(dev))
    out = torch.full((M, N + 8), 7.0, device=dev)[:, :N]                   # strided destination, sentinel columns
    got = ops.linear_split(x.to(dev), ws, n_pad, N, bias.to(dev), residual=res.to(dev), act=act, out=out)
    exact = ops.linear_f32(x.to(dev), W.to(dev), bias.to(dev), residual=res.to(dev), act=act)
    e_split = (got.cpu().double() - ref).abs().max().item()
    e_exact = (exact.cpu().double() - ref).abs().max().item()
    print(f"    split linear M={M} N={N} K={K}: max err vs fp64 {e_split:.2e} (exact fp32 chain {e_exact:.2e})")
    assert e_split < 1e-4 and e_split <= 3.0 * e_exact + 4e-6, (e_split, e_exact)
    assert torch.all(out.as_strided((M, 8), (N + 8, 1), out.storage_offset() + N) == 7.0)


def test_conv_split_randomised_shapes(dev):
    """Seeded sweep over channel counts, ragged image sizes, kernel sizes / strides / upsampling, bias / residual / fused
    GroupNorm: every split-bf16 kernel (im2col, 8-row patch, 16-row patch) against the exact fp32 MFMA convolution."""
    ops = _ops()
    rng = random.Random(2024)
    gen = torch.Generator().manual_seed(2024)
    for case in range(24):
        cin = rng.choice([32, 64, 96, 128, 160, 256])
        cout = rng.choice([64, 68, 128, 192, 256, 320])
        k = rng.choice([1, 3, 3, 3])
        big = case % 6 == 0                                   # a few cases large enough for the 16-row variant
        H, W = (rng.randint(120, 136), rng.randint(250, 270)) if big else (rng.randint(3, 40), rng.randint(3, 40))
        B = 2 if big else rng.randint(1, 3)
        if big:
            cin, cout = rng.choice([32, 64]), rng.choice([64, 128])
        mode = rng.choice(["plain", "plain", "stride2", "ups"]) if k == 3 else "plain"
        use_gn = k == 3 and mode == "plain" and cin % 128 == 0 and rng.random() < 0.7
        use_res, use_bias = rng.random() < 0.5, rng.random() < 0.8
        x = (torch.randn(B, H, W, cin, generator=gen) * 1.5 + 0.2).to(dev)
        w = (torch.randn(cout, cin, k, k, generator=gen) / math.sqrt(cin * k * k)).to(dev)
        bias = torch.randn(cout, generator=gen).to(dev) if use_bias else None
        wp, cpad = ops.pack_conv_weight(w)
        ws = ops.split_conv_weight(wp)
        kw = dict(stride=2 if mode == "stride2" else 1, asym_pad=mode == "stride2", upsample=mode == "ups")
        xin = x
        gn = None
        if use_gn:
            ga, be = torch.randn(cin, generator=gen).to(dev), torch.randn(cin, generator=gen).to(dev)
            xin = ops.groupnorm_swish(x, ga, be, swish=True)
            gn = (ops.groupnorm_stats(x), ga, be, 32, True)
        ref = ops.conv2d_nhwc(xin, wp, cpad, bias, cout, k, **kw)
        res = torch.randn(ref.shape, generator=gen).to(dev) if use_res else None
        if res is not None:
            ref = ref + res
        got = ops.conv2d_nhwc(xin, wp, cpad, bias, cout, k, residual=res, w_split=ws, **kw)
        tol = 2e-5 * max(1.0, ref.abs().max().item())
        assert got.shape == ref.shape and _maxabs(got, ref) < tol, (case, cin, cout, k, H, W, mode, _maxabs(got, ref))
        if k == 3 and mode == "plain":
            bound = ops.gn_out_bound(ga.abs().max().item(), be.abs().max().item(), (cin // 32) * H * W) if use_gn else None
            got3 = ops.conv3x3_nhwc(x, ws, cpad, bias, cout, residual=res, gn=gn, gn_bound=bound)
            assert _maxabs(got3, ref) < tol, (case, cin, cout, H, W, use_gn, _maxabs(got3, ref))


def test_conv_split_weights_reconstruct(dev):
    """The two fp16 planes of the split weight image, divided by the recorded power-of-two scale, add back to the fp32
    weights to 2^-22 relative (tile order, swizzle, scale exponent, trailing max|w| record)."""
    ops = _ops()
    torch.manual_seed(3)
    taps, cin, cpad = 9, 64, 256
    wp = torch.randn(taps, cin, cpad, device=dev) * 0.1
    buf = ops.split_conv_weight(wp)
    n = 2 * taps * cin * cpad
    amax = buf[n:n + 2].view(torch.float32)
    assert float(amax) == float(wp.abs().max())
    ew = 14 - math.floor(math.log2(float(amax)))
    ws = buf[:n].view(taps, cin // 32, cpad // 128, 2, 128, 4, 8).double()
    r = torch.arange(128, device=dev)
    stored = torch.arange(4, device=dev)[None, :] ^ (((r[:, None] >> 2) & 1) << 1)   # logical chunk -> stored position (swz_q, conv_split.hip)
    idx = stored[None, None, None, None, :, :, None].expand(taps, cin // 32, cpad // 128, 2, 128, 4, 8)
    logical = torch.gather(ws, 5, idx)                                               # [.., n, chunk, 8]
    assert float(logical[:, :, :, 0].abs().max()) < 2.0 ** 15 and float(logical[:, :, :, 0].abs().max()) >= 2.0 ** 14
    total = logical.sum(3).reshape(taps, cin // 32, cpad // 128, 128, 32)           # planes summed
    back = total.permute(0, 1, 4, 2, 3).reshape(taps, cin, cpad) * 2.0 ** -ew
    err = (back - wp.double()).abs()
    assert bool((err <= wp.double().abs() * 2.0 ** -22 + float(amax) * 2.0 ** -38).all()), float(err.max())


def test_groupnorm_swish(dev):
    ops = _ops()
    torch.manual_seed(7)
    for C, H in [(128, 16), (256, 8), (512, 4)]:
        x = torch.randn(2, C, H, H) * 2 + 0.5
        g = torch.randn(C); b = torch.randn(C)
        ref = F.group_norm(x, 32, g, b, eps=1e-6)
        xn = x.permute(0, 2, 3, 1).contiguous().to(dev)
        y0 = ops.groupnorm_swish(xn, g.to(dev), b.to(dev), swish=False).permute(0, 3, 1, 2)
        assert _maxabs(y0, ref) < 5e-6
        y1 = ops.groupnorm_swish(xn, g.to(dev), b.to(dev), swish=True).permute(0, 3, 1, 2)
        assert _maxabs(y1, ref * torch.sigmoid(ref)) < 5e-6


def test_gemm_f32_and_softmax(dev):
    ops = _ops()
    torch.manual_seed(8)
    Bn, T, C = 2, 256, 512
    q = torch.randn(Bn, T, C); k = torch.randn(Bn, T, C); v = torch.randn(Bn, T, C)
    s = ops.gemm_f32(q.to(dev), k.to(dev), b_is_nk=True, M=T, N=T, K=C, batch=Bn, lda=C, ldb=C, stride_a=T * C,
                     stride_b=T * C)
    assert _maxabs(s, q @ k.transpose(1, 2)) < 2e-4
    ops.softmax_rows_(s.view(-1, T), C ** -0.5)
    ref = torch.softmax((q @ k.transpose(1, 2)) * C ** -0.5, -1)
    assert _maxabs(s, ref) < 1e-6
    h = ops.gemm_f32(s, v.to(dev), b_is_nk=False, M=T, N=C, K=T, batch=Bn, lda=T, ldb=C, stride_a=T * T,
                     stride_b=T * C)
    assert _maxabs(h, ref @ v) < 1e-5


def test_gemm_f32_nested_batches_and_padded_softmax(dev):
    """Two-level batch (heads inside images) of the fp32 GEMM on a fused q|k|v buffer with a ragged token count: scores,
    softmax with zero-filled pad columns, P.V over the padded key count -- the SigLIP attention data path."""
    ops = _ops()
    torch.manual_seed(18)
    B, Hh, T, hd = 3, 4, 37, 24                       # T % 4 != 0: the padded-key path
    D = Hh * hd
    ldS, Tp = ops.round_up(T, 4), ops.round_up(T, 4)
    qkv = torch.randn(B * T + 4, 3 * D)
    qkv[B * T:] = 0
    qd = qkv.to(dev)
    s = torch.full((B, Hh, T, ldS), float("nan"), device=dev)
    ops.gemm_f32_nested(qd[:, 0:D], qd[:, D:2 * D], s, b_is_nk=True, M=T, N=T, K=hd, batch_in=Hh, batch_out=B, lda=3 * D,
                        ldb=3 * D, ldc=ldS, sa=(hd, T * 3 * D), sb=(hd, T * 3 * D), sc=(T * ldS, Hh * T * ldS))
    q = qkv[:B * T, 0:D].view(B, T, Hh, hd).transpose(1, 2)
    k = qkv[:B * T, D:2 * D].view(B, T, Hh, hd).transpose(1, 2)
    v = qkv[:B * T, 2 * D:].view(B, T, Hh, hd).transpose(1, 2)
    ref_s = q @ k.transpose(-1, -2)
    assert _maxabs(s[..., :T], ref_s) < 1e-4
    scale = hd ** -0.5
    ops.softmax_rows_(s.view(B * Hh * T, ldS), scale, cols=T)
    ref_p = torch.softmax(ref_s * scale, -1)
    assert _maxabs(s[..., :T], ref_p) < 1e-6 and s[..., T:].abs().max().item() == 0.0
    ctx = torch.empty(B * T, D, device=dev)
    ops.gemm_f32_nested(s, qd[:, 2 * D:], ctx, b_is_nk=False, M=T, N=hd, K=Tp, batch_in=Hh, batch_out=B, lda=ldS, ldb=3 * D,
                        ldc=D, sa=(T * ldS, Hh * T * ldS), sb=(hd, T * 3 * D), sc=(hd, T * D))
    ref_ctx = (ref_p @ v).transpose(1, 2).reshape(B * T, D)
    assert _maxabs(ctx, ref_ctx) < 1e-5


def test_lfq_and_layout(dev):
    ops = _ops()
    from oracle.ops_ref import lfq_indices_ref, lfq_entries_ref
    torch.manual_seed(9)
    z = torch.randn(3, 13, 16, 16); z[0, :, 0, 0] = 0.0
    idx = ops.lfq_pack(z.permute(0, 2, 3, 1).reshape(-1, 13).contiguous().to(dev), 13).view(3, 256)
    assert torch.equal(idx.cpu(), lfq_indices_ref(z))
    ent = ops.lfq_unpack(idx, 13).view(3, 16, 16, 13).permute(0, 3, 1, 2)
    assert torch.equal(ent.cpu(), lfq_entries_ref(idx.cpu(), 13))
    img = torch.randn(2, 3, 8, 8)
    nhwc = ops.nchw_to_nhwc(img.to(dev), 4)
    assert torch.equal(nhwc[..., :3].cpu(), img.permute(0, 2, 3, 1)) and nhwc[..., 3].abs().max().item() == 0
    assert torch.equal(ops.nhwc_to_nchw(nhwc, 3).cpu(), img)


@pytest.mark.parametrize("R,N,K", [(16, 2048, 1536), (5, 333, 256), (20, 1536, 8960), (32, 17920, 1536), (1, 64, 32)])
def test_decode_gemv_and_skinny_linear(dev, R, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(R * 31 + N + K)
    x = torch.randn(R, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    ref = x.float() @ w.float().t()
    y = ops.skinny_linear(x.to(dev), w.to(dev), bias=b.to(dev))
    assert _rel(y, ref + b.float()) < 4e-3
    res = torch.randn(R, N, generator=g)
    r2 = res.clone().to(dev)
    ops.skinny_linear(x.to(dev), w.to(dev), resid=r2)
    assert _rel(r2, res + ref.to(torch.bfloat16).float()) < 4e-3


@pytest.mark.gpu
@pytest.mark.parametrize("R,Tmax,length", [(16, 400, 331), (3, 96, 64), (5, 130, 1), (16, 1100, 1027)])
def test_attn_decode_matches_reference(dev, R, Tmax, length):
    """one-token query over the static KV cache == softmax(q K^T / sqrt(d)) V with GQA and a key-validity mask"""
    ops = _ops()
    H, HK, hd = 12, 2, 128
    g = torch.Generator().manual_seed(R + Tmax + length)
    q = torch.randn(R, H * hd, generator=g).to(torch.bfloat16)
    ck = torch.randn(R, HK, Tmax, hd, generator=g).to(torch.bfloat16)
    cv = torch.randn(R, HK, Tmax, hd, generator=g).to(torch.bfloat16)
    valid = (torch.rand(R, Tmax, generator=g) > 0.2).to(torch.uint8)
    valid[:, length - 1] = 1
    ln = torch.tensor([length], dtype=torch.int32)
    for kvd in (None, valid):
        o = ops.attn_decode(q.to(dev), ck.to(dev), cv.to(dev), None if kvd is None else kvd.to(dev), H, HK, hd, Tmax, ln.to(dev))
        qf = q.float().view(R, H, hd)
        kf = ck.float()[:, :, :length].repeat_interleave(H // HK, dim=1)
        vf = cv.float()[:, :, :length].repeat_interleave(H // HK, dim=1)
        s = torch.einsum("rhd,rhtd->rht", qf, kf) / math.sqrt(hd)
        if kvd is not None:
            s = s.masked_fill(kvd[:, None, :length] == 0, float("-inf"))
        ref = torch.einsum("rht,rhtd->rhd", torch.softmax(s, -1), vf).reshape(R, H * hd)
        assert _rel(o, ref) < 1e-2


@pytest.mark.gpu
@pytest.mark.parametrize("R", [16, 5, 24])
def test_decode_fused_finishers_match_unfused_kernels(dev, R):
    """the residual + RMSNorm finisher of the decode step is bit-identical to the separate kernels it replaces and re-zeroes
    the accumulator"""
    ops = _ops()
    Hq, Hk, hd, H, I, Tmax, pos = 12, 2, 128, 1536, 8960, 40, 17
    g = torch.Generator().manual_seed(R)
    cos, sin = ops.rope_tables(Tmax, hd, 1e6, dev)
    pos_dev = torch.tensor([pos], dtype=torch.int32, device=dev)
    lib = ops._l.load()
    # ---- residual + rmsnorm
    a = torch.randn(R, H, generator=g)
    x0 = torch.randn(R, H, generator=g)
    w = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    acc = a.clone().to(dev)
    x = x0.clone().to(dev)
    xn = torch.empty(R, H, dtype=torch.bfloat16, device=dev)
    ops.decode_finish_resid_norm_(acc, x, w, xn, 1e-6)
    x2 = x0.clone().to(dev)
    ops._l.check(lib.ug_skinny_finish(ops._p(a.to(dev).contiguous()), None, None, ops._p(x2), R, H, 1, ops._stream()), "fin")
    xn2, _ = ops.rmsnorm_fwd(x2, w, 1e-6, want_rstd=False)
    assert torch.equal(x, x2) and torch.equal(xn, xn2) and float(acc.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("R,N,K", [(16, 2048, 1536), (16, 17920, 1536), (16, 1536, 8960), (7, 1000, 160), (20, 520, 96)])
def test_gemv_row_major_accumulator(dev, R, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(R, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    acc = torch.zeros(R, N, device=dev)
    ops.gemv_acc_(x.to(dev), w.to(dev), acc)
    ref = x.float() @ w.float().t()
    assert _rel(acc, ref) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("R", [16, 5, 24])
def test_decode_consumer_side_finishing_matches_separate_kernels(dev, R):
    """The five-launch decode layer (raw accumulators finished by their consumers) against the separate kernels, on
    Qwen2.5-1.5B layer shapes; the scratch buffers are reused across iterations like in a captured step."""
    ops = _ops()
    Hq, Hk, hd, H, I, Tmax, pos = 12, 2, 128, 1536, 8960, 40, 17
    g = torch.Generator().manual_seed(100 + R)
    cos, sin = ops.rope_tables(Tmax, hd, 1e6, dev)
    pos_dev = torch.tensor([pos], dtype=torch.int32, device=dev)
    len_dev = torch.tensor([pos + 1], dtype=torch.int32, device=dev)
    nq = (Hq + 2 * Hk) * hd
    mk = lambda n, k: (torch.randn(n, k, generator=g) / math.sqrt(k)).to(torch.bfloat16).to(dev)
    wqkv, wo, wgu, wdown = mk(nq, H), mk(H, Hq * hd), mk(2 * I, H), mk(H, I)
    bias = torch.randn(nq, generator=g).to(torch.bfloat16).to(dev)
    ln1 = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    ln2 = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    z = lambda *shape: torch.zeros(shape, device=dev)
    acc_qkv, acc_o, acc_gu, acc_down, x_mid = z(R, nq), z(R, H), z(R, 2 * I), z(R, H), z(R, H)
    ss_attn, ss_mlp = z(32), z(32)
    ck0 = torch.randn(R, Hk, Tmax, hd, generator=g).to(torch.bfloat16).to(dev)
    cv0 = torch.randn(R, Hk, Tmax, hd, generator=g).to(torch.bfloat16).to(dev)
    valid = (torch.rand(R, Tmax, generator=g) > 0.2).to(torch.uint8).to(dev)
    valid[:, pos] = 1
    for it in range(4):
        x0 = torch.randn(R, H, generator=g).to(dev)
        pend = (0.3 * torch.randn(R, H, generator=g)).to(dev) if it else z(R, H)
        # ---- reference: separate kernels (the > 32-row decode path)
        xr = x0 + pend.to(torch.bfloat16).float()
        xn, _ = ops.rmsnorm_fwd(xr, ln1, 1e-6, want_rstd=False)
        qkv = ops.skinny_linear(xn, wqkv, bias=bias)
        ops.rope_at_(qkv, cos, sin, Hq + Hk, hd, pos_dev)
        ck2, cv2 = ck0.clone(), cv0.clone()
        ops.kv_store(qkv, ck2, cv2, R, 1, Hq, Hk, hd, Tmax, pos_dev, 0)
        o2 = ops.attn_decode(qkv, ck2, cv2, valid, Hq, Hk, hd, Tmax, len_dev)
        xr2 = xr.clone()
        ops.skinny_linear(o2, wo, resid=xr2)
        xn2, _ = ops.rmsnorm_fwd(xr2, ln2, 1e-6, want_rstd=False)
        act2 = ops.swiglu_fwd(ops.skinny_linear(xn2, wgu))
        xr3 = xr2.clone()
        ops.skinny_linear(act2, wdown, resid=xr3)
        # ---- five launches
        x = x0.clone()
        acc_down.copy_(pend)
        ck, cv = ck0.clone(), cv0.clone()
        o = torch.empty(R, Hq * hd, dtype=torch.bfloat16, device=dev)
        ops.decode_gemv_resid_norm_(x, acc_down, ln1, x_mid, ss_attn, wqkv, acc_qkv, zero0=acc_gu, ss_zero=ss_mlp)
        assert _rel(x_mid, xr) < 1e-6 and _rel(ss_attn[:R], (xr * xr).sum(-1)) < 1e-5
        ops.attn_decode_fused(acc_qkv, ss_attn, 1e-6, H, bias, cos, sin, pos_dev, ck, cv, valid, o, Hq, Hk, hd, Tmax)
        assert _rel(ck, ck2) < 1e-2 and _rel(cv, cv2) < 1e-2 and _rel(o, o2) < 2e-2
        ops.decode_gemv_(o, wo, acc_o, zero0=acc_qkv, zero1=acc_down, ss_zero=ss_attn)
        ops.decode_gemv_resid_norm_(x_mid, acc_o, ln2, x, ss_mlp, wgu, acc_gu)
        assert _rel(x, xr2) < 5e-3
        ops.decode_gemv_swiglu_(acc_gu, ss_mlp, 1e-6, H, wdown, acc_down, zero0=acc_o)
        hn = torch.empty(R, H, dtype=torch.bfloat16, device=dev)
        ops.decode_finish_resid_norm_(acc_down, x, ln1, hn, 1e-6)
        assert _rel(x, xr3) < 1e-2
        for t in (acc_qkv, acc_o, acc_down, ss_attn):
            assert float(t.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("N,n,V,cfg", [(3, 256, 8192, True), (2, 16, 20, True), (4, 64, 1000, False)])
def test_maskgit_step_kernel_matches_restatement(dev, N, n, V, cfg):
    """ug_maskgit_step against a float64 restatement of its rule (CFG mix, inverse-CDF draw from supplied uniforms,
    confidence + Gumbel, re-mask the k least confident); draws whose target lies within rounding distance of a CDF
    step are excluded from the exact comparison."""
    ops = _ops()
    g = torch.Generator().manual_seed(N * 1000 + n)
    rows = (2 if cfg else 1) * N * n
    logits = (2.0 * torch.randn(rows, V, generator=g)).to(torch.bfloat16)
    u1, u2 = torch.rand(N, n, generator=g), torch.rand(N, n, generator=g)
    mask_id, off, scale, temp, sched = V + 7, 300, 3.0, 0.6, n // 3
    cur = torch.full((N, n), mask_id, dtype=torch.int64)
    known = torch.rand(N, n, generator=g) < 0.3
    cur[known] = torch.randint(0, V, (int(known.sum()),), generator=g)
    s, nc, ni, mk = ops.maskgit_step(logits.to(dev), N, n, cfg, scale, u1.to(dev), u2.to(dev), cur.to(dev), mask_id, off, sched,
                                     temp, want_masking=True)
    s, nc, ni, mk = s.cpu(), nc.cpu(), ni.cpu(), mk.cpu()
    lf = logits.float()
    mixed = (scale * (lf[:N * n] - lf[N * n:]) + lf[N * n:]) if cfg else lf
    ex = torch.exp((mixed - mixed.max(-1, keepdim=True).values).double())
    cdf = ex.cumsum(-1)
    target = u1.reshape(-1, 1).double() * cdf[:, -1:]
    want = (cdf <= target).sum(-1).clamp(max=V - 1).view(N, n)
    gap = ((cdf - target).abs().min(-1).values / cdf[:, -1]).view(N, n)
    sure = (gap > 1e-4) & ~known
    assert sure.float().mean() > 0.5
    assert torch.equal(s[sure], want[sure]) and torch.equal(s[known], cur[known])
    # confidence / re-masking, teacher-forced on the kernel's own draws
    p = (ex / cdf[:, -1:]).view(N, n, V).gather(-1, s[..., None]).squeeze(-1)
    p = torch.where(known, torch.full_like(p, 3.4e38), p)
    conf = torch.log(p.clamp(min=1e-20)) + temp * -torch.log((-torch.log(u2.double().clamp(min=1e-20))).clamp(min=1e-20))
    k = torch.clamp(torch.minimum((~known).sum(-1, keepdim=True) - 1, torch.tensor(sched)), min=1)
    srt = conf.sort(-1).values
    thr = srt.gather(1, k)
    want_mask = conf < thr
    margin = (conf - thr).abs() > 1e-4
    assert torch.equal(mk[margin], want_mask[margin])
    assert torch.equal(nc, torch.where(mk, torch.tensor(mask_id), s)) and torch.equal(ni, torch.where(mk, torch.tensor(mask_id), s + off))
    assert int(mk.sum(-1).min()) >= 1 and not bool((mk & known).any())


def _unpack_bits(mb):
    """MaskBits -> bool [B, L, L]"""
    w = mb.bits.cpu()
    B, L, nW = w.shape
    cols = torch.arange(nW * 64)
    allow = ((w[:, :, cols // 64] >> (cols % 64)) & 1).bool()
    return allow[:, :, :L]


@pytest.mark.gpu
def test_masks_from_ids_match_reference_builders(dev):
    """ug_attn_mask_from_ids (no dense mask) against the reference builders' outputs (golden G4) and, on longer random
    sequences with pads / several tiles, against the oracle restatement that G4 pins."""
    from helpers import golden
    from oracle import host_ref
    ops = _ops()
    g = golden("g4_masks.pt")
    ids = g["ids"]
    for seq, want, mode in ((g["t2i_seq"], g["t2i_allow"], ops.MASK_T2I), (g["lm_seq"], g["lm_allow"], ops.MASK_LM),
                            (g["mmu_seq"], g["mmu_allow"], ops.MASK_MMU)):
        mb = ops.mask_from_ids(seq.to(dev), ids["pad"], ids["soi"], ids["eoi"], mode)
        assert torch.equal(_unpack_bits(mb), want), mode
    gen = torch.Generator().manual_seed(4)
    B, L, n = 5, 203, 64
    seq = torch.randint(0, 290, (B, L), generator=gen)
    seq[:, -(n + 2)] = ids["soi"]; seq[:, -1] = ids["eoi"]
    for b, npad in enumerate((0, 1, 70, 130, 136)):
        seq[b, :npad] = ids["pad"]
    for mode, ref in ((ops.MASK_T2I, host_ref.mask_predict_next_ref(seq, ids["pad"], ids["soi"], ids["eoi"], rm_pad_in_image=True)),
                      (ops.MASK_LM, host_ref.mask_predict_next_ref(seq, ids["pad"], ids["soi"], ids["eoi"])),
                      (ops.MASK_MMU, host_ref.mask_mmu_ref(seq, ids["eoi"]))):
        mb = ops.mask_from_ids(seq.to(dev), ids["pad"], ids["soi"], ids["eoi"], mode)
        assert torch.equal(_unpack_bits(mb), ref), mode
        dense = ops.mask_compress(host_ref.to_additive(ref).to(dev))
        assert torch.equal(mb.bits.cpu(), dense.bits.cpu()) and torch.equal(mb.tileany.cpu(), dense.tileany.cpu())


@pytest.mark.gpu
def test_t2i_assemble_matches_reference_layout(dev):
    """ug_t2i_assemble against the ids / labels the real UniversalPromptingQwen2.t2i_prompt produced (golden G2 layout)
    and against the oracle layout on ragged / truncated prompts."""
    from helpers import golden
    from oracle import host_ref
    ops = _ops()
    g = golden("g2_tiny_unigen.pt")
    lay, ids = g["layout"], g["ids"]
    got_ids, attn, got_lab = ops.t2i_assemble(lay["t2i_texts"], lay["t2i_in"].to(dev), lay["t2i_lab"].to(dev), lay["max_seq_len"],
                                              ids["pad"], ids["soi"], ids["eoi"], lay["conv_start"], lay["conv_end"])
    assert torch.equal(got_ids.cpu(), lay["ids_t2i"]) and torch.equal(got_lab.cpu(), lay["lab_t2i"])
    gen = torch.Generator().manual_seed(6)
    texts = [torch.randint(0, 290, (k,), generator=gen).tolist() for k in (0, 3, 40, 95, 200)]      # the last two are truncated
    n, L = 32, 96
    img = torch.randint(312, 332, (5, n), generator=gen)
    lab = torch.where(torch.rand(5, n, generator=gen) < 0.5, img, torch.full_like(img, -100))
    lab[0, 0] = ids["pad"]                                                # a label equal to the pad id becomes ignore
    want = host_ref.t2i_layout_ref(texts, img, lab, L, ids["pad"], ids["soi"], ids["eoi"], lay["conv_start"], lay["conv_end"])
    got = ops.t2i_assemble(texts, img.to(dev), lab.to(dev), L, ids["pad"], ids["soi"], ids["eoi"], lay["conv_start"], lay["conv_end"])
    for a, b in zip(got, want):
        assert torch.equal(a.cpu().long(), b.long())


@pytest.mark.gpu
@pytest.mark.parametrize("greedy", [True, False])
def test_ar_sample_kernel_matches_restatement(dev, greedy):
    """ug_ar_sample: CFG mix of bf16-rounded head logits, temperature, argmax / inverse-CDF draw, next-input embedding
    rows, accumulator cleared; against a float64 restatement (draws within rounding of a CDF step excluded)."""
    ops = _ops()
    bsz, V, H, n, P, off = 6, 8192, 256, 10, 40, 100
    g = torch.Generator().manual_seed(17 + greedy)
    acc = 3.0 * torch.randn(2 * bsz, V, generator=g)
    emb = torch.randn(off + V, H, generator=g)
    u = torch.rand(n, bsz, generator=g)
    step = 3
    pos = torch.tensor([P + step], dtype=torch.int32)
    tok = torch.zeros(bsz, 1, dtype=torch.long, device=dev)
    out = torch.zeros(bsz, n, dtype=torch.int32, device=dev)
    x = torch.zeros(2 * bsz, H, device=dev)
    acc_d = acc.clone().to(dev)
    ops.ar_sample_(acc_d, bsz, V, 4.0, 0.7, greedy, None if greedy else u.to(dev), pos.to(dev), P, n, emb.to(dev), off, tok, out, x)
    lf = acc.to(torch.bfloat16).float()
    mixed = ((lf[bsz:] + 4.0 * (lf[:bsz] - lf[bsz:])) / 0.7).double()
    got = tok[:, 0].cpu()
    if greedy:
        want = mixed.argmax(-1)
        top2 = mixed.topk(2, -1).values
        sure = (top2[:, 0] - top2[:, 1]) > 1e-4
    else:
        ex = torch.exp(mixed - mixed.max(-1, keepdim=True).values)
        cdf = ex.cumsum(-1)
        target = u[step].double().reshape(-1, 1) * cdf[:, -1:]
        want = (cdf <= target).sum(-1).clamp(max=V - 1)
        sure = ((cdf - target).abs().min(-1).values / cdf[:, -1]) > 1e-4
    assert sure.sum() >= bsz - 1 and torch.equal(got[sure], want[sure])
    assert torch.equal(out[:, step].cpu().long(), got) and int(out.cpu().abs().sum()) == int(got.sum())
    assert torch.equal(x[:bsz].cpu(), emb[got + off]) and torch.equal(x[bsz:].cpu(), emb[got + off])
    assert float(acc_d.abs().max()) == 0.0


@pytest.mark.gpu
def test_gemm_bf16_randomised_shapes_against_device_fp32(dev):
    """Seeded sweep over shapes that land on every dispatch path (128x128 single / two-stage, staggered 256x256, k-sliced
    tail, k-sliced small outputs), all layouts and epilogues, ragged M / N / K; reference = fp32 matmul on the device."""
    ops = _ops()
    ops.set_gemm_tile_policy(-1)
    rng = torch.Generator().manual_seed(2718)
    cases = [(4400, 4000, 3104), (2048, 1536, 5000), (1536, 1536, 4104), (2300, 1800, 2600), (5000, 2100, 1288), (12336, 1536, 2048),
             (777, 333, 1000), (3100, 3000, 640), (4608, 5120, 1296), (1200, 2500, 9000)]
    gen = torch.Generator(device=dev).manual_seed(1)
    for (M, N, K) in cases:
        mode = int(torch.randint(0, 3, (1,), generator=rng))
        ak, bk = [(False, False), (False, True), (True, True)][mode]
        if not ak and not bk:
            K = K // 8 * 8
        a = (torch.randn(M, K, device=dev, generator=gen) * 0.3).to(torch.bfloat16)
        b = (torch.randn(N, K, device=dev, generator=gen) * 0.3).to(torch.bfloat16)
        ref = a.float() @ b.float().t()
        A = a.t().contiguous() if ak else a
        B = b.t().contiguous() if bk else b
        if ak:       # k-major leading dimensions must be multiples of 8
            A = torch.nn.functional.pad(A, (0, (-M) % 8))
        if bk:
            B = torch.nn.functional.pad(B, (0, (-N) % 8))
        ldc = (N + 7) // 8 * 8
        out = torch.zeros(M, ldc, dtype=torch.bfloat16, device=dev)
        ops.gemm(A, B, out=out, M=M, N=N, K=K, a_kmajor=ak, b_kmajor=bk)
        assert _rel(out[:, :N], ref) < 4e-3, (M, N, K, mode)
        c32 = torch.full((M, ldc), 0.5, dtype=torch.float32, device=dev)
        for beta in (1, 0):
            ops.gemm(A, B, out=c32, M=M, N=N, K=K, a_kmajor=ak, b_kmajor=bk, epilogue=ops.UG_EPI_F32, beta=beta)
            want = ref + 0.5 if beta else ref
            assert _rel(c32[:, :N], want) < 1e-5 * math.sqrt(K) + 1e-6, (M, N, K, mode, beta)
            c32.fill_(0.5)


def test_gemm_k_sliced_on_two_streams_and_under_capture(dev):
    """Library contract (include/unigen_hip.h: ug_create): the k-sliced GEMM forms keep their partials in the calling
    stream's handle, so two streams running such GEMMs at once never share scratch; and no op entry point allocates or
    synchronises -- a GEMM of the k-sliced class issued on a fresh stream under hipGraph capture (no handle may be created
    there) runs in its plain form and replays correctly."""
    ops = _ops()
    ops.set_gemm_tile_policy(-1)
    M, N, K = 2048, 1536, 5000                      # attention-projection weight gradient class: every tile cut along K
    g = torch.Generator(device=dev).manual_seed(11)
    data = []
    for i in range(2):
        a = (torch.randn(K, M, device=dev, generator=g) * 0.5).to(torch.bfloat16)
        b = (torch.randn(K, N, device=dev, generator=g) * 0.5).to(torch.bfloat16)
        data.append((a, b, a.float().t() @ b.float(), torch.zeros(M, N, device=dev)))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    reps = 12
    for _ in range(reps):
        for s, (a, b, _, out) in zip(streams, data):
            with torch.cuda.stream(s):
                ops.gemm(a, b, out=out, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1)
    torch.cuda.synchronize()
    for a, b, ref, out in data:
        assert _rel(out, reps * ref) < 1e-5 * math.sqrt(K) + 1e-6
    handles = {ops._HANDLES[(s.device_index, s.cuda_stream)] for s in streams}
    assert len(handles) == 2 and 0 not in handles
    # under capture: a stream that has no handle yet gets none, the launch takes the un-sliced path, nothing allocates
    a, b, ref, _ = data[0]
    out = torch.zeros(M, N, device=dev)
    n_before = len(ops._HANDLES)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ops.gemm(a, b, out=out, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=1)
    assert len(ops._HANDLES) == n_before
    out.zero_()
    graph.replay(); graph.replay()
    torch.cuda.synchronize()
    assert _rel(out, 2 * ref) < 1e-5 * math.sqrt(K) + 1e-6


@pytest.mark.parametrize("scale_x,scale_w", [(1e4, 1.0), (1e-6, 1.0), (1.0, 1e5), (3e-5, 2e-4), (5e7, 1e-9)])
def test_split_linear_is_scale_invariant(dev, scale_x, scale_w):
    """The power-of-two operand scaling keeps the f16 split fp32-accurate whatever the tensors' magnitudes (fp16 alone covers
    6e-8 .. 65504): relative error vs fp64 stays at the fp32 level from 1e-9 to 5e7, including one 1000x outlier that sets
    the bound while every other value sits three decades below it."""
    ops = _ops()
    gen = torch.Generator().manual_seed(17)
    M, N, K = 192, 256, 1152
    x = torch.randn(M, K, generator=gen) * scale_x
    x[5, 7] = 1000.0 * scale_x
    W = torch.randn(N, K, generator=gen) / math.sqrt(K) * scale_w
    ref = x.double() @ W.double().t()
    ws, n_pad = ops.split_linear_weight(W.to(dev))
    got = ops.linear_split(x.to(dev), ws, n_pad, N).cpu().double()
    exact = ops.linear_f32(x.to(dev), W.to(dev)).cpu().double()
    rel = ((got - ref).norm() / ref.norm()).item()
    rel_exact = ((exact - ref).norm() / ref.norm()).item()
    print(f"    scales x={scale_x:g} w={scale_w:g}: split rel err {rel:.2e}, fp32 chain {rel_exact:.2e}")
    assert rel < 1e-6 and rel <= 3.0 * rel_exact + 1e-7, (rel, rel_exact)


def test_amax_and_saturating_bound(dev):
    """ug_amax_f32 (exact max|x| incl. ragged columns and strided rows), and the behaviour under a WRONG (too small) bound:
    values past the bound's binade saturate at the fp16 maximum instead of turning into infinities / NaNs."""
    ops = _ops()
    gen = torch.Generator().manual_seed(3)
    for rows, cols, pad in ((1, 5, 0), (37, 1152, 0), (129, 70, 6), (4, 3, 1), (1000, 4304, 0)):
        x = torch.randn(rows, cols + pad, generator=gen)
        x[rows // 2, cols // 2] = -77.5
        xd = x.to(dev)[:, :cols]
        assert float(ops.amax(xd)) == float(x[:, :cols].abs().max())
    assert float(ops.amax(torch.zeros(8, 8, device=dev))) == 0.0
    # the entry with its own memset (ug_amax_f32) on a slot full of garbage, and the pooled zeroed slots across a pool refill
    from unigen_hip import lib as ug_lib
    xd = torch.full((3, 7), -2.5, device=dev)
    out = torch.full((1,), 9.0e9, device=dev)
    ug_lib.check(ug_lib.load().ug_amax_f32(xd.data_ptr(), 3, 7, 7, out.data_ptr(), torch.cuda.current_stream().cuda_stream), "ug_amax_f32")
    assert float(out) == 2.5
    bounds = [ops.amax(torch.full((2, 4), float(i % 97 + 1), device=dev)) for i in range(1100)]
    assert [float(b) for b in bounds[::50]] == [float(i % 97 + 1) for i in range(0, 1100, 50)]
    M, N, K = 64, 128, 64
    x = torch.ones(M, K) * 3.0
    x[0, 0] = 1.0e6                                                   # 2^14 / 4 * 1e6 >> 65504 under the bound below
    W = torch.eye(N, K)
    ws, n_pad = ops.split_linear_weight(W.to(dev))
    y = ops.linear_split(x.to(dev), ws, n_pad, N, x_amax=ops.amax_const(4.0, dev)).cpu()
    assert torch.isfinite(y).all() and abs(float(y[1, 1]) - 3.0) < 1e-5 and float(y[0, 0]) < 1.0e6


@pytest.mark.parametrize("B,T,H,hd", [(2, 729, 16, 72), (1, 16, 2, 72), (3, 130, 4, 64), (1, 65, 3, 40), (2, 200, 2, 80)])
def test_siglip_fused_attention_matches_fp64(dev, B, T, H, hd):
    """ug_siglip_attn_f32 (flash-style, scaled two-way f16 split in both contractions) vs fp64 softmax attention and vs the
    materialised-scores path (fp32 MFMA GEMM -> row softmax -> GEMM): fp32-level accuracy, ragged last key tile, head_dim
    not a multiple of 16 or 32, large / small operand magnitudes."""
    ops = _ops()
    gen = torch.Generator().manual_seed(B * 1000 + T + hd)
    D = H * hd
    for mag in (1.0, 300.0, 1e-3):
        qkv = torch.randn(B * T, 3 * D, generator=gen) * mag
        qkv[:, :D] *= 2.0 / mag if mag != 1.0 else 1.0              # keep the scores O(10) so softmax is not degenerate
        scale = hd ** -0.5 / (mag if mag != 1.0 else 1.0) * (0.5 if mag != 1.0 else 1.0)
        q = qkv[:, :D].double().view(B, T, H, hd).transpose(1, 2)
        k = qkv[:, D:2 * D].double().view(B, T, H, hd).transpose(1, 2)
        v = qkv[:, 2 * D:].double().view(B, T, H, hd).transpose(1, 2)
        ref = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).transpose(1, 2).reshape(B * T, D)
        out = torch.empty(B * T, D, device=dev)
        ops.siglip_attn(qkv.to(dev), out, B, T, H, hd, scale)
        err = (out.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
        print(f"    siglip attn B={B} T={T} H={H} hd={hd} |x|~{mag:g}: max err / max|ref| {err:.2e}")
        assert err < 3e-6, (mag, err)


@pytest.mark.parametrize("M,I,K", [(12336, 8960, 1536), (4000, 4096, 512), (771, 512, 256), (300, 8960, 1536), (5000, 1000, 128), (9288, 8960, 64),
                                   (1542, 1024, 96), (200, 2048, 40)])
def test_gemm_swiglu_fused_epilogue_is_bit_identical(dev, M, I, K):
    """ug_gemm_bf16_swiglu (gate_up projection whose epilogue also writes act = bf16(bf16(silu(gate)) * up); a 128 ... 320 x 256 tile =
    128 gate + the same hidden units' 128 up weight rows, gate and up of a unit in one lane) against the two-launch form GEMM ->
    ug_swiglu_fwd: gu and act bit-identical -- fused path (ragged M, several tile heights) and the fallback shapes alike."""
    ops = _ops()
    g = torch.Generator(device=dev).manual_seed(M + I)
    x = (torch.randn(M, K, device=dev, generator=g)).to(torch.bfloat16)
    w = (torch.randn(2 * I, K, device=dev, generator=g) * (K ** -0.5)).to(torch.bfloat16)
    ops.set_gemm_tile_policy(3)                     # the reference GEMM on the 256 x 256 kernel: another kernel, the same values
    try:
        gu_ref = ops.gemm(x, w)
    finally:
        ops.set_gemm_tile_policy(-1)
    act_ref = ops.swiglu_fwd(gu_ref)
    was, ops.FUSED_SWIGLU = ops.FUSED_SWIGLU, True
    try:
        gu, act = ops.gemm_swiglu(x, w)
    finally:
        ops.FUSED_SWIGLU = was
    assert gu.shape == (M, 2 * I) and act.shape == (M, I)
    assert torch.equal(gu, gu_ref) and torch.equal(act, act_ref)
    ref32 = x.float() @ w.float().t()
    assert _rel(gu, ref32) < 4e-3


@pytest.mark.gpu
@pytest.mark.parametrize("R", [16, 5, 1])
def test_decode_single_writer_projections_match_separate_kernels(dev, R):
    """csrc/decode_sw.hip (round 6): the o projection (finished residual add), gate/up + SwiGLU and the head slice as single-writer
    launches on Qwen2.5-1.5B shapes -- against the separate kernels they stand for (RMSNorm -> GEMV -> SwiGLU / residual add) and a
    fp32 evaluation; the pending-accumulator input must equal feeding the folded stream (bit for bit), and a launch repeated on the
    same input must reproduce itself bit for bit (no atomics)."""
    ops = _ops()
    assert ops.decode_sw_supported(1536, 8960, 1536, 128) and not ops.decode_sw_supported(896, 4864, 896, 64)
    H, I, V = 1536, 8960, 8192
    g = torch.Generator().manual_seed(600 + R)
    mk = lambda n, k: (torch.randn(n, k, generator=g) / math.sqrt(k)).to(torch.bfloat16).to(dev)
    wo, wgu, whead = mk(H, H), mk(2 * I, H), mk(V, H)
    ln = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    h0 = torch.randn(R, H, generator=g)
    o = torch.randn(R, H, generator=g).to(torch.bfloat16)
    eps = 1e-6
    # ---- o projection: h += float(bf16(o W^T)), exactly the split-K GEMV + finisher up to the fp32 summation order
    h = h0.clone().to(dev)
    ops.decode_sw_resid_(o.to(dev), wo, h)
    ref = h0 + (o.float() @ wo.float().cpu().t()).to(torch.bfloat16).float()
    assert _rel(h, ref) < 2e-3, _rel(h, ref)
    h_again = h0.clone().to(dev)
    ops.decode_sw_resid_(o.to(dev), wo, h_again)
    assert torch.equal(h, h_again)
    # ---- gate/up: RMSNorm as the reference applies it, SwiGLU in the epilogue
    act = torch.empty(R, I, dtype=torch.bfloat16, device=dev)
    ops.decode_sw_gate_up_(h, ln, eps, wgu, act)
    xn, _ = ops.rmsnorm_fwd(h, ln, eps, want_rstd=False)                       # the training path's kernel: same rounding points
    gu = ops.skinny_linear(xn, wgu)
    act_sep = ops.swiglu_fwd(gu)
    assert _rel(act, act_sep.float()) < 6e-3, _rel(act, act_sep.float())       # (one bf16 ulp where a gate / up sum lands on the other side)
    hf = h.float().cpu()
    xr = (ln.cpu() * (hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + eps))).to(torch.bfloat16).float()
    guf = (xr @ wgu.float().cpu().t()).to(torch.bfloat16).float()
    gate, up = guf[:, :I], guf[:, I:]
    act_ref = (torch.nn.functional.silu(gate).to(torch.bfloat16).float() * up).to(torch.bfloat16).float()
    assert _rel(act, act_ref) < 6e-3, _rel(act, act_ref)
    act2 = torch.empty_like(act)
    ops.decode_sw_gate_up_(h, ln, eps, wgu, act2)
    assert torch.equal(act, act2)
    # ---- pending accumulator: gate_up(h, pend) == gate_up(h + float(bf16(pend))), x_out = that sum
    pend = (0.5 * torch.randn(R, H, generator=g)).to(dev)
    folded = h + pend.to(torch.bfloat16).float()
    x_out = torch.zeros_like(h)
    act3, act4 = torch.empty_like(act), torch.empty_like(act)
    ops.decode_sw_gate_up_(h, ln, eps, wgu, act3, pend=pend, x_out=x_out)
    ops.decode_sw_gate_up_(folded, ln, eps, wgu, act4)
    assert torch.equal(act3, act4) and torch.equal(x_out, folded)
    # ---- head slice (+ position advance)
    logits = torch.zeros(R, V, device=dev)
    pos = torch.tensor([7], dtype=torch.int32, device=dev)
    ln_dev = torch.tensor([8], dtype=torch.int32, device=dev)
    ops.decode_sw_head_(h, ln, eps, whead, logits, pend=pend, advance=(pos, ln_dev))
    ff = folded.float().cpu()
    xr = (ln.cpu() * (ff * torch.rsqrt(ff.pow(2).mean(-1, keepdim=True) + eps))).to(torch.bfloat16).float()
    assert _rel(logits, xr @ whead.float().cpu().t()) < 2e-3
    assert int(pos.item()) == 8 and int(ln_dev.item()) == 9
    logits2 = torch.zeros(R, V, device=dev)
    ops.decode_sw_head_(folded, ln, eps, whead, logits2)
    assert torch.equal(logits, logits2) and int(pos.item()) == 8


@pytest.mark.gpu
def test_decode_single_writer_rejects_what_it_was_not_built_for(dev):
    ops = _ops()
    from unigen_hip.lib import UniGenHipError
    w = torch.zeros(64, 1024, dtype=torch.bfloat16, device=dev)
    with pytest.raises(UniGenHipError):
        ops.decode_sw_resid_(torch.zeros(4, 1024, dtype=torch.bfloat16, device=dev), w, torch.zeros(4, 64, device=dev))      # K != 1536
    with pytest.raises(UniGenHipError):
        ops.decode_sw_gate_up_(torch.zeros(17, 1536, device=dev), torch.ones(1536, device=dev), 1e-6,
                               torch.zeros(64, 1536, dtype=torch.bfloat16, device=dev), torch.zeros(17, 32, dtype=torch.bfloat16, device=dev))   # 17 rows
    h = torch.zeros(4, 1536, device=dev)
    with pytest.raises(UniGenHipError):                                                                                      # x_out must not alias h
        ops.decode_sw_gate_up_(h, torch.ones(1536, device=dev), 1e-6, torch.zeros(64, 1536, dtype=torch.bfloat16, device=dev),
                               torch.zeros(4, 32, dtype=torch.bfloat16, device=dev), pend=torch.zeros(4, 1536, device=dev), x_out=h)


@pytest.mark.gpu
@pytest.mark.parametrize("R,N,K", [(16, 1536, 8960), (5, 100, 1792), (16, 48, 3584)])
def test_decode_kblock_projection_accumulates_and_clears(dev, R, N, K):
    """ug_decode_sw_kblock (the decode step's down projection): acc += x W^T in k-blocks of 1 792 with the seven partial tiles of a
    workgroup pre-reduced in LDS; the clears it carries; a contraction that is not a whole number of k-blocks is refused."""
    ops = _ops()
    from unigen_hip.lib import UniGenHipError
    g = torch.Generator().manual_seed(N + K)
    x = torch.randn(R, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    acc0 = torch.randn(R, N, generator=g)
    acc = acc0.clone().to(dev)
    z0, z1, ss = torch.ones(16, 2048, device=dev), torch.ones(R, 1536, device=dev), torch.ones(32, device=dev)
    ops.decode_sw_kblock_(x.to(dev), w.to(dev), acc, zero0=z0, zero1=z1, ss_zero=ss)
    ref = acc0 + x.float() @ w.float().t()
    assert _rel(acc, ref) < 1e-5, _rel(acc, ref)
    assert float(z0.abs().max()) == 0.0 and float(z1.abs().max()) == 0.0 and float(ss.abs().max()) == 0.0
    with pytest.raises(UniGenHipError):
        ops.decode_sw_kblock_(x[:, :1536].contiguous().to(dev), w[:, :1536].contiguous().to(dev), acc)
