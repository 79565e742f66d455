"""Qwen2.5 backbone engine on the gfx950 kernels: flat parameter storage, decoder-stack forward /
backward, tied lm_head + cross-entropy.  This is the arithmetic behind `UniGen.llm`
(reference: models/unigen.py:56-69 builds transformers' Qwen2ForCausalLM; call sites :274-287).

Memory plan (sized for 288 GB HBM3E, no activation recompute):
  * ONE flat fp32 buffer holds every parameter (q/k/v and gate/up stored fused so one GEMM serves
    each), ONE flat fp32 gradient buffer mirrors it (this is also what the data-parallel
    all-reduce moves), ONE flat bf16 compute copy plus per-matrix transposed bf16 copies feed the
    matrix cores.  `nn.Parameter`s with the reference checkpoint's names are views into the flat
    buffers, so optimizers / state_dict / named_parameters() see the reference layout.
  * precision = the reference's accelerate-DDP bf16-autocast mode: fp32 master weights and residual
    stream, bf16 GEMM operands, fp32 accumulation / statistics / gradients.
"""
import math
import os

import types

import torch
import torch.nn as nn

from . import ops
from .lib import UniGenHipError


class Qwen2Dims:
    def __init__(self, vocab_size, hidden_size=1536, intermediate_size=8960, num_hidden_layers=28,
                 num_attention_heads=12, num_key_value_heads=2, rope_theta=1e6, rms_norm_eps=1e-6,
                 initializer_range=0.02, rope_scaling=None, max_position_embeddings=32768, **_unused):
        self.vocab_size = int(vocab_size)
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.num_hidden_layers = num_hidden_layers
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.head_dim = hidden_size // num_attention_heads
        self.rope_theta = float(rope_theta)
        self.rope_scaling = None
        if rope_scaling:
            self.rope_scaling = dict(rope_scaling, max_position_embeddings=int(max_position_embeddings))
        self.rms_norm_eps = float(rms_norm_eps)
        self.initializer_range = initializer_range
        if self.head_dim != 128:
            raise UniGenHipError(f"attention kernels are specialised for head_dim 128 (got {self.head_dim})")
        if hidden_size % 64 or intermediate_size % 64:
            raise UniGenHipError("hidden / intermediate sizes must be multiples of 64 for the MFMA GEMM")
        self.vocab_pad = ops.round_up(self.vocab_size, 64)
        self.qkv_out = (num_attention_heads + 2 * num_key_value_heads) * self.head_dim


class FlatParams:
    """Flat fp32 master / grad buffers + bf16 compute copies, with named views."""

    ALIGN = 64

    def __init__(self, dims, device):
        d, self.dims, self.device = dims, dims, device
        H, I = d.hidden_size, d.intermediate_size
        self.spec = []   # (key, shape)
        self.spec.append(("embed", (d.vocab_size, H)))
        for i in range(d.num_hidden_layers):
            self.spec += [(f"l{i}.wqkv", (d.qkv_out, H)), (f"l{i}.bqkv", (d.qkv_out,)), (f"l{i}.wo", (H, H)),
                          (f"l{i}.wgu", (2 * I, H)), (f"l{i}.wdown", (H, I)), (f"l{i}.ln1", (H,)), (f"l{i}.ln2", (H,))]
        self.spec.append(("norm", (H,)))
        self.off = {}
        n = 0
        for k, shp in self.spec:
            self.off[k] = (n, shp)
            n += ops.round_up(math.prod(shp), self.ALIGN)
        self.numel = n
        self.master = torch.zeros(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device)
        self.bf16 = torch.zeros(n, dtype=torch.bfloat16, device=device)
        self._seen_version = -1
        self.pending_update = None                 # event of an optimizer update running on a side stream (FusedAdamW overlap)
        ops.register_bf16_mirror(self)
        # gradients of the big per-layer matrices are written by exactly one GEMM per backward pass: after a
        # zero_grad they are not cleared but marked "fresh", and that GEMM overwrites (beta = 0) instead of adding
        self.fresh = set()
        # the tied embedding's gradient is first written by the lm-head weight-gradient GEMM (every vocabulary row), so it is
        # "fresh" too: 983 MB that are neither zero-filled nor read back; writers that only ADD call ensure_zeroed first
        self._big = [k for k, shp in self.spec if len(shp) == 2]
        big = set(self._big)
        self._small_runs = []                      # maximal runs of the remaining (small / multi-writer) tensors
        run = None
        for k, shp in self.spec:
            o, _ = self.off[k]
            end = o + ops.round_up(math.prod(shp), self.ALIGN)
            if k in big:
                run = None
            elif run is None:
                run = [o, end]
                self._small_runs.append(run)
            else:
                run[1] = end
        self._small_table = torch.tensor([[lo, hi - lo] for lo, hi in self._small_runs], dtype=torch.int64, device=device)
        self._small_max = max(hi - lo for lo, hi in self._small_runs)

    def view(self, buf, key):
        o, shp = self.off[key]
        return buf[o:o + math.prod(shp)].view(shp)

    def p(self, key):
        return self.view(self.master, key)

    def g(self, key):
        return self.view(self.grad, key)

    def wait_pending_update(self):
        """Order the current stream behind an overlapped optimizer update before it touches weights or gradients."""
        if self.pending_update is not None:
            torch.cuda.current_stream().wait_event(self.pending_update)
            self.pending_update = None

    def clear_grads(self):
        """Start of a gradient pass after zero_grad: clear what is accumulated into (embedding, norms, biases) and
        mark the big matrices fresh (their single weight-gradient GEMM overwrites them): 5.2 of the 6.2 GB are never
        zero-filled nor read back."""
        self.wait_pending_update()
        ops.zero_ranges_(self.grad, self._small_table, self._small_max)     # one launch instead of one fill per run
        self.fresh = set(self._big)

    def beta_for(self, key):
        """1 (accumulate) or 0 (first and only write of this pass) for the weight-gradient GEMM of `key`."""
        if key in self.fresh:
            self.fresh.discard(key)
            return 0
        return 1

    def ensure_zeroed(self, key):
        """For writers that accumulate into part of `key` (embedding scatter-add, a vocabulary slice of the head): if no
        full overwrite has happened yet in this pass, clear it now."""
        if key in self.fresh:
            self.fresh.discard(key)
            self.g(key).zero_()

    def flush_fresh(self):
        """End of the backbone's backward: a matrix that received no gradient in this pass must read as zero."""
        for k in self.fresh:
            self.g(k).zero_()
        self.fresh = set()

    def w(self, key):
        return self.view(self.bf16, key)

    def refresh_compute_copies(self, force=False):
        """bf16 compute copy of the fp32 master weights (one HBM pass per optimizer step; dgrad / wgrad read
        the same copy through the GEMM's k-major operand mode, so no transposed copies exist)."""
        self.wait_pending_update()
        ver = self.master._version
        if not force and ver == self._seen_version:
            return
        ops.cast_bf16(self.master, self.bf16)
        self._seen_version = ver


def _hf_name_map(dims):
    """reference checkpoint key (under `llm.`) -> (flat key, row slice | None)."""
    d = dims
    hd, Hq, Hk = d.head_dim, d.num_attention_heads, d.num_key_value_heads
    q_end, k_end = Hq * hd, (Hq + Hk) * hd
    m = {"model.embed_tokens.weight": ("embed", None), "model.norm.weight": ("norm", None)}
    for i in range(d.num_hidden_layers):
        p = f"model.layers.{i}."
        m[p + "self_attn.q_proj.weight"] = (f"l{i}.wqkv", (0, q_end))
        m[p + "self_attn.k_proj.weight"] = (f"l{i}.wqkv", (q_end, k_end))
        m[p + "self_attn.v_proj.weight"] = (f"l{i}.wqkv", (k_end, d.qkv_out))
        m[p + "self_attn.q_proj.bias"] = (f"l{i}.bqkv", (0, q_end))
        m[p + "self_attn.k_proj.bias"] = (f"l{i}.bqkv", (q_end, k_end))
        m[p + "self_attn.v_proj.bias"] = (f"l{i}.bqkv", (k_end, d.qkv_out))
        m[p + "self_attn.o_proj.weight"] = (f"l{i}.wo", None)
        m[p + "mlp.gate_proj.weight"] = (f"l{i}.wgu", (0, d.intermediate_size))
        m[p + "mlp.up_proj.weight"] = (f"l{i}.wgu", (d.intermediate_size, 2 * d.intermediate_size))
        m[p + "mlp.down_proj.weight"] = (f"l{i}.wdown", None)
        m[p + "input_layernorm.weight"] = (f"l{i}.ln1", None)
        m[p + "post_attention_layernorm.weight"] = (f"l{i}.ln2", None)
    return m


class _Saved:
    __slots__ = ("h", "rstd1", "xn1", "qkv", "o", "lse", "h_mid", "rstd2", "xn2", "gu", "act")


class Qwen2Engine:
    """Decoder stack + head on the HIP kernels.  Not an nn.Module: the module tree lives in
    `modules.py`; this object owns buffers and orchestrates kernel launches."""

    def __init__(self, dims, device):
        self.dims, self.device = dims, device
        self.fp = FlatParams(dims, device)
        self._rope_cache = {}
        self.err_flag = torch.zeros(1, dtype=torch.int32, device=device)
        self.grad_ready_hook = None      # callable(layer_index | 'embed' | 'norm') fired as grads complete (DDP)

    # ---------------------------------------------------------------- helpers
    def rope(self, L):
        if L not in self._rope_cache:
            self._rope_cache[L] = ops.rope_tables(L, self.dims.head_dim, self.dims.rope_theta, self.device, self.dims.rope_scaling)
        return self._rope_cache[L]

    def check_errors(self):
        """Device-side error flags (out-of-range token id = 1, non-binary attention mask = 2, bad VQ code = 4)."""
        v = int(self.err_flag.item())
        if v:
            self.err_flag.zero_()
            raise UniGenHipError(f"device-side input check failed (flags={v}): 1=token id out of range, "
                                 f"2=attention mask value neither 0 nor <= -1e9, 4=VQ code out of range")

    # ---------------------------------------------------------------- forward
    def layer_fwd(self, i, h, mb, L, save):
        d, fp = self.dims, self.fp
        Hq, Hk, hd = d.num_attention_heads, d.num_key_value_heads, d.head_dim
        cos, sin = self.rope(L)
        xn1, rstd1 = ops.rmsnorm_fwd(h, fp.p(f"l{i}.ln1"), d.rms_norm_eps)
        qkv = ops.gemm_qkv_rope(xn1, fp.w(f"l{i}.wqkv"), fp.w(f"l{i}.bqkv"), cos, sin, L, Hq + Hk, hd)     # projection + RoPE, one launch
        o, lse = ops.attn_fwd(qkv, mb, Hq, Hk, hd)
        h_mid = ops.gemm_nt(o, fp.w(f"l{i}.wo"), epilogue=ops.UG_EPI_RESID, resid=h)
        xn2, rstd2 = ops.rmsnorm_fwd(h_mid, fp.p(f"l{i}.ln2"), d.rms_norm_eps)
        gu, act = ops.gemm_swiglu(xn2, fp.w(f"l{i}.wgu"))           # projection + SwiGLU in its epilogue (one launch; UNIGEN_FUSED_SWIGLU=0: two)
        h_out = ops.gemm_nt(act, fp.w(f"l{i}.wdown"), epilogue=ops.UG_EPI_RESID, resid=h_mid)
        if save is not None:
            s = _Saved()
            s.h, s.rstd1, s.xn1, s.qkv, s.o, s.lse = h, rstd1, xn1, qkv, o, lse
            s.h_mid, s.rstd2, s.xn2, s.gu, s.act = h_mid, rstd2, xn2, gu, act
            save.append(s)
        return h_out

    def stack_fwd(self, h0, mb, L, save):
        """h0 fp32 [B*L, H] -> (h_last fp32, hn bf16 = final-norm output, rstd of the final norm)."""
        self.fp.refresh_compute_copies()
        h = h0
        for i in range(self.dims.num_hidden_layers):
            h = self.layer_fwd(i, h, mb, L, save)
        hn, rstd = ops.rmsnorm_fwd(h, self.fp.p("norm"), self.dims.rms_norm_eps)
        return h, hn, rstd

    # ---------------------------------------------------------------- incremental MaskGIT rounds
    def maskgit_begin(self, h0, mb, L, seg_start):
        """First round of UniGen.t2i_generate: a full forward that keeps every layer's (post-RoPE) q/k/v.  The rows before
        `seg_start` (padding + text) are causal and never see the image segment, so their keys / values are the same in
        every later round; only the segment rows [seg_start, L) are recomputed by maskgit_step.
        h0 fp32 [B*L, H] -> (session, final-norm hidden of the segment rows bf16 [B*S, H])."""
        self.fp.refresh_compute_copies()
        B = h0.shape[0] // L
        saved, h = [], h0
        for i in range(self.dims.num_hidden_layers):
            h = self.layer_fwd(i, h, mb, L, saved)
        sess = types.SimpleNamespace()
        sess.qkv = [s.qkv for s in saved]
        sess.mb, sess.B, sess.L, sess.P, sess.S = mb, B, L, seg_start, L - seg_start
        r = torch.arange(seg_start, L, device=self.device)
        sess.rows = (torch.arange(B, device=self.device)[:, None] * L + r[None, :]).reshape(-1).contiguous()
        cos, sin = self.rope(L)
        sess.cos, sess.sin = cos[seg_start:].contiguous(), sin[seg_start:].contiguous()
        hn, _ = ops.rmsnorm_fwd(h.view(B, L, -1)[:, seg_start:].reshape(B * sess.S, -1).contiguous(), self.fp.p("norm"),
                                self.dims.rms_norm_eps, want_rstd=False)
        return sess, hn

    def maskgit_step(self, sess, seg):
        """seg fp32 [B*S, H] = embeddings of the segment rows this round -> final-norm hidden bf16 [B*S, H].  All GEMMs and
        norms run on the B*S segment rows only; attention runs over the full sequence from the per-layer q/k/v buffers
        (prefix rows cached, segment rows refreshed in place)."""
        d, fp = self.dims, self.fp
        Hq, Hk, hd = d.num_attention_heads, d.num_key_value_heads, d.head_dim
        h = seg
        for i in range(d.num_hidden_layers):
            xn1, _ = ops.rmsnorm_fwd(h, fp.p(f"l{i}.ln1"), d.rms_norm_eps, want_rstd=False)
            qkv_s = ops.gemm_qkv_rope(xn1, fp.w(f"l{i}.wqkv"), fp.w(f"l{i}.bqkv"), sess.cos, sess.sin, sess.S, Hq + Hk, hd)   # row r sits at position seg_start + r % S
            ops.scatter_rows_(qkv_s, sess.rows, sess.qkv[i])
            o, _ = ops.attn_fwd(sess.qkv[i], sess.mb, Hq, Hk, hd)
            o_s = ops.gather_rows(o, sess.rows)
            h_mid = ops.gemm_nt(o_s, fp.w(f"l{i}.wo"), epilogue=ops.UG_EPI_RESID, resid=h)
            xn2, _ = ops.rmsnorm_fwd(h_mid, fp.p(f"l{i}.ln2"), d.rms_norm_eps, want_rstd=False)
            _, act = ops.gemm_swiglu(xn2, fp.w(f"l{i}.wgu"))
            h = ops.gemm_nt(act, fp.w(f"l{i}.wdown"), epilogue=ops.UG_EPI_RESID, resid=h_mid)
        hn, _ = ops.rmsnorm_fwd(h, fp.p("norm"), d.rms_norm_eps, want_rstd=False)
        return hn

    # ---------------------------------------------------------------- backward
    def layer_bwd(self, i, s, dh, mb, L, dh_bf16=None):
        """dh fp32 [M,H]: grad w.r.t. the layer output on entry, w.r.t. the layer input on exit (in place).
        Weight gradients accumulate into the flat fp32 grad buffer.  dh_bf16: bf16(dh) if the producer already has it.
        Returns (dh, bf16(dh)): every RMSNorm backward also emits the bf16 operand of the GEMMs that follow."""
        d, fp = self.dims, self.fp
        Hq, Hk, hd = d.num_attention_heads, d.num_key_value_heads, d.head_dim
        cos, sin = self.rope(L)
        F32 = ops.UG_EPI_F32
        # ---- MLP   (wgrad: both operands k-major over the token axis; dgrad: weight read k-major)
        # The four weight gradients (dW = dY^T X, both operands token-major) are leaves of the backward graph: they are
        # collected and issued as ONE grouped launch at the end of the layer (3 rounds of the chip instead of 2 + 1 + two
        # k-sliced launches, ops.gemm_wgrad_group).
        wg = lambda key, dy, x: (dy, x, fp.g(key), fp.beta_for(key))
        dyd = dh_bf16 if dh_bf16 is not None else ops.cast_bf16(dh)
        w_down = wg(f"l{i}.wdown", dyd, s.act)
        dgu = ops.gemm_swiglu_bwd(dyd, fp.w(f"l{i}.wdown"), s.gu)          # down dgrad + SwiGLU backward in its epilogue: d(act) never stored
        w_gu = wg(f"l{i}.wgu", dgu, s.xn2)
        dxn2 = ops.gemm(dgu, fp.w(f"l{i}.wgu"), b_kmajor=True)
        dyo = ops.rmsnorm_bwd(dxn2, s.h_mid, s.rstd2, fp.p(f"l{i}.ln2"), dh, fp.g(f"l{i}.ln2"), want_bf16=True)
        # ---- attention
        w_o = wg(f"l{i}.wo", dyo, s.o)
        do = ops.gemm(dyo, fp.w(f"l{i}.wo"), b_kmajor=True)
        # RoPE transposed and the bias gradient (column sums) where the attention backward stores dq / dk / dv
        dqkv = ops.attn_bwd(s.qkv, s.o, s.lse, do, mb, Hq, Hk, hd, rope=(cos, sin), dbias=fp.g(f"l{i}.bqkv"))
        w_qkv = wg(f"l{i}.wqkv", dqkv, s.xn1)
        dxn1 = ops.gemm(dqkv, fp.w(f"l{i}.wqkv"), b_kmajor=True)
        dnext = ops.rmsnorm_bwd(dxn1, s.h, s.rstd1, fp.p(f"l{i}.ln1"), dh, fp.g(f"l{i}.ln1"), want_bf16=True)
        # ... plus a slice of the tied head's weight gradient when one is waiting (head_bwd): its 128-k-tile tiles run on the CUs
        # the group's partial third round (714 tiles = 2.79 rounds) leaves idle, so most of the head's 1.7 ms weight-gradient
        # launch disappears from the step
        self._wgrad_tokens = dyd.shape[0]
        ops.gemm_wgrad_group([w_gu, w_down, w_qkv, w_o] + self._head_wgrad_slice())
        return dh, dnext

    def stack_bwd(self, saved, h_last, rstd_last, dhn, mb, L):
        """dhn bf16 [M,H] (grad of the final-norm output) -> dh0 fp32 [M,H]."""
        dh = torch.zeros_like(h_last)
        dh16 = ops.rmsnorm_bwd(dhn, h_last, rstd_last, self.fp.p("norm"), dh, self.fp.g("norm"), want_bf16=True)
        if self.grad_ready_hook:
            self.grad_ready_hook("norm")
        for i in reversed(range(self.dims.num_hidden_layers)):
            dh, dh16 = self.layer_bwd(i, saved[i], dh, mb, L, dh16)
            saved[i] = None                       # release this layer's activations
            if self.grad_ready_hook:
                self.grad_ready_hook(i)
        self.flush_deferred_head()
        self.fp.flush_fresh()
        return dh

    # ---------------------------------------------------------------- head
    def logits_rows(self, hn_rows):
        """hn_rows bf16 [R, H] -> logits bf16 [R, vocab_pad] (columns >= vocab_size are unspecified)."""
        d = self.dims
        out = torch.empty((hn_rows.shape[0], d.vocab_pad), dtype=torch.bfloat16, device=self.device)
        return ops.gemm_nt(hn_rows, self.fp.w("embed"), out=out, N=d.vocab_size, K=d.hidden_size)

    def head_bwd(self, dlogits, hn_rows):
        """dlogits bf16 [R, vocab_pad] (pad columns zero) -> dhn_rows bf16 [R,H]; embed grad accumulated."""
        d, fp = self.dims, self.fp
        R = hn_rows.shape[0]
        sync = getattr(self, "grad_sync", None)
        if sync is not None:
            sync.before_dense_embed_write()       # (data parallel: a table already handed over is waited for and re-exchanged)
        # dW[V,H] += dlogits^T hn : both k-major over the R selected rows (dlogits' leading dim is vocab_pad)
        if self._may_defer_head_wgrad(R):
            # a leaf of the backward graph: handed to the decoder layers' grouped weight-gradient launches slice by slice
            # (layer_bwd), whatever is left is launched by flush_deferred_head before any other writer of the table runs
            self.flush_deferred_head()
            self._deferred_head = [dlogits, hn_rows, fp.beta_for("embed"), 0]
            torch.autograd.Variable._execution_engine.queue_callback(self.flush_deferred_head)
        else:
            ops.gemm(dlogits, hn_rows, out=fp.g("embed"), M=d.vocab_size, N=d.hidden_size, K=R, a_kmajor=True, b_kmajor=True,
                     epilogue=ops.UG_EPI_F32, beta=fp.beta_for("embed"))
        # dhn[R,H] = dlogits[R,V] W[V,H] : W read k-major; its rows >= V come from the zero page and the
        # dlogits pad columns are zero (ug_ce_bwd), so K = V needs no padding
        return ops.gemm(dlogits, fp.w("embed"), M=R, N=d.hidden_size, K=d.vocab_size, b_kmajor=True)


def _head_deferral_methods(cls):
    def _may_defer_head_wgrad(self, R):
        """Only inside a backward pass (the end-of-backward callback is the safety net), on the single-GPU path (with a gradient
        exchange the table's dense part is handed over right after the head instead, unigen_hip/ddp.py), and when the layers'
        grouped launches exist to carry the slices (the grouped form needs a round of tiles)."""
        if os.environ.get("UNIGEN_DEFER_HEAD_WGRAD", "1") != "1" or not torch.is_tensor(self.fp.grad) or not self.fp.grad.is_cuda:
            return False
        sync = getattr(self, "grad_sync", None)
        if sync is not None and sync.active and sync.enabled:
            return False
        d = self.dims
        layer_tiles = sum(-(-a // 256) * -(-b // 256) for a, b in ((2 * d.intermediate_size, d.hidden_size), (d.hidden_size, d.intermediate_size),
                                                                   (d.qkv_out, d.hidden_size), (d.hidden_size, d.hidden_size)))
        return layer_tiles >= ops.WGRAD_GROUP_MIN_TILES and ops.GEMM_POLICY == -1

    def _head_wgrad_slice(self):
        """the next slice of a waiting head weight gradient as a problem of ops.gemm_wgrad_group, or []"""
        st = getattr(self, "_deferred_head", None)
        if not st:
            return []
        dlogits, hn_rows, beta, v0 = st
        V = self.dims.vocab_size
        # Row tiles of the table per layer launch: what fits on the CUs the layer's own tiles leave idle in their last round --
        # idle x (cost of a layer tile / cost of a head tile), costs in k-tiles of 32 plus ~60 for a tile's prologue and fp32
        # epilogue, 80 % of it (measured at the 1.5B shape, 12 336 tokens, 4 096 label rows: 54 idle CUs -> 16 row tiles = 96
        # tiles; 23 -- the whole table over 28 layers -- spills into a fourth round: +45 us per launch).  The rest of the table
        # goes out as one launch at the end of the stack's backward.
        d = self.dims
        layer_tiles = sum(-(-a // 256) * -(-b // 256) for a, b in ((2 * d.intermediate_size, d.hidden_size), (d.hidden_size, d.intermediate_size),
                                                                   (d.qkv_out, d.hidden_size), (d.hidden_size, d.hidden_size)))
        idle = -layer_tiles % 256
        cols = -(-d.hidden_size // 256)
        fit = int(0.8 * idle * (self._wgrad_tokens / 32 + 60) / (hn_rows.shape[0] / 32 + 60) / cols)
        per = int(os.environ.get("UNIGEN_HEAD_SLICE_TILES", "0")) or fit
        if per <= 0:
            return []
        per *= 256
        v1 = min(V, v0 + per)
        prob = (dlogits[:, v0:v1], hn_rows, self.fp.g("embed")[v0:v1], beta)
        if v1 >= V:
            self._deferred_head = None
        else:
            st[3] = v1
        return [prob]

    def flush_deferred_head(self):
        """what is left of a waiting head weight gradient, as one launch (before any other writer of the tied table; end of the
        decoder stack's backward; end of the backward pass)"""
        st = getattr(self, "_deferred_head", None)
        if not st:
            return
        dlogits, hn_rows, beta, v0 = st
        self._deferred_head = None
        V, H = self.dims.vocab_size, self.dims.hidden_size
        ops.gemm(dlogits[:, v0:], hn_rows, out=self.fp.g("embed")[v0:], M=V - v0, N=H, K=hn_rows.shape[0], a_kmajor=True,
                 b_kmajor=True, epilogue=ops.UG_EPI_F32, beta=beta)

    cls._may_defer_head_wgrad, cls._head_wgrad_slice, cls.flush_deferred_head = _may_defer_head_wgrad, _head_wgrad_slice, flush_deferred_head
    return cls


class DecodeState:
    """Static KV cache for autoregressive generation: per layer K,V [rows][HKV][Tmax][128] bf16, the write
    position and visible length as DEVICE ints (so one captured graph serves every step)."""

    def __init__(self, dims, rows, Tmax, device, key_valid=None):
        self.rows, self.Tmax = rows, Tmax
        n, hk, hd = dims.num_hidden_layers, dims.num_key_value_heads, dims.head_dim
        self.k = [torch.zeros((rows, hk, Tmax, hd), dtype=torch.bfloat16, device=device) for _ in range(n)]
        self.v = [torch.zeros((rows, hk, Tmax, hd), dtype=torch.bfloat16, device=device) for _ in range(n)]
        self.pos = torch.zeros(1, dtype=torch.int32, device=device)
        self.len = torch.zeros(1, dtype=torch.int32, device=device)
        self.key_valid = None
        if key_valid is not None:
            kv = torch.ones((rows, Tmax), dtype=torch.uint8, device=device)
            kv[:, :key_valid.shape[1]] = key_valid.to(device=device, dtype=torch.uint8)
            self.key_valid = kv

    def advance(self):
        self.pos.add_(1)
        self.len.add_(1)

    def ensure_accumulators(self, dims, sw=False):
        """Persistent scratch of the five-launch decode layer: one raw fp32 accumulator per split-K projection (row-major
        [rows][N]; each is cleared by a later launch once fully consumed), the second residual-stream buffer, and the row
        sum-of-squares slots that carry the RMSNorm statistics to the accumulators' consumers.  sw: the layer whose o and gate/up
        projections are single-writer launches (csrc/decode_sw.hip) -- no gate/up or o accumulator, a finished bf16 `act` instead,
        and two down-projection accumulators that alternate by layer (the one layer l + 1 still reads is cleared by layer l + 1's
        own down projection)."""
        dev = self.pos.device
        nqkv = (dims.num_attention_heads + 2 * dims.num_key_value_heads) * dims.head_dim
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        if getattr(self, "acc_qkv", None) is None:
            self.acc_qkv, self.acc_down = z(self.rows, nqkv), z(self.rows, dims.hidden_size)
            self.x_mid = z(self.rows, dims.hidden_size)
            self.ss_attn, self.ss_mlp = z(32), z(32)
        if sw and getattr(self, "act", None) is None:
            self.acc_down2, self.zeros = z(self.rows, dims.hidden_size), z(self.rows, dims.hidden_size)
            self.act = torch.empty((self.rows, dims.intermediate_size), dtype=torch.bfloat16, device=dev)
        if not sw and getattr(self, "acc_gu", None) is None:
            self.acc_gu, self.acc_o = z(self.rows, 2 * dims.intermediate_size), z(self.rows, dims.hidden_size)


def _decode_methods(cls):
    def prefill(self, st, embeds, key_valid=None, mask_bits=None):
        """embeds fp32 [rows, P, H] -> final-norm hidden of the LAST position, bf16 [rows, H]; fills the cache.
        mask_bits: compressed [rows, P, P] mask of the prompt (default: causal with `key_valid` columns)."""
        d = self.dims
        R, P, H = embeds.shape
        Hq, Hk, hd = d.num_attention_heads, d.num_key_value_heads, d.head_dim
        self.fp.refresh_compute_copies()
        mb = mask_bits if mask_bits is not None else ops.mask_causal(R, P, self.device, key_valid=key_valid)
        h = embeds.reshape(R * P, H).float().contiguous()
        for i in range(d.num_hidden_layers):
            saved = []
            h = self.layer_fwd(i, h, mb, P, saved)
            ops.kv_store(saved[0].qkv, st.k[i], st.v[i], R, P, Hq, Hk, hd, st.Tmax, None, 0)
        st.pos.fill_(P)
        st.len.fill_(P + 1)
        hn, _ = ops.rmsnorm_fwd(h, self.fp.p("norm"), d.rms_norm_eps, want_rstd=False)
        return hn.view(R, P, H)[:, -1].contiguous()

    def decode_sw(self, st):
        """Whether this decode state runs the layer with single-writer o / gate-up / head launches (csrc/decode_sw.hip)."""
        d = self.dims
        return (getattr(self, "decode_fused", True) and os.environ.get("UNIGEN_DECODE_SW", "1") != "0" and st.rows <= 16
                and ops.decode_sw_supported(d.hidden_size, d.intermediate_size, d.num_attention_heads * d.head_dim, d.head_dim))

    def _decode_layers_sw(self, st, x):
        """The decoder stack of one decode step, five launches per layer (measured forms: profiles/r06_decode_forms.md):
          q/k/v      split-K (operand = the stream + the previous layer's pending down projection; leaves a raw accumulator)
          attention  finishes q/k/v from that accumulator, appends k / v, attends to the cache
          o          single writer: the stream is FINISHED in place
          gate/up    single writer: act = SwiGLU, finished bf16
          down       split-K (k-blocks, LDS pre-reduction) on act into this layer's accumulator; clears the accumulators consumed so far
        -> (stream, pending): the residual stream after the last layer is stream + bf16round(pending)."""
        d, fp = self.dims, self.fp
        Hq, Hk, hd = d.num_attention_heads, d.num_key_value_heads, d.head_dim
        R, H = st.rows, d.hidden_size
        cos, sin = self.rope(st.Tmax)
        st.ensure_accumulators(d, sw=True)
        eps = d.rms_norm_eps
        o = torch.empty((R, Hq * hd), dtype=torch.bfloat16, device=x.device)
        bufs, accd = (x, st.x_mid), (st.acc_down, st.acc_down2)
        n = d.num_hidden_layers
        # down projection: k-blocks of seven slabs with the partial tiles pre-reduced in LDS (5 atomics per output for the 1.5B model
        # instead of 35: 7.6 vs 8.9 us) when the intermediate size is a whole number of them, else one slab per wave
        down = ops.decode_sw_kblock_ if d.intermediate_size % 1792 == 0 else ops.decode_gemv_
        for i in range(n):
            xin, xout = bufs[i & 1], bufs[(i + 1) & 1]
            pend = st.zeros if i == 0 else accd[(i - 1) & 1]
            ops.decode_gemv_resid_norm_(xin, pend, fp.p(f"l{i}.ln1"), xout, st.ss_attn, fp.w(f"l{i}.wqkv"), st.acc_qkv)
            ops.attn_decode_fused(st.acc_qkv, st.ss_attn, eps, H, fp.w(f"l{i}.bqkv"), cos, sin, st.pos, st.k[i], st.v[i],
                                  st.key_valid, o, Hq, Hk, hd, st.Tmax)
            ops.decode_sw_resid_(o, fp.w(f"l{i}.wo"), xout)
            ops.decode_sw_gate_up_(xout, fp.p(f"l{i}.ln2"), eps, fp.w(f"l{i}.wgu"), st.act)
            down(st.act, fp.w(f"l{i}.wdown"), accd[i & 1], zero0=st.acc_qkv, zero1=accd[(i + 1) & 1], ss_zero=st.ss_attn)
        return bufs[n & 1], accd[(n - 1) & 1]

    def decode_step_logits(self, st, x, w_head, logits):
        """decode_step + the head slice in one go (single-writer layer only): logits fp32 [rows, N] = rows `w_head` of the tied
        embedding applied to the final-norm hidden state; advances st.pos / st.len.  The final RMSNorm and the last layer's
        pending residual add ride in the head launch's prologue."""
        stream, pending = self._decode_layers_sw(st, x)
        ops.decode_sw_head_(stream, self.fp.p("norm"), self.dims.rms_norm_eps, w_head, logits, pend=pending, advance=(st.pos, st.len))
        return logits

    def decode_step(self, st, x):
        """x fp32 [rows, H] = embedding of the newest token (updated in place as the residual stream);
        appends its K/V at st.pos, ADVANCES st.pos / st.len by one and returns the final-norm hidden bf16 [rows, H].  No host sync, no
        shape depends on the step: capturable.  Five launches per layer (see include/unigen_hip.h): a split-K
        projection leaves its raw fp32 accumulator behind and the NEXT kernel applies bias / RoPE / residual add /
        RMSNorm / SiLU-mul while it builds its own operand, so kernel boundaries are the only synchronisation."""
        d, fp = self.dims, self.fp
        Hq, Hk, hd = d.num_attention_heads, d.num_key_value_heads, d.head_dim
        R, H, I = st.rows, d.hidden_size, d.intermediate_size
        if not getattr(self, "decode_fused", True) or R > 32 or hd != 128 or min(H, I, Hq * hd) < 256 or H % 32 or I % 32:
            return self._decode_step_wide(st, x)
        eps = d.rms_norm_eps
        hn = torch.empty((R, H), dtype=torch.bfloat16, device=x.device)
        if self.decode_sw(st):
            stream, pending = self._decode_layers_sw(st, x)
            # pending down_proj of the last layer + final RMSNorm (also clears that accumulator)
            ops.decode_finish_resid_norm_(pending, stream, fp.p("norm"), hn, eps, advance=(st.pos, st.len))
            return hn
        cos, sin = self.rope(st.Tmax)
        st.ensure_accumulators(d)
        o = torch.empty((R, Hq * hd), dtype=torch.bfloat16, device=x.device)
        for i in range(d.num_hidden_layers):
            # x (+ pending down_proj of the previous layer) -> x_mid ; q/k/v accumulator ; clears gate_up acc
            ops.decode_gemv_resid_norm_(x, st.acc_down, fp.p(f"l{i}.ln1"), st.x_mid, st.ss_attn, fp.w(f"l{i}.wqkv"), st.acc_qkv,
                                        zero0=st.acc_gu, ss_zero=st.ss_mlp)
            ops.attn_decode_fused(st.acc_qkv, st.ss_attn, eps, H, fp.w(f"l{i}.bqkv"), cos, sin, st.pos, st.k[i], st.v[i],
                                  st.key_valid, o, Hq, Hk, hd, st.Tmax)
            ops.decode_gemv_(o, fp.w(f"l{i}.wo"), st.acc_o, zero0=st.acc_qkv, zero1=st.acc_down, ss_zero=st.ss_attn)
            # x_mid + pending o_proj -> x ; gate/up accumulator
            ops.decode_gemv_resid_norm_(st.x_mid, st.acc_o, fp.p(f"l{i}.ln2"), x, st.ss_mlp, fp.w(f"l{i}.wgu"), st.acc_gu)
            ops.decode_gemv_swiglu_(st.acc_gu, st.ss_mlp, eps, H, fp.w(f"l{i}.wdown"), st.acc_down, zero0=st.acc_o)
        # pending down_proj of the last layer + final RMSNorm (also clears acc_down for the next step)
        ops.decode_finish_resid_norm_(st.acc_down, x, fp.p("norm"), hn, eps, advance=(st.pos, st.len))
        return hn

    def _decode_step_wide(self, st, x):
        """> 32 rows: the GEMV kernel does not apply; split-K GEMMs + separate finishing kernels."""
        d, fp = self.dims, self.fp
        Hq, Hk, hd = d.num_attention_heads, d.num_key_value_heads, d.head_dim
        cos, sin = self.rope(st.Tmax)
        for i in range(d.num_hidden_layers):
            xn, _ = ops.rmsnorm_fwd(x, fp.p(f"l{i}.ln1"), d.rms_norm_eps, want_rstd=False)
            qkv = ops.skinny_linear(xn, fp.w(f"l{i}.wqkv"), bias=fp.w(f"l{i}.bqkv"))
            ops.rope_at_(qkv, cos, sin, Hq + Hk, hd, st.pos)
            ops.kv_store(qkv, st.k[i], st.v[i], st.rows, 1, Hq, Hk, hd, st.Tmax, st.pos, 0)
            o = ops.attn_decode(qkv, st.k[i], st.v[i], st.key_valid, Hq, Hk, hd, st.Tmax, st.len)
            ops.skinny_linear(o, fp.w(f"l{i}.wo"), resid=x)
            xn2, _ = ops.rmsnorm_fwd(x, fp.p(f"l{i}.ln2"), d.rms_norm_eps, want_rstd=False)
            gu = ops.skinny_linear(xn2, fp.w(f"l{i}.wgu"))
            act = ops.swiglu_fwd(gu)
            ops.skinny_linear(act, fp.w(f"l{i}.wdown"), resid=x)
        hn, _ = ops.rmsnorm_fwd(x, fp.p("norm"), d.rms_norm_eps, want_rstd=False)
        st.advance()
        return hn

    def head_slice(self, hn, v0, v1):
        """logits bf16 [rows, v1-v0] for vocabulary rows [v0, v1) of the tied embedding."""
        n = v1 - v0
        out = torch.empty((hn.shape[0], ops.round_up(n, 8)), dtype=torch.bfloat16, device=self.device)
        ops.gemm(hn, self.fp.w("embed")[v0:v1], out=out, N=n, K=self.dims.hidden_size)
        return out[:, :n]

    cls.prefill, cls.decode_step, cls.head_slice = prefill, decode_step, head_slice
    cls.decode_sw, cls._decode_layers_sw, cls.decode_step_logits = decode_sw, _decode_layers_sw, decode_step_logits
    cls._decode_step_wide = _decode_step_wide
    return cls


_decode_methods(Qwen2Engine)
_head_deferral_methods(Qwen2Engine)
