"""nn.Module façade over Qwen2Engine with the attribute paths the reference's callers reach into
(SURVEY.md §8b): `llm.model.embed_tokens(ids)`, `llm.lm_head`, `llm.config.{use_cache,vocab_size}`,
`llm.resize_token_embeddings`, `llm.gradient_checkpointing_enable()`, and parameter names equal to
the reference checkpoint keys (`llm.model.layers.N.self_attn.q_proj.weight`, ...).

Autograd: three coarse `torch.autograd.Function`s (embedding, decoder stack, head+loss).  Weight
gradients are written by the kernels straight into the flat fp32 gradient buffer; every
`nn.Parameter` is a view of the flat master buffer and its `.grad` a view of the flat grad buffer,
so torch optimizers, `named_parameters()` and `zero_grad(set_to_none=True)` behave as usual.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .lib import UniGenHipError
from .qwen2 import Qwen2Dims, Qwen2Engine, _hf_name_map


def anchor_of(engine):
    return engine._anchor


# ------------------------------------------------------------------------------------ autograd
class _EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, ids, engine):
        flat_ids = ids.reshape(-1).contiguous()
        engine.fp.wait_pending_update()
        out = ops.embed_fwd(flat_ids, engine.fp.p("embed"), engine.err_flag)
        ctx.engine, ctx.ids = engine, flat_ids
        if ctx.needs_input_grad[0]:
            engine._lookup_rows_live += flat_ids.numel()     # host-side count: the data-parallel exchange sizes its all-gather from it
        return out.view(*ids.shape, engine.dims.hidden_size)

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.engine
        eng.begin_grad_pass()
        drows = dout.reshape(-1, eng.dims.hidden_size).float().contiguous()
        eng.flush_deferred_head()                # (a head weight gradient still waiting for a layer launch overwrites: it goes first)
        sync = eng.grad_sync
        if sync is not None and sync.wants_lookups() and eng._lookup_rows_kept + ctx.ids.numel() <= eng._lookup_rows_cap:
            # data parallel: the (id, row) pairs are kept aside and exchanged as such at the end of backward; the tied
            # table's dense part (the head's weight gradient) is already travelling (unigen_hip/ddp.py)
            eng._lookup_rows_kept += ctx.ids.numel()
            sync.add_lookup(ctx.ids, drows)
        else:
            if sync is not None:
                sync.before_dense_embed_write()
            eng.fp.ensure_zeroed("embed")
            ops.embed_bwd(ctx.ids, drows, eng.fp.g("embed"))
        if eng.grad_ready_hook:
            eng.grad_ready_hook("embed")
        return None, None, None


class _StackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, h0, engine, mb):
        B, L, H = h0.shape
        need = any(ctx.needs_input_grad)
        saved = [] if need else None
        if need:
            engine._forward_entry()         # world > 1: install the gradient exchange / align the replicas before step 1
            engine._stack_graphs_live += 1  # decoder-stack segments waiting for their backward (see _StackFn.backward)
        h_last, hn, rstd = engine.stack_fwd(h0.reshape(B * L, H).float().contiguous(), mb, L, saved)
        if need:
            ctx.engine, ctx.saved, ctx.h_last, ctx.rstd, ctx.mb, ctx.shape = engine, saved, h_last, rstd, mb, (B, L, H)
        return hn.view(B, L, H)

    @staticmethod
    def backward(ctx, dhn):
        eng = ctx.engine
        B, L, H = ctx.shape
        eng.begin_grad_pass()
        # layer gradients may be handed to the data-parallel exchange as this segment retires them only if no other
        # recorded stack segment (a second forward under the same backward) can still add to them
        last_writer = eng._stack_graphs_live <= 1
        eng._stack_graphs_live = max(0, eng._stack_graphs_live - 1)
        if eng.grad_sync is not None:
            eng.grad_sync.set_overlap(last_writer)
        dhn = dhn.reshape(B * L, H).to(torch.bfloat16).contiguous()
        dh0 = eng.stack_bwd(ctx.saved, ctx.h_last, ctx.rstd, dhn, ctx.mb, L)
        ctx.saved = None
        # the anchor gets a (zero) gradient once per backward: when it is the `_ddp_anchor` Parameter of a model wrapped
        # in DistributedDataParallel this is what tells DDP's reducer that the iteration is complete
        danchor = torch.zeros_like(anchor_of(eng)) if ctx.needs_input_grad[0] else None
        return danchor, dh0.view(B, L, H), None, None


class _HeadLossFn(torch.autograd.Function):
    """losses[s] = mean CE over segment s of the selected rows (tied lm_head + F.cross_entropy,
    reference models/unigen.py:287,310-338); logits exist only for the selected rows."""

    @staticmethod
    def forward(ctx, anchor, hn, engine, idx, labels, bounds):
        B, L, H = hn.shape
        V = engine.dims.vocab_size
        rows = ops.gather_rows(hn.reshape(B * L, H), idx)
        logits = engine.logits_rows(rows)
        outs, stats = [], []
        for (r0, r1) in bounds:
            lc, lse, _, _ = ops.ce_fwd(logits[r0:r1], V, labels[r0:r1])
            outs.append(lc[:1])
            stats.append((lse, lc))
        ctx.engine, ctx.rows, ctx.logits, ctx.idx, ctx.labels, ctx.bounds, ctx.stats = engine, rows, logits, idx, labels, bounds, stats
        ctx.shape = (B, L, H)
        if any(ctx.needs_input_grad):
            engine._head_graphs_live += 1        # dense writers of the tied table waiting for their backward
        return torch.cat(outs)

    @staticmethod
    def backward(ctx, dloss):
        eng = ctx.engine
        B, L, H = ctx.shape
        V = eng.dims.vocab_size
        eng.begin_grad_pass()
        dloss = dloss.float().contiguous()
        for s, (r0, r1) in enumerate(ctx.bounds):
            lse, lc = ctx.stats[s]
            ops.ce_bwd_(ctx.logits[r0:r1], V, ctx.labels[r0:r1], lse, lc, dloss[s:s + 1])
        drows = eng.head_bwd(ctx.logits, ctx.rows)
        eng.head_written()
        ctx.logits = None
        dhn = torch.zeros((B * L, H), dtype=torch.bfloat16, device=drows.device)
        ops.scatter_rows_(drows, ctx.idx, dhn)
        return None, dhn.view(B, L, H), None, None, None, None


class _HeadRowsFn(torch.autograd.Function):
    """logits[rows, v0:v1] of the tied lm_head for the rows `idx` of hn, differentiable: the gradient goes to hn
    (dgrad GEMM) and into the flat embedding-gradient rows [v0, v1) (wgrad GEMM)."""

    @staticmethod
    def forward(ctx, anchor, hn, engine, idx, v0, v1):
        B, L, H = hn.shape
        rows = ops.gather_rows(hn.reshape(B * L, H), idx)
        n = v1 - v0
        npad = ops.round_up(max(n, 1), 8)
        out = torch.empty((rows.shape[0], npad), dtype=torch.bfloat16, device=hn.device)
        ops.gemm(rows, engine.fp.w("embed")[v0:v1], out=out, N=n, K=H)
        ctx.engine, ctx.rows, ctx.idx, ctx.rng, ctx.shape = engine, rows, idx, (v0, v1), (B, L, H)
        if any(ctx.needs_input_grad):
            engine._head_graphs_live += 1
        return out[:, :n]

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.engine
        B, L, H = ctx.shape
        v0, v1 = ctx.rng
        n, R = v1 - v0, ctx.rows.shape[0]
        eng.begin_grad_pass()
        npad = ops.round_up(n, 8)
        dl = torch.zeros((R, npad), dtype=torch.bfloat16, device=dout.device)
        dl[:, :n] = dout
        eng.flush_deferred_head()
        if eng.grad_sync is not None:
            eng.grad_sync.before_dense_embed_write()
        eng.fp.ensure_zeroed("embed")
        ops.gemm(dl, ctx.rows, out=eng.fp.g("embed")[v0:v1], M=n, N=H, K=R, a_kmajor=True, b_kmajor=True,
                 epilogue=ops.UG_EPI_F32, beta=1)
        eng.head_written()
        drows = ops.gemm(dl, eng.fp.w("embed")[v0:v1], M=R, N=H, K=n, b_kmajor=True)
        dhn = torch.zeros((B * L, H), dtype=torch.bfloat16, device=dout.device)
        ops.scatter_rows_(drows, ctx.idx, dhn)
        return None, dhn.view(B, L, H), None, None, None, None


# ------------------------------------------------------------------------------------ lazy logits
class LazyLogits:
    """Stand-in for the dense [B, L, V] logits tensor the reference returns (models/unigen.py:287-290).
    Callers only ever slice it (train.py:926-936, unigen.py:416,507, train_dpo.py:602-609); rows are
    produced on demand by the lm_head GEMM for exactly the positions / vocabulary range requested."""

    def __init__(self, engine, hn, dtype=torch.bfloat16):
        self.engine, self.hn = engine, hn      # hn bf16 [B, L, H]; attached to the autograd graph when it requires grad
        B, L, _ = hn.shape
        self.shape = torch.Size((B, L, engine.dims.vocab_size))
        self.dtype, self.device = dtype, hn.device

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 3

    def _rows(self, bsel, psel, vsel):
        eng = self.engine
        B, L, V = self.shape
        bs = list(range(B))[bsel] if isinstance(bsel, slice) else [int(bsel) % B]
        ps = list(range(L))[psel] if isinstance(psel, slice) else [int(psel) % L]
        v0, v1, vstep = vsel.indices(V) if isinstance(vsel, slice) else (int(vsel) % V, int(vsel) % V + 1, 1)
        if vstep != 1:
            raise UniGenHipError("strided vocabulary slices are not supported")
        idx = (torch.tensor(bs, device=self.device)[:, None] * L + torch.tensor(ps, device=self.device)[None, :]).reshape(-1)
        n = v1 - v0
        out = _HeadRowsFn.apply(eng._anchor, self.hn, eng, idx, v0, v1)          # [R, n] bf16, differentiable
        if self.dtype != torch.bfloat16:
            out = out.to(self.dtype)
        out = out.reshape(len(bs), len(ps), n)
        if not isinstance(vsel, slice):
            out = out[..., 0]
        if not isinstance(psel, slice):
            out = out[:, 0]
        if not isinstance(bsel, slice):
            out = out[0]
        return out

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            raise UniGenHipError("LazyLogits does not support Ellipsis indexing; call .materialize()")
        key = tuple(key) + (slice(None),) * (3 - len(key))
        if not all(isinstance(k, (slice, int)) for k in key):
            return self.materialize()[key]
        return self._rows(*key)

    def materialize(self):
        return self._rows(slice(None), slice(None), slice(None))

    def float(self):
        return self.to(torch.float32)

    def to(self, *a, **k):
        """dtype conversion stays lazy (training/train_dpo.py:602 up-casts the whole tensor before slicing it);
        device moves are no-ops."""
        dt = k.get("dtype")
        for x in a:
            if isinstance(x, torch.dtype):
                dt = x
        return LazyLogits(self.engine, self.hn, dt or self.dtype)

    def chunk(self, n, dim=0):
        B = self.shape[0]
        step = (B + n - 1) // n
        return tuple(self[i:i + step] for i in range(0, B, step))


# ------------------------------------------------------------------------------------ modules
class _ParamHolder(nn.Module):
    """Container giving parameters the reference's dotted names; forward is never called."""


class HipEmbedding(nn.Module):
    def __init__(self, engine, weight):
        super().__init__()
        self.__dict__["_engine"] = engine
        self.weight = weight
        self.num_embeddings, self.embedding_dim = weight.shape

    def forward(self, ids):
        eng = self.__dict__["_engine"]
        if torch.is_grad_enabled():
            eng._forward_entry()            # world > 1: replicas aligned to rank 0 before the first training lookup
        return _EmbedFn.apply(eng._anchor, ids.to(eng.device), eng)


class HipLMHead(nn.Module):
    def __init__(self, engine, weight):
        super().__init__()
        self.__dict__["_engine"] = engine
        self.weight = weight

    def forward(self, hidden):
        eng = self.__dict__["_engine"]
        h = hidden if hidden.dim() == 3 else hidden[None]
        return LazyLogits(eng, h.to(torch.bfloat16).contiguous())


class HipQwen2Model(nn.Module):
    """`llm.model`: embed_tokens / layers / norm with reference parameter names."""

    def __init__(self, engine):
        super().__init__()
        self.__dict__["_engine"] = engine
        d = engine.dims
        params = engine.named_param_views()
        self.embed_tokens = HipEmbedding(engine, params["model.embed_tokens.weight"])
        self.layers = nn.ModuleList()
        for i in range(d.num_hidden_layers):
            layer = _ParamHolder()
            layer.self_attn = _ParamHolder()
            for proj in ("q_proj", "k_proj", "v_proj", "o_proj"):
                holder = _ParamHolder()
                holder.weight = params[f"model.layers.{i}.self_attn.{proj}.weight"]
                if proj != "o_proj":
                    holder.bias = params[f"model.layers.{i}.self_attn.{proj}.bias"]
                setattr(layer.self_attn, proj, holder)
            layer.mlp = _ParamHolder()
            for proj in ("gate_proj", "up_proj", "down_proj"):
                holder = _ParamHolder()
                holder.weight = params[f"model.layers.{i}.mlp.{proj}.weight"]
                setattr(layer.mlp, proj, holder)
            for nm in ("input_layernorm", "post_attention_layernorm"):
                holder = _ParamHolder()
                holder.weight = params[f"model.layers.{i}.{nm}.weight"]
                setattr(layer, nm, holder)
            self.layers.append(layer)
        self.norm = _ParamHolder()
        self.norm.weight = params["model.norm.weight"]

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, output_hidden_states=False,
                return_dict=True, **kwargs):
        eng = self.__dict__["_engine"]
        if (input_ids is None) == (inputs_embeds is None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if inputs_embeds is None:
            inputs_embeds = self.embed_tokens(input_ids)
        B, L, _ = inputs_embeds.shape
        mb = eng.mask_bits(attention_mask, B, L)
        hn = _StackFn.apply(eng._anchor, inputs_embeds, eng, mb)
        return SimpleNamespace(last_hidden_state=hn, past_key_values=None)


class HipQwen2ForCausalLM(nn.Module):
    """`UniGen.llm` (reference: transformers Qwen2ForCausalLM built at models/unigen.py:56-69)."""

    def __init__(self, dims, device, seed=None):
        super().__init__()
        engine = TrainEngine(dims, device)
        self.__dict__["_engine"] = engine
        self.config = SimpleNamespace(vocab_size=dims.vocab_size, hidden_size=dims.hidden_size, use_cache=False,
                                      num_hidden_layers=dims.num_hidden_layers, intermediate_size=dims.intermediate_size,
                                      num_attention_heads=dims.num_attention_heads,
                                      num_key_value_heads=dims.num_key_value_heads, rope_theta=dims.rope_theta,
                                      rms_norm_eps=dims.rms_norm_eps)
        self.model = HipQwen2Model(engine)
        self.lm_head = HipLMHead(engine, self.model.embed_tokens.weight)     # tied
        self.vocab_size = dims.vocab_size
        if seed is not None:
            self.init_weights(seed)

    @property
    def engine(self):
        return self.__dict__["_engine"]

    def init_weights(self, seed):
        """HF Qwen2 init (N(0, initializer_range) matrices, zero biases, unit norms): what
        `Qwen2ForCausalLM(config)` gives the reference at models/unigen.py:65.  Drawn on the host
        parameter by parameter in named_parameters() order so the CPU oracle can reproduce it."""
        g = torch.Generator().manual_seed(seed)
        std = self.engine.dims.initializer_range
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("norm.weight") or name.endswith("layernorm.weight"):
                    p.fill_(1.0)
                elif name.endswith(".bias"):
                    p.zero_()
                else:
                    p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))

    def init_weights_device(self, seed):
        """Same distribution, drawn on the GPU (used for the 1.5 B benchmark model: host draws would take
        minutes).  Deterministic per (seed, device type); identical on every rank."""
        eng = self.engine
        g = torch.Generator(device=eng.device).manual_seed(seed)
        std = eng.dims.initializer_range
        with torch.no_grad():
            eng.fp.master.zero_()
            for key, shp in eng.fp.spec:
                v = eng.fp.p(key)
                if key.endswith(("ln1", "ln2")) or key == "norm":
                    v.fill_(1.0)
                elif key.endswith("bqkv"):
                    v.zero_()
                else:
                    v.normal_(0.0, std, generator=g)

    def gradient_checkpointing_enable(self, *a, **k):
        # 288 GB of HBM holds every activation of the shipped configs; nothing to recompute.
        return None

    def gradient_checkpointing_disable(self):
        return None

    def resize_token_embeddings(self, vocab_size):
        if vocab_size != self.engine.dims.vocab_size:
            raise UniGenHipError("resize_token_embeddings after construction is not implemented: build the model "
                                 "with the final vocab_size (the reference does, models/unigen.py:59-60)")
        return self.model.embed_tokens

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, **kwargs):
        out = self.model(input_ids=input_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds)
        return SimpleNamespace(logits=self.lm_head(out.last_hidden_state), past_key_values=None)


class TrainEngine(Qwen2Engine):
    """Qwen2Engine + the glue the module layer needs (parameter views, grad bookkeeping, masks, data-parallel hook-up)."""

    def __init__(self, dims, device):
        super().__init__(dims, device)
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self._params = None
        self._mask_cache = (None, None)
        # data parallelism (unigen_hip/ddp.py): installed lazily when torch.distributed runs with world > 1
        self.grad_sync = None
        self.auto_data_parallel = True
        self.require_grad_sync = True        # False inside UniGen.no_sync(): gradient-accumulation micro-steps
        self.extra_grad_params = None        # callable -> ordinary Parameters to average when no DDP wrapper does it
        self._in_backward = False
        self._stack_graphs_live = 0
        self._head_graphs_live = 0           # recorded head segments (dense writers of the tied table) waiting for their backward
        self._lookup_rows_live = 0           # embedding-lookup rows recorded on live autograd graphs (host-side count)
        self._lookup_rows_cap = 0            # ... as announced to the exchange at the start of the running backward pass
        self._lookup_rows_kept = 0

    def head_written(self):
        """A head segment has added its weight gradient to the tied table.  When it was the last recorded one, the table's
        dense part is final -- only embedding lookups (kept aside by the exchange) follow -- and is handed over at once."""
        self._head_graphs_live = max(0, self._head_graphs_live - 1)
        if self._head_graphs_live == 0 and self.grad_ready_hook:
            self.grad_ready_hook("head")

    def named_param_views(self):
        if self._params is None:
            self._params = {}
            for name, (key, rows) in _hf_name_map(self.dims).items():
                v, g = self.fp.p(key), self.fp.g(key)
                if rows is not None:
                    v, g = v[rows[0]:rows[1]], g[rows[0]:rows[1]]
                p = nn.Parameter(v, requires_grad=True)
                p.grad = g
                self.__dict__.setdefault("_grad_views", {})[name] = (p, g)
                self._params[name] = p
        return self._params

    # ------------------------------------------------------------------ data parallel
    def _dp_sync(self):
        """The FlatGradSync of this engine; created on first use once torch.distributed is initialised with world > 1.
        Creation also broadcasts rank 0's master weights: DistributedDataParallel does that for the parameters it
        manages, and the flat views are deliberately hidden from it (`UniGen._ddp_params_and_buffers_to_ignore`)."""
        if self.grad_sync is None and self.auto_data_parallel:
            import torch.distributed as dist
            import os
            forced = os.environ.get("UNIGEN_DDP_FORCE", "0") == "1"
            if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced):
                from .ddp import FlatGradSync
                self.grad_sync = FlatGradSync(self, extra_params=lambda: self.extra_grad_params() if self.extra_grad_params else [])
                with torch.no_grad():
                    dist.broadcast(self.fp.master, 0)
                    for p in (self.extra_grad_params() if self.extra_grad_params else []):
                        dist.broadcast(p.data, 0)
                self.fp._seen_version = -1
        return self.grad_sync

    def _forward_entry(self):
        """Start of a training forward.  A backward that raised after arming the exchange (OOM, a device-side check,
        KeyboardInterrupt in a retry loop) never ran its end-of-backward callback: a forward is proof that no backward
        pass is in flight any more, so the next one arms again."""
        self._in_backward = False
        return self._dp_sync()

    def _sync_this_pass(self):
        """False on a gradient-accumulation micro-step: inside `UniGen.no_sync()`, or when accelerate's shared
        GradientState (set by `accelerator.accumulate` / `no_sync`, training/train.py's own mechanism) says so."""
        if not self.require_grad_sync:
            return False
        import sys
        st = sys.modules.get("accelerate.state")
        gs = getattr(st, "GradientState", None) if st is not None else None
        if gs is not None and getattr(gs, "_shared_state", None):
            return bool(gs().sync_gradients)
        return True

    def _end_of_backward(self):
        self._in_backward = False
        self._stack_graphs_live = 0          # graphs that were built and dropped without a backward do not count against the next pass
        self._head_graphs_live = 0
        self._lookup_rows_live = 0
        if self.grad_sync is not None:
            self.grad_sync.finish()

    def begin_grad_pass(self):
        """Called at the start of every backward segment.  If the caller dropped the gradients
        (`optimizer.zero_grad(set_to_none=True)`, reference training/train.py:793) the flat buffer is
        cleared once and every Parameter gets its persistent grad view back.  The first segment of a backward pass also
        arms the data-parallel exchange and queues its completion on the autograd engine's end-of-backward callbacks."""
        self.fp.wait_pending_update()
        views = self.__dict__.get("_grad_views", {})
        if any(p.grad is None for p, _ in views.values() if p.requires_grad):
            if all(p.grad is None for p, _ in views.values() if p.requires_grad):
                self.fp.clear_grads()
            else:                                # only some were dropped: those restart from zero
                for p, g in views.values():
                    if p.grad is None and p.requires_grad:
                        g.zero_()
            for p, g in views.values():
                if p.grad is None and p.requires_grad:
                    p.grad = g
        if not self._in_backward:
            sync = self._dp_sync()
            if sync is not None:
                self._in_backward = True
                self._lookup_rows_cap, self._lookup_rows_kept = self._lookup_rows_live, 0
                sync.begin(enabled=self._sync_this_pass(), lookup_rows=self._lookup_rows_live, heads_live=self._head_graphs_live)
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def mask_bits(self, attention_mask, B, L):
        if attention_mask is None:
            key = ("causal", B, L)
            if self._mask_cache[0] != key:
                self._mask_cache = (key, ops.mask_causal(B, L, self.device))
            return self._mask_cache[1]
        if isinstance(attention_mask, ops.MaskBits):
            return attention_mask
        if attention_mask.dim() == 2:         # [B, L] key-validity mask: causal + padding (HF semantics)
            return ops.mask_causal(B, L, self.device, key_valid=attention_mask.to(self.device) != 0)
        # cache per mask OBJECT (MaskGIT calls forward T times with the same mask); holding the
        # reference keeps the allocator from recycling its address under a different mask
        cached = self._mask_cache[0]
        if not (isinstance(cached, tuple) and cached[0] is attention_mask and cached[1] == attention_mask._version):
            mb = ops.mask_compress(attention_mask.to(self.device), self.err_flag)
            self._mask_cache = ((attention_mask, attention_mask._version), mb)
        return self._mask_cache[1]


# ------------------------------------------------------------------------------------ generic table lookup / CE (gen_projector path)
class _TableEmbedFn(torch.autograd.Function):
    """rows of an ordinary fp32 embedding table (UniGen.gen_embed, reference models/unigen.py:76,82) on the embedding
    kernels; the gradient is scatter-added into a dense table gradient like torch's nn.Embedding."""

    @staticmethod
    def forward(ctx, weight, ids, err_flag):
        flat = ids.reshape(-1).contiguous()
        out = ops.embed_fwd(flat, weight.detach().float().contiguous(), err_flag)
        ctx.ids, ctx.shape = flat, tuple(weight.shape)
        return out.view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, dout):
        dW = torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        ops.embed_bwd(ctx.ids, dout.reshape(-1, ctx.shape[1]).float().contiguous(), dW)
        return dW, None, None


class _CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits [R, V] bf16, labels [R], ignore_index=-100), mean over the kept rows, on the CE kernels
    (img_head logits of the gen_projector path, reference models/unigen.py:301-311)."""

    @staticmethod
    def forward(ctx, logits, labels):
        R, V = logits.shape
        lg = torch.empty((R, ops.round_up(V, 8)), dtype=torch.bfloat16, device=logits.device)      # 16-byte rows for the kernel
        lg[:, :V] = logits
        lc, lse, _, _ = ops.ce_fwd(lg, V, labels)
        ctx.save_for_backward(lg, labels, lse, lc)
        ctx.in_dtype, ctx.V = logits.dtype, V
        return lc[0].clone()

    @staticmethod
    def backward(ctx, dloss):
        lg, labels, lse, lc = ctx.saved_tensors
        g = lg.clone()
        ops.ce_bwd_(g, ctx.V, labels, lse, lc, dloss.float().reshape(1).contiguous())
        return g[:, :ctx.V].to(ctx.in_dtype), None


# ------------------------------------------------------------------------------------ mm_projector
class _LinearFn(torch.autograd.Function):
    """y = x W^T + b with bf16 operands / fp32 accumulate (what nn.Linear does under the reference's bf16
    autocast), dgrad and wgrad on the same GEMM kernel."""

    @staticmethod
    def forward(ctx, x, w, b):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).to(torch.bfloat16).contiguous()
        wb = ops.cast_bf16(w.detach().float().contiguous())
        bb = ops.cast_bf16(b.detach().float().contiguous()) if b is not None else None
        y = ops.gemm(x2, wb, bias=bb)
        ctx.save_for_backward(x2, wb)
        ctx.shp, ctx.has_bias, ctx.in_dtype = shp, b is not None, x.dtype
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, wb = ctx.saved_tensors
        N, K = wb.shape
        dy2 = dy.reshape(-1, N).to(torch.bfloat16)
        if N % 8:                 # 16-byte rows for the GEMM's loads: zero-padded columns (they meet zero-page weight rows)
            pad = torch.zeros((dy2.shape[0], ops.round_up(N, 8)), dtype=torch.bfloat16, device=dy2.device)
            pad[:, :N] = dy2
            dy2 = pad
        else:
            dy2 = dy2.contiguous()
        M = dy2.shape[0]
        dx = (ops.gemm(dy2, wb, M=M, N=K, K=N, b_kmajor=True).view(ctx.shp).to(ctx.in_dtype)
              if ctx.needs_input_grad[0] else None)
        dw = ops.gemm(dy2, x2, M=N, N=K, K=M, a_kmajor=True, b_kmajor=True, epilogue=ops.UG_EPI_F32)
        db = None
        if ctx.has_bias:
            db = torch.zeros(N, dtype=torch.float32, device=dy2.device)
            ops.colsum_(dy2[:, :N], db)
        return dx, dw, db


class _GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xb = x.to(torch.bfloat16).contiguous()
        ctx.save_for_backward(xb)
        return ops.gelu(xb)

    @staticmethod
    def backward(ctx, dy):
        (xb,) = ctx.saved_tensors
        return ops.gelu(xb, dy.to(torch.bfloat16).contiguous())


class HipProjector(nn.Sequential):
    """`UniGen.mm_projector` (reference models/unigen.py:119-128): Linear -> (GELU -> Linear)*; an
    nn.Sequential of ordinary torch modules (state-dict keys `mm_projector.0.weight`, ...) whose forward
    runs on the HIP GEMM / GELU kernels."""

    def forward(self, x):
        for m in self:
            if isinstance(m, nn.Linear):
                x = _LinearFn.apply(x, m.weight, m.bias)
            elif isinstance(m, nn.GELU):
                x = _GeluFn.apply(x)
            else:
                raise UniGenHipError(f"unexpected module in mm_projector: {type(m).__name__}")
        return x
