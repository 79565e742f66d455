"""Fused label log-probabilities for DPO (reference training/train_dpo.py:51-90 `get_batch_logps`, used at :573-647).

The reference slices the dense [2B, L, V] logits at the image positions, up-casts to fp32, takes log_softmax over
V = 159 867 and gathers the label's entry (4.9 GB of fp32 for 10 pairs).  Here the tied lm_head is evaluated for the
image rows only and one cross-entropy kernel pass returns each row's label log-probability; the backward rewrites the
bf16 logits in place as (onehot - softmax) * upstream and runs the head's dgrad / wgrad GEMMs.  Same values as the
reference function applied to `model(...)`'s lazy logits (tests/test_dpo_gpu.py), a maintainer can swap it in:

    from unigen_hip.dpo import get_batch_logps        # instead of the function defined in train_dpo.py
"""
import torch

from . import ops
from .modules import LazyLogits


class _LogpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hn, engine, idx, row_labels, weights, B):
        Bh, L, H = hn.shape
        V = engine.dims.vocab_size
        rows = ops.gather_rows(hn.reshape(Bh * L, H), idx)
        logits = engine.logits_rows(rows)
        _, lse, _, logp = ops.ce_fwd(logits, V, row_labels, want_logp=True)
        ctx.engine, ctx.rows, ctx.logits, ctx.idx, ctx.labels, ctx.lse, ctx.weights = engine, rows, logits, idx, row_labels, lse, weights
        ctx.shape, ctx.B = (Bh, L, H), B
        if any(ctx.needs_input_grad) and hasattr(engine, "_head_graphs_live"):
            engine._head_graphs_live += 1            # a dense writer of the tied table waiting for its backward (modules.py)
        return (logp * weights).view(B, -1).sum(-1)

    @staticmethod
    def backward(ctx, dlogps):
        eng = ctx.engine
        Bh, L, H = ctx.shape
        eng.begin_grad_pass()
        per_row = ctx.weights.view(ctx.B, -1) * dlogps.float().view(ctx.B, 1)
        # d logp / d logits = onehot - softmax = -(softmax - onehot)
        ops.ce_bwd_(ctx.logits, eng.dims.vocab_size, ctx.labels, ctx.lse, None, row_scale=(-per_row).reshape(-1).contiguous())
        drows = eng.head_bwd(ctx.logits, ctx.rows)
        if hasattr(eng, "head_written"):
            eng.head_written()
        ctx.logits = None
        dhn = torch.zeros((Bh * L, H), dtype=torch.bfloat16, device=drows.device)
        ops.scatter_rows_(drows, ctx.idx, dhn)
        return dhn.view(Bh, L, H), None, None, None, None, None


def get_batch_logps(logits, labels, average_log_prob=False, label_pad_token_id=-100, num_vq_tokens=256, t2i_gen_mode="mask"):
    """Drop-in for training/train_dpo.py:get_batch_logps on the lazy logits `UniGen.forward(labels=None)` returns."""
    if not isinstance(logits, LazyLogits):
        raise ops._l.UniGenHipError("unigen_hip.dpo.get_batch_logps expects the lazy logits returned by UniGen.forward")
    if tuple(logits.shape[:-1]) != tuple(labels.shape):
        raise ValueError("Logits (batch and sequence length dim) and labels must have the same shape.")
    eng, hn = logits.engine, logits.hn
    B, L, _ = hn.shape
    n = num_vq_tokens
    dev = hn.device
    lab = labels.to(dev)[:, -(n + 1):-1]
    pos = torch.arange(L - n - 1, L - 1, device=dev)
    if t2i_gen_mode == "ar":                        # logits at position p score the label at p + 1
        lab, pos = lab[:, 1:], pos[:-1]
    keep = lab != label_pad_token_id
    idx = (torch.arange(B, device=dev)[:, None] * L + pos[None, :]).reshape(-1).contiguous()
    row_labels = torch.where(keep, lab, torch.full_like(lab, -100)).reshape(-1).contiguous()
    w = keep.float()
    if average_log_prob:
        w = w / keep.sum(-1, keepdim=True).clamp(min=1)
    return _LogpFn.apply(hn, eng, idx, row_labels, w.reshape(-1).contiguous(), B)
