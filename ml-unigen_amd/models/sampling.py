"""MaskGIT sampling helpers with the reference's public names and results
(reference: models/sampling.py -- 13 public functions incl. the misspelt `get_mask_chedule`).
These are tiny per-step host-orchestrated tensor ops on [B, 256] tensors; the heavy part of a
generation step (the backbone + head) runs on the HIP kernels.
"""
import math
from functools import partial

import torch
import torch.nn.functional as F

_TINY = 1e-20


def log(t, eps=_TINY):
    """log with a floor (reference sampling.py:20-21)."""
    return t.clamp(min=eps).log()


def gumbel_noise(t, generator=None):
    """-log(-log(U)), U ~ U(0,1) drawn with `generator` in t's shape/dtype/device (reference :24-26)."""
    u = torch.zeros_like(t).uniform_(0, 1, generator=generator)
    return -log(-log(u))


def gumbel_sample(t, temperature=1.0, dim=-1, generator=None):
    """argmax of t/temperature + Gumbel noise (reference :29-30)."""
    scaled = t / max(temperature, 1e-10)
    return (scaled + gumbel_noise(t, generator=generator)).argmax(dim=dim)


def top_k(logits, thres=0.9):
    """keep the ceil((1-thres)*V) largest logits of a [B, N, V] tensor, -inf elsewhere (reference :33-38)."""
    k = math.ceil((1 - thres) * logits.shape[-1])
    vals, idx = logits.topk(k, dim=-1)
    out = torch.full_like(logits, float("-inf"))
    out.scatter_(2, idx, vals)
    return out


def mask_by_random_topk(mask_len, probs, temperature=1.0, generator=None):
    """Re-mask the `mask_len` least confident positions, confidence = log p + temperature * Gumbel
    (reference :41-46).  mask_len: [B, 1]; probs: [B, N]; returns bool [B, N]."""
    confidence = log(probs) + temperature * gumbel_noise(probs, generator=generator)
    ordered = torch.sort(confidence, dim=-1).values
    threshold = torch.gather(ordered, 1, mask_len.long())
    return confidence < threshold


def cosine_schedule(t):
    return torch.cos(t * math.pi * 0.5)


def linear_schedule(t):
    return (1 - t).clamp(min=1e-6, max=1.0)


def pow(t, method):
    """`method` = 'pow<exponent>' (reference :59-63)."""
    exponent = float(method.replace("pow", ""))
    return (1.0 - t ** exponent).clamp(min=1e-6, max=1.0)


def sigmoid_schedule(t, start=-3, end=3, tau=1.0, clip_min=1e-6):
    """gamma schedule built from a sigmoid (reference :66-75)."""
    lo = torch.sigmoid(torch.tensor(start / tau))
    hi = torch.sigmoid(torch.tensor(end / tau))
    cur = torch.sigmoid((t * (end - start) + start) / tau)
    return torch.clip((hi - cur) / (hi - lo), clip_min, 1.0)


def get_mask_chedule(method, **schedule_kwargs):
    if method == "cosine":
        return cosine_schedule
    if method == "linear":
        return linear_schedule
    if "pow" in method:
        return partial(pow, method=method)
    if method == "sigmoid":
        return partial(sigmoid_schedule, **schedule_kwargs)
    raise ValueError("Unknown schedule method: {}".format(method))


def top_k_top_p_filtering(logits, top_k=0, top_p=1.0, filter_value=-float("Inf"), min_tokens_to_keep=1):
    """In-place top-k / nucleus filtering of [B, V] logits (reference :90-128)."""
    if top_k > 0:
        k = min(max(top_k, min_tokens_to_keep), logits.size(-1))
        kth = torch.topk(logits, k)[0][..., -1, None]
        logits[logits < kth] = filter_value
    if top_p < 1.0:
        ordered, order = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(ordered, dim=-1), dim=-1)
        drop = cum > top_p
        if min_tokens_to_keep > 1:
            drop[..., :min_tokens_to_keep] = 0
        drop[..., 1:] = drop[..., :-1].clone()    # keep the first token that crosses the threshold
        drop[..., 0] = 0
        logits[drop.scatter(1, order, drop)] = filter_value
    return logits
