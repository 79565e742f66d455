"""SURVEY.md section 8 row a6 on the device: `ug_maskgit_train_mask` / `data.masking.mask_or_random_replace_tokens` against the
real reference's outputs (golden G5, captured with torch.manual_seed(123)) and the pinned oracle on larger shapes.  Integer
outputs: bit-exact."""
import math
import types

import pytest
import torch

from helpers import golden

pytestmark = pytest.mark.gpu
cosine = lambda t: torch.cos(t * math.pi * 0.5)


def test_train_mask_kernel_matches_reference_golden(dev):
    from unigen_hip import ops
    g = golden("g5_sampling.pt")
    torch.manual_seed(g["mask_seed"])
    ts, sc = torch.rand(4), torch.rand(4, 16)                 # the draws the reference made on the CPU generator
    mp = cosine(ts).clip(0.0)
    k = (16 * mp).round().clamp(min=1)
    ids, labels = ops.maskgit_train_mask(g["mask_tokens"].to(dev), sc.to(dev), k.to(dev), 332)
    assert torch.equal(ids.cpu(), g["mask_ids"]) and torch.equal(labels.cpu(), g["mask_labels"])
    assert torch.equal(mp, g["mask_prob"])


@pytest.mark.parametrize("B,n", [(1, 1), (7, 256), (24, 256), (3, 1024), (2, 4000)])
def test_train_mask_kernel_matches_oracle(dev, B, n):
    from oracle import host_ref
    from unigen_hip import ops
    gen = torch.Generator().manual_seed(B * 1000 + n)
    toks = torch.randint(151674, 151674 + 8192, (B, n), generator=gen)
    ts, sc = torch.rand(B, generator=gen), torch.rand(B, n, generator=gen)
    if n >= 256:
        sc[0, 5] = sc[0, 77]                                   # a tie: stable order (lower index first), like a stable argsort
        ts[0] = 0.0                                            # mask_prob 1: everything masked
    want_ids, want_lab, mp = host_ref.maskgit_train_mask_ref(toks, 159866, ts, sc, cosine)
    if n >= 256:                                               # torch.argsort is not stable: restate the tie rule explicitly
        perm = sc.argsort(dim=-1, stable=True)
        mask = perm < (n * mp).round().clamp(min=1).unsqueeze(-1)
        want_ids, want_lab = torch.where(mask, 159866, toks), torch.where(mask, toks, -100)
    k = (n * mp).round().clamp(min=1)
    ids, labels = ops.maskgit_train_mask(toks.to(dev), sc.to(dev), k.to(dev), 159866)
    assert torch.equal(ids.cpu(), want_ids) and torch.equal(labels.cpu(), want_lab)
    assert int((labels != -100).sum(1).cpu()[0]) == int(k[0])


def test_masking_module_is_a_drop_in(dev):
    """data.masking.mask_or_random_replace_tokens: same signature / return tuple as the reference's, same draws in the same
    order from the device generator (timesteps [B], then scores [B, n])."""
    from data.masking import mask_or_random_replace_tokens
    from oracle import host_ref
    cfg = types.SimpleNamespace(training=types.SimpleNamespace(min_masking_rate=0.0, get=lambda k, d=None: d),
                                model=types.SimpleNamespace(codebook_size=8192))
    toks = torch.randint(151674, 151674 + 8192, (16, 256), device=dev)
    torch.manual_seed(99)
    ids, labels, lw, mp = mask_or_random_replace_tokens(toks, 159866, cfg, cosine)
    torch.manual_seed(99)
    ts, sc = torch.rand(16, device=dev), torch.rand(16, 256, device=dev)
    w_ids, w_lab, w_mp = host_ref.maskgit_train_mask_ref(toks.cpu(), 159866, ts.cpu(), sc.cpu(), cosine)
    assert lw is None and torch.equal(ids.cpu(), w_ids) and torch.equal(labels.cpu(), w_lab)
    assert torch.allclose(mp.cpu(), w_mp, atol=1e-6)


def test_masking_options_on_the_device(dev):
    """The optional branches with device tensors: `eval_mask_ratios` through the masking kernel (exactly round(n * ratio) masked
    positions per row, ratios from the configured list), and the training-time contiguous-region branch (one filled rectangle per
    image covering at least the requested number of positions); golden G15 pins both against the real reference on the host."""
    import random
    from data.masking import mask_or_random_replace_tokens

    class Node(dict):
        __getattr__ = dict.__getitem__
    toks = torch.randint(151674, 151674 + 8192, (12, 256), device=dev)
    cfg = types.SimpleNamespace(training=Node(min_masking_rate=0.0, eval_mask_ratios=[0.25, 0.5, 0.9]), model=types.SimpleNamespace(codebook_size=8192))
    random.seed(3)
    ids, labels, lw, mp = mask_or_random_replace_tokens(toks, 159866, cfg, cosine, is_train=False)
    assert lw is None and set(round(float(v), 2) for v in mp.cpu()) <= {0.25, 0.5, 0.9}
    assert torch.equal((labels != -100).sum(1).cpu(), (256 * mp).round().long().cpu())
    assert torch.equal(torch.where(labels != -100, labels, ids).cpu(), toks.cpu()) and bool(((ids == 159866) == (labels != -100)).all())
    cfg = types.SimpleNamespace(training=Node(min_masking_rate=0.0, mask_contiguous_region_prob=1.0), model=types.SimpleNamespace(codebook_size=8192))
    random.seed(4)
    torch.manual_seed(4)
    ids, labels, lw, mp = mask_or_random_replace_tokens(toks, 159866, cfg, cosine)
    m = (labels != -100).view(12, 16, 16).cpu()
    want = (256 * mp).round().clamp(min=1).cpu()
    for b in range(12):
        rows, cols = m[b].any(1).nonzero().flatten(), m[b].any(0).nonzero().flatten()
        assert int(m[b].sum()) == len(rows) * len(cols) >= min(int(want[b]), 256) - 16
    assert bool(((ids == 159866) == (labels != -100)).all())
