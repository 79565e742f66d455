"""A control for the logits gate, from the reference's OWN arithmetic.

north_star asks <= 1e-3 relative for bf16 logits; the GPU tests hold 6.6e-3 on the one- and two-layer fixtures (2.8e-2 at 28 layers)
plus the fp32 yardstick, arguing that two bf16 evaluations of the same network that differ only in summation order are that far apart.
Until round 6 the only evidence for that was the HIP-vs-oracle distance itself.  Here the pinned oracle (bit-identical to the real
reference on CPU, tests/test_oracle_golden.py) runs its bf16-autocast forward twice on G2 / G3 / G16 -- as is, and with the contraction
axis of EVERY nn.Linear permuted (x[..., perm] @ W[:, perm].T: the same mathematics, a different fp32 summation order inside the GEMM,
nothing else touched: norms and every rounding point stay where they are) -- and a third time in fp32.  A second control swaps the
attention implementation for transformers' OTHER Qwen2 attention (`eager_attention_forward`, modeling_qwen2.py: q k^T, fp32 softmax
cast back to bf16, P v) in place of `sdpa`: the reference's config picks either, both are "the reference's eager path".

If the permuted run sat < 1e-3 from the plain run, the GPU gate would be hiding a rounding-point difference.  It does not: see
profiles/r06_logits_control.txt (written by `UNIGEN_WRITE_PROFILES=1 pytest tests/test_logits_control_cpu.py -s`)."""
import os

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import LOGITS_GATE, NORTH_STAR_LOGITS, additive, golden, oracle_lm, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class permuted_linears:
    """Every nn.Linear of `model` contracts over a fixed random permutation of its input features while the context is active."""

    def __init__(self, model, seed=0):
        self.model, self.seed, self.saved = model, seed, []

    def __enter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for m in self.model.modules():
            if isinstance(m, nn.Linear):
                perm = torch.randperm(m.in_features, generator=g)
                self.saved.append((m, m.forward))
                m.forward = (lambda x, m=m, perm=perm: F.linear(x[..., perm], m.weight[:, perm], m.bias))
        return self

    def __exit__(self, *exc):
        for m, f in self.saved:
            m.forward = f
        return False


class eager_attention:
    """F.scaled_dot_product_attention replaced by transformers' eager attention arithmetic (modeling_qwen2.py eager_attention_forward)."""

    def __enter__(self):
        self.saved = F.scaled_dot_product_attention

        def eager(q, k, v, attn_mask=None, dropout_p=0.0, scale=None, is_causal=False):
            w = torch.matmul(q, k.transpose(2, 3)) * scale
            if is_causal:
                L, S = q.shape[-2], k.shape[-2]
                w = w + torch.full((L, S), float("-inf")).triu(1 + S - L)
            elif attn_mask is not None:
                w = w + attn_mask
            w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
            return torch.matmul(w, v)
        F.scaled_dot_product_attention = eager
        return self

    def __exit__(self, *exc):
        F.scaled_dot_product_attention = self.saved
        return False


def _forward(lm, g, autocast, wide):
    from oracle import qwen2_ref
    if wide:
        logits, l1, _, _ = qwen2_ref.unigen_forward_ref(lm, g["input_ids"], additive(g["mask_allow"], torch.float32), g["labels"],
                                                        autocast=autocast, batch_size_t2i=2, num_vq_tokens=256)
        return logits[:, -257:-1].float(), l1.float()
    kw = {k: v for k, v in g["kw"].items() if k != "max_seq_length"}
    logits, l1, l2, l3 = qwen2_ref.unigen_forward_ref(lm, g["input_ids"], additive(g["mask_allow"]), g["labels"], autocast=autocast, **kw)
    return logits.float(), (l1 + l2 + l3).float()


@pytest.mark.parametrize("fixture,wide", [("g2_tiny_unigen.pt", False), ("g3_wide_layer.pt", True), ("g16_wide_layer_L771.pt", True)])
def test_summation_order_alone_moves_reference_logits_beyond_north_star(fixture, wide):
    g = golden(fixture)
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    with torch.no_grad():
        plain, loss_plain = _forward(lm, g, True, wide)
        exact, loss_exact = _forward(lm, g, False, wide)
        dists, dloss = [], []
        for seed in (0, 1, 2):
            with permuted_linears(lm, seed):
                perm, loss_perm = _forward(lm, g, True, wide)
                perm32, _ = _forward(lm, g, False, wide)
            dists.append((rel_err(perm, plain), rel_err(perm, exact), rel_err(perm32, exact)))
            dloss.append(abs(float(loss_perm - loss_plain)) / abs(float(loss_plain)))
        with eager_attention():
            eag, loss_eag = _forward(lm, g, True, wide)
            eag32, _ = _forward(lm, g, False, wide)
            with permuted_linears(lm, 0):
                both, loss_both = _forward(lm, g, True, wide)
        dloss += [abs(float(loss_eag - loss_plain)) / abs(float(loss_plain)), abs(float(loss_both - loss_plain)) / abs(float(loss_plain))]
    d_plain = rel_err(plain, exact)
    lines = [f"{fixture}: {sum(p.numel() for p in lm.parameters()) / 1e6:.1f} M parameters, {len(list(lm.model.layers))} layer(s), logits {tuple(plain.shape)}",
             f"  reference bf16 vs reference fp32                      {d_plain:.3e}"]
    for s, (d_pp, d_pe, d_32) in enumerate(dists):
        lines.append(f"  permuted-contraction bf16 run {s} vs the plain bf16 run   {d_pp:.3e}   (vs fp32 {d_pe:.3e}; the same permutation in fp32 "
                     f"moves the fp32 logits by {d_32:.1e}); loss moves by {dloss[s]:.1e} relative")
    d_eag, d_both = rel_err(eag, plain), rel_err(both, plain)
    lines.append(f"  eager-attention bf16 run vs the plain (sdpa) bf16 run     {d_eag:.3e}   (vs fp32 {rel_err(eag, exact):.3e}; in fp32 the two attentions "
                 f"differ by {rel_err(eag32, exact):.1e})")
    lines.append(f"  eager attention + permuted contractions vs plain          {d_both:.3e}   (vs fp32 {rel_err(both, exact):.3e})")
    worst = max(max(d[0] for d in dists), d_eag, d_both)
    lines.append(f"  => bf16 evaluations of the reference that differ ONLY in summation order / attention implementation: up to {worst:.3e} apart "
                 f"= {worst / NORTH_STAR_LOGITS:.1f} x north_star's 1e-3; GPU gate {LOGITS_GATE:.1e} = {LOGITS_GATE / worst:.2f} x this")
    print("\n".join("    " + ln for ln in lines))
    if os.environ.get("UNIGEN_WRITE_PROFILES") == "1":
        with open(os.path.join(ROOT, "profiles", "r06_logits_control.txt"), "a") as f:
            f.write("\n".join(lines) + "\n\n")
    # the same permutation in fp32 is invisible at this scale: what moves the bf16 logits is rounding, not the mathematics
    assert max(d[2] for d in dists) < 1e-5
    # the reference's own order noise exceeds north_star's bar ...
    assert min(d[0] for d in dists) > NORTH_STAR_LOGITS, dists
    # ... and the gate the GPU tests hold is within 1.5 x of that spread (it would hide a rounding-point difference otherwise)
    assert LOGITS_GATE < 1.5 * worst, (LOGITS_GATE, worst)
    # the loss is insensitive (north_star's 1e-3 holds for it, here as on the GPU)
    assert max(dloss) < 1e-3, dloss
