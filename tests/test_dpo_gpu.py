"""DPO step (BASELINE configs[4], reference training/train_dpo.py:51-90, 573-647) through the drop-in API: the
reference's own get_batch_logps arithmetic applied to the lazy logits must match the CPU oracle, values and
gradients."""
import pytest
import torch
import torch.nn.functional as F

from helpers import additive, golden, oracle_lm
from test_model_gpu import _check, _tiny_unigen, _rel

pytestmark = pytest.mark.gpu


def batch_logps(logits, labels, n):
    """get_batch_logps, 'mask' mode (train_dpo.py:74-90): the oracle's restatement, pinned bit-exactly to the real
    function by tests/golden/g8_dpo_logps.pt (tests/test_oracle_golden.py)."""
    from oracle import host_ref
    return host_ref.batch_logps_ref(logits, labels, n)


def test_dpo_loss_and_grads_match_oracle(dev):
    from oracle import host_ref, qwen2_ref
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    ids = g["ids"]
    gen = torch.Generator().manual_seed(11)
    Bp, L, n = 2, 40, 16
    seq = torch.randint(0, 290, (2 * Bp, L), generator=gen)
    seq[:, -(n + 2)] = ids["soi"]; seq[:, -1] = ids["eoi"]
    seq[0, :4] = ids["pad"]
    img = torch.randint(312, 332, (2 * Bp, n), generator=gen)
    msk = torch.rand(2 * Bp, n, generator=gen) < 0.5
    msk[:, 0] = True
    seq[:, -(n + 1):-1] = torch.where(msk, ids["mask"], img)
    labels = torch.full((2 * Bp, L), -100)
    labels[:, -(n + 1):-1] = torch.where(msk, img, -100)
    mask = additive(host_ref.mask_predict_next_ref(seq, ids["pad"], ids["soi"], ids["eoi"], rm_pad_in_image=True))
    ref_lp = torch.randn(2 * Bp, generator=gen)            # frozen reference-model log-probs (any constants)
    beta = 0.5

    def dpo(lp):
        pi = lp[:Bp] - lp[Bp:]
        rf = ref_lp[:Bp].to(lp.device) - ref_lp[Bp:].to(lp.device)
        return -F.logsigmoid(beta * (pi - rf)).mean()

    lo = qwen2_ref.unigen_forward_ref(lm, seq, mask, None, batch_size_t2i=2 * Bp, autocast=True)
    lp_ref = batch_logps(lo.float(), labels, n)
    loss_ref = dpo(lp_ref)
    loss_ref.backward()
    all_logits = model(input_ids=seq.to(dev), attention_mask=mask.to(torch.bfloat16).to(dev), batch_size_t2i=2 * Bp)
    all_logits = all_logits.to(torch.float32)
    assert tuple(all_logits.shape[:-1]) == tuple(labels.shape)
    lp = batch_logps(all_logits, labels.to(dev), n)
    _check("lp", lp, lp_ref, 2e-3)
    loss = dpo(lp)
    assert abs(loss.item() - loss_ref.item()) < 2e-3 * max(1.0, abs(loss_ref.item()))
    loss.backward()
    ref_g = dict(lm.named_parameters())
    for nme, p in model.llm.named_parameters():
        _check("p.grad", p.grad, ref_g[nme].grad, 6e-2)


def test_fused_batch_logps_matches_reference_function(dev):
    """unigen_hip.dpo.get_batch_logps (head on the image rows only + one CE pass, per-row backward) == the reference
    function applied to the lazy logits: values in both modes / with averaging, and the gradients it sends back."""
    from unigen_hip.dpo import get_batch_logps
    from oracle import host_ref
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    ids = g["ids"]
    gen = torch.Generator().manual_seed(12)
    B, L, n = 4, 40, 16
    seq = torch.randint(0, 290, (B, L), generator=gen)
    seq[:, -(n + 2)] = ids["soi"]; seq[:, -1] = ids["eoi"]
    img = torch.randint(312, 332, (B, n), generator=gen)
    msk = torch.rand(B, n, generator=gen) < 0.5
    msk[:, 0] = True
    seq[:, -(n + 1):-1] = torch.where(msk, ids["mask"], img)
    labels = torch.full((B, L), -100)
    labels[:, -(n + 1):-1] = torch.where(msk, img, -100)
    mask = additive(host_ref.mask_predict_next_ref(seq, ids["pad"], ids["soi"], ids["eoi"], rm_pad_in_image=True)).to(dev)
    up = torch.randn(B, generator=gen).to(dev)
    grads = []
    for fused in (False, True):
        model.zero_grad(set_to_none=True)
        lz = model(input_ids=seq.to(dev), attention_mask=mask, batch_size_t2i=B)
        if fused:
            lp = get_batch_logps(lz, labels.to(dev), num_vq_tokens=n)
        else:
            lp = host_ref.batch_logps_ref(lz.to(torch.float32), labels.to(dev), n)
        (lp * up).sum().backward()
        grads.append((lp.detach().float().cpu(), {k: p.grad.clone() for k, p in model.llm.named_parameters()}))
    _check("grads[1][0]", grads[1][0], grads[0][0], 1e-4)
    for k in grads[0][1]:
        _check("grads[1][1][k]", grads[1][1][k], grads[0][1][k], 2e-2)
    lz = model(input_ids=seq.to(dev), attention_mask=mask, batch_size_t2i=B)
    for mode in ("mask", "ar"):
        for avg in (False, True):
            a = get_batch_logps(lz, labels.to(dev), average_log_prob=avg, num_vq_tokens=n, t2i_gen_mode=mode)
            b = host_ref.batch_logps_ref(lz.to(torch.float32), labels.to(dev), n, average_log_prob=avg, t2i_gen_mode=mode)
            _check("a", a, b, 1e-4)
