"""SigLIP vision tower (HIP fp32 path) vs the golden vector captured from the reference class and vs the CPU
oracle at the real so400m width."""
import pytest
import torch

from helpers import golden

pytestmark = pytest.mark.gpu


def _tower(cfg_kw, seed, dev):
    from models.multimodal_encoder.siglip_encoder import SigLipVisionConfig, SigLipVisionTower
    from oracle import siglip_ref, weights
    cfg = SigLipVisionConfig(**cfg_kw)
    tower = SigLipVisionTower("synthetic-siglip", config=cfg, freeze=True)
    shapes = siglip_ref.siglip_param_shapes(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, 3, cfg.patch_size,
                                            cfg.image_size)
    sd = weights.synth_siglip_state(shapes, seed=seed)
    # the tower already dropped its last layer: its parameters are simply absent from the module
    own = dict(tower.vision_tower.named_parameters())
    with torch.no_grad():
        for k, v in sd.items():
            if k in own:
                own[k].copy_(v)
    assert len(tower.vision_tower.vision_model.encoder.layers) == cfg.num_hidden_layers - 1
    return tower.to(dev), sd, cfg


def test_siglip_small_matches_reference_golden(dev):
    g = golden("g7_siglip.pt")
    tower, _, cfg = _tower(g["cfg"], g["weight_seed"], dev)
    x = torch.rand(2, 3, 56, 56, generator=torch.Generator().manual_seed(g["image_seed"])) * 2 - 1
    out = tower(x.to(dev)).cpu()
    assert out.shape == g["out"].shape == (2, 16, 144)
    assert (out - g["out"]).abs().max().item() < 2e-5
    lst = tower([x[0].to(dev), x[1].to(dev)])
    assert torch.allclose(torch.cat(lst).cpu(), out, atol=1e-6)
    assert tower.hidden_size == 144 and tower.num_patches == 16 and tower.dtype == torch.float32


def test_siglip_so400m_width_vs_oracle(dev):
    from oracle import siglip_ref
    kw = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=3, num_attention_heads=16, image_size=384, patch_size=14)
    tower, sd, cfg = _tower(kw, 43, dev)
    x = torch.rand(1, 3, 384, 384, generator=torch.Generator().manual_seed(44)) * 2 - 1
    ref = siglip_ref.siglip_tower_ref(sd, x, num_layers_total=3, num_heads=16, patch=14)
    out = tower(x.to(dev)).cpu()
    assert out.shape == (1, 729, 1152)
    err = (out - ref).abs().max().item()
    assert err < 5e-4 * max(1.0, ref.abs().max().item()), err


def _unfrozen_grads(cfg_kw, seed, B, dev, gate):
    """tower(images) with freeze=False -> weighted sum -> backward, against torch autograd through the CPU oracle (plain fp32
    torch ops: the reference's arithmetic, models/multimodal_encoder/siglip_encoder.py:152-309 with the last layer dropped)"""
    from models.multimodal_encoder.siglip_encoder import SigLipVisionConfig, SigLipVisionTower
    from oracle import siglip_ref, weights
    cfg = SigLipVisionConfig(**cfg_kw)
    tower = SigLipVisionTower("synthetic-siglip", config=cfg, freeze=False)
    shapes = siglip_ref.siglip_param_shapes(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, 3, cfg.patch_size, cfg.image_size)
    sd = weights.synth_siglip_state(shapes, seed=seed)
    own = dict(tower.vision_tower.named_parameters())
    with torch.no_grad():
        for k, v in sd.items():
            if k in own:
                own[k].copy_(v)
    tower = tower.to(dev)
    assert all(p.requires_grad for p in tower.vision_tower.parameters())
    gen = torch.Generator().manual_seed(seed + 1)
    x = torch.rand(B, 3, cfg.image_size, cfg.image_size, generator=gen) * 2 - 1
    T = (cfg.image_size // cfg.patch_size) ** 2
    w = torch.randn(B, T, cfg.hidden_size, generator=gen)
    out = tower(x.to(dev))
    assert out.requires_grad
    (out * w.to(dev)).sum().backward()
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = siglip_ref.siglip_tower_ref(ref_sd, x, num_layers_total=cfg.num_hidden_layers, num_heads=cfg.num_attention_heads, patch=cfg.patch_size)
    (ref * w).sum().backward()
    assert (out.detach().cpu() - ref.detach()).abs().max().item() < 5e-4 * max(1.0, ref.abs().max().item())
    worst, which, n = 0.0, None, 0
    for k, p in tower.vision_tower.named_parameters():
        rg = ref_sd[k].grad
        if rg is None:                       # post_layernorm: the tower returns the hidden state BEFORE it (:579-590)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        n += 1
        if k.endswith("k_proj.bias"):
            # softmax is invariant to a shift of all scores of a row, so d(loss)/d(k bias) = sum_q (sum_k dS[q, k]) Q[q] is
            # exactly zero: both sides hold rounding noise -- compare it with the q bias gradient's size instead of with itself
            qb = ref_sd[k.replace("k_proj", "q_proj")].grad.norm().item()
            assert rg.norm().item() < 1e-4 * qb and p.grad.norm().item() < 1e-3 * qb, (k, rg.norm().item(), p.grad.norm().item(), qb)
            continue
        e = ((p.grad.cpu() - rg).norm() / (rg.norm() + 1e-30)).item()
        if e > worst:
            worst, which = e, k
    print(f"    unfrozen SigLIP tower ({cfg.hidden_size} wide, {cfg.num_hidden_layers - 1} layers, B={B}): {n} parameter gradients, "
          f"worst relative Frobenius error {worst:.2e} ({which}), gate {gate:.0e}")
    assert n == len(ref_sd) - 2 - 16 and worst < gate, (which, worst)      # every kept parameter (last layer + post_layernorm unused)
    # a frozen tower still takes the inference path and returns the same features
    tower.vision_tower.requires_grad_(False)
    with torch.no_grad():
        assert torch.allclose(tower(x.to(dev)), out.detach(), atol=2e-5, rtol=1e-5)


def test_siglip_unfrozen_backward_small(dev):
    g = golden("g7_siglip.pt")
    _unfrozen_grads(g["cfg"], g["weight_seed"], 2, dev, 1e-4)


def test_siglip_unfrozen_backward_so400m_width(dev):
    """so400m width, two kept layers, ONE image: 729 tokens (not a multiple of 4: every zero-padded contraction of the backward)"""
    kw = dict(hidden_size=1152, intermediate_size=4304, num_hidden_layers=3, num_attention_heads=16, image_size=384, patch_size=14)
    _unfrozen_grads(kw, 45, 1, dev, 1e-4)
