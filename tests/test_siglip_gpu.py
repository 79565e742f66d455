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
