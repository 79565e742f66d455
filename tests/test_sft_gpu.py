"""SFT-path pieces on the GPU (BASELINE configs[2]): mm_projector on the HIP kernels and a UniGen.forward driven by
`input_embeddings` (projected image features spliced between text embeddings) with the mmu_vit mask, against the
CPU oracle under bf16 autocast -- including the gradients that flow back into the projector and the embedding."""
import pytest
import torch
import torch.nn as nn

from helpers import additive, golden, oracle_lm
from test_model_gpu import _check, _tiny_unigen, _rel

pytestmark = pytest.mark.gpu


def test_projector_and_embedding_driven_forward(dev):
    from oracle import host_ref, qwen2_ref
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    model.add_mm_projector(2, 144)
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    ref_proj = nn.Sequential(nn.Linear(144, 256), nn.GELU(), nn.Linear(256, 256))
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in ref_proj.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
        for (n, p), (_, q) in zip(model.mm_projector.named_parameters(), ref_proj.named_parameters()):
            p.copy_(q.to(dev))
    assert [n for n, _ in model.mm_projector.named_parameters()] == ["0.weight", "0.bias", "2.weight", "2.bias"]
    B, n_img, L = 2, 9, 40
    feats = torch.randn(B, n_img, 144, generator=gen)
    pre = torch.randint(0, 290, (B, 5), generator=gen)
    post = torch.randint(0, 290, (B, L - 5 - n_img), generator=gen)
    labels = torch.full((B, L), -100)
    labels[:, 5 + n_img:] = post
    allow = host_ref.mask_mmu_vit_ref(B, L, prefix_length=5, num_tokens=n_img)
    mask = additive(allow)

    def run(embed, proj, mdl_forward, device):
        img = proj(feats.to(device))
        e = torch.cat([embed(pre.to(device)), img.to(torch.float32), embed(post.to(device))], 1)
        return mdl_forward(e)

    # oracle (reference arithmetic: autocast over projector + backbone, as accelerate wraps model.forward; the
    # reference calls mm_projector under the same autocast in train_w_clip_vit.py:803-831)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        img = ref_proj(feats)
    e_ref = torch.cat([lm.model.embed_tokens(pre), img.float(), lm.model.embed_tokens(post)], 1)
    _, _, _, r3 = qwen2_ref.unigen_forward_ref(lm, None, mask, labels, input_embeddings=e_ref, batch_size_mmu=B, autocast=True)
    r3.backward()
    img_h = model.mm_projector(feats.to(dev))
    assert img_h.dtype == torch.bfloat16 and _rel(img_h, img) < 1e-2
    e = torch.cat([model.llm.model.embed_tokens(pre.to(dev)), img_h.float(), model.llm.model.embed_tokens(post.to(dev))], 1)
    _, l1, l2, l3 = model(input_ids=None, input_embeddings=e, attention_mask=mask.to(dev), labels=labels.to(dev), batch_size_mmu=B)
    assert abs(l3.item() - r3.item()) / r3.item() < 1e-3
    l3.backward()
    for (n, p), (_, q) in zip(model.mm_projector.named_parameters(), ref_proj.named_parameters()):
        _check("p.grad", p.grad, q.grad, 5e-2)
    ge = dict(model.llm.named_parameters())["model.embed_tokens.weight"].grad
    _check("ge", ge, lm.model.embed_tokens.weight.grad, 4e-2)


def test_unfrozen_vision_tower_trains_through_the_understanding_path(dev):
    """Reference models/unigen.py:111 builds the tower with freeze=False and training/train_w_clip_vit.py:282,311-312 can make it
    tunable: images -> SigLIP tower (fp32, hand-written backward) -> mm_projector -> backbone -> mmu loss; the gradients that
    arrive in the tower's parameters against the same chain through the CPU oracle (tower in fp32, projector + backbone under
    bf16 autocast)."""
    from models.multimodal_encoder.siglip_encoder import SigLipVisionConfig, SigLipVisionTower
    from oracle import host_ref, qwen2_ref, siglip_ref, weights
    g, g7 = golden("g2_tiny_unigen.pt"), golden("g7_siglip.pt")
    model, _ = _tiny_unigen(g, dev)
    model.add_mm_projector(2, 144)
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    ref_proj = nn.Sequential(nn.Linear(144, 256), nn.GELU(), nn.Linear(256, 256))
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in ref_proj.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * 0.05)
        for (n, p), (_, q) in zip(model.mm_projector.named_parameters(), ref_proj.named_parameters()):
            p.copy_(q.to(dev))
    cfg = SigLipVisionConfig(**g7["cfg"])
    tower = SigLipVisionTower("synthetic-siglip", config=cfg, freeze=False)
    shapes = siglip_ref.siglip_param_shapes(cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, 3, cfg.patch_size, cfg.image_size)
    sd = weights.synth_siglip_state(shapes, seed=g7["weight_seed"])
    own = dict(tower.vision_tower.named_parameters())
    with torch.no_grad():
        for k, v in sd.items():
            if k in own:
                own[k].copy_(v)
    tower = tower.to(dev)
    B, n_img, L = 2, 16, 40
    x = torch.rand(B, 3, 56, 56, generator=gen) * 2 - 1
    pre = torch.randint(0, 290, (B, 5), generator=gen)
    post = torch.randint(0, 290, (B, L - 5 - n_img), generator=gen)
    labels = torch.full((B, L), -100)
    labels[:, 5 + n_img:] = post
    mask = additive(host_ref.mask_mmu_vit_ref(B, L, prefix_length=5, num_tokens=n_img))
    # ---- oracle chain
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    feats_ref = siglip_ref.siglip_tower_ref(ref_sd, x, num_layers_total=cfg.num_hidden_layers, num_heads=cfg.num_attention_heads, patch=14)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        img = ref_proj(feats_ref)
    e_ref = torch.cat([lm.model.embed_tokens(pre), img.float(), lm.model.embed_tokens(post)], 1)
    _, _, _, r3 = qwen2_ref.unigen_forward_ref(lm, None, mask, labels, input_embeddings=e_ref, batch_size_mmu=B, autocast=True)
    r3.backward()
    # ---- HIP chain
    feats = tower(x.to(dev))
    img_h = model.mm_projector(feats)
    e = torch.cat([model.llm.model.embed_tokens(pre.to(dev)), img_h.float(), model.llm.model.embed_tokens(post.to(dev))], 1)
    _, _, _, l3 = model(input_ids=None, input_embeddings=e, attention_mask=mask.to(dev), labels=labels.to(dev), batch_size_mmu=B)
    assert abs(l3.item() - r3.item()) / r3.item() < 1e-3
    l3.backward()
    worst, which, n = 0.0, None, 0
    for k, p in tower.vision_tower.named_parameters():
        rg = ref_sd[k].grad
        if rg is None or k.endswith("k_proj.bias"):         # unused post_layernorm; the exactly-zero key-bias gradient
            continue
        e_ = _rel(p.grad, rg)
        n += 1
        if e_ > worst:
            worst, which = e_, k
    print(f"    gradients reaching {n} tower parameters through projector + backbone: worst rel {worst:.2e} ({which}), gate 5e-2")
    assert n >= 45 and worst < 5e-2, (which, worst)
