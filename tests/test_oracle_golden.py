"""CPU suite, part 1: the oracle (oracle/) against the golden vectors captured from the real
reference (tools/make_golden.py).  This is what pins the oracle; the GPU suite then checks the HIP
path against the oracle and against the same fixtures."""
import math

import pytest

import torch

from helpers import golden, oracle_lm, additive


def test_mask_builders_match_reference():
    from oracle import host_ref
    g = golden("g4_masks.pt")
    ids = g["ids"]
    t2i = host_ref.mask_predict_next_ref(g["t2i_seq"], ids["pad"], ids["soi"], ids["eoi"], rm_pad_in_image=True)
    assert torch.equal(t2i, g["t2i_allow"])
    assert torch.equal(host_ref.to_additive(t2i), g["t2i_additive"])
    assert torch.equal(host_ref.mask_predict_next_ref(g["lm_seq"], ids["pad"], ids["soi"], ids["eoi"]), g["lm_allow"])
    assert torch.equal(host_ref.mask_mmu_ref(g["mmu_seq"], ids["eoi"]), g["mmu_allow"])
    assert torch.equal(host_ref.mask_mmu_vit_ref(2, 24, prefix_length=5, num_tokens=9), g["mmu_vit_allow"])
    # semantics the kernels rely on: no fully blocked row; image rows see every non-pad column
    assert g["t2i_allow"].any(-1).all()


def test_maskgit_train_masking_matches_reference():
    from oracle import host_ref
    g = golden("g5_sampling.pt")
    torch.manual_seed(g["mask_seed"])
    ts, sc = torch.rand(4), torch.rand(4, 16)
    ids, labels, mp = host_ref.maskgit_train_mask_ref(g["mask_tokens"], 332, ts, sc, lambda x: torch.cos(x * math.pi * 0.5))
    assert torch.equal(ids, g["mask_ids"]) and torch.equal(labels, g["mask_labels"]) and torch.equal(mp, g["mask_prob"])


def test_tiny_unigen_oracle_bit_exact_vs_reference():
    from oracle import qwen2_ref
    g = golden("g2_tiny_unigen.pt")
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    mask = additive(g["mask_allow"])
    kw = {k: v for k, v in g["kw"].items() if k != "max_seq_length"}
    for mode, ac in (("fp32", False), ("bf16", True)):
        lm.zero_grad(set_to_none=True)
        logits, l1, l2, l3 = qwen2_ref.unigen_forward_ref(lm, g["input_ids"], mask, g["labels"], autocast=ac, **kw)
        want = g[mode]
        assert torch.equal(logits.to(want["logits"].dtype), want["logits"])
        assert torch.equal(torch.stack([l1, l2, l3]).float(), want["losses"])
        (1.0 * l1 + 0.1 * l2 + 1.0 * l3).backward()
        grads = dict(lm.named_parameters())
        for n, gg in want["grads_small"].items():
            assert torch.equal(grads[n].grad, gg), n
        assert torch.equal(grads["model.embed_tokens.weight"].grad[[0, 5, 300, 303, 304, 312, 320, 332]], want["grad_embed_rows"])
        for n, v in want["grad_norms"].items():
            assert abs(grads[n].grad.norm().item() - v) <= 1e-6 * max(1.0, v), n


def test_magvit_oracle_vs_reference():
    from oracle import magvit_ref, weights
    g = golden("g1_magvit.pt")
    sd = weights.synth_magvit_state(magvit_ref.magvit_param_shapes(), seed=g["weight_seed"])
    assert sum(v.numel() for v in sd.values()) == g["n_params"]
    x = weights.synth_images(2, 256, seed=g["image_seed"])[:1]      # one image keeps the CPU suite short
    with torch.no_grad():
        z = magvit_ref.encode_z_ref(sd, x)
        idx = magvit_ref.get_code_ref(sd, x)
    # mkldnn picks different blockings for different batch sizes / hosts: fp32 round-off level agreement
    assert (z - g["z"][:1]).abs().max().item() < 2e-5
    from oracle.ops_ref import lfq_indices_ref
    safe = (g["z"][:1].abs() > 1e-4)
    bits_ref, bits = (g["z"][:1] > 0), (z > 0)
    assert torch.equal(bits[safe], bits_ref[safe])
    assert (idx != g["indices"][:1]).float().mean().item() <= (~safe).float().mean().item() * 13 + 1e-9
    assert torch.equal(lfq_indices_ref(g["z"]), g["indices"])


def test_maskgit_oracle_reproduces_reference_trajectory():
    """oracle.maskgit_generate_ref with the reference's own randomness (torch.multinomial + uniform_ Gumbel on one
    CPU generator) returns exactly what the real UniGen.t2i_generate returned when the fixture was made."""
    import math
    from oracle import qwen2_ref
    g = golden("g2_tiny_unigen.pt")
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    m = g["maskgit"]
    mask = additive(m["mask_allow"])
    out = qwen2_ref.maskgit_generate_ref(lm, m["input_ids"], m["uncond_ids"], mask, m["scale"], 1.0, m["steps"],
                                         lambda t: torch.cos(t * math.pi * 0.5), 16, g["ids"]["text_vocab"], g["ids"]["mask"],
                                         qwen2_ref.TorchSampler(torch.Generator().manual_seed(m["seed"])))
    assert torch.equal(out, m["result"])


def test_dpo_logps_oracle_matches_reference():
    from oracle import host_ref
    g = golden("g8_dpo_logps.pt")
    for mode in ("mask", "ar"):
        for avg in (False, True):
            got = host_ref.batch_logps_ref(g["logits"], g["labels"], g["n"], average_log_prob=avg, t2i_gen_mode=mode)
            assert torch.equal(got, g[f"{mode}_{int(avg)}"]), (mode, avg)


@pytest.mark.parametrize("fixture", ["g3_wide_layer.pt", "g16_wide_layer_L771.pt"])
def test_wide_layer_oracle_bit_exact_vs_reference(fixture):
    """One decoder layer at the 1.5B model's width (1536 / 8960 / 12:2 heads) on the pt1 sequence shape L = 387 (G3) and on the
    benchmarked L = 771 with left padding (G16, round 5)."""
    from oracle import qwen2_ref
    g = golden(fixture)
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    logits, l1, _, _ = qwen2_ref.unigen_forward_ref(lm, g["input_ids"], additive(g["mask_allow"], torch.float32), g["labels"],
                                                    autocast=True, batch_size_t2i=2, num_vq_tokens=256)
    assert torch.equal(l1, g["loss"])
    assert torch.equal(logits[:, -257:-1:8].to(torch.bfloat16), g["logits_rows"])
    l1.backward()
    for n, p in lm.named_parameters():
        assert abs(p.grad.norm().item() - g["grad_norms"][n]) <= 1e-6 * max(1.0, g["grad_norms"][n]), n


def test_generation_oracles_match_reference_trajectories():
    """G9: ar_generate_ref / mmu_generate_ref return the tokens the real reference's t2i_generate_ar (temperature 1e-6 =
    argmax through its own multinomial) and mmu_generate (temperature 0) returned, in fp32 and under bf16 autocast."""
    from oracle import host_ref, qwen2_ref
    g = golden("g9_generate.pt")
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"], std=g["weight_std"])
    ar, tv = g["ar"], g["ids"]["text_vocab"]
    P, n = ar["P"], ar["n"]
    with torch.no_grad():
        ce, ue = lm.model.embed_tokens(ar["cond"][:, :P]), lm.model.embed_tokens(ar["uncond"][:, :P])
    for mode, ac in (("fp32", False), ("bf16", True)):
        tok, margin = qwen2_ref.ar_generate_ref(lm, ce, ue, n, ar["scale"], tv, key_valid=ar["attention_mask"][:, :P], autocast=ac)
        assert torch.equal(tok.long(), ar[mode]["tokens"]) and len(set(tok[0].tolist())) > 4      # not a degenerate echo
        assert torch.allclose(margin, ar[mode]["margin"])
        mm = g["mmu"]
        mask = host_ref.to_additive(mm["mask_allow"]).to(torch.float32)
        toks, _ = qwen2_ref.mmu_generate_ref(lm, idx=mm["idx"], attention_mask=mask, max_new_tokens=mm["max_new_tokens"], autocast=ac)
        assert toks == mm[mode]["tokens"].tolist() and len(set(toks)) > 6


def test_gen_projector_oracle_matches_reference():
    """G12: unigen_forward_gen_ref vs the real reference built with gen_proj_depth = 2."""
    from oracle import host_ref, qwen2_ref, weights
    g = golden("g12_gen_head.pt")
    cfg = qwen2_ref.Qwen2Cfg(**g["cfg"])
    mask = host_ref.to_additive(g["mask_allow"]).float()
    kw = {k: v for k, v in g["kw"].items() if k != "max_seq_length"}
    for use_dim in (False, True):
        lm = qwen2_ref.RefCausalLM(cfg)
        gen = qwen2_ref.GenHeadRef(g["cfg"]["hidden_size"], g["codebook"], depth=2, use_gen_dim=use_dim, gen_input_dim=16)
        names = [("llm." + k, tuple(p.shape)) for k, p in lm.named_parameters()] + [(k, tuple(p.shape)) for k, p in gen.named_parameters()]
        sd = weights.synth_llm_state(names, seed=g["weight_seed"], std=0.05)
        lm.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("llm.")}, strict=False)
        gen.load_state_dict({k: v for k, v in sd.items() if not k.startswith("llm.")})
        lo, r1, _, _ = qwen2_ref.unigen_forward_gen_ref(lm, gen, g["input_ids"], mask, g["labels"], autocast=True, **kw)
        want = g[f"dim{int(use_dim)}"]
        assert torch.equal(lo.to(torch.bfloat16), want["img_logits"]) and r1.item() == want["loss"].item()


def test_ar_generation_gen_head_oracle_matches_reference():
    """G13: ar_generate_ref(gen=...) returns the tokens of the real reference's t2i_generate_ar on a gen_proj_depth = 2 model
    (models/unigen.py:486-495,512-514), both use_gen_dim settings, fp32 and bf16 autocast."""
    from oracle import qwen2_ref, weights
    g = golden("g13_ar_gen_head.pt")
    cfg = qwen2_ref.Qwen2Cfg(**g["cfg"])
    P, n, tv = g["P"], g["n"], g["ids"]["text_vocab"]
    for use_dim in (False, True):
        lm = qwen2_ref.RefCausalLM(cfg)
        gen = qwen2_ref.GenHeadRef(g["cfg"]["hidden_size"], g["codebook"], depth=2, use_gen_dim=use_dim, gen_input_dim=16)
        names = [("llm." + k, tuple(p.shape)) for k, p in lm.named_parameters()] + [(k, tuple(p.shape)) for k, p in gen.named_parameters()]
        sd = weights.synth_llm_state(names, seed=g["weight_seed"], std=g["weight_std"])
        lm.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("llm.")}, strict=False)
        gen.load_state_dict({k: v for k, v in sd.items() if not k.startswith("llm.")})
        with torch.no_grad():
            ce, ue = lm.model.embed_tokens(g["cond"][:, :P]), lm.model.embed_tokens(g["uncond"][:, :P])
        for mode, ac in (("fp32", False), ("bf16", True)):
            tok, margin = qwen2_ref.ar_generate_ref(lm, ce, ue, n, g["scale"], tv, key_valid=g["attention_mask"][:, :P], autocast=ac, gen=gen)
            want = g[f"dim{int(use_dim)}"][mode]
            assert torch.equal(tok.long(), want["tokens"]) and torch.allclose(margin, want["margin"])
            assert len(set(tok[0].tolist())) > 4 and int(tok.max()) < g["codebook"]


def test_rope_scaling_oracle_matches_reference():
    """G14: the oracle's rope with config.rope_scaling = {"factor": 2.0, "type": "linear" | "dynamic"} (reference
    models/unigen.py:38-40,61-64; max_position_embeddings = 16 < L so the dynamic-NTK base applies) reproduces the real
    reference's logits and losses bit for bit, fp32 and bf16 autocast."""
    from oracle import host_ref, qwen2_ref, weights
    g, g2 = golden("g14_rope_scaling.pt"), golden("g2_tiny_unigen.pt")
    mask = host_ref.to_additive(g2["mask_allow"]).float()
    kw = {k: v for k, v in g2["kw"].items() if k != "max_seq_length"}
    for kind in ("linear", "dynamic"):
        lm = qwen2_ref.RefCausalLM(qwen2_ref.Qwen2Cfg(**g["cfg"], rope_scaling={"factor": g["factor"], "type": kind},
                                                      max_position_embeddings=g["max_position_embeddings"]))
        names = [(n, tuple(p.shape)) for n, p in lm.named_parameters()]
        lm.load_state_dict(weights.synth_llm_state(names, seed=g["weight_seed"]), strict=False)
        for mode, ac in (("fp32", False), ("bf16", True)):
            with torch.no_grad():
                lo, r1, r2, r3 = qwen2_ref.unigen_forward_ref(lm, g2["input_ids"], mask, g2["labels"], autocast=ac, **kw)
            assert torch.equal(lo, g[kind][mode]["logits"].float())
            assert torch.equal(torch.stack([r1, r2, r3]).float(), g[kind][mode]["losses"])
        assert not torch.equal(g[kind]["fp32"]["logits"], g2["fp32"]["logits"])
