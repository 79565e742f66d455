"""Drop-in boundary on the GPU: checkpoints written by the REAL reference's save_pretrained load
through this build's from_pretrained (golden G10) and this build writes the reference's file names; `prepare_inputs_for_mmu`
and `generate` reproduce what the real reference returned (golden G11)."""
import json
import os
import types

import pytest
import torch

from helpers import GOLDEN, additive, fp32_yardstick, golden, llm_config_dir

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("name", ["ckpt_ref_single", "ckpt_ref_sharded"])
def test_reference_written_checkpoint_loads(dev, name):
    """pytorch_model.bin, and pytorch_model-0000i-of-0000N.bin + diffusion_pytorch_model.bin.index.json, both produced by
    the reference's own ModelMixin.save_pretrained (tools/make_golden.py checkpoint)."""
    from models import UniGen
    from oracle import weights
    g = golden("g10_checkpoint.pt")
    model, info = UniGen.from_pretrained(os.path.join(GOLDEN, name), llm_model_path=llm_config_dir(g["cfg"]), device=dev,
                                         output_loading_info=True)
    assert not info["missing_keys"] and not info["unexpected_keys"], info
    sd = model.state_dict()
    assert sorted(sd) == g["keys"]
    want = weights.synth_llm_state([(k[4:], tuple(v.shape)) for k, v in sd.items() if k != "llm.lm_head.weight"], seed=g["weight_seed"])
    for k, v in want.items():
        assert torch.equal(sd["llm." + k].cpu(), v), k
    assert sd["llm.lm_head.weight"].data_ptr() == sd["llm.model.embed_tokens.weight"].data_ptr()
    logits = model(input_ids=g["input_ids"].to(dev), attention_mask=None)[:, -3:, :].float().cpu()
    err = _rel(logits, g["logits_last"])
    print(f"{name}: logits of the loaded model vs the reference's fp32 logits: rel {err:.2e}")
    assert err < 1e-2


def test_save_pretrained_writes_reference_file_names(dev, tmp_path):
    from models import UniGen
    g = golden("g10_checkpoint.pt")
    cfg_dir = llm_config_dir(g["cfg"])
    model = UniGen.from_pretrained(os.path.join(GOLDEN, "ckpt_ref_single"), llm_model_path=cfg_dir, device=dev)
    a = model.state_dict()
    cases = [("bin", dict(safe_serialization=False), {"config.json", "pytorch_model.bin"}),
             ("st", dict(safe_serialization=True), {"config.json", "pytorch_model.safetensors"}),
             ("bin_sharded", dict(safe_serialization=False, max_shard_size="200KB"),
              {"config.json", "diffusion_pytorch_model.bin.index.json", "pytorch_model-00001-of-00003.bin",
               "pytorch_model-00002-of-00003.bin", "pytorch_model-00003-of-00003.bin"}),
             ("st_sharded", dict(safe_serialization=True, max_shard_size=200_000), None)]
    for name, kw, files in cases:
        out = str(tmp_path / name)
        model.save_pretrained(out, **kw)
        got = set(os.listdir(out))
        if files is not None:
            assert got == files, (name, got)
        else:
            assert "diffusion_pytorch_model.safetensors.index.json" in got and "pytorch_model-00001-of-00003.safetensors" in got
        cfg = json.load(open(os.path.join(out, "config.json")))
        assert cfg["_class_name"] == "UniGen" and "device" not in cfg and "init_seed" not in cfg
        again = UniGen.from_pretrained(out, llm_model_path=cfg_dir, device=dev)
        b = again.state_dict()
        assert sorted(a) == sorted(b) and all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a), name
    # the sharded .bin split equals the reference's (same greedy rule): same weight_map as the committed fixture
    mine = json.load(open(str(tmp_path / "bin_sharded" / "diffusion_pytorch_model.bin.index.json")))
    ref = json.load(open(os.path.join(GOLDEN, "ckpt_ref_sharded", "diffusion_pytorch_model.bin.index.json")))
    assert mine["weight_map"] == ref["weight_map"] and mine["metadata"]["total_size"] == ref["metadata"]["total_size"]


def _und_model(g, dev):
    from models import UniGen
    from oracle import weights
    cfg, ids = g["cfg"], g["ids"]
    m = UniGen(w_und_encoder=True, vocab_size=cfg["vocab_size"], llm_vocab_size=ids["text_vocab"], llm_model_path=llm_config_dir(cfg),
               codebook_size=20, num_vq_tokens=16, load_from_pretrained=True, mm_input_dim=g["mm_input_dim"], und_proj_depth=2,
               device=dev, init_seed=1)
    names = [(n, tuple(p.shape)) for n, p in m.named_parameters() if not n.startswith("llm.lm_head")]
    res = m.load_state_dict(weights.synth_llm_state(names, seed=g["weight_seed"], std=g["weight_std"]), strict=False)
    assert not res.unexpected_keys
    return m


def test_prepare_inputs_for_mmu_matches_reference(dev):
    g = golden("g11_mmu_inputs.pt")
    model = _und_model(g, dev)
    t = g["template"]
    tmpl = types.SimpleNamespace(text_tokenizer=types.SimpleNamespace(pad_token_id=t["pad_token_id"]), max_seq_len=t["max_seq_len"],
                                 sptids_dict={k: torch.tensor([v]) for k, v in t["sptids"].items()}, ignore_id=t["ignore_id"],
                                 eos_token_id=t["eos_token_id"], task_token_first=t["task_token_first"])
    x = {k: v.to(dev) for k, v in g["mmu_in"].items()}
    for mode in ("train", "eval"):
        model.train(mode == "train")
        for tag in ("nosys", "sys"):
            want = g[f"mmu_{mode}_{tag}"]
            e, am, lab, p1 = model.prepare_inputs_for_mmu(x["image_feats"], x["spatial_shapes"], x["input_ids"], x["label_ids"], tmpl,
                                                          x["input_ids_system"] if tag == "sys" else None)
            assert torch.equal(am.cpu(), want["attention_mask"]) and torch.equal(lab.cpu(), want["labels"]), (mode, tag)
            assert torch.equal(p1.cpu(), want["part1"])
            assert e.shape == want["embeddings"].shape
            L1 = p1.shape[1]
            n_img = (g["mmu_in"]["spatial_shapes"][:, 0] * g["mmu_in"]["spatial_shapes"][:, 1]).tolist()
            img = torch.zeros(e.shape[:2], dtype=torch.bool)
            for b, n in enumerate(n_img):
                img[b, L1:L1 + n] = True
            assert torch.equal(e.float().cpu()[~img], want["embeddings"][~img]), (mode, tag)     # embedding rows: exact fp32 copies
            err = _rel(e.float().cpu()[img], want["embeddings"][img])                             # projector rows: bf16 GEMM
            print(f"prepare_inputs_for_mmu[{mode},{tag}]: ids/labels/mask exact; projector rows rel {err:.2e}")
            assert err < 1e-2
    model.train()
    e, _, lab, _ = model.prepare_inputs_for_mmu(x["image_feats"], x["spatial_shapes"], x["input_ids"], x["label_ids"], tmpl, None)
    e.float().pow(2).sum().backward()                     # differentiable into the projector and the embedding table
    assert model.mm_projector[0].weight.grad is not None and model.llm.model.embed_tokens.weight.grad.abs().sum() > 0


def test_generate_matches_reference_generate(dev):
    """UniGen.generate vs the real reference (transformers GenerationMixin under UniGen.generate), greedy: ids in ->
    prompt + continuation, embeddings in -> continuation only, eos handling with per-row padding."""
    g = golden("g11_mmu_inputs.pt")
    model = _und_model(g, dev).eval()
    gg = g["generate"]
    prompt, am, new = gg["prompt"].to(dev), gg["attention_mask"].to(dev), gg["max_new_tokens"]
    P = prompt.shape[1]
    pad = g["template"]["pad_token_id"]
    full = model.generate(input_ids=prompt, attention_mask=am, max_new_tokens=new, do_sample=False, use_cache=True,
                          pad_token_id=g["template"]["eos_token_id"]).cpu()
    assert full.shape == gg["full"].shape and torch.equal(full[:, :P], gg["prompt"])
    compared = 0
    for r in range(full.shape[0]):
        for i in range(new):
            if gg["margin"][r, i] < 0.05:
                break
            assert full[r, P + i] == gg["full"][r, P + i], (r, i, full[r, P:], gg["full"][r, P:])
            compared += 1
    assert compared >= 14, compared
    cont = model.generate(input_embeddings=model.llm.model.embed_tokens(prompt), attention_mask=am, max_new_tokens=new,
                          do_sample=False, use_cache=True, pad_token_id=pad).cpu()
    assert cont.shape == gg["cont"].shape and torch.equal(cont, full[:, P:])
    stop = model.generate(input_ids=prompt, attention_mask=am, max_new_tokens=new, do_sample=False, pad_token_id=pad,
                          eos_token_id=gg["eos"]).cpu()
    if torch.equal(full, gg["full"]):
        assert torch.equal(stop, gg["stop"]), (stop, gg["stop"])
    smp = model.generate(input_ids=prompt, attention_mask=am, max_new_tokens=4, do_sample=True, temperature=0.7, top_k=5, top_p=0.9,
                         generator=torch.Generator(device=dev).manual_seed(1))
    assert smp.shape == (2, P + 4) and int(smp.max()) < g["cfg"]["vocab_size"]
    print(f"generate vs reference: {compared}/{2 * new} tokens compared, all equal")


@pytest.mark.parametrize("kind", ["linear", "dynamic"])
def test_rope_scaling_arguments_match_reference(dev, kind):
    """UniGen(scaling_factor=2.0, rope_type=...) (reference models/unigen.py:38-40,61-64 -> config.rope_scaling) against the REAL
    reference's outputs (golden G14; max_position_embeddings = 16 < L = 40 so 'dynamic' takes its NTK branch): the three losses
    <= 1e-3, logits as close to the reference's fp32 logits as its own bf16 path."""
    from models import UniGen
    from oracle import weights
    g, g2 = golden("g14_rope_scaling.pt"), golden("g2_tiny_unigen.pt")
    cfg, ids = dict(g["cfg"], max_position_embeddings=g["max_position_embeddings"]), g["ids"]
    m = UniGen(w_und_encoder=False, vocab_size=cfg["vocab_size"], llm_vocab_size=ids["text_vocab"], llm_model_path=llm_config_dir(cfg),
               codebook_size=20, num_vq_tokens=16, load_from_pretrained=True, scaling_factor=g["factor"], rope_type=kind, device=dev,
               init_seed=1).train()
    names = [(n, tuple(p.shape)) for n, p in m.llm.named_parameters()]
    m.llm.load_state_dict(weights.synth_llm_state(names, seed=g["weight_seed"]), strict=False)
    logits, l1, l2, l3 = m(input_ids=g2["input_ids"].to(dev), attention_mask=additive(g2["mask_allow"]).to(dev), labels=g2["labels"].to(dev),
                           **g2["kw"])
    m.llm.engine.check_errors()
    got = torch.stack([l1, l2, l3]).float().cpu()
    want = g[kind]["bf16"]
    rel = ((got - want["losses"]).abs() / want["losses"]).max().item()
    print(f"    [rope_type={kind}] losses rel {rel:.2e} (gate 1e-3)")
    assert rel < 1e-3
    fp32_yardstick(f"G14 rope_type={kind}", logits.materialize().float().cpu(), want["logits"], g[kind]["fp32"]["logits"])
    # and the unscaled model is measurably somewhere else (the arguments are not ignored)
    dense = logits.materialize().float().cpu()
    d_scaled = ((dense - want["logits"].float()).norm() / want["logits"].float().norm()).item()
    d_plain = ((dense - g2["bf16"]["logits"].float()).norm() / g2["bf16"]["logits"].float().norm()).item()
    print(f"    [rope_type={kind}] distance to the scaled reference {d_scaled:.2e}, to the unscaled one {d_plain:.2e}")
    assert d_scaled < 0.7 * d_plain
