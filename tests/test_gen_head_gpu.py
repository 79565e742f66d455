"""SURVEY.md section 8 row a18 on the GPU: UniGen built with gen_proj_depth = 2 (gen_embed -> gen_projector in, img_head out,
reference models/unigen.py:74-92, 255-270, 301-311) against the REAL reference's outputs (golden G12) -- img_logits, loss,
every gradient -- for both use_gen_dim settings; prepare_inputs_for_t2i / get_gen_embed; MaskGIT generation on this path
against the oracle for the first round."""
import math

import pytest
import torch

from helpers import additive, fp32_yardstick, golden, llm_config_dir

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _model(g, dev, use_dim):
    from models import UniGen
    from oracle import weights
    cfg, ids = g["cfg"], g["ids"]
    m = UniGen(w_und_encoder=False, vocab_size=cfg["vocab_size"], llm_vocab_size=ids["text_vocab"], llm_model_path=llm_config_dir(cfg),
               codebook_size=g["codebook"], num_vq_tokens=g["n"], load_from_pretrained=True, gen_proj_depth=2, use_gen_dim=use_dim,
               gen_input_dim=16, device=dev, init_seed=1)
    names = [(k, tuple(p.shape)) for k, p in m.named_parameters() if k != "_ddp_anchor"]
    sd = weights.synth_llm_state(names, seed=g["weight_seed"], std=0.05)
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and m.config.mask_token_id == g["codebook"]
    return m, sd


@pytest.mark.parametrize("use_dim", [False, True])
def test_gen_projector_forward_backward_matches_reference(dev, use_dim):
    g = golden("g12_gen_head.pt")
    want = g[f"dim{int(use_dim)}"]
    model, sd = _model(g, dev, use_dim)
    model.train()
    assert sorted(k for k, _ in model.named_parameters() if not k.startswith("llm.") and k != "_ddp_anchor") == sorted(want["grad_norms_gen"])
    mask = additive(g["mask_allow"]).to(dev)
    ids, labels = g["input_ids"].to(dev), g["labels"].to(dev)
    img_logits, l1, l2, l3 = model(input_ids=ids, attention_mask=mask, labels=labels, **g["kw"])
    model.llm.engine.check_errors()
    assert l2 == 0. and l3 == 0. and img_logits.shape == want["img_logits"].shape
    lerr = abs(l1.item() - want["loss"].item()) / want["loss"].item()
    rl = _rel(img_logits, want["img_logits"])
    l1.backward()
    params = dict(model.named_parameters())
    worst = 0.0
    for k, v in list(want["grad_norms_gen"].items()) + list(want["grad_norms_llm"].items()):
        gn = params[k].grad.norm().item()
        worst = max(worst, abs(gn - v) / max(v, 1e-8))
    rows = max(_rel(params[k].grad if v.shape == params[k].grad.shape else params[k].grad[:4], v) for k, v in want["grads_gen_head"].items())
    print(f"[gen_proj use_gen_dim={use_dim}] loss rel {lerr:.2e} (gate 1e-3); img_logits rel {rl:.2e}; grad-norm worst rel {worst:.2e}; "
          f"gen-module gradients rel {rows:.2e}")
    assert lerr < 1e-3 and rl < 1e-2 and worst < 2e-2 and rows < 3e-2
    # fp32 logits of the same weights from the oracle's gen-branch forward (bit-identical to the reference, tools/make_golden.py)
    from oracle import qwen2_ref
    lm = qwen2_ref.RefCausalLM(qwen2_ref.Qwen2Cfg(**g["cfg"]))
    lm.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("llm.")}, strict=False)
    gen = qwen2_ref.GenHeadRef(g["cfg"]["hidden_size"], g["codebook"], depth=2, use_gen_dim=use_dim, gen_input_dim=16)
    gen.load_state_dict({k: v for k, v in sd.items() if not k.startswith("llm.") and k != "_ddp_anchor"})
    with torch.no_grad():
        lo32 = qwen2_ref.unigen_forward_gen_ref(lm, gen, g["input_ids"], additive(g["mask_allow"]), None,
                                                batch_size_t2i=g["kw"]["batch_size_t2i"], num_vq_tokens=g["n"], autocast=False)
    fp32_yardstick(f"G12 img_logits use_gen_dim={use_dim}", img_logits, want["img_logits"], lo32)
    # labels=None returns the img_head logits of the t2i rows (reference :264-265)
    only = model(input_ids=ids, attention_mask=mask, batch_size_t2i=g["kw"]["batch_size_t2i"], num_vq_tokens=g["n"])
    assert torch.equal(only, img_logits)
    # prepare_inputs_for_t2i: token embeddings with the image slots replaced by the projected gen embeddings
    n = g["n"]
    e = model.prepare_inputs_for_t2i(ids, n)
    assert torch.equal(e[:, :-(n + 1)], model.llm.model.embed_tokens(ids)[:, :-(n + 1)])
    assert _rel(e[:, -(n + 1):-1], model.get_gen_embed(ids[:, -(n + 1):-1].contiguous())) < 1e-6


def test_gen_projector_maskgit_round_matches_oracle(dev):
    """t2i_generate on the gen_projector path (raw codes, img_head logits, no text-vocabulary offset): first round against
    a restatement built from the oracle's gen-branch forward with the same uniforms; a 3-round run keeps the contract."""
    from oracle import qwen2_ref
    g = golden("g12_gen_head.pt")
    model, sd = _model(g, dev, False)
    model.eval()
    cfg = qwen2_ref.Qwen2Cfg(**g["cfg"])
    lm = qwen2_ref.RefCausalLM(cfg)
    lm.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("llm.")}, strict=False)
    gen = qwen2_ref.GenHeadRef(g["cfg"]["hidden_size"], g["codebook"], depth=2)
    gen.load_state_dict({k: v for k, v in sd.items() if not k.startswith("llm.") and k != "_ddp_anchor"})
    n, CB = g["n"], g["codebook"]
    ids = g["input_ids"].clone()
    ids[:, -(n + 1):-1] = CB                                     # everything masked
    am = additive(g["mask_allow"])
    N = ids.shape[0]
    gen_dev = torch.Generator(device=dev).manual_seed(5)
    state = gen_dev.get_state()
    u = torch.rand((2, N, n), device=dev, generator=gen_dev).cpu()
    gen_dev.set_state(state)
    sched = lambda t: torch.cos(t * math.pi * 0.5)
    got = model.t2i_generate(input_ids=ids.to(dev), attention_mask=am.to(dev), guidance_scale=0, temperature=1.0, timesteps=1,
                             noise_schedule=sched, generator=gen_dev, image_token_num_per_image=n, text_vocab_size=g["ids"]["text_vocab"]).cpu()
    with torch.no_grad():
        lo = qwen2_ref.unigen_forward_gen_ref(lm, gen, ids, am, None, batch_size_t2i=N, num_vq_tokens=n, autocast=True)
    probs = lo[:, -(n + 1):-1].float().softmax(-1)
    cdf = probs.cumsum(-1)
    want = (cdf <= u[0].reshape(N, n, 1) * cdf[..., -1:]).sum(-1).clamp(max=CB - 1)
    sure = (cdf - u[0].reshape(N, n, 1)).abs().min(-1).values > 0.004
    assert sure.float().mean() > 0.6 and torch.equal(got[sure], want[sure])
    full = model.t2i_generate(input_ids=ids.to(dev), attention_mask=am.to(dev), guidance_scale=0, temperature=1.0, timesteps=3,
                              noise_schedule=sched, generator=torch.Generator().manual_seed(1), image_token_num_per_image=n,
                              text_vocab_size=g["ids"]["text_vocab"], incremental=False).cpu()
    assert full.shape == (N, n) and int(full.min()) >= 0 and int(full.max()) < CB


@pytest.mark.parametrize("use_dim", [False, True])
def test_ar_generation_gen_head_matches_reference_trajectory(dev, use_dim):
    """G13: t2i_generate_ar on the gen_projector path (img_head on the last hidden state, gen_projector(gen_embed(token)) as the
    next input, reference models/unigen.py:486-495,512-514) against the tokens of the REAL reference (bf16 autocast, CFG 2.5,
    left-padded prompts, 2-D mask), step by step until the reference's own top-2 margin drops below bf16 noise; eager and with
    the captured decode graph."""
    from models import UniGen
    from oracle import weights
    g = golden("g13_ar_gen_head.pt")
    cfg, ids = g["cfg"], g["ids"]
    m = UniGen(w_und_encoder=False, vocab_size=cfg["vocab_size"], llm_vocab_size=ids["text_vocab"], llm_model_path=llm_config_dir(cfg),
               codebook_size=g["codebook"], num_vq_tokens=g["n"], load_from_pretrained=True, gen_proj_depth=2, use_gen_dim=use_dim,
               gen_input_dim=16, device=dev, init_seed=1)
    names = [(k, tuple(p.shape)) for k, p in m.named_parameters() if k != "_ddp_anchor"]
    res = m.load_state_dict(weights.synth_llm_state(names, seed=g["weight_seed"], std=g["weight_std"]), strict=False)
    assert not res.unexpected_keys
    m.eval()
    want, margin = g[f"dim{int(use_dim)}"]["bf16"]["tokens"], g[f"dim{int(use_dim)}"]["bf16"]["margin"]
    for use_graph in (False, True):
        got = m.t2i_generate_ar(input_ids=g["cond"].to(dev), uncond_input_ids=g["uncond"].to(dev), attention_mask=g["attention_mask"].to(dev),
                                guidance_scale=g["scale"], temperature=1.0, text_vocab_size=ids["text_vocab"],
                                image_token_num_per_image=g["n"], greedy=True, use_graph=use_graph).cpu()
        assert got.shape == want.shape and int(got.min()) >= 0 and int(got.max()) < g["codebook"]
        compared = 0
        for b in range(want.shape[0]):
            for i in range(want.shape[1]):
                if margin[b, i] < 0.3:      # the mix is bf16 arithmetic here: margins come in steps of one bf16 ulp (0.06 .. 0.13)
                    break
                assert int(got[b, i]) == int(want[b, i]), (use_graph, b, i, got[b].tolist(), want[b].tolist())
                compared += 1
        print(f"AR on the gen_projector path vs reference golden (use_gen_dim={use_dim}, graph={use_graph}): {compared}/{want.numel()} "
              f"tokens compared, all equal")
        assert compared >= 8, compared
