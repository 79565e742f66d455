"""Model-level parity on the GPU: the drop-in `models.UniGen` / `models.MAGVITv2` (HIP kernels through
the C ABI) against (a) the golden vectors captured from the real reference and (b) the CPU oracle on
the same seeded inputs.  Tolerances: bit-exact for integer outputs (token indices, argmax decode),
<= 1e-3 relative for bf16 logits / losses (BASELINE.json north_star), stated per assert."""
import math
import os

import pytest
import torch

from helpers import additive, check_logits, fp32_yardstick, golden, llm_config_dir, oracle_lm

pytestmark = pytest.mark.gpu


def _check(tag, a, b, tol):
    """relative Frobenius error of a vs b, printed with its gate (run pytest -s to see the margins)"""
    e = _rel(a, b)
    print(f"    {tag}: rel {e:.2e} (gate {tol:.0e})")
    assert e < tol, (tag, e, tol)
    return e


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _tiny_unigen(g, dev):
    from models import UniGen
    from oracle import weights
    cfg = g["cfg"]
    d = llm_config_dir(cfg)
    ids = g["ids"]
    m = UniGen(w_und_encoder=False, vocab_size=cfg["vocab_size"], llm_vocab_size=ids["text_vocab"], llm_model_path=d,
               codebook_size=20, num_vq_tokens=16, load_from_pretrained=True, device=dev, init_seed=1)
    names = [(n, tuple(p.shape)) for n, p in m.llm.named_parameters()]
    sd = weights.synth_llm_state(names, seed=g["weight_seed"])
    res = m.llm.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    return m, sd


def test_tiny_unigen_step_matches_reference_golden(dev):
    g = golden("g2_tiny_unigen.pt")
    model, sd = _tiny_unigen(g, dev)
    model.train()
    want = g["bf16"]
    mask = additive(g["mask_allow"]).to(dev)
    ids, labels = g["input_ids"].to(dev), g["labels"].to(dev)
    # parameter inventory == the reference checkpoint's (names and shapes)
    assert sorted(n for n, _ in model.llm.named_parameters()) == sorted(want["grad_norms"].keys())
    logits, l1, l2, l3 = model(input_ids=ids, attention_mask=mask, labels=labels, **g["kw"])
    model.llm.engine.check_errors()
    got = torch.stack([l1, l2, l3]).float().cpu()
    assert ((got - want["losses"]).abs() / want["losses"]).max().item() < 1e-3, (got, want["losses"])
    dense = logits.materialize().float().cpu()
    assert dense.shape == want["logits"].shape
    check_logits("G2 dense", dense, want["logits"])
    sl = logits[:2, -17:-1, 312:-1].float().cpu()
    check_logits("G2 lazy slice", sl, want["logits"][:2, -17:-1, 312:-1])
    fp32_yardstick("G2 dense logits", dense, want["logits"], g["fp32"]["logits"])
    loss = 1.0 * l1 + 0.1 * l2 + 1.0 * l3
    loss.backward()
    params = dict(model.llm.named_parameters())
    worst = 0.0
    for n, v in want["grad_norms"].items():
        gn = params[n].grad.norm().item()
        worst = max(worst, abs(gn - v) / max(v, 1e-8))
    assert worst < 2e-2, worst
    for n, gg in want["grads_small"].items():
        _check("params[n].grad", params[n].grad, gg, 3e-2)
    _check("params[\"model.embed_tokens.weight\"].grad[[0", params["model.embed_tokens.weight"].grad[[0, 5, 300, 303, 304, 312, 320, 332]], want["grad_embed_rows"], 3e-2)
    _check("params[\"model.layers.0.self_attn.q_proj.weight\"].grad[:4]", params["model.layers.0.self_attn.q_proj.weight"].grad[:4], want["grad_q0_rows"], 3e-2)
    _check("params[\"model.layers.1.mlp.down_proj.weight\"].grad[:4]", params["model.layers.1.mlp.down_proj.weight"].grad[:4], want["grad_down1_rows"], 3e-2)
    from unigen_hip.optim import FusedAdamW
    decay = [p for n, p in model.named_parameters() if "bias" not in n]
    nodecay = [p for n, p in model.named_parameters() if "bias" in n]
    opt = FusedAdamW([{"params": decay, "weight_decay": 0.01}, {"params": nodecay, "weight_decay": 0.0}], lr=1e-3,
                     betas=(0.9, 0.999), eps=1e-8)
    opt.step()
    # the optimizer pass also refreshed the bf16 compute mirror of the flat master weights (no cast pass follows)
    fp = model.llm.engine.fp
    assert fp._seen_version == fp.master._version
    assert torch.equal(fp.bf16, fp.master.to(torch.bfloat16))
    a = want["adamw"]
    # first Adam step moves every weight by ~lr*sign(g): compare the update, not the weight
    q0 = params["model.layers.0.self_attn.q_proj.weight"][:4].detach().cpu()
    upd, upd_ref = q0 - sd["model.layers.0.self_attn.q_proj.weight"][:4], a["q0_rows"] - sd["model.layers.0.self_attn.q_proj.weight"][:4]
    assert (upd.sign() == upd_ref.sign()).float().mean().item() > 0.97
    assert (params["model.norm.weight"].detach().cpu() - a["norm"]).abs().max().item() < 2e-4
    opt.zero_grad(set_to_none=True)
    assert params["model.norm.weight"].grad is None
    # second backward after zero_grad(set_to_none=True): flat grads are cleared and re-attached
    _, l1b, l2b, l3b = model(input_ids=ids, attention_mask=mask, labels=labels, **g["kw"])
    (l1b + l3b).backward()
    assert params["model.norm.weight"].grad is not None and torch.isfinite(params["model.norm.weight"].grad).all()
    assert l1b.item() < l1.item()           # the step reduced the loss on the same batch


def test_tiny_unigen_vs_cpu_oracle_fresh_batch(dev):
    """Same comparison on inputs that are NOT in the fixture: t2i rows only, longer sequence (3 kv tiles),
    random left padding, causal-text + bidirectional-image mask from the oracle's builder."""
    from oracle import host_ref, qwen2_ref
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    ids = g["ids"]
    gen = torch.Generator().manual_seed(77)
    B, L, n = 3, 150, 16
    seq = torch.randint(0, 290, (B, L), generator=gen)
    for b, npad in enumerate((0, 17, 70)):
        seq[b, :npad] = ids["pad"]
    seq[:, -(n + 2)] = ids["soi"]
    seq[:, -1] = ids["eoi"]
    img = torch.randint(312, 332, (B, n), generator=gen)
    labels = torch.full((B, L), -100)
    m = torch.rand(B, n, generator=gen) < 0.6
    m[:, 0] = True
    seq[:, -(n + 1):-1] = torch.where(m, ids["mask"], img)
    labels[:, -(n + 1):-1] = torch.where(m, img, -100)
    allow = host_ref.mask_predict_next_ref(seq, ids["pad"], ids["soi"], ids["eoi"], rm_pad_in_image=True)
    mask = additive(allow)
    lo, r1, _, _ = qwen2_ref.unigen_forward_ref(lm, seq, mask, labels, batch_size_t2i=B, num_vq_tokens=n, autocast=True)
    r1.backward()
    logits, l1, l2, l3 = model(input_ids=seq.to(dev), attention_mask=mask.to(dev), labels=labels.to(dev),
                               batch_size_t2i=B, num_vq_tokens=n)
    assert l2 == 0. and l3 == 0.
    assert abs(l1.item() - r1.item()) / r1.item() < 1e-3
    check_logits("tiny fresh batch", logits[:, -(n + 1):-1].float(), lo[:, -(n + 1):-1])
    with torch.no_grad():
        lo32 = qwen2_ref.unigen_forward_ref(lm, seq, mask, None, autocast=False)
    fp32_yardstick("tiny fresh batch", logits[:, -(n + 1):-1].float(), lo[:, -(n + 1):-1], lo32[:, -(n + 1):-1])
    top2 = lo[:, -(n + 1):-1].topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 0.05
    am = logits[:, -(n + 1):-1].float().argmax(-1).cpu()
    assert torch.equal(am[clear], lo[:, -(n + 1):-1].argmax(-1)[clear])
    l1.backward()
    ref_g = dict(lm.named_parameters())
    for n_, p in model.llm.named_parameters():
        _check("p.grad", p.grad, ref_g[n_].grad, 4e-2)


def test_magvit_tokens_match_reference_golden(dev):
    from models import MAGVITv2
    from oracle import magvit_ref, weights
    g = golden("g1_magvit.pt")
    vq = MAGVITv2().to(dev).eval()
    shapes = [(n, tuple(p.shape)) for n, p in vq.named_parameters()]
    assert sorted(shapes) == sorted(magvit_ref.magvit_param_shapes())
    sd = weights.synth_magvit_state(shapes, seed=g["weight_seed"])
    res = vq.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all(k.startswith("quantize.") for k in res.missing_keys)
    x = weights.synth_images(2, 256, seed=g["image_seed"]).to(dev)
    z = vq.get_latents(x).cpu()
    idx = vq.get_code(x).cpu()
    zerr = (z - g["z"]).abs().max().item()
    assert zerr < 1e-4, zerr                                  # fp32 fma-chain vs CPU fp32: summation-order noise only
    eps = 4 * zerr + 1e-6
    safe = g["z"].abs() > eps
    assert torch.equal((z > 0)[safe], (g["z"] > 0)[safe])     # every sign bit outside the noise band is exact
    n_diff = (idx != g["indices"]).sum().item()
    n_band = (~safe).permute(0, 2, 3, 1).reshape(2, 256, 13).any(-1).sum().item()
    assert n_diff <= n_band, (n_diff, n_band)
    print(f"magvit: max|dz|={zerr:.2e}, tokens differing {n_diff}/512 (inside the +-{eps:.1e} band: {n_band})")
    assert idx.dtype == torch.int64 and idx.shape == (2, 256)
    zq, idx2 = vq.encode(x)
    assert torch.equal(idx2.cpu(), idx) and set(zq.unique().tolist()) <= {-1.0, 1.0}
    rec = vq.decode_code(g["indices"].to(dev)).cpu()
    assert rec.shape == (2, 3, 256, 256)
    assert (rec[:, :, 96:160, 96:160] - g["rec_crop"]).abs().max().item() < 2e-4
    assert (rec.mean(dim=(2, 3)) - g["rec_mean"]).abs().max().item() < 1e-5


def test_magvit_split_convs_match_exact_fp32_path(dev):
    """The default tokenizer (split-bf16 convolutions, GroupNorm on the conv load path) against the same module with
    every conv on the exact fp32 MFMA chain: latents agree to fp32 summation noise, tokens outside that band are equal,
    reconstructions agree."""
    from models import MAGVITv2
    from oracle import weights
    vq = MAGVITv2().to(dev).eval()
    sd = weights.synth_magvit_state([(n, tuple(p.shape)) for n, p in vq.named_parameters()], seed=77)
    vq.load_state_dict(sd, strict=False)
    x = weights.synth_images(3, 256, seed=9).to(dev)
    assert not vq.exact_fp32_convs
    z_split, idx_split = vq.get_latents(x).cpu(), vq.get_code(x).cpu()
    rec_split = vq.decode_code(idx_split.to(dev)).cpu()
    vq.exact_fp32_convs = True
    z_exact, idx_exact = vq.get_latents(x).cpu(), vq.get_code(x).cpu()
    rec_exact = vq.decode_code(idx_split.to(dev)).cpu()
    dz = (z_split - z_exact).abs().max().item()
    assert dz < 5e-5, dz
    safe = (z_exact.abs() > 4 * dz + 1e-6).permute(0, 2, 3, 1).reshape(3, 256, 13).all(-1)
    assert torch.equal(idx_split[safe], idx_exact[safe]) and safe.float().mean().item() > 0.95
    assert (rec_split - rec_exact).abs().max().item() < 2e-4


def test_ar_generation_kv_cache_matches_oracle(dev):
    """t2i_generate_ar (static KV cache, captured graph) vs the oracle's DynamicCache-style greedy decode:
    tokens must agree step by step until the oracle's own top-2 margin drops below bf16 noise."""
    from oracle import qwen2_ref
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    model.eval()
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    ids = g["ids"]
    n, B, P = 16, 2, 30
    gen = torch.Generator().manual_seed(5)
    cond = torch.randint(0, 290, (B, P + n + 1), generator=gen)
    uncond = torch.randint(0, 290, (B, P + n + 1), generator=gen)
    cond[0, :6] = ids["pad"]; uncond[:, :20] = ids["pad"]                     # left padding like t2i_gen_prompt
    am = torch.cat([cond != ids["pad"], uncond != ids["pad"]]).long()
    am[:, P:] = 1
    with torch.no_grad():
        ce, ue = lm.model.embed_tokens(cond[:, :P]), lm.model.embed_tokens(uncond[:, :P])
    want, margin = qwen2_ref.ar_generate_ref(lm, ce, ue, n, 3.0, ids["text_vocab"], key_valid=am[:, :P])
    for use_graph, fused in ((False, True), (True, True), (True, False)):
        model.llm.engine.decode_fused = fused        # one-launch projections vs the separate-kernel path
        got = model.t2i_generate_ar(input_ids=cond.to(dev), uncond_input_ids=uncond.to(dev), attention_mask=am.to(dev),
                                    guidance_scale=3.0, temperature=1.0, text_vocab_size=ids["text_vocab"],
                                    image_token_num_per_image=n, greedy=True, use_graph=use_graph).cpu()
        assert got.shape == (B, n) and got.dtype == torch.int32
        assert model.llm.engine.last_decode_graph == use_graph
        compared = 0
        for b in range(B):
            for i in range(n):
                if margin[b, i] < 0.05:
                    break                      # an oracle near-tie: later tokens may legitimately diverge
                assert got[b, i].item() == want[b, i].item(), (use_graph, fused, b, i, got[b], want[b])
                compared += 1
        assert compared >= 8, compared


def test_maskgit_generation_first_round_matches_oracle(dev):
    """UniGen.t2i_generate (fused sampler step) vs oracle.maskgit_generate_ref fed the same uniforms: one round must
    agree wherever the oracle's draw is not within bf16 noise of a CDF step; a full 4-round run keeps the contract
    (every position decoded, ids inside the code book)."""
    import math
    from oracle import qwen2_ref
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    model.eval()
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    m, ids = g["maskgit"], g["ids"]
    n, N = 16, m["input_ids"].shape[0]
    sched = lambda t: torch.cos(t * math.pi * 0.5)
    am = additive(m["mask_allow"])
    gen = torch.Generator(device=dev).manual_seed(77)
    state = gen.get_state()
    u = torch.rand((2, N, n), device=dev, generator=gen).cpu()
    gen.set_state(state)
    got = model.t2i_generate(input_ids=m["input_ids"].to(dev), uncond_input_ids=m["uncond_ids"].to(dev), attention_mask=am.to(dev),
                             guidance_scale=m["scale"], temperature=1.0, timesteps=1, noise_schedule=sched, generator=gen,
                             image_token_num_per_image=n, text_vocab_size=ids["text_vocab"]).cpu()
    trace = []
    want = qwen2_ref.maskgit_generate_ref(lm, m["input_ids"], m["uncond_ids"], am, m["scale"], 1.0, 1, sched, n, ids["text_vocab"],
                                          ids["mask"], qwen2_ref.InverseCdfSampler(u[0:1], u[1:2]), autocast=True, trace=trace)
    probs = trace[0]["mixed"].softmax(-1)
    cdf = probs.cumsum(-1)
    gap = (cdf - u[0].reshape(N, n, 1)).abs().min(-1).values
    sure = gap > 0.004
    assert sure.float().mean() > 0.6 and torch.equal(got[sure], want[sure])
    full = model.t2i_generate(input_ids=m["input_ids"].to(dev), uncond_input_ids=m["uncond_ids"].to(dev), attention_mask=am.to(dev),
                              guidance_scale=m["scale"], temperature=1.0, timesteps=4, noise_schedule=sched,
                              generator=torch.Generator().manual_seed(3), image_token_num_per_image=n,
                              text_vocab_size=ids["text_vocab"]).cpu()
    assert full.shape == (N, n) and int(full.min()) >= 0 and int(full.max()) < 20


@pytest.mark.parametrize("fixture", ["g3_wide_layer.pt", "g16_wide_layer_L771.pt"])
def test_wide_layer_matches_reference_golden(dev, fixture):
    """G3 / G16: the real reference's forward + backward through one decoder layer of the 1.5B model's width at L = 387 and (round 5)
    at the benchmarked L = 771, both rows left-padded (290 / 45 pads at L = 771), MaskGIT labels, vs the HIP path: loss,
    image-position logits, every parameter's gradient."""
    from models import UniGen
    from oracle import weights
    g = golden(fixture)
    cfg, ids = g["cfg"], g["ids"]
    m = UniGen(w_und_encoder=False, vocab_size=cfg["vocab_size"], llm_vocab_size=ids["text_vocab"], llm_model_path=llm_config_dir(cfg),
               codebook_size=g["codebook"], num_vq_tokens=256, load_from_pretrained=True, device=dev, init_seed=1)
    names = [(n, tuple(p.shape)) for n, p in m.llm.named_parameters()]
    m.llm.load_state_dict(weights.synth_llm_state(names, seed=g["weight_seed"]), strict=False)
    m.train()
    logits, l1, _, _ = m(input_ids=g["input_ids"].to(dev), attention_mask=additive(g["mask_allow"]).to(dev), labels=g["labels"].to(dev),
                         **g["kw"])
    assert abs(l1.item() - g["loss"].item()) < 1e-3 * g["loss"].item(), (l1.item(), g["loss"].item())
    got = logits[:, -257:-1, :].float().cpu()[:, ::8]
    check_logits(f"{fixture} wide layer", got, g["logits_rows"])
    # the fixture holds the real reference's bf16 logits; its fp32 logits come from the oracle (bit-identical to the reference
    # on CPU, tools/make_golden.py) with the same seeded weights
    from oracle import qwen2_ref
    lm, _ = oracle_lm(cfg, g["weight_seed"])
    with torch.no_grad():
        lo32 = qwen2_ref.unigen_forward_ref(lm, g["input_ids"], additive(g["mask_allow"]), None, autocast=False)[:, -257:-1][:, ::8]
    fp32_yardstick(f"{fixture} wide layer", got, g["logits_rows"], lo32)
    l1.backward()
    params = dict(m.llm.named_parameters())
    for n, v in g["grad_norms"].items():
        gn = params[n].grad.norm().item()
        assert abs(gn - v) <= 2e-2 * max(v, 1e-8), (n, gn, v)
    _check("params[\"model.layers.0.self_attn.o_proj.weight\"].grad[:2]", params["model.layers.0.self_attn.o_proj.weight"].grad[:2], g["grad_o_rows"], 3e-2)
    _check("params[\"model.layers.0.mlp.gate_proj.weight\"].grad[:2]", params["model.layers.0.mlp.gate_proj.weight"].grad[:2], g["grad_gate_rows"], 3e-2)


def test_loss_curve_20_steps_matches_oracle(dev):
    """SURVEY.md section 8d parity gate: the same 20 optimizer steps (fresh synthetic t2i batch each step, fused AdamW vs
    torch.optim.AdamW on the CPU oracle, bf16 autocast) give the same loss curve within 1e-3 relative."""
    from oracle import host_ref, qwen2_ref
    from unigen_hip.optim import FusedAdamW
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    model.train()
    lm, _ = oracle_lm(g["cfg"], g["weight_seed"])
    ids = g["ids"]
    B, L, n = 4, 40, 16
    gen = torch.Generator().manual_seed(123)

    def groups(named):
        named = list(named)
        return [{"params": [p for k, p in named if "bias" not in k], "weight_decay": 0.01},
                {"params": [p for k, p in named if "bias" in k], "weight_decay": 0.0}]
    opt = FusedAdamW(groups(model.named_parameters()), lr=3e-4, betas=(0.9, 0.999), eps=1e-8)
    opt_ref = torch.optim.AdamW(groups(lm.named_parameters()), lr=3e-4, betas=(0.9, 0.999), eps=1e-8)
    got, want = [], []
    for step in range(20):
        seq = torch.randint(0, 290, (B, L), generator=gen)
        seq[:, -(n + 2)] = ids["soi"]; seq[:, -1] = ids["eoi"]
        seq[0, :step % 7] = ids["pad"]
        img = torch.randint(312, 332, (B, n), generator=gen)
        msk = torch.rand(B, n, generator=gen) < 0.6
        msk[:, 0] = True
        seq[:, -(n + 1):-1] = torch.where(msk, ids["mask"], img)
        labels = torch.full((B, L), -100)
        labels[:, -(n + 1):-1] = torch.where(msk, img, -100)
        mask = additive(host_ref.mask_predict_next_ref(seq, ids["pad"], ids["soi"], ids["eoi"], rm_pad_in_image=True))
        _, r1, _, _ = qwen2_ref.unigen_forward_ref(lm, seq, mask, labels, autocast=True, batch_size_t2i=B, num_vq_tokens=n)
        opt_ref.zero_grad(set_to_none=True)
        r1.backward()
        opt_ref.step()
        _, l1, _, _ = model(input_ids=seq.to(dev), attention_mask=mask.to(dev), labels=labels.to(dev), batch_size_t2i=B,
                            max_seq_length=L - n - 3, num_vq_tokens=n)
        opt.zero_grad(set_to_none=True)
        l1.backward()
        opt.step()
        got.append(l1.item()); want.append(r1.item())
    got, want = torch.tensor(got), torch.tensor(want)
    assert ((got - want).abs() / want).max().item() < 1e-3, (got, want)
    assert want[-5:].mean() < want[:5].mean()          # and it is actually training


def test_mmu_generate_kv_cache_matches_recompute(dev):
    """mmu_generate with the static KV cache (one prefill under the prompt's mmu mask + one decode step per token) returns
    the tokens the reference procedure (whole sequence re-run per token, mask grown by one row) returns, greedy."""
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    model.eval()
    P = 30
    idx = g["input_ids"][-1:, :P].to(dev)                       # the mmu row of the fixture: image tokens, then text
    mask = additive(g["mask_allow"][-1:, :P, :P]).to(torch.float32).to(dev)
    slow = model.mmu_generate(idx=idx, attention_mask=mask, max_new_tokens=12, temperature=0.0, use_cache=False)
    fast = model.mmu_generate(idx=idx, attention_mask=mask, max_new_tokens=12, temperature=0.0, use_cache=True)
    slow, fast = [int(t) for t in slow], [int(t) for t in fast]
    assert len(fast) == 12 and fast[:8] == slow[:8], (fast, slow)
    # sampling path and early stop on the end-of-turn token
    out = model.mmu_generate(idx=idx, attention_mask=mask, max_new_tokens=6, temperature=1.0, top_k=5, eot_token=fast[2])
    assert 1 <= len(out) <= 6
    stop = model.mmu_generate(idx=idx, attention_mask=mask, max_new_tokens=12, temperature=0.0, eot_token=fast[2])
    assert [int(t) for t in stop] == fast[:fast.index(fast[2]) + 1]


def test_mmu_generate_batch_matches_per_row_calls(dev):
    """mmu_generate_batch (left-padded rows decoded together, the CoT-V rating loop batched) returns, row by row, what
    mmu_generate returns for that padded row alone (greedy), and cuts each row at its own end-of-turn token."""
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    model.eval()
    pad = g["ids"]["pad"]
    L, lens = 30, [30, 22, 26]
    rows, masks = [], []
    gen = torch.Generator().manual_seed(4)
    for n in lens:
        ids = torch.full((L,), pad, dtype=torch.long)
        ids[L - n:] = torch.randint(0, 290, (n,), generator=gen)
        allow = torch.tril(torch.ones(L, L, dtype=torch.bool))
        allow[:, :L - n] = False                                  # left padding: pad columns blocked for every query
        allow[torch.arange(L - n), torch.arange(L - n)] = True    # (a pad row still attends to itself: no empty rows)
        rows.append(ids); masks.append(allow)
    idx = torch.stack(rows).to(dev)
    mask = additive(torch.stack(masks)).to(torch.float32).to(dev)
    batch = model.mmu_generate_batch(idx=idx, attention_mask=mask, max_new_tokens=10, temperature=0.0)
    assert len(batch) == 3 and all(len(b) == 10 for b in batch)
    for r in range(3):
        single = model.mmu_generate(idx=idx[r:r + 1], attention_mask=mask[r:r + 1], max_new_tokens=10, temperature=0.0)
        assert [int(t) for t in batch[r]][:8] == [int(t) for t in single][:8], (r, batch[r], single)
    eot = int(batch[1][3])
    cut = model.mmu_generate_batch(idx=idx, attention_mask=mask, max_new_tokens=10, temperature=0.0, eot_token=eot)
    for r in range(3):
        full = [int(t) for t in batch[r]]
        want = full[:full.index(eot) + 1] if eot in full else full
        assert [int(t) for t in cut[r]] == want[:len(cut[r])] and (eot not in want or len(cut[r]) == len(want))


def test_checkpoint_round_trip_and_hf_llm_loading(dev, tmp_path):
    """save_pretrained -> from_pretrained keeps every tensor (reference key names, tied head written once per name) and
    the logits; a HF-layout Qwen2 directory (config.json + *.safetensors) loads through llm_model_path with the
    embedding resized to the UniGen vocabulary (models/unigen.py:58-69)."""
    from models import UniGen
    from safetensors.torch import save_file
    g = golden("g2_tiny_unigen.pt")
    model, sd = _tiny_unigen(g, dev)
    model.eval()
    ids, mask = g["input_ids"].to(dev), additive(g["mask_allow"]).to(dev)
    ref = model(input_ids=ids, attention_mask=mask)[:, -5:, :].float().cpu()
    model.save_pretrained(str(tmp_path / "ckpt"))
    again, info = UniGen.from_pretrained(str(tmp_path / "ckpt"), device=dev, output_loading_info=True)
    assert not info["unexpected_keys"] and not info["missing_keys"], info
    a, b = model.state_dict(), again.state_dict()
    assert sorted(a) == sorted(b) and all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a)
    assert torch.equal(again(input_ids=ids, attention_mask=mask)[:, -5:, :].float().cpu(), ref)
    # HF Qwen2 directory with a SMALLER vocabulary than UniGen's (text tokens only)
    cfg = dict(g["cfg"]); text_v = g["ids"]["text_vocab"]
    hf_dir = llm_config_dir(dict(cfg, vocab_size=text_v))
    hf_sd = {k: v.clone() for k, v in sd.items() if k != "lm_head.weight"}
    hf_sd["model.embed_tokens.weight"] = hf_sd["model.embed_tokens.weight"][:text_v].clone()
    save_file(hf_sd, os.path.join(hf_dir, "model.safetensors"))
    m2 = UniGen(w_und_encoder=False, vocab_size=cfg["vocab_size"], llm_vocab_size=text_v, llm_model_path=hf_dir, codebook_size=20,
                num_vq_tokens=16, load_from_pretrained=False, device=dev, init_seed=5)
    own = m2.llm.state_dict()
    assert torch.equal(own["model.embed_tokens.weight"][:text_v].cpu(), sd["model.embed_tokens.weight"][:text_v])
    assert torch.equal(own["model.layers.1.mlp.down_proj.weight"].cpu(), sd["model.layers.1.mlp.down_proj.weight"])
    assert own["model.embed_tokens.weight"].shape[0] == cfg["vocab_size"]


def test_maskgit_incremental_rounds_match_full_recompute(dev):
    """t2i_generate with the prefix keys / values cached across rounds (only the <soi>..<eoi> rows recomputed) returns
    the tokens of the round-by-round full forward, same uniforms."""
    import math
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    model.eval()
    m, ids = g["maskgit"], g["ids"]
    sched = lambda t: torch.cos(t * math.pi * 0.5)
    am = additive(m["mask_allow"]).to(dev)
    outs = []
    for inc in (False, True):
        gen = torch.Generator(device=dev).manual_seed(2024)
        outs.append(model.t2i_generate(input_ids=m["input_ids"].to(dev), uncond_input_ids=m["uncond_ids"].to(dev), attention_mask=am,
                                       guidance_scale=m["scale"], temperature=1.0, timesteps=6, noise_schedule=sched, generator=gen,
                                       image_token_num_per_image=16, text_vocab_size=ids["text_vocab"], incremental=inc).cpu())
    assert torch.equal(outs[0], outs[1]), (outs[0], outs[1])
    # a mask whose prefix rows can see the image segment disables the cache (falls back to full recompute, no error)
    bad = am.clone(); bad[:, :, 0, -3] = 0
    out = model.t2i_generate(input_ids=m["input_ids"].to(dev), uncond_input_ids=m["uncond_ids"].to(dev), attention_mask=bad,
                             guidance_scale=m["scale"], temperature=1.0, timesteps=2, noise_schedule=sched,
                             image_token_num_per_image=16, text_vocab_size=ids["text_vocab"])
    assert out.shape == outs[0].shape


def test_gradient_accumulation_and_fresh_write_semantics(dev):
    """Weight gradients: the first backward after zero_grad(set_to_none=True) OVERWRITES the big matrices (no zero fill),
    a second backward without zero_grad ACCUMULATES, and stale values from an earlier step never leak through."""
    g = golden("g2_tiny_unigen.pt")
    model, _ = _tiny_unigen(g, dev)
    model.train()
    mask = additive(g["mask_allow"]).to(dev)
    ids, labels = g["input_ids"].to(dev), g["labels"].to(dev)

    def run():
        _, l1, l2, l3 = model(input_ids=ids, attention_mask=mask, labels=labels, **g["kw"])
        (l1 + 0.1 * l2 + l3).backward()
    params = dict(model.llm.named_parameters())
    names = ["model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight", "model.layers.1.mlp.gate_proj.weight",
             "model.layers.0.self_attn.o_proj.weight", "model.norm.weight", "model.embed_tokens.weight", "model.layers.0.self_attn.k_proj.bias"]
    run()
    once = {n: params[n].grad.clone() for n in names}
    run()                                                   # accumulate
    for n in names:
        _check("params[n].grad", params[n].grad, 2 * once[n], 2e-3)
    model.zero_grad(set_to_none=True)
    model.llm.engine.fp.grad.fill_(123.0)                   # poison: anything not rewritten or cleared would show
    run()
    for n in names:
        _check("params[n].grad", params[n].grad, once[n], 2e-3)


def test_fused_epilogues_give_the_same_step_as_the_separate_kernels(dev, monkeypatch):
    """Round 4: gate_up + SwiGLU, down dgrad + SwiGLU backward and q/k/v + RoPE run inside GEMM epilogues (ops.gemm_swiglu,
    ops.gemm_swiglu_bwd, ops.gemm_qkv_rope).  The tiny golden model stepped with the fusions on and with the separate kernels
    (ops.FUSED_SWIGLU / FUSED_SWIGLU_BWD off, the projection + ug_rope pair through a pinned-policy-free wrapper): the same losses and
    the same gradients, bit for bit -- the epilogues share their element functions with the kernels they replace."""
    from unigen_hip import ops
    g = golden("g2_tiny_unigen.pt")
    mask = additive(g["mask_allow"]).to(dev)
    ids, labels = g["input_ids"].to(dev), g["labels"].to(dev)
    names = ["model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.k_proj.bias",
             "model.layers.1.mlp.gate_proj.weight", "model.layers.1.mlp.up_proj.weight", "model.layers.1.mlp.down_proj.weight",
             "model.layers.0.post_attention_layernorm.weight", "model.norm.weight"]
    unfused_rope = lambda x, w, b, cos, sin, L, nh, hd: ops.rope_(ops.gemm(x, w, bias=b), cos, sin, L, nh, hd)
    got = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSED_SWIGLU", fused)
        monkeypatch.setattr(ops, "FUSED_SWIGLU_BWD", fused)
        if not fused:
            monkeypatch.setattr(ops, "gemm_qkv_rope", unfused_rope)
        model, _ = _tiny_unigen(g, dev)
        model.train()
        params = dict(model.llm.named_parameters())
        _, l1, l2, l3 = model(input_ids=ids, attention_mask=mask, labels=labels, **g["kw"])
        (l1 + 0.1 * l2 + l3).backward()
        got[fused] = ([float(l1.detach()), float(l2.detach()), float(l3.detach())], {n: params[n].grad.clone() for n in names})
    assert got[True][0] == got[False][0]
    for n in names:
        if not n.endswith("proj.weight"):                        # (scatter-add, bias column sums and norm-weight sums use fp32 atomics: not bit-reproducible run to run)
            _check(n, got[True][1][n], got[False][1][n], 1e-5)
        else:
            assert torch.equal(got[True][1][n], got[False][1][n]), n


def test_deferred_head_weight_gradient_is_the_same_gradient(dev, monkeypatch):
    """Round 4: the tied head's weight gradient is a leaf of the backward graph and rides, slice by slice, on the decoder layers'
    grouped weight-gradient launches (unigen_hip/qwen2.py: head_bwd / _head_wgrad_slice / flush_deferred_head).  Forced on for the
    tiny model (grouped launches whatever the tile count) and compared with the direct launch: first write after zero_grad
    (beta 0), accumulation over a second backward, a poisoned gradient buffer, two forwards under one backward (a second head
    while the first one's slices are still waiting) -- the embedding gradient includes the lookups' scatter-add, which must land
    AFTER the head's overwrite."""
    from unigen_hip import ops
    g = golden("g2_tiny_unigen.pt")
    mask = additive(g["mask_allow"]).to(dev)
    ids, labels = g["input_ids"].to(dev), g["labels"].to(dev)
    monkeypatch.setattr(ops, "WGRAD_GROUP_MIN_TILES", 0)
    names = ["model.embed_tokens.weight", "model.layers.0.self_attn.q_proj.weight", "model.layers.1.mlp.down_proj.weight", "model.norm.weight"]
    got = {}
    for defer in ("0", "1"):
        monkeypatch.setenv("UNIGEN_DEFER_HEAD_WGRAD", defer)
        model, _ = _tiny_unigen(g, dev)
        model.train()
        eng = model.llm.engine
        params = dict(model.llm.named_parameters())
        seen = []
        orig = eng._head_wgrad_slice
        eng._head_wgrad_slice = lambda: (seen.append(1), orig())[1]          # (count the layer launches that were offered a slice)

        def run(twice=False):
            _, l1, l2, l3 = model(input_ids=ids, attention_mask=mask, labels=labels, **g["kw"])
            loss = l1 + 0.1 * l2 + l3
            if twice:
                _, m1, m2, m3 = model(input_ids=ids, attention_mask=mask, labels=labels, **g["kw"])
                loss = loss + 0.5 * (m1 + m3)
            loss.backward()
        run()
        assert eng._deferred_head is None if defer == "1" else not hasattr(eng, "_deferred_head") or eng._deferred_head is None
        first = {n: params[n].grad.clone() for n in names}
        run()                                                   # accumulate on top
        second = {n: params[n].grad.clone() for n in names}
        model.zero_grad(set_to_none=True)
        eng.fp.grad.fill_(123.0)                                # poison
        run(twice=True)
        third = {n: params[n].grad.clone() for n in names}
        got[defer] = (first, second, third)
        assert len(seen) > 0
    for a, b in zip(got["0"], got["1"]):
        for n in names:
            _check(n, b[n], a[n], 1e-5 if n != "model.embed_tokens.weight" else 1e-4)


def test_fused_adamw_overlapped_update_is_equivalent(dev):
    """FusedAdamW(overlap=True) issues the flat-buffer update on a side stream; the engine orders every later weight /
    gradient access behind it, so three steps (with unrelated work and an optimizer state_dict read in between) leave exactly
    the weights, moments and losses of the in-stream update."""
    from unigen_hip.optim import FusedAdamW
    g = golden("g2_tiny_unigen.pt")
    mask = additive(g["mask_allow"]).to(dev)
    ids, labels = g["input_ids"].to(dev), g["labels"].to(dev)
    results = []
    for overlap in (False, True):
        model, _ = _tiny_unigen(g, dev)
        model.train()
        opt = FusedAdamW(model.parameters(), lr=1e-3, overlap=overlap)
        losses = []
        busy = torch.randn(1024, 1024, device=dev)
        for step in range(3):
            _, l1, l2, l3 = model(input_ids=ids, attention_mask=mask, labels=labels, **g["kw"])
            (l1 + 0.1 * l2 + l3).backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            busy = busy @ busy.t() * 1e-3                       # the "next batch's tokenizer": independent of the backbone
            losses.append(l1.item())
        sd = opt.state_dict()
        m0 = sd["state"][0]["exp_avg"].clone()
        results.append((losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, m0))
        assert opt._pending is None or overlap
    (la, wa, ma), (lb, wb, mb) = results
    # (bit equality is not expected even between two in-stream runs: the embedding scatter-add and the norm-weight gradients
    #  use fp32 atomics, whose order varies)
    assert max(abs(a - b) for a, b in zip(la, lb)) < 1e-5 and torch.allclose(ma, mb, rtol=1e-4, atol=1e-7)
    bad = {k: (wa[k].float() - wb[k].float()).abs().max().item() for k in wa if not torch.allclose(wa[k], wb[k], rtol=1e-5, atol=1e-6)}
    assert not bad, bad


def test_fused_adamw_state_dict_round_trip_and_torch_layout(dev):
    """the optimizer state must survive the reference's checkpointing (`accelerator.save_state` /
    utils/checkpoint.py:67-69 save optimizer.state_dict()).  FusedAdamW exports torch.optim.AdamW's layout: a fresh
    FusedAdamW that loads it continues exactly; a torch.optim.AdamW that loads it takes the same next step; a param group
    added after the first step keeps the moments of the existing groups."""
    import copy
    from unigen_hip.optim import FusedAdamW
    g = golden("g2_tiny_unigen.pt")
    mask = additive(g["mask_allow"]).to(dev)
    ids, labels = g["input_ids"].to(dev), g["labels"].to(dev)

    def grads(model):
        model.zero_grad(set_to_none=True)
        _, l1, l2, l3 = model(input_ids=ids, attention_mask=mask, labels=labels, **g["kw"])
        (l1 + 0.1 * l2 + l3).backward()

    model, _ = _tiny_unigen(g, dev)
    model.train()
    named = [(n, p) for n, p in model.named_parameters()]
    groups = lambda: [{"params": [p for n, p in named if "bias" not in n], "weight_decay": 0.01},
                      {"params": [p for n, p in named if "bias" in n], "weight_decay": 0.0}]
    opt = FusedAdamW(groups(), lr=1e-3)
    for _ in range(2):
        grads(model)
        opt.step()
    sd = copy.deepcopy(opt.state_dict())
    assert set(sd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 2.0
    w2 = {n: p.detach().clone() for n, p in named}
    grads(model)                                               # the gradients every continuation below uses
    gsave = {n: p.grad.detach().clone() for n, p in named}
    opt.step()
    want = {n: p.detach().clone() for n, p in named}

    def restore():
        with torch.no_grad():
            for n, p in named:
                p.copy_(w2[n])
                p.grad.copy_(gsave[n])           # in place: the gradients stay the views of the flat buffer
    # (a) a fresh FusedAdamW resumes from the state dict
    restore()
    opt2 = FusedAdamW(groups(), lr=1e-3)
    opt2.load_state_dict(copy.deepcopy(sd))
    opt2.step()
    for n, p in named:
        assert torch.allclose(p, want[n], rtol=1e-6, atol=1e-8), n
    # (b) torch.optim.AdamW loads the same state dict and takes the same step
    restore()
    opt3 = torch.optim.AdamW(groups(), lr=1e-3)
    opt3.load_state_dict(copy.deepcopy(sd))
    opt3.step()
    for n, p in named:
        assert torch.allclose(p, want[n], rtol=2e-5, atol=1e-7), n
    # (c) add_param_group after steps: existing moments are carried into the rebuilt runs
    extra = torch.nn.Parameter(torch.ones(64, device=dev))
    extra.grad = torch.full((64,), 0.5, device=dev)
    probe = next(p for n, p in named if n.endswith("layers.0.self_attn.q_proj.weight"))   # (named[0] may be the gradient-less DDP anchor)
    m_before = opt.state[probe]["exp_avg"].clone()
    opt.add_param_group({"params": [extra], "weight_decay": 0.0})
    opt.step()
    assert not torch.equal(opt.state[probe]["exp_avg"], torch.zeros_like(m_before)) and float(extra.detach()[0]) < 1.0
    assert torch.allclose(opt.state[probe]["exp_avg"], m_before * 0.9 + 0.1 * probe.grad, rtol=1e-4, atol=1e-7)
